"""`KCenter` and `Herding` model-based coreset initialisers (graphslim/sparsification/{model_based_coreset_base,kcenter,
herding}.py): a GCN is trained on the ORIGINAL graph (`fit_with_val(reduced=False)`, args.eval_epochs epochs — on the HIP ops
of graphslim_amd/models/gcn.py), its log-probability outputs are the embeddings, and real train nodes are picked per class by
the greedy k-center / herding rules.  The selection itself is host-side index arithmetic on [n_c, C] matrices, like the
reference's (`torch.cdist` / argsort / argmax / argmin with first-occurrence tie-breaking)."""
import numpy as np
import torch

from ..dataset.utils import save_reduced
from ..evaluation.utils import verbose_time_memory
from .random import Random


def kcenter_select(embeds, idx_train, labels_train, num_class_dict):
    """kcenter.py:8-24: per class, start from the node closest to the class mean, then repeatedly add the node farthest from
    its nearest chosen centre."""
    embeds, lt = torch.as_tensor(embeds).float().cpu(), np.asarray(labels_train)
    picked = []
    for class_id, cnt in num_class_dict.items():
        idx = np.asarray(idx_train)[lt == class_id]
        feature = embeds[idx]
        mean = feature.mean(0, keepdim=True)
        centers = torch.argsort(torch.cdist(feature, mean)[:, 0])[:1].tolist()
        for _ in range(cnt - 1):
            dis_min = torch.cdist(feature, feature[centers]).min(dim=-1)[0]
            centers.append(int(torch.argmax(dis_min)))
        picked.append(idx[centers])
    return np.hstack(picked)


def herding_select(embeds, idx_train, labels_train, num_class_dict):
    """herding.py:8-26: per class, greedily pick the node whose embedding is closest to (i + 1) * mean - sum(selected so far),
    without replacement."""
    embeds, lt = torch.as_tensor(embeds).float().cpu(), np.asarray(labels_train)
    picked = []
    for class_id, cnt in num_class_dict.items():
        idx = np.asarray(idx_train)[lt == class_id]
        features = embeds[idx]
        mean = features.mean(0, keepdim=True)
        selected, left = [], list(range(features.shape[0]))
        for i in range(cnt):
            det = mean * (i + 1) - features[selected].sum(0)
            j = int(torch.argmin(torch.cdist(det, features[left])))
            selected.append(left[j])
            del left[j]
        picked.append(idx[selected])
    return np.hstack(picked)


class _ModelBased(Random):
    """model_based_coreset_base.py:10-64 (the non-sfgc branch)."""
    rule = None

    def select(self, embeds):
        return type(self).rule(embeds, self.idx_train, self.labels_train, self.num_class_dict)

    @verbose_time_memory
    def reduce(self, data, verbose=False, save=True):
        from ..models.gcn import GCN
        args = self.args
        model = GCN(data.feat_full.shape[1], args.hidden, data.nclass, args)          # coreset_base.py:24: condense_model = 'GCN'
        model.fit_with_val(data, train_iters=args.eval_epochs, normadj=True, verbose=verbose, setting=args.setting, reduced=False)
        embeds = model.predict(data.feat_full, data.adj_full).detach().cpu()
        idx = self.select(embeds)
        if self.setting == "trans":
            adj, feat, lab = data.adj_full[np.ix_(idx, idx)], data.feat_full[idx], data.labels_full[idx]
        else:
            # upstream indexes the embeddings of the FULL graph with train-local ids in the inductive branch
            # (model_based_coreset_base.py:45-51); kept as is: idx are local ids into feat_train / adj_train
            adj, feat, lab = data.adj_train[np.ix_(idx, idx)], data.feat_train[idx], data.labels_train[idx]
        if verbose:
            print("selected nodes:", idx.shape[0])
            print("induced edges:", adj.sum())
        data.adj_syn = torch.from_numpy(np.asarray(adj.todense(), dtype=np.float32))
        data.feat_syn = torch.as_tensor(feat).float().cpu()
        data.labels_syn = torch.as_tensor(lab).long().cpu()
        if save:
            save_reduced(data.adj_syn, data.feat_syn, data.labels_syn, args)
        return data


class KCenter(_ModelBased):
    rule = staticmethod(kcenter_select)


class Herding(_ModelBased):
    rule = staticmethod(herding_select)
