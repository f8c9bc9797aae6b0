"""`Random` coreset initialiser (graphslim/sparsification/{random,coreset_base,model_free_coreset_base}.py): picks real
train nodes per class and returns their features as the initial X~.  Host-side numpy, like the reference."""
import numpy as np
import torch

from ..dataset.utils import save_reduced
from ..evaluation.utils import verbose_time_memory
from ..host import generate_labels_syn


class Random:
    def __init__(self, setting, data, args, **kwargs):
        self.data, self.args, self.setting = data, args, setting
        self.device = args.device
        if getattr(data, "labels_syn", None) is not None:             # coreset_base.py:13-20
            self.num_class_dict = data.num_class_dict
        else:
            # coreset_base.py:27-47: every class gets max(int(count*rate), 1), no remainder rule
            lt = np.asarray(data.labels_train)
            _, ncd, _ = generate_labels_syn(lt, args.reduction_rate)
            from collections import Counter
            cnt = Counter(lt.tolist())
            self.num_class_dict = {c: max(int(cnt[c] * args.reduction_rate), 1) for c in ncd}
        self.labels_train = data.labels_train
        self.idx_train = np.arange(len(data.idx_train)) if args.setting == "ind" else np.asarray(data.idx_train)

    def select(self, embeds=None):
        """random.py:9-17."""
        lt = np.asarray(self.labels_train)
        out = []
        for class_id, cnt in self.num_class_dict.items():
            out.append(np.random.permutation(self.idx_train[lt == class_id])[:cnt])
        return np.hstack(out)

    @verbose_time_memory
    def reduce(self, data, verbose=False, save=True):
        """model_free_coreset_base.py:16-61.  With --agg the initial features are the two-hop aggregated ones, A^ (A^ X)
        (upstream forms A^ A^ first, :20-24; same product), computed by two gs_spmm_csr_f32 launches, and the initial
        structure is the identity."""
        args = self.args
        if getattr(args, "agg", False):
            from ..engine import spmm_csr
            from ..host import normalize_adj_sparse
            if not str(args.device).startswith("cuda"):
                raise RuntimeError("graphslim_amd has no CPU path: run with a GPU (-G 0)")
            # upstream multiplies the TRAIN adjacency with feat_full in the inductive branch (:37-41), a shape error; the
            # train features are what that branch means
            adj_s, feat_s, lab_s = (data.adj_full, data.feat_full, data.labels_full) if self.setting == "trans" else \
                (data.adj_train, data.feat_train, data.labels_train)
            rp, col, val = normalize_adj_sparse(adj_s)
            dev = torch.device(args.device)
            i32 = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.int32).to(dev)
            csr = (i32(rp), i32(col), torch.as_tensor(val, dtype=torch.float32).to(dev))
            d = feat_s.shape[1]
            X = torch.zeros(feat_s.shape[0], (d + 3) // 4 * 4, device=dev)
            X[:, :d] = torch.as_tensor(feat_s).to(dev, torch.float32)
            feat_agg = spmm_csr(*csr, spmm_csr(*csr, X))[:, :d].cpu()
            idx = self.select(feat_agg)
            data.feat_syn = feat_agg[idx].float()
            data.adj_syn = torch.eye(len(idx))
            data.labels_syn = torch.as_tensor(lab_s)[idx].long().cpu()
            if verbose:
                print("selected nodes:", idx.shape[0])
                print("induced edges:", data.adj_syn.sum())
            if save:
                save_reduced(data.adj_syn, data.feat_syn, data.labels_syn, args)
            return data
        idx = self.select()
        if self.setting == "trans":
            adj, feat, lab = data.adj_full[np.ix_(idx, idx)], data.feat_full[idx], data.labels_full[idx]
        else:
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
