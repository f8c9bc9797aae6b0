"""`clustering` initialiser (graphslim/coarsening/clustering.py:15-76): per class, scikit-learn's BisectingKMeans (random_state = 0,
the dependency upstream calls, on the host) splits the class's train features into as many clusters as the class has synthetic
nodes; every synthetic node starts at one cluster mean (cluster id order); the initial structure is the identity.
The cluster means are gs_gather_rows_f32 + gs_colsum_f32 (fp64 accumulation) on the GPU."""
import numpy as np
import torch

from .._lib import check, lib, ptr, stream
from ..dataset.utils import save_reduced
from ..evaluation.utils import verbose_time_memory
from ..host import generate_labels_syn


def cluster_ids(x_c, n_clusters):
    """clustering.py:62-65: BisectingKMeans(n_clusters, random_state=0).fit(x_c).predict(x_c) (scikit-learn, host)."""
    from sklearn.cluster import BisectingKMeans
    km = BisectingKMeans(n_clusters=n_clusters, random_state=0)
    km.fit(x_c)
    return km.predict(x_c).astype(np.int64)


class Cluster:
    def __init__(self, setting, data, args, **kwargs):
        self.data, self.args, self.setting = data, args, setting
        # (upstream also renames args.method here, clustering.py:34, which outlives GCondBase.init: see coarsening/averaging.py)

    @verbose_time_memory
    def reduce(self, data, verbose=True, save=True):
        args = self.args
        if not str(args.device).startswith("cuda"):
            raise RuntimeError("graphslim_amd has no CPU path: run with a GPU (-G 0)")
        dev = torch.device(args.device)
        y_train = np.asarray(data.labels_train)
        if getattr(data, "labels_syn", None) is not None:                 # clustering.py:53-57
            ls = data.labels_syn
            y_syn = torch.as_tensor(np.asarray(ls.cpu() if isinstance(ls, torch.Tensor) else ls))
        else:                                                             # prepare_select, clustering.py:78-118
            y_syn = torch.from_numpy(generate_labels_syn(y_train, args.reduction_rate)[0])
        x_host = torch.as_tensor(data.feat_train).float().cpu()
        x_train = x_host.to(dev).contiguous()
        d = x_train.shape[1]
        x_syn = torch.zeros(y_syn.shape[0], d)
        mean = torch.empty(d, dtype=torch.float32, device=dev)
        for c in range(data.nclass):
            ids = np.nonzero(y_train == c)[0]
            n_c = int((y_syn == c).sum())
            if n_c == 0:
                continue
            clusters = cluster_ids(x_host[ids].numpy(), n_c)
            means = torch.zeros(n_c, d)
            for k in range(n_c):                                          # scatter_mean(x_c, clusters, dim=0), clustering.py:66
                members = ids[clusters == k]
                if len(members) == 0:
                    continue
                idx = torch.as_tensor(members, dtype=torch.int32).to(dev)
                rows = torch.empty(len(members), d, dtype=torch.float32, device=dev)
                check(lib().gs_gather_rows_f32(ptr(x_train), d, ptr(idx), len(members), d, ptr(rows), d, stream()))
                check(lib().gs_colsum_f32(ptr(rows), d, len(members), d, ptr(mean), stream()))
                means[k] = (mean.cpu().double() / len(members)).float()
            x_syn[y_syn == c] = means
        data.feat_syn, data.labels_syn = x_syn, y_syn
        data.adj_syn = torch.eye(x_syn.shape[0])
        if save:
            save_reduced(data.adj_syn, data.feat_syn, data.labels_syn, args)
        return data
