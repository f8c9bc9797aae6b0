"""GraphData contract of the reference (`TransAndInd`, graphslim/dataset/loader.py:100-229) over dataset-shaped
synthetic graphs (or real arrays handed in by the caller).  Host-side preparation only: numpy / scipy / torch-CPU,
exactly as the reference does it on the host before anything touches the device.
"""
from __future__ import annotations

import os

import numpy as np
import scipy.sparse as sp
import torch

from .synthetic import SHAPES, make_raw


def ei2csr(edge_index, n):
    """graphslim/dataset/convertor.py:71-75: int64 ones, duplicates summed."""
    ei = edge_index.numpy() if isinstance(edge_index, torch.Tensor) else np.asarray(edge_index)
    return sp.csr_matrix((np.ones(ei.shape[1], dtype=np.int64), (ei[0], ei[1])), shape=(n, n))


def get_dataset(name="cora", args=None, load_path=None):
    """graphslim/dataset/loader.py:39-97.  There is no network on the target boxes, so every dataset name resolves to
    the dataset-shaped synthetic generator (graphslim_amd/dataset/synthetic.py) unless `<load_path>/<name>.npz` holds
    real arrays (x, y, edge_index, train_mask, val_mask, test_mask)."""
    name = {"arxiv": "ogbn-arxiv"}.get(name, name)
    load_path = load_path or getattr(args, "load_path", None)
    raw = None
    if load_path and os.path.exists(os.path.join(load_path, f"{name}.npz")):
        z = np.load(os.path.join(load_path, f"{name}.npz"))
        from types import SimpleNamespace
        raw = SimpleNamespace(x=torch.from_numpy(z["x"]).float(), y=torch.from_numpy(z["y"]).long(),
                              edge_index=torch.from_numpy(z["edge_index"]).long(), num_nodes=int(z["x"].shape[0]),
                              train_mask=torch.from_numpy(z["train_mask"]), val_mask=torch.from_numpy(z["val_mask"]),
                              test_mask=torch.from_numpy(z["test_mask"]), name=name, synthetic=False)
    if raw is None:
        if name not in SHAPES:
            raise ValueError(f"dataset '{name}' is not available offline; known shapes: {sorted(SHAPES)}")
        raw = make_raw(name, seed=getattr(args, "seed", 1), scale=float(os.environ.get("GRAPHSLIM_AMD_SCALE", "1")))
    data = splits(raw)
    out = TransAndInd(data, name)
    out.nclass = int(raw.y.max()) + 1
    if args is not None:
        args.nclass = out.nclass
    return out


def splits(data):
    """graphslim/dataset/utils.py:97-134 ('fixed' split): masks -> index tensors."""
    data.idx_train = data.train_mask.nonzero().view(-1)
    data.idx_val = data.val_mask.nonzero().view(-1)
    data.idx_test = data.test_mask.nonzero().view(-1)
    return data


class TransAndInd:
    """Attribute-for-attribute mirror of graphslim/dataset/loader.py:100-175."""

    def __init__(self, data, dataset, norm=True):
        self.class_dict = None
        self.samplers = None
        self.class_dict2 = None
        self.num_nodes = data.num_nodes
        self.train_mask, self.val_mask, self.test_mask = data.train_mask, data.val_mask, data.test_mask
        self.x, self.y = data.x, data.y
        self.feat_full, self.labels_full = data.x, data.y
        self.edge_index = data.edge_index
        self.adj_full = ei2csr(data.edge_index, data.x.shape[0])
        if dataset in ("flickr", "reddit", "ogbn-arxiv"):
            # loader.py:111-119: symmetrise, then StandardScaler fitted on the TRAIN rows (float64 like sklearn)
            a = self.adj_full
            self.adj_full = ((a + a.T) > 0).astype(np.int64).tocsr()
            coo = self.adj_full.tocoo()
            self.edge_index = torch.from_numpy(np.vstack([coo.row, coo.col]).astype(np.int64))
            xt = self.x[data.idx_train].double()
            mean, std = xt.mean(0), xt.std(0, unbiased=False)
            std[std == 0] = 1.0
            self.feat_full = ((self.x.double() - mean) / std).float()
        if norm and dataset in ("cora", "citeseer", "pubmed"):
            self.feat_full = torch.nn.functional.normalize(self.feat_full, p=1, dim=1)      # loader.py:120-121
        self.idx_train, self.idx_val, self.idx_test = data.idx_train, data.idx_val, data.idx_test
        it, iv, ie = (i.numpy() for i in (self.idx_train, self.idx_val, self.idx_test))
        self.adj_train = self.adj_full[np.ix_(it, it)]
        self.adj_val = self.adj_full[np.ix_(iv, iv)]
        self.adj_test = self.adj_full[np.ix_(ie, ie)]
        self.labels_train, self.labels_val, self.labels_test = (self.labels_full[i] for i in (self.idx_train, self.idx_val, self.idx_test))
        self.feat_train, self.feat_val, self.feat_test = (self.feat_full[i] for i in (self.idx_train, self.idx_val, self.idx_test))
        self.labels_syn = self.feat_syn = self.adj_syn = None
        self.nclass = int(self.labels_full.max()) + 1

    def to(self, device):
        """loader.py:141-158."""
        for k in ("feat_full", "labels_full", "x", "y", "edge_index", "feat_train", "feat_val", "feat_test"):
            setattr(self, k, getattr(self, k).to(device))
        return self

    def class_pools(self, setting):
        """loader.py:188-195: per-class seed pools — global train ids (trans) or local ids of the train sub-graph (ind)."""
        lt = self.labels_train.cpu().numpy()
        if setting == "trans":
            ids = self.idx_train.cpu().numpy()
        else:
            ids = np.arange(len(lt))
        return [ids[lt == c] for c in range(self.nclass)]

    def retrieve_class(self, c, num=256):
        """loader.py:177-185."""
        lt = self.labels_train.cpu().numpy()
        idx = np.arange(len(lt))[lt == c]
        return np.random.permutation(idx)[:num]

    def reset(self):
        self.samplers = None
        self.class_dict2 = None
        self.labels_syn, self.feat_syn, self.adj_syn = None, None, None
