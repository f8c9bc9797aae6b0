"""Raw-file readers for the datasets of the GCond configurations, so that a `load_path` populated by upstream GraphSlim
(i.e. by PyTorch-Geometric's downloaders) works here without PyTorch-Geometric.  Host-side numpy / scipy only.

Two on-disk formats cover cora / citeseer / pubmed / flickr / reddit / ogbn-arxiv (graphslim/dataset/loader.py:39-97):

* **Planetoid** (`<load_path>/<Name>/raw/ind.<name>.{x,tx,allx,y,ty,ally,graph,test.index}`), read by the reference through
  `torch_geometric.datasets.Planetoid` -> `torch_geometric.io.read_planetoid_data` (torch_geometric 2.5-2.7, a third-party
  dependency that is not under /root/reference; restated here from its published algorithm): pickled scipy / numpy
  objects; train = the first len(y) rows, validation = the next 500, test = `test.index`; citeseer's isolated test nodes
  become zero rows; the adjacency lists are symmetrised, self loops removed, duplicates coalesced.
* **GraphSAINT** (`<root>/raw/{adj_full.npz, feats.npy, class_map.json, role.json}`), read by the reference through
  `torch_geometric.datasets.{Flickr, Reddit2}` and, for ogbn-arxiv, by its own `DataGraphSAINT`
  (graphslim/dataset/loader.py:380-445, which also symmetrises the arxiv graph :426-429).

Every reader returns the record `get_dataset` expects: x [N, d] float32, y [N] int64, edge_index [2, E] int64 and boolean
train / val / test masks (torch tensors).
"""
from __future__ import annotations

import json
import os
import pickle
from types import SimpleNamespace

import numpy as np
import scipy.sparse as sp
import torch


def _mask(idx, n):
    m = np.zeros(n, dtype=bool)
    m[np.asarray(idx, dtype=np.int64)] = True
    return torch.from_numpy(m)


def _record(name, x, y, row, col, train, val, test):
    n = x.shape[0]
    ei = np.vstack([row, col]).astype(np.int64)
    return SimpleNamespace(x=torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)),
                           y=torch.from_numpy(np.asarray(y, dtype=np.int64)), edge_index=torch.from_numpy(ei), num_nodes=n,
                           train_mask=_mask(train, n), val_mask=_mask(val, n), test_mask=_mask(test, n), name=name,
                           synthetic=False)


# ---------------------------------------------------------------------------------------------------------------
# Planetoid
# ---------------------------------------------------------------------------------------------------------------
def _unpickle(path):
    with open(path, "rb") as f:
        return pickle.load(f, encoding="latin1")


def _dense(a):
    return np.asarray(a.todense() if sp.issparse(a) else a)


def planetoid_dir(load_path, name):
    """PyG keeps the raw files under <root>/<Name>/raw with the capitalisation it was given; accept the usual spellings."""
    for sub in (name, name.capitalize(), name.lower(), {"cora": "Cora", "citeseer": "CiteSeer", "pubmed": "PubMed"}.get(name, name)):
        d = os.path.join(load_path, sub, "raw")
        if os.path.exists(os.path.join(d, f"ind.{name.lower()}.x")):
            return d
    return None


def read_planetoid(raw_dir, name):
    name = name.lower()
    obj = {k: _unpickle(os.path.join(raw_dir, f"ind.{name}.{k}")) for k in ("x", "tx", "allx", "y", "ty", "ally", "graph")}
    with open(os.path.join(raw_dir, f"ind.{name}.test.index")) as f:
        test_index = np.array([int(line) for line in f.read().split()], dtype=np.int64)
    x, tx, allx = (_dense(obj[k]).astype(np.float32) for k in ("x", "tx", "allx"))
    y, ty, ally = (np.asarray(obj[k]) for k in ("y", "ty", "ally"))
    train_index = np.arange(y.shape[0])
    val_index = np.arange(y.shape[0], y.shape[0] + 500)
    sorted_test = np.sort(test_index)
    if name == "citeseer":
        # some test nodes are isolated and absent from tx / ty: give them zero rows at their place
        span = int(test_index.max() - test_index.min()) + 1
        tx_ext = np.zeros((span, tx.shape[1]), dtype=tx.dtype)
        ty_ext = np.zeros((span, ty.shape[1]), dtype=ty.dtype)
        tx_ext[sorted_test - test_index.min()] = tx
        ty_ext[sorted_test - test_index.min()] = ty
        tx, ty = tx_ext, ty_ext
    feats = np.vstack([allx, tx])
    feats[test_index] = feats[sorted_test]
    onehot = np.vstack([ally, ty])
    labels = onehot.argmax(1)
    labels[test_index] = labels[sorted_test]
    n = feats.shape[0]
    rows, cols = [], []
    for k, nbrs in obj["graph"].items():
        rows += [int(k)] * len(nbrs)
        cols += [int(v) for v in nbrs]
    a = sp.coo_matrix((np.ones(len(rows), dtype=np.int8), (rows, cols)), shape=(n, n)).tocsr()
    a = ((a + a.T) > 0).astype(np.int8).tolil()
    a.setdiag(0)
    coo = a.tocoo()
    order = np.lexsort((coo.col, coo.row))
    return _record(name, feats, labels, coo.row[order], coo.col[order], train_index, val_index, test_index)


# ---------------------------------------------------------------------------------------------------------------
# GraphSAINT
# ---------------------------------------------------------------------------------------------------------------
GRAPHSAINT_FILES = ("adj_full.npz", "feats.npy", "class_map.json", "role.json")


def graphsaint_dir(load_path, name):
    """<load_path>/flickr/raw, <load_path>/reddit/raw (loader.py:52-55), <load_path>/ogbn_arxiv/raw (loader.py:385-386)."""
    for sub in (name, name.replace("-", "_"), name.capitalize()):
        d = os.path.join(load_path, sub, "raw")
        if all(os.path.exists(os.path.join(d, f)) for f in GRAPHSAINT_FILES):
            return d
    return None


def read_graphsaint(raw_dir, name):
    adj = sp.load_npz(os.path.join(raw_dir, "adj_full.npz")).tocsr()
    if name.replace("_", "-") == "ogbn-arxiv":          # loader.py:426-429
        adj = adj + adj.T
        adj.data[:] = 1
    coo = adj.tocoo()
    x = np.load(os.path.join(raw_dir, "feats.npy")).astype(np.float32)
    with open(os.path.join(raw_dir, "class_map.json")) as f:
        cm = json.load(f)
    n = x.shape[0]
    first = next(iter(cm.values()))
    if isinstance(first, list):
        raise NotImplementedError("graphslim_amd: multi-label GraphSAINT datasets (yelp, amazon) are outside the built path")
    y = np.full(n, -1, dtype=np.int64)
    ids = np.fromiter((int(k) for k in cm.keys()), dtype=np.int64, count=len(cm))
    y[ids] = np.fromiter((int(v) for v in cm.values()), dtype=np.int64, count=len(cm))
    if (y < 0).any():
        raise ValueError(f"{raw_dir}/class_map.json does not label every node")
    with open(os.path.join(raw_dir, "role.json")) as f:
        role = json.load(f)
    return _record(name, x, y, coo.row, coo.col, role["tr"], role["va"], role["te"])


def find_and_read(load_path, name):
    """The raw files of `name` under `load_path` in either format, or None."""
    if not load_path:
        return None
    if name in ("cora", "citeseer", "pubmed"):
        d = planetoid_dir(load_path, name)
        return read_planetoid(d, name) if d else None
    d = graphsaint_dir(load_path, name)
    return read_graphsaint(d, name) if d else None
