"""Shared test helpers (CPU side)."""
import os

import numpy as np
import scipy.sparse as sp
import torch

from oracle import gcond_oracle as O


def load_golden(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"), allow_pickle=False)


def T(a, dtype=None):
    t = torch.from_numpy(np.asarray(a))
    return t.to(dtype) if dtype is not None else t


def rebuild_batches(g, tag):
    """Re-create the deterministic batches the golden run of tests/golden/make_golden.py fed to the reference."""
    n, nclass, d, hidden, t, calls0 = [int(v) for v in g[f"{tag}_meta"]]
    a = sp.coo_matrix((np.ones(len(g[f"{tag}_a_row"])), (g[f"{tag}_a_row"], g[f"{tag}_a_col"])), shape=(n, n)).tocsr()
    csr = O.normalize_adj_sparse(a)
    idx_train, y = g[f"{tag}_idx_train"], g[f"{tag}_y"]
    labels_train = y[idx_train]
    sizes = [int(s) for s in g[f"{tag}_sizes"]]
    batches = []
    for c in range(nclass):
        rng = np.random.default_rng(100000 + calls0 + c)
        batch = rng.permutation(idx_train[labels_train == c])[:256]
        batches.append(O.sample_blocks(*csr, batch, sizes, rng))
    return batches, (n, nclass, d, hidden, t), csr


def sbm_graph(n, nclass, d, avg_deg, seed, homophily=0.8):
    """Small symmetric graph with class-correlated features (numpy only)."""
    rng = np.random.default_rng(seed)
    y = rng.integers(0, nclass, n)
    m = n * avg_deg // 2
    src = rng.integers(0, n, m)
    same = rng.random(m) < homophily
    dst = rng.integers(0, n, m)
    order = np.argsort(y, kind="stable")
    starts = np.searchsorted(y[order], np.arange(nclass))
    ends = np.searchsorted(y[order], np.arange(nclass), side="right")
    pick = starts[y[src]] + (rng.random(m) * (ends[y[src]] - starts[y[src]])).astype(np.int64)
    dst = np.where(same, order[np.minimum(pick, n - 1)], dst)
    keep = src != dst
    a = sp.coo_matrix((np.ones(keep.sum()), (src[keep], dst[keep])), shape=(n, n)).tocsr()
    a = ((a + a.T) > 0).astype(np.int64).tocsr()
    mu = rng.normal(size=(nclass, d))
    x = (mu[y] + rng.normal(size=(n, d))).astype(np.float32)
    return a, x, y.astype(np.int64)
