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


class VirtualWorld:
    """`world` ranks as threads of ONE process sharing one GPU, each driving its own engine through the library's sharded entry
    point (gs_gcond_outer_step_sharded): the callback transport of gs_comm sums the ranks' buffers in place (ctypes releases the GIL
    inside the library call; the callback re-enters Python on the calling thread).  All threads use the legacy default stream."""

    def __init__(self, world):
        import threading
        self.world = world
        self.barrier = threading.Barrier(world, timeout=120)
        self.slots = [None] * world
        self.calls = 0

    def comm(self, rank):
        from graphslim_amd.parallel import Comm

        def fn(t):
            self.slots[rank] = t
            self.barrier.wait()
            if rank == 0:
                torch.cuda.synchronize()
                total = torch.stack(self.slots).sum(0)
                for s in self.slots:
                    s.copy_(total)
                torch.cuda.synchronize()
                self.calls += 1
            self.barrier.wait()

        return Comm.callback(fn, rank, self.world)

    def run(self, fns):
        """fns[r]() on thread r; re-raises the first exception (and breaks the barrier so the others do not wait forever)."""
        import threading
        errs = []

        def wrap(r):
            try:
                torch.cuda.set_device(0)
                fns[r]()
            except BaseException as e:      # noqa: BLE001
                errs.append(e)
                self.barrier.abort()

        ts = [threading.Thread(target=wrap, args=(r,)) for r in range(self.world)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        torch.cuda.synchronize()
        if errs:
            raise errs[0]
