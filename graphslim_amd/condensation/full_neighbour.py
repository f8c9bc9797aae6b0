"""Full-neighbour mode of the GCond real-side gradients (SURVEY section 8e "mode F"; dataset/loader.py:204 `sizes = [-1, -1]`,
commented out upstream; BASELINE.json configs[4] "row-sharded SpMM") for SGC with ONE transform.

With ntrans = 1 the condensation model is linear up to the softmax (models/sgc.py:37-50: x W + b, then `nlayers` = 2
propagations), so the class gradient over the 2-hop FULL neighbourhood of a seed batch needs exactly the seed rows of

    S_full = A^ (A^ X)        rs_full = A^ (A^ 1)        (A^ = the normalised adjacency the sampler walks, utils.py:451-458)

and X never changes during a job: the two full-graph SpMMs run ONCE, before the first outer iteration.  They are the
path's row-sharded full-graph product: rank r owns a contiguous row block of A^ (balanced by non-zeros), multiplies it with the
full operand (gs_spmm_csr_f32, width d + 1: the ones column carries the bias factor) and the blocks are all-gathered after each
hop (parallel.RowShard.all_gather_rows; rows are computed by the same kernel in the same order as on one GPU, so the sharded
result is bitwise the single-GPU one).  Per outer iteration the real side then only draws the class seeds and gathers their
rows (gs_gcond_set_full_neighbour) — no neighbour sampling, no sampling noise.

With a ReLU between two transforms (ntrans = 2) the same idea needs one full-graph SpMM per class and iteration in the
backward: declined, DESIGN.md section 6."""
import numpy as np
import torch

from ..engine import spmm_csr
from ..parallel import RowShard


def full_neighbour_operators(csr, X, device, shard=None, hops=2):
    """csr = (rowptr, col, val) of A^ (host arrays, `normalize_adj_sparse`); X [N, d] features.
    Returns (S_full [N, ld] fp32 device tensor with ld = d + 1 padded to a multiple of 4: columns [0, d) = A^^hops X,
    column d = A^^hops 1; rs_full = a contiguous copy of that column) and the seconds the SpMMs took on this rank."""
    import time
    shard = shard or RowShard()
    rowptr, col, val = (np.asarray(a) for a in csr)
    n = len(rowptr) - 1
    d = X.shape[1]
    ld = (d + 1 + 3) // 4 * 4
    cur = torch.zeros(n, ld, dtype=torch.float32, device=device)
    cur[:, :d] = torch.as_tensor(X).to(device, torch.float32)
    cur[:, d] = 1.0
    blocks = RowShard.split_rows(rowptr, shard.world)
    lo, hi = blocks[shard.rank]
    i32 = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.int32).to(device)
    rp = i32(rowptr[lo:hi + 1] - rowptr[lo])
    cl = i32(col[rowptr[lo]:rowptr[hi]])
    vl = torch.as_tensor(np.ascontiguousarray(val[rowptr[lo]:rowptr[hi]]), dtype=torch.float32).to(device)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(hops):
        part = spmm_csr(rp, cl, vl, cur, F=ld)
        cur = shard.all_gather_rows(part, n, blocks)
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    return cur, cur[:, d].contiguous(), dt
