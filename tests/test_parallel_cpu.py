"""Multi-rank host logic on CPU: partitioning, and the sharded driver's collective schedule over a real
torch.distributed process group (gloo, world_size 2) with a recording stand-in for the HIP engine."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_partitions_cover_everything_once():
    from graphslim_amd.parallel import split_classes, split_even
    for n in (1, 7, 40, 41, 454):
        for w in (1, 2, 3, 4, 8):
            r = split_even(n, w)
            assert r[0][0] == 0 and r[-1][1] == n and all(a[1] == b[0] for a, b in zip(r, r[1:]))
            assert max(h - l for l, h in r) - min(h - l for l, h in r) <= 1
    rng = np.random.default_rng(0)
    for C in (3, 7, 40, 41):
        costs = np.minimum(256, rng.integers(5, 3000, C))
        for w in (1, 2, 4, 8):
            r = split_classes(costs, w)
            assert r[0][0] == 0 and r[-1][1] == C and all(a[1] == b[0] for a, b in zip(r, r[1:]))
            if C >= w:
                assert all(h > l for l, h in r)
                loads = [costs[l:h].sum() for l, h in r]
                assert max(loads) <= costs.sum() / w + costs.max()


class FakeEngine:
    """Records the stage calls and fills the all-reduce buffers with rank-dependent values."""

    class S:
        bcap, n_syn = 256, 10

    def __init__(self, rank):
        self.rank, self.calls, self.shapes, self.dev = rank, [], self.S(), "cpu"
        self.bufs = {k: torch.zeros(6, dtype=torch.float64 if k in ("ar_bn2", "ar_l3") else torch.float32)
                     for k in ("ar_bn2", "ar_l3", "e3", "ar_syn", "ar_marg", "ar_w1")}
        for name in ("set_feat", "set_pge", "set_model", "set_forward_reuse", "get_model", "get_pge", "read_loss", "buffer", "feat_syn", "adj_syn",
                     "pge_params", "model_params", "get_blocks"):
            setattr(self, name, None)

    def view(self, name, dtype=torch.float32):
        return self.bufs[name]

    def forward_cached(self):
        return False

    def set_shard(self, *a):
        self.shard = a

    def _fill(self, name):
        self.bufs[name] += float(self.rank + 1)

    def stage_sample(self): self.calls.append("sample")
    def stage_real_grads(self): self.calls.append("real")
    def stage_real_async(self): self.calls.extend(["sample", "real"])
    def pge_forward_part(self, inf, part):
        self.calls.append(f"fwd{int(inf)}{part}")
        if part == 1: self._fill("ar_bn2")
        if part == 2: self._fill("e3")
    def stage_syn_match(self):
        self.calls.append("syn"); self._fill("ar_syn")
    def syn_finish(self): self.calls.append("fin")
    def pge_backward_part(self, a, b, part):
        self.calls.append(f"bwd{part}")
        if part == 1: self._fill("ar_l3")
        if part == 2: self._fill("ar_marg"); self._fill("ar_w1")
    def stage_update(self, epoch): self.calls.append("upd")
    def inner_steps(self): self.calls.append("inner")


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from graphslim_amd.parallel import Collective, ShardedGCond
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    eng = FakeEngine(rank)
    sh = ShardedGCond(eng, rank, world, [500, 30, 300, 256, 10], Collective(dist))
    sh.outer_step(0)      # PGE epoch: ar_w1 exchanged
    a = {k: v.clone() for k, v in eng.bufs.items()}
    sh.outer_step(10)     # feature epoch: ar_w1 untouched by the collective
    q.put((rank, eng.shard, eng.calls, {k: v.tolist() for k, v in a.items()}, eng.bufs["ar_w1"].tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_driver_over_gloo_world2():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=60) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, shard0, calls0, bufs0, w0), (r1, shard1, calls1, bufs1, w1) = res
    assert shard0[0] == 0 and shard0[1] == shard1[0] and shard1[1] == 5          # classes split contiguously
    assert shard0[2:] == (0, 5) and shard1[2:] == (5, 10)                        # pair rows split evenly
    expected = ["sample", "real", "fwd01", "fwd02", "fwd03", "syn", "fin", "bwd1", "bwd2", "bwd3", "upd", "fwd11", "fwd12",
                "fwd13", "inner"]
    assert calls0 == expected * 2 and calls1 == calls0
    # after the first (PGE-epoch) iteration every exchanged buffer holds the SUM over ranks on both ranks
    assert bufs0 == bufs1
    assert bufs0["ar_syn"] == [3.0] * 6 and bufs0["ar_l3"] == [3.0] * 6 and bufs0["ar_w1"] == [3.0] * 6
    # the forward runs twice (train + inference); the stand-in accumulates: (1+2) -> 3, then (3+1)+(3+2) -> 9
    assert bufs0["ar_bn2"][0] == 9.0 and bufs0["e3"][0] == 9.0
    # feature epoch: ar_w1 is filled locally but NOT reduced (3 + own contribution only)
    assert w0 == [4.0] * 6 and w1 == [5.0] * 6
