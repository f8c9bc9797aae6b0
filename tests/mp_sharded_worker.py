#!/usr/bin/env python3
"""Worker of tests/test_gpu_multiprocess.py: one rank of a REAL multi-process sharded GCond run under torch.distributed.
Launched by torch.distributed.run with WORLD_SIZE ranks.  Backend: nccl (= RCCL over xGMI) when every rank has its own GPU,
gloo with all ranks sharing cuda:0 when the box has fewer GPUs than ranks (GS_DIST_BACKEND=gloo GS_SHARE_DEVICE=1).
Every rank drives its ShardedGCond through outer iterations (class-parallel + PGE pair-row sharding, all-reduce(sum)
exchanges); rank 0 also runs the UNSHARDED engine on the same inputs and checks loss / X~ / theta / adj_syn; all ranks check
that the replicas hold bitwise identical state.  Exit code 0 = pass."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def ipc_transport_checks(rank, world, local):
    """The one-shot peer-read all-reduce by itself (gs_comm_create_ipc): sizes from one element to the largest join of the headline
    iteration, fp32 and fp64, odd counts (scalar tails), many exchanges back to back on one stream (slot parity, flag sequence), sums
    checked against the closed form, every rank's result bitwise the same; an exchange beyond the slots goes to the fallback."""
    from graphslim_amd.parallel import Comm
    dev = f"cuda:{local}"
    ok = True
    base = Comm.callback(lambda t: dist.all_reduce(t, op=dist.ReduceOp.SUM), rank, world)
    comm = Comm.ipc(dist, None, slot_bytes=1_100_000, fallback=base)
    g = torch.Generator(device="cpu").manual_seed(7)
    for rep in range(3):
        for n, dt in ((1, torch.float32), (5, torch.float32), (1000, torch.float32), (264_000 + rep, torch.float32), (512, torch.float64),
                      (3, torch.float64), (137_001, torch.float64)):
            parts = [torch.randn(n, generator=g, dtype=dt) for _ in range(world)]      # same draws on every rank
            want = parts[0].clone()
            for q in parts[1:]:
                want += q                                                               # rank order, like the kernel
            t = parts[rank].to(dev)
            comm.allreduce(t, tag=2)
            torch.cuda.synchronize()
            if not torch.equal(t.cpu(), want):
                print(f"rank {rank}: ipc all-reduce of {n} x {dt} differs: max {float((t.cpu() - want).abs().max())}")
                ok = False
    big = torch.full((400_000,), float(rank + 1), device=dev)      # 1.6 MB > the slots: the fallback carries it
    comm.allreduce(big, tag=5)
    torch.cuda.synchronize()
    if not torch.equal(big.cpu(), torch.full((400_000,), world * (world + 1) / 2.0)):
        print(f"rank {rank}: the fallback exchange is wrong")
        ok = False
    if comm.ipc_error() != 0:
        print(f"rank {rank}: ipc exchange timed out waiting for rank {comm.ipc_error() - 1}")
        ok = False
    dist.barrier()
    del comm
    return ok


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", rank))
    backend = os.environ.get("GS_DIST_BACKEND", "nccl")
    if os.environ.get("GS_SHARE_DEVICE") == "1":
        local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local}"))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    import test_gpu_engine as TE
    from graphslim_amd.parallel import Collective, ShardedGCond
    TE.DEV = f"cuda:{local}"
    t = int(os.environ.get("GS_MP_T", "2"))
    p = TE._problem(t, seed=21, nclass=5)

    def build():
        eng = TE.make_engine(p["n"], p["nclass"], p["d"], p["hidden"], t, p["ph"], p["labels_syn"], p["ranges"], p["x"], p["f1"],
                             p["f2"], bcap=p["bcap"], inner=2, lrs=(0.01, 0.005, 0.02), csr=p["csr"], pools=p["pools"],
                             external=False, seed=5)
        eng.set_feat(p["xs"])
        eng.set_pge(p["pge"]["lin"] + p["pge"]["bn"])
        return eng

    ok = True
    if os.environ.get("GS_COMM_TRANSPORT") == "ipc":
        ok = ipc_transport_checks(rank, world, local) and ok
    eng = build()
    sh = ShardedGCond(eng, rank, world, [len(q) for q in p["pools"]], Collective(dist))
    if os.environ.get("GS_COMM_TRANSPORT") == "ipc" and sh.comm.kind != "ipc":
        print(f"rank {rank}: GS_COMM_TRANSPORT=ipc but the engine's transport is {sh.comm.kind}")
        ok = False
    ref = build() if rank == 0 else None
    for epoch in (0, 10):
        eng.set_model(p["theta"])
        if ref is not None:
            ref.set_model(p["theta"])
        for _ in range(2):
            sh.outer_step(epoch)
            if ref is not None:
                ref.outer_step(epoch)
                l0, l1 = float(ref.buffer("loss")[0]), float(eng.buffer("loss")[0])
                if not abs(l1 - l0) <= 2e-5 * abs(l0):
                    print(f"rank 0: loss mismatch {l1} vs {l0}")
                    ok = False
    torch.cuda.synchronize()
    if sh.comm.kind == "ipc" and sh.comm.ipc_error() != 0:
        print(f"rank {rank}: an ipc exchange of the sharded run timed out waiting for rank {sh.comm.ipc_error() - 1}")
        ok = False
    if ref is not None:
        for name, a, b, tol in (("feat_syn", eng.feat_syn, ref.feat_syn, 2e-4), ("adj_syn", eng.adj_syn, ref.adj_syn, 2e-4),
                                ("model", eng.model_params, ref.model_params, 5e-4)):
            dmax = float((a - b).abs().max())
            if not dmax <= tol:
                print(f"rank 0: {name} differs from the unsharded run by {dmax}")
                ok = False
    # replicas must not drift: compare a checksum vector of the learnable state across ranks (bitwise)
    state = torch.cat([eng.feat_syn.reshape(-1), eng.pge_params.reshape(-1), eng.model_params.reshape(-1)]).double()
    chk = torch.stack([state.sum(), (state * torch.arange(1, state.numel() + 1, device=state.device, dtype=torch.float64)).sum()])
    gathered = [torch.zeros_like(chk) for _ in range(world)]
    dist.all_gather(gathered, chk)
    for r, g in enumerate(gathered):
        if not torch.equal(g, gathered[0]):
            print(f"rank {rank}: replica {r} drifted: {g.tolist()} vs {gathered[0].tolist()}")
            ok = False
    flag = torch.tensor([0.0 if ok else 1.0], device=f"cuda:{local}")
    dist.all_reduce(flag)
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print("mp_sharded_worker:", "PASS" if float(flag) == 0 else "FAIL", f"(world {world}, backend {backend})")
    sys.exit(0 if float(flag) == 0 else 1)


if __name__ == "__main__":
    main()
