"""A REAL multi-process sharded GCond run (one process per rank, torch.distributed): VERDICT r01 "next" 5a.
* >= 2 GPUs visible: backend nccl (= RCCL over xGMI), one GPU per rank.
* 1 GPU (the round's test box): the same processes over gloo, all ranks sharing cuda:0 — RCCL refuses several ranks on one
  device, so this leg checks the multi-process protocol (shard assignment, exchange order, replica consistency, agreement
  with the unsharded engine), and the RCCL leg runs wherever the driver has more than one GPU.
The worker (tests/mp_sharded_worker.py) exits non-zero on any mismatch.  Needs an MI355X."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _launch(world, env_extra, t):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", GS_MP_T=str(t), **env_extra)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "mp_sharded_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "PASS" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


@pytest.mark.parametrize("t", [1, 2])
def test_two_processes_sharing_one_gpu_over_gloo(t):
    _launch(2, {"GS_DIST_BACKEND": "gloo", "GS_SHARE_DEVICE": "1"}, t)


def test_three_processes_sharing_one_gpu_over_gloo():
    _launch(3, {"GS_DIST_BACKEND": "gloo", "GS_SHARE_DEVICE": "1"}, 2)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one GPU per rank: runs where >= 2 GPUs are visible")
@pytest.mark.parametrize("world", [2, 4, 8])
def test_rccl_one_gpu_per_rank(world):
    if torch.cuda.device_count() < world:
        pytest.skip(f"{world} GPUs not visible")
    _launch(world, {"GS_DIST_BACKEND": "nccl"}, 2)
