"""SGDD on the GPU (graphslim/condensation/sgdd.py, models/ignr.py): the IGNR structure generator against fp64 autograd of the
oracle's restatement, the optimal-transport regulariser, one outer iteration against the oracle, the replay of the reference's own
run (golden G20) and the drop-in reduce() API.  Needs an MI355X."""
import os

import numpy as np
import pytest
import scipy.sparse as sp
import torch

from oracle import gcond_oracle as O
from oracle import sgdd_oracle as S
from helpers import load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ignr_case(n=14, d=10, mx=9, seed=0):
    torch.manual_seed(seed)
    p = S.init_ignr(d, n, mx)
    for t in "cx":
        for l in range(2):
            p[f"{t}{l}_g"] = 1 + 0.3 * torch.randn(128)
            p[f"{t}{l}_be"] = 0.2 * torch.randn(128)
    return p, torch.randn(n, d)


@pytest.mark.parametrize("n,d", [(14, 10), (37, 6)])
def test_ignr_generator_against_fp64_autograd(n, d):
    """Training-mode structure, every parameter gradient and d / d X~ for a random upstream gradient; the running statistics after
    the pass; the eval-mode (inference) structure from them."""
    from graphslim_amd.structure import IGNRGenerator, IGNR_ORDER
    p, x = _ignr_case(n, d)
    gen = IGNRGenerator(p, d, n, 0.4, 0.01, DEV)
    xd = x.to(DEV).contiguous()
    adj = gen.forward(xd, train=True).cpu()
    p64 = {k: v.double().requires_grad_(True) for k, v in p.items()}
    x64 = x.double().requires_grad_(True)
    run = {k: [v[0].double(), v[1].double()] for k, v in S.new_running().items()}
    ref = S.ignr_structure(x64, p64, run, 0.4, train=True)
    np.testing.assert_allclose(adj.numpy(), ref.detach().numpy(), atol=3e-6)
    assert torch.equal(adj, adj.T) and float(adj.diagonal().abs().max()) == 0.0
    up = torch.randn(n, n)
    names = [k for k in IGNR_ORDER if k not in ("P", "out_w", "out_b")]
    grads = torch.autograd.grad((ref * up.double()).sum(), [p64[k] for k in names] + [x64])
    gflat, dx = gen.backward(up.to(DEV).contiguous())
    got = gen.views(gflat)
    wmax = max(float(g.abs().max()) for g in grads[:-1])
    for k, b in zip(names, grads[:-1]):
        a = got[k].cpu()
        b = b.float().reshape(a.shape)
        if k.endswith("_b") and k[1] in "01":       # biases in front of a batch-statistics BatchNorm: the exact gradient is zero
            assert float(a.abs().max()) <= 1e-4 * max(1.0, wmax), (k, float(a.abs().max()))
            continue
        np.testing.assert_allclose(a.numpy(), b.numpy(), atol=3e-4 * float(b.abs().max()) + 1e-7, err_msg=k)
    np.testing.assert_allclose(dx.cpu().numpy(), grads[-1].float().numpy(), atol=3e-4 * float(grads[-1].abs().max()))
    for key in ("c0", "c1", "x0", "x1"):
        np.testing.assert_allclose(gen.run[key][0].cpu().numpy(), run[key][0].float().numpy(), atol=1e-5)
        np.testing.assert_allclose(gen.run[key][1].cpu().numpy(), run[key][1].float().numpy(), rtol=1e-4, atol=1e-6)
    adj_eval = gen.forward(xd, train=False).cpu()
    with torch.no_grad():
        ref_eval = S.ignr_structure(x64.detach(), {k: v.detach() for k, v in p64.items()}, run, 0.4, train=False)
    np.testing.assert_allclose(adj_eval.numpy(), ref_eval.numpy(), atol=1e-5)
    assert float((adj_eval - adj).abs().max()) > 1e-3          # the two BatchNorm modes give different structures
