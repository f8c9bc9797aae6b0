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


@pytest.mark.parametrize("n,mx,seed", [(14, 9, 0), (37, 40, 1), (60, 25, 2)])
def test_transport_regulariser_against_fp64_autograd(n, mx, seed):
    """opt_loss (ignr.py:184-209) and its gradient wrt the structure and the transport plan: closed form in the eigenbasis
    (rocSOLVER ssyevd + gs_gemm_f32 + spectrum kernels) against fp64 autograd through the reference's own formula — torch.svd,
    U f(S) V^T, torch.linalg.eigh — on a structure the IGNR produces (symmetric, zero diagonal, entries in (0, 1))."""
    from graphslim_amd.structure import TransportRegulariser, pinv_corner
    torch.manual_seed(seed)
    raw = torch.rand(n, n, dtype=torch.float64)
    adj = torch.sigmoid(2 * (raw + raw.T) - 2)
    adj = adj - torch.diag(torch.diag(adj))
    c = (torch.rand(mx, mx) < 0.3).double()
    corner = torch.triu(c, 1) + torch.triu(c, 1).T + 3 * torch.eye(mx, dtype=torch.float64)      # full rank: a well-defined pseudo-inverse
    Lx_inv = S.mx_inv(corner)
    np.testing.assert_allclose(pinv_corner(corner.numpy()), S.mx_inv(corner.float()).numpy(), atol=1e-5)
    P = torch.rand(mx, n, dtype=torch.float64)
    a64, p64 = adj.clone().requires_grad_(True), P.clone().requires_grad_(True)
    val = S.opt_loss(a64, p64, Lx_inv)
    ga, gp = torch.autograd.grad(val, [a64, p64])
    ga = (ga + ga.T) / 2                                        # the structure only moves along symmetric directions
    reg = TransportRegulariser(Lx_inv.float(), n, DEV)
    d_adj = torch.zeros(n, n, device=DEV)
    dP = torch.zeros(mx, n, device=DEV)
    scale = 1e-3
    reg.forward_backward(adj.float().to(DEV).contiguous(), P.float().to(DEV).contiguous(), scale, d_adj, dP)
    assert abs(float(reg.value) - float(val)) <= 2e-3 * abs(float(val)), (float(reg.value), float(val))
    got_a = d_adj.cpu().double() / scale
    got_a = (got_a + got_a.T) / 2
    assert float((got_a - ga).abs().max()) <= 2e-2 * float(ga.abs().max()), (float((got_a - ga).abs().max()), float(ga.abs().max()))
    got_p = dP.cpu().double() / scale
    assert float((got_p - gp).abs().max()) <= 2e-2 * float(gp.abs().max()), (float((got_p - gp).abs().max()), float(gp.abs().max()))
    assert float(reg.acc) == pytest.approx(float(reg.value), rel=1e-6)
