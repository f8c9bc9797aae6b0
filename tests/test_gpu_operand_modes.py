"""The opt-in operand precisions of the row-panel products (gs_gcond_set_operand_mode; VERDICT r01 row *J, BASELINE.json
configs[1] "bf16") against the fp32 CPU oracle, one whole outer iteration at FULL size on engine-sampled blocks:

  bf16x3  split bf16 x3 + fp32 accumulate  — must meet the SAME gates as fp32 (loss 1e-4 rel, gradients 1e-3 of max) at the
          bench configuration (ogbn-arxiv, n'^2 = 206 116 pair rows); at cora r = 0.25 (1 225 pair rows) the loss and d/dX~
          gates hold, but a handful of ReLU / BatchNorm sign flips of pre-activations within 2^-16 of zero move the small
          theta_PGE gradient tensors by up to ~4e-3 of their max (measured: 3.5e-3 on layers.1.weight) — gated at 1e-2 there,
          which is why fp32 stays the engine's default
  bf16    bf16 operands + fp32 accumulate  — reported; gated only against gross breakage (SURVEY trap 15: X~, the norms of
          distance_wb and every accumulator stay fp32, only the MFMA operands of the PGE layer-2 products are rounded)

The measured errors are printed (pytest -s) and asserted; DESIGN.md quotes them.  Needs an MI355X."""
import pytest

from test_gpu_fullsize import run_config

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["cora-r0.25", "ogbn-arxiv"])
@pytest.mark.parametrize("epoch", [10, 0])
def test_split_bf16x3_meets_the_fp32_gates(name, epoch):
    r = run_config(name, epoch, operand_mode="bf16x3")
    print("bf16x3", r)
    assert r["rel_loss"] <= 1e-4, r
    assert r["g_feat"] <= 1e-3, r
    for i, e in r["g_pge"]:
        assert e <= (1e-3 if name == "ogbn-arxiv" else 1e-2), (i, e, r)
    assert r["adj"] <= 1e-4, r


@pytest.mark.parametrize("name", ["cora-r0.25", "cora-r1.0", "ogbn-arxiv", "reddit"])
@pytest.mark.parametrize("epoch", [10, 0])
def test_exact_three_way_split_meets_every_fp32_gate(name, epoch):
    """Operand mode 'bf16x6' (exact 3-way bf16 split, 6 products, fp32 accumulate) is an fp32-accurate way to use the bf16 matrix
    pipe: it must pass the SAME gates as the fp32 path on every configuration — including cora, where the two-way split of
    'bf16x3' misses the 1e-3 gate on two small theta_PGE gradient tensors."""
    from test_gpu_fullsize import check
    r = run_config(name, epoch, operand_mode="bf16x6")
    print("bf16x6", r)
    check(r, adj_tol=1e-3 if (name == "cora-r1.0" and epoch == 0) else 2e-4)


@pytest.mark.parametrize("name", ["cora-r0.25", "ogbn-arxiv"])
def test_bf16_operands_error_is_recorded(name):
    r = run_config(name, 10, operand_mode="bf16")
    print("bf16", r)
    assert r["rel_loss"] <= 5e-2, r          # gross-breakage guards only: see the module docstring
    assert r["g_feat"] <= 0.5, r
    assert r["adj"] <= 5e-2, r
