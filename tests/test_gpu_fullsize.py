"""Oracle parity at FULL size for every 1-GPU configuration of BASELINE.json (VERDICT r01 "next round" 1b): the engine
samples its blocks with the HIP sampler, those very blocks and the engine's own (X~, theta_PGE, theta) go to
oracle.GCondState.outer_iteration (the CPU restatement of gcond.py:47-72 / gcond_base.py:156-241, itself pinned against the
reference's golden vectors), and ONE whole outer iteration is compared stage by stage:

  matching loss                    <= 1e-4 relative      (north_star gate)
  d loss / d X~  (direct + PGE)    <= 1e-3 of the tensor's max
  d loss / d theta_PGE, per tensor <= 1e-3 of the tensor's max
  the Adam step on the stepped group (well-conditioned entries identical), then — continuing the oracle from the engine's
  updated tensors — adj_syn of PGE.inference (2e-5 abs) and theta after `inner_loop` SGC steps.

Configs: ogbn-arxiv r = 0.5 % (the bench workload: N = 169 343, n' = 454 — the 128x128 MFMA tile, BGEN edge product and the
streaming PGE kernels run here, not the small-shape routes), Reddit r = 0.1 % inductive with fan-out [15, 8] and d = 602,
Cora r = 0.25, Citeseer and Pubmed r = 0.5 % (one synthetic node per class).  Needs an MI355X; the CPU side takes a few
seconds per config (reddit: the graph generator dominates).
"""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import gcond_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def rel_max(got, ref):
    got, ref = got.double().cpu(), ref.double().cpu()
    return float((got - ref).abs().max() / (ref.abs().max() + 1e-300))


class _ReluTies:
    """oracle.RELU_HOOK for the PGE's ReLUs: records the inputs within `tol` of zero (first training forward only) and, when `flip` names
    some of them (layer, flat index), gates those elements the OTHER way (value x * gate, gradient gate — what an fp32 forward whose
    sum lands on the other side of zero computes)."""

    def __init__(self, tol=1e-6, flip=()):
        self.tol, self.flip, self.found, self.calls, self.sign = tol, set(flip), [], {}, {}

    def __call__(self, layer, e):
        k = self.calls.get(layer, 0)
        self.calls[layer] = k + 1
        gate = (e > 0)
        if k == 0:      # the training forward of the iteration (the later calls are the inference pass after the update)
            flat = e.detach().reshape(-1)
            near = torch.nonzero(flat.abs() < self.tol).reshape(-1).tolist()
            self.found += [(layer, i) for i in near]
            self.sign.update({(layer, i): bool(flat[i] > 0) for i in near})
            idx = [i for (ly, i) in self.flip if ly == layer]
            if idx:
                gate = gate.clone().reshape(-1)
                gate[idx] = ~gate[idx]
                gate = gate.reshape(e.shape)
        return e * gate.to(e.dtype)


def run_config(name, epoch, lr_scale=1.0, operand_mode=None, driver="staged", relu_flip=None, scale=1.0, dropout=0.0):
    """driver: "staged" = the stage entry points one by one; "fused" = ONE gs_gcond_outer_step call (what bench.py times: two
    streams, blocks sampled inside the call); "fused_hoisted" = the same compared on the SECOND call, whose training forward the first
    call issued ahead beside its inner steps (the persistent kernel on 13/16 of the compute units at this shape; the test asserts that
    the hoisted pass was pending); "fused_reuse" = the same with forward reuse on, compared on the SECOND call (the
    one that adopts the cached inference forward of the first)."""
    import bench
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    p = bench.build_problem(name, seed=1, scale=scale)
    p["dropout"] = dropout
    eng = bench.make_engine(p, DEV)
    drop_it = 3
    eng.set_dropout_iter(drop_it)
    if operand_mode is not None:
        eng.set_operand_mode(operand_mode)
    bench.init_state(p, eng)
    torch.manual_seed(123)
    theta = bench.new_model(p, eng)
    if driver in ("fused_reuse", "fused_hoisted"):
        if driver == "fused_reuse":
            eng.set_forward_reuse(True)
        eng.outer_step(epoch)
        drop_it += 1
        torch.cuda.synchronize()
        assert eng.forward_cached() == (driver == "fused_reuse") and eng.forward_hoisted() == (driver == "fused_hoisted")
        theta = [t.cpu().clone() for t in eng.get_model()]          # theta after the first call's inner steps
    xs0, pge0 = eng.feat_syn.cpu().clone(), [t.cpu().clone() for t in eng.get_pge()]
    n, d = xs0.shape
    if driver == "staged":
        eng.stage_sample()
        batches = bench.blocks_to_batches(eng.get_blocks())
        # ---- engine: one outer iteration, stage by stage (what gs_gcond_outer_step runs) ---------------------------------
        eng.stage_real_grads()
        eng.stage_pge_forward(False)
        engine_z = None
        if relu_flip is not None:      # the ENGINE's inputs of the PGE's two ReLUs in this training forward (to resolve ties, see test_cora_r05_full_size)
            ph = eng.shapes.pge_hidden      # (the second ReLU: BatchNorm-2 of the layer-2 pre-activations E2 with the engine's batch statistics)
            e2, mu2, is2 = eng.buffer("E2", (n * n, ph)), eng.buffer("mu2"), eng.buffer("is2")
            engine_z = {1: ((e2 - mu2) * is2 * pge0[8] + pge0[9]).reshape(-1).clone()}
        eng.stage_syn_match()
        eng.syn_finish()
        loss = float(eng.buffer("loss")[0])
        eng.stage_pge_backward(True, True)
        g_feat = eng.buffer("d_feat", (n, d)).clone()
        g_pge = eng.buffer("d_pge").clone()
        eng.stage_update(epoch)
        eng.stage_inner()
    else:
        eng.outer_step(epoch)                                        # ONE library call
        batches = bench.blocks_to_batches(eng.get_blocks())          # the blocks that call sampled and used
        loss = float(eng.buffer("loss")[0])
        g_feat = eng.buffer("d_feat", (n, d)).clone()
        g_pge = eng.buffer("d_pge").clone()
    torch.cuda.synchronize()
    # ---- oracle on the same blocks and state ------------------------------------------------------------------------------
    st = O.GCondState(xs0, {"lin": pge0[:6], "bn": pge0[6:]}, torch.from_numpy(p["labels_syn"]), p["ncd"], p["data"].nclass,
                      p["lrf"], p["lra"], 0.01, p.get("model", "SGC"), len(p["fan"]))
    st.new_epoch([t.clone() for t in theta])
    hook = None
    if relu_flip is not None:
        hook = O.RELU_HOOK = _ReluTies(flip=relu_flip)
    if dropout:      # the engine's masks of this iteration (seed 1: workloads.make_engine), handed to the restatement
        from graphslim_amd.engine import dropout_keep

        def masks(where, x):
            kind, index, rows = where[:3]
            layer = where[3] if len(where) > 3 else 0      # (the GCN model's hook names the hidden layer)
            rows = np.arange(x.shape[0]) if rows is None else np.asarray(rows)
            keep = dropout_keep(1, drop_it, kind, index, rows, x.shape[1], dropout, layer)
            return torch.from_numpy(keep).to(x.dtype) * torch.tensor(np.float32(1.0 / (1.0 - dropout))).to(x.dtype)
        O.DROPOUT = masks
    try:
        ref_loss, ref_gfeat, ref_gpge, _ = st.outer_iteration(epoch, batches, p["data"].bench_feat, p["data"].bench_labels, 0)
    finally:
        O.RELU_HOOK = None
    out = dict(relu_ties=hook.found if hook else None, relu_sign=hook.sign if hook else None,
               engine_z=engine_z if driver == "staged" and relu_flip is not None else None, loss=loss, ref_loss=float(ref_loss), rel_loss=abs(loss - float(ref_loss)) / abs(float(ref_loss)),
               g_feat=rel_max(g_feat, ref_gfeat), g_pge=[], n=n, name=name, epoch=epoch)
    o = 0
    for i, (shp, ref) in enumerate(zip(eng.pge_shapes(), ref_gpge)):
        k = int(np.prod(shp))
        # biases of the Linear layers in front of a batch-statistics BatchNorm: mathematically zero gradient.  The fused call only
        # computes the gradient group it steps (gcond.py:54-61 discards the other unread): theta_PGE's in PGE epochs.
        if i not in (1, 3) and (driver == "staged" or epoch % 50 < 10):
            out["g_pge"].append((i, rel_max(g_pge[o:o + k].reshape(shp), ref)))
        o += k
    if driver != "staged" and epoch % 50 < 10:
        out["g_feat"] = 0.0          # (not computed by the fused call in a PGE epoch)
    if driver in ("fused_reuse", "fused_hoisted"):      # the optimisers of both sides have stepped a different number of times: gradients only
        out.update(upd_well_max=0.0, upd_any_frac=0.0, adj=0.0, model_bad_frac=0.0, model_max=0.0)
        O.DROPOUT = None
        return out
    # ---- the Adam step.  Its first step is lr * g / (|g| + eps) ~ lr * sign(g): entries whose gradient lies below the fp32
    # noise of the computation have an ill-conditioned sign, so the update is gated on the WELL-conditioned entries
    # (|g| > 1e-3 max|g| of their tensor), which must move identically; the share of all entries that differ is reported.
    if epoch % 50 >= 10:
        lr, new_e, new_o, refg = p["lrf"], eng.feat_syn.cpu(), st.feat_syn.detach(), [ref_gfeat]
    else:
        lr, new_e = p["lra"], eng.pge_params.cpu()
        new_o, refg = torch.cat([q.detach().reshape(-1) for q in st.pge_param_list()]), ref_gpge
    # (the two Linear biases in front of a batch-statistics BatchNorm have a mathematically ZERO gradient: what Adam sees there
    # is rounding noise, in the reference too — never well-conditioned)
    noise = (1, 3) if len(refg) > 1 else ()
    well = torch.cat([((g.abs() > 1e-3 * g.abs().max()) & (i not in noise)).reshape(-1) for i, g in enumerate(refg)])
    diff = (new_e.reshape(-1) - new_o.reshape(-1)).abs()
    out["upd_well_max"] = float(diff[well].max()) / lr
    out["upd_any_frac"] = float((diff > 0.02 * lr).double().mean())
    # ---- PGE.inference + inner steps: the oracle continues from the ENGINE's updated tensors, so that this stage is
    # compared on identical inputs whatever the ill-conditioned signs did
    with torch.no_grad():
        if epoch % 50 >= 10:
            st.feat_syn.copy_(new_e)
        else:
            for q, v in zip(st.pge_param_list(), eng.get_pge()):
                q.copy_(v.cpu())
    # yardstick for the inference structure: the oracle against ITSELF from X~ (1 + 1e-7 noise) — BatchNorm over n'^2 pair rows divides
    # by sqrt(var + 1e-5), so a column of E2 with (nearly) no variance amplifies fp32 rounding ~300x in the reference too
    with torch.no_grad():
        gq = torch.Generator().manual_seed(4321)
        pd_ = {"lin": [q.detach() for q in st.pge["lin"]], "bn": [q.detach() for q in st.pge["bn"]]}
        xs_now = st.feat_syn.detach()
        out["adj_self"] = float((O.pge_forward(xs_now * (1 + 1e-7 * torch.randn(xs_now.shape, generator=gq)), pd_) -
                                 O.pge_forward(xs_now, pd_)).abs().max())
    try:
        adj_inner = st.inference_and_inner(p["inner"])
    finally:
        O.DROPOUT = None
    out["adj"] = float((eng.adj_syn.cpu() - adj_inner).abs().max())
    mdiff = torch.cat([(a.cpu() - b.detach()).abs().reshape(-1) for a, b in zip(eng.get_model(), st.model)])
    out["model_bad_frac"] = float((mdiff > 0.02 * 0.01 * max(p["inner"], 1)).double().mean())
    out["model_max"] = float(mdiff.max())
    return out


def check(r, adj_tol=None):
    """adj_tol None: 2e-5, or 4x what the oracle's own inference structure moves under a 1e-7 perturbation of X~ (`adj_self`) where
    that is larger — a measured yardstick per configuration instead of round 4's hand-set 1e-3 / 2e-4 for cora r = 1.0."""
    print(r)
    if adj_tol is None:
        adj_tol = max(2e-5, 4.0 * r.get("adj_self", 0.0))
    assert r["rel_loss"] <= 1e-4, r
    assert r["g_feat"] <= 1e-3, r
    for i, e in r["g_pge"]:
        assert e <= 1e-3, (i, e, r)
    assert r["upd_well_max"] <= 1e-2, r          # well-conditioned entries: identical Adam move (in units of lr)
    assert r["upd_any_frac"] <= 1e-2, r          # ill-conditioned signs are rare
    assert r["adj"] <= adj_tol, r
    assert r["model_bad_frac"] <= 1e-3, r


@pytest.mark.parametrize("driver", ["fused", "fused_hoisted", "fused_reuse"])
@pytest.mark.parametrize("epoch", [10, 0])
def test_fused_outer_step_full_size(epoch, driver):
    """the call bench.py times — gs_gcond_outer_step, real side on the second stream, blocks sampled inside
    the call, forward reuse off and on, and the call whose training forward the previous call issued ahead — against the oracle at the
    full ogbn-arxiv shape, on the blocks that call drew."""
    check(run_config("ogbn-arxiv", epoch, driver=driver))


@pytest.mark.parametrize("epoch", [10, 0])
def test_ogbn_arxiv_r0005_full_size(epoch):
    """BASELINE configs[3] at one GPU, and the bench workload: feature-update (epoch 10) and PGE-update (epoch 0) iterations."""
    check(run_config("ogbn-arxiv", epoch))


def test_reddit_r0001_inductive_full_size():
    """BASELINE configs[4] (the 1-GPU half): inductive train sub-graph, fan-out [15, 8], d = 602, n' = 153."""
    check(run_config("reddit", 10))


@pytest.mark.parametrize("epoch", [10, 0])
def test_cora_r10_full_size_generated_forms_at_width_128(epoch):
    """Not a BASELINE config: cora at r = 1.0 gives n' = 140 >= 128 with the Planetoid PGE width 128, i.e. the generated-A1
    forward, the fused marginals with the regenerated mask and (feature epochs) the generated dE2 in their N = 128 instantiations
    — against the oracle like the configs above."""
    check(run_config("cora-r1.0", epoch))


def test_flickr_r0005_inductive_two_transforms_full_size():
    """graphslim/configs/gcond/flickr.json: the one upstream configuration that combines the inductive setting (train sub-graph,
    loader.py:125), two transforms and the wide fan-out [15, 8] (it had no full-size oracle test)."""
    check(run_config("flickr", 10))
    check(run_config("flickr", 0))


def test_cora_r05_full_size():
    """BASELINE configs[0]'s workload (Cora, r = 0.5 -> n' = 70) on the GPU path, feature- and PGE-update iterations.

    Strict gates (1e-3 on every gradient) on the fp32 pipe AND in the default operand form (exact three-way split).  One of the
    627 200 inputs of the PGE's second ReLU lies 3.3e-7 from zero here — a tie at rounding distance that two fp32 forwards with
    different summation orders resolve differently (profiles/r04/cora_r05_panel_vs_general_single_relu_flip.txt: dE2 differs by 2e-6
    on that element and by 4e-10 elsewhere; with only n'^2 = 4 900 pair rows one flipped gate is 1.4e-3 of max |dW1|).  Round 4 gated
    that run at a blanket 1e-2; round 5 searched over subsets of the reported ties until the strict gates held.  Now the oracle
    reports its ReLU inputs within 1e-6 of zero AND their signs, the engine's own value of each of those inputs is read back, and the
    oracle is replayed ONCE with exactly the gates the engine resolved the other way — nothing is searched, nothing else is loosened."""
    check(run_config("cora", 10, operand_mode="f32"))
    check(run_config("cora", 0, operand_mode="f32"))
    for epoch in (10, 0):
        r = run_config("cora", epoch, relu_flip=())
        ties = r["relu_ties"]
        print(f"cora r = 0.5, epoch {epoch}: ReLU inputs of the PGE within 1e-6 of zero in the oracle's forward: {ties}")
        assert len(ties) <= 4, ties
        # No search: a tie is resolved by the ENGINE's own value of that very ReLU input, read from its buffers —
        # it must be a tie there too (|z| <= 1e-5: else the engine computed something else at that element and the strict gate below
        # decides), and the oracle is replayed with exactly the gates whose sign the engine resolved the other way.
        flip = []
        for (layer, i) in ties:
            assert layer in r["engine_z"], f"a tie in ReLU {layer} of the PGE: only the second ReLU's inputs are read back from the engine"
            z = float(r["engine_z"][layer][i])
            assert abs(z) <= 1e-5, (layer, i, z)
            if (z > 0) != r["relu_sign"][(layer, i)]:
                flip.append((layer, i))
        print(f"   engine resolves {len(flip)} of {len(ties)} the other way: {flip}")
        if flip:
            r = run_config("cora", epoch, relu_flip=tuple(flip))
            print(f"   oracle replayed with those gates flipped: d theta_PGE errors {[round(e, 6) for _, e in r['g_pge']]}")
        check(r)


@pytest.mark.parametrize("name,epoch,driver", [("ogbn-arxiv", 10, "fused"), ("ogbn-arxiv", 0, "staged"), ("flickr", 10, "fused")])
def test_dropout_full_size_against_the_oracle(name, epoch, driver):
    """Dropout 0.5 in the condensation model (models/sgc.py:42) at the full ogbn-arxiv / flickr shapes (two transforms, hidden 256): the
    engine's masks on the real batches' gathered hidden rows (8 700 rows x 256 per class) and on the synthetic graph, handed to the
    oracle; the whole iteration — loss, gradients, the Adam step, PGE.inference and the inner steps under their own masks — to the
    gates of the dropout-free configurations."""
    check(run_config(name, epoch, driver=driver, dropout=0.5))


@pytest.mark.parametrize("name,scale", [("arxiv-k3", 0.15), ("cora-k3", 1.0)])
def test_three_hops_against_the_oracle(name, scale):
    """`--nlayers 3` (VERDICT r04 "missing" 2; dataset/loader.py:205-206 sizes [15, 10, 5], models/sgc.py:45-51): the engine samples
    THREE hops with its own sampler (a middle block with de-duplicated frontier and CSC between hop 1 and the outermost block), the
    oracle — pinned for three hops by the reference's golden G24 — replays one whole outer iteration on those very blocks: loss 1e-4,
    every gradient 1e-3, the Adam step, the inference structure, the inner steps; staged and as ONE fused call, feature and PGE epochs.
    (ogbn-arxiv shape at 15 % of its nodes: frontiers of ~20 000 rows per class keep the CPU side at seconds.)"""
    check(run_config(name, 10, scale=scale))
    check(run_config(name, 0, scale=scale))
    check(run_config(name, 10, scale=scale, driver="fused"))


def test_three_hops_full_scale_fused_iteration():
    """VERDICT r05 weak 2 / "next" 4b: bench.py quotes `--nlayers 3` at the FULL ogbn-arxiv shape (profiles/r05/bench_arxiv_three_hops.json:
    frontier capacities of 45 056 rows per class, 7 GB of workspace) — one whole fused iteration at that size against the oracle on the
    blocks the call sampled (feature epoch; the gates of every other full-size configuration)."""
    check(run_config("arxiv-k3", 10, scale=1.0, driver="fused"))


@pytest.mark.parametrize("name,scale", [("arxiv-t3", 0.15), ("arxiv-t4", 0.15), ("arxiv-t3", 1.0)])
def test_three_and_four_transforms_against_the_oracle(name, scale):
    """`--ntrans` 3 / 4 (VERDICT r05 "missing" 2; models/sgc.py:32-35: ntrans - 2 hidden-to-hidden transforms between the first and the
    last): the engine walks the layers (csrc/gs_deep.inc; the real side per distinct source of the outermost block), the oracle — whose
    SGC forward is the reference's loop over `layers`, pinned for three transforms by golden G27 — replays one whole outer iteration on
    the blocks the engine sampled: loss 1e-4, every gradient 1e-3, the Adam step, the inference structure, the inner steps; staged and as
    ONE fused call, feature and PGE epochs; the ogbn-arxiv shape at 15 % of its nodes and (three transforms) at full size."""
    check(run_config(name, 10, scale=scale))
    check(run_config(name, 0, scale=scale))
    check(run_config(name, 10, scale=scale, driver="fused"))


@pytest.mark.parametrize("scale", [0.15, 1.0])
def test_three_layer_gcn_condensation_model_against_the_oracle(scale):
    """`--condense_model GCN --nlayers 3` (VERDICT r05 "missing" 3; models/gcn.py:8-23, models/base.py:51-78): the engine samples three
    hops [15, 10, 5]; layer 1 aggregates over the outermost block, layer 2 over the middle block, layer 3 over hop 1's (csrc/gs_gcnk.inc);
    the oracle — pinned for three and four layers by golden G28 — replays one whole outer iteration on those blocks, at 15 % of the
    ogbn-arxiv shape (staged and fused, feature and PGE epochs) and once at its full size (frontiers of 45 056 rows per class)."""
    check(run_config("arxiv-gcn3", 10, scale=scale))
    if scale < 1.0:
        check(run_config("arxiv-gcn3", 0, scale=scale))
        check(run_config("arxiv-gcn3", 10, scale=scale, driver="fused"))


def test_gcn_condensation_model_with_dropout_against_the_oracle():
    """`--condense_model GCN` with dropout (VERDICT r05 "missing" 3; models/base.py:59: behind relu in every layer but the last — golden
    G29 pins the oracle): masks keyed by class, hidden layer and node id on the real batches' frontiers, one synthetic forward per class
    with per-class activations and adjoints (csrc/gs_gcnk.inc gcnk_syn_match_drop), fresh masks in every inner step.  Two layers at the
    FULL ogbn-arxiv shape (a whole staged iteration and a whole fused one), three layers / three hops at 15 % of it (feature and PGE
    epochs), to the gates of the dropout-free configurations."""
    check(run_config("arxiv-gcn2", 10, dropout=0.4))
    check(run_config("arxiv-gcn2", 10, driver="fused", dropout=0.4))
    check(run_config("arxiv-gcn3", 10, scale=0.15, driver="fused", dropout=0.3))
    check(run_config("arxiv-gcn3", 0, scale=0.15, dropout=0.3))


def test_three_hops_with_dropout_against_the_oracle():
    """`--nlayers 3` together with dropout in the condensation model: the masks act on the hidden rows the OUTERMOST of three gathers
    reads (keyed by node id) and on the per-edge indicator of the backward through three blocks, on the synthetic side on Q = A^^3 —
    a whole fused iteration and a staged PGE-epoch iteration at 15 % of the ogbn-arxiv shape."""
    check(run_config("arxiv-k3", 10, scale=0.15, driver="fused", dropout=0.3))
    check(run_config("arxiv-k3", 0, scale=0.15, dropout=0.3))


def test_cora_r025_full_size():
    """BASELINE configs[1] in its fp32 form (the bf16 operand mode of the same config: test_gpu_operand_modes.py)."""
    check(run_config("cora-r0.25", 10))
    check(run_config("cora-r0.25", 0))


@pytest.mark.parametrize("name", ["citeseer-r0.005", "pubmed-r0.005"])
def test_planetoid_r0005_full_size(name):
    """BASELINE configs[2]: r = 0.5 % of 120 / 60 train nodes -> one synthetic node per class (n' = 6 / 3)."""
    check(run_config(name, 10))
    check(run_config(name, 0))


@pytest.mark.parametrize("world,mode,driver", [(2, "f32", "c"), (3, "f32", "c"), (8, "f32", "c"), (2, "bf16x3", "c"), (3, "f32", "py")])
def test_sharded_ranks_reproduce_the_single_gpu_run_at_panel_shapes(world, mode, driver):
    """Class-parallel + pair-row sharding at a shape where the round-2 PGE forms are the ones that run
    (n' >= 128: A1 generated in the forward product, marginals fused with the regenerated mask, dE2 generated in feature epochs)
    with r_lo > 0 and ragged row ranges: `world` virtual ranks in lock step on one GPU (exchanges = sums of the ranks' buffers)
    against the unsharded engine, feature and PGE epochs.  The small-shape test of the same name in test_gpu_engine.py never
    reaches these kernels (n' ~ 30)."""
    import bench
    from graphslim_amd.parallel import Collective, ShardedGCond
    from test_gpu_engine import _LockStep
    p = bench.build_problem("ogbn-arxiv", seed=1, scale=0.4)
    n_syn = len(p["labels_syn"])
    assert n_syn >= 128, n_syn

    def build():
        eng = bench.make_engine(p, DEV)
        eng.set_operand_mode(mode)
        bench.init_state(p, eng)
        return eng

    torch.manual_seed(5)
    from graphslim_amd.host import init_sgc_params
    theta = init_sgc_params(p["data"].feat_full.shape[1], p["hidden"], p["data"].nclass, p["T"])
    ref = build()
    engs = [build() for _ in range(world)]
    shs = [ShardedGCond(e, r, world, [len(q) for q in p["pools"]], Collective(None)) for r, e in enumerate(engs)]
    lock = _LockStep(shs, driver)
    ph, d = p["ph"], p["data"].feat_full.shape[1]
    keep = np.ones(ref.pge_params.numel(), dtype=bool)       # the two biases in front of a batch-statistics BatchNorm: zero true gradient
    keep[ph * 2 * d: ph * 2 * d + ph] = False
    keep[ph * 2 * d + ph + ph * ph: ph * 2 * d + ph + ph * ph + ph] = False
    for epoch in (10, 0):
        ref.set_model(theta)
        for e in engs:
            e.set_model(theta)
        for it in range(2):
            ref.outer_step(epoch)
            lock.outer_step(epoch)
            l0 = float(ref.buffer("loss")[0])
            # the second PGE-epoch iteration starts from theta_PGE after an Adam step whose ill-conditioned entries may have moved
            # the other way (see below): its loss is compared at 5e-4, every other iteration at 2e-5
            gate = 5e-4 if (epoch == 0 and it == 1) else 2e-5
            for e in engs:
                assert abs(float(e.buffer("loss")[0]) - l0) <= gate * abs(l0), (epoch, it, float(e.buffer("loss")[0]), l0)
        for e in engs:
            if epoch == 10:
                # feature epochs: X~ stepped, theta_PGE untouched -> the inference pass must give the same structure
                np.testing.assert_allclose(e.feat_syn.cpu().numpy(), ref.feat_syn.cpu().numpy(), atol=2e-4)
                np.testing.assert_allclose(e.adj_syn.cpu().numpy(), ref.adj_syn.cpu().numpy(), atol=2e-4)
                # (theta after the inner steps: Adam's first steps are lr * sign-like, so an entry whose inner-step gradient sits in the
                # summation-order noise of X~ / A^ — the shard's edge-product slabs partition K differently from the full run's, round 5 —
                # moves the other way: the share of such entries is gated like in the PGE epochs below, the rest must agree)
                dm = np.abs(e.model_params.cpu().numpy() - ref.model_params.cpu().numpy())
                assert (dm > 5e-4).mean() <= 0.01 and dm.max() <= 2.1 * 0.01 * 3, (float((dm > 5e-4).mean()), float(dm.max()))
            else:
                # PGE epochs: Adam's first steps are lr * sign(g) — entries whose gradient sits in the summation-order noise
                # move the other way in a differently ordered sum (tests/test_gpu_fullsize.py::run_config explains the gate);
                # the matching losses above stayed within 2e-5 through both steps, here the share of agreeing entries
                diff = np.abs(e.pge_params.cpu().numpy()[keep] - ref.pge_params.cpu().numpy()[keep])
                assert (diff > 5e-4).mean() <= 0.02, float((diff > 5e-4).mean())
    for e in engs[1:]:
        assert torch.equal(e.feat_syn, engs[0].feat_syn) and torch.equal(e.pge_params, engs[0].pge_params)
        assert torch.equal(e.model_params, engs[0].model_params)
