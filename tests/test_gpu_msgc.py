"""MSGC on the GPU (graphslim/condensation/msgc.py): the edge-list structure generator (csrc/gs_structgen.hip + gs_gemm_f32)
against fp64 autograd, one outer iteration against the oracle, the replay of the reference's own run (golden G18) and the
drop-in reduce() / Evaluator API.  Needs an MI355X."""
import os

import numpy as np
import pytest
import scipy.sparse as sp
import torch

from oracle import gcond_oracle as O
from oracle import msgc_oracle as M
from helpers import load_golden, sbm_graph

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _case(n_per_class=(5, 7, 9), B=3, d=12, hid=64, seed=0):
    from graphslim_amd import host as H
    from graphslim_amd.structure import BatchLayout
    torch.manual_seed(seed)
    nclass = len(n_per_class)
    y_syn = np.repeat(np.arange(nclass), n_per_class)
    rows, cols, batch = H.msgc_edges(y_syn, nclass, B)
    params = H.init_edge_mlp_params(d, hid)
    # non-trivial BatchNorm affine parameters
    params[2], params[3] = 1 + 0.3 * torch.randn(hid), 0.2 * torch.randn(hid)
    params[6], params[7] = 1 + 0.3 * torch.randn(hid), 0.2 * torch.randn(hid)
    x = torch.randn(len(y_syn), d)
    return y_syn, nclass, rows, cols, batch, params, x, BatchLayout(y_syn, nclass, B)


def _ref_structure(x, params, rows, cols, batch, lay):
    """fp64 torch restatement of msgc.py:225-228 in the generator's block-diagonal row layout (before normalisation)."""
    r, c, b = (torch.as_tensor(a) for a in (rows, cols, batch))
    s = M.edge_scores(x, params, r, c)
    pos = torch.as_tensor(lay.pos)
    big = torch.zeros(lay.n_big, lay.n_big, dtype=x.dtype).index_put((pos[b, r], pos[b, c]), s)
    return (big + big.T) / 2


@pytest.mark.parametrize("hid,B", [(64, 3), (256, 2)])
def test_edge_structure_generator_against_fp64_autograd(hid, B):
    """Forward structure and every gradient of the generator (10 parameter tensors and X~) for a random upstream gradient."""
    from graphslim_amd.structure import EdgeStructureGenerator
    y_syn, nclass, rows, cols, batch, params, x, lay = _case(B=B, hid=hid)
    d = x.shape[1]
    gen = EdgeStructureGenerator(params, d, rows, cols, batch, lay, 0.01, DEV, hidden=hid)
    xd = x.to(DEV).contiguous()
    adj = gen.forward(xd).cpu()
    p64 = [p.double().requires_grad_(True) for p in params]
    x64 = x.double().requires_grad_(True)
    ref = _ref_structure(x64, p64, rows, cols, batch, lay)
    np.testing.assert_allclose(adj.numpy(), ref.detach().numpy(), atol=2e-6)
    assert torch.equal(adj, adj.T) and float(adj.diagonal().abs().max()) == 0.0
    up = torch.randn(lay.n_big, lay.n_big)
    grads = torch.autograd.grad((ref * up.double()).sum(), p64 + [x64])
    gflat, dx = gen.backward(up.to(DEV).contiguous())
    got = [g.cpu() for g in gen.views(gflat)] + [dx.cpu()]
    for i, (a, b) in enumerate(zip(got, grads)):
        b = b.float().reshape(a.shape)
        scale = float(b.abs().max())
        if i in (1, 5):       # biases in front of a batch-statistics BatchNorm: the exact gradient is zero
            assert float(a.abs().max()) <= 1e-4 * max(1.0, float(grads[0].abs().max())), (i, float(a.abs().max()))
            continue
        np.testing.assert_allclose(a.numpy(), b.numpy(), atol=2e-4 * scale + 1e-7, err_msg=f"gradient {i}")
    # determinism: a second forward / backward reproduces every bit
    adj2 = gen.forward(xd).cpu()
    g2, dx2 = gen.backward(up.to(DEV).contiguous())
    assert torch.equal(adj, adj2) and torch.equal(g2.cpu(), torch.cat([g.reshape(-1) for g in got[:-1]])) and torch.equal(dx2.cpu(), got[-1])


def test_tile_and_fold_are_adjoint_index_maps():
    from graphslim_amd.structure import EdgeStructureGenerator
    y_syn, nclass, rows, cols, batch, params, x, lay = _case()
    gen = EdgeStructureGenerator(params, x.shape[1], rows, cols, batch, lay, 0.01, DEV, hidden=64)
    big = torch.empty(lay.n_big, x.shape[1], device=DEV)
    gen.tile(x.to(DEV).contiguous(), big)
    assert torch.equal(big.cpu(), x[torch.as_tensor(lay.node_of_pos)])
    assert np.array_equal(lay.labels_big, y_syn[lay.node_of_pos]) and all(np.all(lay.labels_big[lo:hi] == c) for c, (lo, hi) in enumerate(lay.ranges))
    g = torch.randn(lay.n_big, x.shape[1])
    out = torch.empty_like(x).to(DEV)
    gen.fold(g.to(DEV).contiguous(), out)
    ref = torch.zeros_like(x).index_add_(0, torch.as_tensor(lay.node_of_pos), g)
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), atol=1e-5)
    a = torch.randn(lay.n_big, lay.n_big)
    b3 = gen.batched(a.to(DEV)).cpu()
    pos = torch.as_tensor(lay.pos)
    assert torch.equal(b3[1, 2, 5], a[pos[1, 2], pos[1, 5]])


def _golden_setup(g, tag, external_blocks=True, sizes=(4, 3)):
    from graphslim_amd import host as H
    from graphslim_amd.condensation.msgc import MSGCLoop
    from graphslim_amd.engine import EngineShapes, GCondEngine
    from graphslim_amd.structure import BatchLayout
    n, nclass, d, hidden, t, inner, epochs, outer, B = [int(v) for v in g[f"{tag}_meta"]]
    a = sp.coo_matrix((np.ones(len(g[f"{tag}_a_row"])), (g[f"{tag}_a_row"], g[f"{tag}_a_col"])), shape=(n, n)).tocsr()
    x, y, idx_train = g[f"{tag}_x"], g[f"{tag}_y"], g[f"{tag}_idx_train"]
    labels_train = y[idx_train]
    H.seed_everything(1)
    y_syn, n_each = H.msgc_labels(labels_train, int(len(idx_train) * 0.5), nclass)
    params = H.init_edge_mlp_params(d)
    H.init_sgc_params(d, hidden, nclass, t)                       # model constructor draw
    rows, cols, batch = H.msgc_edges(y_syn, nclass, B)
    sel = H.random_init_indices(idx_train, labels_train, {c: int(k) for c, k in enumerate(n_each)})
    lay = BatchLayout(y_syn, nclass, B)
    sh = EngineShapes(n, d, nclass, lay.n_big, t, hidden, 4, 32, sizes[0], sizes[-1], fmid=tuple(sizes[1:-1]))
    eng = GCondEngine(sh, torch.from_numpy(x), None, None, None, lay.labels_big, lay.ranges, lr_feat=0.01, lr_adj=0.01,
                      lr_model=0.01, inner_loop=inner, external_blocks=external_blocks, device=DEV)
    loop = MSGCLoop(eng, params, rows, cols, batch, lay, torch.from_numpy(x[sel]), 0.01, 0.01)
    return dict(a=a, x=x, y=y, idx_train=idx_train, labels_train=labels_train, csr=H.normalize_adj_sparse(a), y_syn=y_syn,
                n_each=n_each, rows=rows, cols=cols, batch=batch, params=params, lay=lay, eng=eng, loop=loop, sel=sel,
                meta=(n, nclass, d, hidden, t, inner, epochs, outer, B))


def _blocks(s, calls, sizes=(4, 3)):
    from graphslim_amd.engine import BlockSet
    nclass = s["meta"][1]
    batches = []
    for c in range(nclass):
        rng = np.random.default_rng(100000 + calls + c)
        batches.append(O.sample_blocks(*s["csr"], rng.permutation(s["idx_train"][s["labels_train"] == c])[:256], list(sizes), rng))
    return batches, BlockSet.from_batches(batches, 32, sizes[0], sizes[-1], DEV, fmid=tuple(sizes[1:-1]))


@pytest.mark.parametrize("tag,epoch,sizes", [("t1", 0, (4, 3)), ("t1", 10, (4, 3)), ("t2", 0, (4, 3)), ("t2", 10, (4, 3)),
                                             ("t1", 10, (4, 3, 2)), ("t2", 0, (4, 3, 2)), ("t2", 10, (3, 2, 2, 2))])
def test_one_outer_iteration_against_the_oracle(golden_dir, tag, epoch, sizes):
    """msgc.py:95-118 once (a structure epoch and a feature epoch): matching loss, d loss / d X~, d loss / d theta of the edge MLP,
    the updated tensors, the normalised [B, n', n'] batch of the inner steps and the model after them.  Round 6: also with `--nlayers`
    3 / 4 (K sampled hops, K propagations through the batch of structures — the dense products on the block-diagonal A^; the oracle's
    K-hop train_class is pinned by golden G24, its MSGC loop by G18)."""
    from graphslim_amd import host as H
    g = load_golden(golden_dir, "g18_msgc")
    s = _golden_setup(g, tag, sizes=sizes)
    n, nclass, d, hidden, t, inner, epochs, outer, B = s["meta"]
    eng, loop = s["eng"], s["loop"]
    model = H.init_sgc_params(d, hidden, nclass, t)
    eng.set_model(model)
    st = M.MSGCState(torch.from_numpy(s["x"][s["sel"]]), s["params"], torch.from_numpy(s["y_syn"]), torch.from_numpy(s["n_each"]),
                     nclass, *(torch.from_numpy(a) for a in (s["rows"], s["cols"], s["batch"])), B, 0.01, 0.01, 0.01, model, nlayers=len(sizes))
    batches, blocks = _blocks(s, 0, sizes)
    eng.set_blocks(blocks)
    loop.step(epoch)
    loss, g_x, g_p, adj_in = st.outer_iteration(epoch, batches, torch.from_numpy(s["x"]), torch.from_numpy(s["y"]).long(), inner)
    assert abs(float(eng.buffer("loss")[0]) - float(loss)) <= 1e-4 * abs(float(loss))
    got_p = [v.cpu() for v in loop.gen.views(loop.gen.grad)]
    for i, (a, b) in enumerate(zip(got_p, g_p)):
        if i in (1, 5) or epoch % 50 >= 10:
            continue          # zero-gradient biases in front of batch-statistics BatchNorm; feature epochs skip the unread parameter gradients
        np.testing.assert_allclose(a.numpy(), b.reshape(a.shape).numpy(), atol=1e-3 * float(b.abs().max()) + 1e-9, err_msg=f"theta {i}")
    if epoch % 50 >= 10:
        np.testing.assert_allclose(loop.gen.dX.cpu().numpy(), g_x.numpy(), atol=1e-3 * float(g_x.abs().max()))
        np.testing.assert_allclose(loop.x.cpu().numpy(), st.feat_syn.detach().numpy(), atol=2e-5)
    np.testing.assert_allclose(loop.inner_structure().cpu().numpy(), adj_in.numpy(), atol=2e-5 if epoch % 50 >= 10 else 2e-3)
    for a, b in zip(eng.get_model(), st.model):
        np.testing.assert_allclose(a.cpu().numpy(), b.detach().numpy(), atol=2e-4 if epoch % 50 >= 10 else 5e-3)


@pytest.mark.parametrize("tag", ["t1", "t2"])
def test_replay_of_the_reference_msgc_run(golden_dir, tag):
    """Golden G18: the reference's own MSGC.__init__ + reduce (3 epochs x 4 outer iterations, all of them structure epochs since
    it % 50 < 10) replayed on the HIP engine + generator with the reference's RNG draw order and the same deterministic blocks:
    labels, edge lists (bit exact), per-iteration matching losses, the final normalised batch and the window averages."""
    from graphslim_amd import host as H
    g = load_golden(golden_dir, "g18_msgc")
    s = _golden_setup(g, tag)
    n, nclass, d, hidden, t, inner, epochs, outer, B = s["meta"]
    assert np.array_equal(s["y_syn"], g[f"{tag}_y_syn"])
    for k in ("rows", "cols", "batch"):
        assert np.array_equal(s[k], g[f"{tag}_{k}"])
    eng, loop = s["eng"], s["loop"]
    calls, losses, xs, adjs = 0, [], [], []
    for it in range(epochs):
        eng.set_model(H.init_sgc_params(d, hidden, nclass, t), keep_optimizer=it > 0)
        for _ in range(outer):
            _, blocks = _blocks(s, calls)
            calls += nclass
            eng.set_blocks(blocks)
            loop.step(it)
            losses.append(float(eng.buffer("loss")[0]))
        xs.append(loop.x.cpu())
        adjs.append(loop.inner_structure().cpu())
    np.testing.assert_allclose(np.array(losses), g[f"{tag}_losses"], rtol=2e-3)
    np.testing.assert_allclose(loop.x.cpu().numpy(), g[f"{tag}_feat_syn"], atol=1e-6)           # structure epochs: X~ untouched
    np.testing.assert_allclose(adjs[-1].numpy(), g[f"{tag}_adj"], atol=3e-3)
    np.testing.assert_allclose((sum(adjs) / len(adjs)).numpy(), g[f"{tag}_best_adj"], atol=3e-3)
    np.testing.assert_allclose((sum(xs) / len(xs)).numpy(), g[f"{tag}_best_x"], atol=1e-6)
    for i, p in enumerate(loop.gen.views()):
        if i in (1, 5):
            continue
        diff = np.abs(p.cpu().numpy() - g[f"{tag}_pge{i}"])
        tol = 1e-3 + 1e-3 * np.abs(g[f"{tag}_pge{i}"])
        assert (diff > tol).mean() <= 2e-3 and diff.max() <= 0.05, (i, (diff > tol).mean(), diff.max())


def test_model_optimiser_state_survives_the_epoch_boundary(golden_dir):
    """msgc.py:90: optimizer_basic_model is built once — the second epoch's first model step continues the Adam moments and the
    step count (a fresh optimiser would take a full lr-sized first step)."""
    from graphslim_amd import host as H
    g = load_golden(golden_dir, "g18_msgc")
    s = _golden_setup(g, "t1")
    n, nclass, d, hidden, t, inner, epochs, outer, B = s["meta"]
    eng, loop = s["eng"], s["loop"]
    outs = {}
    for keep in (True, False):
        s = _golden_setup(g, "t1")
        eng, loop = s["eng"], s["loop"]
        torch.manual_seed(5)
        th = H.init_sgc_params(d, hidden, nclass, t)
        eng.set_model(th)
        _, blocks = _blocks(s, 0)
        eng.set_blocks(blocks)
        loop.step(0)
        eng.set_model(th, keep_optimizer=keep)
        eng.set_blocks(blocks)
        loop.step(0)
        outs[keep] = torch.cat([p.reshape(-1) for p in eng.get_model()]).cpu()
    assert not torch.equal(outs[True], outs[False])


@pytest.mark.parametrize("nlayers", [2, 3])
def test_msgc_reduce_api(tmp_path, nlayers):
    """`-M msgc` end to end through the reference's entry points (examples/train_all.py pattern): cli -> get_dataset ->
    MSGC(...).reduce -> Evaluator.evaluate on the [B, n', n'] batch, with the shipped configs/msgc/cora.json, batch_adj = 16
    from method_config (config.py:252-253) and the `averaging` initialiser that JSON names (given on the command line: the
    click default 'random' beats the JSON's init upstream too, config.py:30-34)."""
    import graphslim  # noqa: F401
    from graphslim.config import cli
    from graphslim.condensation import MSGC
    from graphslim.dataset import get_dataset
    from graphslim.evaluation.eval_agent import Evaluator
    from graphslim_amd.host import seed_everything
    args = cli(["-D", "cora", "-M", "msgc", "-R", "0.25", "-G", "0", "--epochs", "20", "--outer_loop", "2", "--inner_loop", "2",
                "--eval_epochs", "40", "--run_eval", "2", "--init", "averaging", "--nlayers", str(nlayers), "--save_path", str(tmp_path)],
               standalone_mode=False)      # (round 6: `--nlayers 3` — three sampled hops, three propagations — is honoured like upstream)
    assert args.batch_adj == 16 and args.init == "averaging" and args.lr_feat == 1e-4
    args.run_inter_eval = 1
    graph = get_dataset(args.dataset, args)
    seed_everything(args.seed)
    agent = MSGC(setting=args.setting, data=graph, args=args)
    reduced = agent.reduce(graph, verbose=False)
    n = int(graph.feat_train.shape[0] * 0.25)
    assert tuple(reduced.adj_syn.shape) == (16, n, n) and tuple(reduced.feat_syn.shape) == (n, graph.feat_train.shape[1])
    assert tuple(reduced.labels_syn.shape) == (16 * n,)
    adj = reduced.adj_syn.cpu()
    assert torch.isfinite(adj).all() and torch.allclose(adj, adj.transpose(1, 2), atol=1e-6)
    # every structure is sparse: at most 2 links per (node, class) plus the self loop
    assert int((adj[0] != 0).sum()) <= n * (2 * 2 * graph.nclass + 1)
    assert os.path.exists(os.path.join(str(tmp_path), "reduced_graph", "msgc", "adj_cora_0.25_1.pt"))
    mean, _ = Evaluator(args).evaluate(reduced, model_type="GCN", verbose=False)
    assert mean > 0.3, mean


def test_msgc_dropout_has_no_effect_as_upstream(tmp_path, golden_dir):
    """`--dropout` with `-M msgc`: upstream puts the condensation model into eval mode before every train_class and never back
    (msgc.py:99), so no mask is ever drawn — golden G30 records the reference's run with dropout 0.4 reproducing dropout 0 bit for bit.
    Here: accepted, and the condensed graph is bitwise the one of the run without it (two transforms, where GCond / SGDD do draw masks)."""
    import graphslim  # noqa: F401
    from graphslim.config import cli
    from graphslim.condensation import MSGC
    from graphslim.dataset import get_dataset
    from graphslim_amd.host import seed_everything
    g = load_golden(golden_dir, "g30_siblings_dropout")
    assert int(g["msgc_p4_masks_drawn"]) == 0 and np.array_equal(g["msgc_p0_losses"], g["msgc_p4_losses"])
    assert np.array_equal(g["msgc_p0_feat_syn"], g["msgc_p4_feat_syn"])
    out = []
    for extra in ([], ["--dropout", "0.4"]):
        args = cli(["-D", "cora", "-M", "msgc", "-R", "0.25", "-G", "0", "--epochs", "10", "--outer_loop", "2", "--inner_loop", "2", "--ntrans", "2",
                    "--hidden", "64", "--init", "averaging", "--save_path", str(tmp_path)] + extra, standalone_mode=False)
        args.run_inter_eval, args.checkpoints = 1, []
        graph = get_dataset(args.dataset, args)
        seed_everything(args.seed)
        agent = MSGC(setting=args.setting, data=graph, args=args)
        agent.reduce(graph, verbose=False)
        assert agent.engine.shapes.dropout == 0.0
        out.append((agent.feat_syn.clone(), agent.adj_syn.clone()))
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])


@pytest.mark.parametrize("tag", ["t1", "t2"])
def test_block_products_equal_the_dense_products(golden_dir, tag):
    """gs_gcond_set_block_structure: the per-block A^ A^ and dQ A^^T + A^^T dQ products give the iteration the dense products give
    (the structural zeros left out of the sums: same values up to the summation order) — loss, generator gradients, X~ after
    a feature epoch, the model after the inner steps."""
    from graphslim_amd import host as H
    g = load_golden(golden_dir, "g18_msgc")
    outs = []
    for blocks_on in (True, False):
        s = _golden_setup(g, tag)
        n, nclass, d, hidden, t, inner, epochs, outer, B = s["meta"]
        eng, loop = s["eng"], s["loop"]
        if not blocks_on:
            eng.set_block_structure(None)
        torch.manual_seed(11)
        eng.set_model(H.init_sgc_params(d, hidden, nclass, t))
        _, blocks = _blocks(s, 0)
        eng.set_blocks(blocks)
        loop.step(10)
        outs.append((float(eng.buffer("loss")[0]), loop.gen.grad.cpu().clone(), loop.gen.dX.cpu().clone(), loop.x.cpu().clone(),
                     torch.cat([p.reshape(-1) for p in eng.get_model()]).cpu(), loop.inner_structure().cpu()))
    a, b = outs
    assert abs(a[0] - b[0]) <= 1e-5 * abs(b[0])
    for k in (1, 2):
        assert float((a[k] - b[k]).abs().max()) <= 2e-5 * float(b[k].abs().max()) + 1e-9, k
    np.testing.assert_allclose(a[3].numpy(), b[3].numpy(), atol=1e-6)
    np.testing.assert_allclose(a[4].numpy(), b[4].numpy(), atol=1e-5)
    np.testing.assert_allclose(a[5].numpy(), b[5].numpy(), atol=1e-6)


@pytest.mark.parametrize("name,B,epoch", [("cora", 16, 10), ("cora", 16, 0), ("cora-r0.25", 16, 10)])
def test_full_size_iteration_against_the_oracle(name, B, epoch):
    """MSGC at the reference's cora configuration at FULL size (configs/msgc/cora.json: T = 1, batch_adj = 16; d = 1 433, C = 7,
    n' = 70 / 35 -> 1 120 / 560 rows, ~15 k / 7 k directed edges through the width-256 edge MLP) on blocks drawn by the HIP sampler:
    those very blocks go to oracle.MSGCState — matching loss 1e-4, d loss / d X~ and d loss / d theta of the edge MLP 1e-3 of the
    tensor's max, the structure of the inner steps."""
    import bench
    from graphslim_amd import host as H
    from graphslim_amd.condensation.msgc import MSGCLoop
    from graphslim_amd.engine import EngineShapes, GCondEngine
    from graphslim_amd.structure import BatchLayout
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    p = bench.build_problem(name)
    data = p["data"]
    d, nclass = data.feat_full.shape[1], data.nclass
    n_syn = int(data.feat_train.shape[0] * p["rate"])
    H.seed_everything(3)
    y_syn, each = H.msgc_labels(data.labels_train.numpy(), n_syn, nclass)
    params = H.init_edge_mlp_params(d)
    rows, cols, batch = H.msgc_edges(y_syn, nclass, B)
    lay = BatchLayout(y_syn, nclass, B)
    sh = EngineShapes(data.bench_n, d, nclass, lay.n_big, 1, p["hidden"], 4, 256, 10, 5)
    cls_ptr = np.concatenate([[0], np.cumsum([len(q) for q in p["pools"]])])
    inner = 3
    eng = GCondEngine(sh, data.bench_feat, p["csr"], cls_ptr, np.concatenate(p["pools"]), lay.labels_big, lay.ranges, lr_feat=1e-4,
                      lr_adj=1e-4, lr_model=0.01, inner_loop=inner, seed=1, device=DEV)
    sel = H.random_init_indices(data.idx_train.numpy(), data.labels_train.numpy(), {c: int(k) for c, k in enumerate(each)})
    x0 = data.bench_feat[sel].clone()
    loop = MSGCLoop(eng, params, rows, cols, batch, lay, x0, 1e-4, 1e-4)
    model = H.init_sgc_params(d, p["hidden"], nclass, 1)
    eng.set_model(model)
    loop.step(epoch)                                            # a feature epoch (d / d X~, the Adam step) or a structure epoch (d / d theta)
    batches = bench.blocks_to_batches(eng.get_blocks())         # the block set this iteration ran on
    st = M.MSGCState(x0, params, torch.from_numpy(y_syn), torch.from_numpy(each), nclass,
                     *(torch.from_numpy(a) for a in (rows, cols, batch)), B, 1e-4, 1e-4, 0.01, model)
    loss, g_x, g_p, adj_in = st.outer_iteration(epoch, batches, data.bench_feat, data.bench_labels.long(), inner)
    got = float(eng.buffer("loss")[0])
    assert abs(got - float(loss)) <= 1e-4 * abs(float(loss)), (got, float(loss))
    if epoch % 50 < 10:
        for i, (a, b) in enumerate(zip((v.cpu() for v in loop.gen.views(loop.gen.grad)), g_p)):
            if i in (1, 5):
                continue
            assert float((a - b.reshape(a.shape)).abs().max()) <= 1e-3 * float(b.abs().max()) + 1e-12, (i, float((a - b.reshape(a.shape)).abs().max()), float(b.abs().max()))
        assert torch.equal(loop.x.cpu(), x0)                    # X~ is not stepped in a structure epoch
        return
    assert float((loop.gen.dX.cpu() - g_x).abs().max()) <= 1e-3 * float(g_x.abs().max())
    # lr 1e-4 Adam step on X~: entries whose gradient is above the fp32 noise floor move identically
    big = g_x.abs() > 1e-3 * g_x.abs().max()
    assert float((loop.x.cpu() - st.feat_syn.detach())[big].abs().max()) <= 2e-6
    np.testing.assert_allclose(loop.inner_structure().cpu().numpy(), adj_in.numpy(), atol=1e-4)


@pytest.mark.parametrize("preset", [False, True])
def test_g19_averaging_and_clustering_initialisers(golden_dir, preset):
    """--init averaging / clustering (coarsening/averaging.py:33-71, clustering.py:36-76) against the reference's own output
    (golden G19): class means / BisectingKMeans(random_state = 0) cluster means, stand-alone and with MSGC-style preset labels."""
    import types
    from graphslim_amd.coarsening import Average, Cluster
    g = load_golden(golden_dir, "g19_inits")
    x, y, idx = g["x"], g["y"], g["idx_train"]
    tag = "preset" if preset else "alone"
    for name, cls in (("averaging", Average), ("clustering", Cluster)):
        data = types.SimpleNamespace(feat_train=torch.from_numpy(x[idx]), labels_train=torch.from_numpy(y[idx]), nclass=3)
        if preset:
            data.labels_syn = torch.from_numpy(np.repeat(np.arange(3), [3, 4, 5]))
        args = types.SimpleNamespace(device=DEV, reduction_rate=0.25, method=name)
        red = cls("trans", data, args).reduce(data, verbose=False, save=False)
        np.testing.assert_array_equal(np.asarray(red.labels_syn), g[f"{tag}_{name}_labels"])
        np.testing.assert_allclose(red.feat_syn.numpy(), g[f"{tag}_{name}_feat"], atol=2e-6, err_msg=name)
        assert torch.equal(red.adj_syn, torch.eye(len(red.labels_syn)))
