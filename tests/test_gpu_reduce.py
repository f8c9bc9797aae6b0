"""End-to-end checks on the GPU: the evaluation GCN against the oracle, a replay of a whole reference run
(golden g8) through the HIP engine, and the drop-in reduce()/Evaluator API.  Needs an MI355X."""
import os
import subprocess
import sys
import types

import numpy as np
import pytest
import scipy.sparse as sp
import torch

from oracle import gcond_oracle as O
from helpers import T, load_golden, sbm_graph

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _args(**kw):
    from graphslim_amd.utils import accuracy
    import logging
    base = dict(device=DEV, method="gcond", reduction_rate=0.5, verbose=False, lr=0.01, weight_decay=0.0, dropout=0.0,
                nlayers=2, with_bn=False, hidden=16, metric=accuracy, setting="trans", eval_epochs=20, eval_model="GCN",
                run_eval=3, run_inter_eval=2, seed=1, dataset="cora", threshold=0.05, attack=None, save_path="/tmp/gs_test",
                logger=logging.getLogger("gs_test"))
    base.update(kw)
    return types.SimpleNamespace(**base)


def _fake_data(a, x, y, idx_train, idx_val, idx_test, nclass):
    d = types.SimpleNamespace()
    d.adj_full, d.feat_full, d.labels_full = a, torch.from_numpy(x), torch.from_numpy(y)
    d.idx_train, d.idx_val, d.idx_test = (torch.from_numpy(np.asarray(i)) for i in (idx_train, idx_val, idx_test))
    for part, idx in (("train", idx_train), ("val", idx_val), ("test", idx_test)):
        setattr(d, f"adj_{part}", a[np.ix_(idx, idx)])
        setattr(d, f"labels_{part}", d.labels_full[idx])
        setattr(d, f"feat_{part}", d.feat_full[idx])
    d.nclass = nclass
    return d


def test_eval_gcn_follows_the_oracle():
    """models/base.py:80-225 on HIP ops vs oracle.fit_with_val / eval_test with identical parameter draws."""
    from graphslim_amd.models.gcn import GCN
    n, C, d, h = 600, 4, 10, 16
    a, x, y = sbm_graph(n, C, d, 8, seed=4)
    rng = np.random.default_rng(0)
    perm = rng.permutation(n)
    it, iv, ie = np.sort(perm[:100]), np.sort(perm[100:300]), np.sort(perm[300:])
    data = _fake_data(a, x, y, it, iv, ie, C)
    torch.manual_seed(3)
    ns = 24
    data.feat_syn = torch.randn(ns, d)
    adj = torch.rand(ns, ns)
    data.adj_syn = ((adj + adj.T) / 2).fill_diagonal_(0)
    data.labels_syn = torch.arange(ns) % C
    args = _args(hidden=h)
    for iters in (20, 7):
        torch.manual_seed(11)
        model = GCN(d, h, C, args, mode="eval")
        best = model.fit_with_val(data, train_iters=iters, normadj=True, setting="trans", reduced=True)
        acc = model.test(data, setting="trans")
        torch.manual_seed(11)
        O.init_gcn(d, h, C, 2)
        params = O.init_gcn(d, h, C, 2)
        rp, col, val = O.normalize_adj_sparse(a)
        An = O.csr_to_torch(rp, col, val, n)
        trace = []
        ref_best, w = O.fit_with_val(params, data.adj_syn, data.feat_syn, data.labels_syn, data.feat_full, An, T(iv),
                                     data.labels_val, iters, trace=trace)
        ref_acc = O.eval_test(w, data.feat_full, An, T(ie), data.labels_test)
        assert abs(best - ref_best) <= 1.01 / len(iv) and abs(acc - ref_acc) <= 1.01 / len(ie), (best, ref_best, acc, ref_acc)
        for got, ref in zip(model.parameters(), w):
            np.testing.assert_allclose(got.cpu().numpy(), ref.numpy(), atol=3e-3)
    out = model.predict(data.feat_full, data.adj_full)
    with torch.no_grad():
        ref = O.gcn_forward(data.feat_full, An, w)
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), atol=2e-2)


@pytest.mark.parametrize("tag", ["cora_like", "arxiv_like"])
def test_replay_of_a_whole_reference_run(golden_dir, tag):
    """Golden g8: the reference's GCond.__init__ + reduce (12 epochs x 2 outer iterations, inner steps, two checkpoints
    with intermediate evaluation) + Evaluator.evaluate, replayed through the HIP engine and the HIP evaluation GCN with
    the reference's RNG draw order and the same deterministic blocks."""
    from graphslim_amd import host as H
    from graphslim_amd.dataset.utils import sparsify
    from graphslim_amd.engine import BlockSet, EngineShapes, GCondEngine
    from graphslim_amd.models.gcn import GCN
    g = load_golden(golden_dir, "g8_reduce")
    n, nclass, d, hidden, t, inner, epochs, outer = [int(v) for v in g[f"{tag}_meta"]]
    a = sp.coo_matrix((np.ones(len(g[f"{tag}_a_row"])), (g[f"{tag}_a_row"], g[f"{tag}_a_col"])), shape=(n, n)).tocsr()
    x, y = g[f"{tag}_x"], g[f"{tag}_y"]
    idx_train, idx_val, idx_test = g[f"{tag}_idx_train"], g[f"{tag}_idx_val"], g[f"{tag}_idx_test"]
    labels_train = y[idx_train]
    csr = H.normalize_adj_sparse(a)
    ph = 256 if tag == "arxiv_like" else 128
    data = _fake_data(a, x, y, idx_train, idx_val, idx_test, nclass)
    args = _args(hidden=hidden, dataset="ogbn-arxiv" if tag == "arxiv_like" else "cora")
    H.seed_everything(1)
    labels_syn, ncd, ranges = H.generate_labels_syn(labels_train, 0.5)
    pge = H.init_pge_params(d, ph)
    sel = H.random_init_indices(idx_train, labels_train, ncd)
    sh = EngineShapes(n, d, nclass, len(labels_syn), t, hidden, ph, 32, 4, 3)
    eng = GCondEngine(sh, torch.from_numpy(x), None, None, None, labels_syn, [tuple(ranges[c]) for c in range(nclass)],
                      lr_feat=0.01, lr_adj=0.01, lr_model=0.01, inner_loop=inner, external_blocks=True, device=DEV)
    eng.set_feat(torch.from_numpy(x[sel]))
    eng.set_pge(pge)
    H.init_sgc_params(d, hidden, nclass, t)
    calls, loss_avg, best_val, losses, inter, saved = 0, 0.0, 0.0, [], [], None
    for it in range(epochs):
        eng.set_model(H.init_sgc_params(d, hidden, nclass, t))
        for _ in range(outer):
            batches = []
            for c in range(nclass):
                rng = np.random.default_rng(100000 + calls)
                calls += 1
                batch = rng.permutation(idx_train[labels_train == c])[:256]
                batches.append(O.sample_blocks(*csr, batch, [4, 3], rng))
            eng.set_blocks(BlockSet.from_batches(batches, 32, 4, 3, DEV))
            eng.outer_step(it)
        s, _ = eng.read_loss()
        loss_avg = (loss_avg + s) / (nclass * outer)
        if it in (5, 11):
            losses.append(loss_avg)
            data.adj_syn, data.feat_syn, data.labels_syn = eng.adj_syn.clone(), eng.feat_syn.clone(), torch.from_numpy(labels_syn)
            res = []
            for _ in range(2):
                m = GCN(d, hidden, nclass, args, mode="eval")
                res.append([m.fit_with_val(data, train_iters=20, setting="trans", reduced=True), m.test(data)])
            res = np.array(res).T
            inter.append(100 * res[0].mean())
            if res[0].mean() > best_val:
                best_val = res[0].mean()
                saved = (data.adj_syn.clone(), data.feat_syn.clone())
    np.testing.assert_allclose(np.array(losses), g[f"{tag}_loss_avg"], rtol=1e-3)
    np.testing.assert_allclose(eng.feat_syn.cpu().numpy(), g[f"{tag}_feat_syn"], atol=3e-3)
    np.testing.assert_allclose(eng.adj_syn.cpu().numpy(), g[f"{tag}_adj_syn"], atol=3e-3)
    log = str(g[f"{tag}_log"])
    vals = [float(l.split("Val:")[1].split()[0]) for l in log.split("\n") if "Val:" in l]
    np.testing.assert_allclose(inter, vals, atol=100 * 1.01 / len(idx_val))
    # final evaluation (eval_agent.py:234-291): +-0.3 pp gate of BASELINE.json
    data.adj_syn = sparsify("GCN", saved[0].clone(), types.SimpleNamespace(method="gcond", threshold=0.05, dataset="cora"))
    data.feat_syn = saved[1]
    accs = []
    for i in range(3):
        H.seed_everything(1 + i)
        m = GCN(d, hidden, nclass, args, mode="eval")
        m.fit_with_val(data, train_iters=20, setting="trans", reduced=True)
        accs.append(m.test(data))
    assert abs(np.mean(accs) - float(g[f"{tag}_eval_mean"])) <= max(0.003, 1.01 / len(idx_test)), (accs, float(g[f"{tag}_eval_mean"]))


def test_reduce_and_evaluate_api(tmp_path):
    """The drop-in boundary: cli -> get_dataset -> GCond(...).reduce -> Evaluator(...).evaluate on a cora-shaped graph
    (GPU sampler, product configuration), reduced-graph files in the reference's on-disk format."""
    import graphslim  # noqa: F401  (alias package)
    from graphslim.config import cli
    from graphslim.condensation import GCond
    from graphslim.dataset import get_dataset
    from graphslim.evaluation.eval_agent import Evaluator
    args = cli(["-D", "cora", "-M", "gcond", "-R", "0.5", "-G", "0", "--epochs", "20", "--outer_loop", "3", "--inner_loop", "2",
                "--eval_epochs", "40", "--run_eval", "2", "--save_path", str(tmp_path)], standalone_mode=False)
    args.run_inter_eval = 1
    graph = get_dataset(args.dataset, args)
    agent = GCond(setting=args.setting, data=graph, args=args)
    assert agent.nnodes_syn == 70
    reduced = agent.reduce(graph, verbose=args.verbose)
    assert reduced.adj_syn.shape == (70, 70) and reduced.feat_syn.shape == (70, 1433)
    adj = reduced.adj_syn.cpu()
    assert torch.equal(adj, adj.T) and float(adj.diagonal().abs().max()) == 0 and 0 <= float(adj.min()) and float(adj.max()) < 1
    d = os.path.join(str(tmp_path), "reduced_graph", "gcond")
    assert sorted(os.listdir(d)) == ["adj_cora_0.5_1.pt", "feat_cora_0.5_1.pt", "label_cora_0.5_1.pt"]
    lab = torch.load(os.path.join(d, "label_cora_0.5_1.pt"))
    assert lab.dtype == torch.int64 and lab.shape == (70,)
    assert torch.load(os.path.join(d, "adj_cora_0.5_1.pt")).dtype == torch.float32
    assert os.path.exists(os.path.join(str(tmp_path), "reduced_graph", "random", "feat_cora_0.5_1.pt"))   # init graph
    mean, std = Evaluator(args).evaluate(reduced, model_type="GCN", verbose=False)
    assert 0.4 < mean <= 1.0, mean        # 7 classes: chance is 0.14; the SBM graph is learnable


def test_example_script_runs(tmp_path):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "train_gcond.py"), "-D", "citeseer", "-M", "gcond", "-R",
                        "0.25", "-G", "0", "--epochs", "10", "--outer_loop", "2", "--inner_loop", "1", "--eval_epochs", "10",
                        "--run_eval", "1", "--save_path", str(tmp_path)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert os.path.exists(os.path.join(str(tmp_path), "reduced_graph", "gcond", "adj_citeseer_0.25_1.pt"))
    assert "Test Mean Accuracy" in open(os.path.join(str(tmp_path), "logs", "gcond", "citeseer_0.25.log")).read()


def test_cpu_device_is_refused():
    from graphslim_amd.condensation import GCond
    args = _args(device="cpu")
    with pytest.raises(RuntimeError, match="no CPU path"):
        GCond("trans", types.SimpleNamespace(labels_train=torch.zeros(4), feat_train=torch.zeros(4, 2)), args)
