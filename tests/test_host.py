"""Host-side product logic (graphslim_amd/host.py, dataset/) against the reference golden vectors.  CPU only."""
import os

import numpy as np
import pytest
import scipy.sparse as sp
import torch

from graphslim_amd import host as H
from helpers import load_golden


def test_labels_syn_matches_reference(golden_dir):
    g = load_golden(golden_dir, "g2_labels")
    for name in sorted({k.split("__")[0] for k in g.files}):
        ls, ncd, ranges = H.generate_labels_syn(g[name + "__labels_train"], float(g[name + "__rate"]))
        np.testing.assert_array_equal(ls, g[name + "__labels_syn"])
        assert list(ncd.keys()) == list(g[name + "__dict_keys"]) and list(ncd.values()) == list(g[name + "__dict_vals"])
        for c, (lo, hi) in ranges.items():
            assert np.all(ls[lo:hi] == c) and hi - lo == ncd[c]


def test_sparse_norm_bit_exact(golden_dir):
    g = load_golden(golden_dir, "g1_norm")
    n = int(g["n"])
    a = sp.coo_matrix((np.ones(len(g["a_row"])), (g["a_row"], g["a_col"])), shape=(n, n)).tocsr()
    rp, col, val = H.normalize_adj_sparse(a)
    dense = sp.csr_matrix((val, col, rp), shape=(n, n)).toarray()
    np.testing.assert_array_equal(dense, g["sparse_norm_dense"])
    trp, tcol, tval = H.csr_transpose(rp, col, val, n, n)
    np.testing.assert_array_equal(sp.csr_matrix((tval, tcol, trp), shape=(n, n)).toarray(), dense.T)


def test_init_draw_order(golden_dir):
    g = load_golden(golden_dir, "g3_init")
    torch.manual_seed(77)
    H.init_sgc_params(6, 5, 3, 2)
    for i, t in enumerate(H.init_sgc_params(6, 5, 3, 2)):
        np.testing.assert_array_equal(t.numpy(), g[f"sgc2_p{i}"])
    torch.manual_seed(99)
    for i, t in enumerate(H.init_pge_params(6, 128)):
        np.testing.assert_array_equal(t.numpy(), g[f"pge_p{i}"])
    assert H.pge_hidden("ogbn-arxiv", 0.005) == 256 and H.pge_hidden("cora", 0.5) == 128
    assert H.pge_hidden("reddit", 0.01) == 128 and H.pge_hidden("reddit", 0.001) == 256


def test_dataset_shaped_generators_and_contract():
    from graphslim_amd.dataset.loader import TransAndInd, splits
    from graphslim_amd.dataset.synthetic import SHAPES, make_raw
    raw = splits(make_raw("cora", seed=1))
    n, nnz, d, c, ntr, nva, nte = SHAPES["cora"][:7]
    assert raw.x.shape == (n, d) and int(raw.y.max()) + 1 == c
    assert int(raw.train_mask.sum()) == ntr and int(raw.val_mask.sum()) == nva and int(raw.test_mask.sum()) == nte
    data = TransAndInd(raw, "cora")
    assert abs(data.adj_full.nnz - nnz) / nnz < 0.1 and (data.adj_full != data.adj_full.T).nnz == 0
    assert data.adj_full.diagonal().sum() == 0
    np.testing.assert_allclose(data.feat_full.abs().sum(1).numpy(), 1.0, rtol=1e-5)        # L1 rows (loader.py:120-121)
    assert [int((data.labels_train == k).sum()) for k in range(c)] == [20] * c             # Planetoid split
    assert data.adj_train.shape == (ntr, ntr) and data.feat_val.shape == (nva, d)
    pools = data.class_pools("trans")
    assert sum(len(p) for p in pools) == ntr and set(pools[3]) <= set(data.idx_train.tolist())
    raw = splits(make_raw("ogbn-arxiv", seed=1, scale=0.02))
    data = TransAndInd(raw, "ogbn-arxiv")
    tr = data.feat_full[data.idx_train]
    np.testing.assert_allclose(tr.mean(0).numpy(), 0, atol=1e-4)                           # scaler fitted on train rows
    np.testing.assert_allclose(tr.std(0, unbiased=False).numpy(), 1, atol=1e-3)
    assert data.nclass == 40 and data.feat_full.shape[1] == 128
    ls, ncd, _ = H.generate_labels_syn(data.labels_train.numpy(), 0.05)
    assert len(ls) >= 40


def test_g10_sparsify_and_reduced_graph_files(golden_dir, tmp_path):
    """dataset/utils.py:8-66,136-258 of the reference vs graphslim_amd.dataset.utils: thresholds per (model, method,
    dataset), file names / dtypes of the saved condensed graph, and the independent fall-backs of load_reduced."""
    import logging
    from types import SimpleNamespace
    from helpers import load_golden
    from graphslim_amd.dataset import utils as DU
    g = load_golden(golden_dir, "g10_files")
    adj = g["adj"]
    for i, case in enumerate(g["cases"]):
        mt, method, ds, thr = str(case).split("|")
        args = SimpleNamespace(method=method, dataset=ds, threshold=float(thr))
        got = DU.sparsify(mt, torch.from_numpy(adj.copy()), args, verbose=False)
        np.testing.assert_array_equal(got.numpy(), g[f"sparsify{i}"], err_msg=str(case))
    base = dict(method="gcond", dataset="cora", seed=3, save_path=str(tmp_path), device="cpu", setting="trans", attack=None,
                logger=logging.getLogger("t"))
    args = SimpleNamespace(reduction_rate=0.25, **base)
    DU.save_reduced(torch.from_numpy(adj), torch.from_numpy(g["loaded_feat"]), torch.from_numpy(g["loaded_label"]), args)
    files = sorted(os.path.relpath(os.path.join(r, f), tmp_path) for r, _, fs in os.walk(tmp_path) for f in fs)
    assert files == list(g["files"])
    data = SimpleNamespace(feat_full=torch.zeros(60, 5), feat_train=torch.zeros(18, 5), labels_train=torch.from_numpy(g["missing_label"]))
    la, lf, ll = DU.load_reduced(args, data)
    assert [str(t.dtype) for t in (la, lf, ll)] == list(g["loaded_dtypes"])
    np.testing.assert_array_equal(la.numpy(), g["loaded_adj"])
    np.testing.assert_array_equal(lf.numpy(), g["loaded_feat"])
    np.testing.assert_array_equal(ll.numpy(), g["loaded_label"])
    ma, mf, ml = DU.load_reduced(SimpleNamespace(reduction_rate=0.125, **base), data)
    assert bool(g["missing_adj_is_eye"]) and torch.equal(ma, torch.eye(mf.shape[0]))
    assert list(mf.shape) == list(g["missing_feat_shape"])
    np.testing.assert_array_equal(ml.numpy(), g["missing_label"])
    _, mf3, _ = DU.load_reduced(SimpleNamespace(reduction_rate=0.125, **{**base, "setting": "ind"}), data)
    assert list(mf3.shape) == list(g["missing_feat_shape_ind"])


def _write_planetoid(root, name, rng, n_all=60, n_train=12, n_test=20, d=7, C=3, isolated=()):
    """Files in the Planetoid raw layout for a random graph; returns the arrays a reader must reproduce."""
    import pickle
    from collections import defaultdict
    raw = os.path.join(root, name, "raw")
    os.makedirs(raw)
    n = n_all + n_test + len(isolated)
    test_ids = np.setdiff1d(np.arange(n_all, n), np.asarray(isolated, dtype=np.int64))
    test_index = rng.permutation(test_ids)
    feats = (rng.random((n, d)) < 0.3).astype(np.float32)
    labels = rng.integers(0, C, n)
    feats[list(isolated)] = 0
    labels[list(isolated)] = 0
    onehot = np.eye(C, dtype=np.int32)[labels]
    onehot[list(isolated)] = 0
    g = defaultdict(list)
    edges = set()
    for _ in range(4 * n):
        a, b = rng.integers(0, n, 2)
        if a in isolated or b in isolated:
            continue
        edges.add((int(a), int(b)))
    for a, b in edges:                      # adjacency lists with a few duplicates and self loops, like the real files
        g[a].append(b)
        g[b].append(a)
    g[0].append(0)
    g[1] += [g[1][0]] if g[1] else []
    dump = {"x": sp.csr_matrix(feats[:n_train]), "y": onehot[:n_train], "allx": sp.csr_matrix(feats[:n_all]), "ally": onehot[:n_all],
            "tx": sp.csr_matrix(feats[test_index]), "ty": onehot[test_index], "graph": g}
    # (row i of tx / ty belongs to node test.index[i]; the reader moves it to that node's row)
    for k, v in dump.items():
        with open(os.path.join(raw, f"ind.{name}.{k}"), "wb") as f:
            pickle.dump(v, f)
    with open(os.path.join(raw, f"ind.{name}.test.index"), "w") as f:
        f.write("\n".join(str(int(i)) for i in test_index) + "\n")
    und = {(a, b) for a, b in edges if a != b} | {(b, a) for a, b in edges if a != b}
    return feats, labels, und, np.arange(n_train), np.arange(n_train, n_train + 500), test_index, n


def test_planetoid_raw_reader_roundtrip(tmp_path):
    from graphslim_amd.dataset.readers import find_and_read
    rng = np.random.default_rng(0)
    feats, labels, und, tr, va, te, n = _write_planetoid(str(tmp_path), "cora", rng, n_all=560, n_train=12, n_test=20)
    rec = find_and_read(str(tmp_path), "cora")
    assert rec is not None and rec.num_nodes == n
    # tx rows are stored in test.index order and re-placed by node id: features / labels of every node come back
    np.testing.assert_array_equal(rec.x.numpy(), feats)
    np.testing.assert_array_equal(rec.y.numpy(), labels)
    got = set(map(tuple, rec.edge_index.numpy().T.tolist()))
    assert got == und                                        # symmetrised, no self loops, no duplicates
    assert rec.train_mask.nonzero().view(-1).tolist() == tr.tolist()
    assert rec.val_mask.nonzero().view(-1).tolist() == va.tolist()
    assert sorted(rec.test_mask.nonzero().view(-1).tolist()) == sorted(te.tolist())


def test_planetoid_citeseer_isolated_test_nodes(tmp_path):
    from graphslim_amd.dataset.readers import find_and_read
    rng = np.random.default_rng(1)
    feats, labels, und, tr, va, te, n = _write_planetoid(str(tmp_path), "citeseer", rng, n_all=560, n_train=12, n_test=20,
                                                         isolated=(565, 571))
    rec = find_and_read(str(tmp_path), "citeseer")
    assert rec.num_nodes == n
    np.testing.assert_array_equal(rec.x.numpy(), feats)      # the isolated nodes are zero rows with label 0
    np.testing.assert_array_equal(rec.y.numpy(), labels)
    assert not rec.test_mask[565] and not rec.test_mask[571]


def test_graphsaint_raw_reader_and_get_dataset(tmp_path):
    import json
    from types import SimpleNamespace
    from graphslim_amd.dataset import get_dataset
    rng = np.random.default_rng(2)
    n, d, C = 300, 6, 5
    raw = os.path.join(str(tmp_path), "ogbn_arxiv", "raw")
    os.makedirs(raw)
    src, dst = rng.integers(0, n, 900), rng.integers(0, n, 900)
    keep = src != dst
    a = sp.coo_matrix((np.ones(keep.sum()), (src[keep], dst[keep])), shape=(n, n)).tocsr()
    a.data[:] = 1
    sp.save_npz(os.path.join(raw, "adj_full.npz"), a)                 # directed, like the citation graph
    x = rng.normal(size=(n, d)).astype(np.float32)
    np.save(os.path.join(raw, "feats.npy"), x)
    y = rng.integers(0, C, n)
    perm = rng.permutation(n)
    json.dump({str(i): int(y[i]) for i in range(n)}, open(os.path.join(raw, "class_map.json"), "w"))
    json.dump({"tr": perm[:150].tolist(), "va": perm[150:200].tolist(), "te": perm[200:].tolist()}, open(os.path.join(raw, "role.json"), "w"))
    args = SimpleNamespace(load_path=str(tmp_path), seed=1)
    data = get_dataset("ogbn_arxiv", args)
    assert data.num_nodes == n and data.nclass == C and args.nclass == C
    sym = ((a + a.T) > 0).astype(np.int64)
    assert (data.adj_full != sym).nnz == 0                             # loader.py:426-429 symmetrisation
    np.testing.assert_array_equal(np.sort(data.idx_train.numpy()), np.sort(perm[:150]))
    np.testing.assert_array_equal(data.labels_full.numpy(), y)
    # loader.py:113-119: features standardised with the TRAIN rows' statistics
    mu, sd = x[np.sort(perm[:150])].astype(np.float64).mean(0), x[np.sort(perm[:150])].astype(np.float64).std(0)
    np.testing.assert_allclose(data.feat_full.numpy(), (x - mu) / sd, rtol=1e-5, atol=1e-5)


def test_synthetic_fallback_is_loud_and_marked(monkeypatch, capsys):
    """get_dataset must never silently swap in the synthetic generator; split / pre_norm are passed through
    like loader.py:85-87."""
    import types
    import warnings
    from graphslim_amd.dataset.loader import get_dataset, splits
    monkeypatch.setenv("GRAPHSLIM_AMD_SCALE", "1")
    monkeypatch.delenv("GRAPHSLIM_AMD_SYNTHETIC", raising=False)
    args = types.SimpleNamespace(seed=1, load_path=None, pre_norm=False, split="fixed")
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        data = get_dataset("cora", args)
    assert any("SYNTHETIC" in str(x.message) for x in w)
    assert "SYNTHETIC" in capsys.readouterr().err
    assert data.synthetic is True and args.synthetic_data is True
    assert float(data.feat_full.sum(1).max()) > 1.5          # pre_norm = False: rows are NOT L1-normalised
    monkeypatch.setenv("GRAPHSLIM_AMD_SYNTHETIC", "1")
    args.pre_norm = True
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        data = get_dataset("cora", args)
    assert not w and abs(float(data.feat_full.sum(1).max()) - 1.0) < 1e-5
    # records without masks get the reference's class-wise random splits (dataset/utils.py:104-127)
    raw = types.SimpleNamespace(y=torch.arange(300) % 3, num_nodes=300)
    torch.manual_seed(0)
    few = splits(raw, "few")
    assert int(few.train_mask.sum()) == 15 and int(few.val_mask.sum()) == 15 and int(few.test_mask.sum()) == 270
    raw = types.SimpleNamespace(y=torch.arange(300) % 3, num_nodes=300)
    rnd = splits(raw, "random")
    assert int(rnd.train_mask.sum()) == 60 and int(rnd.val_mask.sum()) == 90
    raw = types.SimpleNamespace(y=torch.arange(300) % 3, num_nodes=300)
    dft = splits(raw, "fixed")
    assert int(dft.train_mask.sum()) == 240 and int(dft.val_mask.sum()) == 30 and int(dft.test_mask.sum()) == 30


def test_g14_kcenter_and_herding_selection_match_the_reference(golden_dir):
    """sparsification/kcenter.py:8-24, herding.py:8-26 (incl. first-occurrence tie-breaking on duplicated embeddings):
    bit-exact node ids against the reference's own select() (golden G14)."""
    from helpers import load_golden
    from graphslim_amd.sparsification.model_based import herding_select, kcenter_select
    g = load_golden(golden_dir, "g14_coreset_select")
    ncd = {int(k): int(v) for k, v in zip(g["ncd_keys"], g["ncd_vals"])}
    assert np.array_equal(kcenter_select(g["emb"], g["idx_train"], g["labels_train"], ncd), g["kcenter"])
    assert np.array_equal(herding_select(g["emb"], g["idx_train"], g["labels_train"], ncd), g["herding"])


@pytest.mark.parametrize("rate,n", [("0.1", 14), ("0.5", 70)])
def test_reads_condensed_graphs_written_by_the_reference(golden_dir, tmp_path, rate, n):
    """tests/golden/ref_artifacts/adj_cora_{0.1,0.5}_1.pt are files the REFERENCE's own save_reduced wrote
    (interface/reduced_graph/gcond/ upstream; the label / feat files next to them are corrupt archives upstream).  load_reduced
    must read them as the reference does (dataset/utils.py:155-258): the adjacency as saved — dense fp32 [n', n'], the raw PGE
    sigmoid output: symmetric, zero diagonal, entries in [0, 1) — with the reference's fall-backs for the missing feat / label
    files; sparsify then applies args.threshold (dataset/utils.py:47-48)."""
    import shutil
    import types
    from graphslim_amd.dataset.utils import load_reduced, sparsify
    d = tmp_path / "reduced_graph" / "gcond"
    d.mkdir(parents=True)
    shutil.copy(os.path.join(golden_dir, "ref_artifacts", f"adj_cora_{rate}_1.pt"), d / f"adj_cora_{rate}_1.pt")
    args = types.SimpleNamespace(save_path=str(tmp_path), method="gcond", dataset="cora", reduction_rate=float(rate), seed=1,
                                 device="cpu", setting="trans", threshold=0.05)
    data = types.SimpleNamespace(feat_full=torch.zeros(140, 3), feat_train=torch.zeros(140, 3), labels_train=torch.zeros(140, dtype=torch.long))
    adj, feat, lab = load_reduced(args, data)
    assert adj.dtype == torch.float32 and tuple(adj.shape) == (n, n)
    assert torch.equal(adj, adj.T) and float(adj.diagonal().abs().max()) == 0.0 and 0.0 <= float(adj.min()) and float(adj.max()) < 1.0
    assert feat is data.feat_full and lab is data.labels_train          # the reference's fall-backs
    sp_ = sparsify("GCN", adj.clone(), args)
    assert float(sp_[sp_ > 0].min()) >= 0.05 and int((sp_ > 0).sum()) == int((adj >= 0.05).sum())


def test_g15_appnp_condense_model_is_fenced_like_the_reference_fails(golden_dir):
    """SURVEY section 8 f4 names APPNP as a condense backbone, but upstream's own train_class cannot run it: golden G15 is the
    outcome of the reference's GCondBase.train_class with models.APPNP (TypeError at models/appnp.py:63, for ntrans 1 and 2).
    The package therefore refuses the value, and says why, instead of inventing a behaviour."""
    from helpers import load_golden
    from graphslim_amd.condensation.gcond import GCond
    g = load_golden(golden_dir, "g15_appnp_condense")
    for tag in ("t1", "t2"):
        assert str(g[f"{tag}_type"]) == "TypeError"
        assert "'list' and 'Tensor'" in str(g[f"{tag}_message"])
    agent = GCond.__new__(GCond)
    agent.args = type("A", (), {"condense_model": "APPNP", "verbose": False})()
    with pytest.raises(NotImplementedError, match="TypeError"):
        agent.reduce(None, verbose=False)


def test_g21_multi_label_is_fenced_like_the_reference_fails(golden_dir):
    """configs/gcond/{yelp,amazon}.json exist upstream, but the gradient-matching family cannot start on a multi-label dataset:
    golden G21 is the outcome of the reference's generate_labels_syn on a multi-hot label matrix (TypeError from collections.Counter
    on list rows, gcond_base.py:93).  The package refuses the datasets at configuration time and says why."""
    from helpers import load_golden
    from graphslim_amd import config
    g = load_golden(golden_dir, "g21_multi_label_labels_syn")
    assert str(g["type"]) == "TypeError" and "unhashable type: 'list'" in str(g["message"])
    for ds in ("yelp", "amazon"):
        args = type("A", (), {"dataset": ds, "reduction_rate": -1, "epochs": 100})()
        with pytest.raises(NotImplementedError, match="TypeError"):
            config.setting_config(args)


def test_g22_gcn_condense_model_with_bn_is_fenced_like_the_reference_fails(golden_dir):
    """with_bn + condense_model GCN: golden G22 is the outcome of the reference's check_bn + train_class (RuntimeError from
    BatchNorm1d on the 3-D activations GraphConvolution returns for the dense synthetic adjacency) — no behaviour to match."""
    from helpers import load_golden
    from graphslim_amd.condensation.gcond import GCond
    g = load_golden(golden_dir, "g22_gcn_with_bn_train_class")
    assert str(g["gcn_type"]) == "RuntimeError" and "running_mean should contain" in str(g["gcn_message"])
    agent = GCond.__new__(GCond)
    agent.args = type("A", (), {"condense_model": "GCN", "verbose": False, "nlayers": 2, "ntrans": 2, "with_bn": True, "dis_metric": "ours"})()
    with pytest.raises(NotImplementedError, match="RuntimeError"):
        agent.reduce(None, verbose=False)


def test_g17_evaluation_gcn_with_bn_is_fenced_like_the_reference_fails(golden_dir):
    """with_bn reaches the evaluation model through the shared args (eval_agent.py, gcond_base.py:287-340), but upstream's GCN
    cannot run it on a condensed graph: golden G17 is the outcome of the reference's fit_with_val(reduced=True) — a RuntimeError
    from BatchNorm1d on the 3-D activations of GraphConvolution (layers.py:41-47).  The package refuses and says why; the
    condensation side of with_bn is built (golden G16, tests/test_gpu_engine.py)."""
    from helpers import load_golden
    from graphslim_amd.models.gcn import GCN
    g = load_golden(golden_dir, "g17_eval_gcn_with_bn")
    assert str(g["type"]) == "RuntimeError"
    assert str(g["message"]) == f"running_mean should contain {int(g['n_syn'])} elements not {int(g['hidden'])}"
    args = type("A", (), {"nlayers": 2, "with_bn": True, "device": "cuda:0"})()
    with pytest.raises(NotImplementedError, match="RuntimeError"):
        GCN(10, 16, 4, args, mode="eval")


@pytest.mark.parametrize("tag", ["t1", "t2"])
def test_g18_msgc_host_allocation_is_bit_exact(golden_dir, tag):
    """MSGC's host-side preparation (msgc.py:41-64 labels, :178-222 greedy edge lists with randperm tie breaks) in the product's
    numpy form against the reference's own run: bit-exact integers, with the reference's RNG draw order."""
    from graphslim_amd import host as H
    from graphslim_amd.structure import BatchLayout
    g = load_golden(golden_dir, "g18_msgc")
    n, nclass, d, hidden, t, inner, epochs, outer, B = [int(v) for v in g[f"{tag}_meta"]]
    y, idx = g[f"{tag}_y"], g[f"{tag}_idx_train"]
    H.seed_everything(1)
    y_syn, each = H.msgc_labels(y[idx], int(len(idx) * 0.5), nclass)
    np.testing.assert_array_equal(y_syn, g[f"{tag}_y_syn"])
    assert each.sum() == len(y_syn) == int(len(idx) * 0.5) and each.min() >= 1
    H.init_edge_mlp_params(d)
    H.init_sgc_params(d, hidden, nclass, t)
    rows, cols, batch = H.msgc_edges(y_syn, nclass, B)
    for got, key in ((rows, "rows"), (cols, "cols"), (batch, "batch")):
        np.testing.assert_array_equal(got, g[f"{tag}_{key}"])
    # the class-grouped block-diagonal layout: a bijection whose class ranges are contiguous and B times the per-structure counts
    lay = BatchLayout(y_syn, nclass, B)
    assert sorted(lay.pos.reshape(-1).tolist()) == list(range(B * len(y_syn)))
    assert [hi - lo for lo, hi in lay.ranges] == (B * each).tolist()
    assert np.array_equal(lay.labels_big, np.repeat(np.arange(nclass), B * each))
    assert np.array_equal(np.tile(y_syn, B), g[f"{tag}_labels_out"])


@pytest.mark.parametrize("tag", ["t1", "t2"])
def test_g20_ignr_draw_order(golden_dir, tag):
    """SGDD's RNG draw order on the host (ignr.py:95-149: six Linear, P, `output`, then reset_parameters() over every Linear):
    the `output` head is drawn LAST and never receives a gradient, so its final value in the reference's run is its initial
    draw — bit-exact here; P moves by at most lr per structure-epoch step from its U(0, 1) draw."""
    g = load_golden(golden_dir, "g20_sgdd")
    n, nclass, d, hidden, t, inner, epochs, outer, mx = [int(v) for v in g[f"{tag}_meta"]]
    H.seed_everything(1)
    labels_syn, _, _ = H.generate_labels_syn(g[f"{tag}_y"][g[f"{tag}_idx_train"]], 0.5)
    p = H.init_ignr_params(d, len(labels_syn), mx)
    names = list(g[f"{tag}_param_names"])
    np.testing.assert_array_equal(p["out_w"].numpy(), g[f"{tag}_p{names.index('output.weight')}"])
    np.testing.assert_array_equal(p["out_b"].numpy(), g[f"{tag}_p{names.index('output.bias')}"])
    P_final = g[f"{tag}_p{names.index('P')}"]
    steps = 10 * outer                                  # structure epochs: it % 50 < 10
    assert p["P"].shape == P_final.shape and float(np.abs(p["P"].numpy() - P_final).max()) <= 0.01 * steps * 1.25
    assert 0 <= float(p["P"].min()) and float(p["P"].max()) <= 1


@pytest.mark.parametrize("counts,B,seed", [([1, 1, 1], 2, 0), ([1, 4, 2], 3, 1), ([3, 3, 3, 3], 2, 2), ([2, 7], 4, 3), ([5], 2, 4),
                                            ([1, 9, 1, 6, 2], 2, 5)])
def test_msgc_edge_lists_numpy_form_equals_the_torch_restatement(counts, B, seed):
    """msgc.py:178-222 in the product's numpy form against the oracle's line-by-line torch restatement (itself pinned by G18) on
    label layouts the golden does not hold: singleton classes (no partner inside the class), one class only, many ties — same
    randperm draws, bit-equal edge lists."""
    import torch
    from oracle import msgc_oracle as M
    y = np.repeat(np.arange(len(counts)), counts)
    torch.manual_seed(seed)
    a = H.msgc_edges(y, len(counts), B)
    torch.manual_seed(seed)
    b = M.msgc_edges(torch.from_numpy(y), len(counts), B)
    for u, v in zip(a, b):
        np.testing.assert_array_equal(u, v.numpy())
    rows, cols, batch = a
    assert np.all(rows != cols) and len(rows) % 2 == 0
    # both directions of every link are present within each structure
    for s in range(B):
        m = batch == s
        assert set(zip(rows[m].tolist(), cols[m].tolist())) == set(zip(cols[m].tolist(), rows[m].tolist()))


@pytest.mark.parametrize("n_syn,seed", [(7, 0), (12, 1), (40, 2), (41, 3)])
def test_msgc_label_allocation_equals_the_torch_restatement(n_syn, seed):
    """msgc.py:41-64 (float32 torch arithmetic upstream) in numpy float32 against the oracle's torch form on skewed class counts."""
    from oracle import msgc_oracle as M
    rng = np.random.default_rng(seed)
    nclass = 5
    lab = rng.choice(nclass, size=300, p=[0.5, 0.25, 0.15, 0.07, 0.03])
    y, each = H.msgc_labels(lab, n_syn, nclass)
    y2, each2 = M.msgc_labels(lab, n_syn, nclass)
    np.testing.assert_array_equal(y, y2.numpy())
    np.testing.assert_array_equal(each, each2.numpy())
    assert each.sum() == n_syn and each.min() >= 1
