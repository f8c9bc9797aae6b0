"""Host-side product logic (graphslim_amd/host.py, dataset/) against the reference golden vectors.  CPU only."""
import numpy as np
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
