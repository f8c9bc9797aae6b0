"""The engine's counter-based dropout masks (csrc/gs_kernels.h gs_drop_mult4, restated in graphslim_amd.engine.dropout_keep) — the
statistics F.dropout's Bernoulli draws have (models/sgc.py:42), checked on the restatement; tests/test_gpu_engine.py checks that the
device draws exactly these masks (the oracle is handed them and must reproduce the engine's iteration)."""
import numpy as np
import pytest

from graphslim_amd.engine import dropout_keep


@pytest.mark.parametrize("p", [0.1, 0.5, 0.8])
def test_keep_rate_and_independence(p):
    rows = np.arange(2000)
    a = dropout_keep(1, 0, "syn", 3, rows, 256, p)
    n = a.size
    se = np.sqrt(p * (1 - p) / n)
    assert abs(a.mean() - (1 - p)) <= 5 * se
    # every column and every row keeps at the same rate (no structure along either axis)
    assert np.abs(a.mean(0) - (1 - p)).max() <= 6 * np.sqrt(p * (1 - p) / a.shape[0])
    assert np.abs(a.mean(1) - (1 - p)).max() <= 6 * np.sqrt(p * (1 - p) / a.shape[1])
    # another class, another iteration, another kind, another seed: independent masks (joint keep rate = product of the rates)
    for other in (dropout_keep(1, 0, "syn", 4, rows, 256, p), dropout_keep(1, 1, "syn", 3, rows, 256, p),
                  dropout_keep(1, 0, "real", 3, rows, 256, p), dropout_keep(1, 0, "inner", 3, rows, 256, p),
                  dropout_keep(2, 0, "syn", 3, rows, 256, p)):
        assert (other != a).any()
        joint = (a & other).mean()
        assert abs(joint - (1 - p) ** 2) <= 6 * np.sqrt((1 - p) ** 2 * (1 - (1 - p) ** 2) / n)
    # neighbouring entries of a row share one 64-bit word (16 bits each): still uncorrelated
    x = a.astype(np.float64) - a.mean()
    for lag in (1, 2, 3, 4):
        r = (x[:, :-lag] * x[:, lag:]).mean() / x.var()
        assert abs(r) <= 6 / np.sqrt(n)


def test_masks_are_a_function_of_the_node_id_not_of_its_position():
    """The real batch's mask is keyed by the node's id in the graph: the same node in another position of n_id (or in a class shard of
    another rank) draws the same entries, and the mask of a row does not depend on which other rows are asked for."""
    ids = np.array([7, 169000, 12, 7, 3])
    m = dropout_keep(5, 9, "real", 2, ids, 64, 0.4)
    assert (m[0] == m[3]).all()
    assert (m[1] == dropout_keep(5, 9, "real", 2, [169000], 64, 0.4)[0]).all()
    assert m.dtype == bool and m.shape == (5, 64)
    # the threshold resolves p to 1 / 65536 and never degenerates
    assert dropout_keep(1, 0, "syn", 0, np.arange(50), 64, 1e-9).all()
    assert 0 < dropout_keep(1, 0, "syn", 0, np.arange(5000), 64, 1 - 1e-9).mean() < 1e-3
