"""GPU suite: the query-layer kernels of the C-ABI (rhj_col_filter, rhj_gather_tuples, rhj_pairs_split,
rhj_gather_u64, rhj_rows_filter_equal, rhj_sum_gather) bit-exact against numpy restatements of the
reference loops they replace (Query.cpp:66-74,96-146; structs.cpp:217-243; intermediate.cpp:72-105)."""
import numpy as np
import pytest

from oracle.pyoracle import PAIR, TUPLE

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [1, 63, 64, 65, 1000, 100_003, 3_000_000])
def test_col_filter_and_chain(engine, n):
    rng = np.random.default_rng(n)
    col = rng.integers(0, 1000, n, dtype=np.uint64)
    dcol, drows, drows2 = engine.to_device(col), engine.alloc(8 * n), engine.alloc(8 * n)
    for op, val, f in (("<", 500, lambda v: v < 500), (">", 998, lambda v: v > 998), ("=", 7, lambda v: v == 7), (">", 5000, lambda v: v > 5000)):
        m = engine.col_filter(dcol, None, n, op, val, drows)
        exp = np.nonzero(f(col))[0].astype(np.uint64)
        assert m == len(exp)
        got = np.sort(drows.to_numpy(np.uint64, m))
        assert np.array_equal(got, exp)
        # a second filter applied to the surviving rows (several filters on one alias)
        m2 = engine.col_filter(dcol, drows, m, ">", 100, drows2)
        exp2 = exp[col[exp.astype(np.int64)] > 100]
        assert m2 == len(exp2) and np.array_equal(np.sort(drows2.to_numpy(np.uint64, m2)), exp2)


def test_gather_tuples_split_gather_sum(engine):
    rng = np.random.default_rng(3)
    nrows, n = 50_000, 333_333
    col = rng.integers(0, 1 << 64, nrows, dtype=np.uint64)
    rows = rng.integers(0, nrows, n, dtype=np.uint64)
    dcol, drows, dt = engine.to_device(col), engine.to_device(rows), engine.alloc(16 * n)
    for pos in (False, True):
        engine.gather_tuples(dcol, drows, n, pos, dt)
        t = dt.to_numpy(TUPLE, n)
        assert np.array_equal(t["payload"], col[rows.astype(np.int64)])
        assert np.array_equal(t["key"], np.arange(n, dtype=np.uint64) if pos else rows)
    pairs = np.empty(n, dtype=PAIR)
    pairs["keyR"], pairs["keyS"] = rng.integers(0, 1 << 64, n, dtype=np.uint64), rng.integers(0, 1 << 64, n, dtype=np.uint64)
    dp, dr, ds = engine.to_device(pairs), engine.alloc(8 * n), engine.alloc(8 * n)
    engine.pairs_split(dp, n, dr, ds)
    assert np.array_equal(dr.to_numpy(np.uint64, n), pairs["keyR"]) and np.array_equal(ds.to_numpy(np.uint64, n), pairs["keyS"])
    dd = engine.alloc(8 * n)
    engine.gather_u64(dcol, drows, n, dd)
    assert np.array_equal(dd.to_numpy(np.uint64, n), col[rows.astype(np.int64)])
    assert engine.sum_gather(dcol, drows, n) == int(col[rows.astype(np.int64)].sum(dtype=np.uint64))     # wraps mod 2^64
    assert engine.sum_gather(dcol, drows, 0) == 0


def test_rows_filter_equal(engine):
    rng = np.random.default_rng(4)
    n = 200_001
    colA, colB = rng.integers(0, 50, 1000, dtype=np.uint64), rng.integers(0, 50, 2000, dtype=np.uint64)
    rA, rB = rng.integers(0, 1000, n, dtype=np.uint64), rng.integers(0, 2000, n, dtype=np.uint64)
    d = [engine.to_device(x) for x in (colA, rA, colB, rB)]
    dpos = engine.alloc(8 * n)
    m = engine.rows_filter_equal(d[0], d[1], d[2], d[3], n, dpos)
    exp = np.nonzero(colA[rA.astype(np.int64)] == colB[rB.astype(np.int64)])[0].astype(np.uint64)
    assert m == len(exp) and np.array_equal(np.sort(dpos.to_numpy(np.uint64, m)), exp)
