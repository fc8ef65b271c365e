"""GPU suite: BASELINE config 1 -- the reference's own golden workload small/small.init + small.work
through the GPU engine (radixhashjoin_amd/host/join_gpu = the reference's CLI protocol on top of
rhj_compat / rhj_query), byte-identical to the reference's small/small.result; and the 94 hot-path
calls it makes are the 94 calls the reference makes (tests/golden/small_joins.json, link-time tap)."""
import collections
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
JOIN = os.path.join(ROOT, "radixhashjoin_amd", "host", "join_gpu")


def run_small(tmp_path, mode="host"):
    assert os.path.exists(JOIN), "build with __graft_entry__.build()"
    log = tmp_path / "joins.log"
    env = dict(os.environ, RHJ_JOIN_LOG=str(log), RHJ_QUERY_MODE=mode)
    stdin = open(os.path.join(GOLD, "small", "small.init"), "rb").read() + open(os.path.join(GOLD, "small", "small.work"), "rb").read()
    out = subprocess.run([JOIN], input=stdin, cwd=GOLD, env=env, capture_output=True, timeout=600, check=True).stdout
    return out, log


def test_small_work_byte_identical(tmp_path, small_joins):
    out, log = run_small(tmp_path)
    expected = open(os.path.join(GOLD, "small", "small.result"), "rb").read()
    assert out == expected                                        # 50 lines of SUMs / NULLs
    meta, _ = small_joins
    want = collections.Counter((c["nR"], c["nS"], c["count"]) for c in meta)
    got = collections.Counter(tuple(int(x) for x in line.split()) for line in open(log))
    assert sum(got.values()) == 94
    assert got == want                                            # the same 94 joins, same sizes, same match counts


def test_small_work_device_resident_queries(tmp_path):
    """the same workload with the whole query device-resident (RHJ_QUERY_MODE=device: columns in HBM, device
    filters, position-carrying join inputs, gathered intermediates, device SUMs): same 50 lines"""
    out, _ = run_small(tmp_path, mode="device")
    assert out == open(os.path.join(GOLD, "small", "small.result"), "rb").read()


def test_small_work_level_batched_queries(tmp_path, small_joins):
    """RHJ_QUERY_MODE=batch: ONE thread runs each batch of queries level by level, the joins of a level through
    rhj_join_batch, sixteen per GPU launch.  Same 50 lines, and the same 94 joins with the same counts."""
    out, log = run_small(tmp_path, mode="batch")
    assert out == open(os.path.join(GOLD, "small", "small.result"), "rb").read()
    meta, _ = small_joins
    want = collections.Counter((c["nR"], c["nS"], c["count"]) for c in meta)
    got = collections.Counter(tuple(int(x) for x in line.split()) for line in open(log))
    assert sum(got.values()) == 94 and got == want


@pytest.mark.parametrize("mode", ["host", "device", "batch"])
def test_edge_queries_match_the_reference(mode):
    """corners of Query::run_joins that small.work never reaches: a projected alias that is never joined (sums to 0,
    Query.cpp:198-200 over an empty intermediate), two disconnected joins (the second drops the older columns,
    intermediate.cpp:147-162), the a-b / c-d / b-c order, a predicate between two aliases already joined.  Expected
    lines were printed by the REAL reference (tests/golden/make_edge.py); same-alias predicates are parity-unpinned (the
    reference segfaults on them)."""
    edge = os.path.join(GOLD, "edge")
    stdin = open(os.path.join(edge, "edge.init"), "rb").read() + open(os.path.join(edge, "edge.work"), "rb").read()
    env = dict(os.environ, RHJ_QUERY_MODE=mode)
    out = subprocess.run([JOIN], input=stdin, cwd=GOLD, env=env, capture_output=True, timeout=600, check=True).stdout
    assert out == open(os.path.join(edge, "edge.result"), "rb").read()


def test_join_batch_equals_the_golden_pair_sets(engine, small_joins):
    """rhj_join_batch (sixteen small joins per launch) over the 15 small.work calls whose inputs and pairs are in
    tests/golden/small_joins.npz, twice (the per-join device counters are left clean by the kernel), next to a join that is too
    large for the one-launch path, an empty one and a many-to-many one whose result outgrows 32x the guess"""
    from conftest import small_call_arrays
    from oracle.pyoracle import TUPLE, sorted_pairs
    meta, npz = small_joins
    with_vectors = [i for i, c in enumerate(meta) if c.get("vectors")]
    cases = [small_call_arrays(npz, i) for i in with_vectors]
    rng = np.random.default_rng(4)
    big_R = np.empty(300_000, dtype=TUPLE); big_R["key"] = np.arange(300_000); big_R["payload"] = rng.permutation(300_000)
    big_S = np.empty(400_000, dtype=TUPLE); big_S["key"] = np.arange(400_000); big_S["payload"] = rng.integers(0, 300_000, 400_000)
    dup_R = np.zeros(3_000, dtype=TUPLE); dup_R["key"] = np.arange(3_000); dup_R["payload"] = 7
    dup_S = np.zeros(2_000, dtype=TUPLE); dup_S["key"] = np.arange(2_000); dup_S["payload"] = 7           # 6 * 10^6 pairs from 5000 tuples
    empty = np.zeros(0, dtype=TUPLE)
    joins = [(R, S) for R, S, _ in cases] + [(big_R, big_S), (empty, cases[0][1]), (dup_R, dup_S)] + [(R, S) for R, S, _ in cases[:5]]
    for _ in range(2):
        got = engine.join_batch(joins)
        assert len(got) == len(joins)
        for (R, S, P), g in zip(cases + cases[:0], got[:len(cases)]):
            assert np.array_equal(sorted_pairs(g), sorted_pairs(P))
        k = len(cases)
        assert len(got[k]) == 400_000 and np.array_equal(big_R["payload"][got[k]["keyR"].astype(np.int64)], big_S["payload"][got[k]["keyS"].astype(np.int64)])
        assert len(got[k + 1]) == 0
        assert len(got[k + 2]) == 6_000_000 and len(np.unique(got[k + 2]["keyR"] * np.uint64(2_000) + got[k + 2]["keyS"])) == 6_000_000
        for (R, S, P), g in zip(cases[:5], got[k + 3:]):
            assert np.array_equal(sorted_pairs(g), sorted_pairs(P))


def test_join_batch_from_several_threads_with_a_context_each(small_joins):
    """eight host threads, eight contexts (MainScheduler.cpp:6-14), every one batching the golden joins and running one-pass joins
    (three-launch path) in between: contexts share nothing, results stay the golden pair sets"""
    import threading
    from conftest import small_call_arrays
    from oracle.pyoracle import sorted_pairs
    from radixhashjoin_amd import Engine, Opts
    meta, npz = small_joins
    cases = [small_call_arrays(npz, i) for i, c in enumerate(meta) if c.get("vectors")]
    want = [sorted_pairs(P) for _, _, P in cases]
    rng = np.random.default_rng(9)
    from oracle.pyoracle import TUPLE
    big_R = np.empty(200_000, dtype=TUPLE); big_R["key"] = np.arange(200_000); big_R["payload"] = rng.permutation(200_000)
    big_S = np.empty(300_000, dtype=TUPLE); big_S["key"] = np.arange(300_000); big_S["payload"] = rng.integers(0, 200_000, 300_000)
    errors = []

    def work(k):
        try:
            e = Engine(0)
            for rep in range(3):
                order = list(range(len(cases)))
                np.random.default_rng(k * 10 + rep).shuffle(order)
                got = e.join_batch([(cases[i][0], cases[i][1]) for i in order])
                for i, g in zip(order, got):
                    assert np.array_equal(sorted_pairs(g), want[i]), (k, rep, i)
                p = e.join(big_R, big_S, opts=Opts(1, 6))
                assert len(p) == 300_000 and np.array_equal(big_R["payload"][p["keyR"].astype(np.int64)], big_S["payload"][p["keyS"].astype(np.int64)])
            e.close()
        except Exception as ex:                                   # noqa: BLE001 -- reported by the main thread
            errors.append((k, repr(ex)))

    threads = [threading.Thread(target=work, args=(k,)) for k in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
