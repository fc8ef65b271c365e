"""GPU suite: one-pass joins in three launches (k_hist_fused2 / k_scatter_fused2 / the bucket join publishing its counters to
pinned host memory; rhj_api.hip join_one_pass_fused).  What the reference does per join with HistogramJob / PartitionJob /
JoinJob and five barriers (Result.cpp:90-124, structs.cpp:144-204) is here one histogram launch, one scatter launch whose
workgroups derive the partition boundaries themselves and reserve their output ranges with atomics while a planner workgroup
writes the task list, one join launch.  Against the CPU oracle, against the unfused launch sequence ("join.fused" 0), on one
context over many calls (two copies of the control block, each call's histogram launch zeroes the next call's), count-only,
overflow, empty partitions, skew."""
import numpy as np
import pytest

from oracle.pyoracle import PAIR, TUPLE, sorted_pairs
from radixhashjoin_amd import Engine, Opts
from radixhashjoin_amd.binding import GEN_R, GEN_S_UNIFORM, GEN_S_ZIPF, RhjError

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    e = Engine(0)
    yield e
    e.close()


def join_pairs(e, R, S, plan):
    dR, dS = e.to_device(R), e.to_device(S)
    n = e.join_dev(dR, len(R), dS, len(S), opts=plan)                      # count only
    dO = e.alloc(16 * max(n, 1))
    assert e.join_dev(dR, len(R), dS, len(S), dO, n, opts=plan) == n
    out = dO.to_numpy(PAIR, n)
    for x in (dR, dS, dO):
        x.free()
    return out


@pytest.mark.parametrize("nR,nS,D,bits", [(50_000, 80_000, 50_000, 4), (300_000, 200_000, 300_000, 7), (1_000_000, 1_000_000, 1_000_000, 8),
                                          (700_000, 2_500_000, 50_000, 9), (4_097, 4_095, 300, 1), (100, 70_000, 100, 8),
                                          (3_000_000, 3_000_000, 3_000_000, 9)])
def test_fused_one_pass_equals_oracle_and_unfused(eng, oracle, nR, nS, D, bits):
    R, S = oracle.gen_R(nR, D), oracle.gen_S_counter(nS, D + D // 8, 11)       # duplicates when D < nR, some dangling foreign keys
    exp = sorted_pairs(oracle.join(R, S)) if nR * (nS // max(D, 1) + 1) < 40_000_000 else None
    got = {}
    for fused in (1, 0):
        eng.set_option("join.fused", fused)
        got[fused] = sorted_pairs(join_pairs(eng, R, S, Opts(1, bits)))
        t = eng.timings()
        assert (t["passes"], t["bits1"]) == (1, bits)
    eng.set_option("join.fused", -1)
    assert np.array_equal(got[1], got[0])
    if exp is not None:
        assert np.array_equal(got[1], exp)


def test_launch_count_and_state_over_many_calls(eng, oracle):
    """3 launches instead of 8; the control block (global histograms, cursors) of the next call is zeroed by this call's kernels,
    whatever the two calls' radix widths: fifty joins of changing sizes and plans on one context, each checked by count + checksum"""
    eng.set_option("join.fused", 1)
    eng.set_profiling(True)
    rng = np.random.default_rng(3)
    for i in range(50):
        n = int(rng.integers(30_000, 2_000_000))
        m = int(rng.integers(30_000, 2_000_000))
        bits = int(rng.integers(3, 10))
        dR, dS, dO = eng.alloc(16 * n), eng.alloc(16 * m), eng.alloc(16 * m)
        eng.generate(GEN_R, dR, n, 0, n)
        eng.generate(GEN_S_ZIPF if i % 3 == 0 else GEN_S_UNIFORM, dS, m, 0, n, seed=i + 1, theta_milli=1100)
        exp = eng.expected_pkfk(dS, m)
        cnt = eng.join_dev(dR, n, dS, m, dO, m, opts=Opts(1, bits))
        t = eng.timings()
        launches = sum(v["launches"] for v in t.values() if isinstance(v, dict))
        assert launches == 3 or (i == 0 and launches == 4), t      # (the very first fused call on a context clears the control block)
        assert (cnt, eng.pairs_checksum(dO, cnt)) == exp
        assert 0 < eng.info("last.max_part_R") <= n and eng.info("last.max_part_S") <= m
        for x in (dR, dS, dO):
            x.free()
    eng.set_profiling(False)
    eng.set_option("join.fused", -1)


def test_overflow_reports_the_exact_count_and_the_next_call_is_clean(eng, oracle):
    R, S = oracle.gen_R(200_000, 1_000), oracle.gen_S_counter(150_000, 1_000, 3)       # 200 x 150 pairs per join value
    exp = oracle.join(R, S)
    dR, dS = eng.to_device(R), eng.to_device(S)
    small = eng.alloc(16 * 1000)
    with pytest.raises(RhjError):
        eng.join_dev(dR, len(R), dS, len(S), small, 1000, opts=Opts(1, 6))
    assert eng.join_dev(dR, len(R), dS, len(S), small, 1000, opts=Opts(1, 6), allow_overflow=True) == len(exp)
    dO = eng.alloc(16 * len(exp))
    assert eng.join_dev(dR, len(R), dS, len(S), dO, len(exp), opts=Opts(1, 6)) == len(exp)
    assert np.array_equal(sorted_pairs(dO.to_numpy(PAIR, len(exp))), sorted_pairs(exp))
    for x in (dR, dS, small, dO):
        x.free()


def test_host_pointer_call_takes_the_fused_path_too(eng, oracle):
    """rhj_join (host AoS in, page out) with a one-pass plan: the same three launches behind the two uploads"""
    R, S = oracle.gen_R(400_000, 400_000), oracle.gen_S_counter(900_000, 400_000, 9)
    got = eng.join(R, S)
    t = eng.timings()
    assert t["passes"] == 1 and eng.info("last.pipelined") == 0
    exp = oracle.join(R, S)
    assert np.array_equal(sorted_pairs(got), sorted_pairs(exp))


def test_large_one_pass_join_publishes_through_the_ticket_path(eng):
    """The packed result counter (pairs | finished workgroups << 48) is used while nR * nS < 2^48 and the grid stays below 2^16
    workgroups; a forced one-pass plan beyond that publishes through the ticket + read-back of the counters.  17M x 17M, 9 bits:
    33 K-tuple partitions through the chunked one-table kernel; count and checksum against the generator's closed form."""
    n = 17_000_000
    dR, dS, dO = eng.alloc(16 * n), eng.alloc(16 * n), eng.alloc(16 * n)
    eng.generate(GEN_R, dR, n, 0, n)
    eng.generate(GEN_S_UNIFORM, dS, n, 0, n, seed=5)
    exp = eng.expected_pkfk(dS, n)
    for _ in range(2):                                                       # (twice: the control block alternates)
        cnt = eng.join_dev(dR, n, dS, n, dO, n, opts=Opts(1, 9))
        assert (cnt, eng.pairs_checksum(dO, cnt)) == exp
    t = eng.timings()
    assert (t["passes"], t["bits1"]) == (1, 9)
    small = 1_000_000                                                        # and a small join right behind it (packed again)
    eng.generate(GEN_S_UNIFORM, dS, small, 0, small, seed=6)
    exp = eng.expected_pkfk(dS, small)
    cnt = eng.join_dev(dR, small, dS, small, dO, small)
    assert (cnt, eng.pairs_checksum(dO, cnt)) == exp
    for x in (dR, dS, dO):
        x.free()
