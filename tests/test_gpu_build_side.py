"""GPU suite: which side of a partition becomes the hash table (rhj_kernels.hip build_on_S, DupSniff in rhj_internal.h).
The reference builds on the smaller bucket, S on a tie (JobScheduler.cpp:187); here, where the two sides are within 1/16 of each
other, the side whose sampled join values show fewer duplicates is built -- whichever argument it is.  The pairs must not depend on
any of it: both argument orders, sampling on and off, one-pass and two-pass plans, key / foreign-key, many-to-many and
unique-on-both-sides joins against the oracle; and at sizes the oracle does not reach, count + checksum of the sampled run against
the unsampled one."""
import numpy as np
import pytest

from oracle.pyoracle import PAIR, sorted_pairs
from radixhashjoin_amd import Engine, Opts
from radixhashjoin_amd.binding import GEN_R, GEN_S_UNIFORM

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    e = Engine(0)
    yield e
    e.set_option("join.sniff", -1)
    e.close()


def swapped(p):
    q = np.empty(len(p), dtype=PAIR)
    q["keyR"], q["keyS"] = p["keyS"], p["keyR"]
    return q


@pytest.mark.parametrize("shape", ["key_fk", "many_many", "unique_both", "fk_much_larger"])
@pytest.mark.parametrize("plan", [None, Opts(1, 6), Opts(2, 4, 4)])
def test_pairs_do_not_depend_on_order_or_sampling(eng, oracle, shape, plan):
    n = 120_000
    if shape == "key_fk":
        R, S = oracle.gen_R(n, n), oracle.gen_S_counter(n, n, 7)
    elif shape == "many_many":
        R, S = oracle.gen_R(n, n // 3), oracle.gen_S_counter(n, n // 3, 5)
    elif shape == "unique_both":
        R, S = oracle.gen_R(n, n), oracle.gen_R(n, n)
        S["key"] += np.uint64(1 << 20)
    else:
        R, S = oracle.gen_R(n // 4, n // 4), oracle.gen_S_counter(n, n // 4, 3)
    exp = sorted_pairs(oracle.join(R, S))
    for sniff in (1, 0):
        eng.set_option("join.sniff", sniff)
        got = eng.join(R, S, opts=plan)
        assert np.array_equal(sorted_pairs(got), exp), (shape, sniff)
        got = eng.join(S, R, opts=plan)                                   # the same join with the arguments exchanged
        assert np.array_equal(sorted_pairs(swapped(got)), exp), (shape, sniff, "swapped")


@pytest.mark.parametrize("n", [1_000_000, 24_000_000])
def test_both_orders_at_size_against_the_unsampled_run(eng, n):
    """one-pass (three launches, the counters in the control block) and two-pass (k_hist2d_units samples, k_make_tasks decides)"""
    dR, dS, dO = eng.alloc(16 * n), eng.alloc(16 * n), eng.alloc(16 * n)
    eng.generate(GEN_R, dR, n, 0, n)
    eng.generate(GEN_S_UNIFORM, dS, n, 0, n, seed=9)
    exp_n, exp_c = eng.expected_pkfk(dS, n)
    ref = {}
    for sniff in (0, 1, 1):                                               # (twice with sampling: the control block alternates)
        eng.set_option("join.sniff", sniff)
        cnt = eng.join_dev(dR, n, dS, n, dO, n)
        assert (cnt, eng.pairs_checksum(dO, cnt)) == (exp_n, exp_c)
        cnt = eng.join_dev(dS, n, dR, n, dO, n)                           # foreign-key side first
        chk = eng.pairs_checksum(dO, cnt)
        assert cnt == exp_n
        assert ref.setdefault("swapped", chk) == chk                      # the same pair set as without sampling
    for x in (dR, dS, dO):
        x.free()
