"""GPU suite: rhj_join on inputs of hundreds of MiB takes the PIPELINED path (S uploaded, partitioned and joined in chunks
against the partitioned R while finished pairs travel home).  The pair set must be what one join of the whole
relations yields: checked by exact count + order-insensitive checksum against the closed form of the PK/FK generators
(numpy restatement of SURVEY §8d / App. A), plus the fall-back to the plain path when a rowID does not fit the narrow format."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (np_mix / host_inputs: the generators of SURVEY §8d with numpy)
from radixhashjoin_amd import Engine, TUPLE  # noqa: E402

pytestmark = pytest.mark.gpu
C = np.uint64(0x100000001B3)


def chunks_of(nS):
    """the S chunks rhj_join uses (rhj_api.hip, join_host_pipelined): >= 8 Mi tuples each, at most 12"""
    k0 = min(12, nS // (8 << 20))
    chunk = (-(-nS // k0) + 4095) // 4096 * 4096
    return -(-nS // chunk)


def checksum(pairs):
    return int(np.sum(bench.np_mix(pairs["keyR"] * C ^ bench.np_mix(pairs["keyS"])), dtype=np.uint64))


def expected(S, n):
    """closed form: S tuple {j, mix(k)} matches exactly R row k - 1, k = 1 + mix(j ^ 42) % n"""
    j = S["key"]
    k1 = bench.np_mix(j ^ np.uint64(42)) % np.uint64(n)
    return len(S), int(np.sum(bench.np_mix(k1 * C ^ bench.np_mix(j)), dtype=np.uint64))


@pytest.mark.parametrize("n", [40_000_000, 72_000_000])
def test_pipelined_host_join_equals_the_closed_form(n):
    R, S = bench.host_inputs(n, TUPLE)
    S = S[: n - 1_234_567].copy()                         # |S| != |R|, last chunk ragged
    e = Engine(0)
    try:
        got = e.join(R, S)
        exp_n, exp_c = expected(S, n)
        assert len(got) == exp_n and checksum(got) == exp_c
        assert e.info("last.narrow") == 2 and e.info("last.pipelined") == chunks_of(len(S))
        # a rowID beyond 2^32 on the build side: the pipelined attempt is abandoned, the plain path repeats the join in the
        # 16-byte format; same pairs but for that one rowID
        R2 = R.copy()
        R2["key"][777] = np.uint64((1 << 40) + 777)
        got2 = e.join(R2, S)
        assert len(got2) == exp_n and e.info("last.narrow") == 0 and e.info("last.pipelined") == 0
        fix = got2["keyR"] == np.uint64((1 << 40) + 777)
        got2["keyR"][fix] = np.uint64(777)
        assert checksum(got2) == exp_c
    finally:
        e.close()


def test_pipelined_with_a_one_pass_plan_and_a_small_build_side():
    """|R| = 6M (one 9-bit pass), |S| = 40M: S still goes through in chunks"""
    n, m = 6_000_000, 40_000_000
    i = np.arange(n, dtype=np.uint64)
    R = np.empty(n, dtype=TUPLE)
    R["key"], R["payload"] = i, bench.np_mix(i + np.uint64(1))
    j = np.arange(m, dtype=np.uint64)
    S = np.empty(m, dtype=TUPLE)
    S["key"] = j
    k1 = bench.np_mix(j ^ np.uint64(42)) % np.uint64(n)
    S["payload"] = bench.np_mix(k1 + np.uint64(1))
    S["payload"][::11] ^= np.uint64(1 << 50)                 # every 11th foreign key matches nothing
    hit = np.ones(m, dtype=bool)
    hit[::11] = False
    exp_c = int(np.sum(bench.np_mix(k1[hit] * C ^ bench.np_mix(j[hit])), dtype=np.uint64))
    e = Engine(0)
    try:
        got = e.join(R, S)
        assert e.info("last.pipelined") == chunks_of(m)
        assert len(got) == int(hit.sum()) and checksum(got) == exp_c
    finally:
        e.close()
