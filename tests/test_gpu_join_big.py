"""GPU suite: the bucket-join kernels for partitions whose build side does not fit one 16 B/tuple LDS table --
explicit plans with too few radix bits (BASELINE config 3 names 8+8 bits at 10^9 tuples: 15 K-tuple partitions) and
inputs beyond 2^30 tuples:
    kernel 1  k_join_bkt<1024, 8448>   16-byte entries, build side in chunks, probe side re-read per chunk (any plan)
    kernel 2  k_join_ct                8-byte {48-bit key | index} entries, both sides read once (plans removing >= 16 bits)
Reference semantics: JoinJob::run + Result::join_buckets (JobScheduler.cpp:186-192, Result.cpp:43-76): every
(rowR,rowS) with equal payloads, build side = smaller bucket.  Checked against the CPU oracle (sorted pair sets) and, at
10^9 tuples, by count + checksum against the closed form.  rhj_set_option("join.big_tables", 1) makes the engine use
these kernels whatever the partition sizes, so that small, oracle-sized inputs reach them.
The compact-table kernel also reads the narrow intermediate format ({payload 8 B, rowID 4 B}, k_scatter_wcn) that a join
with a fused two-pass plan uses while rowIDs fit 32 bits: "partition.narrow" 0 / 1 / 2 = 16-byte tuples throughout /
narrow partitions / narrow between the passes too.  Every case below runs in each format."""
import numpy as np
import pytest

from oracle.pyoracle import PAIR, TUPLE, sorted_pairs
from radixhashjoin_amd import Engine, Opts
from radixhashjoin_amd.binding import GEN_CONST, GEN_R, GEN_S_UNIFORM, GEN_S_ZIPF, unmix64

pytestmark = pytest.mark.gpu
BKT_BIG, CT, CT_HALF, CT_WIDE, CT_HALF_WIDE, CT_MID, CT_HALF_MID, CT_13, CT_HALF_MID_G, CT_G13 = 1, 2, 3, 4, 5, 6, 7, 8, 9, 10


def _forced(param):
    e = Engine(0)
    e.set_option("join.big_tables", 1)
    e.set_option("join.big_kernel", param[0])
    e.narrow = param[1]
    e.set_option("partition.narrow", e.narrow)
    return e


@pytest.fixture(scope="module", params=[(BKT_BIG, 0), (CT, 0), (CT_HALF, 0), (CT, 1), (CT, 2), (CT_HALF, 2)],
                ids=["bkt_big", "compact_table", "compact_table_half", "compact_table_narrow1", "compact_table_narrow2",
                     "compact_table_half_narrow2"])
def big(request):
    e = _forced(request.param)
    yield e
    e.close()


# the geometries added in round 3 (same kernel template, other table size / slot rows): a shorter list of cases, chosen at
# their table and task boundaries, plus duplicates, long buckets and the 17-18-bit plans
@pytest.fixture(scope="module", params=[(CT_WIDE, 2), (CT_HALF_WIDE, 2), (CT_MID, 0), (CT_MID, 2), (CT_HALF_MID, 2), (CT_13, 0), (CT_13, 2), (CT_HALF_MID_G, 0), (CT_HALF_MID_G, 2),
                                        (CT_G13, 0), (CT_G13, 2)],
                ids=["compact_table_20slots_narrow2", "compact_table_half_20slots_narrow2", "compact_table_mid",
                     "compact_table_mid_narrow2", "compact_table_half_mid_narrow2", "compact_table_8192_buckets",
                     "compact_table_8192_buckets_narrow2", "compact_table_half_mid_row_guards", "compact_table_half_mid_row_guards_narrow2",
                     "compact_table_13bit_index", "compact_table_13bit_index_narrow2"])
def geom(request):
    e = _forced(request.param)
    yield e
    e.close()


def used_format(engine, plan, wide_rowids=False):
    """the format the last join ran in: the requested narrow level under a fused 8+8 plan, else 16-byte tuples"""
    fused = (plan.passes, plan.bits1, plan.bits2) == (2, 8, 8)
    deep = plan.passes == 2 and plan.bits1 + plan.bits2 > 16 and max(plan.bits1, plan.bits2) <= 9      # 17-18 bits: level 2 only
    exp = 0 if wide_rowids else engine.narrow if fused else 2 if deep and engine.narrow == 2 else 0
    assert engine.info("last.narrow") == exp
    engine.set_option("partition.narrow", engine.narrow)        # re-arm after a fallback


def rel(rng, n, values, key0=0):
    t = np.empty(n, dtype=TUPLE)
    t["key"] = rng.permutation(n).astype(np.uint64) + np.uint64(key0)
    t["payload"] = values
    return t


def few_partitions(values, nlow):
    """payloads whose MIXED value (rhj_mix64, what joins take their radix digits from) is join value << 16 | one of `nlow`
    16-bit patterns chosen BY the value (the same on both sides of a join): a 16-bit radix plan then yields `nlow` large
    partitions -- what 10^9 tuples give every partition -- from an oracle-sized input"""
    lows = np.random.default_rng(nlow).permutation(1 << 16)[:nlow].astype(np.uint64)
    return unmix64((values << np.uint64(16)) | lows[(values % np.uint64(nlow)).astype(np.int64)])


def check(engine, oracle, R, S, plan, wide_rowids=False):
    got = engine.join(R, S, opts=plan)
    exp = oracle.join(R, S)
    assert len(got) == len(exp)
    assert np.array_equal(sorted_pairs(got), sorted_pairs(exp))
    used_format(engine, plan, wide_rowids)


@pytest.mark.parametrize("nR,nS,nlow", [(60_000, 200_000, 3),        # 20 K build / 66 K probe per partition: 2 chunks, 5 tasks
                                        (8_160, 8_192, 1), (8_161, 8_193, 1), (8_960, 8_000, 1),   # the half-size table / task, one beyond (8960: its 20-slot form)
                                        (200_000, 50_000, 4),        # build on S (the smaller bucket), pairs stay (rowR,rowS)
                                        (16_352, 16_384, 1),         # exactly one table, exactly one task
                                        (16_353, 16_385, 1),         # one tuple beyond each
                                        (17_920, 16_384, 1),         # (the table of rounds 1-2, now kernel 8: two chunks here)
                                        (300_000, 300_000, 1500)])   # 200-tuple partitions through the same kernels
def test_pkfk_16_bit_plan(big, oracle, nR, nS, nlow):
    pkfk_case(big, oracle, nR, nS, nlow)


def pkfk_case(eng, oracle, nR, nS, nlow, plan=None):
    rng = np.random.default_rng(nR + nS)
    rv = rng.permutation(1 << 22)[:nR].astype(np.uint64)
    R = rel(rng, nR, few_partitions(rv, nlow))
    S = rel(rng, nS, R["payload"][rng.integers(0, nR, nS)], key0=1 << 31)
    S["payload"][::97] ^= np.uint64(1 << 40)                        # some probe tuples match nothing
    check(eng, oracle, R, S, plan or Opts(2, 8, 8))


@pytest.mark.parametrize("nR,nS,nlow", [(60_000, 200_000, 3),                     # chunks and several tasks per partition
                                        (17_000, 20_480, 1), (8_900, 10_241, 1),   # the 20-slot tasks: exactly one, one beyond
                                        (12_288, 12_288, 1), (12_289, 12_289, 1),  # the 12288-entry geometry: exactly, one beyond
                                        (16_352, 16_384, 1), (16_353, 16_385, 1),  # the 16352-entry table / 16-slot tasks (kernel 2; 17920 for kernel 8)
                                        (17_920, 16_000, 1), (17_921, 18_000, 1),
                                        (6_144, 6_144, 1), (6_145, 6_145, 1),      # ... and its half-size form
                                        (1_100, 1_537, 1), (2_048, 2_049, 1), (3_000, 1_100, 1), (1_030, 5_000, 1),   # ... with rows left empty (the row guards)
                                        (200_000, 50_000, 4)])                     # build on S, pairs stay (rowR,rowS)
def test_pkfk_16_bit_plan_new_geometries(geom, oracle, nR, nS, nlow):
    pkfk_case(geom, oracle, nR, nS, nlow)


def test_new_geometries_duplicates_long_buckets_and_deep_plans(geom, oracle):
    test_duplicates_on_both_sides(geom, oracle)
    test_long_buckets_cooperative_scan(geom, oracle)
    test_17_and_18_bit_plans_narrow(geom, oracle, Opts(2, 8, 9))


@pytest.mark.parametrize("plan", [Opts(2, 8, 8), Opts(2, 9, 9), Opts(2, 9, 8), Opts(2, 8, 9), Opts(2, 10, 10)])
def test_generated_inputs_forced_big(big, oracle, plan):
    R, S = oracle.gen_R(400_000), oracle.gen_S_counter(700_000, 400_000, 7)
    check(big, oracle, R, S, plan)


@pytest.mark.parametrize("plan", [Opts(2, 8, 9), Opts(2, 9, 8), Opts(2, 9, 9)])
def test_17_and_18_bit_plans_narrow(big, oracle, plan):
    """plans beyond 16 bits (what 1.1 - 4.4 * 10^9 tuples per side get): two narrow passes with separate histograms, the 9-bit
    ones with 16-tuple carry lines; large partitions (few low-bit patterns), duplicates, unmatched probes, and the fall-back
    when a rowID does not fit 32 bits"""
    rng = np.random.default_rng(plan.bits2)
    tb = plan.bits1 + plan.bits2
    nR, nS, nlow = 70_000, 160_000, 3
    rv = rng.permutation(1 << 22)[:nR].astype(np.uint64)
    lows = np.random.default_rng(nlow).permutation(1 << tb)[:nlow].astype(np.uint64)
    R = rel(rng, nR, unmix64((rv << np.uint64(tb)) | lows[(rv % np.uint64(nlow)).astype(np.int64)]))      # (mixed value crafted, see few_partitions)
    S = rel(rng, nS, R["payload"][rng.integers(0, nR, nS)], key0=(1 << 32) - nS)
    S["payload"][::97] ^= np.uint64(1 << 50)
    check(big, oracle, R, S, plan)
    check(big, oracle, S, R, plan)
    R["key"][4_321] = np.uint64(1 << 33)
    check(big, oracle, R, S, plan, wide_rowids=True)


def test_duplicates_on_both_sides(big, oracle):
    """several matches per probe tuple: the generic (wavefront, slot) loop; output far larger than the inputs"""
    rng = np.random.default_rng(5)
    R = rel(rng, 60_000, few_partitions(rng.integers(0, 9_000, 60_000).astype(np.uint64), 2))
    S = rel(rng, 50_000, few_partitions(rng.integers(0, 9_000, 50_000).astype(np.uint64), 2), key0=1 << 31)
    check(big, oracle, R, S, Opts(2, 8, 8))


def test_long_buckets_cooperative_scan(big, oracle):
    """a few join values repeated thousands of times on the BUILD side among unique ones: the lanes that hit them
    face buckets far beyond BJ_HEAVY and are served by the whole wavefront"""
    rng = np.random.default_rng(11)
    nb = 30_000
    vals = (rng.permutation(1 << 20)[:nb].astype(np.uint64) + np.uint64(1000)) << np.uint64(16)
    vals[:2500] = 17 << 16                 # one hot value
    vals[2500:3100] = 18 << 16             # another
    B = rel(rng, nb, unmix64(vals))        # (the MIXED values share their low 16 bits: one partition, see few_partitions)
    pv = (rng.permutation(1 << 20)[:90_000].astype(np.uint64) + np.uint64(1000)) << np.uint64(16)
    pv[::9001] = 17 << 16                  # ten probe tuples hit the first hot value ...
    pv[5::30_011] = 18 << 16               # ... three the second
    P = rel(rng, 90_000, unmix64(pv), key0=(1 << 32) - 90_000)                         # rowIDs up to 2^32 - 1: still narrow
    check(big, oracle, B, P, Opts(2, 8, 8))   # B is the smaller side: build
    check(big, oracle, P, B, Opts(2, 8, 8))   # roles swapped: pairs are (rowR,rowS) either way
    P["key"][77_777] = 1 << 32                                                # one rowID beyond: detected on the device,
    check(big, oracle, B, P, Opts(2, 8, 8), wide_rowids=True)                 # the join repeats itself with 16-byte tuples
    check(big, oracle, P, B, Opts(2, 8, 8), wide_rowids=True)


def test_all_equal_keys(big):
    n, m = 20_000, 9_000
    dR, dS = big.alloc(16 * n), big.alloc(16 * m)
    big.generate(GEN_CONST, dR, n, 0, 99 << 16)
    big.generate(GEN_CONST, dS, m, 0, 99 << 16)
    assert big.join_dev(dR, n, dS, m, opts=Opts(2, 8, 8)) == n * m
    used_format(big, Opts(2, 8, 8))
    dO = big.alloc(16 * 500_000)
    assert big.join_dev(dR, n, dS, m, dO, 500_000, opts=Opts(2, 8, 8), allow_overflow=True) == n * m
    part = dO.to_numpy(PAIR, 500_000)
    assert part["keyR"].max() < n and part["keyS"].max() < m
    assert len(np.unique(part["keyR"] * np.uint64(m) + part["keyS"])) == 500_000


def test_full_width_rowids(big, oracle):
    rng = np.random.default_rng(3)
    n = 50_000
    vals = rng.permutation(np.unique(rng.integers(0, 1 << 30, 2 * n)))[:n].astype(np.uint64)     # n distinct values below 2^30
    R = rel(rng, n, few_partitions(vals, 2))
    R["key"] = rng.integers(0, 1 << 63, n, dtype=np.uint64) * np.uint64(2) + np.uint64(1)
    S = rel(rng, 80_000, R["payload"][rng.integers(0, n, 80_000)])
    S["key"] = rng.integers(0, 1 << 63, 80_000, dtype=np.uint64) * np.uint64(2)
    check(big, oracle, R, S, Opts(2, 8, 8), wide_rowids=True)
    S["key"] = np.arange(80_000, dtype=np.uint64)                            # only R's rowIDs are wide
    check(big, oracle, R, S, Opts(2, 8, 8), wide_rowids=True)


def test_under_partitioned_plans_use_the_chunked_kernel(big, oracle):
    """fewer than 16 radix bits: keys do not fit 48 bits, the engine must take the 16-byte-entry kernel whatever
    "join.big_kernel" says -- same pairs"""
    R, S = oracle.gen_R(300_000), oracle.gen_S_counter(200_000, 300_000, 3)
    for plan in (Opts(0), Opts(1, 4), Opts(2, 4, 4)):
        check(big, oracle, R, S, plan)


def test_skewed_probe_side(big):
    nR, nS = 400_000, 2_000_000
    dR, dS, dO = big.alloc(16 * nR), big.alloc(16 * nS), big.alloc(16 * nS)
    big.generate(GEN_R, dR, nR, 0, nR)
    big.generate(GEN_S_ZIPF, dS, nS, 0, nR, seed=3, theta_milli=900)
    exp_n, exp_c = big.expected_pkfk(dS, nS)
    for plan in (Opts(2, 8, 8), Opts(1, 3)):
        assert big.join_dev(dR, nR, dS, nS, dO, nS, opts=plan) == exp_n == nS
        assert big.pairs_checksum(dO, nS) == exp_c
        used_format(big, plan)


def test_bucket_join_stage_big_partitions(big, oracle):
    """the JoinJob stage alone on four 75 K x 50 K partitions made by the oracle's own partitioner"""
    R, S = oracle.gen_R(300_000, 100_000), oracle.gen_S_chain(200_000, 100_000)
    def part2(T):
        d = (T["payload"] & np.uint64(3)).astype(np.int64)
        o = np.argsort(d, kind="stable")
        return T[o], np.concatenate([[0], np.cumsum(np.bincount(d, minlength=4))]).astype(np.uint64)
    Rp, sR = part2(R)
    Sp, sS = part2(S)
    exp = oracle.join(R, S)
    dRp, dSp, dsR, dsS = big.to_device(Rp), big.to_device(Sp), big.to_device(sR), big.to_device(sS)
    n = big.bucket_join(dRp, dsR, dSp, dsS, 4, 2)
    assert n == len(exp)
    dO = big.alloc(16 * n)
    assert big.bucket_join(dRp, dsR, dSp, dsS, 4, 2, dO, n) == n
    assert np.array_equal(sorted_pairs(dO.to_numpy(PAIR, n)), sorted_pairs(exp))


@pytest.mark.parametrize("narrow", [1, 2])
@pytest.mark.parametrize("plan", [Opts(2, 5, 5), Opts(2, 8, 7), Opts(2, 3, 8), Opts(2, 8, 8), Opts(2, 4, 4)])
def test_narrow_format_with_the_one_table_join(engine, oracle, plan, narrow):
    """fused two-pass plans whose partitions fit one 16-byte-entry table (k_join_bkt reading the narrow arrays): the
    automatic choice from 8M tuples on, forced here on oracle-sized inputs; duplicates on both sides, probe tuples without
    a partner, rowIDs up to 2^32 - 1, and the fall-back on a larger one"""
    rng = np.random.default_rng(plan.bits1 * 16 + plan.bits2 + narrow)
    nR, nS = 150_000, 260_000
    R = rel(rng, nR, rng.integers(0, 100_000, nR).astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15))
    S = rel(rng, nS, R["payload"][rng.integers(0, nR, nS)], key0=(1 << 32) - nS)
    S["payload"][::53] ^= np.uint64(1 << 50)
    engine.set_option("partition.narrow", narrow)
    try:
        for wide in (False, True):
            if wide:
                R["key"][12_345] = np.uint64(1 << 32)
            got = engine.join(R, S, opts=plan)
            exp = oracle.join(R, S)
            assert len(got) == len(exp) and np.array_equal(sorted_pairs(got), sorted_pairs(exp))
            assert engine.info("last.narrow") == (0 if wide else narrow) and engine.info("last.join_kernel") == 0
            engine.set_option("partition.narrow", narrow)
    finally:
        engine.set_option("partition.narrow", -1)


@pytest.mark.parametrize("n,kind", [(1_500_000_000, GEN_S_UNIFORM), (1_500_000_000, GEN_S_ZIPF), (2_200_000_000, GEN_S_UNIFORM)])
def test_beyond_the_16_bit_plans_count_and_checksum(engine, n, kind):
    """1.5 and 2.2 * 10^9 tuples per side under the automatic plan (17 and 18 radix bits, narrow format): exact count and
    checksum of the pair set against the closed form"""
    from radixhashjoin_amd.binding import plan as resolve
    free, _ = engine.mem_info()
    if free < 16 * n * 6.2:
        pytest.skip("not enough free HBM")
    p = resolve(n, n)
    assert (p.passes, p.bits1, p.bits2) == (2, 8, 9)                          # 17 bits up to 2.2 * 10^9, 18 beyond
    dR, dS, dO = engine.alloc(16 * n), engine.alloc(16 * n), engine.alloc(16 * n)
    engine.generate(GEN_R, dR, n, 0, n)
    engine.generate(kind, dS, n, 0, n, seed=42, theta_milli=900)
    exp_n, exp_c = engine.expected_pkfk(dS, n)
    try:
        assert engine.join_dev(dR, n, dS, n, dO, n) == exp_n == n
        assert engine.pairs_checksum(dO, n) == exp_c
        assert engine.info("last.narrow") == 2
    finally:
        for b in (dR, dS, dO):
            b.free()
        engine.release_workspace()


@pytest.mark.parametrize("kind,plan,narrow", [(GEN_S_UNIFORM, Opts(2, 8, 8), -1), (GEN_S_ZIPF, Opts(2, 8, 8), -1),
                                              (GEN_S_ZIPF, Opts(), -1), (GEN_S_UNIFORM, Opts(2, 8, 8), 0),
                                              (GEN_S_UNIFORM, Opts(2, 8, 8), 1)])
def test_one_billion_count_and_checksum(engine, kind, plan, narrow):
    """BASELINE configs 3 and 4 at full size through rhj_join_dev with the engine's own kernel choice: 10^9 x 10^9,
    exact count and order-insensitive checksum of the pair set against the closed form (itself pinned to the oracle
    at 2-3 M tuples)"""
    n = 1_000_000_000
    free, _ = engine.mem_info()
    if free < 16 * n * 6.5:
        pytest.skip("not enough free HBM")
    dR, dS, dO = engine.alloc(16 * n), engine.alloc(16 * n), engine.alloc(16 * n)
    engine.generate(GEN_R, dR, n, 0, n)
    engine.generate(kind, dS, n, 0, n, seed=42, theta_milli=900)
    exp_n, exp_c = engine.expected_pkfk(dS, n)
    engine.set_option("partition.narrow", narrow)
    try:
        assert engine.join_dev(dR, n, dS, n, dO, n, opts=plan) == exp_n == n
        assert engine.pairs_checksum(dO, n) == exp_c
        assert engine.info("last.narrow") == (2 if narrow < 0 else narrow)      # rowIDs < 2^32: the default is narrow
    finally:
        engine.set_option("partition.narrow", -1)
    for b in (dR, dS, dO):
        b.free()
    engine.release_workspace()


def test_wide_rowid_fallback_is_per_join(oracle):
    """A rowID >= 2^32 makes ONE join repeat itself in the 16-byte format (the histogram kernel reports it before any
    scatter has run); the next join on the same context is narrow again -- nothing sticks to the context."""
    e = Engine(0)
    try:
        e.set_option("join.big_tables", 1)
        e.set_option("join.big_kernel", CT)
        e.set_option("partition.narrow", 2)       # (inputs this small are not narrow by default); set ONCE, never re-armed below
        rng = np.random.default_rng(3)
        rv = rng.permutation(1 << 22)[:40_000].astype(np.uint64)
        R = rel(rng, 40_000, few_partitions(rv, 2))
        S = rel(rng, 30_000, R["payload"][rng.integers(0, 40_000, 30_000)], key0=1 << 20)
        exp = sorted_pairs(oracle.join(R, S))
        for wide in (False, True, False, True, True, False):
            Sx = S.copy()
            if wide:
                Sx["key"][123] = (1 << 40) + 5
            got = e.join(R, Sx, opts=Opts(2, 8, 8))
            assert e.info("last.narrow") == (0 if wide else 2)
            assert np.array_equal(sorted_pairs(got), sorted_pairs(oracle.join(R, Sx)) if wide else exp)
    finally:
        e.close()


def test_bucket_join_checks_its_radix_bits_contract(engine):
    """rhj_bucket_join routes 16-bit plans with large partitions to the compact-table kernel, which compares only
    payload >> radix_bits: partitions whose payloads do NOT share those low bits are refused, not joined wrongly."""
    from radixhashjoin_amd.binding import RhjError
    n = 40_000
    rng = np.random.default_rng(9)
    t = np.empty(n, dtype=TUPLE)
    t["key"] = np.arange(n, dtype=np.uint64)
    t["payload"] = rng.integers(0, 1 << 40, n).astype(np.uint64)             # low 16 bits differ inside the one "partition"
    d = engine.to_device(t)
    st = engine.to_device(np.array([0, n], dtype=np.uint64))
    out = engine.alloc(16 * 4 * n)
    engine.set_option("join.big_tables", 1)
    try:
        with pytest.raises(RhjError, match="radix_bits"):
            engine.bucket_join(d, st, d, st, 1, 16, out, 4 * n)
        # with the true number of constant bits (none) the same call is a correct self-join on the payloads
        engine.set_option("join.big_tables", -1)
        cnt = engine.bucket_join(d, st, d, st, 1, 0, out, 4 * n)
        assert cnt >= n
    finally:
        engine.set_option("join.big_tables", -1)
    for x in (d, st, out):
        x.free()


@pytest.mark.parametrize("plan", [Opts(1, 8, 0, 2**31 - 1), Opts(2, 4, 4, 2**31 - 1), Opts(2, 8, 8, 1 << 30)])
def test_a_huge_probe_split_from_the_caller(engine, oracle, plan):
    """probe_split is the caller's: the kernels address a task's probe side through a 32-bit buffer descriptor, so values
    above 2^24 are clamped (include/rhj.h) -- same pairs"""
    R, S = oracle.gen_R(120_000), oracle.gen_S_counter(500_000, 120_000, 5)
    got = engine.join(R, S, opts=plan)
    exp = oracle.join(R, S)
    assert len(got) == len(exp) and np.array_equal(sorted_pairs(got), sorted_pairs(exp))


CT_Q12 = 11


@pytest.mark.parametrize("narrow", [0, 2])
@pytest.mark.parametrize("kind,plan", [(CT_G13, Opts(2, 7, 6)), (CT_G13, Opts(2, 7, 7)), (CT_G13, Opts(2, 8, 7)), (CT_G13, Opts(2, 8, 8)),
                                       (CT_Q12, Opts(2, 6, 6)), (CT_Q12, Opts(2, 7, 6)), (CT_Q12, Opts(2, 8, 8))])
def test_narrow_index_compact_tables_under_12_to_16_bit_plans(oracle, kind, plan, narrow):
    """k_join_ct<.., KB = 13> (6144 entries {key of up to 51 bits | 13-bit arrival index}: what plans of 13-15 radix bits take for
    partitions of 2-5 K tuples since round 4) and <.., KB = 12> (4096 entries, keys of up to 52 bits: plans of 12 bits).  Forced
    here onto few, large partitions (chunks of the table, several probe tasks per partition), exactly one table and one beyond,
    duplicates on both sides, unmatched probes; and chosen BY ITSELF -- plan and kernel -- for 40M and 80M tuples per side (7+6 and
    7+7 bits, partitions of 4.9 K tuples) and a 6+6-bit join of 3 K-tuple partitions."""
    tb = plan.bits1 + plan.bits2
    table = 6_144 if kind == CT_G13 else 4_096
    e = Engine(0)
    try:
        e.set_option("join.big_tables", 1)
        e.set_option("join.big_kernel", kind)
        e.set_option("partition.narrow", narrow)
        rng = np.random.default_rng(tb * 10 + narrow + kind)
        for nR, nS, nlow, dup in ((30_000, 50_000, 3, 1), (table, table, 1, 1), (table + 1, table + 6, 1, 1), (40_000, 25_000, 5, 4), (2_500, 9_000, 2, 1)):
            vals = rng.permutation(1 << 22)[:max(nR // dup, 1)].astype(np.uint64)
            rv = vals[rng.integers(0, len(vals), nR)] if dup > 1 else vals[:nR]
            lows = np.random.default_rng(nlow).permutation(1 << tb)[:nlow].astype(np.uint64)
            craft = lambda v: unmix64((v << np.uint64(tb)) | lows[(v % np.uint64(nlow)).astype(np.int64)])
            R = rel(rng, nR, craft(rv))
            sv = rv[rng.integers(0, nR, nS)]
            sv[::41] ^= np.uint64(1 << 21)                                 # some probes match another value or nothing
            S = rel(rng, nS, craft(sv), key0=1 << 31)
            got = e.join(R, S, opts=plan)
            exp = oracle.join(R, S)
            assert len(got) == len(exp) and np.array_equal(sorted_pairs(got), sorted_pairs(exp))
            assert e.info("last.join_kernel") == kind
        # the automatic choice at these sizes: the plan leaves partitions of up to 5120 tuples for the 6144-entry kernel (4.9 K here)
        auto = {(7, 6): 40_000_000, (7, 7): 80_000_000, (6, 6): 12_000_000}.get((plan.bits1, plan.bits2))
        if auto and ((kind == CT_G13 and tb in (13, 14)) or (kind == CT_Q12 and tb == 12)):
            e.set_option("join.big_tables", -1)
            e.set_option("join.big_kernel", -1)
            e.set_option("partition.narrow", -1)
            n = auto
            dR, dS, dO = e.alloc(16 * n), e.alloc(16 * n), e.alloc(16 * n)
            e.generate(GEN_R, dR, n, 0, n)
            e.generate(GEN_S_UNIFORM, dS, n, 0, n, seed=5)
            exp_n, exp_c = e.expected_pkfk(dS, n)
            assert e.join_dev(dR, n, dS, n, dO, n) == exp_n
            assert e.pairs_checksum(dO, n) == exp_c
            t = e.timings()
            assert (t["passes"], t["bits1"] + t["bits2"]) == (2, tb) and e.info("last.join_kernel") == kind
    finally:
        e.close()
