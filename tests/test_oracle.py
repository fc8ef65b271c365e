"""CPU suite: pins the oracle (oracle/rhj_oracle.c) against the reference's golden data.
   (1) known answers produced by the real reference (tests/golden/synthetic.json, tiny_vectors.npz)
   (2) the 94 multiRadixHashJoin calls of small.work (tests/golden/small_joins.*)
   (3) the reference itself, when oracle/_ref is built (page-order byte equality)."""
import numpy as np
import pytest

from conftest import make_inputs, small_call_arrays
from oracle.pyoracle import PAIR, TUPLE, sorted_pairs

CPU_CASES = ["pkfk_1k", "pkfk_1m", "pkfk_4m_x_1m", "pkfk_300k_x_3m", "dup_10k", "dup_100k_50k", "dup_2m_d64k",
             "alleq_300_500", "alleq_7000_9", "tinyR_1_1000", "tinyS_1000_1", "tiny_5_3", "lt_ranges_7_7", "one_one",
             "disjoint_1k", "disjoint_200k", "chunk_edge_6144", "chunk_edge_6145", "small_build_big_probe"]


def test_next_prime_known_values(oracle):
    # SURVEY.md App. A [measured on the reference]; auxFun.cpp:4-22
    known = {0: 2, 1: 2, 2: 5, 3: 5, 4: 5, 5: 7, 6: 7, 7: 11, 8: 11, 23: 29, 24: 29, 3906: 3907, 3907: 3911, 15259: 15263}
    for x, p in known.items():
        assert oracle.next_prime(x) == p


def test_next_prime_vs_reference(oracle, reference):
    for x in list(range(0, 2000)) + [4095, 4096, 65535, 65536, 10**6, 10**6 + 3]:
        assert oracle.next_prime(x) == reference.next_prime(x)


def test_mix_known_value(oracle):
    # splitmix64 with seed 0: first output
    assert oracle.mix(0) == 0xE220A8397B1DCDAF


@pytest.mark.parametrize("name", CPU_CASES)
def test_synthetic_golden(oracle, synthetic_golden, name):
    g = synthetic_golden[name]
    R, S = make_inputs(oracle, g["spec"])
    cnt, chk = oracle.join_count_checksum(R, S)
    assert cnt == g["count"]
    assert f"{chk:016x}" == g["checksum"]
    assert (cnt == 0) == g["head_null"]


def test_synthetic_golden_16m(oracle, synthetic_golden):
    g = synthetic_golden["pkfk_16m"]
    R, S = make_inputs(oracle, g["spec"])
    cnt, chk = oracle.join_count_checksum(R, S)
    assert (cnt, f"{chk:016x}") == (g["count"], g["checksum"])


def test_tiny_vectors_page_order(oracle, tiny_vectors):
    names = sorted({k.split("__")[0] for k in tiny_vectors.files})
    assert len(names) >= 8
    for n in names:
        R, S, P = tiny_vectors[n + "__R"], tiny_vectors[n + "__S"], tiny_vectors[n + "__pairs"]
        got = oracle.join(R.astype(TUPLE), S.astype(TUPLE))
        assert np.array_equal(got, P.astype(PAIR)), n      # same pairs in the reference's own page order


def test_small_work_calls(oracle, small_joins):
    meta, npz = small_joins
    assert len(meta) == 94
    assert sum(c["count"] for c in meta) == 2_171_642
    assert sum(c["nR"] for c in meta) == 480_656 and sum(c["nS"] for c in meta) == 1_194_737
    full = [i for i, c in enumerate(meta) if c.get("vectors")]
    assert len(full) >= 10
    for i in full:
        R, S, P = small_call_arrays(npz, i)
        assert f"{oracle.pairs_checksum(R.view(PAIR)):016x}" == meta[i]["checksum_R"]
        assert f"{oracle.pairs_checksum(S.view(PAIR)):016x}" == meta[i]["checksum_S"]
        got = oracle.join(R, S)
        assert len(got) == meta[i]["count"]
        assert np.array_equal(got, P)
        assert f"{oracle.pairs_checksum(got):016x}" == meta[i]["checksum"]


@pytest.mark.parametrize("n", [0, 1, 5, 7, 8, 9, 1000, 100003])
def test_hash_relation_is_stable_partition(oracle, n):
    # structs.cpp:144-204 must equal the dead-code serial spec structs.cpp:86-109
    R = oracle.gen_R(n, max(n // 3, 1))
    a, ha = oracle.hash_relation(R)
    b, hb = oracle.single_partition(R)
    assert np.array_equal(a, b) and np.array_equal(ha, hb)
    assert int(ha.sum()) == n
    if n:
        d = (a["payload"] & 255).astype(np.int64)
        assert np.all(np.diff(d) >= 0)
        for v in np.unique(d)[:8]:                               # stability: rowIDs ascending inside a bucket
            assert np.all(np.diff(a["key"][d == v].astype(np.int64)) > 0)


@pytest.mark.parametrize("n", [0, 1, 7, 8, 9, 1000, 100003])
def test_hash_relation_vs_reference(oracle, reference, n):
    R = oracle.gen_R(n, max(n // 3, 1))
    a, ha = oracle.hash_relation(R)
    b, hb = reference.hash_relation(R)
    c, hc = reference.single_partition(R)
    assert np.array_equal(a, b) and np.array_equal(ha, hb) and np.array_equal(a, c) and np.array_equal(ha, hc)


def test_random_joins_vs_reference(oracle, reference):
    rng = np.random.default_rng(7)
    for trial in range(40):
        nR, nS = int(rng.integers(0, 3000)), int(rng.integers(0, 3000))
        dom = int(rng.choice([1, 3, 50, 1000, 1 << 20, 1 << 62]))
        R = np.empty(nR, dtype=TUPLE); S = np.empty(nS, dtype=TUPLE)
        R["key"] = rng.permutation(nR); S["key"] = rng.permutation(nS)
        R["payload"] = rng.integers(0, dom, nR, dtype=np.uint64); S["payload"] = rng.integers(0, dom, nS, dtype=np.uint64)
        po = oracle.join(R, S)
        pr, cnt, head_null, _ = reference.join(R, S)
        assert np.array_equal(po, pr)
        assert head_null == (cnt == 0)
        # brute force on the small ones: the pair multiset is what equality of join values gives
        if nR * nS <= 400_000:
            rr, ss = np.nonzero(R["payload"][:, None] == S["payload"][None, :])
            bf = np.empty(len(rr), dtype=PAIR); bf["keyR"] = R["key"][rr]; bf["keyS"] = S["key"][ss]
            assert np.array_equal(sorted_pairs(bf), sorted_pairs(po))


def test_counter_generator_is_pkfk(oracle):
    # the device generator's uniform FK stream (gen v2) restated on the CPU: every S tuple matches one R row
    n = 50_000
    R, S = oracle.gen_R(n), oracle.gen_S_counter(70_000, n, 42)
    cnt, _ = oracle.join_count_checksum(R, S)
    assert cnt == 70_000
