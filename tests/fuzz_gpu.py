#!/usr/bin/env python3
"""tests/fuzz_gpu.py -- randomized differential test of the HIP path against the CPU oracle (run by hand on the GPU box).
Random sizes (incl. tile/unit/table boundaries), value domains, duplicate structure, skew, radix plans and
probe splits; compares (count, checksum) and, for small outputs, the sorted pair sets.  python tests/fuzz_gpu.py [seconds] [seed]
RHJ_FUZZ_BIG=1..11 forces the oversized-partition kernels (1: chunked 16-byte entries, 2 / 3: compact table at full / half
size where the plan allows, 4 / 5 with 20 probe slots, 6 / 7 the 12288- / 6144-entry geometries); with RHJ_FUZZ_NARROW=1 half of
the cases run a 16-, 17- or 18-bit plan that stores its partitions in the narrow {payload, rowID} format (rowIDs start at 10^7 on S; every 16th case has one rowID >= 2^32, which must
send the join back to 16-byte tuples)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import radixhashjoin_amd as rhj
from oracle.pyoracle import Oracle, TUPLE, sorted_pairs

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 12345
rng = np.random.default_rng(seed)
o, e = Oracle(), rhj.Engine(0)
if os.environ.get("RHJ_FUZZ_BIG"):
    e.set_option("join.big_tables", 1)
    e.set_option("join.big_kernel", int(os.environ["RHJ_FUZZ_BIG"]))
EDGE = [1, 2, 7, 8, 9, 63, 64, 65, 2047, 2048, 2049, 4095, 4096, 4097, 4223, 4224, 4225, 8447, 8448, 8449, 32767, 32768, 32769]


def size():
    r = rng.random()
    if r < 0.35:
        return int(rng.choice(EDGE))
    if r < 0.8:
        return int(rng.integers(1, 200_000))
    return int(rng.integers(200_000, 3_000_000))


def rel(n, dom, skew, key0):
    t = np.empty(n, dtype=TUPLE)
    t["key"] = rng.permutation(n).astype(np.uint64) + np.uint64(key0)
    if skew == 0:
        v = rng.integers(0, dom, n, dtype=np.uint64)
    elif skew == 1:      # power-law ranks
        v = (rng.pareto(1.1, n) * 3).astype(np.uint64) % np.uint64(dom)
    else:                # one hot value + uniform rest
        v = rng.integers(0, dom, n, dtype=np.uint64)
        v[rng.random(n) < 0.5] = np.uint64(dom // 2)
    t["payload"] = v
    return t


def shape(R, S):
    """the same re-labelling of the join values on both sides: spread over 64 bits, aligned (multiples of 2^k: what raw-bit radix
    digits choke on), or values whose MIXED form shares its low 16 bits (few, large partitions after rhj_mix64)"""
    mode = int(rng.integers(0, 4))
    for t in (R, S):
        v = t["payload"]
        if mode == 1:
            v = v * np.uint64(0x9E3779B97F4A7C15)
        elif mode == 2:
            v = v << np.uint64(shape.k)
        elif mode == 3:
            v = rhj.unmix64((v << np.uint64(16)) | np.uint64(0xBEEF))
        t["payload"] = v


t0, cases, maxout, narrow_runs, fallbacks = time.time(), 0, 0, 0, 0
while time.time() - t0 < budget:
    nR, nS = size(), size()
    dom = int(rng.choice([1, 2, 5, 100, 4096, 70_000, 1 << 22, 1 << 40]))
    skR, skS = int(rng.integers(0, 3)), int(rng.integers(0, 3))
    R, S = rel(nR, dom, skR, 0), rel(nS, dom, skS, 10**7)
    shape.k = int(rng.integers(0, 40))
    shape(R, S)
    e.set_option("partition.mix", int(rng.random() < 0.85))                 # mostly the default; raw digits now and then
    e.set_option("join.fused", int(rng.random() < 0.8))
    # expected output size first (value counts), so that neither the oracle nor the GPU materialises a huge result
    vr, cr = np.unique(R["payload"], return_counts=True)
    vs, cs = np.unique(S["payload"], return_counts=True)
    common, ir, is_ = np.intersect1d(vr, vs, assume_unique=True, return_indices=True)
    est = int((cr[ir].astype(np.float64) * cs[is_].astype(np.float64)).sum())
    if est > 20_000_000:
        continue
    exp_n, exp_c = o.join_count_checksum(R, S)
    assert exp_n == est
    r = rng.random()
    wide = False
    if os.environ.get("RHJ_FUZZ_NARROW") and rng.random() < 0.5:
        r = 2.0
        e.set_option("partition.narrow", int(rng.integers(1, 3)))
        if cases % 16 == 7:
            (R if rng.random() < 0.5 else S)["key"][int(rng.integers(0, min(nR, nS)))] = np.uint64(1 << 32) + np.uint64(cases)
            wide = True
            exp_n, exp_c = o.join_count_checksum(R, S)
    if r == 2.0:                                                             # 16 bits fused, or a 17-18-bit plan (narrow level 2 only)
        b1, b2 = [(8, 8), (8, 8), (8, 9), (9, 8), (9, 9)][int(rng.integers(0, 5))]
        if b1 + b2 > 16:
            e.set_option("partition.narrow", 2)
        opts = rhj.Opts(2, b1, b2, int(rng.choice([0, 4096, 32768])))
    elif r < 0.4:
        opts = None
    elif r < 0.55:
        opts = rhj.Opts(0, 0, 0, int(rng.choice([0, 4096, 8192])))
    elif r < 0.75:
        opts = rhj.Opts(1, int(rng.integers(1, 11)), 0, int(rng.choice([0, 4096])))
    elif r < 0.9 or not os.environ.get("RHJ_FUZZ_BIG"):
        opts = rhj.Opts(2, int(rng.integers(1, 11)), int(rng.integers(1, 11)), int(rng.choice([0, 4096, 32768])))
    else:                                                                    # >= 16 bits: what the compact-table kernel serves
        opts = rhj.Opts(2, int(rng.integers(8, 11)), int(rng.integers(8, 11)), int(rng.choice([0, 4096, 32768])))
    got = e.join(R, S, opts=opts)
    narrow_runs += 1 if e.info("last.narrow") else 0
    fallbacks += 1 if wide else 0
    ok = (len(got), o.pairs_checksum(got)) == (exp_n, exp_c) and not (wide and e.info("last.narrow"))
    if ok and exp_n <= 300_000:
        ok = np.array_equal(sorted_pairs(got), sorted_pairs(o.join(R, S)))
    if not ok:
        print("MISMATCH", dict(nR=nR, nS=nS, dom=dom, skR=skR, skS=skS, opts=opts, got=len(got), exp=exp_n, seed=seed, case=cases), flush=True)
        sys.exit(1)
    cases += 1
    maxout = max(maxout, exp_n)
print(f"fuzz ok: {cases} random joins in {time.time() - t0:.0f} s, largest output {maxout} pairs, seed {seed}; "
      f"{narrow_runs} ran in the narrow format, {fallbacks} had a rowID >= 2^32")
