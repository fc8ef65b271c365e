#!/usr/bin/env python3
"""Generate tests/golden/*.json|npz from the REAL reference.

Run in the build container (needs /root/reference):
    make -C oracle all _ref/join_tap
    (cd /root/reference && cat small/small.init small/small.work | RHJ_TAP_OUT=/tmp/tap.bin \
        /root/repo/oracle/_ref/join_tap > /tmp/tap.result && cmp /tmp/tap.result small/small.result)
    python tests/golden/make_golden.py --tap /tmp/tap.bin

Outputs (data only: inputs and expected outputs):
  synthetic.json     (case spec) -> (count, checksum, head_null) from oracle/_ref/libref_rhj.so,
                     i.e. Result::multiRadixHashJoin of the reference itself, 8 threads
  tiny_vectors.npz   complete input/output arrays of a few tiny joins (reference page order)
  small_joins.json   the 94 multiRadixHashJoin calls the reference makes while running
                     small/small.work (link-time tap, oracle/ref_tap.cpp): sizes, match count,
                     checksums of R, S and of the pair set
  small_joins.npz    complete inputs/outputs for a size-bounded subset of those calls
"""
import argparse
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle.pyoracle import PAIR, TUPLE, Oracle, Reference  # noqa: E402

# (name, generator spec). Generators are oracle.pyoracle.Oracle.gen_* (SURVEY.md App. A):
#   R: gen_R(nR, D);  S: "chain" gen_S_chain(nS, D) | "disjoint" | "const"
SYNTH = [
    ("pkfk_1k", dict(nR=1000, nS=1000, D=1000, S="chain")),
    ("pkfk_1m", dict(nR=1_000_000, nS=1_000_000, D=1_000_000, S="chain")),
    ("pkfk_16m", dict(nR=16_000_000, nS=16_000_000, D=16_000_000, S="chain")),
    ("pkfk_4m_x_1m", dict(nR=4_000_000, nS=1_000_000, D=4_000_000, S="chain")),
    ("pkfk_300k_x_3m", dict(nR=300_000, nS=3_000_000, D=300_000, S="chain")),
    ("dup_10k", dict(nR=10_000, nS=10_000, D=100, S="chain")),
    ("dup_100k_50k", dict(nR=100_000, nS=50_000, D=1000, S="chain")),
    ("dup_2m_d64k", dict(nR=2_000_000, nS=500_000, D=65_536, S="chain")),
    ("alleq_300_500", dict(nR=300, nS=500, value=7, S="const")),
    ("alleq_7000_9", dict(nR=7000, nS=9, value=123456789, S="const")),
    ("tinyR_1_1000", dict(nR=1, nS=1000, D=1, S="chain")),
    ("tinyS_1000_1", dict(nR=1000, nS=1, D=1000, S="chain")),
    ("tiny_5_3", dict(nR=5, nS=3, D=4, S="chain")),
    ("lt_ranges_7_7", dict(nR=7, nS=7, D=7, S="chain")),
    ("one_one", dict(nR=1, nS=1, D=1, S="chain")),
    ("disjoint_1k", dict(nR=1000, nS=1000, D=1000, S="disjoint")),
    ("disjoint_200k", dict(nR=200_000, nS=150_000, D=200_000, S="disjoint")),
    ("chunk_edge_6144", dict(nR=6144, nS=20_000, D=6144, S="chain")),
    ("chunk_edge_6145", dict(nR=6145, nS=20_000, D=6145, S="chain")),
    ("small_build_big_probe", dict(nR=3000, nS=2_000_000, D=3000, S="chain")),
]
# known answers the surveyor recorded from the real reference at sizes too slow to regenerate
# in every run; same generator (S="chain", D=n)
SURVEY_ONLY = [
    ("pkfk_64m", dict(nR=64_000_000, nS=64_000_000, D=64_000_000, S="chain"), 64_000_000, 0x52B743FBE77A5984),
    ("pkfk_128m", dict(nR=128_000_000, nS=128_000_000, D=128_000_000, S="chain"), 128_000_000, 0xD8E7AE6564CB115F),
]


def make_inputs(o, spec):
    if spec["S"] == "const":
        return o.gen_const(spec["nR"], spec["value"]), o.gen_const(spec["nS"], spec["value"])
    R = o.gen_R(spec["nR"], spec["D"])
    if spec["S"] == "chain":
        S = o.gen_S_chain(spec["nS"], spec["D"])
    elif spec["S"] == "disjoint":
        S = o.gen_S_disjoint(spec["nS"], spec["D"])
    else:
        raise ValueError(spec["S"])
    return R, S


def tuples_checksum(o, t):
    return o.pairs_checksum(t.view(PAIR))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tap", default=None, help="dump written by oracle/_ref/join_tap (RHJ_TAP_OUT)")
    ap.add_argument("--npz-budget-mb", type=float, default=4.0)
    args = ap.parse_args()
    o, r = Oracle(), Reference()

    synth = {}
    for name, spec in SYNTH:
        R, S = make_inputs(o, spec)
        pairs, cnt, head_null, sec = r.join(R, S)
        synth[name] = dict(spec=spec, count=int(cnt), checksum=f"{o.pairs_checksum(pairs):016x}",
                           head_null=bool(head_null), source="oracle/_ref/libref_rhj.so")
        print(name, synth[name]["count"], synth[name]["checksum"], f"{sec:.3f}s")
    for name, spec, cnt, chk in SURVEY_ONLY:
        synth[name] = dict(spec=spec, count=cnt, checksum=f"{chk:016x}", head_null=False, source="SURVEY.md App. A")
    with open(os.path.join(HERE, "synthetic.json"), "w") as f:
        json.dump(synth, f, indent=1)

    tiny = {}
    for name in ("tiny_5_3", "lt_ranges_7_7", "one_one", "tinyR_1_1000", "alleq_300_500"):
        spec = dict(SYNTH)[name]
        R, S = make_inputs(o, spec)
        pairs, cnt, head_null, _ = r.join(R, S)
        tiny[name + "__R"], tiny[name + "__S"], tiny[name + "__pairs"] = R, S, pairs
    rng = np.random.default_rng(20181)
    for i, (nR, nS, dom) in enumerate([(50, 40, 8), (257, 300, 1 << 40), (1000, 33, 100)]):
        R = np.empty(nR, dtype=TUPLE); S = np.empty(nS, dtype=TUPLE)
        R["key"] = rng.permutation(nR); R["payload"] = rng.integers(0, dom, nR, dtype=np.uint64)
        S["key"] = rng.permutation(nS) + 1000; S["payload"] = rng.integers(0, dom, nS, dtype=np.uint64)
        pairs, cnt, _, _ = r.join(R, S)
        tiny[f"rand{i}__R"], tiny[f"rand{i}__S"], tiny[f"rand{i}__pairs"] = R, S, pairs
    np.savez_compressed(os.path.join(HERE, "tiny_vectors.npz"), **tiny)

    if args.tap:
        raw = np.fromfile(args.tap, dtype=np.uint64)
        pos, calls = 0, []
        while pos < len(raw):
            nR, nS, cnt, hn = (int(x) for x in raw[pos:pos + 4]); pos += 4
            R = raw[pos:pos + 2 * nR].view(TUPLE); pos += 2 * nR
            S = raw[pos:pos + 2 * nS].view(TUPLE); pos += 2 * nS
            P = raw[pos:pos + 2 * cnt].view(PAIR); pos += 2 * cnt
            calls.append((nR, nS, cnt, hn, R, S, P))
        calls.sort(key=lambda c: (c[0], c[1], c[2], tuples_checksum(o, c[4])))
        meta, full, budget = [], {}, args.npz_budget_mb * 1e6
        for i, (nR, nS, cnt, hn, R, S, P) in enumerate(calls):
            # the oracle restatement must agree with the reference on the reference's own workload
            po = o.join(R, S)
            assert np.array_equal(po, P), f"oracle != reference on small.work call {i}"
            meta.append(dict(nR=nR, nS=nS, count=cnt, head_null=bool(hn), checksum=f"{o.pairs_checksum(P):016x}",
                             checksum_R=f"{tuples_checksum(o, R):016x}", checksum_S=f"{tuples_checksum(o, S):016x}"))
        # full vectors: smallest-first within budget, but always the largest-output call
        order = sorted(range(len(calls)), key=lambda i: calls[i][0] + calls[i][1] + calls[i][2])
        biggest = max(range(len(calls)), key=lambda i: calls[i][2])
        chosen = []
        for i in [biggest] + order[::7]:
            if i in chosen:
                continue
            nR, nS, cnt = calls[i][:3]
            cost = (nR + nS) * 5 + cnt * 5          # rough compressed size with u32 columns
            if i != biggest:
                if cost > budget:
                    continue
                budget -= cost
            chosen.append(i)
        for i in sorted(chosen):
            nR, nS, cnt, hn, R, S, P = calls[i]
            # values in small/ fit in 32 bits: store columns as u32 to keep the fixture small
            for tag, arr, cols in (("R", R, ("key", "payload")), ("S", S, ("key", "payload")), ("P", P, ("keyR", "keyS"))):
                for c in cols:
                    assert arr[c].max(initial=0) < 2 ** 32
                    full[f"call{i}__{tag}_{c}"] = arr[c].astype(np.uint32)
            meta[i]["vectors"] = True
        with open(os.path.join(HERE, "small_joins.json"), "w") as f:
            json.dump(dict(source="oracle/_ref/join_tap over reference small/small.init + small/small.work "
                                  "(stdout byte-identical to small/small.result)", calls=meta), f, indent=1)
        np.savez_compressed(os.path.join(HERE, "small_joins.npz"), **full)
        print(len(calls), "calls,", len(chosen), "with full vectors; sum matches", sum(c[2] for c in calls))


if __name__ == "__main__":
    main()
