#!/usr/bin/env python3
"""tools/verify_big.py -- large asymmetric / skewed joins through rhj_join_dev, each checked by (count, checksum) against the
closed form, in the narrow and in the 16-byte format (development aid; run on the GPU box)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import radixhashjoin_amd as rhj  # noqa: E402
from radixhashjoin_amd.binding import GEN_R, GEN_S_UNIFORM, GEN_S_ZIPF  # noqa: E402

e = rhj.Engine(0)
CASES = [(300_000_000, 1_000_000_000, GEN_S_ZIPF, 900), (1_000_000_000, 200_000_000, GEN_S_UNIFORM, 0),
         (700_000_000, 700_000_000, GEN_S_ZIPF, 1250), (50_000_000, 900_000_000, GEN_S_ZIPF, 1100),
         (400_000_000, 400_000_000, GEN_S_UNIFORM, 0), (10_000_000, 10_000_000, GEN_S_ZIPF, 900)]
bad = 0
for nR, nS, kind, theta in CASES:
    dR, dS = e.alloc(16 * nR), e.alloc(16 * nS)
    e.generate(GEN_R, dR, nR, 0, nR)
    e.generate(kind, dS, nS, 0, nR, seed=nR % 1000 + 1, theta_milli=theta or 900)
    exp = e.expected_pkfk(dS, nS)
    dO = e.alloc(16 * (exp[0] + 1024))
    for narrow in (-1, 0):
        e.set_option("partition.narrow", narrow)
        e.join_dev(dR, nR, dS, nS, dO, exp[0] + 1024)
        t0 = time.perf_counter()
        cnt = e.join_dev(dR, nR, dS, nS, dO, exp[0] + 1024)
        ms = (time.perf_counter() - t0) * 1e3
        ok = (cnt, e.pairs_checksum(dO, cnt)) == exp
        bad += 0 if ok else 1
        t = e.timings()
        print(json.dumps({"nR": nR, "nS": nS, "zipf_theta_milli": theta, "narrow_requested": narrow, "narrow_used": e.info("last.narrow"),
                          "join_kernel": e.info("last.join_kernel"), "plan": [t["passes"], t["bits1"], t["bits2"]], "ok": ok,
                          "ms": round(ms, 2), "Mtuples_per_s": round((nR + nS) / ms / 1e3)}), flush=True)
    for b in (dR, dS, dO):
        b.free()
    e.release_workspace()
sys.exit(1 if bad else 0)
