#!/usr/bin/env python3
"""tools/size_sweep.py -- throughput of rhj_join_dev under the automatic plan across input sizes, one process (development
aid: looks for cliffs between the plans and join-kernel geometries).  nR = nS = n, PK/FK uniform (or --dist zipf).
   python tools/size_sweep.py [--lo 500000] [--hi 1000000000] [--step 1.4] [--reps 3]
One JSON line per size: plan, join kernel, total kernel ms (HIP events around every launch), wall ms, tuples/s (wall)."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import radixhashjoin_amd as rhj  # noqa: E402
from radixhashjoin_amd.binding import GEN_R, GEN_S_UNIFORM, GEN_S_ZIPF  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--lo", type=int, default=500_000)
ap.add_argument("--hi", type=int, default=1_000_000_000)
ap.add_argument("--step", type=float, default=1.4)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--dist", default="uniform")
ap.add_argument("--sizes", default="")
ap.add_argument("--bits", default="", help="explicit two-pass plan, e.g. 8,8")
ap.add_argument("--big", type=int, default=-1, help="force join.big_tables=1 and this join.big_kernel")
a = ap.parse_args()
sizes = [int(x) for x in a.sizes.split(",")] if a.sizes else []
n = a.lo
while not a.sizes and n <= a.hi:
    sizes.append(int(n))
    n *= a.step
e = rhj.Engine(0)
if a.big >= 0:
    e.set_option("join.big_tables", 1)
    e.set_option("join.big_kernel", a.big)
opts = rhj.Opts(2, *[int(x) for x in a.bits.split(",")]) if a.bits else None
cap = max(sizes)
dR, dS, dO = e.alloc(16 * cap), e.alloc(16 * cap), e.alloc(16 * (cap + 1024))
for n in sizes:
    e.generate(GEN_R, dR, n, 0, n)
    e.generate(GEN_S_ZIPF if a.dist == "zipf" else GEN_S_UNIFORM, dS, n, 0, n, seed=42, theta_milli=900)
    exp = e.expected_pkfk(dS, n)
    e.set_profiling(False)
    cnt = e.join_dev(dR, n, dS, n, dO, n + 1024, opts=opts)
    ok = (cnt, e.pairs_checksum(dO, cnt)) == exp
    e.sync()
    wall = []
    for _ in range(a.reps):
        t0 = time.perf_counter()
        e.join_dev(dR, n, dS, n, dO, n + 1024, opts=opts)
        wall.append((time.perf_counter() - t0) * 1e3)
    e.set_profiling(True)
    e.join_dev(dR, n, dS, n, dO, n + 1024, opts=opts)
    t = e.timings()
    kern = {k: round(t[k]["ms"], 3) for k in ("hist", "scatter", "join") if t[k]["launches"]}
    w = sorted(wall)[len(wall) // 2]
    print(json.dumps({"n": n, "ok": ok, "plan": [t["passes"], t["bits1"], t["bits2"]], "join_kernel": e.info("last.join_kernel"),
                      "narrow": e.info("last.narrow"), "ntasks": t["ntasks"], "kernel_ms": kern, "kernels_total_ms": round(t["total_ms"], 3),
                      "wall_ms": round(w, 3), "Mtuples/s": round(2 * n / w / 1e3)}), flush=True)
