#!/usr/bin/env python3
"""tools/build_side_ab.py -- which side of a partition becomes the hash table, and what it costs to get that wrong.
A key / foreign-key join in both argument orders (R JOIN S and S JOIN R: the same pairs with the columns swapped), timed as
tools/size_sweep.py does.  RHJ_SNIFF=0 switches the duplicate sampling off (then the first argument wins every near tie),
RHJ_BUILD_TIE=63 is the reference's rule (JobScheduler.cpp:187: the smaller bucket, S on a tie).
    python tools/build_side_ab.py [--sizes 1000000,80000000,1000000000] [--reps 5]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import radixhashjoin_amd as rhj
from radixhashjoin_amd.binding import GEN_R, GEN_S_UNIFORM

ap = argparse.ArgumentParser()
ap.add_argument("--sizes", default="1000000,80000000,1000000000")
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
sizes = [int(x) for x in a.sizes.split(",")]
e = rhj.Engine(0)
cap = max(sizes)
dR, dS, dO = e.alloc(16 * cap), e.alloc(16 * cap), e.alloc(16 * (cap + 1024))
for n in sizes:
    e.generate(GEN_R, dR, n, 0, n)
    e.generate(GEN_S_UNIFORM, dS, n, 0, n, seed=42)
    exp_n, _ = e.expected_pkfk(dS, n)
    for order, (A, B) in (("key side first", (dR, dS)), ("foreign-key side first", (dS, dR))):
        e.set_profiling(False)
        cnt = e.join_dev(A, n, B, n, dO, n + 1024)
        wall = []
        for _ in range(a.reps):
            t0 = time.perf_counter()
            e.join_dev(A, n, B, n, dO, n + 1024)
            wall.append((time.perf_counter() - t0) * 1e3)
        e.set_profiling(True)
        e.join_dev(A, n, B, n, dO, n + 1024)
        t = e.timings()
        w = sorted(wall)[len(wall) // 2]
        print(json.dumps({"n": n, "order": order, "count_ok": cnt == exp_n, "plan": [t["passes"], t["bits1"], t["bits2"]],
                          "join_kernel_ms": round(t["join"]["ms"], 3), "hist_ms": round(t["hist"]["ms"], 3), "wall_ms": round(w, 3),
                          "Mtuples/s": round(2 * n / w / 1e3), "sniff": os.environ.get("RHJ_SNIFF", "1"),
                          "tie": os.environ.get("RHJ_BUILD_TIE", "4")}), flush=True)
