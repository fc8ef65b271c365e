#!/usr/bin/env python3
"""tools/kbench.py -- per-kernel micro-benchmark on the GPU box (development aid, not the headline bench).
   python tools/kbench.py [--n 256000000] [--bits1 8] [--bits2 8] [--reps 3] [--dist uniform|zipf]
Prints per-launch ms and algorithmic GB/s of the histogram, scatter and join kernels."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import radixhashjoin_amd as rhj  # noqa: E402
from radixhashjoin_amd.binding import GEN_R, GEN_S_UNIFORM, GEN_S_ZIPF  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=256_000_000)
ap.add_argument("--bits1", type=int, default=8)
ap.add_argument("--bits2", type=int, default=8)
ap.add_argument("--passes", type=int, default=2)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--dist", default="uniform")
ap.add_argument("--label", default="")
ap.add_argument("--probe-split", type=int, default=0)
ap.add_argument("--no-join", action="store_true", help="partition stage only (safe for ablations that corrupt outputs)")
ap.add_argument("--join-only", action="store_true", help="partition once, then time rhj_bucket_join alone")
ap.add_argument("--auto", action="store_true", help="the engine's automatic radix plan instead of --passes/--bits")
ap.add_argument("--big", type=int, default=-1, help="rhj_set_option join.big_kernel (1: chunked 16 B entries, 2: compact table)")
a = ap.parse_args()

e = rhj.Engine(0)
n = a.n
dR, dS, dO = e.alloc(16 * n), e.alloc(16 * n), e.alloc(16 * n)
e.generate(GEN_R, dR, n, 0, n)
e.generate(GEN_S_ZIPF if a.dist == "zipf" else GEN_S_UNIFORM, dS, n, 0, n, seed=42, theta_milli=900)
exp = e.expected_pkfk(dS, n)
opts = None if a.auto else rhj.Opts(a.passes, a.bits1, a.bits2 if a.passes == 2 else 0, a.probe_split)
if a.big >= 0:
    e.set_option("join.big_kernel", a.big)
if a.no_join:
    dP = e.alloc(8 * ((1 << (a.bits1 + (a.bits2 if a.passes == 2 else 0))) + 1))
    e.partition(dR, n, a.bits1, a.bits2 if a.passes == 2 else 0, dO, dP)
    e.set_profiling(True)
    acc = {}
    for _ in range(a.reps):
        e.partition(dR, n, a.bits1, a.bits2 if a.passes == 2 else 0, dO, dP)
        t = e.timings()
        for k in ("hist", "scan", "scatter"):
            acc.setdefault(k, [0.0, 0]); acc[k][0] += t[k]["ms"]; acc[k][1] += t[k]["launches"]
    per = {k: v[0] / max(v[1], 1) for k, v in acc.items()}
    print(json.dumps({"label": a.label, "n": n, "bits": [a.bits1, a.bits2], "hist_ms": round(per["hist"], 3),
                      "scatter_ms": round(per["scatter"], 3), "scatter_GBs(32B/t)": round(32 * n / per["scatter"] / 1e6),
                      "scan_ms": round(per["scan"], 3)}))
    sys.exit(0)
if a.join_only:
    tb = a.bits1 + (a.bits2 if a.passes == 2 else 0)
    pR, pS = e.alloc(16 * n), e.alloc(16 * n)
    sR, sS = e.alloc(8 * ((1 << tb) + 1)), e.alloc(8 * ((1 << tb) + 1))
    e.partition(dR, n, a.bits1, a.bits2 if a.passes == 2 else 0, pR, sR)
    e.partition(dS, n, a.bits1, a.bits2 if a.passes == 2 else 0, pS, sS)
    e.set_profiling(True)
    ms = []
    for _ in range(a.reps + 1):
        cnt = e.bucket_join(pR, sR, pS, sS, 1 << tb, tb, dO, n, probe_split=a.probe_split)
        ms.append(e.timings()["join"]["ms"])
    ok = (cnt, e.pairs_checksum(dO, cnt)) == exp
    ms = ms[1:]
    print(json.dumps({"label": a.label or os.environ.get("RHJ_VARIANT", ""), "n": n, "bits": [a.bits1, a.bits2], "ok": ok,
                      "join_ms": [round(x, 3) for x in ms], "join_GBs(48B/t)": round(48 * n / (sum(ms) / len(ms)) / 1e6)}))
    sys.exit(0 if ok else 1)
e.join_dev(dR, n, dS, n, dO, n, opts=opts)
e.set_profiling(True)
acc = {}
for _ in range(a.reps):
    cnt = e.join_dev(dR, n, dS, n, dO, n, opts=opts)
    t = e.timings()
    for k in ("hist", "scan", "scatter", "tasks", "join", "aux"):
        acc.setdefault(k, [0.0, 0])
        acc[k][0] += t[k]["ms"]; acc[k][1] += t[k]["launches"]
    acc.setdefault("total", [0.0, 0]); acc["total"][0] += t["total_ms"]; acc["total"][1] += 1
    ntasks = t["ntasks"]
ok = (cnt, e.pairs_checksum(dO, cnt)) == exp
per = {k: v[0] / max(v[1], 1) for k, v in acc.items()}
pl = e.timings()
res = {"label": a.label or os.environ.get("RHJ_VARIANT", ""), "n": n, "bits": [pl["bits1"], pl["bits2"]] if a.auto else [a.bits1, a.bits2], "ok": ok,
       "hist_ms": round(per["hist"], 3), "hist_GBs(16B/t)": round(16 * n / per["hist"] / 1e6, 0) if per["hist"] else 0,
       "scatter_ms": round(per["scatter"], 3), "scatter_GBs(32B/t)": round(32 * n / per["scatter"] / 1e6, 0) if per["scatter"] else 0,
       "scan_ms": round(per["scan"], 3),
       "join_ms": round(per["join"], 3), "join_GBs(48B/t)": round(48 * n / per["join"] / 1e6, 0),
       "ntasks": ntasks, "total_ms": round(per["total"], 3), "Mtuples/s": round(2 * n / per["total"] / 1e3, 0)}
print(json.dumps(res))
