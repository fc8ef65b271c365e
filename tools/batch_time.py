#!/usr/bin/env python3
"""tools/batch_time.py -- the 94 joins of small.work (sizes and match counts of tests/golden/small_joins.json, synthetic values) through
rhj_join_batch with 0 / 1 / 3 / 5 / 7 helper threads (RHJ_BATCH_THREADS is read when a context first batches), next to 94 single
rhj_join calls: wall ms of the C calls from ONE thread (development aid; run on the GPU box)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import radixhashjoin_amd as rhj
eng = rhj.Engine(0)
meta = json.load(open("tests/golden/small_joins.json"))["calls"]
rng = np.random.default_rng(1)
cases = []
for c in meta:
    nR, nS, m = c["nR"], c["nS"], max(c["count"], 1)
    D = max(1, int(nR * nS / m))
    Rt = np.empty(nR, dtype=rhj.TUPLE); Rt["key"] = np.arange(nR); Rt["payload"] = rng.integers(0, D, nR, dtype=np.uint64)
    St = np.empty(nS, dtype=rhj.TUPLE); St["key"] = np.arange(nS); St["payload"] = rng.integers(0, D, nS, dtype=np.uint64)
    cases.append((Rt, St))
for thr in (0, 1, 3, 5, 7):
    os.environ["RHJ_BATCH_THREADS"] = str(thr)
    e = rhj.Engine(0)
    e.join_batch(cases, keep_pairs=False)
    ts = []
    for _ in range(7):
        cnt, dt = e.join_batch(cases, keep_pairs=False, timed=True)
        ts.append(dt)
    print("helpers", thr, "batch ms", [round(t * 1e3, 2) for t in sorted(ts)], "pairs", sum(cnt))
    e.close()
for Rt, St in cases: eng.join_count_only_page(Rt, St)
t0 = time.perf_counter()
tot = 0
for _ in range(5):
    for Rt, St in cases: tot += eng.join_count_only_page(Rt, St)
print("single ms", (time.perf_counter() - t0) / 5 * 1e3, "pairs", tot // 5)
