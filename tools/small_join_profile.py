#!/usr/bin/env python3
"""tools/small_join_profile.py -- where the time of a SMALL host-pointer join goes (development aid).
Times rhj_join (C call only) for a few (|R|, |S|, matches) shapes of small.work and prints mean microseconds."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import radixhashjoin_amd as rhj
e = rhj.Engine(0)
rng = np.random.default_rng(1)
shapes = [(1, 1561, 1), (16, 17296, 187), (1561, 3754, 1600), (3754, 14368, 14368), (3754, 39532, 39532), (11500, 3754, 11500),
          (28513, 11115, 84294), (42987, 43131, 1187333)]
for nR, nS, m in shapes:
    D = max(1, int(nR * nS / m))
    R = np.empty(nR, dtype=rhj.TUPLE); R["key"] = np.arange(nR); R["payload"] = rng.integers(0, D, nR, dtype=np.uint64)
    S = np.empty(nS, dtype=rhj.TUPLE); S["key"] = np.arange(nS); S["payload"] = rng.integers(0, D, nS, dtype=np.uint64)
    for _ in range(3):
        e.join_count_only_page(R, S)
    ts = []
    for _ in range(50):
        cnt, dt = e.join_count_only_page(R, S, timed=True)
        ts.append(dt)
    ts.sort()
    kb_in, kb_out = (nR + nS) * 16 / 1024, cnt * 16 / 1024
    print(f"{nR:6d} x {nS:6d} -> {cnt:8d} pairs  in {kb_in:8.1f} KiB  out {kb_out:9.1f} KiB   median {ts[25]*1e6:8.1f} us   min {ts[0]*1e6:8.1f} us", flush=True)
