#!/usr/bin/env python3
"""Timeline of the host-pointer drop-in (rhj_join) on a large join: RHJ_TRACE_JOIN=1 python tools/host_join_trace.py [rows]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("RHJ_TRACE_JOIN", "1")
import bench
import radixhashjoin_amd as rhj
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128_000_000
R, S = bench.host_inputs(n, rhj.TUPLE)
eng = rhj.Engine(0)
eng.join(R[:1_000_000], S[:1_000_000])
for _ in range(4):
    cnt, dt = eng.join_count_only_page(R, S, timed=True)
    print(f"{n} x {n}: {dt * 1e3:.1f} ms, {(32.0 * n + 16.0 * cnt) / dt / 1e9:.1f} GB/s", flush=True)
