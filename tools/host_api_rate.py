#!/usr/bin/env python3
"""tools/host_api_rate.py -- end-to-end rate of the host-pointer drop-in (rhj_join: H2D + kernels + D2H
into a fresh malloc'd result page), timed around the C-ABI call only."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import radixhashjoin_amd as rhj
e = rhj.Engine(0); lib = e.lib; libc = C.CDLL(None); libc.free.argtypes = [C.c_void_p]
rng = np.random.default_rng(1)
for n in (43131, 1_000_000, 16_000_000, 128_000_000):
    R, S = np.empty(n, dtype=rhj.TUPLE), np.empty(n, dtype=rhj.TUPLE)          # PK/FK: every S tuple matches one R tuple
    R["key"] = np.arange(n, dtype=np.uint64); R["payload"] = (R["key"] + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)
    S["key"] = R["key"]; S["payload"] = (rng.integers(0, n, n, dtype=np.uint64) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)
    reps = 20 if n <= 1_000_000 else 3
    tot = 0.0
    for i in range(reps + 1):
        page, cnt = C.c_void_p(), C.c_uint64()
        t = time.perf_counter()
        rc = lib.rhj_join(e.ctx, R.ctypes.data, n, S.ctypes.data, n, None, C.byref(page), C.byref(cnt))
        dt = time.perf_counter() - t
        assert rc == 0 and cnt.value == n
        libc.free(page)
        if i: tot += dt
    dt = tot / reps
    print(f"rhj_join {n} x {n}: {dt*1e3:.2f} ms  {2*n/dt/1e6:.1f} Mtuples/s  ({48*n/dt/1e9:.1f} GB/s over PCIe)", flush=True)
