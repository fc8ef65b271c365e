#!/usr/bin/env python3
"""tools/verify_host.py -- full-size results checked by code OUTSIDE librhj_hip.so.

bench.py and the GPU suite verify the 10^9-tuple runs with two of the product's own kernels (k_checksum over the pairs against
k_expected_pkfk over S), both pinned to the oracle only at oracle-sized inputs.  This script closes that loop once per
workload: it runs rhj_join_dev at full size, copies S and the pair buffer to the HOST, and checks the pair set from first
principles with numpy -- no k_checksum, no k_expected_pkfk, nothing of the oracle either (it does not scale to 10^9):

    R[i] = {i, mix(i + 1)} is a primary key over ranks 1..n, every S tuple {j, mix(k_j)} has exactly one partner, so the result
    must be exactly { (k_j - 1, j) : j in [0, nS) }, which holds iff
      (1) count == nS,
      (2) the keyS column is a permutation of 0..nS-1 (every S tuple reported once, none twice),
      (3) for every pair (r, s):  S.payload[s] == mix(r + 1) == R.payload[r]      (the two tuples really have equal join values).
    Then the order-insensitive checksum of SURVEY App. A is recomputed on the host over the pairs AND from S alone (closed form),
    and both are compared with what the device kernels report.

    python tools/verify_host.py --n 1000000000 --dist uniform|zipf [--out gpurun_out/verify.jsonl]
Host memory: 40 n bytes (S, pairs, a bitmap); run time: minutes (random gathers over S in numpy)."""
import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import radixhashjoin_amd as rhj  # noqa: E402
from radixhashjoin_amd.binding import GEN_R, GEN_S_UNIFORM, GEN_S_ZIPF  # noqa: E402

M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def mix(z):
    """splitmix64 step of SURVEY §8d (numpy, wrapping)"""
    with np.errstate(over="ignore"):
        z = z + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def unmix(x):
    with np.errstate(over="ignore"):
        x = x ^ (x >> np.uint64(31)) ^ (x >> np.uint64(62))
        x = x * np.uint64(0x319642B2D24D8EC3)
        x = x ^ (x >> np.uint64(27)) ^ (x >> np.uint64(54))
        x = x * np.uint64(0x96DE1B173F119089)
        x = x ^ (x >> np.uint64(30)) ^ (x >> np.uint64(60))
        return x - np.uint64(0x9E3779B97F4A7C15)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1_000_000_000)
    ap.add_argument("--dist", choices=["uniform", "zipf"], default="uniform")
    ap.add_argument("--out", default="")
    ap.add_argument("--threads", type=int, default=12)
    a = ap.parse_args()
    n = a.n
    t_start = time.perf_counter()
    e = rhj.Engine(0)
    dR, dS, dO = e.alloc(16 * n), e.alloc(16 * n), e.alloc(16 * (n + 1024))
    e.generate(GEN_R, dR, n, 0, n)
    e.generate(GEN_S_ZIPF if a.dist == "zipf" else GEN_S_UNIFORM, dS, n, 0, n, seed=42, theta_milli=900)
    dev_expected = e.expected_pkfk(dS, n)                    # what the suite trusts: (count, checksum) from k_expected_pkfk
    cnt = e.join_dev(dR, n, dS, n, dO, n + 1024)
    t = e.timings()
    dev_checksum = e.pairs_checksum(dO, cnt)                 # ... and k_checksum over the pairs
    # R on the host: only a sample (its closed form mix(i + 1) is what check (3) uses; the sample pins the generator)
    Rs = dR.to_numpy(rhj.TUPLE, 1 << 20)
    assert np.array_equal(Rs["key"], np.arange(1 << 20, dtype=np.uint64)) and np.array_equal(Rs["payload"], mix(Rs["key"] + np.uint64(1)))
    dR.free()
    S = dS.to_numpy(rhj.TUPLE, n)                            # 16 n bytes
    pairs = dO.to_numpy(rhj.PAIR, cnt)                       # 16 cnt bytes
    t_copied = time.perf_counter()
    say = lambda what: print(f"[verify_host {time.perf_counter() - t_start:7.1f} s] {what}", file=sys.stderr, flush=True)
    say(f"join done (count {cnt}), S and pairs on the host")
    assert np.array_equal(S["key"][:1 << 20], np.arange(1 << 20, dtype=np.uint64))
    Sp = np.ascontiguousarray(S["payload"])
    seen = np.zeros(n, dtype=np.uint8)
    CH = 1 << 25

    def chunk(lo):
        hi = min(lo + CH, cnt)
        r, s = pairs["keyR"][lo:hi], pairs["keyS"][lo:hi]
        in_range = bool((s < np.uint64(n)).all() and (r < np.uint64(n)).all())
        si = s.astype(np.int64)
        equal = bool(np.array_equal(Sp[si], mix(r + np.uint64(1)))) if in_range else False      # (3)
        with np.errstate(over="ignore"):
            chk = int(np.sum(mix((r * np.uint64(0x100000001B3)) ^ mix(s)), dtype=np.uint64))
        return in_range, equal, chk

    with ThreadPoolExecutor(a.threads) as pool:
        parts = list(pool.map(chunk, range(0, max(cnt, 1), CH)))
    say("pairs checked against S, checksum over the pairs done")
    in_range = all(p[0] for p in parts)
    equal = all(p[1] for p in parts)
    host_checksum_pairs = sum(p[2] for p in parts) & 0xFFFFFFFFFFFFFFFF
    # (2) keyS is a permutation of 0..n-1: mark every s once (single pass, no threads: plain fancy assignment), then all marked
    # and count == n  ==>  no s twice
    if in_range:
        for lo in range(0, cnt, CH):
            seen[pairs["keyS"][lo:lo + CH].astype(np.int64)] = 1
    permutation = bool(in_range and cnt == n and seen.all())
    say("permutation check done")
    # closed form from S alone, on the host: pair of S tuple j is (unmix(payload_j) - 1, j)
    def closed(lo):
        hi = min(lo + CH, n)
        with np.errstate(over="ignore"):
            k = unmix(Sp[lo:hi])
            j = S["key"][lo:hi]
            return int(np.sum(mix(((k - np.uint64(1)) * np.uint64(0x100000001B3)) ^ mix(j)), dtype=np.uint64))
    with ThreadPoolExecutor(a.threads) as pool:
        host_checksum_closed = sum(pool.map(closed, range(0, n, CH))) & 0xFFFFFFFFFFFFFFFF
    res = {"workload": f"{n} x {n} {a.dist} uint64 PK/FK, rhj_join_dev, automatic plan", "plan": [t["passes"], t["bits1"], t["bits2"]],
           "narrow": e.info("last.narrow"), "join_kernel": e.info("last.join_kernel"),
           "count_device": cnt, "count_expected": n,
           "host_checks": {"rowIDs_in_range": in_range, "keyS_is_a_permutation_of_all_S_rows": permutation,
                           "join_values_equal_for_every_pair": equal},
           "checksum_host_over_pairs": f"{host_checksum_pairs:016x}", "checksum_host_closed_form_from_S": f"{host_checksum_closed:016x}",
           "checksum_device_k_checksum": f"{dev_checksum:016x}", "checksum_device_k_expected_pkfk": f"{dev_expected[1]:016x}",
           "count_device_k_expected_pkfk": dev_expected[0],
           "seconds": {"device_and_copies": round(t_copied - t_start, 1), "host_checks": round(time.perf_counter() - t_copied, 1)}}
    res["verified_on_host"] = bool(cnt == n and in_range and permutation and equal and
                                   host_checksum_pairs == host_checksum_closed == dev_checksum == dev_expected[1] and dev_expected[0] == n)
    line = json.dumps(res)
    print(line, flush=True)
    if a.out:
        with open(a.out, "a") as f:
            f.write(line + "\n")
    sys.exit(0 if res["verified_on_host"] else 1)


if __name__ == "__main__":
    main()
