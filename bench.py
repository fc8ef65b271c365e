#!/usr/bin/env python3
"""bench.py -- join throughput of the MI355X radix hash join on BASELINE.json's headline workload.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one complete radixHashJoin of the synthetic workload with inputs already resident in
HBM: partition R, partition S (2 passes of 8+8 radix bits by default, BASELINE config 3 as named), bucket
build/probe, result pairs written to HBM, exact result count read back.  Rank 0 prints ONE JSON
line.  `value` = (|R|+|S|) tuples joined per second, whole job (all ranks).

N == 1 : 1B x 1B uniform uint64 PK/FK join (BASELINE.json configs[2], the configuration the metric's
         target is quoted on); every step's pair set is verified by (count, checksum) against the
         closed form (rhj_expected_pkfk_dev), which tests/ pin to the CPU oracle.
N  > 1 : weak scaling: the global relations (N x 1B rows each) are range-sharded by row over the
         ranks; one RCCL all-to-all over xGMI redistributes tuples by owner radix bits, then every
         rank joins its partitions locally (radixhashjoin_amd/sharded.py).  Started WITHOUT a launcher
         (`python bench.py --gpus N`, no WORLD_SIZE in the environment) the script starts its own N ranks:
         it runs `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same args>` as a
         CHILD process before anything here has touched the GPU, and relays rank 0's JSON line and the exit code.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
SCATTER_BYTES_PER_TUPLE = 32   # algorithmic: 16 B tuple read + 16 B tuple write per partition pass
HIST_BYTES_PER_TUPLE = 8       # algorithmic: join value read for the histogram


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def source_blobs():
    """git blob ids of the kernel sources (what `git hash-object` prints; computed here, the GPU box has no .git): profiles/
    traffic.json carries the ids of the sources it was measured on, and `roofline.traffic` is only replayed from it while they
    are the sources of the library that runs"""
    import hashlib
    ids = {}
    d = os.path.join(ROOT, "radixhashjoin_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            data = open(os.path.join(d, f), "rb").read()
            ids[f] = hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()
    return ids


def self_launch(args_list, gpus):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves.  Nothing in this process has touched the
    GPU (torch is not even imported yet), the ranks are a child process tree, and this process only waits, relays the
    child's output (rank 0 prints the JSON line) and returns its exit code -- never an exec over a process that holds a GPU."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + args_list
    p = subprocess.run(cmd, env=env)
    return p.returncode


def np_mix(z):
    """splitmix64 step of SURVEY §8d on a numpy uint64 array (wrapping arithmetic)"""
    import numpy as np
    z = z + np.uint64(0x9E3779B97F4A7C15)
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def host_inputs(n, tuple_dtype, seed=42):
    """host AoS relations of the SURVEY §8d generators (R[i] = {i, mix(i+1)}; counter-based uniform FK: S[j] =
    {j, mix(1 + mix(j ^ seed) % n)}), made with numpy: the GPU legs of this file do not touch oracle/"""
    import numpy as np
    i = np.arange(n, dtype=np.uint64)
    R = np.empty(n, dtype=tuple_dtype)
    R["key"] = i
    R["payload"] = np_mix(i + np.uint64(1))
    S = np.empty(n, dtype=tuple_dtype)
    S["key"] = i
    S["payload"] = np_mix(np.uint64(1) + np_mix(i ^ np.uint64(seed)) % np.uint64(n))
    return R, S


def cpu_baseline(sample_n, reps=5):
    """SURVEY §8d protocol: the reference's pthread CPU path (oracle/_ref, compiled from the reference's own sources) on
    this host's cores, NUM_OF_THREADS = 8 (the reference's default) and = 1, one warm-up + median of `reps` runs each, on a
    bounded sample of the same workload family.  Falls back to the scalar oracle port where oracle/_ref is absent."""
    import statistics
    from oracle import pyoracle
    o = pyoracle.Oracle()
    R, S = o.gen_R(sample_n), o.gen_S_counter(sample_n, sample_n, 42)
    out = {"unit": "tuples/s", "host_cpus": os.cpu_count(), "cpu_model": cpu_model(),
           "sample": f"{sample_n} x {sample_n} uniform uint64 PK/FK join, one multiRadixHashJoin call per run, "
                     f"1 warm-up + median of {reps}"}
    if pyoracle.ref_available():
        runs = {}
        for threads in (8, 1):
            if not pyoracle.ref_available(threads):
                continue
            ref = pyoracle.Reference(threads)
            assert ref.num_threads == threads
            ref.join(R[:1_000_000], S[:1_000_000], want_pairs=False)           # warm-up
            secs = []
            for _ in range(reps):
                _, cnt, _, sec = ref.join(R, S, want_pairs=False)
                assert cnt == sample_n
                secs.append(sec)
            runs[threads] = secs
        best = 8 if 8 in runs else 1
        out.update({"value": 2 * sample_n / statistics.median(runs[best]), "cores": best, "kind": "reference",
                    "runs_s": {f"{t}_threads": [round(x, 3) for x in v] for t, v in runs.items()},
                    "tuples_per_s": {f"{t}_threads": 2 * sample_n / statistics.median(v) for t, v in runs.items()}})
    else:
        secs = []
        for _ in range(reps):
            t0 = time.perf_counter()
            cnt, _ = o.join_count_checksum(R, S)
            secs.append(time.perf_counter() - t0)
            assert cnt == sample_n
        out.update({"value": 2 * sample_n / statistics.median(secs), "cores": 1, "kind": "port",
                    "runs_s": {"1_threads": [round(x, 3) for x in secs]}})
    return out


def shard_rank_kernel_time(eng, torch, dev, world, n, dist_kind):
    """BASELINE config 5 as far as ONE GPU can measure it: this process plays rank 0 of `world` ranks of the sharded join
    (radixhashjoin_amd/sharded.py; n rows of R and of S per rank, rowIDs global): every rank's shards are generated, their
    class histograms gathered, the class ranges cut, every rank's shard split into the 12-byte wire format -- and rank 0
    receives what the all-to-all would deliver (device-side copies of the other ranks' slices), partitions it and joins it.
    Returns the device time of rank 0's kernels (HIP events of the engine, the figure `kernel_ms_per_step` reports for the
    single-GPU step) next to the bytes that would cross xGMI.  Verified: rank 0's pairs against the closed form."""
    import numpy as np
    from radixhashjoin_amd.binding import (GEN_R, GEN_S_UNIFORM, GEN_S_ZIPF, SHARD_PLAIN, narrow_bytes, narrow_key_offset, shard_plan)
    from radixhashjoin_amd.sharded import balanced_cuts
    SHIFT, BITS = 20, 8
    nglob = n * world
    kind = GEN_S_ZIPF if dist_kind == "zipf" else GEN_S_UNIFORM
    acc = {"stats": 0.0, "split": 0.0, "partition": 0.0, "join": 0.0}

    def collect(tag):
        t = eng.timings()
        acc[tag] += sum(v["ms"] for v in t.values() if isinstance(v, dict))

    def shards(rank):
        R = torch.empty((n, 2), dtype=torch.int64, device=dev)
        S = torch.empty((n, 2), dtype=torch.int64, device=dev)
        eng.generate(GEN_R, R, n, row0=rank * n, D=nglob)
        eng.generate(kind, S, n, row0=rank * n, D=nglob, seed=42, theta_milli=900)
        return R, S

    eng.set_profiling(True)
    hists, kmins, kmaxs = [], [], []
    for rank in range(world):                                   # the all-gather: every rank's class counts and rowID ranges
        R, S = shards(rank)
        for side, rel in ((0, R), (1, S)):
            h, lo, hi = eng.shard_stats(side, rel, n, SHIFT, BITS)
            hists.append(h); kmins.append(lo); kmaxs.append(hi)
        del R, S
    cuts = balanced_cuts(np.sum(hists, axis=0).tolist(), world)
    lo_c, hi_c = cuts[0], cuts[1]
    recv = [sum(int(hists[2 * r + side][cuts[d]:cuts[d + 1]].sum()) for r in range(world)) for side in (0, 1) for d in range(world)]
    mode, plan = shard_plan(max(recv[:world]), max(recv[world:]), None)
    if not mode:
        return {"skipped": "sizes outside the narrow sharded path"}
    plain = max(kmaxs) < (1 << 32)
    if plain:
        mode = SHARD_PLAIN
    row0 = {side: [0 if plain else kmins[2 * r + side] for r in range(world)] for side in (0, 1)}
    pieces = {0: [], 1: []}
    sent_bytes = 0
    exp_cnt = exp_chk = 0
    for rank in range(world):                                   # every rank splits; rank 0 keeps what it would receive
        R, S = shards(rank)
        for side, rel in ((0, R), (1, S)):
            buf = torch.empty(max(narrow_bytes(n), 16), dtype=torch.uint8, device=dev)
            eng.shard_stats(side, rel, n, SHIFT, BITS)          # (the unit tables the split uses; what the rank itself runs first)
            if rank == 0:
                collect("stats")
            eng.shard_split(side, rel, n, SHIFT, BITS, row0[side][rank], buf)
            if rank == 0:
                collect("split")
            st = np.concatenate([[0], np.cumsum(hists[2 * rank + side])])
            x, y = int(st[lo_c]), int(st[hi_c])
            koff = narrow_key_offset(n)
            pieces[side].append((buf[8 * x:8 * y].view(torch.int64).clone(), buf[koff + 4 * x:koff + 4 * y].view(torch.int32).clone()))
            if rank == 0:
                sent_bytes += 12 * (n - (y - x))
            del buf
        del R, S
    seg = {side: [0] for side in (0, 1)}
    arr = {}
    for side in (0, 1):
        for P, K in pieces[side]:
            seg[side].append(seg[side][-1] + P.shape[0])
        arr[side] = (torch.cat([p for p, _ in pieces[side]]), torch.cat([k for _, k in pieces[side]]))
    del pieces
    mR, mS = seg[0][-1], seg[1][-1]
    for side in (0, 1):
        eng.shard_partition(side, arr[side][0], arr[side][1], seg[side][-1], seg[side], row0[side], plan, mode)
        collect("partition")
    out = torch.empty((max(mR, mS) + 1024, 2), dtype=torch.int64, device=dev)
    cnt = eng.shard_join(out, out.shape[0])
    collect("join")
    eng.set_profiling(False)
    # rank 0 owns classes [lo_c, hi_c): its pairs are exactly the S tuples of those classes (PK/FK, every S tuple matches once)
    chk = eng.pairs_checksum(out, cnt)
    # closed form over what rank 0 received of S: payload = mix(k) matches R row k - 1; rowID = row0 + local
    Sx = torch.empty((mS, 2), dtype=torch.int64, device=dev)
    base = torch.zeros(mS, dtype=torch.int64, device=dev)
    for r in range(world):
        base[seg[1][r]:seg[1][r + 1]] = row0[1][r] if row0[1][r] < (1 << 63) else row0[1][r] - (1 << 64)
    Sx[:, 0] = arr[1][1].to(torch.int64) & 0xFFFFFFFF
    Sx[:, 0] += base
    Sx[:, 1] = arr[1][0]
    exp_cnt, exp_chk = eng.expected_pkfk(Sx, mS)
    total = sum(acc.values())
    return {"world_emulated": world, "rows_per_rank": n, "dist": dist_kind, "rowid_mode": {1: "tagged", 2: "global16", 3: "plain"}[mode],
            "local_plan": [plan.passes, plan.bits1, plan.bits2], "recv_tuples_rank0": [mR, mS],
            "rank0_kernel_ms": {k: round(v, 3) for k, v in acc.items()}, "rank0_kernel_ms_total": round(total, 3),
            "exchange_bytes_per_rank": sent_bytes, "bytes_per_tuple_sent": 12,
            "xgmi_ms_at_153GBps_per_link": round(sent_bytes / max(world - 1, 1) / 153e9 * 1e3, 2),
            "verified": (cnt, chk) == (exp_cnt, exp_chk),
            "note": "one GPU playing rank 0 of the sharded join with real segments from every emulated rank; device time of its "
                    "kernels; the exchange itself is modelled (direct peer links, all busy)"}


def extras_host(eng, which="all"):
    """N == 1 only, outside the timed region and BEFORE the headline's buffers exist: the host-pointer legs (the drop-in call end
    to end, the 94 joins of small.work single / batched / from 8 threads).  They run first because a process that has just
    released a few hundred GB of HBM sees its device-to-host copies run at half rate for a while ([measured] round 3: 126 ms
    instead of 87-94 for the 128M leg when it ran after the 10^9-tuple legs) -- a property of the runtime, not of the join."""
    import numpy as np
    import radixhashjoin_amd as rhj
    res = {}
    # the drop-in as the reference calls it: host AoS in, one malloc'd result page out (PCIe inclusive, pageable memory)
    n = 128_000_000 if which == "all" else 16_000_000
    Rh, Sh = host_inputs(n, rhj.TUPLE)
    eng.join(Rh[:1_000_000], Sh[:1_000_000])
    secs = []
    for _ in range(3):
        cnt, dt_call = eng.join_count_only_page(Rh, Sh, timed=True)          # the C call alone (the page is freed outside)
        secs.append(dt_call)
    sec = sorted(secs)[1]
    res[f"end_to_end_rhj_join_{n // 1_000_000}Mx{n // 1_000_000}M"] = {"ms": sec * 1e3, "tuples_per_s": 2 * n / sec, "matches": cnt,
                                            "pcie_GBps": (32.0 * n + 16.0 * cnt) / sec / 1e9, "runs_ms": [round(x * 1e3, 1) for x in secs],
                                            "s_chunks_pipelined": eng.info("last.pipelined"),
                                            "note": "H2D of both inputs from pageable memory + kernels + D2H of the result page; pcie_GBps = (input + "
                                                    "result bytes) / wall time, above one direction's wire rate when download overlaps upload"}
    del Rh, Sh

    # config 1's joins: the 94 multiRadixHashJoin calls the reference makes on small.work (sizes from the link-time tap,
    # tests/golden/small_joins.json), synthetic inputs of those sizes and match counts, through the host-pointer call
    meta = json.load(open(os.path.join(ROOT, "tests", "golden", "small_joins.json")))["calls"]
    rng = np.random.default_rng(1)
    cases = []
    for c in meta:
        nR, nS, m = c["nR"], c["nS"], max(c["count"], 1)
        D = max(1, int(nR * nS / m))
        Rt = np.empty(nR, dtype=rhj.TUPLE); Rt["key"] = np.arange(nR); Rt["payload"] = rng.integers(0, D, nR, dtype=np.uint64)
        St = np.empty(nS, dtype=rhj.TUPLE); St["key"] = np.arange(nS); St["payload"] = rng.integers(0, D, nS, dtype=np.uint64)
        cases.append((Rt, St))
    for Rt, St in cases:
        eng.join_count_only_page(Rt, St)
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        for Rt, St in cases:
            eng.join_count_only_page(Rt, St)
    sec = (time.perf_counter() - t0) / reps
    res["small_work_94_joins_rhj_join"] = {"total_ms": sec * 1e3, "mean_us_per_join": sec / len(cases) * 1e6,
                                           "tuples_per_s": sum(len(a) + len(b) for a, b in cases) / sec}
    # ... and through rhj_join_batch from the same ONE thread: sixteen joins per launch, one staged upload and one
    # synchronisation per sixteen (the kernel writes the pairs into pinned host memory itself)
    eng.join_batch(cases, keep_pairs=False)
    secs_b = []
    for _ in range(reps):
        cnts, dt_b = eng.join_batch(cases, keep_pairs=False, timed=True)
        secs_b.append(dt_b)
    sec_b = sorted(secs_b)[len(secs_b) // 2]
    single_counts = [eng.join_count_only_page(Rt, St) for Rt, St in cases]
    res["small_work_94_joins_rhj_join_batch"] = {"total_ms": sec_b * 1e3, "mean_us_per_join": sec_b / len(cases) * 1e6,
                                                 "tuples_per_s": sum(len(a) + len(b) for a, b in cases) / sec_b,
                                                 "same_counts_as_rhj_join": [int(c) for c in cnts] == [int(c) for c in single_counts],
                                                 "note": "ONE caller thread, the C call alone (pages freed outside), median of %d" % reps}
    # the same 94 joins the way the reference issues them: 8 query threads (join.cpp:42-43, MainScheduler.cpp:6-14), each
    # with its own scheduler = its own rhj_ctx and HIP stream; ctypes releases the GIL inside the C call
    import threading
    nthr = 8
    engines = [rhj.Engine(0) for _ in range(nthr)]
    for k, e2 in enumerate(engines):
        for Rt, St in cases[k::nthr]:
            e2.join_count_only_page(Rt, St)
    def work(k, reps_):
        for _ in range(reps_):
            for Rt, St in cases[k::nthr]:
                engines[k].join_count_only_page(Rt, St)
    t0 = time.perf_counter()
    thr = [threading.Thread(target=work, args=(k, reps)) for k in range(nthr)]
    for t in thr:
        t.start()
    for t in thr:
        t.join()
    sec8 = (time.perf_counter() - t0) / reps
    for e2 in engines:
        e2.close()
    res["small_work_94_joins_rhj_join_8_query_threads"] = {"total_ms": sec8 * 1e3, "mean_us_per_join": sec8 / len(cases) * 1e6,
                                                           "note": "wall time of the 94 calls spread over 8 host threads, one context each"}
    return res


def extras(eng, torch, dev, steps, which="all"):
    """N == 1 only, outside the timed region: the other BASELINE configs and the host-pointer drop-in, so that every
    number DESIGN.md quotes is on the driver's record.  Each leg verifies its pair set (count + checksum)."""
    import numpy as np
    import radixhashjoin_amd as rhj
    from radixhashjoin_amd.binding import GEN_R, GEN_S_UNIFORM, GEN_S_ZIPF
    res = {}

    def timed_join(R, S, n, out, opts, reps):
        eng.join_dev(R, n, S, n, out, out.shape[0], opts=opts)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            cnt = eng.join_dev(R, n, S, n, out, out.shape[0], opts=opts)
        torch.cuda.synchronize()
        return cnt, (time.perf_counter() - t0) / reps

    # config 2: 1M x 1M uniform, one 8-bit pass (what the automatic plan picks), device-resident
    n = 1_000_000
    R = torch.empty((n, 2), dtype=torch.int64, device=dev)
    S = torch.empty((n, 2), dtype=torch.int64, device=dev)
    out = torch.empty((n + 1024, 2), dtype=torch.int64, device=dev)
    eng.generate(GEN_R, R, n, D=n)
    eng.generate(GEN_S_UNIFORM, S, n, D=n, seed=42)
    exp = eng.expected_pkfk(S, n)
    cnt, sec = timed_join(R, S, n, out, rhj.Opts(1, 8, 0), 200)
    res["c2_1Mx1M_8bit"] = {"ms": sec * 1e3, "tuples_per_s": 2 * n / sec,
                            "verified": (cnt, eng.pairs_checksum(out, cnt)) == exp}
    del R, S, out

    # join values that defeat radix digits taken from raw low bits (multiples of 2^16 / 2^28, k * 65536 + const, dense i + 1
    # -- SURVEY §8d's "dense variant"): the SAME PK/FK pair set re-labelled (rhj_remap_keys_dev), automatic plan, next to the
    # uniform workload of the same size.  The engine partitions on rhj_mix64(payload) (include/rhj.h), so these cost what
    # uniform keys cost; with raw bits (rhj_set_option "partition.mix" 0) the k << 16 case is ONE partition of 64M tuples.
    n = 64_000_000 if which == "all" else 8_000_000
    R = torch.empty((n, 2), dtype=torch.int64, device=dev)
    S = torch.empty((n, 2), dtype=torch.int64, device=dev)
    out = torch.empty((n + 1024, 2), dtype=torch.int64, device=dev)
    leg = {}
    for name, shift, add in (("uniform", None, 0), ("k<<16", 16, 0), ("k<<28", 28, 0), ("k*65536+const", 16, 12345), ("dense_i+1", 0, 0)):
        eng.generate(GEN_R, R, n, D=n)
        eng.generate(GEN_S_UNIFORM, S, n, D=n, seed=42)
        exp = eng.expected_pkfk(S, n)                   # the pair set does not depend on how the join values are labelled
        if shift is not None:
            eng.remap_keys(R, n, shift, add)
            eng.remap_keys(S, n, shift, add)
        cnt, sec = timed_join(R, S, n, out, None, 10)
        leg[name] = {"ms": sec * 1e3, "tuples_per_s": 2 * n / sec, "max_partition_R": eng.info("last.max_part_R"),
                     "verified": (cnt, eng.pairs_checksum(out, cnt)) == exp}
    for name in leg:
        leg[name]["vs_uniform"] = leg[name]["ms"] / leg["uniform"]["ms"]
    res[f"aligned_join_values_{n // 1_000_000}Mx{n // 1_000_000}M_auto"] = leg
    del R, S, out

    # config 4: 1B x 1B Zipf(0.9) foreign key, named 8+8 plan (= the automatic plan) and 9+9
    n = 1_000_000_000
    free, _ = eng.mem_info()
    if which == "all" and free > 16 * n * 6.5:
        R = torch.empty((n, 2), dtype=torch.int64, device=dev)
        S = torch.empty((n, 2), dtype=torch.int64, device=dev)
        out = torch.empty((n + 1024, 2), dtype=torch.int64, device=dev)
        eng.generate(GEN_R, R, n, D=n)
        eng.generate(GEN_S_ZIPF, S, n, D=n, seed=42, theta_milli=900)
        exp = eng.expected_pkfk(S, n)
        res["c4_1Bx1B_zipf0.9"] = {}
        for name, opts in (("8+8", rhj.Opts(2, 8, 8)), ("9+9", rhj.Opts(2, 9, 9))):
            cnt, sec = timed_join(R, S, n, out, opts, steps)
            res["c4_1Bx1B_zipf0.9"][name] = {"ms": sec * 1e3, "tuples_per_s": 2 * n / sec,
                                             "verified": (cnt, eng.pairs_checksum(out, cnt)) == exp}
        # the headline workload with the arguments EXCHANGED (foreign-key relation first), and what the choice of the build side
        # is worth: where a partition's two sides are within 1/16 of each other the engine builds the hash table on the side
        # whose sampled join values show fewer duplicates ("join.sniff", include/rhj.h) -- so the order of the arguments does
        # not matter; with the sampling off the FIRST argument wins such ties (fast for R JOIN S, slow for S JOIN R).
        eng.generate(GEN_S_UNIFORM, S, n, D=n, seed=42)
        exp_n, _ = eng.expected_pkfk(S, n)
        leg = {}
        for name, sniff, A, B in (("R_join_S", -1, R, S), ("S_join_R", -1, S, R), ("S_join_R_sampling_off", 0, S, R)):
            eng.set_option("join.sniff", sniff)
            cnt, sec = timed_join(A, B, n, out, rhj.Opts(2, 8, 8), steps)
            leg[name] = {"ms": sec * 1e3, "tuples_per_s": 2 * n / sec, "count_ok": cnt == exp_n}
        eng.set_option("join.sniff", -1)
        res["build_side_1Bx1B_uniform"] = leg
        del R, S, out
    eng.release_workspace()
    torch.cuda.empty_cache()

    # BASELINE config 5, per rank: what one GPU of an 8-GPU job computes at 10^9 rows per rank (uniform), and of a 4-GPU job
    free, _ = eng.mem_info()
    if which == "all" and free > 200e9:
        for world_ in (8, 4):
            res[f"c5_sharded_rank0_of_{world_}_1Bx1B_per_rank"] = shard_rank_kernel_time(eng, torch, dev, world_, 1_000_000_000, "uniform")
            eng.release_workspace()
            torch.cuda.empty_cache()

    # beyond the 16-bit plans (the engine's own scaling axis; the reference has one fixed 8-bit pass, Result.cpp:5,91):
    # 1.5 and 2.2 * 10^9 tuples per side under the automatic plan (17 bits = 9+8, narrow format), HBM permitting
    from radixhashjoin_amd.binding import plan as rhj_plan
    for n, kinds in ((1_500_000_000, (("uniform", GEN_S_UNIFORM), ("zipf0.9", GEN_S_ZIPF))), (2_200_000_000, (("uniform", GEN_S_UNIFORM),))):
        free, _ = eng.mem_info()
        if which != "all" or free < 16 * n * 6.3:
            continue
        R = torch.empty((n, 2), dtype=torch.int64, device=dev)
        S = torch.empty((n, 2), dtype=torch.int64, device=dev)
        out = torch.empty((n + 1024, 2), dtype=torch.int64, device=dev)
        eng.generate(GEN_R, R, n, D=n)
        p_ = rhj_plan(n, n)
        for name, kind in kinds:
            eng.generate(kind, S, n, D=n, seed=42, theta_milli=900)
            exp = eng.expected_pkfk(S, n)
            cnt, sec = timed_join(R, S, n, out, None, 3)
            res[f"{n / 1e9:.1f}Bx{n / 1e9:.1f}B_{name}_auto"] = {
                "plan": f"{p_.passes}-pass ({p_.bits1}+{p_.bits2} bit)", "ms": sec * 1e3, "tuples_per_s": 2 * n / sec,
                "narrow": eng.info("last.narrow"), "join_kernel": eng.info("last.join_kernel"),
                "verified": (cnt, eng.pairs_checksum(out, cnt)) == exp}
        del R, S, out
        eng.release_workspace()
        torch.cuda.empty_cache()

    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--tuples", type=int, default=1_000_000_000, help="rows of R and of S per GPU")
    ap.add_argument("--bits1", type=int, default=8)
    ap.add_argument("--bits2", type=int, default=8)
    ap.add_argument("--passes", type=int, default=2, choices=[1, 2], help="1: single pass of --bits1 bits (BASELINE config 2)")
    ap.add_argument("--dist", choices=["uniform", "zipf"], default="uniform")
    ap.add_argument("--cpu-sample", type=int, default=128_000_000, help="rows per side of the CPU baseline sample (0 = skip)")
    ap.add_argument("--no-extras", action="store_true", help="skip the other BASELINE configs / end-to-end legs (N == 1)")
    ap.add_argument("--extras", choices=["all", "small"], default="all", help="small: skip the 1B Zipf leg, 16M end-to-end")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-auto", action="store_true", help="skip the extra (untimed-for-value) run under the automatic radix plan")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(sys.argv[1:], args.gpus))

    import torch
    import radixhashjoin_amd as rhj
    from radixhashjoin_amd.binding import GEN_R, GEN_S_UNIFORM, GEN_S_ZIPF

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if os.environ.get("RHJ_BENCH_BACKEND", "nccl") != "nccl":
        local_rank = 0                              # rehearsal: every rank on the one visible GPU
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        # RHJ_BENCH_BACKEND=gloo rehearses the multi-rank path with several ranks on ONE GPU (payloads staged
        # through the host); the real runs use RCCL
        backend = os.environ.get("RHJ_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    rdev = dev if (dist is None or dist.get_backend() == "nccl") else torch.device("cpu")
    # one explicit (non-null) stream shared by torch ops, RCCL hand-offs and the engine's kernels
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    eng = rhj.Engine(local_rank)
    eng.set_stream(stream.cuda_stream)
    n = args.tuples
    nglobal = n * world
    opts = rhj.Opts(2, args.bits1, args.bits2) if args.passes == 2 else rhj.Opts(1, args.bits1, 0)
    early = extras_host(eng, args.extras) if world == 1 and not args.no_extras else {}

    # inputs resident in HBM, generated on device (SURVEY §8d generators; 16 B AoS tuples)
    R = torch.empty((n, 2), dtype=torch.int64, device=dev)
    S = torch.empty((n, 2), dtype=torch.int64, device=dev)
    eng.generate(GEN_R, R, n, row0=rank * n, D=nglobal)
    eng.generate(GEN_S_ZIPF if args.dist == "zipf" else GEN_S_UNIFORM, S, n, row0=rank * n, D=nglobal, seed=42,
                 theta_milli=900)
    exp_cnt, exp_chk = eng.expected_pkfk(S, n)        # local part of the closed-form expectation

    if world == 1:
        out = torch.empty((n + 1024, 2), dtype=torch.int64, device=dev)
        eng.reserve(n, n, opts)

        def step(o=opts):
            return eng.join_dev(R, n, S, n, out, out.shape[0], opts=o), out
    else:
        from radixhashjoin_amd.sharded import ShardedJoin
        sj = ShardedJoin(eng, dist.group.WORLD, local_opts=opts)

        def step():
            return sj.join(R, n, S, n)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    eng.set_profiling(2)                               # HIP events around every launch; read ONCE, after the timed steps
    if world > 1:
        sj.kernel_ms = {}
    kt = {"hist": [0.0, 0], "scan": [0.0, 0], "scatter": [0.0, 0], "tasks": [0.0, 0], "join": [0.0, 0], "aux": [0.0, 0]}
    sc_pass = {1: [0.0, 0], 2: [0.0, 0]}               # scatter launches by pass (a two-pass join: R 1, R 2, S 1, S 2)
    barrier()
    t0 = time.perf_counter()
    cnt, res = 0, None
    for _ in range(args.steps):
        cnt, res = step()
    barrier()
    dt = time.perf_counter() - t0
    if world == 1:
        lt = eng.launch_timings()                      # every launch of the timed steps, in launch order
        for kind, ms in lt:
            if kind in kt:
                kt[kind][0] += ms
                kt[kind][1] += 1
        sc = [ms for kind, ms in lt if kind == "scatter"]
        if args.passes == 2 and len(sc) == 4 * args.steps:       # per step: R pass 1, R pass 2, S pass 1, S pass 2
            for i, ms in enumerate(sc):
                sc_pass[1 + (i & 1)][0] += ms
                sc_pass[1 + (i & 1)][1] += 1
    if world > 1:
        # per-kernel device time of the sharded path: `steps` MORE steps, outside the timed region (reading the HIP events of
        # an engine call synchronises after it, which the timed steps must not do)
        sj.collect_timings = True
        sj.kernel_ms = {}
        for _ in range(args.steps):
            cnt, res = step()
        sj.collect_timings = False
        for k in kt:
            kt[k] = list(sj.kernel_ms.get(k, [0.0, 0]))
        barrier()
    eng.set_profiling(False)

    # verification of the last step (outside the timed region): exact count + order-insensitive checksum
    ok = True
    if not args.no_verify:
        chk = eng.pairs_checksum(res, cnt)
        if dist is not None:
            v = torch.tensor([cnt, exp_cnt, chk - (1 << 64) if chk >= (1 << 63) else chk,
                              exp_chk - (1 << 64) if exp_chk >= (1 << 63) else exp_chk], dtype=torch.int64, device=rdev)
            dist.all_reduce(v)                          # wrapping int64 sums == sums mod 2^64
            ok = bool(v[0] == v[1]) and bool(v[2] == v[3])
        else:
            ok = (cnt == exp_cnt) and (chk == exp_chk)
    tmax = torch.tensor([dt], dtype=torch.float64, device=rdev)
    ranks_counted = 1
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        one = torch.ones(1, dtype=torch.int64, device=rdev)        # every rank adds 1: did the collective see N ranks?
        dist.all_reduce(one)
        ranks_counted = int(one.item())
    dt = float(tmax.item())

    # N == 1 only, outside the timed region: the same workload under the engine's automatic radix plan
    auto = None
    if world == 1 and not args.no_auto:
        from radixhashjoin_amd.binding import plan as rhj_plan
        ap_ = rhj_plan(n, n)
        if (ap_.passes, ap_.bits1, ap_.bits2) != (opts.passes, opts.bits1, opts.bits2):
            step(ap_)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                c2, r2 = step(ap_)
            torch.cuda.synchronize()
            d2 = time.perf_counter() - t1
            ok2 = args.no_verify or (c2 == exp_cnt and eng.pairs_checksum(r2, c2) == exp_chk)
            auto = {"plan": f"{ap_.passes}-pass ({ap_.bits1}+{ap_.bits2} bit)", "value": 2.0 * n * args.steps / d2,
                    "unit": "tuples/s", "ms_per_step": d2 / args.steps * 1e3, "verified": bool(ok2)}
            ok = ok and bool(ok2)

    if rank == 0:
        ms_step = dt / args.steps * 1e3
        value = 2.0 * nglobal * args.steps / dt
        sc_ms = kt["scatter"][0] / max(kt["scatter"][1], 1)
        sc_all_ms = sc_ms
        tuples_per_launch = n                                   # one launch scatters one relation shard once
        # The two passes of a narrow two-pass join are two instantiations of the scatter kernel that move different bytes
        # (pass 1: 16-byte tuples in, 12 B out; pass 2: 12 B in and out).  `roofline` prices the DOMINANT one -- pass 1, the
        # slower -- on its own launches; the other is in `variants`.
        by_pass = world == 1 and args.passes == 2 and sc_pass[1][1] and sc_pass[2][1]
        if by_pass:
            sc_ms = sc_pass[1][0] / sc_pass[1][1]
        achieved = SCATTER_BYTES_PER_TUPLE * tuples_per_launch / (sc_ms * 1e-3) / 1e9 if sc_ms else 0.0
        # format of the intermediate (rhj_get_info "last.narrow"): inside a join with a fused two-pass plan the scatter
        # writes {payload 8 B, rowID 4 B} arrays, so the bytes a launch must MOVE are fewer than SURVEY §8d's 32 B/tuple
        narrow = eng.info("last.narrow")
        moved_per_tuple = {0: 32.0, 1: (32.0 + 28.0) / 2, 2: (28.0 + 24.0) / 2}[narrow] if args.passes == 2 else 32.0
        pass_moved = {1: {0: 32.0, 1: 32.0, 2: 28.0}[narrow], 2: {0: 32.0, 1: 28.0, 2: 24.0}[narrow]}
        if by_pass:
            moved_per_tuple = pass_moved[1]
        moved = moved_per_tuple * tuples_per_launch / (sc_ms * 1e-3) / 1e9 if sc_ms else 0.0
        part_ms = (kt["hist"][0] + kt["scan"][0] + kt["scatter"][0]) / args.steps
        npass_tuples = 2 * args.passes * n                      # 2 relations x passes
        traffic, traffic_note, traffic_by_pass = None, None, {}
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            if tj.get("tuples") == n and tj.get("bits") == [args.bits1, args.bits2] and args.passes == 2:
                if tj.get("source_blobs") == source_blobs():
                    traffic_by_pass = {1: tj.get("scatter_pass1_hbm_bytes_per_launch"), 2: tj.get("scatter_pass2_hbm_bytes_per_launch")}
                    traffic = traffic_by_pass[1] if by_pass and traffic_by_pass[1] else tj.get("scatter_hbm_bytes_per_launch")
                    traffic_note = (f"{tj.get('source')}: two rocprofv3 --pmc passes of this command (FETCH_SIZE, WRITE_SIZE; profiles/"
                                    "run_profile.sh), taken on the kernel sources this library was built from (blob ids match); "
                                    "replayed, not measured by this run")
                else:
                    traffic_note = ("null: profiles/traffic.json was measured on other kernel sources than the ones in "
                                    "radixhashjoin_amd/csrc now (blob ids differ) -- re-run profiles/run_profile.sh + summarize.py")
        line = {
            "metric": "join throughput (build+probe tuples/s)", "value": value, "unit": "tuples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "verified": ok,
            "config": {"workload": f"{n} x {n} {args.dist} uint64 PK/FK radix hash join per GPU, "
                                   + (f"2-pass ({args.bits1}+{args.bits2} bit)" if args.passes == 2 else f"1-pass ({args.bits1} bit)")
                                   + " radix, inputs and pairs resident in HBM",
                       "tuples_R_global": nglobal, "tuples_S_global": nglobal, "matches_last_step_rank0": cnt,
                       "exchange": "none (single GPU)" if world == 1 else
                                   sj.transport() + " by balanced owner class ranges, wire format " +
                                   {"narrow12": "{payload 8 B, shard-local rowID 4 B}", "tuple16": "16-byte tuples"}[sj.stats["format"]]},
            "roofline": {"bound": "hbm",
                         "kernel": (("k_scatter_wcn<in_narrow=false>: pass 1 of a narrow two-pass join, the dominant kernel "
                                     "(16-byte tuples in, {payload 8 B, rowID 4 B} out)") if by_pass and narrow == 2 else
                                    ("k_scatter_wcn" if narrow else "k_scatter_wc")
                                    + " (line-aligned write-combining scatter-partition, one pass over one relation)"),
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic,
                         "traffic_replayed_from": traffic_note,
                         "variants": {f"pass {p_}": {"avg_launch_ms": sc_pass[p_][0] / sc_pass[p_][1],
                                                     "achieved": SCATTER_BYTES_PER_TUPLE * n / (sc_pass[p_][0] / sc_pass[p_][1] * 1e-3) / 1e9,
                                                     "frac": SCATTER_BYTES_PER_TUPLE * n / (sc_pass[p_][0] / sc_pass[p_][1] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                                     "moved_bytes_per_launch": pass_moved[p_] * n,
                                                     "achieved_moved": pass_moved[p_] * n / (sc_pass[p_][0] / sc_pass[p_][1] * 1e-3) / 1e9,
                                                     "traffic": traffic_by_pass.get(p_)}
                                      for p_ in (1, 2)} if by_pass else None,
                         "avg_launch_ms_all_scatter_launches": sc_all_ms,
                         "algorithmic_bytes_per_launch": SCATTER_BYTES_PER_TUPLE * tuples_per_launch,
                         "avg_launch_ms": sc_ms,
                         "intermediate_format": {0: "16 B tuples", 1: "16 B tuples, then {payload 8 B, rowID 4 B} partitions",
                                                 2: "{payload 8 B, rowID 4 B} arrays after both passes"}[narrow],
                         "moved_bytes_per_launch": moved_per_tuple * tuples_per_launch,
                         "achieved_moved": moved, "frac_moved": moved / HBM_PEAK_GBS,
                         "partition_pass_GBps": (40.0 * npass_tuples / (part_ms * 1e-3) / 1e9) if part_ms else 0.0},
            "kernel_ms_per_step": {k: v[0] / args.steps for k, v in kt.items()},
        }
        line["kernel_GBps"] = {"join (48 B per tuple pair: 16 B per input tuple + 16 B per pair)":
                               48.0 * n / (kt["join"][0] / max(kt["join"][1], 1) * 1e-3) / 1e9 if kt["join"][0] else 0.0,
                               "join, bytes moved (%d B per input tuple + 16 B per pair)" % (12 if narrow else 16):
                               ((24.0 if narrow else 32.0) + 16.0) * n / (kt["join"][0] / max(kt["join"][1], 1) * 1e-3) / 1e9
                               if kt["join"][0] else 0.0,
                               "hist (16 B per tuple)": 16.0 * n / (kt["hist"][0] / max(kt["hist"][1], 1) * 1e-3) / 1e9 if kt["hist"][0] else 0.0}
        if world > 1:
            line["sharded"] = {"wire_format": sj.stats["format"], "exchange_bytes_per_rank": sj.stats["exchange_bytes_sent"],
                               "bytes_per_tuple_sent": {"narrow12": 12, "tuple16": 16}[sj.stats["format"]],
                               "recv_tuples_rank0": [sj.stats["recv_R"], sj.stats["recv_S"]],
                               "local_plan": sj.stats.get("plan"), "rowid_mode": sj.stats.get("rowid_mode"), "backend": dist.get_backend(),
                               "ranks_seen": dist.get_world_size(), "ranks_in_all_reduce": ranks_counted,
                               "devices_visible": torch.cuda.device_count(),
                               "kernel_ms_per_step_rank0": {k: v[0] / args.steps for k, v in kt.items()},
                               "kernel_ms_note": "device time of rank 0's kernels, from extra steps outside the timed region"}
        if auto is not None:
            line["auto_plan"] = auto
        else:
            line["auto_plan"] = "same as the named plan"
        if world == 1 and not args.no_extras:
            R = S = out = res = None                      # the headline inputs are done with: HBM back for the other configs
            eng.release_workspace()
            torch.cuda.empty_cache()
            line["other_configs"] = dict(early, **extras(eng, torch, dev, args.steps, args.extras))
        if world == 1 and args.cpu_sample > 0:
            line["cpu_baseline"] = cpu_baseline(args.cpu_sample)
            line["gpu_over_cpu"] = value / line["cpu_baseline"]["value"]
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if not ok:
        raise SystemExit("VERIFICATION FAILED: pair set differs from the closed form")


if __name__ == "__main__":
    main()
