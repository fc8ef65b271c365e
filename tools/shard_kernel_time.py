#!/usr/bin/env python3
"""Per-rank kernel time of the sharded join WITHOUT the contention of a multi-rank rehearsal on one GPU: one process plays
rank 0 of `world` ranks (the other ranks' shards are generated and split too, untimed, so that rank 0 receives real
segments), the all-to-all is a device-side copy, and rank 0's engine calls are timed with the engine's HIP events.
Compared with rhj_join_dev on the same number of tuples (one GPU joining what one rank joins).

    python tools/shard_kernel_time.py [--world 2] [--tuples 200000000] [--dist uniform|zipf]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=2)
    ap.add_argument("--tuples", type=int, default=200_000_000)
    ap.add_argument("--dist", default="uniform")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--plain", action="store_true", help="rowIDs below 2^32 (world x tuples < 2^32): nothing to restore at the receiver")
    a = ap.parse_args()
    import numpy as np
    import torch
    import radixhashjoin_amd as rhj
    from radixhashjoin_amd.binding import GEN_R, GEN_S_UNIFORM, GEN_S_ZIPF, narrow_bytes, narrow_key_offset, shard_plan
    from radixhashjoin_amd.sharded import balanced_cuts

    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    eng = rhj.Engine(0)
    eng.set_stream(stream.cuda_stream)
    eng.set_profiling(True)
    W, n, SHIFT, BITS = a.world, a.tuples, 20, 8
    nglob = n * W
    acc = {}

    def collect(tag):
        t = eng.timings()
        ms = sum(v["ms"] for v in t.values() if isinstance(v, dict))
        acc[tag] = acc.get(tag, 0.0) + ms

    def gen(rank):
        R = torch.empty((n, 2), dtype=torch.int64, device=dev)
        S = torch.empty((n, 2), dtype=torch.int64, device=dev)
        eng.generate(GEN_R, R, n, row0=rank * n, D=nglob)
        eng.generate(GEN_S_ZIPF if a.dist == "zipf" else GEN_S_UNIFORM, S, n, row0=rank * n, D=nglob, seed=42, theta_milli=900)
        return R, S

    results = []
    for rep in range(a.reps + 1):
        acc.clear()
        sent = {0: [], 1: []}
        hists = []
        for rank in range(W):
            R, S = gen(rank)
            for side, rel in ((0, R), (1, S)):
                hist, kmin, kmax = eng.shard_stats(side, rel, n, SHIFT, BITS)
                if rank == 0:
                    collect("stats")
                hists.append(hist)
                buf = torch.empty(narrow_bytes(n), dtype=torch.uint8, device=dev)
                if a.plain:
                    kmin = 0
                eng.shard_split(side, rel, n, SHIFT, BITS, kmin, buf)
                if rank == 0:
                    collect("split")
                else:
                    eng.sync()
                st = np.concatenate([[0], np.cumsum(hist)])
                sent[side].append((buf, st, kmin))
            del R, S
        cuts = balanced_cuts(np.sum(hists, axis=0).tolist(), W)
        lo, hi = cuts[0], cuts[1]
        recv = {}
        for side in (0, 1):
            Ps, Ks, off = [], [], [0]
            for buf, st, _ in sent[side]:
                x, y = int(st[lo]), int(st[hi])
                koff = narrow_key_offset(n)
                Ps.append(buf[8 * x:8 * y].view(torch.int64))
                Ks.append(buf[koff + 4 * x:koff + 4 * y].view(torch.int32))
                off.append(off[-1] + y - x)
            recv[side] = (torch.cat(Ps), torch.cat(Ks), off)
        row0 = {s: [x[2] for x in sent[s]] for s in (0, 1)}
        del sent
        mR, mS = recv[0][2][-1], recv[1][2][-1]
        mode, plan = shard_plan(mR, mS, None)
        assert mode, (mR, mS, plan)
        if a.plain:
            mode = 3
        for side in (0, 1):
            eng.shard_partition(side, recv[side][0], recv[side][1], recv[side][2][-1], recv[side][2], row0[side], plan, mode)
            collect("partition")
        out = torch.empty((max(mR, mS) + 1024, 2), dtype=torch.int64, device=dev)
        cnt = eng.shard_join(out, out.shape[0])
        collect("join")
        kern = eng.info("last.join_kernel")
        del recv, out
        if rep:
            results.append(dict(acc))
    # one GPU joining as many tuples as one rank joins
    R, S = torch.empty((n, 2), dtype=torch.int64, device=dev), torch.empty((n, 2), dtype=torch.int64, device=dev)
    eng.generate(GEN_R, R, n, D=n)
    eng.generate(GEN_S_ZIPF if a.dist == "zipf" else GEN_S_UNIFORM, S, n, D=n, seed=42, theta_milli=900)
    out = torch.empty((n + 1024, 2), dtype=torch.int64, device=dev)
    single = []
    for rep in range(a.reps + 1):
        eng.join_dev(R, n, S, n, out, out.shape[0])
        t = eng.timings()
        if rep:
            single.append(sum(v["ms"] for v in t.values() if isinstance(v, dict)))
    med = lambda xs: sorted(xs)[len(xs) // 2]
    shard = {k: med([r[k] for r in results]) for k in results[0]}
    total = sum(shard.values())
    print(json.dumps({"world": W, "tuples_per_rank": n, "dist": a.dist, "rank0_kernel_ms": shard, "rank0_kernel_ms_total": total,
                      "join_kernel": kern, "rowid_mode": {1: "tagged", 2: "global16", 3: "plain"}[mode], "recv": [mR, mS], "plan": [plan.passes, plan.bits1, plan.bits2],
                      "single_gpu_join_dev_kernel_ms": med(single), "ratio": total / med(single), "matches_rank0": cnt}))


if __name__ == "__main__":
    main()
