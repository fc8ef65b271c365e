"""GPU suite: the multi-GPU driver (radixhashjoin_amd/sharded.py) with the REAL engine.

A gpurun box has one MI355X, so world_size 2 (and 3: class ranges need no power of two) run as ranks that share
cuda:0, with gloo moving the exchange through the host (RHJ_BENCH_BACKEND=gloo does the same for bench.py).  Everything
except the transport is what an 8-GPU job runs: rhj_shard_stats, count matrix, rhj_shard_split into the 12-byte wire
format, uneven all_to_all_single of payloads and rowIDs, rhj_shard_partition of what arrived, rhj_shard_join (cases with a
two-pass local plan), or the 16-byte fallback rhj_partition_at / all-to-all / rhj_join_dev (the others).  The union of the
ranks' pair sets must equal the CPU oracle's join of the global relations."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(rank, world, port, n_per_rank, D, zipf, opts, q, shift=0, rowid_mode=None):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import radixhashjoin_amd as rhj
    from radixhashjoin_amd.binding import GEN_R, GEN_S_UNIFORM, GEN_S_ZIPF
    from radixhashjoin_amd.sharded import ShardedJoin
    from oracle.pyoracle import Oracle, TUPLE
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    eng = rhj.Engine(0)                                   # fails loudly without the HIP library / GPU
    nglob = n_per_rank * world
    R = torch.empty((n_per_rank, 2), dtype=torch.int64, device=dev)
    S = torch.empty((n_per_rank, 2), dtype=torch.int64, device=dev)
    eng.generate(GEN_R, R, n_per_rank, row0=rank * n_per_rank + shift, D=D)                  # shift: rowIDs beyond 2^32
    eng.generate(GEN_S_ZIPF if zipf else GEN_S_UNIFORM, S, n_per_rank, row0=rank * n_per_rank + shift, D=D, seed=42, theta_milli=zipf or 0)
    eng.sync()
    sj = ShardedJoin(eng, dist.group.WORLD, local_opts=rhj.Opts(*opts) if opts else None, rowid_mode=rowid_mode)
    cnt, out = sj.join(R, n_per_rank, S, n_per_rank)
    torch.cuda.synchronize()
    pairs = out[:cnt].cpu().numpy().view(np.uint64)
    shards = [None] * world
    dist.all_gather_object(shards, (R.cpu().numpy().view(np.uint64), S.cpu().numpy().view(np.uint64), pairs,
                                    sj.stats["recv_R"] + sj.stats["recv_S"], sj.stats["format"] + ":" + str(sj.stats.get("rowid_mode"))))
    if rank == 0:
        o = Oracle()
        def tup(parts):
            a = np.concatenate(parts)
            t = np.empty(len(a), dtype=TUPLE)
            t["key"], t["payload"] = a[:, 0], a[:, 1]
            return t
        Rg, Sg = tup([s[0] for s in shards]), tup([s[1] for s in shards])
        assert np.array_equal(Rg["key"], np.arange(nglob, dtype=np.uint64) + np.uint64(shift))    # rowIDs stay global
        exp = o.join(Rg, Sg)
        allp = np.concatenate([s[2] for s in shards])
        a = allp[np.lexsort((allp[:, 1], allp[:, 0]))]
        e = np.stack([exp["keyR"], exp["keyS"]], axis=1)
        e = e[np.lexsort((e[:, 1], e[:, 0]))]
        recv = [s[3] for s in shards]
        q.put((len(allp), len(exp), bool(np.array_equal(a, e)), max(recv) / (sum(recv) / world), shards[0][4]))
    dist.barrier()
    eng.close()
    dist.destroy_process_group()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,n_per_rank,D,zipf,opts,fmt,shift,mode", [
    (2, 1_500_000, 3_000_000, 0, None, "tuple16:None", 0, None),           # PK/FK, automatic local plan (one pass: 16-byte fallback)
    (2, 400_000, 100_000, 0, (2, 4, 4), "narrow12:plain", 0, None),        # duplicates, forced two-pass plan: 12-byte wire format
    (3, 300_000, 900_000, 1250, None, "tuple16:None", 0, None),            # skew, 3 ranks
    (3, 500_000, 1_500_000, 1250, (2, 6, 6), "narrow12:plain", 0, None),   # skew, 3 ranks, narrow
    (2, 3_000_000, 6_000_000, 0, (2, 8, 8), "narrow12:plain", 0, None),    # the 8+8 plan of the large configurations
    (3, 400_000, 300_000, 0, (2, 5, 5), "narrow12:tagged", 1 << 33, None),   # rowIDs beyond 2^32: sender tags
    (2, 500_000, 1_000_000, 0, (2, 8, 8), "narrow12:global16", 1 << 34, 2)])  # ... or 16-byte final partitions (what 8 x 10^9 rows use)
def test_sharded_join_real_engine(world, n_per_rank, D, zipf, opts, fmt, shift, mode):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, n_per_rank, D, zipf, opts, q, shift, mode)) for r in range(world)]
    for p in procs:
        p.start()
    got, exp, same, imbalance, used = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert got == exp and same
    assert used == fmt
    if zipf:
        assert imbalance <= 1.3, imbalance


def rccl_worker(port, n, D, opts, mode_rows, q):
    """ONE rank over the real RCCL backend ("nccl"): the whole sharded schedule with the collectives addressed to oneself --
    device all_gather of the count matrix, asynchronous all_to_all_single of int64 payloads and int32 rowIDs, work.wait()
    on the shared stream -- i.e. the torch.distributed / RCCL calls an 8-GPU job makes, on the one GPU a box has."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    import radixhashjoin_amd as rhj
    from radixhashjoin_amd.binding import GEN_R, GEN_S_UNIFORM
    from radixhashjoin_amd.sharded import ShardedJoin
    from oracle.pyoracle import Oracle, TUPLE
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    eng = rhj.Engine(0)
    eng.set_stream(stream.cuda_stream)
    R = torch.empty((n, 2), dtype=torch.int64, device=dev)
    S = torch.empty((n, 2), dtype=torch.int64, device=dev)
    eng.generate(GEN_R, R, n, row0=mode_rows, D=D)
    eng.generate(GEN_S_UNIFORM, S, n, row0=mode_rows, D=D, seed=42)
    sj = ShardedJoin(eng, dist.group.WORLD, local_opts=rhj.Opts(*opts), force_exchange=True)
    cnt, out = sj.join(R, n, S, n)
    torch.cuda.synchronize()
    pairs = out[:cnt].cpu().numpy().view(np.uint64)
    o = Oracle()
    def tup(t):
        a = t.cpu().numpy().view(np.uint64)
        x = np.empty(len(a), dtype=TUPLE)
        x["key"], x["payload"] = a[:, 0], a[:, 1]
        return x
    exp = o.join(tup(R), tup(S))
    a = pairs[np.lexsort((pairs[:, 1], pairs[:, 0]))]
    e = np.stack([exp["keyR"], exp["keyS"]], axis=1)
    e = e[np.lexsort((e[:, 1], e[:, 0]))]
    q.put((len(pairs), len(exp), bool(np.array_equal(a, e)), sj.stats["format"], sj.stats.get("rowid_mode"), sj.transport()))
    eng.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("n,D,opts,row0,mode", [(600_000, 200_000, (2, 5, 5), 0, "plain"),             # rowIDs < 2^32
                                                (600_000, 600_000, (2, 6, 5), 1 << 33, "tagged")])   # rowIDs beyond: sender tags
def test_sharded_schedule_over_rccl_single_rank(n, D, opts, row0, mode):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=rccl_worker, args=(free_port(), n, D, opts, row0, q))
    p.start()
    got, exp, same, fmt, rowid_mode, transport = q.get(timeout=600)
    p.join(timeout=120)
    assert p.exitcode == 0
    assert got == exp and same
    assert fmt == "narrow12" and rowid_mode == mode and "RCCL" in transport


def rccl_worker_big(port, n, max_msg, q):
    """one rank over RCCL with a segment larger than one message may be (sharded.py _a2a): verified by count + checksum"""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    import radixhashjoin_amd as rhj
    from radixhashjoin_amd.binding import GEN_R, GEN_S_UNIFORM
    from radixhashjoin_amd.sharded import ShardedJoin
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    eng = rhj.Engine(0)
    eng.set_stream(stream.cuda_stream)
    R = torch.empty((n, 2), dtype=torch.int64, device=dev)
    S = torch.empty((n, 2), dtype=torch.int64, device=dev)
    eng.generate(GEN_R, R, n, row0=0, D=n)
    eng.generate(GEN_S_UNIFORM, S, n, row0=0, D=n, seed=42)
    exp = eng.expected_pkfk(S, n)
    sj = ShardedJoin(eng, dist.group.WORLD, force_exchange=True)
    if max_msg:
        sj.max_msg_bytes = max_msg
    cnt, out = sj.join(R, n, S, n)
    torch.cuda.synchronize()
    got = (cnt, eng.pairs_checksum(out, cnt))
    q.put((got, exp, sj.stats["format"], sj.stats.get("rowid_mode"), sj.transport()))
    eng.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("n,max_msg", [(100_000_000, 0), (30_000_000, 16 << 20)])
def test_rccl_exchange_of_a_large_self_segment(n, max_msg):
    """10^8 rows = 800 MB of payloads to oneself: two rounds of at most 512 MiB per message (a single 1.6 GB message to
    oneself comes back wrong from this image's RCCL, see sharded.py _a2a); 3 x 10^7 rows in rounds of 16 MiB"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=rccl_worker_big, args=(free_port(), n, max_msg, q))
    p.start()
    got, exp, fmt, rowid_mode, transport = q.get(timeout=600)
    p.join(timeout=120)
    assert p.exitcode == 0
    assert tuple(got) == tuple(exp) and got[0] == n
    assert fmt == "narrow12" and rowid_mode == "plain" and "RCCL" in transport
