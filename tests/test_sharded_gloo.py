"""CPU suite, world_size 2 and 4 over gloo: the multi-GPU driver's exchange logic
(radixhashjoin_amd/sharded.py).  The ranks have no GPU here, so the engine is replaced by a
test double whose two operations are restated with numpy / the CPU oracle; what is under test is
the owner split, the count + tuple all-to-all and the sharded-result contract."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleEngine:
    """test double for radixhashjoin_amd.Engine on CPU tensors (duck-typed: partition_at, join_dev)"""

    def __init__(self):
        from oracle.pyoracle import Oracle
        self.o = Oracle()

    @staticmethod
    def _np(t, n):
        return t.numpy()[:n]

    def partition_at(self, d_in, n, shift, bits, d_out, d_part_start):
        a = self._np(d_in, n)
        dig = ((a[:, 1].astype(np.uint64) >> np.uint64(shift)) & np.uint64((1 << bits) - 1)).astype(np.int64)
        order = np.argsort(dig, kind="stable")
        d_out.numpy()[:n] = a[order]
        d_part_start.numpy()[:] = np.concatenate([[0], np.cumsum(np.bincount(dig, minlength=1 << bits))])

    def histogram(self, d_rel, n, shift, bits, d_hist):
        a = self._np(d_rel, n)
        dig = ((a[:, 1].astype(np.uint64) >> np.uint64(shift)) & np.uint64((1 << bits) - 1)).astype(np.int64)
        d_hist.numpy()[:] = np.bincount(dig, minlength=1 << bits)

    # the owner classes are bits of rhj_mix64(payload) (include/rhj.h); the 16-byte split leaves the tuples as they are
    def owner_split(self, d_in, n, shift, bits, d_out, d_class_start):
        from radixhashjoin_amd.binding import mix64
        a = self._np(d_in, n)
        dig = ((mix64(a[:, 1].astype(np.uint64)) >> np.uint64(shift)) & np.uint64((1 << bits) - 1)).astype(np.int64)
        order = np.argsort(dig, kind="stable")
        d_out.numpy()[:n] = a[order]
        d_class_start.numpy()[:] = np.concatenate([[0], np.cumsum(np.bincount(dig, minlength=1 << bits))])

    def owner_histogram(self, d_rel, n, shift, bits, d_hist):
        from radixhashjoin_amd.binding import mix64
        a = self._np(d_rel, n)
        dig = ((mix64(a[:, 1].astype(np.uint64)) >> np.uint64(shift)) & np.uint64((1 << bits) - 1)).astype(np.int64)
        d_hist.numpy()[:] = np.bincount(dig, minlength=1 << bits)

    def partition(self, d_in, n, bits1, bits2, d_out, d_part_start):
        # layout documented in include/rhj.h: pass-1 digit major, pass-2 digit minor
        a = self._np(d_in, n)
        p = (a[:, 1].astype(np.uint64) & np.uint64((1 << (bits1 + bits2)) - 1)).astype(np.int64)
        oid = ((p & ((1 << bits1) - 1)) << bits2) | (p >> bits1)
        order = np.argsort(oid, kind="stable")
        d_out.numpy()[:n] = a[order]
        d_part_start.numpy()[:] = np.concatenate([[0], np.cumsum(np.bincount(oid, minlength=1 << (bits1 + bits2)))])

    def bucket_join(self, d_Rp, d_startR, d_Sp, d_startS, nparts, radix_bits, d_out=None, capacity=0, probe_split=0,
                    allow_overflow=False):
        nR, nS = int(d_startR.numpy()[nparts]), int(d_startS.numpy()[nparts])
        # partitions must be aligned for a bucket join: same digit at the same index on both sides
        for t, st in ((d_Rp, d_startR), (d_Sp, d_startS)):
            a, b = self._np(t, nR if t is d_Rp else nS), st.numpy()
            low = (a[:, 1].astype(np.uint64) & np.uint64((1 << radix_bits) - 1)).astype(np.int64)
            for k in (0, nparts // 2, nparts - 1):
                seg = low[b[k]:b[k + 1]]
                assert len(np.unique(seg)) <= 1
        return self.join_dev(d_Rp, nR, d_Sp, nS, d_out, capacity)

    # ---- the narrow wire format (include/rhj.h "multi-GPU stage entry points"), restated with numpy ----------------
    def shard_stats(self, side, d_rel, n, shift, bits):
        from radixhashjoin_amd.binding import mix64
        a = self._np(d_rel, n).view(np.uint64)
        dig = ((mix64(a[:, 1]) >> np.uint64(shift)) & np.uint64((1 << bits) - 1)).astype(np.int64)
        self._dig = getattr(self, "_dig", {})
        self._dig[side] = dig
        return (np.bincount(dig, minlength=1 << bits).astype(np.int64), int(a[:, 0].min()) if n else 0,
                int(a[:, 0].max()) if n else 0)

    def shard_split(self, side, d_rel, n, shift, bits, key_base, d_narrow_out, d_class_start=None):
        from radixhashjoin_amd.binding import mix64, narrow_key_offset
        a = self._np(d_rel, n).view(np.uint64)
        order = np.argsort(self._dig[side], kind="stable")
        local = a[order, 0] - np.uint64(key_base)
        assert np.all(local < np.uint64(1 << 32))
        buf = d_narrow_out.numpy()
        buf[:8 * n] = np.ascontiguousarray(mix64(a[order, 1])).view(np.uint8)      # the wire carries the mixed value
        koff = narrow_key_offset(n)
        buf[koff:koff + 4 * n] = local.astype(np.uint32).view(np.uint8)

    def shard_partition(self, side, d_payloads, d_rowids, m, seg_off, row0, plan, mode):
        assert plan.passes == 2 and seg_off[0] == 0 and seg_off[-1] == m and mode in (1, 2, 3)
        assert mode != 3 or not any(row0)
        sender = np.repeat(np.arange(len(seg_off) - 1), np.diff(np.asarray(seg_off, dtype=np.int64)))
        self._recv = getattr(self, "_recv", {})
        self._recv[side] = (d_payloads.numpy()[:m].view(np.uint64).copy(), d_rowids.numpy()[:m].view(np.uint32).copy(), sender, row0)

    def shard_join(self, d_out=None, capacity=0, allow_overflow=False):
        from oracle.pyoracle import TUPLE
        rel = []
        for side in (0, 1):
            P, K, sender, row0 = self._recv[side]
            t = np.empty(len(P), dtype=TUPLE)
            t["payload"] = P
            t["key"] = np.asarray(row0, dtype=np.uint64)[sender] + K.astype(np.uint64)       # global rowID = sender's base + local
            rel.append(t)
        p = self.o.join(rel[0], rel[1])
        k = min(len(p), capacity)
        if d_out is not None and k:
            d_out.numpy()[:k, 0] = p["keyR"][:k].view(np.int64)
            d_out.numpy()[:k, 1] = p["keyS"][:k].view(np.int64)
        return len(p)

    def join_dev(self, d_R, nR, d_S, nS, d_out=None, capacity=0, opts=None, allow_overflow=False):
        from oracle.pyoracle import TUPLE
        R = np.ascontiguousarray(self._np(d_R, nR)).view(np.uint64).reshape(-1, 2)
        S = np.ascontiguousarray(self._np(d_S, nS)).view(np.uint64).reshape(-1, 2)
        Rt = np.empty(nR, dtype=TUPLE); Rt["key"], Rt["payload"] = R[:, 0], R[:, 1]
        St = np.empty(nS, dtype=TUPLE); St["key"], St["payload"] = S[:, 0], S[:, 1]
        p = self.o.join(Rt, St)
        k = min(len(p), capacity)
        if d_out is not None and k:
            d_out.numpy()[:k, 0] = p["keyR"][:k].view(np.int64)
            d_out.numpy()[:k, 1] = p["keyS"][:k].view(np.int64)
        return len(p)


def zipf_payloads(o, n, D, theta, seed):
    """Zipf(theta) ranks over [1, D] by inverse CDF of the continuous approximation, mapped through mix() like
    the device generator (rhj_generate_dev kind 2); numpy only, the same on every rank"""
    rng = np.random.default_rng(seed)
    e = 1.0 - theta
    u = rng.random(n)
    r = np.clip(np.power(1.0 + u * (np.power(D + 1.0, e) - 1.0), 1.0 / e).astype(np.int64), 1, D)
    lut = np.array([o.mix(int(k)) for k in range(D + 1)], dtype=np.uint64)
    return lut[r]


def worker(rank, world, port, n_per_rank, dup, narrow, q, zipf=None, balance=True, aligned=0):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle.pyoracle import Oracle
    from radixhashjoin_amd.sharded import ShardedJoin
    o = Oracle()
    nglob = n_per_rank * world
    D = max(nglob // dup, 1)
    Rg, Sg = o.gen_R(nglob, D), o.gen_S_counter(nglob, D, 42)       # global relations, rows range-sharded
    if zipf is not None:
        Sg["payload"] = zipf_payloads(o, nglob, D, zipf, 7)         # skewed foreign key, same array on every rank
    if aligned:                                                     # join values k << aligned: raw bits [20,28) all zero
        from radixhashjoin_amd.binding import unmix64
        for t in (Rg, Sg):
            t["payload"] = unmix64(t["payload"]) << np.uint64(aligned)
    lo, hi = rank * n_per_rank, (rank + 1) * n_per_rank

    def shard(t):
        a = np.empty((hi - lo, 2), dtype=np.uint64)
        a[:, 0], a[:, 1] = t["key"][lo:hi], t["payload"][lo:hi]
        return torch.from_numpy(a.view(np.int64))

    # narrow: the 12-byte wire format (needs a two-pass local plan: forced here, the sizes are tiny); else 16-byte tuples
    from radixhashjoin_amd import Opts
    sj = ShardedJoin(OracleEngine(), dist.group.WORLD, balance=balance, narrow=narrow, local_opts=Opts(2, 4, 4) if narrow else None)
    if os.environ.get("RHJ_TEST_MAX_MSG"):                          # the exchange in rounds of at most this many bytes per message
        sj.max_msg_bytes = int(os.environ["RHJ_TEST_MAX_MSG"])
    cnt, out = sj.join(shard(Rg), n_per_rank, shard(Sg), n_per_rank)
    if narrow and n_per_rank >= 4_000:
        assert sj.stats["format"] == "narrow12", sj.stats
        assert sj.stats["exchange_bytes_sent"] == 12 * (2 * n_per_rank - sj.stats["kept_local"]), sj.stats
    if not narrow:
        assert sj.stats["format"] == "tuple16"
    pairs = out.numpy()[:cnt].view(np.uint64)
    # every pair this rank produced belongs to its owner class
    assert np.all(sj.owner_of(Rg["payload"][pairs[:, 0].astype(np.int64)]) == rank)
    gathered, received = [None] * world, [None] * world
    dist.all_gather_object(gathered, pairs)
    dist.all_gather_object(received, sj.stats["recv_R"] + sj.stats["recv_S"])
    sj.stats["received_per_rank"] = received
    if rank == 0:
        allp = np.concatenate(gathered)
        exp = o.join(Rg, Sg)
        a = allp[np.lexsort((allp[:, 1], allp[:, 0]))]
        e = np.stack([exp["keyR"], exp["keyS"]], axis=1)
        e = e[np.lexsort((e[:, 1], e[:, 0]))]
        q.put((len(allp), len(exp), bool(np.array_equal(a, e)), sj.stats))
    dist.barrier()
    dist.destroy_process_group()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_world(world, *args):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port) + args[:3] + (q,) + args[3:]) for r in range(world)]
    for p in procs:
        p.start()
    got, exp, same, stats = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got == exp and same
    return stats


@pytest.mark.parametrize("world,n_per_rank,dup,narrow", [(2, 20_000, 1, True), (2, 5_000, 4, False), (4, 6_000, 2, True),
                                                         (2, 1_000, 1, True), (3, 4_000, 1, True), (3, 4_000, 3, False)])
def test_sharded_join_equals_global_join(world, n_per_rank, dup, narrow):
    run_world(world, n_per_rank, dup, narrow)


@pytest.mark.parametrize("world,n_per_rank,narrow,max_msg", [(2, 20_000, True, 4096), (3, 4_000, False, 1000), (4, 6_000, True, 64)])
def test_exchange_in_rounds(world, n_per_rank, narrow, max_msg, monkeypatch):
    """no message of the exchange above max_msg_bytes (sharded.py _a2a: RCCL's limits on this image): segments larger than
    that travel in several rounds, the own segment by a local copy -- same pair set"""
    monkeypatch.setenv("RHJ_TEST_MAX_MSG", str(max_msg))
    run_world(world, n_per_rank, 1, narrow)


@pytest.mark.parametrize("narrow", [True, False])
def test_aligned_join_values_spread_over_ranks(narrow):
    """join values that are multiples of 2^28 (raw owner bits [20,28) all zero: with classes taken from the raw payload every
    tuple went to rank 0, and balanced_cuts cannot split a class): classes are bits of rhj_mix64(payload), so every rank
    receives its share even with equal-width class ranges -- and the pair set is the global join's"""
    stats = run_world(4, 6_000, 1, narrow, None, False, 28)
    r = stats["received_per_rank"]
    assert max(r) / (sum(r) / len(r)) <= 1.15, r


def test_skewed_join_values_are_balanced_over_ranks():
    """Zipf(1.4) foreign key: the hottest join value alone is ~a third of S.  Equal-width class ranges (a static
    radix map) overload the rank that owns it; ranges cut from the all-gathered class histogram keep every rank
    within 1.3x of the mean, with the same pair set."""
    world, n = 4, 12_000

    def imbalance(stats):
        r = stats["received_per_rank"]
        return max(r) / (sum(r) / len(r))

    static = imbalance(run_world(world, n, 1, True, 1.4, False))
    balanced = imbalance(run_world(world, n, 1, True, 1.4, True))
    assert static > 1.3, static                    # the input really is skewed enough to matter
    assert balanced <= 1.3, balanced
