"""GPU suite: the multi-GPU stage entry points of include/rhj.h (rhj_shard_stats / _split / _partition / _join) in ONE
process: `world` ranks are emulated by running the sender calls once per shard, doing the all-to-all with numpy slices on
the host, and running the receiver calls once per owner -- every kernel an 8-GPU job runs (class histogram with rowID range,
class split into the 12-byte wire format, segmented fused two-pass partition with sender tags, tagged bucket joins) against
the CPU oracle's join of the GLOBAL relations.  RowIDs are global and exceed 2^32 (the tags must resolve them); the real
collectives are covered by tests/test_gpu_sharded.py and tests/test_sharded_gloo.py."""
import numpy as np
import pytest

from oracle.pyoracle import TUPLE
from radixhashjoin_amd import Engine, Opts
from radixhashjoin_amd.binding import (SHARD_GLOBAL16, SHARD_PLAIN, SHARD_TAGGED, mix64, narrow_bytes, narrow_key_offset, shard_plan,
                                       unmix64)
from radixhashjoin_amd.sharded import balanced_cuts

pytestmark = pytest.mark.gpu
SHIFT, BITS = 20, 8
BKT, CT, CT_HALF = 0, 2, 3


def few_partitions(values, nlow):
    """payloads whose MIXED value (rhj_mix64: what the engine partitions by) is join value << 16 | one of `nlow` 16-bit
    patterns chosen by the value: large partitions under a 16-bit plan"""
    lows = np.random.default_rng(nlow).permutation(1 << 16)[:nlow].astype(np.uint64)
    return unmix64((values << np.uint64(16)) | lows[(values % np.uint64(nlow)).astype(np.int64)])


def sharded_join(eng, shardsR, shardsS, plan, mode):
    """the schedule of radixhashjoin_amd/sharded.py with the collectives replaced by host slicing; returns all pairs"""
    world = len(shardsR)
    C = 1 << BITS
    sent = {0: [], 1: []}                # per side: per sender (payloads sorted by class, local rowIDs, class histogram, base)
    for side, shards in ((0, shardsR), (1, shardsS)):
        for t in shards:
            n = len(t)
            d = eng.to_device(t)
            hist, kmin, kmax = eng.shard_stats(side, d, n, SHIFT, BITS)
            dig = ((mix64(t["payload"]) >> np.uint64(SHIFT)) & np.uint64(C - 1)).astype(np.int64)       # classes: bits of the mixed value
            assert np.array_equal(hist, np.bincount(dig, minlength=C))
            assert (kmin, kmax) == (int(t["key"].min()), int(t["key"].max()))
            buf = eng.alloc(max(narrow_bytes(n), 16))
            starts = eng.alloc(8 * (C + 1))
            if mode == SHARD_PLAIN:
                kmin = 0                                                     # rowIDs travel as they are
            eng.shard_split(side, d, n, SHIFT, BITS, kmin, buf, starts)
            raw = buf.to_numpy(np.uint8, narrow_bytes(n))
            P = raw[:8 * n].view(np.uint64).copy()
            K = raw[narrow_key_offset(n):narrow_key_offset(n) + 4 * n].view(np.uint32).copy()
            st = starts.to_numpy(np.uint64, C + 1)
            assert np.array_equal(st, np.concatenate([[0], np.cumsum(hist)]).astype(np.uint64))
            # the split is a stable-enough class partition of exactly the shard's tuples, rowIDs local to the shard
            for c in (0, C // 2, C - 1):
                seg = slice(int(st[c]), int(st[c + 1]))
                assert np.all(((P[seg] >> np.uint64(SHIFT)) & np.uint64(C - 1)) == c)
            a = np.sort(np.stack([P, K.astype(np.uint64) + np.uint64(kmin)], axis=1).view([("p", "<u8"), ("k", "<u8")]).ravel(), order=["k"])
            b = np.sort(np.stack([mix64(t["payload"]), t["key"]], axis=1).view([("p", "<u8"), ("k", "<u8")]).ravel(), order=["k"])   # the wire carries the mixed value
            assert np.array_equal(a, b)
            sent[side].append((P, K, st, kmin))
            for x in (d, buf, starts):
                x.free()
    total = np.zeros(C, dtype=np.int64)
    for side in (0, 1):
        for (_, _, st, _) in sent[side]:
            total += np.diff(st.astype(np.int64))
    cuts = balanced_cuts(total.tolist(), world)
    pairs = []
    for owner in range(world):
        lo, hi = cuts[owner], cuts[owner + 1]
        for side in (0, 1):
            Ps, Ks, off = [], [], [0]
            for (P, K, st, _) in sent[side]:
                a, b = int(st[lo]), int(st[hi])
                Ps.append(P[a:b]); Ks.append(K[a:b]); off.append(off[-1] + b - a)
            m = off[-1]
            dP = eng.to_device(np.concatenate(Ps) if m else np.zeros(1, dtype=np.uint64))
            dK = eng.to_device(np.concatenate(Ks) if m else np.zeros(1, dtype=np.uint32))
            eng.shard_partition(side, dP, dK, m, off, [x[3] for x in sent[side]], plan, mode)
            eng.sync()
            dP.free(); dK.free()
        cnt = eng.shard_join(None, 0)                                       # count only
        out = eng.alloc(16 * max(cnt, 1))
        assert eng.shard_join(out, cnt) == cnt
        pairs.append(out.to_numpy(np.uint64, 2 * cnt).reshape(-1, 2))
        out.free()
    return np.concatenate(pairs)


def global_relations(rng, world, n_per, nlow, dup, stride=5 << 30):
    """R, S as lists of shards; rowIDs global, shard r's in [r * stride, ...): with the default beyond 2^32 from rank 1 on"""
    nglob = n_per * world
    vals = rng.permutation(1 << 24)[:max(nglob // dup, 1)].astype(np.uint64)
    rv = vals[rng.integers(0, len(vals), nglob)] if dup > 1 else vals[:nglob]
    sv = vals[rng.integers(0, len(vals), nglob)]
    sv[::53] ^= np.uint64(1 << 45)                                            # some foreign keys match nothing
    def shards(v):
        out = []
        for r in range(world):
            t = np.empty(n_per, dtype=TUPLE)
            t["key"] = rng.permutation(n_per).astype(np.uint64) + np.uint64(r * stride + 12345)
            pv = v[r * n_per:(r + 1) * n_per]
            t["payload"] = few_partitions(pv, nlow) if nlow else pv * np.uint64(0x9E3779B97F4A7C15)
            out.append(t)
        return out
    return shards(rv), shards(sv)


@pytest.mark.parametrize("world,n_per,nlow,dup,plan,kernel,mode", [
    (3, 60_000, 0, 1, Opts(2, 4, 4), BKT, SHARD_TAGGED),           # one-table kernel resolving sender tags; 3 ranks
    (2, 50_000, 0, 3, Opts(2, 5, 3), BKT, SHARD_TAGGED),           # duplicates on both sides
    (8, 20_000, 4, 1, Opts(2, 8, 8), CT, SHARD_GLOBAL16),          # compact-table kernel, 8 ranks, 40 K-tuple partitions (chunks, split tasks)
    (4, 30_000, 12, 2, Opts(2, 8, 8), CT, SHARD_GLOBAL16),         # duplicates: the generic (wavefront, slot) loop
    (2, 40_000, 9, 1, Opts(2, 8, 8), CT_HALF, SHARD_GLOBAL16),     # half-size compact table
    (5, 9_000, 3, 1, Opts(2, 8, 8), CT, SHARD_GLOBAL16),
    (3, 40_000, 0, 2, Opts(2, 6, 6), BKT, SHARD_GLOBAL16),         # 16-byte final partitions into the one-table kernel works too
    (4, 30_000, 5, 1, Opts(2, 8, 8), CT, SHARD_PLAIN),             # rowIDs below 2^32: narrow end to end, nothing to restore
    (3, 50_000, 0, 2, Opts(2, 5, 5), BKT, SHARD_PLAIN),
    (3, 40_000, 4, 1, Opts(2, 8, 9), CT, SHARD_PLAIN),             # 17- and 18-bit local plans (receivers beyond 1.1 * 10^9): PLAIN only
    (2, 30_000, 0, 2, Opts(2, 9, 9), BKT, SHARD_PLAIN),
])
def test_shard_stage_calls_equal_global_join(oracle, world, n_per, nlow, dup, plan, kernel, mode):
    rng = np.random.default_rng(world * 1000 + n_per)
    Rs, Ss = global_relations(rng, world, n_per, nlow, dup, stride=(1 << 30) if mode == SHARD_PLAIN else (5 << 30))
    suggested, rplan = shard_plan(n_per, n_per, plan)
    assert suggested == (SHARD_PLAIN if plan.bits1 + plan.bits2 > 16 else SHARD_TAGGED)    # (sizes this small: one table per partition)
    eng = Engine(0)
    try:
        if kernel != BKT:
            eng.set_option("join.big_tables", 1)
            eng.set_option("join.big_kernel", kernel)
        got = sharded_join(eng, Rs, Ss, rplan, mode)
        assert eng.info("last.join_kernel") == kernel
    finally:
        eng.close()
    exp = oracle.join(np.concatenate(Rs), np.concatenate(Ss))
    assert len(got) == len(exp)
    a = got[np.lexsort((got[:, 1], got[:, 0]))]
    e = np.stack([exp["keyR"], exp["keyS"]], axis=1)
    e = e[np.lexsort((e[:, 1], e[:, 0]))]
    assert np.array_equal(a, e)
    assert (got[:, 0].max() >= (1 << 32)) == (mode != SHARD_PLAIN)          # (else) the receiver really had to restore wide rowIDs


def test_shard_split_refuses_a_base_that_does_not_fit(engine):
    t = np.zeros(2000, dtype=TUPLE)
    t["key"] = np.arange(2000, dtype=np.uint64) * np.uint64(1 << 22)         # span 2^33
    d = engine.to_device(t)
    _, kmin, kmax = engine.shard_stats(0, d, len(t), SHIFT, BITS)
    assert kmax - kmin >= (1 << 32)
    buf = engine.alloc(narrow_bytes(len(t)))
    from radixhashjoin_amd.binding import RhjError
    with pytest.raises(RhjError):
        engine.shard_split(0, d, len(t), SHIFT, BITS, kmin, buf)
    with pytest.raises(RhjError):
        engine.shard_split(0, d, len(t), SHIFT, BITS, kmin + 1, buf)          # base above the smallest rowID
    d.free(); buf.free()


def test_lopsided_segments(oracle):
    """every owner receives (nearly) everything from ONE sender and nothing from the others: empty sender segments, empty
    pass-1 units, sizes far from n / world"""
    world, n_per = 4, 50_000
    rng = np.random.default_rng(77)
    Rs, Ss = [], []
    for r in range(world):
        # class = bits [20, 28) of the mixed payload = value bits [4, 12): shard r only has classes [64 r, 64 r + 64)
        hi = rng.permutation(1 << 12)[:n_per // 8].astype(np.uint64)                     # value bits [12, 24)
        vals = (hi[rng.integers(0, len(hi), n_per)] << np.uint64(12)) | (np.uint64(64 * r) + rng.integers(0, 64, n_per).astype(np.uint64)) << np.uint64(4) \
            | rng.integers(0, 16, n_per).astype(np.uint64)
        for dst, v in ((Rs, vals), (Ss, vals[rng.permutation(n_per)])):
            t = np.empty(n_per, dtype=TUPLE)
            t["key"] = rng.permutation(n_per).astype(np.uint64) + np.uint64(r * (5 << 30))
            t["payload"] = unmix64((v << np.uint64(16)) | np.uint64(0xBEEF))
            dst.append(t)
    mode, plan = shard_plan(n_per, n_per, Opts(2, 8, 8))
    eng = Engine(0)
    try:
        got = sharded_join(eng, Rs, Ss, plan, SHARD_TAGGED)
    finally:
        eng.close()
    exp = oracle.join(np.concatenate(Rs), np.concatenate(Ss))
    assert len(got) == len(exp)
    a = got[np.lexsort((got[:, 1], got[:, 0]))]
    e = np.stack([exp["keyR"], exp["keyS"]], axis=1)
    assert np.array_equal(a, e[np.lexsort((e[:, 1], e[:, 0]))])


def test_split_of_another_relation_than_stats_saw_is_reported(engine):
    """rhj_shard_split checks key_base against the rowID range rhj_shard_stats found; a DIFFERENT relation handed to the split
    may still hold a rowID that does not fit 32 bits from key_base: detected on the device, reported by this context's
    rhj_shard_join (the tuples sent are incomplete)"""
    from radixhashjoin_amd.binding import RhjError
    rng = np.random.default_rng(5)
    n = 40_000
    t = np.empty(n, dtype=TUPLE)
    t["key"] = rng.permutation(n).astype(np.uint64) + np.uint64(1000)
    t["payload"] = rng.integers(0, 1 << 40, n).astype(np.uint64)
    other = t.copy()
    other["key"][1234] = np.uint64(1000 + (1 << 33))
    mode, plan = shard_plan(n, n, Opts(2, 4, 4))
    for bad_side in (None, 0):
        bufs = []
        for side in (0, 1):
            d = engine.to_device(t)
            hist, kmin, kmax = engine.shard_stats(side, d, n, SHIFT, BITS)
            buf = engine.alloc(narrow_bytes(n))
            dsplit = engine.to_device(other) if side == bad_side else d
            engine.shard_split(side, dsplit, n, SHIFT, BITS, kmin, buf)
            raw = buf.to_numpy(np.uint8, narrow_bytes(n))
            bufs.append((raw[:8 * n].view(np.uint64).copy(), raw[narrow_key_offset(n):narrow_key_offset(n) + 4 * n].view(np.uint32).copy(), kmin))
        for side, (P, K, kmin) in enumerate(bufs):
            dP, dK = engine.to_device(P), engine.to_device(K)
            engine.shard_partition(side, dP, dK, n, [0, n], [kmin], plan, SHARD_TAGGED)
            engine.sync()
        if bad_side is None:
            assert engine.shard_join(None, 0) == n                      # (the relation joined with itself: unique payloads)
        else:
            with pytest.raises(RhjError, match="did not fit 32 bits"):
                engine.shard_join(None, 0)


def peer_split_join(eng, shardsR, shardsS, plan, mode, check_against_send_buffers=True):
    """the sharded schedule with rhj_shard_split_peer: every sender writes its classes straight into the owners' receive arrays
    (here: buffers of the one GPU standing in for the peers' HBM) -- no send buffer, no all-to-all -- then the owners run the
    receiver calls as usual.  Returns all pairs."""
    world, C = len(shardsR), 1 << BITS
    hists = {0: [], 1: []}
    kmins = {0: [], 1: []}
    for side, shards in ((0, shardsR), (1, shardsS)):
        for t in shards:
            d = eng.to_device(t)
            h, kmin, _ = eng.shard_stats(side, d, len(t), SHIFT, BITS)
            hists[side].append(h)
            kmins[side].append(0 if mode == SHARD_PLAIN else kmin)
            d.free()
    total = np.sum(hists[0], axis=0) + np.sum(hists[1], axis=0)
    cuts = balanced_cuts(total.tolist(), world)
    owner = np.zeros(C, dtype=np.uint8)
    for r in range(world):
        owner[cuts[r]:cuts[r + 1]] = r
    recvP, recvK, seg_off = {}, {}, {}
    for side in (0, 1):
        recvP[side], recvK[side], seg_off[side] = [], [], []
        for r in range(world):
            off = [0]
            for s in range(world):
                off.append(off[-1] + int(hists[side][s][cuts[r]:cuts[r + 1]].sum()))
            seg_off[side].append(off)
            recvP[side].append(eng.alloc(8 * max(off[-1], 1)))
            recvK[side].append(eng.alloc(4 * max(off[-1], 1)))
    for side, shards in ((0, shardsR), (1, shardsS)):
        for s, t in enumerate(shards):
            n = len(t)
            d = eng.to_device(t)
            eng.shard_stats(side, d, n, SHIFT, BITS)
            dst = np.zeros(C, dtype=np.uint64)
            for r in range(world):
                at = seg_off[side][r][s]
                for c in range(cuts[r], cuts[r + 1]):
                    dst[c] = at
                    at += int(hists[side][s][c])
            eng.shard_split_peer(side, d, n, SHIFT, BITS, kmins[side][s], owner, dst, recvP[side], recvK[side])
            eng.sync()
            if check_against_send_buffers:                   # the same tuples, at the same places, as send buffer + slicing delivers
                buf = eng.alloc(max(narrow_bytes(n), 16))
                eng.shard_split(side, d, n, SHIFT, BITS, kmins[side][s], buf)
                raw = buf.to_numpy(np.uint8, narrow_bytes(n))
                P = raw[:8 * n].view(np.uint64)
                K = raw[narrow_key_offset(n):narrow_key_offset(n) + 4 * n].view(np.uint32)
                st = np.concatenate([[0], np.cumsum(hists[side][s])]).astype(np.int64)
                for r in range(world):
                    a, b = int(st[cuts[r]]), int(st[cuts[r + 1]])
                    lo = seg_off[side][r][s]
                    m = seg_off[side][r][-1]
                    gotP = recvP[side][r].to_numpy(np.uint64, max(m, 1))[lo:lo + b - a]
                    gotK = recvK[side][r].to_numpy(np.uint32, max(m, 1))[lo:lo + b - a]
                    # class by class the same {payload, rowID} tuples (the order INSIDE a class is the order in which a tile's
                    # lanes won their LDS atomics: unspecified, and different from launch to launch)
                    for c in range(cuts[r], cuts[r + 1]):
                        x, y = int(st[c]) - a, int(st[c + 1]) - a
                        g = np.stack([gotP[x:y], gotK[x:y].astype(np.uint64)], axis=1)
                        w = np.stack([P[a + x:a + y], K[a + x:a + y].astype(np.uint64)], axis=1)
                        assert np.array_equal(g[np.lexsort((g[:, 1], g[:, 0]))], w[np.lexsort((w[:, 1], w[:, 0]))]), (side, s, r, c)
                buf.free()
            d.free()
    pairs = []
    for r in range(world):
        for side in (0, 1):
            m = seg_off[side][r][-1]
            eng.shard_partition(side, recvP[side][r], recvK[side][r], m, seg_off[side][r], kmins[side], plan, mode)
            eng.sync()
        cnt = eng.shard_join(None, 0)
        out = eng.alloc(16 * max(cnt, 1))
        assert eng.shard_join(out, cnt) == cnt
        pairs.append(out.to_numpy(np.uint64, 2 * cnt).reshape(-1, 2))
        out.free()
    for side in (0, 1):
        for b in recvP[side] + recvK[side]:
            b.free()
    return np.concatenate(pairs)


@pytest.mark.parametrize("world,n_per,nlow,dup,plan,kernel,mode", [
    (3, 60_000, 0, 1, Opts(2, 4, 4), BKT, SHARD_TAGGED),
    (8, 20_000, 4, 1, Opts(2, 8, 8), CT, SHARD_GLOBAL16),
    (4, 30_000, 5, 2, Opts(2, 8, 8), CT, SHARD_PLAIN),
    (2, 300_000, 0, 1, Opts(2, 6, 6), BKT, SHARD_PLAIN),          # several tiles per unit and several units per sender
])
def test_peer_mapped_class_split_equals_global_join(oracle, world, n_per, nlow, dup, plan, kernel, mode):
    """rhj_shard_split_peer (DESIGN §9 of round 3, built): sender kernels store into the owners' arrays through a table of
    per-destination base pointers; class by class the same tuples as send buffer + all-to-all delivers, and the global join's
    pair set"""
    rng = np.random.default_rng(world * 77 + n_per)
    Rs, Ss = global_relations(rng, world, n_per, nlow, dup, stride=(1 << 30) // 4 if mode == SHARD_PLAIN else (5 << 30))
    _, rplan = shard_plan(n_per, n_per, plan)
    eng = Engine(0)
    try:
        if kernel != BKT:
            eng.set_option("join.big_tables", 1)
            eng.set_option("join.big_kernel", kernel)
        got = peer_split_join(eng, Rs, Ss, rplan, mode)
    finally:
        eng.close()
    exp = oracle.join(np.concatenate(Rs), np.concatenate(Ss))
    assert len(got) == len(exp)
    a = got[np.lexsort((got[:, 1], got[:, 0]))]
    e = np.stack([exp["keyR"], exp["keyS"]], axis=1)
    assert np.array_equal(a, e[np.lexsort((e[:, 1], e[:, 0]))])
