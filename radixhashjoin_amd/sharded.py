"""Multi-GPU radix hash join: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

The reference has no distributed path (single process, pthreads).  Sharding:
both relations are range-sharded by row over the ranks (rowIDs stay global).  The join needs exactly
ONE exchange step, an all-to-all of tuples by OWNER CLASS:

    class(tuple) = (mix64(payload) >> owner_shift) & (C - 1)  C = 2^fine_bits classes, C >> world; mix64 = rhj_mix64, the
                                                              bijection the engine's joins partition by (include/rhj.h):
                                                              join values that share their raw bits [20,28) -- multiples
                                                              of 2^28, small dense keys -- still spread over all ranks
    owner(class) = the rank whose contiguous class range [cut[r], cut[r+1]) holds it

Schedule of the narrow path (include/rhj.h "multi-GPU stage entry points"; all compute in the C-ABI engine):

  1. rhj_shard_stats on the shards of R and S: class histogram + rowID range (one 16 B/tuple read each);
  2. ONE all_gather of {sizes, rowID ranges, the 2 x C class counts} (tiny).  From it every rank derives, identically,
     the class ranges -- balanced so that every rank receives about (|R|+|S|) / world tuples whatever the skew of
     the join values (a static `payload bits -> rank` map sends a Zipf foreign key's hot values, and everything
     hashed next to them, to one rank) --, the send / receive counts, and the local radix plan (from the LARGEST
     receive count, so that all ranks use the same bits);
  3. rhj_shard_split: the class split writes the NARROW WIRE FORMAT, payload 8 B + (rowID - shard's smallest rowID) 4 B
     in two arrays: 12 B/tuple cross xGMI instead of 16;
  4. all_to_all_single of the payload array and of the rowID array per relation (RCCL: world-1 direct peer sends
     over xGMI, all links busy at once, no multi-hop); R is on the wire while S is being split;
  5. rhj_shard_partition: the local fused two-pass partition of what arrived (one 8 B/tuple histogram read, two
     scatter passes); pass-1 units are cut at the sender segments of the receive buffer, so pass 2 knows every tuple's
     sender and restores global rowIDs (rowIDs all below 2^32: nothing to restore; else a sender tag in dead payload
     bits for the one-table join, or 16-byte tuples with global rowIDs for the compact-table join: include/rhj.h);
     the partitioning of R overlaps the transfer of S;
  6. rhj_shard_join: the bucket join.
Why the class split is NOT pass 1 of the local plan: two 8-bit passes separate 16 bits in all, and the
ownership of a tuple uses up log2(world) of them -- 8 x 10^9 tuples need 19 bits, three passes, wherever the exchange sits.

Everything is ordered on ONE stream (the engine is bound to torch's current stream; RCCL work objects make that
stream wait): no host synchronisation except the count all_gather (sizes must be known on the host) and the result
count.  Results stay sharded: rank d holds the pairs whose join value belongs to one of its classes; the global
result is the disjoint union (counts add up, no reduction).

Fallback (16-byte tuples on the wire; `rhj_shard_plan` says 0, or a shard's rowIDs span 2^32 or more, or more than 16
ranks): rhj_owner_histogram + rhj_owner_split (class of the mixed value, tuples unchanged) -> all-to-all -> rhj_join_dev
locally.

The engine is duck-typed so the exchange logic can be tested on CPU ranks with the gloo backend
(tests/test_sharded_gloo.py) and the whole path with the real engine and several ranks on one GPU
(tests/test_gpu_sharded.py, tests/test_gpu_bench_multirank.py).
"""
import os

import numpy as np
import torch
import torch.distributed as dist

OWNER_SHIFT_DEFAULT = 20      # above the 2 x 10 bits a local two-pass plan can use (PART_MAX_BITS = 10)
MAX_NARROW_WORLD = 16         # sender tags are 4 bits (include/rhj.h)


def balanced_cuts(weights, world):
    """Contiguous class ranges of near-equal weight: cut[r] = first class of rank r, cut[world] = len(weights).
    Greedy on the prefix sums: rank r ends at the class boundary closest to (r+1)/world of the total.  Every rank
    computes this from the same gathered histogram, so all agree without another message."""
    n = len(weights)
    total = sum(weights)
    cuts, acc, c = [0], 0, 0
    for r in range(1, world):
        target = total * r / world
        while c < n and acc + weights[c] / 2 <= target:     # take class c if its midpoint lies before the target
            acc += weights[c]
            c += 1
        c = max(c, cuts[-1])                                  # ranges may be empty, never reversed
        cuts.append(c)
    cuts.append(n)
    return cuts


def _i64(x):
    """uint64 value -> the int64 with the same bits (torch has no uint64 collectives)"""
    x = int(x) & ((1 << 64) - 1)
    return x - (1 << 64) if x >= (1 << 63) else x


def _u64(x):
    return int(x) & ((1 << 64) - 1)


class ShardedJoin:
    def __init__(self, engine, group=None, local_opts=None, owner_shift=OWNER_SHIFT_DEFAULT, fine_bits=None,
                 balance=True, narrow=True, bind_stream=True, force_exchange=False, rowid_mode=None):
        self.engine = engine
        self.group = group if group is not None else dist.group.WORLD
        self.world = dist.get_world_size(self.group)
        self.rank = dist.get_rank(self.group)
        if fine_bits is None:                # 256 classes: >= 32 per rank up to 8 ranks; the narrow split takes <= 8 bits
            fine_bits = 8
        if (1 << fine_bits) < self.world or owner_shift + fine_bits > 64:
            raise ValueError("need at least one owner class per rank inside the 64 payload bits")
        self.fine_bits = fine_bits
        self.nclasses = 1 << fine_bits
        self.owner_shift = owner_shift
        self.balance = balance               # False: equal-width class ranges (a static radix map)
        self.cuts = [self.nclasses * r // self.world for r in range(self.world + 1)]
        self.local_opts = local_opts
        self.narrow = narrow                 # False: always exchange 16-byte tuples
        self.bind_stream = bind_stream       # the engine launches on torch's current stream (no fences needed)
        self.rowid_mode = rowid_mode         # tests: SHARD_TAGGED / SHARD_GLOBAL16 instead of what rhj_shard_plan suggests for the sizes
        self.force_exchange = force_exchange # world == 1: run the whole schedule anyway (collectives with oneself): the
        #                                      one-GPU box's way of executing the RCCL code path of a multi-GPU job
        # no single message of the exchange above this many bytes (see _a2a); RHJ_SHARD_MAX_MSG overrides (tests: small values)
        self.max_msg_bytes = int(os.environ.get("RHJ_SHARD_MAX_MSG", 512 << 20))
        self.collect_timings = False         # True: sum the engine's per-kernel HIP-event timings over the calls of a join
        #                                      (synchronises after every engine call: for profiling steps, not timed ones)
        self.kernel_ms = {}
        self.stats = {}
        self._bound = None

    def owner_of(self, payload):
        """rank that owns the join values `payload` (numpy uint64 array) under the class ranges of the last join"""
        from .binding import mix64
        cls = ((mix64(payload) >> np.uint64(self.owner_shift)) & np.uint64(self.nclasses - 1)).astype(np.int64)
        return np.searchsorted(np.asarray(self.cuts[1:], dtype=np.int64), cls, side="right")

    # ------------------------------------------------------------------------------------------------
    def join(self, R, nR, S, nS, out=None):
        """Local shards in ([n,2] int64 tensors of {rowID, payload}), local share of the result out:
        (count, [count,2] tensor of {rowR,rowS}, global rowIDs)."""
        dev = R.device
        self._bind(dev)
        if self.world == 1 and not self.force_exchange:
            return self._local_join_whole(R, nR, S, nS, out)
        eng = self.engine
        can_narrow = self.narrow and hasattr(eng, "shard_stats") and self.world <= MAX_NARROW_WORLD
        if can_narrow:
            self._fence_torch(dev)                     # (fences are no-ops once the engine shares torch's stream, see _bind)
            histR, minR, maxR = eng.shard_stats(0, R, nR, self.owner_shift, self.fine_bits)
            self._collect()
            histS, minS, maxS = eng.shard_stats(1, S, nS, self.owner_shift, self.fine_bits)
            self._collect()
        else:
            histR, histS = self._class_histogram(R, nR), self._class_histogram(S, nS)
            minR = maxR = minS = maxS = 0
        meta = self._gather_meta([nR, nS, _i64(minR), _i64(maxR), _i64(minS), _i64(maxS), 1 if can_narrow else 0], histR, histS)
        (inR, outR), (inS, outS) = self._plan_exchange(meta)
        mR, mS = sum(outR), sum(outS)
        self.stats = {"recv_R": mR, "recv_S": mS, "cuts": list(self.cuts), "format": "tuple16"}
        # one decision for all ranks, from gathered numbers only
        head = meta["head"]
        spanok = all(_u64(head[r][3]) - _u64(head[r][2]) < (1 << 32) and _u64(head[r][5]) - _u64(head[r][4]) < (1 << 32)
                     for r in range(self.world))
        narrow = all(head[r][6] for r in range(self.world)) and spanok
        plan, mode = None, 0
        if narrow:
            from .binding import SHARD_PLAIN, shard_plan
            mode, plan = shard_plan(max(meta["recvR"]), max(meta["recvS"]), self.local_opts)
            # rowIDs that are all below 2^32 travel as they are (key_base 0 on every rank): the receiver has nothing to restore
            small = all(_u64(head[r][3]) < (1 << 32) and _u64(head[r][5]) < (1 << 32) for r in range(self.world))
            if mode == SHARD_PLAIN and not small:
                mode = 0                               # a 17-18-bit local plan cannot restore rowIDs: 16-byte fallback
            elif mode and self.rowid_mode and mode != SHARD_PLAIN:
                mode = self.rowid_mode
            elif mode and small:
                mode = SHARD_PLAIN
        if not mode:
            return self._join_tuple16(R, nR, S, nS, inR, outR, inS, outS, out)

        from .binding import narrow_bytes, narrow_key_offset
        plain = mode == SHARD_PLAIN
        row0R = [0 if plain else _u64(head[r][2]) for r in range(self.world)]
        row0S = [0 if plain else _u64(head[r][4]) for r in range(self.world)]
        self.stats["format"] = "narrow12"
        self.stats["rowid_mode"] = {1: "tagged", 2: "global16", 3: "plain"}[mode]
        self.stats["kept_local"] = inR[self.rank] + inS[self.rank]
        self.stats["exchange_bytes_sent"] = 12 * ((nR - inR[self.rank]) + (nS - inS[self.rank]))
        self.stats["plan"] = (plan.passes, plan.bits1, plan.bits2)

        def split_and_send(side, rel, n, key_base, in_splits, out_splits):
            buf = torch.empty(max(narrow_bytes(n), 16), dtype=torch.uint8, device=dev)
            self._fence_torch(dev)
            eng.shard_split(side, rel, n, self.owner_shift, self.fine_bits, key_base, buf)
            self._fence_engine()
            koff = narrow_key_offset(n)
            P = buf[:8 * n].view(torch.int64)
            K = buf[koff:koff + 4 * n].view(torch.int32)
            m = sum(out_splits)
            rP = torch.empty(max(m, 1), dtype=torch.int64, device=dev)
            rK = torch.empty(max(m, 1), dtype=torch.int32, device=dev)
            w1 = self._a2a(rP[:m], P, out_splits, in_splits, async_op=True)
            w2 = self._a2a(rK[:m], K, out_splits, in_splits, async_op=True)
            return rP, rK, m, (w1, w2), buf          # buf must stay alive until the transfer has finished

        hR = split_and_send(0, R, nR, row0R[self.rank], inR, outR)      # R on the wire ...
        hS = split_and_send(1, S, nS, row0S[self.rank], inS, outS)      # ... while S is being split

        def seg_offsets(out_splits):
            off = [0]
            for c in out_splits:
                off.append(off[-1] + c)
            return off

        for w in hR[3]:
            if w is not None:
                w.wait()                               # (stream-level wait: the host runs on)
        self._fence_torch(dev)
        eng.shard_partition(0, hR[0], hR[1], mR, seg_offsets(outR), row0R, plan, mode)   # R's local passes overlap the S transfer
        self._collect()
        for w in hS[3]:
            if w is not None:
                w.wait()
        self._fence_torch(dev)
        eng.shard_partition(1, hS[0], hS[1], mS, seg_offsets(outS), row0S, plan, mode)
        self._collect()
        cap = out.shape[0] if out is not None else max(mR, mS) + 1024
        if out is None:
            out = torch.empty((cap, 2), dtype=torch.int64, device=dev)
        cnt = eng.shard_join(out, cap, allow_overflow=True)
        if cnt > cap:                                  # more pairs than guessed: the exact size is known now
            out = torch.empty((cnt, 2), dtype=torch.int64, device=dev)
            cnt = eng.shard_join(out, cnt)
        self._collect()
        return cnt, out

    # -- fallback: 16-byte tuples on the wire ------------------------------------------------------------
    def _join_tuple16(self, R, nR, S, nS, inR, outR, inS, outS, out):
        eng, dev = self.engine, R.device
        self.stats["exchange_bytes_sent"] = 16 * ((nR - inR[self.rank]) + (nS - inS[self.rank]))

        def split_and_send(rel, n, in_splits, out_splits):
            staged = torch.empty((max(n, 1), 2), dtype=torch.int64, device=dev)
            bounds = torch.empty(self.nclasses + 1, dtype=torch.int64, device=dev)
            self._fence_torch(dev)
            eng.owner_split(rel, n, self.owner_shift, self.fine_bits, staged, bounds)
            self._fence_engine()
            m = sum(out_splits)
            recv = torch.empty((max(m, 1), 2), dtype=torch.int64, device=dev)
            work = self._a2a(recv[:m], staged[:n], out_splits, in_splits, async_op=True)
            return recv, m, work, staged

        hR = split_and_send(R, nR, inR, outR)
        hS = split_and_send(S, nS, inS, outS)
        for h in (hR, hS):
            if h[2] is not None:
                h[2].wait()
        return self._local_join_whole(hR[0], hR[1], hS[0], hS[1], out)

    def _class_histogram(self, rel, n):
        hist = torch.empty(self.nclasses, dtype=torch.int64, device=rel.device)
        self._fence_torch(rel.device)
        self.engine.owner_histogram(rel, n, self.owner_shift, self.fine_bits, hist)
        self._fence_engine()
        return hist.cpu().numpy()

    # -- step 2: count matrix -> class ranges and exchange sizes -----------------------------------------
    def _gather_meta(self, head, histR, histS):
        """ONE all_gather of this rank's {head words, 2 x C class counts}; returns the gathered numbers (host)."""
        mine = torch.tensor(list(head) + [int(x) for x in histR] + [int(x) for x in histS], dtype=torch.int64)
        on_dev = dist.get_backend(self.group) == "nccl"
        if on_dev:
            mine = mine.to(torch.device("cuda", torch.cuda.current_device()))
        allv = torch.empty(self.world * mine.numel(), dtype=torch.int64, device=mine.device)
        dist.all_gather_into_tensor(allv, mine, group=self.group)
        allv = allv.cpu().view(self.world, -1)                               # host sync: sizes must be known
        nh, C = len(head), self.nclasses
        return {"head": [[int(x) for x in allv[r, :nh]] for r in range(self.world)],
                "hist": allv[:, nh:].reshape(self.world, 2, C)}

    def _plan_exchange(self, meta):
        """Sets self.cuts; returns, per relation, (in_splits, out_splits): tuples this rank sends to / receives from every
        rank; meta gets "recvR" / "recvS": what EVERY rank receives (the same lists on all ranks)."""
        allh = meta["hist"]
        if self.balance:
            self.cuts = balanced_cuts(allh.sum(dim=(0, 1)).tolist(), self.world)
        cuts = self.cuts
        splits = []
        for rel in (0, 1):
            send = [int(allh[self.rank, rel, cuts[d]:cuts[d + 1]].sum()) for d in range(self.world)]
            recv = [int(allh[src, rel, cuts[self.rank]:cuts[self.rank + 1]].sum()) for src in range(self.world)]
            splits.append((send, recv))
            meta["recvR" if rel == 0 else "recvS"] = [int(allh[:, rel, cuts[d]:cuts[d + 1]].sum()) for d in range(self.world)]
        # the longest segment any rank sends to any rank (rows): the number of rounds of the exchange must be the same everywhere
        self._max_seg = max(int(allh[src, rel, cuts[d]:cuts[d + 1]].sum())
                            for src in range(self.world) for rel in (0, 1) for d in range(self.world))
        return splits

    def _local_join_whole(self, Rx, mR, Sx, mS, out):
        cap = out.shape[0] if out is not None else max(mR, mS) + 1024
        if out is None:
            out = torch.empty((cap, 2), dtype=torch.int64, device=Rx.device)
        self._fence_torch(Rx.device)              # the received tuples have landed (collective complete)
        cnt = self.engine.join_dev(Rx, mR, Sx, mS, out, cap, opts=self.local_opts, allow_overflow=True)
        if cnt > cap:                              # more pairs than guessed: exact size is known now
            out = torch.empty((cnt, 2), dtype=torch.int64, device=Rx.device)
            cnt = self.engine.join_dev(Rx, mR, Sx, mS, out, cnt, opts=self.local_opts)
        self._collect()
        return cnt, out

    # -- ordering ------------------------------------------------------------------------------------------
    def _bind(self, dev):
        """Put the engine on torch's current stream: torch ops, RCCL hand-offs (work.wait()) and the engine's kernels are
        then ordered by the stream itself and the path has no host-side fence."""
        if dev.type != "cuda" or not self.bind_stream or not hasattr(self.engine, "set_stream"):
            return
        cur = torch.cuda.current_stream(dev).cuda_stream
        if cur == 0:                     # torch's default stream has no handle to hand over: the engine keeps its own stream
            self._bound = None           # and the hand-offs are fenced on the host (tests; bench.py sets a real stream)
            return
        if self._bound != cur:
            self.engine.set_stream(cur)
            self._bound = cur

    def _fence_torch(self, dev):
        if dev.type == "cuda" and self._bound is None:
            torch.cuda.current_stream(dev).synchronize()

    def _fence_engine(self):
        if self._bound is None:
            sync = getattr(self.engine, "sync", None)
            if sync is not None:
                sync()
        self._collect()

    def _collect(self):
        if not self.collect_timings or not hasattr(self.engine, "timings"):
            return
        t = self.engine.timings()            # of the engine call that just finished (synchronises its stream)
        for k, v in t.items():
            if isinstance(v, dict):
                acc = self.kernel_ms.setdefault(k, [0.0, 0])
                acc[0] += v["ms"]
                acc[1] += v["launches"]

    class _Works:
        """the work objects of one exchange (several collectives when it was cut into rounds)"""
        def __init__(self, works):
            self.works = [w for w in works if w is not None]

        def wait(self):
            for w in self.works:
                w.wait()

    def _a2a(self, out, inp, out_splits, in_splits, async_op=False):
        """The all-to-all of one array (rows of `inp` in destination order, `in_splits[d]` rows to rank d; `out_splits[r]` rows
        from rank r land in `out` in source order).

        * What this rank keeps (1 / world of the tuples) never enters the collective: a device copy on the current stream.
        * No message is larger than `max_msg_bytes` (default 512 MiB): larger segments go in rounds, each round one
          `all_to_all` over views of the two arrays.  [measured, this image] RCCL's copy of a rank's own segment is wrong above
          about 1 GiB (torch's bundled RCCL: `all_to_all_single` of 1.6 GB on one rank returns at once with other bytes) or never
          finishes (system RCCL 2.27.7: grouped `ncclSend` + `ncclRecv` to self of 1.6 GB); peer segments of 10^9-row shards are
          1 - 4 GB (`tools/ab`-style probe: `tests/test_gpu_sharded.py::test_rccl_exchange_of_a_large_self_segment`).
        * With a backend that cannot move device memory (gloo rehearsal of the multi-rank path on a single-GPU box) the
          rounds are staged through host memory."""
        rank, world = self.rank, self.world
        row_bytes = inp.element_size()
        for dim in inp.shape[1:]:
            row_bytes *= int(dim)
        # world == 1 with force_exchange (the one-GPU way of executing the RCCL path): the own segment goes through the
        # collective after all, in rounds like a peer's
        via_self = world == 1 and self.force_exchange
        in_off = [0]
        for c in in_splits:
            in_off.append(in_off[-1] + c)
        out_off = [0]
        for c in out_splits:
            out_off.append(out_off[-1] + c)
        n_self = in_splits[rank]
        assert n_self == out_splits[rank]
        if n_self and not via_self:
            out[out_off[rank]:out_off[rank] + n_self].copy_(inp[in_off[rank]:in_off[rank] + n_self], non_blocking=True)
        if world == 1 and not via_self:
            return None
        step = max(1, int(self.max_msg_bytes) // max(row_bytes, 1))
        peer = lambda d: d != rank or via_self
        longest = self._max_seg                             # the same number of rounds on every rank (a round may be empty here)
        gloo = dist.get_backend(self.group) == "gloo"       # no list all_to_all, no device memory: contiguous host pieces
        works = []
        for lo in range(0, max(longest, 1), step):
            def piece(t, off, cnt, d):
                a = min(lo, cnt) if peer(d) else 0
                b = min(lo + step, cnt) if peer(d) else 0
                return t[off + a:off + b]
            ins = [piece(inp, in_off[d], in_splits[d], d) for d in range(world)]
            outs = [piece(out, out_off[r], out_splits[r], r) for r in range(world)]
            if gloo:
                h_in = torch.cat([t.cpu() for t in ins])
                h_out = torch.empty((sum(t.shape[0] for t in outs),) + tuple(out.shape[1:]), dtype=out.dtype)
                dist.all_to_all_single(h_out, h_in, [t.shape[0] for t in outs], [t.shape[0] for t in ins], group=self.group)
                at = 0
                for t in outs:
                    if t.shape[0]:
                        t.copy_(h_out[at:at + t.shape[0]])
                    at += t.shape[0]
            else:
                works.append(dist.all_to_all(outs, [t.contiguous() for t in ins], group=self.group, async_op=async_op))
        return self._Works(works) if async_op and works else None

    def transport(self):
        """what moves the tuples: for bench.py's config.exchange"""
        b = dist.get_backend(self.group)
        return "RCCL all-to-all over xGMI" if b == "nccl" else f"{b} all-to-all staged through host memory (single-GPU rehearsal)"
