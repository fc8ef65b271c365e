"""Multi-GPU radix hash join: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

The reference has no distributed path (single process, pthreads).  Sharding:
both relations are range-sharded by row over the ranks (rowIDs stay global).  The join needs exactly
ONE exchange step, an all-to-all of tuples by OWNER CLASS:

    class(tuple) = (payload >> owner_shift) & (C - 1)         C = 2^fine_bits classes, C >> world
    owner(class) = the rank whose contiguous class range [cut[r], cut[r+1]) holds it

  1. every rank splits its shards of R and S by class with one scatter-partition pass each
     (rhj_partition_at): the tuples of a class, hence of a destination, become contiguous; the pass
     also yields the shard's class histogram;
  2. ONE all_gather of the (world x 2 x C) count matrix (tiny).  From it every rank derives, identically,
     the class ranges: balanced so that every rank receives about (|R|+|S|) / world tuples whatever the skew
     of the join values (a static `payload bits -> rank` map sends a Zipf foreign key's hot values, and everything
     hashed next to them, to one rank) -- and the send / receive sizes of the tuple exchange;
  3. ONE all_to_all_single of the tuples per relation (RCCL: world-1 direct peer sends over xGMI, all links
     busy at once, no multi-hop); the exchange of S is in flight while the received R is radix-partitioned locally;
  4. every rank now holds ALL tuples of its classes of both relations and runs the normal
     single-GPU join (1-2 radix passes on the low bits + LDS bucket join) locally.
     Class bits lie above every bit the local plan can use (2 * 10), so the local plan is untouched.
Results stay sharded: rank d holds the pairs whose join value belongs to one of its classes; the global
result is the disjoint union (counts add up, no reduction).

All compute runs in the C-ABI engine passed in (HIP kernels); torch supplies device memory and the
collectives.  The engine is duck-typed (partition_at, partition, bucket_join, join_dev) so the exchange logic
can be tested on CPU ranks with the gloo backend (tests/test_sharded_gloo.py) and the whole path with the real
engine and several ranks on one GPU (tests/test_gpu_sharded.py).
"""
import torch
import torch.distributed as dist

OWNER_SHIFT_DEFAULT = 20      # above the 2 x 10 bits a local two-pass plan can use (PART_MAX_BITS = 10)


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


class ShardedJoin:
    def __init__(self, engine, group=None, local_opts=None, owner_shift=OWNER_SHIFT_DEFAULT, fine_bits=None,
                 balance=True):
        self.engine = engine
        self.group = group if group is not None else dist.group.WORLD
        self.world = dist.get_world_size(self.group)
        self.rank = dist.get_rank(self.group)
        if fine_bits is None:                # >= 32 classes per rank, within the full-rate range of the scatter kernel
            fine_bits = min(9, max(8, (self.world - 1).bit_length() + 5))
        if (1 << fine_bits) < self.world or owner_shift + fine_bits > 64:
            raise ValueError("need at least one owner class per rank inside the 64 payload bits")
        self.fine_bits = fine_bits
        self.nclasses = 1 << fine_bits
        self.owner_shift = owner_shift
        self.balance = balance               # False: equal-width class ranges (a static radix map)
        self.cuts = [self.nclasses * r // self.world for r in range(self.world + 1)]
        self.local_opts = local_opts
        self.staged_local_join = True        # False: one rhj_join_dev call after both exchanges (no S-transfer overlap)
        self.collect_timings = False         # True: sum the engine's per-kernel HIP-event timings over the calls of a join
        self.kernel_ms = {}
        self.stats = {}

    def owner_of(self, payload):
        """rank that owns the join values `payload` (numpy uint64 array) under the class ranges of the last join"""
        import numpy as np
        cls = ((payload >> np.uint64(self.owner_shift)) & np.uint64(self.nclasses - 1)).astype(np.int64)
        return np.searchsorted(np.asarray(self.cuts[1:], dtype=np.int64), cls, side="right")

    # -- step 1: class split of one shard (compute) -------------------------------------------------
    def split(self, rel, n):
        """One scatter-partition pass by owner class.  Returns (staged [n,2] tensor with each class's tuples
        contiguous, in class order; the shard's class histogram as an int64 tensor of nclasses)."""
        dev = rel.device
        staged = torch.empty((max(n, 1), 2), dtype=torch.int64, device=dev)
        bounds = torch.empty(self.nclasses + 1, dtype=torch.int64, device=dev)
        self._fence_torch(dev)                    # rel / buffers produced by torch ops are complete
        self.engine.partition_at(rel, n, self.owner_shift, self.fine_bits, staged, bounds)
        self._fence_engine()                      # staged / bounds complete before torch and RCCL touch them
        return staged, (bounds[1:] - bounds[:-1]).contiguous()

    def class_histogram(self, rel, n):
        """The shard's class histogram alone (rhj_histogram at the owner bits): int64 tensor of nclasses."""
        hist = torch.empty(self.nclasses, dtype=torch.int64, device=rel.device)
        self._fence_torch(rel.device)
        self.engine.histogram(rel, n, self.owner_shift, self.fine_bits, hist)
        self._fence_engine()
        return hist

    # -- step 2: count matrix -> class ranges and exchange sizes -----------------------------------
    def plan_exchange(self, histR, histS):
        """all_gather of this rank's (2 x C) class histogram.  Sets self.cuts; returns, per relation,
        (in_splits, out_splits): tuples this rank sends to / receives from every rank."""
        mine = torch.stack([histR, histS])                                   # [2, C]
        via_host = mine.is_cuda and dist.get_backend(self.group) == "gloo"   # rehearsal: several ranks on one GPU
        if via_host:
            mine = mine.cpu()
        mine = mine.reshape(-1).contiguous()
        allh = torch.empty(self.world * mine.numel(), dtype=mine.dtype, device=mine.device)
        dist.all_gather_into_tensor(allh, mine, group=self.group)
        allh = allh.cpu().view(self.world, 2, self.nclasses)                 # host sync: sizes must be known
        if self.balance:
            self.cuts = balanced_cuts(allh.sum(dim=(0, 1)).tolist(), self.world)
        lo, hi = self.cuts[self.rank], self.cuts[self.rank + 1]
        splits = []
        for rel in (0, 1):
            send = [int(allh[self.rank, rel, self.cuts[d]:self.cuts[d + 1]].sum()) for d in range(self.world)]
            recv = [int(allh[src, rel, lo:hi].sum()) for src in range(self.world)]
            splits.append((send, recv))
        return splits

    # -- step 3: tuple exchange (communication) ----------------------------------------------------
    def start_exchange(self, staged, n, in_splits, out_splits):
        """Launches the all-to-all of `staged`; returns a handle for finish_exchange.  With RCCL the
        transfer proceeds on the communicator's stream while the caller keeps launching kernels."""
        m = int(sum(out_splits))
        recv = torch.empty((max(m, 1), 2), dtype=torch.int64, device=staged.device)
        work = self._a2a(recv[:m], staged[:n], out_splits, in_splits, async_op=True)
        return recv, m, work, staged              # staged must stay alive until the transfer has finished

    @staticmethod
    def finish_exchange(handle):
        recv, m, work, _staged = handle
        if work is not None:
            work.wait()
        return recv, m

    def join(self, R, nR, S, nS, out=None):
        """Local shards in, local share of the result out: (count, [count,2] tensor of {rowR,rowS}).

        Schedule (communication on the RCCL stream, kernels on the engine's stream):
            class histogram of S | split R | count matrix | exchange R  ||  split S | exchange S  ||  partition received R |
            partition received S | bucket join
        i.e. the R transfer overlaps the class split of S and the S transfer the local radix partitioning of R (stage
        entry points rhj_histogram / rhj_partition_at / rhj_partition / rhj_bucket_join of the C-ABI).  The class ranges
        need the histogram of BOTH relations (the skewed one is usually S), hence the cheap histogram-only pass over S
        first (8 B/tuple algorithmic against 40 for the split)."""
        if self.world == 1:
            return self._local_join_whole(R, nR, S, nS, out)
        if hasattr(self.engine, "histogram"):
            hS_ = self.class_histogram(S, nS)
            stagedR, hR_ = self.split(R, nR)
            (inR, outR), (inS, outS) = self.plan_exchange(hR_, hS_)
            hR = self.start_exchange(stagedR, nR, inR, outR)   # R on the wire ...
            stagedS, _ = self.split(S, nS)                     # ... while S is being split
            hS = self.start_exchange(stagedS, nS, inS, outS)
        else:
            stagedR, hR_ = self.split(R, nR)
            stagedS, hS_ = self.split(S, nS)
            (inR, outR), (inS, outS) = self.plan_exchange(hR_, hS_)
            hR = self.start_exchange(stagedR, nR, inR, outR)   # both relations on the wire ...
            hS = self.start_exchange(stagedS, nS, inS, outS)
        mR, mS = hR[1], hS[1]
        self.stats = {"recv_R": mR, "recv_S": mS, "cuts": list(self.cuts)}
        if not self.staged_local_join or not hasattr(self.engine, "partition"):
            Rx, _ = self.finish_exchange(hR)
            Sx, _ = self.finish_exchange(hS)
            return self._local_join_whole(Rx, mR, Sx, mS, out)
        plan = self._plan(mR, mS)
        dev = R.device
        Rx, _ = self.finish_exchange(hR)
        hR = stagedR = None                                    # the send buffer of R can be recycled now
        if plan.passes == 0 or mR == 0 or mS == 0:
            Sx, _ = self.finish_exchange(hS)
            return self._local_join_whole(Rx, mR, Sx, mS, out)
        b1, b2 = plan.bits1, (plan.bits2 if plan.passes == 2 else 0)
        nparts = 1 << (b1 + b2)
        partR = torch.empty((max(mR, 1), 2), dtype=torch.int64, device=dev)
        psR = torch.empty(nparts + 1, dtype=torch.int64, device=dev)
        self._fence_torch(dev)                                 # R has landed; S is still in flight
        self.engine.partition(Rx, mR, b1, b2, partR, psR)      # local radix passes over R overlap the S transfer
        self._collect()
        Sx, _ = self.finish_exchange(hS)
        hS = stagedS = None
        partS = torch.empty((max(mS, 1), 2), dtype=torch.int64, device=dev)
        psS = torch.empty(nparts + 1, dtype=torch.int64, device=dev)
        self._fence_torch(dev)
        self.engine.partition(Sx, mS, b1, b2, partS, psS)
        self._collect()
        cap = out.shape[0] if out is not None else max(mR, mS) + 1024
        if out is None:
            out = torch.empty((cap, 2), dtype=torch.int64, device=dev)
        self._fence_torch(dev)
        cnt = self.engine.bucket_join(partR, psR, partS, psS, nparts, b1 + b2, out, cap,
                                      probe_split=plan.probe_split, allow_overflow=True)
        if cnt > cap:
            out = torch.empty((cnt, 2), dtype=torch.int64, device=dev)
            self._fence_torch(dev)
            cnt = self.engine.bucket_join(partR, psR, partS, psS, nparts, b1 + b2, out, cnt, probe_split=plan.probe_split)
        self._collect()
        return cnt, out

    def _plan(self, mR, mS):
        from .binding import plan as resolve
        return resolve(mR, mS, self.local_opts)

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

    # The engine launches on its own HIP stream unless it was given torch's (rhj_set_stream); these two
    # host-side fences make the hand-offs correct either way.  They cost microseconds per join.
    def _fence_torch(self, dev):
        if dev.type == "cuda":
            torch.cuda.current_stream(dev).synchronize()

    def _fence_engine(self):
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

    def _a2a(self, out, inp, out_splits, in_splits, async_op=False):
        """all_to_all_single; with a backend that cannot move device memory (gloo rehearsal of the
        multi-rank path on a single-GPU box) the payload is staged through host memory."""
        if inp.is_cuda and dist.get_backend(self.group) == "gloo":
            h_in, h_out = inp.cpu(), torch.empty(out.shape, dtype=out.dtype)
            dist.all_to_all_single(h_out, h_in, out_splits, in_splits, group=self.group)
            out.copy_(h_out)
            return None
        return dist.all_to_all_single(out, inp, out_splits, in_splits, group=self.group, async_op=async_op)
