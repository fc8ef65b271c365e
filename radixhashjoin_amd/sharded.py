"""Multi-GPU radix hash join: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

The reference has no distributed path (single process, pthreads).  Sharding:
both relations are range-sharded by row over the ranks (rowIDs stay global).  The join needs exactly
ONE exchange step, an all-to-all of tuples by OWNER RADIX BITS:

    owner(tuple) = (payload >> owner_shift) & (world - 1)        (world is a power of two)

  1. every rank splits its shard of R (then S) by owner bits with one scatter-partition pass
     (rhj_partition_at): tuples for rank d become contiguous, d = 0..world-1;
  2. a tiny all-to-all of the per-destination counts, then ONE all_to_all_single of the tuples
     (RCCL: world-1 direct peer sends over xGMI, all links busy at once, no multi-hop); the
     exchange of R is in flight while S is being split, the exchange of S while the received R is
     radix-partitioned locally;
  3. every rank now holds ALL tuples of its owner class of both relations and runs the normal
     single-GPU join (rhj_join_dev: 1-2 radix passes on the low bits + LDS bucket join) locally.
     Owner bits lie above every bit the local plan can use (2 * 10), so the local plan is untouched.
Results stay sharded: rank d holds the pairs whose join value belongs to owner class d; the global
result is the disjoint union (counts add up, no reduction).

All compute runs in the C-ABI engine passed in (HIP kernels); torch supplies device memory and the
collective.  The engine is duck-typed (partition_at, join_dev) so the exchange logic can be tested
on CPU ranks with the gloo backend (tests/test_sharded_gloo.py).
"""
import torch
import torch.distributed as dist

OWNER_SHIFT_DEFAULT = 20      # above the 2 x 10 bits a local two-pass plan can use (PART_MAX_BITS = 10)


class ShardedJoin:
    def __init__(self, engine, group=None, local_opts=None, owner_shift=OWNER_SHIFT_DEFAULT):
        self.engine = engine
        self.group = group if group is not None else dist.group.WORLD
        self.world = dist.get_world_size(self.group)
        self.rank = dist.get_rank(self.group)
        if self.world & (self.world - 1):
            raise ValueError("world size must be a power of two (owner = radix bits)")
        self.owner_bits = self.world.bit_length() - 1
        self.owner_shift = owner_shift
        self.local_opts = local_opts
        self.staged_local_join = True        # False: one rhj_join_dev call after both exchanges (no S-transfer overlap)
        self.collect_timings = False         # True: sum the engine's per-kernel HIP-event timings over the calls of a join
        self.kernel_ms = {}
        self.stats = {}

    # -- step 1: owner split of one shard (compute) ------------------------------------------------
    def split(self, rel, n):
        """One scatter-partition pass by owner bits.  Returns (staged [n,2] tensor with each destination's
        tuples contiguous, per-destination counts as a python list)."""
        dev = rel.device
        staged = torch.empty((max(n, 1), 2), dtype=torch.int64, device=dev)
        bounds = torch.empty(self.world + 1, dtype=torch.int64, device=dev)
        self._fence_torch(dev)                    # rel / buffers produced by torch ops are complete
        self.engine.partition_at(rel, n, self.owner_shift, self.owner_bits, staged, bounds)
        self._fence_engine()                      # staged / bounds complete before torch and RCCL touch them
        return staged, (bounds[1:] - bounds[:-1]).contiguous()

    # -- step 2: count exchange + tuple exchange (communication) ----------------------------------
    def start_exchange(self, staged, n, send_counts):
        """Launches the all-to-all of `staged`; returns a handle for finish_exchange.  With RCCL the
        transfer proceeds on the communicator's stream while the caller keeps launching kernels."""
        recv_counts = torch.empty_like(send_counts)
        self._a2a(recv_counts, send_counts, None, None)
        in_splits = send_counts.tolist()          # host sync: the exchange sizes must be known
        out_splits = recv_counts.tolist()
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

    def exchange(self, rel, n):
        """rel: [n,2] int64 tensor of {rowID, join value} on this rank's device.
        Returns (received [m,2] tensor, m): every tuple of the global relation owned by this rank."""
        if self.world == 1:
            return rel, n
        staged, counts = self.split(rel, n)
        return self.finish_exchange(self.start_exchange(staged, n, counts))

    def join(self, R, nR, S, nS, out=None):
        """Local shards in, local share of the result out: (count, [count,2] tensor of {rowR,rowS}).

        Schedule (communication on the RCCL stream, kernels on the engine's stream):
            split R | exchange R  ||  split S | exchange S  ||  partition received R | partition received S | bucket join
        i.e. the R transfer overlaps the owner split of S, and the S transfer overlaps the local radix
        partitioning of R (stage entry points rhj_partition / rhj_bucket_join of the C-ABI)."""
        if self.world == 1:
            return self._local_join_whole(R, nR, S, nS, out)
        stagedR, cR = self.split(R, nR)
        hR = self.start_exchange(stagedR, nR, cR)              # R tuples on the wire ...
        stagedS, cS = self.split(S, nS)                        # ... while S is being split
        hS = self.start_exchange(stagedS, nS, cS)
        mR, mS = hR[1], hS[1]                                  # received sizes are known from the count exchange
        self.stats = {"recv_R": mR, "recv_S": mS}
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
