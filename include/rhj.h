/*
 * rhj.h -- C-ABI of librhj_hip.so: the MI355X (gfx950) radix hash join engine.
 *
 * This is the drop-in boundary for the hot path of pelekoudasq/radixHashJoin,
 *     void Result::multiRadixHashJoin(JobScheduler&, relation&, relation&)   (Result.h:30, Result.cpp:90-124)
 * and for the job bodies it fans out (HistogramJob / PartitionJob / JoinJob, JobScheduler.cpp:149-192).
 * Plain pointers and sizes only; no HIP, torch or C++ types.  Every entry point cites the reference
 * interface it replaces.  The reference-side binding is shown in INTEGRATION.md; the C++ host mirror
 * of the reference surface that calls this ABI lives in radixhashjoin_amd/host/.
 *
 * Conventions
 *   - all functions return RHJ_OK (0) or a negative rhj_status; rhj_last_error(ctx) has the text.
 *   - "d_" arguments are DEVICE pointers (HBM), everything else is host memory.
 *   - a context owns one HIP stream and a grow-only HBM workspace.  A context is NOT thread-safe:
 *     create one per calling thread, exactly like each query thread of the reference owns a
 *     private JobScheduler (MainScheduler.cpp:6-14).  Different contexts may run concurrently.
 *   - all arithmetic is 64-bit unsigned integer; pair ORDER in outputs is unspecified (SURVEY §8a:
 *     no consumer of Result observes it); the pair MULTISET is bit-exact with the reference.
 */
#ifndef RHJ_H
#define RHJ_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RHJ_ABI_VERSION 3

/* layout-identical to `struct tuple` (structs.h:33-36): key = rowID, payload = join value */
typedef struct { uint64_t key; uint64_t payload; } rhj_tuple;
/* layout-identical to `struct key_tuple` (Result.h:9-12) */
typedef struct { uint64_t keyR; uint64_t keyS; } rhj_pair;

typedef struct rhj_ctx rhj_ctx;

typedef enum {
    RHJ_OK = 0,
    RHJ_E_INVALID = -1,    /* bad argument */
    RHJ_E_NODEVICE = -2,   /* no usable HIP device */
    RHJ_E_HIP = -3,        /* a HIP call failed (rhj_last_error has hipGetErrorString) */
    RHJ_E_NOMEM = -4,      /* HBM or host allocation failed */
    RHJ_E_OVERFLOW = -5    /* d_out too small: *out_count holds the exact size needed, pairs beyond capacity dropped */
} rhj_status;

/* Radix plan.  The reference hard-codes one 8-bit pass (HASH_LSB, Result.cpp:5,91); here the
 * number of passes and bits per pass are run-time knobs.  0/0 with passes=-1 = automatic:
 * no partitioning when the smaller input fits one LDS hash table, else the fewest bits such
 * that the average build partition fits one LDS table, split over at most two passes. */
/* WHICH bits.  Inside rhj_join / rhj_join_dev (and the multi-GPU stage calls) the radix digits are bits of
 *     h = rhj_mix64(payload),   a BIJECTIVE 64-bit mix (splitmix64's finaliser),
 * not the raw low payload bits: join values that are multiples of 2^16, share their low bits, or differ only in their high
 * bits would otherwise fall into ONE partition, and a partition far beyond an LDS table is joined in
 * (probe tasks) x (build chunks) table builds -- quadratic where the reference stays linear (its per-bucket table hashes the
 * whole value modulo a prime, Result.cpp:43-58).  The first kernel that touches a caller's tuple replaces the payload by h
 * (histogram digit of h; the scatter writes h); every later kernel compares h, and h == h' <=> payload == payload', so the
 * pair multiset is unchanged and only rowIDs are ever reported.  Cost: one mix per tuple in two HBM-bound kernels.
 * What remains: a join value REPEATED n times on both sides yields n^2 pairs under any plan (so does the reference), and
 * values chosen as rhj_unmix64(k << 16) defeat this fixed mix like any fixed hash; a partition whose build side exceeds the
 * table is then joined in ceil(build / table) x ceil(probe / probe_split) table builds, correctly.
 * The public STAGE calls rhj_histogram / rhj_partition / rhj_partition_at / rhj_bucket_join keep RAW payload bits (their
 * bucket order is the reference's).  rhj_set_option("partition.mix", 0) restores raw bits inside joins (A/B runs, tests). */
uint64_t rhj_mix64(uint64_t x);
uint64_t rhj_unmix64(uint64_t h);      /* rhj_unmix64(rhj_mix64(x)) == x */

typedef struct {
    int32_t passes;        /* -1 auto, 0, 1 or 2 */
    int32_t bits1;         /* radix bits of pass 1 (bits [0,bits1) of h, see above), 1..10; 0 = auto */
    int32_t bits2;         /* radix bits of pass 2 (bits [bits1,bits1+bits2) of h), 1..10; 0 = auto */
    int32_t probe_split;   /* max probe tuples per join task (skew/load balance); 0 = auto; values above 2^24 act as 2^24 */
} rhj_opts;

/* per-kernel device time of the LAST rhj_join / rhj_join_dev / stage call, from HIP events on the
 * context's stream (only filled while profiling is enabled, see rhj_set_profiling). */
typedef enum {
    RHJ_K_HIST = 0,        /* radix histogram           (HistogramJob::run) */
    RHJ_K_SCAN = 1,        /* prefix sums               (PartitionJob::run prefix + structs.cpp:168-173) */
    RHJ_K_SCATTER = 2,     /* scatter-partition         (PartitionJob::run scatter + structs.cpp:183-194) */
    RHJ_K_TASKS = 3,       /* join task list            (the JoinJob scheduling loop, Result.cpp:98-107) */
    RHJ_K_JOIN = 4,        /* bucket build+probe+write  (JoinJob::run / Result::join_buckets / add_result) */
    RHJ_K_AUX = 5,         /* unit tables, memsets, checksum, generators */
    RHJ_K_COUNT = 6
} rhj_kernel_kind;

typedef struct {
    double   ms[RHJ_K_COUNT];        /* summed device ms per kind */
    uint32_t launches[RHJ_K_COUNT];  /* launches per kind */
    double   total_ms;               /* first launch start -> last launch end */
    int32_t  passes, bits1, bits2;   /* the plan that ran */
    uint64_t ntasks;                 /* join tasks executed */
} rhj_timings;

/* ---- lifetime: replaces JobScheduler::init / stop / destroy for the join path
 *      (JobScheduler.cpp:67-86, 140-146, 89-97) --------------------------------------------- */
int  rhj_abi_version(void);
int  rhj_device_count(void);
int  rhj_init(int device, rhj_ctx **out_ctx);
void rhj_destroy(rhj_ctx *ctx);
const char *rhj_last_error(const rhj_ctx *ctx);          /* ctx may be NULL: last global error */
/* run on a caller-owned hipStream_t (e.g. torch's current stream); NULL = context's own stream */
int  rhj_set_stream(rhj_ctx *ctx, void *hip_stream);
/* enabled: 0 off; 1 every launch of a call is timed with a HIP event pair (rhj_get_timings / rhj_get_launch_timings report the
 * LAST call); 2 the same, and the launches of successive calls ACCUMULATE until profiling is set again (a benchmark reads the
 * events once, after its timed loop, instead of after every step) */
int  rhj_set_profiling(rhj_ctx *ctx, int enabled);
/* tuning / test knobs; results never depend on them.  "join.big_tables": -1 (default) choose the bucket-join kernel by
 * the average build partition, 0 always the one-table kernel, 1 always an oversized-partition kernel;
 * "join.big_kernel": -1 automatic, 1 the chunked 16-byte-entry kernel, 2 / 3 the compact-table kernel at full / half size
 * where the plan allows, 4 / 5 the same with 20 instead of 16 probe slots per thread (narrow partitions; else 2 / 3), 6 / 7 the 12288-entry geometry and its half-size form (6144 entries), 8 the full-size table with half the buckets (17920 entries in 8192 buckets instead of 16352 in 16384), 9 the 6144-entry geometry skipping the slot rows a partition leaves empty, 10 the same with 13-bit arrival indices (keys of up to 51 bits: what plans of 13-15 radix bits take by themselves for partitions of 2-5 K tuples), 11 a 4096-entry table with 12-bit arrival indices (plans of 12 bits);
 * "partition.narrow": -1 automatic, 0 never, 1 / 2: inside a join with a two-pass plan, partitions (1) and the
 * intermediate of the two passes (2; the only level of 17-18-bit plans) are stored as {payload 8 B, rowID 4 B} while every
 * rowID is below 2^32 (a larger one is detected on the device -- by the first histogram kernel -- and THAT join repeats itself
 * in the 16-byte format; the next join tries the narrow format again);
 * "partition.mix": -1 automatic (= 1 unless RHJ_MIX=0 is in the environment), 1: joins take their radix digits from
 * rhj_mix64(payload), 0: from the raw payload (see rhj_opts);
 * "join.sniff": -1 automatic (= 1 unless RHJ_SNIFF=0), 1: a partitioned join samples the join values of both relations for
 * duplicates while it counts them, and where the two sides of a partition are within 1/16 of each other in size the side with
 * fewer duplicates becomes the hash table (the reference builds on the smaller bucket, S on a tie: JobScheduler.cpp:187; a table
 * without duplicates answers every probe tuple with one match); 0: the first relation wins such a tie.  Same pairs either way. */
int  rhj_set_option(rhj_ctx *ctx, const char *name, int64_t value);
/* what the last join did: "last.narrow" (0 / 1 / 2, see above), "last.join_kernel" (0 one-table, 1 chunked, 2 / 3
 * compact table full / half size, 4 / 5 the same with 20 probe slots per thread, 6 / 7 the 12288- / 6144-entry geometries, 8 / 9 / 10 see "join.big_kernel", -1 none: direct small join or empty input), "last.pipelined" (the number
 * of S chunks the last rhj_join streamed through the device while finished pairs travelled home; 0: the plain path),
 * "last.max_part_R" / "last.max_part_S" (tuples in the largest partition of each side the last partitioned join saw; 0 for
 * an unpartitioned one), "partition.mix" (0 / 1: what joins on this context do) */
int  rhj_get_info(rhj_ctx *ctx, const char *name, int64_t *value);
int  rhj_get_timings(rhj_ctx *ctx, rhj_timings *out);
/* the same per LAUNCH, in launch order: kinds[i] (rhj_kernel_kind) and ms[i] of the first min(*n, capacity) timed spans of the
 * last call; *n = how many there were.  (A fused two-pass join scatters R pass 1, R pass 2, S pass 1, S pass 2 in that order:
 * bench.py prices the two scatter variants separately with this.) */
int  rhj_get_launch_timings(rhj_ctx *ctx, int32_t *kinds, double *ms, uint32_t capacity, uint32_t *n);
int  rhj_sync(rhj_ctx *ctx);                             /* JobScheduler::barrier (JobScheduler.cpp:103-122) */
/* pre-size / release the HBM workspace (otherwise grown on demand) */
int  rhj_reserve(rhj_ctx *ctx, uint64_t nR, uint64_t nS, const rhj_opts *opts);
int  rhj_release_workspace(rhj_ctx *ctx);
void rhj_default_opts(rhj_opts *opts);
/* the plan rhj_join would use for these sizes (host logic only, no device needed).  rhj_join_dev plans the same
 * except for build sides of 21 K - 51 K tuples, which it partitions with one pass where rhj_join stays unpartitioned
 * (one launch matters more when the inputs still have to cross PCIe); rhj_get_timings reports the plan that ran. */
int  rhj_plan(uint64_t nR, uint64_t nS, const rhj_opts *in, rhj_opts *resolved);

/* ---- the drop-in: replaces the body of Result::multiRadixHashJoin (Result.cpp:90-124) ----
 * Host AoS in, one result page out.  *out_page is NULL when there is no match (Result::isEmpty,
 * Result.cpp:16-18) else a malloc() block laid out like one reference result page
 * (Result.cpp:21-35): 8 bytes `next` pointer (= NULL) followed by *out_count rhj_pair.
 * The caller owns it and releases it with free() (as ~Result does, Result.cpp:127-133).
 * Inputs are neither modified nor retained. */
int rhj_join(rhj_ctx *ctx, const rhj_tuple *R, uint64_t nR, const rhj_tuple *S, uint64_t nS,
             const rhj_opts *opts, void **out_page, uint64_t *out_count);

/* ---- several joins in one call (MainScheduler.cpp:6-30 / join.cpp:42-50 keep 8 queries in flight) -------------
 * out_pages[i] / out_counts[i] are what rhj_join(ctx, joins[i].R, .., NULL, &page, &count) would return, for every i.  Joins small
 * enough for the one-launch path (every join of small.work: <= 43 K tuples) run SIXTEEN PER LAUNCH: their inputs are staged
 * into one pinned buffer by a few helper threads and cross PCIe in one copy, one kernel launch joins them (grid.y = join), the
 * pairs land in pinned host memory written by the kernel itself, one synchronisation -- instead of two copies, a launch and a
 * synchronisation per join.  Larger joins of the list take the rhj_join path one by one.  On an error every page already
 * produced is freed and all counts are zero. */
typedef struct { const rhj_tuple *R; uint64_t nR; const rhj_tuple *S; uint64_t nS; } rhj_join_desc;
int rhj_join_batch(rhj_ctx *ctx, uint32_t n, const rhj_join_desc *joins, void **out_pages, uint64_t *out_counts);

/* ---- device-resident variant (inputs/outputs already in HBM).  d_out may be NULL with
 * out_capacity 0 to count only.  Returns RHJ_E_OVERFLOW (and the exact *out_count) when
 * out_capacity is too small; call again with a larger buffer. */
int rhj_join_dev(rhj_ctx *ctx, const rhj_tuple *d_R, uint64_t nR, const rhj_tuple *d_S, uint64_t nS,
                 const rhj_opts *opts, rhj_pair *d_out, uint64_t out_capacity, uint64_t *out_count);

/* ---- stage entry points (device pointers), one per reference job body ---------------------
 * rhj_histogram: HistogramJob::run over the whole relation + the reduction of structs.cpp:168-173:
 *   d_hist[b] = #{ i : ((payload_i >> shift) & (2^bits-1)) == b },  d_hist has 2^bits uint64. */
int rhj_histogram(rhj_ctx *ctx, const rhj_tuple *d_rel, uint64_t n, int shift, int bits, uint64_t *d_hist);
/* rhj_prefix: the exclusive prefix of PartitionJob::run (JobScheduler.cpp:163-169):
 *   d_start[0]=0, d_start[b+1]=d_start[b]+d_hist[b]; d_start has nbins+1 uint64. */
int rhj_prefix(rhj_ctx *ctx, const uint64_t *d_hist, uint64_t nbins, uint64_t *d_start);
/* rhj_partition: relation_info::hash_relation (structs.cpp:144-204) generalised to one or two passes:
 *   d_out = tuples of d_in grouped by partition id  p = payload & (2^(bits1+bits2)-1)  laid out in the
 *   order  (p & (2^bits1-1)) * 2^bits2 + (p >> bits1)   [pass-1 digit major, pass-2 digit minor; with
 *   bits2 == 0 this is the reference's bucket order];  d_part_start[k], k in [0, 2^(bits1+bits2)], are
 *   the partition boundaries in that order.  Order of tuples INSIDE a partition is unspecified. */
int rhj_partition(rhj_ctx *ctx, const rhj_tuple *d_in, uint64_t n, int bits1, int bits2,
                  rhj_tuple *d_out, uint64_t *d_part_start);
/* rhj_partition_at: ONE scatter-partition pass on payload bits [shift, shift+bits): d_out grouped by that
 *   digit, d_part_start[2^bits + 1].  Used by the multi-GPU driver to split a shard by owner bits before
 *   the RCCL all-to-all; the owner bits lie above every bit the local plan uses. */
int rhj_partition_at(rhj_ctx *ctx, const rhj_tuple *d_in, uint64_t n, int shift, int bits,
                     rhj_tuple *d_out, uint64_t *d_part_start);
/* rhj_owner_histogram / rhj_owner_split: rhj_histogram / rhj_partition_at with the digit taken from bits [shift, shift+bits)
 *   of rhj_mix64(payload) instead of the payload; tuples are written UNCHANGED.  The multi-GPU owner split of 16-byte
 *   tuples (the wire format when the narrow one does not apply): the receiver runs rhj_join_dev on what arrives. */
int rhj_owner_histogram(rhj_ctx *ctx, const rhj_tuple *d_rel, uint64_t n, int shift, int bits, uint64_t *d_hist);
int rhj_owner_split(rhj_ctx *ctx, const rhj_tuple *d_in, uint64_t n, int shift, int bits,
                    rhj_tuple *d_out, uint64_t *d_class_start);
/* rhj_bucket_join: the JoinJob loop of Result.cpp:98-107 + JoinJob::run + Result::join_buckets +
 *   add_result: for every partition k with both sides non-empty, build an LDS hash table on the smaller
 *   side (S when |R_k| >= |S_k|, JobScheduler.cpp:187) and probe with the other; emit (rowR,rowS).
 *   radix_bits = number of low payload bits that are constant inside a partition (0 if unpartitioned). */
int rhj_bucket_join(rhj_ctx *ctx, const rhj_tuple *d_Rp, const uint64_t *d_startR,
                    const rhj_tuple *d_Sp, const uint64_t *d_startS, uint64_t nparts, int radix_bits,
                    int probe_split, rhj_pair *d_out, uint64_t out_capacity, uint64_t *out_count);

/* ---- multi-GPU stage entry points (the reference has no distributed path) -------------------
 * One process per GPU; both relations range-sharded by row (structs.cpp:146-161 applied across GPUs instead of threads).
 * These calls are the COMPUTE of a sharded join; the two collectives between them (an all-gather of the class
 * histograms, an all-to-all of the tuples over RCCL / xGMI) belong to the host, which may be C++ with rccl.h or Python
 * with torch.distributed (radixhashjoin_amd/sharded.py runs exactly this schedule):
 *
 *   1. rhj_shard_stats  (R, side 0), (S, side 1)     class histogram of the shard at bits [shift, shift+bits) of
 *                                                    h = rhj_mix64(payload) and the range of its rowIDs   [16 B/tuple read]
 *   2. all-gather {histograms, rowID ranges}  ->  every rank derives the same contiguous class range per owner and its
 *      send / receive counts; key_base = the shard's smallest rowID (the narrow wire format needs max - min < 2^32)
 *   3. rhj_shard_split  per relation                 class split straight into the NARROW WIRE FORMAT:
 *         payloads  uint64[n]  at d_narrow_out, as h = rhj_mix64(payload): what every later stage works on   (8 B/tuple)
 *         rowIDs    uint32[n]  at d_narrow_out + rhj_narrow_key_offset(n), value = rowID - key_base   (4 B/tuple)
 *      tuples of one class contiguous, classes in order: 12 B/tuple cross xGMI instead of 16      [16 B read + 12 B written]
 *   4. all-to-all of the payload array and of the rowID array (same element counts; destination d gets classes
 *      [cut[d], cut[d+1]))
 *   5. rhj_shard_partition per relation              the local fused two-pass radix partition of what arrived (one histogram
 *      read of the payloads, two scatter passes).  The receive buffer is nseg sender segments; pass-1 units are cut at the
 *      segment boundaries, so that pass 2 knows the sender of every tuple it moves and can restore GLOBAL rowIDs
 *      (row0[sender] + local rowID), in the way `mode` names:
 *        RHJ_SHARD_PLAIN     every rowID of both relations is < 2^32 and every rank split with key_base 0: nothing to restore,
 *                            narrow partitions, the kernels of a single-GPU join
 *        RHJ_SHARD_TAGGED    narrow partitions, the sender number in the low 4 payload bits (dead by then), resolved by the
 *                            one-table bucket join (partitions that fit one LDS table)
 *        RHJ_SHARD_GLOBAL16  pass 2 writes 16-byte tuples with global rowIDs (larger partitions: the compact-table kernel has
 *                            no register left to carry tags in)
 *   6. rhj_shard_join                                bucket join of the two partitioned sides (Result.cpp:43-76 per bucket):
 *      global rowIDs, as if one GPU had joined everything.
 * Results stay sharded (every rank holds the pairs of the join values it owns).
 * rhj_shard_plan returns RHJ_SHARD_TAGGED or RHJ_SHARD_GLOBAL16 for sizes / plans that fit this path (the host may use
 * RHJ_SHARD_PLAIN instead when the gathered rowID ranges allow it), RHJ_SHARD_PLAIN for 17-18-bit local plans, or 0: fall back
 * to exchanging 16-byte tuples (rhj_owner_histogram + rhj_owner_split + all-to-all + rhj_join_dev).
 * Limits: at most 16 ranks (nseg), class bits <= 8, fewer than 2^32 tuples received per relation, a two-pass local plan;
 * 17-18-bit plans (receivers beyond 1.1 * 10^9 tuples) only as RHJ_SHARD_PLAIN, which is what rhj_shard_plan returns for them. */
#define RHJ_SHARD_TAGGED 1
#define RHJ_SHARD_GLOBAL16 2
#define RHJ_SHARD_PLAIN 3
uint64_t rhj_narrow_key_offset(uint64_t n);               /* byte offset of the rowID array inside a narrow buffer of n tuples */
uint64_t rhj_narrow_bytes(uint64_t n);                    /* bytes of a narrow buffer of n tuples (<= 16 n for n >= 1024) */
int rhj_shard_plan(uint64_t nR, uint64_t nS, const rhj_opts *in, rhj_opts *resolved);   /* mode / 0 / negative rhj_status */
/* hist: HOST array of 2^bits counts; key_min / key_max: HOST words (may be NULL).  Synchronises.  side: 0 or 1 -- two sets of
 * unit tables, so that R and S can both be between their rhj_shard_stats and their rhj_shard_split. */
int rhj_shard_stats(rhj_ctx *ctx, int side, const rhj_tuple *d_rel, uint64_t n, int shift, int bits, uint64_t *hist,
                    uint64_t *key_min, uint64_t *key_max);
/* asynchronous; d_narrow_out has rhj_narrow_bytes(n) bytes; d_class_start (device, 2^bits + 1, may be NULL) gets the class
 * boundaries inside the output.  Same d_rel / n / shift / bits as the rhj_shard_stats call of this side (key_base is checked
 * against the rowID range that call found; a different relation whose rowIDs do not fit is detected on the device and
 * reported by the rhj_shard_join of this context: RHJ_E_INVALID). */
int rhj_shard_split(rhj_ctx *ctx, int side, const rhj_tuple *d_rel, uint64_t n, int shift, int bits, uint64_t key_base,
                    void *d_narrow_out, uint64_t *d_class_start);
/* rhj_shard_split with NO send buffer and NO all-to-all: class c of this rank's shard is written straight into the receive
 * arrays of the rank that owns it -- peer_payloads[owner[c]] (uint64) / peer_rowids[owner[c]] (uint32), device pointers of
 * this process: a peer's HBM opened with rhj_ipc_open (xGMI peer stores), or local buffers -- starting at element
 * dst_class_start[c] of those arrays (from the gathered count matrix: where sender `rank`'s segment begins in the owner's
 * arrays + the sender's classes of that owner before c).  owner / dst_class_start: HOST arrays of 2^bits entries.
 * Same d_rel / n / shift / bits / key_base rules as rhj_shard_split; the receiver must not read its arrays before every
 * sender's call has completed (a barrier on the stream, e.g. a 1-word all-reduce), then runs rhj_shard_partition as usual.
 * HBM traffic per tuple: 16 B read + 12 B written, against + 12 B read + 12 B written by the all-to-all's copy. */
int rhj_shard_split_peer(rhj_ctx *ctx, int side, const rhj_tuple *d_rel, uint64_t n, int shift, int bits, uint64_t key_base,
                         const uint8_t *owner, const uint64_t *dst_class_start, void *const *peer_payloads, void *const *peer_rowids,
                         int nranks);
/* HBM of another process of the node mapped into this one (hipIpc*): export a 64-byte handle of a device allocation, open it
 * elsewhere, close it.  For the peer arrays of rhj_shard_split_peer.  Unverified on the one-GPU boxes of this pool. */
int rhj_ipc_export(rhj_ctx *ctx, void *d_ptr, void *handle64);
int rhj_ipc_open(rhj_ctx *ctx, const void *handle64, void **d_ptr);
int rhj_ipc_close(rhj_ctx *ctx, void *d_ptr);
/* asynchronous; seg_off: HOST array of nseg + 1 offsets into the received arrays (seg_off[0] = 0, seg_off[nseg] = m),
 * segment s = what rank s sent; row0: HOST array of nseg rowID bases (the key_base each rank split this relation with);
 * plan: the resolved two-pass plan rhj_shard_plan returned a mode for; plan and mode the same on both sides */
int rhj_shard_partition(rhj_ctx *ctx, int side, const uint64_t *d_payloads, const uint32_t *d_rowids, uint64_t m, int nseg,
                        const uint64_t *seg_off, const uint64_t *row0, const rhj_opts *plan, int mode);
/* count / overflow behaviour of rhj_join_dev */
int rhj_shard_join(rhj_ctx *ctx, rhj_pair *d_out, uint64_t out_capacity, uint64_t *out_count);

/* ---- utilities -------------------------------------------------------------------------- */
/* order-insensitive checksum of SURVEY.md App. A over a device pair array:
 *   sum over pairs of mix(keyR * 0x100000001B3 ^ mix(keyS))  (mod 2^64), mix = splitmix64 step */
int rhj_pairs_checksum_dev(rhj_ctx *ctx, const rhj_pair *d_pairs, uint64_t n, uint64_t *checksum);
/* synthetic inputs generated in HBM:  kind 0: R[i] = {i+row0, mix(1 + (i+row0) % D)}
 *   kind 1: uniform FK, counter based: S[j] = {j+row0, mix(1 + mix((j+row0) ^ seed) % D)}
 *   kind 2: Zipf(theta) FK: rank r in [1,D] by inverse-CDF of the continuous approximation from
 *           u = mix((j+row0) ^ seed) / 2^64; payload = mix(r)            (theta given as theta_milli/1000)
 *   kind 3: disjoint: S[j] = {j+row0, mix(D + 1 + j + row0)}
 *   kind 4: constant: T[i] = {i+row0, D} */
int rhj_generate_dev(rhj_ctx *ctx, int kind, rhj_tuple *d_out, uint64_t n, uint64_t row0, uint64_t D,
                     uint64_t seed, int theta_milli);
/* Re-labels the join values of a GENERATED relation in place: payload = (k << shift) + add for a payload rhj_mix64(k)
 *   (kinds 0-3 above).  Applied to both sides of a PK/FK pair of relations it leaves the pair set -- hence
 *   rhj_expected_pkfk_dev's answer, taken BEFORE the call -- unchanged while the join values become dense (shift 0: the
 *   value range of the reference's small/ data), multiples of 2^shift, or k * 2^shift + const. */
int rhj_remap_keys_dev(rhj_ctx *ctx, rhj_tuple *d_rel, uint64_t n, int shift, uint64_t add);
/* closed-form expectation for PK/FK inputs (R of kind 0 with D == |R| global, unique payloads):
 *   every S tuple {j, mix(k)} matches exactly R row k-1: *count = n, *checksum = sum mix((k-1)*0x100000001B3 ^ mix(j)).
 *   Computed by one streaming pass over S that inverts mix(); does not run the join. */
int rhj_expected_pkfk_dev(rhj_ctx *ctx, const rhj_tuple *d_S, uint64_t n, uint64_t *count, uint64_t *checksum);

/* ---- query-layer kernels (the steps immediately before and after the hot path) -------------
 * With the columns of the stored relations resident in HBM these make a whole query device-resident:
 * filters, join-input construction, intermediate-result maintenance and the SUM projections never
 * cross PCIe; only counts and 8-byte sums reach the host.  All arrays are uint64 in HBM.  A NULL row list
 * (d_rows / d_rowsA / d_rowsB / d_rows_in) stands for the identity 0..n-1 (an alias without filters).
 *
 * rhj_col_filter: the filter loops of Query::run_filters (Query.cpp:96-146).  d_rows_in == NULL means
 *   "all rows 0..n_in-1".  Keeps the rows r with  d_col[r] <op> value,  op in {'<','>','='}; writes them
 *   (unordered) to d_rows_out (capacity n_in) and their number to *n_out. */
int rhj_col_filter(rhj_ctx *ctx, const uint64_t *d_col, const uint64_t *d_rows_in, uint64_t n_in, int op,
                   uint64_t value, uint64_t *d_rows_out, uint64_t *n_out);
/* rhj_gather_tuples: relation::foo / create_relation (structs.cpp:217-243) without the host round trip:
 *   d_tuples[i] = { key = key_is_position ? i : d_rows[i],  payload = d_col[d_rows[i]] }.
 *   key_is_position = 1 builds a POSITION-CARRYING join input: the join's pairs then name intermediate rows
 *   directly, which replaces the de-duplication of structs.cpp:238-241 and the rescans of
 *   intermediate.cpp:52-87 by one gather. */
int rhj_gather_tuples(rhj_ctx *ctx, const uint64_t *d_col, const uint64_t *d_rows, uint64_t n, int key_is_position,
                      rhj_tuple *d_tuples);
/* rhj_pairs_split: getVector (intermediate.cpp:92-105): d_r[i] = pairs[i].keyR, d_s[i] = pairs[i].keyS */
int rhj_pairs_split(rhj_ctx *ctx, const rhj_pair *d_pairs, uint64_t n, uint64_t *d_r, uint64_t *d_s);
/* rhj_gather_u64: d_dst[i] = d_src[d_idx[i]]  (re-materialises one intermediate column after a join) */
int rhj_gather_u64(rhj_ctx *ctx, const uint64_t *d_src, const uint64_t *d_idx, uint64_t n, uint64_t *d_dst);
/* rhj_rows_filter_equal: a predicate between two aliases that are BOTH in the intermediate already
 *   (intermediate.cpp:72-87,169-180) or a same-alias predicate (parse_table, intermediate.cpp:11-44):
 *   keeps the positions e with d_colA[d_rowsA[e]] == d_colB[d_rowsB[e]] in d_pos_out (capacity n). */
int rhj_rows_filter_equal(rhj_ctx *ctx, const uint64_t *d_colA, const uint64_t *d_rowsA, const uint64_t *d_colB,
                          const uint64_t *d_rowsB, uint64_t n, uint64_t *d_pos_out, uint64_t *n_out);
/* rhj_sum_gather: column_proj (Query.cpp:66-74): *sum = sum of d_col[d_rows[i]] (mod 2^64) */
int rhj_sum_gather(rhj_ctx *ctx, const uint64_t *d_col, const uint64_t *d_rows, uint64_t n, uint64_t *sum);

/* raw HBM helpers so a plain C/C++ host (no HIP headers) can use the device-resident API.  rhj_dev_free must be
 * given the context that allocated the block; released blocks are kept by the context for re-use (its work is
 * ordered on one stream) and go back to the device with rhj_release_workspace / rhj_destroy. */
int rhj_dev_alloc(rhj_ctx *ctx, uint64_t bytes, void **d_ptr);
int rhj_dev_free(rhj_ctx *ctx, void *d_ptr);
int rhj_copy_h2d(rhj_ctx *ctx, void *d_dst, const void *src, uint64_t bytes);
int rhj_copy_d2h(rhj_ctx *ctx, void *dst, const void *d_src, uint64_t bytes);
int rhj_dev_mem_info(rhj_ctx *ctx, uint64_t *free_bytes, uint64_t *total_bytes);

#ifdef __cplusplus
}
#endif
#endif /* RHJ_H */
