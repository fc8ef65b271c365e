// rhj_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the radix hash join.
//
// Kernel                 replaces (reference file:line)                          bound
// k_hist_units           HistogramJob::run            JobScheduler.cpp:149-155     HBM read  (16 B/tuple)
// k_scan_units           hist reduce + range prefix   structs.cpp:168-173,
//                                                     JobScheduler.cpp:163-169     latency (KBs)
// k_hist2d_units         the same for both passes of a two-pass plan at once     HBM read  (16 B/tuple, once)
// k_scatter_wc           PartitionJob scatter + the   JobScheduler.cpp:170-174,
// (k_scatter_units_pipe) serial merge-gather          structs.cpp:183-194          HBM read+write (32 B/tuple)
// k_make_tasks           JoinJob scheduling loop      Result.cpp:98-107            latency
// k_join_bkt             JoinJob::run, join_buckets,  JobScheduler.cpp:186-192,
//                        add_result / addAll          Result.cpp:43-76, 21-35      HBM read+write (16 B/tuple + 16 B/pair)
//                        (partitions that fit one 16 B/entry LDS table; DIRECT: small unpartitioned joins, no task list)
// k_join_ct              the same for partitions of 2 K ... 17.9 K build tuples under plans that remove >= 16 payload
//                        bits: 8 B entries, both sides read once; six table geometries         vector ALU + LDS latency
// k_*2                   the one-pass kernels with grid.y = relation: R and S of a join through the same launches
// k_scatter_wcn          the same scatter writing the narrow {payload 8 B, rowID 4 B} format inside a join: 32-tuple carry lines for
//                        <= 8-bit passes, 16-tuple lines for 9-bit passes (17-18-bit plans)                HBM read+write (28 / 24 B/tuple)
// k_hist_units_n         HistogramJob::run over a narrow payload array (pass 2 of 17-18-bit plans)          HBM read (8 B/tuple)
// multi-GPU receiver / sender (no reference counterpart: the reference is one process):
// k_seg_units            pass-1 units cut at the sender segments of a received shard
// k_hist2d_units<true>   the two-pass histogram over received payloads                                       HBM read (8 B/tuple)
// k_scatter_wc_n         last pass in front of the compact-table join: narrow in, 16-byte tuples with global rowIDs out
// k_join_bkt<..TAGGED>   the one-table join resolving {sender tag, shard-local rowID} into global rowIDs
// k_check_radix          contract check of the public rhj_bucket_join (radix_bits must describe the partitions)
//
// No MFMA anywhere: the path is 64-bit integer hashing and data movement, bounded by HBM.
#include "rhj_internal.h"

#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <string>
#include <type_traits>
#include <vector>

namespace {

struct __align__(16) Tup { u64 key; u64 payload; };   // reference structs.h:33-36
struct __align__(16) Pair { u64 r; u64 s; };          // reference Result.h:9-12

__device__ __forceinline__ u64 mix64(u64 z)
{
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

__device__ __forceinline__ u64 unmix64(u64 x)      // inverse of mix64
{
    x ^= (x >> 31) ^ (x >> 62);
    x *= 0x319642B2D24D8EC3ULL;
    x ^= (x >> 27) ^ (x >> 54);
    x *= 0x96DE1B173F119089ULL;
    x ^= (x >> 30) ^ (x >> 60);
    return x - 0x9E3779B97F4A7C15ULL;
}

// inclusive scan over the 64 lanes of a wavefront
__device__ __forceinline__ u32 wave_incl_scan(u32 v, int lane)
{
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        u32 t = __shfl_up(v, off, 64);
        if (lane >= off) v += t;
    }
    return v;
}

// Workgroup exclusive scan. wsum: LDS scratch of THREADS/64 words. Ends with a barrier, so wsum
// may be reused immediately by the caller.
template <int THREADS, bool TRAILING_SYNC = true>
__device__ __forceinline__ u32 block_excl_scan(u32 v, u32 *wsum, u32 &total, int tid = (int)threadIdx.x)
{
    // (tid: a caller inside a long loop passes an opaque copy of the thread index, so that &wsum[w] is recomputed here
    // instead of being hoisted out of the loop and spilled)
    constexpr int NW = THREADS / 64;
    const int lane = tid & 63, w = tid >> 6;
    const u32 inc = wave_incl_scan(v, lane);
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    u32 base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < NW; i++) {
        const u32 s = wsum[i];
        if (i < w) base += s;
        tot += s;
    }
    if (TRAILING_SYNC) __syncthreads();      // (false: the caller does not touch wsum before its own next barrier)
    total = tot;
    return base + inc - v;
}

// unit u -> segment s with unit_start[s] <= u < unit_start[s+1]  (unit_start has nseg+1 entries)
__device__ __forceinline__ u32 find_segment(const u32 *__restrict__ unit_start, u32 nseg, u32 u)
{
    u32 lo = 0, hi = nseg;     // invariant: unit_start[lo] <= u < unit_start[hi]
    while (hi - lo > 1) {
        const u32 mid = (lo + hi) >> 1;
        if (unit_start[mid] <= u) lo = mid; else hi = mid;
    }
    return lo;
}

// ------------------------------------------------------------------------------------------------
// unit tables
// ------------------------------------------------------------------------------------------------
__global__ void k_init_single_segment(u64 n, u64 L, u64 *seg_start, u32 *unit_start)
{
    if (threadIdx.x == 0) {
        seg_start[0] = 0; seg_start[1] = n;
        unit_start[0] = 0; unit_start[1] = (u32)((n + L - 1) / L);
    }
}

// Pass-1 units of a relation that arrived in SEGMENTS (multi-GPU receiver: segment s = the tuples sender s sent): every
// segment is cut into the same number of units (units past a short segment's end are empty), so unit u belongs to segment
// u / units_per_seg and no unit straddles two senders.  Also writes the one-segment tables {0, n}, {0, units}.
constexpr int SEG_MAX = 16;
struct SegPlan { u32 nseg, units_per_seg; u64 off[SEG_MAX + 1]; u64 L[SEG_MAX]; };

__global__ void __launch_bounds__(256) k_seg_units(SegPlan sp, u64 *__restrict__ unit_rng, u64 *__restrict__ seg_start,
                                                   u32 *__restrict__ unit_start)
{
    const u32 total = sp.nseg * sp.units_per_seg;
    for (u32 u = blockIdx.x * 256 + threadIdx.x; u <= total; u += gridDim.x * 256) {
        if (u == total) { unit_rng[u] = sp.off[sp.nseg]; continue; }
        const u32 s = u / sp.units_per_seg, j = u % sp.units_per_seg;
        const u64 b = sp.off[s] + (u64)j * sp.L[s];
        unit_rng[u] = b < sp.off[s + 1] ? b : sp.off[s + 1];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        seg_start[0] = 0; seg_start[1] = sp.off[sp.nseg];
        unit_start[0] = 0; unit_start[1] = total;
    }
}

// ---- both relations of a join through the same launches (one-pass plans: mid-size joins are launch-bound) -----------
// blockIdx.y selects the relation; the kernels below are the single-relation bodies called with that relation's arguments.
struct PassRel {                      // one relation's side of a partition pass
    const Tup *in;
    Tup *out;
    u64 *seg_start;                   // {0, n}
    u32 *unit_start;                  // {0, units}
    u32 *unit_hist;
    u64 *unit_base;
    u64 *part_start;
    u64 *scan_tmp;
    u64 n, L;
    u32 max_units;
};
struct PassPair { PassRel r[2]; int mix; };

// zero8 (optional): the eight 64-bit join counters, cleared here so that the join phase needs no memset of its own
__global__ void k_init_single_segment2(PassPair a, u64 *__restrict__ zero8)
{
    const PassRel &x = a.r[blockIdx.x];
    if (threadIdx.x == 0) {
        x.seg_start[0] = 0; x.seg_start[1] = x.n;
        x.unit_start[0] = 0; x.unit_start[1] = (u32)((x.n + x.L - 1) / x.L);
    }
    if (zero8 != nullptr && blockIdx.x == 0 && threadIdx.x < 8) zero8[threadIdx.x] = 0;
}

// unit_start[s] = sum_{s'<s} ceil(size(s') / L), one workgroup, nseg arbitrary
__global__ void __launch_bounds__(1024) k_make_units(const u64 *__restrict__ seg_start, u32 nseg, u64 L,
                                                     u32 *__restrict__ unit_start)
{
    __shared__ u32 wsum[16];
    __shared__ u32 carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (u32 base = 0; base < nseg; base += 1024) {
        const u32 s = base + threadIdx.x;
        u32 v = 0;
        if (s < nseg) v = (u32)((seg_start[s + 1] - seg_start[s] + L - 1) / L);
        u32 tot;
        const u32 ex = block_excl_scan<1024>(v, wsum, tot);
        const u32 c = carry;
        if (s < nseg) unit_start[s] = c + ex;
        __syncthreads();
        if (threadIdx.x == 0) carry = c + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) unit_start[nseg] = carry;
}

// ------------------------------------------------------------------------------------------------
// K1: radix histogram per unit.  HistogramJob::run (JobScheduler.cpp:149-155): hist[payload & mask]++
// over a row range.  16 B/lane coalesced loads (the rowID rides along in the same 128 B line),
// LDS histogram per workgroup, one flush per unit.
// ------------------------------------------------------------------------------------------------
// the same over a payload array (a narrow relation: 8 B/tuple), 8 loads in flight per lane
__global__ void __launch_bounds__(PART_THREADS)
k_hist_units_n(const u64 *__restrict__ inP, const u64 *__restrict__ seg_start, const u32 *__restrict__ unit_start,
               u32 nseg, u64 L, int shift, int bits, u32 *__restrict__ unit_hist)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u32 *cnt = reinterpret_cast<u32 *>(smem);
    const u32 nbins = 1u << bits, mask = nbins - 1, u = blockIdx.x;
    if (u >= unit_start[nseg]) return;
    u32 lo = 0, hi = nseg;                              // find_segment (defined below)
    while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if (unit_start[mid] <= u) lo = mid; else hi = mid; }
    const u32 s = lo;
    const u64 beg = seg_start[s] + (u64)(u - unit_start[s]) * L;
    const u64 send = seg_start[s + 1];
    const u64 end = (beg + L < send) ? beg + L : send;
    for (u32 b = threadIdx.x; b < nbins; b += PART_THREADS) cnt[b] = 0;
    __syncthreads();
    u64 i = beg + threadIdx.x;
    for (; i + 7ull * PART_THREADS < end; i += 8ull * PART_THREADS) {
        u64 v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = inP[i + (u64)k * PART_THREADS];
#pragma unroll
        for (int k = 0; k < 8; k++) atomicAdd(&cnt[(u32)(v[k] >> shift) & mask], 1u);
    }
    for (; i < end; i += PART_THREADS) atomicAdd(&cnt[(u32)(inP[i] >> shift) & mask], 1u);
    __syncthreads();
    u32 *out = unit_hist + (u64)u * nbins;
    for (u32 b = threadIdx.x; b < nbins; b += PART_THREADS) out[b] = cnt[b];
}

// DupSniff (rhj_internal.h): m = mix64(payload)
__device__ __forceinline__ void sniff_sample(const DupSniff &sn, u64 m)
{
    if (sn.sel_bits > 0 && (m >> (64 - sn.sel_bits)) != 0) return;
    const u64 m2 = mix64(m ^ 0xA5A5A5A5A5A5A5A5ull);             // (bits of its own for the slot: the sampled values share their top bits)
    (void)__hip_atomic_fetch_add(&sn.tab[(u32)m2 & (SNIFF_SLOTS - 1)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void
dev_hist_units(const Tup *__restrict__ in, const u64 *__restrict__ seg_start, const u32 *__restrict__ unit_start,
               u32 nseg, u64 L, int shift, int bits, u32 *__restrict__ unit_hist, const u32 u, u64 *__restrict__ minmax = nullptr,
               int mix = 0, const DupSniff sn = DupSniff())
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u32 *cnt = reinterpret_cast<u32 *>(smem);
    const u32 nbins = 1u << bits, mask = nbins - 1;
    if (u >= unit_start[nseg]) return;
    const u32 s = find_segment(unit_start, nseg, u);
    const u64 beg = seg_start[s] + (u64)(u - unit_start[s]) * L;
    const u64 send = seg_start[s + 1];
    const u64 end = (beg + L < send) ? beg + L : send;

    for (u32 b = threadIdx.x; b < nbins; b += PART_THREADS) cnt[b] = 0;
    __syncthreads();

    u64 i = beg + threadIdx.x;
    u64 kmin = ~0ull, kmax = 0;                        // range of the rowIDs (minmax != nullptr: the multi-GPU sender's class histogram)
    // mix != 0: the digit comes from mix64(payload) (inside a join, see MIX_* in rhj_internal.h)
    auto dig = [&](u64 p) -> u32 {
        const u64 m = mix ? mix64(p) : p;
        if (sn.tab != nullptr) sniff_sample(sn, mix ? m : mix64(p));
        return (u32)(m >> shift) & mask;
    };
    // 4 independent 16 B loads in flight per lane
    for (; i + 3ull * PART_THREADS < end; i += 4ull * PART_THREADS) {
        const Tup t0 = in[i], t1 = in[i + PART_THREADS], t2 = in[i + 2 * PART_THREADS], t3 = in[i + 3 * PART_THREADS];
        atomicAdd(&cnt[dig(t0.payload)], 1u);
        atomicAdd(&cnt[dig(t1.payload)], 1u);
        atomicAdd(&cnt[dig(t2.payload)], 1u);
        atomicAdd(&cnt[dig(t3.payload)], 1u);
        if (minmax != nullptr) {
            const u64 a = t0.key < t1.key ? t0.key : t1.key, b = t2.key < t3.key ? t2.key : t3.key;
            const u64 c = t0.key > t1.key ? t0.key : t1.key, d = t2.key > t3.key ? t2.key : t3.key;
            const u64 lo = a < b ? a : b, hi = c > d ? c : d;
            kmin = lo < kmin ? lo : kmin;
            kmax = hi > kmax ? hi : kmax;
        }
    }
    for (; i < end; i += PART_THREADS) {
        const Tup t = in[i];
        atomicAdd(&cnt[dig(t.payload)], 1u);
        kmin = t.key < kmin ? t.key : kmin;
        kmax = t.key > kmax ? t.key : kmax;
    }
    if (minmax != nullptr) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const u64 a = __shfl_down(kmin, off, 64), b = __shfl_down(kmax, off, 64);
            kmin = a < kmin ? a : kmin;
            kmax = b > kmax ? b : kmax;
        }
        if ((threadIdx.x & 63) == 0 && beg < end) { atomicMin(&minmax[0], kmin); atomicMax(&minmax[1], kmax); }
    }
    __syncthreads();
    u32 *out = unit_hist + (u64)u * nbins;
    for (u32 b = threadIdx.x; b < nbins; b += PART_THREADS) out[b] = cnt[b];
}

__global__ void __launch_bounds__(PART_THREADS)
k_hist_units(const Tup *__restrict__ in, const u64 *__restrict__ seg_start, const u32 *__restrict__ unit_start,
             u32 nseg, u64 L, int shift, int bits, u32 *__restrict__ unit_hist, u64 *__restrict__ minmax, int mix, DupSniff sn)
{
    dev_hist_units(in, seg_start, unit_start, nseg, L, shift, bits, unit_hist, blockIdx.x, minmax, mix, sn);
}

__global__ void __launch_bounds__(PART_THREADS) k_hist_units2(PassPair a, int shift, int bits)
{
    const PassRel &x = a.r[blockIdx.y];
    if (blockIdx.x >= x.max_units) return;
    dev_hist_units(x.in, x.seg_start, x.unit_start, 1u, x.L, shift, bits, x.unit_hist, blockIdx.x, nullptr, a.mix);
}


// ------------------------------------------------------------------------------------------------
// K1 (fused two-pass form): ONE read of the input yields the histograms of BOTH passes.
//   hist1[u][d1]            per pass-1 unit, as k_hist_units
//   hist2[(d1*NG + g)][d2]  per pass-2 unit: pass-2 units are defined as "the piece of bucket d1 written by
//                           group g of pass-1 units" (a contiguous range of the pass-1 output whose bounds
//                           come from the pass-1 cursors), so every tuple's pass-2 unit is known here.
// The (d1,d2) counts of a unit live in LDS as packed 16-bit counters (2^(b1+b2) x 2 B <= 128 KiB).  A counter
// is drained to the global table when it reaches 2^15 (the thread whose add moved it from 0x7FFF takes
// 0x8000 out again), so a half can never carry into its neighbour: at most threads x 4 adds are in flight.
// Saves the second histogram read: 16 B/tuple of HBM traffic per relation.
// ------------------------------------------------------------------------------------------------
constexpr int H2_THREADS = 1024;

// unit_rng (optional): explicit pass-1 units, unit u = rows [unit_rng[u], unit_rng[u+1]) (the multi-GPU receiver cuts its
// pass-1 units at the sender segments of the receive buffer -- k_seg_units -- so that a group of pass-1 units, hence a
// pass-2 unit, holds tuples of ONE sender).  Null: unit u = rows [u * L, (u+1) * L).  Group = u / units_per_group either way.

// IN_NARROW: the input is a payload array (8 B/tuple; a received narrow shard), no rowIDs to inspect.
// the planners' question: which side would rather be the hash table, and how lopsided may a partition be before size decides?
// Called by EVERY thread of the workgroup (barriers inside); sh: two words of LDS.
//   no sampling                       -> R, ties of 1/2^tie_shift
//   both sides sampled                -> the side with the lower duplicate rate, ties of 1/2^tie_shift; and when the other side's
//                                        rate is at least four times as high and at least 1/16 of its sample are duplicates, the
//                                        cleaner side is built unless it is more than TWICE as large ([measured] Zipf(0.9) foreign
//                                        key, wall ms with ties of 1/16 / 1/2 / 1/1: 1M 0.174 / 0.084 / 0.070, 3M 0.479 / 0.141 /
//                                        0.134, 6M 0.965 / 0.293 / 0.296 -- most partitions of a skewed foreign key are a little
//                                        SMALLER than the key side's, and a table of duplicates is long buckets; uniform: no change)
struct BuildRule { bool prefer_R; int shift; };
__device__ __forceinline__ BuildRule sniff_rule(const SniffVerdict &sv, u32 *sh, int tie_shift)
{
    BuildRule r{true, tie_shift};
    if (sv.tab == nullptr) return r;
    if (threadIdx.x < 2) sh[threadIdx.x] = 0;
    __syncthreads();
#pragma unroll
    for (int side = 0; side < 2; side++) {
        u32 d = 0;
        const uint4 *t4 = reinterpret_cast<const uint4 *>(sv.tab + (size_t)side * SNIFF_SLOTS);
#pragma unroll 8
        for (u32 i = threadIdx.x; i < SNIFF_SLOTS / 4; i += blockDim.x) {           // (independent 16-byte loads: in flight together)
            const uint4 c = t4[i];
            d += (c.x > 1 ? c.x - 1 : 0) + (c.y > 1 ? c.y - 1 : 0) + (c.z > 1 ? c.z - 1 : 0) + (c.w > 1 ? c.w - 1 : 0);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) d += __shfl_down(d, off, 64);
        if ((threadIdx.x & 63) == 0 && d) atomicAdd(&sh[side], d);
    }
    __syncthreads();
    const u64 dR = sh[0], dS = sh[1];
    __syncthreads();                                             // (sh belongs to the caller again)
    const u64 a = dR * (u64)sv.expect_S, b = dS * (u64)sv.expect_R;          // the two rates, cross-multiplied
    r.prefer_R = a <= b;
    const u64 lo = r.prefer_R ? a : b, hi = r.prefer_R ? b : a;
    const u64 worse_d = r.prefer_R ? dS : dR, worse_e = r.prefer_R ? sv.expect_S : sv.expect_R;
    if (tie_shift < 63 && hi >= 4 * lo && worse_d * 16 >= worse_e && worse_d != 0) r.shift = 0;
    return r;
}

template <bool IN_NARROW>
__global__ void __launch_bounds__(H2_THREADS)
k_hist2d_units(const Tup *__restrict__ in, const u64 *__restrict__ inP, u64 n, u64 L, int b1, int b2, u32 units_per_group,
               u32 ngroups, u32 *__restrict__ hist1, u32 *__restrict__ hist2, u64 key_base, u32 *__restrict__ wide,
               const u64 *__restrict__ unit_rng, int mix, DupSniff sn)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const u32 nb1 = 1u << b1, nb2 = 1u << b2, nbin = nb1 * nb2;
    u32 *tab = reinterpret_cast<u32 *>(smem);                 // nbin / 2 words, two 16-bit counters each
    u32 *sum1 = tab + (nbin >> 1);                            // nb1
    const u32 u = blockIdx.x;
    u64 beg, end;
    if (unit_rng != nullptr) { beg = unit_rng[u]; end = unit_rng[u + 1]; }
    else {
        beg = (u64)u * L;
        if (beg >= n) return;
        end = (beg + L < n) ? beg + L : n;
    }
    const u32 grp = u / units_per_group;
    const int tid = threadIdx.x, lane = tid & 63;
    const u32 m1 = nb1 - 1, m2 = nb2 - 1;

    for (u32 i = tid; i < (nbin >> 1) + nb1; i += H2_THREADS) tab[i] = 0;
    __syncthreads();

    auto count = [&](u64 payload) {
        if (!IN_NARROW && mix) payload = mix64(payload);      // (16-byte input inside a join: digits of the mixed payload)
        if (sn.tab != nullptr) sniff_sample(sn, IN_NARROW || mix ? payload : mix64(payload));   // (received payloads are mixed already)
        const u32 d1 = (u32)payload & m1, d2 = (u32)(payload >> b1) & m2;
        const u32 bin = (d1 << b2) | d2, sh = (bin & 1u) * 16u;
        const u32 old = atomicAdd(&tab[bin >> 1], 1u << sh);
        if (((old >> sh) & 0xFFFFu) == 0x7FFFu) {             // this add made it 2^15: drain 2^15 to the global tables
            atomicSub(&tab[bin >> 1], 0x8000u << sh);
            atomicAdd(&hist2[((u64)(d1 * ngroups + grp) << b2) + d2], 0x8000u);
            atomicAdd(&sum1[d1], 0x8000u);
        }
    };
    u64 i = beg + tid;
    if constexpr (IN_NARROW) {
        for (; i + 7ull * H2_THREADS < end; i += 8ull * H2_THREADS) {        // 8 x 8 B loads in flight per lane
            u64 v[8];
#pragma unroll
            for (int k = 0; k < 8; k++) v[k] = inP[i + (u64)k * H2_THREADS];
#pragma unroll
            for (int k = 0; k < 8; k++) count(v[k]);
        }
        for (; i < end; i += H2_THREADS) count(inP[i]);
    } else {
        // wide (optional): the narrow format is wanted downstream -- this kernel sees every rowID anyway, so a rowID that
        // does not fit 32 bits (after subtracting key_base) is reported here, before any scatter has run
        u32 hi = 0;
        for (; i + 3ull * H2_THREADS < end; i += 4ull * H2_THREADS) {
            const Tup t0 = in[i], t1 = in[i + H2_THREADS], t2 = in[i + 2 * H2_THREADS], t3 = in[i + 3 * H2_THREADS];
            count(t0.payload); count(t1.payload); count(t2.payload); count(t3.payload);
            hi |= (u32)((t0.key - key_base) >> 32) | (u32)((t1.key - key_base) >> 32) | (u32)((t2.key - key_base) >> 32) |
                  (u32)((t3.key - key_base) >> 32);
        }
        for (; i < end; i += H2_THREADS) { const Tup t = in[i]; count(t.payload); hi |= (u32)((t.key - key_base) >> 32); }
        if (wide != nullptr && __ballot(hi != 0) != 0 && lane == 0) atomicOr(wide, 1u);
    }
    __syncthreads();

    // flush: 64 consecutive bins per wavefront instruction (one 256 B row segment of hist2 when nb2 >= 64)
    for (u32 bin = tid; bin < nbin; bin += H2_THREADS) {
        const u32 c = (tab[bin >> 1] >> ((bin & 1u) * 16u)) & 0xFFFFu;
        const u32 d1 = bin >> b2, d2 = bin & m2;
        if (c) atomicAdd(&hist2[((u64)(d1 * ngroups + grp) << b2) + d2], c);
        if (nb2 >= 64) {                                       // whole wavefront shares d1: reduce, one LDS add
            u32 r = c;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) r += __shfl_down(r, off, 64);
            if (lane == 0 && r) atomicAdd(&sum1[d1], r);
        } else if (c) {
            atomicAdd(&sum1[d1], c);
        }
    }
    __syncthreads();
    for (u32 d = tid; d < nb1; d += H2_THREADS) hist1[(u64)u * nb1 + d] = sum1[d];
}

// rng[d1*NG + g] = first output index of bucket d1 written by group g of pass-1 units (= the pass-1 cursor of
// the group's first unit); rng[nb1*NG] = n; unit_start2[d1] = d1*NG (NG pass-2 units per bucket).
__global__ void k_make_group_ranges(const u64 *__restrict__ unit_base1, u32 nb1, u32 units_per_group, u32 ngroups,
                                    u64 n, u64 *__restrict__ rng, u32 *__restrict__ unit_start2)
{
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    const u32 total = nb1 * ngroups;
    if (i < total) {
        const u32 d1 = i / ngroups, g = i % ngroups;
        rng[i] = unit_base1[(u64)g * units_per_group * nb1 + d1];
    }
    if (i == total) rng[total] = n;
    if (i <= nb1) unit_start2[i] = i * ngroups;
}

// ------------------------------------------------------------------------------------------------
// K2: per segment, turn unit histograms into absolute write cursors.
//   part_start[s*nbins + d] = seg_start[s] + sum_{d'<d} total(s,d')         (the global histogram's
//       exclusive prefix, structs.cpp:168-173 + JobScheduler.cpp:163-169)
//   unit_base[u*nbins + d]  = part_start[s*nbins+d] + sum_{u'<u in s} hist[u'][d]  (each range's own
//       cursor: what PartitionJob's local prefix + the bucket-major/range-minor merge amount to)
// One workgroup (1024 threads = G groups x nbins digits) per segment; nbins <= 1024.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void
dev_scan_units(const u64 *__restrict__ seg_start, const u32 *__restrict__ unit_start, u32 nseg, int bits,
               const u32 *__restrict__ unit_hist, u64 *__restrict__ unit_base, u64 *__restrict__ part_start, u64 n_total,
               const u32 s)
{
    // 64-bit sums throughout: a segment (a whole relation, or one pass-1 bucket of a skewed multi-billion-tuple
    // input) may hold 2^32 tuples or more; only a UNIT's counts (<= L tuples) are 32-bit
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u64 *part = reinterpret_cast<u64 *>(smem);          // [G][nbins] partial sums
    __shared__ u64 wtot[16];
    const u32 nbins = 1u << bits;
    const u32 G = 1024u / nbins;                        // >= 1
    const u32 d = threadIdx.x & (nbins - 1), g = threadIdx.x >> bits;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const u32 us = unit_start[s], ue = unit_start[s + 1];
    const u32 nu = ue - us, per = (nu + G - 1) / G;
    const u32 gb = us + ((g * per < nu) ? g * per : nu);
    const u32 ge = us + (((g + 1) * per < nu) ? (g + 1) * per : nu);

    u64 sum = 0;
    for (u32 u = gb; u < ge; u += 8) {                  // (eight independent loads per step: see dev_scan1_partial)
        u32 h[8];
#pragma unroll
        for (int q = 0; q < 8; q++) h[q] = u + q < ge ? unit_hist[(u64)(u + q) * nbins + d] : 0u;
#pragma unroll
        for (int q = 0; q < 8; q++) sum += h[q];
    }
    part[g * nbins + d] = sum;
    __syncthreads();
    u64 tot_d = 0, before = 0;
    for (u32 k = 0; k < G; k++) {
        const u64 p = part[k * nbins + d];
        tot_d += p;
        if (k < g) before += p;
    }
    // workgroup exclusive scan of the digit totals (held by the threads of group 0)
    const u64 v = g == 0 ? tot_d : 0ull;
    u64 inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const u64 t = __shfl_up(inc, off, 64);
        if (lane >= off) inc += t;
    }
    if (lane == 63) wtot[w] = inc;
    __syncthreads();
    u64 ex = inc - v;
    for (int i = 0; i < w; i++) ex += wtot[i];
    // threads of group 0 hold the digit-exclusive prefix; publish through LDS for the other groups
    if (g == 0) part[d] = ex;       // part[0][*] no longer needed: every thread has read its column
    __syncthreads();
    const u64 pstart = seg_start[s] + part[d];
    if (g == 0) {
        part_start[(u64)s * nbins + d] = pstart;
        if (s == nseg - 1 && d == nbins - 1) part_start[(u64)nseg * nbins] = n_total;
    }
    u64 run = pstart + before;
    for (u32 u = gb; u < ge; u += 8) {
        u32 h[8];
#pragma unroll
        for (int q = 0; q < 8; q++) h[q] = u + q < ge ? unit_hist[(u64)(u + q) * nbins + d] : 0u;
#pragma unroll
        for (int q = 0; q < 8; q++)
            if (u + q < ge) { unit_base[(u64)(u + q) * nbins + d] = run; run += h[q]; }
    }
}

__global__ void __launch_bounds__(1024)
k_scan_units(const u64 *__restrict__ seg_start, const u32 *__restrict__ unit_start, u32 nseg, int bits,
             const u32 *__restrict__ unit_hist, u64 *__restrict__ unit_base, u64 *__restrict__ part_start, u64 n_total)
{
    dev_scan_units(seg_start, unit_start, nseg, bits, unit_hist, unit_base, part_start, n_total, blockIdx.x);
}

__global__ void __launch_bounds__(1024) k_scan_units2(PassPair a, int bits)
{
    const PassRel &x = a.r[blockIdx.x];
    dev_scan_units(x.seg_start, x.unit_start, 1u, bits, x.unit_hist, x.unit_base, x.part_start, x.n, 0u);
}


// K2 for ONE long segment (pass 1: up to ~2048 units): the same result as k_scan_units, computed by
// up to SCAN_SLICES workgroups in three short launches instead of one workgroup walking every unit.  The number of
// slices is ~sqrt(units): the middle kernel walks the slices serially (one dependent load each), the outer two walk
// the units of a slice (measured at 245 units: 64 slices 8.9 us for the middle kernel alone).
constexpr u32 SCAN_SLICES = 64;

__device__ __forceinline__ void slice_range(u32 nu, u32 nsl, u32 k, u32 &ub, u32 &ue)
{
    const u32 per = (nu + nsl - 1) / nsl;
    ub = k * per < nu ? k * per : nu;
    ue = (k + 1) * per < nu ? (k + 1) * per : nu;
}

// partial[k][d] = sum of hist[u][d] over the units of slice k
__device__ __forceinline__ void
dev_scan1_partial(const u32 *__restrict__ unit_start, int bits, const u32 *__restrict__ unit_hist, u64 *__restrict__ partial)
{
    const u32 nbins = 1u << bits, nu = unit_start[1];
    u32 ub, ue;
    slice_range(nu, gridDim.x, blockIdx.x, ub, ue);
    for (u32 d = threadIdx.x; d < nbins; d += 1024) {
        // (eight independent loads per step: these three kernels are nothing but memory latency -- [measured] 20M x 20M, 1024
        // units x 128 digits: 24 / 28 / 19 us each with one load in flight per thread)
        u64 sum = 0;
        for (u32 u = ub; u < ue; u += 8) {
            u32 h[8];
#pragma unroll
            for (int q = 0; q < 8; q++) h[q] = u + q < ue ? unit_hist[(u64)(u + q) * nbins + d] : 0u;
#pragma unroll
            for (int q = 0; q < 8; q++) sum += h[q];
        }
        partial[(u64)blockIdx.x * nbins + d] = sum;
    }
}

__global__ void __launch_bounds__(1024)
k_scan1_partial(const u32 *__restrict__ unit_start, int bits, const u32 *__restrict__ unit_hist, u64 *__restrict__ partial)
{
    dev_scan1_partial(unit_start, bits, unit_hist, partial);
}

__global__ void __launch_bounds__(1024) k_scan1_partial2(PassPair a, int bits)
{
    const PassRel &x = a.r[blockIdx.y];
    dev_scan1_partial(x.unit_start, bits, x.unit_hist, x.scan_tmp);
}

// partial[k][d] <- part_start[d] + sum_{k'<k} partial[k'][d];  part_start[d] = exclusive prefix of the digit totals
__device__ __forceinline__ void
dev_scan1_mid(int bits, u32 nsl, u64 *__restrict__ partial, u64 *__restrict__ part_start, u64 n_total)
{
    __shared__ u64 wtot[16];
    const u32 nbins = 1u << bits;                         // <= 1024
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const u32 d = threadIdx.x;
    u64 tot = 0;
    if (d < nbins)
        for (u32 k = 0; k < nsl; k += 8) {
            u64 c[8];
#pragma unroll
            for (int q = 0; q < 8; q++) c[q] = k + q < nsl ? partial[(u64)(k + q) * nbins + d] : 0ull;
#pragma unroll
            for (int q = 0; q < 8; q++) tot += c[q];
        }
    u64 inc = tot;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const u64 t = __shfl_up(inc, off, 64);
        if (lane >= off) inc += t;
    }
    if (lane == 63) wtot[w] = inc;
    __syncthreads();
    u64 pre = 0;
    for (int i = 0; i < w; i++) pre += wtot[i];
    if (d < nbins) {
        u64 run = pre + inc - tot;
        part_start[d] = run;
        if (d == nbins - 1) part_start[nbins] = n_total;
        for (u32 k = 0; k < nsl; k += 8) {
            u64 c[8];
#pragma unroll
            for (int q = 0; q < 8; q++) c[q] = k + q < nsl ? partial[(u64)(k + q) * nbins + d] : 0ull;
#pragma unroll
            for (int q = 0; q < 8; q++)
                if (k + q < nsl) { partial[(u64)(k + q) * nbins + d] = run; run += c[q]; }
        }
    }
}

__global__ void __launch_bounds__(1024)
k_scan1_mid(int bits, u32 nsl, u64 *__restrict__ partial, u64 *__restrict__ part_start, u64 n_total)
{
    dev_scan1_mid(bits, nsl, partial, part_start, n_total);
}

__global__ void __launch_bounds__(1024) k_scan1_mid2(PassPair a, int bits, u32 nsl)
{
    const PassRel &x = a.r[blockIdx.x];
    dev_scan1_mid(bits, nsl, x.scan_tmp, x.part_start, x.n);
}

__device__ __forceinline__ void
dev_scan1_final(const u32 *__restrict__ unit_start, int bits, const u32 *__restrict__ unit_hist,
                const u64 *__restrict__ partial, u64 *__restrict__ unit_base)
{
    const u32 nbins = 1u << bits, nu = unit_start[1];
    u32 ub, ue;
    slice_range(nu, gridDim.x, blockIdx.x, ub, ue);
    for (u32 d = threadIdx.x; d < nbins; d += 1024) {
        u64 run = partial[(u64)blockIdx.x * nbins + d];
        for (u32 u = ub; u < ue; u += 8) {
            u32 h[8];
#pragma unroll
            for (int q = 0; q < 8; q++) h[q] = u + q < ue ? unit_hist[(u64)(u + q) * nbins + d] : 0u;
#pragma unroll
            for (int q = 0; q < 8; q++)
                if (u + q < ue) { unit_base[(u64)(u + q) * nbins + d] = run; run += h[q]; }
        }
    }
}

__global__ void __launch_bounds__(1024)
k_scan1_final(const u32 *__restrict__ unit_start, int bits, const u32 *__restrict__ unit_hist,
              const u64 *__restrict__ partial, u64 *__restrict__ unit_base)
{
    dev_scan1_final(unit_start, bits, unit_hist, partial, unit_base);
}

__global__ void __launch_bounds__(1024) k_scan1_final2(PassPair a, int bits)
{
    const PassRel &x = a.r[blockIdx.y];
    dev_scan1_final(x.unit_start, bits, x.unit_hist, x.scan_tmp, x.unit_base);
}


// ------------------------------------------------------------------------------------------------
// K3 (tile-sort form), used only for 10-bit passes, whose carry lines do not fit LDS beside a tile.
// PartitionJob::run's scatter (JobScheduler.cpp:170-174) fused with the serial merge-gather of
// structs.cpp:183-194: tuples go straight to their final slot of R'.  Per 4096-tuple tile: coalesced 16 B
// loads -> LDS rank per digit -> tile re-ordered by digit in LDS -> each digit's run written as
// consecutive 16 B stores.  (Its runs start at arbitrary 16 B offsets: 3.4-3.9 TB/s)
//   * the next tile's 16 B/lane loads are issued into a second register set before the current
//     tile is processed (plain loads survive __syncthreads on gfx950), so HBM latency overlaps
//     the LDS phases and the stores of the current tile;
//   * 3 workgroup barriers per tile instead of 7: one wavefront does the nbins-wide exclusive scan
//     and maintains the unit's global write cursors (gbase) and gdelta[d] = gbase[d] - excl[d],
//     so a staged tuple at LDS slot i goes to out[gdelta[digit] + i].
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(PART_THREADS, 4)
k_scatter_units_pipe(const Tup *__restrict__ in, Tup *__restrict__ out, const u64 *__restrict__ seg_start,
                     const u32 *__restrict__ unit_start, u32 nseg, u64 L, int shift, int bits,
                     const u64 *__restrict__ unit_base, int mix)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const u32 nbins = 1u << bits, mask = nbins - 1;
    Tup *tile = reinterpret_cast<Tup *>(smem);                               // PART_TILE * 16 B
    u64 *gbase = reinterpret_cast<u64 *>(smem + (size_t)PART_TILE * 16);     // nbins * 8
    u64 *gdelta = gbase + nbins;                                             // nbins * 8
    u32 *cnt = reinterpret_cast<u32 *>(gdelta + nbins);                      // nbins * 4
    u32 *excl = cnt + nbins;                                                 // nbins * 4

    const u32 u = blockIdx.x;
    if (u >= unit_start[nseg]) return;
    const u32 s = find_segment(unit_start, nseg, u);
    const u64 beg = seg_start[s] + (u64)(u - unit_start[s]) * L;
    const u64 send = seg_start[s + 1];
    const u64 end = (beg + L < send) ? beg + L : send;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u32 bpl = nbins >= 64 ? nbins >> 6 : 1;                            // bins per lane of the scanning wave
    auto dig = [&](u64 p) -> u32 { return (u32)((mix == MIX_DIGIT ? mix64(p) : p) >> shift) & mask; };   // (mix: see dev_scatter_wc)

    for (u32 b = tid; b < nbins; b += PART_THREADS) { gbase[b] = unit_base[(u64)u * nbins + b]; cnt[b] = 0; }
    __syncthreads();

    auto load_tile = [&](Tup (&t)[PART_TPT], u64 tb) {
        const u32 ntile = (end - tb < (u64)PART_TILE) ? (u32)(end - tb) : (u32)PART_TILE;
        const Tup *__restrict__ tp = in + tb;            // wave-uniform base + 32-bit lane offsets
#pragma unroll
        for (int k = 0; k < PART_TPT; k++) {
            const u32 i = k * PART_THREADS + tid;
            if (i < ntile) t[k] = tp[i];
        }
    };
    auto process = [&](Tup (&t)[PART_TPT], u64 tb) {
        const u32 ntile = (end - tb < (u64)PART_TILE) ? (u32)(end - tb) : (u32)PART_TILE;
        u32 rk[PART_TPT];
#pragma unroll
        for (int k = 0; k < PART_TPT; k++) {
            const u32 i = k * PART_THREADS + tid;
            if (i < ntile) {
                if (mix == MIX_STORE) t[k].payload = mix64(t[k].payload);
                rk[k] = atomicAdd(&cnt[dig(t[k].payload)], 1u);
            }
        }
        __syncthreads();
        if (wave == 0) {
            const u32 b0 = lane * bpl;
            u32 loc = 0;
            for (u32 j = 0; j < bpl; j++) if (b0 + j < nbins) loc += cnt[b0 + j];
            u32 ex = wave_incl_scan(loc, lane) - loc;
            for (u32 j = 0; j < bpl; j++) {
                const u32 b = b0 + j;
                if (b < nbins) {
                    const u32 c = cnt[b];
                    const u64 g = gbase[b];
                    excl[b] = ex;
                    gdelta[b] = g - ex;
                    gbase[b] = g + c;
                    cnt[b] = 0;
                    ex += c;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < PART_TPT; k++) {
            const u32 i = k * PART_THREADS + tid;
            if (i < ntile) tile[excl[dig(t[k].payload)] + rk[k]] = t[k];
        }
        __syncthreads();
#pragma unroll
        for (int h = 0; h < PART_TPT; h += 4) {      // two halves: bounds the live registers of this phase
#pragma unroll
            for (int k = h; k < h + 4; k++) {
                const u32 i = k * PART_THREADS + tid;
                if (i < ntile) {
                    const Tup v = tile[i];
                    out[gdelta[dig(v.payload)] + i] = v;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    Tup ta[PART_TPT], tb_[PART_TPT];
    u64 cur = beg;
    if (cur < end) load_tile(ta, cur);
    while (cur < end) {
        u64 nxt = cur + PART_TILE;
        if (nxt < end) load_tile(tb_, nxt);
        process(ta, cur);
        cur = nxt;
        if (cur >= end) break;
        nxt = cur + PART_TILE;
        if (nxt < end) load_tile(ta, nxt);
        process(tb_, cur);
        cur = nxt;
    }
}


// ------------------------------------------------------------------------------------------------
// K3 (line-aligned write-combining form) -- the production scatter kernel.
//
// Measured on MI355X (profiles/): the tile-sort scatter above moves the ideal number of
// bytes but runs at 3.3-3.9 TB/s because every digit run starts and ends at an arbitrary 16 B
// offset, so about half of the 128 B lines it touches are written in two pieces by two different
// tiles, microseconds apart, after the first piece has already left the L2.  The same kernel with
// line-aligned runs sustains 5.2 TB/s at any fan-out.  This form therefore writes ONLY full,
// 128 B-aligned lines (8 tuples), except for the first and last line of each digit of a unit:
//   cb[d][8]   carry line of digit d in LDS: the tuples of the not yet complete line, at slot
//              (global index & 7)
//   per tile   new tuples of digit d get global indices [g0, e).  With a = first line boundary
//              >= g0 and b = last line boundary <= e:
//                 [g0, a)  "head"   -> cb[d] directly (completes the carried line -> flushed as one
//                                      128 B store by 8 adjacent lanes)
//                 [a, b)   "middle" -> staged in LDS in digit order, flushed as whole lines
//                 [b, e)   "tail"   -> staged, then moved to cb[d] after the flush (carried on)
// One workgroup of 1024 threads per CU (tile 64 KiB + carry lines nbins*128 B), next tile's 16 B/lane
// loads prefetched into a second register set, 4 workgroup barriers per tile.
// ------------------------------------------------------------------------------------------------
// (multi-GPU receiver) pass-2 unit u holds tuples of sender (u % ngroups) / div; see k_scatter_wcn
struct WnTag { u32 ngroups, div, bits; };   // bits == 0: no tagging
constexpr u32 TAG_BITS = 4, TAG_MAX = 1u << TAG_BITS;   // sender tags in the low payload bits: <= 16 ranks (== SEG_MAX)
constexpr int FUSE_STRIDE64_FWD = 16;                                       // (= FUSE_STRIDE64, defined with the fused kernels below)
constexpr int WC_THREADS = 1024, WC_TPT = 4;                                // geometry for 9-bit passes (tile = THREADS * WC_TPT)
constexpr int WC_THREADS_SMALL = 512;                                       // <= 8 bits: two workgroups per CU
constexpr int WC_MAX_BITS = 9;

// IN_NARROW (multi-GPU receiver, last pass in front of the compact-table join): the input is a narrow relation whose rowIDs are
// local to the sender's shard; unit u holds tuples of ONE sender (see WnTag) and the 16-byte tuples written carry the GLOBAL
// rowID key_add + rowID32 -- the compact-table kernels then run unchanged on what one GPU would have partitioned itself.
template <int THREADS, bool IN_NARROW = false>
__device__ __forceinline__ void
dev_scatter_wc(const Tup *__restrict__ in, Tup *__restrict__ out, const u64 *__restrict__ seg_start,
               const u32 *__restrict__ unit_start, u32 nseg, u64 L, int shift, int bits,
               const u64 *__restrict__ unit_base, const u64 *__restrict__ unit_rng, u32 n_rng_units, const u32 u,
               const u64 *__restrict__ inP = nullptr, const u32 *__restrict__ inK = nullptr, u64 key_add = 0, int mix = 0,
               const u32 *__restrict__ hist_rows = nullptr, u64 *__restrict__ cursor = nullptr, u64 n_single = 0,
               const u32 *pbase = nullptr)
{
    // cursor (one-pass joins, k_scatter_fused2): no unit_base table and no scan -- the unit reserves its range of every digit
    // with ONE atomicAdd of its histogram row (hist_rows, from k_hist_fused2) on the digit's cursor, which starts every call at
    // zero: the range begins pbase[digit] (LDS: the partition's start, which the workgroup has just derived from the global
    // histogram) + what the atomicAdd returned.  Units then lie in a partition in arrival order: unspecified, like the
    // order inside a partition has always been.  The unit is rows [u * L, (u + 1) * L) of the n_single tuples.
    // mix (16-byte input only, see MIX_* in rhj_internal.h): MIX_STORE -- first pass inside a join: the payload becomes
    // mix64(payload) as it is loaded, and that is what is written; MIX_DIGIT -- the digit comes from mix64(payload), the tuple
    // is written as it came (the multi-GPU owner split of 16-byte tuples)
    constexpr int TILE = THREADS * WC_TPT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const u32 nbins = 1u << bits, mask = nbins - 1;
    Tup *tile = reinterpret_cast<Tup *>(smem);                               // TILE * 16
    Tup *cb = tile + TILE;                                                // nbins * 8 * 16
    u64 *gnext = reinterpret_cast<u64 *>(cb + (size_t)nbins * 8);            // next global index per digit
    u64 *A = gnext + nbins;                                                  // staging slot i -> global index A[d] + i
    u64 *LB = A + nbins;                                                     // carry-line flush: (line base | first slot), ~0 = none
    u32 *cnt = reinterpret_cast<u32 *>(LB + nbins);
    u32 *P = cnt + nbins;                                                    // sexcl | heads << 16 | (g0 & 7) << 20
    u32 *LO = P + nbins;                                                     // first valid slot of the carried line
    u32 *T = LO + nbins;                                                     // staged slots below T[d] are whole lines of digit d
    u32 *mtot = T + nbins;                                                   // staged tuples of this tile
    u32 *wsc = mtot + 4;                                                     // THREADS/64 wave totals

    u64 beg, end;
    if (unit_rng != nullptr) {               // explicit unit ranges (fused two-pass plan: units of pass 2 are
        if (u >= n_rng_units) return;        // the pieces of a bucket written by groups of pass-1 units)
        beg = unit_rng[u];
        end = unit_rng[u + 1];
    } else if (cursor != nullptr) {
        beg = (u64)u * L;
        if (beg >= n_single) return;
        end = (beg + L < n_single) ? beg + L : n_single;
    } else {
        if (u >= unit_start[nseg]) return;
        const u32 s = find_segment(unit_start, nseg, u);
        beg = seg_start[s] + (u64)(u - unit_start[s]) * L;
        const u64 send = seg_start[s + 1];
        end = (beg + L < send) ? beg + L : send;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    auto dig = [&](u64 p) -> u32 { return (u32)((mix == MIX_DIGIT ? mix64(p) : p) >> shift) & mask; };

    // (the first tile is requested BEFORE the cursors are fetched: with cursors reserved by device-scope atomics -- one-pass joins --
    // the two round trips would otherwise follow each other in a workgroup that has one tile to move)
    Tup ta[WC_TPT], tb_[WC_TPT];
    auto load_first = [&](Tup (&t)[WC_TPT]) {
        if (beg >= end) return;
        const u32 last = ((end - beg < (u64)TILE) ? (u32)(end - beg) : (u32)TILE) - 1u;
#pragma unroll
        for (int k = 0; k < WC_TPT; k++) {
            const u32 i = k * THREADS + tid;
            if constexpr (IN_NARROW) { const u64 j = beg + (i < last ? i : last); t[k].payload = inP[j]; t[k].key = key_add + inK[j]; }
            else t[k] = in[beg + (i < last ? i : last)];
        }
    };
    load_first(ta);
    for (u32 b = tid; b < nbins; b += THREADS) {
        const u64 g = cursor != nullptr ? pbase[b] + atomicAdd((unsigned long long *)&cursor[(size_t)b * FUSE_STRIDE64_FWD],
                                                               (unsigned long long)hist_rows[(u64)u * nbins + b])
                                        : unit_base[(u64)u * nbins + b];
        gnext[b] = g;
        LO[b] = (u32)g & 7u;
        cnt[b] = 0;
    }
    __syncthreads();

    // Loads are unconditional (slots past the end of a short tile re-read its last tuple; process() ignores them): with a
    // branch per load the compiler cannot count the loads in flight and waits for vmcnt(0) -- i.e. for the tile it has
    // just prefetched -- before it touches the current one.
    auto load_tile = [&](Tup (&t)[WC_TPT], u64 tb) {
        const u32 last = ((end - tb < (u64)TILE) ? (u32)(end - tb) : (u32)TILE) - 1u;      // tb < end
        const Tup *__restrict__ tp = in + tb;
#pragma unroll
        for (int k = 0; k < WC_TPT; k++) {
            const u32 i = k * THREADS + tid;
            if constexpr (IN_NARROW) { const u64 j = tb + (i < last ? i : last); t[k].payload = inP[j]; t[k].key = key_add + inK[j]; }
            else t[k] = tp[i < last ? i : last];
        }
    };
    // FULL: the tile has TILE tuples (every tile of a unit but its last): no per-tuple range checks
    auto process = [&](Tup (&t)[WC_TPT], u64 tb, auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        const u32 ntile = FULL ? (u32)TILE : (u32)(end - tb);
        u32 rk[WC_TPT], dg[WC_TPT];
        if (mix == MIX_STORE) {
#pragma unroll
            for (int k = 0; k < WC_TPT; k++) t[k].payload = mix64(t[k].payload);
        }
#pragma unroll
        for (int k = 0; k < WC_TPT; k++) {
            const u32 i = k * THREADS + tid;
            dg[k] = dig(t[k].payload);
            if (FULL || i < ntile) rk[k] = atomicAdd(&cnt[dg[k]], 1u);
        }
        __syncthreads();                                                     // B1: counts complete
        {   // plan of this tile, one digit per thread (nbins <= 512 <= THREADS): staged count m per digit,
            // workgroup exclusive scan of m, then the per-digit routing words
            const u32 b = tid;
            u32 c = 0, m = 0;
            u64 g0 = 0, e = 0, a = 0;
            if (b < nbins) {
                c = cnt[b];
                g0 = gnext[b]; e = g0 + c; a = (g0 + 7) & ~7ull;
                m = (e >= a) ? (u32)(e - a) : 0u;
            }
            const u32 inc = wave_incl_scan(m, lane);
            if (lane == 63) wsc[wave] = inc;
            __syncthreads();                                                 // Sx: wave totals visible
            if (b < nbins) {
                u32 sx = inc - m;
                for (int i = 0; i < wave; i++) sx += wsc[i];
                const bool crossed = e >= a;
                const u32 heads = crossed ? (u32)(a - g0) : c;               // tuples that go straight to cb
                P[b] = sx | (heads << 16) | (((u32)g0 & 7u) << 28);
                A[b] = a - sx;
                if (crossed && ((u32)g0 & 7u)) { LB[b] = (a - 8) | LO[b]; LO[b] = 0; }
                else LB[b] = ~0ull;
                gnext[b] = e;
                cnt[b] = 0;
                // staged slot i of digit b has global index (a - sx) + i; it is part of a whole line iff that is below the
                // last line boundary (e & ~7): i < sx + ((e & ~7) - a), when any line was crossed at all
                T[b] = (crossed && (e & ~7ull) > a) ? sx + (u32)((e & ~7ull) - a) : sx;
                if (b == nbins - 1) *mtot = sx + m;
            }
        }
        __syncthreads();                                                     // S2: plan visible
#pragma unroll
        for (int k = 0; k < WC_TPT; k++) {
            const u32 i = k * THREADS + tid;
            if (FULL || i < ntile) {
                const u32 d = dg[k];
                const u32 p = P[d], heads = (p >> 16) & 0xfffu;
                // one LDS store with a selected address (cb and tile are one array: cb = tile + TILE)
                const u32 slot = rk[k] < heads ? (u32)TILE + d * 8 + (p >> 28) + rk[k] : (p & 0xffffu) + rk[k] - heads;
                tile[slot] = t[k];
            }
        }
        __syncthreads();                                                     // D: cb heads + staging complete
        for (u32 q = tid; q < nbins * 8; q += THREADS) {                  // completed carry lines: 8 lanes = one 128 B line
            const u32 d = q >> 3, j = q & 7;
            const u64 x = LB[d];
            if (x != ~0ull && j >= ((u32)x & 7u)) out[(x & ~7ull) + j] = cb[q];
        }
        const u32 mt = *mtot;
        u32 keep = 0;
#pragma unroll
        for (int k = 0; k < WC_TPT; k++) {
            const u32 i = k * THREADS + tid;
            if (i < mt) {
                const Tup v = tile[i];
                const u32 d = dig(v.payload);
                if (i < T[d]) out[A[d] + i] = v;                              // whole lines [a, b)
                else keep |= 1u << k;                                         // tail [b, e): carried on
            }
        }
        __syncthreads();                                                     // F: carry lines read, tails may overwrite them
        if (keep) {
#pragma unroll
            for (int k = 0; k < WC_TPT; k++) {
                if (keep & (1u << k)) {
                    const u32 i = k * THREADS + tid;
                    const Tup v = tile[i];
                    const u32 d = dig(v.payload);
                    cb[d * 8 + ((u32)(A[d] + i) & 7u)] = v;
                }
            }
        }
    };

    u64 cur = beg;
    while (cur < end) {
        // the prefetch is issued on every path (past the unit's end: the current tile again, unused), so that the wait for
        // the current tile is "all but the 4 loads just issued" instead of vmcnt(0)
        u64 nxt = cur + TILE;
        load_tile(tb_, nxt < end ? nxt : cur);
        if (nxt <= end) process(ta, cur, std::true_type{}); else process(ta, cur, std::false_type{});
        cur = nxt;
        if (cur >= end) break;
        nxt = cur + TILE;
        load_tile(ta, nxt < end ? nxt : cur);
        if (nxt <= end) process(tb_, cur, std::true_type{}); else process(tb_, cur, std::false_type{});
        cur = nxt;
    }
    __syncthreads();
    // unit end: the still incomplete line of every digit (shared with the next unit's first line)
    for (u32 q = tid; q < nbins * 8; q += THREADS) {
        const u32 d = q >> 3, j = q & 7;
        const u64 g = gnext[d];
        if (j >= LO[d] && j < ((u32)g & 7u)) out[(g & ~7ull) + j] = cb[q];
    }
}

template <int THREADS>
__global__ void __launch_bounds__(THREADS)
k_scatter_wc(const Tup *__restrict__ in, Tup *__restrict__ out, const u64 *__restrict__ seg_start,
             const u32 *__restrict__ unit_start, u32 nseg, u64 L, int shift, int bits,
             const u64 *__restrict__ unit_base, const u64 *__restrict__ unit_rng, u32 n_rng_units, int mix)
{
    dev_scatter_wc<THREADS>(in, out, seg_start, unit_start, nseg, L, shift, bits, unit_base, unit_rng, n_rng_units, blockIdx.x,
                            nullptr, nullptr, 0, mix);
}

// narrow in (explicit unit ranges, one sender per unit), 16-byte tuples with global rowIDs out
template <int THREADS>
__global__ void __launch_bounds__(THREADS)
k_scatter_wc_n(const u64 *__restrict__ inP, const u32 *__restrict__ inK, Tup *__restrict__ out, int shift, int bits,
               const u64 *__restrict__ unit_base, const u64 *__restrict__ unit_rng, u32 n_rng_units,
               const u64 *__restrict__ key_bases, WnTag tag, const u32 *__restrict__ skip)
{
    if (skip != nullptr && *skip != 0) return;
    const u64 add = key_bases[(blockIdx.x % tag.ngroups) / tag.div];
    dev_scatter_wc<THREADS, true>(nullptr, out, nullptr, nullptr, 0u, 0, shift, bits, unit_base, unit_rng, n_rng_units, blockIdx.x,
                                  inP, inK, add);
}

template <int THREADS>
__global__ void __launch_bounds__(THREADS) k_scatter_wc2(PassPair a, int shift, int bits)
{
    const PassRel &x = a.r[blockIdx.y];
    if (blockIdx.x >= x.max_units) return;
    dev_scatter_wc<THREADS>(x.in, x.out, x.seg_start, x.unit_start, 1u, x.L, shift, bits, x.unit_base, (const u64 *)nullptr, 0u,
                            blockIdx.x, nullptr, nullptr, 0, a.mix);
}

// Which side of a partition becomes the hash table.  The reference builds on the smaller bucket, S on a tie (JobScheduler.cpp:187).
// Here a tie is "within 1/2^tie_shift of each other": the two choices then cost the same to build and to probe, and R is taken --
// in a primary-key / foreign-key join written R JOIN S that is the side without duplicates, whose table answers every probe
// tuple with exactly one match ([measured] timestamps inside k_join_bkt, 10^6 x 10^6: tasks that build on the side with
// duplicates take 26 us, the others 20; with |R_k| ~ |S_k| the old rule made it a coin flip per partition).  Same pairs either way.
__device__ __forceinline__ bool build_on_S(u64 nr, u64 ns, int tie_shift, bool prefer_R)
{
    return prefer_R ? nr >= ns + (ns >> tie_shift) : nr + (nr >> tie_shift) >= ns;
}

// ---- one-pass joins in THREE launches (mid-size joins are launch-bound: 10^6 x 10^6 was 8 dependent launches for 77 us of
// kernel time) ------------------------------------------------------------------------------------------------------------
//   k_hist_fused2     per-unit histograms of R and S (grid.y = relation; unit ranges computed, no unit tables), each row also
//                     added to the relation's global histogram (fire-and-forget atomics; the launch boundary orders them)
//   k_scatter_fused2  the write-combining scatter.  EVERY workgroup turns its relation's global histogram into partition starts
//                     itself (nbins x FUSE_COPIES loads that hit L2 and one workgroup scan: ~2 us, side by side in all
//                     workgroups -- a "last workgroup" doing it once at the end of the histogram launch was ~10 us of one
//                     workgroup with the chip idle), and reserves its output ranges with one atomicAdd per digit on a cursor
//                     that counts from zero.  Workgroup (0, 0) moves no tuples: it is the PLANNER -- both relations'
//                     boundaries to HBM for the join, the JOIN TASK LIST and its counters (what k_init_single_segment2, three
//                     scan launches, k_make_tasks and a memset did), while the others scatter.
//   k_join_bkt        as before; its last workgroup publishes the counters to pinned host memory (no D2H copy)
// FuseCtl lives in HBM in TWO copies used by alternate calls: the histogram launch of a call zeroes the copy of the NEXT call
// (plain stores, nobody else touches it), so no launch ever waits for a clean-up (a failed call makes the host clear both).
struct FuseCtl {
    u32 *ghist;          // [2][FUSE_COPIES][512] global digit histograms of R, S (this call's copy)
    u64 *cursor;         // [2][512] tuples of the digit already placed
    u32 *ghist_next;     // the next call's copy, zeroed by this call's k_hist_fused2
    u64 *cursor_next;
    u32 *sniff, *sniff_next; // [2][SNIFF_SLOTS] counters of the sampled join values (DupSniff), this call's (null: no sampling) and the next's
    int sel_bits[2];
};
struct FuseTasks { u32 probe_split, max_tasks, table_tuples, units_per_wg; JoinTask *tasks; u64 *counters; u64 *host_pub; int tie_shift; SniffVerdict sniff; };
// Every counter on its own 128-byte line, the global histograms in FUSE_COPIES copies (unit u adds to copy u mod FUSE_COPIES):
// device-scope atomics on one LINE are served one behind the other ([measured] 245 units x 256 digits x 2 relations of
// atomics on 16 lines: 75 us for a 9 us histogram), on different lines side by side.
constexpr int FUSE_MAX_BINS = 512, FUSE_COPIES = 4, FUSE_STRIDE32 = 32, FUSE_STRIDE64 = 16;
static_assert(FUSE_STRIDE64 == FUSE_STRIDE64_FWD, "cursor stride");

__global__ void __launch_bounds__(PART_THREADS)
k_hist_fused2(PassPair a, int shift, int bits, FuseCtl fc, FuseTasks ft)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u32 *cnt = reinterpret_cast<u32 *>(smem);                      // nbins
    const PassRel &x = a.r[blockIdx.y];
    const u32 nbins = 1u << bits, mask = nbins - 1;
    const int tid = threadIdx.x;
    {   // the NEXT call's control block, and this call's join counters (k_scatter_fused2's planner and the join write them later)
        const u32 gthreads = gridDim.x * gridDim.y * PART_THREADS, gid = (blockIdx.y * gridDim.x + blockIdx.x) * PART_THREADS + tid;
        for (u32 i = gid; i < 2u * FUSE_COPIES * FUSE_MAX_BINS; i += gthreads) fc.ghist_next[(size_t)i * FUSE_STRIDE32] = 0;
        for (u32 i = gid; i < 2u * FUSE_MAX_BINS; i += gthreads) fc.cursor_next[(size_t)i * FUSE_STRIDE64] = 0;
        for (u32 i = gid; i < 2u * SNIFF_SLOTS; i += gthreads) fc.sniff_next[i] = 0;
        if (gid < 8) ft.counters[gid] = 0;
    }
    DupSniff sn;
    if (fc.sniff != nullptr) {
        sn.tab = fc.sniff + (size_t)blockIdx.y * SNIFF_SLOTS;
        sn.sel_bits = fc.sel_bits[blockIdx.y];
    }
    // a workgroup counts ft.units_per_wg consecutive units (a row each, for the scatter) and adds their SUM to the global
    // histogram once: device-scope atomics are the scarce thing here ([measured] ~7 per ns over the whole chip)
    u32 *wtot_ = cnt + nbins;                                        // this workgroup's sum over its units
    auto dig = [&](u64 p) -> u32 {
        const u64 m = a.mix ? mix64(p) : p;
        if (sn.tab != nullptr) sniff_sample(sn, a.mix ? m : mix64(p));
        return (u32)(m >> shift) & mask;
    };
    const u32 u0 = blockIdx.x * ft.units_per_wg;
    if ((u64)u0 * x.L >= x.n) return;
    for (u32 b = tid; b < nbins; b += PART_THREADS) wtot_[b] = 0;
    for (u32 u = u0; u < u0 + ft.units_per_wg && (u64)u * x.L < x.n; u++) {
        const u64 beg = (u64)u * x.L, end = (beg + x.L < x.n) ? beg + x.L : x.n;
        for (u32 b = tid; b < nbins; b += PART_THREADS) cnt[b] = 0;
        __syncthreads();
        u64 i = beg + tid;
        // eight loads in flight per lane: the launch is at most one 512-thread workgroup per CU (the scatter wants few, large
        // units).  [measured] 8.5 * 10^6 x 8.5 * 10^6, four -> eight: 74 -> 69 us; requesting the next eight before counting the
        // current ones: no further change -- what is left is the LDS atomics, one wavefront instruction per ~90 cycles per CU,
        // the same rate k_hist2d_units runs at with 10^9 tuples
        for (; i + 7ull * PART_THREADS < end; i += 8ull * PART_THREADS) {
            u64 p[8];
#pragma unroll
            for (int q = 0; q < 8; q++) p[q] = x.in[i + (u64)q * PART_THREADS].payload;
#pragma unroll
            for (int q = 0; q < 8; q++) atomicAdd(&cnt[dig(p[q])], 1u);
        }
        for (; i < end; i += PART_THREADS) atomicAdd(&cnt[dig(x.in[i].payload)], 1u);
        __syncthreads();
        u32 *row = x.unit_hist + (u64)u * nbins;
        for (u32 b = tid; b < nbins; b += PART_THREADS) {       // (thread tid owns bins tid, tid + 512: no barrier needed for wtot_)
            const u32 c = cnt[b];
            row[b] = c;
            wtot_[b] += c;
        }
        __syncthreads();
    }
    u32 *gh = fc.ghist + (size_t)(blockIdx.y * FUSE_COPIES + (blockIdx.x & (FUSE_COPIES - 1))) * FUSE_MAX_BINS * FUSE_STRIDE32;
    for (u32 b = tid; b < nbins; b += PART_THREADS) {
        const u32 c = wtot_[b];
        if (c) (void)__hip_atomic_fetch_add(&gh[(size_t)b * FUSE_STRIDE32], c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// grid (1 + units, 2): workgroup (0, 0) plans the join, (0, 1) has nothing to do, (1 + u, rel) scatters unit u of relation rel
template <int THREADS>
__global__ void __launch_bounds__(THREADS) k_scatter_fused2(PassPair a, int shift, int bits, FuseCtl fc, FuseTasks ft)
{
    static_assert(THREADS >= FUSE_MAX_BINS || THREADS == WC_THREADS_SMALL, "one digit per thread");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ u32 pbase[FUSE_MAX_BINS];
    __shared__ u32 wsum[THREADS / 64];
    __shared__ u64 wmax[2][THREADS / 64];
    const u32 nbins = 1u << bits;                                   // <= THREADS (8 bits at most with 512 threads: wc_threads_for)
    const int tid = threadIdx.x;
    auto total_of = [&](int rel) -> u32 {                           // digit tid's tuples in relation rel (the launch boundary made them visible)
        u32 c = 0;
        if ((u32)tid < nbins) {
#pragma unroll
            for (int cp = 0; cp < FUSE_COPIES; cp++) c += fc.ghist[((size_t)(rel * FUSE_COPIES + cp) * FUSE_MAX_BINS + tid) * FUSE_STRIDE32];
        }
        return c;
    };
    if (blockIdx.x != 0) {
        const PassRel &x = a.r[blockIdx.y];
        const u32 u = blockIdx.x - 1;
        if ((u64)u * x.L >= x.n) return;
        u32 tot;
        const u32 ex = block_excl_scan<THREADS>(total_of((int)blockIdx.y), wsum, tot);
        if ((u32)tid < nbins) pbase[tid] = ex;                      // (read back by the same thread: digit b = tid)
        dev_scatter_wc<THREADS>(x.in, x.out, nullptr, nullptr, 1u, x.L, shift, bits, nullptr, (const u64 *)nullptr, 0u, u, nullptr, nullptr, 0,
                                a.mix, x.unit_hist, fc.cursor + (size_t)blockIdx.y * FUSE_MAX_BINS * FUSE_STRIDE64, x.n, pbase);
        return;
    }
    if (blockIdx.y != 0) return;
    // ---- the planner: partition boundaries of R and S, then the join task list (k_make_tasks's rules; one workgroup sees
    // every partition) -------------------------------------------------------------------------------------------------
    u64 *st0 = reinterpret_cast<u64 *>(smem), *st1 = st0 + FUSE_MAX_BINS + 1;      // (the tile area: this workgroup moves no tuples)
    {
        const u32 c0 = total_of(0), c1 = total_of(1);
        u32 tot;
        const u32 e0 = block_excl_scan<THREADS>(c0, wsum, tot);
        const u32 e1 = block_excl_scan<THREADS>(c1, wsum, tot);
        if ((u32)tid < nbins) {
            st0[tid] = e0; a.r[0].part_start[tid] = e0;
            st1[tid] = e1; a.r[1].part_start[tid] = e1;
        }
        if (tid == 0) {
            st0[nbins] = a.r[0].n; a.r[0].part_start[nbins] = a.r[0].n;
            st1[nbins] = a.r[1].n; a.r[1].part_start[nbins] = a.r[1].n;
        }
    }
    __syncthreads();
    const u32 k = tid;
    u64 nr = 0, ns = 0, r0 = 0, s0 = 0;
    if (k < nbins) { r0 = st0[k]; nr = st0[k + 1] - r0; s0 = st1[k]; ns = st1[k + 1] - s0; }
    u64 mr = nr, ms = ns;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const u64 p = __shfl_down(mr, off, 64), q = __shfl_down(ms, off, 64);
        mr = p > mr ? p : mr;
        ms = q > ms ? q : ms;
    }
    if ((tid & 63) == 0) { wmax[0][tid >> 6] = mr; wmax[1][tid >> 6] = ms; }
    __syncthreads();
    u64 maxR = 0, maxS = 0;
    for (int i = 0; i < THREADS / 64; i++) { maxR = wmax[0][i] > maxR ? wmax[0][i] : maxR; maxS = wmax[1][i] > maxS ? wmax[1][i] : maxS; }
    u32 nt = 0, bis = 0;
    u64 pbeg = 0, plen = 0, bbeg = 0, blen = 0;
    const BuildRule rule = sniff_rule(ft.sniff, wsum, ft.tie_shift);      // (wsum: free between the scans)
    if (k < nbins && nr != 0 && ns != 0) {
        const u64 meanR = a.r[0].n / nbins + 1, meanS = a.r[1].n / nbins + 1;
        const bool skewR = maxR > 16 * meanR, skewS = maxS > 16 * meanS;
        bool build_S = build_on_S(nr, ns, rule.shift, rule.prefer_R);         // (+ the skew exception of k_make_tasks)
        if (skewS && !skewR && nr <= 2 * (u64)ft.table_tuples) build_S = false;
        if (skewR && !skewS && ns <= 2 * (u64)ft.table_tuples) build_S = true;
        if (build_S) { pbeg = r0; plen = nr; bbeg = s0; blen = ns; bis = 1; }
        else         { pbeg = s0; plen = ns; bbeg = r0; blen = nr; bis = 0; }
        nt = (u32)((plen + ft.probe_split - 1) / ft.probe_split);
    }
    u32 tot;
    const u32 ex = block_excl_scan<THREADS>(nt, wsum, tot);
    u32 slot = ex;
    for (u32 j = 0; j < nt; j++, slot++) {
        if (slot >= ft.max_tasks) break;
        JoinTask t;
        t.pbeg = pbeg + (u64)j * ft.probe_split;
        const u64 rem = plen - (u64)j * ft.probe_split;
        t.plen = (u32)(rem < ft.probe_split ? rem : ft.probe_split);
        t.part = k;
        t.bbeg = bbeg;
        t.blen = (u32)blen;
        t.build_is_S = bis;
        ft.tasks[slot] = t;
    }
    if (tid == 0) {
        const u64 v[3] = {(u64)(tot < ft.max_tasks ? tot : ft.max_tasks), maxR, maxS};       // ntasks, largest partitions
        for (int i = 0; i < 3; i++) {
            ft.counters[1 + i] = v[i];
            if (ft.host_pub != nullptr) __hip_atomic_store(&ft.host_pub[1 + i], v[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // (see bj_count_packed)
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K2n: the write-combining scatter with a NARROW intermediate format (internal to a join; never seen at the C-ABI).
// A partition pass at 10^9 tuples runs at the HBM copy rate ([measured] tools/copybench: a plain 16 B/tuple copy with
// this kernel's shape takes as long as the scatter), so the only way down is fewer bytes: between the passes of a
// two-pass plan and between the last pass and the join a tuple is {payload 8 B, rowID 4 B} in two arrays (rowIDs of a
// relation of < 2^32 tuples; a larger rowID raises *overflow and the join re-runs in the 16-byte format).
//   P[i] (u64) payloads, K[i] (u32) rowIDs, same index i as the 16-byte layout (part_start is unchanged).
// Writes cost per 128-byte line touched, not per byte ([measured] copybench: 64 B + 32 B pieces per 8 tuples are 25 %
// SLOWER than 128 B lines, 128 B + 64 B per 16 tuples equal, 256 B + 128 B per 32 tuples 17 % faster), so the carry
// line of a digit is GR = 32 tuples: 2 payload lines + 1 rowID line.  Same algorithm as dev_scatter_wc with 8 -> GR;
// staging and carry live in LDS as two arrays (8 B + 4 B per slot); one workgroup of 1024 threads per CU.
//   IN_NARROW: the input is already narrow (pass 2 of a plan whose pass 1 wrote narrow).
// ------------------------------------------------------------------------------------------------
// Two geometries: <GR = 32, 1024 threads x 4> for passes of <= 8 bits (above), and <GR = 16, 896 threads x 4> for 9-bit passes
// (plans of 17-18 bits, beyond 1.1 * 10^9 tuples per side): 512 carry lines of 16 tuples (one payload line + half a rowID line:
// 0.125 lines per tuple, what the 16-byte scatter pays) = 96 KiB + a 3584-slot stage = 158 KiB of LDS.
constexpr int WN_THREADS = 1024, WN_TPT = 4, WN_GR = 32, WN_MAX_BITS = 8;
constexpr int WN9_THREADS = 896, WN9_TPT = 4, WN9_GR = 16, WN9_MAX_BITS = 9;   // 14 wavefronts x 4 tuples: a 3584-slot stage is
                                                                                // what 160 KiB leave beside 512 x 16-tuple carry lines

// key_base (16-byte input): the rowID stored is key - key_base (a shard of a range-sharded relation sends rowIDs local to
//   the shard; the receiver adds the sender's base again in the join, see WnTag).
// tag (explicit unit ranges only): unit u carries tuples of ONE sender (the multi-GPU receiver cuts its pass-1 units at the
//   sender segments of the receive buffer, so group g = u % tag.ngroups of pass-1 units belongs to sender g / tag.div); the
//   low tag.bits bits of a payload -- constant inside the partition from here on, hence dead -- are replaced by that
//   sender number, which the join kernels turn back into a global rowID (row0[sender] + local rowID).

// PEER (multi-GPU sender, rhj_shard_split_peer): digit d's tuples go straight into the receive arrays of the rank that owns
// class d -- peer.P[owner[d]] / peer.K[owner[d]], the peers' HBM mapped into this process -- at the index the receiver's
// layout gives them: the unit cursors are moved by delta[d] = (where this sender's class d starts in its owner's arrays) -
// (where it starts in a local send buffer), so every index below is an index into the OWNER's arrays and the carry lines
// are aligned to the owner's 128-byte lines.  No send buffer, no all-to-all: 12 B written and 12 B read per tuple less.
struct WnPeer {
    const u64 *delta;            // [nbins] (device)
    const unsigned char *owner;  // [nbins] (device): destination rank of every class
    u64 *P[SEG_MAX];             // payload arrays of the ranks
    u32 *K[SEG_MAX];             // rowID arrays of the ranks
};

template <bool IN_NARROW, int GR_ = WN_GR, int TPT_ = WN_TPT, int THREADS_ = WN_THREADS, bool PEER = false>
__global__ void __launch_bounds__(THREADS_)
k_scatter_wcn(const Tup *__restrict__ in, const u64 *__restrict__ inP, const u32 *__restrict__ inK,
              u64 *__restrict__ outP, u32 *__restrict__ outK, const u64 *__restrict__ seg_start,
              const u32 *__restrict__ unit_start, u32 nseg, u64 L, int shift, int bits,
              const u64 *__restrict__ unit_base, const u64 *__restrict__ unit_rng, u32 n_rng_units,
              u32 *__restrict__ overflow, u64 key_base, WnTag tag, int mix, WnPeer peer)
{
    static_assert(!PEER || !IN_NARROW, "the peer split reads the rank's own 16-byte shard");
    // a rowID that does not fit 32 bits has been seen (by the histogram kernel or by an earlier workgroup of this pass):
    // the join is going to repeat itself in the 16-byte format, nothing written from here on will be read
    if (overflow != nullptr && __builtin_nontemporal_load(overflow) != 0) return;
    constexpr int THREADS = THREADS_, TPT = TPT_, TILE = THREADS * TPT, GR = GR_;
    constexpr u64 GM = GR - 1;
    using KeyT = typename std::conditional<IN_NARROW, u32, u64>::type;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const u32 nbins = 1u << bits, mask = nbins - 1;
    u64 *sp = reinterpret_cast<u64 *>(smem);                                 // payloads: TILE staging slots, then nbins carry lines
    u64 *gnext = sp + TILE + (size_t)nbins * GR;                             // next global index per digit
    u64 *A = gnext + nbins;                                                  // staging slot i -> global index A[d] + i
    u64 *LB = A + nbins;                                                     // carry-line flush: (line base | first slot), ~0 = none
    u32 *sk = reinterpret_cast<u32 *>(LB + nbins);                           // rowIDs, same slot numbering as sp
    u32 *cnt = sk + TILE + (size_t)nbins * GR;
    u32 *P = cnt + nbins;                                                    // sexcl | heads << 16 | (g0 & GM) << 24
    u32 *LO = P + nbins;                                                     // first valid slot of the carried line
    u32 *T = LO + nbins;                                                     // staged slots below T[d] are whole lines of digit d
    u32 *mtot = T + nbins;
    u32 *wsc = mtot + 4;
    u32 *own = wsc + THREADS / 64;                                           // PEER: owner rank of every digit
    u64 **pP = reinterpret_cast<u64 **>((reinterpret_cast<uintptr_t>(own + (PEER ? nbins : 0)) + 7) & ~(uintptr_t)7);   // PEER: the ranks' arrays
    u32 **pK = reinterpret_cast<u32 **>(pP + (PEER ? SEG_MAX : 0));

    const u32 u = blockIdx.x;
    u64 beg, end;
    if (unit_rng != nullptr) {
        if (u >= n_rng_units) return;
        beg = unit_rng[u];
        end = unit_rng[u + 1];
    } else {
        if (u >= unit_start[nseg]) return;
        const u32 s = find_segment(unit_start, nseg, u);
        beg = seg_start[s] + (u64)(u - unit_start[s]) * L;
        const u64 send = seg_start[s + 1];
        end = (beg + L < send) ? beg + L : send;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u64 tag_keep = ~(((u64)1 << tag.bits) - 1);                        // tag.bits == 0: keeps every bit
    const u64 tag_or = tag.bits ? (u64)((u % tag.ngroups) / tag.div) : 0ull;
    if constexpr (PEER) {
        for (u32 b = tid; b < nbins; b += THREADS) own[b] = peer.owner[b];
        if (tid < SEG_MAX) { pP[tid] = peer.P[tid]; pK[tid] = peer.K[tid]; }
    }
    // where digit d's tuples are written: the one output relation, or the arrays of the rank that owns class d
    auto dstP = [&](u32 d) -> u64 * { if constexpr (PEER) return pP[own[d]]; else return outP; };
    auto dstK = [&](u32 d) -> u32 * { if constexpr (PEER) return pK[own[d]]; else return outK; };

    for (u32 b = tid; b < nbins; b += THREADS) {
        const u64 g = unit_base[(u64)u * nbins + b] + (PEER ? peer.delta[b] : 0ull);
        gnext[b] = g;
        LO[b] = (u32)(g & GM);
        cnt[b] = 0;
    }
    __syncthreads();

    u32 ovf = 0;                                                             // any rowID >= 2^32 seen (16-byte input only)
    auto load_tile = [&](u64 (&pay)[TPT], KeyT (&key)[TPT], u64 tb) {
        const u32 last = ((end - tb < (u64)TILE) ? (u32)(end - tb) : (u32)TILE) - 1u;      // tb < end
#pragma unroll
        for (int k = 0; k < TPT; k++) {                                      // unconditional loads (see dev_scatter_wc)
            const u32 i0 = k * THREADS + tid, i = i0 < last ? i0 : last;
            if constexpr (IN_NARROW) { pay[k] = inP[tb + i]; key[k] = inK[tb + i]; }
            else { const Tup v = in[tb + i]; pay[k] = v.payload; key[k] = v.key - key_base; }
        }
    };
    auto process = [&](u64 (&pay)[TPT], KeyT (&key)[TPT], u64 tb, auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        const u32 ntile = FULL ? (u32)TILE : (u32)(end - tb);
        u32 rk[TPT], dg[TPT];
        if constexpr (!IN_NARROW) {                                          // first pass inside a join (MIX_STORE): the narrow
            if (mix) {                                                       // format carries mix64(payload) from here on
#pragma unroll
                for (int k = 0; k < TPT; k++) pay[k] = mix64(pay[k]);
            }
        }
#pragma unroll
        for (int k = 0; k < TPT; k++) {
            const u32 i = k * THREADS + tid;
            dg[k] = (u32)(pay[k] >> shift) & mask;
            if (FULL || i < ntile) {
                rk[k] = atomicAdd(&cnt[dg[k]], 1u);
                if constexpr (!IN_NARROW) ovf |= (u32)(key[k] >> 32);
            }
        }
        __syncthreads();                                                     // B1: counts complete
        {
            const u32 b = tid;
            u32 c = 0, m = 0;
            u64 g0 = 0, e = 0, a = 0;
            if (b < nbins) {
                c = cnt[b];
                g0 = gnext[b]; e = g0 + c; a = (g0 + GM) & ~GM;
                m = (e >= a) ? (u32)(e - a) : 0u;
            }
            const u32 inc = wave_incl_scan(m, lane);
            if (lane == 63) wsc[wave] = inc;
            __syncthreads();                                                 // Sx: wave totals visible
            if (b < nbins) {
                u32 sx = inc - m;
                for (int i = 0; i < wave; i++) sx += wsc[i];
                const bool crossed = e >= a;
                const u32 heads = crossed ? (u32)(a - g0) : c;               // tuples that go straight to the carry line (< GR)
                P[b] = sx | (heads << 16) | ((u32)(g0 & GM) << 24);
                A[b] = a - sx;
                if (crossed && (g0 & GM)) { LB[b] = (a - GR) | LO[b]; LO[b] = 0; }
                else LB[b] = ~0ull;
                gnext[b] = e;
                cnt[b] = 0;
                T[b] = (crossed && (e & ~GM) > a) ? sx + (u32)((e & ~GM) - a) : sx;
                if (b == nbins - 1) *mtot = sx + m;
            }
        }
        __syncthreads();                                                     // S2: plan visible
#pragma unroll
        for (int k = 0; k < TPT; k++) {
            const u32 i = k * THREADS + tid;
            if (FULL || i < ntile) {
                const u32 d = dg[k];
                const u32 p = P[d], heads = (p >> 16) & 0xffu;
                const u32 slot = rk[k] < heads ? (u32)TILE + d * GR + (p >> 24) + rk[k] : (p & 0xffffu) + rk[k] - heads;
                sp[slot] = (pay[k] & tag_keep) | tag_or;                     // (the digit of a staged payload is re-derived from bits
                sk[slot] = (u32)key[k];                                      //  >= shift >= tag.bits: untouched by the tag)
            }
        }
        __syncthreads();                                                     // D: carry heads + staging complete
        for (u32 q = tid; q < nbins * GR; q += THREADS) {                    // completed carry lines: GR lanes = 2 + 1 lines
            const u32 d = q / GR, j = q % GR;
            const u64 x = LB[d];
            if (x != ~0ull && j >= (u32)(x & GM)) {
                const u64 o = (x & ~GM) + j;
                __builtin_nontemporal_store(sp[TILE + q], &dstP(d)[o]);
                __builtin_nontemporal_store(sk[TILE + q], &dstK(d)[o]);
            }
        }
        const u32 mt = *mtot;
        u32 keep = 0;
#pragma unroll
        for (int k = 0; k < TPT; k++) {
            const u32 i = k * THREADS + tid;
            if (i < mt) {
                const u64 v = sp[i];
                const u32 d = (u32)(v >> shift) & mask;
                if (i < T[d]) { const u64 o = A[d] + i; __builtin_nontemporal_store(v, &dstP(d)[o]); __builtin_nontemporal_store(sk[i], &dstK(d)[o]); }      // whole lines [a, b)
                else keep |= 1u << k;                                                       // tail [b, e): carried on
            }
        }
        __syncthreads();                                                     // F: carry lines read, tails may overwrite them
        if (keep) {
#pragma unroll
            for (int k = 0; k < TPT; k++) {
                if (keep & (1u << k)) {
                    const u32 i = k * THREADS + tid;
                    const u64 v = sp[i];
                    const u32 kk = sk[i];
                    const u32 d = (u32)(v >> shift) & mask;
                    const u32 c = (u32)TILE + d * GR + (u32)((A[d] + i) & GM);
                    sp[c] = v;
                    sk[c] = kk;
                }
            }
        }
    };

    u64 pa[TPT], pb[TPT];
    KeyT ka[TPT], kb[TPT];
    u64 cur = beg;
    if (cur < end) load_tile(pa, ka, cur);
    while (cur < end) {
        u64 nxt = cur + TILE;                                                // (prefetch on every path: see dev_scatter_wc)
        load_tile(pb, kb, nxt < end ? nxt : cur);
        if (nxt <= end) process(pa, ka, cur, std::true_type{}); else process(pa, ka, cur, std::false_type{});
        cur = nxt;
        if (cur >= end) break;
        nxt = cur + TILE;
        load_tile(pa, ka, nxt < end ? nxt : cur);
        if (nxt <= end) process(pb, kb, cur, std::true_type{}); else process(pb, kb, cur, std::false_type{});
        cur = nxt;
    }
    __syncthreads();
    // unit end: the still incomplete line of every digit (shared with the next unit's first line)
    for (u32 q = tid; q < nbins * GR; q += THREADS) {
        const u32 d = q / GR, j = q % GR;
        const u64 g = gnext[d];
        if (j >= LO[d] && j < (u32)(g & GM)) {
            const u64 o = (g & ~GM) + j;
            __builtin_nontemporal_store(sp[TILE + q], &dstP(d)[o]);
            __builtin_nontemporal_store(sk[TILE + q], &dstK(d)[o]);
        }
    }
    if constexpr (!IN_NARROW) {
        if (__ballot(ovf != 0) != 0 && lane == 0) atomicOr(overflow, 1u);
    }
}

// Contract check of the public stage call rhj_bucket_join before it routes to the compact-table kernel, which compares
// only (payload >> radix_bits): inside every partition the low radix_bits payload bits of both sides must be one value.
// One workgroup per partition (grid-stride); *bad is OR-ed with 1 otherwise.
__global__ void __launch_bounds__(256)
k_check_radix(const Tup *__restrict__ R, const u64 *__restrict__ startR, const Tup *__restrict__ S, const u64 *__restrict__ startS,
              u64 nparts, int radix_bits, u64 *__restrict__ bad)
{
    const u64 mask = ((u64)1 << radix_bits) - 1;
    u64 diff = 0;
    for (u64 k = blockIdx.x; k < nparts; k += gridDim.x) {
        const u64 r0 = startR[k], r1 = startR[k + 1], s0 = startS[k], s1 = startS[k + 1];
        if (r0 == r1 && s0 == s1) continue;
        const u64 ref = r0 < r1 ? R[r0].payload : S[s0].payload;
        for (u64 i = r0 + threadIdx.x; i < r1; i += 256) diff |= (R[i].payload ^ ref) & mask;
        for (u64 i = s0 + threadIdx.x; i < s1; i += 256) diff |= (S[i].payload ^ ref) & mask;
    }
    if (__ballot(diff != 0) != 0 && (threadIdx.x & 63) == 0) atomicOr((unsigned long long *)bad, 1ull);
}

// d_hist[b] = d_start[b+1] - d_start[b]
__global__ void k_diff_hist(const u64 *__restrict__ start, u64 nbins, u64 *__restrict__ hist)
{
    const u64 b = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (b < nbins) hist[b] = start[b + 1] - start[b];
}

// exclusive prefix of a uint64 histogram, one workgroup, any nbins (JobScheduler.cpp:163-169)
__global__ void __launch_bounds__(1024) k_prefix(const u64 *__restrict__ hist, u64 nbins, u64 *__restrict__ start)
{
    __shared__ u64 wtot[16];
    __shared__ u64 carry;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (u64 base = 0; base < nbins; base += 1024) {
        const u64 b = base + threadIdx.x;
        const u64 v = (b < nbins) ? hist[b] : 0;
        u64 inc = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const u64 t = __shfl_up(inc, off, 64);
            if (lane >= off) inc += t;
        }
        if (lane == 63) wtot[w] = inc;
        __syncthreads();
        u64 pre = 0, tot = 0;
        for (int i = 0; i < 16; i++) { const u64 x = wtot[i]; if (i < w) pre += x; tot += x; }
        const u64 c = carry;
        if (b < nbins) start[b] = c + pre + inc - v;
        __syncthreads();
        if (threadIdx.x == 0) carry = c + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) start[nbins] = carry;
}

// ------------------------------------------------------------------------------------------------
// Join task list.  The loop of Result.cpp:98-107 schedules one JoinJob per bucket with both sides
// non-empty; here a partition whose probe side is longer than probe_split is cut into several tasks
// (probe-side load balance under skew).  Build side = S when |R_k| >= |S_k| (JobScheduler.cpp:187).
// One atomic per workgroup reserves its tasks' slots.
// ------------------------------------------------------------------------------------------------
// largest partition of each side (skew indicator for k_make_tasks): stats[0] = max |R_k|, stats[1] = max |S_k|
__global__ void __launch_bounds__(256)
k_part_max(const u64 *__restrict__ startR, const u64 *__restrict__ startS, u64 nparts, u64 *__restrict__ stats)
{
    u64 mr = 0, ms = 0;
    for (u64 k = (u64)blockIdx.x * 256 + threadIdx.x; k < nparts; k += (u64)gridDim.x * 256) {
        const u64 a = startR[k + 1] - startR[k], b = startS[k + 1] - startS[k];
        mr = a > mr ? a : mr;
        ms = b > ms ? b : ms;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const u64 a = __shfl_down(mr, off, 64), b = __shfl_down(ms, off, 64);
        mr = a > mr ? a : mr;
        ms = b > ms ? b : ms;
    }
    if ((threadIdx.x & 63) == 0) { atomicMax(&stats[0], mr); atomicMax(&stats[1], ms); }
}

__global__ void __launch_bounds__(1024)
k_make_tasks(const u64 *__restrict__ startR, const u64 *__restrict__ startS, u64 nparts, u32 probe_split,
             JoinTask *__restrict__ tasks, u32 *__restrict__ ntasks, u32 max_tasks, u64 *__restrict__ stats,
             u32 table_tuples, int own_max, int tie_shift, SniffVerdict sniff)
{
    __shared__ u32 wsum[16];
    __shared__ u32 gbase;
    __shared__ u64 wmax[2][16];
    const u64 k = (u64)blockIdx.x * 1024 + threadIdx.x;
    const BuildRule rule = sniff_rule(sniff, wsum, tie_shift);
    __syncthreads();
    u64 maxR = 0, maxS = 0;
    if (own_max) {                                     // one workgroup covers every partition: k_part_max's job done here
        u64 mr = k < nparts ? startR[k + 1] - startR[k] : 0, ms = k < nparts ? startS[k + 1] - startS[k] : 0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const u64 a = __shfl_down(mr, off, 64), b = __shfl_down(ms, off, 64);
            mr = a > mr ? a : mr;
            ms = b > ms ? b : ms;
        }
        if ((threadIdx.x & 63) == 0) { wmax[0][threadIdx.x >> 6] = mr; wmax[1][threadIdx.x >> 6] = ms; }
        __syncthreads();
        for (int i = 0; i < 16; i++) { maxR = wmax[0][i] > maxR ? wmax[0][i] : maxR; maxS = wmax[1][i] > maxS ? wmax[1][i] : maxS; }
        if (threadIdx.x == 0) { stats[0] = maxR; stats[1] = maxS; }
    } else { maxR = stats[0]; maxS = stats[1]; }
    u32 nt = 0, bis = 0;
    u64 pbeg = 0, plen = 0, bbeg = 0, blen = 0;
    if (k < nparts) {
        const u64 r0 = startR[k], nr = startR[k + 1] - r0, s0 = startS[k], ns = startS[k + 1] - s0;
        if (nr != 0 && ns != 0) {
            // Reference rule: build on the smaller bucket (S when |R_k| >= |S_k|, JobScheduler.cpp:187).  Exception:
            // when ONE relation's partition sizes are heavily skewed (its largest partition > 16x the mean: repeated
            // join values, e.g. a Zipf foreign key) and the other's are balanced, its small partitions are small
            // tables full of duplicates: long buckets that serialise the few probe lanes hitting them.  If the
            // balanced side fits one or two LDS tables, build on it instead: one-compare probes, same pairs.
            const u64 meanR = startR[nparts] / nparts + 1, meanS = startS[nparts] / nparts + 1;
            const bool skewR = maxR > 16 * meanR, skewS = maxS > 16 * meanS;
            bool build_S = build_on_S(nr, ns, rule.shift, rule.prefer_R);
            if (skewS && !skewR && nr <= 2 * (u64)table_tuples) build_S = false;     // at most two build chunks
            if (skewR && !skewS && ns <= 2 * (u64)table_tuples) build_S = true;
            if (build_S) { pbeg = r0; plen = nr; bbeg = s0; blen = ns; bis = 1; }
            else         { pbeg = s0; plen = ns; bbeg = r0; blen = nr; bis = 0; }
            nt = (u32)((plen + probe_split - 1) / probe_split);
            // a task addresses its build range with 32 bits: report instead of truncating (host: RHJ_E_INVALID)
            if (blen >> 32) { nt = 0; atomicMax(&stats[3], blen); }
        }
    }
    u32 tot;
    const u32 ex = block_excl_scan<1024>(nt, wsum, tot);
    if (threadIdx.x == 0) gbase = tot ? atomicAdd(ntasks, tot) : 0u;
    __syncthreads();
    u32 slot = gbase + ex;
    for (u32 j = 0; j < nt; j++, slot++) {
        if (slot >= max_tasks) break;                  // cannot happen: max_tasks is an upper bound
        JoinTask t;
        t.pbeg = pbeg + (u64)j * probe_split;
        const u64 rem = plen - (u64)j * probe_split;
        t.plen = (u32)(rem < probe_split ? rem : probe_split);
        t.part = (u32)k;
        t.bbeg = bbeg;
        t.blen = (u32)blen;              // < 2^32: larger build partitions were reported through stats[3] above
        t.build_is_S = bis;
        tasks[slot] = t;
    }
}

// ------------------------------------------------------------------------------------------------
// K4 (bucketized LDS table): bucket build + probe + result write.
// JoinJob::run + Result::join_buckets (Result.cpp:43-76) + the page appends of add_result/addAll
// (Result.cpp:21-35, 78-84, 111-121).  The reference builds bucket[next_prime(n)] + chain[n]; the result set
// does not depend on the hash (full 64-bit equality), so the table here is shaped by latency on a 256-CU part:
//   * 512 threads and <= 80 KiB LDS: two workgroups per CU overlap each other's global-load and
//     atomic round trips;
//   * the table is the build chunk itself re-ordered by hash bucket (LDS counting sort: per-bucket
//     counts by ds_add_rtn, in-place exclusive scan, place), so there is no head/next storage:
//     probe = read off[h], off[h+1], compare the keys in between (full 64-bit equality);
//   * the task descriptor carries everything (no dependent loads of partition boundaries) and the
//     first probe tile is requested from HBM before the table is built;
//   * output compaction: per probe slot a wavefront ballot + mbcnt prefix when no lane has more
//     than one match (foreign-key case), a shuffle scan otherwise; one 64-lane scan over the
//     (slot, wavefront) totals; one global atomicAdd per 4096-tuple probe tile reserves the pairs.
// ------------------------------------------------------------------------------------------------
constexpr u32 BJ_HEAVY = 24;          // bucket length above which a lone lane asks its wavefront for help
constexpr int BJ_HEAVY_LANES = 16;     // ... unless more than this many lanes of the wavefront are in that position

__device__ __forceinline__ u64 bj_readlane64(u64 v, int srclane)
{
    const u32 lo = __builtin_amdgcn_readlane((u32)v, srclane), hi = __builtin_amdgcn_readlane((u32)(v >> 32), srclane);
    return ((u64)hi << 32) | lo;
}

template <int BBITS>
__device__ __forceinline__ u32 bj_bucket(u64 v, int radix_bits)
{
    // (one 32-bit multiply of the folded value: a 64 x 64-bit product is four quarter-rate multiplies, and this function runs
    // four times per tuple in kernels bound by the instructions they issue)
    const u64 x = v >> radix_bits;
    return (((u32)x ^ (u32)(x >> 32)) * 0x9E3779B1u) >> (32 - BBITS);
}

// Two geometries of the same kernel:
//   <512, 4224, 11, 8>   two workgroups per CU (76 KiB LDS each): partitions that fit one table (the planned case)
//   <1024, 8448, 12, 4>  one workgroup per CU (152 KiB LDS): used when the AVERAGE build partition exceeds 4224 tuples
//                        (explicit plans such as 8+8 bits at 10^9 tuples): half as many build chunks, so half as
//                        many re-probes of the probe side
// DIRECT: no task list -- the inputs are ONE unpartitioned pair of relations and workgroup b probes tuples
// [b * dsplit, (b+1) * dsplit) of the probe side against the whole build side (small joins: the launch sequence
// k_part_max / k_make_tasks / task-list read would cost more than the join itself).
// host_count / done (optional): the LAST workgroup to finish copies the result count to pinned host memory and resets the
// device counters for the next call -- with the pairs written straight into pinned host memory too, a small join needs no
// device-to-host copy and no memset at all (H2D, H2D, kernel, synchronise).
// host_out / host_cap: the first host_cap pairs are ALSO stored in pinned host memory (same positions as in `out`), so a
// result that outgrows the landing zone needs no second run: the rest is fetched from `out`.
struct DirectJoin { u32 nb, np, build_is_S, split; u64 *host_count; u32 *done; Pair *host_out; u64 host_cap; };

// Task-list launches (not DIRECT) with dj.host_count set: the LAST workgroup of the launch copies the seven join counters
// (count, ntasks, largest partitions, error words) to pinned host memory, so that the host learns the result count from its
// stream synchronisation alone, without a device-to-host copy behind the kernel (one-pass joins: ~10 us of a ~90 us join).
__device__ __forceinline__ void bj_publish(const DirectJoin &dj, const u64 *__restrict__ counters)
{
    // No fences: every atomicAdd of this workgroup on the result counter has RETURNED (its value placed the pairs) before
    // the ticket is taken, the last workgroup reads the counters with device-scope atomic loads, and the host reads the pinned
    // block only after the stream has synchronised (the end of a kernel releases at system scope).  (A fence per workgroup is
    // an L2 write-back of the pairs it has just stored: [measured] 35 -> 52 us for the 10^6 x 10^6 join.)
    // (the first wavefront: lane 0 takes the ticket, seven lanes copy one counter each -- seven device-scope loads one behind the
    // other were ~5 us at the end of a 35 us kernel)
    if (dj.host_count == nullptr || threadIdx.x >= 64) return;
    u32 ticket = 0;
    if (threadIdx.x == 0) ticket = __hip_atomic_fetch_add(dj.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ticket = (u32)__builtin_amdgcn_readfirstlane((int)ticket);
    if (ticket == gridDim.x - 1) {
        if (threadIdx.x < 7)
            __hip_atomic_store(&dj.host_count[threadIdx.x], __hip_atomic_load(&counters[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (threadIdx.x == 0) __hip_atomic_store(dj.done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// PACKED result counter (one-pass joins; dj.host_count set, dj.done null): counters[0] = pairs | (workgroups that are done) << 48.
// The LAST atomicAdd a workgroup makes on the counter -- the one that reserves the range of its last tile's pairs -- also counts
// the workgroup as done, so the workgroup whose add returns `grid - 1` in the upper bits knows the final count (what came back
// + its own) and stores it to the pinned host word: no ticket atomic and no read-back of the counter behind the last tile
// ([measured] two device round trips, ~5 us of a 36 us kernel at 10^6 x 10^6).  The host asks for it only when the grid is
// below 2^16 workgroups and nR * nS < 2^48; the counters are zeroed by k_hist_fused2 of the same call.  Thread 0 only.
constexpr int BJ_PK_SHIFT = 48;
__device__ __forceinline__ u64 bj_count_packed(const DirectJoin &dj, u64 *__restrict__ counter, u32 add, bool last)
{
    const u64 v = atomicAdd((unsigned long long *)counter, (unsigned long long)add + (last ? 1ull << BJ_PK_SHIFT : 0ull));
    const u64 base = v & ((1ull << BJ_PK_SHIFT) - 1);
    if (last && (v >> BJ_PK_SHIFT) == (u64)gridDim.x - 1)
        __hip_atomic_store(&dj.host_count[0], base + add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return base;
}

// A partitioned relation as the join kernels read it: 16-byte tuples, or the narrow {payload 8 B, rowID 4 B} arrays that
// k_scatter_wcn writes (NARROW: the build phase then reads 8 B per tuple and the rowID re-fetch 4 B instead of 16 + 16)
// Buffer descriptors over a partition slice (k_join_ct): a load is then `descriptor + lane offset (one VGPR for all slot rows)
// + scalar row offset`, with no vector instruction per load for the address, and a lane beyond `n` tuples reads 0 (the
// range check of the hardware) -- against an add, a compare, a select and a 64-bit address per load with flat addresses.
typedef u32 v2u32 __attribute__((ext_vector_type(2)));
typedef u32 v4u32 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_srd(const void *base, u32 bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000);
}
template <bool NARROW> struct RelView;
template <> struct RelView<false> {
    typedef u64 Rid;
    struct Both { u64 key, payload; };
    const Tup *__restrict__ t;
    __device__ __forceinline__ RelView at(u64 off) const { return RelView{t + off}; }
    __device__ __forceinline__ u64 payload(u32 i) const { return t[i].payload; }
    __device__ __forceinline__ Rid rowid(u32 i) const { return t[i].key; }
    __device__ __forceinline__ Both both(u32 i) const { const Tup v = t[i]; return Both{v.key, v.payload}; }
    struct Buf {                                           // n tuples from tuple `first` on; row = first tuple of a slot row
        __amdgpu_buffer_rsrc_t d;
        __device__ __forceinline__ u64 payload(u32 row, u32 tid) const
        { const v2u32 v = __builtin_amdgcn_raw_buffer_load_b64(d, (int)(tid * 16u + 8u), (int)(row * 16u), 0); return (u64)v.x | ((u64)v.y << 32); }
        __device__ __forceinline__ Rid rowid(u32 row, u32 tid) const
        { const v2u32 v = __builtin_amdgcn_raw_buffer_load_b64(d, (int)(tid * 16u), (int)(row * 16u), 0); return (u64)v.x | ((u64)v.y << 32); }
        __device__ __forceinline__ Both both(u32 row, u32 tid) const
        { const v4u32 v = __builtin_amdgcn_raw_buffer_load_b128(d, (int)(tid * 16u), (int)(row * 16u), 0);
          return Both{(u64)v.x | ((u64)v.y << 32), (u64)v.z | ((u64)v.w << 32)}; }
    };
    __device__ __forceinline__ Buf buf(u32 first, u32 n) const { return Buf{make_srd(t + first, n * 16u)}; }
};
template <> struct RelView<true> {
    typedef u32 Rid;
    struct Both { u32 key; u64 payload; };
    const u64 *__restrict__ p;
    const u32 *__restrict__ k;
    __device__ __forceinline__ RelView at(u64 off) const { return RelView{p + off, k + off}; }
    __device__ __forceinline__ u64 payload(u32 i) const { return p[i]; }
    __device__ __forceinline__ Rid rowid(u32 i) const { return k[i]; }
    __device__ __forceinline__ Both both(u32 i) const { return Both{k[i], p[i]}; }
    struct Buf {
        __amdgpu_buffer_rsrc_t dp, dk;
        __device__ __forceinline__ u64 payload(u32 row, u32 tid) const
        { const v2u32 v = __builtin_amdgcn_raw_buffer_load_b64(dp, (int)(tid * 8u), (int)(row * 8u), 0); return (u64)v.x | ((u64)v.y << 32); }
        __device__ __forceinline__ Rid rowid(u32 row, u32 tid) const
        { return __builtin_amdgcn_raw_buffer_load_b32(dk, (int)(tid * 4u), (int)(row * 4u), 0); }
        __device__ __forceinline__ Both both(u32 row, u32 tid) const { return Both{rowid(row, tid), payload(row, tid)}; }
    };
    __device__ __forceinline__ Buf buf(u32 first, u32 n) const { return Buf{make_srd(p + first, n * 8u), make_srd(k + first, n * 4u)}; }
};

// TAGGED (multi-GPU receiver, NARROW only): the low TAG_BITS bits of every payload hold the number of the rank the tuple
// came from (written by the last partition pass, k_scatter_wcn's WnTag) and its rowID is local to that rank's shard: the
// rowID a pair reports is tag_base[side][tag] + rowID32 (tag_base: 2 x 16 u64, R's bases then S's).  Payloads are compared
// with the tag bits forced to 1 on both sides.  The payload stays in registers in this kernel, so the tags cost nothing;
// the compact-table kernel has no register to carry them in (three more VGPRs through its probe phase spill 44; fetching
// them again costs a second pass over the build payloads: measured 20 against 10 ms at 10^9 tuples): partitions for that
// kernel get their global rowIDs from the last partition pass instead (k_scatter_wc<.., IN_NARROW>, 16-byte tuples out).
// skip (optional): a device word that is non-zero when this join is going to be repeated in another format.
// BATCH (DIRECT only): ONE launch runs up to 16 independent small joins, blockIdx.y = join.  Everything a DIRECT launch gets
// as kernel arguments comes from batch[blockIdx.y] instead (BatchJoinDesc, rhj_internal.h); every join publishes its own
// count and leaves its own counters zeroed, as a DIRECT launch does.  (rhj_join_batch: MainScheduler's "several queries at
// once", MainScheduler.cpp:6-30, for joins so small that launch and copy latencies are their whole cost.)
template <int THREADS, int CHUNK, int BBITS, int EPT, bool DIRECT, bool NARROW = false, bool TAGGED = false, bool BATCH = false>
__global__ void __launch_bounds__(THREADS, 4)
k_join_bkt(RelView<NARROW> R, RelView<NARROW> S, const JoinTask *__restrict__ tasks,
           const u32 *__restrict__ ntasks, int radix_bits, Pair *__restrict__ out, u64 out_capacity,
           u64 *__restrict__ out_count, DirectJoin dj, const u64 *__restrict__ tag_base = nullptr,
           const u32 *__restrict__ skip = nullptr, const BatchJoinDesc *__restrict__ batch = nullptr)
{
    static_assert(!TAGGED || (NARROW && !DIRECT), "sender tags exist in the narrow format only");
    static_assert(!BATCH || (DIRECT && !NARROW), "batched launches are direct joins of 16-byte tuples");
    if (skip != nullptr && *skip != 0) return;
    u32 nblocks = gridDim.x;                                                 // workgroups of THIS join
    if constexpr (BATCH) {
        const BatchJoinDesc b = batch[blockIdx.y];
        if (blockIdx.x >= b.nblocks) return;
        nblocks = b.nblocks;
        R = RelView<false>{(const Tup *)b.R};
        S = RelView<false>{(const Tup *)b.S};
        out = (Pair *)b.out;
        out_capacity = b.cap;
        out_count = b.count;
        dj.nb = b.nb; dj.np = b.np; dj.build_is_S = b.build_is_S; dj.split = b.split;
        dj.host_count = b.host_count; dj.done = b.done; dj.host_out = (Pair *)b.host_out; dj.host_cap = b.host_cap;
    }
    constexpr u64 TM = TAGGED ? (u64)(TAG_MAX - 1) : 0ull;                    // compare payloads with these bits set
    constexpr int NB = 1 << BBITS;
    constexpr int NW = THREADS / 64;
    constexpr int TILE = THREADS * EPT;
    constexpr int BPT = (CHUNK + THREADS - 1) / THREADS;            // build tuples per thread
    static_assert(EPT * NW == 64, "the (slot, wavefront) totals are scanned by one 64-lane wavefront");
    static_assert(CHUNK + 4 < 16384, "bucket start, match mask and match count of a probe slot share one register");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u64 *keys = reinterpret_cast<u64 *>(smem);                               // CHUNK * 8
    u64 *rids = keys + CHUNK;                                             // CHUNK * 8
    u32 *off = reinterpret_cast<u32 *>(rids + CHUNK);                     // NB + 1 (+ pad to 16 B)
    u32 *wtot = off + NB + 4;                                                // 64: [slot][wave] match totals
    u32 *wsum = wtot + 64;                                                   // NW scan scratch
    u64 *gres = reinterpret_cast<u64 *>(wsum + NW);                          // 1
    u64 *tbase = gres + 1;                                                   // TAGGED: 2 x TAG_MAX rowID bases (R's, then S's)

    JoinTask task;
    if (DIRECT) {
        task.pbeg = (u64)blockIdx.x * dj.split;
        task.plen = dj.np - (u32)task.pbeg < dj.split ? dj.np - (u32)task.pbeg : dj.split;
        task.bbeg = 0;
        task.blen = dj.nb;
        task.build_is_S = dj.build_is_S;
    } else {
        task = tasks[blockIdx.x];                                            // (the list has gridDim.x slots: both loads in flight together)
        const u32 nt = *ntasks;
        if (blockIdx.x >= nt) {
            if (dj.host_count != nullptr && dj.done == nullptr) { if (threadIdx.x == 0) (void)bj_count_packed(dj, out_count, 0u, true); }
            else bj_publish(dj, out_count);
            return;
        }
    }
    const bool packed = !DIRECT && dj.host_count != nullptr && dj.done == nullptr;
    const bool build_is_S = task.build_is_S != 0;
    typedef typename RelView<NARROW>::Both Both;
    const RelView<NARROW> B = (build_is_S ? S : R).at(task.bbeg);
    const RelView<NARROW> P = (build_is_S ? R : S).at(task.pbeg);
    const u32 nb = task.blen, np = task.plen;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (TAGGED && tid < (int)(2 * TAG_MAX)) tbase[tid] = tag_base[tid];     // (visible after the first barrier of the build)
    const u32 btoff = build_is_S ? TAG_MAX : 0u, ptoff = build_is_S ? 0u : TAG_MAX;
    // rowID a pair reports for a probe tuple
    auto probe_rowid = [&](const Both &t) -> u64 { return TAGGED ? (u64)t.key + tbase[ptoff + ((u32)t.payload & (u32)TM)] : (u64)t.key; };

    // Loads go through buffer descriptors (see RelView::Buf): no address arithmetic and no branch per load, lanes past the
    // end read 0 and are ignored by the `i < n` tests below.  (np * 16 < 2^32: a task's probe side is at most DIRECT_MAX_PROBE.)
    const typename RelView<NARROW>::Buf PB = P.buf(0, np);
    // first probe tile: in flight while the table is built
    Both p[EPT];
#pragma unroll
    for (int k = 0; k < EPT; k++) p[k] = PB.both((u32)k * THREADS, (u32)tid);

    for (u32 cb = 0; cb < nb; cb += CHUNK) {
        const u32 nc = (nb - cb < (u32)CHUNK) ? nb - cb : (u32)CHUNK;
        // ---- build: counting sort of the chunk by hash bucket ---------------------------------
        for (u32 h = tid; h <= NB; h += THREADS) off[h] = 0;
        Both bt[BPT];
        u32 br[BPT];
        const typename RelView<NARROW>::Buf BBf = B.buf(cb, nc);
#pragma unroll
        for (int k = 0; k < BPT; k++) bt[k] = BBf.both((u32)k * THREADS, (u32)tid);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < BPT; k++) {
            const u32 i = (u32)k * THREADS + tid;
            if (i < nc) br[k] = atomicAdd(&off[bj_bucket<BBITS>(bt[k].payload, radix_bits)], 1u);
        }
        __syncthreads();
        {   // in-place exclusive scan of the NB bucket counts (NB / THREADS consecutive buckets per thread)
            constexpr int PER = NB / THREADS;
            u32 c[PER], loc = 0;
#pragma unroll
            for (int j = 0; j < PER; j++) { c[j] = off[tid * PER + j]; loc += c[j]; }
            u32 tot;
            u32 ex = block_excl_scan<THREADS>(loc, wsum, tot);
#pragma unroll
            for (int j = 0; j < PER; j++) { off[tid * PER + j] = ex; ex += c[j]; }
            if (tid == THREADS - 1) off[NB] = ex;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < BPT; k++) {
            const u32 i = (u32)k * THREADS + tid;
            if (i < nc) {
                const u32 pos = off[bj_bucket<BBITS>(bt[k].payload, radix_bits)] + br[k];
                keys[pos] = bt[k].payload | TM;
                rids[pos] = TAGGED ? (u64)bt[k].key + tbase[btoff + ((u32)bt[k].payload & (u32)TM)] : (u64)bt[k].key;
            }
        }
        __syncthreads();

        // ---- probe ------------------------------------------------------------------------------
        for (u32 tb = 0; tb < np; tb += TILE) {
            if (tb != 0 || cb != 0) {
#pragma unroll
                for (int k = 0; k < EPT; k++) p[k] = PB.both(tb + (u32)k * THREADS, (u32)tid);
            }
            // The first four entries of a probe tuple's bucket are read unconditionally and together (entries past the bucket's end
            // belong to the next bucket or to the rowID array behind the table: masked by len) and WHICH of them match is kept, so
            // the write phase below neither hashes nor compares again; only a longer bucket is walked entry by entry.
            // A slot's result in one register: bucket start (14 bits) | which of the first four entries match (4) | matches beyond
            // them, or the whole count of a bucket scanned by the wavefront together (14).
            u32 cf[EPT];
            u32 coopbits = 0;                                                 // slots whose heavy buckets this wavefront scans together
            auto matches = [](u32 v) -> u32 { return (u32)__popc((v >> 14) & 15u) + (v >> 18); };
#pragma unroll
            for (int k = 0; k < EPT; k++) {
                const u32 i = tb + (u32)k * THREADS + tid;
                u32 lo = 0, len = 0;
                if (i < np) {
                    const u32 h = bj_bucket<BBITS>(p[k].payload, radix_bits);
                    lo = off[h]; len = off[h + 1] - lo;
                }
                const u64 pk = p[k].payload | TM;
                const u64 e0 = keys[lo], e1 = keys[lo + 1], e2 = keys[lo + 2], e3 = keys[lo + 3];
                u32 m4 = (len > 0 && e0 == pk ? 1u : 0u) | (len > 1 && e1 == pk ? 2u : 0u) | (len > 2 && e2 == pk ? 4u : 0u) |
                         (len > 3 && e3 == pk ? 8u : 0u);
                u32 xc = 0;
                if (__ballot(len > 4)) {                                      // (wavefront-uniform) somebody's bucket goes on
                    // A few lanes facing a long bucket (duplicate-heavy build side, e.g. Zipf FK as build) would
                    // serialise the whole workgroup: those buckets are scanned by all 64 lanes together.
                    unsigned long long heavy = __ballot(len > BJ_HEAVY);
                    const bool coop = heavy != 0 && __popcll(heavy) <= BJ_HEAVY_LANES;
                    if (!coop || len <= BJ_HEAVY)
                        for (u32 j = lo + 4; j < lo + len; j++) xc += keys[j] == pk ? 1u : 0u;
                    if (coop) {
                        coopbits |= 1u << k;
                        while (heavy) {
                            const int leader = __ffsll((long long)heavy) - 1;
                            heavy &= heavy - 1;
                            const u64 key = bj_readlane64(pk, leader);
                            const u32 l = __builtin_amdgcn_readlane(lo, leader), hh = l + __builtin_amdgcn_readlane(len, leader);
                            u32 tot = 0;
                            for (u32 j = l; j < hh; j += 64) {
                                const bool m = (j + lane < hh) && keys[j + lane] == key;
                                tot += (u32)__popcll(__ballot(m));
                            }
                            if (lane == leader) { m4 = 0; xc = tot; }
                        }
                    }
                }
                cf[k] = lo | (m4 << 14) | (xc << 18);
            }
            // per slot: the wavefront's matches.  (No prefix over the lanes: the pairs of a slot are written ROUND-MAJOR -- first every
            // lane's first match, side by side, then every lane's second ... -- so a lane's position in a round is its rank among
            // the lanes that still have a match (ballot + mbcnt), and every store instruction of the wavefront covers one
            // contiguous range.  [measured] lane-major positions (a lane's matches side by side, lanes by prefix sum): tasks that
            // build on the side with duplicates -- half the tasks of 10^6 x 10^6 -- spent 9.3 us writing pairs against 1.7-4.4 us
            // for tasks with one match per probe tuple, and set the kernel's time.)
            u32 dup = 0;
#pragma unroll
            for (int k = 0; k < EPT; k++) dup |= matches(cf[k]);
            if (__ballot(dup > 1) == 0) {                                     // foreign-key case
#pragma unroll
                for (int k = 0; k < EPT; k++) {
                    const unsigned long long m = __ballot(cf[k] >> 14);
                    if (lane == 0) wtot[k * NW + w] = (u32)__popcll(m);
                }
            } else if (__ballot(dup > 7) == 0) {                              // a few duplicates: three bit planes per slot
#pragma unroll
                for (int k = 0; k < EPT; k++) {
                    const u32 c = matches(cf[k]);
                    u32 tot = 0;
#pragma unroll
                    for (int b = 0; b < 3; b++) tot += (u32)__popcll(__ballot((c >> b) & 1u)) << b;
                    if (lane == 0) wtot[k * NW + w] = tot;
                }
            } else {
#pragma unroll
                for (int k = 0; k < EPT; k++) {
                    const u32 tot = __shfl(wave_incl_scan(matches(cf[k]), lane), 63, 64);
                    if (lane == 0) wtot[k * NW + w] = tot;
                }
            }
            __syncthreads();
            // every wavefront scans the 64 (slot, wave) totals itself: no second barrier
            const u32 mine = wtot[lane];
            const u32 inc64 = wave_incl_scan(mine, lane);
            const u32 tile_total = __shfl(inc64, 63, 64);
            const bool last_tile = packed && cb + (u32)CHUNK >= nb && tb + (u32)TILE >= np;
            if (tid == 0 && (tile_total || last_tile))
                *gres = packed ? bj_count_packed(dj, out_count, tile_total, last_tile) : atomicAdd(out_count, (u64)tile_total);
            __syncthreads();
            if (tile_total && out != nullptr) {
                const u64 g = *gres;
                const u32 wu = (u32)__builtin_amdgcn_readfirstlane(w);
                auto put = [&](u64 o, u64 prow, u64 brow) {
                    if (o < out_capacity) {
                        Pair pr;
                        if (build_is_S) { pr.r = prow; pr.s = brow; }         // orderFlag, Result.cpp:64-68
                        else            { pr.r = brow; pr.s = prow; }
                        out[o] = pr;
                        if (DIRECT && o < dj.host_cap) dj.host_out[o] = pr;
                    }
                };
                auto rank_in = [&](unsigned long long m) -> u32 { return __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u)); };
                u32 sb[EPT];                                                  // (wavefront-uniform) the slot's next free pair, relative to g
#pragma unroll
                for (int k = 0; k < EPT; k++) sb[k] = (u32)__builtin_amdgcn_readlane((int)(inc64 - mine), k * NW + wu);
                // the matches among the first four entries need no second look at the keys: one round per match, every slot of
                // every lane writing its next one (ONE round in a foreign-key join)
                for (;;) {
                    u32 more = 0;
#pragma unroll
                    for (int k = 0; k < EPT; k++) {
                        const u32 m4 = (coopbits >> k) & 1u ? 0u : (cf[k] >> 14) & 15u;
                        const unsigned long long bal = __ballot(m4 != 0);
                        if (m4) {
                            put(g + sb[k] + rank_in(bal), probe_rowid(p[k]), rids[(cf[k] & 0x3fffu) + (u32)__builtin_ctz(m4)]);
                            cf[k] &= ~((m4 & (0u - m4)) << 14);
                        }
                        sb[k] += (u32)__popcll(bal);
                        more |= m4 & (m4 - 1);
                    }
                    if (!__ballot(more != 0)) break;
                }
#pragma unroll
                for (int k = 0; k < EPT; k++) {
                    const u32 lo = cf[k] & 0x3fffu;
                    if (!((coopbits >> k) & 1u)) {                            // (wavefront-uniform)
                        u32 xc = cf[k] >> 18;
                        if (!__ballot(xc != 0)) continue;
                        u32 j = lo + 4;                                       // the rest of a longer bucket, a round per match again
                        for (;;) {
                            const unsigned long long bal = __ballot(xc != 0);
                            if (!bal) break;
                            if (xc) {
                                while (keys[j] != (p[k].payload | TM)) j++;   // (xc matches lie ahead: counted in step C)
                                put(g + sb[k] + rank_in(bal), probe_rowid(p[k]), rids[j]);
                                j++;
                                xc--;
                            }
                            sb[k] += (u32)__popcll(bal);
                        }
                        continue;
                    }
                    // a slot with buckets scanned by the whole wavefront: lane-major positions
                    const u32 c = matches(cf[k]);
                    u64 o = g + sb[k] + (wave_incl_scan(c, lane) - c);
                    u32 hi = 0;
                    if (c) hi = off[bj_bucket<BBITS>(p[k].payload, radix_bits) + 1];
                    const bool is_heavy = hi > lo + BJ_HEAVY;
                    unsigned long long heavy = __ballot(is_heavy);
                    if (c && !is_heavy) {
                        for (u32 j = lo; j < hi; j++)
                            if (keys[j] == (p[k].payload | TM)) put(o++, probe_rowid(p[k]), rids[j]);
                    }
                    const unsigned long long lt = (1ull << lane) - 1ull;
                    while (heavy) {
                        const int leader = __ffsll((long long)heavy) - 1;
                        heavy &= heavy - 1;
                        const u64 key = bj_readlane64(p[k].payload | TM, leader);
                        const u64 pkey = bj_readlane64(probe_rowid(p[k]), leader);
                        u64 ob = bj_readlane64(o, leader);
                        const u32 l = __builtin_amdgcn_readlane(lo, leader), hh = __builtin_amdgcn_readlane(hi, leader);
                        for (u32 j = l; j < hh; j += 64) {
                            const bool m = (j + lane < hh) && keys[j + lane] == key;
                            const unsigned long long bal = __ballot(m);
                            if (m) put(ob + (u64)__popcll(bal & lt), pkey, rids[j + lane]);
                            ob += (u64)__popcll(bal);
                        }
                    }
                }
            }
            // wtot / gres are rewritten only after the next tile's first barrier: safe without another one
        }
        __syncthreads();         // the table is rebuilt by the next chunk
    }
    if (packed) {
        if (tid == 0 && (nb == 0 || np == 0)) (void)bj_count_packed(dj, out_count, 0u, true);      // (no tile ran: a task has both sides, but)
    } else if (!DIRECT && dj.host_count != nullptr) {
        __syncthreads();
        bj_publish(dj, out_count);
    }
    if (DIRECT && dj.host_count != nullptr) {
        __syncthreads();                                                     // every store / atomic of this workgroup has been issued
        if (tid == 0) {
            __threadfence_system();                                          // ... and is visible before `done` says so
            if (atomicAdd(dj.done, 1u) == nblocks - 1) {                     // last workgroup of the launch (BATCH: of this join)
                __threadfence_system();
                *dj.host_count = atomicExch(out_count, 0ull);                // publish, and leave the counters zeroed
                *dj.done = 0;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K4 (compact-table form): bucket join for partitions whose build side does not fit a 16 B/tuple LDS table but
// whose radix plan has removed >= 16 payload bits (BASELINE config 3: 8+8 bits at 10^9 tuples, 15.3 K-tuple
// partitions).  Same job as k_join_bkt (JoinJob::run + Result::join_buckets, Result.cpp:43-76, + add_result /
// addAll), organised so that BOTH sides are read from HBM exactly once and every tuple is inserted / probed once:
//
//   * inside a partition all payloads share their low radix_bits bits, so (payload >> radix_bits) < 2^48 decides
//     equality: a table entry is 8 bytes, {48-bit key | 16-bit arrival index of the build tuple}.  16352 entries +
//     16384 bucket offsets fill the 160 KiB LDS of one workgroup per CU: the whole 15.3 K-tuple build side is ONE
//     table (k_join_bkt needs two 8448-tuple chunks and re-reads the probe side per chunk: 64 GB moved for 48 GB
//     algorithmic, 2.0 TB/s, round 1).  The bucket count is what the probe phase pays for: a wavefront walks its 256
//     buckets of a tile in lock step, so a tile costs as many compare rounds as its LONGEST bucket has entries --
//     7 with 8192 buckets of 1.9 entries on average, 4.6 with 16384 of 0.93 ([measured] 10^9 x 10^9: join kernel
//     10.1 -> 9.3 ms for 1568 table entries fewer; 17920 entries in 8192 buckets remain as JK_CT_13).
//   * the probe side streams through a 3-tile register ring; a probe tuple leaves in registers its rowID and its
//     matches as {first table position of its bucket, bit mask of the matching entries} (the register file, 512 KiB
//     per CU, is the largest memory there is: it holds a whole 16 K-tuple probe task).  1024 threads x 128 VGPRs:
//     four wavefronts per SIMD hide the LDS round trips of the probe phase better than two with twice the slots
//     each ([measured] 1B x 1B: 10.2 against 12.8 ms narrow, 12.3 against 13.8 ms with 16-byte tuples).
//     Several matches per probe tuple (duplicates on the build side: every second partition of a PK/FK join builds on
//     the foreign-key side, JobScheduler.cpp:187) cost nothing extra.
//   * when every probe is done the keys are dead: the build rowIDs (re-fetched from the partition, see below) are
//     written over the table IN TABLE ORDER (each thread kept the 16-bit position of the entries it placed; 4-byte words
//     in the narrow format), and the pairs (rowR,rowS) are completed from LDS: the first match of all 16 slots in one
//     batch of LDS reads.  One global atomicAdd per task reserves the output (issued before the rowIDs go into the table);
//     matches are compacted per wavefront slot by ballot + mbcnt; consecutive lanes store consecutive 16 B pairs; further
//     matches of a slot (duplicates on the build side) follow in a second pass.
//   * buckets of more than 16 entries (a join value repeated many times on the build side, e.g. skew) do not fit the
//     mask: such (wavefront, tile)s go through a generic loop that re-reads the slot's probe tuple, scans long buckets
//     with all 64 lanes, reserves its own output range and fetches build rowIDs from the partition in L2/HBM by the
//     16-bit arrival index every entry carries -- correct for any input, off the fast path's registers.
// Bucket counts are 16-bit halves of 32-bit LDS words (ds_add_rtn on the word; a half cannot carry: <= 17920 per
// workgroup), so 16384 buckets cost 32 KiB.  They lie at LDS offset 0 (their byte offset is their address), the table
// behind them, the scan scratch last (a compare round may read 15 entries past a bucket's end: never past the allocation).
// ------------------------------------------------------------------------------------------------
constexpr int CT_THREADS = 1024, CT_CHUNK = 16352, CT_BUCKET_BITS = 14, CT_EPT = 16, CT_PT = 4, CT_DEPTH = 3;
// JK_CT_13 (the full-size geometry of rounds 2 and 3 until the bucket count was doubled): 17920 entries, 8192 buckets.  For
// partitions of 15.3 - 16.8 K build tuples (1.005 - 1.1 * 10^9 tuples under 16 bits), which the 16352-entry table would
// build in two chunks; the 20-slot kernel (probe side beyond 16 K) keeps this table too.
constexpr int CT13_CHUNK = 17920, CT13_BUCKET_BITS = 13;
// the same kernel at half size, for partitions of up to 8960 build tuples (3 ... 5.5 * 10^8 tuples under a 16-bit plan):
// 512 threads, 80 KiB LDS -> TWO workgroups per CU, which overlap each other's memory and LDS phases; the per-thread
// register picture (18 build slots, 16 probe slots, 128 VGPRs) is unchanged.  The kernel's cost per task does not shrink
// with the partition (every slot row is walked), so the full-size geometry is 2-3x too expensive there (measured at
// 3 * 10^8: 8.9 ms against 5.1 ms for the chunked 16-byte-entry kernel).
constexpr int CTH_THREADS = 512, CTH_CHUNK = 8160, CTH_BUCKET_BITS = 13;      // (8960 entries in 4096 buckets until round 3: the probe tasks
// of 8192 tuples bound the partition size anyway, and twice the buckets are worth more than the last 800 entries)
constexpr int CTHW_CHUNK = 8960, CTHW_BUCKET_BITS = 12;                           // the 20-slot form keeps the larger table
// ... and with 20 probe slots per thread instead of 16 (narrow format only; 12 spilled VGPRs): partitions whose probe side is
// just beyond one 16-slot task (2.2 * 10^9 tuples under 17 or 18 bits: 16.8 K / 8.4 K per partition) would otherwise be cut into
// two tasks that both build the whole table
constexpr int CT_EPT_WIDE = 20;
// ... and a middle geometry: 12288 entries, 12 build and 12 probe slots per thread (1024 threads, one workgroup per CU).  The
// kernel's cost per task follows its slot rows, not the partition: partitions of 8.4 - 11.5 K tuples (5.5 - 7.5 * 10^8 tuples under
// 16 bits, 1.1 - 1.5 * 10^9 under 17) paid for 18 + 16 rows in the full-size geometry ([measured] join kernel 8.5 -> 6.6 ms at
// 6 * 10^8, 17.6 -> 14.1 at 1.5 * 10^9; 16 + 16 rows for the 15.3 K-tuple partitions of 10^9 tuples: 10.09 -> 10.02, not kept).
constexpr int CTM_CHUNK = 12288, CTM_EPT = 12, CTM_BUCKET_BITS = 14, CTHM_BUCKET_BITS = 13;
constexpr int CTHM_CHUNK = 6144;                    // ... and at half size (512 threads, two workgroups per CU): 4.2 - 5.8 K-tuple partitions
constexpr u32 CT_NONE = 0xFFFFu;
constexpr int CT_MIN_RADIX_BITS = 16;       // keys must fit 48 bits
// JK_CT_G13: the 6144-entry geometry with row guards and 13-bit arrival indices: keys of up to 51 bits, i.e. plans of 13-15
// radix bits (1.6 * 10^7 ... 1.3 * 10^8 tuples per side), whose 2-4 K-tuple partitions the one-table kernel served until round 4
constexpr int CT13_KB = 13, CT13_MIN_RADIX_BITS = 64 - (64 - CT13_KB);
// JK_CT_Q12: a 4096-entry table with 12-bit arrival indices (keys of up to 52 bits) in 4096 buckets, 8 + 8 slot rows per thread, row
// guards, 41 KiB of LDS: plans of exactly 12 bits (8.4 * 10^6 ... 1.6 * 10^7 tuples per side, partitions of 2-3.8 K tuples)
constexpr int CTQ_CHUNK = 4096, CTQ_BUCKET_BITS = 12, CTQ_EPT = 8, CTQ_KB = 12;
constexpr u32 CT_MASK_BITS = 16;            // a probe records its matches as a bit mask over a bucket of at most this many entries

// 32-bit Fibonacci hash of the folded key: one quarter-rate multiply instead of the four of a 64-bit product (the probe
// phase is latency-bound)
template <int BBITS>
__device__ __forceinline__ u32 ct_bucket(u64 key)
{
    return (((u32)key ^ (u32)(key >> 32)) * 0x9E3779B1u) >> (32 - BBITS);
}

// Pair stores of the fast path through a per-wavefront BUFFER DESCRIPTOR (base = the wavefront's first output slot, records =
// what the output buffer has left from there: a pair past the capacity is dropped by the hardware's range check): a store
// is descriptor + lane offset (mbcnt * 16) + scalar running offset, against a 64-bit address computed per lane and a
// capacity compare + branch per store.  Round 3 tried this and got wrong pairs in 2 of 15 instantiations; the cause
// (tools/srd_store_hazard.hip, profiles/r04_srd_store_hazard_probe.txt, [measured] on gfx950): a VALU write to the DATA registers
// of a buffer_store_dwordx4 right behind the store races with the store's read of them --
//     soffset an SGPR:            0 wait states 1.4 % of the pairs wrong, 1 wait state clean (LLVM inserts none: its hazard table
//                                 exempts stores whose soffset is a register, GCNHazardRecognizer::createsVALUHazard)
//     soffset the literal 0:      0 wait states 19 % wrong, 1 wait state STILL 1.5 % wrong (LLVM inserts exactly one)
// -- not a range problem (bases 5 GiB into an allocation, records clamped to 2^32 - 1: clean).  So the store is issued from
// inline assembly together with its own `s_nop 1` (two wait states), SGPR soffset.
#ifndef RHJ_CT_SRD_STORES
#define RHJ_CT_SRD_STORES 1
#endif
__device__ __forceinline__ void srd_store_pair(u64 r, u64 s_, __amdgpu_buffer_rsrc_t rsrc, u32 voff, u32 soff)
{
    const v4u32 d = {(u32)r, (u32)(r >> 32), (u32)s_, (u32)(s_ >> 32)};
    asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 1" : : "v"(d), "v"(voff), "s"(rsrc), "s"(soff));
}

// STAMPS: tuning aid (RHJ_CT_STAMPS=1): thread 0 of the first workgroups records s_memrealtime (100 MHz) at phase
// boundaries into `stamps` (CT_NSTAMP words per workgroup); never set in production launches.
constexpr int CT_NSTAMP = 16;

// skip (optional): a device word that is non-zero when this join is going to be repeated in another format (a rowID did
// not fit the narrow format): nothing to do then.
// KB: bits of the arrival index in a table entry {key | index}: 16 by default (keys of 48 bits: plans that remove >= 16 payload bits);
// 13 for the 6144-entry geometry under plans of 13-15 bits (keys of up to 51 bits: the partitions of 1.6 * 10^7 ... 1.3 * 10^8 tuples)
template <int THREADS, int CHUNK, int BBITS, int EPT, bool STAMPS, bool NARROW, bool GUARD = false, int KB = 16>
__global__ void __launch_bounds__(THREADS, THREADS * (CHUNK <= 8960 ? 2 : 1) / 256)   // wavefronts per SIMD: 2 (256 registers per lane) or 4 (128)
k_join_ct(const RelView<NARROW> R, const RelView<NARROW> S, const JoinTask *__restrict__ tasks,
          const u32 *__restrict__ ntasks, int radix_bits, Pair *__restrict__ out, u64 out_capacity,
          u64 *__restrict__ out_count, u64 *__restrict__ stamps, u32 nstamp_wgs, const u32 *__restrict__ skip)
{
    if (skip != nullptr && *skip != 0) return;
    int stamp_i = 0;
    auto stamp = [&]() {
        if (STAMPS && threadIdx.x == 0 && blockIdx.x < nstamp_wgs && stamp_i < CT_NSTAMP)
            stamps[(u64)blockIdx.x * CT_NSTAMP + stamp_i++] = __builtin_amdgcn_s_memrealtime();
    };
    stamp();                                                                 // 0: start
    constexpr int NB = 1 << BBITS;
    constexpr int NW = THREADS / 64;
    constexpr u32 KM = (1u << KB) - 1u;                             // index part of an entry
    static_assert(CHUNK <= (1 << KB), "the arrival index of every table entry must fit KB bits");
    constexpr int BPT = (CHUNK + THREADS - 1) / THREADS;            // build tuples per thread
    constexpr int BB = BPT;                                         // build loads in flight per lane: the whole build side
    constexpr int WPT = NB / 2 / THREADS;                           // packed counter words per thread in the scan
    constexpr int PT = CT_PT, NT = EPT / PT, DEPTH = NARROW ? CT_DEPTH : 1;      // probe tile: PT slots; ring of DEPTH tiles ([measured] narrow: 2, 3, 4 tiles alike, 10.1-10.2 ms; 16-byte tuples: 1 tile 11.8 ms, 2 or 3 tiles 12.4 -- they spill)
    static_assert(EPT % PT == 0 && EPT <= 32 && NB % (2 * THREADS) == 0 && CHUNK < (int)CT_NONE && BPT % BB == 0 && NW <= 64,
                  "geometry");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // the bucket offsets first: their LDS addresses are then the byte offset itself (behind the table they cost two more
    // address instructions per access: the base does not fit the 16-bit immediate of a ds instruction)
    u32 *off32 = reinterpret_cast<u32 *>(smem);                              // NB/2 + 2 words of two 16-bit halves
    const unsigned short *off16 = reinterpret_cast<const unsigned short *>(off32);   // off16[h], h in [0, NB]
    u64 *ent = reinterpret_cast<u64 *>(off32 + NB / 2 + 2);                  // CHUNK entries {key48 | idx16}, bucket order
    // ... later the CHUNK build rowIDs in TABLE order (4 bytes each in the narrow format)
    // (a compare round reads up to CT_MASK_BITS - 1 entries past a bucket's end: behind the table lie >= 128 bytes of scratch)
    u32 *wsum = reinterpret_cast<u32 *>(ent + CHUNK);                        // NW scan scratch
    u32 *wtot = wsum + NW;                                                   // NW match totals
    u64 *gres = reinterpret_cast<u64 *>(wtot + NW);                          // 1
    u64 *dummy = gres + 1;                                                   // target of the LDS writes of out-of-range slots:
    // range checks select an address instead of branching, so that the reads / atomics of a whole batch are in
    // flight together (a branch per slot costs one exposed LDS round trip each: measured 35 of them per build phase)

    const u32 nt = *ntasks;
    if (blockIdx.x >= nt) return;
    const JoinTask task = tasks[blockIdx.x];
    const bool build_is_S = task.build_is_S != 0;
    typedef typename RelView<NARROW>::Rid Rid;
    typedef typename RelView<NARROW>::Both Both;
    Rid *rid = reinterpret_cast<Rid *>(ent);
    Rid *rdummy = reinterpret_cast<Rid *>(dummy);
    const RelView<NARROW> B = (build_is_S ? S : R).at(task.bbeg);
    const RelView<NARROW> P = (build_is_S ? R : S).at(task.pbeg);
    const u32 nb = task.blen, np = task.plen;                                // np <= THREADS * EPT (host: probe_split)
    const typename RelView<NARROW>::Buf PB = P.buf(0, np);                   // (np * 16 < 2^32)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int rb = radix_bits;
    const unsigned long long lt = (1ull << lane) - 1ull;
    // slot k of this thread is probe tuple k * THREADS + tid; slots k < nv are valid (nv: computed where the probe phase starts)
    const int tid0 = tid;
    // Slot row k holds tuples [k * THREADS, (k + 1) * THREADS) of the chunk / task: whether a row is in use is the same for
    // every thread (a scalar compare and branch).  GUARD: rows beyond the partition are skipped, so a partition that fills
    // half the rows pays for half of them and one geometry serves a range of partition sizes at a cost that follows the
    // partition ([measured] 6144-entry geometry, 2 * 10^8 tuples, 3 K-tuple partitions: join kernel 2.04 ms against 2.83
    // unguarded and 2.58 for k_join_bkt).  The branches cost full partitions 10 % (they end the batches of loads and LDS
    // operations: 10^9 tuples 9.1 -> 10.2 ms), so kernels for full tables are instantiated without them.
#define USED(k, n) (!GUARD || (u32)(k) * (u32)THREADS < (n))
    for (u32 cb = 0; cb < nb; cb += CHUNK) {
        const u32 nc = (nb - cb < (u32)CHUNK) ? nb - cb : (u32)CHUNK;
        const typename RelView<NARROW>::Buf BB_ = B.buf(cb, nc);              // this chunk of the build side: lanes past nc read 0
        // Addresses and range predicates of the 18 build and 16 probe slots depend only on the thread index: left
        // alone they are hoisted out of this loop into > 100 live registers.  An opaque copy of the thread index per
        // iteration keeps those one-instruction recomputations next to their uses.
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        const int nvb = nc > (u32)tid ? (int)((nc - (u32)tid + THREADS - 1) / THREADS) : 0;   // valid build slots
        // ---- build: count per bucket (rank = value before the add), keep {key | rank} and the rowID in registers ----
#pragma unroll
        for (int j = 0; j < WPT; j++) off32[tid * WPT + j] = 0;
        if (tid == 0) off32[NB / 2] = 0;
        __syncthreads();
        u64 kr[BPT];
#pragma unroll
        for (int k0 = 0; k0 < BPT; k0 += BB) {
            u64 bt[BB];
#pragma unroll
            for (int k = k0; k < k0 + BB; k++) {
                bt[k - k0] = 0;
                if (USED(k, nc)) bt[k - k0] = BB_.payload((u32)k * THREADS, (u32)tid);
            }
#pragma unroll
            for (int k = k0; k < k0 + BB; k++) {
                kr[k] = 0;
                if (!USED(k, nc)) continue;
                const u64 key = bt[k - k0] >> rb;
                const u32 h = k < nvb ? ct_bucket<BBITS>(key) : (u32)NB + 2u, sh = (h & 1u) * 16u;   // NB + 2: a padding word
                kr[k] = (key << KB) | ((atomicAdd(&off32[h >> 1], 1u << sh) >> sh) & 0xFFFFu);     // (the rank inside a bucket: < CHUNK <= 2^KB)
            }
        }
        stamp();                                                             // 1: build side loaded and counted
        __syncthreads();
        {   // in-place exclusive scan of the NB packed bucket counts
            u32 wd[WPT], loc = 0;
#pragma unroll
            for (int j = 0; j < WPT; j++) { wd[j] = off32[tid * WPT + j]; loc += (wd[j] & 0xFFFFu) + (wd[j] >> 16); }
            u32 tot;
            u32 ex = block_excl_scan<THREADS, false>(loc, wsum, tot, tid);
#pragma unroll
            for (int j = 0; j < WPT; j++) {
                const u32 c0 = wd[j] & 0xFFFFu, c1 = wd[j] >> 16;
                off32[tid * WPT + j] = ex | ((ex + c0) << 16);
                ex += c0 + c1;
            }
            if (tid == THREADS - 1) off32[NB / 2] = ex;                      // off16[NB] = nc
        }
        __syncthreads();
        u32 ppos[(BPT + 1) / 2];                                             // table position of each build tuple, two per register
        {
            int tp = tid0;                                                   // (see above: no shared index temporaries)
            asm volatile("" : "+v"(tp));
            const int nvp = nc > (u32)tp ? (int)((nc - (u32)tp + THREADS - 1) / THREADS) : 0;
#pragma unroll
            for (int k = 0; k < BPT; k++) {
                if (!USED(k, nc)) { if (!(k & 1)) ppos[k >> 1] = 0; continue; }
                const u64 key = kr[k] >> KB;
                const u32 pos = off16[ct_bucket<BBITS>(key)] + ((u32)kr[k] & KM);
                *(k < nvp ? &ent[pos] : dummy) = (key << KB) | (u64)((u32)k * THREADS + tp);
                if (k & 1) ppos[k >> 1] |= pos << 16; else ppos[k >> 1] = pos;
            }
        }
        __syncthreads();
        stamp();                                                             // 2: table ready

        // ---- probe: the task's probe tuples stream through a ring of DEPTH register tiles --------------------------
        Rid prid[EPT];                                                       // probe rowIDs
        u32 mi[EPT];                                                         // matches of a slot: first table position of its bucket |
                                                                             // (bit b: entry lo + b matches) << 16; 0 = none
        u32 deferred = 0;                                                    // wave-uniform: slots left to the generic loop
        u32 ctot = 0;                                                        // matches of this lane
        Both ring[DEPTH][PT];
        asm volatile("" : "+v"(tid));
        // (valid probe slots of this thread: recomputed here -- carried from the kernel's start it sat in a spill slot, and
        // 8 bytes of scratch per thread are 0.5 GB of HBM writes per launch)
        const int nv = np > (u32)tid ? (int)((np - (u32)tid + THREADS - 1) / THREADS) : 0;
#pragma unroll
        for (int t = 0; t < DEPTH && t < NT; t++) {
            if (!USED(t * PT, np)) continue;
#pragma unroll
            for (int s = 0; s < PT; s++) ring[t][s] = PB.both((u32)(t * PT + s) * THREADS, (u32)tid);
        }
#pragma unroll
        for (int t = 0; t < NT; t++) {
            u32 khi[PT], klo[PT];                                            // key << 16, to compare with an entry's upper 48 bits
            u32 lo[PT], len[PT], m[PT], maxlen = 0;
            if (!USED(t * PT, np)) {                                         // the task ends before this tile (same for every thread)
#pragma unroll
                for (int s = 0; s < PT; s++) { prid[t * PT + s] = 0; mi[t * PT + s] = 0; }
                continue;
            }
#pragma unroll
            for (int s = 0; s < PT; s++) {
                const int k = t * PT + s;
                prid[k] = ring[t % DEPTH][s].key;
                const u64 key = ring[t % DEPTH][s].payload >> rb;
                const u32 h = ct_bucket<BBITS>(key);
                khi[s] = (u32)(key >> (32 - KB)); klo[s] = (u32)key << KB;
                lo[s] = off16[h]; m[s] = 0;
                len[s] = k < nv ? off16[h + 1] - lo[s] : 0u;
                maxlen = len[s] > maxlen ? len[s] : maxlen;
            }
            if (t + DEPTH < NT && USED((t + DEPTH) * PT, np)) {               // the slot is free: next tile on its way
#pragma unroll
                for (int s = 0; s < PT; s++)
                    ring[t % DEPTH][s] = PB.both((u32)((t + DEPTH) * PT + s) * THREADS, (u32)tid);
            }
            const bool longb = __ballot(maxlen > CT_MASK_BITS) != 0;         // a long bucket somewhere: the generic loop
            if (!longb) {
                for (u32 j = 0; __ballot(j < maxlen) != 0; j++) {            // PT independent LDS reads per round
                    // the PT reads first, then the PT compares: written as one loop the compiler issues read, wait, compare per
                    // slot -- four LDS round trips per round behind one another (seen in the ISA: one destination register pair)
                    u64 e[PT];
#pragma unroll
                    for (int s = 0; s < PT; s++) e[s] = ent[lo[s] + j];      // (past the bucket's end: some other entry, ignored)
#pragma unroll
                    for (int s = 0; s < PT; s++) {
                        // (no `j < len` here: an entry of another bucket has another key; what lies behind the table's end is
                        // masked off below)
                        const u32 xl = (u32)e[s] ^ klo[s];
                        if ((u32)(e[s] >> 32) == khi[s] && xl <= KM) m[s] |= 0x10000u << j;
                    }
                }
#pragma unroll
                for (int s = 0; s < PT; s++) m[s] &= ((1u << len[s]) - 1u) << 16;
            }
#pragma unroll
            for (int s = 0; s < PT; s++) {
                const int k = t * PT + s;
                if (longb) { deferred |= 1u << k; m[s] = 0; }
                ctot += (u32)__popc(m[s] >> 16);
                mi[k] = m[s] ? (m[s] | lo[s]) : 0u;
            }
        }
        u32 wave_total = ctot;                                               // matches found by this wavefront on the fast path
#pragma unroll
        for (int o2 = 32; o2 > 0; o2 >>= 1) wave_total += __shfl_xor(wave_total, o2, 64);

        stamp();                                                             // 3: this wavefront's probes done
        // ---- generic loop: (wavefront, slot)s with duplicates on the build side or long buckets.  Own output range,
        // build rowIDs from the partition in global memory; the table is still valid here. ----
        while (deferred) {
            const int k = __ffs((int)deferred) - 1;
            deferred &= deferred - 1;
            const Both pt = P.both(k < nv ? (u32)k * THREADS + tid : 0u);
            const u64 key = pt.payload >> rb;
            u32 lo = 0, hi = 0;
            if (k < nv) { const u32 h = ct_bucket<BBITS>(key); lo = off16[h]; hi = off16[h + 1]; }
            unsigned long long heavy = __ballot(hi - lo > BJ_HEAVY);
            const bool coop = heavy != 0 && __popcll(heavy) <= BJ_HEAVY_LANES;
            const bool serial = !coop || hi - lo <= BJ_HEAVY;
            u32 c = 0;
            if (serial) for (u32 j = lo; j < hi; j++) c += ((ent[j] >> KB) == key) ? 1u : 0u;
            if (coop) {
                unsigned long long hv = heavy;
                while (hv) {
                    const int leader = __ffsll((long long)hv) - 1;
                    hv &= hv - 1;
                    const u64 lkey = bj_readlane64(key, leader);
                    const u32 l = __builtin_amdgcn_readlane(lo, leader), hh = __builtin_amdgcn_readlane(hi, leader);
                    u32 tot = 0;
                    for (u32 j = l; j < hh; j += 64) {
                        const bool mt = (j + lane < hh) && (ent[j + lane] >> KB) == lkey;
                        tot += (u32)__popcll(__ballot(mt));
                    }
                    if (lane == leader) c = tot;
                }
            }
            const u32 ic = wave_incl_scan(c, lane);
            const u32 tot = __shfl(ic, 63, 64);
            if (tot == 0) continue;
            u64 base = 0;
            if (lane == 0) base = atomicAdd(out_count, (u64)tot);
            base = bj_readlane64(base, 0);
            if (out == nullptr) continue;
            u64 o = base + ic - c;
            if (c && serial) {
                for (u32 j = lo; j < hi; j++) {
                    const u64 e = ent[j];
                    if ((e >> KB) == key) {
                        if (o < out_capacity) {
                            const u64 br = B.rowid(cb + ((u32)e & KM));
                            Pair pr;
                            if (build_is_S) { pr.r = pt.key; pr.s = br; } else { pr.r = br; pr.s = pt.key; }
                            out[o] = pr;
                        }
                        o++;
                    }
                }
            }
            if (coop) {
                unsigned long long hv = heavy;
                while (hv) {
                    const int leader = __ffsll((long long)hv) - 1;
                    hv &= hv - 1;
                    const u64 lkey = bj_readlane64(key, leader), lprid = bj_readlane64((u64)pt.key, leader);
                    u64 ob = bj_readlane64(o, leader);
                    const u32 l = __builtin_amdgcn_readlane(lo, leader), hh = __builtin_amdgcn_readlane(hi, leader);
                    for (u32 j = l; j < hh; j += 64) {
                        u64 e = 0;
                        bool mt = false;
                        if (j + lane < hh) { e = ent[j + lane]; mt = (e >> KB) == lkey; }
                        const unsigned long long bal = __ballot(mt);
                        const u64 dst = ob + (u64)__popcll(bal & lt);
                        if (mt && dst < out_capacity) {
                            const u64 br = B.rowid(cb + ((u32)e & KM));
                            Pair pr;
                            if (build_is_S) { pr.r = lprid; pr.s = br; } else { pr.r = br; pr.s = lprid; }
                            out[dst] = pr;
                        }
                        ob += (u64)__popcll(bal);
                    }
                }
            }
        }

        // ---- output of the FK case: table -> build rowIDs, one reservation per task, pairs completed from LDS ------
        stamp();                                                             // 4: generic loop done
        // The build rowIDs are needed once the table is dead.  Holding them in registers from the build phase on would
        // take 70 more VGPRs through the probe phase (tried: the allocator spills them at their definition and the
        // build loads serialise behind the scratch stores); they are fetched again here instead, 8 of every 16 bytes of
        // a partition this CU streamed a few microseconds ago, while the barrier and the reservation go on.
        Rid brid[BPT];
        {
            int tq = tid0;
            asm volatile("" : "+v"(tq));
#pragma unroll
            for (int k = 0; k < BPT; k++) {
                brid[k] = 0;
                if (USED(k, nc)) brid[k] = BB_.rowid((u32)k * THREADS, (u32)tq);
            }
        }
        stamp();                                                             // 5: rowID loads issued
        int tq2 = tid0;                                                      // (opaque: &wtot[w], &wtot[lane] are not worth a spill slot)
        asm volatile("" : "+v"(tq2));
        if ((tq2 & 63) == 0) wtot[tq2 >> 6] = wave_total;
        __syncthreads();                                                     // every wavefront is done with the table
        stamp();                                                             // 6: barrier passed
        // the one reservation of the task is on its way to L2 while the rowIDs go into the table
        const u32 mine = (tq2 & 63) < NW ? wtot[tq2 & 63] : 0u;
        const u32 inc = wave_incl_scan(mine, lane);
        const u32 chunk_total = __shfl(inc, NW - 1, 64);
        const u32 wbase = __shfl(inc - mine, w, 64);
        u64 reserved = 0;
        if (tid == 0 && chunk_total) reserved = atomicAdd(out_count, (u64)chunk_total);
        {
            int td = tid0;
            asm volatile("" : "+v"(td));
            const int nvd = nc > (u32)td ? (int)((nc - (u32)td + THREADS - 1) / THREADS) : 0;
#pragma unroll
            for (int k = 0; k < BPT; k++)
                if (USED(k, nc))
                    *(k < nvd ? &rid[(k & 1) ? (ppos[k >> 1] >> 16) : (ppos[k >> 1] & 0xFFFFu)] : rdummy) = brid[k];   // table order
        }
        stamp();                                                             // 7: rowIDs in LDS
        if (tid == 0 && chunk_total) *gres = reserved;
        __syncthreads();
        stamp();                                                             // 8: output reserved
        if (out != nullptr && wave_total) {
            u64 o = *gres + wbase;                                           // next output slot of this wavefront
            // narrow partitions only (SRD stores): with 64-bit rowIDs the four-register data tuples of the stores cost the 16-byte-
            // tuple instantiations 32 more spilled VGPRs
            constexpr bool SRD = NARROW && RHJ_CT_SRD_STORES != 0;
            // (wave-uniform: the descriptor and the running offset live in SGPRs)
            // (the builtin returns int: without the casts a low word >= 2^31 -- a pair index beyond 2^31, 2.15 * 10^9 pairs -- sign-extends
            // into the high word; that, not the hazard, was what still failed the 2.2 * 10^9 checksum)
            const u64 o_first = ((u64)(u32)__builtin_amdgcn_readfirstlane((u32)(o >> 32)) << 32) | (u64)(u32)__builtin_amdgcn_readfirstlane((u32)o);
            const u64 o_left = out_capacity > o_first ? out_capacity - o_first : 0;
            const __amdgpu_buffer_rsrc_t orsrc = make_srd(out + o_first, o_left > 0x0FFFFFFFull ? 0xFFFFFFF0u : (u32)o_left * 16u);
            u32 orun = 0;                                                    // pairs this wavefront has stored so far
            // The first match of every slot: EPT independent LDS reads in flight together (one read, wait, store per slot
            // left 16 LDS round trips per thread exposed behind one another).
            Rid br0[EPT];
#pragma unroll
            for (int k = 0; k < EPT; k++) {
                const u32 mk = mi[k] >> 16;
                br0[k] = 0;
                if (USED(k, np)) br0[k] = rid[(mi[k] & 0xFFFFu) + (mk ? (u32)__ffs((int)mk) - 1u : 0u)];
            }
            // Pass 1 stores every slot's first match: ballot + mbcnt compaction, consecutive lanes -> consecutive pairs.  In
            // the FK case (unique build keys) that is everything.
            u32 more = 0;
#pragma unroll
            for (int k = 0; k < EPT; k++) {
                if (!USED(k, np)) continue;
                u32 mask = mi[k] >> 16;
                const unsigned long long bal = __ballot(mask != 0);
                if (mask) {
                    mask &= mask - 1;
                    if constexpr (SRD) {
                    const u32 lanepos = __builtin_amdgcn_mbcnt_hi((u32)(bal >> 32), __builtin_amdgcn_mbcnt_lo((u32)bal, 0u));
                    if (build_is_S) srd_store_pair((u64)prid[k], (u64)br0[k], orsrc, lanepos * 16u, orun * 16u);   // orderFlag, Result.cpp:64-68
                    else            srd_store_pair((u64)br0[k], (u64)prid[k], orsrc, lanepos * 16u, orun * 16u);
                    } else {
                    const u64 dst = o + (u64)__popcll(bal & lt);
                    if (dst < out_capacity) {
                        Pair pr;
                        if (build_is_S) { pr.r = prid[k]; pr.s = br0[k]; }     // orderFlag, Result.cpp:64-68
                        else            { pr.r = br0[k]; pr.s = prid[k]; }
                        // plain (not nontemporal) stores: a wavefront's pair run continues in the next slot's store a
                        // microsecond later; with the nt hint the shared 128 B line at the seam went to HBM twice
                        // ([measured] WRITE_SIZE 18.32 GB per 10^9 pairs against 16.03, same kernel time within 1 %)
                        out[dst] = pr;
                    }
                    }
                    mi[k] = (mi[k] & 0xFFFFu) | (mask << 16);
                    more |= mask;
                }
                if constexpr (SRD) orun += (u32)__popcll(bal); else o += (u64)__popcll(bal);
            }
            // Pass 2: further matches of a slot (duplicates on the build side), behind the first matches of the wavefront --
            // the order of the pairs is free.  Round r stores the (r+2)-th match of every lane that has one.
            if (__ballot(more != 0) != 0) {
#pragma unroll
                for (int k = 0; k < EPT; k++) {
                    if (!USED(k, np)) continue;
                    const u32 lo = mi[k] & 0xFFFFu;
                    u32 mask = mi[k] >> 16;
                    for (unsigned long long bal = __ballot(mask != 0); bal != 0; bal = __ballot(mask != 0)) {
                        if (mask) {
                            const u32 bpos = (u32)__ffs((int)mask) - 1;
                            mask &= mask - 1;
                            if constexpr (SRD) {
                            const u32 lanepos = __builtin_amdgcn_mbcnt_hi((u32)(bal >> 32), __builtin_amdgcn_mbcnt_lo((u32)bal, 0u));
                            const u64 br = rid[lo + bpos];
                            if (build_is_S) srd_store_pair((u64)prid[k], br, orsrc, lanepos * 16u, orun * 16u);
                            else            srd_store_pair(br, (u64)prid[k], orsrc, lanepos * 16u, orun * 16u);
                            } else {
                            const u64 dst = o + (u64)__popcll(bal & lt);
                            if (dst < out_capacity) {
                                const u64 br = rid[lo + bpos];
                                Pair pr;
                                if (build_is_S) { pr.r = prid[k]; pr.s = br; }
                                else            { pr.r = br; pr.s = prid[k]; }
                                out[dst] = pr;
                            }
                            }
                        }
                        if constexpr (SRD) orun += (u32)__popcll(bal); else o += (u64)__popcll(bal);
                    }
                }
            }
        }
        stamp();                                                             // 9: this wavefront's pairs stored (issued)
        if (cb + CHUNK < nb) __syncthreads();                                // the next chunk rebuilds the table over rid
    }
#undef USED
}

// ------------------------------------------------------------------------------------------------
// utilities
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ u64 block_sum_u64(u64 v, u64 *wtot)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0) wtot[threadIdx.x >> 6] = v;
    __syncthreads();
    u64 t = 0;
    if (threadIdx.x == 0) for (int i = 0; i < (int)(blockDim.x >> 6); i++) t += wtot[i];
    return t;
}

__global__ void __launch_bounds__(256) k_checksum(const Pair *__restrict__ p, u64 n, u64 *__restrict__ sum)
{
    __shared__ u64 wtot[4];
    u64 acc = 0;
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256) {
        const Pair x = p[i];
        acc += mix64(x.r * 0x100000001B3ULL ^ mix64(x.s));
    }
    const u64 t = block_sum_u64(acc, wtot);
    if (threadIdx.x == 0) atomicAdd(sum, t);
}

__global__ void __launch_bounds__(256) k_expected_pkfk(const Tup *__restrict__ S, u64 n, u64 *__restrict__ sum)
{
    __shared__ u64 wtot[4];
    u64 acc = 0;
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256) {
        const Tup x = S[i];
        const u64 k = unmix64(x.payload);              // payload = mix(k), R row = k - 1
        acc += mix64((k - 1) * 0x100000001B3ULL ^ mix64(x.key));
    }
    const u64 t = block_sum_u64(acc, wtot);
    if (threadIdx.x == 0) atomicAdd(sum, t);
}

// payload = (k << shift) + add for a generated payload mix64(k): the same PK/FK pair set over join values that are dense
// (shift 0), multiples of 2^shift, or k * 2^shift + const -- the key shapes that defeat radix digits taken from raw bits
__global__ void __launch_bounds__(256) k_remap_keys(Tup *__restrict__ t, u64 n, int shift, u64 add)
{
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256)
        t[i].payload = (unmix64(t[i].payload) << shift) + add;
}

__global__ void __launch_bounds__(256)
k_generate(int kind, Tup *__restrict__ out, u64 n, u64 row0, u64 D, u64 seed, double theta)
{
    const double e = 1.0 - theta;
    const double span = (kind == 2) ? (pow((double)D + 1.0, e) - 1.0) : 0.0;
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256) {
        const u64 row = row0 + i;
        Tup t;
        t.key = row;
        switch (kind) {
        case 0: t.payload = mix64(1 + row % D); break;
        case 1: t.payload = mix64(1 + mix64(row ^ seed) % D); break;
        case 2: {
            const double uu = (double)(mix64(row ^ seed) >> 11) * (1.0 / 9007199254740992.0);   // [0,1)
            double x = pow(1.0 + uu * span, 1.0 / e);
            u64 r = (u64)x;
            if (r < 1) r = 1;
            if (r > D) r = D;
            t.payload = mix64(r);
            break;
        }
        case 3: t.payload = mix64(D + 1 + row); break;
        default: t.payload = D; break;
        }
        out[i] = t;
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
size_t scan_tmp_bytes(int bits) { return (size_t)SCAN_SLICES * ((size_t)8 << bits); }

// sizes within 1/2^this of each other are a tie when a partition's build side is chosen (build_on_S)
int build_tie_shift()
{
    static const int v = getenv("RHJ_BUILD_TIE") ? atoi(getenv("RHJ_BUILD_TIE")) : 4;      // tuning aid: 63 = the reference's rule exactly
    return v < 1 ? 1 : v > 63 ? 63 : v;
}

size_t part_lds_bytes(int bits)
{
    const size_t nbins = (size_t)1 << bits;
    return (size_t)PART_TILE * 16 + nbins * (8 + 8 + 4 + 4);
}

static size_t wc_lds_bytes(int bits, int threads)
{
    const size_t nbins = (size_t)1 << bits;
    return (size_t)threads * WC_TPT * 16 + nbins * (128 + 8 + 8 + 8 + 4 + 4 + 4 + 4) + 16 + (size_t)(threads / 64) * 4;
}

static size_t wn_lds_bytes(int bits, int gr = WN_GR, int tpt = WN_TPT, int threads = WN_THREADS, bool peer = false)
{
    const size_t nbins = (size_t)1 << bits;
    return ((size_t)threads * tpt + nbins * gr) * 12 + nbins * (8 + 8 + 8 + 4 + 4 + 4 + 4) + 16 + (size_t)(threads / 64) * 4 +
           (peer ? nbins * 4 + 8 + (size_t)SEG_MAX * 16 : 0);
}

static int wc_threads_for(int bits)
{
    static const int force = getenv("RHJ_WC_THREADS") ? atoi(getenv("RHJ_WC_THREADS")) : 0;   // tuning aid
    if (force == 512 || force == 1024) return force;
    return bits <= 8 ? WC_THREADS_SMALL : WC_THREADS;
}

constexpr int BJ2_THREADS = 1024, BJ2_CHUNK = 8448, BJ2_BUCKET_BITS = 12, BJ2_EPT = 4;

// probe tuples per task / build tuples per table of each join kernel (host plan)
u32 join_probe_split(int kind)
{
    return kind == JK_CT || kind == JK_CT_13 ? (u32)(CT_THREADS * CT_EPT) : kind == JK_CT_HALF ? (u32)(CTH_THREADS * CT_EPT) :
           kind == JK_CT_WIDE ? (u32)(CT_THREADS * CT_EPT_WIDE) : kind == JK_CT_HALF_WIDE ? (u32)(CTH_THREADS * CT_EPT_WIDE) :
           kind == JK_CT_MID ? (u32)(CT_THREADS * CTM_EPT) : kind == JK_CT_HALF_MID || kind == JK_CT_HALF_MID_G || kind == JK_CT_G13 ? (u32)(CTH_THREADS * CTM_EPT) :
           kind == JK_CT_Q12 ? (u32)(CTH_THREADS * CTQ_EPT) : 0u;
}
u32 join_table_tuples(int kind)
{
    return kind == JK_CT ? (u32)CT_CHUNK : kind == JK_CT_13 || kind == JK_CT_WIDE ? (u32)CT13_CHUNK : kind == JK_CT_HALF ? (u32)CTH_CHUNK : kind == JK_CT_HALF_WIDE ? (u32)CTHW_CHUNK :
           kind == JK_CT_MID ? (u32)CTM_CHUNK : kind == JK_CT_HALF_MID || kind == JK_CT_HALF_MID_G || kind == JK_CT_G13 ? (u32)CTHM_CHUNK : kind == JK_CT_Q12 ? (u32)CTQ_CHUNK :
           kind == JK_BKT_BIG ? (u32)BJ2_CHUNK : (u32)BJ_CHUNK;
}
int join_ct_min_radix_bits(int kind) { return kind == JK_CT_G13 ? CT13_MIN_RADIX_BITS : kind == JK_CT_Q12 ? CTQ_KB : CT_MIN_RADIX_BITS; }

static size_t bj_lds_bytes(int threads, int chunk, int bbits)
{
    return (size_t)chunk * 16 + ((size_t)(1 << bbits) + 4) * 4 + 64 * 4 + (size_t)(threads / 64) * 4 + 16 + 2 * TAG_MAX * 8;
}

// Per device (a process may drive several GPUs through different contexts) and exactly once: contexts of
// different host threads launch concurrently (MainScheduler's 8 query threads), so the flag may only become
// visible after every attribute has been applied -- std::call_once blocks the other callers until then.
static int current_device_slot()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    return dev;
}

static size_t ct_lds_bytes(int threads = CT_THREADS, int chunk = CT_CHUNK, int bbits = CT_BUCKET_BITS)
{
    return (size_t)chunk * 8 + ((size_t)(1 << bbits) / 2 + 2 + 2 * (threads / 64)) * 4 + 24 + (threads < 1024 ? 64 : 0);   // (>= 128 B behind the table)
}

// hipFuncSetAttribute results are kept: a refused LDS size would otherwise surface later as an anonymous launch failure.
// rhj_api.hip reads the text through launch_attr_error() in check_launch().
static std::mutex g_attr_mu;
static std::string g_attr_error;

const char *launch_attr_error()
{
    std::lock_guard<std::mutex> lk(g_attr_mu);
    return g_attr_error.empty() ? nullptr : g_attr_error.c_str();
}

static void set_lds(const void *fn, size_t bytes, const char *name)
{
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) return;
    (void)hipGetLastError();
    std::lock_guard<std::mutex> lk(g_attr_mu);
    if (g_attr_error.empty())
        g_attr_error = std::string("hipFuncSetAttribute(") + name + ", dynamic LDS " + std::to_string(bytes) + " B): " + hipGetErrorString(e);
}
#define SET_LDS(fn, bytes) set_lds(reinterpret_cast<const void *>(&fn), (bytes), #fn)

static void allow_big_lds()
{
    static std::once_flag done[64];
    std::call_once(done[current_device_slot()], [] {
    SET_LDS(k_scatter_units_pipe, part_lds_bytes(PART_MAX_BITS));
    SET_LDS(k_scatter_wc<WC_THREADS>, wc_lds_bytes(WC_MAX_BITS, WC_THREADS));
    SET_LDS(k_scatter_wc<WC_THREADS_SMALL>, wc_lds_bytes(WC_MAX_BITS, WC_THREADS_SMALL));
    SET_LDS(k_scatter_wc2<WC_THREADS>, wc_lds_bytes(WC_MAX_BITS, WC_THREADS));
    SET_LDS(k_scatter_wc2<WC_THREADS_SMALL>, wc_lds_bytes(WC_MAX_BITS, WC_THREADS_SMALL));
    SET_LDS(k_scatter_fused2<WC_THREADS>, wc_lds_bytes(WC_MAX_BITS, WC_THREADS));
    SET_LDS(k_scatter_fused2<WC_THREADS_SMALL>, wc_lds_bytes(WC_MAX_BITS, WC_THREADS_SMALL));
    SET_LDS((k_join_bkt<BJ_THREADS, BJ_CHUNK, BJ_BUCKET_BITS, BJ_EPT, false>), bj_lds_bytes(BJ_THREADS, BJ_CHUNK, BJ_BUCKET_BITS));
    SET_LDS((k_join_bkt<BJ_THREADS, BJ_CHUNK, BJ_BUCKET_BITS, BJ_EPT, true>), bj_lds_bytes(BJ_THREADS, BJ_CHUNK, BJ_BUCKET_BITS));
    SET_LDS((k_join_bkt<BJ_THREADS, BJ_CHUNK, BJ_BUCKET_BITS, BJ_EPT, true, false, false, true>), bj_lds_bytes(BJ_THREADS, BJ_CHUNK, BJ_BUCKET_BITS));
    SET_LDS((k_join_bkt<BJ_THREADS, BJ_CHUNK, BJ_BUCKET_BITS, BJ_EPT, false, true>), bj_lds_bytes(BJ_THREADS, BJ_CHUNK, BJ_BUCKET_BITS));
    SET_LDS((k_join_bkt<BJ2_THREADS, BJ2_CHUNK, BJ2_BUCKET_BITS, BJ2_EPT, false>), bj_lds_bytes(BJ2_THREADS, BJ2_CHUNK, BJ2_BUCKET_BITS));
    SET_LDS((k_join_ct<CT_THREADS, CT_CHUNK, CT_BUCKET_BITS, CT_EPT, false, false>), ct_lds_bytes());
    SET_LDS((k_join_ct<CTH_THREADS, CTH_CHUNK, CTH_BUCKET_BITS, CT_EPT, false, false>), ct_lds_bytes(CTH_THREADS, CTH_CHUNK, CTH_BUCKET_BITS));
    SET_LDS((k_join_ct<CT_THREADS, CT_CHUNK, CT_BUCKET_BITS, CT_EPT, true, false>), ct_lds_bytes());
    SET_LDS((k_join_ct<CT_THREADS, CT_CHUNK, CT_BUCKET_BITS, CT_EPT, false, true>), ct_lds_bytes());
    SET_LDS((k_join_ct<CTH_THREADS, CTH_CHUNK, CTH_BUCKET_BITS, CT_EPT, false, true>), ct_lds_bytes(CTH_THREADS, CTH_CHUNK, CTH_BUCKET_BITS));
    SET_LDS((k_join_ct<CT_THREADS, CT13_CHUNK, CT13_BUCKET_BITS, CT_EPT_WIDE, false, true>), ct_lds_bytes(CT_THREADS, CT13_CHUNK, CT13_BUCKET_BITS));
    SET_LDS((k_join_ct<CT_THREADS, CT13_CHUNK, CT13_BUCKET_BITS, CT_EPT, false, true>), ct_lds_bytes(CT_THREADS, CT13_CHUNK, CT13_BUCKET_BITS));
    SET_LDS((k_join_ct<CT_THREADS, CT13_CHUNK, CT13_BUCKET_BITS, CT_EPT, false, false>), ct_lds_bytes(CT_THREADS, CT13_CHUNK, CT13_BUCKET_BITS));
    SET_LDS((k_join_ct<CT_THREADS, CTM_CHUNK, CTM_BUCKET_BITS, CTM_EPT, false, true>), ct_lds_bytes(CT_THREADS, CTM_CHUNK, CTM_BUCKET_BITS));
    SET_LDS((k_join_ct<CT_THREADS, CTM_CHUNK, CTM_BUCKET_BITS, CTM_EPT, false, false>), ct_lds_bytes(CT_THREADS, CTM_CHUNK, CTM_BUCKET_BITS));
    SET_LDS((k_join_ct<CTH_THREADS, CTHM_CHUNK, CTHM_BUCKET_BITS, CTM_EPT, false, true>), ct_lds_bytes(CTH_THREADS, CTHM_CHUNK, CTHM_BUCKET_BITS));
    SET_LDS((k_join_ct<CTH_THREADS, CTHM_CHUNK, CTHM_BUCKET_BITS, CTM_EPT, false, false>), ct_lds_bytes(CTH_THREADS, CTHM_CHUNK, CTHM_BUCKET_BITS));
    SET_LDS((k_join_ct<CTH_THREADS, CTHM_CHUNK, CTHM_BUCKET_BITS, CTM_EPT, false, true, true>), ct_lds_bytes(CTH_THREADS, CTHM_CHUNK, CTHM_BUCKET_BITS));
    SET_LDS((k_join_ct<CTH_THREADS, CTHM_CHUNK, CTHM_BUCKET_BITS, CTM_EPT, false, false, true>), ct_lds_bytes(CTH_THREADS, CTHM_CHUNK, CTHM_BUCKET_BITS));
    SET_LDS((k_join_ct<CTH_THREADS, CTHM_CHUNK, CTHM_BUCKET_BITS, CTM_EPT, false, true, true, CT13_KB>), ct_lds_bytes(CTH_THREADS, CTHM_CHUNK, CTHM_BUCKET_BITS));
    SET_LDS((k_join_ct<CTH_THREADS, CTQ_CHUNK, CTQ_BUCKET_BITS, CTQ_EPT, false, true, true, CTQ_KB>), ct_lds_bytes(CTH_THREADS, CTQ_CHUNK, CTQ_BUCKET_BITS));
    SET_LDS((k_join_ct<CTH_THREADS, CTQ_CHUNK, CTQ_BUCKET_BITS, CTQ_EPT, false, false, true, CTQ_KB>), ct_lds_bytes(CTH_THREADS, CTQ_CHUNK, CTQ_BUCKET_BITS));
    SET_LDS((k_join_ct<CTH_THREADS, CTHM_CHUNK, CTHM_BUCKET_BITS, CTM_EPT, false, false, true, CT13_KB>), ct_lds_bytes(CTH_THREADS, CTHM_CHUNK, CTHM_BUCKET_BITS));
    SET_LDS((k_join_ct<CTH_THREADS, CTHW_CHUNK, CTHW_BUCKET_BITS, CT_EPT_WIDE, false, true>), ct_lds_bytes(CTH_THREADS, CTHW_CHUNK, CTHW_BUCKET_BITS));
    SET_LDS((k_join_bkt<BJ_THREADS, BJ_CHUNK, BJ_BUCKET_BITS, BJ_EPT, false, true, true>), bj_lds_bytes(BJ_THREADS, BJ_CHUNK, BJ_BUCKET_BITS));
    SET_LDS(k_scatter_wc_n<WC_THREADS>, wc_lds_bytes(WC_MAX_BITS, WC_THREADS));
    SET_LDS(k_scatter_wc_n<WC_THREADS_SMALL>, wc_lds_bytes(WC_MAX_BITS, WC_THREADS_SMALL));
    SET_LDS(k_scatter_wcn<false>, wn_lds_bytes(WN_MAX_BITS));
    SET_LDS(k_scatter_wcn<true>, wn_lds_bytes(WN_MAX_BITS));
    SET_LDS((k_scatter_wcn<false, WN_GR, WN_TPT, WN_THREADS, true>), wn_lds_bytes(WN_MAX_BITS, WN_GR, WN_TPT, WN_THREADS, true));
    SET_LDS((k_scatter_wcn<false, WN9_GR, WN9_TPT, WN9_THREADS>), wn_lds_bytes(WN9_MAX_BITS, WN9_GR, WN9_TPT, WN9_THREADS));
    SET_LDS((k_scatter_wcn<true, WN9_GR, WN9_TPT, WN9_THREADS>), wn_lds_bytes(WN9_MAX_BITS, WN9_GR, WN9_TPT, WN9_THREADS));
    });
}

void launch_init_single_segment(hipStream_t st, u64 n, u64 L, u64 *d_seg_start, u32 *d_unit_start)
{
    hipLaunchKernelGGL(k_init_single_segment, dim3(1), dim3(64), 0, st, n, L, d_seg_start, d_unit_start);
}

void launch_make_units(hipStream_t st, const u64 *d_seg_start, u32 nseg, u64 L, u32 *d_unit_start)
{
    hipLaunchKernelGGL(k_make_units, dim3(1), dim3(1024), 0, st, d_seg_start, nseg, L, d_unit_start);
}

void launch_hist_units(hipStream_t st, const void *d_in, const PassGeom &g, const u64 *d_seg_start,
                       const u32 *d_unit_start, u32 *d_unit_hist, u64 *d_minmax, const DupSniff &sniff)
{
    if (g.max_units == 0) return;
    hipLaunchKernelGGL(k_hist_units, dim3(g.max_units), dim3(PART_THREADS), ((size_t)4 << g.bits), st,
                       (const Tup *)d_in, d_seg_start, d_unit_start, g.nseg, g.L, g.shift, g.bits, d_unit_hist, d_minmax, g.mix, sniff);
}

void launch_hist_units_narrow(hipStream_t st, const void *d_inP, const PassGeom &g, const u64 *d_seg_start,
                              const u32 *d_unit_start, u32 *d_unit_hist)
{
    if (g.max_units == 0) return;
    hipLaunchKernelGGL(k_hist_units_n, dim3(g.max_units), dim3(PART_THREADS), ((size_t)4 << g.bits), st, (const u64 *)d_inP,
                       d_seg_start, d_unit_start, g.nseg, g.L, g.shift, g.bits, d_unit_hist);
}

void launch_scan_units(hipStream_t st, const PassGeom &g, const u64 *d_seg_start, const u32 *d_unit_start,
                       const u32 *d_unit_hist, u64 *d_unit_base, u64 *d_part_start, u64 *d_scan_tmp)
{
    const size_t nbins = (size_t)1 << g.bits;
    if (g.nseg == 1 && g.max_units > 2 * SCAN_SLICES && d_scan_tmp != nullptr) {
        u32 nsl = 8;
        while (nsl < SCAN_SLICES && nsl * nsl < g.max_units) nsl++;
        hipLaunchKernelGGL(k_scan1_partial, dim3(nsl), dim3(1024), 0, st, d_unit_start, g.bits, d_unit_hist, d_scan_tmp);
        hipLaunchKernelGGL(k_scan1_mid, dim3(1), dim3(1024), 0, st, g.bits, nsl, d_scan_tmp, d_part_start, g.n);
        hipLaunchKernelGGL(k_scan1_final, dim3(nsl), dim3(1024), 0, st, d_unit_start, g.bits, d_unit_hist,
                           d_scan_tmp, d_unit_base);
        return;
    }
    const size_t G = 1024 / nbins;
    hipLaunchKernelGGL(k_scan_units, dim3(g.nseg), dim3(1024), G * nbins * 8, st, d_seg_start, d_unit_start,
                       g.nseg, g.bits, d_unit_hist, d_unit_base, d_part_start, g.n);
}

void launch_scatter_units(hipStream_t st, const void *d_in, void *d_out, const PassGeom &g,
                          const u64 *d_seg_start, const u32 *d_unit_start, const u64 *d_unit_base)
{
    if (g.max_units == 0) return;
    allow_big_lds();
    if (g.bits <= WC_MAX_BITS) {
        if (wc_threads_for(g.bits) == WC_THREADS_SMALL)
            hipLaunchKernelGGL(k_scatter_wc<WC_THREADS_SMALL>, dim3(g.max_units), dim3(WC_THREADS_SMALL),
                               wc_lds_bytes(g.bits, WC_THREADS_SMALL), st, (const Tup *)d_in, (Tup *)d_out, d_seg_start,
                               d_unit_start, g.nseg, g.L, g.shift, g.bits, d_unit_base, (const u64 *)nullptr, 0u, g.mix);
        else
            hipLaunchKernelGGL(k_scatter_wc<WC_THREADS>, dim3(g.max_units), dim3(WC_THREADS), wc_lds_bytes(g.bits, WC_THREADS),
                               st, (const Tup *)d_in, (Tup *)d_out, d_seg_start, d_unit_start, g.nseg, g.L, g.shift, g.bits,
                               d_unit_base, (const u64 *)nullptr, 0u, g.mix);
        return;
    }
    // 10-bit pass: carry lines (2^10 x 128 B) do not fit LDS beside a tile -> tile-sort form
    hipLaunchKernelGGL(k_scatter_units_pipe, dim3(g.max_units), dim3(PART_THREADS), part_lds_bytes(g.bits), st,
                       (const Tup *)d_in, (Tup *)d_out, d_seg_start, d_unit_start, g.nseg, g.L, g.shift, g.bits,
                       d_unit_base, g.mix);
}

// One partition pass over BOTH relations of a join, each kernel launched once (grid.y = relation).  bits <= WC_MAX_BITS.
// phase: 0 unit tables, 1 histogram, 2 scan, 3 scatter (separate calls so that the host can time them per kind).
void launch_pass_pair(hipStream_t st, const PassPairHost &h, int shift, int bits, int phase)
{
    allow_big_lds();
    PassPair a;
    a.mix = h.mix;
    u32 mu = 0;
    for (int i = 0; i < 2; i++) {
        const PassSide &x = h.side[i];
        a.r[i] = PassRel{(const Tup *)x.in, (Tup *)x.out, x.seg_start, x.unit_start, x.unit_hist, x.unit_base, x.part_start,
                         x.scan_tmp, x.g.n, x.g.L, x.g.max_units};
        mu = x.g.max_units > mu ? x.g.max_units : mu;
    }
    if (mu == 0) return;
    if (phase == 0) {
        hipLaunchKernelGGL(k_init_single_segment2, dim3(2), dim3(64), 0, st, a, h.zero8);
    } else if (phase == 1) {
        hipLaunchKernelGGL(k_hist_units2, dim3(mu, 2), dim3(PART_THREADS), ((size_t)4 << bits), st, a, shift, bits);
    } else if (phase == 2) {
        if (mu > 2 * SCAN_SLICES) {
            u32 nsl = 8;
            while (nsl < SCAN_SLICES && nsl * nsl < mu) nsl++;
            hipLaunchKernelGGL(k_scan1_partial2, dim3(nsl, 2), dim3(1024), 0, st, a, bits);
            hipLaunchKernelGGL(k_scan1_mid2, dim3(2), dim3(1024), 0, st, a, bits, nsl);
            hipLaunchKernelGGL(k_scan1_final2, dim3(nsl, 2), dim3(1024), 0, st, a, bits);
        } else {
            const size_t nbins = (size_t)1 << bits, G = 1024 / nbins;
            hipLaunchKernelGGL(k_scan_units2, dim3(2), dim3(1024), G * nbins * 8, st, a, bits);
        }
    } else if (wc_threads_for(bits) == WC_THREADS_SMALL) {
        hipLaunchKernelGGL(k_scatter_wc2<WC_THREADS_SMALL>, dim3(mu, 2), dim3(WC_THREADS_SMALL), wc_lds_bytes(bits, WC_THREADS_SMALL),
                           st, a, shift, bits);
    } else {
        hipLaunchKernelGGL(k_scatter_wc2<WC_THREADS>, dim3(mu, 2), dim3(WC_THREADS), wc_lds_bytes(bits, WC_THREADS), st, a, shift, bits);
    }
}

// One-pass join, partition phase in two launches (k_hist_fused2, k_scatter_fused2).  d_ctl: FUSE_CTL_BYTES of HBM, zero
// between calls (see FuseCtl).  phase 0: histogram + boundaries + task list; phase 1: scatter.
constexpr size_t FUSE_CURSOR_BYTES = (size_t)2 * FUSE_MAX_BINS * FUSE_STRIDE64 * 8;
constexpr size_t FUSE_GHIST_BYTES = (size_t)2 * FUSE_COPIES * FUSE_MAX_BINS * FUSE_STRIDE32 * 4;
constexpr size_t FUSE_SNIFF_BYTES = (size_t)2 * SNIFF_SLOTS * 4;
constexpr size_t FUSE_COPY_BYTES = FUSE_CURSOR_BYTES + FUSE_GHIST_BYTES + FUSE_SNIFF_BYTES;
size_t fuse_ctl_bytes() { return 2 * FUSE_COPY_BYTES + 256; }
u32 *fuse_join_ticket(void *d_ctl) { return (u32 *)((unsigned char *)d_ctl + 2 * FUSE_COPY_BYTES + 128); }
void launch_fused_pass(hipStream_t st, const PassPairHost &h, int bits, int phase, int parity, void *d_ctl, u32 probe_split, u32 max_tasks,
                       u32 table_tuples, JoinTask *d_tasks, u64 *d_counters, u64 *host_pub, bool sniff)
{
    allow_big_lds();
    PassPair a;
    a.mix = h.mix;
    u32 mu = 0;
    for (int i = 0; i < 2; i++) {
        const PassSide &x = h.side[i];
        a.r[i] = PassRel{(const Tup *)x.in, (Tup *)x.out, x.seg_start, x.unit_start, x.unit_hist, x.unit_base, x.part_start,
                         x.scan_tmp, x.g.n, x.g.L, x.g.max_units};
        const u32 units = (u32)((x.g.n + x.g.L - 1) / x.g.L);
        mu = units > mu ? units : mu;
    }
    if (mu == 0) mu = 1;                                                     // (both relations empty: the ticket logic still runs)
    FuseCtl fc;
    unsigned char *mine = (unsigned char *)d_ctl + (size_t)(parity & 1) * FUSE_COPY_BYTES, *next = (unsigned char *)d_ctl + (size_t)(~parity & 1) * FUSE_COPY_BYTES;
    fc.cursor = (u64 *)mine;
    fc.ghist = (u32 *)(mine + FUSE_CURSOR_BYTES);
    fc.cursor_next = (u64 *)next;
    fc.ghist_next = (u32 *)(next + FUSE_CURSOR_BYTES);
    fc.sniff = sniff ? (u32 *)(mine + FUSE_CURSOR_BYTES + FUSE_GHIST_BYTES) : nullptr;
    fc.sniff_next = (u32 *)(next + FUSE_CURSOR_BYTES + FUSE_GHIST_BYTES);
    SniffVerdict sv;
    for (int i = 0; i < 2; i++) fc.sel_bits[i] = sniff_sel_bits(h.side[i].g.n);
    if (sniff) {
        sv.tab = fc.sniff;
        sv.expect_R = (u32)(h.side[0].g.n >> fc.sel_bits[0]);
        sv.expect_S = (u32)(h.side[1].g.n >> fc.sel_bits[1]);
    }
    u32 k = 1;
    if (phase == 0) {
        // units per workgroup: as few global-histogram atomics as a full chip allows (>= ~2 workgroups per CU stay)
        static const u32 forced_k = getenv("RHJ_FUSE_K") ? (u32)atoi(getenv("RHJ_FUSE_K")) : 0u;      // tuning aid
        // ([measured] 10^6 x 10^6, 8 bits: k = 1 / 2 / 4 -> 27 / 24 / 24 us; 4 * 10^6, 9 bits: k = 1 / 2 / 4 / 8 -> 81 / 61 / 55 / 53 us):
        // at most ~32 K atomics per launch while at least 64 workgroups per relation remain, at most 8 units each
        k = (u32)(((u64)mu * ((u64)2 << bits) + 32767) / 32768);
        if (k > mu / 64) k = mu / 64;
        if (k > 8) k = 8;
        if (forced_k) k = forced_k;
        if (k < 1) k = 1;
    }
    const FuseTasks ft{probe_split, max_tasks, table_tuples, k, d_tasks, d_counters, host_pub, build_tie_shift(), sv};
    if (phase == 0) {
        hipLaunchKernelGGL(k_hist_fused2, dim3((mu + k - 1) / k, 2), dim3(PART_THREADS), ((size_t)8 << bits), st, a, 0, bits, fc, ft);
    } else if (wc_threads_for(bits) == WC_THREADS_SMALL && bits <= 8) {
        hipLaunchKernelGGL(k_scatter_fused2<WC_THREADS_SMALL>, dim3(mu + 1, 2), dim3(WC_THREADS_SMALL), wc_lds_bytes(bits, WC_THREADS_SMALL),
                           st, a, 0, bits, fc, ft);
    } else {
        hipLaunchKernelGGL(k_scatter_fused2<WC_THREADS>, dim3(mu + 1, 2), dim3(WC_THREADS), wc_lds_bytes(bits, WC_THREADS), st, a, 0, bits, fc, ft);
    }
}

bool fused_two_pass_ok(int b1, int b2) { return b1 >= 1 && b2 >= 1 && b1 <= WC_MAX_BITS && b2 <= WC_MAX_BITS && b1 + b2 <= 16; }

void launch_hist2d_units(hipStream_t st, const void *d_in, bool in_narrow, u64 n, u64 L, u32 units, int b1, int b2,
                         u32 units_per_group, u32 ngroups, u32 *d_hist1, u32 *d_hist2, u64 key_base, u32 *d_wide,
                         const u64 *d_unit_rng, int mix, const DupSniff &sniff)
{
    static std::once_flag once[64];
    const size_t lds = ((size_t)1 << (b1 + b2)) * 2 + ((size_t)4 << b1);
    std::call_once(once[current_device_slot()], [] {
        SET_LDS(k_hist2d_units<false>, ((size_t)1 << 16) * 2 + ((size_t)4 << WC_MAX_BITS));
        SET_LDS(k_hist2d_units<true>, ((size_t)1 << 16) * 2 + ((size_t)4 << WC_MAX_BITS));
    });
    if (units == 0) return;
    if (in_narrow)
        hipLaunchKernelGGL(k_hist2d_units<true>, dim3(units), dim3(H2_THREADS), lds, st, (const Tup *)nullptr, (const u64 *)d_in, n,
                           L, b1, b2, units_per_group, ngroups, d_hist1, d_hist2, (u64)0, (u32 *)nullptr, d_unit_rng, 0, sniff);
    else
        hipLaunchKernelGGL(k_hist2d_units<false>, dim3(units), dim3(H2_THREADS), lds, st, (const Tup *)d_in, (const u64 *)nullptr, n,
                           L, b1, b2, units_per_group, ngroups, d_hist1, d_hist2, key_base, d_wide, d_unit_rng, mix, sniff);
}

// pass-1 units cut at segment boundaries (multi-GPU receiver); d_unit_rng gets nseg * units_per_seg + 1 entries
void launch_seg_units(hipStream_t st, u32 nseg, const u64 *seg_off, const u64 *seg_L, u32 units_per_seg, u64 *d_unit_rng,
                      u64 *d_seg_start, u32 *d_unit_start)
{
    SegPlan sp{};
    sp.nseg = nseg;
    sp.units_per_seg = units_per_seg;
    for (u32 i = 0; i <= nseg && i <= SEG_MAX; i++) sp.off[i] = seg_off[i];
    for (u32 i = 0; i < nseg && i < SEG_MAX; i++) sp.L[i] = seg_L[i];
    const u32 total = nseg * units_per_seg + 1;
    hipLaunchKernelGGL(k_seg_units, dim3((total + 255) / 256), dim3(256), 0, st, sp, d_unit_rng, d_seg_start, d_unit_start);
}
int seg_max() { return SEG_MAX; }

void launch_make_group_ranges(hipStream_t st, const u64 *d_unit_base1, u32 nb1, u32 units_per_group, u32 ngroups, u64 n,
                              u64 *d_rng, u32 *d_unit_start2)
{
    const u32 total = nb1 * ngroups + 1;
    hipLaunchKernelGGL(k_make_group_ranges, dim3((total + 255) / 256), dim3(256), 0, st, d_unit_base1, nb1, units_per_group,
                       ngroups, n, d_rng, d_unit_start2);
}

void launch_scatter_ranges(hipStream_t st, const void *d_in, void *d_out, u32 nunits, int shift, int bits,
                           const u64 *d_unit_base, const u64 *d_rng)
{
    if (nunits == 0) return;
    allow_big_lds();
    if (wc_threads_for(bits) == WC_THREADS_SMALL)
        hipLaunchKernelGGL(k_scatter_wc<WC_THREADS_SMALL>, dim3(nunits), dim3(WC_THREADS_SMALL), wc_lds_bytes(bits, WC_THREADS_SMALL),
                           st, (const Tup *)d_in, (Tup *)d_out, (const u64 *)nullptr, (const u32 *)nullptr, 0u, (u64)0, shift,
                           bits, d_unit_base, d_rng, nunits, 0);
    else
        hipLaunchKernelGGL(k_scatter_wc<WC_THREADS>, dim3(nunits), dim3(WC_THREADS), wc_lds_bytes(bits, WC_THREADS), st,
                           (const Tup *)d_in, (Tup *)d_out, (const u64 *)nullptr, (const u32 *)nullptr, 0u, (u64)0, shift, bits,
                           d_unit_base, d_rng, nunits, 0);
}

// last pass of the multi-GPU receiver in front of the compact-table join: narrow in (rowIDs at narrow_k_offset(n)), 16-byte
// tuples with global rowIDs out; d_key_bases: 16 u64 on the device, one per sender
void launch_scatter_ranges_n2a(hipStream_t st, const void *d_in, void *d_out, u64 n, u32 nunits, int shift, int bits,
                               const u64 *d_unit_base, const u64 *d_rng, const u64 *d_key_bases, u32 tag_groups, u32 tag_div,
                               const u32 *d_skip)
{
    if (nunits == 0) return;
    allow_big_lds();
    const u64 *iP = (const u64 *)d_in;
    const u32 *iK = (const u32 *)((const unsigned char *)d_in + narrow_k_offset(n));
    const WnTag tag{tag_groups, tag_div, 0u};
    if (wc_threads_for(bits) == WC_THREADS_SMALL)
        hipLaunchKernelGGL(k_scatter_wc_n<WC_THREADS_SMALL>, dim3(nunits), dim3(WC_THREADS_SMALL), wc_lds_bytes(bits, WC_THREADS_SMALL),
                           st, iP, iK, (Tup *)d_out, shift, bits, d_unit_base, d_rng, nunits, d_key_bases, tag, d_skip);
    else
        hipLaunchKernelGGL(k_scatter_wc_n<WC_THREADS>, dim3(nunits), dim3(WC_THREADS), wc_lds_bytes(bits, WC_THREADS), st, iP, iK,
                           (Tup *)d_out, shift, bits, d_unit_base, d_rng, nunits, d_key_bases, tag, d_skip);
}

// Narrow-format scatters (k_scatter_wcn).  A narrow relation of n tuples lives in one buffer of >= 16 n bytes: payloads
// (u64) at offset 0, rowIDs (u32) at narrow_k_offset(n).
bool narrow_pass_ok(int bits) { return bits >= 1 && bits <= WN_MAX_BITS; }
bool narrow_pass9_ok(int bits) { return bits >= 1 && bits <= WN9_MAX_BITS; }

void launch_scatter_units_narrow(hipStream_t st, const void *d_in, void *d_out, u64 n, const PassGeom &g,
                                 const u64 *d_seg_start, const u32 *d_unit_start, const u64 *d_unit_base, u32 *d_overflow,
                                 u64 key_base)
{
    if (g.max_units == 0) return;
    allow_big_lds();
    hipLaunchKernelGGL(k_scatter_wcn<false>, dim3(g.max_units), dim3(WN_THREADS), wn_lds_bytes(g.bits), st, (const Tup *)d_in,
                       (const u64 *)nullptr, (const u32 *)nullptr, (u64 *)d_out,
                       (u32 *)((unsigned char *)d_out + narrow_k_offset(n)), d_seg_start, d_unit_start, g.nseg, g.L, g.shift,
                       g.bits, d_unit_base, (const u64 *)nullptr, 0u, d_overflow, key_base, WnTag{1u, 1u, 0u}, g.mix, WnPeer{});
}

// the multi-GPU sender's class split straight into the owners' receive arrays (k_scatter_wcn<.., PEER>): d_delta / d_owner:
// 2^bits entries on the device; peersP / peersK: nranks device pointers (this rank's own arrays among them)
void launch_scatter_units_narrow_peer(hipStream_t st, const void *d_in, const PassGeom &g, const u64 *d_seg_start,
                                      const u32 *d_unit_start, const u64 *d_unit_base, u32 *d_overflow, u64 key_base,
                                      const u64 *d_delta, const unsigned char *d_owner, void *const *peersP, void *const *peersK,
                                      int nranks)
{
    if (g.max_units == 0) return;
    allow_big_lds();
    WnPeer peer{};
    peer.delta = d_delta;
    peer.owner = d_owner;
    for (int i = 0; i < nranks && i < SEG_MAX; i++) { peer.P[i] = (u64 *)peersP[i]; peer.K[i] = (u32 *)peersK[i]; }
    hipLaunchKernelGGL((k_scatter_wcn<false, WN_GR, WN_TPT, WN_THREADS, true>), dim3(g.max_units), dim3(WN_THREADS),
                       wn_lds_bytes(g.bits, WN_GR, WN_TPT, WN_THREADS, true), st, (const Tup *)d_in, (const u64 *)nullptr,
                       (const u32 *)nullptr, (u64 *)nullptr, (u32 *)nullptr, d_seg_start, d_unit_start, g.nseg, g.L, g.shift, g.bits,
                       d_unit_base, (const u64 *)nullptr, 0u, d_overflow, key_base, WnTag{1u, 1u, 0u}, g.mix, peer);
}

// One narrow-output pass over segments cut into units (run_pass form): 16-byte or narrow input, <= 8 bits (32-tuple lines)
// or 9 bits (16-tuple lines).  d_inK: rowIDs of a narrow input.
void launch_scatter_units_narrow_any(hipStream_t st, const void *d_in, const u32 *d_inK, void *d_outP, u32 *d_outK, const PassGeom &g,
                                     const u64 *d_seg_start, const u32 *d_unit_start, const u64 *d_unit_base, u32 *d_overflow)
{
    if (g.max_units == 0) return;
    allow_big_lds();
    const bool in_narrow = d_inK != nullptr;
    const WnTag notag{1u, 1u, 0u};
#define WCN_ARGS (const Tup *)(in_narrow ? nullptr : d_in), (const u64 *)(in_narrow ? d_in : nullptr), d_inK, (u64 *)d_outP, d_outK,    \
                 d_seg_start, d_unit_start, g.nseg, g.L, g.shift, g.bits, d_unit_base, (const u64 *)nullptr, 0u, d_overflow, (u64)0, notag, g.mix, WnPeer{}
    if (g.bits <= WN_MAX_BITS) {
        if (in_narrow) hipLaunchKernelGGL(k_scatter_wcn<true>, dim3(g.max_units), dim3(WN_THREADS), wn_lds_bytes(g.bits), st, WCN_ARGS);
        else hipLaunchKernelGGL(k_scatter_wcn<false>, dim3(g.max_units), dim3(WN_THREADS), wn_lds_bytes(g.bits), st, WCN_ARGS);
    } else {
        const size_t lds = wn_lds_bytes(g.bits, WN9_GR, WN9_TPT, WN9_THREADS);
        if (in_narrow) hipLaunchKernelGGL((k_scatter_wcn<true, WN9_GR, WN9_TPT, WN9_THREADS>), dim3(g.max_units), dim3(WN9_THREADS), lds, st, WCN_ARGS);
        else hipLaunchKernelGGL((k_scatter_wcn<false, WN9_GR, WN9_TPT, WN9_THREADS>), dim3(g.max_units), dim3(WN9_THREADS), lds, st, WCN_ARGS);
    }
#undef WCN_ARGS
}

// explicit unit ranges [d_rng[u], d_rng[u+1]); tag_groups / tag_div != 0: the low TAG_BITS bits of every payload written are
// replaced by (u % tag_groups) / tag_div (see WnTag)
void launch_scatter_ranges_narrow(hipStream_t st, const void *d_in, bool in_narrow, void *d_out, u64 n, u32 nunits, int shift,
                                  int bits, const u64 *d_unit_base, const u64 *d_rng, u32 *d_overflow, u32 tag_groups,
                                  u32 tag_div, const u32 *d_inK)
{
    if (nunits == 0) return;
    allow_big_lds();
    u64 *oP = (u64 *)d_out;
    u32 *oK = (u32 *)((unsigned char *)d_out + narrow_k_offset(n));
    const WnTag tag = tag_div ? WnTag{tag_groups, tag_div, TAG_BITS} : WnTag{1u, 1u, 0u};
    if (in_narrow)
        hipLaunchKernelGGL(k_scatter_wcn<true>, dim3(nunits), dim3(WN_THREADS), wn_lds_bytes(bits), st, (const Tup *)nullptr,
                           (const u64 *)d_in, d_inK ? d_inK : (const u32 *)((const unsigned char *)d_in + narrow_k_offset(n)), oP, oK,
                           (const u64 *)nullptr, (const u32 *)nullptr, 0u, (u64)0, shift, bits, d_unit_base, d_rng, nunits,
                           d_overflow, (u64)0, tag, 0, WnPeer{});
    else
        hipLaunchKernelGGL(k_scatter_wcn<false>, dim3(nunits), dim3(WN_THREADS), wn_lds_bytes(bits), st, (const Tup *)d_in,
                           (const u64 *)nullptr, (const u32 *)nullptr, oP, oK, (const u64 *)nullptr, (const u32 *)nullptr, 0u,
                           (u64)0, shift, bits, d_unit_base, d_rng, nunits, d_overflow, (u64)0, tag, 0, WnPeer{});
}
int tag_bits() { return (int)TAG_BITS; }

void launch_check_radix(hipStream_t st, const void *d_R, const u64 *d_startR, const void *d_S, const u64 *d_startS, u64 nparts,
                        int radix_bits, u64 *d_bad)
{
    const unsigned grid = (unsigned)(nparts < 4096 ? nparts : 4096);
    hipLaunchKernelGGL(k_check_radix, dim3(grid ? grid : 1), dim3(256), 0, st, (const Tup *)d_R, d_startR, (const Tup *)d_S, d_startS,
                       nparts, radix_bits, d_bad);
}

void launch_diff_hist(hipStream_t st, const u64 *d_start, u64 nbins, u64 *d_hist)
{
    hipLaunchKernelGGL(k_diff_hist, dim3((unsigned)((nbins + 255) / 256)), dim3(256), 0, st, d_start, nbins, d_hist);
}

void launch_prefix(hipStream_t st, const u64 *d_hist, u64 nbins, u64 *d_start)
{
    hipLaunchKernelGGL(k_prefix, dim3(1), dim3(1024), 0, st, d_hist, nbins, d_start);
}

void launch_make_tasks(hipStream_t st, const u64 *d_startR, const u64 *d_startS, u64 nparts, u32 probe_split,
                       JoinTask *d_tasks, u32 *d_ntasks, u32 max_tasks, u64 *d_stats, int kind, const SniffVerdict &sniff)
{
    const int own_max = nparts <= 1024 ? 1 : 0;        // a single workgroup of k_make_tasks sees every partition
    if (!own_max) {
        u64 g = (nparts + 255) / 256;
        if (g > 1024) g = 1024;
        hipLaunchKernelGGL(k_part_max, dim3((unsigned)g), dim3(256), 0, st, d_startR, d_startS, nparts, d_stats);   // d_stats zeroed by the caller
    }
    hipLaunchKernelGGL(k_make_tasks, dim3((unsigned)((nparts + 1023) / 1024)), dim3(1024), 0, st, d_startR, d_startS,
                       nparts, probe_split, d_tasks, d_ntasks, max_tasks, d_stats,
                       join_table_tuples(kind), own_max, build_tie_shift(), sniff);
}

void launch_join(hipStream_t st, const void *d_R, const u64 *d_startR, const void *d_S, const u64 *d_startS,
                 const JoinTask *d_tasks, const u32 *d_ntasks, u32 grid, int radix_bits,
                 void *d_out, u64 out_capacity, u64 *d_out_count, int kind, const u32 *d_RK, const u32 *d_SK,
                 const u64 *d_tag_base, const u32 *d_skip, u64 *host_pub, u32 *d_done)
{
    if (grid == 0) return;
    allow_big_lds();
    DirectJoin pub{};                                                        // (JK_BKT / JK_BKT_BIG over 16-byte tuples only)
    pub.host_count = host_pub;
    pub.done = d_done;
    const RelView<false> vR{(const Tup *)d_R}, vS{(const Tup *)d_S};
    if (d_RK != nullptr) {                                                   // narrow partitions (k_scatter_wcn): d_R, d_S are payload arrays
        const RelView<true> nR{(const u64 *)d_R, d_RK}, nS{(const u64 *)d_S, d_SK};
        Pair *o = (Pair *)d_out;
#define LAUNCH_BKT_N(TG) hipLaunchKernelGGL((k_join_bkt<BJ_THREADS, BJ_CHUNK, BJ_BUCKET_BITS, BJ_EPT, false, true, TG>), dim3(grid), \
            dim3(BJ_THREADS), bj_lds_bytes(BJ_THREADS, BJ_CHUNK, BJ_BUCKET_BITS), st, nR, nS, d_tasks, d_ntasks, radix_bits, o,       \
            out_capacity, d_out_count, DirectJoin{}, d_tag_base, d_skip)
#define LAUNCH_CT_N(T, C, B, E) hipLaunchKernelGGL((k_join_ct<T, C, B, E, false, true>), dim3(grid), dim3(T),                        \
            ct_lds_bytes(T, C, B), st, nR, nS, d_tasks, d_ntasks, radix_bits, o, out_capacity, d_out_count, (u64 *)nullptr, 0u, d_skip)
        const bool tg = d_tag_base != nullptr;                              // sender tags: the one-table kernel only (the host sees to it)
        if (kind == JK_BKT) { if (tg) LAUNCH_BKT_N(true); else LAUNCH_BKT_N(false); }
        else if (kind == JK_CT_HALF) LAUNCH_CT_N(CTH_THREADS, CTH_CHUNK, CTH_BUCKET_BITS, CT_EPT);
        else if (kind == JK_CT_HALF_WIDE) LAUNCH_CT_N(CTH_THREADS, CTHW_CHUNK, CTHW_BUCKET_BITS, CT_EPT_WIDE);
        else if (kind == JK_CT_WIDE) LAUNCH_CT_N(CT_THREADS, CT13_CHUNK, CT13_BUCKET_BITS, CT_EPT_WIDE);
        else if (kind == JK_CT_13) LAUNCH_CT_N(CT_THREADS, CT13_CHUNK, CT13_BUCKET_BITS, CT_EPT);
        else if (kind == JK_CT_MID) LAUNCH_CT_N(CT_THREADS, CTM_CHUNK, CTM_BUCKET_BITS, CTM_EPT);
        else if (kind == JK_CT_HALF_MID) LAUNCH_CT_N(CTH_THREADS, CTHM_CHUNK, CTHM_BUCKET_BITS, CTM_EPT);
        else if (kind == JK_CT_Q12)
            hipLaunchKernelGGL((k_join_ct<CTH_THREADS, CTQ_CHUNK, CTQ_BUCKET_BITS, CTQ_EPT, false, true, true, CTQ_KB>), dim3(grid), dim3(CTH_THREADS),
                               ct_lds_bytes(CTH_THREADS, CTQ_CHUNK, CTQ_BUCKET_BITS), st, nR, nS, d_tasks, d_ntasks, radix_bits, o, out_capacity,
                               d_out_count, (u64 *)nullptr, 0u, d_skip);
        else if (kind == JK_CT_G13)
            hipLaunchKernelGGL((k_join_ct<CTH_THREADS, CTHM_CHUNK, CTHM_BUCKET_BITS, CTM_EPT, false, true, true, CT13_KB>), dim3(grid), dim3(CTH_THREADS),
                               ct_lds_bytes(CTH_THREADS, CTHM_CHUNK, CTHM_BUCKET_BITS), st, nR, nS, d_tasks, d_ntasks, radix_bits, o, out_capacity,
                               d_out_count, (u64 *)nullptr, 0u, d_skip);
        else if (kind == JK_CT_HALF_MID_G)
            hipLaunchKernelGGL((k_join_ct<CTH_THREADS, CTHM_CHUNK, CTHM_BUCKET_BITS, CTM_EPT, false, true, true>), dim3(grid), dim3(CTH_THREADS),
                               ct_lds_bytes(CTH_THREADS, CTHM_CHUNK, CTHM_BUCKET_BITS), st, nR, nS, d_tasks, d_ntasks, radix_bits, o, out_capacity,
                               d_out_count, (u64 *)nullptr, 0u, d_skip);
        else LAUNCH_CT_N(CT_THREADS, CT_CHUNK, CT_BUCKET_BITS, CT_EPT);     // JK_CT (the host never asks for another kind here)
#undef LAUNCH_BKT_N
#undef LAUNCH_CT_N
        return;
    }
    if (kind == JK_BKT) {
        hipLaunchKernelGGL((k_join_bkt<BJ_THREADS, BJ_CHUNK, BJ_BUCKET_BITS, BJ_EPT, false>), dim3(grid), dim3(BJ_THREADS),
                           bj_lds_bytes(BJ_THREADS, BJ_CHUNK, BJ_BUCKET_BITS), st, vR, vS,
                           d_tasks, d_ntasks, radix_bits, (Pair *)d_out, out_capacity, d_out_count, pub);
        return;
    }
    if (kind == JK_BKT_BIG) {
        hipLaunchKernelGGL((k_join_bkt<BJ2_THREADS, BJ2_CHUNK, BJ2_BUCKET_BITS, BJ2_EPT, false>), dim3(grid), dim3(BJ2_THREADS),
                           bj_lds_bytes(BJ2_THREADS, BJ2_CHUNK, BJ2_BUCKET_BITS), st, vR, vS,
                           d_tasks, d_ntasks, radix_bits, (Pair *)d_out, out_capacity, d_out_count, pub);
        return;
    }
    if (kind == JK_CT_HALF_MID) {
        hipLaunchKernelGGL((k_join_ct<CTH_THREADS, CTHM_CHUNK, CTHM_BUCKET_BITS, CTM_EPT, false, false>), dim3(grid), dim3(CTH_THREADS),
                           ct_lds_bytes(CTH_THREADS, CTHM_CHUNK, CTHM_BUCKET_BITS), st, vR, vS, d_tasks,
                           d_ntasks, radix_bits, (Pair *)d_out, out_capacity, d_out_count, (u64 *)nullptr, 0u, (const u32 *)nullptr);
        return;
    }
    if (kind == JK_CT_Q12) {
        hipLaunchKernelGGL((k_join_ct<CTH_THREADS, CTQ_CHUNK, CTQ_BUCKET_BITS, CTQ_EPT, false, false, true, CTQ_KB>), dim3(grid), dim3(CTH_THREADS),
                           ct_lds_bytes(CTH_THREADS, CTQ_CHUNK, CTQ_BUCKET_BITS), st, vR, vS, d_tasks,
                           d_ntasks, radix_bits, (Pair *)d_out, out_capacity, d_out_count, (u64 *)nullptr, 0u, (const u32 *)nullptr);
        return;
    }
    if (kind == JK_CT_G13) {
        hipLaunchKernelGGL((k_join_ct<CTH_THREADS, CTHM_CHUNK, CTHM_BUCKET_BITS, CTM_EPT, false, false, true, CT13_KB>), dim3(grid), dim3(CTH_THREADS),
                           ct_lds_bytes(CTH_THREADS, CTHM_CHUNK, CTHM_BUCKET_BITS), st, vR, vS, d_tasks,
                           d_ntasks, radix_bits, (Pair *)d_out, out_capacity, d_out_count, (u64 *)nullptr, 0u, (const u32 *)nullptr);
        return;
    }
    if (kind == JK_CT_HALF_MID_G) {
        hipLaunchKernelGGL((k_join_ct<CTH_THREADS, CTHM_CHUNK, CTHM_BUCKET_BITS, CTM_EPT, false, false, true>), dim3(grid), dim3(CTH_THREADS),
                           ct_lds_bytes(CTH_THREADS, CTHM_CHUNK, CTHM_BUCKET_BITS), st, vR, vS, d_tasks,
                           d_ntasks, radix_bits, (Pair *)d_out, out_capacity, d_out_count, (u64 *)nullptr, 0u, (const u32 *)nullptr);
        return;
    }
    if (kind == JK_CT_MID) {
        hipLaunchKernelGGL((k_join_ct<CT_THREADS, CTM_CHUNK, CTM_BUCKET_BITS, CTM_EPT, false, false>), dim3(grid), dim3(CT_THREADS),
                           ct_lds_bytes(CT_THREADS, CTM_CHUNK, CTM_BUCKET_BITS), st, vR, vS, d_tasks,
                           d_ntasks, radix_bits, (Pair *)d_out, out_capacity, d_out_count, (u64 *)nullptr, 0u, (const u32 *)nullptr);
        return;
    }
    if (kind == JK_CT_13) {
        hipLaunchKernelGGL((k_join_ct<CT_THREADS, CT13_CHUNK, CT13_BUCKET_BITS, CT_EPT, false, false>), dim3(grid), dim3(CT_THREADS),
                           ct_lds_bytes(CT_THREADS, CT13_CHUNK, CT13_BUCKET_BITS), st, vR, vS, d_tasks,
                           d_ntasks, radix_bits, (Pair *)d_out, out_capacity, d_out_count, (u64 *)nullptr, 0u, (const u32 *)nullptr);
        return;
    }
    if (kind == JK_CT_HALF) {
        hipLaunchKernelGGL((k_join_ct<CTH_THREADS, CTH_CHUNK, CTH_BUCKET_BITS, CT_EPT, false, false>), dim3(grid), dim3(CTH_THREADS),
                           ct_lds_bytes(CTH_THREADS, CTH_CHUNK, CTH_BUCKET_BITS), st, vR, vS, d_tasks,
                           d_ntasks, radix_bits, (Pair *)d_out, out_capacity, d_out_count, (u64 *)nullptr, 0u, (const u32 *)nullptr);
        return;
    }
    static const bool want_stamps = getenv("RHJ_CT_STAMPS") != nullptr;
    const u32 nw = grid < 4096 ? grid : 4096;
    u64 *d_st = nullptr;
    if (want_stamps && hipMalloc(&d_st, (size_t)nw * CT_NSTAMP * 8) != hipSuccess) { (void)hipGetLastError(); d_st = nullptr; }
    if (d_st != nullptr) {                                                   // tuning aid: phase timeline of the first workgroups
        (void)hipMemsetAsync(d_st, 0, (size_t)nw * CT_NSTAMP * 8, st);
        hipLaunchKernelGGL((k_join_ct<CT_THREADS, CT_CHUNK, CT_BUCKET_BITS, CT_EPT, true, false>), dim3(grid), dim3(CT_THREADS),
                           ct_lds_bytes(), st, vR, vS, d_tasks, d_ntasks, radix_bits,
                           (Pair *)d_out, out_capacity, d_out_count, d_st, nw, (const u32 *)nullptr);
        std::vector<u64> h((size_t)nw * CT_NSTAMP);
        (void)hipMemcpyAsync(h.data(), d_st, h.size() * 8, hipMemcpyDeviceToHost, st);
        (void)hipStreamSynchronize(st);
        (void)hipFree(d_st);
        double acc[CT_NSTAMP] = {};
        u32 used = 0;
        for (u32 g = 256; g < nw; g++) {                                     // skip the first wave of workgroups (cold start)
            const u64 *r = &h[(size_t)g * CT_NSTAMP];
            if (!r[0] || !r[9] || r[10]) continue;                           // single-chunk tasks: 10 stamps
            used++;
            for (int i = 1; i < 10; i++) acc[i] += (double)(r[i] - r[i - 1]) * 0.01;  // 100 MHz -> us
        }
        if (used) {
            static const char *nm[10] = {"", "desc+build.load", "sort", "probe", "generic", "rid.issue", "barrier", "rid.dump",
                                         "reserve", "store"};
            double tot = 0;
            fprintf(stderr, "[k_join_ct timeline, us, mean of %u single-chunk workgroups]", used);
            for (int i = 1; i < 10; i++) { fprintf(stderr, " %s=%.2f", nm[i], acc[i] / used); tot += acc[i] / used; }
            fprintf(stderr, " total=%.2f\n", tot);
        }
        return;
    }
    hipLaunchKernelGGL((k_join_ct<CT_THREADS, CT_CHUNK, CT_BUCKET_BITS, CT_EPT, false, false>), dim3(grid), dim3(CT_THREADS),
                       ct_lds_bytes(), st, vR, vS, d_tasks, d_ntasks, radix_bits,
                       (Pair *)d_out, out_capacity, d_out_count, (u64 *)nullptr, 0u, (const u32 *)nullptr);
}

// Unpartitioned join of two small relations in ONE launch: build side = S when nR >= nS (JobScheduler.cpp:187).
void launch_join_direct(hipStream_t st, const void *d_R, u64 nR, const void *d_S, u64 nS, void *d_out, u64 out_capacity,
                        u64 *d_out_count, u64 *host_count, u32 *d_done, void *host_out, u64 host_cap)
{
    allow_big_lds();
    DirectJoin dj;
    dj.host_count = host_count;
    dj.done = d_done;
    dj.host_out = (Pair *)host_out;
    dj.host_cap = host_out ? host_cap : 0;
    dj.build_is_S = nR >= nS + (nS >> build_tie_shift()) ? 1u : 0u;      // (the smaller side; near ties: R -- see build_on_S)
    dj.nb = (u32)(dj.build_is_S ? nS : nR);
    dj.np = (u32)(dj.build_is_S ? nR : nS);
    dj.split = (u32)BJ_TILE;
    const u32 grid = (dj.np + dj.split - 1) / dj.split;
    hipLaunchKernelGGL((k_join_bkt<BJ_THREADS, BJ_CHUNK, BJ_BUCKET_BITS, BJ_EPT, true>), dim3(grid), dim3(BJ_THREADS),
                       bj_lds_bytes(BJ_THREADS, BJ_CHUNK, BJ_BUCKET_BITS), st, RelView<false>{(const Tup *)d_R},
                       RelView<false>{(const Tup *)d_S}, (const JoinTask *)nullptr, (const u32 *)nullptr, 0, (Pair *)d_out,
                       out_capacity, d_out_count, dj);
}

// up to 16 direct joins in one launch: d_batch = BatchJoinDesc[njoins] in HBM, max_blocks = the largest nblocks among them
void launch_join_batch(hipStream_t st, const BatchJoinDesc *d_batch, u32 njoins, u32 max_blocks)
{
    if (njoins == 0 || max_blocks == 0) return;
    allow_big_lds();
    hipLaunchKernelGGL((k_join_bkt<BJ_THREADS, BJ_CHUNK, BJ_BUCKET_BITS, BJ_EPT, true, false, false, true>), dim3(max_blocks, njoins),
                       dim3(BJ_THREADS), bj_lds_bytes(BJ_THREADS, BJ_CHUNK, BJ_BUCKET_BITS), st, RelView<false>{nullptr},
                       RelView<false>{nullptr}, (const JoinTask *)nullptr, (const u32 *)nullptr, 0, (Pair *)nullptr, (u64)0,
                       (u64 *)nullptr, DirectJoin{}, (const u64 *)nullptr, (const u32 *)nullptr, d_batch);
}
u32 join_direct_tile() { return (u32)BJ_TILE; }

static unsigned stream_grid(u64 n)
{
    u64 g = (n + 255) / 256;
    if (g > 4096) g = 4096;
    if (g == 0) g = 1;
    return (unsigned)g;
}

void launch_checksum(hipStream_t st, const void *d_pairs, u64 n, u64 *d_sum)
{
    hipLaunchKernelGGL(k_checksum, dim3(stream_grid(n)), dim3(256), 0, st, (const Pair *)d_pairs, n, d_sum);
}

void launch_expected_pkfk(hipStream_t st, const void *d_S, u64 n, u64 *d_sum)
{
    hipLaunchKernelGGL(k_expected_pkfk, dim3(stream_grid(n)), dim3(256), 0, st, (const Tup *)d_S, n, d_sum);
}

void launch_remap_keys(hipStream_t st, void *d_rel, u64 n, int shift, u64 add)
{
    hipLaunchKernelGGL(k_remap_keys, dim3(stream_grid(n)), dim3(256), 0, st, (Tup *)d_rel, n, shift, add);
}

void launch_generate(hipStream_t st, int kind, void *d_out, u64 n, u64 row0, u64 D, u64 seed, double theta)
{
    hipLaunchKernelGGL(k_generate, dim3(stream_grid(n)), dim3(256), 0, st, kind, (Tup *)d_out, n, row0, D, seed, theta);
}
