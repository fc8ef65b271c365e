// rhj_internal.h -- shared between rhj_kernels.hip (device code + launchers) and rhj_api.hip (C-ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned long long u64;
typedef unsigned int u32;

// ---- partition pass geometry ------------------------------------------------------------------
// A pass partitions every SEGMENT of the input (pass 1: the whole relation; pass 2: each pass-1
// bucket) by `bits` radix bits.  A segment is cut into UNITS of at most L tuples; one workgroup
// owns one unit in the histogram kernel and again in the scatter kernel (the reference's row
// ranges, structs.cpp:146-161, with ranges = units instead of 8 threads).
constexpr int PART_THREADS = 512;                 // 8 wavefronts
constexpr int PART_TPT = 8;                       // tuples per thread per tile
constexpr int PART_TILE = PART_THREADS * PART_TPT;  // 4096 tuples = 64 KiB LDS staging
constexpr int PART_MAX_BITS = 10;                 // k_scan_units: nbins <= 1024 threads
constexpr u32 PART_TARGET_UNITS = 2048;           // ~8 units per CU

// ---- bucket join geometry: bucketized LDS table, 512 threads, two workgroups per CU ------------
constexpr int BJ_THREADS = 512;
constexpr int BJ_CHUNK = 4224;                    // build tuples per LDS table: 66 KiB keys+rowids
constexpr int BJ_BUCKET_BITS = 11;                // 2048 hash buckets (offsets: 8 KiB)
constexpr int BJ_EPT = 8;                         // probe tuples per thread per tile
constexpr int BJ_TILE = BJ_THREADS * BJ_EPT;      // 4096
constexpr int BJ_FIT = BJ_CHUNK * 15 / 16;        // plan: average build partition <= 3960 tuples
constexpr u32 BJ_MAX_PROBE_SPLIT = 1u << 24;      // probe tuples per join task at most (a caller's larger probe_split is clamped: same pairs, more tasks)
// under a plan of >= 16 bits, average build partitions of CT_GUARDED_FROM ... CT_GUARDED_UPTO tuples go to the compact-table
// kernel's 6144-entry geometry with row guards (k_join_ct<.., GUARD>): 4 ... 10 of its 12 slot rows in use
constexpr int CT_GUARDED_FROM = 2048, CT_GUARDED_UPTO = 5120;
// small joins run unpartitioned in one launch (k_join_bkt DIRECT): every 4096-tuple probe tile re-builds the table chunks
constexpr u64 DIRECT_MAX_BUILD = 12ull * BJ_CHUNK; // build side of at most 12 table chunks ...
constexpr u64 DIRECT_MAX_PROBE = 131072;          // ... probed by at most 32 workgroups
// ... with inputs already in HBM (rhj_join_dev) at most 5: a workgroup walks the chunks one after the other (20 us + 15 us
// per chunk), and one 4-6-bit pass + join is ~105 us whatever the size ([measured] 50K x 50K: 0.21 ms direct, 0.11 partitioned)
constexpr u64 DIRECT_MAX_BUILD_DEV = 5ull * BJ_CHUNK;

struct JoinTask {           // one workgroup's work: probe range [pbeg, pbeg+plen) against build range [bbeg, bbeg+blen)
    u64 pbeg;               // absolute index into the probe-side array
    u32 plen;
    u32 part;               // partition id
    u64 bbeg;               // absolute index into the build-side array
    u32 blen;
    u32 build_is_S;         // 1: build on S, probe with R (|R_k| >= |S_k|, JobScheduler.cpp:187)
};

struct PassGeom {
    u64 n;          // tuples in the relation
    u64 L;          // max tuples per unit
    u32 nseg;       // segments in this pass
    u32 max_units;  // grid size upper bound: floor(n / L) + nseg
    int shift;      // digit = (payload >> shift) & (nbins-1)
    int bits;
    int mix = 0;    // MIX_*: where the digit comes from when the input is the caller's 16-byte relation
};
// Inside a join the radix digits (and, multi-GPU, the owner classes) come from mix64(payload), a BIJECTIVE 64-bit mix
// (splitmix64's finaliser), not from the raw low payload bits: join values that are multiples of 2^16, or that differ in
// their high bits only, would otherwise all land in one partition (the reference does not degrade there: its bucket table
// hashes the whole value modulo a prime, Result.cpp:43-58).  The FIRST kernels that touch the caller's tuples apply it
// (MIX_STORE: histogram digit of the mixed value; the scatter writes the mixed value), every later kernel -- pass 2, the
// bucket joins -- works on the mixed value as if it were the payload: a bijection keeps equality, so the pair set does
// not change, and only rowIDs are reported.  The public stage calls (rhj_histogram / rhj_partition / rhj_partition_at /
// rhj_bucket_join) keep raw bits: their bucket order is documented.  MIX_DIGIT: digit of the mixed value, tuple written
// as it came (rhj_shard_split16: the multi-GPU owner split of 16-byte tuples, joined by rhj_join_dev on the receiver).
constexpr int MIX_NONE = 0, MIX_STORE = 1, MIX_DIGIT = 2;

// launchers (all asynchronous on `st`)
void launch_init_single_segment(hipStream_t st, u64 n, u64 L, u64 *d_seg_start, u32 *d_unit_start);
// DupSniff: which side of a join has duplicate join values -- asked of the data, by the histogram kernels that read every tuple
// anyway.  A tuple whose mix64(payload) has sel_bits leading zero bits is SAMPLED (a fixed subset of the VALUES, so every
// duplicate of a sampled value is sampled too; sel_bits such that 256-512 tuples of the relation are) and counted in one of
// SNIFF_SLOTS counters of its side by a hash of the value: fire-and-forget atomics, nothing waits for them.  The task planners
// (k_make_tasks, the planner workgroup of k_scatter_fused2) sum max(0, counter - 1) per side -- duplicates, plus the few
// chance meetings of two values in a slot, the same for both sides -- compare the two RATES, and let the side with fewer
// duplicates be the hash table when the sizes are near (build_on_S in rhj_kernels.hip).  The counters are zero when a join
// starts (the caller clears them, or the previous join's histogram launch did).  tab == nullptr: no sampling / the first relation.
constexpr u32 SNIFF_SLOTS = 16384, SNIFF_TARGET = 512;
struct DupSniff { u32 *tab = nullptr; int sel_bits = 0; };
struct SniffVerdict { const u32 *tab = nullptr; u32 expect_R = 0, expect_S = 0; };   // tab: [2 sides][SNIFF_SLOTS]; expected samples
inline int sniff_sel_bits(u64 n) { int s = 0; while ((n >> s) > SNIFF_TARGET) s++; return s; }
int build_tie_shift();                             // sizes within 1/2^this of each other are a tie (RHJ_BUILD_TIE, default 4: 1/16)
void launch_make_units(hipStream_t st, const u64 *d_seg_start, u32 nseg, u64 L, u32 *d_unit_start);
// d_minmax (may be null): two u64, atomicMin / atomicMax of the rowIDs seen
void launch_hist_units(hipStream_t st, const void *d_in, const PassGeom &g, const u64 *d_seg_start,
                       const u32 *d_unit_start, u32 *d_unit_hist, u64 *d_minmax = nullptr, const DupSniff &sniff = DupSniff());
void launch_scan_units(hipStream_t st, const PassGeom &g, const u64 *d_seg_start, const u32 *d_unit_start,
                       const u32 *d_unit_hist, u64 *d_unit_base, u64 *d_part_start, u64 *d_scan_tmp);
void launch_scatter_units(hipStream_t st, const void *d_in, void *d_out, const PassGeom &g,
                          const u64 *d_seg_start, const u32 *d_unit_start, const u64 *d_unit_base);
void launch_diff_hist(hipStream_t st, const u64 *d_start, u64 nbins, u64 *d_hist);
void launch_check_radix(hipStream_t st, const void *d_R, const u64 *d_startR, const void *d_S, const u64 *d_startS, u64 nparts,
                        int radix_bits, u64 *d_bad);
void launch_prefix(hipStream_t st, const u64 *d_hist, u64 nbins, u64 *d_start);
void launch_make_tasks(hipStream_t st, const u64 *d_startR, const u64 *d_startS, u64 nparts, u32 probe_split,
                       JoinTask *d_tasks, u32 *d_ntasks, u32 max_tasks, u64 *d_stats, int kind, const SniffVerdict &sniff = SniffVerdict());
void launch_join(hipStream_t st, const void *d_R, const u64 *d_startR, const void *d_S, const u64 *d_startS,
                 const JoinTask *d_tasks, const u32 *d_ntasks, u32 grid, int radix_bits,
                 void *d_out, u64 out_capacity, u64 *d_out_count, int kind, const u32 *d_RK = nullptr, const u32 *d_SK = nullptr,
                 const u64 *d_tag_base = nullptr, const u32 *d_skip = nullptr, u64 *host_pub = nullptr, u32 *d_done = nullptr);
void launch_join_direct(hipStream_t st, const void *d_R, u64 nR, const void *d_S, u64 nS, void *d_out, u64 out_capacity,
                        u64 *d_out_count, u64 *host_count = nullptr, u32 *d_done = nullptr, void *host_out = nullptr,
                        u64 host_cap = 0);
// one join of a batched direct launch (k_join_bkt<.., BATCH>): what launch_join_direct passes as kernel arguments, per join
struct BatchJoinDesc {
    const void *R, *S;          // 16-byte tuples in HBM
    void *out;                  // pairs in HBM
    u64 cap;                    // ... capacity (pairs)
    u64 *count;                 // device result counter of this join (zero before, zero after)
    u32 nblocks, nb, np, build_is_S, split, pad;
    u64 *host_count;            // pinned host: the count, published by the join's last workgroup
    u32 *done;                  // device ticket of this join (zero before, zero after)
    void *host_out;             // pinned host landing zone of this join's first host_cap pairs
    u64 host_cap;
};
void launch_join_batch(hipStream_t st, const BatchJoinDesc *d_batch, u32 njoins, u32 max_blocks);
u32 join_direct_tile();
void launch_checksum(hipStream_t st, const void *d_pairs, u64 n, u64 *d_sum);
void launch_generate(hipStream_t st, int kind, void *d_out, u64 n, u64 row0, u64 D, u64 seed, double theta);
void launch_expected_pkfk(hipStream_t st, const void *d_S, u64 n, u64 *d_sum);
void launch_remap_keys(hipStream_t st, void *d_rel, u64 n, int shift, u64 add);
// one partition pass over both relations of a join with shared launches (one-pass plans)
struct PassSide {
    const void *in;
    void *out;
    u64 *seg_start;
    u32 *unit_start;
    u32 *unit_hist;
    u64 *unit_base;
    u64 *part_start;
    u64 *scan_tmp;
    PassGeom g;
};
struct PassPairHost { PassSide side[2]; u64 *zero8 = nullptr; /* eight 64-bit words cleared by the first launch, or null */ int mix = 0; };
void launch_pass_pair(hipStream_t st, const PassPairHost &h, int shift, int bits, int phase);
// one-pass joins in three launches (see k_hist_fused2): phase 0 histograms, phase 1 scatter + boundaries + task list (same
// arguments to both); parity: which of the two copies of the control block this call uses (the caller alternates)
size_t fuse_ctl_bytes();
u32 *fuse_join_ticket(void *d_ctl);
// sniff: sample the join values for duplicates (DupSniff; the counters live in the control block)
void launch_fused_pass(hipStream_t st, const PassPairHost &h, int bits, int phase, int parity, void *d_ctl, u32 probe_split, u32 max_tasks,
                       u32 table_tuples, JoinTask *d_tasks, u64 *d_counters, u64 *host_pub, bool sniff = false);
constexpr int PASS_PAIR_MAX_BITS = 9;            // the write-combining scatter's range
bool fused_two_pass_ok(int b1, int b2);
// bucket-join kernels: JK_BKT partitions that fit one 4224-tuple table (two workgroups per CU); JK_BKT_BIG 8448-tuple
// chunks, probe side re-read per chunk (any radix plan); JK_CT compact 8-byte entries, both sides read once
// (plans that remove >= 16 payload bits)
enum JoinKernel { JK_BKT = 0, JK_BKT_BIG = 1, JK_CT = 2, JK_CT_HALF = 3, JK_CT_WIDE = 4, JK_CT_HALF_WIDE = 5, JK_CT_MID = 6, JK_CT_HALF_MID = 7, JK_CT_13 = 8,
                  JK_CT_HALF_MID_G = 9, JK_CT_G13 = 10, JK_CT_Q12 = 11, JK_LAST = JK_CT_Q12 };
inline bool jk_is_ct(int k) { return k >= JK_CT && k <= JK_LAST; }                           // a compact-table geometry (48-bit keys: needs >= 16 radix bits)
inline bool jk_ct_narrow_only(int k) { return k == JK_CT_WIDE || k == JK_CT_HALF_WIDE; }    // 20 probe slots: {payload, rowID} partitions only
// _WIDE: 20 probe slots per thread (narrow format only); _MID: a 12288-entry table and 12 slots per thread (partitions of 8.4 - 11.5 K);
// _HALF_MID: the same at half size, 6144 entries, 512 threads, two workgroups per CU (partitions of 4.2 - 5.8 K)
u32 join_probe_split(int kind);      // probe tuples per task the kernel holds at most (0: no limit of its own)
u32 join_table_tuples(int kind);     // build tuples per LDS table
int join_ct_min_radix_bits(int kind = JK_CT);   // 16: keys of 48 bits beside a 16-bit arrival index; JK_CT_G13: 13 (51 + 13 bits); JK_CT_Q12: 12
// in_narrow: d_in is a payload array (u64).  key_base / d_wide (16-byte input): d_wide (may be null) is OR-ed with 1 when some
// rowID - key_base does not fit 32 bits.  d_unit_rng (may be null): explicit pass-1 units (launch_seg_units).
void launch_hist2d_units(hipStream_t st, const void *d_in, bool in_narrow, u64 n, u64 L, u32 units, int b1, int b2,
                         u32 units_per_group, u32 ngroups, u32 *d_hist1, u32 *d_hist2, u64 key_base, u32 *d_wide,
                         const u64 *d_unit_rng, int mix = 0, const DupSniff &sniff = DupSniff());
void launch_seg_units(hipStream_t st, u32 nseg, const u64 *seg_off, const u64 *seg_L, u32 units_per_seg, u64 *d_unit_rng,
                      u64 *d_seg_start, u32 *d_unit_start);
int seg_max();                                             // segments (= ranks) a receiver can tell apart: 16
int tag_bits();                                            // low payload bits that carry the sender number at the receiver
void launch_make_group_ranges(hipStream_t st, const u64 *d_unit_base1, u32 nb1, u32 units_per_group, u32 ngroups, u64 n,
                              u64 *d_rng, u32 *d_unit_start2);
void launch_scatter_ranges(hipStream_t st, const void *d_in, void *d_out, u32 nunits, int shift, int bits,
                           const u64 *d_unit_base, const u64 *d_rng);
// narrow intermediate format (k_scatter_wcn): payloads (u64) at offset 0 of a buffer of >= 16 n bytes, rowIDs (u32) here
inline size_t narrow_k_offset(u64 n) { return ((size_t)n * 8 + 255) & ~(size_t)255; }
constexpr int RHJ_RETRY_WIDE = 1000;                    // internal: join_phase saw the narrow-format overflow flag
constexpr u64 NARROW_MIN_TUPLES = 1024;                 // 12 n + 256 <= 16 n
constexpr u64 NARROW_AUTO_MIN_TUPLES = 8000000;         // automatic choice: larger side at least this ([measured] 4M: 0.45 ms
                                                        // either way; 16M ... 256M: 5-8 % faster narrow; 10^9: 19 %)
bool narrow_pass_ok(int bits);                          // the 32-tuple-line geometry: <= 8 bits
bool narrow_pass9_ok(int bits);                         // ... or the 16-tuple-line geometry: <= 9 bits
void launch_hist_units_narrow(hipStream_t st, const void *d_inP, const PassGeom &g, const u64 *d_seg_start,
                              const u32 *d_unit_start, u32 *d_unit_hist);
void launch_scatter_units_narrow_any(hipStream_t st, const void *d_in, const u32 *d_inK, void *d_outP, u32 *d_outK, const PassGeom &g,
                                     const u64 *d_seg_start, const u32 *d_unit_start, const u64 *d_unit_base, u32 *d_overflow);
void launch_scatter_units_narrow(hipStream_t st, const void *d_in, void *d_out, u64 n, const PassGeom &g,
                                 const u64 *d_seg_start, const u32 *d_unit_start, const u64 *d_unit_base, u32 *d_overflow,
                                 u64 key_base = 0);
void launch_scatter_units_narrow_peer(hipStream_t st, const void *d_in, const PassGeom &g, const u64 *d_seg_start,
                                      const u32 *d_unit_start, const u64 *d_unit_base, u32 *d_overflow, u64 key_base,
                                      const u64 *d_delta, const unsigned char *d_owner, void *const *peersP, void *const *peersK,
                                      int nranks);
void launch_scatter_ranges_narrow(hipStream_t st, const void *d_in, bool in_narrow, void *d_out, u64 n, u32 nunits, int shift,
                                  int bits, const u64 *d_unit_base, const u64 *d_rng, u32 *d_overflow, u32 tag_groups = 0,
                                  u32 tag_div = 0, const u32 *d_inK = nullptr);   // d_inK: narrow input whose rowID array is not at narrow_k_offset(n)
const char *launch_attr_error();                       // text of the first refused hipFuncSetAttribute, or null
void launch_scatter_ranges_n2a(hipStream_t st, const void *d_in, void *d_out, u64 n, u32 nunits, int shift, int bits,
                               const u64 *d_unit_base, const u64 *d_rng, const u64 *d_key_bases, u32 tag_groups, u32 tag_div,
                               const u32 *d_skip);
size_t scan_tmp_bytes(int bits);
size_t part_lds_bytes(int bits);
