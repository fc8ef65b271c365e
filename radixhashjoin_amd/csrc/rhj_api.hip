// rhj_api.hip -- the C-ABI of include/rhj.h: context, HBM workspace, radix plan, orchestration.
// Host orchestration replaces the JobScheduler dispatch of the join path: what the reference does
// with 5 barriers and <= 288 heap Job objects per join is a fixed sequence of
// asynchronous kernel launches on one HIP stream with a single host sync for the result count.
#include "../../include/rhj.h"
#include "rhj_internal.h"

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <sys/mman.h>
#include <thread>
#include <vector>

namespace {

thread_local std::string g_last_error;

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
};

struct Prof {
    bool on = false;
    bool keep = false;                     // rhj_set_profiling(ctx, 2): the spans of successive calls accumulate
    std::vector<hipEvent_t> pool;          // event pairs, reused
    std::vector<int> kinds;                // kind per recorded pair
    size_t used = 0;
};

}  // namespace

// Host -> HBM copies of PAGEABLE caller memory (the reference allocates relations with new[]).  [measured, one MI355X host]
// hipMemcpy from pageable memory 15-18 GB/s, from pinned memory 57.5 GB/s; hipHostRegister of the caller's array 60 ms per GiB
// (more than the copy itself); memcpy pageable -> pinned 24-30 GB/s per thread.  So: STAGE_BUFS pinned 16 MiB buffers filled
// by STAGE_WORKERS persistent threads (whole chunks each, > 100 GB/s together), DMA'd out in order by the calling thread on the
// copy stream: the wire runs at the pinned rate.
constexpr size_t STAGE_BYTES = (size_t)16 << 20;
constexpr int STAGE_BUFS = 8, STAGE_WORKERS = 6;

struct Stager {
    void *buf[STAGE_BUFS] = {nullptr};
    hipEvent_t ev[STAGE_BUFS] = {nullptr};
    std::vector<std::thread> workers;
    std::mutex mu;
    std::condition_variable cv;
    bool quit = false;
    u64 job_id = 0;                        // bumped for every copy; workers pick the job up once
    int device = 0;
    // the current job
    const char *src = nullptr;
    size_t bytes = 0, nchunks = 0;
    std::atomic<size_t> next{0}, issued{0}, chunk0{0};
    std::atomic<int> active{0};
    std::vector<std::atomic<unsigned char>> filled;
    size_t total_chunks = 0;               // chunks DMA'd since the buffers were created (buffer = chunk number % STAGE_BUFS)

    void work()
    {
        (void)hipSetDevice(device);
        u64 seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return quit || job_id != seen; });
                if (quit) return;
                seen = job_id;
            }
            for (;;) {
                const size_t c = next.fetch_add(1);
                if (c >= nchunks) break;
                const size_t g = chunk0.load() + c;                      // global chunk number: buffer g % STAGE_BUFS
                if (g >= (size_t)STAGE_BUFS) {                            // the buffer's previous chunk must be on the device
                    while (issued.load(std::memory_order_acquire) + STAGE_BUFS <= g) std::this_thread::yield();
                    (void)hipEventSynchronize(ev[g % STAGE_BUFS]);
                }
                const size_t off = c * STAGE_BYTES, len = bytes - off < STAGE_BYTES ? bytes - off : STAGE_BYTES;
                memcpy(buf[g % STAGE_BUFS], src + off, len);
                filled[c].store(1, std::memory_order_release);
            }
            active.fetch_sub(1);
        }
    }
};

// A few helper threads that copy a list of memory segments side by side (rhj_join_batch: staging the inputs of sixteen small
// joins into pinned memory, filling sixteen result pages).  One job at a time; the caller copies too.
struct CopySeg { void *dst; const void *src; size_t bytes; };
struct CopyPool {
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv, idle;
    const CopySeg *segs = nullptr;
    size_t nseg = 0;
    std::atomic<size_t> next{0};
    u64 gen = 0;
    int active = 0;
    bool quit = false;
    explicit CopyPool(int n)
    {
        try {
            for (int i = 0; i < n; i++) th.emplace_back([this] { work(); });
        } catch (...) {}                                   // fewer helpers (or none): the caller copies everything itself
    }
    ~CopyPool()
    {
        { std::lock_guard<std::mutex> lk(mu); quit = true; }
        cv.notify_all();
        for (std::thread &t : th) t.join();
    }
    void drain()
    {
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= nseg) return;
            if (segs[i].src) memcpy(segs[i].dst, segs[i].src, segs[i].bytes);
            else memset(segs[i].dst, 0, segs[i].bytes);              // (no source: first touch of fresh pages)
        }
    }
    void work()
    {
        u64 seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return quit || gen != seen; });
                if (quit) return;
                seen = gen;
            }
            drain();
            {
                std::lock_guard<std::mutex> lk(mu);
                active--;
            }
            idle.notify_all();
        }
    }
    void run(const std::vector<CopySeg> &list)
    {
        if (list.empty()) return;
        {
            std::lock_guard<std::mutex> lk(mu);
            segs = list.data();
            nseg = list.size();
            next.store(0);
            active = (int)th.size();
            gen++;
        }
        cv.notify_all();
        drain();
        std::unique_lock<std::mutex> lk(mu);
        idle.wait(lk, [&] { return active == 0; });
    }
};

struct rhj_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    std::string err;
    Prof prof;
    rhj_timings last{};
    // workspace (grow-only)
    DevBuf in_R, in_S;                 // H2D staging of host inputs (rhj_join)
    DevBuf part_R, part_S, part_tmp;   // partitioned relations, pass-1 intermediate
    DevBuf ps_R, ps_S, ps_1;           // partition boundaries (final R, final S, pass-1 scratch)
    DevBuf seg0, unit_start, unit_hist, unit_base;
    DevBuf seg0_b, unit_start_b, unit_hist_b, unit_base_b, scan_tmp_b;   // second relation of a paired pass
    DevBuf hist2_b, grp_rng_b, unit_start2_b, ps_1_b, part_tmp_b;        // ... and of a two-pass join partitioned on two streams
    hipStream_t aux_stream = nullptr;  // S of a mid-size two-pass join is partitioned here while R runs on `stream`
    hipEvent_t aux_ev[2] = {nullptr, nullptr};
    DevBuf tasks, counters;            // counters: [0] u64 out_count, [1] u32 ntasks (+pad), [2] u64 checksum
    DevBuf out_pairs;                  // rhj_join's device result buffer
    DevBuf small_out;                  // small-join path: 64-byte header {count} + pairs, fetched in one D2H
    bool small_hdr_clean = false;      // the header has been zeroed behind the previous small join
    DevBuf hist_tmp, scan_tmp, hist2, grp_rng, unit_start2, seg_rng, tag_base;
    DevBuf shard_ps[2], shard_mm;      // multi-GPU sender: class boundaries per side, rowID {min, max} per side
    // rhj_join_batch: several small joins per launch
    struct CopyPool *pool = nullptr;   // helper threads for host memcpy (staging in, pages out)
    // (two slots: the host work of one group of joins overlaps the GPU work of the next)
    unsigned char *b_stage[2] = {nullptr, nullptr};  // pinned: descriptors + the inputs of one group of joins
    size_t b_stage_cap[2] = {0, 0};
    unsigned char *b_land[2] = {nullptr, nullptr}, *b_land_dev[2] = {nullptr, nullptr};   // pinned: 16 counts (128 B) + the group's first pairs, written by the kernel itself
    size_t b_land_cap[2] = {0, 0};
    DevBuf b_in[2], b_out[2], b_cnt[2];   // device: staged blob, pair buffers, per-join {count, ticket}
    hipEvent_t b_ev[2] = {nullptr, nullptr};
    DevBuf fuse_ctl;                   // one-pass joins in three launches: two copies of {global histograms, digit cursors, sample counters} + the join's ticket
    bool fuse_clean = false;           // ... whose copy for the next call the kernels leave zeroed (false: the next call clears both first)
    int fuse_parity = 0;               // the copy the next call uses
    // which side of a join has duplicate join values (DupSniff, rhj_internal.h): the sample counters of R and S for a two-pass join
    // (a one-pass join keeps them in its control block)
    DevBuf sniff_tab;
    int sniff_side = -1;               // partition_relation_fused: the side it is counting for (-1: no sampling)
    u64 sniff_n[2] = {0, 0};
    bool sniff_ready = false;          // both sides of the current join were sampled: k_make_tasks may ask
    bool shard_sniffed[2] = {false, false};   // rhj_shard_partition sampled this side's received join values
    u64 *h_pub = nullptr, *h_pub_dev = nullptr;   // pinned: the join counters as the bucket join's last workgroup publishes them
    int opt_fused = -1;                // -1: automatic (RHJ_FUSE env, default 1), 0 / 1
    int opt_sniff = -1;                // -1: automatic (RHJ_SNIFF env, default 1), 0 / 1: sample the join values for duplicates (DupSniff)
    std::vector<u64> shard_ps_host[2]; // the class boundaries of the last rhj_shard_stats of each side (host copy)
    DevBuf shard_peer_tab;             // rhj_shard_split_peer: delta[2^bits] u64 + owner[2^bits] u8 per side
    DevBuf shard_wide;                 // u32 per side: rhj_shard_split met a rowID - key_base >= 2^32 (checked by rhj_shard_join)
    // rhj_dev_alloc / rhj_dev_free keep released blocks for re-use (all work of a context is ordered on its one
    // stream, so a block may be handed out again while kernels that used it are still queued): a device-resident
    // query allocates and frees a dozen arrays per join, and hipMalloc/hipFree would synchronise every time
    std::vector<std::pair<void *, size_t>> free_blocks;
    size_t free_bytes = 0;
    // pinned staging for host -> HBM copies of pageable caller memory (rhj_join)
    Stager *stager = nullptr;
    unsigned char *h_counts = nullptr;     // pinned: the join counters of every S chunk of a pipelined rhj_join (64 B each)
    std::vector<hipEvent_t> chunk_ev;
    hipStream_t down_stream = nullptr;     // result chunks travel home while later S chunks are still on their way in
    hipStream_t copy_stream = nullptr;     // uploads of rhj_join: S travels while R is being partitioned
    hipEvent_t up_ev[2] = {nullptr, nullptr};
    // state of the last partition phase (consumed by join_phase)
    const void *cur_R = nullptr, *cur_S = nullptr;
    const u64 *cur_psR = nullptr, *cur_psS = nullptr;
    u64 cur_nparts = 0, cur_nR = 0, cur_nS = 0;
    int cur_radix_bits = 0;
    u32 cur_probe_split = 0;
    int last_join_kind = -1;
    int last_pipelined = 0;            // S chunks of the last rhj_join (0: not pipelined)
    u64 last_max_part[2] = {0, 0};     // largest partition of R / S the last task list saw (0: direct join)
    bool counters_clean = false;       // the 64-byte join counters are zero (cleared by the partition phase's first launch)
    int cur_narrow = 0;                // partitions are in the narrow {payload, rowID} format (k_scatter_wcn); 2: so was the intermediate
    DevBuf narrow_flag;                // u32: a rowID >= 2^32 met a narrow scatter -> the join re-runs in the 16-byte format
    // ... the next join tries the narrow format again; consecutive fall-backs make the context skip the attempt for the next
    // 2, 4, ... 32 eligible joins (a caller whose rowIDs are always wide pays one extra histogram per relation now and then)
    u32 narrow_fail_streak = 0, narrow_skip = 0;
    bool narrow_off_once = false;      // the repeat of a join that fell back
    // multi-GPU receiver (rhj_shard_partition / rhj_shard_join): the partitions carry sender tags
    int shard_nseg = 0, shard_mode[2] = {0, 0};
    bool shard_side_done[2] = {false, false};
    u64 shard_n[2] = {0, 0}, shard_kmin[2] = {0, 0}, shard_kmax[2] = {0, 0};
    rhj_opts shard_plan{};
    // pinned host landing zone of the small-join path: 64-byte header {count}, then up to 128 KiB of result pairs
    unsigned char *h_land = nullptr;
    unsigned char *h_land_dev = nullptr;   // the same memory as the device addresses it
    // tuning / test knobs (rhj_set_option)
    int opt_big_tables = -1;           // -1: by average build partition size, 0: never, 1: always use an oversized-partition kernel
    int opt_big_kernel = -1;           // -1: automatic, JK_BKT_BIG: never the compact-table kernel
    int opt_narrow = -1;               // -1: automatic (RHJ_NARROW env: 0, 1 = last pass only, 2), 0: never, 1 / 2: that level
    int opt_mix = -1;                  // -1: automatic (RHJ_MIX env, default 1), 0: radix digits from the raw low payload bits (rounds 1-3),
                                       // 1: from mix64(payload) (MIX_STORE, rhj_internal.h)
};

namespace {

int fail(rhj_ctx *ctx, int code, const std::string &msg)
{
    g_last_error = msg;
    if (ctx) ctx->err = msg;
    return code;
}

#define HIPCHK(ctx, call)                                                                          \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(ctx, e_ == hipErrorOutOfMemory ? RHJ_E_NOMEM : RHJ_E_HIP,                  \
                        std::string(#call) + ": " + hipGetErrorString(e_));                        \
    } while (0)

#define RHJCHK(call)                                                                               \
    do {                                                                                           \
        int r_ = (call);                                                                           \
        if (r_ != RHJ_OK) return r_;                                                               \
    } while (0)

int ensure(rhj_ctx *ctx, DevBuf &b, size_t bytes)
{
    if (bytes <= b.cap) return RHJ_OK;
    if (b.p) { HIPCHK(ctx, hipFree(b.p)); b.p = nullptr; b.cap = 0; }
    // round up to 2 MiB so that repeated slightly-larger requests do not re-allocate every time
    size_t want = (bytes + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) {
        b.p = nullptr;
        (void)hipGetLastError();
        return fail(ctx, RHJ_E_NOMEM, "hipMalloc(" + std::to_string(want) + " B): " + hipGetErrorString(e));
    }
    b.cap = want;
    return RHJ_OK;
}

void release(DevBuf &b)
{
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
}

void stager_free(Stager *st)
{
    {
        std::lock_guard<std::mutex> lk(st->mu);
        st->quit = true;
    }
    st->cv.notify_all();
    for (std::thread &t : st->workers) t.join();
    for (int i = 0; i < STAGE_BUFS; i++) {
        if (st->buf[i]) (void)hipHostFree(st->buf[i]);
        if (st->ev[i]) (void)hipEventDestroy(st->ev[i]);
    }
    delete st;
}

void stager_destroy(rhj_ctx *ctx)
{
    if (ctx->stager) stager_free(ctx->stager);
    ctx->stager = nullptr;
}

// Built completely before the context sees it: a stager with fewer than STAGE_WORKERS workers would leave h2d_staged
// waiting for chunks nobody fills.  Null when a pinned buffer, an event or a thread cannot be had (the caller then copies
// straight from the pageable source).
Stager *stager_create(int device)
{
    Stager *st = new (std::nothrow) Stager();
    if (!st) return nullptr;
    st->device = device;
    bool ok = true;
    for (int i = 0; i < STAGE_BUFS && ok; i++)
        ok = hipHostMalloc(&st->buf[i], STAGE_BYTES, hipHostMallocDefault) == hipSuccess &&
             hipEventCreateWithFlags(&st->ev[i], hipEventDisableTiming) == hipSuccess;
    try {
        for (int i = 0; i < STAGE_WORKERS && ok; i++) st->workers.emplace_back([st] { st->work(); });
    } catch (...) {                                        // std::system_error: no more threads -- never through the C ABI
        ok = false;
    }
    if (!ok) {
        (void)hipGetLastError();
        stager_free(st);
        return nullptr;
    }
    return st;
}

// asynchronous with respect to the device: returns when the last chunk's DMA has been ENQUEUED on `stream` (the source has
// been read completely by then)
int h2d_staged(rhj_ctx *ctx, void *d_dst, const void *src, size_t bytes, hipStream_t stream)
{
    if (bytes < 4 * STAGE_BYTES) {
        HIPCHK(ctx, hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, stream));
        return RHJ_OK;
    }
    if (!ctx->stager) ctx->stager = stager_create(ctx->device);
    if (!ctx->stager) {                                                   // no pinned ring to be had: the plain (slower) copy
        HIPCHK(ctx, hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, stream));
        return RHJ_OK;
    }
    Stager *st = ctx->stager;
    const size_t nchunks = (bytes + STAGE_BYTES - 1) / STAGE_BYTES;
    {
        std::lock_guard<std::mutex> lk(st->mu);
        st->src = (const char *)src;
        st->bytes = bytes;
        st->nchunks = nchunks;
        st->filled = std::vector<std::atomic<unsigned char>>(nchunks);
        for (auto &f : st->filled) f.store(0);
        st->next.store(0);
        st->chunk0.store(st->total_chunks);
        st->issued.store(st->total_chunks);
        st->active.store(STAGE_WORKERS);
        st->job_id++;
    }
    st->cv.notify_all();
    int rc = RHJ_OK;
    for (size_t c = 0; c < nchunks; c++) {
        while (!st->filled[c].load(std::memory_order_acquire)) std::this_thread::yield();
        const size_t g = st->total_chunks + c, off = c * STAGE_BYTES, len = bytes - off < STAGE_BYTES ? bytes - off : STAGE_BYTES;
        hipError_t e = hipMemcpyAsync((char *)d_dst + off, st->buf[g % STAGE_BUFS], len, hipMemcpyHostToDevice, stream);
        if (e == hipSuccess) e = hipEventRecord(st->ev[g % STAGE_BUFS], stream);
        if (e != hipSuccess && rc == RHJ_OK) rc = fail(ctx, RHJ_E_HIP, std::string("staged upload: ") + hipGetErrorString(e));
        st->issued.store(g + 1, std::memory_order_release);              // (on an error too: the workers must not wait for ever)
    }
    while (st->active.load() != 0) std::this_thread::yield();             // every worker has left the job
    st->total_chunks += nchunks;
    return rc;
}

int use_device(rhj_ctx *ctx)
{
    if (!ctx) return fail(nullptr, RHJ_E_INVALID, "null context");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    return RHJ_OK;
}

// ---- profiling: one event pair per launch, resolved lazily -------------------------------------
struct Span {
    rhj_ctx *c;
    Span(rhj_ctx *ctx, int kind) : c(ctx)
    {
        if (!c->prof.on) return;
        Prof &p = c->prof;
        if (p.used * 2 + 2 > p.pool.size()) {
            hipEvent_t a, b;
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { p.on = false; return; }
            p.pool.push_back(a);
            p.pool.push_back(b);
        }
        p.kinds.push_back(kind);
        (void)hipEventRecord(p.pool[p.used * 2], c->stream);
    }
    ~Span()
    {
        if (!c->prof.on) return;
        Prof &p = c->prof;
        (void)hipEventRecord(p.pool[p.used * 2 + 1], c->stream);
        p.used++;
    }
};

void prof_reset(rhj_ctx *ctx)
{
    if (!ctx->prof.keep) {
        ctx->prof.used = 0;
        ctx->prof.kinds.clear();
    }
    memset(&ctx->last, 0, sizeof(ctx->last));
}

int check_launch(rhj_ctx *ctx, const char *what)
{
    hipError_t e = hipGetLastError();
    if (const char *attr = launch_attr_error())          // a kernel was refused its LDS size: say so instead of "invalid argument"
        return fail(ctx, RHJ_E_HIP, std::string(what) + ": " + attr);
    if (e != hipSuccess) return fail(ctx, RHJ_E_HIP, std::string(what) + " launch: " + hipGetErrorString(e));
    return RHJ_OK;
}

// tuning environment variables: unset, empty or non-numeric -> the default; numeric values are clamped to [lo, hi]
u64 env_u64(const char *name, u64 dflt, u64 lo, u64 hi)
{
    const char *v = getenv(name);
    if (!v || !*v) return dflt;
    char *end = nullptr;
    const u64 x = strtoull(v, &end, 10);
    if (end == v) return dflt;
    return x < lo ? lo : x > hi ? hi : x;
}

// whether joins sample their join values for duplicates (DupSniff; RHJ_SNIFF=0: no, the first relation wins every near tie)
bool sniff_on(const rhj_ctx *ctx)
{
    static const bool on = env_u64("RHJ_SNIFF", 1, 0, 1) != 0;
    return ctx->opt_sniff >= 0 ? ctx->opt_sniff != 0 : on;
}

int ilog2_ceil(u64 x)
{
    int b = 0;
    while (((u64)1 << b) < x) b++;
    return b;
}

// ---- radix plan (host logic) --------------------------------------------------------------------
// Reference: one fixed 8-bit pass (Result.cpp:5,91).  Here: the fewest radix bits such that the
// average build-side partition fills at most 15/16 of one LDS hash table (BJ_CHUNK), in <= 2 passes.
// device_resident: the inputs are in HBM already (rhj_join_dev); else the host-pointer call, whose small path is one launch
// largest average partition (tuples, either side) the 6144-entry kernel with 13-bit arrival indices is chosen for
u64 g13_upto()
{
    // 15/16 of its table: the row guards stop paying beyond CT_GUARDED_UPTO, but half as many partitions still do ([measured, wall
    // ms, 5120 -> 5760] 43M 1.852 -> 1.760 . 46M 1.894 -> 1.710 . 86M 3.364 -> 3.164 . 93M 3.459 -> 3.363 . 170M 6.732 -> 6.097 . 185M 7.082 -> 6.297)
    static const u64 v = env_u64("RHJ_G13_UPTO", (u64)join_table_tuples(JK_CT_G13) * 15 / 16, 2048, 5760);      // tuning aid
    return v;
}

bool default_join_kernels(const rhj_ctx *ctx) { return ctx->opt_big_kernel < 0 && ctx->opt_big_tables < 0; }

// ct13_ok: the join may pick the 6144-entry compact-table kernel for plans of 13-15 bits (default kernel choice, not the
// multi-GPU receiver): the plan then leaves partitions of up to CT_GUARDED_UPTO tuples
int resolve_plan(u64 nR, u64 nS, const rhj_opts *in, rhj_opts *out, bool device_resident = false, bool ct13_ok = true)
{
    rhj_opts o;
    if (in) o = *in; else rhj_default_opts(&o);
    const u64 nb = nR < nS ? nR : nS;
    const u64 fit = (u64)BJ_FIT;
    if (o.passes < -1 || o.passes > 2 || o.bits1 < 0 || o.bits2 < 0 || o.bits1 > PART_MAX_BITS ||
        o.bits2 > PART_MAX_BITS || o.probe_split < 0)
        return RHJ_E_INVALID;
    if (o.passes == -1) {
        if (o.bits1 > 0) o.passes = o.bits2 > 0 ? 2 : 1;
        else {
            int bits = 0;
            const u64 np_ = nR < nS ? nS : nR;
            // small joins stay unpartitioned (one launch, every probe tile re-builds the few table chunks): a partition
            // pass costs ~6 launches per relation, more than such a join itself
            const bool direct = nb <= (device_resident ? DIRECT_MAX_BUILD_DEV : DIRECT_MAX_BUILD) && np_ <= DIRECT_MAX_PROBE;
            if (nb > (u64)BJ_CHUNK && !direct) bits = ilog2_ceil((nb + fit - 1) / fit);
            if (bits == 0) o.passes = 0;
            else if (bits <= 9) { o.passes = 1; o.bits1 = bits; }
            // Up to two and a half 8448-tuple chunks per partition after ONE 9-bit pass (three launches; the chunked
            // 16-byte-entry kernel joins): a two-pass plan costs ~27 launches, 0.2 ms of fixed latency that such a join does
            // not have to spare.  [measured, round 4, wall ms one pass / two] 8.5M 0.43 / 0.50 . 10M 0.51 / 0.54 . 12M 0.61 / 0.59
            else if (nb <= (u64)512 * 5 * join_table_tuples(JK_BKT_BIG) * 15 / 32) { o.passes = 1; o.bits1 = 9; }
            else {
                // Two passes.  Up to 16 bits both histograms come from ONE read of the input (k_hist2d_units); a 17- or
                // 18-bit plan re-reads each relation once more just to count (10.7 instead of 5.4 ms per 10^9-tuple join).
                // Partitions that 16 bits leave larger than one 16-byte-entry table go to the compact-table bucket join
                // (one 17920-entry table, both sides read once), which costs less than that extra read: measured at
                // 10^9 x 10^9, 8+8 bits 43 ms against 49 ms for 9+9.  Beyond 1.1 * 10^9 tuples per side a 16-bit partition
                // no longer fits ONE compact table (chunks: probe payloads streamed again per chunk) and 9+9 bits (the
                // widest line-aligned write-combining scatter) takes over: 2.2 * 10^9 x 2.2 * 10^9, 121.6 against 125.3 ms.
                // 17 bits (9+8) while the average partition stays inside one compact table (its probe side inside one task of 16 or
                // 20 slots per thread), 18 (9+9) beyond; both run in the narrow format with 16-tuple carry lines in their 9-bit passes (k_scatter_wcn
                // <GR = 16>) and a second, 8 B/tuple histogram read of the narrow intermediate.
                if (bits > 16) bits = nb <= (u64)65536 * 16800 ? 16 : nb <= (u64)131072 * 16800 ? 17 : 18;
                // One bit fewer where that leaves partitions for the 6144-entry compact-table kernel (k_join_ct<.., KB = 13>, plans
                // of 13-15 bits, average partitions of up to 15/16 of its 6144 entries on both sides): half as many partitions,
                // tasks and histogram rows.  [measured, one box, wall ms, bits as above -> one fewer] 34M 1.538 -> 1.431 . 40M 1.722 ->
                // 1.579 . 66M 2.814 -> 2.538 . 83M 3.397 -> 2.971 . 135M 5.692 -> 5.091 . 165M 6.729 -> 6.084
                static const bool ct13_env = env_u64("RHJ_CT13", 1, 0, 1) != 0;
                if (ct13_ok && ct13_env && bits - 1 >= join_ct_min_radix_bits(JK_CT_G13) && bits - 1 < join_ct_min_radix_bits(JK_CT) &&
                    (nb >> (bits - 1)) <= g13_upto() && (np_ >> (bits - 1)) <= g13_upto())
                    bits -= 1;
                o.passes = 2; o.bits1 = (bits + 1) / 2; o.bits2 = bits / 2;
                // 17 bits: the 9-bit pass second -- from the narrow intermediate it costs 5.3 ms per 10^9 tuples, from 16-byte
                // tuples 7.9 ([measured] 9+8 against 8+9 at 1.5 * 10^9: 68.4 against 65.2 ms)
                if (bits == 17) { o.bits1 = 8; o.bits2 = 9; }
            }
        }
    }
    if (o.passes == 0) { o.bits1 = 0; o.bits2 = 0; }
    if (o.passes == 1) { if (o.bits1 == 0) o.bits1 = 8; o.bits2 = 0; }
    if (o.passes == 2) { if (o.bits1 == 0) o.bits1 = 8; if (o.bits2 == 0) o.bits2 = 8; }
    if (o.probe_split == 0) {
        // enough tasks to fill 256 CUs even when nothing is partitioned, at most 32 Ki probe tuples each
        const u64 np = nR > nS ? nR : nS;
        u64 ps = (np + 1023) / 1024;
        ps = (ps + BJ_TILE - 1) / BJ_TILE * BJ_TILE;
        if (ps < (u64)BJ_TILE) ps = BJ_TILE;
        if (ps > 32768) ps = 32768;
        o.probe_split = (int32_t)ps;
    }
    *out = o;
    return RHJ_OK;
}

// whether the joins of this context take their radix digits from mix64(payload) (MIX_STORE) or from the raw payload
int join_mix(const rhj_ctx *ctx)
{
    static const int env = (int)env_u64("RHJ_MIX", 1, 0, 1);
    return (ctx->opt_mix >= 0 ? ctx->opt_mix : env) ? MIX_STORE : MIX_NONE;
}

PassGeom make_geom(u64 n, u32 nseg, int shift, int bits, u64 target_units = PART_TARGET_UNITS)
{
    PassGeom g;
    g.n = n;
    static const u64 forced_units = env_u64("RHJ_UNITS", 0, 1, 65536);                                   // tuning aid
    if (forced_units) target_units = forced_units;
    u64 L = (n + target_units - 1) / target_units;
    L = (L + PART_TILE - 1) / PART_TILE * PART_TILE;
    if (L < (u64)PART_TILE) L = PART_TILE;
    g.L = L;
    g.nseg = nseg;
    g.max_units = (u32)(n / L) + nseg;
    g.shift = shift;
    g.bits = bits;
    return g;
}

// One partition pass over all segments: histogram -> scan -> scatter.
// `whole_relation`: the pass has ONE segment [0,n) and d_seg_start is this context's seg0 buffer, which the pass then
// fills together with the unit table in one launch.
int run_pass(rhj_ctx *ctx, const void *d_in, void *d_out, u64 n, const u64 *d_seg_start, u32 nseg, int shift,
             int bits, u64 *d_part_start, bool whole_relation = false, int mix = MIX_NONE)
{
    PassGeom g = make_geom(n, nseg, shift, bits);
    g.mix = mix;
    const size_t nbins = (size_t)1 << bits;
    RHJCHK(ensure(ctx, ctx->unit_start, ((size_t)nseg + 1) * 4));
    RHJCHK(ensure(ctx, ctx->unit_hist, (size_t)g.max_units * nbins * 4));
    RHJCHK(ensure(ctx, ctx->unit_base, (size_t)g.max_units * nbins * 8));
    RHJCHK(ensure(ctx, ctx->scan_tmp, scan_tmp_bytes(bits)));
    u32 *unit_start = (u32 *)ctx->unit_start.p;
    {
        Span s(ctx, RHJ_K_AUX);
        if (whole_relation) launch_init_single_segment(ctx->stream, n, g.L, (u64 *)ctx->seg0.p, unit_start);
        else launch_make_units(ctx->stream, d_seg_start, nseg, g.L, unit_start);
    }
    {
        Span s(ctx, RHJ_K_HIST);
        launch_hist_units(ctx->stream, d_in, g, d_seg_start, unit_start, (u32 *)ctx->unit_hist.p);
    }
    {
        Span s(ctx, RHJ_K_SCAN);
        launch_scan_units(ctx->stream, g, d_seg_start, unit_start, (const u32 *)ctx->unit_hist.p,
                          (u64 *)ctx->unit_base.p, d_part_start, (u64 *)ctx->scan_tmp.p);
    }
    {
        Span s(ctx, RHJ_K_SCATTER);
        launch_scatter_units(ctx->stream, d_in, d_out, g, d_seg_start, unit_start, (const u64 *)ctx->unit_base.p);
    }
    return check_launch(ctx, "partition pass");
}

// One pass over BOTH relations with shared launches (grid.y = relation): 6 launches instead of 12.  Mid-size joins
// (BASELINE config 2: 1M x 1M, one 8-bit pass) are bound by launch count and by 244-workgroup kernels that fill the
// chip only halfway.
int run_pass_pair(rhj_ctx *ctx, const void *d_R, u64 nR, void *outR, u64 *psR, const void *d_S, u64 nS, void *outS, u64 *psS,
                  int bits)
{
    const size_t nbins = (size_t)1 << bits;
    PassPairHost h;
    DevBuf *seg[2] = {&ctx->seg0, &ctx->seg0_b}, *ust[2] = {&ctx->unit_start, &ctx->unit_start_b},
           *uh[2] = {&ctx->unit_hist, &ctx->unit_hist_b}, *ub[2] = {&ctx->unit_base, &ctx->unit_base_b},
           *sc[2] = {&ctx->scan_tmp, &ctx->scan_tmp_b};
    const void *in[2] = {d_R, d_S};
    void *out[2] = {outR, outS};
    u64 *ps[2] = {psR, psS};
    const u64 n[2] = {nR, nS};
    for (int i = 0; i < 2; i++) {
        const PassGeom g = make_geom(n[i], 1, 0, bits);
        RHJCHK(ensure(ctx, *seg[i], 64));
        RHJCHK(ensure(ctx, *ust[i], 16));
        RHJCHK(ensure(ctx, *uh[i], (size_t)g.max_units * nbins * 4));
        RHJCHK(ensure(ctx, *ub[i], (size_t)g.max_units * nbins * 8));
        RHJCHK(ensure(ctx, *sc[i], scan_tmp_bytes(bits)));
        h.side[i] = PassSide{in[i], out[i], (u64 *)seg[i]->p, (u32 *)ust[i]->p, (u32 *)uh[i]->p, (u64 *)ub[i]->p, ps[i],
                             (u64 *)sc[i]->p, g};
    }
    // the first launch also clears the join counters (one memset less in front of a join that is ~10 launches in all)
    RHJCHK(ensure(ctx, ctx->counters, 64));
    h.zero8 = (u64 *)ctx->counters.p;
    h.mix = join_mix(ctx);                               // (only joins come through here)
    ctx->counters_clean = true;
    static const int kinds[4] = {RHJ_K_AUX, RHJ_K_HIST, RHJ_K_SCAN, RHJ_K_SCATTER};
    for (int phase = 0; phase < 4; phase++) {
        Span s(ctx, kinds[phase]);
        launch_pass_pair(ctx->stream, h, 0, bits, phase);
    }
    return check_launch(ctx, "paired partition pass");
}

// Two passes with ONE histogram read (k_hist2d_units): used when both passes fit the write-combining scatter
// and b1 + b2 <= 16.  Pass-2 units = pieces of each pass-1 bucket written by groups of pass-1 units.
// narrow: 0 = 16-byte tuples throughout; 1 = pass 2 writes the narrow format (what the join kernel then reads); 2 = pass 1 too
// Input: 16-byte tuples (in.aos), or -- the multi-GPU receiver -- narrow arrays that arrived in nseg sender segments
// (in.P / in.K / in.seg_off; narrow is then 2): pass-1 units are cut at the segment boundaries and pass 2 writes the sender
// number into the low payload bits (k_scatter_wcn's WnTag), which the join kernels resolve into global rowIDs.
// the scratch tables one relation's fused two-pass partition works in, and the stream it runs on
struct PartScratch {
    DevBuf *seg0, *unit_start, *unit_hist, *unit_base, *scan_tmp, *hist2, *grp_rng, *unit_start2, *ps_1, *part_tmp, *seg_rng;
    hipStream_t st;
};
PartScratch first_scratch(rhj_ctx *ctx)
{
    return PartScratch{&ctx->seg0, &ctx->unit_start, &ctx->unit_hist, &ctx->unit_base, &ctx->scan_tmp, &ctx->hist2, &ctx->grp_rng,
                       &ctx->unit_start2, &ctx->ps_1, &ctx->part_tmp, &ctx->seg_rng, ctx->stream};
}
PartScratch second_scratch(rhj_ctx *ctx)
{
    return PartScratch{&ctx->seg0_b, &ctx->unit_start_b, &ctx->unit_hist_b, &ctx->unit_base_b, &ctx->scan_tmp_b, &ctx->hist2_b, &ctx->grp_rng_b,
                       &ctx->unit_start2_b, &ctx->ps_1_b, &ctx->part_tmp_b, &ctx->seg_rng, ctx->aux_stream};
}

struct FusedIn {
    const void *aos = nullptr;
    const u64 *P = nullptr;
    const u32 *K = nullptr;
    int nseg = 0;
    const u64 *seg_off = nullptr;      // host, nseg + 1
    // what pass 2 writes (segmented input): 0 narrow, rowIDs as they came; 1 narrow, sender tag in the low payload bits;
    // 2 16-byte tuples with global rowIDs key_bases[sender] + rowID32 (key_bases: device, 16 u64)
    int final_form = 0;
    const u64 *key_bases = nullptr;
};

// mix (16-byte input only): MIX_STORE inside a join -- the histogram and pass 1 take their digits from mix64(payload) and pass 1
// writes the mixed value, so that pass 2 and the bucket join work on it unchanged
int partition_relation_fused(rhj_ctx *ctx, const FusedIn &in, u64 n, int b1, int b2, void *d_out, u64 *d_ps, int narrow = 0,
                             int mix = MIX_NONE, const PartScratch *scratch = nullptr)
{
    // the scratch tables and the stream of this relation: the context's first set on its stream, or -- R and S of a mid-size
    // join partitioned side by side -- the second set on the auxiliary stream (partition_phase)
    const PartScratch sc = scratch ? *scratch : first_scratch(ctx);
    DevBuf &x_seg0 = *sc.seg0, &x_unit_start = *sc.unit_start, &x_unit_hist = *sc.unit_hist, &x_unit_base = *sc.unit_base,
           &x_scan_tmp = *sc.scan_tmp, &x_hist2 = *sc.hist2, &x_grp_rng = *sc.grp_rng, &x_unit_start2 = *sc.unit_start2,
           &x_ps_1 = *sc.ps_1, &x_part_tmp = *sc.part_tmp, &x_seg_rng = *sc.seg_rng;
    const hipStream_t st = sc.st;
    const bool segs = in.P != nullptr;
    // 1024 pass-1 units instead of 2048: every unit flushes a 2^(b1+b2)-bin table, and the scatter does not care
    // ([measured] at 10^9 tuples: histogram 2.54 against 2.74 ms, scatter within noise)
    PassGeom g1 = make_geom(n, 1, 0, b1, PART_TARGET_UNITS / 2);
    g1.mix = segs ? MIX_NONE : mix;
    static const u32 want_groups = (u32)env_u64("RHJ_GROUPS", 16, 1, 64);                          // tuning aid
    u32 units1, per, ngroups, groups_per_seg = 0;
    u64 segL[16] = {0};
    if (segs) {
        if (in.nseg < 1 || in.nseg > seg_max() || narrow != 2 || b1 < tag_bits() || in.final_form < 0 || in.final_form > 2 ||
            (in.final_form == 2 && !in.key_bases))
            return fail(ctx, RHJ_E_INVALID, "segmented input: 1..16 segments, narrow format, pass 1 of at least 4 bits");
        groups_per_seg = want_groups / (u32)in.nseg ? want_groups / (u32)in.nseg : 1u;
        ngroups = groups_per_seg * (u32)in.nseg;
        per = (u32)(PART_TARGET_UNITS / 2) / ngroups ? (u32)(PART_TARGET_UNITS / 2) / ngroups : 1u;
        const u32 units_per_seg = groups_per_seg * per;
        units1 = units_per_seg * (u32)in.nseg;
        for (int s_ = 0; s_ < in.nseg; s_++) {
            const u64 len = in.seg_off[s_ + 1] - in.seg_off[s_];
            u64 L = (len + units_per_seg - 1) / units_per_seg;
            L = (L + PART_TILE - 1) / PART_TILE * PART_TILE;
            segL[s_] = L ? L : (u64)PART_TILE;
        }
    } else {
        units1 = (u32)((n + g1.L - 1) / g1.L);
        per = units1 ? (units1 + want_groups - 1) / want_groups : 1;
        ngroups = units1 ? (units1 + per - 1) / per : 1;
    }
    const size_t nb1 = (size_t)1 << b1, nb2 = (size_t)1 << b2;
    const u32 units2 = (u32)(nb1 * ngroups);
    RHJCHK(ensure(ctx, x_seg0, 64));
    RHJCHK(ensure(ctx, x_unit_start, 16));
    RHJCHK(ensure(ctx, x_unit_hist, (size_t)(units1 + 1) * nb1 * 4));
    RHJCHK(ensure(ctx, x_unit_base, (size_t)((units1 + 1) * nb1 > units2 * nb2 ? (units1 + 1) * nb1 : units2 * nb2) * 8));
    RHJCHK(ensure(ctx, x_scan_tmp, scan_tmp_bytes(b1)));
    RHJCHK(ensure(ctx, x_hist2, (size_t)units2 * nb2 * 4));
    RHJCHK(ensure(ctx, x_grp_rng, ((size_t)units2 + 1) * 8));
    RHJCHK(ensure(ctx, x_unit_start2, (nb1 + 1) * 4));
    RHJCHK(ensure(ctx, x_part_tmp, (size_t)(n ? n : 1) * 16));
    RHJCHK(ensure(ctx, x_ps_1, (nb1 + 1) * 8));
    if (segs) RHJCHK(ensure(ctx, x_seg_rng, ((size_t)units1 + 1) * 8));
    u64 *seg0 = (u64 *)x_seg0.p;
    u32 *unit_start1 = (u32 *)x_unit_start.p;
    const u64 *rng1 = segs ? (const u64 *)x_seg_rng.p : nullptr;
    u32 *wide = narrow ? (u32 *)ctx->narrow_flag.p : nullptr;
    PassGeom gs = g1;                                   // (segmented: the scan only needs the unit count)
    gs.max_units = units1;
    {
        Span s(ctx, RHJ_K_AUX);
        if (segs) launch_seg_units(st, (u32)in.nseg, in.seg_off, segL, groups_per_seg * per, (u64 *)x_seg_rng.p, seg0, unit_start1);
        else launch_init_single_segment(st, n, g1.L, seg0, unit_start1);          // seg0 = {0,n}, unit_start1 = {0, units1}
        HIPCHK(ctx, hipMemsetAsync(x_hist2.p, 0, (size_t)units2 * nb2 * 4, st));
    }
    DupSniff sn;
    if (ctx->sniff_side >= 0) {                         // a join's relation: sample its join values for duplicates
        const int side = ctx->sniff_side;
        sn.tab = (u32 *)ctx->sniff_tab.p + (size_t)side * SNIFF_SLOTS;
        sn.sel_bits = sniff_sel_bits(n);
        ctx->sniff_n[side] = n;
        Span s(ctx, RHJ_K_AUX);
        HIPCHK(ctx, hipMemsetAsync(sn.tab, 0, (size_t)SNIFF_SLOTS * 4, st));
    }
    {
        Span s(ctx, RHJ_K_HIST);                        // (16-byte input: also reports a rowID that does not fit the narrow format)
        launch_hist2d_units(st, segs ? (const void *)in.P : in.aos, segs, n, g1.L, units1, b1, b2, per, ngroups,
                            (u32 *)x_unit_hist.p, (u32 *)x_hist2.p, 0, wide, rng1, g1.mix, sn);
    }
    {
        Span s(ctx, RHJ_K_SCAN);
        launch_scan_units(st, gs, seg0, unit_start1, (const u32 *)x_unit_hist.p, (u64 *)x_unit_base.p,
                          (u64 *)x_ps_1.p, (u64 *)x_scan_tmp.p);
    }
    {
        Span s(ctx, RHJ_K_SCATTER);
        if (segs)
            launch_scatter_ranges_narrow(st, in.P, true, x_part_tmp.p, n, units1, 0, b1, (const u64 *)x_unit_base.p,
                                         rng1, wide, 0, 0, in.K);
        else if (narrow == 2)
            launch_scatter_units_narrow(st, in.aos, x_part_tmp.p, n, g1, seg0, unit_start1, (const u64 *)x_unit_base.p,
                                        wide);
        else
            launch_scatter_units(st, in.aos, x_part_tmp.p, g1, seg0, unit_start1, (const u64 *)x_unit_base.p);
    }
    {
        Span s(ctx, RHJ_K_AUX);
        launch_make_group_ranges(st, (const u64 *)x_unit_base.p, (u32)nb1, per, ngroups, n, (u64 *)x_grp_rng.p,
                                 (u32 *)x_unit_start2.p);
    }
    PassGeom g2;
    g2.n = n; g2.L = 0; g2.nseg = (u32)nb1; g2.max_units = units2; g2.shift = b1; g2.bits = b2;
    {
        Span s(ctx, RHJ_K_SCAN);
        launch_scan_units(st, g2, (const u64 *)x_ps_1.p, (const u32 *)x_unit_start2.p, (const u32 *)x_hist2.p,
                          (u64 *)x_unit_base.p, d_ps, nullptr);
    }
    {
        Span s(ctx, RHJ_K_SCATTER);
        if (segs && in.final_form == 2)
            launch_scatter_ranges_n2a(st, x_part_tmp.p, d_out, n, units2, b1, b2, (const u64 *)x_unit_base.p,
                                      (const u64 *)x_grp_rng.p, in.key_bases, ngroups, groups_per_seg, wide);
        else if (narrow)
            launch_scatter_ranges_narrow(st, x_part_tmp.p, narrow == 2, d_out, n, units2, b1, b2,
                                         (const u64 *)x_unit_base.p, (const u64 *)x_grp_rng.p, wide,
                                         segs && in.final_form == 1 ? ngroups : 0u, segs && in.final_form == 1 ? groups_per_seg : 0u);
        else
            launch_scatter_ranges(st, x_part_tmp.p, d_out, units2, b1, b2, (const u64 *)x_unit_base.p,
                                  (const u64 *)x_grp_rng.p);
    }
    return check_launch(ctx, "fused two-pass partition");
}

int partition_relation_fused(rhj_ctx *ctx, const void *d_in, u64 n, int b1, int b2, void *d_out, u64 *d_ps, int narrow = 0,
                             int mix = MIX_NONE, const PartScratch *scratch = nullptr)
{
    FusedIn in;
    in.aos = d_in;
    return partition_relation_fused(ctx, in, n, b1, b2, d_out, d_ps, narrow, mix, scratch);
}

// Two narrow passes with SEPARATE histograms (plans of 17-18 bits: 2^(b1+b2) packed counters do not fit the LDS, so the
// histogram of pass 2 is a second read -- of the narrow intermediate, 8 B/tuple): 16-byte tuples -> narrow part_tmp -> narrow
// d_out.  A rowID >= 2^32 raises ctx->narrow_flag in pass 1 (every later kernel of the join returns at once).
// d_inK != nullptr: the input is narrow already (d_in = payloads, d_inK = rowIDs: a received shard whose rowIDs are global).
int partition_relation_narrow2(rhj_ctx *ctx, const void *d_in, u64 n, int b1, int b2, void *d_out, u64 *d_ps,
                               const u32 *d_inK = nullptr, int mix = MIX_NONE)
{
    RHJCHK(ensure(ctx, ctx->seg0, 64));
    RHJCHK(ensure(ctx, ctx->part_tmp, (size_t)(n ? n : 1) * 16));
    RHJCHK(ensure(ctx, ctx->ps_1, (((size_t)1 << b1) + 1) * 8));
    u32 *wide = (u32 *)ctx->narrow_flag.p;
    for (int pass = 0; pass < 2; pass++) {
        const int bits = pass ? b2 : b1, shift = pass ? b1 : 0;
        const u32 nseg = pass ? 1u << b1 : 1u;
        const u64 *seg_start = pass ? (const u64 *)ctx->ps_1.p : (const u64 *)ctx->seg0.p;
        u64 *part_start = pass ? d_ps : (u64 *)ctx->ps_1.p;
        PassGeom g = make_geom(n, nseg, shift, bits);
        g.mix = (!pass && !d_inK) ? mix : MIX_NONE;       // (the 16-byte input of pass 1; from there on the mixed value is the payload)
        const size_t nbins = (size_t)1 << bits;
        RHJCHK(ensure(ctx, ctx->unit_start, ((size_t)nseg + 1) * 4));
        RHJCHK(ensure(ctx, ctx->unit_hist, (size_t)g.max_units * nbins * 4));
        RHJCHK(ensure(ctx, ctx->unit_base, (size_t)g.max_units * nbins * 8));
        RHJCHK(ensure(ctx, ctx->scan_tmp, scan_tmp_bytes(bits)));
        u32 *unit_start = (u32 *)ctx->unit_start.p;
        {
            Span s(ctx, RHJ_K_AUX);
            if (!pass) launch_init_single_segment(ctx->stream, n, g.L, (u64 *)ctx->seg0.p, unit_start);
            else launch_make_units(ctx->stream, seg_start, nseg, g.L, unit_start);
        }
        DupSniff sn;
        if (!pass && !d_inK && ctx->sniff_side >= 0) {  // a join's relation: sample its join values for duplicates
            sn.tab = (u32 *)ctx->sniff_tab.p + (size_t)ctx->sniff_side * SNIFF_SLOTS;
            sn.sel_bits = sniff_sel_bits(n);
            ctx->sniff_n[ctx->sniff_side] = n;
            Span s(ctx, RHJ_K_AUX);
            HIPCHK(ctx, hipMemsetAsync(sn.tab, 0, (size_t)SNIFF_SLOTS * 4, ctx->stream));
        }
        {
            Span s(ctx, RHJ_K_HIST);
            if (!pass && !d_inK) launch_hist_units(ctx->stream, d_in, g, seg_start, unit_start, (u32 *)ctx->unit_hist.p, nullptr, sn);
            else launch_hist_units_narrow(ctx->stream, pass ? ctx->part_tmp.p : d_in, g, seg_start, unit_start, (u32 *)ctx->unit_hist.p);
        }
        {
            Span s(ctx, RHJ_K_SCAN);
            launch_scan_units(ctx->stream, g, seg_start, unit_start, (const u32 *)ctx->unit_hist.p, (u64 *)ctx->unit_base.p,
                              part_start, (u64 *)ctx->scan_tmp.p);
        }
        {
            Span s(ctx, RHJ_K_SCATTER);
            void *out = pass ? d_out : ctx->part_tmp.p;
            launch_scatter_units_narrow_any(ctx->stream, pass ? ctx->part_tmp.p : d_in,
                                            pass ? (const u32 *)((const unsigned char *)ctx->part_tmp.p + narrow_k_offset(n)) : d_inK,
                                            out, (u32 *)((unsigned char *)out + narrow_k_offset(n)), g, seg_start, unit_start,
                                            (const u64 *)ctx->unit_base.p, wide);
        }
    }
    return check_launch(ctx, "two-pass narrow partition");
}

// Partition one relation with `passes` passes into d_out; boundaries into d_ps[2^(b1+b2) + 1].
int partition_relation(rhj_ctx *ctx, const void *d_in, u64 n, int passes, int b1, int b2, void *d_out, u64 *d_ps, int mix = MIX_NONE)
{
    if (passes == 2 && fused_two_pass_ok(b1, b2) && n > 0 && n < ((u64)1 << 32))
        return partition_relation_fused(ctx, d_in, n, b1, b2, d_out, d_ps, 0, mix);
    RHJCHK(ensure(ctx, ctx->seg0, 64));
    u64 *seg0 = (u64 *)ctx->seg0.p;                 // {0, n}: written by the first pass itself
    if (passes == 1) return run_pass(ctx, d_in, d_out, n, seg0, 1, 0, b1, d_ps, true, mix);
    RHJCHK(ensure(ctx, ctx->part_tmp, (size_t)(n ? n : 1) * 16));
    RHJCHK(ensure(ctx, ctx->ps_1, (((size_t)1 << b1) + 1) * 8));
    RHJCHK(run_pass(ctx, d_in, ctx->part_tmp.p, n, seg0, 1, 0, b1, (u64 *)ctx->ps_1.p, true, mix));
    return run_pass(ctx, ctx->part_tmp.p, d_out, n, (const u64 *)ctx->ps_1.p, 1u << b1, b1, b2, d_ps);
}

// Which bucket-join kernel a partitioned join runs (JoinKernel).
// Average build partition larger than one 4224-tuple table (an explicit plan with too few bits, or more than 2^30
// tuples): the compact-table kernel when the plan removed enough payload bits for 48-bit keys, else 8448-tuple
// chunks.  The compact-table kernel keeps a task's probe rowIDs in registers, so a task is at most that many tuples.
// allow13: plans of 13-15 bits may take the compact-table kernel with 13-bit arrival indices (keys of up to 51 bits) for
// average partitions of 2-5 K tuples (not the multi-GPU receiver, whose sender tags the one-table kernel resolves).
int choose_join_kind(const rhj_ctx *ctx, u64 nR, u64 nS, u64 nparts, int radix_bits, bool narrow = false, bool allow13 = true)
{
    const u64 nbuild = nR < nS ? nR : nS;
    if (radix_bits < join_ct_min_radix_bits(JK_CT) && radix_bits >= join_ct_min_radix_bits(JK_CT_Q12) && allow13 &&
        (ctx->opt_big_kernel < 0 || ctx->opt_big_kernel == JK_CT_Q12) && (radix_bits == join_ct_min_radix_bits(JK_CT_Q12) || ctx->opt_big_kernel == JK_CT_Q12)) {
        // plans of exactly 12 bits (and a forced kernel 11 from 12 bits on): the 4096-entry table with 12-bit arrival indices
        static const bool on12 = env_u64("RHJ_CT13", 1, 0, 1) != 0;
        const u64 np12 = nR < nS ? nS : nR, ab12 = nbuild / nparts, ap12 = np12 / nparts;
        const bool fits12 = ab12 <= (u64)join_table_tuples(JK_CT_Q12) * 15 / 16 && ap12 <= (u64)join_probe_split(JK_CT_Q12) * 15 / 16;
        if ((ctx->opt_big_kernel == JK_CT_Q12 && ctx->opt_big_tables == 1) ||
            (on12 && ctx->opt_big_tables < 0 && ab12 > (u64)CT_GUARDED_FROM && fits12))
            return JK_CT_Q12;
    }
    if (radix_bits < join_ct_min_radix_bits(JK_CT) && radix_bits >= join_ct_min_radix_bits(JK_CT_G13) && allow13 &&
        (ctx->opt_big_kernel < 0 || ctx->opt_big_kernel == JK_CT_G13)) {
        static const bool on = env_u64("RHJ_CT13", 1, 0, 1) != 0;              // tuning aid: 0 = the one-table kernel as before
        const u64 nprobe13 = nR < nS ? nS : nR, ab13 = nbuild / nparts, ap13 = nprobe13 / nparts;
        const bool fits13 = ab13 <= (u64)join_table_tuples(JK_CT_G13) * 15 / 16 && ap13 <= (u64)join_probe_split(JK_CT_G13) * 15 / 16 &&
                            ab13 <= g13_upto() && ap13 <= g13_upto();
        if ((ctx->opt_big_kernel == JK_CT_G13 && ctx->opt_big_tables == 1) ||
            (on && ctx->opt_big_tables < 0 && ab13 > (u64)CT_GUARDED_FROM && fits13))
            return JK_CT_G13;
    }
    // ... and from half a table on when the plan allows the compact-table kernel: its 6144-entry geometry with row guards
    // handles a tuple in two thirds of the one-table kernel's time ([measured] 1.5 - 2.7 * 10^8 tuples under 16 bits)
    const bool ct_ok = radix_bits >= join_ct_min_radix_bits(JK_CT) && ctx->opt_big_kernel != JK_BKT_BIG;
    const u64 big_from = ct_ok ? (u64)CT_GUARDED_FROM : (u64)BJ_CHUNK;
    if (!(ctx->opt_big_tables == 1 || (ctx->opt_big_tables < 0 && nbuild / nparts > big_from))) return JK_BKT;
    if (radix_bits < join_ct_min_radix_bits(JK_CT) || ctx->opt_big_kernel == JK_BKT_BIG) return JK_BKT_BIG;
    if (jk_is_ct(ctx->opt_big_kernel) && !jk_ct_narrow_only(ctx->opt_big_kernel)) return ctx->opt_big_kernel;
    if (ctx->opt_big_kernel == JK_CT_WIDE) return narrow ? JK_CT_WIDE : JK_CT_13;
    if (ctx->opt_big_kernel == JK_CT_HALF_WIDE) return narrow ? JK_CT_HALF_WIDE : JK_CT_HALF;
    // the compact-table kernel at half size (two workgroups per CU) while the average partition fits its table
    // AND its 8192-tuple probe tasks (a partition cut into two tasks builds its table twice)
    const u64 nprobe = nR < nS ? nS : nR, ab = nbuild / nparts, ap = nprobe / nparts;
    auto fits = [&](int k) { return ab <= (u64)join_table_tuples(k) * 15 / 16 && ap <= (u64)join_probe_split(k) * 15 / 16; };
    // 12 + 12 slot rows, 6144 entries: partitions of up to 5.76 K tuples; with row guards while two or more rows stay empty
    if (fits(JK_CT_HALF_MID)) return ab <= (u64)CT_GUARDED_UPTO && ap <= (u64)CT_GUARDED_UPTO ? JK_CT_HALF_MID_G : JK_CT_HALF_MID;
    if (fits(JK_CT_HALF)) return JK_CT_HALF;
    // 20 probe slots per thread (narrow partitions only) before a partition's probe side is cut into two tasks that build the
    // table twice: [measured] 2.2 * 10^9 x 2.2 * 10^9, join kernel 32.5 ms with two 16-slot tasks per partition
    if (narrow && fits(JK_CT_HALF_WIDE)) return JK_CT_HALF_WIDE;
    if (fits(JK_CT_MID)) return JK_CT_MID;           // 12 + 12 slot rows instead of 18 + 16 for partitions of up to 11.5 K tuples
    if (fits(JK_CT)) return JK_CT;                   // 16352 entries in 16384 buckets (up to 1.005 * 10^9 tuples under 16 bits)
    if (fits(JK_CT_13)) return JK_CT_13;             // 17920 entries in 8192 buckets
    if (narrow && fits(JK_CT_WIDE)) return JK_CT_WIDE;
    return JK_CT_13;
}

bool narrow_fused_plan(const rhj_opts &plan)
{
    return fused_two_pass_ok(plan.bits1, plan.bits2) && narrow_pass_ok(plan.bits1) && narrow_pass_ok(plan.bits2);
}

// The narrow intermediate format applies to fused two-pass plans (both passes <= 8 bits) whose bucket join is the one-table
// kernel or the compact-table kernel, for relations of < 2^32 tuples (rowIDs that CAN be 32 bits; whether they are is
// found out on the device).  Returns the level (0 = not at all).
int narrow_level(const rhj_ctx *ctx, u64 nR, u64 nS, const rhj_opts &plan)
{
    static const int env = (int)env_u64("RHJ_NARROW", 2, 0, 2);
    const int want = ctx->opt_narrow >= 0 ? ctx->opt_narrow : env;
    if (want <= 0 || ctx->narrow_off_once || plan.passes != 2) return 0;
    if (ctx->opt_narrow < 0 && ctx->narrow_skip > 0) return 0;              // backing off after repeated wide rowIDs (automatic mode only)
    const bool fused = narrow_fused_plan(plan);
    // plans beyond 16 bits: two narrow passes with separate histograms (level 2 only), 9-bit passes in the 16-tuple-line geometry
    if (!fused && !(want >= 2 && plan.bits1 + plan.bits2 > 16 && narrow_pass9_ok(plan.bits1) && narrow_pass9_ok(plan.bits2))) return 0;
    const u64 lo = nR < nS ? nR : nS, hi = nR < nS ? nS : nR;
    if (lo < NARROW_MIN_TUPLES || hi >= ((u64)1 << 32)) return 0;
    const int tb = plan.bits1 + plan.bits2;
    const int kind = choose_join_kind(ctx, nR, nS, (u64)1 << tb, tb, true);
    if (kind == JK_BKT_BIG) return 0;
    // the narrow scatter has one 1024-thread workgroup per CU and 32-tuple lines to start and finish per digit and unit:
    // below a few million tuples its fixed costs outweigh the bytes it saves (forced levels, used by the tests, skip this)
    static const u64 min_auto = env_u64("RHJ_NARROW_MIN", NARROW_AUTO_MIN_TUPLES, 0, ~0ull);
    if (ctx->opt_narrow < 0 && hi < min_auto) return 0;
    return want >= 2 ? 2 : 1;
}

// Small unpartitioned joins run as ONE launch without a task list (k_join_bkt DIRECT).
bool is_direct(const rhj_ctx *ctx, u64 nparts, u64 nR, u64 nS)
{
    return nparts == 1 && ctx->opt_big_tables != 1 && (nR < nS ? nR : nS) <= DIRECT_MAX_BUILD &&
           (nR < nS ? nS : nR) <= DIRECT_MAX_PROBE;
}

// R and S of a fused two-pass plan side by side on two streams: device-resident joins (nothing to upload in between) of at
// most 2^28 tuples per side (the second pass-1 intermediate costs 16 B per tuple of S), not while profiling (spans are timed
// on one stream).  RHJ_TWO_STREAMS=0 switches it off.
bool two_streams_ok(const rhj_ctx *ctx, u64 nR, u64 nS, const rhj_opts &plan, std::function<int()> *before_S)
{
    static const bool on = env_u64("RHJ_TWO_STREAMS", 1, 0, 1) != 0;
    if (!on || ctx->prof.on || (before_S && *before_S)) return false;
    if (plan.passes != 2 || !fused_two_pass_ok(plan.bits1, plan.bits2)) return false;
    if (ctx->cur_narrow && !narrow_fused_plan(plan)) return false;
    const u64 hi = nR > nS ? nR : nS, lo = nR > nS ? nS : nR;
    return lo > 0 && hi <= ((u64)1 << 28);
}

// Partition phase of a join: leaves ctx->cur_* describing partitioned R and S.
// before_S (optional, consumed by the first call that gets it): invoked once, after the kernels that partition R have been
// enqueued and before anything reads S -- rhj_join uploads S there, so that S crosses PCIe while R is being partitioned
// (plans that push both relations through the same launches call it first).
int partition_phase(rhj_ctx *ctx, const void *d_R, u64 nR, const void *d_S, u64 nS, const rhj_opts &plan,
                    std::function<int()> *before_S = nullptr)
{
    ctx->sniff_ready = false;
    auto s_ready = [&]() -> int {
        if (!before_S || !*before_S) return RHJ_OK;
        std::function<int()> f;
        f.swap(*before_S);
        return f();
    };
    ctx->cur_nR = nR;
    ctx->cur_nS = nS;
    ctx->cur_probe_split = (u32)plan.probe_split;
    ctx->last.passes = plan.passes;
    ctx->last.bits1 = plan.bits1;
    ctx->last.bits2 = plan.bits2;
    ctx->counters_clean = false;
    ctx->cur_narrow = narrow_level(ctx, nR, nS, plan);
    if (ctx->cur_narrow) {
        RHJCHK(ensure(ctx, ctx->narrow_flag, 64));
        Span s(ctx, RHJ_K_AUX);
        HIPCHK(ctx, hipMemsetAsync(ctx->narrow_flag.p, 0, 64, ctx->stream));
    }
    if (plan.passes == 0) {
        RHJCHK(s_ready());
        RHJCHK(ensure(ctx, ctx->ps_R, 64));
        RHJCHK(ensure(ctx, ctx->ps_S, 64));
        if (!is_direct(ctx, 1, nR, nS)) {                               // boundaries {0, n} for the task list (the direct launch needs none)
            Span s(ctx, RHJ_K_AUX);
            launch_init_single_segment(ctx->stream, nR, PART_TILE, (u64 *)ctx->ps_R.p, (u32 *)((u64 *)ctx->ps_R.p + 4));
            launch_init_single_segment(ctx->stream, nS, PART_TILE, (u64 *)ctx->ps_S.p, (u32 *)((u64 *)ctx->ps_S.p + 4));
        }
        ctx->cur_R = d_R;
        ctx->cur_S = d_S;
        ctx->cur_nparts = 1;
        ctx->cur_radix_bits = 0;
    } else {
        const int tb = plan.bits1 + (plan.passes == 2 ? plan.bits2 : 0);
        const size_t np = (size_t)1 << tb;
        const int mix = join_mix(ctx);
        RHJCHK(ensure(ctx, ctx->ps_R, (np + 1) * 8));
        RHJCHK(ensure(ctx, ctx->ps_S, (np + 1) * 8));
        RHJCHK(ensure(ctx, ctx->part_R, (size_t)(nR ? nR : 1) * 16));
        RHJCHK(ensure(ctx, ctx->part_S, (size_t)(nS ? nS : 1) * 16));
        if (ctx->cur_narrow && !narrow_fused_plan(plan)) {
            const bool sniff = sniff_on(ctx);
            if (sniff) RHJCHK(ensure(ctx, ctx->sniff_tab, (size_t)2 * SNIFF_SLOTS * 4));
            ctx->sniff_side = sniff ? 0 : -1;
            int prc = partition_relation_narrow2(ctx, d_R, nR, plan.bits1, plan.bits2, ctx->part_R.p, (u64 *)ctx->ps_R.p, nullptr, mix);
            if (prc == RHJ_OK) prc = s_ready();
            ctx->sniff_side = sniff ? 1 : -1;
            if (prc == RHJ_OK)
                prc = partition_relation_narrow2(ctx, d_S, nS, plan.bits1, plan.bits2, ctx->part_S.p, (u64 *)ctx->ps_S.p, nullptr, mix);
            ctx->sniff_side = -1;
            RHJCHK(prc);
            ctx->sniff_ready = sniff;
        } else if (two_streams_ok(ctx, nR, nS, plan, before_S)) {
            // Mid-size joins are launch-bound (a fused two-pass partition is ~11 short dependent launches per relation): R and S
            // are independent until the join, so S is partitioned on a second stream, in a second set of scratch tables, while R
            // runs on the first -- the gaps between one relation's launches are filled by the other's kernels.
            if (!ctx->aux_stream) {
                HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->aux_stream, hipStreamNonBlocking));
                for (int i = 0; i < 2; i++) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->aux_ev[i], hipEventDisableTiming));
            }
            const bool sniff = sniff_on(ctx);
            if (sniff) RHJCHK(ensure(ctx, ctx->sniff_tab, (size_t)2 * SNIFF_SLOTS * 4));
            HIPCHK(ctx, hipEventRecord(ctx->aux_ev[0], ctx->stream));             // (the inputs, the cleared flag: everything so far)
            HIPCHK(ctx, hipStreamWaitEvent(ctx->aux_stream, ctx->aux_ev[0], 0));
            const PartScratch s2 = second_scratch(ctx);
            ctx->sniff_side = sniff ? 0 : -1;
            int prc = partition_relation_fused(ctx, d_R, nR, plan.bits1, plan.bits2, ctx->part_R.p, (u64 *)ctx->ps_R.p, ctx->cur_narrow, mix);
            ctx->sniff_side = sniff ? 1 : -1;
            if (prc == RHJ_OK)
                prc = partition_relation_fused(ctx, d_S, nS, plan.bits1, plan.bits2, ctx->part_S.p, (u64 *)ctx->ps_S.p, ctx->cur_narrow, mix, &s2);
            ctx->sniff_side = -1;
            RHJCHK(prc);
            ctx->sniff_ready = sniff;
            HIPCHK(ctx, hipEventRecord(ctx->aux_ev[1], ctx->aux_stream));
            HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->aux_ev[1], 0));
        } else if (ctx->cur_narrow) {
            const bool sniff = sniff_on(ctx);
            if (sniff) RHJCHK(ensure(ctx, ctx->sniff_tab, (size_t)2 * SNIFF_SLOTS * 4));
            ctx->sniff_side = sniff ? 0 : -1;
            int prc = partition_relation_fused(ctx, d_R, nR, plan.bits1, plan.bits2, ctx->part_R.p, (u64 *)ctx->ps_R.p, ctx->cur_narrow, mix);
            if (prc == RHJ_OK) prc = s_ready();
            ctx->sniff_side = sniff ? 1 : -1;
            if (prc == RHJ_OK)
                prc = partition_relation_fused(ctx, d_S, nS, plan.bits1, plan.bits2, ctx->part_S.p, (u64 *)ctx->ps_S.p, ctx->cur_narrow, mix);
            ctx->sniff_side = -1;
            RHJCHK(prc);
            ctx->sniff_ready = sniff;
        } else if (plan.passes == 1 && plan.bits1 <= PASS_PAIR_MAX_BITS) {
            RHJCHK(s_ready());
            RHJCHK(run_pass_pair(ctx, d_R, nR, ctx->part_R.p, (u64 *)ctx->ps_R.p, d_S, nS, ctx->part_S.p, (u64 *)ctx->ps_S.p,
                                 plan.bits1));
        } else {
            RHJCHK(partition_relation(ctx, d_R, nR, plan.passes, plan.bits1, plan.bits2, ctx->part_R.p, (u64 *)ctx->ps_R.p, mix));
            RHJCHK(s_ready());
            RHJCHK(partition_relation(ctx, d_S, nS, plan.passes, plan.bits1, plan.bits2, ctx->part_S.p, (u64 *)ctx->ps_S.p, mix));
        }
        ctx->cur_R = ctx->part_R.p;
        ctx->cur_S = ctx->part_S.p;
        ctx->cur_nparts = np;
        ctx->cur_radix_bits = tb;
    }
    ctx->cur_psR = (const u64 *)ctx->ps_R.p;
    ctx->cur_psS = (const u64 *)ctx->ps_S.p;
    return check_launch(ctx, "partition phase");
}

// Join phase on explicit partitioned inputs.  Synchronises to read the exact result count.
// narrow: the partitions are narrow arrays; tag_base (device, 2 x 16 u64; narrow only): the low payload bits name the sender
// of a tuple and rowIDs are local to it (multi-GPU receiver).  allow_direct: the boundaries are known to be {0, n}.
int join_phase_on(rhj_ctx *ctx, const void *d_Rp, const u64 *d_psR, u64 nR, const void *d_Sp, const u64 *d_psS,
                  u64 nS, u64 nparts, int radix_bits, u32 probe_split, void *d_out, u64 cap, u64 *out_count, bool narrow = false,
                  const u64 *d_tag_base = nullptr, bool allow_direct = true, bool check_radix = false, bool keep_count = false,
                  bool enqueue_only = false, bool allow13 = true)
{
    // keep_count: the result counter goes on from where the previous join on this context left it (the pairs of several joins
    // land behind one another in d_out); enqueue_only: no read-back, the caller collects the counters itself
    if (probe_split == 0) probe_split = 32768;
    // a task addresses its build range with 32 bits; k_make_tasks reports any partition whose build side is larger
    // (counters[5], checked below) whatever the plan
    const int kind = choose_join_kind(ctx, nR, nS, nparts, radix_bits, narrow, allow13);
    if (narrow && kind != JK_BKT && !jk_is_ct(kind))
        return fail(ctx, RHJ_E_INVALID, "no bucket-join kernel for narrow partitions under this plan");
    if (join_probe_split(kind) && probe_split > join_probe_split(kind)) probe_split = join_probe_split(kind);
    // the kernels address a task's probe side through a buffer descriptor of 32-bit byte size (16 B per tuple)
    if (probe_split > BJ_MAX_PROBE_SPLIT) probe_split = BJ_MAX_PROBE_SPLIT;
    const u64 max_tasks64 = nparts + (nR + nS) / probe_split + 1;
    if (max_tasks64 > 0x7fffffffull) return fail(ctx, RHJ_E_INVALID, "too many join tasks");
    const u32 max_tasks = (u32)max_tasks64;
    RHJCHK(ensure(ctx, ctx->tasks, (size_t)max_tasks * sizeof(JoinTask)));
    RHJCHK(ensure(ctx, ctx->counters, 64));
    u64 *d_count = (u64 *)ctx->counters.p;
    u32 *d_ntasks = (u32 *)(d_count + 1);
    if (keep_count) {
        Span s(ctx, RHJ_K_AUX);
        HIPCHK(ctx, hipMemsetAsync((unsigned char *)ctx->counters.p + 8, 0, 56, ctx->stream));
    } else if (!ctx->counters_clean) {               // (a paired partition pass has cleared them already)
        Span s(ctx, RHJ_K_AUX);
        HIPCHK(ctx, hipMemsetAsync(ctx->counters.p, 0, 64, ctx->stream));
    }
    ctx->counters_clean = false;
    const bool direct = allow_direct && !narrow && is_direct(ctx, nparts, nR, nS);
    ctx->last_join_kind = direct ? -1 : kind;
    if (check_radix && !direct && jk_is_ct(kind)) {       // counters[6]: the partitions break the radix_bits contract
        Span s(ctx, RHJ_K_AUX);
        launch_check_radix(ctx->stream, d_Rp, d_psR, d_Sp, d_psS, nparts, radix_bits, d_count + 6);
    }
    if (direct) {
        Span s(ctx, RHJ_K_JOIN);                                     // small unpartitioned join: one launch, no task list
        launch_join_direct(ctx->stream, d_Rp, nR, d_Sp, nS, d_out, d_out ? cap : 0, d_count);
    } else {
        {
            Span s(ctx, RHJ_K_TASKS);
            SniffVerdict sv;                                        // (the partition phase of THIS join sampled both sides: ask them)
            if (ctx->sniff_ready && ctx->sniff_n[0] == nR && ctx->sniff_n[1] == nS) {
                sv.tab = (const u32 *)ctx->sniff_tab.p;
                sv.expect_R = (u32)(nR >> sniff_sel_bits(nR));
                sv.expect_S = (u32)(nS >> sniff_sel_bits(nS));
            }
            ctx->sniff_ready = false;
            launch_make_tasks(ctx->stream, d_psR, d_psS, nparts, probe_split, (JoinTask *)ctx->tasks.p, d_ntasks, max_tasks,
                              d_count + 2, kind, sv);               // counters[2..3]: largest partition of R, S
        }
        {
            Span s(ctx, RHJ_K_JOIN);
            launch_join(ctx->stream, d_Rp, d_psR, d_Sp, d_psS, (const JoinTask *)ctx->tasks.p, d_ntasks, max_tasks,
                        radix_bits, d_out, d_out ? cap : 0, d_count, kind,
                        narrow ? (const u32 *)((const unsigned char *)d_Rp + narrow_k_offset(nR)) : nullptr,
                        narrow ? (const u32 *)((const unsigned char *)d_Sp + narrow_k_offset(nS)) : nullptr,
                        d_tag_base, narrow ? (const u32 *)ctx->narrow_flag.p : nullptr);
        }
    }
    RHJCHK(check_launch(ctx, "join phase"));
    if (enqueue_only) return RHJ_OK;
    u64 host[7] = {0, 0, 0, 0, 0, 0, 0};       // count, ntasks, max |R_k|, max |S_k|, (checksum scratch), oversized build side, contract
    u32 wide_rowid = 0;
    HIPCHK(ctx, hipMemcpyAsync(host, ctx->counters.p, sizeof(host), hipMemcpyDeviceToHost, ctx->stream));
    if (narrow) HIPCHK(ctx, hipMemcpyAsync(&wide_rowid, ctx->narrow_flag.p, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (wide_rowid) return RHJ_RETRY_WIDE;                       // a rowID did not fit 32 bits: nothing of this run is valid
    *out_count = host[0];
    ctx->last.ntasks = (u32)(host[1] & 0xffffffffu);
    ctx->last_max_part[0] = direct ? 0 : host[2];
    ctx->last_max_part[1] = direct ? 0 : host[3];
    if (host[5])
        return fail(ctx, RHJ_E_INVALID, "a partition's build side has " + std::to_string(host[5]) +
                                        " tuples (>= 2^32): use more radix bits");
    if (host[6])
        return fail(ctx, RHJ_E_INVALID, "rhj_bucket_join: the payloads of a partition do not share their low radix_bits bits "
                                        "(radix_bits must be the number of low payload bits that are constant inside every partition)");
    return RHJ_OK;
}

int join_phase(rhj_ctx *ctx, void *d_out, u64 cap, u64 *out_count)
{
    return join_phase_on(ctx, ctx->cur_R, ctx->cur_psR, ctx->cur_nR, ctx->cur_S, ctx->cur_psS, ctx->cur_nS,
                         ctx->cur_nparts, ctx->cur_radix_bits, ctx->cur_probe_split, d_out, cap, out_count, ctx->cur_narrow != 0);
}

// bookkeeping of the narrow-format back-off (rhj.h "partition.narrow"), shared by the plain and the pipelined host path
void narrow_note_fallback(rhj_ctx *ctx)                  // a join met a rowID >= 2^32 in the narrow format
{
    ctx->narrow_fail_streak++;
    ctx->narrow_skip = ctx->narrow_fail_streak < 2 ? 0u : (ctx->narrow_fail_streak > 5 ? 32u : 1u << (ctx->narrow_fail_streak - 1));
}
void narrow_note_done(rhj_ctx *ctx, const rhj_opts &plan, bool tried_narrow)   // a join ran to its end in one format
{
    if (tried_narrow) ctx->narrow_fail_streak = 0;
    else if (ctx->narrow_skip > 0 && plan.passes == 2) ctx->narrow_skip--;
}

// One-pass joins (plans of <= 9 bits: 2 * 10^4 ... 8 * 10^6 build tuples) in THREE launches and no device-to-host copy:
// k_hist_fused2 (histograms of both relations, the join counters cleared, the next call's control block zeroed),
// k_scatter_fused2 (both relations; every workgroup derives the partition boundaries itself, a planner workgroup writes them
// and the join task list meanwhile), the bucket join, which publishes its count to pinned host memory.  Was: unit tables, histogram, three scan launches, scatter, k_make_tasks,
// join, a 56-byte D2H copy -- 8 dependent launches for 77 us of kernel time at 10^6 x 10^6 (BASELINE config 2).
bool fused_one_pass_ok(const rhj_ctx *ctx, u64 nR, u64 nS, const rhj_opts &plan)
{
    static const int env = (int)env_u64("RHJ_FUSE", 1, 0, 1);
    if (!(ctx->opt_fused >= 0 ? ctx->opt_fused : env)) return false;
    return plan.passes == 1 && plan.bits1 >= 1 && plan.bits1 <= PASS_PAIR_MAX_BITS && nR > 0 && nS > 0 && nR < ((u64)1 << 32) &&
           nS < ((u64)1 << 32);
}

int join_one_pass_fused(rhj_ctx *ctx, const void *d_R, u64 nR, const void *d_S, u64 nS, const rhj_opts &plan, void *d_out, u64 cap,
                        u64 *out_count, std::function<int()> *before_S)
{
    const int bits = plan.bits1;
    const size_t nbins = (size_t)1 << bits;
    if (before_S && *before_S) {                             // (both relations go through the same launches: S must be there)
        std::function<int()> f;
        f.swap(*before_S);
        RHJCHK(f());
    }
    ctx->cur_nR = nR;
    ctx->cur_nS = nS;
    ctx->last.passes = 1;
    ctx->last.bits1 = bits;
    ctx->last.bits2 = 0;
    ctx->cur_narrow = 0;
    ctx->counters_clean = false;
    int kind = choose_join_kind(ctx, nR, nS, nbins, bits, false);
    u32 probe_split = plan.probe_split ? (u32)plan.probe_split : 32768u;
    if (join_probe_split(kind) && probe_split > join_probe_split(kind)) probe_split = join_probe_split(kind);
    if (probe_split > BJ_MAX_PROBE_SPLIT) probe_split = BJ_MAX_PROBE_SPLIT;
    ctx->cur_probe_split = probe_split;
    const u64 max_tasks64 = nbins + (nR + nS) / probe_split + 1;
    if (max_tasks64 > 0x7fffffffull) return fail(ctx, RHJ_E_INVALID, "too many join tasks");
    const u32 max_tasks = (u32)max_tasks64;
    PassPairHost h;
    DevBuf *uh[2] = {&ctx->unit_hist, &ctx->unit_hist_b};
    const void *in[2] = {d_R, d_S};
    const u64 n[2] = {nR, nS};
    RHJCHK(ensure(ctx, ctx->ps_R, (nbins + 1) * 8));
    RHJCHK(ensure(ctx, ctx->ps_S, (nbins + 1) * 8));
    RHJCHK(ensure(ctx, ctx->part_R, (size_t)nR * 16));
    RHJCHK(ensure(ctx, ctx->part_S, (size_t)nS * 16));
    void *out[2] = {ctx->part_R.p, ctx->part_S.p};
    u64 *ps[2] = {(u64 *)ctx->ps_R.p, (u64 *)ctx->ps_S.p};
    // units: the scatter's workgroups of R and S together fill the chip ONCE (9 bits: one 1024-thread workgroup per CU, below two
    // of 512 threads) -- every workgroup of k_scatter_fused2 first derives the partition starts and reserves its ranges with
    // 2^bits atomics, so a second round of workgroups pays that again ([measured] 3 * 10^6 x 3 * 10^6, 9 bits, units per
    // relation 733 / 256 / 128: scatter 85 / 54 / 44 us, the join 0.191 / 0.157 / 0.148 ms)
    const u64 target_units = bits >= 9 ? 128 : 256;
    for (int i = 0; i < 2; i++) {
        const PassGeom g = make_geom(n[i], 1, 0, bits, target_units);
        RHJCHK(ensure(ctx, *uh[i], (size_t)g.max_units * nbins * 4));
        h.side[i] = PassSide{in[i], out[i], nullptr, nullptr, (u32 *)uh[i]->p, nullptr, ps[i], nullptr, g};
    }
    h.mix = join_mix(ctx);
    RHJCHK(ensure(ctx, ctx->tasks, (size_t)max_tasks * sizeof(JoinTask)));
    RHJCHK(ensure(ctx, ctx->counters, 64));
    if (ctx->fuse_ctl.cap < fuse_ctl_bytes()) { RHJCHK(ensure(ctx, ctx->fuse_ctl, fuse_ctl_bytes())); ctx->fuse_clean = false; }
    if (!ctx->h_pub) {
        HIPCHK(ctx, hipHostMalloc((void **)&ctx->h_pub, 64, hipHostMallocDefault));
        if (hipHostGetDevicePointer((void **)&ctx->h_pub_dev, ctx->h_pub, 0) != hipSuccess || !ctx->h_pub_dev) {
            (void)hipGetLastError();
            (void)hipHostFree(ctx->h_pub);
            ctx->h_pub = nullptr;
            return fail(ctx, RHJ_E_HIP, "no device address for the pinned counter block");
        }
    }
    if (!ctx->fuse_clean) {
        Span s(ctx, RHJ_K_AUX);
        HIPCHK(ctx, hipMemsetAsync(ctx->fuse_ctl.p, 0, fuse_ctl_bytes(), ctx->stream));
    }
    ctx->fuse_clean = false;                                 // until this call has run to its end
    u64 *d_count = (u64 *)ctx->counters.p;
    const int parity = ctx->fuse_parity;
    ctx->fuse_parity ^= 1;
    for (int phase = 0; phase < 2; phase++) {
        Span s(ctx, phase == 0 ? RHJ_K_HIST : RHJ_K_SCATTER);
        launch_fused_pass(ctx->stream, h, bits, phase, parity, ctx->fuse_ctl.p, probe_split, max_tasks, join_table_tuples(kind),
                          (JoinTask *)ctx->tasks.p, d_count, ctx->h_pub_dev, sniff_on(ctx));
    }
    // the packed result counter (rhj_kernels.hip bj_count_packed) while pairs and workgroups fit its two fields
    const bool packed = max_tasks < (1u << 16) && (double)nR * (double)nS < 2.8e14;
    ctx->cur_R = ctx->part_R.p;
    ctx->cur_S = ctx->part_S.p;
    ctx->cur_psR = ps[0];
    ctx->cur_psS = ps[1];
    ctx->cur_nparts = nbins;
    ctx->cur_radix_bits = bits;
    ctx->last_join_kind = kind;
    volatile u64 *pub = ctx->h_pub;
    pub[0] = ~0ull;
    {
        Span s(ctx, RHJ_K_JOIN);
        launch_join(ctx->stream, ctx->cur_R, ps[0], ctx->cur_S, ps[1], (const JoinTask *)ctx->tasks.p, (const u32 *)(d_count + 1), max_tasks, bits,
                    d_out, d_out ? cap : 0, d_count, kind, nullptr, nullptr, nullptr, nullptr, ctx->h_pub_dev,
                    packed ? nullptr : fuse_join_ticket(ctx->fuse_ctl.p));
    }
    RHJCHK(check_launch(ctx, "one-pass join"));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (pub[0] == ~0ull) return fail(ctx, RHJ_E_HIP, "the bucket join did not publish its counters");
    ctx->fuse_clean = true;
    *out_count = pub[0];
    ctx->last.ntasks = (u32)(pub[1] & 0xffffffffu);
    ctx->last_max_part[0] = pub[2];
    ctx->last_max_part[1] = pub[3];
    return RHJ_OK;
}

// partition + join.  A run in the narrow format whose histogram kernel met a rowID >= 2^32 costs two histogram launches
// (every later kernel of the run returns at once) and is repeated in the 16-byte format; the fall-back is per join.
int partition_and_join(rhj_ctx *ctx, const void *d_R, u64 nR, const void *d_S, u64 nS, const rhj_opts &plan, void *d_out,
                       u64 cap, u64 *out_count, std::function<int()> *before_S = nullptr)
{
    if (fused_one_pass_ok(ctx, nR, nS, plan)) return join_one_pass_fused(ctx, d_R, nR, d_S, nS, plan, d_out, cap, out_count, before_S);
    int rc = partition_phase(ctx, d_R, nR, d_S, nS, plan, before_S);
    if (rc != RHJ_OK) { ctx->counters_clean = false; return rc; }
    const bool tried_narrow = ctx->cur_narrow != 0;
    rc = join_phase(ctx, d_out, cap, out_count);
    if (rc != RHJ_RETRY_WIDE) {
        if (rc == RHJ_OK && !ctx->narrow_off_once) narrow_note_done(ctx, plan, tried_narrow);   // (off_once: the repeat of a join that fell back)
        return rc;
    }
    narrow_note_fallback(ctx);
    ctx->narrow_off_once = true;
    rc = partition_phase(ctx, d_R, nR, d_S, nS, plan);
    ctx->narrow_off_once = false;
    if (rc != RHJ_OK) { ctx->counters_clean = false; return rc; }
    return join_phase(ctx, d_out, cap, out_count);
}

// one relation under a resolved two-pass (or one-pass) plan; narrow: the level narrow_level() returned for the join
int partition_side(rhj_ctx *ctx, const void *d_in, u64 n, const rhj_opts &plan, int narrow, void *part, u64 *ps)
{
    const int mix = join_mix(ctx);
    if (narrow && !narrow_fused_plan(plan)) return partition_relation_narrow2(ctx, d_in, n, plan.bits1, plan.bits2, part, ps, nullptr, mix);
    if (narrow) return partition_relation_fused(ctx, d_in, n, plan.bits1, plan.bits2, part, ps, narrow, mix);
    return partition_relation(ctx, d_in, n, plan.passes, plan.bits1, plan.bits2, part, ps, mix);
}

}  // namespace

static std::mutex g_pool_sizes_mu;
static std::map<void *, size_t> g_pool_sizes;             // size of every live rhj_dev_alloc block (all contexts)

// hooks for rhj_query.hip (same shared library, separate translation unit)
int rhj_internal_use_device(rhj_ctx *ctx) { return use_device(ctx); }
hipStream_t rhj_internal_stream(rhj_ctx *ctx) { return ctx->stream; }
int rhj_internal_fail(rhj_ctx *ctx, int code, const char *msg) { return fail(ctx, code, msg); }
void *rhj_internal_counters(rhj_ctx *ctx)
{
    ctx->counters_clean = false;                     // the caller is about to use them as scratch
    return ensure(ctx, ctx->counters, 64) == RHJ_OK ? ctx->counters.p : nullptr;
}

// =================================================================================================
// C-ABI
// =================================================================================================
extern "C" {

int rhj_abi_version(void) { return RHJ_ABI_VERSION; }

// the bijection joins partition by (rhj_kernels.hip mix64 / unmix64: splitmix64's finaliser and its inverse)
uint64_t rhj_mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

uint64_t rhj_unmix64(uint64_t x)
{
    x ^= (x >> 31) ^ (x >> 62);
    x *= 0x319642B2D24D8EC3ULL;
    x ^= (x >> 27) ^ (x >> 54);
    x *= 0x96DE1B173F119089ULL;
    x ^= (x >> 30) ^ (x >> 60);
    return x - 0x9E3779B97F4A7C15ULL;
}

int rhj_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

void rhj_default_opts(rhj_opts *o)
{
    if (!o) return;
    o->passes = -1;
    o->bits1 = 0;
    o->bits2 = 0;
    o->probe_split = 0;
}

int rhj_plan(uint64_t nR, uint64_t nS, const rhj_opts *in, rhj_opts *resolved)
{
    if (!resolved) return RHJ_E_INVALID;
    return resolve_plan(nR, nS, in, resolved);
}

const char *rhj_last_error(const rhj_ctx *ctx) { return ctx ? ctx->err.c_str() : g_last_error.c_str(); }

int rhj_init(int device, rhj_ctx **out_ctx)
{
    if (!out_ctx) return fail(nullptr, RHJ_E_INVALID, "out_ctx is null");
    *out_ctx = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return fail(nullptr, RHJ_E_NODEVICE, "no HIP device available");
    }
    if (device < 0 || device >= n) return fail(nullptr, RHJ_E_INVALID, "device index out of range");
    rhj_ctx *ctx = new rhj_ctx();
    ctx->device = device;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) {
        (void)hipGetLastError();
        delete ctx;
        return fail(nullptr, RHJ_E_HIP, "cannot create a stream on the device");
    }
    ctx->stream = ctx->own_stream;
    *out_ctx = ctx;
    return RHJ_OK;
}

int rhj_release_workspace(rhj_ctx *ctx)
{
    RHJCHK(use_device(ctx));
    (void)hipStreamSynchronize(ctx->stream);
    DevBuf *all[] = {&ctx->in_R, &ctx->in_S, &ctx->part_R, &ctx->part_S, &ctx->part_tmp, &ctx->ps_R, &ctx->ps_S,
                     &ctx->ps_1, &ctx->seg0, &ctx->unit_start, &ctx->unit_hist, &ctx->unit_base, &ctx->seg0_b,
                     &ctx->unit_start_b, &ctx->unit_hist_b, &ctx->unit_base_b, &ctx->scan_tmp_b, &ctx->tasks,
                     &ctx->counters, &ctx->out_pairs, &ctx->small_out, &ctx->hist_tmp, &ctx->scan_tmp, &ctx->hist2,
                     &ctx->grp_rng, &ctx->unit_start2, &ctx->narrow_flag, &ctx->seg_rng, &ctx->tag_base,
                     &ctx->shard_ps[0], &ctx->shard_ps[1], &ctx->shard_mm, &ctx->shard_wide, &ctx->shard_peer_tab, &ctx->fuse_ctl, &ctx->sniff_tab, &ctx->hist2_b, &ctx->grp_rng_b,
                     &ctx->unit_start2_b, &ctx->ps_1_b, &ctx->part_tmp_b, &ctx->b_in[0], &ctx->b_out[0], &ctx->b_cnt[0],
                     &ctx->b_in[1], &ctx->b_out[1], &ctx->b_cnt[1]};
    ctx->fuse_clean = false;
    for (DevBuf *b : all) release(*b);
    ctx->small_hdr_clean = false;
    for (auto &b : ctx->free_blocks) {
        { std::lock_guard<std::mutex> lk(g_pool_sizes_mu); g_pool_sizes.erase(b.first); }
        (void)hipFree(b.first);
    }
    ctx->free_blocks.clear();
    ctx->free_bytes = 0;
    if (ctx->h_land) { (void)hipHostFree(ctx->h_land); ctx->h_land = nullptr; }
    stager_destroy(ctx);
    if (ctx->h_counts) { (void)hipHostFree(ctx->h_counts); ctx->h_counts = nullptr; }
    if (ctx->h_pub) { (void)hipHostFree(ctx->h_pub); ctx->h_pub = nullptr; ctx->h_pub_dev = nullptr; }
    for (hipEvent_t e : ctx->chunk_ev) (void)hipEventDestroy(e);
    ctx->chunk_ev.clear();
    return RHJ_OK;
}

void rhj_destroy(rhj_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)rhj_release_workspace(ctx);
    for (hipEvent_t ev : ctx->prof.pool) (void)hipEventDestroy(ev);
    for (int i = 0; i < 2; i++) if (ctx->up_ev[i]) (void)hipEventDestroy(ctx->up_ev[i]);
    if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
    if (ctx->aux_stream) (void)hipStreamDestroy(ctx->aux_stream);
    for (int i = 0; i < 2; i++) if (ctx->aux_ev[i]) (void)hipEventDestroy(ctx->aux_ev[i]);
    if (ctx->down_stream) (void)hipStreamDestroy(ctx->down_stream);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx->pool;
    for (int i = 0; i < 2; i++) {
        if (ctx->b_stage[i]) (void)hipHostFree(ctx->b_stage[i]);
        if (ctx->b_land[i]) (void)hipHostFree(ctx->b_land[i]);
        if (ctx->b_ev[i]) (void)hipEventDestroy(ctx->b_ev[i]);
    }
    delete ctx;
}

int rhj_set_stream(rhj_ctx *ctx, void *hip_stream)
{
    RHJCHK(use_device(ctx));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return RHJ_OK;
}

int rhj_set_option(rhj_ctx *ctx, const char *name, int64_t value)
{
    if (!ctx || !name) return fail(ctx, RHJ_E_INVALID, "rhj_set_option: null argument");
    const std::string n(name);
    if (n == "join.big_tables" && value >= -1 && value <= 1) { ctx->opt_big_tables = (int)value; return RHJ_OK; }
    if (n == "join.big_kernel" && (value == -1 || (value >= JK_BKT_BIG && value <= JK_LAST))) {
        ctx->opt_big_kernel = (int)value;
        return RHJ_OK;
    }
    if (n == "partition.narrow" && value >= -1 && value <= 2) {
        ctx->opt_narrow = (int)value;
        ctx->narrow_fail_streak = ctx->narrow_skip = 0;
        return RHJ_OK;
    }
    if (n == "partition.mix" && value >= -1 && value <= 1) { ctx->opt_mix = (int)value; return RHJ_OK; }
    if (n == "join.fused" && value >= -1 && value <= 1) { ctx->opt_fused = (int)value; return RHJ_OK; }
    if (n == "join.sniff" && value >= -1 && value <= 1) { ctx->opt_sniff = (int)value; return RHJ_OK; }
    return fail(ctx, RHJ_E_INVALID, "rhj_set_option: unknown option or value: " + n);
}

int rhj_get_info(rhj_ctx *ctx, const char *name, int64_t *value)
{
    if (!ctx || !name || !value) return fail(ctx, RHJ_E_INVALID, "rhj_get_info: null argument");
    const std::string n(name);
    if (n == "last.narrow") { *value = ctx->cur_narrow; return RHJ_OK; }
    if (n == "last.join_kernel") { *value = ctx->last_join_kind; return RHJ_OK; }
    if (n == "last.pipelined") { *value = ctx->last_pipelined; return RHJ_OK; }
    if (n == "last.max_part_R") { *value = (int64_t)ctx->last_max_part[0]; return RHJ_OK; }
    if (n == "last.max_part_S") { *value = (int64_t)ctx->last_max_part[1]; return RHJ_OK; }
    if (n == "partition.mix") { *value = join_mix(ctx) != MIX_NONE; return RHJ_OK; }
    return fail(ctx, RHJ_E_INVALID, "rhj_get_info: unknown name: " + n);
}

int rhj_set_profiling(rhj_ctx *ctx, int enabled)
{
    if (!ctx) return RHJ_E_INVALID;
    ctx->prof.on = enabled != 0;
    ctx->prof.keep = false;
    prof_reset(ctx);                       // (also the start of an accumulating series)
    ctx->prof.keep = enabled == 2;
    return RHJ_OK;
}

int rhj_get_timings(rhj_ctx *ctx, rhj_timings *out)
{
    if (!ctx || !out) return RHJ_E_INVALID;
    RHJCHK(use_device(ctx));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    rhj_timings t = ctx->last;
    for (int k = 0; k < RHJ_K_COUNT; k++) { t.ms[k] = 0; t.launches[k] = 0; }
    Prof &p = ctx->prof;
    for (size_t i = 0; i < p.used; i++) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, p.pool[2 * i], p.pool[2 * i + 1]) != hipSuccess) { (void)hipGetLastError(); continue; }
        t.ms[p.kinds[i]] += ms;
        t.launches[p.kinds[i]]++;
    }
    if (p.used) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, p.pool[0], p.pool[2 * (p.used - 1) + 1]) == hipSuccess) t.total_ms = ms;
    }
    *out = t;
    return RHJ_OK;
}

int rhj_get_launch_timings(rhj_ctx *ctx, int32_t *kinds, double *ms, uint32_t capacity, uint32_t *n)
{
    if (!ctx || !n || (capacity && (!kinds || !ms))) return RHJ_E_INVALID;
    RHJCHK(use_device(ctx));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    Prof &p = ctx->prof;
    *n = (uint32_t)p.used;
    for (size_t i = 0; i < p.used && i < capacity; i++) {
        float t = 0;
        if (hipEventElapsedTime(&t, p.pool[2 * i], p.pool[2 * i + 1]) != hipSuccess) { (void)hipGetLastError(); t = 0; }
        kinds[i] = p.kinds[i];
        ms[i] = t;
    }
    return RHJ_OK;
}

int rhj_sync(rhj_ctx *ctx)
{
    RHJCHK(use_device(ctx));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return RHJ_OK;
}

int rhj_reserve(rhj_ctx *ctx, uint64_t nR, uint64_t nS, const rhj_opts *opts)
{
    RHJCHK(use_device(ctx));
    rhj_opts plan;
    if (resolve_plan(nR, nS, opts, &plan, false, default_join_kernels(ctx)) != RHJ_OK) return fail(ctx, RHJ_E_INVALID, "bad rhj_opts");
    if (plan.passes >= 1) {
        RHJCHK(ensure(ctx, ctx->part_R, (size_t)(nR ? nR : 1) * 16));
        RHJCHK(ensure(ctx, ctx->part_S, (size_t)(nS ? nS : 1) * 16));
    }
    if (plan.passes == 2) RHJCHK(ensure(ctx, ctx->part_tmp, (size_t)((nR > nS ? nR : nS) + 1) * 16));
    return RHJ_OK;
}

int rhj_join_dev(rhj_ctx *ctx, const rhj_tuple *d_R, uint64_t nR, const rhj_tuple *d_S, uint64_t nS,
                 const rhj_opts *opts, rhj_pair *d_out, uint64_t out_capacity, uint64_t *out_count)
{
    RHJCHK(use_device(ctx));
    if (!out_count) return fail(ctx, RHJ_E_INVALID, "out_count is null");
    *out_count = 0;
    prof_reset(ctx);
    if (nR == 0 || nS == 0) return RHJ_OK;            // nothing to schedule (Result.cpp:101 never fires)
    if (!d_R || !d_S) return fail(ctx, RHJ_E_INVALID, "null input relation");
    rhj_opts plan;
    if (resolve_plan(nR, nS, opts, &plan, true, default_join_kernels(ctx)) != RHJ_OK) return fail(ctx, RHJ_E_INVALID, "bad rhj_opts");
    RHJCHK(partition_and_join(ctx, d_R, nR, d_S, nS, plan, d_out, d_out ? out_capacity : 0, (u64 *)out_count));
    if (d_out && *out_count > out_capacity) return fail(ctx, RHJ_E_OVERFLOW, "result buffer too small");
    return RHJ_OK;
}

namespace {

// A large result page is fresh mmap'd memory: every 4 KiB (or, with transparent huge pages, 2 MiB) page must
// be faulted in before the device-to-host copy can land, and a DMA that takes those faults itself runs at
// ~18 GB/s.  PagePrefault allocates the page for the optimistic size early and touches it from a few helper
// threads WHILE the inputs are copied in and the kernels run; the final copy then runs at the PCIe rate.
struct PagePrefault {
    unsigned char *page = nullptr;
    size_t pairs = 0;
    std::vector<std::thread> workers;
    std::atomic<bool> stop{false};         // drop(): the result turned out empty / too large -- stop touching at once
    void start(size_t npairs)
    {
        const size_t bytes = 8 + npairs * 16;
        if (npairs * 16 < ((size_t)64 << 20)) return;             // small pages: not worth threads
        page = (unsigned char *)malloc(bytes);
        if (!page) return;
        pairs = npairs;
        const uintptr_t lo = ((uintptr_t)page + ((uintptr_t)2 << 20) - 1) & ~(((uintptr_t)2 << 20) - 1);
        const uintptr_t hi = ((uintptr_t)page + bytes) & ~(((uintptr_t)2 << 20) - 1);
        if (hi > lo) (void)madvise((void *)lo, hi - lo, MADV_HUGEPAGE);
        const int nt = 4;
        for (int t = 0; t < nt; t++)
            workers.emplace_back([=] {
                const size_t from = bytes / nt * t, to = t + 1 == nt ? bytes : bytes / nt * (t + 1);
                for (size_t o = from; o < to && !stop.load(std::memory_order_relaxed); o += 4096)
                    ((volatile unsigned char *)page)[o] = 0;
            });
    }
    void wait() { for (std::thread &w : workers) w.join(); workers.clear(); }
    void drop() { stop.store(true); wait(); free(page); page = nullptr; pairs = 0; }
    ~PagePrefault() { wait(); }
};

}  // namespace

namespace {

// rhj_join for small inputs (the 94 joins of small.work are <= 43 K tuples).  Every asynchronous operation
// costs 5-10 us of latency on this platform whatever its size (measured: 2 H2D + memset + kernel + 2 D2H = 48 us for a
// 3754 x 14368 join whose kernel runs ~10 us), so the sequence is cut to THREE operations and one host synchronisation:
//   H2D R, H2D S, ONE kernel (k_join_bkt DIRECT) that stores the pairs into HBM AND, the first MiB of them, straight into
//   pinned host memory, and whose last workgroup publishes the result count there and zeroes the device counters for the
//   next call.  One memcpy fills the exact-size page; what a larger result has beyond the first MiB is fetched from HBM
//   straight into the page (no second run: the pairs have the same positions in both places).
constexpr size_t LAND_BYTES = (size_t)1 << 20;

int join_small_host(rhj_ctx *ctx, const rhj_tuple *R, u64 nR, const rhj_tuple *S, u64 nS, void **out_page, u64 *out_count)
{
    static const bool trace = getenv("RHJ_TRACE_SMALL") != nullptr;       // tuning aid: host-side timeline on stderr
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<double, std::micro>(b - a).count();
    };
    const auto t0 = now();
    const u64 guess = (nR > nS ? nR : nS) + 1024, landcap = LAND_BYTES / 16;
    RHJCHK(ensure(ctx, ctx->in_R, (size_t)nR * 16));
    RHJCHK(ensure(ctx, ctx->in_S, (size_t)nS * 16));
    if (!ctx->h_land || !ctx->h_land_dev) {
        if (ctx->h_land) { (void)hipHostFree(ctx->h_land); ctx->h_land = nullptr; }      // an earlier attempt got the memory but no device address
        ctx->h_land_dev = nullptr;
        HIPCHK(ctx, hipHostMalloc((void **)&ctx->h_land, 64 + LAND_BYTES, hipHostMallocDefault));
        const hipError_t e = hipHostGetDevicePointer((void **)&ctx->h_land_dev, ctx->h_land, 0);
        if (e != hipSuccess || !ctx->h_land_dev) {
            (void)hipHostFree(ctx->h_land);
            ctx->h_land = nullptr;
            ctx->h_land_dev = nullptr;
            return fail(ctx, RHJ_E_HIP, std::string("hipHostGetDevicePointer: ") + hipGetErrorString(e));
        }
    }
    if (ctx->small_out.cap < 64 + (size_t)guess * 16) {
        RHJCHK(ensure(ctx, ctx->small_out, 64 + (size_t)guess * 16));
        ctx->small_hdr_clean = false;
    }
    HIPCHK(ctx, hipMemcpyAsync(ctx->in_R.p, R, (size_t)nR * 16, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ctx->in_S.p, S, (size_t)nS * 16, hipMemcpyHostToDevice, ctx->stream));
    ctx->last.passes = 0;
    ctx->last.bits1 = ctx->last.bits2 = 0;
    ctx->last.ntasks = 0;
    ctx->last_join_kind = -1;
    ctx->cur_narrow = 0;
    u64 count = 0, landed = 0;
    const u64 unpublished = ~0ull;                                            // the kernel's last workgroup overwrites it with the count
    for (int attempt = 0; attempt < 2; attempt++) {
        *(volatile u64 *)ctx->h_land = unpublished;
        // device buffer: {64-byte header: [0] count, [8] finished workgroups | pairs}; the first MiB of pairs and the count
        // land in pinned host memory as well, written by the kernel itself
        unsigned char *d_hdr = (unsigned char *)ctx->small_out.p;
        const u64 dcap = (ctx->small_out.cap - 64) / 16;
        if (!ctx->small_hdr_clean) HIPCHK(ctx, hipMemsetAsync(d_hdr, 0, 64, ctx->stream));
        ctx->small_hdr_clean = false;
        {
            Span s(ctx, RHJ_K_JOIN);
            launch_join_direct(ctx->stream, ctx->in_R.p, nR, ctx->in_S.p, nS, d_hdr + 64, dcap, (u64 *)d_hdr,
                               (u64 *)ctx->h_land_dev, (u32 *)(d_hdr + 8), ctx->h_land_dev + 64, landcap);
        }
        RHJCHK(check_launch(ctx, "direct join"));
        const auto t1 = now();
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        ctx->small_hdr_clean = true;                                      // the last workgroup zeroed the counters
        count = *(volatile const u64 *)ctx->h_land;
        if (count == unpublished) return fail(ctx, RHJ_E_HIP, "the direct join kernel did not publish its result count");
        landed = dcap < landcap ? dcap : landcap;
        if (trace) fprintf(stderr, "[small join %llu x %llu -> %llu] enqueue %.1f  sync %.1f us\n", (unsigned long long)nR,
                           (unsigned long long)nS, (unsigned long long)count, us(t0, t1), us(t1, now()));
        if (count <= dcap) break;                                         // every pair is in HBM (and the first MiB on the host)
        if (attempt == 1) return fail(ctx, RHJ_E_HIP, "result count changed between join phases");
        RHJCHK(ensure(ctx, ctx->small_out, 64 + (size_t)count * 16));     // more pairs than the buffer holds: exact size known now
        ctx->small_hdr_clean = false;
    }
    unsigned char *page = nullptr;
    if (count) {
        page = (unsigned char *)malloc(8 + (size_t)count * 16);
        if (!page) return fail(ctx, RHJ_E_NOMEM, "malloc of the result page failed");
        memset(page, 0, 8);                                               // bucket_info::next = nullptr (Result.h:14-17)
        const u64 got = count < landed ? count : landed;
        memcpy(page + 8, ctx->h_land + 64, (size_t)got * 16);
        if (count > got) {                                                // the rest straight from HBM into the page
            hipError_t e = hipMemcpyAsync(page + 8 + got * 16, (const unsigned char *)ctx->small_out.p + 64 + got * 16,
                                          (size_t)(count - got) * 16, hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
            if (e != hipSuccess) { free(page); return fail(ctx, RHJ_E_HIP, std::string("result copy: ") + hipGetErrorString(e)); }
        }
    }
    *out_page = page;                                                     // nullptr when nothing matched (Result::isEmpty)
    *out_count = count;
    return RHJ_OK;
}

}  // namespace

namespace {

// rhj_join for inputs of hundreds of MiB: PCIe is full duplex, so the call should cost about max(upload, download), not their
// sum.  R is uploaded and partitioned once; S goes through in CHUNKS -- R join S = union of R join S_i --: while chunk i+1 is
// being staged and uploaded, chunk i is partitioned (same radix plan) and joined against the partitioned R, its pairs landing
// behind the earlier chunks' in the pair buffer (the result counter simply runs on), and a downloader thread brings finished
// ranges of pairs home into the result page.  Returns RHJ_NOT_PIPELINED when the call should take the plain path instead
// (small inputs, a rowID that does not fit the narrow format, more pairs than the optimistic page holds).
constexpr int RHJ_NOT_PIPELINED = 1001;
constexpr int RHJ_NOT_PIPELINED_WIDE = 1002;           // ... because a rowID did not fit the narrow format: the repeat runs 16-byte
constexpr u64 PIPE_MIN_CHUNK = (u64)8 << 20;           // tuples per S chunk at least
constexpr int PIPE_MAX_CHUNKS = 16;

int join_host_pipelined(rhj_ctx *ctx, const rhj_tuple *R, u64 nR, const rhj_tuple *S, u64 nS, const rhj_opts &plan,
                        void **out_page, u64 *out_count)
{
    static const bool off = getenv("RHJ_NO_PIPELINE") != nullptr;                 // tuning aid: A/B against the plain path
    static const bool trace = getenv("RHJ_TRACE_JOIN") != nullptr;
    static const u64 max_chunks = env_u64("RHJ_PIPE_CHUNKS", 12, 2, PIPE_MAX_CHUNKS);   // tuning aid
    static const u64 min_chunk = env_u64("RHJ_PIPE_MIN_CHUNK", PIPE_MIN_CHUNK, 1 << 16, 1 << 26);   // (tests: pipelining at small sizes)
    int K = (int)(nS / min_chunk < max_chunks ? nS / min_chunk : max_chunks);
    if (off || plan.passes < 1 || nS < 4 * min_chunk || K < 2 || nR < min_chunk / 2) return RHJ_NOT_PIPELINED;
    const u64 chunk = ((nS + K - 1) / K + 4095) / 4096 * 4096;
    K = (int)((nS + chunk - 1) / chunk);
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<double, std::milli>(b - a).count();
    };
    const auto t0 = now();
    const u64 cap = (nR > nS ? nR : nS) + 1024;
    PagePrefault pre;
    pre.start((size_t)cap);
    if (!pre.page) return RHJ_NOT_PIPELINED;
    const int narrow = narrow_level(ctx, nR, chunk, plan);
    const int tb = plan.bits1 + (plan.passes == 2 ? plan.bits2 : 0);
    const size_t np = (size_t)1 << tb;
    auto setup = [&]() -> int {
        RHJCHK(ensure(ctx, ctx->in_R, (size_t)nR * 16));
        RHJCHK(ensure(ctx, ctx->in_S, (size_t)nS * 16));
        RHJCHK(ensure(ctx, ctx->part_R, (size_t)nR * 16));
        RHJCHK(ensure(ctx, ctx->part_S, (size_t)chunk * 16));
        RHJCHK(ensure(ctx, ctx->ps_R, (np + 1) * 8));
        RHJCHK(ensure(ctx, ctx->ps_S, (np + 1) * 8));
        RHJCHK(ensure(ctx, ctx->out_pairs, (size_t)cap * 16));
        RHJCHK(ensure(ctx, ctx->counters, 64));
        RHJCHK(ensure(ctx, ctx->narrow_flag, 64));
        if (!ctx->copy_stream) {
            HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
            for (int i = 0; i < 2; i++) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->up_ev[i], hipEventDisableTiming));
        }
        if (!ctx->down_stream) HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->down_stream, hipStreamNonBlocking));
        if (!ctx->h_counts) HIPCHK(ctx, hipHostMalloc((void **)&ctx->h_counts, (size_t)PIPE_MAX_CHUNKS * 128, hipHostMallocDefault));
        while ((int)ctx->chunk_ev.size() < K) {
            hipEvent_t e;
            HIPCHK(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
            ctx->chunk_ev.push_back(e);
        }
        return RHJ_OK;
    };
    int rc = setup();
    if (rc != RHJ_OK) { pre.drop(); return rc; }
    const u64 dcap = ctx->out_pairs.cap / 16 < pre.pairs ? ctx->out_pairs.cap / 16 : pre.pairs;
    unsigned char *page = pre.page;

    // downloader: finished ranges of pairs -> result page, on its own stream, while later chunks are still coming in
    std::mutex mu;
    std::condition_variable cv;
    std::vector<std::pair<u64, u64>> todo;                 // (first pair, pairs)
    bool closing = false, dl_failed = false;
    std::chrono::steady_clock::time_point t_faulted = t0, t_dl_first = t0;
    std::thread downloader([&] {
        (void)hipSetDevice(ctx->device);
        pre.wait();                                        // the helper threads still WRITE into the page while they fault it in
        t_faulted = now();
        for (;;) {
            std::pair<u64, u64> job;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return closing || !todo.empty(); });
                if (todo.empty()) return;
                job = todo.front();
                todo.erase(todo.begin());
            }
            hipError_t e = hipMemcpyAsync(page + 8 + job.first * 16, (const unsigned char *)ctx->out_pairs.p + job.first * 16,
                                          (size_t)job.second * 16, hipMemcpyDeviceToHost, ctx->down_stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->down_stream);
            if (e != hipSuccess) dl_failed = true;
            if (t_dl_first == t0) t_dl_first = now();
        }
    });
    auto finish_downloader = [&] {
        { std::lock_guard<std::mutex> lk(mu); closing = true; }
        cv.notify_all();
        downloader.join();
    };

    bool abandon = false, saw_wide = false;                // wide rowID / more pairs than the page holds: take the plain path
    std::string errtext;
    u64 handed = 0, count = 0;
    int collected = 0;
    auto collect = [&](int j) {                            // chunk j's join has finished: hand its pairs to the downloader
        const u64 *h = (const u64 *)(ctx->h_counts + (size_t)j * 128);
        const u32 wide = *(const u32 *)(ctx->h_counts + (size_t)j * 128 + 64);
        if (wide || h[0] > dcap) { abandon = true; saw_wide = wide != 0; return; }
        if (h[5]) { errtext = "a partition's build side has " + std::to_string(h[5]) + " tuples (>= 2^32): use more radix bits"; return; }
        count = h[0];
        ctx->last.ntasks += (u32)(h[1] & 0xffffffffu);
        if (count > handed) {
            { std::lock_guard<std::mutex> lk(mu); todo.emplace_back(handed, count - handed); }
            cv.notify_all();
            handed = count;
        }
    };
    auto body = [&]() -> int {
        ctx->cur_narrow = narrow;
        ctx->last.passes = plan.passes;
        ctx->last.bits1 = plan.bits1;
        ctx->last.bits2 = plan.bits2;
        ctx->counters_clean = false;
        HIPCHK(ctx, hipMemsetAsync(ctx->narrow_flag.p, 0, 64, ctx->stream));
        HIPCHK(ctx, hipMemsetAsync(ctx->counters.p, 0, 64, ctx->stream));
        HIPCHK(ctx, hipEventRecord(ctx->up_ev[0], ctx->stream));           // (earlier work of this context may still read in_R / in_S)
        HIPCHK(ctx, hipStreamWaitEvent(ctx->copy_stream, ctx->up_ev[0], 0));
        RHJCHK(h2d_staged(ctx, ctx->in_R.p, R, (size_t)nR * 16, ctx->copy_stream));
        HIPCHK(ctx, hipEventRecord(ctx->up_ev[0], ctx->copy_stream));
        HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->up_ev[0], 0));
        RHJCHK(partition_side(ctx, ctx->in_R.p, nR, plan, narrow, ctx->part_R.p, (u64 *)ctx->ps_R.p));
        for (int i = 0; i < K && !abandon && errtext.empty(); i++) {
            const u64 off_i = (u64)i * chunk, n_i = nS - off_i < chunk ? nS - off_i : chunk;
            void *d_Si = (unsigned char *)ctx->in_S.p + off_i * 16;
            RHJCHK(h2d_staged(ctx, d_Si, S + off_i, (size_t)n_i * 16, ctx->copy_stream));
            HIPCHK(ctx, hipEventRecord(ctx->up_ev[1], ctx->copy_stream));
            HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->up_ev[1], 0));
            RHJCHK(partition_side(ctx, d_Si, n_i, plan, narrow, ctx->part_S.p, (u64 *)ctx->ps_S.p));
            u64 unused = 0;
            RHJCHK(join_phase_on(ctx, ctx->part_R.p, (const u64 *)ctx->ps_R.p, nR, ctx->part_S.p, (const u64 *)ctx->ps_S.p, n_i, np, tb,
                                 (u32)plan.probe_split, ctx->out_pairs.p, dcap, &unused, narrow != 0, nullptr, false, false, i > 0, true));
            HIPCHK(ctx, hipMemcpyAsync(ctx->h_counts + (size_t)i * 128, ctx->counters.p, 64, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(ctx, hipMemcpyAsync(ctx->h_counts + (size_t)i * 128 + 64, ctx->narrow_flag.p, 4, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(ctx, hipEventRecord(ctx->chunk_ev[i], ctx->stream));
            while (collected < i && hipEventQuery(ctx->chunk_ev[collected]) == hipSuccess) collect(collected++);
        }
        for (; collected < K && !abandon && errtext.empty(); collected++) {
            HIPCHK(ctx, hipEventSynchronize(ctx->chunk_ev[collected]));
            collect(collected);
        }
        return RHJ_OK;
    };
    rc = body();
    const auto t1 = now();
    finish_downloader();
    (void)hipStreamSynchronize(ctx->stream);               // nothing of this call is in flight when it returns
    (void)hipGetLastError();
    if (rc == RHJ_OK && !errtext.empty()) rc = fail(ctx, RHJ_E_INVALID, errtext);
    if (rc == RHJ_OK && dl_failed) rc = fail(ctx, RHJ_E_HIP, "result copy failed");
    if (rc != RHJ_OK || abandon || count == 0) {
        pre.page = page;                                   // (drop() frees it)
        pre.drop();
        if (rc != RHJ_OK) return rc;
        if (abandon && saw_wide) { narrow_note_fallback(ctx); return RHJ_NOT_PIPELINED_WIDE; }
        if (abandon) return RHJ_NOT_PIPELINED;
        narrow_note_done(ctx, plan, narrow != 0);
        *out_page = nullptr;                               // no match: head stays nullptr (Result::isEmpty)
        *out_count = 0;
        return RHJ_OK;
    }
    narrow_note_done(ctx, plan, narrow != 0);
    pre.page = nullptr;                                    // the caller's now
    ctx->last_pipelined = K;
    memset(page, 0, 8);                                    // bucket_info::next = nullptr (Result.h:14-17)
    if (trace)
        fprintf(stderr, "[rhj_join pipelined %llu x %llu -> %llu, %d chunks of S] enqueued %.1f  tail (last join + download) %.1f  total %.1f ms"
                        "  (result page faulted in at %.1f, first range home at %.1f)\n",
                (unsigned long long)nR, (unsigned long long)nS, (unsigned long long)count, K, ms(t0, t1), ms(t1, now()), ms(t0, now()),
                ms(t0, t_faulted), ms(t0, t_dl_first));
    *out_page = page;
    *out_count = count;
    return RHJ_OK;
}

}  // namespace

int rhj_join(rhj_ctx *ctx, const rhj_tuple *R, uint64_t nR, const rhj_tuple *S, uint64_t nS,
             const rhj_opts *opts, void **out_page, uint64_t *out_count)
{
    RHJCHK(use_device(ctx));
    if (!out_page || !out_count) return fail(ctx, RHJ_E_INVALID, "null output argument");
    *out_page = nullptr;
    *out_count = 0;
    prof_reset(ctx);
    ctx->last_pipelined = 0;
    if (nR == 0 || nS == 0) return RHJ_OK;
    if (!R || !S) return fail(ctx, RHJ_E_INVALID, "null input relation");
    rhj_opts plan;
    if (resolve_plan(nR, nS, opts, &plan, false, default_join_kernels(ctx)) != RHJ_OK) return fail(ctx, RHJ_E_INVALID, "bad rhj_opts");
    if (plan.passes == 0 && is_direct(ctx, 1, nR, nS)) return join_small_host(ctx, R, nR, S, nS, out_page, (u64 *)out_count);
    {
        const int prc = join_host_pipelined(ctx, R, nR, S, nS, plan, out_page, (u64 *)out_count);
        if (prc != RHJ_NOT_PIPELINED && prc != RHJ_NOT_PIPELINED_WIDE) return prc;
        prof_reset(ctx);
        if (prc == RHJ_NOT_PIPELINED_WIDE) {                // the pipelined attempt has met a wide rowID (and recorded it): the
            ctx->narrow_off_once = true;                    // plain path below goes straight to 16-byte tuples
        }
    }
    struct OffOnce { rhj_ctx *c; ~OffOnce() { c->narrow_off_once = false; } } off_once{ctx};
    // optimistic capacity: a foreign-key join yields about max(|R|,|S|) pairs; the count is exact
    // either way, and an overflow only repeats the join phase (partitions stay in the workspace)
    u64 cap = (nR > nS ? nR : nS) + 1024;
    static const bool trace = getenv("RHJ_TRACE_JOIN") != nullptr;        // tuning aid: host-side timeline on stderr
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<double, std::milli>(b - a).count();
    };
    const auto t0 = now();
    std::chrono::steady_clock::time_point tR = t0, tS = t0, tK = t0;
    PagePrefault pre;
    pre.start((size_t)cap);                           // host page being faulted in while the GPU side proceeds
    int rc = RHJ_OK;
    u64 count = 0;
    auto body = [&]() -> int {
        RHJCHK(ensure(ctx, ctx->in_R, (size_t)nR * 16));
        RHJCHK(ensure(ctx, ctx->in_S, (size_t)nS * 16));
        RHJCHK(ensure(ctx, ctx->out_pairs, (size_t)cap * 16));
        u64 dcap = ctx->out_pairs.cap / 16;
        // uploads on their own stream: R first; S while the kernels that partition R run (the host is busy staging S by then)
        if (!ctx->copy_stream) {
            HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
            for (int i = 0; i < 2; i++) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->up_ev[i], hipEventDisableTiming));
        }
        HIPCHK(ctx, hipEventRecord(ctx->up_ev[0], ctx->stream));          // (earlier work of this context may still read in_R)
        HIPCHK(ctx, hipStreamWaitEvent(ctx->copy_stream, ctx->up_ev[0], 0));
        RHJCHK(h2d_staged(ctx, ctx->in_R.p, R, (size_t)nR * 16, ctx->copy_stream));
        tR = now();
        HIPCHK(ctx, hipEventRecord(ctx->up_ev[0], ctx->copy_stream));
        HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->up_ev[0], 0));
        std::function<int()> upload_S = [&]() -> int {
            RHJCHK(h2d_staged(ctx, ctx->in_S.p, S, (size_t)nS * 16, ctx->copy_stream));
            HIPCHK(ctx, hipEventRecord(ctx->up_ev[1], ctx->copy_stream));
            HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->up_ev[1], 0));
            tS = now();
            return RHJ_OK;
        };
        RHJCHK(partition_and_join(ctx, ctx->in_R.p, nR, ctx->in_S.p, nS, plan, ctx->out_pairs.p, dcap, &count, &upload_S));
        if (count > dcap) {
            RHJCHK(ensure(ctx, ctx->out_pairs, (size_t)count * 16));
            dcap = ctx->out_pairs.cap / 16;
            u64 again = 0;
            RHJCHK(join_phase(ctx, ctx->out_pairs.p, dcap, &again));
            if (again != count) return fail(ctx, RHJ_E_HIP, "result count changed between join phases");
        }
        return RHJ_OK;
    };
    rc = body();
    tK = now();
    if (rc != RHJ_OK || count == 0) {                 // count == 0: head stays nullptr (Result::isEmpty)
        pre.drop();
        return rc;
    }
    unsigned char *page = nullptr;
    if (pre.page && count <= pre.pairs) {             // the pre-faulted block is large enough (it may be larger than
        pre.wait();                                   // needed: the block is the caller's to free() either way)
        page = pre.page;
        pre.page = nullptr;
    } else {
        pre.drop();
        page = (unsigned char *)malloc(8 + (size_t)count * 16);
        if (!page) return fail(ctx, RHJ_E_NOMEM, "malloc of the result page failed");
        if ((size_t)count * 16 >= ((size_t)64 << 20)) {
            const uintptr_t lo = ((uintptr_t)page + ((uintptr_t)2 << 20) - 1) & ~(((uintptr_t)2 << 20) - 1);
            const uintptr_t hi = ((uintptr_t)page + 8 + (size_t)count * 16) & ~(((uintptr_t)2 << 20) - 1);
            if (hi > lo) (void)madvise((void *)lo, hi - lo, MADV_HUGEPAGE);
        }
    }
    memset(page, 0, 8);                               // bucket_info::next = nullptr (Result.h:14-17)
    hipError_t e = hipMemcpyAsync(page + 8, ctx->out_pairs.p, (size_t)count * 16, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { free(page); return fail(ctx, RHJ_E_HIP, std::string("result copy: ") + hipGetErrorString(e)); }
    if (trace)
        fprintf(stderr, "[rhj_join %llu x %llu -> %llu] R staged %.1f  S staged %.1f  kernels + count %.1f  result copy %.1f  total %.1f ms\n",
                (unsigned long long)nR, (unsigned long long)nS, (unsigned long long)count, ms(t0, tR), ms(tR, tS), ms(tS, tK),
                ms(tK, now()), ms(t0, now()));
    *out_page = page;
    *out_count = count;
    return RHJ_OK;
}

// ---- several small joins per launch (MainScheduler.cpp:6-30 / join.cpp:42-50: the reference keeps 8 queries in
// flight; their joins are so small -- <= 43 K tuples on small.work -- that launch, copy and synchronisation latencies are
// their whole cost) ------------------------------------------------------------------------------------------------------
namespace {

constexpr u32 BATCH_MAX = 16;                           // joins per launch (grid.y)
constexpr size_t BATCH_PIECE = (size_t)256 << 10;       // host copies are cut into pieces of this size for the helper threads

int ensure_pinned(rhj_ctx *ctx, unsigned char **p, size_t *cap, size_t bytes, unsigned char **dev = nullptr)
{
    if (*cap >= bytes) return RHJ_OK;
    if (*p) { (void)hipHostFree(*p); *p = nullptr; *cap = 0; }
    size_t want = (size_t)8 << 20;
    while (want < bytes) want <<= 1;
    HIPCHK(ctx, hipHostMalloc((void **)p, want, hipHostMallocDefault));
    *cap = want;
    if (dev) {
        if (hipHostGetDevicePointer((void **)dev, *p, 0) != hipSuccess || !*dev) {
            (void)hipGetLastError();
            (void)hipHostFree(*p);
            *p = nullptr; *cap = 0;
            return fail(ctx, RHJ_E_HIP, "no device address for pinned host memory");
        }
    }
    return RHJ_OK;
}

void add_pieces(std::vector<CopySeg> &v, void *dst, const void *src, size_t bytes)
{
    for (size_t o = 0; o < bytes; o += BATCH_PIECE)
        v.push_back(CopySeg{(char *)dst + o, (const char *)src + o, bytes - o < BATCH_PIECE ? bytes - o : BATCH_PIECE});
}

// One group of <= BATCH_MAX direct joins = ONE staged upload, ONE launch, ONE wait.  A context has two slots (staging buffer,
// device buffers, landing zone, counters, event), so that the host work of a group -- staging its inputs, filling its
// result pages -- overlaps the GPU work of its neighbour:   stage(g+1) | GPU(g)   and   pages(g) | GPU(g+1).
struct BatchSlot {
    u32 k = 0;
    const u32 *idx = nullptr;
    size_t in_bytes = 0, out_off[BATCH_MAX] = {0}, land_off[BATCH_MAX] = {0};
    u64 dcap[BATCH_MAX] = {0}, lcap[BATCH_MAX] = {0};
    u32 max_blocks = 0;
};

int batch_stage(rhj_ctx *ctx, int sl, BatchSlot &b, const rhj_join_desc *J, const u32 *idx, u32 k)
{
    const size_t A = 256;
    auto up = [&](size_t x) { return (x + A - 1) & ~(A - 1); };
    size_t in_off[BATCH_MAX][2];
    size_t in_bytes = up(BATCH_MAX * sizeof(BatchJoinDesc)), out_bytes = 0, land_bytes = 128;
    b.k = k;
    b.idx = idx;
    for (u32 i = 0; i < k; i++) {
        const rhj_join_desc &j = J[idx[i]];
        in_off[i][0] = in_bytes; in_bytes += up((size_t)j.nR * 16);
        in_off[i][1] = in_bytes; in_bytes += up((size_t)j.nS * 16);
        const u64 guess = (j.nR > j.nS ? j.nR : j.nS) + 1024;
        b.lcap[i] = guess;                                  // what a foreign-key join yields lands in pinned memory ...
        b.dcap[i] = guess * 32;                             // ... a many-to-many result still fits the device buffer
        b.out_off[i] = out_bytes; out_bytes += up((size_t)b.dcap[i] * 16);
        b.land_off[i] = land_bytes; land_bytes += up((size_t)b.lcap[i] * 16);
    }
    b.in_bytes = in_bytes;
    RHJCHK(ensure_pinned(ctx, &ctx->b_stage[sl], &ctx->b_stage_cap[sl], in_bytes));
    RHJCHK(ensure_pinned(ctx, &ctx->b_land[sl], &ctx->b_land_cap[sl], land_bytes, &ctx->b_land_dev[sl]));
    RHJCHK(ensure(ctx, ctx->b_in[sl], in_bytes));
    RHJCHK(ensure(ctx, ctx->b_out[sl], out_bytes));
    if (!ctx->b_cnt[sl].p) {
        RHJCHK(ensure(ctx, ctx->b_cnt[sl], BATCH_MAX * 16));
        HIPCHK(ctx, hipMemsetAsync(ctx->b_cnt[sl].p, 0, BATCH_MAX * 16, ctx->stream));  // the kernels leave it zeroed from here on
    }
    if (!ctx->b_ev[sl]) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->b_ev[sl], hipEventDisableTiming));
    BatchJoinDesc *desc = (BatchJoinDesc *)ctx->b_stage[sl];
    volatile u64 *hcount = (volatile u64 *)ctx->b_land[sl];
    std::vector<CopySeg> segs;
    const u32 tile = join_direct_tile();
    b.max_blocks = 0;
    for (u32 i = 0; i < k; i++) {
        const rhj_join_desc &j = J[idx[i]];
        BatchJoinDesc &d = desc[i];
        d.R = (const unsigned char *)ctx->b_in[sl].p + in_off[i][0];
        d.S = (const unsigned char *)ctx->b_in[sl].p + in_off[i][1];
        d.out = (unsigned char *)ctx->b_out[sl].p + b.out_off[i];
        d.cap = b.dcap[i];
        d.count = (u64 *)((unsigned char *)ctx->b_cnt[sl].p + (size_t)i * 16);
        d.done = (u32 *)((unsigned char *)ctx->b_cnt[sl].p + (size_t)i * 16 + 8);
        d.build_is_S = j.nR >= j.nS + (j.nS >> build_tie_shift()) ? 1u : 0u;      // JobScheduler.cpp:187, near ties: R (build_on_S)
        d.nb = (u32)(d.build_is_S ? j.nS : j.nR);
        d.np = (u32)(d.build_is_S ? j.nR : j.nS);
        d.split = tile;
        d.nblocks = (d.np + tile - 1) / tile;
        d.pad = 0;
        d.host_count = (u64 *)ctx->b_land_dev[sl] + i;
        d.host_out = ctx->b_land_dev[sl] + b.land_off[i];
        d.host_cap = b.lcap[i];
        b.max_blocks = d.nblocks > b.max_blocks ? d.nblocks : b.max_blocks;
        hcount[i] = ~0ull;
        add_pieces(segs, ctx->b_stage[sl] + in_off[i][0], j.R, (size_t)j.nR * 16);
        add_pieces(segs, ctx->b_stage[sl] + in_off[i][1], j.S, (size_t)j.nS * 16);
    }
    ctx->pool->run(segs);
    return RHJ_OK;
}

int batch_launch(rhj_ctx *ctx, int sl, const BatchSlot &b)
{
    HIPCHK(ctx, hipMemcpyAsync(ctx->b_in[sl].p, ctx->b_stage[sl], b.in_bytes, hipMemcpyHostToDevice, ctx->stream));
    {
        Span s(ctx, RHJ_K_JOIN);
        launch_join_batch(ctx->stream, (const BatchJoinDesc *)ctx->b_in[sl].p, b.k, b.max_blocks);
    }
    RHJCHK(check_launch(ctx, "batched direct join"));
    HIPCHK(ctx, hipEventRecord(ctx->b_ev[sl], ctx->stream));
    return RHJ_OK;
}

// the group's launch has finished (the caller waited for its event): counts, pages.  retry: joins whose result outgrew the
// device buffer (they take the single-join path afterwards)
int batch_finish(rhj_ctx *ctx, int sl, const BatchSlot &b, void **pages, uint64_t *counts, std::vector<u32> &retry)
{
    volatile u64 *hcount = (volatile u64 *)ctx->b_land[sl];
    std::vector<CopySeg> segs, touch;
    struct Tail { void *dst; const void *src; size_t bytes; };
    std::vector<Tail> tails;                                   // what lies beyond a join's landing zone: fetched from HBM
    for (u32 i = 0; i < b.k; i++) {
        const u64 count = hcount[i];
        if (count == ~0ull) {
            (void)hipMemsetAsync(ctx->b_cnt[sl].p, 0, BATCH_MAX * 16, ctx->stream);
            return fail(ctx, RHJ_E_HIP, "a batched join did not publish its result count");
        }
        pages[b.idx[i]] = nullptr;
        counts[b.idx[i]] = count;
        if (count == 0) continue;                            // head stays nullptr (Result::isEmpty)
        if (count > b.dcap[i]) { retry.push_back(b.idx[i]); continue; }
        unsigned char *page = (unsigned char *)malloc(8 + (size_t)count * 16);
        if (!page) return fail(ctx, RHJ_E_NOMEM, "malloc of a result page failed");
        memset(page, 0, 8);                                  // bucket_info::next = nullptr (Result.h:14-17)
        pages[b.idx[i]] = page;
        const u64 got = count < b.lcap[i] ? count : b.lcap[i];
        add_pieces(segs, page + 8, ctx->b_land[sl] + b.land_off[i], (size_t)got * 16);
        if (count > got) {                                   // beyond the landing zone: straight from HBM into the page
            unsigned char *dst = page + 8 + got * 16;
            const size_t bytes = (size_t)(count - got) * 16;
            tails.push_back(Tail{dst, (const unsigned char *)ctx->b_out[sl].p + b.out_off[i] + got * 16, bytes});
            if (bytes >= ((size_t)1 << 20)) {
                // a DMA into fresh pages takes the page faults itself and runs at a third of the wire rate: fault the range in
                // first, from the helper threads, in huge pages where the kernel grants them
                const uintptr_t lo = ((uintptr_t)dst + ((uintptr_t)2 << 20) - 1) & ~(((uintptr_t)2 << 20) - 1);
                const uintptr_t hi = ((uintptr_t)dst + bytes) & ~(((uintptr_t)2 << 20) - 1);
                if (hi > lo) (void)madvise((void *)lo, hi - lo, MADV_HUGEPAGE);
                for (size_t o = 0; o < bytes; o += BATCH_PIECE)
                    touch.push_back(CopySeg{dst + o, nullptr, bytes - o < BATCH_PIECE ? bytes - o : BATCH_PIECE});
            }
        }
    }
    ctx->pool->run(touch);
    if (!tails.empty()) {
        // on its own stream: the next group's kernel is already running on ctx->stream (this group's has finished: its event)
        if (!ctx->down_stream) HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->down_stream, hipStreamNonBlocking));
        for (const Tail &t : tails) HIPCHK(ctx, hipMemcpyAsync(t.dst, t.src, t.bytes, hipMemcpyDeviceToHost, ctx->down_stream));
    }
    ctx->pool->run(segs);                                    // the landing zones -> pages, while the tails travel
    if (!tails.empty()) HIPCHK(ctx, hipStreamSynchronize(ctx->down_stream));
    return RHJ_OK;
}

}  // namespace

int rhj_join_batch(rhj_ctx *ctx, uint32_t n, const rhj_join_desc *joins, void **out_pages, uint64_t *out_counts)
{
    RHJCHK(use_device(ctx));
    if (n && (!joins || !out_pages || !out_counts)) return fail(ctx, RHJ_E_INVALID, "rhj_join_batch: null argument");
    prof_reset(ctx);
    ctx->last_pipelined = 0;
    for (u32 i = 0; i < n; i++) { out_pages[i] = nullptr; out_counts[i] = 0; }
    auto free_all = [&]() { for (u32 i = 0; i < n; i++) { free(out_pages[i]); out_pages[i] = nullptr; out_counts[i] = 0; } };
    std::vector<u32> small, single, retry;
    for (u32 i = 0; i < n; i++) {
        const rhj_join_desc &j = joins[i];
        if (j.nR == 0 || j.nS == 0) continue;
        if (!j.R || !j.S) return fail(ctx, RHJ_E_INVALID, "rhj_join_batch: null input relation");
        rhj_opts plan;
        if (resolve_plan(j.nR, j.nS, nullptr, &plan, false, default_join_kernels(ctx)) != RHJ_OK) return fail(ctx, RHJ_E_INVALID, "bad rhj_opts");
        (plan.passes == 0 && is_direct(ctx, 1, j.nR, j.nS) ? small : single).push_back(i);
    }
    if (!small.empty()) {
        if (!ctx->pool) ctx->pool = new CopyPool((int)env_u64("RHJ_BATCH_THREADS", 3, 0, 15));
        const size_t G = (small.size() + BATCH_MAX - 1) / BATCH_MAX;
        BatchSlot slot[2];
        auto group_k = [&](size_t g) { return (u32)(small.size() - g * BATCH_MAX < BATCH_MAX ? small.size() - g * BATCH_MAX : BATCH_MAX); };
        int rc = batch_stage(ctx, 0, slot[0], joins, &small[0], group_k(0));
        if (rc == RHJ_OK) rc = batch_launch(ctx, 0, slot[0]);
        for (size_t g = 0; g < G && rc == RHJ_OK; g++) {
            const int cur = (int)(g & 1), nxt = cur ^ 1;
            if (g + 1 < G) rc = batch_stage(ctx, nxt, slot[nxt], joins, &small[(g + 1) * BATCH_MAX], group_k(g + 1));   // | GPU(g)
            if (rc == RHJ_OK && hipEventSynchronize(ctx->b_ev[cur]) != hipSuccess) rc = fail(ctx, RHJ_E_HIP, "batched join: event wait failed");
            if (rc == RHJ_OK && g + 1 < G) rc = batch_launch(ctx, nxt, slot[nxt]);
            if (rc == RHJ_OK) rc = batch_finish(ctx, cur, slot[cur], out_pages, out_counts, retry);                        // | GPU(g+1)
        }
        if (rc != RHJ_OK) { (void)hipStreamSynchronize(ctx->stream); free_all(); return rc; }
    }
    ctx->last.passes = 0;
    ctx->last.bits1 = ctx->last.bits2 = 0;
    ctx->last_join_kind = -1;
    ctx->cur_narrow = 0;
    single.insert(single.end(), retry.begin(), retry.end());
    for (u32 i : single) {                                   // too large for the one-launch path (or a result beyond 32x the guess)
        const rhj_join_desc &j = joins[i];
        const int rc = rhj_join(ctx, j.R, j.nR, j.S, j.nS, nullptr, &out_pages[i], &out_counts[i]);
        if (rc != RHJ_OK) { free_all(); return rc; }
    }
    return RHJ_OK;
}

// ---- stage entry points ---------------------------------------------------------------------------
int rhj_partition(rhj_ctx *ctx, const rhj_tuple *d_in, uint64_t n, int bits1, int bits2, rhj_tuple *d_out,
                  uint64_t *d_part_start)
{
    RHJCHK(use_device(ctx));
    if (bits1 < 1 || bits1 > PART_MAX_BITS || bits2 < 0 || bits2 > PART_MAX_BITS || !d_out || !d_part_start || (n && !d_in))
        return fail(ctx, RHJ_E_INVALID, "bad rhj_partition argument");
    prof_reset(ctx);
    return partition_relation(ctx, d_in, n, bits2 ? 2 : 1, bits1, bits2, d_out, (u64 *)d_part_start);
}

static int partition_at(rhj_ctx *ctx, const rhj_tuple *d_in, uint64_t n, int shift, int bits, rhj_tuple *d_out,
                        uint64_t *d_part_start, int mix, const char *who)
{
    RHJCHK(use_device(ctx));
    if (bits < 1 || bits > PART_MAX_BITS || shift < 0 || shift + bits > 64 || !d_out || !d_part_start || (n && !d_in))
        return fail(ctx, RHJ_E_INVALID, std::string("bad argument: ") + who);
    prof_reset(ctx);
    RHJCHK(ensure(ctx, ctx->seg0, 64));
    return run_pass(ctx, d_in, d_out, n, (const u64 *)ctx->seg0.p, 1, shift, bits, (u64 *)d_part_start, true, mix);
}

int rhj_partition_at(rhj_ctx *ctx, const rhj_tuple *d_in, uint64_t n, int shift, int bits, rhj_tuple *d_out,
                     uint64_t *d_part_start)
{
    return partition_at(ctx, d_in, n, shift, bits, d_out, d_part_start, MIX_NONE, "rhj_partition_at");
}

int rhj_owner_split(rhj_ctx *ctx, const rhj_tuple *d_in, uint64_t n, int shift, int bits, rhj_tuple *d_out,
                    uint64_t *d_class_start)
{
    return partition_at(ctx, d_in, n, shift, bits, d_out, d_class_start, MIX_DIGIT, "rhj_owner_split");
}

static int histogram_of(rhj_ctx *ctx, const rhj_tuple *d_rel, uint64_t n, int shift, int bits, uint64_t *d_hist, int mix,
                        const char *who)
{
    RHJCHK(use_device(ctx));
    if (bits < 1 || bits > PART_MAX_BITS || shift < 0 || shift + bits > 64 || !d_hist || (n && !d_rel))
        return fail(ctx, RHJ_E_INVALID, std::string("bad argument: ") + who);
    prof_reset(ctx);
    const size_t nbins = (size_t)1 << bits;
    RHJCHK(ensure(ctx, ctx->seg0, 64));
    RHJCHK(ensure(ctx, ctx->hist_tmp, (nbins + 1) * 8));
    u64 *seg0 = (u64 *)ctx->seg0.p;
    PassGeom g = make_geom(n, 1, shift, bits);
    g.mix = mix;
    RHJCHK(ensure(ctx, ctx->unit_start, 8));
    RHJCHK(ensure(ctx, ctx->unit_hist, (size_t)g.max_units * nbins * 4));
    RHJCHK(ensure(ctx, ctx->unit_base, (size_t)g.max_units * nbins * 8));
    RHJCHK(ensure(ctx, ctx->scan_tmp, scan_tmp_bytes(bits)));
    {
        Span s(ctx, RHJ_K_AUX);
        launch_init_single_segment(ctx->stream, n, g.L, seg0, (u32 *)ctx->unit_start.p);
    }
    {
        Span s(ctx, RHJ_K_HIST);
        launch_hist_units(ctx->stream, d_rel, g, seg0, (const u32 *)ctx->unit_start.p, (u32 *)ctx->unit_hist.p);
    }
    {
        Span s(ctx, RHJ_K_SCAN);
        launch_scan_units(ctx->stream, g, seg0, (const u32 *)ctx->unit_start.p, (const u32 *)ctx->unit_hist.p,
                          (u64 *)ctx->unit_base.p, (u64 *)ctx->hist_tmp.p, (u64 *)ctx->scan_tmp.p);
        launch_diff_hist(ctx->stream, (const u64 *)ctx->hist_tmp.p, nbins, (u64 *)d_hist);
    }
    return check_launch(ctx, who);
}

int rhj_histogram(rhj_ctx *ctx, const rhj_tuple *d_rel, uint64_t n, int shift, int bits, uint64_t *d_hist)
{
    return histogram_of(ctx, d_rel, n, shift, bits, d_hist, MIX_NONE, "rhj_histogram");
}

int rhj_owner_histogram(rhj_ctx *ctx, const rhj_tuple *d_rel, uint64_t n, int shift, int bits, uint64_t *d_hist)
{
    return histogram_of(ctx, d_rel, n, shift, bits, d_hist, MIX_DIGIT, "rhj_owner_histogram");
}

int rhj_prefix(rhj_ctx *ctx, const uint64_t *d_hist, uint64_t nbins, uint64_t *d_start)
{
    RHJCHK(use_device(ctx));
    if (!d_hist || !d_start || nbins == 0) return fail(ctx, RHJ_E_INVALID, "bad rhj_prefix argument");
    prof_reset(ctx);
    {
        Span s(ctx, RHJ_K_SCAN);
        launch_prefix(ctx->stream, (const u64 *)d_hist, nbins, (u64 *)d_start);
    }
    return check_launch(ctx, "rhj_prefix");
}

int rhj_bucket_join(rhj_ctx *ctx, const rhj_tuple *d_Rp, const uint64_t *d_startR, const rhj_tuple *d_Sp,
                    const uint64_t *d_startS, uint64_t nparts, int radix_bits, int probe_split, rhj_pair *d_out,
                    uint64_t out_capacity, uint64_t *out_count)
{
    RHJCHK(use_device(ctx));
    if (!d_startR || !d_startS || !out_count || nparts == 0 || radix_bits < 0 || radix_bits > 40 || probe_split < 0)
        return fail(ctx, RHJ_E_INVALID, "bad rhj_bucket_join argument");
    prof_reset(ctx);
    // sizes are the last boundary of each side
    u64 ends[2] = {0, 0}, begs[2] = {0, 0};
    HIPCHK(ctx, hipMemcpyAsync(&ends[0], d_startR + nparts, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(&ends[1], d_startS + nparts, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(&begs[0], d_startR, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(&begs[1], d_startS, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    *out_count = 0;
    if (ends[0] <= begs[0] || ends[1] <= begs[1]) return RHJ_OK;
    // the one-launch small-join path assumes boundaries {0, n}; the compact-table kernel trusts radix_bits: checked on the device
    RHJCHK(join_phase_on(ctx, d_Rp, (const u64 *)d_startR, ends[0], d_Sp, (const u64 *)d_startS, ends[1], nparts,
                         radix_bits, (u32)probe_split, d_out, out_capacity, (u64 *)out_count, false, nullptr,
                         begs[0] == 0 && begs[1] == 0, true));
    if (d_out && *out_count > out_capacity) return fail(ctx, RHJ_E_OVERFLOW, "result buffer too small");
    return RHJ_OK;
}


// ---- multi-GPU stage entry points ---------------------------------------------------------------------
// Sender: class histogram (+ rowID range) of a shard, then the class split into the narrow wire format; receiver: the fused
// two-pass partition of what arrived (sender segments -> sender tags) and the bucket join that resolves the tags.
namespace {

struct ShardTables { DevBuf *seg0, *unit_start, *unit_hist, *unit_base, *scan_tmp; };

ShardTables shard_tables(rhj_ctx *ctx, int side)
{
    if (side == 0) return {&ctx->seg0, &ctx->unit_start, &ctx->unit_hist, &ctx->unit_base, &ctx->scan_tmp};
    return {&ctx->seg0_b, &ctx->unit_start_b, &ctx->unit_hist_b, &ctx->unit_base_b, &ctx->scan_tmp_b};
}

}  // namespace

uint64_t rhj_narrow_key_offset(uint64_t n) { return narrow_k_offset(n); }
uint64_t rhj_narrow_bytes(uint64_t n) { return (narrow_k_offset(n) + n * 4 + 255) & ~(uint64_t)255; }

int rhj_shard_plan(uint64_t nR, uint64_t nS, const rhj_opts *in, rhj_opts *resolved)
{
    if (!resolved) return RHJ_E_INVALID;
    rhj_opts o;
    if (resolve_plan(nR, nS, in, &o, true, false) != RHJ_OK) return RHJ_E_INVALID;     // (the receiver's join: sender tags or 16-byte partitions)
    *resolved = o;
    if (o.passes != 2 || nR < NARROW_MIN_TUPLES || nS < NARROW_MIN_TUPLES || nR >= ((u64)1 << 32) || nS >= ((u64)1 << 32)) return 0;
    if (!narrow_fused_plan(o))                          // 17-18 bits: no sender-aligned pass-2 units, so only rowIDs that need no restoring
        return o.bits1 + o.bits2 > 16 && narrow_pass9_ok(o.bits1) && narrow_pass9_ok(o.bits2) ? RHJ_SHARD_PLAIN : 0;
    if (o.bits1 < tag_bits()) return 0;
    rhj_ctx probe;                                      // default options: which kernel would join partitions of this size
    const int tb = o.bits1 + o.bits2;
    const int kind = choose_join_kind(&probe, nR, nS, (u64)1 << tb, tb, false, false);
    if (kind == JK_BKT) return RHJ_SHARD_TAGGED;
    if (jk_is_ct(kind) && !jk_ct_narrow_only(kind)) return RHJ_SHARD_GLOBAL16;
    return 0;
}

int rhj_shard_stats(rhj_ctx *ctx, int side, const rhj_tuple *d_rel, uint64_t n, int shift, int bits, uint64_t *hist,
                    uint64_t *key_min, uint64_t *key_max)
{
    RHJCHK(use_device(ctx));
    if ((side != 0 && side != 1) || bits < 1 || bits > 8 || shift < 0 || shift + bits > 64 || !hist || (n && !d_rel))
        return fail(ctx, RHJ_E_INVALID, "bad rhj_shard_stats argument");
    prof_reset(ctx);
    const size_t nbins = (size_t)1 << bits;
    const ShardTables t = shard_tables(ctx, side);
    PassGeom g = make_geom(n, 1, shift, bits);
    g.mix = MIX_STORE;                                   // classes are bits of mix64(payload): see rhj.h
    RHJCHK(ensure(ctx, *t.seg0, 64));
    RHJCHK(ensure(ctx, *t.unit_start, 16));
    RHJCHK(ensure(ctx, *t.unit_hist, (size_t)g.max_units * nbins * 4));
    RHJCHK(ensure(ctx, *t.unit_base, (size_t)g.max_units * nbins * 8));
    RHJCHK(ensure(ctx, *t.scan_tmp, scan_tmp_bytes(bits)));
    RHJCHK(ensure(ctx, ctx->shard_ps[side], (nbins + 1) * 8));
    RHJCHK(ensure(ctx, ctx->shard_mm, 64));
    RHJCHK(ensure(ctx, ctx->shard_wide, 64));
    HIPCHK(ctx, hipMemsetAsync((u32 *)ctx->shard_wide.p + side, 0, 4, ctx->stream));
    u64 *mm = (u64 *)ctx->shard_mm.p + 2 * side;
    const u64 init[2] = {~0ull, 0ull};
    {
        Span s(ctx, RHJ_K_AUX);
        HIPCHK(ctx, hipMemcpyAsync(mm, init, 16, hipMemcpyHostToDevice, ctx->stream));
        launch_init_single_segment(ctx->stream, n, g.L, (u64 *)t.seg0->p, (u32 *)t.unit_start->p);
    }
    {
        Span s(ctx, RHJ_K_HIST);
        launch_hist_units(ctx->stream, d_rel, g, (const u64 *)t.seg0->p, (const u32 *)t.unit_start->p, (u32 *)t.unit_hist->p, mm);
    }
    {
        Span s(ctx, RHJ_K_SCAN);
        launch_scan_units(ctx->stream, g, (const u64 *)t.seg0->p, (const u32 *)t.unit_start->p, (const u32 *)t.unit_hist->p,
                          (u64 *)t.unit_base->p, (u64 *)ctx->shard_ps[side].p, (u64 *)t.scan_tmp->p);
    }
    RHJCHK(check_launch(ctx, "rhj_shard_stats"));
    std::vector<u64> ps(nbins + 1);
    u64 hmm[2] = {0, 0};
    HIPCHK(ctx, hipMemcpyAsync(ps.data(), ctx->shard_ps[side].p, (nbins + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(hmm, mm, 16, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (size_t b = 0; b < nbins; b++) hist[b] = n ? ps[b + 1] - ps[b] : 0;
    ctx->shard_ps_host[side] = ps;
    if (key_min) *key_min = n ? hmm[0] : 0;
    if (key_max) *key_max = n ? hmm[1] : 0;
    ctx->shard_n[side] = n;
    ctx->shard_kmin[side] = n ? hmm[0] : 0;
    ctx->shard_kmax[side] = n ? hmm[1] : 0;
    return RHJ_OK;
}

int rhj_shard_split(rhj_ctx *ctx, int side, const rhj_tuple *d_rel, uint64_t n, int shift, int bits, uint64_t key_base,
                    void *d_narrow_out, uint64_t *d_class_start)
{
    RHJCHK(use_device(ctx));
    if ((side != 0 && side != 1) || bits < 1 || bits > 8 || shift < 0 || shift + bits > 64 || !d_narrow_out || (n && !d_rel))
        return fail(ctx, RHJ_E_INVALID, "bad rhj_shard_split argument");
    if (ctx->shard_n[side] != n || !ctx->shard_ps[side].p)
        return fail(ctx, RHJ_E_INVALID, "rhj_shard_split: call rhj_shard_stats for this side and relation first");
    if (n && (key_base > ctx->shard_kmin[side] || ctx->shard_kmax[side] - key_base >= ((u64)1 << 32)))
        return fail(ctx, RHJ_E_INVALID, "rhj_shard_split: rowID - key_base must lie in [0, 2^32) for every tuple of the shard "
                                        "(rhj_shard_stats reports the range)");
    prof_reset(ctx);
    const size_t nbins = (size_t)1 << bits;
    const ShardTables t = shard_tables(ctx, side);
    PassGeom g = make_geom(n, 1, shift, bits);
    g.mix = MIX_STORE;                                   // the wire carries mix64(payload); the receiver never mixes again
    {
        // the range check above holds for the relation rhj_shard_stats saw; a DIFFERENT relation handed in here may still hold a
        // rowID whose offset from key_base does not fit 32 bits: the kernel raises this side's word of shard_wide (cleared by
        // rhj_shard_stats) and rhj_shard_join of this context reports it
        Span s(ctx, RHJ_K_SCATTER);
        launch_scatter_units_narrow(ctx->stream, d_rel, d_narrow_out, n, g, (const u64 *)t.seg0->p, (const u32 *)t.unit_start->p,
                                    (const u64 *)t.unit_base->p, (u32 *)ctx->shard_wide.p + side, key_base);
    }
    if (d_class_start)
        HIPCHK(ctx, hipMemcpyAsync(d_class_start, ctx->shard_ps[side].p, (nbins + 1) * 8, hipMemcpyDeviceToDevice, ctx->stream));
    return check_launch(ctx, "rhj_shard_split");
}

// The class split of rhj_shard_split written STRAIGHT INTO THE OWNERS' RECEIVE ARRAYS (peer-mapped HBM over xGMI; on one GPU:
// any device buffers), instead of into a local send buffer that an all-to-all then copies.
int rhj_shard_split_peer(rhj_ctx *ctx, int side, const rhj_tuple *d_rel, uint64_t n, int shift, int bits, uint64_t key_base,
                         const uint8_t *owner, const uint64_t *dst_class_start, void *const *peer_payloads, void *const *peer_rowids,
                         int nranks)
{
    RHJCHK(use_device(ctx));
    if ((side != 0 && side != 1) || bits < 1 || bits > 8 || shift < 0 || shift + bits > 64 || (n && !d_rel) || !owner || !dst_class_start ||
        !peer_payloads || !peer_rowids || nranks < 1 || nranks > seg_max())
        return fail(ctx, RHJ_E_INVALID, "bad rhj_shard_split_peer argument");
    const size_t nbins = (size_t)1 << bits;
    if (ctx->shard_n[side] != n || ctx->shard_ps_host[side].size() != nbins + 1)
        return fail(ctx, RHJ_E_INVALID, "rhj_shard_split_peer: call rhj_shard_stats for this side and relation first");
    if (n && (key_base > ctx->shard_kmin[side] || ctx->shard_kmax[side] - key_base >= ((u64)1 << 32)))
        return fail(ctx, RHJ_E_INVALID, "rhj_shard_split_peer: rowID - key_base must lie in [0, 2^32) for every tuple of the shard");
    for (size_t c = 0; c < nbins; c++)
        if (owner[c] >= nranks || !peer_payloads[owner[c]] || !peer_rowids[owner[c]])
            return fail(ctx, RHJ_E_INVALID, "rhj_shard_split_peer: a class is owned by a rank without receive arrays");
    prof_reset(ctx);
    const ShardTables t = shard_tables(ctx, side);
    PassGeom g = make_geom(n, 1, shift, bits);
    g.mix = MIX_STORE;
    // delta[c] = (index of this rank's class c in its owner's arrays) - (index it would have in a local send buffer)
    const size_t tab_bytes = nbins * 8 + nbins;
    RHJCHK(ensure(ctx, ctx->shard_peer_tab, 2 * ((tab_bytes + 255) & ~(size_t)255)));
    unsigned char *d_tab = (unsigned char *)ctx->shard_peer_tab.p + (size_t)side * ((tab_bytes + 255) & ~(size_t)255);
    std::vector<unsigned char> tab(tab_bytes);
    for (size_t c = 0; c < nbins; c++) {
        const u64 d = dst_class_start[c] - ctx->shard_ps_host[side][c];          // (mod 2^64: added to the cursors, wraps back)
        memcpy(&tab[c * 8], &d, 8);
        tab[nbins * 8 + c] = owner[c];
    }
    // (a synchronous copy: the table is a stack object of this call)
    HIPCHK(ctx, hipMemcpyAsync(d_tab, tab.data(), tab_bytes, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    {
        Span s(ctx, RHJ_K_SCATTER);
        launch_scatter_units_narrow_peer(ctx->stream, d_rel, g, (const u64 *)t.seg0->p, (const u32 *)t.unit_start->p,
                                         (const u64 *)t.unit_base->p, (u32 *)ctx->shard_wide.p + side, key_base, (const u64 *)d_tab,
                                         d_tab + nbins * 8, peer_payloads, peer_rowids, nranks);
    }
    return check_launch(ctx, "rhj_shard_split_peer");
}

// Peer mapping of HBM between the processes of one node (one process per GPU): the owner of a receive array exports a handle,
// the senders open it and pass the resulting pointer to rhj_shard_split_peer.  Thin wrappers over hipIpc*; UNVERIFIED on this
// pool (one GPU per box: there is no second process with a GPU to open a handle in).
int rhj_ipc_export(rhj_ctx *ctx, void *d_ptr, void *handle64)
{
    RHJCHK(use_device(ctx));
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "handle size");
    if (!d_ptr || !handle64) return fail(ctx, RHJ_E_INVALID, "bad rhj_ipc_export argument");
    HIPCHK(ctx, hipIpcGetMemHandle((hipIpcMemHandle_t *)handle64, d_ptr));
    return RHJ_OK;
}

int rhj_ipc_open(rhj_ctx *ctx, const void *handle64, void **d_ptr)
{
    RHJCHK(use_device(ctx));
    if (!d_ptr || !handle64) return fail(ctx, RHJ_E_INVALID, "bad rhj_ipc_open argument");
    hipIpcMemHandle_t h;
    memcpy(&h, handle64, sizeof h);
    HIPCHK(ctx, hipIpcOpenMemHandle(d_ptr, h, hipIpcMemLazyEnablePeerAccess));
    return RHJ_OK;
}

int rhj_ipc_close(rhj_ctx *ctx, void *d_ptr)
{
    RHJCHK(use_device(ctx));
    if (!d_ptr) return RHJ_OK;
    HIPCHK(ctx, hipIpcCloseMemHandle(d_ptr));
    return RHJ_OK;
}

int rhj_shard_partition(rhj_ctx *ctx, int side, const uint64_t *d_payloads, const uint32_t *d_rowids, uint64_t m, int nseg,
                        const uint64_t *seg_off, const uint64_t *row0, const rhj_opts *plan, int mode)
{
    RHJCHK(use_device(ctx));
    if ((side != 0 && side != 1) || !plan || plan->passes != 2 || !seg_off || !row0 || nseg < 1 || nseg > seg_max() ||
        seg_off[0] != 0 || seg_off[nseg] != m || (m && (!d_payloads || !d_rowids)) ||
        (mode != RHJ_SHARD_TAGGED && mode != RHJ_SHARD_GLOBAL16 && mode != RHJ_SHARD_PLAIN))
        return fail(ctx, RHJ_E_INVALID, "bad rhj_shard_partition argument");
    if (mode == RHJ_SHARD_PLAIN)
        for (int i = 0; i < nseg; i++)
            if (row0[i] != 0) return fail(ctx, RHJ_E_INVALID, "RHJ_SHARD_PLAIN: every rank must have split with key_base 0");
    const bool fused = narrow_fused_plan(*plan) && plan->bits1 >= tag_bits();
    const bool deep = !narrow_fused_plan(*plan) && plan->bits1 + plan->bits2 > 16 && narrow_pass9_ok(plan->bits1) && narrow_pass9_ok(plan->bits2);
    if (!(fused || (deep && mode == RHJ_SHARD_PLAIN)) || m >= ((u64)1 << 32))
        return fail(ctx, RHJ_E_INVALID, "rhj_shard_partition: not a plan of the narrow format for this mode (see rhj_shard_plan)");
    for (int i = 0; i < nseg; i++)
        if (seg_off[i] > seg_off[i + 1]) return fail(ctx, RHJ_E_INVALID, "rhj_shard_partition: segment offsets must not decrease");
    prof_reset(ctx);
    const int tb = plan->bits1 + plan->bits2;
    const size_t np = (size_t)1 << tb;
    DevBuf &part = side == 0 ? ctx->part_R : ctx->part_S;
    DevBuf &ps = side == 0 ? ctx->ps_R : ctx->ps_S;
    RHJCHK(ensure(ctx, part, (size_t)(m ? m : 1) * 16));
    RHJCHK(ensure(ctx, ps, (np + 1) * 8));
    RHJCHK(ensure(ctx, ctx->narrow_flag, 64));
    {
        Span s(ctx, RHJ_K_AUX);
        HIPCHK(ctx, hipMemsetAsync(ctx->narrow_flag.p, 0, 64, ctx->stream));
    }
    ctx->counters_clean = false;
    u64 bases[16] = {0};
    for (int i = 0; i < nseg; i++) bases[i] = row0[i];
    RHJCHK(ensure(ctx, ctx->tag_base, 2 * sizeof(bases)));
    u64 *d_bases = (u64 *)ctx->tag_base.p + 16 * side;
    HIPCHK(ctx, hipMemcpyAsync(d_bases, bases, sizeof(bases), hipMemcpyHostToDevice, ctx->stream));   // (pageable source: staged before the call returns)
    ctx->shard_sniffed[side] = false;
    if (m == 0) {                                       // nothing arrived: all boundaries 0
        HIPCHK(ctx, hipMemsetAsync(ps.p, 0, (np + 1) * 8, ctx->stream));
    } else {
        FusedIn in;
        in.P = (const u64 *)d_payloads;
        in.K = (const u32 *)d_rowids;
        in.nseg = nseg;
        in.seg_off = (const u64 *)seg_off;
        in.final_form = mode == RHJ_SHARD_TAGGED ? 1 : mode == RHJ_SHARD_GLOBAL16 ? 2 : 0;
        in.key_bases = d_bases;
        if (fused) {
            const bool sniff = sniff_on(ctx);                                   // (the received join values are sampled for duplicates too)
            if (sniff) RHJCHK(ensure(ctx, ctx->sniff_tab, (size_t)2 * SNIFF_SLOTS * 4));
            ctx->sniff_side = sniff ? side : -1;
            const int prc = partition_relation_fused(ctx, in, m, plan->bits1, plan->bits2, part.p, (u64 *)ps.p, 2);
            ctx->sniff_side = -1;
            RHJCHK(prc);
            ctx->shard_sniffed[side] = sniff;
        }
        else RHJCHK(partition_relation_narrow2(ctx, d_payloads, m, plan->bits1, plan->bits2, part.p, (u64 *)ps.p, (const u32 *)d_rowids));
    }
    ctx->shard_side_done[side] = true;
    ctx->shard_n[side] = m;
    ctx->shard_nseg = nseg;
    ctx->shard_mode[side] = mode;
    ctx->shard_plan = *plan;
    return RHJ_OK;
}

int rhj_shard_join(rhj_ctx *ctx, rhj_pair *d_out, uint64_t out_capacity, uint64_t *out_count)
{
    RHJCHK(use_device(ctx));
    if (!out_count) return fail(ctx, RHJ_E_INVALID, "bad rhj_shard_join argument");
    if (!ctx->shard_side_done[0] || !ctx->shard_side_done[1] || ctx->shard_mode[0] != ctx->shard_mode[1])
        return fail(ctx, RHJ_E_INVALID, "rhj_shard_join: rhj_shard_partition both sides (0 and 1) in the same mode first");
    *out_count = 0;
    prof_reset(ctx);
    const u64 mR = ctx->shard_n[0], mS = ctx->shard_n[1];
    if (mR == 0 || mS == 0) return RHJ_OK;
    const int mode = ctx->shard_mode[0];
    const int tb = ctx->shard_plan.bits1 + ctx->shard_plan.bits2;
    const bool narrow = mode != RHJ_SHARD_GLOBAL16;
    if (mode == RHJ_SHARD_TAGGED && choose_join_kind(ctx, mR, mS, (u64)1 << tb, tb, false, false) != JK_BKT)
        return fail(ctx, RHJ_E_INVALID, "rhj_shard_join: partitions this large need RHJ_SHARD_GLOBAL16 (see rhj_shard_plan)");
    ctx->cur_narrow = narrow ? 2 : 0;
    ctx->sniff_ready = ctx->shard_sniffed[0] && ctx->shard_sniffed[1];
    ctx->last.passes = 2;
    ctx->last.bits1 = ctx->shard_plan.bits1;
    ctx->last.bits2 = ctx->shard_plan.bits2;
    int rc = join_phase_on(ctx, ctx->part_R.p, (const u64 *)ctx->ps_R.p, mR, ctx->part_S.p, (const u64 *)ctx->ps_S.p, mS,
                           (u64)1 << tb, tb, (u32)ctx->shard_plan.probe_split, d_out, d_out ? out_capacity : 0, (u64 *)out_count, narrow,
                           mode == RHJ_SHARD_TAGGED ? (const u64 *)ctx->tag_base.p : nullptr, false, false, false, false, false);
    if (rc == RHJ_RETRY_WIDE) return fail(ctx, RHJ_E_HIP, "rhj_shard_join: unexpected wide-rowID flag");
    RHJCHK(rc);
    if (ctx->shard_wide.p) {                               // what this context's own rhj_shard_split calls met (the stream is idle here)
        u32 w[2] = {0, 0};
        HIPCHK(ctx, hipMemcpy(w, ctx->shard_wide.p, 8, hipMemcpyDeviceToHost));
        if (w[0] | w[1])
            return fail(ctx, RHJ_E_INVALID, "rhj_shard_split: a rowID - key_base did not fit 32 bits -- the relation differs from the "
                                            "one given to rhj_shard_stats for that side; what was sent is incomplete");
    }
    if (d_out && *out_count > out_capacity) return fail(ctx, RHJ_E_OVERFLOW, "result buffer too small");
    return RHJ_OK;
}

// ---- utilities --------------------------------------------------------------------------------------
static int reduce_to_host(rhj_ctx *ctx, uint64_t *host_out, void (*launch)(hipStream_t, const void *, u64, u64 *),
                          const void *d_in, u64 n)
{
    RHJCHK(ensure(ctx, ctx->counters, 64));
    ctx->counters_clean = false;
    u64 *d_sum = (u64 *)ctx->counters.p + 4;
    HIPCHK(ctx, hipMemsetAsync(d_sum, 0, 8, ctx->stream));
    if (n) {
        Span s(ctx, RHJ_K_AUX);
        launch(ctx->stream, d_in, n, d_sum);
    }
    RHJCHK(check_launch(ctx, "reduction"));
    HIPCHK(ctx, hipMemcpyAsync(host_out, d_sum, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return RHJ_OK;
}

int rhj_pairs_checksum_dev(rhj_ctx *ctx, const rhj_pair *d_pairs, uint64_t n, uint64_t *checksum)
{
    RHJCHK(use_device(ctx));
    if (!checksum || (n && !d_pairs)) return fail(ctx, RHJ_E_INVALID, "bad rhj_pairs_checksum_dev argument");
    return reduce_to_host(ctx, checksum, launch_checksum, d_pairs, n);
}

int rhj_expected_pkfk_dev(rhj_ctx *ctx, const rhj_tuple *d_S, uint64_t n, uint64_t *count, uint64_t *checksum)
{
    RHJCHK(use_device(ctx));
    if (!checksum || !count || (n && !d_S)) return fail(ctx, RHJ_E_INVALID, "bad rhj_expected_pkfk_dev argument");
    *count = n;
    return reduce_to_host(ctx, checksum, launch_expected_pkfk, d_S, n);
}

int rhj_generate_dev(rhj_ctx *ctx, int kind, rhj_tuple *d_out, uint64_t n, uint64_t row0, uint64_t D, uint64_t seed,
                     int theta_milli)
{
    RHJCHK(use_device(ctx));
    if (kind < 0 || kind > 4 || (n && !d_out) || (kind != 4 && D == 0) || (kind == 2 && theta_milli == 1000))
        return fail(ctx, RHJ_E_INVALID, "bad rhj_generate_dev argument");
    if (n) {
        Span s(ctx, RHJ_K_AUX);
        launch_generate(ctx->stream, kind, d_out, n, row0, D, seed, theta_milli / 1000.0);
    }
    return check_launch(ctx, "rhj_generate_dev");
}

int rhj_remap_keys_dev(rhj_ctx *ctx, rhj_tuple *d_rel, uint64_t n, int shift, uint64_t add)
{
    RHJCHK(use_device(ctx));
    if (shift < 0 || shift > 63 || (n && !d_rel)) return fail(ctx, RHJ_E_INVALID, "bad rhj_remap_keys_dev argument");
    if (n) {
        Span s(ctx, RHJ_K_AUX);
        launch_remap_keys(ctx->stream, d_rel, n, shift, add);
    }
    return check_launch(ctx, "rhj_remap_keys_dev");
}

// block sizes are rounded up (64 KiB steps, 1/8 steps above 8 MiB) so that released blocks fit later requests
static size_t pool_round(uint64_t bytes)
{
    size_t b = bytes ? (size_t)bytes : 16;
    size_t step = (size_t)64 << 10;
    while (step * 8 < b) step <<= 1;
    return (b + step - 1) / step * step;
}
constexpr size_t POOL_KEEP_BYTES = (size_t)8 << 30;       // released blocks kept per context at most

int rhj_dev_alloc(rhj_ctx *ctx, uint64_t bytes, void **d_ptr)
{
    RHJCHK(use_device(ctx));
    if (!d_ptr) return fail(ctx, RHJ_E_INVALID, "d_ptr is null");
    *d_ptr = nullptr;
    const size_t want = pool_round(bytes);
    size_t best = (size_t)-1;
    for (size_t i = 0; i < ctx->free_blocks.size(); i++)
        if (ctx->free_blocks[i].second == want) { best = i; break; }
    if (best != (size_t)-1) {
        *d_ptr = ctx->free_blocks[best].first;
        ctx->free_bytes -= want;
        ctx->free_blocks[best] = ctx->free_blocks.back();
        ctx->free_blocks.pop_back();
        return RHJ_OK;
    }
    hipError_t e = hipMalloc(d_ptr, want);
    if (e != hipSuccess && !ctx->free_blocks.empty()) {           // make room and try once more
        (void)hipGetLastError();
        for (auto &b : ctx->free_blocks) {
            { std::lock_guard<std::mutex> lk(g_pool_sizes_mu); g_pool_sizes.erase(b.first); }
            (void)hipFree(b.first);
        }
        ctx->free_blocks.clear();
        ctx->free_bytes = 0;
        e = hipMalloc(d_ptr, want);
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        *d_ptr = nullptr;
        return fail(ctx, RHJ_E_NOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
    }
    std::lock_guard<std::mutex> lk(g_pool_sizes_mu);
    g_pool_sizes[*d_ptr] = want;
    return RHJ_OK;
}

int rhj_dev_free(rhj_ctx *ctx, void *d_ptr)
{
    RHJCHK(use_device(ctx));
    if (!d_ptr) return RHJ_OK;
    size_t sz = 0;
    {
        std::lock_guard<std::mutex> lk(g_pool_sizes_mu);
        auto it = g_pool_sizes.find(d_ptr);
        if (it != g_pool_sizes.end()) sz = it->second;
    }
    if (sz && ctx->free_bytes + sz <= POOL_KEEP_BYTES && ctx->free_blocks.size() < 256) {
        ctx->free_blocks.emplace_back(d_ptr, sz);                 // kept for re-use; returned to the device by
        ctx->free_bytes += sz;                                    // rhj_release_workspace / rhj_destroy
        return RHJ_OK;
    }
    { std::lock_guard<std::mutex> lk(g_pool_sizes_mu); g_pool_sizes.erase(d_ptr); }
    HIPCHK(ctx, hipFree(d_ptr));
    return RHJ_OK;
}

int rhj_copy_h2d(rhj_ctx *ctx, void *d_dst, const void *src, uint64_t bytes)
{
    RHJCHK(use_device(ctx));
    if (bytes) {
        HIPCHK(ctx, hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    return RHJ_OK;
}

int rhj_copy_d2h(rhj_ctx *ctx, void *dst, const void *d_src, uint64_t bytes)
{
    RHJCHK(use_device(ctx));
    if (bytes) {
        HIPCHK(ctx, hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    return RHJ_OK;
}

int rhj_dev_mem_info(rhj_ctx *ctx, uint64_t *free_bytes, uint64_t *total_bytes)
{
    RHJCHK(use_device(ctx));
    size_t f = 0, t = 0;
    HIPCHK(ctx, hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    return RHJ_OK;
}

}  // extern "C"
