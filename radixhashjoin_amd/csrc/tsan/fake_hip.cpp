// fake_hip.cpp -- the host-only stand-in behind tsan/hip/hip_runtime.h, and stand-ins for the kernel launchers of
// rhj_internal.h.  See the header: test infrastructure for `make -C radixhashjoin_amd/csrc tsan`, nothing of the product.
#include "../rhj_internal.h"

#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>

// ---- streams: one thread each, executing its queue in order --------------------------------------------------------
struct FakeStream {
    std::mutex mu;
    std::condition_variable cv, idle;
    std::deque<std::function<void()>> q;
    bool quit = false, busy = false;
    std::thread th;
    FakeStream() : th([this] { run(); }) {}
    ~FakeStream()
    {
        { std::lock_guard<std::mutex> lk(mu); quit = true; }
        cv.notify_all();
        th.join();
    }
    void run()
    {
        for (;;) {
            std::function<void()> f;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return quit || !q.empty(); });
                if (q.empty()) return;
                f = std::move(q.front());
                q.pop_front();
                busy = true;
            }
            f();
            {
                std::lock_guard<std::mutex> lk(mu);
                busy = false;
            }
            idle.notify_all();
        }
    }
    void push(std::function<void()> f)
    {
        { std::lock_guard<std::mutex> lk(mu); q.push_back(std::move(f)); }
        cv.notify_all();
    }
    void drain()
    {
        std::unique_lock<std::mutex> lk(mu);
        idle.wait(lk, [&] { return q.empty() && !busy; });
    }
};

static FakeStream *null_stream()
{
    static FakeStream *s = new FakeStream();          // (leaked on purpose: alive for every static destructor)
    return s;
}
static FakeStream *S(hipStream_t st) { return st ? st : null_stream(); }
void fake_enqueue(hipStream_t st, std::function<void()> f) { S(st)->push(std::move(f)); }

// ---- events: "everything enqueued on the stream before the record has run" ------------------------------------------
struct FakeEvent {
    std::mutex mu;
    std::condition_variable cv;
    unsigned long long recorded = 0, done = 0;
};

hipError_t hipIpcGetMemHandle(hipIpcMemHandle_t *h, void *p) { memcpy(h->reserved, &p, sizeof p); return hipSuccess; }
hipError_t hipIpcOpenMemHandle(void **p, hipIpcMemHandle_t h, unsigned) { memcpy(p, h.reserved, sizeof *p); return hipSuccess; }
hipError_t hipIpcCloseMemHandle(void *) { return hipSuccess; }
hipError_t hipGetLastError() { return hipSuccess; }
const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "success" : "fake hip error"; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
hipError_t hipMalloc(void **p, size_t bytes) { *p = malloc(bytes ? bytes : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void *p) { free(p); return hipSuccess; }
hipError_t hipHostMalloc(void **p, size_t bytes, unsigned) { *p = malloc(bytes ? bytes : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
hipError_t hipHostGetDevicePointer(void **dev, void *host, unsigned) { *dev = host; return hipSuccess; }
hipError_t hipMemGetInfo(size_t *f, size_t *t) { *f = *t = (size_t)64 << 30; return hipSuccess; }
hipError_t hipMemcpy(void *dst, const void *src, size_t bytes, hipMemcpyKind)
{
    null_stream()->drain();
    memcpy(dst, src, bytes);
    return hipSuccess;
}
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t bytes, hipMemcpyKind, hipStream_t st)
{
    S(st)->push([=] { memcpy(dst, src, bytes); });    // the "DMA engine": reads the source when the stream gets there
    return hipSuccess;
}
hipError_t hipMemsetAsync(void *dst, int value, size_t bytes, hipStream_t st)
{
    S(st)->push([=] { memset(dst, value, bytes); });
    return hipSuccess;
}
hipError_t hipStreamCreateWithFlags(hipStream_t *st, unsigned) { *st = new FakeStream(); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t st) { delete st; return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t st) { S(st)->drain(); return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t *ev) { *ev = new FakeEvent(); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t *ev, unsigned) { return hipEventCreate(ev); }
hipError_t hipEventDestroy(hipEvent_t ev) { delete ev; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t ev, hipStream_t st)
{
    unsigned long long seq;
    { std::lock_guard<std::mutex> lk(ev->mu); seq = ++ev->recorded; }
    S(st)->push([=] {
        std::lock_guard<std::mutex> lk(ev->mu);           // (notify under the lock: a waiter may destroy the event as soon as it returns)
        if (ev->done < seq) ev->done = seq;
        ev->cv.notify_all();
    });
    return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t ev)
{
    std::unique_lock<std::mutex> lk(ev->mu);
    const unsigned long long want = ev->recorded;
    ev->cv.wait(lk, [&] { return ev->done >= want; });
    return hipSuccess;
}
hipError_t hipEventQuery(hipEvent_t ev)
{
    std::lock_guard<std::mutex> lk(ev->mu);
    return ev->done >= ev->recorded ? hipSuccess : hipErrorNotReady;
}
hipError_t hipStreamWaitEvent(hipStream_t st, hipEvent_t ev, unsigned)
{
    unsigned long long want;
    { std::lock_guard<std::mutex> lk(ev->mu); want = ev->recorded; }
    S(st)->push([=] {
        std::unique_lock<std::mutex> lk(ev->mu);
        ev->cv.wait(lk, [&] { return ev->done >= want; });
    });
    return hipSuccess;
}
hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0.001f; return hipSuccess; }

// ---- the launchers of rhj_internal.h ---------------------------------------------------------------------------------
// Host-logic helpers keep plausible values; kernels are nothing, except that a "bucket join" leaves a result behind: it
// appends FAKE_PAIRS pairs {i, ~i} (i = position in the pair buffer) and moves the result counter on, in stream order --
// what the host paths read back, hand to the downloader and copy into the result page.
static const u64 FAKE_PAIRS = (u64)1 << 20;
struct FPair { u64 r, s; };

bool fused_two_pass_ok(int b1, int b2) { return b1 >= 1 && b2 >= 1 && b1 <= 9 && b2 <= 9 && b1 + b2 <= 16; }
u32 join_probe_split(int kind) { return jk_is_ct(kind) ? 16384u : 0u; }
u32 join_table_tuples(int kind) { return jk_is_ct(kind) ? 16352u : kind == JK_BKT_BIG ? 8448u : (u32)BJ_CHUNK; }
int join_ct_min_radix_bits(int kind) { return kind == JK_CT_G13 ? 13 : kind == JK_CT_Q12 ? 12 : 16; }
int seg_max() { return 16; }
int tag_bits() { return 4; }
bool narrow_pass_ok(int bits) { return bits >= 1 && bits <= 8; }
bool narrow_pass9_ok(int bits) { return bits >= 1 && bits <= 9; }
const char *launch_attr_error() { return nullptr; }
size_t scan_tmp_bytes(int bits) { return (size_t)64 * ((size_t)8 << bits); }
size_t part_lds_bytes(int) { return 0; }

void launch_init_single_segment(hipStream_t, u64, u64, u64 *, u32 *) {}
void launch_make_units(hipStream_t, const u64 *, u32, u64, u32 *) {}
void launch_hist_units(hipStream_t, const void *, const PassGeom &, const u64 *, const u32 *, u32 *, u64 *, const DupSniff &) {}
void launch_scan_units(hipStream_t, const PassGeom &, const u64 *, const u32 *, const u32 *, u64 *, u64 *, u64 *) {}
void launch_scatter_units(hipStream_t, const void *, void *, const PassGeom &, const u64 *, const u32 *, const u64 *) {}
void launch_diff_hist(hipStream_t, const u64 *, u64, u64 *) {}
void launch_check_radix(hipStream_t, const void *, const u64 *, const void *, const u64 *, u64, int, u64 *) {}
void launch_prefix(hipStream_t, const u64 *, u64, u64 *) {}
void launch_make_tasks(hipStream_t st, const u64 *, const u64 *, u64, u32, JoinTask *, u32 *d_ntasks, u32, u64 *, int, const SniffVerdict &)
{
    fake_enqueue(st, [=] { *d_ntasks = 1; });
}
static void fake_join(void *d_out, u64 cap, u64 *d_out_count, u64 *host_count, void *host_out, u64 host_cap)
{
    const u64 at = *d_out_count;
    for (u64 i = at; i < at + FAKE_PAIRS; i++) {
        const FPair p{i, ~i};
        if (d_out && i < cap) ((FPair *)d_out)[i] = p;
        if (host_out && i < host_cap) ((FPair *)host_out)[i] = p;
    }
    if (host_count) { *host_count = at + FAKE_PAIRS; *d_out_count = 0; }     // (the direct kernel publishes and re-zeroes)
    else *d_out_count = at + FAKE_PAIRS;
}
void launch_join(hipStream_t st, const void *, const u64 *, const void *, const u64 *, const JoinTask *, const u32 *, u32, int,
                 void *d_out, u64 out_capacity, u64 *d_out_count, int, const u32 *, const u32 *, const u64 *, const u32 *, u64 *host_pub,
                 u32 *)
{
    fake_enqueue(st, [=] {
        fake_join(d_out, out_capacity, d_out_count, nullptr, nullptr, 0);
        if (host_pub) for (int i = 0; i < 7; i++) host_pub[i] = d_out_count[i];      // (the last workgroup publishes the counters)
    });
}
int build_tie_shift() { return 4; }
size_t fuse_ctl_bytes() { return 12352; }
u32 *fuse_join_ticket(void *d_ctl) { return (u32 *)((unsigned char *)d_ctl + 12288) + 1; }
void launch_fused_pass(hipStream_t st, const PassPairHost &, int, int phase, int, void *, u32, u32, u32, JoinTask *, u64 *d_counters, u64 *, bool)
{
    if (phase == 0) fake_enqueue(st, [=] { memset(d_counters, 0, 64); });           // (the histogram launch clears the join counters)
}
void launch_join_direct(hipStream_t st, const void *, u64, const void *, u64, void *d_out, u64 out_capacity, u64 *d_out_count,
                        u64 *host_count, u32 *, void *host_out, u64 host_cap)
{
    fake_enqueue(st, [=] { fake_join(d_out, out_capacity, d_out_count, host_count, host_out, host_cap); });
}
void launch_join_batch(hipStream_t st, const BatchJoinDesc *d_batch, u32 njoins, u32)
{
    fake_enqueue(st, [=] {                                          // (the descriptors arrive with the staged upload)
        for (u32 i = 0; i < njoins; i++) {
            const BatchJoinDesc &b = d_batch[i];
            fake_join(b.out, b.cap, b.count, b.host_count, b.host_out, b.host_cap);
        }
    });
}
u32 join_direct_tile() { return 4096; }
void launch_checksum(hipStream_t, const void *, u64, u64 *) {}
void launch_generate(hipStream_t, int, void *, u64, u64, u64, u64, double) {}
void launch_expected_pkfk(hipStream_t, const void *, u64, u64 *) {}
void launch_remap_keys(hipStream_t, void *, u64, int, u64) {}
void launch_pass_pair(hipStream_t st, const PassPairHost &h, int, int, int phase)
{
    u64 *z = h.zero8;
    if (phase == 0 && z) fake_enqueue(st, [=] { memset(z, 0, 64); });      // (the first launch clears the join counters)
}
void launch_hist2d_units(hipStream_t, const void *, bool, u64, u64, u32, int, int, u32, u32, u32 *, u32 *, u64, u32 *, const u64 *, int, const DupSniff &) {}
void launch_seg_units(hipStream_t, u32, const u64 *, const u64 *, u32, u64 *, u64 *, u32 *) {}
void launch_make_group_ranges(hipStream_t, const u64 *, u32, u32, u32, u64, u64 *, u32 *) {}
void launch_scatter_ranges(hipStream_t, const void *, void *, u32, int, int, const u64 *, const u64 *) {}
void launch_hist_units_narrow(hipStream_t, const void *, const PassGeom &, const u64 *, const u32 *, u32 *) {}
void launch_scatter_units_narrow_any(hipStream_t, const void *, const u32 *, void *, u32 *, const PassGeom &, const u64 *, const u32 *,
                                     const u64 *, u32 *) {}
void launch_scatter_units_narrow(hipStream_t, const void *, void *, u64, const PassGeom &, const u64 *, const u32 *, const u64 *, u32 *,
                                 u64) {}
void launch_scatter_ranges_narrow(hipStream_t, const void *, bool, void *, u64, u32, int, int, const u64 *, const u64 *, u32 *, u32, u32,
                                  const u32 *) {}
void launch_scatter_units_narrow_peer(hipStream_t, const void *, const PassGeom &, const u64 *, const u32 *, const u64 *, u32 *, u64,
                                      const u64 *, const unsigned char *, void *const *, void *const *, int) {}
void launch_scatter_ranges_n2a(hipStream_t, const void *, void *, u64, u32, int, int, const u64 *, const u64 *, const u64 *, u32, u32,
                               const u32 *) {}
