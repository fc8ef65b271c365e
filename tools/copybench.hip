// tools/copybench.hip -- which 16 B/lane copy shapes reach the HBM copy ceiling on MI355X (development aid).
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/copybench tools/copybench.hip && gpurun_out/copybench [GiB]
// Shapes: the scatter kernel's (one contiguous region per workgroup, 32 KiB tiles, next tile prefetched)
// against tile-interleaved and grid-stride copies, with and without nontemporal hints.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>

typedef unsigned long long u64;
typedef unsigned int u32;
struct __attribute__((aligned(16))) Tup { u64 key, payload; };

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <bool NT> __device__ __forceinline__ Tup ld(const Tup *p)
{
    if (NT) { Tup t; t.key = __builtin_nontemporal_load(&p->key); t.payload = __builtin_nontemporal_load(&p->payload); return t; }
    return *p;
}
template <bool NT> __device__ __forceinline__ void st(Tup *p, const Tup &t)
{
    if (NT) { __builtin_nontemporal_store(t.key, &p->key); __builtin_nontemporal_store(t.payload, &p->payload); }
    else *p = t;
}

// MODE 0: workgroup u copies the contiguous region [u*L, (u+1)*L) tile by tile (the scatter's shape)
// MODE 1: workgroup u copies tiles u, u+G, u+2G, ...
template <int THREADS, int TPT, int MODE, bool NTL, bool NTS, bool PREFETCH>
__global__ void __launch_bounds__(THREADS) k_copy(const Tup *__restrict__ in, Tup *__restrict__ out, u64 n, u64 L)
{
    constexpr u64 TILE = (u64)THREADS * TPT;
    const u32 u = blockIdx.x, G = gridDim.x, tid = threadIdx.x;
    const u64 ntiles_unit = (MODE == 0) ? (L + TILE - 1) / TILE : (n / TILE + G - 1 - u) / G;
    auto base = [&](u64 j) -> u64 { return MODE == 0 ? (u64)u * L + j * TILE : (j * G + u) * TILE; };
    auto lim = [&](u64 j) -> u64 {
        if (MODE == 0) { u64 e = (u64)(u + 1) * L; return e < n ? e : n; }
        return n;
    };
    Tup a[TPT], b[TPT];
    auto load = [&](Tup (&t)[TPT], u64 j) {
        const u64 tb = base(j), e = lim(j);
#pragma unroll
        for (int k = 0; k < TPT; k++) { const u64 i = tb + (u64)k * THREADS + tid; if (i < e) t[k] = ld<NTL>(in + i); }
    };
    auto store = [&](Tup (&t)[TPT], u64 j) {
        const u64 tb = base(j), e = lim(j);
#pragma unroll
        for (int k = 0; k < TPT; k++) { const u64 i = tb + (u64)k * THREADS + tid; if (i < e) st<NTS>(out + i, t[k]); }
    };
    if (!PREFETCH) {
        for (u64 j = 0; j < ntiles_unit; j++) { load(a, j); store(a, j); }
        return;
    }
    u64 j = 0;
    if (j < ntiles_unit) load(a, j);
    while (j < ntiles_unit) {
        if (j + 1 < ntiles_unit) load(b, j + 1);
        store(a, j);
        if (++j >= ntiles_unit) break;
        if (j + 1 < ntiles_unit) load(a, j + 1);
        store(b, j);
        ++j;
    }
}

// read-only: sum of payloads (one atomic per workgroup)
template <int THREADS, int TPT, int MODE>
__global__ void __launch_bounds__(THREADS) k_read(const Tup *__restrict__ in, u64 *__restrict__ sink, u64 n, u64 L)
{
    constexpr u64 TILE = (u64)THREADS * TPT;
    const u32 u = blockIdx.x, G = gridDim.x, tid = threadIdx.x;
    const u64 nt = (MODE == 0) ? (L + TILE - 1) / TILE : (n / TILE + G - 1 - u) / G;
    u64 acc = 0;
    for (u64 j = 0; j < nt; j++) {
        const u64 tb = MODE == 0 ? (u64)u * L + j * TILE : (j * G + u) * TILE;
        u64 e = MODE == 0 ? (u64)(u + 1) * L : n; if (e > n) e = n;
        Tup t[TPT];
#pragma unroll
        for (int k = 0; k < TPT; k++) { const u64 i = tb + (u64)k * THREADS + tid; t[k] = (i < e) ? in[i] : Tup{0, 0}; }
#pragma unroll
        for (int k = 0; k < TPT; k++) acc += t[k].payload ^ t[k].key;
    }
    if (acc == 0x1234567) atomicAdd(sink, acc);
}

// write-only
template <int THREADS, int TPT, int MODE>
__global__ void __launch_bounds__(THREADS) k_write(Tup *__restrict__ out, u64 n, u64 L)
{
    constexpr u64 TILE = (u64)THREADS * TPT;
    const u32 u = blockIdx.x, G = gridDim.x, tid = threadIdx.x;
    const u64 nt = (MODE == 0) ? (L + TILE - 1) / TILE : (n / TILE + G - 1 - u) / G;
    for (u64 j = 0; j < nt; j++) {
        const u64 tb = MODE == 0 ? (u64)u * L + j * TILE : (j * G + u) * TILE;
        u64 e = MODE == 0 ? (u64)(u + 1) * L : n; if (e > n) e = n;
#pragma unroll
        for (int k = 0; k < TPT; k++) { const u64 i = tb + (u64)k * THREADS + tid; if (i < e) out[i] = Tup{i, j}; }
    }
}


// ---- narrow (12 B/tuple, structure of arrays) experiments -----------------------------------------------
// tuples/s is what matters here: AoS16 moves 32 B per tuple copied, SoA12 moves 24 B
// PAIR: a lane handles two adjacent tuples (16 B payload load + 8 B rowid load)
template <int THREADS, int TPT, bool PAIR>
__global__ void __launch_bounds__(THREADS) k_copy_soa(const u64 *__restrict__ inP, const u32 *__restrict__ inK,
                                                      u64 *__restrict__ outP, u32 *__restrict__ outK, u64 n, u64 L)
{
    constexpr u64 TILE = (u64)THREADS * TPT;
    const u32 u = blockIdx.x, tid = threadIdx.x;
    const u64 beg = (u64)u * L, end = beg + L < n ? beg + L : n;
    for (u64 tb = beg; tb < end; tb += TILE) {
        if (PAIR) {
            ulonglong2 p[TPT / 2]; uint2 k2[TPT / 2];
#pragma unroll
            for (int k = 0; k < TPT / 2; k++) {
                const u64 i = tb + (u64)k * 2 * THREADS + 2 * tid;
                if (i < end) { p[k] = *reinterpret_cast<const ulonglong2 *>(inP + i); k2[k] = *reinterpret_cast<const uint2 *>(inK + i); }
            }
#pragma unroll
            for (int k = 0; k < TPT / 2; k++) {
                const u64 i = tb + (u64)k * 2 * THREADS + 2 * tid;
                if (i < end) { *reinterpret_cast<ulonglong2 *>(outP + i) = p[k]; *reinterpret_cast<uint2 *>(outK + i) = k2[k]; }
            }
        } else {
            u64 p[TPT]; u32 kk[TPT];
#pragma unroll
            for (int k = 0; k < TPT; k++) { const u64 i = tb + (u64)k * THREADS + tid; if (i < end) { p[k] = inP[i]; kk[k] = inK[i]; } }
#pragma unroll
            for (int k = 0; k < TPT; k++) { const u64 i = tb + (u64)k * THREADS + tid; if (i < end) { outP[i] = p[k]; outK[i] = kk[k]; } }
        }
    }
}

// AoS16 in -> scattered 8-tuple lines out (the write pattern of the scatter: 8 adjacent lanes own one line at a
// pseudo-random line index).  NARROW: line = 64 B of payloads + 32 B of rowids in two arrays; else one 128 B line.
template <int THREADS, int TPT, bool NARROW>
__global__ void __launch_bounds__(THREADS) k_scatter_lines(const Tup *__restrict__ in, Tup *__restrict__ out, u64 *__restrict__ outP,
                                                           u32 *__restrict__ outK, u64 n, u64 L)
{
    constexpr u64 TILE = (u64)THREADS * TPT;
    const u32 u = blockIdx.x, tid = threadIdx.x;
    const u64 beg = (u64)u * L, end = beg + L < n ? beg + L : n;
    const u64 nlines = n / 8;
    for (u64 tb = beg; tb < end; tb += TILE) {
        Tup t[TPT];
#pragma unroll
        for (int k = 0; k < TPT; k++) { const u64 i = tb + (u64)k * THREADS + tid; if (i < end) t[k] = in[i]; }
#pragma unroll
        for (int k = 0; k < TPT; k++) {
            const u64 i = tb + (u64)k * THREADS + tid;
            if (i < end) {
                const u64 line = ((i >> 3) * 0x9E3779B97F4A7C15ull >> 20) % nlines;          // a permutation-ish map of lines
                const u64 o = line * 8 + (i & 7);
                if (NARROW) { outP[o] = t[k].payload; outK[o] = (u32)t[k].key; }
                else out[o] = t[k];
            }
        }
    }
}

// AoS16 in -> the scatter's real write pattern: 256 sequential streams per workgroup, each advancing by one chunk of GR
// tuples at a time.  NARROW: chunk = GR*8 B of payloads + GR*4 B of rowids in two arrays; else GR*16 B in one.
template <int THREADS, int TPT, bool NARROW, int GR, int MODE = 0>
__global__ void __launch_bounds__(THREADS) k_stream_lines(const Tup *__restrict__ in, Tup *__restrict__ out, u64 *__restrict__ outP,
                                                          u32 *__restrict__ outK, u64 n, u64 L)
{
    constexpr u64 TILE = (u64)THREADS * TPT;
    const u32 u = blockIdx.x, tid = threadIdx.x, G = gridDim.x;
    const u64 beg = (u64)u * L, end = beg + L < n ? beg + L : n;
    const u64 cpud = (L / GR + 255) / 256, cpd = (u64)G * cpud;
    u64 j = 0;
    for (u64 tb = beg; tb < end; tb += TILE, j++) {
        Tup t[TPT];
        // MODE 1: the READ side is tile-interleaved over the workgroups (tile j of workgroup u = global tile j * G + u)
        const u64 rb = MODE == 0 ? tb : (j * G + u) * TILE;
#pragma unroll
        for (int k = 0; k < TPT; k++) { const u64 i = rb + (u64)k * THREADS + tid; if (tb + (u64)k * THREADS + tid < end && i < n) t[k] = in[i]; }
#pragma unroll
        for (int k = 0; k < TPT; k++) {
            const u32 it = k * THREADS + tid;
            if (tb + it < end) {
                const u64 cg = j * (TILE / GR) + it / GR;
                const u64 o = ((cg & 255) * cpd + (u64)u * cpud + (cg >> 8)) * GR + it % GR;
                if (NARROW) { outP[o] = t[k].payload; outK[o] = (u32)t[k].key; }
                else out[o] = t[k];
            }
        }
    }
}

// AoS16 in -> SoA12 out, linear (the byte mix of a narrow-output pass without any scatter): T threads, tiles of T*TPT
template <int THREADS, int TPT>
__global__ void __launch_bounds__(THREADS) k_aos_to_soa(const Tup *__restrict__ in, u64 *__restrict__ outP, u32 *__restrict__ outK, u64 n, u64 L)
{
    constexpr u64 TILE = (u64)THREADS * TPT;
    const u32 u = blockIdx.x, tid = threadIdx.x;
    const u64 beg = (u64)u * L, end = beg + L < n ? beg + L : n;
    for (u64 tb = beg; tb < end; tb += TILE) {
        Tup t[TPT];
#pragma unroll
        for (int k = 0; k < TPT; k++) { const u64 i = tb + (u64)k * THREADS + tid; if (i < end) t[k] = in[i]; }
#pragma unroll
        for (int k = 0; k < TPT; k++) { const u64 i = tb + (u64)k * THREADS + tid; if (i < end) { outP[i] = t[k].payload; outK[i] = (u32)t[k].key; } }
    }
}

template <typename F> static double time_ms(F f, int reps = 5)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); CK(hipDeviceSynchronize());
    std::vector<float> ms;
    for (int r = 0; r < reps; r++) {
        CK(hipEventRecord(e0, 0)); f(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float m; CK(hipEventElapsedTime(&m, e0, e1)); ms.push_back(m);
    }
    std::sort(ms.begin(), ms.end());
    return ms[ms.size() / 2];
}

int main(int argc, char **argv)
{
    const double gib = argc > 1 ? atof(argv[1]) : 14.9;
    const u64 n = (u64)(gib * (1ull << 30) / 16) / 8192 * 8192;
    Tup *in, *out; u64 *sink;
    CK(hipMalloc(&in, n * 16)); CK(hipMalloc(&out, n * 16)); CK(hipMalloc(&sink, 8));
    CK(hipMemset(in, 1, n * 16)); CK(hipMemset(out, 0, n * 16)); CK(hipMemset(sink, 0, 8));
    auto report = [&](const char *name, double ms, double bytes) {
        printf("%-58s %8.3f ms  %7.0f GB/s\n", name, ms, bytes / ms / 1e6); fflush(stdout);
    };
    report("hipMemcpyDtoD", time_ms([&] { CK(hipMemcpyAsync(out, in, n * 16, hipMemcpyDeviceToDevice, 0)); }), 32.0 * n);

#define RUN_COPY(T, P, M, NL, NS, PF, G, LDS)                                                                     \
    {                                                                                                                \
        const u64 L = ((n + (G) - 1) / (G) + (T) * (P) - 1) / ((T) * (P)) * ((T) * (P));                          \
        char nm[128];                                                                                                \
        snprintf(nm, sizeof nm, "copy T=%d tpt=%d %s ntl=%d nts=%d pf=%d G=%d lds=%dK", T, P,                     \
                 M ? "interleaved" : "contiguous", NL, NS, PF, (int)(G), (int)((LDS) >> 10));                      \
        report(nm, time_ms([&] { hipLaunchKernelGGL((k_copy<T, P, M, NL, NS, PF>), dim3(G), dim3(T), LDS, 0, in, out, n, L); }), \
               32.0 * n);                                                                                            \
    }
    // the scatter's shape: 512 threads, 4 x 16 B, 2 WGs/CU (LDS-limited), 2048 units
    RUN_COPY(512, 4, 0, false, false, true, 2048, 76 << 10)
    RUN_COPY(512, 4, 0, false, false, true, 512, 76 << 10)
    RUN_COPY(512, 4, 1, false, false, true, 2048, 76 << 10)
    RUN_COPY(512, 4, 1, false, false, true, 512, 76 << 10)
    RUN_COPY(512, 4, 0, true, true, true, 2048, 76 << 10)
    RUN_COPY(512, 4, 1, true, true, true, 2048, 76 << 10)
    RUN_COPY(512, 4, 0, false, true, true, 2048, 76 << 10)
    RUN_COPY(512, 4, 0, true, false, true, 2048, 76 << 10)
    RUN_COPY(512, 4, 0, false, false, false, 2048, 76 << 10)
    RUN_COPY(512, 4, 1, false, false, false, 2048, 76 << 10)
    // occupancy: no LDS cap
    RUN_COPY(512, 4, 0, false, false, true, 2048, 0)
    RUN_COPY(512, 4, 1, false, false, true, 2048, 0)
    RUN_COPY(256, 4, 1, false, false, false, 4096, 0)
    RUN_COPY(256, 1, 1, false, false, false, 8192, 0)
    RUN_COPY(256, 4, 1, true, true, false, 4096, 0)
    RUN_COPY(1024, 4, 0, false, false, true, 2048, 150 << 10)
    RUN_COPY(1024, 4, 1, false, false, true, 2048, 150 << 10)
    RUN_COPY(1024, 2, 0, false, false, true, 2048, 150 << 10)
    RUN_COPY(512, 8, 0, false, false, true, 2048, 76 << 10)
    RUN_COPY(512, 2, 0, false, false, true, 2048, 76 << 10)
    RUN_COPY(256, 4, 0, false, false, true, 2048, 38 << 10)
    RUN_COPY(256, 8, 0, false, false, true, 2048, 38 << 10)

#define RUN_RW(T, P, M, G, LDS)                                                                                   \
    {                                                                                                                \
        const u64 L = ((n + (G) - 1) / (G) + (T) * (P) - 1) / ((T) * (P)) * ((T) * (P));                          \
        char nm[128];                                                                                                \
        snprintf(nm, sizeof nm, "read  T=%d tpt=%d %s G=%d lds=%dK", T, P, M ? "interleaved" : "contiguous", (int)(G), (int)((LDS) >> 10)); \
        report(nm, time_ms([&] { hipLaunchKernelGGL((k_read<T, P, M>), dim3(G), dim3(T), LDS, 0, in, sink, n, L); }), 16.0 * n); \
        snprintf(nm, sizeof nm, "write T=%d tpt=%d %s G=%d lds=%dK", T, P, M ? "interleaved" : "contiguous", (int)(G), (int)((LDS) >> 10)); \
        report(nm, time_ms([&] { hipLaunchKernelGGL((k_write<T, P, M>), dim3(G), dim3(T), LDS, 0, out, n, L); }), 16.0 * n); \
    }
    RUN_RW(512, 4, 0, 2048, 76 << 10)
    RUN_RW(512, 4, 1, 2048, 76 << 10)
    RUN_RW(512, 4, 0, 2048, 0)
    RUN_RW(512, 4, 1, 2048, 0)
    RUN_RW(256, 4, 1, 8192, 0)
    {   // narrow experiments: reuse the two buffers as (P, K) pairs
        u64 *inP = reinterpret_cast<u64 *>(in), *outP = reinterpret_cast<u64 *>(out);
        u32 *inK = reinterpret_cast<u32 *>(inP + n), *outK = reinterpret_cast<u32 *>(outP + n);
        const int G = 2048;
        const u64 L = ((n + G - 1) / G + 2047) / 2048 * 2048;
        auto rep = [&](const char *name, double ms) { printf("%-58s %8.3f ms  %7.2f Gtuples/s\n", name, ms, n / ms / 1e6); fflush(stdout); };
        rep("AoS16 copy T=512 tpt=4 contiguous (reference point)",
            time_ms([&] { hipLaunchKernelGGL((k_copy<512, 4, 0, false, false, true>), dim3(G), dim3(512), 76 << 10, 0, in, out, n, L); }));
        rep("SoA12 copy T=512 tpt=4 single", time_ms([&] { hipLaunchKernelGGL((k_copy_soa<512, 4, false>), dim3(G), dim3(512), 76 << 10, 0, inP, inK, outP, outK, n, L); }));
        rep("SoA12 copy T=512 tpt=8 single", time_ms([&] { hipLaunchKernelGGL((k_copy_soa<512, 8, false>), dim3(G), dim3(512), 76 << 10, 0, inP, inK, outP, outK, n, L); }));
        rep("SoA12 copy T=512 tpt=4 pair", time_ms([&] { hipLaunchKernelGGL((k_copy_soa<512, 4, true>), dim3(G), dim3(512), 76 << 10, 0, inP, inK, outP, outK, n, L); }));
        rep("SoA12 copy T=512 tpt=8 pair", time_ms([&] { hipLaunchKernelGGL((k_copy_soa<512, 8, true>), dim3(G), dim3(512), 76 << 10, 0, inP, inK, outP, outK, n, L); }));
        rep("AoS16 -> scattered 128 B lines", time_ms([&] { hipLaunchKernelGGL((k_scatter_lines<512, 4, false>), dim3(G), dim3(512), 76 << 10, 0, in, out, outP, outK, n, L); }));
        rep("AoS16 -> scattered 64 B + 32 B lines (narrow)", time_ms([&] { hipLaunchKernelGGL((k_scatter_lines<512, 4, true>), dim3(G), dim3(512), 76 << 10, 0, in, out, outP, outK, n, L); }));
#define RUN_SL(NARROW, GR)                                                                                          \
    {                                                                                                                \
        const u64 n2 = n - (64ull << 20), L2 = ((n2 + G - 1) / G + 2047) / 2048 * 2048;                              \
        const u64 top = 256ull * G * ((L2 / GR + 255) / 256) * GR;             /* one past the largest index written */ \
        if (top > n) { fprintf(stderr, "stream test would overrun: %llu > %llu\n", top, n); exit(1); }               \
        rep("AoS16 -> 256 streams/WG, " #NARROW " GR=" #GR, time_ms([&] { hipLaunchKernelGGL((k_stream_lines<512, 4, NARROW, GR>), dim3(G), dim3(512), 76 << 10, 0, in, out, outP, outK, n2, L2); }) * (double)n / (double)n2); \
    }
        rep("AoS16 -> SoA12 linear T=512 tpt=4 lds=76K", time_ms([&] { hipLaunchKernelGGL((k_aos_to_soa<512, 4>), dim3(G), dim3(512), 76 << 10, 0, in, outP, outK, n, L); }));
        rep("AoS16 -> SoA12 linear T=1024 tpt=4 lds=150K", time_ms([&] { hipLaunchKernelGGL((k_aos_to_soa<1024, 4>), dim3(G), dim3(1024), 150 << 10, 0, in, outP, outK, n, L); }));
        rep("AoS16 -> SoA12 linear T=256 tpt=4 lds=0", time_ms([&] { hipLaunchKernelGGL((k_aos_to_soa<256, 4>), dim3(G * 4), dim3(256), 0, 0, in, outP, outK, n, (L + 3) / 4 / 1024 * 1024 + 1024); }));
#define RUN_SLM(T, MODE)                                                                                             \
    {                                                                                                                \
        const u64 n2 = n - (64ull << 20), L2 = ((n2 + G - 1) / G + 4095) / 4096 * 4096;                              \
        const u64 top = 256ull * G * ((L2 / 32 + 255) / 256) * 32;                                                   \
        if (top > n) { fprintf(stderr, "stream test would overrun\n"); exit(1); }                                    \
        rep("AoS16 -> 256 streams/WG narrow GR=32, T=" #T " lds=150K, read " #MODE, time_ms([&] { hipLaunchKernelGGL((k_stream_lines<T, 4, true, 32, MODE>), dim3(G), dim3(T), 150 << 10, 0, in, out, outP, outK, n2, L2); }) * (double)n / (double)n2); \
    }
        RUN_SLM(1024, 0) RUN_SLM(1024, 1) RUN_SLM(1024, 0) RUN_SLM(1024, 1)
        RUN_SL(false, 8) RUN_SL(false, 4) RUN_SL(false, 16) RUN_SL(true, 8) RUN_SL(true, 16) RUN_SL(true, 32)
    }
    return 0;
}
