// tools/srd_store_hazard.hip -- what went wrong when k_join_ct's pair stores went through a buffer descriptor (DESIGN §4.4, round 3:
// wrong pairs in 2 of 15 instantiations, "3 % of the pairs, a different 3 % every run"; the experiment was dropped unexplained).
// Two hypotheses, one probe each:
//
//  A  a WAIT STATE is missing: `buffer_store_dwordx4 vdata, voff, srsrc, soff` reads its four data VGPRs some cycles after it
//     issues; a VALU instruction right behind it that overwrites one of them races with that read.  The ISA documents the
//     hazard (VMEM store of more than 64 bits of data, then a VALU write of the data registers: 1 wait state), and LLVM's hazard
//     recogniser inserts the s_nop -- EXCEPT when the instruction's soffset operand is an SGPR, where it assumes the hardware
//     interlocks (GCNHazardRecognizer::createsVALUHazard).  The failing kernels were exactly the ones with the running output
//     offset in soffset.  Probe: the instruction pair in inline asm, soffset an SGPR / the literal 0, with 0, 1 and 2 s_nop
//     between store and overwrite; every stored pair checked.
//  B  the 32-bit RANGE of a descriptor (num_records) or of voffset + soffset wraps on outputs beyond 4 GiB.
//     Probe: per-wavefront descriptors whose base lies 5 GiB into a 6 GiB allocation, records = what is left / clamped to
//     2^32 - 1; every stored pair checked.
//
//     hipcc --offload-arch=gfx950 -O2 tools/srd_store_hazard.hip -o /tmp/srd_hazard && /tmp/srd_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned long long u64;
typedef unsigned int u32;
typedef u32 v4u __attribute__((ext_vector_type(4)));

#define HIPOK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

constexpr int ROUNDS = 32;         // stores per lane

// out[(wave * ROUNDS + r) * 64 + lane] = {tag(wave, r, lane), ~tag}: 16 bytes per lane per round, each wavefront its own descriptor
template <int NOPS, bool SOFF_SGPR>
__global__ void __launch_bounds__(1024) k_hazard(unsigned char *out, u32 waves_total, u64 *clobber_sink)
{
    const u32 lane = threadIdx.x & 63, wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (wave >= waves_total) return;
    unsigned char *base = out + (u64)wave * ROUNDS * 64 * 16;
    const u32 base_lo = __builtin_amdgcn_readfirstlane((u32)(u64)base), base_hi = __builtin_amdgcn_readfirstlane((u32)((u64)base >> 32));
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(((u64)base_hi << 32) | base_lo), 0, ROUNDS * 64 * 16, 0x00020000);
    u32 junk = 0;
    for (int r = 0; r < ROUNDS; r++) {
        const u64 tag = ((u64)wave << 20) | ((u64)r << 8) | lane;
        const u32 d0 = (u32)tag, d1 = (u32)(tag >> 32), d2 = ~d0, d3 = ~d1;
        const u32 soff = __builtin_amdgcn_readfirstlane((u32)r * 64u * 16u);
        const u32 voff = SOFF_SGPR ? lane * 16u : lane * 16u + (u32)r * 64u * 16u;
        // the store, then AT ONCE a VALU write of all four data registers (what the compiler scheduled behind the store in the
        // failing kernels: the next slot's pair being assembled in the same registers)
#define STORE_THEN_CLOBBER(NOPSTR, SOFFSTR, ...)                                                                                    \
        asm volatile("v_mov_b32 v20, %1\n\tv_mov_b32 v21, %2\n\tv_mov_b32 v22, %3\n\tv_mov_b32 v23, %4\n\ts_nop 7\n\t"                  \
                     "buffer_store_dwordx4 v[20:23], %5, %6, " SOFFSTR " offen\n\t" NOPSTR                                            \
                     "v_mov_b32 v20, 0xdeadbeef\n\tv_mov_b32 v21, 0xdeadbeef\n\tv_mov_b32 v22, 0xdeadbeef\n\tv_mov_b32 v23, 0xdeadbeef\n\t" \
                     "v_add_u32 %0, v20, v23\n"                                                                                    \
                     : "=v"(junk) : "v"(d0), "v"(d1), "v"(d2), "v"(d3), "v"(voff), "s"(rsrc), __VA_ARGS__ : "v20", "v21", "v22", "v23", "memory")
        if constexpr (SOFF_SGPR) {
            if constexpr (NOPS == 0) STORE_THEN_CLOBBER("", "%7", "s"(soff));
            else if constexpr (NOPS == 1) STORE_THEN_CLOBBER("s_nop 0\n\t", "%7", "s"(soff));
            else STORE_THEN_CLOBBER("s_nop 1\n\t", "%7", "s"(soff));
        } else {
            if constexpr (NOPS == 0) STORE_THEN_CLOBBER("", "0", "s"(soff));
            else if constexpr (NOPS == 1) STORE_THEN_CLOBBER("s_nop 0\n\t", "0", "s"(soff));
            else STORE_THEN_CLOBBER("s_nop 1\n\t", "0", "s"(soff));
        }
    }
    if (junk == 0x12345u) *clobber_sink = junk;
}

static u64 check(const std::vector<u64> &h, u32 waves)
{
    u64 bad = 0;
    for (u32 w = 0; w < waves; w++)
        for (int r = 0; r < ROUNDS; r++)
            for (u32 l = 0; l < 64; l++) {
                const u64 tag = ((u64)w << 20) | ((u64)r << 8) | l;
                const size_t i = (((size_t)w * ROUNDS + r) * 64 + l) * 2;
                if (h[i] != tag || h[i + 1] != ~tag) bad++;
            }
    return bad;
}

// hypothesis B: the same stores through the compiler's builtin, bases beyond 4 GiB, records clamped or exact
__global__ void __launch_bounds__(1024) k_range(unsigned char *out, u64 first_byte, u32 waves_total, int clamp)
{
    const u32 lane = threadIdx.x & 63, wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (wave >= waves_total) return;
    unsigned char *base = out + first_byte + (u64)wave * ROUNDS * 64 * 16;
    const u32 base_lo = __builtin_amdgcn_readfirstlane((u32)(u64)base), base_hi = __builtin_amdgcn_readfirstlane((u32)((u64)base >> 32));
    const u64 left = (u64)(waves_total - wave) * ROUNDS * 64 * 16;            // bytes from this wavefront's base to the end
    const u32 records = clamp ? (left > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)left) : (u32)(ROUNDS * 64 * 16);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(((u64)base_hi << 32) | base_lo), 0, (int)records, 0x00020000);
    for (int r = 0; r < ROUNDS; r++) {
        const u64 tag = ((u64)wave << 20) | ((u64)r << 8) | lane;
        const v4u v = {(u32)tag, (u32)(tag >> 32), ~(u32)tag, ~(u32)(tag >> 32)};
        __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, (int)(lane * 16u), (int)__builtin_amdgcn_readfirstlane((u32)r * 64u * 16u), 0);
    }
}

int main()
{
    const u32 waves = 256 * 16 * 8;                        // 8 workgroups of 16 wavefronts per CU
    const size_t bytes = (size_t)waves * ROUNDS * 64 * 16;
    unsigned char *d;
    u64 *sink;
    HIPOK(hipMalloc(&d, bytes));
    HIPOK(hipMalloc(&sink, 8));
    std::vector<u64> h(bytes / 8);
    int failed = 0;
#define RUN(NOPS, SGPR, name)                                                                                                   \
    do {                                                                                                                        \
        u64 worst = 0;                                                                                                          \
        for (int rep = 0; rep < 5; rep++) {                                                                                     \
            HIPOK(hipMemset(d, 0, bytes));                                                                                      \
            hipLaunchKernelGGL((k_hazard<NOPS, SGPR>), dim3(waves / 16), dim3(1024), 0, 0, d, waves, sink);                   \
            HIPOK(hipDeviceSynchronize());                                                                                      \
            HIPOK(hipMemcpy(h.data(), d, bytes, hipMemcpyDeviceToHost));                                                        \
            const u64 bad = check(h, waves);                                                                                    \
            worst = bad > worst ? bad : worst;                                                                                  \
        }                                                                                                                       \
        printf("A  %-58s wrong pairs (worst of 5 runs): %llu of %llu\n", name, worst, (u64)waves * ROUNDS * 64);               \
        if (worst && NOPS >= 2) failed = 1;                                                                                     \
    } while (0)
    RUN(0, true, "soffset = SGPR, NO wait state before the VALU overwrite");
    RUN(1, true, "soffset = SGPR, s_nop 0 (1 wait state)");
    RUN(2, true, "soffset = SGPR, s_nop 1 (2 wait states)");
    RUN(0, false, "soffset = 0 (offset in voffset), NO wait state");
    RUN(1, false, "soffset = 0 (offset in voffset), s_nop 0 (1 wait state)");
    RUN(2, false, "soffset = 0 (offset in voffset), s_nop 1 (2 wait states)");
    HIPOK(hipFree(d));
    // B: 6 GiB allocation, stores start 5 GiB in
    const u64 first = 5ull << 30;
    const u32 wavesB = 256 * 16 * 2;
    const size_t bytesB = (size_t)wavesB * ROUNDS * 64 * 16;
    unsigned char *big;
    HIPOK(hipMalloc(&big, first + bytesB));
    std::vector<u64> hb(bytesB / 8);
    for (int clamp = 0; clamp < 2; clamp++) {
        HIPOK(hipMemset(big + first, 0, bytesB));
        hipLaunchKernelGGL(k_range, dim3(wavesB / 16), dim3(1024), 0, 0, big, first, wavesB, clamp);
        HIPOK(hipDeviceSynchronize());
        HIPOK(hipMemcpy(hb.data(), big + first, bytesB, hipMemcpyDeviceToHost));
        const u64 bad = check(hb, wavesB);
        printf("B  base 5 GiB into the allocation, records %-28s wrong pairs: %llu of %llu\n",
               clamp ? "= bytes left, clamped to 2^32-1" : "= this wavefront's 32 KiB", bad, (u64)wavesB * ROUNDS * 64);
        if (bad) failed = 1;
    }
    HIPOK(hipFree(big));
    return failed;
}
