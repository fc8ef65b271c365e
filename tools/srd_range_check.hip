// tools/srd_range_check.hip -- does the range check of a raw buffer descriptor account for the scalar offset of a buffer_load?
// (k_join_ct / k_join_bkt address slot rows as descriptor + lane offset + scalar row offset and rely on lanes past the end reading 0.)
//     hipcc --offload-arch=gfx950 -O2 tools/srd_range_check.hip -o /tmp/srd && /tmp/srd      -> "SRD range check includes soffset: ok" on MI355X
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64; typedef unsigned int u32;
typedef u32 v2u __attribute__((ext_vector_type(2)));
__global__ void k(const u64 *p, u32 n, u64 *out) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, (int)(n * 8), 0x00020000);
    for (int k2 = 0; k2 < 4; k2++) {
        v2u v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)(threadIdx.x * 8), (int)(k2 * 64 * 8), 0);
        out[k2 * 64 + threadIdx.x] = (u64)v.x | ((u64)v.y << 32);
    }
}
int main() {
    u64 *d, *o; hipMalloc(&d, 4096 * 8); hipMalloc(&o, 256 * 8);
    u64 h[4096]; for (int i = 0; i < 4096; i++) h[i] = 1000 + i;
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    u32 n = 150;                       // rows of 64: row 2 partial (128..149 valid), row 3 entirely out of range
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, n, o);
    u64 r[256]; hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 256; i++) { u64 e = i < (int)n ? 1000 + i : 0; if (r[i] != e) { if (bad < 5) printf("i=%d got %llu want %llu\n", i, r[i], e); bad++; } }
    printf(bad ? "SRD RANGE CHECK: %d mismatches\n" : "SRD range check includes soffset: ok (%d)\n", bad);
    return bad != 0;
}
