// rhj_query.hip -- query-layer kernels and their C-ABI (include/rhj.h, "query-layer kernels"): the steps
// immediately before and after the hot path, so that a whole query stays in HBM.
// All of them are streaming or gather kernels over uint64 arrays: bounded by HBM / L2, no LDS tiling needed.
#include "../../include/rhj.h"
#include "rhj_internal.h"

#include <string>

// provided by rhj_api.hip
int rhj_internal_use_device(rhj_ctx *ctx);
hipStream_t rhj_internal_stream(rhj_ctx *ctx);
int rhj_internal_fail(rhj_ctx *ctx, int code, const char *msg);
void *rhj_internal_counters(rhj_ctx *ctx);      // >= 64 bytes of zeroable device scratch, or nullptr on failure

namespace {

struct __align__(16) Tup { u64 key; u64 payload; };
struct __align__(16) Pair { u64 r; u64 s; };

// unordered stream compaction: wavefront ballot + mbcnt prefix, one global atomic per wavefront
__device__ __forceinline__ void emit_if(bool keep, u64 value, u64 *__restrict__ out, u64 *__restrict__ cursor)
{
    const unsigned long long m = __ballot(keep);
    if (m == 0) return;
    const int lane = threadIdx.x & 63;
    u64 base = 0;
    if (lane == __ffsll((long long)m) - 1) base = atomicAdd(cursor, (u64)__popcll(m));
    base = ((u64)__builtin_amdgcn_readlane((u32)(base >> 32), __ffsll((long long)m) - 1) << 32) |
           __builtin_amdgcn_readlane((u32)base, __ffsll((long long)m) - 1);
    if (keep) out[base + __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u))] = value;
}

__global__ void __launch_bounds__(256)
k_col_filter(const u64 *__restrict__ col, const u64 *__restrict__ rows_in, u64 n, int op, u64 value,
             u64 *__restrict__ rows_out, u64 *__restrict__ cursor)
{
    const u64 stride = (u64)gridDim.x * 256;
    const u64 nround = (n + stride - 1) / stride * stride;           // every lane runs the same trip count (ballot)
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < nround; i += stride) {
        bool keep = false;
        u64 r = 0;
        if (i < n) {
            r = rows_in ? rows_in[i] : i;
            const u64 v = col[r];
            keep = op == '>' ? v > value : op == '<' ? v < value : v == value;
        }
        emit_if(keep, r, rows_out, cursor);
    }
}

__global__ void __launch_bounds__(256)
k_rows_filter_equal(const u64 *__restrict__ colA, const u64 *__restrict__ rowsA, const u64 *__restrict__ colB,
                    const u64 *__restrict__ rowsB, u64 n, u64 *__restrict__ pos_out, u64 *__restrict__ cursor)
{
    const u64 stride = (u64)gridDim.x * 256;
    const u64 nround = (n + stride - 1) / stride * stride;
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < nround; i += stride) {
        const bool keep = i < n && colA[rowsA ? rowsA[i] : i] == colB[rowsB ? rowsB[i] : i];
        emit_if(keep, i, pos_out, cursor);
    }
}

__global__ void __launch_bounds__(256)
k_gather_tuples(const u64 *__restrict__ col, const u64 *__restrict__ rows, u64 n, int key_is_position, Tup *__restrict__ out)
{
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256) {
        const u64 r = rows ? rows[i] : i;
        Tup t;
        t.key = key_is_position ? i : r;
        t.payload = col[r];
        out[i] = t;
    }
}

__global__ void __launch_bounds__(256)
k_pairs_split(const Pair *__restrict__ p, u64 n, u64 *__restrict__ r, u64 *__restrict__ s)
{
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256) {
        const Pair x = p[i];
        r[i] = x.r;
        s[i] = x.s;
    }
}

__global__ void __launch_bounds__(256)
k_gather_u64(const u64 *__restrict__ src, const u64 *__restrict__ idx, u64 n, u64 *__restrict__ dst)
{
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256) dst[i] = src[idx[i]];
}

__global__ void __launch_bounds__(256)
k_sum_gather(const u64 *__restrict__ col, const u64 *__restrict__ rows, u64 n, u64 *__restrict__ sum)
{
    __shared__ u64 wtot[4];
    u64 acc = 0;
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256) acc += col[rows ? rows[i] : i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) wtot[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(sum, wtot[0] + wtot[1] + wtot[2] + wtot[3]);
}

unsigned grid_for(u64 n)
{
    u64 g = (n + 255) / 256;
    if (g > 2048) g = 2048;
    if (g == 0) g = 1;
    return (unsigned)g;
}

int check(rhj_ctx *ctx, const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return rhj_internal_fail(ctx, RHJ_E_HIP, (std::string(what) + ": " + hipGetErrorString(e)).c_str());
    return RHJ_OK;
}

// runs `launch(cursor)` with a zeroed device counter and returns its final value
template <typename F> int with_counter(rhj_ctx *ctx, uint64_t *host_out, const char *what, F launch)
{
    u64 *cursor = (u64 *)rhj_internal_counters(ctx);
    if (!cursor) return RHJ_E_NOMEM;
    cursor += 5;                                                      // slots 0..4 belong to the join / checksum paths
    hipStream_t st = rhj_internal_stream(ctx);
    if (hipMemsetAsync(cursor, 0, 8, st) != hipSuccess) return rhj_internal_fail(ctx, RHJ_E_HIP, "memset");
    launch(cursor, st);
    int rc = check(ctx, what);
    if (rc != RHJ_OK) return rc;
    if (hipMemcpyAsync(host_out, cursor, 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
        return rhj_internal_fail(ctx, RHJ_E_HIP, what);
    return RHJ_OK;
}

}  // namespace

extern "C" {

int rhj_col_filter(rhj_ctx *ctx, const uint64_t *d_col, const uint64_t *d_rows_in, uint64_t n_in, int op,
                   uint64_t value, uint64_t *d_rows_out, uint64_t *n_out)
{
    int rc = rhj_internal_use_device(ctx);
    if (rc != RHJ_OK) return rc;
    if (!n_out || (op != '<' && op != '>' && op != '=') || (n_in && (!d_col || !d_rows_out)))
        return rhj_internal_fail(ctx, RHJ_E_INVALID, "bad rhj_col_filter argument");
    *n_out = 0;
    if (n_in == 0) return RHJ_OK;
    return with_counter(ctx, n_out, "rhj_col_filter", [&](u64 *cursor, hipStream_t st) {
        hipLaunchKernelGGL(k_col_filter, dim3(grid_for(n_in)), dim3(256), 0, st, (const u64 *)d_col, (const u64 *)d_rows_in,
                           (u64)n_in, op, (u64)value, (u64 *)d_rows_out, cursor);
    });
}

int rhj_rows_filter_equal(rhj_ctx *ctx, const uint64_t *d_colA, const uint64_t *d_rowsA, const uint64_t *d_colB,
                          const uint64_t *d_rowsB, uint64_t n, uint64_t *d_pos_out, uint64_t *n_out)
{
    int rc = rhj_internal_use_device(ctx);
    if (rc != RHJ_OK) return rc;
    if (!n_out || (n && (!d_colA || !d_colB || !d_pos_out)))
        return rhj_internal_fail(ctx, RHJ_E_INVALID, "bad rhj_rows_filter_equal argument");
    *n_out = 0;
    if (n == 0) return RHJ_OK;
    return with_counter(ctx, n_out, "rhj_rows_filter_equal", [&](u64 *cursor, hipStream_t st) {
        hipLaunchKernelGGL(k_rows_filter_equal, dim3(grid_for(n)), dim3(256), 0, st, (const u64 *)d_colA, (const u64 *)d_rowsA,
                           (const u64 *)d_colB, (const u64 *)d_rowsB, (u64)n, (u64 *)d_pos_out, cursor);
    });
}

int rhj_sum_gather(rhj_ctx *ctx, const uint64_t *d_col, const uint64_t *d_rows, uint64_t n, uint64_t *sum)
{
    int rc = rhj_internal_use_device(ctx);
    if (rc != RHJ_OK) return rc;
    if (!sum || (n && !d_col)) return rhj_internal_fail(ctx, RHJ_E_INVALID, "bad rhj_sum_gather argument");
    *sum = 0;
    if (n == 0) return RHJ_OK;
    return with_counter(ctx, sum, "rhj_sum_gather", [&](u64 *acc, hipStream_t st) {
        hipLaunchKernelGGL(k_sum_gather, dim3(grid_for(n)), dim3(256), 0, st, (const u64 *)d_col, (const u64 *)d_rows, (u64)n, acc);
    });
}

int rhj_gather_tuples(rhj_ctx *ctx, const uint64_t *d_col, const uint64_t *d_rows, uint64_t n, int key_is_position,
                      rhj_tuple *d_tuples)
{
    int rc = rhj_internal_use_device(ctx);
    if (rc != RHJ_OK) return rc;
    if (n && (!d_col || !d_tuples)) return rhj_internal_fail(ctx, RHJ_E_INVALID, "bad rhj_gather_tuples argument");
    if (n == 0) return RHJ_OK;
    hipLaunchKernelGGL(k_gather_tuples, dim3(grid_for(n)), dim3(256), 0, rhj_internal_stream(ctx), (const u64 *)d_col,
                       (const u64 *)d_rows, (u64)n, key_is_position, (Tup *)d_tuples);
    return check(ctx, "rhj_gather_tuples");
}

int rhj_pairs_split(rhj_ctx *ctx, const rhj_pair *d_pairs, uint64_t n, uint64_t *d_r, uint64_t *d_s)
{
    int rc = rhj_internal_use_device(ctx);
    if (rc != RHJ_OK) return rc;
    if (n && (!d_pairs || !d_r || !d_s)) return rhj_internal_fail(ctx, RHJ_E_INVALID, "bad rhj_pairs_split argument");
    if (n == 0) return RHJ_OK;
    hipLaunchKernelGGL(k_pairs_split, dim3(grid_for(n)), dim3(256), 0, rhj_internal_stream(ctx), (const Pair *)d_pairs, (u64)n,
                       (u64 *)d_r, (u64 *)d_s);
    return check(ctx, "rhj_pairs_split");
}

int rhj_gather_u64(rhj_ctx *ctx, const uint64_t *d_src, const uint64_t *d_idx, uint64_t n, uint64_t *d_dst)
{
    int rc = rhj_internal_use_device(ctx);
    if (rc != RHJ_OK) return rc;
    if (n && (!d_src || !d_idx || !d_dst)) return rhj_internal_fail(ctx, RHJ_E_INVALID, "bad rhj_gather_u64 argument");
    if (n == 0) return RHJ_OK;
    hipLaunchKernelGGL(k_gather_u64, dim3(grid_for(n)), dim3(256), 0, rhj_internal_stream(ctx), (const u64 *)d_src,
                       (const u64 *)d_idx, (u64)n, (u64 *)d_dst);
    return check(ctx, "rhj_gather_u64");
}

}  // extern "C"
