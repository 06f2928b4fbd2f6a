// Grouping gather / scatter in the reference's (B,C,N) layout:
//   group_points_kernel_fast       lib/src/group_points_gpu.cu:47-66
//   group_points_grad_kernel_fast  lib/src/group_points_gpu.cu:8-25
//
// HBM-bound byte movers: the forward writes b*c*npoints*nsample*4 bytes and reads
// only b*c*n*4 + b*npoints*nsample*4.  The reference launches one thread per output float and
// re-reads idx once per channel with an uncoalesced gather from global memory.  Here a
// workgroup owns (sample, tile of GP_TILE idx entries, chunk of GP_CH channels): the idx tile is
// read ONCE into registers, the GP_CH feature rows are staged in LDS (n floats each), the random
// gather is served by LDS, and every global store is a coalesced 16-byte store.
#include "cmf_common.h"
#include "../../include/cmflow_hip.h"

constexpr int GP_THREADS = 256;
// channels per workgroup: 8 when the rows are short (8 rows of n floats in LDS), 2 for long rows so that
// several workgroups still fit on a CU (occupancy hides the LDS gather latency)
constexpr int GP_VEC = 4;             // idx entries per thread per step (one float4 store)
constexpr int GP_STEPS = 4;           // steps per thread -> GP_TILE = 256*4*4 = 4096 entries
constexpr int GP_TILE = GP_THREADS * GP_VEC * GP_STEPS;
constexpr int GP_MAX_N_LDS = 8192;    // rows staged in LDS up to this n

template <bool ROWS_IN_LDS, int GP_CH>
__global__ __launch_bounds__(GP_THREADS) void group_points_kernel(
    int c, int n, int total /* npoints*nsample */, int tiles_per_sample,
    const float *__restrict__ points, const int *__restrict__ idx, float *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) float rows[];   // [GP_CH][n] when ROWS_IN_LDS
    const int tile = blockIdx.x % tiles_per_sample;
    const int bs = blockIdx.x / tiles_per_sample;
    const int c0 = blockIdx.y * GP_CH;
    const int nch = min(GP_CH, c - c0);
    const int e0 = tile * GP_TILE;
    const int *ix = idx + (size_t)bs * total;
    const float *src = points + ((size_t)bs * c + c0) * n;
    float *dst = out + ((size_t)bs * c + c0) * total;

    if (ROWS_IN_LDS) {
        for (int i = threadIdx.x; i < nch * n; i += GP_THREADS) rows[i] = src[i];
    }
    // this thread's idx entries, loaded once and reused for every channel
    int my[GP_STEPS][GP_VEC];
    const bool vec_ok = (total % GP_VEC) == 0;
#pragma unroll
    for (int s = 0; s < GP_STEPS; ++s) {
        const int e = e0 + (s * GP_THREADS + threadIdx.x) * GP_VEC;
        if (vec_ok && e + GP_VEC <= total) {
            const int4 v = *reinterpret_cast<const int4 *>(ix + e);
            my[s][0] = v.x; my[s][1] = v.y; my[s][2] = v.z; my[s][3] = v.w;
        } else {
#pragma unroll
            for (int j = 0; j < GP_VEC; ++j) my[s][j] = (e + j < total) ? ix[e + j] : 0;
        }
    }
    if (ROWS_IN_LDS) __syncthreads();

    for (int ch = 0; ch < nch; ++ch) {
        const float *row = ROWS_IN_LDS ? rows + (size_t)ch * n : src + (size_t)ch * n;
        float *o = dst + (size_t)ch * total;
#pragma unroll
        for (int s = 0; s < GP_STEPS; ++s) {
            const int e = e0 + (s * GP_THREADS + threadIdx.x) * GP_VEC;
            if (vec_ok && e + GP_VEC <= total) {
                float4 v;
                v.x = row[my[s][0]]; v.y = row[my[s][1]]; v.z = row[my[s][2]]; v.w = row[my[s][3]];
                *reinterpret_cast<float4 *>(o + e) = v;
            } else {
#pragma unroll
                for (int j = 0; j < GP_VEC; ++j)
                    if (e + j < total) o[e + j] = row[my[s][j]];
            }
        }
    }
}

extern "C" int cmf_group_points(int b, int c, int n, int npoints, int nsample,
                                const float *points, const int *idx, float *out, void *stream)
{
    CMF_CHECK_ARG(b >= 0 && c >= 0 && n >= 0 && npoints >= 0 && nsample >= 0);
    const long long total = (long long)npoints * nsample;
    if (b == 0 || c == 0 || total == 0) return 0;
    CMF_CHECK_ARG(points && idx && out && n > 0 && total < (1LL << 31));
    const int tiles = cmf_divup(total, GP_TILE);
    hipStream_t st = (hipStream_t)stream;
    if (n <= 1024) {
        dim3 grid((unsigned)(tiles * (long long)b), cmf_divup(c, 8));
        hipLaunchKernelGGL((group_points_kernel<true, 8>), grid, dim3(GP_THREADS), (size_t)8 * n * sizeof(float), st,
                           c, n, (int)total, tiles, points, idx, out);
    } else if (n <= GP_MAX_N_LDS) {
        dim3 grid((unsigned)(tiles * (long long)b), cmf_divup(c, 2));
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute((const void *)group_points_kernel<true, 2>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, 2 * GP_MAX_N_LDS * 4);
            attr_set = true;
        }
        hipLaunchKernelGGL((group_points_kernel<true, 2>), grid, dim3(GP_THREADS), (size_t)2 * n * sizeof(float), st,
                           c, n, (int)total, tiles, points, idx, out);
    } else {
        dim3 grid((unsigned)(tiles * (long long)b), cmf_divup(c, 8));
        hipLaunchKernelGGL((group_points_kernel<false, 8>), grid, dim3(GP_THREADS), 0, st,
                           c, n, (int)total, tiles, points, idx, out);
    }
    return cmf_launch_status();
}

// ---------------------------------------------------------------------------------------------
// Backward: grad_points[b,c,idx[b,p,s]] += grad_out[b,c,p,s].
// The reference issues one global fp32 atomicAdd per element (538 M atomics for one scale of
// mse_layer2, order undefined).  Here the scatter becomes a GATHER over the inverse index of idx
// (cmf_build_inverse: for every target point the entries that reference it, ascending): a workgroup
// owns (sample, GG_CH channels), stages one grad_out row (npoints*nsample floats) in LDS with coalesced
// 16-byte loads, and thread j sums row[inv[t]] over its segment -- no atomics, fixed summation order
// (bit-reproducible and bit-equal to the CPU oracle's scan-order sum), each grad_points element written
// once with a coalesced store.
// ---------------------------------------------------------------------------------------------
int cmf_build_inverse_rows(int b, int n, int P, int S, const int *idx, int *offsets, int *inv, void *stream, int entries);

constexpr int GG_THREADS = 256;
constexpr int GG_CH = 8;
constexpr int GG_MAX_TOTAL_LDS = 12288;    // grad_out row + inverse list staged in LDS up to this many entries

template <bool IN_LDS>
__global__ __launch_bounds__(GG_THREADS) void group_points_grad_kernel(
    int c, int n, int total, const float *__restrict__ grad_out, const int *__restrict__ offsets,
    const int *__restrict__ inv, float *__restrict__ grad_points)
{
    extern __shared__ __attribute__((aligned(16))) int sm[];       // IN_LDS: [total] inv | [total] row
    const int bs = blockIdx.x;
    const int c0 = blockIdx.y * GG_CH;
    const int nch = min(GG_CH, c - c0);
    const int *off = offsets + (size_t)bs * (n + 1);
    const int *lst = inv + (size_t)bs * total;
    const float *g = grad_out + ((size_t)bs * c + c0) * total;
    float *gp = grad_points + ((size_t)bs * c + c0) * n;
    int *linv = sm;
    float *row = reinterpret_cast<float *>(sm + (IN_LDS ? total : 0));
    if (IN_LDS) {
        for (int i = threadIdx.x; i < total; i += GG_THREADS) linv[i] = lst[i];
    }
    for (int ch = 0; ch < nch; ++ch) {
        const float *src = g + (size_t)ch * total;
        if (IN_LDS) {
            __syncthreads();
            if ((total & 3) == 0)
                for (int i = threadIdx.x * 4; i < total; i += GG_THREADS * 4) *(float4 *)(row + i) = *(const float4 *)(src + i);
            else
                for (int i = threadIdx.x; i < total; i += GG_THREADS) row[i] = src[i];
            __syncthreads();
        }
        for (int j = threadIdx.x; j < n; j += GG_THREADS) {
            const int beg = off[j], end = off[j + 1];
            float acc = 0.f;
            for (int t = beg; t < end; ++t) acc += IN_LDS ? row[linv[t]] : src[lst[t]];
            gp[(size_t)ch * n + j] += acc;
        }
    }
}

extern "C" int cmf_group_points_grad(int b, int c, int n, int npoints, int nsample,
                                     const float *grad_out, const int *idx, float *grad_points, void *stream)
{
    CMF_CHECK_ARG(b >= 0 && c >= 0 && n >= 0 && npoints >= 0 && nsample >= 0);
    const long long total = (long long)npoints * nsample;
    if (b == 0 || c == 0 || total == 0) return 0;
    CMF_CHECK_ARG(grad_out && idx && grad_points && n > 0 && total < (1LL << 31));
    hipStream_t st = (hipStream_t)stream;
    // stream-ordered scratch for the inverse index (nothing is retained after the call)
    int *scratch = nullptr;
    const size_t n_off = (size_t)b * (n + 1), n_inv = (size_t)b * total;
    hipError_t e = hipMallocAsync((void **)&scratch, (n_off + n_inv) * sizeof(int), st);
    if (e != hipSuccess) return (int)e;
    int *offsets = scratch, *inv = scratch + n_off;
    int err = cmf_build_inverse_rows(b, n, npoints, nsample, idx, offsets, inv, stream, (int)total);
    if (!err) {
        dim3 grid(b, cmf_divup(c, GG_CH));
        if (total <= GG_MAX_TOTAL_LDS) {
            static bool attr_set = false;
            if (!attr_set) {
                (void)hipFuncSetAttribute((const void *)group_points_grad_kernel<true>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, GG_MAX_TOTAL_LDS * 8);
                attr_set = true;
            }
            hipLaunchKernelGGL(group_points_grad_kernel<true>, grid, dim3(GG_THREADS), (size_t)total * 8, st,
                               c, n, (int)total, grad_out, offsets, inv, grad_points);
        } else {
            hipLaunchKernelGGL(group_points_grad_kernel<false>, grid, dim3(GG_THREADS), 0, st,
                               c, n, (int)total, grad_out, offsets, inv, grad_points);
        }
        err = cmf_launch_status();
    }
    (void)hipFreeAsync(scratch, st);
    return err;
}
