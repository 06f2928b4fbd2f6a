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
constexpr int GP_CH = 8;              // channels per workgroup
constexpr int GP_VEC = 4;             // idx entries per thread per step (one float4 store)
constexpr int GP_STEPS = 4;           // steps per thread -> GP_TILE = 256*4*4 = 4096 entries
constexpr int GP_TILE = GP_THREADS * GP_VEC * GP_STEPS;
constexpr int GP_MAX_N_LDS = 4096;    // rows staged in LDS up to this n (GP_CH*n*4 = 128 KiB)

template <bool ROWS_IN_LDS>
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
    dim3 grid((unsigned)(tiles * (long long)b), cmf_divup(c, GP_CH));
    hipStream_t st = (hipStream_t)stream;
    if (n <= GP_MAX_N_LDS) {
        const size_t lds = (size_t)GP_CH * n * sizeof(float);
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute((const void *)group_points_kernel<true>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, GP_CH * GP_MAX_N_LDS * 4);
            attr_set = true;
        }
        hipLaunchKernelGGL(group_points_kernel<true>, grid, dim3(GP_THREADS), lds, st,
                           c, n, (int)total, tiles, points, idx, out);
    } else {
        hipLaunchKernelGGL(group_points_kernel<false>, grid, dim3(GP_THREADS), 0, st,
                           c, n, (int)total, tiles, points, idx, out);
    }
    return cmf_launch_status();
}

// ---------------------------------------------------------------------------------------------
// Backward: grad_points[b,c,idx[b,p,s]] += grad_out[b,c,p,s].
// The reference issues one global fp32 atomicAdd per element (538 M atomics for one scale of
// mse_layer2).  Here a workgroup owns (sample, GG_CH channels): it accumulates into an LDS copy
// of its rows with LDS atomics (ds_add_f32), reading grad_out with coalesced 16-byte loads, and
// finally adds the rows to grad_points with plain stores (each row has exactly one owner).
// Summation order inside a row is unordered, like the reference's atomics.
// ---------------------------------------------------------------------------------------------
constexpr int GG_THREADS = 256;
constexpr int GG_MAX_LDS_FLOATS = 32768;   // 128 KiB

__global__ __launch_bounds__(GG_THREADS) void group_points_grad_kernel(
    int c, int n, int total, int ch_per_block,
    const float *__restrict__ grad_out, const int *__restrict__ idx, float *__restrict__ grad_points)
{
    extern __shared__ __attribute__((aligned(16))) float acc[];    // [ch_per_block][n]
    const int bs = blockIdx.x;
    const int c0 = blockIdx.y * ch_per_block;
    const int nch = min(ch_per_block, c - c0);
    const int *ix = idx + (size_t)bs * total;
    const float *g = grad_out + ((size_t)bs * c + c0) * total;
    float *gp = grad_points + ((size_t)bs * c + c0) * n;

    for (int i = threadIdx.x; i < nch * n; i += GG_THREADS) acc[i] = 0.f;
    __syncthreads();
    const bool vec_ok = (total % 4) == 0;
    if (vec_ok) {
        for (int e = threadIdx.x * 4; e < total; e += GG_THREADS * 4) {
            const int4 j = *reinterpret_cast<const int4 *>(ix + e);
            for (int ch = 0; ch < nch; ++ch) {
                const float4 v = *reinterpret_cast<const float4 *>(g + (size_t)ch * total + e);
                float *row = acc + (size_t)ch * n;
                atomicAdd(row + j.x, v.x);
                atomicAdd(row + j.y, v.y);
                atomicAdd(row + j.z, v.z);
                atomicAdd(row + j.w, v.w);
            }
        }
    } else {
        for (int e = threadIdx.x; e < total; e += GG_THREADS) {
            const int j = ix[e];
            for (int ch = 0; ch < nch; ++ch) atomicAdd(acc + (size_t)ch * n + j, g[(size_t)ch * total + e]);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nch * n; i += GG_THREADS) gp[i] += acc[i];
}

// n too large for LDS rows: global atomics, one thread per element (the reference's scheme).
__global__ __launch_bounds__(GG_THREADS) void group_points_grad_global_kernel(
    int c, int n, int total, const float *__restrict__ grad_out, const int *__restrict__ idx,
    float *__restrict__ grad_points)
{
    const int bs = blockIdx.z, ch = blockIdx.y;
    const int e = blockIdx.x * GG_THREADS + threadIdx.x;
    if (e >= total) return;
    atomicAdd(grad_points + ((size_t)bs * c + ch) * n + idx[(size_t)bs * total + e],
              grad_out[((size_t)bs * c + ch) * total + e]);
}

extern "C" int cmf_group_points_grad(int b, int c, int n, int npoints, int nsample,
                                     const float *grad_out, const int *idx, float *grad_points, void *stream)
{
    CMF_CHECK_ARG(b >= 0 && c >= 0 && n >= 0 && npoints >= 0 && nsample >= 0);
    const long long total = (long long)npoints * nsample;
    if (b == 0 || c == 0 || total == 0) return 0;
    CMF_CHECK_ARG(grad_out && idx && grad_points && n > 0 && total < (1LL << 31));
    hipStream_t st = (hipStream_t)stream;
    if (n <= GG_MAX_LDS_FLOATS) {
        int chpb = GG_MAX_LDS_FLOATS / n;
        if (chpb > 8) chpb = 8;
        // keep >= ~1024 workgroups in flight when the problem allows it
        while (chpb > 1 && (long long)b * cmf_divup(c, chpb) < 1024) chpb >>= 1;
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute((const void *)group_points_grad_kernel,
                                hipFuncAttributeMaxDynamicSharedMemorySize, GG_MAX_LDS_FLOATS * 4);
            attr_set = true;
        }
        dim3 grid(b, cmf_divup(c, chpb));
        hipLaunchKernelGGL(group_points_grad_kernel, grid, dim3(GG_THREADS), (size_t)chpb * n * sizeof(float), st,
                           c, n, (int)total, chpb, grad_out, idx, grad_points);
    } else {
        dim3 grid(cmf_divup(total, GG_THREADS), c, b);
        hipLaunchKernelGGL(group_points_grad_global_kernel, grid, dim3(GG_THREADS), 0, st,
                           c, n, (int)total, grad_out, idx, grad_points);
    }
    return cmf_launch_status();
}
