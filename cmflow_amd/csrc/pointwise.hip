// Elementwise / reduction kernels around the GEMMs of the point-major path: BatchNorm statistics
// and folding, the set-conv gather (first conv hoisted per point), BN+ReLU+max over the ball,
// and their backward passes.  All are HBM-bound streams over [positions, channels] matrices with
// 16-byte accesses; per-channel reductions are tree-reduced per 128-row tile into a partial buffer
// [tiles][2][C] (deterministic: no atomics) and finished by cmf_bn_finalize / cmf_colsum_finalize.
#include <algorithm>
#include <cstdint>
#include "cmf_common.h"
#include "../../include/cmflow_hip.h"

constexpr int PW_THREADS = 256;
constexpr int PW_ROWS = 128;        // rows per tile == cmf_gemm's BM, so partial buffers are interchangeable

// ---------------------------------------------------------------------------------------------
// BN finalize.  partial [tiles][2][C] (sum, sumsq) over `count` rows ->
//   train: mean, invstd (biased var), running stats update (unbiased var, momentum), a = g*invstd,
//          c = b - mean*a.        eval (tiles == 0): a, c from the running statistics.
// ---------------------------------------------------------------------------------------------
// Cross-tile reduction shared by the two finalize kernels: a 256-thread workgroup owns 16 columns of
// the [tiles][ncols] partial matrix; 16 tile-lanes stride over the tiles (coalesced 64-byte reads),
// accumulate in double, and are combined through LDS in a fixed order (deterministic).
constexpr int FIN_COLS = 16, FIN_LANES = 16;
__device__ __forceinline__ double fin_reduce(int tiles, int ncols, int col, const float *__restrict__ partial, double *sh)
{
    const int cl = threadIdx.x % FIN_COLS, tl = threadIdx.x / FIN_COLS;
    double s = 0.0;
    if (col < ncols) {
        int t = tl;
        for (; t + 7 * FIN_LANES < tiles; t += 8 * FIN_LANES) {       // 8 loads in flight; order of the sum is fixed
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = partial[(size_t)(t + u * FIN_LANES) * ncols + col];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += (double)v[u];
        }
        for (; t < tiles; t += FIN_LANES) s += (double)partial[(size_t)t * ncols + col];
    }
    sh[tl * FIN_COLS + cl] = s;
    __syncthreads();
    double r = 0.0;
    if (tl == 0)
        for (int l = 0; l < FIN_LANES; ++l) r += sh[l * FIN_COLS + cl];
    __syncthreads();
    return r;          // valid for tl == 0
}

// The same reduction for 4 adjacent columns per workgroup (float4 reads; every caller inside the library has ncols % 4 == 0
// and 16-byte aligned rows): all 256 threads stride over the TILES, 8 loads in flight each -- 2048 tiles per round of
// memory latency instead of 128.  [With 16 tile-lanes a [4096][64] partial matrix (M = 524288 rows) took 32 dependent
// rounds: 16-40 us per call, twice per BN layer and direction, on the critical path of every set-conv chain.]
// The 256 per-thread sums are folded by a butterfly over the wave (fixed pattern) and the 4 waves in order: deterministic.
template <int G>                                   // G column groups of 4 at once (group g starts at col4[g])
__device__ __forceinline__ void fin_reduce4(int tiles, int ncols, const int (&col4)[G], const float *__restrict__ partial,
                                            double (*sh)[G][4], double (&out)[G][4])
{
    double s[G][4];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) s[g][i] = 0.0;
    int t = threadIdx.x;
    constexpr int U = 8 / G;
    for (; t + (U - 1) * 256 < tiles; t += U * 256) {
        float4 v[U][G];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int g = 0; g < G; ++g) v[u][g] = *(const float4 *)(partial + (size_t)(t + u * 256) * ncols + col4[g]);
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int g = 0; g < G; ++g) { s[g][0] += (double)v[u][g].x; s[g][1] += (double)v[u][g].y; s[g][2] += (double)v[u][g].z; s[g][3] += (double)v[u][g].w; }
    }
    for (; t < tiles; t += 256)
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const float4 v = *(const float4 *)(partial + (size_t)t * ncols + col4[g]);
            s[g][0] += (double)v.x; s[g][1] += (double)v.y; s[g][2] += (double)v.z; s[g][3] += (double)v.w;
        }
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) s[g][i] += __shfl_xor(s[g][i], off, 64);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int i = 0; i < 4; ++i) sh[wave][g][i] = s[g][i];
    __syncthreads();
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) out[g][i] = ((sh[0][g][i] + sh[1][g][i]) + sh[2][g][i]) + sh[3][g][i];
}

// The same reduction for 16 adjacent columns per 1024-thread workgroup (round 4): 256 row slots x 4 lanes of float4, so a wave
// reads 16 rows x 64 contiguous bytes (whole sectors) and 4096 tiles are ONE round of 16 loads per thread.  [The 4-column
// form above reads 16 bytes of every row per workgroup -- neighbouring workgroups re-fetch the same sectors: 55 us for the
// [4096][2560] matrix of a 524288-row data gradient, 760 GB/s; profiles/r04_finalize_probe.txt.]  Fixed order: per thread
// ascending tiles, butterfly over the 16 row slots of a wave, the 16 waves in order.  Result in res[G][16] (LDS).
constexpr int FIN16_THREADS = 1024;
template <int G>
__device__ __forceinline__ void fin_reduce16(int tiles, int ncols, const int (&col16)[G], const float *__restrict__ partial,
                                             double (*sh)[G][16], double (*res)[16])
{
    const int rl = threadIdx.x >> 2, cl = threadIdx.x & 3;
    double s[G][4];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) s[g][i] = 0.0;
    constexpr int U = 16 / G;
    constexpr int RS = FIN16_THREADS / 4;                            // row slots
    int t = rl;
    for (; t + (U - 1) * RS < tiles; t += U * RS) {
        float4 v[U][G];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int g = 0; g < G; ++g) v[u][g] = *(const float4 *)(partial + (size_t)(t + u * RS) * ncols + col16[g] + 4 * cl);
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int g = 0; g < G; ++g) { s[g][0] += (double)v[u][g].x; s[g][1] += (double)v[u][g].y; s[g][2] += (double)v[u][g].z; s[g][3] += (double)v[u][g].w; }
    }
    for (; t < tiles; t += RS)
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const float4 v = *(const float4 *)(partial + (size_t)t * ncols + col16[g] + 4 * cl);
            s[g][0] += (double)v.x; s[g][1] += (double)v.y; s[g][2] += (double)v.z; s[g][3] += (double)v.w;
        }
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int off = 32; off >= 4; off >>= 1) s[g][i] += __shfl_xor(s[g][i], off, 64);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) < 4)
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int i = 0; i < 4; ++i) sh[wave][g][4 * cl + i] = s[g][i];
    __syncthreads();
    if (threadIdx.x < 16 * G) {
        const int g = threadIdx.x >> 4, k = threadIdx.x & 15;
        double r = 0.0;
#pragma unroll
        for (int w = 0; w < FIN16_THREADS / 64; ++w) r += sh[w][g][k];
        res[g][k] = r;
    }
    __syncthreads();
}

// Which fold form a call takes.  The 16-column form reads whole sectors and needs one round of loads, but it has only ncols / 16
// (fat, 1024-thread) workgroups: it wins from about 2^20 partial sums up (4096 tiles x 256 columns: 4.1 against 5.4 us; 4096 x 2560:
// 16 against 55 us), below that the two are equal in isolation (tools/fin_sweep.py) and inside the step, next to other streams'
// kernels, the few fat workgroups of a narrow matrix (the first encoder's 32-64 channels) ran 2-3x slower than the 4-column form.
static inline bool fin_wide(long long tiles, long long ncols)
{
#ifdef CMF_FIN_EXPERIMENT
    if (const char *e = getenv("CMF_FIN_WIDE")) return atoi(e) != 0;
#endif
    return tiles * ncols >= (1ll << 20);
}
static inline bool fin_wide_bn(int tiles, int C) { return fin_wide(tiles, 2ll * C); }
static inline bool fin_wide_cs(int tiles, int ncols) { return fin_wide(tiles, ncols); }

// per-channel tail of bn_finalize from the channel's two sums
__device__ __forceinline__ void bn_finalize_channel(
    int ch, double s1, double s2, double count, const float *__restrict__ gamma, const float *__restrict__ beta,
    float eps, float momentum, float *__restrict__ running_mean, float *__restrict__ running_var, float *__restrict__ mean_out,
    float *__restrict__ invstd_out, float *__restrict__ a_out, float *__restrict__ c_out)
{
    double mean = s1 / count;
    double var = s2 / count - mean * mean;
    if (var < 0.0) var = 0.0;
    if (running_mean) {
        const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
        running_mean[ch] = (float)((1.0 - momentum) * running_mean[ch] + momentum * mean);
        running_var[ch] = (float)((1.0 - momentum) * running_var[ch] + momentum * unbiased);
    }
    const double invstd = 1.0 / sqrt(var + (double)eps);
    const double a = (gamma ? (double)gamma[ch] : 1.0) * invstd;
    if (mean_out) mean_out[ch] = (float)mean;
    if (invstd_out) invstd_out[ch] = (float)invstd;
    a_out[ch] = (float)a;
    c_out[ch] = (float)((beta ? (double)beta[ch] : 0.0) - mean * a);
}

// bn_finalize for C % 4 == 0: one workgroup per 4 channels, sums and sums of squares reduced together
__device__ __forceinline__ void bn_finalize4_body(
    int tiles, int C, double count, const float *__restrict__ partial,
    const float *__restrict__ gamma, const float *__restrict__ beta,
    float eps, float momentum, float *__restrict__ running_mean,
    float *__restrict__ running_var, float *__restrict__ mean_out,
    float *__restrict__ invstd_out, float *__restrict__ a_out, float *__restrict__ c_out,
    long long *__restrict__ num_batches_tracked, const int bx)
{
    __shared__ double sh[4][2][4];
    if (num_batches_tracked && bx == 0 && threadIdx.x == 0) *num_batches_tracked += 1;
    const int c0 = bx * 4;
    const int col4[2] = {c0, C + c0};
    double r[2][4];
    fin_reduce4<2>(tiles, 2 * C, col4, partial, sh, r);
    if (threadIdx.x >= 4) return;
    const int i = threadIdx.x;
    const double s1 = i == 0 ? r[0][0] : i == 1 ? r[0][1] : i == 2 ? r[0][2] : r[0][3];
    const double s2 = i == 0 ? r[1][0] : i == 1 ? r[1][1] : i == 2 ? r[1][2] : r[1][3];
    bn_finalize_channel(c0 + i, s1, s2, count, gamma, beta, eps, momentum, running_mean, running_var, mean_out, invstd_out, a_out, c_out);
}

// ... and for C % 16 == 0: one 1024-thread workgroup per 16 channels
__device__ __forceinline__ void bn_finalize16_body(
    int tiles, int C, double count, const float *__restrict__ partial,
    const float *__restrict__ gamma, const float *__restrict__ beta,
    float eps, float momentum, float *__restrict__ running_mean,
    float *__restrict__ running_var, float *__restrict__ mean_out,
    float *__restrict__ invstd_out, float *__restrict__ a_out, float *__restrict__ c_out,
    long long *__restrict__ num_batches_tracked, const int bx)
{
    __shared__ double sh[FIN16_THREADS / 64][2][16];
    __shared__ double res[2][16];
    if (num_batches_tracked && bx == 0 && threadIdx.x == 0) *num_batches_tracked += 1;
    const int c0 = bx * 16;
    const int col16[2] = {c0, C + c0};
    fin_reduce16<2>(tiles, 2 * C, col16, partial, sh, res);
    if (threadIdx.x >= 16) return;
    bn_finalize_channel(c0 + threadIdx.x, res[0][threadIdx.x], res[1][threadIdx.x], count, gamma, beta, eps, momentum, running_mean,
                        running_var, mean_out, invstd_out, a_out, c_out);
}

__global__ __launch_bounds__(FIN16_THREADS) void bn_finalize16_kernel(
    int tiles, int C, double count, const float *__restrict__ partial, const float *__restrict__ gamma, const float *__restrict__ beta,
    float eps, float momentum, float *__restrict__ running_mean, float *__restrict__ running_var, float *__restrict__ mean_out,
    float *__restrict__ invstd_out, float *__restrict__ a_out, float *__restrict__ c_out, long long *__restrict__ num_batches_tracked)
{
    bn_finalize16_body(tiles, C, count, partial, gamma, beta, eps, momentum, running_mean, running_var, mean_out, invstd_out, a_out, c_out,
                       num_batches_tracked, blockIdx.x);
}

__global__ __launch_bounds__(FIN16_THREADS) void bn_finalize16_batch_kernel(const CmfBatch<CmfBnFinArgs> b)
{
    const CmfBnFinArgs &p = b.a[blockIdx.y];
    if ((int)blockIdx.x * 16 >= p.C) return;
    bn_finalize16_body(p.tiles, p.C, p.count, p.partial, p.gamma, p.beta, p.eps, p.momentum, p.rmean, p.rvar, p.mean_out, p.invstd_out,
                       p.a_out, p.c_out, p.nbt, blockIdx.x);
}

__global__ __launch_bounds__(256) void bn_finalize4_kernel(
    int tiles, int C, double count, const float *__restrict__ partial, const float *__restrict__ gamma, const float *__restrict__ beta,
    float eps, float momentum, float *__restrict__ running_mean, float *__restrict__ running_var, float *__restrict__ mean_out,
    float *__restrict__ invstd_out, float *__restrict__ a_out, float *__restrict__ c_out, long long *__restrict__ num_batches_tracked)
{
    bn_finalize4_body(tiles, C, count, partial, gamma, beta, eps, momentum, running_mean, running_var, mean_out, invstd_out, a_out, c_out,
                      num_batches_tracked, blockIdx.x);
}

__global__ __launch_bounds__(256) void bn_finalize4_batch_kernel(const CmfBatch<CmfBnFinArgs> b)
{
    const CmfBnFinArgs &p = b.a[blockIdx.y];
    if ((int)blockIdx.x * 4 >= p.C) return;
    bn_finalize4_body(p.tiles, p.C, p.count, p.partial, p.gamma, p.beta, p.eps, p.momentum, p.rmean, p.rvar, p.mean_out, p.invstd_out,
                      p.a_out, p.c_out, p.nbt, blockIdx.x);
}

int cmf_bn_finalize_batch(int n, const CmfBnFinArgs *a, hipStream_t st)
{
    CMF_CHECK_ARG(n >= 1 && n <= CMF_MAX_BATCH && a);
    CmfBatch<CmfBnFinArgs> b;
    int cmax = 0;
    for (int i = 0; i < n; ++i) {
        CMF_CHECK_ARG(a[i].tiles > 0 && a[i].C > 0 && a[i].C % 4 == 0 && a[i].partial && (uintptr_t)a[i].partial % 16 == 0 && a[i].a_out && a[i].c_out);
        b.a[i] = a[i];
        cmax = std::max(cmax, a[i].C);
    }
    // every item's channel count a multiple of 16 and a large partial matrix among them: the coalesced 16-column form (fin_wide)
    bool wide = true;
    int tmax = 0;
    for (int i = 0; i < n; ++i) { wide = wide && a[i].C % 16 == 0; tmax = std::max(tmax, a[i].tiles); }
    if (wide && fin_wide_bn(tmax, cmax)) hipLaunchKernelGGL(bn_finalize16_batch_kernel, dim3(cmax / 16, n), dim3(FIN16_THREADS), 0, st, b);
    else hipLaunchKernelGGL(bn_finalize4_batch_kernel, dim3(cmax / 4, n), dim3(256), 0, st, b);
    return cmf_launch_status();
}

__global__ __launch_bounds__(256) void bn_finalize_kernel(
    int tiles, int C, double count, const float *__restrict__ partial,
    const float *__restrict__ gamma, const float *__restrict__ beta,
    float eps, float momentum, float *__restrict__ running_mean,
    float *__restrict__ running_var, float *__restrict__ mean_out,
    float *__restrict__ invstd_out, float *__restrict__ a_out, float *__restrict__ c_out,
    long long *__restrict__ num_batches_tracked)
{
    __shared__ double sh[FIN_COLS * FIN_LANES];
    if (num_batches_tracked && blockIdx.x == 0 && threadIdx.x == 0) *num_batches_tracked += 1;
    const int ch = blockIdx.x * FIN_COLS + threadIdx.x % FIN_COLS;
    double mean = 0.0, var = 1.0;
    if (tiles > 0) {
        const double s1 = fin_reduce(tiles, 2 * C, ch, partial, sh);                 // row layout [2][C]: sums
        const double s2 = fin_reduce(tiles, 2 * C, C + ch, partial, sh);             // then sums of squares
        mean = s1 / count;
        var = s2 / count - mean * mean;
        if (var < 0.0) var = 0.0;
    }
    if (threadIdx.x >= FIN_COLS || ch >= C) return;
    if (tiles > 0) {
        if (running_mean) {
            const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
            running_mean[ch] = (float)((1.0 - momentum) * running_mean[ch] + momentum * mean);
            running_var[ch] = (float)((1.0 - momentum) * running_var[ch] + momentum * unbiased);
        }
    } else { mean = running_mean[ch]; var = running_var[ch]; }
    const double invstd = 1.0 / sqrt(var + (double)eps);
    const double a = (gamma ? (double)gamma[ch] : 1.0) * invstd;
    if (mean_out) mean_out[ch] = (float)mean;
    if (invstd_out) invstd_out[ch] = (float)invstd;
    a_out[ch] = (float)a;
    c_out[ch] = (float)((beta ? (double)beta[ch] : 0.0) - mean * a);
}

extern "C" int cmf_bn_finalize(int tiles, int C, double count, const float *partial, const float *gamma,
                               const float *beta, float eps, float momentum, float *running_mean,
                               float *running_var, float *mean_out, float *invstd_out, float *a_out,
                               float *c_out, long long *num_batches_tracked, void *stream)
{
    CMF_CHECK_ARG(C > 0 && a_out && c_out && (tiles == 0 ? (running_mean && running_var) : partial != nullptr));
    if (fin_wide_bn(tiles, C) && C % 16 == 0 && (uintptr_t)partial % 16 == 0) {
        hipLaunchKernelGGL(bn_finalize16_kernel, dim3(C / 16), dim3(FIN16_THREADS), 0, (hipStream_t)stream, tiles, C, count,
                           partial, gamma, beta, eps, momentum, running_mean, running_var, mean_out, invstd_out, a_out, c_out,
                           num_batches_tracked);
        return cmf_launch_status();
    }
    if (tiles > 0 && C % 4 == 0 && (uintptr_t)partial % 16 == 0) {
        hipLaunchKernelGGL(bn_finalize4_kernel, dim3(C / 4), dim3(256), 0, (hipStream_t)stream, tiles, C, count,
                           partial, gamma, beta, eps, momentum, running_mean, running_var, mean_out, invstd_out, a_out, c_out,
                           num_batches_tracked);
        return cmf_launch_status();
    }
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(cmf_divup(C, FIN_COLS)), dim3(256), 0, (hipStream_t)stream, tiles, C, count,
                       partial, gamma, beta, eps, momentum, running_mean, running_var, mean_out, invstd_out, a_out, c_out,
                       num_batches_tracked);
    return cmf_launch_status();
}

// column sums of the partial buffer: out[2][C] = sum_t partial[t][2][C]  (dbeta, dgamma in backward)
// acc0 / acc1 (optional): accumulate columns [0,C) into acc0 and [C,2C) into acc1 -- the BN-backward sums go
// straight into beta.grad / gamma.grad (one launch instead of a reduction plus two tiny torch adds).
__global__ __launch_bounds__(256) void colsum_finalize_kernel(int tiles, int C2, const float *__restrict__ partial, float *__restrict__ out,
                                                              int C, float *__restrict__ acc0, float *__restrict__ acc1, int store = 0)
{
    __shared__ double sh[FIN_COLS * FIN_LANES];
    const int col = blockIdx.x * FIN_COLS + threadIdx.x % FIN_COLS;
    const double s = fin_reduce(tiles, C2, col, partial, sh);
    if (threadIdx.x < FIN_COLS && col < C2) {
        out[col] = (float)s;
        if (acc0 && col < C) acc0[col] = store ? (float)s : acc0[col] + (float)s;
        if (acc1 && col >= C && col < 2 * C) acc1[col - C] = store ? (float)s : acc1[col - C] + (float)s;
    }
}

__device__ __forceinline__ void colsum_finalize4_body(int tiles, int C2, const float *__restrict__ partial, float *__restrict__ out,
                                                      int C, float *__restrict__ acc0, float *__restrict__ acc1, int store, const int bx)
{
    __shared__ double sh[4][1][4];
    const int col4[1] = {bx * 4};
    double r[1][4];
    fin_reduce4<1>(tiles, C2, col4, partial, sh, r);
    if (threadIdx.x >= 4) return;
    const int i = threadIdx.x, col = col4[0] + i;
    const double s = i == 0 ? r[0][0] : i == 1 ? r[0][1] : i == 2 ? r[0][2] : r[0][3];
    out[col] = (float)s;
    if (acc0 && col < C) acc0[col] = store ? (float)s : acc0[col] + (float)s;
    if (acc1 && col >= C && col < 2 * C) acc1[col - C] = store ? (float)s : acc1[col - C] + (float)s;
}

__device__ __forceinline__ void colsum_finalize16_body(int tiles, int C2, const float *__restrict__ partial, float *__restrict__ out,
                                                       int C, float *__restrict__ acc0, float *__restrict__ acc1, int store, const int bx)
{
    __shared__ double sh[FIN16_THREADS / 64][1][16];
    __shared__ double res[1][16];
    const int col16[1] = {bx * 16};
    fin_reduce16<1>(tiles, C2, col16, partial, sh, res);
    if (threadIdx.x >= 16) return;
    const int col = col16[0] + threadIdx.x;
    const double s = res[0][threadIdx.x];
    out[col] = (float)s;
    if (acc0 && col < C) acc0[col] = store ? (float)s : acc0[col] + (float)s;
    if (acc1 && col >= C && col < 2 * C) acc1[col - C] = store ? (float)s : acc1[col - C] + (float)s;
}

__global__ __launch_bounds__(FIN16_THREADS) void colsum_finalize16_kernel(int tiles, int C2, const float *__restrict__ partial, float *__restrict__ out,
                                                                          int C, float *__restrict__ acc0, float *__restrict__ acc1, int store)
{
    colsum_finalize16_body(tiles, C2, partial, out, C, acc0, acc1, store, blockIdx.x);
}

__global__ __launch_bounds__(FIN16_THREADS) void colsum_finalize16_batch_kernel(const CmfBatch<CmfColsumArgs> b)
{
    const CmfColsumArgs &p = b.a[blockIdx.y];
    if ((int)blockIdx.x * 16 >= p.ncols) return;
    colsum_finalize16_body(p.tiles, p.ncols, p.partial, p.out, p.C, p.acc0, p.acc1, p.store, blockIdx.x);
}

__global__ __launch_bounds__(256) void colsum_finalize4_kernel(int tiles, int C2, const float *__restrict__ partial, float *__restrict__ out,
                                                               int C, float *__restrict__ acc0, float *__restrict__ acc1, int store)
{
    colsum_finalize4_body(tiles, C2, partial, out, C, acc0, acc1, store, blockIdx.x);
}

__global__ __launch_bounds__(256) void colsum_finalize4_batch_kernel(const CmfBatch<CmfColsumArgs> b)
{
    const CmfColsumArgs &p = b.a[blockIdx.y];
    if ((int)blockIdx.x * 4 >= p.ncols) return;
    colsum_finalize4_body(p.tiles, p.ncols, p.partial, p.out, p.C, p.acc0, p.acc1, p.store, blockIdx.x);
}

int cmf_colsum_batch(int n, const CmfColsumArgs *a, hipStream_t st)
{
    CMF_CHECK_ARG(n >= 1 && n <= CMF_MAX_BATCH && a);
    CmfBatch<CmfColsumArgs> b;
    int cmax = 0;
    for (int i = 0; i < n; ++i) {
        CMF_CHECK_ARG(a[i].tiles > 0 && a[i].ncols > 0 && a[i].ncols % 4 == 0 && a[i].partial && (uintptr_t)a[i].partial % 16 == 0 && a[i].out);
        b.a[i] = a[i];
        cmax = std::max(cmax, a[i].ncols);
    }
    bool wide = true;
    int tmax = 0;
    for (int i = 0; i < n; ++i) { wide = wide && a[i].ncols % 16 == 0; tmax = std::max(tmax, a[i].tiles); }
    if (wide && fin_wide_cs(tmax, cmax)) hipLaunchKernelGGL(colsum_finalize16_batch_kernel, dim3(cmax / 16, n), dim3(FIN16_THREADS), 0, st, b);
    else hipLaunchKernelGGL(colsum_finalize4_batch_kernel, dim3(cmax / 4, n), dim3(256), 0, st, b);
    return cmf_launch_status();
}

static int launch_colsum(int tiles, int ncols, const float *partial, float *out, int C, float *a0, float *a1, int store, void *stream)
{
    if (fin_wide_cs(tiles, ncols) && ncols % 16 == 0 && (uintptr_t)partial % 16 == 0)
        hipLaunchKernelGGL(colsum_finalize16_kernel, dim3(ncols / 16), dim3(FIN16_THREADS), 0, (hipStream_t)stream, tiles, ncols, partial, out, C, a0, a1, store);
    else if (ncols % 4 == 0 && (uintptr_t)partial % 16 == 0)
        hipLaunchKernelGGL(colsum_finalize4_kernel, dim3(ncols / 4), dim3(256), 0, (hipStream_t)stream, tiles, ncols, partial, out, C, a0, a1, store);
    else
        hipLaunchKernelGGL(colsum_finalize_kernel, dim3(cmf_divup(ncols, FIN_COLS)), dim3(256), 0, (hipStream_t)stream,
                           tiles, ncols, partial, out, C, a0, a1, store);
    return cmf_launch_status();
}

extern "C" int cmf_colsum_finalize(int tiles, int C, const float *partial, float *out, float *acc0, float *acc1, void *stream)
{
    CMF_CHECK_ARG(tiles > 0 && C > 0 && partial && out);
    return launch_colsum(tiles, 2 * C, partial, out, C, acc0, acc1, 0, stream);
}

extern "C" int cmf_colsum(int tiles, int ncols, const float *partial, float *out, int C, float *acc0, float *acc1, void *stream)
{
    CMF_CHECK_ARG(tiles > 0 && ncols > 0 && partial && out);
    return launch_colsum(tiles, ncols, partial, out, C, acc0, acc1, 0, stream);
}

// internal (csrc/setconv_block.hip): the same reduction with the first 2*C columns STORED to dst0 / dst1 instead of added --
// a block whose BN gradients go to caller buffers gets them from this launch instead of two device-to-device copies
int cmf_colsum_store(int tiles, int ncols, const float *partial, float *out, int C, float *dst0, float *dst1, void *stream)
{
    CMF_CHECK_ARG(tiles > 0 && ncols > 0 && partial && out);
    return launch_colsum(tiles, ncols, partial, out, C, dst0, dst1, 1, stream);
}

// dW_xyz of the set-conv first layer from column sums (see cmflow_hip.h)
__device__ __forceinline__ void setconv_dwx_body(int C, float inv_count, int train, const float *__restrict__ bwd5,
                                                 const float *__restrict__ fwd, const float *__restrict__ a,
                                                 const float *__restrict__ mean, const float *__restrict__ invstd, float *__restrict__ dwx,
                                                 int ld, int accumulate)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float s1 = bwd5[c], s2 = bwd5[C + c];
    for (int k = 0; k < 3; ++k) {
        const float q = bwd5[(2 + k) * C + c];
        float g = q;
        if (train) {
            const float u = fwd[3 * C + k], tz = fwd[k * C + c];
            g = q - (s1 * inv_count) * u - (s2 * inv_count) * invstd[c] * (tz - mean[c] * u);
        }
        float *dst = dwx + (size_t)c * ld + k;
        *dst = accumulate ? *dst + a[c] * g : a[c] * g;
    }
}

__global__ void setconv_dwx_kernel(int C, float inv_count, int train, const float *__restrict__ bwd5,
                                   const float *__restrict__ fwd, const float *__restrict__ a,
                                   const float *__restrict__ mean, const float *__restrict__ invstd, float *__restrict__ dwx,
                                   int ld, int accumulate)
{
    setconv_dwx_body(C, inv_count, train, bwd5, fwd, a, mean, invstd, dwx, ld, accumulate);
}

__global__ void setconv_dwx_batch_kernel(const CmfBatch<CmfDwxArgs> b)
{
    const CmfDwxArgs &p = b.a[blockIdx.y];
    setconv_dwx_body(p.C, p.inv_count, p.train, p.bwd5, p.fwd, p.a, p.mean, p.invstd, p.dwx, p.ld, p.accumulate);
}

int cmf_setconv_dwx_batch(int n, const CmfDwxArgs *a, hipStream_t st)
{
    CMF_CHECK_ARG(n >= 1 && n <= CMF_MAX_BATCH && a);
    CmfBatch<CmfDwxArgs> b;
    int cmax = 0;
    for (int i = 0; i < n; ++i) {
        const CmfDwxArgs &q = a[i];
        CMF_CHECK_ARG(q.C > 0 && q.bwd5 && q.a && q.dwx && q.ld >= 3 && (!q.train || (q.fwd && q.mean && q.invstd)));
        b.a[i] = q;
        cmax = std::max(cmax, q.C);
    }
    hipLaunchKernelGGL(setconv_dwx_batch_kernel, dim3(cmf_divup(cmax, 64), n), dim3(64), 0, st, b);
    return cmf_launch_status();
}

extern "C" int cmf_setconv_dwx(int C, float inv_count, int train, const float *bwd5, const float *fwd,
                               const float *a, const float *mean, const float *invstd, float *dwx, int ld, int accumulate,
                               void *stream)
{
    CMF_CHECK_ARG(C > 0 && bwd5 && a && dwx && ld >= 3 && (!train || (fwd && mean && invstd)));
    hipLaunchKernelGGL(setconv_dwx_kernel, dim3(cmf_divup(C, 64)), dim3(64), 0, (hipStream_t)stream,
                       C, inv_count, train, bwd5, fwd, a, mean, invstd, dwx, ld, accumulate);
    return cmf_launch_status();
}

// ---------------------------------------------------------------------------------------------
// Tile-shaped streaming helper: a workgroup owns PW_ROWS rows x all C columns; thread t owns the
// float4 column group (t % CG) and the rows (t / CG) + k*RL.  Column partial sums are reduced over
// the RL row-lanes through LDS and written to partial[tile][2][C].
// ---------------------------------------------------------------------------------------------
struct TileMap { int cg, rl, col, r0; };
__device__ __forceinline__ TileMap tile_map(int C)
{
    TileMap t;
    t.cg = C / 4; t.rl = PW_THREADS / t.cg; t.col = (threadIdx.x % t.cg) * 4; t.r0 = threadIdx.x / t.cg;
    return t;
}
__device__ __forceinline__ void tile_reduce_store(float4 s1, float4 s2, const TileMap &tm, int C, float *partial, float *red,
                                                  long long tile = -1)
{
    if (tile < 0) tile = blockIdx.x;
    // red: [rl][2][C]
    float *r = red + (size_t)tm.r0 * 2 * C;
    *(float4 *)(r + tm.col) = s1;
    *(float4 *)(r + C + tm.col) = s2;
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += PW_THREADS) {
        float s = 0.f;
        for (int l = 0; l < tm.rl; ++l) s += red[(size_t)l * 2 * C + i];
        partial[(size_t)tile * 2 * C + i] = s;
    }
}
// Workgroup id -> XCD is round-robin (id % 8), each XCD with its own L2.  A kernel whose tiles gather from per-sample
// sources wants the tiles of one sample on ONE XCD, or every XCD fetches every sample's sources from HBM: XCD x gets the
// contiguous range of logical tiles [x * q + min(x, r), ...) with q = tiles / 8, r = tiles % 8, walked in launch order.
__device__ __forceinline__ long long xcd_major_tile(long long bid, long long tiles)
{
    const long long q = tiles / 8, r = tiles % 8, x = bid % 8, slot = bid / 8;
    return x * q + (x < r ? x : r) + slot;
}
static inline bool tile_ok(int C) { return C % 4 == 0 && C >= 4 && C / 4 <= PW_THREADS && PW_THREADS % (C / 4) == 0; }
static inline size_t tile_lds(int C) { return (size_t)(PW_THREADS / (C / 4)) * 2 * C * sizeof(float); }

// ---------------------------------------------------------------------------------------------
// Set-conv / cost-volume gather with the first conv hoisted:
//   z[b,p,s,:] = act( ysrc[b, idx[b,p,s], :] + (yctr ? yctr[b,p,:] : 0) + Wx (xyz_src[idx] - xyz_ctr[p]) )
// plus BN partial statistics of z, and the relative coordinates dxyz (B,P,S,4) (4th lane 0) that the
// WeightNet and the weight-gradient GEMM consume.   (radarflow_util.py:148-151 / :207-214)
// ---------------------------------------------------------------------------------------------
// accumulators of one thread (its 4 columns over its rows of the tile)
struct GaSums { float4 s1, s2, tz0, tz1, tz2; float ud0, ud1, ud2; };

// The rows of a tile whose 128 rows all exist, specialised by what the call needs (CTR: centre rows added, ST: BN partial sums,
// XS: the z * d_k sums of the set-conv dW_xyz).  No bounds checks, no 64-bit divisions and no branches in the loop: the generic
// loop below (ragged last tile, narrow C) carried four `row / S` long divisions and four validity branches per step, and the
// kernel was bound by instruction issue, not by its stores.
template <bool CTR, bool ST, bool XS, bool WR>
__device__ __forceinline__ void ga_full_rows(const TileMap &tm, int C, const float *__restrict__ ysrc, int ld_src, const float *__restrict__ yctr,
                                             int ld_ctr, const float (&wx)[4][3], int act, float *__restrict__ zt /* z + row0 * C + col */,
                                             const float4 *sd, const int *ssrc, const int *sctr, GaSums &a)
{
    const float *ys = ysrc + tm.col;
    const float *yc = CTR ? yctr + tm.col : nullptr;
    // [requesting the gathers of step i + 1 before the stores of step i (two register sets) was measured equal: 227-238 us against
    //  234 us at 524288 rows; the kernel runs at 4.7 TB/s of stores next to 1 GB of gathers from L2, a plain fill at 6.8 TB/s]
    for (int rb = tm.r0; rb < PW_ROWS; rb += 4 * tm.rl) {
        float4 v[4], cc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = rb + u * tm.rl;
            v[u] = *(const float4 *)(ys + (size_t)ssrc[r] * ld_src);
            if (CTR) cc[u] = *(const float4 *)(yc + (size_t)sctr[r] * ld_ctr);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = rb + u * tm.rl;
            const float4 d4 = sd[r];
            const float dx = d4.x, dy = d4.y, dz = d4.z;
            float4 o = v[u];
            if (CTR) { o.x += cc[u].x; o.y += cc[u].y; o.z += cc[u].z; o.w += cc[u].w; }
            o.x += fmaf(wx[0][2], dz, fmaf(wx[0][1], dy, wx[0][0] * dx));
            o.y += fmaf(wx[1][2], dz, fmaf(wx[1][1], dy, wx[1][0] * dx));
            o.z += fmaf(wx[2][2], dz, fmaf(wx[2][1], dy, wx[2][0] * dx));
            o.w += fmaf(wx[3][2], dz, fmaf(wx[3][1], dy, wx[3][0] * dx));
            if (act == 2) { o.x = o.x > 0.f ? o.x : 0.1f * o.x; o.y = o.y > 0.f ? o.y : 0.1f * o.y;
                            o.z = o.z > 0.f ? o.z : 0.1f * o.z; o.w = o.w > 0.f ? o.w : 0.1f * o.w; }
            if (WR) *(float4 *)(zt + (size_t)r * C) = o;
            if (ST) {
                // (explicit fused multiply-adds: every instantiation of this body -- stored / statistics-only / batched -- must sum alike)
                a.s1.x += o.x; a.s1.y += o.y; a.s1.z += o.z; a.s1.w += o.w;
                a.s2.x = fmaf(o.x, o.x, a.s2.x); a.s2.y = fmaf(o.y, o.y, a.s2.y); a.s2.z = fmaf(o.z, o.z, a.s2.z); a.s2.w = fmaf(o.w, o.w, a.s2.w);
            }
            if (XS) {
                a.tz0.x = fmaf(o.x, dx, a.tz0.x); a.tz0.y = fmaf(o.y, dx, a.tz0.y); a.tz0.z = fmaf(o.z, dx, a.tz0.z); a.tz0.w = fmaf(o.w, dx, a.tz0.w);
                a.tz1.x = fmaf(o.x, dy, a.tz1.x); a.tz1.y = fmaf(o.y, dy, a.tz1.y); a.tz1.z = fmaf(o.z, dy, a.tz1.z); a.tz1.w = fmaf(o.w, dy, a.tz1.w);
                a.tz2.x = fmaf(o.x, dz, a.tz2.x); a.tz2.y = fmaf(o.y, dz, a.tz2.y); a.tz2.z = fmaf(o.z, dz, a.tz2.z); a.tz2.w = fmaf(o.w, dz, a.tz2.w);
                a.ud0 += dx; a.ud1 += dy; a.ud2 += dz;
            }
        }
    }
}

// (body + single / batch entry: cmf_common.h "batched launches"; bx / nbx = this problem's block index and block count)
template <bool CTR, bool ST, bool XS, bool WR = true>   // centre rows added / BN partial sums / z * d_k sums / z written (the call's NULL pointers)
__device__ __forceinline__ void group_affine_body(
    int n_src, int P, int S, int C, long long rows,
    const float *__restrict__ ysrc, int ld_src, const float *__restrict__ yctr, int ld_ctr,
    const float *__restrict__ xyz_src, const float *__restrict__ xyz_ctr,
    const float *__restrict__ Wx, int ldw, const int *__restrict__ idx, int act,
    float *__restrict__ z, float *__restrict__ dxyz, float *__restrict__ partial, float *__restrict__ partial_x, const int bx, const int nbx)
{
    extern __shared__ __attribute__((aligned(16))) float red[];
    const TileMap tm = tile_map(C);
    GaSums acc;
    acc.s1 = acc.s2 = acc.tz0 = acc.tz1 = acc.tz2 = make_float4(0.f, 0.f, 0.f, 0.f);   // tz*: sum z*dx, z*dy, z*dz per channel
    acc.ud0 = acc.ud1 = acc.ud2 = 0.f;                                                  // sum dx, dy, dz (column group 0 only)
    float wx[4][3];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int k = 0; k < 3; ++k) wx[j][k] = Wx[(size_t)(tm.col + j) * ldw + k];
    // tiles of one sample on one XCD: the sample's n_src x C source rows (512 KB at C = 512) are fetched from HBM by that
    // L2 only (round 2 counted 1.41x the algorithmic bytes: all eight XCDs pulled all sources)
    const long long tile = xcd_major_tile(bx, nbx);
    const long long row0 = tile * PW_ROWS;
    // Per-row quantities (source row, centre row, relative xyz) are computed ONCE per tile into LDS: with C = 512 the 128 threads
    // that share a row used to issue the same idx load and six scalar xyz loads each -- nine memory instructions per
    // useful 16-byte gather, and the kernel was bound by their issue rate (2.5 TB/s of stores at C = 512).
    __shared__ float4 sd[PW_ROWS];
    __shared__ int ssrc[PW_ROWS], sctr[PW_ROWS];
    for (int r = threadIdx.x; r < PW_ROWS; r += PW_THREADS) {
        const long long row = row0 + r;
        if (row < rows) {
            const int j = idx[row];
            const long long bp = row / S;
            const int b = (int)(bp / P);
            const float *xs = xyz_src + ((size_t)b * n_src + j) * 3;
            const float *xc = xyz_ctr + (size_t)bp * 3;
            const float4 d4 = make_float4(xs[0] - xc[0], xs[1] - xc[1], xs[2] - xc[2], 0.f);
            sd[r] = d4;
            ssrc[r] = b * n_src + j;
            sctr[r] = (int)bp;
            if (dxyz) *(float4 *)(dxyz + (size_t)row * 4) = d4;
        }
    }
    __syncthreads();
    if (row0 + PW_ROWS <= rows && 4 * tm.rl <= PW_ROWS)
        ga_full_rows<CTR, ST, XS, WR>(tm, C, ysrc, ld_src, yctr, ld_ctr, wx, act, z + (size_t)row0 * C + tm.col, sd, ssrc, sctr, acc);
    else
    // the generic loop (ragged last tile; C < 32: fewer than four row slots' worth of rows per step); 4 rows per step in flight
    for (int rb = tm.r0; rb < PW_ROWS; rb += 4 * tm.rl) {
        bool ok[4];
        float4 v[4], cc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = rb + u * tm.rl;
            ok[u] = (r < PW_ROWS) && row0 + r < rows;
            v[u] = *(const float4 *)(ysrc + (size_t)(ok[u] ? ssrc[r] : 0) * ld_src + tm.col);
            cc[u] = CTR ? *(const float4 *)(yctr + (size_t)(ok[u] ? sctr[r] : 0) * ld_ctr + tm.col) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (!ok[u]) continue;
            const int r = rb + u * tm.rl;
            const float4 d4 = sd[r];
            const float dx = d4.x, dy = d4.y, dz = d4.z;
            float4 o = v[u];
            o.x += cc[u].x; o.y += cc[u].y; o.z += cc[u].z; o.w += cc[u].w;
            o.x += fmaf(wx[0][2], dz, fmaf(wx[0][1], dy, wx[0][0] * dx));
            o.y += fmaf(wx[1][2], dz, fmaf(wx[1][1], dy, wx[1][0] * dx));
            o.z += fmaf(wx[2][2], dz, fmaf(wx[2][1], dy, wx[2][0] * dx));
            o.w += fmaf(wx[3][2], dz, fmaf(wx[3][1], dy, wx[3][0] * dx));
            if (act == 2) { o.x = o.x > 0.f ? o.x : 0.1f * o.x; o.y = o.y > 0.f ? o.y : 0.1f * o.y;
                            o.z = o.z > 0.f ? o.z : 0.1f * o.z; o.w = o.w > 0.f ? o.w : 0.1f * o.w; }
            if (WR) *(float4 *)(z + (size_t)(row0 + r) * C + tm.col) = o;
            if (ST) {
                acc.s1.x += o.x; acc.s1.y += o.y; acc.s1.z += o.z; acc.s1.w += o.w;
                acc.s2.x = fmaf(o.x, o.x, acc.s2.x); acc.s2.y = fmaf(o.y, o.y, acc.s2.y); acc.s2.z = fmaf(o.z, o.z, acc.s2.z); acc.s2.w = fmaf(o.w, o.w, acc.s2.w);
            }
            if (XS) {
                acc.tz0.x = fmaf(o.x, dx, acc.tz0.x); acc.tz0.y = fmaf(o.y, dx, acc.tz0.y); acc.tz0.z = fmaf(o.z, dx, acc.tz0.z); acc.tz0.w = fmaf(o.w, dx, acc.tz0.w);
                acc.tz1.x = fmaf(o.x, dy, acc.tz1.x); acc.tz1.y = fmaf(o.y, dy, acc.tz1.y); acc.tz1.z = fmaf(o.z, dy, acc.tz1.z); acc.tz1.w = fmaf(o.w, dy, acc.tz1.w);
                acc.tz2.x = fmaf(o.x, dz, acc.tz2.x); acc.tz2.y = fmaf(o.y, dz, acc.tz2.y); acc.tz2.z = fmaf(o.z, dz, acc.tz2.z); acc.tz2.w = fmaf(o.w, dz, acc.tz2.w);
                acc.ud0 += dx; acc.ud1 += dy; acc.ud2 += dz;
            }
        }
    }
    const float4 s1 = acc.s1, s2 = acc.s2, tz0 = acc.tz0, tz1 = acc.tz1, tz2 = acc.tz2;
    const float ud0 = acc.ud0, ud1 = acc.ud1, ud2 = acc.ud2;
    if (ST) tile_reduce_store(s1, s2, tm, C, partial, red, tile);
    if (XS) {
        // same tree as tile_reduce_store, for the 3 z*d_k column sums and the 3 scalar d_k sums: row
        // layout of partial_x is [tz0[C] | tz1[C] | tz2[C] | u0 u1 u2 0]
        __syncthreads();
        float *r = red + (size_t)tm.r0 * 2 * C;          // reuse [rl][2][C]: two passes
        float *px = partial_x + (size_t)tile * (3 * C + 4);
        *(float4 *)(r + tm.col) = tz0; *(float4 *)(r + C + tm.col) = tz1;
        __syncthreads();
        for (int i = threadIdx.x; i < 2 * C; i += PW_THREADS) {
            float sum = 0.f;
            for (int l = 0; l < tm.rl; ++l) sum += red[(size_t)l * 2 * C + i];
            px[i] = sum;
        }
        __syncthreads();
        *(float4 *)(r + tm.col) = tz2;
        if (tm.col == 0) *(float4 *)(r + C) = make_float4(ud0, ud1, ud2, 0.f);
        __syncthreads();
        for (int i = threadIdx.x; i < C + 4; i += PW_THREADS) {
            float sum = 0.f;
            for (int l = 0; l < tm.rl; ++l) sum += red[(size_t)l * 2 * C + i];
            px[2 * C + i] = sum;
        }
    }
}

template <bool CTR, bool ST, bool XS, bool WR = true>
__global__ __launch_bounds__(PW_THREADS) void group_affine_kernel(
    int n_src, int P, int S, int C, long long rows,
    const float *__restrict__ ysrc, int ld_src, const float *__restrict__ yctr, int ld_ctr,
    const float *__restrict__ xyz_src, const float *__restrict__ xyz_ctr,
    const float *__restrict__ Wx, int ldw, const int *__restrict__ idx, int act,
    float *__restrict__ z, float *__restrict__ dxyz, float *__restrict__ partial, float *__restrict__ partial_x)
{
    group_affine_body<CTR, ST, XS, WR>(n_src, P, S, C, rows, ysrc, ld_src, yctr, ld_ctr, xyz_src, xyz_ctr, Wx, ldw, idx, act, z, dxyz, partial,
                                       partial_x, blockIdx.x, gridDim.x);
}

// the narrow blocks' first layer (no centre rows, statistics + z * d_k sums, z written) for up to CMF_MAX_BATCH blocks of one width
__global__ __launch_bounds__(PW_THREADS) void group_affine_batch_kernel(const CmfBatch<CmfGroupAffineArgs> b)
{
    const CmfGroupAffineArgs &p = b.a[blockIdx.y];
    const long long rows = (long long)p.b * p.P * p.S;
    const int tiles = (int)((rows + PW_ROWS - 1) / PW_ROWS);
    if ((int)blockIdx.x >= tiles) return;
    group_affine_body<false, true, true, true>(p.n_src, p.P, p.S, p.C, rows, p.ysrc, p.ld_src, nullptr, 0, p.xyz_src, p.xyz_ctr, p.Wx, p.ldw, p.idx, 0,
                                               p.z, p.dxyz, p.partial, p.partial_x, blockIdx.x, tiles);
}

int cmf_group_affine_batch(int n, const CmfGroupAffineArgs *a, hipStream_t st)
{
    CMF_CHECK_ARG(n >= 1 && n <= CMF_MAX_BATCH && a);
    CmfBatch<CmfGroupAffineArgs> b;
    int tmax = 0;
    for (int i = 0; i < n; ++i) {
        const CmfGroupAffineArgs &q = a[i];
        CMF_CHECK_ARG(q.b > 0 && q.n_src > 0 && q.P > 0 && q.S > 0 && tile_ok(q.C) && q.C == a[0].C && q.C >= 4 && q.ld_src % 4 == 0);
        CMF_CHECK_ARG(q.ysrc && q.xyz_src && q.xyz_ctr && q.Wx && q.idx && q.z && q.dxyz && q.partial && q.partial_x);
        b.a[i] = q;
        tmax = std::max(tmax, cmf_divup((long long)q.b * q.P * q.S, PW_ROWS));
    }
    hipLaunchKernelGGL(group_affine_batch_kernel, dim3(tmax, n), dim3(PW_THREADS), tile_lds(a[0].C), st, b);
    return cmf_launch_status();
}

extern "C" int cmf_group_affine(int b, int n_src, int P, int S, int C,
                                const float *ysrc, int ld_src, const float *yctr, int ld_ctr,
                                const float *xyz_src, const float *xyz_ctr, const float *Wx, int ldw,
                                const int *idx, int act, float *z, float *dxyz, float *partial, float *partial_x, void *stream)
{
    CMF_CHECK_ARG(b >= 0 && n_src > 0 && P > 0 && S > 0 && tile_ok(C));
    if (b == 0) return 0;
    // z == NULL (with partial): the statistics only -- the tensor itself is formed again where it is consumed (cmf_gemm_gather_affine)
    CMF_CHECK_ARG(ysrc && xyz_src && xyz_ctr && Wx && idx && (z || (partial && !yctr)) && ld_src % 4 == 0 && (!yctr || ld_ctr % 4 == 0));
    CMF_CHECK_ARG(!partial_x || (partial && C >= 4));
    const long long rows = (long long)b * P * S;
    const int tiles = cmf_divup(rows, PW_ROWS);
#define CMF_GA_LAUNCH(CTR, ST, XS)                                                                                            \
    hipLaunchKernelGGL((group_affine_kernel<CTR, ST, XS>), dim3(tiles), dim3(PW_THREADS), partial ? tile_lds(C) : 0, (hipStream_t)stream, \
                       n_src, P, S, C, rows, ysrc, ld_src, yctr, ld_ctr, xyz_src, xyz_ctr, Wx, ldw, idx, act, z, dxyz, partial, partial_x)
    if (!z) {
        if (partial_x) hipLaunchKernelGGL((group_affine_kernel<false, true, true, false>), dim3(tiles), dim3(PW_THREADS), tile_lds(C), (hipStream_t)stream,
                                          n_src, P, S, C, rows, ysrc, ld_src, yctr, ld_ctr, xyz_src, xyz_ctr, Wx, ldw, idx, act, z, dxyz, partial, partial_x);
        else hipLaunchKernelGGL((group_affine_kernel<false, true, false, false>), dim3(tiles), dim3(PW_THREADS), tile_lds(C), (hipStream_t)stream,
                                n_src, P, S, C, rows, ysrc, ld_src, yctr, ld_ctr, xyz_src, xyz_ctr, Wx, ldw, idx, act, z, dxyz, partial, partial_x);
        return cmf_launch_status();
    }
    if (yctr) { if (partial_x) CMF_GA_LAUNCH(true, true, true); else if (partial) CMF_GA_LAUNCH(true, true, false); else CMF_GA_LAUNCH(true, false, false); }
    else { if (partial_x) CMF_GA_LAUNCH(false, true, true); else if (partial) CMF_GA_LAUNCH(false, true, false); else CMF_GA_LAUNCH(false, false, false); }
#undef CMF_GA_LAUNCH
    return cmf_launch_status();
}

// What the gathering GEMM (cmf_gemm_gather_affine, gemm.hip) needs of the neighbour lists: the source row and the relative
// coordinates of every slot, and the coordinate columns of the first conv as planes.
__global__ __launch_bounds__(PW_THREADS) void group_prep_kernel(int n_src, int P, int S, int C, long long rows, const float *__restrict__ xyz_src,
                                                                const float *__restrict__ xyz_ctr, const float *__restrict__ Wx, int ldw,
                                                                const int *__restrict__ idx, int *__restrict__ rows_out,
                                                                float *__restrict__ dxyz, float *__restrict__ wx3)
{
    const long long gid = (long long)blockIdx.x * PW_THREADS + threadIdx.x;
    if (gid < 3 * C) { const int k = (int)(gid / C), c = (int)(gid - (long long)k * C); wx3[gid] = Wx[(size_t)c * ldw + k]; }
    if (gid >= rows) return;
    const int j = idx[gid];
    const long long bp = gid / S;
    const int b = (int)(bp / P);
    const float *xs = xyz_src + ((size_t)b * n_src + j) * 3;
    const float *xc = xyz_ctr + (size_t)bp * 3;
    rows_out[gid] = b * n_src + j;
    *(float4 *)(dxyz + (size_t)gid * 4) = make_float4(xs[0] - xc[0], xs[1] - xc[1], xs[2] - xc[2], 0.f);
}

extern "C" int cmf_group_prep(int b, int n_src, int P, int S, int C, const float *xyz_src, const float *xyz_ctr, const float *Wx, int ldw,
                              const int *idx, int *rows, float *dxyz, float *wx3, void *stream)
{
    CMF_CHECK_ARG(b >= 0 && n_src > 0 && P > 0 && S > 0 && C > 0 && ldw >= 3);
    if (b == 0) return 0;
    CMF_CHECK_ARG(xyz_src && xyz_ctr && Wx && idx && rows && dxyz && wx3 && ((uintptr_t)dxyz & 15) == 0);
    const long long n = (long long)b * P * S;
    CMF_CHECK_ARG(n < (1ll << 31) && (long long)b * n_src < (1ll << 31));
    const long long work = n > 3ll * C ? n : 3ll * C;
    hipLaunchKernelGGL(group_prep_kernel, dim3((unsigned)cmf_divup(work, PW_THREADS)), dim3(PW_THREADS), 0, (hipStream_t)stream,
                       n_src, P, S, C, n, xyz_src, xyz_ctr, Wx, ldw, idx, rows, dxyz, wx3);
    return cmf_launch_status();
}

// The neighbour slots in inverse-index order (sorted by source point, ascending slot inside a point's list) for
// cmf_gemm_dx_gather_sum: position m = sample * entries + t holds slot perm[m] = sample * entries + inv[sample][t], its source point
// and its relative coordinates.
__global__ __launch_bounds__(PW_THREADS) void group_perm_kernel(long long total, int entries, const int *__restrict__ inv, const int *__restrict__ rows,
                                                                const float *__restrict__ dxyz, int *__restrict__ perm, int *__restrict__ pts,
                                                                float *__restrict__ dxyz2)
{
    const long long m = (long long)blockIdx.x * PW_THREADS + threadIdx.x;
    if (m >= total) return;
    const long long r = m / entries * entries + inv[m];
    perm[m] = (int)r;
    pts[m] = rows[r];
    *(float4 *)(dxyz2 + (size_t)m * 4) = *(const float4 *)(dxyz + (size_t)r * 4);
}

extern "C" int cmf_group_perm(int b, int entries, const int *inv, const int *rows, const float *dxyz, int *perm, int *pts, float *dxyz2, void *stream)
{
    CMF_CHECK_ARG(b >= 0 && entries > 0);
    if (b == 0) return 0;
    CMF_CHECK_ARG(inv && rows && dxyz && perm && pts && dxyz2 && (((uintptr_t)dxyz | (uintptr_t)dxyz2) & 15) == 0);
    const long long total = (long long)b * entries;
    CMF_CHECK_ARG(total < (1ll << 31));
    hipLaunchKernelGGL(group_perm_kernel, dim3((unsigned)cmf_divup(total, PW_THREADS)), dim3(PW_THREADS), 0, (hipStream_t)stream, total, entries, inv, rows,
                       dxyz, perm, pts, dxyz2);
    return cmf_launch_status();
}

// ---------------------------------------------------------------------------------------------
// out[p, :] = max_s relu(a*z[p,s,:] + c)   (radarflow_util.py:151-155 fused: BN + ReLU + max over the ball)
// argmax (uint8, first maximum) is kept for the backward pass.  out may be a column slice (ldo).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void bn_relu_maxpool_body(
    long long P, int S, int C, const float *__restrict__ z, const float *__restrict__ a, const float *__restrict__ c,
    float *__restrict__ out, long long ldo, unsigned char *__restrict__ argmax, float *__restrict__ zsel, const int bx, const int nbx)
{
    const int cg = C / 4;
    for (long long i = (long long)bx * PW_THREADS + threadIdx.x; i < P * cg; i += (long long)nbx * PW_THREADS) {
        const long long p = i / cg;
        const int col = (int)(i - p * cg) * 4;
        const float4 sa = *(const float4 *)(a + col), sc = *(const float4 *)(c + col);
        float4 best = make_float4(-1.f, -1.f, -1.f, -1.f), zb = make_float4(0.f, 0.f, 0.f, 0.f);
        uchar4 bi = make_uchar4(0, 0, 0, 0);
        const float *src = z + (size_t)p * S * C + col;
        auto take = [&](const float4 v, int s) {
            const float x = fmaxf(fmaf(sa.x, v.x, sc.x), 0.f), y = fmaxf(fmaf(sa.y, v.y, sc.y), 0.f);
            const float zz = fmaxf(fmaf(sa.z, v.z, sc.z), 0.f), w = fmaxf(fmaf(sa.w, v.w, sc.w), 0.f);
            if (x > best.x) { best.x = x; bi.x = (unsigned char)s; zb.x = v.x; }
            if (y > best.y) { best.y = y; bi.y = (unsigned char)s; zb.y = v.y; }
            if (zz > best.z) { best.z = zz; bi.z = (unsigned char)s; zb.z = v.z; }
            if (w > best.w) { best.w = w; bi.w = (unsigned char)s; zb.w = v.w; }
        };
        int s = 0;
        for (; s + 4 <= S; s += 4) {                         // four slots' rows in flight (a runtime-length loop of load -> compare is one round trip per slot)
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *(const float4 *)(src + (size_t)(s + u) * C);
#pragma unroll
            for (int u = 0; u < 4; ++u) take(v[u], s + u);
        }
        for (; s < S; ++s) take(*(const float4 *)(src + (size_t)s * C), s);
        *(float4 *)(out + (size_t)p * ldo + col) = best;
        if (argmax) *(uchar4 *)(argmax + (size_t)p * C + col) = bi;
        // the pre-activations at the arg-max slots, (P, C) contiguous: what the backward pass needs of z per point (it would otherwise
        // gather them as 4-byte reads from P * C different 32-byte sectors)
        if (zsel) *(float4 *)(zsel + (size_t)p * C + col) = zb;
    }
}

__global__ __launch_bounds__(PW_THREADS) void bn_relu_maxpool_kernel(
    long long P, int S, int C, const float *__restrict__ z, const float *__restrict__ a, const float *__restrict__ c,
    float *__restrict__ out, long long ldo, unsigned char *__restrict__ argmax, float *__restrict__ zsel)
{
    bn_relu_maxpool_body(P, S, C, z, a, c, out, ldo, argmax, zsel, blockIdx.x, gridDim.x);
}

__global__ __launch_bounds__(PW_THREADS) void bn_relu_maxpool_batch_kernel(const CmfBatch<CmfPoolArgs> b)
{
    const CmfPoolArgs &p = b.a[blockIdx.y];
    if ((int)blockIdx.x >= p.grid) return;
    bn_relu_maxpool_body(p.P, p.S, p.C, p.z, p.a, p.c, p.out, p.ldo, p.argmax, p.zsel, blockIdx.x, p.grid);
}

static int pool_grid(long long P, int C) { return (int)std::min<long long>((P * (C / 4) + PW_THREADS - 1) / PW_THREADS, 8192); }

int cmf_bn_relu_maxpool_batch(int n, CmfPoolArgs *a, hipStream_t st)
{
    CMF_CHECK_ARG(n >= 1 && n <= CMF_MAX_BATCH && a);
    CmfBatch<CmfPoolArgs> b;
    int gmax = 0;
    for (int i = 0; i < n; ++i) {
        CmfPoolArgs &q = a[i];
        CMF_CHECK_ARG(q.P > 0 && q.S > 0 && q.S <= 255 && q.C % 4 == 0 && q.ldo % 4 == 0 && q.z && q.a && q.c && q.out);
        q.grid = pool_grid(q.P, q.C);
        b.a[i] = q;
        gmax = std::max(gmax, q.grid);
    }
    hipLaunchKernelGGL(bn_relu_maxpool_batch_kernel, dim3(gmax, n), dim3(PW_THREADS), 0, st, b);
    return cmf_launch_status();
}

int cmf_bn_relu_maxpool_sel(long long P, int S, int C, const float *z, const float *a, const float *c,
                            float *out, long long ldo, unsigned char *argmax, float *zsel, void *stream);
extern "C" int cmf_bn_relu_maxpool(long long P, int S, int C, const float *z, const float *a, const float *c,
                                   float *out, long long ldo, unsigned char *argmax, void *stream)
{
    return cmf_bn_relu_maxpool_sel(P, S, C, z, a, c, out, ldo, argmax, nullptr, stream);
}

// ... also keeping the pre-activations at the arg-max slots: zsel (P, C) or NULL (internal: the set-conv block calls)
int cmf_bn_relu_maxpool_sel(long long P, int S, int C, const float *z, const float *a, const float *c,
                            float *out, long long ldo, unsigned char *argmax, float *zsel, void *stream)
{
    CMF_CHECK_ARG(P >= 0 && S > 0 && S <= 255 && C % 4 == 0 && ldo % 4 == 0);
    if (P == 0) return 0;
    CMF_CHECK_ARG(z && a && c && out);
    const int grid = pool_grid(P, C);
    hipLaunchKernelGGL(bn_relu_maxpool_kernel, dim3(grid), dim3(PW_THREADS), 0, (hipStream_t)stream, P, S, C, z, a, c, out, ldo, argmax, zsel);
    return cmf_launch_status();
}

// Backward of the fused BN+ReLU+max:  dU[p,s,:] = (s == argmax[p,:] && a*z+c > 0) ? dout[p,:] : 0, with the
// BN-backward partial sums  s1 = sum dU,  s2 = sum dU * (z - mean) * invstd.
__global__ __launch_bounds__(PW_THREADS) void maxpool_bwd_kernel(
    long long rows, int S, int C, const float *__restrict__ dout, long long ldd, const float *__restrict__ z,
    const float *__restrict__ a, const float *__restrict__ c, const float *__restrict__ mean,
    const float *__restrict__ invstd, const unsigned char *__restrict__ argmax, float *__restrict__ dU,
    float *__restrict__ partial)
{
    extern __shared__ __attribute__((aligned(16))) float red[];
    const TileMap tm = tile_map(C);
    const float4 sa = *(const float4 *)(a + tm.col), sc = *(const float4 *)(c + tm.col);
    const float4 mu = *(const float4 *)(mean + tm.col), is = *(const float4 *)(invstd + tm.col);
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
    const long long row0 = (long long)blockIdx.x * PW_ROWS;
    for (int r = tm.r0; r < PW_ROWS; r += tm.rl) {
        const long long row = row0 + r;
        if (row >= rows) break;
        const long long p = row / S;
        const int s = (int)(row - p * S);
        const uchar4 am = *(const uchar4 *)(argmax + (size_t)p * C + tm.col);
        const float4 g = *(const float4 *)(dout + (size_t)p * ldd + tm.col);
        const float4 v = *(const float4 *)(z + (size_t)row * C + tm.col);
        float4 d;
        d.x = (am.x == s && fmaf(sa.x, v.x, sc.x) > 0.f) ? g.x : 0.f;
        d.y = (am.y == s && fmaf(sa.y, v.y, sc.y) > 0.f) ? g.y : 0.f;
        d.z = (am.z == s && fmaf(sa.z, v.z, sc.z) > 0.f) ? g.z : 0.f;
        d.w = (am.w == s && fmaf(sa.w, v.w, sc.w) > 0.f) ? g.w : 0.f;
        *(float4 *)(dU + (size_t)row * C + tm.col) = d;
        s1.x += d.x; s1.y += d.y; s1.z += d.z; s1.w += d.w;
        s2.x += d.x * ((v.x - mu.x) * is.x); s2.y += d.y * ((v.y - mu.y) * is.y);
        s2.z += d.z * ((v.z - mu.z) * is.z); s2.w += d.w * ((v.w - mu.w) * is.w);
    }
    tile_reduce_store(s1, s2, tm, C, partial, red);
}

extern "C" int cmf_maxpool_bwd(long long P, int S, int C, const float *dout, long long ldd, const float *z,
                               const float *a, const float *c, const float *mean, const float *invstd,
                               const unsigned char *argmax, float *dU, float *partial, void *stream)
{
    CMF_CHECK_ARG(P >= 0 && S > 0 && tile_ok(C) && ldd % 4 == 0);
    if (P == 0) return 0;
    CMF_CHECK_ARG(dout && z && a && c && mean && invstd && argmax && dU && partial);
    const long long rows = P * S;
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(cmf_divup(rows, PW_ROWS)), dim3(PW_THREADS), tile_lds(C), (hipStream_t)stream,
                       rows, S, C, dout, ldd, z, a, c, mean, invstd, argmax, dU, partial);
    return cmf_launch_status();
}

// The same backward per POINT instead of per neighbour row: g[p,:] = dout[p,:] * [a * z[p, argmax[p,:], :] + c > 0] (the
// only slot of a point that receives a gradient is its argmax) with the BN-backward partial sums per 128 points --
// identical sums to maxpool_bwd's (the other rows contribute exact zeros), P rows touched instead of P * S.  The consumer
// (cmf_thin_bwd_layer_pooled) forms dU[p,s,:] = (s == argmax[p,:]) ? g[p,:] : 0 on the fly.
__device__ __forceinline__ void maxpool_bwd_point_body(
    long long P, int S, int C, const float *__restrict__ dout, long long ldd, const float *__restrict__ z,
    const float *__restrict__ a, const float *__restrict__ c, const float *__restrict__ mean,
    const float *__restrict__ invstd, const unsigned char *__restrict__ argmax, float *__restrict__ g,
    float *__restrict__ partial, int sel, const int bx)
{
    extern __shared__ __attribute__((aligned(16))) float red[];
    const TileMap tm = tile_map(C);
    const float4 sa = *(const float4 *)(a + tm.col), sc = *(const float4 *)(c + tm.col);
    const float4 mu = *(const float4 *)(mean + tm.col), is = *(const float4 *)(invstd + tm.col);
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
    const long long p0 = (long long)bx * PW_ROWS;
    for (int r = tm.r0; r < PW_ROWS; r += tm.rl) {
        const long long p = p0 + r;
        if (p >= P) break;
        const float4 gi = *(const float4 *)(dout + (size_t)p * ldd + tm.col);
        float vx, vy, vz, vw;
        if (sel) {                                       // z = the selected pre-activations per point (P, C): the set-conv chain keeps only those
            const float4 zs = *(const float4 *)(z + (size_t)p * C + tm.col);
            vx = zs.x; vy = zs.y; vz = zs.z; vw = zs.w;
        } else {
            const uchar4 am = *(const uchar4 *)(argmax + (size_t)p * C + tm.col);
            const float *zp = z + (size_t)p * S * C + tm.col;
            vx = zp[(size_t)am.x * C]; vy = zp[(size_t)am.y * C + 1]; vz = zp[(size_t)am.z * C + 2]; vw = zp[(size_t)am.w * C + 3];
        }
        float4 d;
        d.x = fmaf(sa.x, vx, sc.x) > 0.f ? gi.x : 0.f;
        d.y = fmaf(sa.y, vy, sc.y) > 0.f ? gi.y : 0.f;
        d.z = fmaf(sa.z, vz, sc.z) > 0.f ? gi.z : 0.f;
        d.w = fmaf(sa.w, vw, sc.w) > 0.f ? gi.w : 0.f;
        *(float4 *)(g + (size_t)p * C + tm.col) = d;
        s1.x += d.x; s1.y += d.y; s1.z += d.z; s1.w += d.w;
        s2.x += d.x * ((vx - mu.x) * is.x); s2.y += d.y * ((vy - mu.y) * is.y);
        s2.z += d.z * ((vz - mu.z) * is.z); s2.w += d.w * ((vw - mu.w) * is.w);
    }
    tile_reduce_store(s1, s2, tm, C, partial, red, bx);
}

__global__ __launch_bounds__(PW_THREADS) void maxpool_bwd_point_kernel(
    long long P, int S, int C, const float *__restrict__ dout, long long ldd, const float *__restrict__ z,
    const float *__restrict__ a, const float *__restrict__ c, const float *__restrict__ mean,
    const float *__restrict__ invstd, const unsigned char *__restrict__ argmax, float *__restrict__ g,
    float *__restrict__ partial, int sel)
{
    maxpool_bwd_point_body(P, S, C, dout, ldd, z, a, c, mean, invstd, argmax, g, partial, sel, blockIdx.x);
}

__global__ __launch_bounds__(PW_THREADS) void maxpool_bwd_point_batch_kernel(const CmfBatch<CmfPoolBwdArgs> b)
{
    const CmfPoolBwdArgs &p = b.a[blockIdx.y];
    if ((long long)blockIdx.x * PW_ROWS >= p.P) return;
    maxpool_bwd_point_body(p.P, p.S, p.C, p.dout, p.ldd, p.z, p.a, p.c, p.mean, p.invstd, p.argmax, p.g, p.partial, p.sel, blockIdx.x);
}

int cmf_maxpool_bwd_point_batch(int n, const CmfPoolBwdArgs *a, hipStream_t st)
{
    CMF_CHECK_ARG(n >= 1 && n <= CMF_MAX_BATCH && a);
    CmfBatch<CmfPoolBwdArgs> b;
    long long pmax = 0;
    for (int i = 0; i < n; ++i) {
        const CmfPoolBwdArgs &q = a[i];
        CMF_CHECK_ARG(q.P > 0 && q.S > 0 && q.S <= 255 && tile_ok(q.C) && q.C == a[0].C && q.ldd % 4 == 0);
        CMF_CHECK_ARG(q.dout && q.z && q.a && q.c && q.mean && q.invstd && (q.sel || q.argmax) && q.g && q.partial);
        b.a[i] = q;
        pmax = std::max(pmax, q.P);
    }
    hipLaunchKernelGGL(maxpool_bwd_point_batch_kernel, dim3(cmf_divup(pmax, PW_ROWS), n), dim3(PW_THREADS), tile_lds(a[0].C), st, b);
    return cmf_launch_status();
}

extern "C" int cmf_maxpool_bwd_point(long long P, int S, int C, const float *dout, long long ldd, const float *z,
                                     const float *a, const float *c, const float *mean, const float *invstd,
                                     const unsigned char *argmax, float *g, float *partial, void *stream)
{
    CMF_CHECK_ARG(P >= 0 && S > 0 && S <= 255 && tile_ok(C) && ldd % 4 == 0);
    if (P == 0) return 0;
    CMF_CHECK_ARG(dout && z && a && c && mean && invstd && argmax && g && partial);
    hipLaunchKernelGGL(maxpool_bwd_point_kernel, dim3(cmf_divup(P, PW_ROWS)), dim3(PW_THREADS), tile_lds(C), (hipStream_t)stream,
                       P, S, C, dout, ldd, z, a, c, mean, invstd, argmax, g, partial, 0);
    return cmf_launch_status();
}

// internal (setconv_block.hip): the same with the pre-activation of every point's argmax slot given directly (zsel: P x C)
int cmf_maxpool_bwd_point_sel(long long P, int C, const float *dout, long long ldd, const float *zsel, const float *a, const float *c,
                              const float *mean, const float *invstd, float *g, float *partial, void *stream)
{
    CMF_CHECK_ARG(P >= 0 && tile_ok(C) && ldd % 4 == 0);
    if (P == 0) return 0;
    CMF_CHECK_ARG(dout && zsel && a && c && mean && invstd && g && partial);
    hipLaunchKernelGGL(maxpool_bwd_point_kernel, dim3(cmf_divup(P, PW_ROWS)), dim3(PW_THREADS), tile_lds(C), (hipStream_t)stream,
                       P, 1, C, dout, ldd, zsel, a, c, mean, invstd, nullptr, g, partial, 1);
    return cmf_launch_status();
}

// ---------------------------------------------------------------------------------------------
// y = relu(a*z + c) materialised (chain ends; out may be a column slice of a concat buffer)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void affine_relu_body(long long M, int C, const float *__restrict__ z, long long ldz,
                                                 const float *__restrict__ a, const float *__restrict__ c,
                                                 float *__restrict__ out, long long ldo)
{
    const int cg = C / 4;
    for (long long i = (long long)blockIdx.x * PW_THREADS + threadIdx.x; i < M * cg; i += (long long)gridDim.x * PW_THREADS) {
        const long long m = i / cg;
        const int col = (int)(i - m * cg) * 4;
        const float4 sa = *(const float4 *)(a + col), sc = *(const float4 *)(c + col);
        float4 v = *(const float4 *)(z + (size_t)m * ldz + col);
        v.x = fmaxf(fmaf(sa.x, v.x, sc.x), 0.f); v.y = fmaxf(fmaf(sa.y, v.y, sc.y), 0.f);
        v.z = fmaxf(fmaf(sa.z, v.z, sc.z), 0.f); v.w = fmaxf(fmaf(sa.w, v.w, sc.w), 0.f);
        *(float4 *)(out + (size_t)m * ldo + col) = v;
    }
}

__global__ __launch_bounds__(PW_THREADS) void affine_relu_kernel(long long M, int C, const float *__restrict__ z, long long ldz,
                                                                 const float *__restrict__ a, const float *__restrict__ c,
                                                                 float *__restrict__ out, long long ldo)
{
    affine_relu_body(M, C, z, ldz, a, c, out, ldo);
}

__global__ __launch_bounds__(PW_THREADS) void affine_relu_batch_kernel(const CmfBatch<CmfAffineArgs> b)
{
    const CmfAffineArgs &p = b.a[blockIdx.y];
    affine_relu_body(p.M, p.C, p.z, p.ldz, p.a, p.c, p.out, p.ldo);          // grid-stride over the problem's elements
}

int cmf_affine_relu_batch(int n, const CmfAffineArgs *a, hipStream_t st)
{
    CMF_CHECK_ARG(n >= 1 && n <= CMF_MAX_BATCH && a);
    CmfBatch<CmfAffineArgs> b;
    long long work = 0;
    for (int i = 0; i < n; ++i) {
        CMF_CHECK_ARG(a[i].M > 0 && a[i].C % 4 == 0 && a[i].ldz % 4 == 0 && a[i].ldo % 4 == 0 && a[i].z && a[i].a && a[i].c && a[i].out);
        b.a[i] = a[i];
        work = std::max(work, a[i].M * (a[i].C / 4));
    }
    const int grid = (int)std::min<long long>((work + PW_THREADS - 1) / PW_THREADS, 8192);
    hipLaunchKernelGGL(affine_relu_batch_kernel, dim3(grid, n), dim3(PW_THREADS), 0, st, b);
    return cmf_launch_status();
}

extern "C" int cmf_affine_relu(long long M, int C, const float *z, long long ldz, const float *a, const float *c,
                               float *out, long long ldo, void *stream)
{
    CMF_CHECK_ARG(M >= 0 && C % 4 == 0 && ldz % 4 == 0 && ldo % 4 == 0);
    if (M == 0) return 0;
    CMF_CHECK_ARG(z && a && c && out);
    const int grid = (int)std::min<long long>((M * (C / 4) + PW_THREADS - 1) / PW_THREADS, 8192);
    hipLaunchKernelGGL(affine_relu_kernel, dim3(grid), dim3(PW_THREADS), 0, (hipStream_t)stream, M, C, z, ldz, a, c, out, ldo);
    return cmf_launch_status();
}

// dU = dY * [a*z + c > 0] with the BN-backward partial sums (stand-alone form of cmf_gemm's bwd_mode 1,
// for gradients that do not come out of a GEMM: concat slices, global max, ...).
__device__ __forceinline__ void act_bwd_stats_body(
    long long rows, int C, const float *__restrict__ dY, long long ldy, const float *__restrict__ z, long long ldz,
    const float *__restrict__ a, const float *__restrict__ c, const float *__restrict__ mean,
    const float *__restrict__ invstd, float *__restrict__ dU, float *__restrict__ partial)
{
    extern __shared__ __attribute__((aligned(16))) float red[];
    const TileMap tm = tile_map(C);
    const float4 sa = *(const float4 *)(a + tm.col), sc = *(const float4 *)(c + tm.col);
    const float4 mu = *(const float4 *)(mean + tm.col), is = *(const float4 *)(invstd + tm.col);
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
    const long long row0 = (long long)blockIdx.x * PW_ROWS;
    for (int r = tm.r0; r < PW_ROWS; r += tm.rl) {
        const long long row = row0 + r;
        if (row >= rows) break;
        const float4 g = *(const float4 *)(dY + (size_t)row * ldy + tm.col);
        const float4 v = *(const float4 *)(z + (size_t)row * ldz + tm.col);
        float4 d;
        d.x = fmaf(sa.x, v.x, sc.x) > 0.f ? g.x : 0.f; d.y = fmaf(sa.y, v.y, sc.y) > 0.f ? g.y : 0.f;
        d.z = fmaf(sa.z, v.z, sc.z) > 0.f ? g.z : 0.f; d.w = fmaf(sa.w, v.w, sc.w) > 0.f ? g.w : 0.f;
        *(float4 *)(dU + (size_t)row * C + tm.col) = d;
        s1.x += d.x; s1.y += d.y; s1.z += d.z; s1.w += d.w;
        s2.x += d.x * ((v.x - mu.x) * is.x); s2.y += d.y * ((v.y - mu.y) * is.y);
        s2.z += d.z * ((v.z - mu.z) * is.z); s2.w += d.w * ((v.w - mu.w) * is.w);
    }
    tile_reduce_store(s1, s2, tm, C, partial, red);
}

__global__ __launch_bounds__(PW_THREADS) void act_bwd_stats_kernel(
    long long rows, int C, const float *__restrict__ dY, long long ldy, const float *__restrict__ z, long long ldz,
    const float *__restrict__ a, const float *__restrict__ c, const float *__restrict__ mean,
    const float *__restrict__ invstd, float *__restrict__ dU, float *__restrict__ partial)
{
    act_bwd_stats_body(rows, C, dY, ldy, z, ldz, a, c, mean, invstd, dU, partial);
}

__global__ __launch_bounds__(PW_THREADS) void act_bwd_stats_batch_kernel(const CmfBatch<CmfActBwdArgs> b)
{
    const CmfActBwdArgs &p = b.a[blockIdx.y];
    if ((long long)blockIdx.x * PW_ROWS >= p.rows) return;
    act_bwd_stats_body(p.rows, p.C, p.dY, p.ldy, p.z, p.ldz, p.a, p.c, p.mean, p.invstd, p.dU, p.partial);
}

int cmf_act_bwd_stats_batch(int n, const CmfActBwdArgs *a, hipStream_t st)
{
    CMF_CHECK_ARG(n >= 1 && n <= CMF_MAX_BATCH && a);
    CmfBatch<CmfActBwdArgs> b;
    long long rmax = 0;
    for (int i = 0; i < n; ++i) {
        CMF_CHECK_ARG(a[i].rows > 0 && a[i].C == a[0].C && tile_ok(a[i].C) && a[i].ldy % 4 == 0 && a[i].ldz % 4 == 0 && a[i].dY && a[i].z && a[i].a &&
                      a[i].c && a[i].mean && a[i].invstd && a[i].dU && a[i].partial);
        b.a[i] = a[i];
        rmax = std::max(rmax, a[i].rows);
    }
    hipLaunchKernelGGL(act_bwd_stats_batch_kernel, dim3(cmf_divup(rmax, PW_ROWS), n), dim3(PW_THREADS), tile_lds(a[0].C), st, b);
    return cmf_launch_status();
}

extern "C" int cmf_act_bwd_stats(long long M, int C, const float *dY, long long ldy, const float *z, long long ldz,
                                 const float *a, const float *c, const float *mean, const float *invstd,
                                 float *dU, float *partial, void *stream)
{
    CMF_CHECK_ARG(M >= 0 && tile_ok(C) && ldy % 4 == 0 && ldz % 4 == 0);
    if (M == 0) return 0;
    CMF_CHECK_ARG(dY && z && a && c && mean && invstd && dU && partial);
    hipLaunchKernelGGL(act_bwd_stats_kernel, dim3(cmf_divup(M, PW_ROWS)), dim3(PW_THREADS), tile_lds(C), (hipStream_t)stream,
                       M, C, dY, ldy, z, ldz, a, c, mean, invstd, dU, partial);
    return cmf_launch_status();
}

// Train-mode BatchNorm backward, in place:  dZ = a * (dU - s1/M - zhat * s2/M),  zhat = (z-mean)*invstd,
// (s1, s2) = sums[2][C].  Eval mode (sums == nullptr): dZ = a * dU.
__global__ __launch_bounds__(PW_THREADS) void bn_bwd_apply_kernel(
    long long M, int C, float *__restrict__ dU, const float *__restrict__ z, long long ldz, const float *__restrict__ a,
    const float *__restrict__ mean, const float *__restrict__ invstd, const float *__restrict__ sums, float inv_count)
{
    const int cg = C / 4;
    for (long long i = (long long)blockIdx.x * PW_THREADS + threadIdx.x; i < M * cg; i += (long long)gridDim.x * PW_THREADS) {
        const long long m = i / cg;
        const int col = (int)(i - m * cg) * 4;
        const float4 sa = *(const float4 *)(a + col);
        float4 d = *(float4 *)(dU + (size_t)m * C + col);
        if (sums) {
            const float4 mu = *(const float4 *)(mean + col), is = *(const float4 *)(invstd + col);
            const float4 t1 = *(const float4 *)(sums + col), t2 = *(const float4 *)(sums + C + col);
            const float4 v = *(const float4 *)(z + (size_t)m * ldz + col);
            float al, be, ga;                                    // cmf_common.h: shared with the GEMM that fuses this pass
            cmf_bnb_coef(sa.x, mu.x, is.x, t1.x, t2.x, inv_count, al, be, ga); d.x = cmf_bnb_apply(d.x, v.x, mu.x, al, be, ga);
            cmf_bnb_coef(sa.y, mu.y, is.y, t1.y, t2.y, inv_count, al, be, ga); d.y = cmf_bnb_apply(d.y, v.y, mu.y, al, be, ga);
            cmf_bnb_coef(sa.z, mu.z, is.z, t1.z, t2.z, inv_count, al, be, ga); d.z = cmf_bnb_apply(d.z, v.z, mu.z, al, be, ga);
            cmf_bnb_coef(sa.w, mu.w, is.w, t1.w, t2.w, inv_count, al, be, ga); d.w = cmf_bnb_apply(d.w, v.w, mu.w, al, be, ga);
        } else { d.x *= sa.x; d.y *= sa.y; d.z *= sa.z; d.w *= sa.w; }
        *(float4 *)(dU + (size_t)m * C + col) = d;
    }
}

extern "C" int cmf_bn_bwd_apply(long long M, int C, float *dU, const float *z, long long ldz, const float *a,
                                const float *mean, const float *invstd, const float *sums, void *stream)
{
    CMF_CHECK_ARG(M >= 0 && C % 4 == 0 && ldz % 4 == 0);
    if (M == 0) return 0;
    CMF_CHECK_ARG(dU && a && (!sums || (z && mean && invstd)));
    const int grid = (int)std::min<long long>((M * (C / 4) + PW_THREADS - 1) / PW_THREADS, 8192);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid), dim3(PW_THREADS), 0, (hipStream_t)stream,
                       M, C, dU, z, ldz, a, mean, invstd, sums, (float)(1.0 / (double)M));
    return cmf_launch_status();
}

// ---- global feature: max over the points, broadcast back and concatenated ------------------------------------
// cmflow.py:76-81,89-91 (Backbone): gfeat = max_n f[b,n,:]; out = cat(f, gfeat expanded over n).  As torch ops that is
// a strided reduction, an expand and a cat in forward and a scatter + two adds + a reduction in backward; here one
// kernel per direction.  A workgroup owns 64 channels of one sample: 16 threads (float4 each) span the channels, 16 row
// groups walk the rows; partial results are folded across the row groups in LDS in fixed order.  arg keeps the FIRST row
// attaining the maximum -- the row torch.max's gradient goes to.
constexpr int GM_THREADS = 256;
constexpr int GM_CH = 64;

__global__ __launch_bounds__(GM_THREADS) void global_max_cat_kernel(
    int N, int C, const float *__restrict__ f, long long ldf, float *__restrict__ out, long long ldo, int *__restrict__ arg)
{
    __shared__ float smax[GM_THREADS / 16][GM_CH];
    __shared__ int sarg[GM_THREADS / 16][GM_CH];
    const int b = blockIdx.y, c0 = blockIdx.x * GM_CH;
    const int tc = (threadIdx.x % 16) * 4, rg = threadIdx.x / 16;
    const int c = c0 + tc;
    const bool live = c < C;
    const float *fb = f + (long long)b * N * ldf;
    float *ob = out + (long long)b * N * ldo;
    float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int am[4] = {0, 0, 0, 0};
    if (live)
        for (int n = rg; n < N; n += GM_THREADS / 16) {
            const float4 v = *(const float4 *)(fb + (long long)n * ldf + c);
            *(float4 *)(ob + (long long)n * ldo + c) = v;
            const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (vv[i] > m[i]) { m[i] = vv[i]; am[i] = n; }
        }
#pragma unroll
    for (int i = 0; i < 4; ++i) { smax[rg][tc + i] = m[i]; sarg[rg][tc + i] = am[i]; }
    __syncthreads();
    if (threadIdx.x < GM_CH) {
        float best = smax[0][threadIdx.x];
        int ba = sarg[0][threadIdx.x];
        for (int g = 1; g < GM_THREADS / 16; ++g) {
            const float v = smax[g][threadIdx.x];
            const int a = sarg[g][threadIdx.x];
            if (v > best || (v == best && a < ba)) { best = v; ba = a; }
        }
        smax[0][threadIdx.x] = best;
        if (c0 + (int)threadIdx.x < C) arg[(long long)b * C + c0 + threadIdx.x] = ba;
    }
    __syncthreads();
    if (live) {
        const float4 g = make_float4(smax[0][tc], smax[0][tc + 1], smax[0][tc + 2], smax[0][tc + 3]);
        for (int n = rg; n < N; n += GM_THREADS / 16) *(float4 *)(ob + (long long)n * ldo + C + c) = g;
    }
}

__global__ __launch_bounds__(GM_THREADS) void global_max_cat_grad_kernel(
    int N, int C, const float *__restrict__ dout, long long ldd, const int *__restrict__ arg, float *__restrict__ df, long long ldf)
{
    __shared__ float ssum[GM_THREADS / 16][GM_CH];
    const int b = blockIdx.y, c0 = blockIdx.x * GM_CH;
    const int tc = (threadIdx.x % 16) * 4, rg = threadIdx.x / 16;
    const int c = c0 + tc;
    const bool live = c < C;
    const float *db = dout + (long long)b * N * ldd;
    float *fb = df + (long long)b * N * ldf;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    if (live)
        for (int n = rg; n < N; n += GM_THREADS / 16) {
            const float4 v = *(const float4 *)(db + (long long)n * ldd + C + c);
            s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
        }
#pragma unroll
    for (int i = 0; i < 4; ++i) ssum[rg][tc + i] = s[i];
    __syncthreads();
    if (threadIdx.x < GM_CH) {
        float t = ssum[0][threadIdx.x];
        for (int g = 1; g < GM_THREADS / 16; ++g) t += ssum[g][threadIdx.x];
        ssum[0][threadIdx.x] = t;
    }
    __syncthreads();
    if (live) {
        const int4 a = *(const int4 *)(arg + (long long)b * C + c);
        const float4 g = make_float4(ssum[0][tc], ssum[0][tc + 1], ssum[0][tc + 2], ssum[0][tc + 3]);
        for (int n = rg; n < N; n += GM_THREADS / 16) {
            float4 v = *(const float4 *)(db + (long long)n * ldd + c);
            if (n == a.x) v.x += g.x;
            if (n == a.y) v.y += g.y;
            if (n == a.z) v.z += g.z;
            if (n == a.w) v.w += g.w;
            *(float4 *)(fb + (long long)n * ldf + c) = v;
        }
    }
}

extern "C" int cmf_global_max_cat(int B, int N, int C, const float *f, long long ldf, float *out, long long ldo, int *arg, void *stream)
{
    CMF_CHECK_ARG(B >= 0 && N > 0 && C > 0 && C % 4 == 0 && ldf >= C && ldo >= 2 * C && ldf % 4 == 0 && ldo % 4 == 0 && B < 65536);
    if (B == 0) return 0;
    CMF_CHECK_ARG(f && out && arg && (((uintptr_t)f | (uintptr_t)out | (uintptr_t)arg) & 15) == 0);
    hipLaunchKernelGGL(global_max_cat_kernel, dim3(cmf_divup(C, GM_CH), B), dim3(GM_THREADS), 0, (hipStream_t)stream, N, C, f, ldf, out, ldo, arg);
    return cmf_launch_status();
}

extern "C" int cmf_global_max_cat_grad(int B, int N, int C, const float *dout, long long ldd, const int *arg, float *df, long long ldf,
                                       void *stream)
{
    CMF_CHECK_ARG(B >= 0 && N > 0 && C > 0 && C % 4 == 0 && ldf >= C && ldd >= 2 * C && ldf % 4 == 0 && ldd % 4 == 0 && B < 65536);
    if (B == 0) return 0;
    CMF_CHECK_ARG(dout && df && arg && (((uintptr_t)dout | (uintptr_t)df | (uintptr_t)arg) & 15) == 0);
    hipLaunchKernelGGL(global_max_cat_grad_kernel, dim3(cmf_divup(C, GM_CH), B), dim3(GM_THREADS), 0, (hipStream_t)stream, N, C, dout, ldd, arg,
                       df, ldf);
    return cmf_launch_status();
}

// ---- stacked first-conv weights -----------------------------------------------------------------------------
// The feature half of the first convs of an encoder's scales runs as ONE GEMM over the shared input (radarflow_util.py
// :132-139 by linearity): its weight is the scales' [O1][3 + cin] conv weights without the three xyz columns, stacked by
// rows, with the first n_tail input channels moved behind the rest and zero-padded to Kp columns.  Built every step (the
// optimizer changes the weights) -- as torch ops that was 4 slices + 4 cats + a cat + a pad per call, and the way back
// 2 x (add_, copy_) per scale: ~40 tiny launches per step, all of them alone on the GPU at the start of the forward /
// the end of the backward pass.  Here one launch per direction.
struct StackArgs { int n_w, O1, cin, n_tail, Kp; const float *w[8]; float *g[8]; };

__global__ __launch_bounds__(256) void stack_first_conv_kernel(const StackArgs a, float *__restrict__ wf)
{
    const long long total = (long long)a.n_w * a.O1 * a.Kp;
    const int head = a.cin - a.n_tail;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % a.Kp);
        const long long row = i / a.Kp;
        const int r = (int)(row % a.O1), s = (int)(row / a.O1);
        float v = 0.f;
        if (c < a.cin) {
            const int src = c < head ? 3 + a.n_tail + c : 3 + (c - head);
            v = a.w[s][(long long)r * (a.cin + 3) + src];
        }
        wf[i] = v;
    }
}

__global__ __launch_bounds__(256) void unstack_first_conv_grad_kernel(const StackArgs a, const float *__restrict__ dwf)
{
    const long long total = (long long)a.n_w * a.O1 * a.cin;
    const int head = a.cin - a.n_tail;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % a.cin);
        const long long row = i / a.cin;
        const int r = (int)(row % a.O1), s = (int)(row / a.O1);
        const int dst = c < head ? 3 + a.n_tail + c : 3 + (c - head);
        a.g[s][(long long)r * (a.cin + 3) + dst] += dwf[row * a.Kp + c];
    }
}

extern "C" int cmf_stack_first_conv(int n_w, int O1, int cin, int n_tail, int Kp, const float *const *w, float *wf, void *stream)
{
    CMF_CHECK_ARG(n_w >= 1 && n_w <= 8 && O1 > 0 && cin > 0 && n_tail >= 0 && n_tail <= cin && Kp >= cin && w && wf);
    StackArgs a;
    a.n_w = n_w; a.O1 = O1; a.cin = cin; a.n_tail = n_tail; a.Kp = Kp;
    for (int i = 0; i < 8; ++i) { a.w[i] = i < n_w ? w[i] : nullptr; a.g[i] = nullptr; }
    for (int i = 0; i < n_w; ++i) CMF_CHECK_ARG(w[i] != nullptr);
    const long long total = (long long)n_w * O1 * Kp;
    hipLaunchKernelGGL(stack_first_conv_kernel, dim3((unsigned)std::min<long long>((total + 255) / 256, 4096)), dim3(256), 0,
                       (hipStream_t)stream, a, wf);
    return cmf_launch_status();
}

extern "C" int cmf_unstack_first_conv_grad(int n_w, int O1, int cin, int n_tail, int Kp, const float *dwf, float *const *g, void *stream)
{
    CMF_CHECK_ARG(n_w >= 1 && n_w <= 8 && O1 > 0 && cin > 0 && n_tail >= 0 && n_tail <= cin && Kp >= cin && dwf && g);
    StackArgs a;
    a.n_w = n_w; a.O1 = O1; a.cin = cin; a.n_tail = n_tail; a.Kp = Kp;
    for (int i = 0; i < 8; ++i) { a.w[i] = nullptr; a.g[i] = i < n_w ? g[i] : nullptr; }
    for (int i = 0; i < n_w; ++i) CMF_CHECK_ARG(g[i] != nullptr);
    const long long total = (long long)n_w * O1 * cin;
    hipLaunchKernelGGL(unstack_first_conv_grad_kernel, dim3((unsigned)std::min<long long>((total + 255) / 256, 4096)), dim3(256), 0,
                       (hipStream_t)stream, a, dwf);
    return cmf_launch_status();
}
