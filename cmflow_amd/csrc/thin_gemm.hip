// "Thin" fp32 MFMA GEMMs for the narrow layers of the set-conv blocks (32 / 64 channels: the whole first
// encoder, the tail of every second-encoder block).  With K <= 64 the tiled kernel of gemm.hip is pure
// latency: a 128-row tile has two K-chunks of work but still pays the global->LDS staging, ~10 workgroup
// barriers and an LDS-transposed epilogue (7-8 us per workgroup, 14-46 us per launch, ~190 launches per
// training step, all on the critical path of the encoder chains).  These layers are HBM-bound streams of
// [rows, <=64] matrices, so here each WAVEFRONT works alone: operands go straight from global memory into
// MFMA fragment registers (no LDS staging, no barriers in the main part), one 32-row tile per wave.
//
//   thin_fwd : C[M,N]  = epi( pro(A)[M,K] @ W[N,K]^T )        A rows k-contiguous, W = conv weight (out,in)
//   thin_dx  : C[M,N]  = epi( A[M,K] @ B[K,N] )               B = W[K=out][N=in]  (data gradient)
//   thin_dw  : C[N,K] += sum over rows of A[m,N]^T B'[m,K]    (weight gradient; per-wave slabs, fixed-order reduce)
//
// Fragment maps of v_mfma_f32_32x32x2_f32 (cdna_hip_programming.md 3): A: lane l holds A[i=l&31][k=l>>5],
// B: lane l holds B[k=l>>5][j=l&31]; C/D: col = l&31, row = (r&3) + 8*(r>>2) + 4*(l>>5).  As in gemm.hip a lane
// loads 4 consecutive k of its row at once (lanes 0-31: k0..k0+3, lanes 32-63: k0+4..k0+7) and feeds 4 MFMAs.
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include "cmf_common.h"
#include "../../include/cmflow_hip.h"
#include "gemm_args.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TG_THREADS = 256;          // 4 waves = 128 rows per workgroup: same partial-statistics tiling as gemm.hip

__device__ __forceinline__ float thin_act(float v, int act)
{
    if (act == 1) return v > 0.f ? v : 0.f;
    if (act == 2) return v > 0.f ? v : 0.1f * v;
    if (act == 3) return 1.0f / (1.0f + __expf(-v));
    return v;
}

// epilogue shared by thin_fwd / thin_dx: bias, activation, backward masks, column statistics, store
template <int NT>
__device__ __forceinline__ void thin_epilogue(const GemmArgs &p, f32x16 (&acc)[NT], int m0, int lane, int wave, float *red, int bx)
{
    const int h = lane >> 5, cl = lane & 31;
    const bool want_stats = p.stats != nullptr;
    const bool want_q = want_stats && p.bwd_mode != 0 && p.dxyz != nullptr;
    const int nstat = want_q ? 5 : 2;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = nt * 32 + cl;
        const bool n_ok = n < p.N;
        float bias = 0.f, ea = 0.f, ec = 0.f, em = 0.f, ei = 0.f;
        if (n_ok) {
            if (p.bias) bias = p.bias[n];
            if (p.bwd_mode == 1) { ea = p.ea[n]; ec = p.ec[n]; em = p.emean[n]; ei = p.einvstd[n]; }
        }
        float s1 = 0.f, s2 = 0.f, q0 = 0.f, q1 = 0.f, q2 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (m < p.M && n_ok) {
                float x = thin_act(acc[nt][r] + bias, p.act);
                if (p.bwd_mode) {
                    const float z = p.Z[(long long)m * p.ldz + n];
                    if (p.bwd_mode == 1) {
                        x = (fmaf(ea, z, ec) > 0.f) ? x : 0.f;
                        s1 += x; s2 += x * ((z - em) * ei);
                        if (want_q) { const float4 d = *(const float4 *)(p.dxyz + (long long)m * 4); q0 += x * d.x; q1 += x * d.y; q2 += x * d.z; }
                    } else {
                        x = z > 0.f ? x : (p.bwd_mode == 2 ? 0.1f * x : 0.f);
                        if (want_stats) s1 += x;
                        if (want_q) { const float4 d = *(const float4 *)(p.dxyz + (long long)m * 4); q0 += x * d.x; q1 += x * d.y; q2 += x * d.z; }
                    }
                } else if (want_stats) { s1 += x; s2 = fmaf(x, x, s2); }         // (explicit: the full-tile body must sum the same way)
                float *dst = p.C + (long long)m * p.ldc + n;
                *dst = p.accumulate ? *dst + x : x;
            }
        }
        if (want_stats) {
            s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
            if (want_q) { q0 += __shfl_xor(q0, 32, 64); q1 += __shfl_xor(q1, 32, 64); q2 += __shfl_xor(q2, 32, 64); }
            if (lane < 32) {
                float *r0 = red + (size_t)wave * 5 * 64 + n;            // [4 waves][5][64]
                r0[0] = s1; r0[64] = s2;
                if (want_q) { r0[128] = q0; r0[192] = q1; r0[256] = q2; }
            }
        }
    }
    if (want_stats) {
        __syncthreads();
        for (int i = threadIdx.x; i < nstat * p.N; i += TG_THREADS) {
            const int which = i / p.N, n = i - which * p.N;
            const float s = red[(0 * 5 + which) * 64 + n] + red[(1 * 5 + which) * 64 + n] +
                            red[(2 * 5 + which) * 64 + n] + red[(3 * 5 + which) * 64 + n];
            p.stats[((long long)bx * nstat + which) * p.N + n] = s;
        }
    }
}

// A fragments of one 32-row tile: lane (row = l&31, h = l>>5) loads A[row][8j + 4h .. +3], j < KS
template <int KS>
__device__ __forceinline__ void thin_load_a(const GemmArgs &p, int m0, int lane, float4 (&a)[KS])
{
    const int row = m0 + (lane & 31), h = lane >> 5;
    const bool ok = row < p.M;
    const float *src = p.A + (long long)(ok ? row : 0) * p.lda + 4 * h;
#pragma unroll
    for (int j = 0; j < KS; ++j) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) v = *(const float4 *)(src + 8 * j);
        if (p.pro_a) {
            const float4 sa = *(const float4 *)(p.pro_a + 8 * j + 4 * h), sc = *(const float4 *)(p.pro_c + 8 * j + 4 * h);
            v.x = fmaxf(fmaf(sa.x, v.x, sc.x), 0.f); v.y = fmaxf(fmaf(sa.y, v.y, sc.y), 0.f);
            v.z = fmaxf(fmaf(sa.z, v.z, sc.z), 0.f); v.w = fmaxf(fmaf(sa.w, v.w, sc.w), 0.f);
            if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        a[j] = v;
    }
}

// Forward form, full tiles (the model's case): every load of the wave -- its 32 input rows, the whole weight matrix as fragments, the
// prologue constants -- is issued before the first wait, and the NT accumulators advance side by side.  [The general body below
// loads fragment by fragment behind bounds checks: a chain of 12+ dependent round trips per wave, 2.7 TB/s at 524288 rows.]  The
// MFMA sequence per accumulator, the statistics' summation order and the stores are those of the general body: bit-identical.
template <int KS, int NT>
__device__ __forceinline__ void thin_fwd_fast(const GemmArgs &p, const int bx, float *red)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m0 = bx * 128 + wave * 32;
    const int h = lane >> 5, cl = lane & 31;
    float4 a[KS], b[NT][KS], sa[KS], sc[KS];
    const float *src = p.A + (long long)(m0 + cl) * p.lda + 4 * h;
#pragma unroll
    for (int j = 0; j < KS; ++j) a[j] = *(const float4 *)(src + 8 * j);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const float *w = p.B + (long long)(nt * 32 + cl) * p.ldb + 4 * h;
#pragma unroll
        for (int j = 0; j < KS; ++j) b[nt][j] = *(const float4 *)(w + 8 * j);
    }
    const bool pro = p.pro_a != nullptr;
    if (pro) {
#pragma unroll
        for (int j = 0; j < KS; ++j) { sa[j] = *(const float4 *)(p.pro_a + 8 * j + 4 * h); sc[j] = *(const float4 *)(p.pro_c + 8 * j + 4 * h); }
    }
    float bias[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bias[nt] = p.bias ? p.bias[nt * 32 + cl] : 0.f;
    if (pro) {
#pragma unroll
        for (int j = 0; j < KS; ++j) {
            a[j].x = fmaxf(fmaf(sa[j].x, a[j].x, sc[j].x), 0.f); a[j].y = fmaxf(fmaf(sa[j].y, a[j].y, sc[j].y), 0.f);
            a[j].z = fmaxf(fmaf(sa[j].z, a[j].z, sc[j].z), 0.f); a[j].w = fmaxf(fmaf(sa[j].w, a[j].w, sc[j].w), 0.f);
        }
    }
    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
#pragma unroll
    for (int j = 0; j < KS; ++j) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j].x, b[nt][j].x, acc[nt], 0, 0, 0);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j].y, b[nt][j].y, acc[nt], 0, 0, 0);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j].z, b[nt][j].z, acc[nt], 0, 0, 0);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j].w, b[nt][j].w, acc[nt], 0, 0, 0);
    }
    const bool want_stats = p.stats != nullptr;
    const float slope = p.act == 1 ? 0.f : 0.1f;                        // act 1 / 2 as one slope select (0 * x would turn -inf into NaN)
    float *crow = p.C + (long long)(m0 + 4 * h) * p.ldc + cl;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float x = acc[nt][r] + bias[nt];
            if (p.act == 1 || p.act == 2) x = x > 0.f ? x : (p.act == 1 ? 0.f : slope * x);
            else if (p.act == 3) x = 1.0f / (1.0f + __expf(-x));
            s1 += x; s2 = fmaf(x, x, s2);
            crow[(long long)((r & 3) + 8 * (r >> 2)) * p.ldc + nt * 32] = x;
        }
        if (want_stats) {
            s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
            if (lane < 32) { float *r0 = red + (size_t)wave * 5 * 64 + nt * 32 + cl; r0[0] = s1; r0[64] = s2; }
        }
    }
    if (want_stats) {
        __syncthreads();
        for (int i = threadIdx.x; i < 2 * p.N; i += TG_THREADS) {
            const int which = i / p.N, n = i - which * p.N;
            const float s = red[(0 * 5 + which) * 64 + n] + red[(1 * 5 + which) * 64 + n] +
                            red[(2 * 5 + which) * 64 + n] + red[(3 * 5 + which) * 64 + n];
            p.stats[((long long)bx * 2 + which) * p.N + n] = s;
        }
    }
}

// C = epi(pro(A) @ W^T): K = 8*KS <= 64, N <= 32*NT
// (body + single / batch entry: cmf_common.h "batched launches")
template <int KS, int NT>
__device__ __forceinline__ void thin_fwd_body(const GemmArgs &p, const int bx)
{
    __shared__ float red[4 * 5 * 64];
    if (p.N == 32 * NT && (long long)bx * 128 + 128 <= p.M && !p.bwd_mode && !p.accumulate && !p.thin_general) {     // (workgroup-uniform)
        thin_fwd_fast<KS, NT>(p, bx, red);
        return;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m0 = bx * 128 + wave * 32;
    const int h = lane >> 5, cl = lane & 31;
    float4 a[KS];
    thin_load_a<KS>(p, m0, lane, a);
    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
        const int n = nt * 32 + cl;
        const float *w = p.B + (long long)(n < p.N ? n : 0) * p.ldb + 4 * h;       // W[n][8j + 4h ..]
#pragma unroll
        for (int j = 0; j < KS; ++j) {
            float4 b = *(const float4 *)(w + 8 * j);
            if (n >= p.N) b = make_float4(0.f, 0.f, 0.f, 0.f);
            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j].x, b.x, acc[nt], 0, 0, 0);
            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j].y, b.y, acc[nt], 0, 0, 0);
            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j].z, b.z, acc[nt], 0, 0, 0);
            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j].w, b.w, acc[nt], 0, 0, 0);
        }
    }
    thin_epilogue<NT>(p, acc, m0, lane, wave, red, bx);
}

template <int KS, int NT>
__global__ __launch_bounds__(TG_THREADS) void thin_fwd_kernel(const GemmArgs p) { thin_fwd_body<KS, NT>(p, blockIdx.x); }

template <int KS, int NT>
__global__ __launch_bounds__(TG_THREADS) void thin_fwd_batch_kernel(const CmfBatch<GemmArgs> b)
{
    const GemmArgs &p = b.a[blockIdx.y];
    if ((long long)blockIdx.x * 128 >= p.M) return;
    thin_fwd_body<KS, NT>(p, blockIdx.x);
}

// C = epi(A @ B), B stored [K][N] (row stride ldb): the data-gradient form dZ @ W
template <int KS, int NT>
__global__ __launch_bounds__(TG_THREADS) void thin_dx_kernel(const GemmArgs p)
{
    __shared__ float red[4 * 5 * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m0 = blockIdx.x * 128 + wave * 32;
    const int h = lane >> 5, cl = lane & 31;
    float4 a[KS];
    thin_load_a<KS>(p, m0, lane, a);
    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
        const int n = nt * 32 + cl;
        const bool ok = n < p.N;
        const float *b0 = p.B + (ok ? n : 0);
#pragma unroll
        for (int j = 0; j < KS; ++j) {
            const int k = 8 * j + 4 * h;                                  // rows k..k+3 of B, column n (coalesced over lanes)
            float bx = b0[(long long)(k + 0) * p.ldb], by = b0[(long long)(k + 1) * p.ldb];
            float bz = b0[(long long)(k + 2) * p.ldb], bw = b0[(long long)(k + 3) * p.ldb];
            if (!ok) { bx = by = bz = bw = 0.f; }
            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j].x, bx, acc[nt], 0, 0, 0);
            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j].y, by, acc[nt], 0, 0, 0);
            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j].z, bz, acc[nt], 0, 0, 0);
            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j].w, bw, acc[nt], 0, 0, 0);
        }
    }
    thin_epilogue<NT>(p, acc, m0, lane, wave, red, blockIdx.x);
}

// Weight gradient: slab[wave][n][k] = sum over the wave's row range of A[m][n] * B'[m][k], A = dZ (rows of MT*32
// channels), B = layer input (rows of KT*32 channels, optionally B' = relu(prob_a[k]*B + prob_c[k])).
// One MFMA step consumes a row PAIR: lanes 0-31 read row 2s, lanes 32-63 row 2s+1, one dword each, coalesced.
template <int MT, int KT>
__global__ __launch_bounds__(TG_THREADS) void thin_dw_kernel(const GemmArgs p, const int rows_per_wg)
{
    // One slab per WORKGROUP: its four waves take a quarter of the workgroup's rows each (16 rows = 8 row pairs in
    // flight per wave per step) and their accumulators are summed through LDS in fixed order (3 + 2, then 1 + 0).
    // [The first version gave every wave its own slab and 256 rows with 8 rows in flight: a 64x64 gradient over
    //  16384 rows took 45 us, pure load latency.]
    __shared__ float red[2][MT * KT * 16 * CMF_WAVE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = lane >> 5, cl = lane & 31;
    const long long wg_begin = (long long)blockIdx.x * rows_per_wg;
    const long long wg_end = wg_begin + rows_per_wg < p.K ? wg_begin + rows_per_wg : p.K;     // p.K = number of rows (contraction)
    const int rpw = ((rows_per_wg + 3) / 4 + 1) / 2 * 2;
    const long long r_begin = wg_begin + (long long)wave * rpw;
    const long long r_end = r_begin + rpw < wg_end ? r_begin + rpw : wg_end;
    float pa[KT], pc[KT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
        const int k = kt * 32 + cl;
        pa[kt] = (p.prob_a && k < p.N) ? p.prob_a[k] : 1.f;
        pc[kt] = (p.prob_a && k < p.N) ? p.prob_c[k] : 0.f;
    }
    f32x16 acc[MT][KT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < KT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    constexpr int U = 8;                                                           // row pairs in flight
    for (long long r0 = r_begin; r0 < r_end; r0 += 2 * U) {
        float av[U][MT], bv[U][KT];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long row = r0 + 2 * u + h;
            const bool ok = row < r_end;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int n = i * 32 + cl;
                av[u][i] = (ok && n < p.M) ? p.A[row * p.lda + n] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < KT; ++j) {
                const int k = j * 32 + cl;
                float v = (ok && k < p.N) ? p.B[row * p.ldb + k] : 0.f;
                if (p.prob_a) { v = fmaxf(fmaf(pa[j], v, pc[j]), 0.f); if (!(ok && k < p.N)) v = 0.f; }
                bv[u][j] = v;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < KT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][i], bv[u][j], acc[i][j], 0, 0, 0);
    }
    // fixed-order reduction of the four waves: (w2, w3) -> LDS, added by (w0, w1); w1 -> LDS, added by w0
    auto put = [&](float *dst) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < KT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) dst[((i * KT + j) * 16 + r) * CMF_WAVE + lane] = acc[i][j][r];
    };
    auto add = [&](const float *src) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < KT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] += src[((i * KT + j) * 16 + r) * CMF_WAVE + lane];
    };
    if (wave >= 2) put(red[wave - 2]);
    __syncthreads();
    if (wave < 2) add(red[wave]);
    __syncthreads();
    if (wave == 1) put(red[0]);
    __syncthreads();
    if (wave != 0) return;
    add(red[0]);
    float *slab = p.C + (long long)blockIdx.x * p.M * p.ldc;                       // [M=out ch][ldc = N=in ch]
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < KT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, k = j * 32 + cl;
                if (n < p.M && k < p.N) slab[(long long)n * p.ldc + k] = acc[i][j][r];
            }
}

// ---- dispatch (called from cmf_gemm) ---------------------------------------------------------------------
template <int KS>
static int launch_fwd_dx(const GemmArgs &g, bool dx, hipStream_t st)
{
    const dim3 grid((g.M + 127) / 128), block(TG_THREADS);
    if (g.N <= 32) {
        if (dx) hipLaunchKernelGGL((thin_dx_kernel<KS, 1>), grid, block, 0, st, g);
        else hipLaunchKernelGGL((thin_fwd_kernel<KS, 1>), grid, block, 0, st, g);
    } else {
        if (dx) hipLaunchKernelGGL((thin_dx_kernel<KS, 2>), grid, block, 0, st, g);
        else hipLaunchKernelGGL((thin_fwd_kernel<KS, 2>), grid, block, 0, st, g);
    }
    return cmf_launch_status();
}

static std::atomic<int> &thin_general_flag()
{
    static std::atomic<int> mode{(getenv("CMF_THIN_GENERAL") && getenv("CMF_THIN_GENERAL")[0] == '1') ? 1 : 0};
    return mode;
}
static int thin_general_mode() { return thin_general_flag().load(std::memory_order_relaxed); }
// diagnostics / tests: 1 = the narrow forward layers take their general body on full tiles too; returns the previous setting
extern "C" int cmf_thin_general(int on)
{
    return thin_general_flag().exchange(on ? 1 : 0);
}

// n <= CMF_MAX_BATCH forward GEMMs C = epi(pro(A) W^T) with the SAME (N, K) in one launch (rows may differ)
int cmf_thin_fwd_batch(int n, const GemmArgs *g, hipStream_t st)
{
    CMF_CHECK_ARG(n >= 1 && n <= CMF_MAX_BATCH && g);
    CmfBatch<GemmArgs> b;
    int max_m = 0;
    for (int i = 0; i < n; ++i) {
        const GemmArgs &q = g[i];
        CMF_CHECK_ARG(q.N == g[0].N && q.K == g[0].K && q.split_k == 1 && q.K % 8 == 0 && q.K >= 8 && q.K <= 64 && q.N <= 64 && q.M >= 1 &&
                      !q.prob_a && q.lda % 4 == 0 && ((uintptr_t)q.A % 16 == 0) && q.ldb % 4 == 0 && ((uintptr_t)q.B % 16 == 0) &&
                      (!q.pro_a || (((uintptr_t)q.pro_a | (uintptr_t)q.pro_c) % 16 == 0)));
        b.a[i] = q;
        b.a[i].thin_general = thin_general_mode();
        max_m = std::max(max_m, q.M);
    }
    const dim3 grid((max_m + 127) / 128, n), block(TG_THREADS);
    const int ks = g[0].K / 8, nt = g[0].N <= 32 ? 1 : 2;
#define CMF_TF(KS_) do { if (nt == 1) hipLaunchKernelGGL((thin_fwd_batch_kernel<KS_, 1>), grid, block, 0, st, b); \
                         else hipLaunchKernelGGL((thin_fwd_batch_kernel<KS_, 2>), grid, block, 0, st, b); } while (0)
    switch (ks) {
        case 1: CMF_TF(1); break;
        case 2: CMF_TF(2); break;
        case 4: CMF_TF(4); break;
        case 8: CMF_TF(8); break;
        default: return (int)hipErrorInvalidValue;
    }
#undef CMF_TF
    for (int i = 0; i < n; ++i) cmf_gemm_count_flops(2.0 * g[i].M * g[i].N * g[i].K);
    return cmf_launch_status();
}

// returns -1 when the shape is not a thin one (the caller falls back to the tiled kernel)
int cmf_thin_gemm(const GemmArgs &g_in, int a_t, int b_t, hipStream_t st)
{
    GemmArgs g = g_in;
    g.thin_general = thin_general_mode();
    if (g.split_k == 1 && !a_t && g.K % 8 == 0 && g.K >= 8 && g.K <= 64 && g.N <= 64 && g.M >= 1 && !g.prob_a &&
        g.lda % 4 == 0 && ((uintptr_t)g.A % 16 == 0) && (b_t ? (g.ldb % 4 == 0 && (uintptr_t)g.B % 16 == 0) : true) &&
        (!g.pro_a || (((uintptr_t)g.pro_a | (uintptr_t)g.pro_c) % 16 == 0))) {
        switch (g.K / 8) {
            case 1: return launch_fwd_dx<1>(g, !b_t, st);
            case 2: return launch_fwd_dx<2>(g, !b_t, st);
            case 4: return launch_fwd_dx<4>(g, !b_t, st);
            case 8: return launch_fwd_dx<8>(g, !b_t, st);
            default: return -1;
        }
    }
    if (g.split_k > 1 && a_t && !b_t && g.M <= 64 && g.N <= 64 && !g.pro_a) {
        // weight gradient: g.M = out channels, g.N = in channels, g.K = rows; g.C = slab workspace [split][M][N]
        const int rpw = (int)((((long long)g.K + g.split_k - 1) / g.split_k + 7) / 8 * 8);      // rows per workgroup = per slab
        const dim3 grid(g.split_k), block(TG_THREADS);
        const int mt = g.M <= 32 ? 1 : 2, kt = g.N <= 32 ? 1 : 2;
        if (mt == 1 && kt == 1) hipLaunchKernelGGL((thin_dw_kernel<1, 1>), grid, block, 0, st, g, rpw);
        else if (mt == 1) hipLaunchKernelGGL((thin_dw_kernel<1, 2>), grid, block, 0, st, g, rpw);
        else if (kt == 1) hipLaunchKernelGGL((thin_dw_kernel<2, 1>), grid, block, 0, st, g, rpw);
        else hipLaunchKernelGGL((thin_dw_kernel<2, 2>), grid, block, 0, st, g, rpw);
        return cmf_launch_status();
    }
    return -1;
}

// ---------------------------------------------------------------------------------------------------------------
// One backward layer of a narrow [linear + BN + ReLU] stack in ONE pass over the rows (cmf_thin_bwd_layer).
//
// As separate kernels the layer is: BN backward in place (read dU, read z, write dZ), the weight gradient (read dZ, read
// the layer input) and the data gradient (read dZ, read the input again for its ReLU mask, write dU_in): 8 row streams
// for the 4 that are needed.  With <= 64 channels all of it is HBM time, so here a workgroup walks its rows once:
//   dZ      = a (dU - s1/M - zhat s2/M)                 formed in registers from dU and z, never stored
//   dU_in   = mask_in(dZ @ W)  (+ the BN-backward partial sums of the layer below, + the dxyz sums)      [thin_dx's part]
//   slab    = sum over the workgroup's rows of dZ^T act_in(x)                                            [thin_dw's part]
// dZ is needed with lanes <-> rows for the first product and lanes <-> channels for the second; the second read of dU / z
// comes from the cache (the same wave touched those lines a few hundred cycles earlier).  The input x is read once, in
// the accumulator layout (lane = channel, registers = rows): it is the mask of the first product's epilogue and, after
// the activation, the B operand of the second, whose contraction index (the row pair of one MFMA step) may be
// assigned freely: register r pairs rows rho(r) and rho(r) + 4, exactly the rows the accumulator layout keeps in r.
// ---------------------------------------------------------------------------------------------------------------
struct ThinBwdArgs {
    long long rows; int cout, cin;
    const float *dU; long long lddu;               // [rows][cout] gradient w.r.t. the layer's BN output, masked by its ReLU
    const float *z; long long ldz;                 // [rows][cout] the layer's pre-BN output
    const float *a, *mean, *invstd, *sums;         // BN of the layer; sums = (s1, s2)[2][cout] or null (eval: dZ = a dU)
    float inv_count;
    const float *w; long long ldw;                 // [cout][cin]
    const float *x; long long ldx;                 // [rows][cin] layer input: pre-BN output of the layer below (in_mode 1) or activated (0)
    int in_mode;
    const float *a_in, *c_in, *mean_in, *invstd_in;
    const float *dxyz;                             // in_mode 1, optional
    float *dx; long long lddx;                     // [rows][cin] or null (no data gradient, no statistics)
    float *stats;                                  // in_mode 1: [tiles128][2 or 5][cin]
    float *slabs;                                  // [gridDim.x][cout][cin] or null (no weight gradient)
    int tiles_per_wg;
    // pooled form (MODE bit 3): dU is not stored -- row m = (p, s) of it is (s == pool_am[p][:]) ? pool_g[p][:] : 0
    const float *pool_g; const unsigned char *pool_am; int pool_S;
};

// LDS of the fused layer
template <int NTO, int NTI>
struct ThinBwdLds {
    static constexpr int LDO = NTO * 32 + 4;
    static constexpr int RED = 2 * NTO * NTI * 16 * CMF_WAVE, DZT = 4 * 32 * LDO;
    float tile[RED > DZT ? RED : DZT];     // per-wave dZ tile [32 rows][cout]; after the row loop the cross-wave sums of the slab
    float wt[NTI * 32 * LDO];              // the weight transposed ([cin][cout]): B operand of the data gradient as one 16-byte read
    float4 dq[4][32];                      // per wave: dxyz of its rows
    float sred[2][4][5][NTI * 32];         // two buffers of per-wave column sums
    float ko[4][NTO * 32], ki[4][NTI * 32];// layer: a, mean, invstd * s2 / M, s1 / M; input layer: a, c, mean, invstd
};

__device__ __forceinline__ float tb_ld(const float *base, unsigned byte_off) { return *(const float *)((const char *)base + byte_off); }
__device__ __forceinline__ float4 tb_ld4(const float *base, unsigned byte_off) { return *(const float4 *)((const char *)base + byte_off); }

// One 128-row tile (32 rows per wave).  FULL: every row and channel of the tile exists -- no predicates around the loads.
// Tile bases are wave-uniform (scalar registers); everything per lane is a 32-bit BYTE offset from them.
// MODE >= 0 fixes the run-time switches at compile time (both products wanted; bit 0: train-mode BN, bit 1: in_mode,
// bit 2: dxyz sums, bit 3: dU comes from the max-pool backward per point): the chain of a set-conv block only uses
// those; MODE < 0 reads them from the arguments (and has no pooled form).
template <int NTO, int NTI, bool FULL, int MODE>
__device__ __forceinline__ void thin_bwd_tile(const ThinBwdArgs &p, ThinBwdLds<NTO, NTI> &S, f32x16 (&accw)[NTO][NTI], long long tile,
                                              unsigned live, int parity, int lane, int wave)
{
    constexpr int LDO = ThinBwdLds<NTO, NTI>::LDO;
    const int h = lane >> 5, cl = lane & 31;
    const bool train = MODE >= 0 ? (MODE & 1) != 0 : p.sums != nullptr;
    const int in_mode = MODE >= 0 ? (MODE >> 1) & 1 : p.in_mode;
    const bool want_q = MODE >= 0 ? (MODE & 4) != 0 : (p.in_mode == 1 && p.dxyz != nullptr);
    const bool has_dx = MODE >= 0 || p.dx != nullptr, has_dw = MODE >= 0 || p.slabs != nullptr;
    const int nstat = want_q ? 5 : 2;
    const int ks = FULL ? NTO * 4 : p.cout >> 3;
    const unsigned cin = (unsigned)p.cin;
    const float *dUt = p.dU + tile * 128 * p.lddu, *zt = train ? p.z + tile * 128 * p.ldz : nullptr;
    const float *xt = p.x + tile * 128 * p.ldx;
    float *dxt = has_dx ? p.dx + tile * 128 * p.lddx : nullptr;
    float *dzt = S.tile + wave * 32 * LDO;
    const unsigned w0 = (unsigned)wave * 32u;
    // Per-lane offsets (LDS and global) and the row strides are made opaque once per tile: as loop invariants the compiler
    // materialises every (offset + constant) address of the tile body as a live register of its own and spills them;
    // behind the empty asm they belong to this iteration, the constants fold into the instructions' offset fields and the
    // row bases stay in scalar registers.
    int oA = (lane & 31) * LDO + 4 * h, oC = 4 * h * LDO + cl, oW = cl * LDO + 4 * h, oK = 4 * h, oN = cl;
    int ldx_s = (int)p.ldx, lddx_s = (int)p.lddx;
    unsigned lx = (4u * h * (unsigned)p.ldx + (unsigned)cl) * 4u, ldx_ = (4u * h * (unsigned)p.lddx + (unsigned)cl) * 4u;
    asm volatile("" : "+v"(oA), "+v"(oC), "+v"(oW), "+v"(oK), "+v"(oN), "+v"(lx), "+v"(ldx_), "+s"(ldx_s), "+s"(lddx_s));
    // ---- layer input in the accumulator layout (lane = channel, register r = rows rho(r) + 4h): mask of the data
    //      gradient, B operand of the weight gradient.  Issued first: it is consumed last. ----
    float xc[NTI][16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const unsigned rho = (unsigned)((r & 3) + 8 * (r >> 2));
        const float *xr = xt + (long long)((int)(w0 + rho) * ldx_s);          // wave-uniform row base
#pragma unroll
        for (int j = 0; j < NTI; ++j) {
            if (FULL) xc[j][r] = tb_ld(xr + j * 32, lx);
            else {
                const unsigned m = w0 + rho + 4u * h, n = (unsigned)(j * 32 + cl);
                xc[j][r] = (m < live && n < cin) ? tb_ld(xr + j * 32, lx) : 0.f;
            }
        }
    }
    if (want_q && lane < 32) {
        const unsigned m = w0 + (unsigned)lane;
        S.dq[wave][lane] = (FULL || m < live) ? *(const float4 *)(p.dxyz + (tile * 128 + m) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // ---- dZ of the wave's 32 rows, rows on the lanes: A operand of the data gradient; parked in LDS for the second product ----
    const unsigned mrow = w0 + (unsigned)(lane & 31);
    const bool ok = FULL || mrow < live;
    const unsigned ou = ((ok ? mrow : 0u) * (unsigned)p.lddu + 4u * h) * 4u, oz = ((ok ? mrow : 0u) * (unsigned)p.ldz + 4u * h) * 4u;
    // pooled form: point and slot of this lane's row; byte offset of its 4 channels in the per-point arrays
    unsigned pool_o = 0, pool_s = 0;
    if (MODE >= 0 && (MODE & 8)) {
        const long long m = tile * 128 + mrow;
        const long long pt = m / p.pool_S;
        pool_s = (unsigned)(m - pt * p.pool_S);
        pool_o = (unsigned)((pt * p.cout + 4 * h) * 4);
    }
    f32x16 acc[NTI];
#pragma unroll
    for (int j = 0; j < NTI; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    constexpr int CH = NTO * 4;                  // every k step of the tile is loaded up front: one exposed load latency per tile
    for (int kb = 0; kb < ks; kb += CH) {
        float4 dv[CH], zv[CH];
#pragma unroll
        for (int ku = 0; ku < CH; ++ku) {
            const int k8 = kb + ku;
            dv[ku] = zv[ku] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k8 < ks) {
                if (MODE >= 0 && (MODE & 8)) {
                    const float4 g4 = tb_ld4(p.pool_g, pool_o + 32u * k8);
                    const uchar4 am = *(const uchar4 *)(p.pool_am + (pool_o >> 2) + 8u * k8);
                    dv[ku] = make_float4(am.x == pool_s ? g4.x : 0.f, am.y == pool_s ? g4.y : 0.f, am.z == pool_s ? g4.z : 0.f,
                                         am.w == pool_s ? g4.w : 0.f);
                } else dv[ku] = tb_ld4(dUt, ou + 32u * k8);
                if (train) zv[ku] = tb_ld4(zt, oz + 32u * k8);
            }
        }
        // (32 -> 32 channels: a compiler fence keeps the loads of all k steps in front of the arithmetic -- the scheduler otherwise sinks
        //  each step's loads to their use: 35.4 -> 32.4 us.  The wider forms have no registers for that at their occupancy: 46 -> 47.5 us)
        if (NTO == 1 && NTI == 1) asm volatile("" ::: "memory");
#pragma unroll
        for (int ku = 0; ku < CH; ++ku) {
            const int k8 = kb + ku;
            if (k8 >= ks) break;
            float4 d = dv[ku];
            const float4 sa = *(const float4 *)&S.ko[0][oK + 8 * k8];
            if (train) {
                const float4 v = zv[ku];
                const float4 mu = *(const float4 *)&S.ko[1][oK + 8 * k8], u = *(const float4 *)&S.ko[2][oK + 8 * k8],
                             t1 = *(const float4 *)&S.ko[3][oK + 8 * k8];
                d.x = sa.x * (d.x - t1.x - (v.x - mu.x) * u.x);
                d.y = sa.y * (d.y - t1.y - (v.y - mu.y) * u.y);
                d.z = sa.z * (d.z - t1.z - (v.z - mu.z) * u.z);
                d.w = sa.w * (d.w - t1.w - (v.w - mu.w) * u.w);
            } else { d.x *= sa.x; d.y *= sa.y; d.z *= sa.z; d.w *= sa.w; }
            if (!ok) d = make_float4(0.f, 0.f, 0.f, 0.f);
            if (has_dw) *(float4 *)(dzt + oA + 8 * k8) = d;
            if (has_dx) {
#pragma unroll
                for (int j = 0; j < NTI; ++j) {
                    const float4 b = *(const float4 *)&S.wt[oW + j * 32 * LDO + 8 * k8];
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(d.x, b.x, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(d.y, b.y, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(d.z, b.z, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(d.w, b.w, acc[j], 0, 0, 0);
                }
            }
        }
    }
    if (has_dx) {
        // epilogue of the data gradient: ReLU mask of the layer below, its BN-backward partial sums, store
        float (*sr)[5][NTI * 32] = S.sred[parity];
#pragma unroll
        for (int j = 0; j < NTI; ++j) {
            const unsigned n = (unsigned)(j * 32 + cl);
            const bool nok = FULL || n < cin;
            const float ia = S.ki[0][oN + j * 32], ic = S.ki[1][oN + j * 32], im = S.ki[2][oN + j * 32], ii = S.ki[3][oN + j * 32];
            float s1 = 0.f, s2 = 0.f, q0 = 0.f, q1 = 0.f, q2 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const unsigned rho = (unsigned)((r & 3) + 8 * (r >> 2)), mr = rho + 4u * h;
                if (FULL || (w0 + mr < live && nok)) {
                    float v = acc[j][r];
                    if (in_mode == 1) {
                        const float zz = xc[j][r];
                        v = (fmaf(ia, zz, ic) > 0.f) ? v : 0.f;
                        s1 += v; s2 += v * ((zz - im) * ii);
                        if (want_q) { const float4 dd = S.dq[wave][mr]; q0 += v * dd.x; q1 += v * dd.y; q2 += v * dd.z; }
                    }
                    float *dr = dxt + (long long)((int)(w0 + rho) * lddx_s) + j * 32;
                    *(float *)((char *)dr + ldx_) = v;
                }
            }
            if (in_mode == 1) {
                s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
                if (want_q) { q0 += __shfl_xor(q0, 32, 64); q1 += __shfl_xor(q1, 32, 64); q2 += __shfl_xor(q2, 32, 64); }
                if (lane < 32) {
                    sr[wave][0][j * 32 + cl] = s1; sr[wave][1][j * 32 + cl] = s2;
                    if (want_q) { sr[wave][2][j * 32 + cl] = q0; sr[wave][3][j * 32 + cl] = q1; sr[wave][4][j * 32 + cl] = q2; }
                }
            }
        }
        if (in_mode == 1) {
            __syncthreads();                // one barrier per tile: the buffer written now is read below, the other one two tiles apart
            for (int i = threadIdx.x; i < nstat * p.cin; i += TG_THREADS) {
                const int which = i / p.cin, n = i - which * p.cin;
                p.stats[(tile * nstat + which) * p.cin + n] = sr[0][which][n] + sr[1][which][n] + sr[2][which][n] + sr[3][which][n];
            }
        }
    }
    if (has_dw) {
        // ---- weight gradient: channels on the lanes, MFMA step r contracts the row pair (rho(r), rho(r) + 4) ----
        if (in_mode == 1) {
#pragma unroll
            for (int j = 0; j < NTI; ++j) {
                const float ia = S.ki[0][oN + j * 32], ic = S.ki[1][oN + j * 32];
#pragma unroll
                for (int r = 0; r < 16; ++r) xc[j][r] = fmaxf(fmaf(ia, xc[j][r], ic), 0.f);
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float *src = dzt + oC + ((r & 3) + 8 * (r >> 2)) * LDO;
#pragma unroll
            for (int i = 0; i < NTO; ++i) {
                const float dz = src[i * 32];
#pragma unroll
                for (int j = 0; j < NTI; ++j)
                    accw[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(dz, xc[j][r], accw[i][j], 0, 0, 0);
            }
        }
    }
}

// FULL (decided by the host): rows % 128 == 0, cout == NTO * 32, cin == NTI * 32 -- the kernel then contains no ragged-edge code
template <int NTO, int NTI, int MODE, bool FULL>
__device__ __forceinline__ void thin_bwd_layer_body(const ThinBwdArgs &p, const int bx)
{
    constexpr int LDO = ThinBwdLds<NTO, NTI>::LDO;
    __shared__ ThinBwdLds<NTO, NTI> S;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // uniform: row bases stay scalar
    const int h = lane >> 5, cl = lane & 31;
    const bool train = p.sums != nullptr;
    if (threadIdx.x < NTO * 32) {
        const int n = threadIdx.x;
        const bool ok = n < p.cout;
        S.ko[0][n] = ok ? p.a[n] : 0.f;
        S.ko[1][n] = (ok && train) ? p.mean[n] : 0.f;
        S.ko[2][n] = (ok && train) ? p.invstd[n] * (p.sums[p.cout + n] * p.inv_count) : 0.f;
        S.ko[3][n] = (ok && train) ? p.sums[n] * p.inv_count : 0.f;
    } else if (threadIdx.x >= 64 && threadIdx.x < 64 + NTI * 32) {
        const int n = threadIdx.x - 64;
        const bool ok = n < p.cin && p.in_mode == 1;
        S.ki[0][n] = ok ? p.a_in[n] : 0.f; S.ki[1][n] = ok ? p.c_in[n] : 0.f;
        S.ki[2][n] = ok ? p.mean_in[n] : 0.f; S.ki[3][n] = ok ? p.invstd_in[n] : 0.f;
    }
    if (p.dx)
        for (int i = threadIdx.x; i < NTO * 32 * NTI * 32; i += TG_THREADS) {
            const int k = i / (NTI * 32), n = i - k * (NTI * 32);
            S.wt[n * LDO + k] = (k < p.cout && n < p.cin) ? p.w[(long long)k * p.ldw + n] : 0.f;
        }
    __syncthreads();
    f32x16 accw[NTO][NTI];
#pragma unroll
    for (int i = 0; i < NTO; ++i)
#pragma unroll
        for (int j = 0; j < NTI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) accw[i][j][r] = 0.f;
    for (int t = 0; t < p.tiles_per_wg; ++t) {
        const long long tile = (long long)bx * p.tiles_per_wg + t;
        if (tile * 128 >= p.rows) break;                                         // uniform over the workgroup
        const long long left = p.rows - tile * 128;
        thin_bwd_tile<NTO, NTI, FULL, MODE>(p, S, accw, tile, (FULL || left >= 128) ? 128u : (unsigned)left, t & 1, lane, wave);
    }
    if (!p.slabs) return;
    __syncthreads();                                                             // the dZ tiles are dead: LDS becomes the reduction buffer
    float (*red)[NTO * NTI * 16 * CMF_WAVE] = (float (*)[NTO * NTI * 16 * CMF_WAVE])S.tile;
    // fixed-order reduction of the four waves: (w2, w3) -> LDS, added by (w0, w1); w1 -> LDS, added by w0
    auto put = [&](float *dst) {
#pragma unroll
        for (int i = 0; i < NTO; ++i)
#pragma unroll
            for (int j = 0; j < NTI; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) dst[((i * NTI + j) * 16 + r) * CMF_WAVE + lane] = accw[i][j][r];
    };
    auto add = [&](const float *src) {
#pragma unroll
        for (int i = 0; i < NTO; ++i)
#pragma unroll
            for (int j = 0; j < NTI; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) accw[i][j][r] += src[((i * NTI + j) * 16 + r) * CMF_WAVE + lane];
    };
    if (wave >= 2) put(red[wave - 2]);
    __syncthreads();
    if (wave < 2) add(red[wave]);
    __syncthreads();
    if (wave == 1) put(red[0]);
    __syncthreads();
    if (wave != 0) return;
    add(red[0]);
    float *slab = p.slabs + (long long)bx * p.cout * p.cin;
#pragma unroll
    for (int i = 0; i < NTO; ++i)
#pragma unroll
        for (int j = 0; j < NTI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, k = j * 32 + cl;
                if (n < p.cout && k < p.cin) slab[(long long)n * p.cin + k] = accw[i][j][r];
            }
}

template <int NTO, int NTI, int MODE, bool FULL>
__global__ __launch_bounds__(TG_THREADS, NTI == 1 ? 3 : 2) void thin_bwd_layer_kernel(const ThinBwdArgs p)
{
    thin_bwd_layer_body<NTO, NTI, MODE, FULL>(p, blockIdx.x);
}

// batch form: every problem has the same slab count (gridDim.x), see thin_bwd_layer_batch
template <int NTO, int NTI, int MODE, bool FULL>
__global__ __launch_bounds__(TG_THREADS, NTI == 1 ? 3 : 2) void thin_bwd_layer_batch_kernel(const CmfBatch<ThinBwdArgs> b)
{
    const ThinBwdArgs &p = b.a[blockIdx.y];
    if ((long long)blockIdx.x * p.tiles_per_wg * 128 >= p.rows) return;        // past this problem's last slab
    thin_bwd_layer_body<NTO, NTI, MODE, FULL>(p, blockIdx.x);
}

// ---------------------------------------------------------------------------------------------------------------
// The same fused backward layer for a WIDE input: cout = 64, cin a multiple of 128 (the third conv of a second-encoder
// block, 256 -> 64 channels per neighbour row).  As tiled GEMMs its two products are the worst shapes of the step -- a
// data gradient with K = 64 (four K-chunks per 128 x 128 tile: all prologue and epilogue) and a weight gradient with 64
// output rows: 39 TF each, 0.87 ms for the largest scale, plus the max-pool backward and the BN backward as passes of
// their own.  Here a workgroup owns a range of rows and a HALF of 128 input channels (grid.y): dZ of a 128-row tile is
// formed once into the per-wave LDS tiles (from the per-point pooled gradient or a stored dU), then for each of its two
// 64-channel chunks the wave runs the masked data gradient (A from the LDS tile, B = W^T chunk resident in LDS) and the
// weight gradient (register r = row pair, as above).  The weight-gradient accumulators of both chunks (128 registers)
// stay live across the workgroup's tiles; dZ is recomputed by the two channel halves (its inputs are 1/5 of the traffic).
// ---------------------------------------------------------------------------------------------------------------
constexpr int TBW_LDO = 68;
struct ThinBwdWideLds {
    float tile[4 * 32 * TBW_LDO];           // per-wave dZ tile [32 rows][64]
    float wt[64 * TBW_LDO];                 // W^T of this workgroup's 64 input channels: [n][k]
    float sred[4][2][64];                   // per-wave column sums of the current tile
    float ko[4][64], ki[4][64];
};

// One workgroup = (a range of 128-row tiles) x (ONE chunk of 64 input channels): 64 weight-gradient accumulators per lane
// leave room to request the next tile's layer input while the current tile computes.  The cin / 64 workgroups of a row range
// get consecutive slots on the SAME XCD (id % 8), so the dZ inputs they all read come from that XCD's L2 after the first.
// timing ablations (tools/diag builds with -DCMF_TBW_DIAG=bits, results invalid): 1 no dZ inputs (constants), 2 no dx stores, 4 no per-tile
// statistics barriers / stores, 8 no weight-gradient MFMAs, 16 no data-gradient MFMAs.  [Also tried, round 6: the two co-resident workgroups at different wave priorities so
// that their phases fall out of step -- 512-522 against 522-525 us at 524288 rows, noise elsewhere.]
#ifndef CMF_TBW_DIAG
#define CMF_TBW_DIAG 0
#endif
template <int MODE>                          // bit 0: train-mode BN, bit 3: pooled dU (in_mode 1, both products, whole tiles)
__global__ __launch_bounds__(TG_THREADS, 2) void thin_bwd_wide_kernel(const ThinBwdArgs p, const int nchunk, const int nslab)
{
    constexpr int LDO = TBW_LDO;
    constexpr bool train = (MODE & 1) != 0, pooled = (MODE & 8) != 0;
    __shared__ ThinBwdWideLds S;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // uniform: row bases stay scalar
    const int h = lane >> 5, cl = lane & 31;
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int chunk = q % nchunk, slab = (q / nchunk) * 8 + xcd;
    if (slab >= nslab) return;
    const int col0 = chunk * 64;                           // first input channel of this workgroup
    if (threadIdx.x < 64) {
        const int n = threadIdx.x;
        S.ko[0][n] = p.a[n];
        S.ko[1][n] = train ? p.mean[n] : 0.f;
        S.ko[2][n] = train ? p.invstd[n] * (p.sums[64 + n] * p.inv_count) : 0.f;
        S.ko[3][n] = train ? p.sums[n] * p.inv_count : 0.f;
    } else if (threadIdx.x < 128) {
        const int n = threadIdx.x - 64;
        S.ki[0][n] = p.a_in[col0 + n]; S.ki[1][n] = p.c_in[col0 + n];
        S.ki[2][n] = p.mean_in[col0 + n]; S.ki[3][n] = p.invstd_in[col0 + n];
    }
    for (int i = threadIdx.x; i < 64 * 64; i += TG_THREADS) {
        const int k = i >> 6, n = i & 63;
        S.wt[n * LDO + k] = p.w[(long long)k * p.ldw + col0 + n];
    }
    __syncthreads();
    float *dzt = S.tile + wave * 32 * LDO;
    f32x16 accw[2][2];                                      // [cout block][cin block]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) accw[i][j][r] = 0.f;
    const unsigned w0 = (unsigned)wave * 32u;
    const unsigned lane_x = (4u * h * (unsigned)p.ldx + (unsigned)cl) * 4u, lane_dx = (4u * h * (unsigned)p.lddx + (unsigned)cl) * 4u;
    const long long tile0 = (long long)slab * p.tiles_per_wg;
    const long long ntile = (p.rows + 127) / 128;

    // layer input of a tile in the accumulator layout (row bases are wave-uniform -> scalar registers)
    auto load_x = [&](long long tile, float (&dst)[2][16]) {
        unsigned lx = lane_x;
        int ldx_s = (int)p.ldx;
        asm volatile("" : "+v"(lx), "+s"(ldx_s));
        const float *xt = p.x + tile * 128 * p.ldx + col0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float *xr = xt + (long long)((int)(w0 + (unsigned)((r & 3) + 8 * (r >> 2))) * ldx_s);
#pragma unroll
            for (int j = 0; j < 2; ++j) dst[j][r] = tb_ld(xr + j * 32, lx);
        }
    };
    float xc[2][16];
    if (tile0 < ntile) load_x(tile0, xc);

    for (int t = 0; t < p.tiles_per_wg; ++t) {
        const long long tile = tile0 + t;
        if (tile >= ntile) break;
        // per-lane offsets made opaque once per tile (see thin_bwd_tile)
        int oA = (lane & 31) * LDO + 4 * h, oC = 4 * h * LDO + cl, oW = cl * LDO + 4 * h, oK = 4 * h, oN = cl;
        unsigned ldx_ = lane_dx;
        int lddx_s = (int)p.lddx;
        asm volatile("" : "+v"(oA), "+v"(oC), "+v"(oW), "+v"(oK), "+v"(oN), "+v"(ldx_), "+s"(lddx_s));
        const float *zt = train ? p.z + tile * 128 * p.ldz : nullptr;
        float *dxt = p.dx + tile * 128 * p.lddx + col0;
        // ---- dZ of the wave's 32 rows (rows on the lanes) into its LDS tile ----
        {
            const unsigned mrow = w0 + (unsigned)(lane & 31);
            const unsigned oz = (mrow * (unsigned)p.ldz + 4u * h) * 4u;
            unsigned ou = 0, pool_s = 0;
            const float *src = nullptr;
            if (pooled) {
                const long long m = tile * 128 + mrow;
                const long long pt = m / p.pool_S;
                pool_s = (unsigned)(m - pt * p.pool_S);
                ou = (unsigned)((pt * 64 + 4 * h) * 4);
                src = p.pool_g;
            } else {
                ou = (mrow * (unsigned)p.lddu + 4u * h) * 4u;
                src = p.dU + tile * 128 * p.lddu;
            }
            float4 dv[8], zv[8];
#pragma unroll
            for (int k8 = 0; k8 < 8; ++k8) {
                if (CMF_TBW_DIAG & 1) { dv[k8] = make_float4(1.f, 2.f, 3.f, 4.f); zv[k8] = make_float4(0.5f, 0.25f, 0.125f, 1.f); continue; }
                if (pooled) {
                    const float4 g4 = tb_ld4(src, ou + 32u * k8);
                    const uchar4 am = *(const uchar4 *)(p.pool_am + (ou >> 2) + 8u * k8);
                    dv[k8] = make_float4(am.x == pool_s ? g4.x : 0.f, am.y == pool_s ? g4.y : 0.f, am.z == pool_s ? g4.z : 0.f,
                                         am.w == pool_s ? g4.w : 0.f);
                } else dv[k8] = tb_ld4(src, ou + 32u * k8);
                zv[k8] = train ? tb_ld4(zt, oz + 32u * k8) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int k8 = 0; k8 < 8; ++k8) {
                float4 d = dv[k8];
                const float4 sa = *(const float4 *)&S.ko[0][oK + 8 * k8];
                if (train) {
                    const float4 v = zv[k8];
                    const float4 mu = *(const float4 *)&S.ko[1][oK + 8 * k8], u = *(const float4 *)&S.ko[2][oK + 8 * k8],
                                 t1 = *(const float4 *)&S.ko[3][oK + 8 * k8];
                    d.x = sa.x * (d.x - t1.x - (v.x - mu.x) * u.x);
                    d.y = sa.y * (d.y - t1.y - (v.y - mu.y) * u.y);
                    d.z = sa.z * (d.z - t1.z - (v.z - mu.z) * u.z);
                    d.w = sa.w * (d.w - t1.w - (v.w - mu.w) * u.w);
                } else { d.x *= sa.x; d.y *= sa.y; d.z *= sa.z; d.w *= sa.w; }
                *(float4 *)(dzt + oA + 8 * k8) = d;
            }
        }
        // the next tile's layer input is requested now and lands while this tile computes
        float xn[2][16];
        const bool more = t + 1 < p.tiles_per_wg && tile + 1 < ntile;
        if (more) load_x(tile + 1, xn);
        // data gradient: A from the LDS tile, B = W^T chunk
        f32x16 acc[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
        for (int k8 = 0; k8 < 8; ++k8) {
            const float4 d = *(const float4 *)(dzt + oA + 8 * k8);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float4 b = *(const float4 *)&S.wt[oW + j * 32 * LDO + 8 * k8];
                if (CMF_TBW_DIAG & 16) { acc[j][k8] += d.x * b.x; continue; }
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(d.x, b.x, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(d.y, b.y, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(d.z, b.z, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(d.w, b.w, acc[j], 0, 0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = oN + j * 32;
            const float ia = S.ki[0][n], ic = S.ki[1][n], im = S.ki[2][n], ii = S.ki[3][n];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float zz = xc[j][r];
                const float v = (fmaf(ia, zz, ic) > 0.f) ? acc[j][r] : 0.f;
                s1 += v; s2 += v * ((zz - im) * ii);
                float *dr = dxt + (long long)((int)(w0 + (unsigned)((r & 3) + 8 * (r >> 2))) * lddx_s) + j * 32;
                if (!(CMF_TBW_DIAG & 2)) *(float *)((char *)dr + ldx_) = v;
                else asm volatile("" :: "v"(v), "v"(dr));
                xc[j][r] = fmaxf(fmaf(ia, zz, ic), 0.f);           // activated: B operand of the weight gradient
            }
            s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
            if (lane < 32) { S.sred[wave][0][n] = s1; S.sred[wave][1][n] = s2; }
        }
        // weight gradient
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float *src = dzt + oC + ((r & 3) + 8 * (r >> 2)) * LDO;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float dz = src[i * 32];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (CMF_TBW_DIAG & 8) { accw[i][j][r] += dz * xc[j][r]; continue; }
                    accw[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(dz, xc[j][r], accw[i][j], 0, 0, 0);
                }
            }
        }
        if (!(CMF_TBW_DIAG & 4)) {
        __syncthreads();
        if (threadIdx.x < 128) {
            const int which = threadIdx.x >> 6, n = threadIdx.x & 63;
            p.stats[(tile * 2 + which) * p.cin + col0 + n] = S.sred[0][which][n] + S.sred[1][which][n] + S.sred[2][which][n] + S.sred[3][which][n];
        }
        __syncthreads();                    // the sums are rewritten by the next tile
        }
        if (more) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) xc[j][r] = xn[j][r];
        }
    }
    __syncthreads();                        // tiles are dead: their LDS becomes the reduction buffer (2 x 4 blocks x 4 KB)
    float (*red)[4 * 16 * CMF_WAVE] = (float (*)[4 * 16 * CMF_WAVE])S.tile;
    static_assert(sizeof(S.tile) >= 2 * 4 * 16 * CMF_WAVE * sizeof(float), "reduction buffer");
    auto put = [&](float *dst) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) dst[(((i * 2 + j) * 16) + r) * CMF_WAVE + lane] = accw[i][j][r];
    };
    auto add = [&](const float *src) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) accw[i][j][r] += src[(((i * 2 + j) * 16) + r) * CMF_WAVE + lane];
    };
    if (wave >= 2) put(red[wave - 2]);
    __syncthreads();
    if (wave < 2) add(red[wave]);
    __syncthreads();
    if (wave == 1) put(red[0]);
    __syncthreads();
    if (wave != 0) return;
    add(red[0]);
    float *slabp = p.slabs + (long long)slab * 64 * p.cin + col0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, k = j * 32 + cl;
                slabp[(long long)n * p.cin + k] = accw[i][j][r];
            }
}

extern "C" int cmf_thin_bwd_wide_supported(int cout, int cin) { return cout == 64 && cin >= 128 && cin % 64 == 0 && cin <= 1024; }

// row slabs of the wide form: one round of 512 resident workgroups over (row slabs) x (cin / 64 channel chunks)
extern "C" int cmf_thin_bwd_wide_slabs(long long rows, int cin, int *tiles_per_wg)
{
    const long long tiles = (rows + 127) / 128;
    long long split = 512 / (cin / 64 > 0 ? cin / 64 : 1);
    if (split < 1) split = 1;
    if (split > tiles) split = tiles;
    if (split < 1) split = 1;
    const long long tpw = (tiles + split - 1) / split;
    if (tiles_per_wg) *tiles_per_wg = (int)tpw;
    return (int)((tiles + tpw - 1) / tpw);
}

// dU given ([rows][64], row stride lddu) or pooled (pool_g != NULL: rows = P * pool_S, see cmf_thin_bwd_layer_pooled)
extern "C" int cmf_thin_bwd_wide_layer(long long rows, int cin, const float *dU, long long lddu, const float *pool_g,
                                       const unsigned char *pool_am, int pool_S, const float *z, long long ldz,
                                       const float *a, const float *mean, const float *invstd, const float *sums,
                                       const float *w, long long ldw, const float *x, long long ldx,
                                       const float *a_in, const float *c_in, const float *mean_in, const float *invstd_in,
                                       float *dx, long long lddx, float *stats, float *dw, long long lddw, int accumulate, float *slabs,
                                       void *stream)
{
    const bool pooled = pool_g != nullptr;
    CMF_CHECK_ARG(rows >= 0 && rows % 128 == 0 && cmf_thin_bwd_wide_supported(64, cin));
    if (rows == 0) return 0;
    CMF_CHECK_ARG((pooled || dU) && a && w && x && (!sums || (z && mean && invstd)) && dx && dw && stats && slabs);
    CMF_CHECK_ARG(a_in && c_in && mean_in && invstd_in);
    CMF_CHECK_ARG(pooled || (lddu % 4 == 0 && (uintptr_t)dU % 16 == 0));
    CMF_CHECK_ARG(!sums || (ldz % 4 == 0 && (uintptr_t)z % 16 == 0));
    CMF_CHECK_ARG(128ll * std::max({lddu, ldz, ldx, lddx}) * 4 < (1ll << 31));
    if (pooled) CMF_CHECK_ARG(pool_am && pool_S > 0 && rows % pool_S == 0 && (uintptr_t)pool_g % 16 == 0 && (uintptr_t)pool_am % 4 == 0 &&
                              (rows / pool_S) * 64 * 4 < (1ll << 32));
    ThinBwdArgs p;
    p.rows = rows; p.cout = 64; p.cin = cin; p.dU = dU; p.lddu = lddu; p.z = z; p.ldz = ldz;
    p.a = a; p.mean = mean; p.invstd = invstd; p.sums = sums; p.inv_count = (float)(1.0 / (double)rows);
    p.w = w; p.ldw = ldw; p.x = x; p.ldx = ldx; p.in_mode = 1;
    p.a_in = a_in; p.c_in = c_in; p.mean_in = mean_in; p.invstd_in = invstd_in; p.dxyz = nullptr;
    p.dx = dx; p.lddx = lddx; p.stats = stats; p.slabs = slabs;
    p.pool_g = pool_g; p.pool_am = pool_am; p.pool_S = pool_S;
    const int nslab = cmf_thin_bwd_wide_slabs(rows, cin, &p.tiles_per_wg);
    hipStream_t st = (hipStream_t)stream;
    const int nchunk = cin / 64;
    const dim3 grid(8 * ((nslab + 7) / 8) * nchunk), block(TG_THREADS);
    const int mode = (sums ? 1 : 0) | (pooled ? 8 : 0);
    switch (mode) {
        case 0: hipLaunchKernelGGL((thin_bwd_wide_kernel<0>), grid, block, 0, st, p, nchunk, nslab); break;
        case 1: hipLaunchKernelGGL((thin_bwd_wide_kernel<1>), grid, block, 0, st, p, nchunk, nslab); break;
        case 8: hipLaunchKernelGGL((thin_bwd_wide_kernel<8>), grid, block, 0, st, p, nchunk, nslab); break;
        default: hipLaunchKernelGGL((thin_bwd_wide_kernel<9>), grid, block, 0, st, p, nchunk, nslab); break;
    }
    int err = cmf_launch_status();
    if (err) return err;
    cmf_gemm_count_flops(4.0 * rows * 64 * cin);
    return cmf_splitk_reduce(64, cin, nslab, slabs, dw, lddw, accumulate, st);
}

// slabs of the fused layer for `rows` rows: workgroups take whole 128-row tiles.  One round of resident workgroups: three
// per CU for <= 32 input channels (<= 168 registers), two otherwise -- 1024 slabs of 4 tiles ran 1.33 rounds, a third of
// the chip idle in the second.  Never more than dw_split's thin rule max(2, min(rows / 128, 1024)) sizes the workspace for.
static int thin_bwd_slabs_for(long long rows, int cap, int *tiles_per_wg)
{
    const long long tiles = (rows + 127) / 128;
    long long split = rows / 128 < cap ? rows / 128 : cap;
    if (split < 2) split = 2;
    const long long tpw = (tiles + split - 1) / split;
    if (tiles_per_wg) *tiles_per_wg = (int)tpw;
    return (int)((tiles + tpw - 1) / tpw);
}

// upper bound of the slab count (what a caller sizes `slabs` with)
extern "C" int cmf_thin_bwd_slabs(long long rows, int *tiles_per_wg) { return thin_bwd_slabs_for(rows, 768, tiles_per_wg); }

extern "C" int cmf_thin_bwd_supported(int cout, int cin)
{
    return cout >= 8 && cout <= 64 && cout % 8 == 0 && cin >= 1 && cin <= 64;
}

static int thin_bwd_layer_impl(long long rows, int cout, int cin, const float *dU, long long lddu, const float *pool_g,
                               const unsigned char *pool_am, int pool_S, const float *z, long long ldz,
                               const float *a, const float *mean, const float *invstd, const float *sums,
                               const float *w, long long ldw, const float *x, long long ldx, int in_mode,
                               const float *a_in, const float *c_in, const float *mean_in, const float *invstd_in, const float *dxyz,
                               float *dx, long long lddx, float *stats, float *dw, long long lddw, int accumulate, float *slabs,
                               void *stream)
{
    const bool pooled = pool_g != nullptr;
    CMF_CHECK_ARG(rows >= 0 && cmf_thin_bwd_supported(cout, cin) && (in_mode == 0 || in_mode == 1));
    if (rows == 0) return 0;
    CMF_CHECK_ARG((pooled || dU) && a && w && x && (!sums || (z && mean && invstd)) && (dx || dw));
    CMF_CHECK_ARG(pooled || (lddu % 4 == 0 && (uintptr_t)dU % 16 == 0));
    CMF_CHECK_ARG(!sums || (ldz % 4 == 0 && (uintptr_t)z % 16 == 0));
    CMF_CHECK_ARG(((uintptr_t)a | (uintptr_t)mean | (uintptr_t)invstd | (uintptr_t)sums) % 16 == 0);
    CMF_CHECK_ARG(in_mode == 0 || !dx || (a_in && c_in && mean_in && invstd_in && stats));
    CMF_CHECK_ARG(in_mode == 0 || (a_in && c_in));
    CMF_CHECK_ARG(!dw || slabs);
    // per-lane offsets inside a tile (and inside the per-point arrays of the pooled form) are 32-bit byte offsets
    CMF_CHECK_ARG(128ll * std::max({lddu, ldz, ldx, lddx}) * 4 < (1ll << 31));
    const bool full = rows % 128 == 0 && cout % 32 == 0 && cin % 32 == 0;
    if (pooled) CMF_CHECK_ARG(pool_am && pool_S > 0 && rows % pool_S == 0 && (uintptr_t)pool_g % 16 == 0 && (uintptr_t)pool_am % 4 == 0 &&
                              dx && dw && full && (rows / pool_S) * cout * 4 < (1ll << 32));
    ThinBwdArgs p;
    p.rows = rows; p.cout = cout; p.cin = cin; p.dU = dU; p.lddu = lddu; p.z = z; p.ldz = ldz;
    p.a = a; p.mean = mean; p.invstd = invstd; p.sums = sums; p.inv_count = (float)(1.0 / (double)rows);
    p.w = w; p.ldw = ldw; p.x = x; p.ldx = ldx; p.in_mode = in_mode;
    p.a_in = a_in; p.c_in = c_in; p.mean_in = mean_in; p.invstd_in = invstd_in; p.dxyz = in_mode == 1 ? dxyz : nullptr;
    p.dx = dx; p.lddx = lddx; p.stats = stats; p.slabs = dw ? slabs : nullptr;
    p.pool_g = pool_g; p.pool_am = pool_am; p.pool_S = pool_S;
    const int nslab = thin_bwd_slabs_for(rows, cin > 32 ? 512 : 768, &p.tiles_per_wg);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(nslab), block(TG_THREADS);
    // the set-conv chains always want both products of whole tiles: their switch combinations are compiled in (MODE,
    // FULL); anything else runs the generic instantiation
    int mode = -1;
    if (dx && dw && full) mode = (sums ? 1 : 0) | (in_mode ? 2 : 0) | ((in_mode && dxyz) ? 4 : 0) | (pooled ? 8 : 0);
#define CMF_TB_MODE(NO, NI)                                                                              \
    switch (mode) {                                                                                     \
        case 0: hipLaunchKernelGGL((thin_bwd_layer_kernel<NO, NI, 0, true>), grid, block, 0, st, p); break;   \
        case 1: hipLaunchKernelGGL((thin_bwd_layer_kernel<NO, NI, 1, true>), grid, block, 0, st, p); break;   \
        case 2: hipLaunchKernelGGL((thin_bwd_layer_kernel<NO, NI, 2, true>), grid, block, 0, st, p); break;   \
        case 3: hipLaunchKernelGGL((thin_bwd_layer_kernel<NO, NI, 3, true>), grid, block, 0, st, p); break;   \
        case 6: hipLaunchKernelGGL((thin_bwd_layer_kernel<NO, NI, 6, true>), grid, block, 0, st, p); break;   \
        case 7: hipLaunchKernelGGL((thin_bwd_layer_kernel<NO, NI, 7, true>), grid, block, 0, st, p); break;   \
        case 10: hipLaunchKernelGGL((thin_bwd_layer_kernel<NO, NI, 10, true>), grid, block, 0, st, p); break; \
        case 11: hipLaunchKernelGGL((thin_bwd_layer_kernel<NO, NI, 11, true>), grid, block, 0, st, p); break; \
        default:                                                                                        \
            if (pooled) return (int)hipErrorInvalidValue;     /* the pooled form feeds a BN + ReLU layer (in_mode 1, no dxyz) */ \
            hipLaunchKernelGGL((thin_bwd_layer_kernel<NO, NI, -1, false>), grid, block, 0, st, p); break; \
    }
    if (cout > 32) {
        if (cin > 32) { CMF_TB_MODE(2, 2) } else { CMF_TB_MODE(2, 1) }
    } else {
        if (cin > 32) { CMF_TB_MODE(1, 2) } else { CMF_TB_MODE(1, 1) }
    }
#undef CMF_TB_MODE
    int err = cmf_launch_status();
    if (err) return err;
    cmf_gemm_count_flops((dx ? 2.0 : 0.0) * rows * cout * cin + (dw ? 2.0 : 0.0) * rows * cout * cin);
    if (dw) return cmf_splitk_reduce(cout, cin, nslab, slabs, dw, lddw, accumulate, st);
    return 0;
}

// n <= CMF_MAX_BATCH fused backward layers in ONE launch: dense dU, both products, whole 128-row tiles, the same channel
// widths and the same mode (train / eval BN, BN + ReLU or activated input) for all -- the per-point tails of the scales of
// an encoder call.  Launches the layer kernel only; c[i].nslab = the slab count of problem i, to be summed by the caller
// (cmf_splitk_reduce_batch).
int cmf_thin_bwd_layer_batch(int n, CmfThinBwdCall *c, hipStream_t st)
{
    CMF_CHECK_ARG(n >= 1 && n <= CMF_MAX_BATCH && c);
    CmfBatch<ThinBwdArgs> b;
    int max_slab = 0, mode0 = -1;
    for (int i = 0; i < n; ++i) {
        const CmfThinBwdCall &q = c[i];
        CMF_CHECK_ARG(q.rows > 0 && q.rows % 128 == 0 && q.cout == c[0].cout && q.cin == c[0].cin && q.cout % 32 == 0 && q.cin % 32 == 0 &&
                      q.cout <= 64 && q.cin <= 64 && (q.in_mode == 0 || q.in_mode == 1));
        const bool pooled = q.pool_g != nullptr;
        CMF_CHECK_ARG((pooled || q.dU) && q.a && q.w && q.x && q.dx && q.dw && q.slabs && (!q.sums || (q.z && q.mean && q.invstd)));
        CMF_CHECK_ARG((pooled || (q.lddu % 4 == 0 && (uintptr_t)q.dU % 16 == 0)) && (!q.sums || (q.ldz % 4 == 0 && (uintptr_t)q.z % 16 == 0)));
        if (pooled) CMF_CHECK_ARG(q.pool_am && q.pool_S > 0 && q.rows % q.pool_S == 0 && (uintptr_t)q.pool_g % 16 == 0 && (uintptr_t)q.pool_am % 4 == 0 &&
                                  q.in_mode == 1 && !q.dxyz && (q.rows / q.pool_S) * q.cout * 4 < (1ll << 32));
        CMF_CHECK_ARG(((uintptr_t)q.a | (uintptr_t)q.mean | (uintptr_t)q.invstd | (uintptr_t)q.sums) % 16 == 0);
        CMF_CHECK_ARG(q.in_mode == 0 || (q.a_in && q.c_in && q.mean_in && q.invstd_in && q.stats));
        CMF_CHECK_ARG(128ll * std::max({q.lddu, q.ldz, q.ldx, q.lddx}) * 4 < (1ll << 31));
        const int mode = (q.sums ? 1 : 0) | (q.in_mode ? 2 : 0) | ((q.in_mode && q.dxyz) ? 4 : 0) | (pooled ? 8 : 0);
        if (i == 0) mode0 = mode;
        CMF_CHECK_ARG(mode == mode0);
        ThinBwdArgs &p = b.a[i];
        p.rows = q.rows; p.cout = q.cout; p.cin = q.cin; p.dU = q.dU; p.lddu = q.lddu; p.z = q.z; p.ldz = q.ldz;
        p.a = q.a; p.mean = q.mean; p.invstd = q.invstd; p.sums = q.sums; p.inv_count = (float)(1.0 / (double)q.rows);
        p.w = q.w; p.ldw = q.ldw; p.x = q.x; p.ldx = q.ldx; p.in_mode = q.in_mode;
        p.a_in = q.a_in; p.c_in = q.c_in; p.mean_in = q.mean_in; p.invstd_in = q.invstd_in; p.dxyz = q.in_mode == 1 ? q.dxyz : nullptr;
        p.dx = q.dx; p.lddx = q.lddx; p.stats = q.stats; p.slabs = q.slabs;
        p.pool_g = q.pool_g; p.pool_am = q.pool_am; p.pool_S = q.pool_S;
        c[i].nslab = thin_bwd_slabs_for(q.rows, q.cin > 32 ? 512 : 768, &p.tiles_per_wg);
        max_slab = std::max(max_slab, c[i].nslab);
        cmf_gemm_count_flops(4.0 * q.rows * q.cout * q.cin);
    }
    const dim3 grid(max_slab, n), block(TG_THREADS);
    // the slot-level layers of the narrow blocks' bodies (train-mode BN): 64 <- 32 channels from the pooled gradient (mode 11) and
    // 32 <- 32 with the dxyz sums (mode 7); other switch combinations of those forms have no batched instantiation
    if (mode0 == 11 || mode0 == 7 || (mode0 & 12)) {
        if (mode0 == 11 && c[0].cout == 64 && c[0].cin == 32) hipLaunchKernelGGL((thin_bwd_layer_batch_kernel<2, 1, 11, true>), grid, block, 0, st, b);
        else if (mode0 == 7 && c[0].cout == 32 && c[0].cin == 32) hipLaunchKernelGGL((thin_bwd_layer_batch_kernel<1, 1, 7, true>), grid, block, 0, st, b);
        else return (int)hipErrorInvalidValue;
        return cmf_launch_status();
    }
#define CMF_TBB(NO, NI)                                                                                              \
    switch (mode0) {                                                                                                \
        case 0: hipLaunchKernelGGL((thin_bwd_layer_batch_kernel<NO, NI, 0, true>), grid, block, 0, st, b); break;   \
        case 1: hipLaunchKernelGGL((thin_bwd_layer_batch_kernel<NO, NI, 1, true>), grid, block, 0, st, b); break;   \
        case 2: hipLaunchKernelGGL((thin_bwd_layer_batch_kernel<NO, NI, 2, true>), grid, block, 0, st, b); break;   \
        default: hipLaunchKernelGGL((thin_bwd_layer_batch_kernel<NO, NI, 3, true>), grid, block, 0, st, b); break;  \
    }
    if (c[0].cout > 32) {
        if (c[0].cin > 32) { CMF_TBB(2, 2) } else { CMF_TBB(2, 1) }
    } else {
        if (c[0].cin > 32) { CMF_TBB(1, 2) } else { CMF_TBB(1, 1) }
    }
#undef CMF_TBB
    return cmf_launch_status();
}

extern "C" int cmf_thin_bwd_layer(long long rows, int cout, int cin, const float *dU, long long lddu, const float *z, long long ldz,
                                  const float *a, const float *mean, const float *invstd, const float *sums,
                                  const float *w, long long ldw, const float *x, long long ldx, int in_mode,
                                  const float *a_in, const float *c_in, const float *mean_in, const float *invstd_in, const float *dxyz,
                                  float *dx, long long lddx, float *stats, float *dw, long long lddw, int accumulate, float *slabs,
                                  void *stream)
{
    CMF_CHECK_ARG(dU);
    return thin_bwd_layer_impl(rows, cout, cin, dU, lddu, nullptr, nullptr, 0, z, ldz, a, mean, invstd, sums, w, ldw, x, ldx, in_mode,
                               a_in, c_in, mean_in, invstd_in, dxyz, dx, lddx, stats, dw, lddw, accumulate, slabs, stream);
}

extern "C" int cmf_thin_bwd_layer_pooled(long long P, int S, int cout, int cin, const float *g, const unsigned char *argmax,
                                         const float *z, const float *a, const float *mean, const float *invstd, const float *sums,
                                         const float *w, const float *x, const float *a_in, const float *c_in, const float *mean_in,
                                         const float *invstd_in, float *dx, float *stats, float *dw, int accumulate, float *slabs,
                                         void *stream)
{
    CMF_CHECK_ARG(P >= 0 && S > 0 && g && argmax);
    return thin_bwd_layer_impl(P * S, cout, cin, nullptr, cout, g, argmax, S, z, cout, a, mean, invstd, sums, w, cin, x, cin, 1,
                               a_in, c_in, mean_in, invstd_in, nullptr, dx, cin, stats, dw, cin, accumulate, slabs, stream);
}
