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
__device__ __forceinline__ void thin_epilogue(const GemmArgs &p, f32x16 (&acc)[NT], int m0, int lane, int wave, float *red)
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
                } else if (want_stats) { s1 += x; s2 += x * x; }
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
            p.stats[((long long)blockIdx.x * nstat + which) * p.N + n] = s;
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

// C = epi(pro(A) @ W^T): K = 8*KS <= 64, N <= 32*NT
template <int KS, int NT>
__global__ __launch_bounds__(TG_THREADS) void thin_fwd_kernel(const GemmArgs p)
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
    thin_epilogue<NT>(p, acc, m0, lane, wave, red);
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
    thin_epilogue<NT>(p, acc, m0, lane, wave, red);
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

// returns -1 when the shape is not a thin one (the caller falls back to the tiled kernel)
int cmf_thin_gemm(const GemmArgs &g, int a_t, int b_t, hipStream_t st)
{
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
