// The narrow set-conv block (first encoder: 6 -> 32 -> 32 -> 64 channels per neighbour slot, radarflow_util.py:144-155 with the
// channels of models/cmflow.py:23) as a REGISTER CHAIN: a wave forms 32 neighbour rows of the first layer from the gathered
// per-point rows, runs the two 1x1 convolutions on the matrix cores and reduces the ball's maximum -- no activation goes to
// memory between the layers (the per-layer kernels wrote and re-read [rows, 32 / 32 / 64] tensors: 1 KB per row of the
// forward pass for 24 bytes of input).
//
// Chaining without a transposition: the layers are evaluated TRANSPOSED, Z^T[c_out][row] = W[c_out][c_in] . X^T[c_in][row],
// with v_mfma_f32_32x32x2_f32's A = weights (i = c_out, k = c_in) and B = activations (k = c_in, j = row).  The result tile
// (C/D layout: lane (j = l & 31, h = l >> 5), register r <-> channel c(r, h) = (r & 3) + 8 (r >> 2) + 4 h of row j) is, register by
// register, already the B operand of the next layer's MFMA steps: step r contracts the channel pair {c(r, 0), c(r, 1)} when lane
// (j, h) supplies its register r, and the weights are pre-loaded in the matching order, A_r[lane (i, h)] = W[i][c(r, h)].
// That pair order -- {0,4} {1,5} {2,6} {3,7} {8,12} ... -- is exactly the order in which thin_fwd / cmf_gemm consume k (a lane's
// 16-byte fragment feeds four steps), so every layer output is bit-identical to the per-layer kernels' (tests/test_gpu_gemm.py).
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include "cmf_common.h"
#include "../../include/cmflow_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

struct ChainArgs {
    long long M;                 // neighbour rows = B * N * S
    int N, S;                    // points per sample, slots per point
    int blocks_per_wave;
    const int *idx;              // (M) source point of every slot, inside its sample
    const float *xyz;            // (B, N, 3)
    const float *y; long long ldy;      // (B * N, 32) per-point rows of the hoisted first conv
    const float *wx; long long ldwx;    // (32, 3) coordinate columns of the first conv
    const float *a0, *c0, *a1, *c1, *a2, *c2;     // folded BatchNorm of the three layers (32 | 32 | 64)
    const float *w2, *w3;        // (32, 32), (64, 32) dense
    float *out; long long ldo;   // (B * N, 64): max over the ball of relu(bn(z3))
};

constexpr int CH_THREADS = 256;
// LDS constants: a0 c0 a1 c1 (32 each) | a2 c2 (64 each) | wx planes (3 x 32)
constexpr int CH_A0 = 0, CH_C0 = 32, CH_A1 = 64, CH_C1 = 96, CH_A2 = 128, CH_C2 = 192, CH_WX = 256, CH_CONST = 352;

__device__ __forceinline__ float bnrelu(float a, float z, float c) { return fmaxf(fmaf(a, z, c), 0.f); }

template <int S>
__global__ __launch_bounds__(CH_THREADS, 2) void setconv_chain_infer_kernel(const ChainArgs p)
{
    __shared__ __attribute__((aligned(16))) float cst[CH_CONST];
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
    for (int i = tid; i < CH_CONST; i += CH_THREADS) {
        float v;
        if (i < 32) v = p.a0[i]; else if (i < 64) v = p.c0[i - 32]; else if (i < 96) v = p.a1[i - 64]; else if (i < 128) v = p.c1[i - 96];
        else if (i < 192) v = p.a2[i - 128]; else if (i < 256) v = p.c2[i - 192];
        else { const int q = i - 256; v = p.wx[(long long)(q & 31) * p.ldwx + (q >> 5)]; }      // plane d of channel ch: wx[ch][d]
        cst[i] = v;
    }
    // weights in step order: A_r[lane (i, h)] = W[i][c(r, h)]
    float w2p[16], w3p[2][16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int ch = (r & 3) + 8 * (r >> 2) + 4 * h;
        w2p[r] = p.w2[j * 32 + ch];
        w3p[0][r] = p.w3[j * 32 + ch];
        w3p[1][r] = p.w3[(32 + j) * 32 + ch];
    }
    __syncthreads();
    const long long wave = (long long)blockIdx.x * (CH_THREADS / 64) + (tid >> 6);
    const long long nblocks = p.M / 32;
    for (int t = 0; t < p.blocks_per_wave; ++t) {
        const long long blk = wave * p.blocks_per_wave + t;
        if (blk >= nblocks) break;
        const long long m = blk * 32 + j;                               // this lane's neighbour row
        const long long pt = m / S;                                     // its centre point (global)
        const long long smp = pt / p.N;                                 // its sample
        const long long src = smp * p.N + p.idx[m];
        const float *xs = p.xyz + src * 3, *xc = p.xyz + pt * 3;
        const float dx = xs[0] - xc[0], dy = xs[1] - xc[1], dz = xs[2] - xc[2];
        // ---- layer 1 (hoisted conv): z1 = y[src] + (wx . d), x1 = relu(bn0(z1)); channels c(r, h) of row j ----
        float x1[16];
        const float *yr = p.y + src * p.ldy + 4 * h;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 yv = *(const f32x4 *)(yr + 8 * g);
            const f32x4 w0 = *(const f32x4 *)(cst + CH_WX + 8 * g + 4 * h), w1 = *(const f32x4 *)(cst + CH_WX + 32 + 8 * g + 4 * h),
                        w2 = *(const f32x4 *)(cst + CH_WX + 64 + 8 * g + 4 * h);
            const f32x4 sa = *(const f32x4 *)(cst + CH_A0 + 8 * g + 4 * h), sc = *(const f32x4 *)(cst + CH_C0 + 8 * g + 4 * h);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float z = yv[q] + fmaf(w2[q], dz, fmaf(w1[q], dy, w0[q] * dx));          // group_affine_kernel's operations, in its order
                x1[4 * g + q] = bnrelu(sa[q], z, sc[q]);
            }
        }
        // ---- layer 2: z2^T = W2 . x1^T ----
        f32x16 acc2;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(w2p[r], x1[r], acc2, 0, 0, 0);
        float x2[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 sa = *(const f32x4 *)(cst + CH_A1 + 8 * g + 4 * h), sc = *(const f32x4 *)(cst + CH_C1 + 8 * g + 4 * h);
#pragma unroll
            for (int q = 0; q < 4; ++q) x2[4 * g + q] = bnrelu(sa[q], acc2[4 * g + q], sc[q]);
        }
        // ---- layer 3: z3^T = W3 . x2^T (two blocks of 32 output channels), relu(bn2), max over the S rows of a point ----
        f32x16 acc3[2];
#pragma unroll
        for (int ob = 0; ob < 2; ++ob) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc3[ob][r] = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc3[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(w3p[ob][r], x2[r], acc3[ob], 0, 0, 0);
        }
#pragma unroll
        for (int ob = 0; ob < 2; ++ob)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 sa = *(const f32x4 *)(cst + CH_A2 + 32 * ob + 8 * g + 4 * h), sc = *(const f32x4 *)(cst + CH_C2 + 32 * ob + 8 * g + 4 * h);
                f32x4 v;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float y3 = bnrelu(sa[q], acc3[ob][4 * g + q], sc[q]);
#pragma unroll
                    for (int off = 1; off < S; off <<= 1) y3 = fmaxf(y3, __shfl_xor(y3, off, 64));     // rows of one point: S consecutive lanes of a half
                    v[q] = y3;
                }
                if ((j & (S - 1)) == 0) *(f32x4 *)(p.out + pt * p.ldo + 32 * ob + 8 * g + 4 * h) = v;
            }
    }
}

}  // namespace

// internal (setconv_block.hip): the inference form of the block up to the max over the ball, when the shape is the chain's
bool cmf_setconv_chain_supported(int N, int S, int O1, int C2, int C3, long long M)
{
    static const bool on = !(getenv("CMF_CHAIN") && getenv("CMF_CHAIN")[0] == '0');
    return on && O1 == 32 && C2 == 32 && C3 == 64 && (S == 4 || S == 8 || S == 16 || S == 32) && M % 32 == 0 && ((long long)N * S) % 32 == 0;
}

int cmf_setconv_chain_infer(long long M, int N, int S, const int *idx, const float *xyz, const float *y, long long ldy, const float *wx,
                            long long ldwx, const float *bn0, const float *bn1, const float *bn2, const float *w2, const float *w3, float *out,
                            long long ldo, void *stream)
{
    CMF_CHECK_ARG(M > 0 && idx && xyz && y && wx && bn0 && bn1 && bn2 && w2 && w3 && out && ldy % 4 == 0 && ldo % 4 == 0);
    CMF_CHECK_ARG((((uintptr_t)y | (uintptr_t)out) & 15) == 0);
    ChainArgs a;
    a.M = M; a.N = N; a.S = S; a.idx = idx; a.xyz = xyz; a.y = y; a.ldy = ldy; a.wx = wx; a.ldwx = ldwx;
    a.a0 = bn0 + 2 * 32; a.c0 = bn0 + 3 * 32; a.a1 = bn1 + 2 * 32; a.c1 = bn1 + 3 * 32; a.a2 = bn2 + 2 * 64; a.c2 = bn2 + 3 * 64;
    a.w2 = w2; a.w3 = w3; a.out = out; a.ldo = ldo;
    const long long blocks = M / 32;
    // ~3 waves per SIMD of work at the largest scale; a wave's weight fragments (12 KB from L2) are amortised over its blocks
    a.blocks_per_wave = (int)std::max<long long>(1, std::min<long long>(16, blocks / 3072));
    const long long waves = (blocks + a.blocks_per_wave - 1) / a.blocks_per_wave;
    const unsigned grid = (unsigned)((waves + CH_THREADS / 64 - 1) / (CH_THREADS / 64));
    hipStream_t st = (hipStream_t)stream;
    switch (S) {
    case 4:  hipLaunchKernelGGL(setconv_chain_infer_kernel<4>, dim3(grid), dim3(CH_THREADS), 0, st, a); break;
    case 8:  hipLaunchKernelGGL(setconv_chain_infer_kernel<8>, dim3(grid), dim3(CH_THREADS), 0, st, a); break;
    case 16: hipLaunchKernelGGL(setconv_chain_infer_kernel<16>, dim3(grid), dim3(CH_THREADS), 0, st, a); break;
    case 32: hipLaunchKernelGGL(setconv_chain_infer_kernel<32>, dim3(grid), dim3(CH_THREADS), 0, st, a); break;
    default: return (int)hipErrorInvalidValue;
    }
    return cmf_launch_status();
}
