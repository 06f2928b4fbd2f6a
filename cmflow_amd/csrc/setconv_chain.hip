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

// MODE of the chain kernel
//   CH_INFER  folded BatchNorm: layers 1-3 + max over the ball -> out                         (inference)
//   CH_STATS2 layers 1-2: column (sum, sum of squares) of z2 per wave -> partial              (training, pass 2)
//   CH_STATS3 layers 1-3: the same of z3                                                       (training, pass 3)
//   CH_POOL   layers 1-3 + max over the ball -> out, the first maximum's slot (argmax) and its raw z3 (zsel)   (training, pass 4)
//   CH_BWD3   layers 1-3 recomputed; dz3 from the pooled gradient; dW3 slabs; dU2 = mask(dz3 W3) with its BN-backward sums
//   CH_BWD2   layers 1-2 recomputed; dz2 from dU2; dW2 slabs; dU1 = mask(dz2 W2) with its BN-backward and dxyz sums
enum { CH_INFER = 0, CH_STATS2 = 1, CH_STATS3 = 2, CH_POOL = 3, CH_BWD3 = 4, CH_BWD2 = 5 };

struct ChainArgs {
    long long M;                 // neighbour rows = B * N * S
    int N, S, lgS;               // points per sample, slots per point (4 / 8 / 16 / 32) and its log2
    int blocks_per_wave;
    const float *dxyz;           // (M, 4) relative coordinates of every slot (cmf_group_affine's by-product), or NULL: formed from xyz
    const int *idx;              // (M) source point of every slot, inside its sample
    const float *xyz;            // (B, N, 3)
    const float *y; long long ldy;      // (B * N, 32) per-point rows of the hoisted first conv
    const float *wx; long long ldwx;    // (32, 3) coordinate columns of the first conv
    const float *bn0, *bn1, *bn2;       // BatchNorm blocks of the three layers: mean | invstd | a | c  (4 x 32 | 4 x 32 | 4 x 64)
    const float *w2, *w3;        // (32, 32), (64, 32) dense
    float *out; long long ldo;   // (B * N, 64): max over the ball of relu(bn(z3))
    float *zsel; unsigned char *argmax;          // CH_POOL: (B * N, 64) each
    float *partial;              // statistics rows, one per WAVE: [2][32] (CH_STATS2, CH_BWD3), [2][64] (CH_STATS3), [5][32] (CH_BWD2)
    const float *g;              // CH_BWD3: (B * N, 64) pooled gradient, already masked by the ReLU at the argmax (cmf_maxpool_bwd_point)
    const float *sums;           // BN-backward sums (s1 | s2) of the layer being differentiated, or NULL (eval-mode BN: dZ = a dU)
    float inv_count;             // 1 / M
    const float *dU_in;          // CH_BWD2: dU2 (M, 32)
    float *dU_out;               // CH_BWD3: dU2, CH_BWD2: dU1 (M, 32)
    float *slabs;                // weight-gradient slabs, one per WORKGROUP: [64][32] (CH_BWD3), [32][32] (CH_BWD2)
};

constexpr int CH_THREADS = 256;
constexpr int CH_LDW = 33;       // row pitch of weight / transposition tiles in LDS (odd: a column walk touches every bank)

__device__ __forceinline__ float bnrelu(float a, float z, float c) { return fmaxf(fmaf(a, z, c), 0.f); }
// lane permutations inside a row of 16 lanes as DPP modifiers (no LDS crossbar): xor 1, xor 2, mirror inside 8, mirror inside 16
template <int CTRL>
__device__ __forceinline__ float ch_dpp(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// maximum over the S consecutive lanes of a point's rows (S = 4 / 8 / 16 / 32, groups aligned to S): every lane of the group gets it
__device__ __forceinline__ float ch_group_max(float v, int S)
{
    v = fmaxf(v, ch_dpp<0xB1>(v));
    v = fmaxf(v, ch_dpp<0x4E>(v));
    if (S >= 8) v = fmaxf(v, ch_dpp<0x141>(v));
    if (S >= 16) v = fmaxf(v, ch_dpp<0x140>(v));
    if (S >= 32) v = fmaxf(v, __shfl_xor(v, 16, 64));
    return v;
}
__device__ __forceinline__ int ch_of(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }     // channel of accumulator register r in lane half h

// LDS image (floats).  Constants: the three BN blocks as loaded (bn0 at 0, bn1 at 128, bn2 at 256: mean | invstd | a | c), the
// coordinate planes wx[d][32] at 512, the backward coefficients ko[4][64] at 608 (a | mean | invstd * s2 / M | s1 / M of the layer
// being differentiated).  Backward modes add the weights (pitch 33) and a transposition tile per wave.
constexpr int CH_BN0 = 0, CH_BN1 = 128, CH_BN2 = 256, CH_WX = 512, CH_KO = 608, CH_CONST = 864;
constexpr int CH_W2S = CH_CONST, CH_W3S = CH_W2S + 32 * CH_LDW, CH_TT = CH_W3S + 64 * CH_LDW;     // weights, then 4 x (96 x 33) tiles
constexpr int CH_TILE = 96 * CH_LDW;
constexpr int CH_LDS_FWD = CH_TT, CH_LDS_BWD = CH_TT + 4 * CH_TILE;

template <int MODE>
__device__ __forceinline__ void setconv_chain_body(const ChainArgs &p, const int bx)
{
    constexpr bool BWD = MODE == CH_BWD3 || MODE == CH_BWD2;
    constexpr bool L3 = MODE == CH_INFER || MODE == CH_STATS3 || MODE == CH_POOL || MODE == CH_BWD3;      // third layer evaluated
    extern __shared__ __attribute__((aligned(16))) float cst[];
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int S = p.S, lgS = p.lgS;
    for (int i = tid; i < CH_KO; i += CH_THREADS) {
        float v;
        if (i < 128) v = p.bn0[i]; else if (i < 256) v = p.bn1[i - 128]; else if (i < 512) v = p.bn2[i - 256];
        else { const int q = i - 512; v = p.wx[(long long)(q & 31) * p.ldwx + (q >> 5)]; }      // plane d of channel ch: wx[ch][d]
        cst[i] = v;
    }
    if (BWD) {
        // coefficients of dZ = a (dU - s1/M - (z - mean) invstd s2/M) for the layer being differentiated (thin_bwd_layer's form)
        const int C = MODE == CH_BWD3 ? 64 : 32;
        const float *bn = MODE == CH_BWD3 ? p.bn2 : p.bn1;
        for (int n = tid; n < C; n += CH_THREADS) {
            const bool train = p.sums != nullptr;
            cst[CH_KO + n] = bn[2 * C + n];
            cst[CH_KO + 64 + n] = train ? bn[n] : 0.f;
            cst[CH_KO + 128 + n] = train ? bn[C + n] * (p.sums[C + n] * p.inv_count) : 0.f;
            cst[CH_KO + 192 + n] = train ? p.sums[n] * p.inv_count : 0.f;
        }
    }
    // the weights live in LDS at an odd pitch (A operands are read per MFMA step: a register copy per wave costs 48 registers = a wave of occupancy)
    for (int i = tid; i < 32 * 32; i += CH_THREADS) cst[CH_W2S + (i >> 5) * CH_LDW + (i & 31)] = p.w2[i];
    if (L3)
        for (int i = tid; i < 64 * 32; i += CH_THREADS) cst[CH_W3S + (i >> 5) * CH_LDW + (i & 31)] = p.w3[i];
    __syncthreads();
    // per-lane statistics over the wave's blocks (lane = row of a block, register = channel): reduced over the lanes at the end
    constexpr int NST = MODE == CH_STATS2 ? 16 : MODE == CH_STATS3 ? 32 : BWD ? 16 : 1;
    float t1[NST], t2[NST], q0[MODE == CH_BWD2 ? 16 : 1], q1[MODE == CH_BWD2 ? 16 : 1], q2[MODE == CH_BWD2 ? 16 : 1];
#pragma unroll
    for (int r = 0; r < NST; ++r) t1[r] = t2[r] = 0.f;
    if (MODE == CH_BWD2) {
#pragma unroll
        for (int r = 0; r < 16; ++r) q0[MODE == CH_BWD2 ? r : 0] = q1[MODE == CH_BWD2 ? r : 0] = q2[MODE == CH_BWD2 ? r : 0] = 0.f;
    }
    // weight-gradient accumulators (C/D layout: lane (ci = j, h), register r <-> output channel c(r, h) of the block)
    f32x16 dwa[MODE == CH_BWD3 ? 2 : 1];
    if (BWD) {
#pragma unroll
        for (int ob = 0; ob < (MODE == CH_BWD3 ? 2 : 1); ++ob)
#pragma unroll
            for (int r = 0; r < 16; ++r) dwa[ob][r] = 0.f;
    }
    float *tt = cst + CH_TT + wv * CH_TILE;                           // this wave's transposition tile: [96 channels][33]

    const int wave = bx * (CH_THREADS / 64) + wv;               // (rows < 2^31: 32-bit index arithmetic, wave-uniform where it can be)
    const int nblocks = (int)(p.M / 32);
    const int blk0 = wave * p.blocks_per_wave;
    const int nb = max(0, min(p.blocks_per_wave, nblocks - blk0));
    const int rows_per_sample = p.N << lgS;
    // The gather of a block is a chain of two dependent loads (slot -> source point -> its rows): the source point of block t + 2 and the
    // rows of block t + 1 are requested while block t computes
    f32x4 yv[4], yn[4];
    f32x4 dv = {0.f, 0.f, 0.f, 0.f}, dn = dv;                           // (dx, dy, dz, 0) of this lane's row: block t, block t + 1
    int in1 = 0, in2 = 0;                                               // idx of blocks t + 1, t + 2 (this lane's row)
    auto rows_of = [&](int blk, int id, f32x4 (&yo)[4], f32x4 &dd) {
        const int m = blk * 32 + j, pt = m >> lgS;
        const int smp = __builtin_amdgcn_readfirstlane((blk * 32) / rows_per_sample);      // a block lies inside one sample (N S % 32 == 0)
        const long long src = (long long)smp * p.N + id;
        const float *yr = p.y + src * p.ldy + 4 * h;
#pragma unroll
        for (int g = 0; g < 4; ++g) yo[g] = *(const f32x4 *)(yr + 8 * g);
        if (p.dxyz) dd = *(const f32x4 *)(p.dxyz + (long long)m * 4);
        else {
            const float *xs = p.xyz + src * 3, *xc = p.xyz + (long long)pt * 3;
            dd[0] = xs[0] - xc[0]; dd[1] = xs[1] - xc[1]; dd[2] = xs[2] - xc[2];
        }
    };
    if (nb > 0) rows_of(blk0, p.idx[(long long)blk0 * 32 + j], yv, dv);
    if (nb > 1) in1 = p.idx[(long long)(blk0 + 1) * 32 + j];
    for (int t = 0; t < nb; ++t) {
        asm volatile("" ::: "memory");                                  // (the LDS-resident weights and constants are re-read per block, not hoisted into registers)
        const int blk = blk0 + t;
        const long long m = (long long)blk * 32 + j;                    // this lane's neighbour row
        const long long pt = m >> lgS;                                  // its centre point (global)
        const int slot = (int)m & (S - 1);
        if (t + 2 < nb) in2 = p.idx[(long long)(blk + 2) * 32 + j];
        if (t + 1 < nb) rows_of(blk + 1, in1, yn, dn);
        const float dx = dv[0], dy = dv[1], dz = dv[2];
        // ---- layer 1 (hoisted conv): z1 = y[src] + (wx . d), x1 = relu(bn0(z1)); channels c(r, h) of row j ----
        float z1[MODE == CH_BWD2 ? 16 : 1], x1[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 w0 = *(const f32x4 *)(cst + CH_WX + 8 * g + 4 * h), w1 = *(const f32x4 *)(cst + CH_WX + 32 + 8 * g + 4 * h),
                        wz = *(const f32x4 *)(cst + CH_WX + 64 + 8 * g + 4 * h);
            const f32x4 sa = *(const f32x4 *)(cst + CH_BN0 + 64 + 8 * g + 4 * h), sc = *(const f32x4 *)(cst + CH_BN0 + 96 + 8 * g + 4 * h);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float z = yv[g][q] + fmaf(wz[q], dz, fmaf(w1[q], dy, w0[q] * dx));       // group_affine_kernel's operations, in its order
                if (MODE == CH_BWD2) z1[MODE == CH_BWD2 ? 4 * g + q : 0] = z;
                x1[4 * g + q] = bnrelu(sa[q], z, sc[q]);
            }
        }
        if (MODE == CH_BWD2) {                                          // x1 is the B operand of dW2 at the end of the block: parked in the wave's tile now
#pragma unroll
            for (int r = 0; r < 16; ++r) tt[(64 + ch_of(r, h)) * CH_LDW + j] = x1[r];
        }
        // (the next block's rows take the place of this block's: they are not read below)
#pragma unroll
        for (int g = 0; g < 4; ++g) yv[g] = yn[g];
        dv = dn;
        in1 = in2;
        // ---- layer 2: z2^T = W2 . x1^T ----
        f32x16 acc2;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(cst[CH_W2S + j * CH_LDW + ch_of(r, h)], x1[r], acc2, 0, 0, 0);
        }
        if (MODE == CH_STATS2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { t1[r % NST] += acc2[r]; t2[r % NST] += acc2[r] * acc2[r]; }
            continue;
        }
        float x2[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 sa = *(const f32x4 *)(cst + CH_BN1 + 64 + 8 * g + 4 * h), sc = *(const f32x4 *)(cst + CH_BN1 + 96 + 8 * g + 4 * h);
#pragma unroll
            for (int q = 0; q < 4; ++q) x2[4 * g + q] = bnrelu(sa[q], acc2[4 * g + q], sc[q]);
        }
        if (MODE == CH_BWD3) {                                          // x2 is the B operand of dW3 at the end of the block
#pragma unroll
            for (int r = 0; r < 16; ++r) tt[(64 + ch_of(r, h)) * CH_LDW + j] = x2[r];
        }
        // ---- layer 3: z3^T = W3 . x2^T (two blocks of 32 output channels) ----
        f32x16 acc3[L3 ? 2 : 1];
        if (L3) {
#pragma unroll
            for (int ob = 0; ob < 2; ++ob) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc3[L3 ? ob : 0][r] = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    acc3[L3 ? ob : 0] = __builtin_amdgcn_mfma_f32_32x32x2f32(cst[CH_W3S + (32 * ob + j) * CH_LDW + ch_of(r, h)], x2[r], acc3[L3 ? ob : 0], 0, 0, 0);
                }
            }
        }
        if (MODE == CH_STATS3) {
#pragma unroll
            for (int ob = 0; ob < 2; ++ob)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = acc3[L3 ? ob : 0][r];
                    t1[(16 * ob + r) % NST] += v; t2[(16 * ob + r) % NST] += v * v;
                }
            continue;
        }
        if (MODE == CH_INFER || MODE == CH_POOL) {
            // relu(bn2), max over the S rows of a point (S consecutive lanes of a half); CH_POOL: the FIRST maximum's slot and its raw z3.
            // The point's first lane stores all three as 16-byte pieces.
            const int gbase = (h << 5) + (j & ~(S - 1));                // first lane of this row's point inside the wave
            const unsigned long long gmask = S == 32 ? 0xFFFFFFFFull : ((1ull << S) - 1ull);
            const bool leader = (j & (S - 1)) == 0;
#pragma unroll
            for (int ob = 0; ob < 2; ++ob)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 sa = *(const f32x4 *)(cst + CH_BN2 + 128 + 32 * ob + 8 * g + 4 * h), sc = *(const f32x4 *)(cst + CH_BN2 + 192 + 32 * ob + 8 * g + 4 * h);
                    f32x4 v, zs;
                    unsigned am = 0;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float z3 = acc3[L3 ? ob : 0][4 * g + q];
                        const float own = bnrelu(sa[q], z3, sc[q]);
                        const float y3 = ch_group_max(own, S);
                        v[q] = y3;
                        if (MODE == CH_POOL) {
                            const unsigned long long eq = __ballot(own == y3);
                            const int first = __builtin_ctzll((eq >> gbase) & gmask);          // (own == max for at least one lane of the group)
                            zs[q] = ch_group_max(slot == first ? z3 : -__builtin_inff(), S);   // exactly one lane of the group contributes
                            am |= (unsigned)first << (8 * q);
                        }
                    }
                    if (leader) {
                        const long long o = pt * 64 + 32 * ob + 8 * g + 4 * h;
                        *(f32x4 *)(p.out + pt * p.ldo + 32 * ob + 8 * g + 4 * h) = v;
                        if (MODE == CH_POOL) { *(f32x4 *)(p.zsel + o) = zs; *(unsigned *)(p.argmax + o) = am; }
                    }
                }
            continue;
        }
        if (MODE == CH_BWD3) {
            // ---- dz3 = a2 (du3 - s1/M - (z3 - mean) invstd s2/M), du3 = the pooled gradient at the argmax slot, 0 elsewhere ----
            f32x16 dz3[2];
#pragma unroll
            for (int ob = 0; ob < 2; ++ob)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c0 = 32 * ob + 8 * g + 4 * h;
                    const f32x4 gv = *(const f32x4 *)(p.g + pt * 64 + c0);
                    const unsigned am = *(const unsigned *)(p.argmax + pt * 64 + c0);
                    const f32x4 sa = *(const f32x4 *)(cst + CH_KO + c0), mu = *(const f32x4 *)(cst + CH_KO + 64 + c0),
                                u = *(const f32x4 *)(cst + CH_KO + 128 + c0), s1 = *(const f32x4 *)(cst + CH_KO + 192 + c0);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float d = ((int)((am >> (8 * q)) & 255u) == slot) ? gv[q] : 0.f;
                        const float z = acc3[L3 ? ob : 0][4 * g + q];
                        dz3[ob][4 * g + q] = p.sums ? sa[q] * (d - s1[q] - (z - mu[q]) * u[q]) : d * sa[q];
                    }
                }
            // ---- dx2^T = W3^T . dz3^T: step (ob, r) contracts the output-channel pair {32 ob + c(r, 0), 32 ob + c(r, 1)} ----
            f32x16 dx2;
#pragma unroll
            for (int r = 0; r < 16; ++r) dx2[r] = 0.f;
#pragma unroll
            for (int ob = 0; ob < 2; ++ob)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    dx2 = __builtin_amdgcn_mfma_f32_32x32x2f32(cst[CH_W3S + (32 * ob + ch_of(r, h)) * CH_LDW + j], dz3[ob][r], dx2, 0, 0, 0);
            // ---- dU2 = dx2 masked by relu(bn1(z2)), its BN-backward sums; stored for the next pass ----
            {
                float *drow = p.dU_out + m * 32 + 4 * h;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 sa = *(const f32x4 *)(cst + CH_BN1 + 64 + 8 * g + 4 * h), sc = *(const f32x4 *)(cst + CH_BN1 + 96 + 8 * g + 4 * h),
                                mu = *(const f32x4 *)(cst + CH_BN1 + 8 * g + 4 * h), is = *(const f32x4 *)(cst + CH_BN1 + 32 + 8 * g + 4 * h);
                    f32x4 o;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float z = acc2[4 * g + q];
                        const float x = fmaf(sa[q], z, sc[q]) > 0.f ? dx2[4 * g + q] : 0.f;
                        t1[(4 * g + q) % NST] += x; t2[(4 * g + q) % NST] += x * ((z - mu[q]) * is[q]);
                        o[q] = x;
                    }
                    *(f32x4 *)(drow + 8 * g) = o;
                }
            }
            // ---- dW3 += dz3^T x2 over the block's 32 rows: both operands through the wave's LDS tile (channel-major, rows along a line) ----
#pragma unroll
            for (int ob = 0; ob < 2; ++ob)
#pragma unroll
                for (int r = 0; r < 16; ++r) tt[(32 * ob + ch_of(r, h)) * CH_LDW + j] = dz3[ob][r];
            // (one wave: LDS operations execute in order, the reads below see the writes above)
#pragma unroll
            for (int s = 0; s < 16; ++s) {                               // step s contracts rows {2 s, 2 s + 1}
                const float b = tt[(64 + j) * CH_LDW + 2 * s + h];       // B[k = row][j = input channel]
#pragma unroll
                for (int ob = 0; ob < 2; ++ob)
                    dwa[MODE == CH_BWD3 ? ob : 0] = __builtin_amdgcn_mfma_f32_32x32x2f32(tt[(32 * ob + j) * CH_LDW + 2 * s + h], b, dwa[MODE == CH_BWD3 ? ob : 0], 0, 0, 0);
            }
            continue;
        }
        if (MODE == CH_BWD2) {
            // ---- dz2 = a1 (dU2 - s1/M - (z2 - mean) invstd s2/M) ----
            float dz2[16];
            const float *urow = p.dU_in + m * 32 + 4 * h;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c0 = 8 * g + 4 * h;
                const f32x4 dv = *(const f32x4 *)(urow + 8 * g);
                const f32x4 sa = *(const f32x4 *)(cst + CH_KO + c0), mu = *(const f32x4 *)(cst + CH_KO + 64 + c0),
                            u = *(const f32x4 *)(cst + CH_KO + 128 + c0), s1 = *(const f32x4 *)(cst + CH_KO + 192 + c0);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float z = acc2[4 * g + q];
                    dz2[4 * g + q] = p.sums ? sa[q] * (dv[q] - s1[q] - (z - mu[q]) * u[q]) : dv[q] * sa[q];
                }
            }
            // ---- dx1^T = W2^T . dz2^T ----
            f32x16 dx1;
#pragma unroll
            for (int r = 0; r < 16; ++r) dx1[r] = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                dx1 = __builtin_amdgcn_mfma_f32_32x32x2f32(cst[CH_W2S + ch_of(r, h) * CH_LDW + j], dz2[r], dx1, 0, 0, 0);
            // ---- dU1 = dx1 masked by relu(bn0(z1)), BN-backward sums and the three dxyz sums (the xyz-weight gradient) ----
            {
                float *drow = p.dU_out + m * 32 + 4 * h;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 sa = *(const f32x4 *)(cst + CH_BN0 + 64 + 8 * g + 4 * h), sc = *(const f32x4 *)(cst + CH_BN0 + 96 + 8 * g + 4 * h),
                                mu = *(const f32x4 *)(cst + CH_BN0 + 8 * g + 4 * h), is = *(const f32x4 *)(cst + CH_BN0 + 32 + 8 * g + 4 * h);
                    f32x4 o;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int r = 4 * g + q;
                        const float z = z1[MODE == CH_BWD2 ? r : 0];
                        const float x = fmaf(sa[q], z, sc[q]) > 0.f ? dx1[r] : 0.f;
                        t1[r % NST] += x; t2[r % NST] += x * ((z - mu[q]) * is[q]);
                        q0[MODE == CH_BWD2 ? r : 0] += x * dx; q1[MODE == CH_BWD2 ? r : 0] += x * dy; q2[MODE == CH_BWD2 ? r : 0] += x * dz;
                        o[q] = x;
                    }
                    *(f32x4 *)(drow + 8 * g) = o;
                }
            }
            // ---- dW2 += dz2^T x1 ----
#pragma unroll
            for (int r = 0; r < 16; ++r) tt[ch_of(r, h) * CH_LDW + j] = dz2[r];
#pragma unroll
            for (int s = 0; s < 16; ++s)
                dwa[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(tt[j * CH_LDW + 2 * s + h], tt[(64 + j) * CH_LDW + 2 * s + h], dwa[0], 0, 0, 0);
            continue;
        }
    }
    // ---- statistics: the lanes of a half hold the 32 rows of every block -> sum over them (fixed tree), one row per wave ----
    if (MODE == CH_STATS2 || MODE == CH_STATS3 || BWD) {
        constexpr int C = MODE == CH_STATS3 ? 64 : 32;
        constexpr int NROW = MODE == CH_BWD2 ? 5 : 2;
        float *prow = p.partial + wave * NROW * C;
        auto fold = [&](float v) {
#pragma unroll
            for (int off = 1; off < 32; off <<= 1) v += __shfl_xor(v, off, 64);
            return v;
        };
#pragma unroll
        for (int r = 0; r < NST; ++r) {
            const int c = (r >> 4) * 32 + ch_of(r & 15, h);
            const float a = fold(t1[r]), b = fold(t2[r]);
            if (j == 0) { prow[c] = a; prow[C + c] = b; }
            if (MODE == CH_BWD2) {
                const float e0 = fold(q0[MODE == CH_BWD2 ? r : 0]), e1 = fold(q1[MODE == CH_BWD2 ? r : 0]), e2 = fold(q2[MODE == CH_BWD2 ? r : 0]);
                if (j == 0) { prow[2 * C + c] = e0; prow[3 * C + c] = e1; prow[4 * C + c] = e2; }
            }
        }
    }
    // ---- weight-gradient slab of the workgroup: the four waves' accumulators summed in wave order through LDS ----
    if (BWD) {
        constexpr int NOB = MODE == CH_BWD3 ? 2 : 1;
        __syncthreads();                                                 // every wave is done with its tile and the weights
        float *red = cst + CH_W2S;                                       // [4 waves][NOB * 32 * 32]
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[wv * NOB * 1024 + (32 * ob + ch_of(r, h)) * 32 + j] = dwa[ob][r];
        __syncthreads();
        float *slab = p.slabs + (long long)bx * NOB * 1024;
        for (int i = tid; i < NOB * 1024; i += CH_THREADS) slab[i] = ((red[i] + red[NOB * 1024 + i]) + red[2 * NOB * 1024 + i]) + red[3 * NOB * 1024 + i];
    }
}

// blocks of 32 rows a wave walks: >= 4 (one statistics row per wave must fit the caller's one-row-per-128-rows scratch; the weight
// fragments and the slab of a wave are amortised), ~2-3 waves per SIMD at the largest scale
int chain_blocks_per_wave(long long blocks, bool bwd)
{
    const long long target = bwd ? 2048 : 3072;
    return (int)std::max<long long>(4, std::min<long long>(bwd ? 32 : 16, (blocks + target - 1) / target));
}

template <int MODE>
__global__ __launch_bounds__(CH_THREADS, 2) void setconv_chain_kernel(const ChainArgs p)
{
    setconv_chain_body<MODE>(p, blockIdx.x);
}

// inference of up to CMF_MAX_BATCH blocks in one launch (cmf_common.h "batched launches"): blockIdx.y = block
struct ChainBatch { ChainArgs a[CMF_MAX_BATCH]; int grid[CMF_MAX_BATCH]; };
__global__ __launch_bounds__(CH_THREADS, 2) void setconv_chain_infer_batch_kernel(const ChainBatch b)
{
    if ((int)blockIdx.x >= b.grid[blockIdx.y]) return;
    setconv_chain_body<CH_INFER>(b.a[blockIdx.y], blockIdx.x);
}

template <int MODE>
int chain_launch(ChainArgs &a, void *stream)
{
    const long long blocks = a.M / 32;
    constexpr bool BWD = MODE == CH_BWD3 || MODE == CH_BWD2;
    // ~3 waves per SIMD of work at the largest scale; a wave's weight fragments are amortised over its blocks
    a.blocks_per_wave = chain_blocks_per_wave(blocks, BWD);
    const long long waves = (blocks + a.blocks_per_wave - 1) / a.blocks_per_wave;
    const unsigned grid = (unsigned)((waves + CH_THREADS / 64 - 1) / (CH_THREADS / 64));
    const size_t lds = (size_t)(BWD ? CH_LDS_BWD : CH_LDS_FWD) * sizeof(float);
    if (BWD) {
        static CmfPerDevice once;
        int dev;
        if (once.need(dev)) {
            if (hipFuncSetAttribute((const void *)setconv_chain_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
                return (int)hipGetLastError();
            once.done(dev);
        }
    }
    hipLaunchKernelGGL(setconv_chain_kernel<MODE>, dim3(grid), dim3(CH_THREADS), lds, (hipStream_t)stream, a);
    return cmf_launch_status();
}

}  // namespace

// internal (setconv_block.hip): the inference chains of n blocks (cmf_setconv_chain_infer's arguments per block) in ONE launch
int cmf_setconv_chain_infer_batch(int n, const CmfChainInferArgs *q, void *stream)
{
    CMF_CHECK_ARG(n >= 1 && n <= CMF_MAX_BATCH && q);
    ChainBatch b;
    int gmax = 0;
    for (int i = 0; i < n; ++i) {
        const CmfChainInferArgs &c = q[i];
        CMF_CHECK_ARG(c.M > 0 && c.M < (1ll << 31) && c.idx && c.xyz && c.y && c.wx && c.bn0 && c.bn1 && c.bn2 && c.w2 && c.w3 && c.ldy % 4 == 0 &&
                      ((uintptr_t)c.y & 15) == 0 && c.out && c.ldo % 4 == 0 && ((uintptr_t)c.out & 15) == 0);
        ChainArgs &a = b.a[i];
        a = ChainArgs{};
        a.M = c.M; a.N = c.N; a.S = c.S; a.lgS = c.S == 4 ? 2 : c.S == 8 ? 3 : c.S == 16 ? 4 : 5; a.idx = c.idx; a.xyz = c.xyz; a.y = c.y; a.ldy = c.ldy;
        a.wx = c.wx; a.ldwx = c.ldwx; a.bn0 = c.bn0; a.bn1 = c.bn1; a.bn2 = c.bn2; a.w2 = c.w2; a.w3 = c.w3; a.out = c.out; a.ldo = c.ldo;
        a.inv_count = (float)(1.0 / (double)c.M);
        const long long blocks = a.M / 32;
        a.blocks_per_wave = chain_blocks_per_wave(blocks, false);
        const long long waves = (blocks + a.blocks_per_wave - 1) / a.blocks_per_wave;
        b.grid[i] = (int)((waves + CH_THREADS / 64 - 1) / (CH_THREADS / 64));
        gmax = std::max(gmax, b.grid[i]);
    }
    hipLaunchKernelGGL(setconv_chain_infer_batch_kernel, dim3(gmax, n), dim3(CH_THREADS), (size_t)CH_LDS_FWD * sizeof(float), (hipStream_t)stream, b);
    return cmf_launch_status();
}

// internal (setconv_block.hip): the block's neighbour-slot layers as chain passes, when the shape is the chain's
bool cmf_setconv_chain_supported(int N, int S, int O1, int C2, int C3, long long M)
{
    static const bool on = !(getenv("CMF_CHAIN") && getenv("CMF_CHAIN")[0] == '0');
    return on && O1 == 32 && C2 == 32 && C3 == 64 && (S == 4 || S == 8 || S == 16 || S == 32) && M % 32 == 0 && ((long long)N * S) % 32 == 0;
}
// rows of `partial` / slabs a pass writes (the caller sizes scratch with the maxima)
long long cmf_setconv_chain_waves(long long M, int backward)
{
    const long long blocks = M / 32;
    const long long bpw = chain_blocks_per_wave(blocks, backward != 0);
    const long long waves = (blocks + bpw - 1) / bpw;
    return (waves + 3) / 4 * 4;
}

// mode: CH_* above.  Pointers a mode does not use may be NULL.
int cmf_setconv_chain_pass(int mode, long long M, int N, int S, const int *idx, const float *xyz, const float *dxyz, const float *y, long long ldy, const float *wx,
                           long long ldwx, const float *bn0, const float *bn1, const float *bn2, const float *w2, const float *w3, float *out,
                           long long ldo, float *zsel, unsigned char *argmax, float *partial, const float *g, const float *sums,
                           const float *dU_in, float *dU_out, float *slabs, void *stream)
{
    CMF_CHECK_ARG(M > 0 && M < (1ll << 31) && idx && xyz && y && wx && bn0 && bn1 && bn2 && w2 && w3 && ldy % 4 == 0 && ((uintptr_t)y & 15) == 0);
    CMF_CHECK_ARG(!dxyz || ((uintptr_t)dxyz & 15) == 0);
    ChainArgs a{};
    a.M = M; a.N = N; a.S = S; a.lgS = S == 4 ? 2 : S == 8 ? 3 : S == 16 ? 4 : 5; a.dxyz = dxyz; a.idx = idx; a.xyz = xyz; a.y = y; a.ldy = ldy; a.wx = wx; a.ldwx = ldwx;
    a.bn0 = bn0; a.bn1 = bn1; a.bn2 = bn2; a.w2 = w2; a.w3 = w3; a.out = out; a.ldo = ldo; a.zsel = zsel; a.argmax = argmax;
    a.partial = partial; a.g = g; a.sums = sums; a.inv_count = (float)(1.0 / (double)M); a.dU_in = dU_in; a.dU_out = dU_out; a.slabs = slabs;
    switch (mode) {
    case CH_INFER:  CMF_CHECK_ARG(out && ldo % 4 == 0 && ((uintptr_t)out & 15) == 0); return chain_launch<CH_INFER>(a, stream);
    case CH_STATS2: CMF_CHECK_ARG(partial); return chain_launch<CH_STATS2>(a, stream);
    case CH_STATS3: CMF_CHECK_ARG(partial); return chain_launch<CH_STATS3>(a, stream);
    case CH_POOL:   CMF_CHECK_ARG(out && zsel && argmax && ldo % 4 == 0 && ((uintptr_t)out & 15) == 0); return chain_launch<CH_POOL>(a, stream);
    case CH_BWD3:   CMF_CHECK_ARG(g && argmax && dU_out && partial && slabs && (((uintptr_t)g | (uintptr_t)dU_out | (uintptr_t)argmax) & 15) == 0); return chain_launch<CH_BWD3>(a, stream);
    case CH_BWD2:   CMF_CHECK_ARG(dU_in && dU_out && partial && slabs && (((uintptr_t)dU_in | (uintptr_t)dU_out) & 15) == 0); return chain_launch<CH_BWD2>(a, stream);
    default: return (int)hipErrorInvalidValue;
    }
}

int cmf_setconv_chain_infer(long long M, int N, int S, const int *idx, const float *xyz, const float *y, long long ldy, const float *wx,
                            long long ldwx, const float *bn0, const float *bn1, const float *bn2, const float *w2, const float *w3, float *out,
                            long long ldo, void *stream)
{
    return cmf_setconv_chain_pass(0, M, N, S, idx, xyz, nullptr, y, ldy, wx, ldwx, bn0, bn1, bn2, w2, w3, out, ldo, nullptr, nullptr, nullptr, nullptr,
                                  nullptr, nullptr, nullptr, nullptr, stream);
}
