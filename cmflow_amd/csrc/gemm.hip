// fp32 MFMA GEMM for the point-major CMFlow path (every 1x1 conv of the reference is a GEMM on
// [positions, channels] matrices; utils/model_utils/radarflow_util.py:151-159,215-221,253-285).
//
//     C[M,N] = epi( pro(A)[M,K] * B[K,N] )
//
// * v_mfma_f32_32x32x2_f32: exact fp32 (bit-equal to an fmaf chain), 64 FLOP/clk/SIMD = the chip's
//   157 TF fp32 peak (no TF32/xf32 on gfx950; bf16 would break the 1e-4 parity bound).
// * 256 threads = 4 wavefronts; block tile BM x BN x 16, 3 workgroups per CU.  Interior tiles stage operands with
//   LDS-direct loads (global_load_lds_dwordx4: no VGPR round trip; 3 stages, prefetch distance 2, swizzled
//   unpadded LDS image); edge tiles / ragged K go global -> registers -> LDS (double buffered, rows padded to
//   BK+4 floats).  One s_barrier per K-chunk either way.
// * K-permutation trick: a lane reads 4 consecutive k of its row with ONE ds_read_b128 (lanes
//   0-31 take k0..k0+3, lanes 32-63 take k0+4..k0+7) and feeds them to 4 MFMAs; step t of the 4
//   contracts k in {k0+t, k0+4+t}.  A and B use the same map, so the sum is just reordered.
// * prologue (fused BN+ReLU of the producer layer): A'[m,k] = relu(pa[k]*A[m,k] + pc[k]).
// * epilogues: +bias[n]; relu / leaky(0.1) / sigmoid; per-column (sum, sum of squares) partials
//   per row tile for train-mode BatchNorm; backward masks (see cmf_gemm docs in cmflow_hip.h).
// * XCD-aware tile order: the column tiles that share an A row-panel run back to back on ONE XCD
//   (blockIdx -> XCD is round-robin), so the panel is fetched from HBM once and re-read from L2.
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include "cmf_common.h"
#include "../../include/cmflow_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int G_THREADS = 256;
// Fixed design choices, each measured against its alternative in an earlier round (DESIGN.md "ruled out"):
//  * kinds 0 / 1 (plain store, forward) store straight from the accumulator layout (direct_epilogue): 6.7 vs 10.4 us per tile;
//  * the backward kinds (2-5) of the 128 x 128 kernels finish their tiles per wave (wave_epilogue);
//  * the chunk barrier sits in the middle of a chunk's MFMAs, the first fragment reads of the next chunk run under the second half;
//  * s_setprio 1 for the main loop, 0 for the epilogue.
#ifndef CMF_GEMM_BK
#define CMF_GEMM_BK 16
#endif
constexpr int G_BK = CMF_GEMM_BK;
constexpr int G_KT = G_BK / 4;            // threads covering one k-contiguous row of a chunk
constexpr int G_RP = 256 / G_KT;          // rows staged per pass by the 256 threads
constexpr int G_LDS_LD = G_BK + 4;      // 20 floats (BK=16) / 36 (BK=32): row strides whose 16-B slots tile all 64 banks -> conflict-free b128 reads

#include "gemm_args.h"
// timing diagnostics (GemmArgs::diag) are compiled in only with -DCMF_GEMM_DIAG (tools/diag builds): in the shipped kernels the
// main loop carries no run-time switches
#ifdef CMF_GEMM_DIAG
#define GDIAG(p) ((p).diag)
#else
#define GDIAG(p) 0
#endif

__device__ __forceinline__ float act_fn(float v, int act)
{
    if (act == 1) return v > 0.f ? v : 0.f;
    if (act == 2) return v > 0.f ? v : 0.1f * v;
    if (act == 3) return 1.0f / (1.0f + __expf(-v));
    return v;
}

// ---- LDS-direct staging (interior tiles, K a multiple of the chunk) ------------------------------------------
// global_load_lds_dwordx4 (gfx950) moves 16 bytes per lane straight from global memory into LDS at
// M0-base + lane*16: no VGPR round trip, no ds_write, and -- the point -- no s_waitcnt vmcnt in the middle of the
// MFMA stream (measured: the register-staged loop loses ~20 % of the MFMA rate to exactly that; the same loop with
// the staging removed runs 120-135 TF).  Three stages, prefetch distance two.  The LDS image is dictated by the
// instruction (consecutive lanes -> consecutive 16-byte slots), so rows are unpadded and bank conflicts are
// avoided by choosing WHICH global 16 bytes a lane fetches: slot(row, kq) = row*4 + (kq ^ swz(row)).  swz makes
// the 8 (32-bank model) and the 16 (64-bank model) lanes that ds_read_b128 serves together hit distinct slots.
constexpr int G_STAGES = 3;
__device__ __forceinline__ int g_swz(int row) { return (((row >> 2) & 1) << 1) | (((row >> 1) & 1) ^ ((row >> 3) & 1)); }
#define CMF_WAIT_VMCNT(n) __builtin_amdgcn_s_waitcnt(((n) & 15) | (((n) >> 4) << 14) | 0x0F70)
typedef const __attribute__((address_space(1))) void *g_gptr;
typedef __attribute__((address_space(3))) void *g_lptr;
typedef float f32x4 __attribute__((ext_vector_type(4)));
// Fragment reads of the direct loop are issued as inline asm: the compiler's waitcnt insertion treats every LDS
// read as possibly aliasing the in-flight LDS-direct loads and would put s_waitcnt vmcnt(0) in front of it
// (prefetch distance zero).  Which stage is complete is known here (wait_prev + barrier), so the reads are
// invisible to that pass and their own completion is awaited explicitly (g_lds_wait + g_pin).
__device__ __forceinline__ unsigned g_lds_addr(const float *p) { return (unsigned)(size_t)(__attribute__((address_space(3))) const float *)p; }
__device__ __forceinline__ f32x4 g_lds_read128(unsigned addr)
{
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
template <int OFF>
__device__ __forceinline__ float g_lds_read32(unsigned addr)
{
    float v;
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
__device__ __forceinline__ void g_lds_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void g_pin(f32x4 &v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void g_keep(const f32x16 &v) { asm volatile("" :: "v"(v)); }
__device__ __forceinline__ unsigned long long g_where()         // HW_REG_XCC_ID[3:0] << 32 | HW_REG_HW_ID
{
    return ((unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) << 32) |
           (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
}

// A_T: A stored [K][M] (contraction-major) instead of [M][K].  B_T: B stored [N][K] (i.e. W[out][in],
// the forward layout) instead of [K][N].
// 0 raw store (split-K slabs, plain GEMM); 1 forward (bias / none-ReLU-leaky activation / BN statistics); 2 backward
// through BN + ReLU; 3 backward through (leaky) ReLU; 4 / 5 = 2 / 3 with the three dxyz column sums
__host__ __device__ __forceinline__ int epilogue_kind(const GemmArgs &p)
{
    if (p.split_k > 1) return 0;
    const int q = (p.stats && p.dxyz) ? 2 : 0;
    if (p.bwd_mode == 1) return 2 + q;
    if (p.bwd_mode) return 3 + q;
    return (p.bias || p.act || p.stats) ? 1 : 0;
}

// EPI: the epilogue kind compiled into the fast path of this instantiation (the host picks the kernel by the call's
// kind, epilogue_kind()); tiles the fast path does not take (edges, unaligned rows, C += ...) use the generic loop.
// GATHER: the A rows are gathered from a per-point matrix and the set-conv first layer's coordinate term + BN + ReLU are applied
// to the fragments (GemmArgs ga_*): LDS-direct loop only, full tiles only (the host checks).
// GMODE 2: the same for the B operand of the weight gradient (B[K][N] = the activated first layer, K = neighbour slots): register-staged
// loop only, the rows' source indices requested one chunk ahead.  GMODE 3: the producer's Z rows of the kind-4 backward epilogue
// (p.Z = the per-point matrix, rows through ga_rows, + the coordinate term).  GMODE 4: the same data gradient over rows in
// inverse-index order, reduced over runs of equal source points in the epilogue instead of stored (seg_epilogue).
template <int BM, int BN, bool A_T, bool B_T, int EPI = 0, int GMODE = 0>
__global__ __launch_bounds__(G_THREADS, (BM > 128 || BN > 128) ? 2 : 3) void gemm_kernel(const GemmArgs p)
{
    constexpr bool GATHER = GMODE == 1, GATHER_B = GMODE == 2, GATHER_Z = GMODE == 3 || GMODE == 4, GATHER_S = GMODE == 4;
    static_assert(!GATHER_Z || (!A_T && !B_T && EPI == 4 && (BM == 128 || (BM == 256 && GMODE == 4)) && BN == 128), "gathered Z rows: the kind-4 data gradient");
    static_assert(!GATHER || (!A_T && B_T), "gathering A operand: A[M][K] W[N][K] layout");
    static_assert(!GATHER_B || (A_T && !B_T), "gathering B operand: A[K][M] B[K][N] layout");
    constexpr int WARPS_M = (BM == 64) ? 2 : ((BN >= 128) ? 2 : 4);
    constexpr int WARPS_N = 4 / WARPS_M;
    constexpr int WM = BM / WARPS_M, WN = BN / WARPS_N;     // wave tile
    constexpr int TM = WM / 32, TN = WN / 32;               // 32x32 MFMA tiles per wave
    static_assert(TM >= 1 && TN >= 1, "tile");
    // LDS images.  An operand whose global rows run along the contraction index (A[M][K], B[N][K]) is
    // kept row-major [rows][36] and read with one ds_read_b128 per 4 k.  An operand stored
    // contraction-major in global memory (A[K][M], B[K][N]) keeps that order in LDS, [32][rows+4]:
    // its staging stores are then contiguous ds_write_b128 (a transposing store would be a 16-way bank
    // conflict) and fragments are read with conflict-free ds_read_b32 (32 consecutive floats per k).
    constexpr int A_LD = A_T ? BM + 4 : G_LDS_LD, A_SZ = A_T ? G_BK * (BM + 4) : BM * G_LDS_LD;
    constexpr int B_LD = B_T ? G_LDS_LD : BN + 4, B_SZ = B_T ? BN * G_LDS_LD : G_BK * (BN + 4);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *As = smem;                                       // [2][A_SZ]
    float *Bs = smem + 2 * A_SZ;                            // [2][B_SZ]

    // ---- XCD-aware tile mapping ----
    // Workgroups are dealt round-robin to the 8 XCDs (id % 8), each with its own L2.  split_k == 1: the
    // column tiles sharing an A row-panel run back to back on ONE XCD.  split_k > 1 (weight gradients:
    // huge contraction, few output tiles): ALL output tiles of one K-slab run on one XCD, so both
    // operand slabs are fetched from HBM once and re-read from that XCD's L2 by the other tiles.
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
    int tm, tn, split;
    {
        const int id = blockIdx.x, nx = 8;
        const int xcd = id % nx, slot = id / nx;
        if (p.split_k > 1) {
            const int tiles = tiles_m * tiles_n;
            split = xcd + nx * (slot / tiles);
            const int t = slot % tiles;
            tm = t / tiles_n; tn = t % tiles_n;
            if (split >= p.split_k) return;
        } else {
            const int per = (tiles_m + nx - 1) / nx;            // row panels per XCD
            split = 0;
            tm = xcd * per + slot / tiles_n;
            tn = slot % tiles_n;
            if (tm >= tiles_m || slot / tiles_n >= per) return;
        }
    }
    const unsigned long long t_start = p.trace ? wall_clock64() : 0ull;
    const int kchunks_total = (p.K + G_BK - 1) / G_BK;
    const int kchunks_per = (kchunks_total + p.split_k - 1) / p.split_k;
    const int kc_begin = split * kchunks_per;
    const int kc_end = min(kchunks_total, kc_begin + kchunks_per);
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave index in an SGPR: LDS-direct destinations and per-wave branches stay scalar
    const int wm = wid / WARPS_N, wn = wid % WARPS_N;

    // ---- staging maps ----
    // non-transposed operand (rows of 32 contiguous k): thread -> (row = tid/8 + 32*i, k4 = (tid%8)*4)
    // transposed operand   (rows of contiguous m/n for one k): thread -> (k = tid/32 + 8*i, x4 = (tid%32)*4)
    // transposed staging: TPR threads cover one k-row (BM or BN contiguous floats), RPI rows per pass
    constexpr int A_TPR = BM / 4, A_RPI = G_THREADS / A_TPR;
    constexpr int B_TPR = BN / 4, B_RPI = G_THREADS / B_TPR;
    constexpr int A_IT = A_T ? (G_BK / A_RPI) : (BM / G_RP);
    constexpr int B_IT = B_T ? (BN / G_RP) : (G_BK / B_RPI);
    float4 ra[A_IT], rb[B_IT];
    // gathering B (GMODE 2): source rows of this thread's B rows of the chunk requested next, their relative coordinates, and the
    // three coordinate weights of the thread's four columns
    int gbrow[GATHER_B ? B_IT : 1];
    float4 gbd[GATHER_B ? B_IT : 1], gbw[3];
    // BN backward fused into the A staging of the weight-gradient layout (GemmArgs::bnb_*): interior tiles only, the host
    // checks the shape.  A thread stages 4 fixed columns m of every chunk, so its coefficients are loop constants.
    float4 rz[A_T ? A_IT : 1];
    const bool bnbA = A_T && p.bnb_z != nullptr;
    float bal[4] = {1.f, 1.f, 1.f, 1.f}, bbe[4] = {0.f, 0.f, 0.f, 0.f}, bga[4] = {0.f, 0.f, 0.f, 0.f}, bmu[4] = {0.f, 0.f, 0.f, 0.f};
    int k0_loaded = 0;
    const int bnb_writers = tiles_n < G_BK ? tiles_n : G_BK;
    if (A_T && bnbA) {
        const int m = m0 + (tid % A_TPR) * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bmu[j] = p.bnb_mean[m + j];
            cmf_bnb_coef(p.bnb_a[m + j], bmu[j], p.bnb_invstd[m + j], p.bnb_sums[m + j], p.bnb_sums[p.M + m + j], p.bnb_ic,
                         bal[j], bbe[j], bga[j]);
        }
    }
    float4 psa = make_float4(1.f, 1.f, 1.f, 1.f), psc = make_float4(0.f, 0.f, 0.f, 0.f);   // A prologue (per chunk)
    float4 qsa = make_float4(1.f, 1.f, 1.f, 1.f), qsc = make_float4(0.f, 0.f, 0.f, 0.f);   // B prologue (per thread)
    unsigned kmask = 0xF;                               // which of this thread's 4 k are < K (edge chunks)
    const bool proA = !A_T && p.pro_a != nullptr;
    const bool proB = !B_T && p.prob_a != nullptr;
    const bool edge_mn = (m0 + BM > p.M) || (n0 + BN > p.N);
    if (proB) {
        const int n = n0 + (tid % B_TPR) * 4;
        float t[8] = {1, 1, 1, 1, 0, 0, 0, 0};
        for (int j = 0; j < 4; ++j) if (n + j < p.N) { t[j] = p.prob_a[n + j]; t[4 + j] = p.prob_c[n + j]; }
        qsa = make_float4(t[0], t[1], t[2], t[3]); qsc = make_float4(t[4], t[5], t[6], t[7]);
    }

    // Loads only ISSUE here (no arithmetic on the results), so all of a chunk's global loads are
    // in flight together under the MFMAs of the previous chunk; the prologue math runs in store_*.
    auto load_tiles = [&](int kc) {
        const int k0 = kc * G_BK;
        const bool fast = !edge_mn && (k0 + G_BK <= p.K);
        k0_loaded = k0;
        if (fast) {
            if (!A_T) {
                const int kk = k0 + (tid % G_KT) * 4;
                if (proA) { psa = *(const float4 *)(p.pro_a + kk); psc = *(const float4 *)(p.pro_c + kk); }
                const float *src = p.A + (long long)(m0 + (tid / G_KT)) * p.lda + kk;
#pragma unroll
                for (int i = 0; i < A_IT; ++i) ra[i] = *(const float4 *)(src + (long long)G_RP * i * p.lda);
            } else {
                const float *src = p.A + (long long)(k0 + tid / A_TPR) * p.lda + m0 + (tid % A_TPR) * 4;
#pragma unroll
                for (int i = 0; i < A_IT; ++i) ra[i] = *(const float4 *)(src + (long long)A_RPI * i * p.lda);
                if (bnbA) {
                    const float *zs = p.bnb_z + (long long)(k0 + tid / A_TPR) * p.ldbz + m0 + (tid % A_TPR) * 4;
#pragma unroll
                    for (int i = 0; i < A_IT; ++i) rz[A_T ? i : 0] = *(const float4 *)(zs + (long long)A_RPI * i * p.ldbz);
                }
            }
            if (B_T) {
                const float *src = p.B + (long long)(n0 + (tid / G_KT)) * p.ldb + k0 + (tid % G_KT) * 4;
#pragma unroll
                for (int i = 0; i < B_IT; ++i) rb[i] = *(const float4 *)(src + (long long)G_RP * i * p.ldb);
            } else if (GATHER_B) {
#pragma unroll
                for (int i = 0; i < B_IT; ++i) {
                    const int k = k0 + tid / B_TPR + B_RPI * i;
                    rb[GATHER_B ? i : 0] = *(const float4 *)(p.B + (long long)gbrow[GATHER_B ? i : 0] * p.ldb + n0 + (tid % B_TPR) * 4);
                    gbd[GATHER_B ? i : 0] = *(const float4 *)(p.ga_dxyz + (long long)k * 4);
                    gbrow[GATHER_B ? i : 0] = k + G_BK < p.K ? p.ga_rows[k + G_BK] : 0;      // for the next chunk: a chunk's MFMAs ahead of its use
                }
            } else {
                const float *src = p.B + (long long)(k0 + tid / B_TPR) * p.ldb + n0 + (tid % B_TPR) * 4;
#pragma unroll
                for (int i = 0; i < B_IT; ++i) rb[i] = *(const float4 *)(src + (long long)B_RPI * i * p.ldb);
            }
            kmask = 0xF;
            return;
        }
        if (GATHER_B) __builtin_trap();                         // (the host only sends full tiles and whole chunks)
        // ---- edge path: bounds-checked, zero-filled ----
        auto ld4 = [&](const float *src, int valid) {
            float t[4] = {0.f, 0.f, 0.f, 0.f};
            if (valid >= 4) return *(const float4 *)src;
            for (int j = 0; j < 4; ++j) if (j < valid) t[j] = src[j];
            return make_float4(t[0], t[1], t[2], t[3]);
        };
        if (!A_T) {
            const int kk = k0 + (tid % G_KT) * 4;
            const int kv = max(0, min(4, p.K - kk));
            kmask = (1u << kv) - 1u;
            if (proA) { psa = ld4(p.pro_a + kk, kv); psc = ld4(p.pro_c + kk, kv); }
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                const int m = m0 + (tid / G_KT) + G_RP * i;
                ra[i] = (m < p.M && kv > 0) ? ld4(p.A + (long long)m * p.lda + kk, kv) : make_float4(0.f, 0.f, 0.f, 0.f);
                if (m >= p.M) ra[i].x = __builtin_nanf("");            // marks a row that must store zeros
            }
        } else {
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                const int k = k0 + tid / A_TPR + A_RPI * i, m = m0 + (tid % A_TPR) * 4;
                ra[i] = (k < p.K && m < p.M) ? ld4(p.A + (long long)k * p.lda + m, p.M - m) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        if (B_T) {
            const int kk = k0 + (tid % G_KT) * 4;
            const int kv = max(0, min(4, p.K - kk));
#pragma unroll
            for (int i = 0; i < B_IT; ++i) {
                const int n = n0 + (tid / G_KT) + G_RP * i;
                rb[i] = (n < p.N && kv > 0) ? ld4(p.B + (long long)n * p.ldb + kk, kv) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        } else {
#pragma unroll
            for (int i = 0; i < B_IT; ++i) {
                const int k = k0 + tid / B_TPR + B_RPI * i, n = n0 + (tid % B_TPR) * 4;
                rb[i] = (k < p.K && n < p.N) ? ld4(p.B + (long long)k * p.ldb + n, p.N - n) : make_float4(0.f, 0.f, 0.f, 0.f);
                if (!(k < p.K)) rb[i].x = __builtin_nanf("");          // out-of-range k row: must store zeros
            }
        }
    };
    auto pro4 = [&](float4 v, const float4 &sa, const float4 &sc) {
        v.x = fmaxf(fmaf(sa.x, v.x, sc.x), 0.f); v.y = fmaxf(fmaf(sa.y, v.y, sc.y), 0.f);
        v.z = fmaxf(fmaf(sa.z, v.z, sc.z), 0.f); v.w = fmaxf(fmaf(sa.w, v.w, sc.w), 0.f);
        return v;
    };
    auto store_A = [&](int buf) {
        float *dst = As + buf * A_SZ;
        if (!A_T) {
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                float4 v = ra[i];
                if (proA) {
                    const bool dead = v.x != v.x;                   // NaN marker from the edge path
                    v = pro4(v, psa, psc);
                    if (dead) v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (kmask != 0xF) { if (!(kmask & 1)) v.x = 0.f; if (!(kmask & 2)) v.y = 0.f;
                                        if (!(kmask & 4)) v.z = 0.f; if (!(kmask & 8)) v.w = 0.f; }
                } else if (v.x != v.x && edge_mn) v = make_float4(0.f, 0.f, 0.f, 0.f);
                *(float4 *)(dst + ((tid / G_KT) + G_RP * i) * G_LDS_LD + (tid % G_KT) * 4) = v;
            }
        } else {
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                const int k = tid / A_TPR + A_RPI * i, m = (tid % A_TPR) * 4;
                float4 v = ra[i];
                if (bnbA) {
                    const float4 zz = rz[A_T ? i : 0];
                    v.x = cmf_bnb_apply(v.x, zz.x, bmu[0], bal[0], bbe[0], bga[0]); v.y = cmf_bnb_apply(v.y, zz.y, bmu[1], bal[1], bbe[1], bga[1]);
                    v.z = cmf_bnb_apply(v.z, zz.z, bmu[2], bal[2], bbe[2], bga[2]); v.w = cmf_bnb_apply(v.w, zz.w, bmu[3], bal[3], bbe[3], bga[3]);
                    // the column tiles that share this row panel stage the same chunk: each writes the rows k with
                    // k % tiles == tn (one slow writer among them would set the pace of a one-round launch)
                    if (p.bnb_out && (k % bnb_writers) == tn) *(float4 *)(p.bnb_out + (long long)(k0_loaded + k) * p.ldbo + m0 + m) = v;
                }
                *(float4 *)(dst + k * A_LD + m) = v;
            }
        }
    };
    auto store_B = [&](int buf) {
        float *dst = Bs + buf * B_SZ;
        if (B_T) {
#pragma unroll
            for (int i = 0; i < B_IT; ++i)
                *(float4 *)(dst + ((tid / G_KT) + G_RP * i) * G_LDS_LD + (tid % G_KT) * 4) = rb[i];
        } else {
#pragma unroll
            for (int i = 0; i < B_IT; ++i) {
                const int k = tid / B_TPR + B_RPI * i, n = (tid % B_TPR) * 4;
                float4 v = rb[i];
                if (GATHER_B) {                                     // z = y + (wx0 dx + wx1 dy + wx2 dz): group_affine_kernel's operations, in its order
                    const float4 d = gbd[GATHER_B ? i : 0];
                    v.x += fmaf(gbw[2].x, d.z, fmaf(gbw[1].x, d.y, gbw[0].x * d.x)); v.y += fmaf(gbw[2].y, d.z, fmaf(gbw[1].y, d.y, gbw[0].y * d.x));
                    v.z += fmaf(gbw[2].z, d.z, fmaf(gbw[1].z, d.y, gbw[0].z * d.x)); v.w += fmaf(gbw[2].w, d.z, fmaf(gbw[1].w, d.y, gbw[0].w * d.x));
                }
                if (proB) {
                    const bool dead = v.x != v.x;
                    v = pro4(v, qsa, qsc);
                    if (dead) v = make_float4(0.f, 0.f, 0.f, 0.f);
                    // columns n >= N hold relu(qsc)=0-filled scale (qsa=1,qsc=0 -> relu(0)=0): already zero
                } else if (v.x != v.x) v = make_float4(0.f, 0.f, 0.f, 0.f);
                *(float4 *)(dst + k * B_LD + n) = v;
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- main loop, LDS-direct variant ----
    constexpr int D_ASLOTS = BM * (G_BK / 4), D_BSLOTS = BN * (G_BK / 4);     // 16-byte slots per operand chunk
    constexpr int D_ANI = D_ASLOTS / G_THREADS, D_BNI = D_BSLOTS / G_THREADS; // load instructions per thread
    constexpr int D_STAGE = (D_ASLOTS + D_BSLOTS) * 4 + (GATHER ? 96 : 32);   // floats: A | B | pro_a[16] pro_c[16] (| wx0[16] wx1[16] wx2[16])
    const bool direct = G_BK == 16 && !edge_mn && (p.K % G_BK == 0) && !p.no_direct && !bnbA && !GATHER_B;
    if (GATHER_B) {
#pragma unroll
        for (int i = 0; i < B_IT; ++i) gbrow[GATHER_B ? i : 0] = p.ga_rows[kc_begin * G_BK + tid / B_TPR + B_RPI * i];
#pragma unroll
        for (int c = 0; c < 3; ++c) gbw[c] = *(const float4 *)(p.ga_wx + (long long)c * p.N + n0 + (tid % B_TPR) * 4);
    }
    if (direct && kc_begin < kc_end) {
        const long long pro_delta = proA ? (long long)(p.pro_c - p.pro_a) : 0ll;
        if (GATHER && !proA) __builtin_trap();
        // gathering A: the source row of each of this lane's D_ANI staging slots (fixed for the tile), and the base of the
        // constant block a lane of wave 0 requests per chunk (pro_a | pro_c | wx0 | wx1 | wx2, 16 floats each)
        long long asrc[D_ANI];
        const float *gconst = nullptr;
        if (GATHER) {
#pragma unroll
            for (int q = 0; q < D_ANI; ++q) asrc[q] = (long long)p.ga_rows[m0 + (((q * 4 + wid) * 64 + lane) >> 2)] * p.lda;
            if (!(B_T && BN > 128)) gconst = (lane < 4 ? p.pro_a : lane < 8 ? p.pro_c : p.ga_wx + (long long)((lane >> 2) - 2) * p.K) + 4 * (lane & 3);
        }
        if (GATHER_S) {
#pragma unroll
            for (int q = 0; q < D_ANI; ++q) asrc[q] = (long long)p.ga_arows[m0 + (((q * 4 + wid) * 64 + lane) >> 2)] * p.lda;
        }
        // 256-column tiles of the gathering forward GEMM: the W rows through a buffer descriptor
        constexpr bool WIDE_B = GATHER && B_T && BN > 128;
        __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void *)(p.B + (long long)n0 * p.ldb), 0, -1, 0x00020000);
        const unsigned offB = (unsigned)((((wid * 64 + lane) >> 2) * (int)p.ldb + 4 * ((lane & 3) ^ g_swz((wid * 64 + lane) >> 2))) * 4);
        // A plain s_barrier: __syncthreads() carries a workgroup fence, which makes the compiler drain EVERY outstanding
        // LDS-direct load (vmcnt(0)) -- the prefetch distance would collapse to zero.  Visibility of the stage that is
        // consumed next is established explicitly: each wave waits for its own loads of that stage (wait_prev), then
        // the barrier; nothing else writes LDS in this loop.
        auto issue = [&](int kc, int st) {
            const int k0 = kc * G_BK;
            float *sa = smem + st * D_STAGE, *sb = sa + D_ASLOTS * 4, *sp = sb + D_BSLOTS * 4;
#pragma unroll
            for (int q = 0; q < D_ANI; ++q) {
                const int grp = q * 4 + wid, sl = grp * 64 + lane;
                const float *g;
                if (!A_T) { const int row = sl >> 2; g = p.A + ((GATHER || GATHER_S) ? asrc[q] : (long long)(m0 + row) * p.lda) + k0 + 4 * ((sl & 3) ^ g_swz(row)); }
                else { const int k = sl / (BM / 4); g = p.A + (long long)(k0 + k) * p.lda + m0 + 4 * ((sl % (BM / 4)) ^ (((k >> 2) & 1) * 8)); }
                __builtin_amdgcn_global_load_lds((g_gptr)g, (g_lptr)(sa + grp * 256), 16, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < D_BNI; ++q) {
                const int grp = q * 4 + wid, sl = grp * 64 + lane;
                if (WIDE_B) {                                       // one lane offset + scalar offsets (the rows of request q lie 64 rows below those of request 0, same swizzle)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (g_lptr)(sb + grp * 256), 16, offB, (unsigned)((q * 64 * (int)p.ldb + k0) * 4), 0, 0);
                    continue;
                }
                const float *g;
                if (B_T) { const int row = sl >> 2; g = p.B + (long long)(n0 + row) * p.ldb + k0 + 4 * ((sl & 3) ^ g_swz(row)); }
                else { const int k = sl / (BN / 4); g = p.B + (long long)(k0 + k) * p.ldb + n0 + 4 * ((sl % (BN / 4)) ^ (((k >> 2) & 1) * 8)); }
                __builtin_amdgcn_global_load_lds((g_gptr)g, (g_lptr)(sb + grp * 256), 16, 0, 0);
            }
            if (GATHER) {
                if (WIDE_B) {                                       // (the lane's source formed here: a 64-bit pointer kept across the loop is two registers this form does not have)
                    int l = lane;
                    asm volatile("" : "+v"(l));
                    if (wid == 0 && l < 20) {
                        const float *gc = (l < 4 ? p.pro_a : l < 8 ? p.pro_c : p.ga_wx + (long long)((l >> 2) - 2) * p.K) + 4 * (l & 3);
                        __builtin_amdgcn_global_load_lds((g_gptr)(gc + k0), (g_lptr)sp, 16, 0, 0);
                    }
                } else
                if (wid == 0 && lane < 20) __builtin_amdgcn_global_load_lds((g_gptr)(gconst + k0), (g_lptr)sp, 16, 0, 0);
            } else
            if (proA && wid == 0 && lane < 8)                       // this chunk's 16 prologue scales and shifts
                __builtin_amdgcn_global_load_lds((g_gptr)(p.pro_a + (lane < 4 ? 0ll : pro_delta) + k0 + 4 * (lane & 3)), (g_lptr)sp, 16, 0, 0);
        };
        // all but the newest issue() of this wave complete (vector-memory loads retire in order)
        auto wait_prev = [&](bool newest_outstanding) {
            if (!newest_outstanding) { CMF_WAIT_VMCNT(0); return; }
            if (proA && wid == 0) { CMF_WAIT_VMCNT(D_ANI + D_BNI + 1); } else { CMF_WAIT_VMCNT(D_ANI + D_BNI); }   // (wave 0: + the constants' request)
        };
        const int nch = kc_end - kc_begin;
        issue(kc_begin, 0);
        if (nch > 1) issue(kc_begin + 1, 1);
        wait_prev(nch > 1);
        __builtin_amdgcn_s_barrier();
        const int frow = lane & 31, h = lane >> 5;
        int arow[TM], aswz[TM], brow[TN], bswz[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) { arow[i] = wm * WM + i * 32 + frow; aswz[i] = g_swz(arow[i]); }
#pragma unroll
        for (int j = 0; j < TN; ++j) { brow[j] = wn * WN + j * 32 + frow; bswz[j] = g_swz(brow[j]); }
        float qa[TN], qc[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + brow[j];
            qa[j] = (proB && n < p.N) ? p.prob_a[n] : 1.f; qc[j] = (proB && n < p.N) ? p.prob_c[n] : 0.f;
        }
        // fragment loader for one k8 step of stage (sa, sb, sp): issues the reads, does not wait
        f32x4 af[2][TM], bf[2][TN], pa4[2], pc4[2];
        // gathering A: ONE set of per-k constants (both fragment sets pass through it: a set's prologue is applied as soon as the
        // set has landed, before the other set's reads are requested) and the rows' relative coordinates
        f32x4 gw0, gw1, gw2, gdq[TM];
        if (GATHER) {
#pragma unroll
            for (int i = 0; i < TM; ++i) gdq[i] = *(const f32x4 *)(p.ga_dxyz + (long long)(m0 + arow[i]) * 4);
        }
        auto read_frags = [&](const float *sa, const float *sb, const float *sp, int k8, int w) {
            const int kq = k8 / 4 + h;                                          // this lane's 4 consecutive k: 4*kq .. 4*kq+3
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if (!A_T) af[w][i] = g_lds_read128(g_lds_addr(sa + (arow[i] * 4 + (kq ^ aswz[i])) * 4));
                else {
                    const unsigned q = g_lds_addr(sa + (4 * kq) * BM + ((((arow[i] >> 2) ^ (h * 8)) << 2) | (arow[i] & 3)));
                    af[w][i].x = g_lds_read32<0>(q); af[w][i].y = g_lds_read32<BM * 4>(q);
                    af[w][i].z = g_lds_read32<BM * 8>(q); af[w][i].w = g_lds_read32<BM * 12>(q);
                }
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if (B_T) bf[w][j] = g_lds_read128(g_lds_addr(sb + (brow[j] * 4 + (kq ^ bswz[j])) * 4));
                else {
                    const unsigned q = g_lds_addr(sb + (4 * kq) * BN + ((((brow[j] >> 2) ^ (h * 8)) << 2) | (brow[j] & 3)));
                    bf[w][j].x = g_lds_read32<0>(q); bf[w][j].y = g_lds_read32<BN * 4>(q);
                    bf[w][j].z = g_lds_read32<BN * 8>(q); bf[w][j].w = g_lds_read32<BN * 12>(q);
                }
            }
            if (GATHER) {
                pa4[0] = g_lds_read128(g_lds_addr(sp + 4 * kq)); pc4[0] = g_lds_read128(g_lds_addr(sp + 16 + 4 * kq));
                gw0 = g_lds_read128(g_lds_addr(sp + 32 + 4 * kq)); gw1 = g_lds_read128(g_lds_addr(sp + 48 + 4 * kq));
                gw2 = g_lds_read128(g_lds_addr(sp + 64 + 4 * kq));
            } else
            if (proA) { pa4[w] = g_lds_read128(g_lds_addr(sp + 4 * kq)); pc4[w] = g_lds_read128(g_lds_addr(sp + 16 + 4 * kq)); }
        };
        auto pin_frags = [&](int w) {
#pragma unroll
            for (int i = 0; i < TM; ++i) g_pin(af[w][i]);
#pragma unroll
            for (int j = 0; j < TN; ++j) g_pin(bf[w][j]);
            if (GATHER) { g_pin(pa4[0]); g_pin(pc4[0]); g_pin(gw0); g_pin(gw1); g_pin(gw2); }
            else if (proA) { g_pin(pa4[w]); g_pin(pc4[w]); }
        };
        // z = y + (wx0 dx + wx1 dy + wx2 dz), a = relu(pa z + pc): the operations of group_affine_kernel followed by the A prologue,
        // in their order (bit-identical to the materialised path)
        auto gather_prologue = [&](int w) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const float dx = gdq[i].x, dy = gdq[i].y, dz = gdq[i].z;
                f32x4 z = af[w][i];
                z.x += fmaf(gw2.x, dz, fmaf(gw1.x, dy, gw0.x * dx)); z.y += fmaf(gw2.y, dz, fmaf(gw1.y, dy, gw0.y * dx));
                z.z += fmaf(gw2.z, dz, fmaf(gw1.z, dy, gw0.z * dx)); z.w += fmaf(gw2.w, dz, fmaf(gw1.w, dy, gw0.w * dx));
                af[w][i].x = fmaxf(fmaf(pa4[0].x, z.x, pc4[0].x), 0.f); af[w][i].y = fmaxf(fmaf(pa4[0].y, z.y, pc4[0].y), 0.f);
                af[w][i].z = fmaxf(fmaf(pa4[0].z, z.z, pc4[0].z), 0.f); af[w][i].w = fmaxf(fmaf(pa4[0].w, z.w, pc4[0].w), 0.f);
            }
        };
        auto prologue_step = [&](int w) {
            if (!GATHER && proA) {                                              // fused BN + ReLU of the producer layer
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    af[w][i].x = fmaxf(fmaf(pa4[w].x, af[w][i].x, pc4[w].x), 0.f); af[w][i].y = fmaxf(fmaf(pa4[w].y, af[w][i].y, pc4[w].y), 0.f);
                    af[w][i].z = fmaxf(fmaf(pa4[w].z, af[w][i].z, pc4[w].z), 0.f); af[w][i].w = fmaxf(fmaf(pa4[w].w, af[w][i].w, pc4[w].w), 0.f);
                }
            }
            if (proB) {
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    bf[w][j].x = fmaxf(fmaf(qa[j], bf[w][j].x, qc[j]), 0.f); bf[w][j].y = fmaxf(fmaf(qa[j], bf[w][j].y, qc[j]), 0.f);
                    bf[w][j].z = fmaxf(fmaf(qa[j], bf[w][j].z, qc[j]), 0.f); bf[w][j].w = fmaxf(fmaf(qa[j], bf[w][j].w, qc[j]), 0.f);
                }
            }
        };
        auto mfma_step = [&](int w) {
            prologue_step(w);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[w][i].x, bf[w][j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[w][i].y, bf[w][j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[w][i].z, bf[w][j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[w][i].w, bf[w][j].w, acc[i][j], 0, 0, 0);
                }
        };
        int st = 0;
#ifndef CMF_NO_SETPRIO
        __builtin_amdgcn_s_setprio(1);
#endif
        // Software-pipelined form: the barrier sits in the MIDDLE of a chunk's MFMAs.  When a wave arrives there it has read
        // all of chunk c (its second fragment set was awaited just before), so the barrier still means "stage c is free and
        // chunk c + 1 is visible" -- but the first fragment reads of chunk c + 1 are issued right behind it and land under
        // the 16 MFMAs of the second half of chunk c, instead of being waited for with nothing to issue.
        {
            {
                const float *sa = smem, *sb = sa + D_ASLOTS * 4, *sp = sb + D_BSLOTS * 4;
                read_frags(sa, sb, sp, 0, 0);
                g_lds_wait(); pin_frags(0);
                if (GATHER) gather_prologue(0);
            }
            for (int c = 0; c < nch; ++c) {
                // the second fragment set of THIS chunk is requested here, not at the end of the previous iteration: the reads are
                // inline asm the compiler cannot see completing, and at a loop back-edge it may copy fragment registers between
                // its per-block assignments -- a copy of a register whose LDS data has not landed yet copies garbage (found in
                // gemm_persist.hip: one launch in ten multiplied a stale fragment)
                {
                    const float *sa = smem + st * D_STAGE, *sb = sa + D_ASLOTS * 4, *sp = sb + D_BSLOTS * 4;
                    read_frags(sa, sb, sp, 8, 1);
                }
                mfma_step(0);
                g_lds_wait(); pin_frags(1);                                         // every LDS read of chunk c by this wave is complete
                if (GATHER) gather_prologue(1);
                if (c + 2 < nch && !(GDIAG(p) & 1)) issue(kc_begin + c + 2, st == 0 ? 2 : st - 1);   // (st + 2) % 3: the stage of chunk c - 1
                const int sn = st == 2 ? 0 : st + 1;
                const float *sa = smem + sn * D_STAGE, *sb = sa + D_ASLOTS * 4, *sp = sb + D_BSLOTS * 4;
                if (c + 1 < nch) {
                    if (!(GDIAG(p) & 3)) wait_prev(c + 2 < nch);
                    if (!(GDIAG(p) & 4)) __builtin_amdgcn_s_barrier();
                    read_frags(sa, sb, sp, 0, 0);                                   // in flight under the second half of chunk c
                }
                mfma_step(1);
                if (c + 1 < nch) { g_lds_wait(); pin_frags(0); if (GATHER) gather_prologue(0); }
                st = sn;
            }
            __builtin_amdgcn_s_barrier();                                            // the epilogue reuses the staging buffers
        }
        __builtin_amdgcn_s_setprio(0);
    } else
    if (GATHER || GATHER_S) __builtin_trap();               // (the host only sends shapes the LDS-direct loop takes)
    else
    if (kc_begin < kc_end) {
        load_tiles(kc_begin);
        store_A(0); store_B(0);
        __syncthreads();
        int buf = 0;
        const int frow = lane & 31, fk = (lane >> 5) * 4;
        for (int kc = kc_begin; kc < kc_end; ++kc) {
            const bool more = kc + 1 < kc_end;
            if (more) load_tiles(kc + 1);                           // global loads in flight under the MFMAs
            const float *a_s = As + buf * A_SZ + (A_T ? fk * A_LD + wm * WM + frow : (wm * WM + frow) * A_LD + fk);
            const float *b_s = Bs + buf * B_SZ + (B_T ? (wn * WN + frow) * B_LD + fk : fk * B_LD + wn * WN + frow);
#pragma unroll
            for (int k8 = 0; k8 < G_BK; k8 += 8) {
                float4 af[TM], bf[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    if (!A_T) af[i] = *(const float4 *)(a_s + i * 32 * A_LD + k8);
                    else { const float *q = a_s + k8 * A_LD + i * 32;
                           af[i] = make_float4(q[0], q[A_LD], q[2 * A_LD], q[3 * A_LD]); }
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (B_T) bf[j] = *(const float4 *)(b_s + j * 32 * B_LD + k8);
                    else { const float *q = b_s + k8 * B_LD + j * 32;
                           bf[j] = make_float4(q[0], q[B_LD], q[2 * B_LD], q[3 * B_LD]); }
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
                    }
                // stage the NEXT chunk into the other LDS buffer three quarters of the way through this
                // chunk's MFMAs: its global loads have had ~3/4 of a chunk to land, and the LDS writes
                // retire under the remaining MFMAs instead of in front of the barrier.
                if (k8 == (G_BK >= 32 ? G_BK - 16 : G_BK - 8) && more) { store_A(buf ^ 1); store_B(buf ^ 1); }
            }
            // LDS-only barrier: __syncthreads() also drains every outstanding global store (vmcnt(0)), and with the fused BN
            // backward this loop has stores in flight (the dZ by-product); the next chunk's loads were consumed by store_*()
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            buf ^= 1;
        }
    }

    const unsigned long long t_main = p.trace ? wall_clock64() : 0ull;
    unsigned long long t_e[3] = {0ull, 0ull, 0ull};     // trace: first transposition visible, band-0 stores issued, last band done
    if (GDIAG(p) & 8) {                                   // timing diagnostic: no epilogue at all (accumulators kept live)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) g_keep(acc[i][j]);
        return;
    }
    // ---- epilogue ----
    // The accumulators (C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)) are
    // transposed through LDS so that every global access of the epilogue is a coalesced 16-byte access:
    // a lane-per-column store tail is store-ISSUE bound (64 dword stores + 64 dword loads per lane).
    constexpr int T_LD = BN + 4;
    float *tile = smem;                                  // [WM][T_LD]: one wave-row band per pass, fits inside the staging buffers
    float *Cout = p.C + (p.split_k > 1 ? (long long)split * p.M * p.ldc : 0);
    const bool want_stats = p.stats != nullptr && p.split_k == 1;
    constexpr int TPR = BN / 4;                          // threads per tile row
    constexpr int RPP = G_THREADS / TPR;                 // rows per pass
    const int col = (tid % TPR) * 4, n = n0 + col;
    const bool vec = (p.ldc % 4 == 0) && (((uintptr_t)Cout) % 16 == 0) && (n + 3 < p.N) &&
                     (!p.bwd_mode || (p.ldz % 4 == 0 && ((uintptr_t)p.Z) % 16 == 0));
    // per-column constants: c0 = bias (forward; the backward modes carry neither bias nor activation) or ea (mode 1)
    // (generic loop only: the fast path keeps them in LDS)
    float c0[4] = {0, 0, 0, 0}, ec[4] = {0, 0, 0, 0}, em[4] = {0, 0, 0, 0}, ei[4] = {0, 0, 0, 0};
    float s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    float qs[3][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    const bool want_q = want_stats && p.bwd_mode != 0 && p.dxyz != nullptr;    // + column sums of out * dxyz_k: the xyz-weight gradient
    const int nstat = want_q ? 5 : 2;
    // One output row of the epilogue: activation / backward mask, statistics, store.  z: the producer's stored
    // pre-activation (backward modes), dd: the row's (dx,dy,dz,0).
    auto finish_row = [&](float (&v)[4], const float (&z)[4], const float4 dd) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float x = p.bwd_mode ? v[q] : act_fn(v[q] + c0[q], p.act);
            if (p.bwd_mode == 1) {
                x = (fmaf(c0[q], z[q], ec[q]) > 0.f) ? x : 0.f;
                s1[q] += x; s2[q] += x * ((z[q] - em[q]) * ei[q]);
                if (want_q) { qs[0][q] += x * dd.x; qs[1][q] += x * dd.y; qs[2][q] += x * dd.z; }
            } else if (p.bwd_mode == 2 || p.bwd_mode == 3) {
                x = z[q] > 0.f ? x : (p.bwd_mode == 2 ? 0.1f * x : 0.f);
                if (want_stats) s1[q] += x;                                           // column sums: bias gradient
                if (want_q) { qs[0][q] += x * dd.x; qs[1][q] += x * dd.y; qs[2][q] += x * dd.z; }
            }
            else if (want_stats) { s1[q] += x; s2[q] += x * x; }
            v[q] = x;
        }
    };
    // workgroup barrier for LDS traffic only: __syncthreads() carries a fence that also waits for every outstanding
    // global store (vmcnt(0))
    auto lds_barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    // ---- fast path (workgroup-uniform): interior tile, 16-byte aligned rows, no read-modify-write of C ----
    // gfx950 counts loads AND stores in one in-order counter (vmcnt): a load issued after a store cannot be awaited
    // without awaiting that store's acknowledgement from L2 (1-3 us under load), and __syncthreads() waits for every
    // outstanding store too.  The straightforward row loop (load Z row, compute, store, next row) therefore ran as a
    // chain of 16 store round trips: ~40 us per tile, as long as the whole K = 256 main loop (measured per workgroup,
    // tools/gemm_timeline.py).  Here every global load of the epilogue is issued BEFORE the first store -- the
    // producer's Z tile into registers, the dxyz rows into LDS -- the row loops are straight-line code, and the
    // workgroup barriers are plain s_barrier behind an LDS-only wait.
    const bool fast_epi = (p.ldc % 4 == 0) && (((uintptr_t)Cout) % 16 == 0) && (n0 + BN <= p.N) && (m0 + BM <= p.M) &&
                          !(p.accumulate && p.split_k == 1) && p.act != 3 &&
                          (!p.bwd_mode || (p.ldz % 4 == 0 && ((uintptr_t)p.Z) % 16 == 0));
    // Bands of 32 rows (one MFMA tile row of the owning waves): 4 (BN = 128) or 2 (BN = 64) rows per thread and band.
    //   transpose -> read rows from LDS -> [Z(band) has landed] compute -> ISSUE Z(band+1) -> store band
    // so the only loads ever waited on are older than every outstanding store.  Band 0 is peeled and the other bands are
    // a real loop: every path into the loop body then has the same queue shape (NIT loads, NIT stores), which lets the
    // compiler's waitcnt pass emit vmcnt(NIT) instead of the vmcnt(0) a merged state forces.  The body is instantiated
    // per epilogue KIND (0 raw store: split-K slabs / plain GEMM; 1 forward: bias, none / ReLU / leaky activation as one
    // slope select, BN statistics; 2 backward through BN + ReLU with the two BN sums; 3 backward through (leaky) ReLU with
    // column sums) and selected once per workgroup -- with the modes decided per element the body is thousands of
    // instructions and 140 registers.
    auto fast_epilogue = [&](auto kind_c) {
        constexpr int KIND = decltype(kind_c)::value;
        constexpr bool USE_Z = KIND >= 2, WQ = KIND >= 4, BNR = KIND == 2 || KIND == 4;
        constexpr int NB = BM / 32, NIT = 32 / RPP;
        static_assert(32 % RPP == 0 && NIT >= 1 && TM <= 4, "band rows");
        const int rr = tid / TPR;
        f32x4 zp[NIT];
        const float *zrow = p.Z + (long long)(m0 + rr) * p.ldz + n;
        float *crow = Cout + (long long)(m0 + rr) * p.ldc + n;
        auto load_z = [&](int band) {
#pragma unroll
            for (int it = 0; it < NIT; ++it) zp[it] = *(const f32x4 *)(zrow + (long long)(band * 32 + it * RPP) * p.ldz);
        };
        if (USE_Z) load_z(0);
        float *dq = smem + 32 * T_LD;                                           // [BM][4] behind the transposition tile
        float *cst = dq + BM * 4;                                               // [4][BN] column constants (kinds >= 1)
        if (WQ && tid < BM) *(f32x4 *)(dq + tid * 4) = *(const f32x4 *)(p.dxyz + (long long)(m0 + tid) * 4);
        // the per-column constants live in LDS and are re-read per band: kept in registers for the whole epilogue they
        // (16) and the statistics (8-20) pushed the backward kinds to 150 registers = 2 workgroups per CU
        if (KIND == 1 && tid < BN) cst[tid] = p.bias ? p.bias[n0 + tid] : 0.f;
        if (BNR && tid < BN) {
            cst[tid] = p.ea[n0 + tid]; cst[BN + tid] = p.ec[n0 + tid];
            cst[2 * BN + tid] = p.emean[n0 + tid]; cst[3 * BN + tid] = p.einvstd[n0 + tid];
        }
        const float slope = KIND == 1 ? (p.act == 1 ? 0.f : (p.act == 2 ? 0.1f : 1.f)) : (p.bwd_mode == 2 ? 0.1f : 0.f);
        auto spill_tile = [&](const f32x16 (&a)[TN]) {                          // this wave's 32 rows x WN columns -> LDS
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    tile[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * T_LD + wn * WN + j * 32 + (lane & 31)] = a[j][r];
        };
        auto do_band = [&](int band) {
            if (wm == band / TM) {
                const int ti = band % TM;                // static accumulator index per branch (a dynamic one would go to scratch)
                if (ti == 0) spill_tile(acc[0]);
                else if (TM > 1 && ti == 1) spill_tile(acc[1 % TM]);
                else if (TM > 2 && ti == 2) spill_tile(acc[2 % TM]);
                else if (TM > 3) spill_tile(acc[3 % TM]);
            }
            lds_barrier();
            if (p.trace && band == 0) t_e[0] = wall_clock64();
            f32x4 t4[NIT];
#pragma unroll
            for (int it = 0; it < NIT; ++it) t4[it] = *(const f32x4 *)(tile + (rr + it * RPP) * T_LD + col);
            if (KIND != 0) {
                f32x4 k0 = {0.f, 0.f, 0.f, 0.f}, k1 = k0, k2 = k0, k3 = k0;
                if (KIND == 1 || BNR) k0 = *(const f32x4 *)(cst + col);
                if (BNR) { k1 = *(const f32x4 *)(cst + BN + col); k2 = *(const f32x4 *)(cst + 2 * BN + col); k3 = *(const f32x4 *)(cst + 3 * BN + col); }
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    f32x4 d4 = {0.f, 0.f, 0.f, 0.f};
                    if (WQ) d4 = *(const f32x4 *)(dq + (band * 32 + rr + it * RPP) * 4);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float x = t4[it][q];
                        if (KIND == 1) {
                            x += k0[q];
                            x = x > 0.f ? x : (slope == 0.f ? 0.f : slope * x);      // select, not 0 * x: -inf must give 0 like torch.relu
                            s1[q] += x; s2[q] += x * x;
                        } else if (BNR) {
                            const float z = zp[it][q];
                            x = (fmaf(k0[q], z, k1[q]) > 0.f) ? x : 0.f;
                            s1[q] += x; s2[q] += x * ((z - k2[q]) * k3[q]);
                        } else {
                            x = zp[it][q] > 0.f ? x : (slope == 0.f ? 0.f : slope * x);
                            s1[q] += x;
                        }
                        if (WQ) { qs[0][q] += x * d4.x; qs[1][q] += x * d4.y; qs[2][q] += x * d4.z; }
                        t4[it][q] = x;
                    }
                }
            }
            if (USE_Z && band + 1 < NB) load_z(band + 1);                       // issued ahead of this band's stores
#pragma unroll
            for (int it = 0; it < NIT; ++it) *(f32x4 *)(crow + (long long)(band * 32 + it * RPP) * p.ldc) = t4[it];
        };
        do_band(0);
        if (p.trace) t_e[1] = wall_clock64();
#pragma unroll 1
        for (int band = 1; band < NB; ++band) {
            lds_barrier();                               // previous band fully consumed
            do_band(band);
        }
        if (p.trace) t_e[2] = wall_clock64();
        if (want_stats) lds_barrier();                   // all tile reads done: the LDS is reused for the column reduction
    };
    // ---- direct form of the fast path (kinds 0 / 1): no transposition.  In the accumulator layout a store instruction
    // of register r already writes two full 128-byte row segments (lanes 0-31: row rho(r), lanes 32-63: row rho(r) + 4), a
    // lane owns ONE column per accumulator block (its statistics are plain per-lane sums), and the producer's Z tile is read
    // with the same map -- every load of the epilogue is issued before its first store, there is no band loop, no LDS round
    // trip and no barrier until the column sums of the wave rows are combined.
    bool stats_done = false;
    auto direct_epilogue = [&](auto kind_c) {
        constexpr int KIND = decltype(kind_c)::value;
        constexpr bool USE_Z = KIND >= 2, WQ = KIND >= 4, BNR = KIND == 2 || KIND == 4;
        const int h = lane >> 5, cl = lane & 31;
        // wave-uniform tile coordinates in scalar registers: row bases are scalar, a lane adds ONE offset (its half-wave's
        // 4 rows down, its column) -- per-lane 64-bit addresses for the 2 x 32 rows would take 128 registers
        const int wms = __builtin_amdgcn_readfirstlane(wm), wns = __builtin_amdgcn_readfirstlane(wn);
        const int rbase = m0 + wms * WM;                              // + i * 32 + rho(r)   (+ 4 h per lane)
        const int cbase = n0 + wns * WN;                              // + j * 32            (+ cl per lane)
        const unsigned lane_c = (unsigned)((4 * h * (int)p.ldc + cl) * 4), lane_z = (unsigned)((4 * h * (int)p.ldz + cl) * 4);
        float *dq = smem;                                             // [BM][4] dxyz rows of the tile
        if (WQ) {
            if (tid < BM) *(f32x4 *)(dq + tid * 4) = *(const f32x4 *)(p.dxyz + (long long)(m0 + tid) * 4);
        }
        // the producer's Z rows of one block row (32 rows of the wave tile); block row i + 1 is requested before block row i
        // is stored, so no load is ever waited on behind a store (one in-order counter for both on gfx950)
        float zv[2][USE_Z ? TN : 1][16];
        auto load_z = [&](int i, float (&dst)[USE_Z ? TN : 1][16]) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float *zr = p.Z + (long long)(rbase + i * 32 + (r & 3) + 8 * (r >> 2)) * p.ldz + cbase;
#pragma unroll
                for (int j = 0; j < TN; ++j) dst[j][r] = *(const float *)((const char *)(zr + j * 32) + lane_z);
            }
        };
        if (USE_Z) load_z(0, zv[0]);
        float k0[TN], k1[TN], k2[TN], k3[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            k0[j] = k1[j] = k2[j] = k3[j] = 0.f;
            const int c = cbase + j * 32 + cl;
            if (KIND == 1) k0[j] = p.bias ? p.bias[c] : 0.f;
            if (BNR) { k0[j] = p.ea[c]; k1[j] = p.ec[c]; k2[j] = p.emean[c]; k3[j] = p.einvstd[c]; }
        }
        const float slope = KIND == 1 ? (p.act == 1 ? 0.f : (p.act == 2 ? 0.1f : 1.f)) : (p.bwd_mode == 2 ? 0.1f : 0.f);
        if (WQ) lds_barrier();
        float t1[TN], t2[TN], q0[TN], q1[TN], q2[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) t1[j] = t2[j] = q0[j] = q1[j] = q2[j] = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            float xs[TN][16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ru = wms * WM + i * 32 + (r & 3) + 8 * (r >> 2);            // row inside the tile, before the lane's + 4 h
                f32x4 d4 = {0.f, 0.f, 0.f, 0.f};
                if (WQ) d4 = *(const f32x4 *)(dq + (ru + 4 * h) * 4);
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    float x = acc[i][j][r];
                    if (KIND == 1) {
                        x += k0[j];
                        x = x > 0.f ? x : (slope == 0.f ? 0.f : slope * x);
                        t1[j] += x; t2[j] += x * x;
                    } else if (BNR) {
                        const float z = zv[i & 1][USE_Z ? j : 0][r];
                        x = (fmaf(k0[j], z, k1[j]) > 0.f) ? x : 0.f;
                        t1[j] += x; t2[j] += x * ((z - k2[j]) * k3[j]);
                    } else if (KIND >= 2) {
                        x = zv[i & 1][USE_Z ? j : 0][r] > 0.f ? x : (slope == 0.f ? 0.f : slope * x);
                        t1[j] += x;
                    }
                    if (WQ) { q0[j] += x * d4[0]; q1[j] += x * d4[1]; q2[j] += x * d4[2]; }
                    xs[j][r] = x;
                }
            }
            if (USE_Z && i + 1 < TM) load_z(i + 1, zv[(i + 1) & 1]);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float *cr = Cout + (long long)(m0 + wms * WM + i * 32 + (r & 3) + 8 * (r >> 2)) * p.ldc + cbase;
#pragma unroll
                for (int j = 0; j < TN; ++j) *(float *)((char *)(cr + j * 32) + lane_c) = xs[j][r];
            }
        }
        if (want_stats) {
            // per-lane column sums -> the two half-waves -> the WARPS_M wave rows (LDS, fixed order)
            float *red = smem + BM * 4;                               // [WARPS_M][nstat][BN]
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                float v1 = t1[j] + __shfl_xor(t1[j], 32, 64), v2 = t2[j] + __shfl_xor(t2[j], 32, 64);
                const int c = wns * WN + j * 32 + cl;
                if (h == 0) { red[(wms * nstat + 0) * BN + c] = v1; red[(wms * nstat + 1) * BN + c] = v2; }
                if (WQ) {
                    const float w0 = q0[j] + __shfl_xor(q0[j], 32, 64), w1 = q1[j] + __shfl_xor(q1[j], 32, 64),
                                w2 = q2[j] + __shfl_xor(q2[j], 32, 64);
                    if (h == 0) { red[(wms * nstat + 2) * BN + c] = w0; red[(wms * nstat + 3) * BN + c] = w1; red[(wms * nstat + 4) * BN + c] = w2; }
                }
            }
            lds_barrier();
            // (partial sums are consumed per 128-row tile, cmf_gemm_tiles_m: a 256-row tile's two wave rows are one partial row each)
            constexpr int SPT = BM > 128 ? BM / 128 : 1, WPS = WARPS_M / SPT;
            for (int c = tid; c < SPT * nstat * BN; c += G_THREADS) {
                const int sp = c / (nstat * BN), which = (c / BN) % nstat, cc = c % BN;
                float sum = 0.f;
#pragma unroll
                for (int g = 0; g < WPS; ++g) sum += red[((sp * WPS + g) * nstat + which) * BN + cc];
                p.stats[(((long long)tm * SPT + sp) * nstat + which) * p.N + n0 + cc] = sum;
            }
        }
        stats_done = true;
    };
    // ---- wave-private form of the fast path (backward kinds): every wave transposes ITS OWN 32 x 64 block
    // rows through a private LDS tile and finishes them alone -- LDS operations of one wave execute in order, so between
    // the accumulator stores and the row reads there is no barrier and no wait, the four waves never meet until the column
    // sums are combined, and nobody idles while two of them spill a band (the banded form above: 2.5 us until the first
    // band is visible + 2.4 us per band behind workgroup barriers, 13.7 us per K = 256 data-gradient tile against a 15.6 us
    // main loop).  Rows are still read and written as 16-byte pieces of 256-byte segments (16 lanes per row, 4 rows per
    // instruction); the producer's Z rows of block row i + 1 are requested before block row i is stored.
    auto wave_epilogue = [&](auto kind_c) {
        constexpr int KIND = decltype(kind_c)::value;
        constexpr bool USE_Z = KIND >= 2, WQ = KIND >= 4, BNR = KIND == 2 || KIND == 4;
        static_assert(WN == 64 && TN == 2, "wave tile 64 columns wide");
        constexpr int W_LD = WN;                                      // 64 floats = one pass over the 64 banks: the 32 lanes of a ds_write_b32 group
                                                                      // write one row, the 16 lanes of a ds_read_b128 group read whole rows -- no padding needed
        constexpr int W_TILE = 32 * W_LD, W_SZ = W_TILE + WM * 4 + (GATHER_Z ? 7 : 4) * WN;   // per wave: transposition tile | dxyz rows | column constants (| wx planes)
        float *wt = smem + wid * W_SZ, *wdq = wt + W_TILE;
        const int rl = lane >> 4, c4 = (lane & 15) * 4;               // row inside a group of 4, first of the lane's 4 columns
        const int cb = wn * WN + c4;                                  // tile-local column; global: n0 + cb
        // wave-uniform row bases in scalar registers, ONE 32-bit per-lane byte offset per matrix (the lane's row of the group
        // of 4 and its columns): per-lane 64-bit addresses for 16 rows x 2 matrices would take 64 registers
        const int wms = __builtin_amdgcn_readfirstlane(wm), wns = __builtin_amdgcn_readfirstlane(wn);
        const int rbase = m0 + wms * WM;
        const float *zbase = GATHER_Z ? p.Z + n0 + wns * WN : p.Z + (long long)rbase * p.ldz + n0 + wns * WN;
        float *cbase = Cout + (long long)rbase * p.ldc + n0 + wns * WN;
        const unsigned lane_z = (unsigned)((rl * (int)p.ldz + c4) * 4), lane_c = (unsigned)((rl * (int)p.ldc + c4) * 4);
        // Z rows of ONE half block row (16 rows: 4 per lane) at a time, requested one half ahead: the next half's loads are
        // issued after this half's arithmetic and BEFORE its stores, so no load is ever waited on behind a store
        f32x4 zp[4];
        int zi[GATHER_Z ? 4 : 1];                                    // gathered Z: the source rows of the half requested next (one half ahead of their use)
        auto load_zi = [&](int g) {
#pragma unroll
            for (int u = 0; u < (GATHER_Z ? 4 : 0); ++u) zi[GATHER_Z ? u : 0] = p.ga_rows[rbase + g * 16 + u * 4 + rl];
        };
        auto load_z = [&](int g) {                                    // g = 2 * block row + half
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (GDIAG(p) & 32) { zp[u] = f32x4{1.f, 1.f, 1.f, 1.f}; continue; }            // timing diagnostic: no Z loads
                if (GATHER_Z) zp[u] = *(const f32x4 *)(zbase + (long long)zi[GATHER_Z ? u : 0] * p.ldz + c4);
                else zp[u] = *(const f32x4 *)((const char *)(zbase + (long long)(g * 16 + u * 4) * p.ldz) + lane_z);
            }
            if (GATHER_Z && g + 1 < 2 * TM) load_zi(g + 1);
        };
        if (GATHER_Z) load_zi(0);
        if (USE_Z) load_z(0);
        // per-column constants and the wave's dxyz rows: wave-private LDS (re-read per half: 16 registers less)
        float *wk = wdq + WM * 4;                                     // [4][WN]
        if (WQ) {
#pragma unroll
            for (int q = lane; q < WM; q += 64) *(f32x4 *)(wdq + q * 4) = *(const f32x4 *)(p.dxyz + (long long)(rbase + q) * 4);
        }
        if (KIND == 1) wk[lane] = p.bias ? p.bias[n0 + wns * WN + lane] : 0.f;
        if (BNR) {
            const int c = n0 + wns * WN + lane;
            wk[lane] = p.ea[c]; wk[WN + lane] = p.ec[c]; wk[2 * WN + lane] = p.emean[c]; wk[3 * WN + lane] = p.einvstd[c];
            if (GATHER_Z) { wk[4 * WN + lane] = p.ga_wx[c]; wk[5 * WN + lane] = p.ga_wx[(long long)p.N + c]; wk[6 * WN + lane] = p.ga_wx[2ll * p.N + c]; }
        }
        const float slope = KIND == 1 ? (p.act == 1 ? 0.f : (p.act == 2 ? 0.1f : 1.f)) : (p.bwd_mode == 2 ? 0.1f : 0.f);
        // compiler fences (no instructions): without them the scheduler hoists the LDS reads and Z loads of later halves above
        // the arithmetic of earlier ones -- 256 registers and spills instead of ~150
        auto fence = [&]() { asm volatile("" ::: "memory"); };
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    wt[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * W_LD + j * 32 + (lane & 31)] = acc[i][j][r];
            if (p.trace && i == 0) t_e[0] = wall_clock64();
            fence();
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {                           // two halves of 16 rows
                f32x4 t4[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) t4[u] = *(const f32x4 *)(wt + (rl + (hf * 4 + u) * 4) * W_LD + c4);
                if (KIND != 0) {
                    f32x4 k0 = {0.f, 0.f, 0.f, 0.f}, k1 = k0, k2 = k0, k3 = k0;
                    if (KIND == 1 || BNR) k0 = *(const f32x4 *)(wk + c4);
                    if (BNR) { k1 = *(const f32x4 *)(wk + WN + c4); k2 = *(const f32x4 *)(wk + 2 * WN + c4); k3 = *(const f32x4 *)(wk + 3 * WN + c4); }
                    f32x4 g0 = k0, g1 = k0, g2 = k0;
                    if (GATHER_Z) { g0 = *(const f32x4 *)(wk + 4 * WN + c4); g1 = *(const f32x4 *)(wk + 5 * WN + c4); g2 = *(const f32x4 *)(wk + 6 * WN + c4); }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        f32x4 d4 = {0.f, 0.f, 0.f, 0.f};
                        if (WQ) d4 = *(const f32x4 *)(wdq + (i * 32 + rl + (hf * 4 + u) * 4) * 4);
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            float x = t4[u][q];
                            if (KIND == 1) {
                                x += k0[q];
                                x = x > 0.f ? x : (slope == 0.f ? 0.f : slope * x);
                                s1[q] += x; s2[q] += x * x;
                            } else if (BNR) {
                                // (gathered: z = y + (wx0 dx + wx1 dy + wx2 dz), group_affine_kernel's operations in its order)
                                const float z = GATHER_Z ? zp[u][q] + fmaf(g2[q], d4.z, fmaf(g1[q], d4.y, g0[q] * d4.x)) : zp[u][q];
                                x = (fmaf(k0[q], z, k1[q]) > 0.f) ? x : 0.f;
                                s1[q] += x; s2[q] += x * ((z - k2[q]) * k3[q]);
                            } else {
                                x = zp[u][q] > 0.f ? x : (slope == 0.f ? 0.f : slope * x);
                                s1[q] += x;
                            }
                            if (WQ) { qs[0][q] += x * d4.x; qs[1][q] += x * d4.y; qs[2][q] += x * d4.z; }
                            t4[u][q] = x;
                        }
                    }
                }
                // pin the running sums here: otherwise the scheduler sinks the statistics below the stores and spills this
                // half's values and Z rows to scratch for them (121 spilled registers)
                if (KIND != 0) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        asm volatile("" : "+v"(s1[q]));
                        if (KIND == 1 || BNR) asm volatile("" : "+v"(s2[q]));
                        if (WQ) asm volatile("" : "+v"(qs[0][q]), "+v"(qs[1][q]), "+v"(qs[2][q]));
                    }
                }
                fence();
                if (USE_Z && 2 * i + hf + 1 < 2 * TM) load_z(2 * i + hf + 1);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (GDIAG(p) & 16) { asm volatile("" :: "v"(t4[u])); continue; }                // timing diagnostic: no C stores
                    *(f32x4 *)((char *)(cbase + (long long)(i * 32 + (hf * 4 + u) * 4) * p.ldc) + lane_c) = t4[u];
                }
                fence();
            }
            if (p.trace && i == 0) t_e[1] = wall_clock64();
        }
        if (p.trace) t_e[2] = wall_clock64();
        if (want_stats) {
            // the lane's 16 rows -> the 4 row groups of the wave (lanes l, l^16, l^32, l^48) -> the WARPS_M wave rows (LDS)
            lds_barrier();                                            // every wave is done with its tile: the LDS is reused
            float *red = smem;                                        // [WARPS_M][nstat][BN]
            auto fold = [&](float v) { v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64); return v; };
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float v1 = fold(s1[q]), v2 = fold(s2[q]);
                if (rl == 0) { red[(wm * nstat + 0) * BN + cb + q] = v1; red[(wm * nstat + 1) * BN + cb + q] = v2; }
                if (WQ) {
                    const float w0 = fold(qs[0][q]), w1 = fold(qs[1][q]), w2 = fold(qs[2][q]);
                    if (rl == 0) { red[(wm * nstat + 2) * BN + cb + q] = w0; red[(wm * nstat + 3) * BN + cb + q] = w1; red[(wm * nstat + 4) * BN + cb + q] = w2; }
                }
            }
            lds_barrier();
            constexpr int SPT = BM > 128 ? BM / 128 : 1, WPS = WARPS_M / SPT;
            for (int c = tid; c < SPT * nstat * BN; c += G_THREADS) {
                const int sp = c / (nstat * BN), which = (c / BN) % nstat, cc = c % BN;
                float sum = 0.f;
#pragma unroll
                for (int g = 0; g < WPS; ++g) sum += red[((sp * WPS + g) * nstat + which) * BN + cc];
                p.stats[(((long long)tm * SPT + sp) * nstat + which) * p.N + n0 + cc] = sum;
            }
        }
        stats_done = true;
    };
    // ---- segmented form of the kind-4 epilogue (GMODE 4): the output rows are neighbour slots sorted by source point, and what the
    // model needs of this data gradient is only its sum per source point (the scatter of the grouping's backward pass) and the five
    // column statistics.  A lane owns ONE column of the wave's 64 x 64 tile and walks its 64 rows: mask and statistics per element as
    // in wave_epilogue, a running sum that is stored to ga_pieces[(point + range)] and reset whenever the source point changes --
    // (point, range) pairs are monotone along the rows, so point + range numbers the pieces uniquely and the consumer adds the pieces
    // of a point in range order (deterministic).  The 1 GB the M x N gradient took at the largest scale is neither written nor read.
    auto seg_epilogue = [&](auto only_when_called) {
        static_assert(WM % 64 == 0 && WN == 64 && TN == 2, "wave tile 64 h x 64");
        constexpr int S_SZ = 32 * 64 + 64 * 4;                        // per wave: transposition tile | dxyz rows (of one 64-row half)
        float *wt = smem + wid * S_SZ, *wdq = wt + 32 * 64;
        const int wms = __builtin_amdgcn_readfirstlane(wm), wns = __builtin_amdgcn_readfirstlane(wn);
        const int c = n0 + wns * WN + lane;                           // this lane's column
        const float k0 = p.ea[c], k1 = p.ec[c], k2 = p.emean[c], k3 = p.einvstd[c];
        const float g0 = p.ga_wx[c], g1 = p.ga_wx[(long long)p.N + c], g2 = p.ga_wx[2ll * p.N + c];
        const float *ycol = p.Z + c;
        float *pcol = p.ga_pieces + c;
        float t1 = 0.f, t2 = 0.f, q0 = 0.f, q1 = 0.f, q2 = 0.f;
        // a wave tile of 128 rows (256-row workgroup tiles) is two 64-row halves, each a range of its own in `pieces`
#pragma unroll
        for (int hv = 0; hv < WM / 64; ++hv) {
        const int rbase = m0 + wms * WM + hv * 64;
        if (hv > 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the previous half's reads of wdq are complete
        // the relative coordinates of row `lane` stay in the lane and are read per row with v_readlane (scalar operands of the row's
        // arithmetic: no LDS read, no vector registers)
        const f32x4 dq4 = *(const f32x4 *)(p.dxyz + (long long)(rbase + lane) * 4);
        const int dqx = __float_as_int(dq4[0]), dqy = __float_as_int(dq4[1]), dqz = __float_as_int(dq4[2]);
        // the source point of row `lane` stays in the lane (read per row with v_readlane); the rows where it changes as a wave mask
        const int ptv = p.ga_rows[rbase + lane];
        const int ptb = __shfl_up(ptv, 1, 64);
        const unsigned long long starts = __ballot(lane == 0 || ptv != ptb);
        const long long range = rbase / 64;
        float seg = 0.f;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // (wave-private LDS: the wave's own writes above, in order)
        if (p.trace && hv == 0) t_e[0] = wall_clock64();
        int pt_cur = 0;
        // the per-point rows of a whole 32-row block row requested at once (32 loads in flight; rows of one run re-read the same 256
        // bytes), the second block row's while the first is walked.  [Eight at a time, one group ahead, each group of 8 rows exposed a
        // full load round trip under the other workgroups' operand streams: 12-14 us per block row, 30 us per tile's epilogue against a
        // 27 us main loop -- tools/dxsum_probe.py, DXSUM_TIMELINE=1]
        float ya[32], yb[32];
        // (a 64-bit scalar multiply-add per row was 9 of a load's 10 instructions; with the matrix below 4 GB the row offset is one
        //  32-bit scalar product and the lane's column sits in the buffer instruction's offset register)
        const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void *)p.Z, 0, -1, 0x00020000);
        const unsigned ycol_off = (unsigned)c * 4u, ldz4 = (unsigned)p.ldz * 4u;
        auto load_y = [&](int row0, float (&y)[32]) {
            if (GDIAG(p) & 32) {                                     // timing diagnostic: no per-point loads
#pragma unroll
                for (int u = 0; u < 32; ++u) y[u] = 1.f;
                return;
            }
            if (p.ga_y32) {
#pragma unroll
                for (int u = 0; u < 32; ++u)
                    y[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ry, ycol_off, (unsigned)__builtin_amdgcn_readlane(ptv, row0 + u) * ldz4, 0));
            } else {
#pragma unroll
                for (int u = 0; u < 32; ++u) y[u] = ycol[(long long)__builtin_amdgcn_readlane(ptv, row0 + u) * p.ldz];
            }
        };
        auto rows8 = [&](int i, int row0, const float *y) {           // rows row0 .. row0 + 7 of the half (block row i of it); y: their 8 values
            // the group's 8 accumulator values out of the LDS tile up front: behind the run-boundary branches below each read would be
            // issued and awaited row by row
            float xr[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) xr[u] = wt[(row0 + u - i * 32) * 64 + lane];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int row = row0 + u;
                if ((starts >> row) & 1ull) {                        // wave-uniform: a run of equal source points begins
                    if (row > 0) { pcol[((long long)pt_cur + range) * p.N] = seg; t1 += seg; }
                    seg = 0.f; pt_cur = __builtin_amdgcn_readlane(ptv, row);
                }
                const float ddx = __int_as_float(__builtin_amdgcn_readlane(dqx, row));
                const float ddy = __int_as_float(__builtin_amdgcn_readlane(dqy, row));
                const float ddz = __int_as_float(__builtin_amdgcn_readlane(dqz, row));
                float x = xr[u];
                const float z = y[u] + fmaf(g2, ddz, fmaf(g1, ddy, g0 * ddx));
                x = (fmaf(k0, z, k1) > 0.f) ? x : 0.f;
                t2 += x * ((z - k2) * k3);
                q0 += x * ddx; q1 += x * ddy; q2 += x * ddz;
                seg += x;
            }
        };
        load_y(0, ya);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    wt[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 64 + j * 32 + (lane & 31)] = acc[(2 * hv + i) % TM][j][r];
            if (i == 0) load_y(32, yb);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // (compiler fences between the groups of 8 rows: without them the scheduler hoists every LDS read of the block row above
            //  the arithmetic and spills)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                asm volatile("" ::: "memory");
                if (GDIAG(p) & 64) { seg += ya[g] + yb[g]; continue; }       // timing diagnostic: no row walk
                rows8(i, i * 32 + g * 8, (i == 0 ? ya : yb) + g * 8);
            }
            asm volatile("" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the tile is read before the next block row overwrites it
            if (p.trace && hv == 0 && i == 0) t_e[1] = wall_clock64();
        }
        pcol[((long long)pt_cur + range) * p.N] = seg; t1 += seg;    // (s1 = the sum of the runs' sums)
        }
        if (p.trace) t_e[2] = wall_clock64();
        // statistics: one partial row per 128 rows like the other epilogues -- the wave rows of a 128-row slice through LDS
        lds_barrier();
        float *red = smem;                                            // [WARPS_M][5][BN]
        const int cc = wns * WN + lane;
        red[(wms * 5 + 0) * BN + cc] = t1; red[(wms * 5 + 1) * BN + cc] = t2;
        red[(wms * 5 + 2) * BN + cc] = q0; red[(wms * 5 + 3) * BN + cc] = q1; red[(wms * 5 + 4) * BN + cc] = q2;
        lds_barrier();
        constexpr int SPT = BM > 128 ? BM / 128 : 1, WPS = WARPS_M / SPT;
        for (int e = tid; e < SPT * 5 * BN; e += G_THREADS) {
            const int sp = e / (5 * BN), which = (e / BN) % 5, col = e % BN;
            float sum = red[((sp * WPS) * 5 + which) * BN + col];
#pragma unroll
            for (int g = 1; g < WPS; ++g) sum += red[((sp * WPS + g) * 5 + which) * BN + col];
            p.stats[(((long long)tm * SPT + sp) * 5 + which) * p.N + n0 + col] = sum;
        }
        stats_done = true;
    };
    if constexpr (GATHER_S) seg_epilogue(0);
    else
    if (fast_epi && epilogue_kind(p) == EPI) {
        if constexpr (BM >= 128 && EPI <= 1) direct_epilogue(std::integral_constant<int, EPI>{});
        else if constexpr (BM >= 128 && BN == 128) wave_epilogue(std::integral_constant<int, EPI>{});
        else fast_epilogue(std::integral_constant<int, EPI>{});
    } else {
    if (p.split_k == 1)
        for (int q = 0; q < 4; ++q)
            if (n + q < p.N) {
                if (p.bwd_mode == 1) { c0[q] = p.ea[n + q]; ec[q] = p.ec[n + q]; em[q] = p.emean[n + q]; ei[q] = p.einvstd[n + q]; }
                else if (p.bias && p.bwd_mode == 0) c0[q] = p.bias[n + q];
            }
#pragma unroll
    for (int band = 0; band < WARPS_M; ++band) {
        if (band > 0) __syncthreads();                   // previous band fully consumed
        if (wm == band) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                        const int c = wn * WN + j * 32 + (lane & 31);
                        tile[row * T_LD + c] = acc[i][j][r];
                    }
        }
        __syncthreads();
        for (int r0 = tid / TPR; r0 < WM; r0 += RPP) {
            const int m = m0 + band * WM + r0;
            if (m >= p.M) break;
            const float4 t4 = *(const float4 *)(tile + r0 * T_LD + col);
            float v[4] = {t4.x, t4.y, t4.z, t4.w};
            if (p.split_k == 1) {
                float z[4] = {0, 0, 0, 0};
                float4 dd = make_float4(0.f, 0.f, 0.f, 0.f);
                if (want_q) dd = *(const float4 *)(p.dxyz + (long long)m * 4);
                if (p.bwd_mode) {
                    const float *zp = p.Z + (long long)m * p.ldz + n;
                    if (vec) { const float4 z4 = *(const float4 *)zp; z[0] = z4.x; z[1] = z4.y; z[2] = z4.z; z[3] = z4.w; }
                    else for (int q = 0; q < 4; ++q) if (n + q < p.N) z[q] = zp[q];
                }
                finish_row(v, z, dd);
            }
            float *dst = Cout + (long long)m * p.ldc + n;
            if (vec) {
                float4 o = make_float4(v[0], v[1], v[2], v[3]);
                if (p.accumulate && p.split_k == 1) { const float4 c4 = *(const float4 *)dst; o.x += c4.x; o.y += c4.y; o.z += c4.z; o.w += c4.w; }
                *(float4 *)dst = o;
            } else {
                for (int q = 0; q < 4; ++q)
                    if (n + q < p.N) dst[q] = (p.accumulate && p.split_k == 1) ? dst[q] + v[q] : v[q];
            }
        }
    }
    if (want_stats) __syncthreads();                     // all tile reads done: reuse LDS for the column reduction
    }
    if (want_stats && !stats_done) {
        float *red = smem;                               // [RPP][nstat][BN]
        const int rg = tid / TPR;
        *(float4 *)(red + (rg * nstat + 0) * BN + col) = make_float4(s1[0], s1[1], s1[2], s1[3]);
        *(float4 *)(red + (rg * nstat + 1) * BN + col) = make_float4(s2[0], s2[1], s2[2], s2[3]);
        if (want_q)
            for (int k = 0; k < 3; ++k)
                *(float4 *)(red + (rg * nstat + 2 + k) * BN + col) = make_float4(qs[k][0], qs[k][1], qs[k][2], qs[k][3]);
        lds_barrier();
        for (int c = tid; c < nstat * BN; c += G_THREADS) {
            const int which = c / BN, cc = c % BN;
            float sum = 0.f;
#pragma unroll
            for (int g = 0; g < RPP; ++g) sum += red[(g * nstat + which) * BN + cc];
            // partial sums are consumed per 128-row tile (cmf_gemm_tiles_m): a 256-row tile stores its sums in the first of
            // its two slots and zeros in the second
            constexpr int SPT = BM > 128 ? BM / 128 : 1;
            if (n0 + cc < p.N) {
                p.stats[(((long long)tm * SPT) * nstat + which) * p.N + n0 + cc] = sum;
#pragma unroll
                for (int e = 1; e < SPT; ++e)
                    if ((long long)(tm * SPT + e) * 128 < p.M) p.stats[(((long long)tm * SPT + e) * nstat + which) * p.N + n0 + cc] = 0.f;
            }
        }
    }
    if (p.trace && tid == 0) {
        unsigned long long *r = p.trace + 8ull * blockIdx.x;
        r[0] = t_start; r[1] = t_main; r[2] = wall_clock64();
        r[3] = g_where(); r[4] = t_e[0]; r[5] = t_e[1]; r[6] = t_e[2]; r[7] = 0ull;
    }
}

// Sum split-K slabs: C[m,n] (+)= sum_s P[s][m][n]   (fixed order -> deterministic weight gradients)
__device__ __forceinline__ void splitk_reduce_body(long long total, int splits, int N, long long ldc, int accumulate,
                                                   const float *__restrict__ P, float *__restrict__ C)
{
    const bool v4 = (N % 4 == 0) && (ldc % 4 == 0);
    if (v4) {
        const long long total4 = total / 4;
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long long)gridDim.x * 256) {
            // 8 slab loads in flight per thread (a runtime-length loop of dependent load->add pairs is
            // latency bound: 42 us for 33 MB); the summation ORDER stays k = 0,1,2,... (deterministic)
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
            const float *src = P + i * 4;
            int k = 0;
            for (; k + 8 <= splits; k += 8) {
                float4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = *(const float4 *)(src + (long long)(k + u) * total);
#pragma unroll
                for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
            }
            for (; k < splits; ++k) {
                const float4 v = *(const float4 *)(src + (long long)k * total);
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
            const long long e = i * 4, m = e / N, n = e - m * N;
            float4 *dst = (float4 *)(C + m * ldc + n);
            if (accumulate) { const float4 o = *dst; s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w; }
            *dst = s;
        }
        return;
    }
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        float s = 0.f;
        for (int k = 0; k < splits; ++k) s += P[(long long)k * total + i];
        const long long m = i / N, n = i - m * N;
        float *dst = C + m * ldc + n;
        *dst = accumulate ? *dst + s : s;
    }
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(long long total, int splits, int N, long long ldc, int accumulate,
                                                            const float *__restrict__ P, float *__restrict__ C)
{
    splitk_reduce_body(total, splits, N, ldc, accumulate, P, C);
}

__global__ __launch_bounds__(256) void splitk_reduce_batch_kernel(const CmfBatch<CmfSplitkArgs> b)
{
    const CmfSplitkArgs &p = b.a[blockIdx.y];
    splitk_reduce_body((long long)p.M * p.N, p.split_k, p.N, p.ldc, p.accumulate, p.workspace, p.C);      // grid-stride
}

// The same sum for a SMALL output with MANY slabs (the thin layers' weight gradients: 32 x 32 ... 64 x 64 outputs, up to
// 1024 slabs).  One thread per output element walks all slabs serially -- 128 dependent rounds of loads for 512 threads,
// 38 us for 8 MB; here 16 lanes share an output float4, each sums every 16th slab (8 loads in flight), and the 16 partial
// sums are folded in lane order through LDS: a fixed order again, so the result stays deterministic.
constexpr int SKW_LANES = 16, SKW_OUT = 256 / SKW_LANES;
__device__ __forceinline__ void splitk_reduce_wide_body(long long total, int splits, int N, long long ldc, int accumulate,
                                                        const float *__restrict__ P, float *__restrict__ C)
{
    __shared__ float4 part[SKW_OUT][SKW_LANES + 1];
    const int o = threadIdx.x / SKW_LANES, l = threadIdx.x % SKW_LANES;
    const long long i = (long long)blockIdx.x * SKW_OUT + o;              // output float4
    const long long total4 = total / 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < total4) {
        const float *src = P + i * 4;
        int k = l;
        for (; k + 7 * SKW_LANES < splits; k += 8 * SKW_LANES) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *(const float4 *)(src + (long long)(k + u * SKW_LANES) * total);
#pragma unroll
            for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
        }
        for (; k < splits; k += SKW_LANES) {
            const float4 v = *(const float4 *)(src + (long long)k * total);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    part[o][l] = s;
    __syncthreads();
    if (l == 0 && i < total4) {
        float4 t = part[o][0];
        for (int u = 1; u < SKW_LANES; ++u) { const float4 v = part[o][u]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
        const long long e = i * 4, m = e / N, n = e - m * N;
        float4 *dst = (float4 *)(C + m * ldc + n);
        if (accumulate) { const float4 q = *dst; t.x += q.x; t.y += q.y; t.z += q.z; t.w += q.w; }
        *dst = t;
    }
}

__global__ __launch_bounds__(256) void splitk_reduce_wide_kernel(long long total, int splits, int N, long long ldc, int accumulate,
                                                                 const float *__restrict__ P, float *__restrict__ C)
{
    splitk_reduce_wide_body(total, splits, N, ldc, accumulate, P, C);
}

__global__ __launch_bounds__(256) void splitk_reduce_wide_batch_kernel(const CmfBatch<CmfSplitkArgs> b)
{
    const CmfSplitkArgs &p = b.a[blockIdx.y];
    if ((long long)blockIdx.x * SKW_OUT * 4 >= (long long)p.M * p.N) return;      // whole workgroup: before the barrier inside
    splitk_reduce_wide_body((long long)p.M * p.N, p.split_k, p.N, p.ldc, p.accumulate, p.workspace, p.C);
}

static bool splitk_wide_ok(long long total, int N, long long ldc, int split_k, const float *workspace, const float *C)
{
    return N % 4 == 0 && ldc % 4 == 0 && total <= 16384 && split_k >= 4 * SKW_LANES && (((uintptr_t)C | (uintptr_t)workspace) & 15) == 0;
}

// n <= CMF_MAX_BATCH slab sums in one launch (cmf_common.h "batched launches"); every problem must take the same kernel
int cmf_splitk_reduce_batch(int n, const CmfSplitkArgs *a, hipStream_t st)
{
    CMF_CHECK_ARG(n >= 1 && n <= CMF_MAX_BATCH && a);
    CmfBatch<CmfSplitkArgs> b;
    const bool wide = splitk_wide_ok((long long)a[0].M * a[0].N, a[0].N, a[0].ldc, a[0].split_k, a[0].workspace, a[0].C);
    long long tmax = 0;
    for (int i = 0; i < n; ++i) {
        const long long total = (long long)a[i].M * a[i].N;
        CMF_CHECK_ARG(a[i].workspace && a[i].C && a[i].split_k >= 1 && total > 0);
        CMF_CHECK_ARG(splitk_wide_ok(total, a[i].N, a[i].ldc, a[i].split_k, a[i].workspace, a[i].C) == wide);
        b.a[i] = a[i];
        tmax = std::max(tmax, total);
    }
    if (wide) hipLaunchKernelGGL(splitk_reduce_wide_batch_kernel, dim3((unsigned)((tmax / 4 + SKW_OUT - 1) / SKW_OUT), n), dim3(256), 0, st, b);
    else hipLaunchKernelGGL(splitk_reduce_batch_kernel, dim3((unsigned)std::min<long long>((tmax + 255) / 256, 4096), n), dim3(256), 0, st, b);
    return cmf_launch_status();
}

// C[M][N] (+)= the sum of `split_k` slabs [M][N] in slab order (internal; also used by thin_gemm.hip)
int cmf_splitk_reduce(int M, int N, int split_k, const float *workspace, float *C, long long ldc, int accumulate, hipStream_t st)
{
    const long long total = (long long)M * N;
    if (splitk_wide_ok(total, N, ldc, split_k, workspace, C)) {
        hipLaunchKernelGGL(splitk_reduce_wide_kernel, dim3((unsigned)((total / 4 + SKW_OUT - 1) / SKW_OUT)), dim3(256), 0, st,
                           total, split_k, N, ldc, accumulate, workspace, C);
        return cmf_launch_status();
    }
    const int grid = (int)std::min<long long>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(grid), dim3(256), 0, st, total, split_k, N, ldc, accumulate, workspace, C);
    return cmf_launch_status();
}

// ---- live timing of the tiled kernel (bench.py's `roofline` object) ------------------------------------------------
// Between cmf_gemm_profile_begin and _end every launch of gemm_kernel whose 2*M*N*K reaches `min_flops` is bracketed by
// a HIP event pair recorded on the stream the kernel is launched on -- inside the library, so the launches issued from
// cmf_setconv_forward / _backward (on side streams, from the block calls' host threads) are covered like the ones a
// caller issues directly.  The FLOPs of ALL launches (thin kernels included) are counted next to the bracketed ones,
// so the caller can state which share of the work the measured launches carry.
#include <mutex>
#include <vector>
namespace {
struct GemmProfile {
    std::mutex mu;
    bool on = false;
    double min_flops = 0.0, flops_timed = 0.0, flops_all = 0.0, flops_eligible = 0.0;
    long long launches_all = 0, launches_eligible = 0;
    int sample_every = 1;                            // bracket every n-th eligible launch (cmf_gemm_profile_sampling)
    size_t used = 0;
    std::vector<hipEvent_t> events;                  // pairs: [2i] before, [2i+1] after; kept across sessions
    std::vector<cmf_gemm_launch_record> recs;        // one per bracketed launch (shape, layout, epilogue kind)
};
GemmProfile g_gprof;

// -> index of the event pair to record around this launch, or -1
long long gprof_open(double flops, hipStream_t st, const GemmArgs *a = nullptr, int layout = 0, int bm = 0, int bn = 0)
{
    if (!g_gprof.on) return -1;
    std::lock_guard<std::mutex> lock(g_gprof.mu);
    if (!g_gprof.on) return -1;
    g_gprof.flops_all += flops;
    ++g_gprof.launches_all;
    if (flops < g_gprof.min_flops) return -1;
    g_gprof.flops_eligible += flops;
    if ((g_gprof.launches_eligible++ % g_gprof.sample_every) != 0) return -1;       // counted, not bracketed
    if (g_gprof.used * 2 == g_gprof.events.size()) {
        hipEvent_t a = nullptr, b = nullptr;
        // timing events without the system-scope release / acquire a default event carries: bracketing a launch must not flush caches
        // around it (measured: the bracketed timed region ran 0.15-0.2 ms per step slower than the unbracketed regions behind it)
        if (hipEventCreateWithFlags(&a, hipEventDisableSystemFence) != hipSuccess || hipEventCreateWithFlags(&b, hipEventDisableSystemFence) != hipSuccess) {
            (void)hipGetLastError(); return -1;
        }
        g_gprof.events.push_back(a); g_gprof.events.push_back(b);
    }
    const size_t i = g_gprof.used++;
    g_gprof.flops_timed += flops;
    if (g_gprof.recs.size() <= i) g_gprof.recs.resize(i + 1);
    cmf_gemm_launch_record &r = g_gprof.recs[i];
    r.M = a ? a->M : 0; r.N = a ? a->N : 0; r.K = a ? a->K : 0; r.layout = layout; r.split_k = a ? a->split_k : 0;
    r.kind = a ? epilogue_kind(*a) : 0; r.bm = bm; r.bn = bn; r.ms = 0.f;
    (void)hipEventRecord(g_gprof.events[2 * i], st);
    return (long long)i;
}
void gprof_count(double flops)
{
    if (!g_gprof.on) return;
    std::lock_guard<std::mutex> lock(g_gprof.mu);
    g_gprof.flops_all += flops;
    ++g_gprof.launches_all;
}
void gprof_close(long long i, hipStream_t st)
{
    if (i < 0) return;
    std::lock_guard<std::mutex> lock(g_gprof.mu);
    (void)hipEventRecord(g_gprof.events[2 * (size_t)i + 1], st);
}
}  // namespace
void cmf_gemm_count_flops(double flops) { gprof_count(flops); }     // thin_gemm.hip's fused layer (counted, never bracketed)

// ---- diagnostics: per-workgroup timeline of the NEXT tiled launch (tools/gemm_timeline.py) -------------------------
namespace {
unsigned long long *g_trace_buf = nullptr;
long long g_trace_cap = 0, g_trace_n = 0;
bool g_trace_armed = false;
}
extern "C" int cmf_gemm_trace_arm(void) { g_trace_armed = true; return 0; }
// The record buffer (8 x u64 per workgroup, zeroed on the launch's stream) if the next launch is to be traced, else NULL
static unsigned long long *trace_take(unsigned grid, hipStream_t st)
{
    if (!g_trace_armed) return nullptr;
    g_trace_armed = false;                               // diagnostics only: one launch, single-threaded use
    if ((long long)grid > g_trace_cap) {
        if (g_trace_buf) (void)hipFree(g_trace_buf);
        g_trace_cap = grid;
        if (hipMalloc((void **)&g_trace_buf, (size_t)g_trace_cap * 64) != hipSuccess) { g_trace_buf = nullptr; g_trace_cap = 0; }
    }
    if (!g_trace_buf) return nullptr;
    (void)hipMemsetAsync(g_trace_buf, 0, (size_t)grid * 64, st);
    g_trace_n = grid;
    return g_trace_buf;
}
// Copies the records of the traced launch (8 x u64 per workgroup) to host memory; returns the workgroup count.
extern "C" long long cmf_gemm_trace_read(unsigned long long *host_out, long long max_workgroups)
{
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    const long long n = g_trace_n < max_workgroups ? g_trace_n : max_workgroups;
    if (n > 0 && host_out && hipMemcpy(host_out, g_trace_buf, (size_t)n * 64, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return g_trace_n;
}

extern "C" int cmf_gemm_profile_begin(double min_flops)
{
    std::lock_guard<std::mutex> lock(g_gprof.mu);
    g_gprof.on = true; g_gprof.min_flops = min_flops; g_gprof.flops_timed = g_gprof.flops_all = 0.0;
    g_gprof.launches_all = 0; g_gprof.used = 0; g_gprof.flops_eligible = 0.0; g_gprof.launches_eligible = 0;
    return 0;
}

// Bracket only every n-th launch that reaches min_flops (default 1: all of them).  An event pair is two barrier packets on the launch
// stream: around every one of a training step's 43 large GEMM launches they cost 0.13-0.18 ms per step (bench.py: the bracketed region
// against the unbracketed regions behind it); every 4th keeps the per-launch figures and a quarter of that.  All eligible launches are
// still counted (cmf_gemm_profile_eligible).  Process-wide; call before cmf_gemm_profile_begin.
extern "C" int cmf_gemm_profile_sampling(int every)
{
    CMF_CHECK_ARG(every >= 1);
    std::lock_guard<std::mutex> lock(g_gprof.mu);
    g_gprof.sample_every = every;
    return 0;
}

// launches >= min_flops of the last window (bracketed or not) and their 2*M*N*K sum
extern "C" int cmf_gemm_profile_eligible(long long *launches, double *flops)
{
    std::lock_guard<std::mutex> lock(g_gprof.mu);
    if (launches) *launches = g_gprof.launches_eligible;
    if (flops) *flops = g_gprof.flops_eligible;
    return 0;
}

// Synchronises the device, sums the bracketed durations.  Any output pointer may be NULL.
extern "C" int cmf_gemm_profile_end(long long *launches_timed, double *ms_timed, double *flops_timed, long long *launches_all,
                                    double *flops_all)
{
    std::lock_guard<std::mutex> lock(g_gprof.mu);
    g_gprof.on = false;
    if (hipDeviceSynchronize() != hipSuccess) return (int)hipGetLastError();
    double ms = 0.0;
    for (size_t i = 0; i < g_gprof.used; ++i) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, g_gprof.events[2 * i], g_gprof.events[2 * i + 1]) != hipSuccess) return (int)hipGetLastError();
        ms += t;
        g_gprof.recs[i].ms = t;
    }
    if (launches_timed) *launches_timed = (long long)g_gprof.used;
    if (ms_timed) *ms_timed = ms;
    if (flops_timed) *flops_timed = g_gprof.flops_timed;
    if (launches_all) *launches_all = g_gprof.launches_all;
    if (flops_all) *flops_all = g_gprof.flops_all;
    return 0;
}

// The records of the bracketed launches of the last closed window (call after cmf_gemm_profile_end); returns their number.
extern "C" long long cmf_gemm_profile_records(cmf_gemm_launch_record *out, long long max_records)
{
    std::lock_guard<std::mutex> lock(g_gprof.mu);
    const long long n = (long long)g_gprof.used < max_records ? (long long)g_gprof.used : max_records;
    for (long long i = 0; i < n && out; ++i) out[i] = g_gprof.recs[(size_t)i];
    return (long long)g_gprof.used;
}

template <int BM, int BN, bool A_T, bool B_T, int EPI = 0, int GMODE = 0>
static int launch(const GemmArgs &a, hipStream_t st)
{
    constexpr bool GATHER = GMODE == 1;
    const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + BN - 1) / BN;
    const int per = (tiles_m + 7) / 8;
    dim3 grid(a.split_k > 1 ? 8 * ((a.split_k + 7) / 8) * tiles_m * tiles_n : 8 * per * tiles_n);
    const size_t lds_reg = (size_t)2 * ((A_T ? G_BK * (BM + 4) : BM * G_LDS_LD) + (B_T ? BN * G_LDS_LD : G_BK * (BN + 4))) * sizeof(float);
    const size_t lds_dir = (size_t)G_STAGES * ((BM + BN) * (G_BK / 4) * 4 + (GATHER ? 96 : 32)) * sizeof(float);
    const size_t lds = lds_reg > lds_dir ? lds_reg : lds_dir;
    // the dynamic-LDS limit is a per-device attribute of the function: set once per (instantiation, device); cmf_gemm is
    // entered concurrently by the host threads of cmf_setconv_*_multi, hence the atomics
    static std::atomic<unsigned> set_mask[4];
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned bit = 1u << (dev & 31);
    if (dev >= 128 || !(set_mask[dev >> 5].load(std::memory_order_acquire) & bit)) {
        if (hipFuncSetAttribute((const void *)gemm_kernel<BM, BN, A_T, B_T, EPI, GMODE>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess) return (int)hipGetLastError();
        if (dev < 128) set_mask[dev >> 5].fetch_or(bit, std::memory_order_release);
    }
    const long long pe = gprof_open(2.0 * a.M * a.N * a.K, st, &a, (A_T ? 2 : 0) | (B_T ? 1 : 0), BM, BN);
    if (unsigned long long *tb = trace_take(grid.x, st)) {
        GemmArgs t = a;
        t.trace = tb;
        hipLaunchKernelGGL((gemm_kernel<BM, BN, A_T, B_T, EPI, GMODE>), grid, dim3(G_THREADS), lds, st, t);
        gprof_close(pe, st);
        return cmf_launch_status();
    }
    hipLaunchKernelGGL((gemm_kernel<BM, BN, A_T, B_T, EPI, GMODE>), grid, dim3(G_THREADS), lds, st, a);
    gprof_close(pe, st);
    return cmf_launch_status();
}

// timing-only ablation bits of the diagnostics builds (-DCMF_GEMM_DIAG; results invalid): read once
static int gemm_diag_rt()
{
    static const int v = getenv("CMF_GEMM_DIAG_RT") ? atoi(getenv("CMF_GEMM_DIAG_RT")) : 0;
    return v;
}

// 256 x 128 tiles for tall interior shapes (env CMF_GEMM_TALL=0: the 128 x 128 tiles everywhere, A/B)
static int gemm_tall_mode()
{
    static const int mode = getenv("CMF_GEMM_TALL") ? atoi(getenv("CMF_GEMM_TALL")) : 1;
    return mode;
}

// 128 x 256 tiles for the gathering forward GEMM (env CMF_GEMM_WIDE=0: 128 x 128, A/B; 2: also below two workgroups per CU, tests)
static int gemm_wide_mode()
{
    static const int mode = getenv("CMF_GEMM_WIDE") ? atoi(getenv("CMF_GEMM_WIDE")) : 1;
    return mode;
}

extern "C" int cmf_gemm(int M, int N, int K, int a_t, int b_t,
                        const float *A, long long lda, const float *B, long long ldb, float *C, long long ldc,
                        const float *pro_a, const float *pro_c, const float *prob_a, const float *prob_c,
                        const float *bias, int act, float *stats,
                        int bwd_mode, const float *Z, long long ldz,
                        const float *ea, const float *ec, const float *emean, const float *einvstd,
                        const float *dxyz, int split_k, float *workspace, int accumulate, void *stream)
{
    CMF_CHECK_ARG(M >= 0 && N >= 0 && K >= 0 && split_k >= 1);
    if (M == 0 || N == 0) return 0;
    CMF_CHECK_ARG(A && B && C);
    CMF_CHECK_ARG(!(a_t && pro_a));                                  // A prologue needs the [M][K] layout
    CMF_CHECK_ARG(!(b_t && prob_a));
    CMF_CHECK_ARG(split_k == 1 || (workspace && !stats && !bwd_mode && !bias && !act));
    CMF_CHECK_ARG(!bwd_mode || (!bias && !act));                     // a backward epilogue is mask * accumulator
    // 16-byte vector loads need aligned rows
    CMF_CHECK_ARG(lda % 4 == 0 && ldb % 4 == 0 && ((uintptr_t)A % 16 == 0) && ((uintptr_t)B % 16 == 0));
    CMF_CHECK_ARG(!pro_a || (((uintptr_t)pro_a | (uintptr_t)pro_c) % 16 == 0));
    hipStream_t st = (hipStream_t)stream;
    GemmArgs g{};
    g.M = M; g.N = N; g.K = K; g.A = A; g.lda = lda; g.B = B; g.ldb = ldb;
    g.C = split_k > 1 ? workspace : C; g.ldc = split_k > 1 ? N : ldc;
    g.pro_a = pro_a; g.pro_c = pro_c; g.prob_a = prob_a; g.prob_c = prob_c;
    g.bias = bias; g.act = act; g.stats = stats; g.bwd_mode = bwd_mode; g.Z = Z; g.ldz = ldz;
    g.ea = ea; g.ec = ec; g.emean = emean; g.einvstd = einvstd; g.dxyz = dxyz; g.split_k = split_k;
    g.accumulate = split_k > 1 ? 0 : accumulate;
    static const int no_direct = (getenv("CMF_GEMM_NO_DIRECT") && getenv("CMF_GEMM_NO_DIRECT")[0] == '1') ? 1 : 0;
    // weight-gradient layout (both operands contraction-major) with the producer's BN + ReLU applied to B: the register-staged
    // loop applies it once per staged element, the LDS-direct loop on every wave's fragments inside the MFMA stream --
    // measured in the step, isolated: 256 x 512 x 524288 / 262144 / 131072 at 117.6 / 117.9 / 114.4 TF staged against
    // 111.7 / 110.3 / 108.7 direct; the short contractions (K = 16384) lose 2-3 % and stay direct.
    // (when the persistent kernel takes the call -- below -- the prologue sits between single MFMAs and costs nothing)
    const bool staged_dw = a_t && !b_t && prob_a && K >= 32768;
    g.no_direct = no_direct ? 1 : 0;
    static const int diag_rt = getenv("CMF_GEMM_DIAG_RT") ? atoi(getenv("CMF_GEMM_DIAG_RT")) : 0;
    g.diag = diag_rt;
    g.trace = nullptr;
    int err = cmf_thin_gemm(g, a_t, b_t, st);            // narrow layers (<= 64 channels): barrier-free per-wave kernels
    if (err > 0) return err;
    const bool thin_done = (err == 0);
    if (thin_done) gprof_count(2.0 * M * N * K);         // thin kernels: counted in flops_all, never bracketed
    const bool wide = N > 64, tall = M > 64;
    // instantiated (layout, kind) pairs: forward GEMMs (A[M][K], W[N][K]) carry kind 0 / 1, data gradients (dZ[M][N], W[N][K])
    // kind 0 / 2 / 3, weight gradients and the rest kind 0; any other pairing runs the kind-0 kernel, whose generic loop
    // handles every epilogue
    const int kind = epilogue_kind(g);
    // 256 x 128 tiles (wave tile 128 x 64, 2 workgroups per CU): half the operand bytes per flop through L2 and LDS-direct, a quarter
    // fewer LDS fragment reads, tiles twice as long against the same prologue / epilogue (round 5; tools/lab/gemm_lab.hip: + 5-20 % on the
    // model's plain shapes).  Interior shapes only (the LDS-direct loop); enough tiles to fill 2 workgroups per CU.
    const int tall_mode = gemm_tall_mode();
    const long long work_tiles = (long long)(M / 256) * ((N + 127) / 128) * g.split_k;
#define CMF_PICK(AT, BT, EP)                                                                               \
    (big ? launch<256, 128, AT, BT, EP>(g, st) :                                                          \
     tall ? (wide ? launch<128, 128, AT, BT, EP>(g, st) : launch<128, 64, AT, BT, EP>(g, st)) \
          : (wide ? launch<64, 128, AT, BT, EP>(g, st) : launch<64, 64, AT, BT, EP>(g, st)))
    const int pgrid = thin_done ? 0 : cmf_pgemm_grid(g, a_t, b_t, kind);
    if (!pgrid && staged_dw) g.no_direct = 1;
    const bool big = tall_mode && (!a_t || tall_mode == 3) && M % 256 == 0 && N % 128 == 0 && K % G_BK == 0 && work_tiles >= 384 && !g.no_direct && !(accumulate && g.split_k == 1);
    if (thin_done)         err = 0;
    else if (pgrid) {
        const long long pe = gprof_open(2.0 * M * N * K, st, &g, (a_t ? 2 : 0) | (b_t ? 1 : 0) | 4, 128, 128);      // layout bit 2: persistent kernel
        g.trace = trace_take((unsigned)pgrid, st);
        err = cmf_pgemm_launch(g, a_t, b_t, kind, pgrid, st);
        gprof_close(pe, st);
    }
    else if (!a_t && b_t)  err = kind == 1 ? CMF_PICK(false, true, 1) : CMF_PICK(false, true, 0);
    else if (!a_t && !b_t) err = kind == 2 ? CMF_PICK(false, false, 2) : kind == 3 ? CMF_PICK(false, false, 3) :
                                 kind == 4 ? CMF_PICK(false, false, 4) : kind == 5 ? CMF_PICK(false, false, 5) : CMF_PICK(false, false, 0);
    else if (a_t && !b_t)  err = CMF_PICK(true, false, 0);
    else                   err = CMF_PICK(true, true, 0);
#undef CMF_PICK
    if (err) return err;
    if (split_k > 1) return cmf_splitk_reduce(M, N, g.split_k, workspace, C, ldc, accumulate, st);     // (the persistent kernel may have lowered it)
    return 0;
}

// Forward GEMM of the set-conv layer BEHIND the hoisted first conv, with that first layer formed in the A-operand path instead of
// being materialised (inference: nothing of it is needed afterwards):
//   C[m][n] = sum_k relu( pro_a[k] * ( Y[rows[m]][k] + wx3[0][k] dx_m + wx3[1][k] dy_m + wx3[2][k] dz_m ) + pro_c[k] ) * W[n][k]
// Y (.., K) per-point rows of pitch ldy, rows[M] the source row of every neighbour slot, dxyz (M,4) its relative coordinates,
// wx3 (3,K) the coordinate columns of the first conv as planes.  The same operations, in the same order, as cmf_group_affine
// followed by cmf_gemm with the A prologue: bit-identical output, without the (M,K) tensor (2 GB at the second encoder's scales)
// written and read back.  stats (optional): the output's BN partial sums like cmf_gemm's.  M, N multiples of 128, K of 16; all
// pointers 16-byte aligned.
extern "C" int cmf_gemm_gather_affine(int M, int N, int K, const float *Y, long long ldy, const int *rows, const float *dxyz,
                                      const float *wx3, const float *pro_a, const float *pro_c, const float *W, long long ldw,
                                      float *C, long long ldc, float *stats, void *stream)
{
    CMF_CHECK_ARG(M > 0 && N > 0 && K > 0 && M % 128 == 0 && N % 128 == 0 && K % G_BK == 0);
    CMF_CHECK_ARG(Y && rows && dxyz && wx3 && pro_a && pro_c && W && C && ldy % 4 == 0 && ldw % 4 == 0 && ldc % 4 == 0);
    CMF_CHECK_ARG((((uintptr_t)Y | (uintptr_t)dxyz | (uintptr_t)wx3 | (uintptr_t)pro_a | (uintptr_t)pro_c | (uintptr_t)W | (uintptr_t)C) & 15) == 0);
    hipStream_t st = (hipStream_t)stream;
    GemmArgs g{};
    g.M = M; g.N = N; g.K = K; g.A = Y; g.lda = ldy; g.B = W; g.ldb = ldw; g.C = C; g.ldc = ldc;
    g.pro_a = pro_a; g.pro_c = pro_c; g.ga_rows = rows; g.ga_dxyz = dxyz; g.ga_wx = wx3; g.split_k = 1;
    g.stats = stats;                                    // train mode: [M / 128][2][N] partial sums of the output
    g.diag = gemm_diag_rt();
    // (256 x 128 tiles measured here, round 5: 111 against 118 TF -- the per-row state of the gathering loads and the fragment prologue
    //  do not fit 256 registers beside a 128 x 64 wave tile; CMF_GEMM_TALL=2 keeps that form reachable)
    if (gemm_tall_mode() == 2 && M % 256 == 0 && (long long)(M / 256) * (N / 128) >= 384)
        return stats ? launch<256, 128, false, true, 1, 1>(g, st) : launch<256, 128, false, true, 0, 1>(g, st);
    // 128 x 256 tiles (wave tile 64 x 128): the fragment arithmetic of the gathered operand -- 16 elements per lane and chunk either
    // way -- then sits beside 64 MFMAs instead of 32, and a row panel is gathered once for 256 output columns
    if (gemm_wide_mode() && N % 256 == 0 && ((long long)(M / 128) * (N / 256) >= 512 || gemm_wide_mode() == 2))
        return stats ? launch<128, 256, false, true, 1, 1>(g, st) : launch<128, 256, false, true, 0, 1>(g, st);
    return stats ? launch<128, 128, false, true, 1, 1>(g, st) : launch<128, 128, false, true, 0, 1>(g, st);
}

// Data gradient through the layer behind the hoisted first conv, masked by the first layer's BN + ReLU and with its BN-backward and
// dxyz partial sums (cmf_gemm's backward kind with dxyz), where the first layer's pre-activations are formed from the per-point rows
// in the epilogue instead of being read back:   dU[m][k] = (dZ @ W)[m][k] * [ea[k] z + ec[k] > 0],  z = Y[rows[m]][k] + wx3[:,k] . dxyz[m]
// -- bit-identical to cmf_gemm(bwd_mode 1, Z = the materialised tensor, dxyz) with the non-persistent kernel; statistics
// [tiles_m][5][cin].  M, cin multiples of 128, cout of 16.
extern "C" int cmf_gemm_dx_gather(int M, int cin, int cout, const float *dZ, long long ldz, const float *W, long long ldw,
                                  float *dU, long long ldu, const float *Y, long long ldy, const int *rows, const float *dxyz,
                                  const float *wx3, const float *ea, const float *ec, const float *emean, const float *einvstd,
                                  float *stats, void *stream)
{
    CMF_CHECK_ARG(M > 0 && cin > 0 && cout > 0 && M % 128 == 0 && cin % 128 == 0 && cout % G_BK == 0);
    CMF_CHECK_ARG(dZ && W && dU && Y && rows && dxyz && wx3 && ea && ec && emean && einvstd && stats);
    CMF_CHECK_ARG(ldz % 4 == 0 && ldw % 4 == 0 && ldu % 4 == 0 && ldy % 4 == 0);
    CMF_CHECK_ARG((((uintptr_t)dZ | (uintptr_t)W | (uintptr_t)dU | (uintptr_t)Y | (uintptr_t)dxyz) & 15) == 0);
    hipStream_t st = (hipStream_t)stream;
    GemmArgs g{};
    g.M = M; g.N = cin; g.K = cout; g.A = dZ; g.lda = ldz; g.B = W; g.ldb = ldw; g.C = dU; g.ldc = ldu;
    g.stats = stats; g.bwd_mode = 1; g.Z = Y; g.ldz = ldy; g.ea = ea; g.ec = ec; g.emean = emean; g.einvstd = einvstd; g.dxyz = dxyz;
    g.ga_rows = rows; g.ga_dxyz = dxyz; g.ga_wx = wx3; g.split_k = 1;
    g.diag = gemm_diag_rt();
    return launch<128, 128, false, false, 4, 3>(g, st);
}

// The same data gradient when only its per-source-point sums are needed (the grouping's backward pass sums it over the slots that
// reference a point): the rows walk the slots in inverse-index order -- arows[m] = the slot (row of dZ) at position m, pts[m] = its
// source point (row of Y), dxyz2 (M,4) its relative coordinates, all three in that order -- and instead of dU the kernel writes, for
// every run of equal source points inside a 64-row range, the run's column sums to pieces[(point + m / 64)][cin]
// (P + M / 64 rows suffice).  The sum of a point's pieces, in range order, equals the sum of cmf_gemm_dx_gather's rows over its slots
// up to fp32 association; the five statistics are the same sums in another order.
// internal (setconv_block.hip): the number of source points behind the next cmf_gemm_dx_gather_sum call of this host thread (0 = unknown)
static thread_local long long g_dxsum_points_hint = 0;
void cmf_gemm_dx_gather_sum_hint(long long points) { g_dxsum_points_hint = points; }

extern "C" int cmf_gemm_dx_gather_sum(int M, int cin, int cout, const float *dZ, long long ldz, const float *W, long long ldw,
                                      const float *Y, long long ldy, const int *arows, const int *pts, const float *dxyz2,
                                      const float *wx3, const float *ea, const float *ec, const float *emean, const float *einvstd,
                                      float *pieces, float *stats, void *stream)
{
    CMF_CHECK_ARG(M > 0 && cin > 0 && cout > 0 && M % 128 == 0 && cin % 128 == 0 && cout % G_BK == 0);
    CMF_CHECK_ARG(dZ && W && Y && arows && pts && dxyz2 && wx3 && ea && ec && emean && einvstd && pieces && stats);
    CMF_CHECK_ARG(ldz % 4 == 0 && ldw % 4 == 0 && ldy % 4 == 0);
    CMF_CHECK_ARG((((uintptr_t)dZ | (uintptr_t)W | (uintptr_t)Y | (uintptr_t)dxyz2) & 15) == 0);
    hipStream_t st = (hipStream_t)stream;
    GemmArgs g{};
    g.M = M; g.N = cin; g.K = cout; g.A = dZ; g.lda = ldz; g.B = W; g.ldb = ldw; g.C = pieces; g.ldc = cin;
    g.stats = stats; g.bwd_mode = 1; g.Z = Y; g.ldz = ldy; g.ea = ea; g.ec = ec; g.emean = emean; g.einvstd = einvstd; g.dxyz = dxyz2;
    g.ga_rows = pts; g.ga_dxyz = dxyz2; g.ga_wx = wx3; g.ga_arows = arows; g.ga_pieces = pieces; g.split_k = 1;
    // every source point has at least one slot, so a point index is below M; the set-conv blocks pass theirs (B * N points: cmf_gemm_dx_gather_sum_hint)
    const long long npts = g_dxsum_points_hint > 0 ? g_dxsum_points_hint : (long long)M;
    g.ga_y32 = (npts * ldy + cin) * 4 < (1ll << 32) ? 1 : 0;
    g_dxsum_points_hint = 0;                            // (a hint serves one call)
    g.diag = gemm_diag_rt();
    // (256-row tiles -- wave tile 128 x 64 = two 64-row ranges of `pieces` -- measured here, round 5: 99.7 against 105 TF at 524288 rows:
    //  the serial row walk of the epilogue doubles per wave while only two workgroups per CU are left to cover it; CMF_GEMM_TALL=2: A/B)
    if (gemm_tall_mode() == 2 && M % 256 == 0 && (long long)(M / 256) * (cin / 128) >= 384) return launch<256, 128, false, false, 4, 4>(g, st);
    return launch<128, 128, false, false, 4, 4>(g, st);
}

// The split count that fills the chip with the 256 x 128 tiles of cmf_gemm_dw_gather (two workgroups per CU: 512 slabs x tiles, slabs
// dealt to the 8 XCDs); 0 = no preference (shape not tileable that way, or CMF_GEMM_TALL=0): the caller's usual choice.
extern "C" int cmf_gemm_dw_gather_split(int cout, int cin, long long nrows)
{
    if (!gemm_tall_mode() || cout <= 0 || cin <= 0 || cout % 256 || cin % 128 || nrows % G_BK) return 0;
    const long long tiles = (long long)(cout / 256) * (cin / 128), chunks = nrows / G_BK;
    int s = (int)(512 / tiles) / 8 * 8;
    while (s >= 8 && chunks / s < 8) s -= 8;
    return s >= 16 ? s : 0;
}

// Weight gradient of the set-conv layer behind the hoisted first conv with that first layer formed in the B-operand staging instead
// of being read back:   dW[cout][cin] (+)= sum_r dZ[r][cout] * relu( prob_a[k] * ( Y[rows[r]][k] + wx3[:,k] . dxyz[r] ) + prob_c[k] )
// -- the same operations in the same order as cmf_gemm(a_t = 1, b_t = 0, prob) on the materialised tensor in the register-staged
// loop (bit-identical).  cout, cin multiples of 128, nrows a multiple of 16.
extern "C" int cmf_gemm_dw_gather(int cout, int cin, long long nrows, const float *dZ, long long ldz, const float *Y, long long ldy,
                                  const int *rows, const float *dxyz, const float *wx3, const float *prob_a, const float *prob_c,
                                  float *dW, long long lddw, int split_k, float *workspace, int accumulate, void *stream)
{
    CMF_CHECK_ARG(cout > 0 && cin > 0 && nrows > 0 && nrows < (1ll << 31) && split_k >= 1);
    CMF_CHECK_ARG(cout % 128 == 0 && cin % 128 == 0 && nrows % G_BK == 0);
    CMF_CHECK_ARG(dZ && Y && rows && dxyz && wx3 && prob_a && prob_c && dW && (split_k == 1 || workspace));
    CMF_CHECK_ARG(ldz % 4 == 0 && ldy % 4 == 0 && lddw % 4 == 0);
    CMF_CHECK_ARG((((uintptr_t)dZ | (uintptr_t)Y | (uintptr_t)dxyz | (uintptr_t)wx3 | (uintptr_t)dW) & 15) == 0);
    hipStream_t st = (hipStream_t)stream;
    GemmArgs g{};
    g.M = cout; g.N = cin; g.K = (int)nrows; g.A = dZ; g.lda = ldz; g.B = Y; g.ldb = ldy;
    g.C = split_k > 1 ? workspace : dW; g.ldc = split_k > 1 ? cin : lddw;
    g.prob_a = prob_a; g.prob_c = prob_c; g.split_k = split_k; g.accumulate = split_k > 1 ? 0 : accumulate;
    g.ga_rows = rows; g.ga_dxyz = dxyz; g.ga_wx = wx3; g.no_direct = 1;
    // 256 x 128 tiles when the caller took the split count that goes with them (cmf_gemm_dw_gather_split): the gathered rows are
    // formed once for 256 output rows and a staged element feeds 64 MFMAs per wave instead of 32 -- 110-114 TF against 100-108 at the
    // second encoder's scales (tools/dwg_probe.py); the slab partition differs from the 128 x 128 form's, the sums inside a slab do not
    const bool tall = split_k > 1 && split_k == cmf_gemm_dw_gather_split(cout, cin, nrows);
    const int err = tall ? launch<256, 128, true, false, 0, 2>(g, st) : launch<128, 128, true, false, 0, 2>(g, st);
    if (err) return err;
    if (split_k > 1) return cmf_splitk_reduce(cout, cin, split_k, workspace, dW, lddw, accumulate, st);
    return 0;
}

// cmf_gemm_dw_gather with the train-mode BatchNorm backward of the output gradient formed while the A operand is staged (the bnb_*
// path of cmf_gemm_dw_bn_bwd) -- BOTH operands of the second encoder's largest weight gradient are then formed in the staging loop:
//   dZ = al * dU + be * (Z - mean) + ga   (written to dZ_out for the data-gradient GEMM),   B = relu(prob (Y[rows] + wx3 . dxyz))
// and the stand-alone cmf_bn_bwd_apply pass over (rows, cout) -- read dU, read Z, write dZ -- disappears.  Same operations in the same
// order as cmf_bn_bwd_apply followed by cmf_gemm_dw_gather: bit-identical results.
extern "C" int cmf_gemm_dw_gather_bn_bwd(int cout, int cin, long long nrows, const float *dU, long long ldu, const float *Z, long long ldz,
                                         const float *a, const float *mean, const float *invstd, const float *sums, float *dZ_out, long long ldo,
                                         const float *Y, long long ldy, const int *rows, const float *dxyz, const float *wx3,
                                         const float *prob_a, const float *prob_c, float *dW, long long lddw, int split_k, float *workspace,
                                         int accumulate, void *stream)
{
    CMF_CHECK_ARG(cout > 0 && cin > 0 && nrows > 0 && nrows < (1ll << 31) && split_k >= 1);
    CMF_CHECK_ARG(cout % 128 == 0 && cin % 128 == 0 && nrows % G_BK == 0);
    CMF_CHECK_ARG(dU && Z && a && mean && invstd && sums && dZ_out && dZ_out != dU && Y && rows && dxyz && wx3 && prob_a && prob_c && dW && (split_k == 1 || workspace));
    CMF_CHECK_ARG(ldu % 4 == 0 && ldz % 4 == 0 && ldo % 4 == 0 && ldy % 4 == 0 && lddw % 4 == 0);
    CMF_CHECK_ARG((((uintptr_t)dU | (uintptr_t)Z | (uintptr_t)dZ_out | (uintptr_t)Y | (uintptr_t)dxyz | (uintptr_t)wx3 | (uintptr_t)dW) & 15) == 0);
    hipStream_t st = (hipStream_t)stream;
    GemmArgs g{};
    g.M = cout; g.N = cin; g.K = (int)nrows; g.A = dU; g.lda = ldu; g.B = Y; g.ldb = ldy;
    g.C = split_k > 1 ? workspace : dW; g.ldc = split_k > 1 ? cin : lddw;
    g.prob_a = prob_a; g.prob_c = prob_c; g.split_k = split_k; g.accumulate = split_k > 1 ? 0 : accumulate;
    g.ga_rows = rows; g.ga_dxyz = dxyz; g.ga_wx = wx3; g.no_direct = 1;
    g.bnb_z = Z; g.ldbz = ldz; g.bnb_a = a; g.bnb_mean = mean; g.bnb_invstd = invstd; g.bnb_sums = sums;
    g.bnb_ic = (float)(1.0 / (double)nrows); g.bnb_out = dZ_out; g.ldbo = ldo;
    const bool tall = split_k > 1 && split_k == cmf_gemm_dw_gather_split(cout, cin, nrows);
    const int err = tall ? launch<256, 128, true, false, 0, 2>(g, st) : launch<128, 128, true, false, 0, 2>(g, st);
    if (err) return err;
    if (split_k > 1) return cmf_splitk_reduce(cout, cin, split_k, workspace, dW, lddw, accumulate, st);
    return 0;
}

// Weight gradient of a layer whose output gradient still has to go through the train-mode BatchNorm backward:
//   dZ = al * dU + be * (Z - mean) + ga  (cmf_common.h cmf_bnb_coef of a / mean / invstd / sums, 1 / rows),   dW (+)= dZ^T @ act(X)
// formed while the A operand is staged -- the stand-alone cmf_bn_bwd_apply pass (read dU, read Z, write dZ) disappears and dZ
// is written to dZ_out (a buffer of its own: other workgroups still read dU) as a by-product for the data-gradient GEMM that
// follows.  Same operations as cmf_bn_bwd_apply + cmf_gemm(a_t = 1, b_t = 0) on the register-staged loop: bit-identical
// results.  cout, cin multiples of 128, rows a multiple of 16; X activated by (prob_a, prob_c) or as stored (both NULL).
extern "C" int cmf_gemm_dw_bn_bwd(int cout, int cin, long long rows, const float *dU, long long ldu, const float *Z, long long ldz,
                                  const float *a, const float *mean, const float *invstd, const float *sums, float *dZ_out, long long ldo,
                                  const float *X, long long ldx, const float *prob_a, const float *prob_c,
                                  float *dW, long long lddw, int split_k, float *workspace, int accumulate, void *stream)
{
    CMF_CHECK_ARG(cout > 0 && cin > 0 && rows > 0 && rows < (1ll << 31) && split_k >= 1);
    CMF_CHECK_ARG(cout % 128 == 0 && cin % 128 == 0 && rows % G_BK == 0);
    CMF_CHECK_ARG(dU && Z && a && mean && invstd && sums && X && dW && (split_k == 1 || workspace));
    CMF_CHECK_ARG(ldu % 4 == 0 && ldz % 4 == 0 && ldx % 4 == 0 && (!dZ_out || ldo % 4 == 0));
    CMF_CHECK_ARG((((uintptr_t)dU | (uintptr_t)Z | (uintptr_t)X | (uintptr_t)dZ_out) & 15) == 0);
    hipStream_t st = (hipStream_t)stream;
    GemmArgs g{};
    g.M = cout; g.N = cin; g.K = (int)rows; g.A = dU; g.lda = ldu; g.B = X; g.ldb = ldx;
    g.C = split_k > 1 ? workspace : dW; g.ldc = split_k > 1 ? cin : lddw;
    g.prob_a = prob_a; g.prob_c = prob_c; g.split_k = split_k; g.accumulate = split_k > 1 ? 0 : accumulate;
    g.bnb_z = Z; g.ldbz = ldz; g.bnb_a = a; g.bnb_mean = mean; g.bnb_invstd = invstd; g.bnb_sums = sums;
    g.bnb_ic = (float)(1.0 / (double)rows); g.bnb_out = dZ_out; g.ldbo = ldo;
    g.no_direct = 1;
    static const int diag_rt = getenv("CMF_GEMM_DIAG_RT") ? atoi(getenv("CMF_GEMM_DIAG_RT")) : 0;
    g.diag = diag_rt;
    const int err = launch<128, 128, true, false, 0>(g, st);
    if (err) return err;
    if (split_k > 1) return cmf_splitk_reduce(cout, cin, split_k, workspace, dW, lddw, accumulate, st);
    return 0;
}

// tiles_m of the forward tile config for (M): the caller sizes the stats partial buffer with it
extern "C" int cmf_gemm_tiles_m(int M) { return M > 64 ? (M + 127) / 128 : 1; }
