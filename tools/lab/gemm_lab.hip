// Lab: large-tile fp32 MFMA GEMM variants for the plain forward form  C[M][N] = A[M][K] * W[N][K]^T  (round 5).
// Stand-alone (no torch): hipcc --offload-arch=gfx950 -O3 -std=c++17 gemm_lab.hip -o gemm_lab ; ./gemm_lab [shape ...]
// Each variant is timed with HIP events over `reps` launches and checked against a naive kernel on sampled rows.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <array>
#include <cmath>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) void *g_gptr;
typedef __attribute__((address_space(3))) void *g_lptr;

struct Args {
    int M, N, K;
    const float *A; long long lda;
    const float *B; long long ldb;
    float *C; long long ldc;
    unsigned *counter;      // persistent variants: tile counter (zero before launch)
    int tiles_m, tiles_n;
    int flags;              // 1: no epilogue stores (timing only)
};

__device__ __forceinline__ int g_swz(int row) { return (((row >> 2) & 1) << 1) | (((row >> 1) & 1) ^ ((row >> 3) & 1)); }
#define WAIT_VMCNT(n) __builtin_amdgcn_s_waitcnt(((n) & 15) | (((n) >> 4) << 14) | 0x0F70)
__device__ __forceinline__ unsigned lds_addr(const float *p) { return (unsigned)(size_t)(__attribute__((address_space(3))) const float *)p; }
__device__ __forceinline__ f32x4 lds_read128(unsigned addr)
{
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
__device__ __forceinline__ void lds_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void keep(const f32x16 &v) { asm volatile("" :: "v"(v)); }
__device__ __forceinline__ void pin(f32x4 &v) { asm volatile("" : "+v"(v)); }

// BM x BN x 16 block tile, WGM x WGN waves (wave tile BM/WGM x BN/WGN), STAGES LDS stages of LDS-direct loads (prefetch distance
// STAGES - 1), MINW = waves per SIMD the register budget is sized for; PERSIST: grid = resident workgroups, tiles claimed from a counter.
template <int BM, int BN, int WGM, int WGN, int STAGES, int MINW, bool PERSIST>
__global__ __launch_bounds__(64 * WGM * WGN, MINW) void lab_kernel(const Args p)
{
    constexpr int NW = WGM * WGN, NT = 64 * NW;
    constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 32, TN = WN / 32;
    constexpr int ASLOTS = BM * 4, BSLOTS = BN * 4;
    constexpr int ANI = ASLOTS / NT, BNI = BSLOTS / NT;
    static_assert(ASLOTS % NT == 0 && BSLOTS % NT == 0 && ANI >= 1 && BNI >= 1, "staging map");
    constexpr int STAGE_F = (ASLOTS + BSLOTS) * 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid / WGN, wn = wid % WGN;
    const int frow = lane & 31, h = lane >> 5;
    const int nch = p.K / 16;
    __shared__ unsigned next_tile;

    int arow[TM], aswz[TM], brow[TN], bswz[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) { arow[i] = wm * WM + i * 32 + frow; aswz[i] = g_swz(arow[i]); }
#pragma unroll
    for (int j = 0; j < TN; ++j) { brow[j] = wn * WN + j * 32 + frow; bswz[j] = g_swz(brow[j]); }

    const int ntiles = p.tiles_m * p.tiles_n;
    int tile;
    if (PERSIST) {
        if (tid == 0) next_tile = atomicAdd(p.counter, 1u);
        __syncthreads();
        tile = (int)next_tile;
    } else {
        // XCD-aware static order: the column tiles of one row panel back to back on one XCD
        const int id = blockIdx.x, xcd = id % 8, slot = id / 8;
        const int per = (p.tiles_m + 7) / 8;
        const int tm_ = xcd * per + slot / p.tiles_n;
        if (tm_ >= p.tiles_m || slot / p.tiles_n >= per) return;
        tile = tm_ * p.tiles_n + slot % p.tiles_n;
    }
    while (tile < ntiles) {
        const int tm = tile / p.tiles_n, tn = tile % p.tiles_n;
        const int m0 = tm * BM, n0 = tn * BN;
        // per-lane source pointers of this thread's staging slots (fixed for the tile)
        const float *asrc[ANI], *bsrc[BNI];
#pragma unroll
        for (int q = 0; q < ANI; ++q) {
            const int sl = (q * NW + wid) * 64 + lane, row = sl >> 2;
            asrc[q] = p.A + (long long)(m0 + row) * p.lda + 4 * ((sl & 3) ^ g_swz(row));
        }
#pragma unroll
        for (int q = 0; q < BNI; ++q) {
            const int sl = (q * NW + wid) * 64 + lane, row = sl >> 2;
            bsrc[q] = p.B + (long long)(n0 + row) * p.ldb + 4 * ((sl & 3) ^ g_swz(row));
        }
        auto issue = [&](int kc, int st) {
            float *sa = smem + st * STAGE_F, *sb = sa + ASLOTS * 4;
#pragma unroll
            for (int q = 0; q < ANI; ++q)
                __builtin_amdgcn_global_load_lds((g_gptr)(asrc[q] + kc * 16), (g_lptr)(sa + (q * NW + wid) * 256), 16, 0, 0);
#pragma unroll
            for (int q = 0; q < BNI; ++q)
                __builtin_amdgcn_global_load_lds((g_gptr)(bsrc[q] + kc * 16), (g_lptr)(sb + (q * NW + wid) * 256), 16, 0, 0);
        };
        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        // prologue: STAGES - 1 chunks in flight
#pragma unroll
        for (int s = 0; s < STAGES - 1; ++s)
            if (s < nch) issue(s, s);
        // wait for chunk 0: all but the (min(nch, STAGES-1) - 1) newest issues
        {
            const int newer = (nch < STAGES - 1 ? nch : STAGES - 1) - 1;
            if (newer >= 2) WAIT_VMCNT(2 * (ANI + BNI)); else if (newer == 1) WAIT_VMCNT(ANI + BNI); else WAIT_VMCNT(0);
        }
        __builtin_amdgcn_s_barrier();
        f32x4 af[2][TM], bf[2][TN];
        auto read_frags = [&](const float *sa, const float *sb, int k8, int w) {
            const int kq = k8 / 4 + h;
#pragma unroll
            for (int i = 0; i < TM; ++i) af[w][i] = lds_read128(lds_addr(sa + (arow[i] * 4 + (kq ^ aswz[i])) * 4));
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[w][j] = lds_read128(lds_addr(sb + (brow[j] * 4 + (kq ^ bswz[j])) * 4));
        };
        auto pin_frags = [&](int w) {
#pragma unroll
            for (int i = 0; i < TM; ++i) pin(af[w][i]);
#pragma unroll
            for (int j = 0; j < TN; ++j) pin(bf[w][j]);
        };
        auto mfma_step = [&](int w) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[w][i][t], bf[w][j][t], acc[i][j], 0, 0, 0);
        };
        int st = 0;
        {
            const float *sa = smem, *sb = sa + ASLOTS * 4;
            read_frags(sa, sb, 0, 0);
            lds_wait(); pin_frags(0);
        }
        for (int c = 0; c < nch; ++c) {
            {
                const float *sa = smem + st * STAGE_F, *sb = sa + ASLOTS * 4;
                read_frags(sa, sb, 8, 1);
            }
            mfma_step(0);
            lds_wait(); pin_frags(1);                       // every LDS read of chunk c by this wave is complete
            // the stage of chunk c - 1 (read by everyone: all waves passed the barrier of chunk c - 1 -> c ... see below) takes chunk c + STAGES - 1
            if (c + STAGES - 1 < nch) issue(c + STAGES - 1, (st + STAGES - 1) % STAGES);
            const int sn = (st + 1) % STAGES;
            if (c + 1 < nch) {
                // chunk c + 1 must have landed: all but the newest min(STAGES - 2, remaining) issues complete
                int newer = nch - (c + 2); if (newer > STAGES - 2) newer = STAGES - 2;
                if (newer >= 2) WAIT_VMCNT(2 * (ANI + BNI)); else if (newer == 1) WAIT_VMCNT(ANI + BNI); else WAIT_VMCNT(0);
                __builtin_amdgcn_s_barrier();
                const float *sa = smem + sn * STAGE_F, *sb = sa + ASLOTS * 4;
                read_frags(sa, sb, 0, 0);
            }
            mfma_step(1);
            if (c + 1 < nch) { lds_wait(); pin_frags(0); }
            st = sn;
        }
        // ---- epilogue: straight from the accumulator layout (col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)) ----
        const int next = PERSIST ? 0 : ntiles;
        if (PERSIST) {
            __builtin_amdgcn_s_barrier();                   // every wave is out of the main loop: LDS stages and next_tile are free
            if (tid == 0) next_tile = atomicAdd(p.counter, 1u);
        }
        if (p.flags & 1) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) keep(acc[i][j]);
        } else {
            const unsigned lane_c = (unsigned)((4 * h * (int)p.ldc + frow) * 4);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float *cr = p.C + (long long)(m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2)) * p.ldc + n0 + wn * WN;
#pragma unroll
                    for (int j = 0; j < TN; ++j) *(float *)((char *)(cr + j * 32) + lane_c) = acc[i][j][r];
                }
        }
        if (PERSIST) { __syncthreads(); tile = (int)next_tile; } else tile = next;
    }
}


// P2: persistent, static XCD-aware tile schedule, ONE operand pipeline running straight through tile boundaries, epilogue through a
// wave-private 16 x 64 LDS tile as 16-byte stores of whole 256-byte row segments.  vmcnt protocol at a tile boundary: every load in
// flight is awaited BEFORE the epilogue's stores are issued, and the first two chunks of the next tile skip their load waits (their
// chunks are known to have landed), so the first counted wait behind the stores comes 2.5 chunks later.
template <int BM, int BN, int WGM, int WGN, int STAGES>
__global__ __launch_bounds__(64 * WGM * WGN, (WGM * WGN) / 4) void p2_kernel(const Args p)
{
    constexpr int NW = WGM * WGN, NT = 64 * NW;
    constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 32, TN = WN / 32;
    static_assert(WN == 64, "epilogue: 64-column wave tiles");
    constexpr int ASLOTS = BM * 4, BSLOTS = BN * 4;
    constexpr int ANI = ASLOTS / NT, BNI = BSLOTS / NT, LPC = ANI + BNI;      // loads per chunk and thread
    constexpr int STAGE_F = (ASLOTS + BSLOTS) * 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid / WGN, wn = wid % WGN;
    const int frow = lane & 31, h = lane >> 5;
    const int nch = p.K / 16;
    float *wt = smem + STAGES * STAGE_F + wid * 1024;            // 16 x 64 wave-private epilogue tile

    int arow[TM], aswz[TM], brow[TN], bswz[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) { arow[i] = wm * WM + i * 32 + frow; aswz[i] = g_swz(arow[i]); }
#pragma unroll
    for (int j = 0; j < TN; ++j) { brow[j] = wn * WN + j * 32 + frow; bswz[j] = g_swz(brow[j]); }

    // static schedule: XCD x owns row panels [x per, (x + 1) per); its tiles in (panel, column) order are dealt to its workgroups
    const int xcd = blockIdx.x % 8, slot = blockIdx.x / 8, nslots = gridDim.x / 8;
    const int per = (p.tiles_m + 7) / 8;
    const int pan_end = min(p.tiles_m, (xcd + 1) * per) - xcd * per;     // panels of this XCD
    const int xt = pan_end > 0 ? pan_end * p.tiles_n : 0;                // tiles of this XCD
    const int my_tiles = slot < xt ? (xt - slot + nslots - 1) / nslots : 0;
    if (my_tiles == 0) return;
    auto tile_mn = [&](int i, int &m0, int &n0) {
        const int j = slot + i * nslots;
        m0 = (xcd * per + j / p.tiles_n) * BM; n0 = (j % p.tiles_n) * BN;
    };
    // issue cursor
    const float *asrc[ANI], *bsrc[BNI];
    int it_tile = 0, it_c = 0;
    auto set_src = [&](int i) {
        int m0, n0; tile_mn(i, m0, n0);
#pragma unroll
        for (int q = 0; q < ANI; ++q) {
            const int sl = (q * NW + wid) * 64 + lane, row = sl >> 2;
            asrc[q] = p.A + (long long)(m0 + row) * p.lda + 4 * ((sl & 3) ^ g_swz(row));
        }
#pragma unroll
        for (int q = 0; q < BNI; ++q) {
            const int sl = (q * NW + wid) * 64 + lane, row = sl >> 2;
            bsrc[q] = p.B + (long long)(n0 + row) * p.ldb + 4 * ((sl & 3) ^ g_swz(row));
        }
    };
    set_src(0);
    const int G = my_tiles * nch;                              // chunks of this workgroup
    int g_issued = 0;
    auto issue_next = [&](int st) {                            // the next chunk of the stream into stage st
        float *sa = smem + st * STAGE_F, *sb = sa + ASLOTS * 4;
#pragma unroll
        for (int q = 0; q < ANI; ++q)
            __builtin_amdgcn_global_load_lds((g_gptr)(asrc[q] + it_c * 16), (g_lptr)(sa + (q * NW + wid) * 256), 16, 0, 0);
#pragma unroll
        for (int q = 0; q < BNI; ++q)
            __builtin_amdgcn_global_load_lds((g_gptr)(bsrc[q] + it_c * 16), (g_lptr)(sb + (q * NW + wid) * 256), 16, 0, 0);
        ++g_issued;
        if (++it_c == nch) { it_c = 0; ++it_tile; if (it_tile < my_tiles) set_src(it_tile); }
    };
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
        if (s < G) issue_next(s);
    {
        const int newer = (G < STAGES - 1 ? G : STAGES - 1) - 1;
        if (newer >= 2) WAIT_VMCNT(2 * LPC); else if (newer == 1) WAIT_VMCNT(LPC); else WAIT_VMCNT(0);
    }
    __builtin_amdgcn_s_barrier();
    f32x4 af[2][TM], bf[2][TN];
    auto read_frags = [&](const float *sa, const float *sb, int k8, int w) {
        const int kq = k8 / 4 + h;
#pragma unroll
        for (int i = 0; i < TM; ++i) af[w][i] = lds_read128(lds_addr(sa + (arow[i] * 4 + (kq ^ aswz[i])) * 4));
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[w][j] = lds_read128(lds_addr(sb + (brow[j] * 4 + (kq ^ bswz[j])) * 4));
    };
    auto pin_frags = [&](int w) {
#pragma unroll
        for (int i = 0; i < TM; ++i) pin(af[w][i]);
#pragma unroll
        for (int j = 0; j < TN; ++j) pin(bf[w][j]);
    };
    auto mfma_step = [&](int w) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[w][i][t], bf[w][j][t], acc[i][j], 0, 0, 0);
    };
    int st = 0;
    {
        const float *sa = smem, *sb = sa + ASLOTS * 4;
        read_frags(sa, sb, 0, 0);
        lds_wait(); pin_frags(0);
    }
    int ct = 0, cc = 0;                                         // compute cursor: tile, chunk in tile
    int m0, n0; tile_mn(0, m0, n0);
    for (int g = 0; g < G; ++g) {
        {
            const float *sa = smem + st * STAGE_F, *sb = sa + ASLOTS * 4;
            read_frags(sa, sb, 8, 1);
        }
        mfma_step(0);
        lds_wait(); pin_frags(1);
        if (g_issued < G) issue_next((st + STAGES - 1) % STAGES);
        const int sn = (st + 1) % STAGES;
        if (g + 1 < G) {
            if (!(ct > 0 && cc < 2)) {                          // (chunks 1 and 2 of a later tile were awaited before its predecessor's stores)
                const int newer = g_issued - (g + 2);           // chunks issued after chunk g + 1
                if (newer >= 2) WAIT_VMCNT(2 * LPC); else if (newer == 1) WAIT_VMCNT(LPC); else WAIT_VMCNT(0);
            }
            __builtin_amdgcn_s_barrier();
            const float *sa = smem + sn * STAGE_F, *sb = sa + ASLOTS * 4;
            read_frags(sa, sb, 0, 0);
        }
        mfma_step(1);
        if (g + 1 < G) { lds_wait(); pin_frags(0); }
        st = sn;
        if (++cc == nch) {
            // ---- tile done: every load in flight is awaited, then the stores go out behind nothing this wave will wait for soon ----
            WAIT_VMCNT(0);
            if (!(p.flags & 1)) {
                const int rl = lane >> 4, c4 = (lane & 15) * 4;
                // wave-uniform row bases (scalar), ONE 32-bit per-lane byte offset
                float *cbase = p.C + (long long)(((p.flags & 2) ? 0 : m0) + wm * WM) * p.ldc + ((p.flags & 2) ? 0 : n0) + wn * WN;
                unsigned lane_c = (unsigned)((rl * (int)p.ldc + c4) * 4);
                asm volatile("" : "+v"(lane_c));
                float *wr = wt + (4 * h) * 64 + frow;                      // this lane's write column
                const float *rd = wt + rl * 64 + c4;
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
                        for (int j = 0; j < TN; ++j)
#pragma unroll
                            for (int rr = 0; rr < 8; ++rr)
                                wr[((rr & 3) + 8 * (rr >> 2)) * 64 + j * 32] = acc[i][j][8 * hf + rr];
                        asm volatile("" ::: "memory");
                        f32x4 v[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) v[u] = *(const f32x4 *)(rd + (4 * u) * 64);
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (p.flags & 4) __builtin_nontemporal_store(v[u], (f32x4 *)((char *)(cbase + (long long)(i * 32 + hf * 16 + 4 * u) * p.ldc) + lane_c));
                            else *(f32x4 *)((char *)(cbase + (long long)(i * 32 + hf * 16 + 4 * u) * p.ldc) + lane_c) = v[u];
                        asm volatile("" ::: "memory");
                    }
            } else {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) keep(acc[i][j]);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
            cc = 0; ++ct;
            if (ct < my_tiles) tile_mn(ct, m0, n0);
        }
    }
}
template <int BM, int BN, int WGM, int WGN, int STAGES>
static void launch_p2(const Args &a, int grid, size_t lds, hipStream_t st)
{
    hipLaunchKernelGGL((p2_kernel<BM, BN, WGM, WGN, STAGES>), dim3(grid), dim3(64 * WGM * WGN), lds, st, a);
}

__global__ void naive_rows(int N, int K, const float *A, long long lda, const float *B, long long ldb, float *out, const int *rows, int nrows)
{
    const int r = blockIdx.y, n = blockIdx.x * 256 + threadIdx.x;
    if (r >= nrows || n >= N) return;
    const float *a = A + (long long)rows[r] * lda, *b = B + (long long)n * ldb;
    double s = 0.0;
    for (int k = 0; k < K; ++k) s += (double)a[k] * (double)b[k];
    out[(long long)r * N + n] = (float)s;
}

__global__ void fill_kernel(float *p, long long n, unsigned seed)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
        p[i] = ((x & 0xFFFFFF) / 16777216.0f) * 2.f - 1.f;
    }
}

struct Variant { const char *name; int bm, bn, threads, stages, persist; /* 2: p2 */ void (*launch)(const Args &, int grid, size_t lds, hipStream_t); const void *fn; };

template <int BM, int BN, int WGM, int WGN, int STAGES, int MINW, bool PERSIST>
static void launch_v(const Args &a, int grid, size_t lds, hipStream_t st)
{
    hipLaunchKernelGGL((lab_kernel<BM, BN, WGM, WGN, STAGES, MINW, PERSIST>), dim3(grid), dim3(64 * WGM * WGN), lds, st, a);
}
#define VAR(name, BM, BN, WGM, WGN, ST, MINW, P) \
    Variant{name, BM, BN, 64 * WGM * WGN, ST, P, launch_v<BM, BN, WGM, WGN, ST, MINW, P>, (const void *)lab_kernel<BM, BN, WGM, WGN, ST, MINW, P>}

int main(int argc, char **argv)
{
    std::vector<Variant> vars = {
        VAR("128x128 4w(2x2) s3 3wg/cu static", 128, 128, 2, 2, 3, 3, false),
        VAR("256x256 8w(2x4) s3 static", 256, 256, 2, 4, 3, 2, false),
        VAR("256x256 8w(2x4) s4 static", 256, 256, 2, 4, 4, 2, false),
        VAR("256x256 8w(2x4) s4 persist", 256, 256, 2, 4, 4, 2, true),
        VAR("256x256 4w(2x2) s4 static", 256, 256, 2, 2, 4, 1, false),
        VAR("256x256 4w(2x2) s4 persist", 256, 256, 2, 2, 4, 1, true),
        VAR("256x128 8w(4x2) s3 2wg/cu static", 256, 128, 4, 2, 3, 4, false),
        VAR("256x128 4w(2x2) s3 2wg/cu static", 256, 128, 2, 2, 3, 2, false),
        Variant{"P2 256x256 8w(2x4) s4", 256, 256, 512, 4, 2, launch_p2<256, 256, 2, 4, 4>, (const void *)p2_kernel<256, 256, 2, 4, 4>},
        Variant{"P2 256x256 8w(2x4) s3", 256, 256, 512, 3, 2, launch_p2<256, 256, 2, 4, 3>, (const void *)p2_kernel<256, 256, 2, 4, 3>},
        VAR("256x128 4w(2x2) s4 2wg/cu static", 256, 128, 2, 2, 4, 2, false),
        VAR("128x256 4w(2x2) s4 2wg/cu static", 128, 256, 2, 2, 4, 2, false),
    };
    std::vector<std::array<int, 3>> shapes = {{524288, 256, 512}, {131072, 512, 512}, {16384, 2048, 1024}, {65536, 256, 512}};
    if (argc > 1) {
        shapes.clear();
        for (int i = 1; i < argc; ++i) { int m, n, k; if (sscanf(argv[i], "%dx%dx%d", &m, &n, &k) == 3) shapes.push_back({m, n, k}); }
    }
    const char *only = getenv("LAB_ONLY");
    int dev = 0; CK(hipSetDevice(dev));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, dev));
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs\n", prop.name, cus);
    unsigned *counter; CK(hipMalloc(&counter, 64));
    for (auto &s : shapes) {
        const int M = s[0], N = s[1], K = s[2];
        float *A, *B, *C; CK(hipMalloc(&A, (size_t)M * K * 4)); CK(hipMalloc(&B, (size_t)N * K * 4)); CK(hipMalloc(&C, (size_t)M * N * 4));
        hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, A, (long long)M * K, 1u);
        hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, B, (long long)N * K, 7u);
        // reference rows
        const int nrows = 64;
        std::vector<int> rows(nrows);
        for (int i = 0; i < nrows; ++i) rows[i] = (int)(((long long)i * 2654435761ll + 12345) % M);
        rows[0] = 0; rows[1] = M - 1; rows[2] = 255; rows[3] = 256;
        int *drows; float *ref, *href = new float[(size_t)nrows * N], *hc = new float[N];
        CK(hipMalloc(&drows, nrows * 4)); CK(hipMalloc(&ref, (size_t)nrows * N * 4));
        CK(hipMemcpy(drows, rows.data(), nrows * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(naive_rows, dim3((N + 255) / 256, nrows), dim3(256), 0, 0, N, K, A, (long long)K, B, (long long)K, ref, drows, nrows);
        CK(hipMemcpy(href, ref, (size_t)nrows * N * 4, hipMemcpyDeviceToHost));
        printf("== %d x %d x %d\n", M, N, K);
        for (auto &v : vars) {
            if (only && !strstr(v.name, only)) continue;
            if (M % v.bm || N % v.bn) { printf("  %-40s (shape not a multiple of the tile)\n", v.name); continue; }
            Args a{M, N, K, A, K, B, K, C, N, counter, M / v.bm, N / v.bn, getenv("LAB_FLAGS") ? atoi(getenv("LAB_FLAGS")) : 0};
            const size_t lds = (size_t)v.stages * (v.bm + v.bn) * 64 + (v.persist == 2 ? (size_t)(v.threads / 64) * 4096 : 0);
            CK(hipFuncSetAttribute(v.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            int occ = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, v.fn, v.threads, lds));
            int grid = v.persist ? occ * cus : 8 * ((a.tiles_m + 7) / 8) * a.tiles_n;
            if (v.persist == 1 && grid > a.tiles_m * a.tiles_n) grid = a.tiles_m * a.tiles_n;
            CK(hipMemset(C, 0, (size_t)M * N * 4));
            auto once = [&]() { if (v.persist) CK(hipMemsetAsync(counter, 0, 4, 0)); v.launch(a, grid, lds, 0); };
            once(); CK(hipDeviceSynchronize());
            double maxerr = 0.0;
            for (int i = 0; i < nrows; ++i) {
                CK(hipMemcpy(hc, C + (size_t)rows[i] * N, (size_t)N * 4, hipMemcpyDeviceToHost));
                for (int n = 0; n < N; ++n) maxerr = fmax(maxerr, fabs((double)hc[n] - (double)href[(size_t)i * N + n]));
            }
            once(); once();
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            const int reps = 10;
            CK(hipEventRecord(e0, 0));
            for (int r = 0; r < reps; ++r) once();
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms = 0.f; CK(hipEventElapsedTime(&ms, e0, e1));
            const double tf = 2.0 * M * N * K / (ms / reps * 1e-3) / 1e12;
            printf("  %-40s %7.1f us  %6.1f TF  occ %d grid %d  lds %zu  maxerr %.2e%s\n", v.name, ms / reps * 1e3, tf, occ, grid, lds, maxerr,
                   maxerr > 1e-3 ? "  <-- WRONG" : "");
            fflush(stdout);
        }
        CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(C)); CK(hipFree(drows)); CK(hipFree(ref)); delete[] href; delete[] hc;
    }
    return 0;
}
