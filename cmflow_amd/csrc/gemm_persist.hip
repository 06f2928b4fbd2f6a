// Persistent form of the fp32 MFMA GEMM for the data-gradient / forward 1x1 convolutions of the set-conv stacks
// (utils/model_utils/radarflow_util.py:151-153,215-221 and their autograd backward):
//
//     C[M,N] = epi( pro(A)[M,K] * B )        A[M][K] row-major, B = B[K][N] (data gradient) or W[N][K] (forward)
//
// Why a second kernel next to gemm.hip's: at K = 256 a 128 x 128 tile is 16 chunks of MFMAs (~37 us with three workgroups
// sharing a CU) followed by an epilogue that reads 64 KB of the producer's Z, masks, sums statistics and writes 64 KB --
// 20-23 us during which the workgroup issues no MFMA (profiles/r03_gemm_timeline_bwd.txt: on average 1.8 of a CU's three
// workgroups are in their main loop).  Here a workgroup is PERSISTENT (grid = 2 per CU, tiles claimed from per-XCD
// counters) and keeps TWO accumulator sets: while tile t+1 accumulates, the finished tile t is written out in eight 8-row
// sub-pieces, one per chunk, placed between the MFMAs of chunks 1..8 of tile t+1 -- the Z rows of a sub-piece arrive in a
// wave-private LDS buffer by LDS-direct requests issued two chunks earlier, the arithmetic runs in the accumulator layout
// (a lane owns one column: constants in registers, statistics as per-lane sums), the results leave through the same buffer
// as 16-byte stores -- and the operand pipeline (LDS-direct, three stages, prefetch distance two) runs straight through tile
// boundaries.  At two waves per SIMD each wave owns 256 registers: 64 + 64 accumulators, 32 fragment registers, the
// sub-piece's working set.  Same MFMA sequence per output element as gemm.hip (same chunk order, same K-permutation): the
// outputs are BIT-IDENTICAL to the non-persistent kernel's; the column statistics are summed in another (fixed) order.
// Every streaming access is a BUFFER instruction (scalar descriptor + scalar offset + one 32-bit lane offset) and every LDS
// access inline asm with explicit waits; two rules learnt the hard way are written where they apply (no asynchronous LDS read
// in flight at a control-flow edge; vmcnt waits count younger loads only).
//
// Scope: interior problems only (M, N multiples of 128, K a multiple of 16 with >= P_MIN_CHUNKS chunks per tile, 16-byte
// aligned rows, no C +=); by default the data gradients with a backward epilogue (cmf_pgemm_grid); everything else stays
// on gemm.hip's kernels.
#include <atomic>
#include <cstdlib>
#include <type_traits>
#include "cmf_common.h"
#include "gemm_args.h"

typedef float pf32x16 __attribute__((ext_vector_type(16)));
typedef float pf32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) void *p_gptr;
typedef __attribute__((address_space(3))) void *p_lptr;

constexpr int P_THREADS = 256, P_BM = 128, P_BN = 128, P_BK = 16, P_NST = 3;
constexpr int P_SLOTS = 128 * (P_BK / 4);                 // 16-byte slots of one operand chunk (A and B alike)
constexpr int P_STAGE = 2 * P_SLOTS * 4 + 32;             // floats: A | B | pro_a[16] pro_c[16]
// per wave: dxyz rows of its 64 tile rows (kinds 4, 5) | 8 x 64 row buffers: the producer's Z rows of a sub-piece arrive here (LDS-direct,
// two buffers; backward kinds), the sub-piece's outputs leave through the same rows (one buffer is enough without Z)
__host__ __device__ constexpr int p_wdq(int epi) { return epi >= 4 ? 64 * 4 : 0; }
__host__ __device__ constexpr int p_wz(int epi) { return epi >= 2 ? 2 * 8 * 64 : 8 * 64; }
constexpr int P_PROK = 1024;                              // longest contraction with an A prologue: scale | shift tables in LDS
constexpr int P_RED = 2 * 5 * 128;                        // [wave row][statistic][column]
__host__ __device__ constexpr int p_lds_floats(int epi, bool pro_a)      // + the claimed tile (one word, padded)
{
    return P_NST * P_STAGE + 4 * (p_wdq(epi) + p_wz(epi)) + P_RED + 4 + (pro_a ? 2 * P_PROK : 0);
}
constexpr int P_MIN_CHUNKS = 12;                          // chunks 0..10 of a tile carry the previous tile's epilogue
constexpr int P_NSUB = 8;                                 // 8-row sub-pieces of a wave's 64 x 64 tile

#define P_WAIT_VMCNT(n) __builtin_amdgcn_s_waitcnt(((n) & 15) | (((n) >> 4) << 14) | 0x0F70)

__device__ __forceinline__ int p_swz(int row) { return (((row >> 2) & 1) << 1) | (((row >> 1) & 1) ^ ((row >> 3) & 1)); }
__device__ __forceinline__ unsigned p_lds_addr(const float *p) { return (unsigned)(size_t)(__attribute__((address_space(3))) const float *)p; }
// Every LDS access of this kernel is inline asm: the compiler's waitcnt pass treats an LDS access it can see as possibly
// aliasing the in-flight LDS-direct loads and drains them (vmcnt(0)) first.  Completion is awaited explicitly.
template <int OFF>
__device__ __forceinline__ pf32x4 p_read128(unsigned addr)
{
    pf32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
template <int OFF>
__device__ __forceinline__ float p_read32(unsigned addr)
{
    float v;
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
template <int OFF>
__device__ __forceinline__ void p_write32(unsigned addr, float v) { asm volatile("ds_write_b32 %0, %1 offset:%2" :: "v"(addr), "v"(v), "n"(OFF) : "memory"); }
template <int OFF>
__device__ __forceinline__ void p_write128(unsigned addr, pf32x4 v) { asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(addr), "v"(v), "n"(OFF) : "memory"); }
__device__ __forceinline__ void p_lds_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void p_pin(pf32x4 &v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void p_pin1(float &v) { asm volatile("" : "+v"(v)); }
#define P_SB() __builtin_amdgcn_sched_barrier(0)

// Streaming accesses (operand requests, the producer's Z, the output) are BUFFER instructions: a descriptor in four scalar
// registers (rebuilt per tile from a uniform pointer), a scalar byte offset and ONE 32-bit per-lane offset.  With plain
// pointers loop strength reduction turns every stream of the peeled loop into a per-lane 64-bit address: two registers
// each, spilled, reloaded behind a vmcnt(0) in the middle of the MFMA stream.  A tile's rows lie within 4 GB of its first.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t p_rsrc(const void *q) { return __builtin_amdgcn_make_buffer_rsrc((void *)q, 0, -1, 0x00020000); }
__device__ __forceinline__ int p_spin(int v) { v = __builtin_amdgcn_readfirstlane(v); asm volatile("" : "+s"(v)); return v; }   // opaque scalar

// EPI (gemm.hip epilogue_kind): 0 raw store (plain GEMMs, split-K slabs), 1 forward (bias, none / ReLU / leaky, BN statistics),
// 2 backward through BN + ReLU, 3 backward through (leaky) ReLU, 4 / 5 the same with the three dxyz column sums.
// A_T: A stored [K][M] (weight-gradient layout).  B_T: B stored [N][K] (forward layout) instead of [K][N].
// PRO: the producer's BN + ReLU on an operand: A'[m,k] = relu(pa[k] A + pc[k]) (A[M][K]) or B'[k,n] = relu(qa[n] B + qc[n]) (A_T).
template <bool A_T, bool B_T, int EPI, bool PRO>
__global__ __launch_bounds__(P_THREADS, 2) void pgemm_kernel(const GemmArgs p, int *__restrict__ sched)
{
    constexpr bool BNR = EPI == 2 || EPI == 4, WQ = EPI >= 4, USE_Z = EPI >= 2;
    constexpr bool PROA = PRO && !A_T, PROB = PRO && A_T;
    constexpr int NSTAT = WQ ? 5 : 2;
    constexpr int P_WDQ = p_wdq(EPI), P_WZ = p_wz(EPI), P_WSZ = P_WDQ + P_WZ;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1;                               // wave tile 64 x 64 at (wm, wn)
    const int h = lane >> 5, cl = lane & 31;

    // ---- tile walk ----
    // Workgroup b lives on XCD b % 8 (round-robin dispatch; a placement assumed for speed only).  The row panels are dealt to
    // the XCDs in 8 contiguous ranges; inside a range the tiles are numbered column tile fastest, so the column tiles that
    // share an A row panel are worked on at the same time by workgroups of ONE XCD.  Tiles are CLAIMED, not dealt: one
    // counter per XCD (sched[0..7], zero at launch; the last workgroup to leave zeroes the block again).  With a static deal
    // the two workgroups of a CU do not advance at the same pace -- the older one wins the arbitration for the SIMD's matrix
    // pipe -- and finished 180 us of a 1290 us kernel apart (tools/pgemm_timeline.py, profiles/r04_pgemm_timeline.txt); claimed,
    // they end within 15 us of each other.  The kernel's duration did not change (a CU's throughput is the same either way);
    // claiming is kept because it bounds the tail at one tile whatever the shape and the placement.  A tile is claimed one
    // tile ahead: wave 0 adds to the counter in a tile's first chunk, the result goes through LDS (behind two chunk
    // barriers) and is the next tile from chunk 4 on.
    // split_k > 1 (weight gradients: huge contraction, few output tiles): slab s lives on XCD s % 8 and ALL output tiles of a
    // slab are consecutive slots there, so both operand slabs are fetched from HBM once and re-read from that XCD's L2.
    const int tiles_m = p.M / P_BM, tiles_n = p.N / P_BN;
    const int xcd = blockIdx.x & 7;
    const int per = (tiles_m + 7) / 8;
    const int tiles = tiles_m * tiles_n;
    const int nslots = p.split_k > 1 ? tiles * ((p.split_k - xcd + 7) / 8) : max(0, min(per, tiles_m - xcd * per)) * tiles_n;
    const unsigned sched_lds = (unsigned)((P_NST * P_STAGE + 4 * P_WSZ + P_RED) * 4);
    int claim = 0;                                                       // wave 0, lane 0: the counter value a claim returned
    auto claim_issue = [&]() { if (wid == 0 && lane == 0) claim = __hip_atomic_fetch_add(sched + xcd, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto claim_publish = [&]() { if (wid == 0 && lane == 0) p_write32<0>(sched_lds, __builtin_bit_cast(float, claim)); };
    auto claim_read = [&]() {                                            // behind a workgroup barrier that follows claim_publish
        float v = p_read32<0>(sched_lds);
        p_lds_wait(); p_pin1(v);
        return __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v));
    };
    auto leave = [&]() {                                                 // every workgroup, once: the last one resets the block
        if (tid == 0) {
            const int d = __hip_atomic_fetch_add(sched + 8, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (d == (int)gridDim.x - 1) {
#pragma unroll
                for (int i = 0; i < 9; ++i) __hip_atomic_store(sched + i, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    };
    claim_issue();
    claim_publish();
    p_lds_wait();
    __builtin_amdgcn_s_barrier();
    int slot_c = claim_read();
    if (slot_c >= nslots) { leave(); return; }
    const int nch = p.K / P_BK / p.split_k;                             // chunks per tile (the host makes the division exact)
    const unsigned long long t_start = p.trace ? wall_clock64() : 0ull;
    unsigned long long t_first = 0ull, t_drain = 0ull;

    // ---- per-lane pieces of the LDS-direct source addresses (bytes) ----
    // A (and B in the [N][K] layout): 16-byte slot sl = (q*4 + wid)*64 + lane -> row = sl >> 2, k-quad (sl & 3) ^ swz(row)
    // B[K][N]: slot sl -> k = sl / 32, x = sl % 32 -> column quad x ^ (((k >> 2) & 1) * 8)
    // row-major operand (A[M][K], B[N][K]): rows of 16 k; K-major operand (A[K][M], B[K][N]): 16 rows of 128 columns
    auto rm_off = [&](int ld) { const int row = wid * 16 + (lane >> 2); return (unsigned)((row * ld + 4 * ((lane & 3) ^ p_swz(row))) * 4); };
    auto km_off = [&](int ld) { const int k = wid * 2 + (lane >> 5); return (unsigned)((k * ld + 4 * ((lane & 31) ^ (((k >> 2) & 1) * 8))) * 4); };
    const unsigned offA = A_T ? km_off((int)p.lda) : rm_off((int)p.lda), offB = B_T ? rm_off((int)p.ldb) : km_off((int)p.ldb);

    // ---- fragment read addresses (bytes inside a stage) ----
    // row-major image [128][16], swizzled: row r, k-quad kq at slot r*4 + (kq ^ swz(r)); set w reads kq = 2w + h with one
    // ds_read_b128 (index = w; the second block row / column is + 32 rows = + 2048 bytes, same swizzle).
    // K-major image [16][128]: k rows of 512 bytes, column quads swizzled by (k >> 2) & 1 = h; 4 x ds_read_b32 per fragment
    // (index = block, set w is + 4096 bytes).
    unsigned fa[2], fb[2];
    {
        auto rm_frag = [&](int row, int w) { return (unsigned)((row * 4 + ((2 * w + h) ^ p_swz(row))) * 16); };
        auto km_frag = [&](int row) { return (unsigned)((4 * h) * 512 + ((((row >> 2) ^ (h * 8)) << 2) | (row & 3)) * 4); };
        const int arow = wm * 64 + cl, brow = wn * 64 + cl;
        fa[0] = A_T ? km_frag(arow) : rm_frag(arow, 0); fa[1] = A_T ? km_frag(arow + 32) : rm_frag(arow, 1);
        fb[0] = P_SLOTS * 16 + (B_T ? rm_frag(brow, 0) : km_frag(brow)); fb[1] = P_SLOTS * 16 + (B_T ? rm_frag(brow, 1) : km_frag(brow + 32));
    }
    // ---- wave-private epilogue LDS (byte addresses) ----
    const unsigned wbase = (unsigned)((P_NST * P_STAGE + wid * P_WSZ) * 4);
    const unsigned w_dq = wbase + (unsigned)(h * 64);                                  // dxyz row 8 S + rr + 4 h: + (8 S + rr) * 16
    const unsigned w_za = wbase + (unsigned)(P_WDQ * 4 + h * 1024 + cl * 4);           // Z buffer, accumulator layout: + b * 2048 + rr * 256 + j * 128
    const unsigned w_zr = wbase + (unsigned)(P_WDQ * 4 + lane * 16);                   // Z buffer, row layout (16 bytes per lane): + b * 2048 + u * 1024
    const unsigned w_fold = wbase + (unsigned)(P_WDQ * 4 + cl * 4);                    // + (which * 2 + j) * 128 (over the idle Z buffers)
    const unsigned red_base = (unsigned)((P_NST * P_STAGE + 4 * P_WSZ) * 4);
    const unsigned pro_lds = (unsigned)((P_NST * P_STAGE + 4 * P_WSZ + P_RED + 4) * 4);      // [2][P_PROK]: scale | shift by contraction index

    pf32x16 acc[2][2], accp[2][2];
    pf32x4 af[2][2], bf[2][2];
    float zq[4][2];                                                     // the producer's Z at the sub-piece's 4 registers x 2 block columns (transient)
    float kc[4][2];                                                     // ea, ec, mean, invstd of the lane's two columns (previous tile)
    float t1[2] = {0.f, 0.f}, t2[2] = {0.f, 0.f}, qs[3][2] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
    const float slope = EPI == 1 ? (p.act == 1 ? 0.f : (p.act == 2 ? 0.1f : 1.f)) : (p.bwd_mode == 2 ? 0.1f : 0.f);
    pf32x4 pa4[2], pc4[2];                                               // A prologue: scale / shift of the fragment's 4 contraction indices
    float qa[2] = {1.f, 1.f}, qc[2] = {0.f, 0.f};                        // B prologue: scale / shift of the lane's two columns
    const bool want_stats = p.stats != nullptr;

    // tile coordinates: cur (accumulating), prv (being written out)
    auto tile_of = [&](int s, int &tm, int &tn, int &sp) {
        if (p.split_k > 1) { sp = xcd + 8 * (s / tiles); const int t = s % tiles; tm = t / tiles_n; tn = t % tiles_n; }
        else { sp = 0; tm = xcd * per + s / tiles_n; tn = s % tiles_n; }
    };
    int tm_c, tn_c, sp_c, tm_p = 0, tn_p = 0, sp_p = 0, tm_n = 0, tn_n = 0, sp_n = 0, n_done = 0;
    bool has_next = false;
    tile_of(slot_c, tm_c, tn_c, sp_c);
    // operand stream: descriptors of the tile's A row panel / B column panel, scalar byte offsets of the NEXT chunk to request
    __amdgpu_buffer_rsrc_t rA, rB;
    int soA = 0, soB = 0;
    const int strideA_q = (A_T ? 8 : 64) * (int)p.lda * 4, strideB_q = (B_T ? 64 : 8) * (int)p.ldb * 4;   // q = 1: + 64 rows (row-major) / + 8 k-rows (K-major)
    const int chunkA = A_T ? P_BK * (int)p.lda * 4 : P_BK * 4, chunkB = B_T ? P_BK * 4 : P_BK * (int)p.ldb * 4;   // bytes per chunk along K
    auto set_stream = [&](int tm, int tn, int sp) {
        const long long kb = (long long)sp * nch * P_BK;                // first contraction index of the slab
        rA = p_rsrc(A_T ? p.A + kb * p.lda + tm * P_BM : p.A + (long long)tm * P_BM * p.lda + kb);
        rB = p_rsrc(B_T ? p.B + (long long)tn * P_BN * p.ldb + kb : p.B + kb * p.ldb + tn * P_BN);
        soA = 0; soB = 0;
    };
    set_stream(tm_c, tn_c, sp_c);
    int st_issue = 0;                                                  // stage the next request goes to
    auto issue_part = [&](int q) {                                     // the four requests of a chunk, one per call
        float *sa = smem + st_issue * P_STAGE, *sb = sa + P_SLOTS * 4;
        if (q == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (p_lptr)(sa + wid * 256), 16, offA, soA, 0, 0);
        if (q == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (p_lptr)(sa + (4 + wid) * 256), 16, offA, soA + strideA_q, 0, 0);
        if (q == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (p_lptr)(sb + wid * 256), 16, offB, soB, 0, 0);
        if (q == 3) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (p_lptr)(sb + (4 + wid) * 256), 16, offB, soB + strideB_q, 0, 0);
            soA = p_spin(soA + chunkA); soB = p_spin(soB + chunkB);
            st_issue = st_issue == 2 ? 0 : st_issue + 1;
        }
    };
    auto issue = [&]() { issue_part(0); issue_part(1); issue_part(2); issue_part(3); };
    constexpr int DMA = 4;                                              // LDS-direct requests per wave and chunk

    // fragment reads of set w (k-quads 2w + h) from the stage at byte offset sbyte: issued, not awaited.  kq: byte offset of the
    // chunk's first contraction index in the prologue tables (A prologue only).
    auto read_frags = [&](unsigned sbyte, unsigned kq, auto wc) {
        constexpr int w = decltype(wc)::value;
        if (!A_T) { const unsigned a = sbyte + fa[w]; af[w][0] = p_read128<0>(a); af[w][1] = p_read128<2048>(a); }
        else {
            const unsigned a0 = sbyte + fa[0], a1 = sbyte + fa[1];
            af[w][0].x = p_read32<w * 4096 + 0>(a0); af[w][0].y = p_read32<w * 4096 + 512>(a0);
            af[w][0].z = p_read32<w * 4096 + 1024>(a0); af[w][0].w = p_read32<w * 4096 + 1536>(a0);
            af[w][1].x = p_read32<w * 4096 + 0>(a1); af[w][1].y = p_read32<w * 4096 + 512>(a1);
            af[w][1].z = p_read32<w * 4096 + 1024>(a1); af[w][1].w = p_read32<w * 4096 + 1536>(a1);
        }
        if (B_T) { const unsigned b = sbyte + fb[w]; bf[w][0] = p_read128<0>(b); bf[w][1] = p_read128<2048>(b); }
        else {
            const unsigned b0 = sbyte + fb[0], b1 = sbyte + fb[1];
            bf[w][0].x = p_read32<w * 4096 + 0>(b0); bf[w][0].y = p_read32<w * 4096 + 512>(b0);
            bf[w][0].z = p_read32<w * 4096 + 1024>(b0); bf[w][0].w = p_read32<w * 4096 + 1536>(b0);
            bf[w][1].x = p_read32<w * 4096 + 0>(b1); bf[w][1].y = p_read32<w * 4096 + 512>(b1);
            bf[w][1].z = p_read32<w * 4096 + 1024>(b1); bf[w][1].w = p_read32<w * 4096 + 1536>(b1);
        }
        if (PROA) { const unsigned q = pro_lds + kq + (unsigned)(h * 16); pa4[w] = p_read128<w * 32>(q); pc4[w] = p_read128<w * 32 + P_PROK * 4>(q); }
    };
    auto pin_frags = [&](int w) { p_pin(af[w][0]); p_pin(af[w][1]); p_pin(bf[w][0]); p_pin(bf[w][1]); if (PROA) { p_pin(pa4[w]); p_pin(pc4[w]); } };
    // MFMA number m (0..15) of a half chunk: component t = m / 4 of fragment set w against accumulator block (i, j) = (m / 2 % 2,
    // m % 2).  first: the tile's first products (C = 0); with `keep` the finished tile's block moves to the second accumulator
    // set right in front of the MFMA that overwrites it (in practice the compiler renames the registers instead).
    // The fillers of a chunk -- operand requests, fragment reads, the previous tile's epilogue -- are placed after named MFMAs and
    // fenced with sched_barrier(0) on their far side, a handful of instructions at a time; the MFMAs themselves are free to be
    // scheduled around them.  [Measured on one box: every MFMA fenced on both sides (strict "one MFMA, one filler" order) was
    // equal or 1-3 % slower; blocks of 4 MFMAs followed by 50-60 filler instructions were equal too: with two waves per SIMD
    // the placement inside a chunk is not what limits the kernel (see the ablations in DESIGN.md).]
    auto mf = [&](int w, int m, bool first, bool keep) {
        const int t = m >> 2, i = (m >> 1) & 1, j = m & 1;
        if ((m & 3) == 0) {                                             // the prologue of component t, right in front of its four MFMAs
            if (PROA) { af[w][0][t] = fmaxf(fmaf(pa4[w][t], af[w][0][t], pc4[w][t]), 0.f); af[w][1][t] = fmaxf(fmaf(pa4[w][t], af[w][1][t], pc4[w][t]), 0.f); }
            if (PROB) { bf[w][0][t] = fmaxf(fmaf(qa[0], bf[w][0][t], qc[0]), 0.f); bf[w][1][t] = fmaxf(fmaf(qa[1], bf[w][1][t], qc[1]), 0.f); }
        }
        if (first) { const pf32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                     if (keep) accp[i][j] = acc[i][j];
                     acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[w][i][t], bf[w][j][t], z, 0, 0, 0); }
        else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[w][i][t], bf[w][j][t], acc[i][j], 0, 0, 0);
    };

    // ---- epilogue of the PREVIOUS tile, in pieces ----
    // Arithmetic in the ACCUMULATOR layout: register r of block (I, j) holds row 32 I + (r & 3) + 8 (r >> 2) + 4 h, column
    // 32 j + cl, so a lane owns ONE column per block column -- its statistics are plain per-lane sums and its four constants
    // sit in registers.  Sub-piece S = 4 I + Q: registers 4 Q .. 4 Q + 3 of blocks (I, 0), (I, 1) = rows 8 S + rr + 4 h of the
    // wave tile.  Memory in the ROW layout, through a wave-private 8 x 64 LDS buffer: the producer's Z rows of sub-piece S
    // are requested LDS-direct two chunks ahead (two 1 KB requests, no registers, the HBM latency of a whole chunk and a
    // half hidden -- measured: Z loaded into registers one chunk ahead cost 5 % of the kernel, eight 4-byte loads and
    // stores per chunk another 4 %), read back element-wise, the results overwrite them in place and leave as 16-byte
    // pieces of 256-byte row segments.
    __amdgpu_buffer_rsrc_t rZ, rC;                                       // descriptors at the wave tile's first row, first column
    const unsigned lane_zr = (unsigned)(((lane >> 4) * (int)p.ldz + (lane & 15) * 4) * 4), lane_cr = (unsigned)(((lane >> 4) * (int)p.ldc + (lane & 15) * 4) * 4);
    const int ldzb = (int)p.ldz * 4, ldcb = (int)p.ldc * 4;
    int z_so = 0, c_so = 0;                                              // byte offsets of the next sub-piece to request / to store (rows 8 S)
    auto set_prev = [&]() {
        const long long r0 = (long long)tm_p * P_BM + wm * 64;
        const int n0 = tn_p * P_BN + wn * 64;
        if (USE_Z) rZ = p_rsrc(p.Z + r0 * p.ldz + n0);
        rC = p_rsrc(p.C + ((long long)sp_p * p.M + r0) * p.ldc + n0);   // split-K: slab sp_p of the workspace
        z_so = 0; c_so = 0;
    };
    // (a) per-tile constants: the lane's two columns of (ea, ec, mean, invstd) -> registers; the wave's 64 dxyz rows -> LDS
    pf32x4 dqv;
    auto const_load = [&]() {
        const int c = tn_p * P_BN + wn * 64 + cl;
        if (BNR) {
#pragma unroll
            for (int j = 0; j < 2; ++j) { kc[0][j] = p.ea[c + 32 * j]; kc[1][j] = p.ec[c + 32 * j]; kc[2][j] = p.emean[c + 32 * j]; kc[3][j] = p.einvstd[c + 32 * j]; }
        }
        if (EPI == 1) { kc[0][0] = p.bias ? p.bias[c] : 0.f; kc[0][1] = p.bias ? p.bias[c + 32] : 0.f; }
        if (WQ) dqv = *(const pf32x4 *)(p.dxyz + ((long long)tm_p * P_BM + wm * 64 + lane) * 4);
    };
    constexpr int CONST_VM = (BNR ? 8 : 0) + (WQ ? 1 : 0);             // (the forward kind's 0 or 2 bias loads are not counted: under-counting is safe)
    auto const_store = [&]() { if (WQ) p_write128<0>(wbase + (unsigned)(lane * 16), dqv); };
    // Z rows of sub-piece S -> buffer S & 1 (rows 8 S .. 8 S + 3 and + 4 .. + 7: 16 lanes per row)
    auto zdma = [&](auto sc) {
        constexpr int S = decltype(sc)::value;
        float *zb = smem + P_NST * P_STAGE + wid * P_WSZ + P_WDQ + (S & 1) * 512;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rZ, (p_lptr)zb, 16, lane_zr, z_so, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rZ, (p_lptr)(zb + 256), 16, lane_zr, z_so + 4 * ldzb, 0, 0);
        z_so = p_spin(z_so + 8 * ldzb);
    };
    auto z_read = [&](auto sc) {                                         // after the vmcnt wait that covers zdma(S)
        constexpr int S = decltype(sc)::value, B = (S & 1) * 2048;
        if (!USE_Z) return;
        zq[0][0] = p_read32<B + 0>(w_za); zq[0][1] = p_read32<B + 128>(w_za); zq[1][0] = p_read32<B + 256>(w_za); zq[1][1] = p_read32<B + 384>(w_za);
        zq[2][0] = p_read32<B + 512>(w_za); zq[2][1] = p_read32<B + 640>(w_za); zq[3][0] = p_read32<B + 768>(w_za); zq[3][1] = p_read32<B + 896>(w_za);
    };
    auto z_pin = [&]() {
        if (!USE_Z) return;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) { p_pin1(zq[rr][0]); p_pin1(zq[rr][1]); }
    };
    pf32x4 d4[2];
    auto sub_lds_read = [&](auto sc, int rr0) {                          // dxyz rows of registers rr0, rr0 + 1 (two rows per read: h)
        constexpr int S = decltype(sc)::value;
        if (WQ) {
            if (rr0 == 0) { d4[0] = p_read128<(8 * S + 0) * 16>(w_dq); d4[1] = p_read128<(8 * S + 1) * 16>(w_dq); }
            else          { d4[0] = p_read128<(8 * S + 2) * 16>(w_dq); d4[1] = p_read128<(8 * S + 3) * 16>(w_dq); }
        }
    };
    auto sub_pin = [&]() { if (WQ) { p_pin(d4[0]); p_pin(d4[1]); } };
    // (b) arithmetic of one element (register rr, block column j) of sub-piece S (same operations per element as gemm.hip's
    // epilogues); sub_write puts the results of registers rr0, rr0 + 1 over the Z values in the buffer
    float xo[2][2];
    auto sub_elem = [&](auto sc, int rr, int j) {
        constexpr int S = decltype(sc)::value, I = S >> 2, Q = S & 3;
        float x = j == 0 ? accp[I][0][4 * Q + rr] : accp[I][1][4 * Q + rr];
        const float z = USE_Z ? zq[rr][j] : 0.f;
        if (EPI == 0) { }
        else if (EPI == 1) {
            x += kc[0][j];
            x = x > 0.f ? x : (slope == 0.f ? 0.f : slope * x);         // select, not 0 * x: -inf must give 0 like torch.relu
            t1[j] += x; t2[j] += x * x;
        } else if (BNR) {
            x = (fmaf(kc[0][j], z, kc[1][j]) > 0.f) ? x : 0.f;
            t1[j] += x; t2[j] += x * ((z - kc[2][j]) * kc[3][j]);
        } else {
            x = z > 0.f ? x : (slope == 0.f ? 0.f : slope * x);
            t1[j] += x;
        }
        if (WQ) { qs[0][j] += x * d4[rr & 1].x; qs[1][j] += x * d4[rr & 1].y; qs[2][j] += x * d4[rr & 1].z; }
        xo[rr & 1][j] = x;
    };
    auto sub_write = [&](auto sc, int rr0) {
        constexpr int B = USE_Z ? (decltype(sc)::value & 1) * 2048 : 0;
        if (rr0 == 0) { p_write32<B + 0>(w_za, xo[0][0]); p_write32<B + 128>(w_za, xo[0][1]); p_write32<B + 256>(w_za, xo[1][0]); p_write32<B + 384>(w_za, xo[1][1]); }
        else          { p_write32<B + 512>(w_za, xo[0][0]); p_write32<B + 640>(w_za, xo[0][1]); p_write32<B + 768>(w_za, xo[1][0]); p_write32<B + 896>(w_za, xo[1][1]); }
    };
    pf32x4 t4[2];
    auto out_read = [&](auto sc) { constexpr int B = USE_Z ? (decltype(sc)::value & 1) * 2048 : 0; t4[0] = p_read128<B>(w_zr); t4[1] = p_read128<B + 1024>(w_zr); };
    auto out_store = [&]() {                                             // behind the lgkmcnt wait + pin of t4
        typedef unsigned pu32x4 __attribute__((ext_vector_type(4)));
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pu32x4, t4[0]), rC, lane_cr, c_so, 0);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pu32x4, t4[1]), rC, lane_cr, c_so + 4 * ldcb, 0);
        c_so = p_spin(c_so + 8 * ldcb);
    };
    // (c) statistics: the lane's 32 rows -> the two half-waves (h = 0 + h = 1, what t + shfl_xor(t, 32) gives) ->
    // [wave row][statistic][column] in LDS; after a workgroup barrier the two wave rows are added and stored per 128-row tile
    auto stats_fold = [&]() {
        if (EPI == 0) return;
        float v[NSTAT][2];
        v[0][0] = t1[0]; v[0][1] = t1[1]; v[1][0] = t2[0]; v[1][1] = t2[1];
        if (WQ) {
#pragma unroll
            for (int k = 0; k < 3; ++k) { v[(2 + k) % NSTAT][0] = qs[k][0]; v[(2 + k) % NSTAT][1] = qs[k][1]; }
        }
        float u[NSTAT][2];
        if (h == 1) {
            p_write32<0 * 128>(w_fold, v[0][0]); p_write32<1 * 128>(w_fold, v[0][1]); p_write32<2 * 128>(w_fold, v[1][0]); p_write32<3 * 128>(w_fold, v[1][1]);
            if (WQ) { p_write32<4 * 128>(w_fold, v[2 % NSTAT][0]); p_write32<5 * 128>(w_fold, v[2 % NSTAT][1]); p_write32<6 * 128>(w_fold, v[3 % NSTAT][0]);
                      p_write32<7 * 128>(w_fold, v[3 % NSTAT][1]); p_write32<8 * 128>(w_fold, v[4 % NSTAT][0]); p_write32<9 * 128>(w_fold, v[4 % NSTAT][1]); }
        }
        u[0][0] = p_read32<0 * 128>(w_fold); u[0][1] = p_read32<1 * 128>(w_fold); u[1][0] = p_read32<2 * 128>(w_fold); u[1][1] = p_read32<3 * 128>(w_fold);
        if (WQ) { u[2 % NSTAT][0] = p_read32<4 * 128>(w_fold); u[2 % NSTAT][1] = p_read32<5 * 128>(w_fold); u[3 % NSTAT][0] = p_read32<6 * 128>(w_fold);
                  u[3 % NSTAT][1] = p_read32<7 * 128>(w_fold); u[4 % NSTAT][0] = p_read32<8 * 128>(w_fold); u[4 % NSTAT][1] = p_read32<9 * 128>(w_fold); }
        p_lds_wait();
#pragma unroll
        for (int k = 0; k < NSTAT; ++k) { p_pin1(u[k][0]); p_pin1(u[k][1]); }
        if (h == 0) {
            const unsigned a = red_base + (unsigned)((wm * NSTAT * 128 + wn * 64 + cl) * 4);
            p_write32<0 * 512>(a, v[0][0] + u[0][0]); p_write32<0 * 512 + 128>(a, v[0][1] + u[0][1]);
            p_write32<1 * 512>(a, v[1][0] + u[1][0]); p_write32<1 * 512 + 128>(a, v[1][1] + u[1][1]);
            if (WQ) { p_write32<2 * 512>(a, v[2 % NSTAT][0] + u[2 % NSTAT][0]); p_write32<2 * 512 + 128>(a, v[2 % NSTAT][1] + u[2 % NSTAT][1]);
                      p_write32<3 * 512>(a, v[3 % NSTAT][0] + u[3 % NSTAT][0]); p_write32<3 * 512 + 128>(a, v[3 % NSTAT][1] + u[3 % NSTAT][1]);
                      p_write32<4 * 512>(a, v[4 % NSTAT][0] + u[4 % NSTAT][0]); p_write32<4 * 512 + 128>(a, v[4 % NSTAT][1] + u[4 % NSTAT][1]); }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) { t1[j] = 0.f; t2[j] = 0.f; qs[0][j] = 0.f; qs[1][j] = 0.f; qs[2][j] = 0.f; }
    };
    auto stats_store = [&]() {                                          // after a workgroup barrier behind stats_fold
        if (EPI == 0) return;
        for (int c = tid; c < NSTAT * 128; c += P_THREADS) {
            const int which = c >> 7, cc = c & 127;
            const unsigned a = red_base + (unsigned)((which * 128 + cc) * 4);
            float v0 = p_read32<0>(a), v1 = p_read32<NSTAT * 128 * 4>(a);
            p_lds_wait();
            p_pin1(v0); p_pin1(v1);
            p.stats[((long long)tm_p * NSTAT + which) * p.N + tn_p * P_BN + cc] = v0 + v1;
        }
    };

    // ---- one chunk of the main loop ----
    // HP: a previous tile is being written out.  CI: position in the tile (0..10 peeled, -1 rolled: no epilogue work).
    // last (rolled chunks): the tile's final chunk -- the tile becomes the "previous" one and its first Z rows are requested.
    int st_read = 0;                                                    // stage of the chunk being multiplied
    unsigned kq_cur = 0;                                                // its place in the prologue tables (bytes)
    auto chunk = [&](auto hpc, auto cic, bool do_issue, bool last) {
        constexpr bool HP = decltype(hpc)::value;
        constexpr int CI = decltype(cic)::value;
        constexpr bool SUB = HP && CI >= 1 && CI <= P_NSUB;
        constexpr int S = SUB ? CI - 1 : 0;
        using SC = std::integral_constant<int, S>;
        using SC2 = std::integral_constant<int, (S + 2 < P_NSUB ? S + 2 : 0)>;
        // vector-memory LOADS of this wave that are YOUNGER than the requests the chunk barrier has to cover (those were issued in
        // the first half of the previous chunk): the second half of the previous chunk.  Must never be over-counted; stores are
        // not counted at all (loads retire in order among themselves -- nothing here relies on how stores retire relative to
        // loads; the price is that a wave also waits for its two 16-byte stores of the previous chunk: measured, none).
        //   final chunk of a tile: 2 (Z rows 0); chunk 0: constants + 2 (Z rows 1); chunks 1..6: 2 (Z rows S + 2)
        constexpr int ZD = USE_Z ? 2 : 0;
        constexpr int VM_PREV = !HP ? 0 : (CI == 0 ? ZD : (CI == 1 ? CONST_VM + ZD : (CI >= 2 && CI <= 7 ? ZD : 0)));
        // ---- first half: fragment set 0 ----
        // (set 1 of THIS chunk is requested here, not at the end of the previous chunk: no asynchronous LDS read may be in
        // flight at a control-flow edge -- at the loop back-edges the compiler copies fragment registers between its
        // per-block assignments, and a copy of a register whose data has not landed yet copies garbage: one launch in ten
        // multiplied a stale fragment, tools/pgemm_race.py)
        read_frags((unsigned)(st_read * P_STAGE * 4), kq_cur, std::integral_constant<int, 1>{});
        if (CI == 2) claim_publish();
        if (CI == 4) { const int sl = claim_read(); has_next = sl < nslots; if (has_next) tile_of(sl, tm_n, tn_n, sp_n); }
        P_SB();
        mf(0, 0, CI == 0, HP);
        if (do_issue) issue_part(0);
        P_SB();
        mf(0, 1, CI == 0, HP);
        if (do_issue) issue_part(1);
        P_SB();
        mf(0, 2, CI == 0, HP);
        if (do_issue) issue_part(2);
        P_SB();
        mf(0, 3, CI == 0, HP);
        if (do_issue) issue_part(3);
        P_SB();
        mf(0, 4, false, false); mf(0, 5, false, false);
        if (SUB) sub_lds_read(SC{}, 0);
        P_SB();
        mf(0, 6, false, false); mf(0, 7, false, false); mf(0, 8, false, false); mf(0, 9, false, false);
        mf(0, 10, false, false); mf(0, 11, false, false); mf(0, 12, false, false); mf(0, 13, false, false);
        mf(0, 14, false, false); mf(0, 15, false, false);
        p_lds_wait(); pin_frags(1);                                     // every LDS read of this chunk by this wave is complete
        if (SUB) sub_pin();
        // the next chunk's operands have landed (this wave's requests, then everybody's) -- and with them everything this
        // wave requested earlier: the Z rows of this chunk's sub-piece (two chunks old)
        if (do_issue) { P_WAIT_VMCNT(DMA + VM_PREV); } else { P_WAIT_VMCNT(VM_PREV); }
        __builtin_amdgcn_s_barrier();
        const int sn = st_read == 2 ? 0 : st_read + 1;
        const unsigned sbyte = (unsigned)(sn * P_STAGE * 4);
        kq_cur = (CI < 0 && last) ? 0u : kq_cur + (unsigned)(P_BK * 4); // the next chunk's place in the prologue tables (0: the next tile)
        read_frags(sbyte, kq_cur, std::integral_constant<int, 0>{});    // in flight under the second half
        if (SUB) z_read(SC{});
        P_SB();
        // ---- second half: fragment set 1 ----
        mf(1, 0, false, false);
        if (CI == 0) claim_issue();       // wave 0: one returning operation more than VM_PREV of chunk 1 counts (under-counting is safe)
        if (HP && CI == 0) const_load();
        P_SB();
        mf(1, 1, false, false);
        if (HP && CI == P_NSUB + 1 && want_stats) stats_fold();
        if (HP && CI == P_NSUB + 2 && want_stats) stats_store();
        P_SB();
        mf(1, 2, false, false);
        if (SUB) { p_lds_wait(); z_pin(); }
        P_SB();
        mf(1, 3, false, false);
        if (SUB) sub_elem(SC{}, 0, 0);
        P_SB();
        mf(1, 4, false, false);
        if (SUB) sub_elem(SC{}, 0, 1);
        P_SB();
        mf(1, 5, false, false);
        if (SUB) sub_elem(SC{}, 1, 0);
        P_SB();
        mf(1, 6, false, false);
        if (SUB) { sub_elem(SC{}, 1, 1); sub_write(SC{}, 0); sub_lds_read(SC{}, 2); }
        P_SB();
        mf(1, 7, false, false);
        mf(1, 8, false, false);
        if (SUB) { if (WQ) { p_lds_wait(); sub_pin(); } sub_elem(SC{}, 2, 0); }
        P_SB();
        mf(1, 9, false, false);
        if (SUB) sub_elem(SC{}, 2, 1);
        P_SB();
        mf(1, 10, false, false);
        if (SUB) sub_elem(SC{}, 3, 0);
        P_SB();
        mf(1, 11, false, false);
        if (SUB) { sub_elem(SC{}, 3, 1); sub_write(SC{}, 2); out_read(SC{}); }
        if (HP && CI == 0) const_store();
        if (CI < 0 && last) { tm_p = tm_c; tn_p = tn_c; sp_p = sp_c; set_prev(); if (USE_Z) zdma(std::integral_constant<int, 0>{}); }
        P_SB();
        mf(1, 12, false, false); mf(1, 13, false, false); mf(1, 14, false, false); mf(1, 15, false, false);
        p_lds_wait(); pin_frags(0);
        if (SUB) { p_pin(t4[0]); p_pin(t4[1]); out_store(); }
        if (HP && CI == 0 && USE_Z) zdma(std::integral_constant<int, 1>{});
        if (SUB && S + 2 < P_NSUB && USE_Z) zdma(SC2{});
        st_read = sn;
    };

    // ---- pipeline start: two chunks requested, the first one visible, its fragments on the way ----
    issue(); issue();
    if (PROA) {                                                         // scale | shift of every contraction index -> LDS, once
        for (int k = tid; k < p.K; k += P_THREADS) { p_write32<0>(pro_lds + (unsigned)(k * 4), p.pro_a[k]); p_write32<P_PROK * 4>(pro_lds + (unsigned)(k * 4), p.pro_c[k]); }
        p_lds_wait();
        P_WAIT_VMCNT(0);
    } else { P_WAIT_VMCNT(DMA); }
    __builtin_amdgcn_s_barrier();
    read_frags(0u, 0u, std::integral_constant<int, 0>{});
    p_lds_wait(); pin_frags(0);

    auto tile_body = [&](auto hpc) {
        using HPC = decltype(hpc);
        chunk(hpc, std::integral_constant<int, 0>{}, true, false);
        chunk(hpc, std::integral_constant<int, 1>{}, true, false);
        chunk(hpc, std::integral_constant<int, 2>{}, true, false);
        chunk(hpc, std::integral_constant<int, 3>{}, true, false);
        chunk(hpc, std::integral_constant<int, 4>{}, true, false);      // has_next, (tm_n, tn_n) known from here on
        chunk(hpc, std::integral_constant<int, 5>{}, true, false);
        chunk(hpc, std::integral_constant<int, 6>{}, true, false);
        chunk(hpc, std::integral_constant<int, 7>{}, true, false);
        chunk(hpc, std::integral_constant<int, 8>{}, true, false);
        chunk(hpc, std::integral_constant<int, 9>{}, true, false);
#pragma unroll 1
        for (int c = 10; c < nch; ++c) {
            if (c + 2 == nch) set_stream(tm_n, tn_n, sp_n);             // the next two requests belong to the next tile
            const bool go = c + 2 < nch || has_next;
            if (c == 10) chunk(hpc, std::integral_constant<int, 10>{}, go, false);
            else chunk(HPC{}, std::integral_constant<int, -1>{}, go, c + 1 == nch);
        }
    };

    auto load_q = [&]() {                                               // B prologue of the tile's columns (the first MFMA waits for it)
        if (PROB) { const int c = tn_c * P_BN + wn * 64 + cl; qa[0] = p.prob_a[c]; qa[1] = p.prob_a[c + 32]; qc[0] = p.prob_c[c]; qc[1] = p.prob_c[c + 32]; }
    };
    load_q();
    tile_body(std::false_type{});
    if (p.trace) t_first = wall_clock64();
    n_done = 1;
    while (has_next) {
        tm_c = tm_n; tn_c = tn_n; sp_c = sp_n;   // (the finished tile became the previous one in its final chunk; its accumulators move at
        load_q();                                //  the head of the next tile's first chunk)
        tile_body(std::true_type{});
        ++n_done;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) accp[i][j] = acc[i][j];

    if (p.trace) t_drain = wall_clock64();
    // ---- drain: the last tile's epilogue, nothing to hide it behind (its Z rows 0 are on their way) ----
    const_load();
    if (USE_Z) zdma(std::integral_constant<int, 1>{});
    const_store();
    auto drain_sub = [&](auto sc) {
        constexpr int S = decltype(sc)::value;
        // Z rows S have landed: younger LOADS than their request are (S = 0) the constants + Z rows 1, (S >= 1) Z rows S + 1
        if (USE_Z) { if (S == 0) { P_WAIT_VMCNT(CONST_VM + 2); } else if (S + 1 < P_NSUB) { P_WAIT_VMCNT(2); } else { P_WAIT_VMCNT(0); } }
        z_read(sc); sub_lds_read(sc, 0);
        p_lds_wait(); z_pin(); sub_pin();
        sub_elem(sc, 0, 0); sub_elem(sc, 0, 1); sub_elem(sc, 1, 0); sub_elem(sc, 1, 1); sub_write(sc, 0);
        sub_lds_read(sc, 2);
        p_lds_wait(); sub_pin();
        sub_elem(sc, 2, 0); sub_elem(sc, 2, 1); sub_elem(sc, 3, 0); sub_elem(sc, 3, 1); sub_write(sc, 2);
        out_read(sc);
        p_lds_wait(); p_pin(t4[0]); p_pin(t4[1]);
        out_store();
        if (S + 2 < P_NSUB && USE_Z) zdma(std::integral_constant<int, (S + 2 < P_NSUB ? S + 2 : 0)>{});
    };
    drain_sub(std::integral_constant<int, 0>{}); drain_sub(std::integral_constant<int, 1>{});
    drain_sub(std::integral_constant<int, 2>{}); drain_sub(std::integral_constant<int, 3>{});
    drain_sub(std::integral_constant<int, 4>{}); drain_sub(std::integral_constant<int, 5>{});
    drain_sub(std::integral_constant<int, 6>{}); drain_sub(std::integral_constant<int, 7>{});
    if (want_stats) {
        stats_fold();
        p_lds_wait();
        __builtin_amdgcn_s_barrier();
        stats_store();
    }
    leave();
    if (p.trace && tid == 0) {                                           // diagnostics (tools/pgemm_timeline.py)
        unsigned long long *r = p.trace + 8ull * blockIdx.x;
        r[0] = t_start; r[1] = t_drain; r[2] = wall_clock64();
        r[3] = ((unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) << 32) | (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
        r[4] = t_first; r[5] = (unsigned long long)n_done; r[6] = 0ull; r[7] = 0ull;
    }
}

// ---- host side ----
namespace {
std::atomic<int> g_pmode{-1};        // -1: from the environment on first use; 0 off; 1 auto (default); 2 wherever the shape allows
std::atomic<int> g_pgrid{0};         // 0: two workgroups per CU
int pmode()
{
    int m = g_pmode.load(std::memory_order_relaxed);
    if (m < 0) {
        const char *e = getenv("CMF_GEMM_PERSIST");
        m = e ? atoi(e) : 1;
        if (m < 0 || m > 2) m = 1;
        g_pmode.store(m);
    }
    return m;
}
}  // namespace

// mode: 0 never, 1 where it pays (default), 2 wherever the shape allows (tests); grid: workgroups (multiple of 8), 0 = 2 per CU
extern "C" int cmf_gemm_persist_config(int mode, int grid)
{
    if (mode < 0 || mode > 2 || grid < 0 || (grid % 8) != 0) return -1;
    g_pmode.store(mode); g_pgrid.store(grid);
    return 0;
}

// One block of tile counters per (device, stream): kernels of one stream run one after the other and each leaves the block
// zeroed, so no launch ever sees another one's counts (two launches sharing a block would have to run concurrently).
#include <map>
#include <mutex>
static int *sched_block(hipStream_t st)
{
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, int *> *table = new std::map<std::pair<int, hipStream_t>, int *>();   // leaked on purpose
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    int *&e = (*table)[std::make_pair(dev, st)];
    if (!e) {
        // zeroed IN STREAM ORDER on the stream whose launches use the block (side streams are non-blocking: nothing orders them behind a
        // null-stream hipMemset), once, ahead of that stream's first persistent launch
        if (hipMalloc((void **)&e, 64) != hipSuccess || hipMemsetAsync(e, 0, 64, st) != hipSuccess) { (void)hipGetLastError(); e = nullptr; }
    }
    return e;
}

template <bool A_T, bool B_T, int EPI, bool PRO>
static int plaunch(const GemmArgs &a, int grid, hipStream_t st)
{
    const size_t lds = (size_t)p_lds_floats(EPI, PRO && !A_T) * sizeof(float);
    static std::atomic<unsigned> set_mask[4];
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned bit = 1u << (dev & 31);
    if (dev >= 128 || !(set_mask[dev >> 5].load(std::memory_order_acquire) & bit)) {
        if (hipFuncSetAttribute((const void *)pgemm_kernel<A_T, B_T, EPI, PRO>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return (int)hipGetLastError();
        if (dev < 128) set_mask[dev >> 5].fetch_or(bit, std::memory_order_release);
    }
    int *sched = sched_block(st);
    if (!sched) return (int)hipErrorOutOfMemory;
    hipLaunchKernelGGL((pgemm_kernel<A_T, B_T, EPI, PRO>), dim3(grid), dim3(P_THREADS), lds, st, a, sched);
    return cmf_launch_status();
}

// Workgroups of the persistent kernel for this call, or 0 when the call is not one it takes (the caller then uses gemm.hip's
// kernels).  kind: gemm.hip epilogue_kind of the call.  Instantiated (layout, kind) pairs: data gradients A[M][K] B[K][N] with
// kinds 0, 2-5; forward A[M][K] W[N][K] with kinds 0, 1 (+ the A prologue); weight gradients A[K][M] B[K][N] as split-K slabs
// (+ the B prologue) -- for those g.split_k may be LOWERED so that tiles x slabs fills the grid exactly once.
int cmf_pgemm_grid(GemmArgs &g, int a_t, int b_t, int kind)
{
    const int mode = pmode();
    if (mode == 0 || (a_t && b_t)) return 0;
    if (g.M % P_BM || g.N % P_BN || g.K % P_BK || g.accumulate || g.no_direct || g.diag || g.bnb_z) return 0;
    if (g.ldc % 4 || (uintptr_t)g.C % 16) return 0;
    if (!a_t && !b_t && !(kind == 0 || (kind >= 2 && kind <= 5))) return 0;
    if (!a_t && b_t && kind > 1) return 0;
    if (a_t && (kind != 0 || g.pro_a)) return 0;
    if (!a_t && g.prob_a) return 0;
    if (g.pro_a && (g.K > P_PROK || a_t || !b_t)) return 0;
    if (kind >= 2 && (g.ldz % 4 || (uintptr_t)g.Z % 16)) return 0;
    if (kind >= 4 && !g.dxyz) return 0;
    if (kind == 1 && g.act == 3) return 0;                            // sigmoid heads: generic epilogue
    if (mode == 1) {
        // Default: the data gradients with a backward epilogue only.  Same-box A/B against the tiled kernel (tools/pgemm_bench.py,
        // profiles/r04_pgemm_bench.txt): data gradients + 3-10 % at K = 256 (+ 1-3 % at K = 512); weight gradients with the B
        // prologue + 3-4 % (without: equal); forward without the A prologue + 0-3 % (with it: - 2-7 %) -- and inside the
        // training step, where four chains share the chip, only the data gradients moved the step (21.5 vs 21.7 ms; with
        // the weight-gradient and forward forms added: 21.5-21.7 either way).  Mode 2 keeps every form reachable (tests).
        if (a_t || b_t || kind < 2) return 0;
    }
    // 32-bit byte offsets inside a tile's panels: operand streams (rows of the panel + the whole contraction), 64 output rows
    const long long lim = 1ll << 31;
    const long long spanA = a_t ? ((long long)g.K + 16) * g.lda * 4 : 128 * g.lda * 4 + (long long)g.K * 4;
    const long long spanB = b_t ? 128 * g.ldb * 4 + (long long)g.K * 4 : ((long long)g.K + 16) * g.ldb * 4;
    if (spanA >= lim || spanB >= lim || 72 * g.ldz * 4 >= lim || 72 * g.ldc * 4 >= lim) return 0;
    int dev = 0;
    (void)hipGetDevice(&dev);
    static std::atomic<int> cus[64];
    int n = dev < 64 ? cus[dev].load(std::memory_order_relaxed) : 0;
    if (!n) { hipDeviceProp_t pr; if (hipGetDeviceProperties(&pr, dev) != hipSuccess) { (void)hipGetLastError(); return 0; } n = pr.multiProcessorCount; if (dev < 64) cus[dev].store(n); }
    int grid = g_pgrid.load(std::memory_order_relaxed);
    if (grid <= 0) grid = 2 * n;
    grid = (grid + 7) / 8 * 8;
    const long long tiles = (long long)(g.M / P_BM) * (g.N / P_BN);
    const int chunks = g.K / P_BK;
    if (a_t) {
        // weight gradient: one (tile, slab) per workgroup -- the largest slab count <= the caller's (its workspace holds that
        // many) that is a multiple of 8 (slabs are dealt to the XCDs), divides the contraction and keeps tiles x slabs <= grid
        if (g.split_k < 8 || tiles > grid / 8) return 0;
        int s = (int)std::min<long long>(g.split_k, grid / tiles) / 8 * 8;
        while (s >= 8 && (chunks % s || chunks / s < P_MIN_CHUNKS)) s -= 8;
        if (s < 8) return 0;
        if (mode == 1 && tiles * s < grid) return 0;                  // the grid would not be full
        g.split_k = s;
        return grid;
    }
    if (g.split_k != 1 || chunks < P_MIN_CHUNKS) return 0;
    if (mode == 1 && tiles < 3ll * grid) return 0;                   // fewer than three tiles per workgroup: the pipeline never fills
    return grid;
}

int cmf_pgemm_launch(const GemmArgs &g, int a_t, int b_t, int kind, int grid, hipStream_t st)
{
    if (a_t) return g.prob_a ? plaunch<true, false, 0, true>(g, grid, st) : plaunch<true, false, 0, false>(g, grid, st);
    if (b_t) {
        if (kind == 0) return g.pro_a ? plaunch<false, true, 0, true>(g, grid, st) : plaunch<false, true, 0, false>(g, grid, st);
        return g.pro_a ? plaunch<false, true, 1, true>(g, grid, st) : plaunch<false, true, 1, false>(g, grid, st);
    }
    switch (kind) {
    case 0: return plaunch<false, false, 0, false>(g, grid, st);
    case 2: return plaunch<false, false, 2, false>(g, grid, st);
    case 3: return plaunch<false, false, 3, false>(g, grid, st);
    case 4: return plaunch<false, false, 4, false>(g, grid, st);
    default: return plaunch<false, false, 5, false>(g, grid, st);
    }
}
