// Neighbour search kernels: ball query (lib/src/ball_query_gpu.cu:9-45) and the torch-level
// kNN of utils/model_utils/radarflow_util.py:8-30,88-99.
//
// Both decide integer outputs from fp32 compares, so the arithmetic is CANONICAL:
// every operation individually rounded, no FMA contraction, except the kNN dot product which
// is the k-ordered FMA chain of a GEMM (bit-equal to torch-CPU matmul).  This file is built
// with -ffp-contract=off and the pragma below; tests/test_build.py checks the ISA.
#include <cstdlib>
#include <mutex>
#include <algorithm>
#include "cmf_common.h"
#include "../../include/cmflow_hip.h"

#pragma clang fp contract(off)

// ---------------------------------------------------------------------------------------------
// Ball query.  One lane per centre; the cloud is staged through LDS in tiles and every lane
// scans it in index order (all lanes read the same LDS address -> broadcast, conflict free).
// The scan order is the semantics (first nsample hits in index order), so there is no
// reordering to exploit; the kernel is latency bound at N=256.
// ---------------------------------------------------------------------------------------------
constexpr int BQ_TILE = 1024;       // points per LDS tile (16 KiB as float4)
constexpr int BQ_MAX_NS_LDS = 64;   // hit lists up to this nsample are buffered in LDS and written coalesced

// One lane per centre.  Points are staged in LDS as float4 so a point is ONE ds_read_b128 that every lane
// of the wave reads at the same address (broadcast, conflict free); the scan is unrolled by 4 so the LDS
// latency of the next points hides under the distance arithmetic.  Hits are collected per lane in LDS
// ([slot][lane] layout: conflict free) and written out at the end with coalesced stores -- the 64
// centres of a wave own one contiguous 64*nsample block of idx.
template <bool HITS_IN_LDS>
__global__ __launch_bounds__(CMF_WAVE) void ball_query_kernel(
    int n, int m, float radius2, int nsample,
    const float *__restrict__ new_xyz, const float *__restrict__ xyz, int *__restrict__ idx)
{
    __shared__ float4 tile[BQ_TILE];
    __shared__ int hits[HITS_IN_LDS ? BQ_MAX_NS_LDS * CMF_WAVE : 1];
    const int bs = blockIdx.y;
    const int lane = threadIdx.x;
    const int pt = blockIdx.x * CMF_WAVE + lane;
    const bool live = pt < m;
    const float *pts = xyz + (size_t)bs * n * 3;
    float cx = 0.f, cy = 0.f, cz = 0.f;
    if (live) {
        const float *c = new_xyz + ((size_t)bs * m + pt) * 3;
        cx = c[0]; cy = c[1]; cz = c[2];
    }
    int *out = idx + ((size_t)bs * m + (live ? pt : 0)) * nsample;
    int cnt = live ? 0 : nsample;       // dead lanes count as finished
    int first = 0;

    auto dist2 = [&](const float4 p) {
        const float dx = cx - p.x;
        const float dy = cy - p.y;
        const float dz = cz - p.z;
        const float xx = dx * dx;
        const float yy = dy * dy;
        const float zz = dz * dz;
        const float s = xx + yy;
        return s + zz;
    };
    auto take = [&](bool hit, int k) {
        if (hit && cnt < nsample) {
            if (cnt == 0) first = k;
            if (HITS_IN_LDS) hits[cnt * CMF_WAVE + lane] = k; else out[cnt] = k;
            ++cnt;
        }
    };

    for (int base = 0; base < n; base += BQ_TILE) {
        const int len = min(BQ_TILE, n - base);
        __syncthreads();
        for (int i = lane; i < len; i += CMF_WAVE) {
            const float *q = pts + (size_t)(base + i) * 3;
            tile[i] = make_float4(q[0], q[1], q[2], 0.f);
        }
        __syncthreads();
        if (__all(cnt >= nsample)) break;           // wave-uniform early exit
        int k = 0;
        for (; k + 4 <= len; k += 4) {
            const float4 p0 = tile[k], p1 = tile[k + 1], p2 = tile[k + 2], p3 = tile[k + 3];
            const bool h0 = dist2(p0) < radius2, h1 = dist2(p1) < radius2;
            const bool h2 = dist2(p2) < radius2, h3 = dist2(p3) < radius2;
            // hits are rare (<= nsample of n): one wave-uniform branch skips the bookkeeping for 4 points
            if (__any(h0 | h1 | h2 | h3)) {
                take(h0, base + k); take(h1, base + k + 1); take(h2, base + k + 2); take(h3, base + k + 3);
            }
        }
        for (; k < len; ++k) take(dist2(tile[k]) < radius2, base + k);
    }
    // pad with the first hit (ball_query_gpu.cu:37-41); an empty ball leaves idx untouched
    if (!HITS_IN_LDS) {
        if (live && cnt > 0)
            for (int l = cnt; l < nsample; ++l) out[l] = first;
        return;
    }
    if (live && cnt > 0)
        for (int l = cnt; l < nsample; ++l) hits[l * CMF_WAVE + lane] = first;
    // coalesced write-out of the wave's 64 x nsample block; lanes with an empty ball keep idx as it was
    const unsigned long long nonempty = __ballot(live && cnt > 0);
    const int total = CMF_WAVE * nsample;
    int *blk = idx + ((size_t)bs * m + (size_t)blockIdx.x * CMF_WAVE) * nsample;
    for (int e = lane; e < total; e += CMF_WAVE) {
        const int c = e / nsample, sl = e - c * nsample;
        if ((nonempty >> c) & 1ull) blk[e] = hits[sl * CMF_WAVE + c];
    }
}

// Multi-wave variant: the NW waves of a workgroup share the same 64 centres and scan consecutive index ranges of the
// cloud (wave w: points [w*seg, (w+1)*seg)), each collecting up to nsample hits of its range in LDS; the write-out
// concatenates the lists in wave order, which IS index order, truncated at nsample and padded with the first hit.
// One wave per 64 centres leaves the chip at 1-2 waves per SIMD (N = 256: 256 waves in total; N = 4096: 2048) and
// the scan is a dependent instruction stream: splitting the range gives NW times the waves and 1/NW the loop length.
constexpr int BQM_NW = 4;
constexpr int BQM_TILE = 256;       // points per wave per LDS tile

typedef float bq_f2 __attribute__((ext_vector_type(2)));

// HITS_GLOBAL: the per-wave hit lists live in a global scratch buffer instead of LDS.  At nsample = 64 the lists
// are 8 KB per wave and cap the kernel at 3 waves per SIMD; hits are rare events (<= nsample stores per lane over
// the whole scan), so spilling them costs nothing and lets the scan run at full occupancy.
template <bool HITS_GLOBAL>
__global__ __launch_bounds__(BQM_NW *CMF_WAVE) void ball_query_multi_kernel(
    int n, int m, float radius2, int nsample, int seg,
    const float *__restrict__ new_xyz, const float *__restrict__ xyz, int *__restrict__ idx,
    unsigned short *__restrict__ scratch)
{
    // points are staged as PAIRS, (x0,x1,y0,y1) + (z0,z1), and the distance of both is evaluated with packed fp32
    // instructions (v_pk_add_f32 / v_pk_mul_f32: two individually rounded operations per instruction -- the same
    // canonical arithmetic at half the VALU issue)
    __shared__ float4 txy[BQM_NW][BQM_TILE / 2];
    __shared__ bq_f2 tz[BQM_NW][BQM_TILE / 2];
    __shared__ unsigned short hits_lds[HITS_GLOBAL ? 1 : BQM_NW * BQ_MAX_NS_LDS * CMF_WAVE];
    __shared__ int cntw[BQM_NW][CMF_WAVE];
    const int bs = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int pt = blockIdx.x * CMF_WAVE + lane;
    const bool live = pt < m;
    const float *pts = xyz + (size_t)bs * n * 3;
    // hit lists of the 4 waves, [wave][slot][lane]
    unsigned short *hits = HITS_GLOBAL ? scratch + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (BQM_NW * BQ_MAX_NS_LDS * CMF_WAVE)
                                       : hits_lds;
    unsigned short *mine = hits + w * (BQ_MAX_NS_LDS * CMF_WAVE);
    float cx = 0.f, cy = 0.f, cz = 0.f;
    if (live) {
        const float *c = new_xyz + ((size_t)bs * m + pt) * 3;
        cx = c[0]; cy = c[1]; cz = c[2];
    }
    const bq_f2 cxx = {cx, cx}, cyy = {cy, cy}, czz = {cz, cz};
    int cnt = live ? 0 : nsample;
    auto dist2 = [&](const float4 a, const bq_f2 z) {
        const bq_f2 px = {a.x, a.y}, py = {a.z, a.w};
        const bq_f2 dx = cxx - px;
        const bq_f2 dy = cyy - py;
        const bq_f2 dz = czz - z;
        const bq_f2 xx = dx * dx;
        const bq_f2 yy = dy * dy;
        const bq_f2 zz = dz * dz;
        const bq_f2 s = xx + yy;
        return s + zz;
    };
    auto take = [&](bool hit, int k) {
        if (hit && cnt < nsample) { mine[cnt * CMF_WAVE + lane] = (unsigned short)k; ++cnt; }
    };
    const int r0 = min(n, w * seg), r1 = min(n, r0 + seg);
    const float far = __builtin_inff();                         // filler of an odd tail: never inside a ball
    for (int base = r0; base < r1; base += BQM_TILE) {
        const int len = min(BQM_TILE, r1 - base);
        const int pairs = (len + 1) / 2;
        for (int i = lane; i < pairs; i += CMF_WAVE) {            // this wave's own tile: no workgroup barrier needed
            const float *q = pts + (size_t)(base + 2 * i) * 3;
            const bool two = 2 * i + 1 < len;
            txy[w][i] = make_float4(q[0], two ? q[3] : far, q[1], two ? q[4] : far);
            tz[w][i] = bq_f2{q[2], two ? q[5] : far};
        }
        __builtin_amdgcn_wave_barrier();
        if (__all(cnt >= nsample)) break;
        int k = 0;
        for (; k + 2 <= pairs; k += 2) {
            const bq_f2 d0 = dist2(txy[w][k], tz[w][k]), d1 = dist2(txy[w][k + 1], tz[w][k + 1]);
            const bool h0 = d0.x < radius2, h1 = d0.y < radius2, h2 = d1.x < radius2, h3 = d1.y < radius2;
            if (__any(h0 | h1 | h2 | h3)) {
                take(h0, base + 2 * k); take(h1, base + 2 * k + 1); take(h2, base + 2 * k + 2); take(h3, base + 2 * k + 3);
            }
        }
        for (; k < pairs; ++k) {
            const bq_f2 d0 = dist2(txy[w][k], tz[w][k]);
            take(d0.x < radius2, base + 2 * k); take(d0.y < radius2, base + 2 * k + 1);
        }
        __builtin_amdgcn_wave_barrier();
    }
    cntw[w][lane] = live ? cnt : 0;
    if (HITS_GLOBAL) __threadfence_block();
    __syncthreads();
    // write-out: the workgroup's 64 x nsample block of idx, coalesced; centres with an empty ball keep idx as it was
    const int total = CMF_WAVE * nsample;
    const int valid = min(CMF_WAVE, m - blockIdx.x * CMF_WAVE);
    int *blk = idx + ((size_t)bs * m + (size_t)blockIdx.x * CMF_WAVE) * nsample;
    for (int e = tid; e < total; e += BQM_NW * CMF_WAVE) {
        const int c = e / nsample, sl = e - c * nsample;
        if (c >= valid) break;
        int start = 0, val = -1, first = -1;
#pragma unroll
        for (int q = 0; q < BQM_NW; ++q) {
            const int cq = cntw[q][c];
            const unsigned short *lst = hits + q * (BQ_MAX_NS_LDS * CMF_WAVE);
            if (first < 0 && cq > 0) first = lst[c];
            if (val < 0 && sl < start + cq) val = lst[(sl - start) * CMF_WAVE + c];
            start += cq;
        }
        if (first >= 0) blk[e] = val >= 0 ? val : first;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Ball query over a cell grid, for large clouds (4096 <= N <= 8192; config 5 of BASELINE: N = 4096, nsample 64).
//
// "The first nsample points within the radius IN INDEX ORDER" makes the scan order part of the result, which is why the
// kernels above walk the whole cloud per centre (N^2 distance evaluations).  The order can be restored afterwards
// instead: (1) per sample, the points are binned into cubic cells of edge >= 1.001 * radius (at most 16 per axis) --
// a centre's hits can only lie in the 27 cells around its own; (2) one WAVE per centre evaluates the distance (same
// canonical arithmetic) of the points of those cells, lane-parallel, in whatever order the cells hold them, and sets bit
// `index` of an N-bit map in LDS for every hit; (3) the map is read back in index order: lane l owns words l, l + 64, ...,
// a wave prefix sum of the population counts gives every lane the rank of its first hit, and the first nsample set
// bits are written out -- exactly the list the scan would have produced, padded with the first hit like
// ball_query_gpu.cu:37-41; an empty ball leaves idx untouched.  Work per centre: the points of 27 cells instead of N.
// ---------------------------------------------------------------------------------------------------------------
constexpr int BQG_MAXC = 16;                 // cells per axis
constexpr int BQG_CELLS = BQG_MAXC * BQG_MAXC * BQG_MAXC;
constexpr int BQG_THREADS = 1024;
constexpr int BQG_MAX_N = 8192;              // bitmap of 256 words per wave
struct BqGridHeader { float mn[3], inv; int g[3], pad; };      // per sample, in scratch

__device__ __forceinline__ int bqg_cell(float v, float mn, float inv, int g)     // raw cell coordinate, clamped to [-1, g]
{
    const float t = floorf((v - mn) * inv);
    return t < -1.f ? -1 : (t > (float)g ? g : (int)t);
}

// one workgroup per sample: bounding box, cell of every point, counting sort (order inside a cell is arbitrary)
__global__ __launch_bounds__(BQG_THREADS) void bq_grid_build_kernel(int n, float radius, const float *__restrict__ xyz,
                                                                    BqGridHeader *__restrict__ hdr, int *__restrict__ cell_start,
                                                                    float4 *__restrict__ spts)
{
    __shared__ float red[6][BQG_THREADS / 64];
    __shared__ int cnt[BQG_CELLS + 1];
    __shared__ BqGridHeader H;
    const int bs = blockIdx.x, tid = threadIdx.x;
    const float *pts = xyz + (size_t)bs * n * 3;
    float lo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()}, hi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    for (int i = tid; i < n; i += BQG_THREADS)
#pragma unroll
        for (int k = 0; k < 3; ++k) { const float v = pts[(size_t)i * 3 + k]; lo[k] = fminf(lo[k], v); hi[k] = fmaxf(hi[k], v); }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { lo[k] = fminf(lo[k], __shfl_xor(lo[k], off, 64)); hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off, 64)); }
        if ((tid & 63) == 0) { red[k][tid >> 6] = lo[k]; red[3 + k][tid >> 6] = hi[k]; }
    }
    __syncthreads();
    if (tid == 0) {
        float ext[3];
        for (int k = 0; k < 3; ++k) {
            float a = red[k][0], b = red[3 + k][0];
            for (int w = 1; w < BQG_THREADS / 64; ++w) { a = fminf(a, red[k][w]); b = fmaxf(b, red[3 + k][w]); }
            H.mn[k] = a; ext[k] = b - a;
        }
        // edge > radius (|dq| < 1 cell whatever the rounding) and at most 16 cells per axis.  [A finer grid (8192 cells of
        // any shape) was measured slower: the query is bound by its instruction count per (y, z) row, not by candidates.]
        const float cs = fmaxf(radius * 1.001f, fmaxf(ext[0], fmaxf(ext[1], ext[2])) / (float)BQG_MAXC);
        H.inv = 1.0f / cs;
        for (int k = 0; k < 3; ++k) {
            const int g = (int)floorf(ext[k] * H.inv) + 1;
            H.g[k] = g < 1 ? 1 : (g > BQG_MAXC ? BQG_MAXC : g);
        }
        H.pad = 0;
        hdr[bs] = H;
    }
    for (int i = tid; i <= BQG_CELLS; i += BQG_THREADS) cnt[i] = 0;
    __syncthreads();
    const int gx = H.g[0], gy = H.g[1], gz = H.g[2];
    auto cell_of = [&](int i) {
        const float *q = pts + (size_t)i * 3;
        const int cx = min(gx - 1, max(0, bqg_cell(q[0], H.mn[0], H.inv, gx)));
        const int cy = min(gy - 1, max(0, bqg_cell(q[1], H.mn[1], H.inv, gy)));
        const int cz = min(gz - 1, max(0, bqg_cell(q[2], H.mn[2], H.inv, gz)));
        return (cz * gy + cy) * gx + cx;
    };
    for (int i = tid; i < n; i += BQG_THREADS) atomicAdd(&cnt[cell_of(i) + 1], 1);
    __syncthreads();
    const int ncell = gx * gy * gz;
    {   // inclusive scan of cnt[1 .. BQG_CELLS]: 4 cells per thread, wave scan, then the 16 wave totals
        constexpr int PER = BQG_CELLS / BQG_THREADS;
        __shared__ int wsum[BQG_THREADS / 64];
        int v[PER], run = 0;
#pragma unroll
        for (int k = 0; k < PER; ++k) { run += cnt[1 + PER * tid + k]; v[k] = run; }
        int incl = run;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(incl, off, 64); if ((tid & 63) >= off) incl += t; }
        if ((tid & 63) == 63) wsum[tid >> 6] = incl;
        __syncthreads();
        int before = incl - run;
        for (int w = 0; w < (tid >> 6); ++w) before += wsum[w];
#pragma unroll
        for (int k = 0; k < PER; ++k) cnt[1 + PER * tid + k] = before + v[k];
    }
    __syncthreads();
    int *cs_out = cell_start + (size_t)bs * (BQG_CELLS + 1);
    for (int c = tid; c <= ncell; c += BQG_THREADS) cs_out[c] = cnt[c];
    __syncthreads();
    float4 *sp = spts + (size_t)bs * n;
    for (int i = tid; i < n; i += BQG_THREADS) {
        const int slot = atomicAdd(&cnt[cell_of(i)], 1);                           // cursor of the cell (cnt[c] = its start)
        const float *q = pts + (size_t)i * 3;
        sp[slot] = make_float4(q[0], q[1], q[2], __int_as_float(i));
    }
}

// one wave per centre (4 per workgroup)
__global__ __launch_bounds__(256) void bq_grid_query_kernel(int n, int m, float radius2, int nsample, const float *__restrict__ new_xyz,
                                                            const BqGridHeader *__restrict__ hdr, const int *__restrict__ cell_start,
                                                            const float4 *__restrict__ spts, int *__restrict__ idx, int zero_empty)
{
    __shared__ unsigned bits[4][BQG_MAX_N / 32];
    const int bs = blockIdx.y, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int pt = blockIdx.x * 4 + w;
    if (pt >= m) return;                                                           // wave-uniform; no workgroup barrier below
    const BqGridHeader H = hdr[bs];
    const int *cst = cell_start + (size_t)bs * (BQG_CELLS + 1);
    const float4 *sp = spts + (size_t)bs * n;
    const float *c = new_xyz + ((size_t)bs * m + pt) * 3;
    const float cx = c[0], cy = c[1], cz = c[2];
    const int words = (n + 31) / 32;
    unsigned *bm = bits[w];
    for (int i = lane; i < words; i += 64) bm[i] = 0u;
    __builtin_amdgcn_wave_barrier();
    const int qx = bqg_cell(cx, H.mn[0], H.inv, H.g[0]), qy = bqg_cell(cy, H.mn[1], H.inv, H.g[1]), qz = bqg_cell(cz, H.mn[2], H.inv, H.g[2]);
    const int x0 = max(0, qx - 1), x1 = min(H.g[0] - 1, qx + 1);
    for (int z = max(0, qz - 1); z <= min(H.g[2] - 1, qz + 1); ++z)
        for (int y = max(0, qy - 1); y <= min(H.g[1] - 1, qy + 1); ++y) {
            if (x0 > x1) continue;
            const int row = (z * H.g[1] + y) * H.g[0];
            const int e0 = cst[row + x0], e1 = cst[row + x1 + 1];                  // the x-neighbours are one contiguous range
            for (int e = e0 + lane; e < e1; e += 64) {
                const float4 p = sp[e];
                const float dx = cx - p.x;
                const float dy = cy - p.y;
                const float dz = cz - p.z;
                const float xx = dx * dx;
                const float yy = dy * dy;
                const float zz = dz * dz;
                const float s2 = xx + yy;
                if (s2 + zz < radius2) {
                    const int k = __float_as_int(p.w);
                    atomicOr(&bm[k >> 5], 1u << (k & 31));
                }
            }
        }
    __builtin_amdgcn_wave_barrier();
    // read the map back in index order
    int *out = idx + ((size_t)bs * m + pt) * nsample;
    int total = 0, first = -1;
    for (int base = 0; base < words && total < nsample; base += 64) {
        const int wi = base + lane;
        unsigned v = wi < words ? bm[wi] : 0u;
        const int cntl = __popc(v);
        int incl = cntl;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(incl, off, 64); if (lane >= off) incl += t; }
        int pos = total + incl - cntl;
        const unsigned long long any = __ballot(cntl > 0);
        if (first < 0 && any) {
            const int fl = __ffsll((long long)any) - 1;                            // lowest lane with a hit holds the smallest index
            const unsigned fv = __shfl(v, fl, 64);
            first = (base + fl) * 32 + (__ffs((int)fv) - 1);
        }
        while (v && pos < nsample) {
            const int bit = __ffs((int)v) - 1;
            out[pos++] = wi * 32 + bit;
            v &= v - 1;
        }
        total += __shfl(incl, 63, 64);
    }
    if (total > 0 && total < nsample)
        for (int l = total + lane; l < nsample; l += 64) out[l] = first;           // ball_query_gpu.cu:37-41
    if (total == 0 && zero_empty)                                                  // idx not pre-zeroed by the caller (cmf_query_and_group)
        for (int l = lane; l < nsample; l += 64) out[l] = 0;
}

// ---------------------------------------------------------------------------------------------------------------
// Ball query for small clouds (n <= 1024: every shape of the model), ballot form.  The scan kernels above give a lane
// one centre and walk the cloud in a dependent loop (n = 256: 64 + points per wave, ~10 us per launch whatever the
// batch).  Here a WAVE owns a centre and the cloud lives in its registers: lane l holds points 4l .. 4l+3 of every
// 256-point chunk, evaluates their distances (same canonical arithmetic), and four ballots per chunk ARE the hit list in
// index order -- a hit's slot is (hits of earlier chunks) + (hits in lower lanes) + (own earlier hits): two mbcnt
// pairs and a few adds, no loop over points, no atomics.  A wave serves G consecutive centres from the same registers
// and collects their lists in a wave-private LDS block; the write-out of the G x nsample block is coalesced.
//   first nsample hits in index order, padded with the first hit, strict d2 < r^2 (ball_query_gpu.cu:29-41)
// ---------------------------------------------------------------------------------------------------------------
constexpr int BQB_WAVES = 4;                 // waves per workgroup (independent of each other)
constexpr int BQB_CHUNK = 256;               // points per register chunk (4 per lane)
constexpr int BQB_MAX_N = 1024;
constexpr size_t BQB_LDS_LIMIT = 64 * 1024;     // dynamic LDS of one launch (no hipFuncSetAttribute)

template <int NCH>
struct BqbCloud { float x[NCH][4], y[NCH][4], z[NCH][4]; };

template <int NCH>
__device__ __forceinline__ void bqb_load_cloud(const float *__restrict__ pts, int n, int lane, BqbCloud<NCH> &c)
{
    const float far = __builtin_inff();                       // out-of-range slots: never inside a ball (inf < r2 is false)
#pragma unroll
    for (int k = 0; k < NCH; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int p = k * BQB_CHUNK + 4 * lane + j;
            const bool ok = p < n;
            const float *q = pts + (size_t)(ok ? p : 0) * 3;
            c.x[k][j] = ok ? q[0] : far; c.y[k][j] = ok ? q[1] : far; c.z[k][j] = ok ? q[2] : far;
        }
}

// One centre: fills lst[0 .. nsample) (wave-private LDS) and returns the number of hits found before the list was full
// (0: empty ball, lst untouched).
template <int NCH>
__device__ __forceinline__ int bqb_centre(const BqbCloud<NCH> &c, float cx, float cy, float cz, float radius2, int nsample, int lane,
                                          int *lst)
{
    int cnt = 0;                                               // wave-uniform
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        if (cnt >= nsample) break;
        bool h[4];
        unsigned long long B[4];
        int lower = 0, tot = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float dx = cx - c.x[k][j];
            const float dy = cy - c.y[k][j];
            const float dz = cz - c.z[k][j];
            const float xx = dx * dx;
            const float yy = dy * dy;
            const float zz = dz * dz;
            const float sxy = xx + yy;
            const float d2 = sxy + zz;
            h[j] = d2 < radius2;
            B[j] = __ballot(h[j]);
            lower += (int)__builtin_amdgcn_mbcnt_hi((unsigned)(B[j] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)B[j], 0u));
            tot += __popcll(B[j]);
        }
        if (tot) {                                             // wave-uniform
            int pos = cnt + lower;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (h[j]) { if (pos < nsample) lst[pos] = k * BQB_CHUNK + 4 * lane + j; ++pos; }
            cnt += tot;
        }
    }
    __builtin_amdgcn_wave_barrier();
    if (cnt > 0 && cnt < nsample) {
        const int first = lst[0];
        for (int sl = cnt + lane; sl < nsample; sl += CMF_WAVE) lst[sl] = first;
        __builtin_amdgcn_wave_barrier();
    }
    return cnt;
}

// idx-only form (cmf_ball_query): G centres per wave; empty balls leave idx untouched (ball_query_gpu.cu:29-41 with the
// caller's pre-zeroed idx, lib/pointnet2_utils.py:246)
template <int NCH>
__global__ __launch_bounds__(BQB_WAVES *CMF_WAVE) void ball_query_ballot_kernel(
    int n, int m, float radius2, int nsample, int G, int zero_empty, const float *__restrict__ new_xyz, const float *__restrict__ xyz,
    int *__restrict__ idx)
{
    extern __shared__ int bqb_lds[];                          // [BQB_WAVES][G * nsample]
    const int bs = blockIdx.y, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int pt0 = (blockIdx.x * BQB_WAVES + w) * G;
    if (pt0 >= m) return;                                     // wave-uniform; no workgroup barrier anywhere
    BqbCloud<NCH> cloud;
    bqb_load_cloud<NCH>(xyz + (size_t)bs * n * 3, n, lane, cloud);
    int *lst = bqb_lds + w * G * nsample;
    const int gv = min(G, m - pt0);
    unsigned empty = 0u;
    for (int g = 0; g < gv; ++g) {
        const float *cc = new_xyz + ((size_t)bs * m + pt0 + g) * 3;
        if (bqb_centre<NCH>(cloud, cc[0], cc[1], cc[2], radius2, nsample, lane, lst + g * nsample) == 0) empty |= 1u << g;
    }
    __builtin_amdgcn_wave_barrier();
    int *out = idx + ((size_t)bs * m + pt0) * nsample;
    for (int e = lane; e < gv * nsample; e += CMF_WAVE) {
        if (!((empty >> (e / nsample)) & 1u)) out[e] = lst[e];
        else if (zero_empty) out[e] = 0;                        // idx not pre-zeroed by the caller (cmf_query_and_group)
    }
}

// The ball queries of ONE MultiScaleEncoder call (radarflow_util.py:111-118: four scales (r, nsample) over the same centres and
// cloud; models/cmflow.py:21-22: (2,4) (4,8) (8,16) (16,32)) in ONE launch: a wave loads the cloud and its G centres once and runs
// the hit-list construction per scale -- the same bqb_centre as the single-scale kernel, so every list is bit-identical to a
// cmf_ball_query call of its own.  blockIdx.z selects one of up to two (centres, cloud) pairs (the two clouds of the first encoder).
struct BqMultiArgs {
    int n, m, nq, G, zero_empty;
    float r2[4];
    int ns[4], lst_off[5];                // per scale: nsample, offset of its lists inside a wave's LDS block (in units of G ints)
    const float *new_xyz[2], *xyz[2];
    int *idx[2][4];
};

template <int NCH>
__global__ __launch_bounds__(BQB_WAVES *CMF_WAVE) void ball_query_multi_ballot_kernel(const BqMultiArgs a)
{
    extern __shared__ int bqb_lds[];                          // [BQB_WAVES][G * sum(nsample)]
    const int bs = blockIdx.y, cl = blockIdx.z, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int pt0 = (blockIdx.x * BQB_WAVES + w) * a.G;
    if (pt0 >= a.m) return;                                   // wave-uniform; no workgroup barrier anywhere
    BqbCloud<NCH> cloud;
    bqb_load_cloud<NCH>(a.xyz[cl] + (size_t)bs * a.n * 3, a.n, lane, cloud);
    int *lst = bqb_lds + w * a.G * a.lst_off[a.nq];
    const int gv = min(a.G, a.m - pt0);
    unsigned empty[4] = {0u, 0u, 0u, 0u};
    for (int g = 0; g < gv; ++g) {
        const float *cc = a.new_xyz[cl] + ((size_t)bs * a.m + pt0 + g) * 3;
        const float cx = cc[0], cy = cc[1], cz = cc[2];
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (q < a.nq && bqb_centre<NCH>(cloud, cx, cy, cz, a.r2[q], a.ns[q], lane, lst + a.G * a.lst_off[q] + g * a.ns[q]) == 0) empty[q] |= 1u << g;
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (q >= a.nq) break;
        const int ns = a.ns[q];
        int *out = a.idx[cl][q] + ((size_t)bs * a.m + pt0) * ns;
        const int *src = lst + a.G * a.lst_off[q];
        for (int e = lane; e < gv * ns; e += CMF_WAVE) {
            if (!((empty[q] >> (e / ns)) & 1u)) out[e] = src[e];
            else if (a.zero_empty) out[e] = 0;
        }
    }
}

// QueryAndGroup.forward (lib/pointnet2_utils.py:269-292) in ONE launch: ball query + grouped xyz relative to the centre
// + grouped features, written in the reference's (B, 3 + C, M, nsample) layout.  The wave that found the G lists
// gathers them: entry e of its G x nsample block is contiguous in every channel plane, so each channel is one (or a
// few) coalesced 256-byte stores per wave; the feature rows (n floats per channel) are read through the caches.
// An empty ball groups point 0 (the reference's pre-zeroed idx).
template <int NCH>
__global__ __launch_bounds__(BQB_WAVES *CMF_WAVE) void query_and_group_kernel(
    int n, int m, float radius2, int nsample, int G, int c, int use_xyz, const float *__restrict__ new_xyz,
    const float *__restrict__ xyz, const float *__restrict__ features, int *__restrict__ idx, float *__restrict__ out)
{
    extern __shared__ int bqb_lds[];
    const int bs = blockIdx.y, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int pt0 = (blockIdx.x * BQB_WAVES + w) * G;
    if (pt0 >= m) return;
    const float *pts = xyz + (size_t)bs * n * 3;
    BqbCloud<NCH> cloud;
    bqb_load_cloud<NCH>(pts, n, lane, cloud);
    int *lst = bqb_lds + w * G * nsample;
    const int gv = min(G, m - pt0);
    for (int g = 0; g < gv; ++g) {
        const float *cc = new_xyz + ((size_t)bs * m + pt0 + g) * 3;
        if (bqb_centre<NCH>(cloud, cc[0], cc[1], cc[2], radius2, nsample, lane, lst + g * nsample) == 0)
            for (int sl = lane; sl < nsample; sl += CMF_WAVE) lst[g * nsample + sl] = 0;
    }
    __builtin_amdgcn_wave_barrier();
    const int ctot = (use_xyz ? 3 : 0) + c;
    const size_t plane = (size_t)m * nsample;                  // one channel of one sample
    const size_t e0 = (size_t)pt0 * nsample;
    float *ob = out + (size_t)bs * ctot * plane + e0;
    const float *fb = features ? features + (size_t)bs * c * n : nullptr;
    for (int e = lane; e < gv * nsample; e += CMF_WAVE) {
        const int id = lst[e];
        if (idx) idx[(size_t)bs * plane + e0 + e] = id;
        float *o = ob + e;
        if (use_xyz) {
            const float *cc = new_xyz + ((size_t)bs * m + pt0 + e / nsample) * 3;
            const float *q = pts + (size_t)id * 3;
            o[0] = q[0] - cc[0]; o[plane] = q[1] - cc[1]; o[2 * plane] = q[2] - cc[2];
            o += 3 * plane;
        }
        int ch = 0;
        for (; ch + 4 <= c; ch += 4) {                          // 4 gathers in flight
            const float v0 = fb[(size_t)ch * n + id], v1 = fb[(size_t)(ch + 1) * n + id];
            const float v2 = fb[(size_t)(ch + 2) * n + id], v3 = fb[(size_t)(ch + 3) * n + id];
            o[(size_t)ch * plane] = v0; o[(size_t)(ch + 1) * plane] = v1; o[(size_t)(ch + 2) * plane] = v2; o[(size_t)(ch + 3) * plane] = v3;
        }
        for (; ch < c; ++ch) o[(size_t)ch * plane] = fb[(size_t)ch * n + id];
    }
}

// centres per wave: enough waves to fill the chip (>= ~4096), rows of at least 64 entries for the write-out
static int bqb_group(int b, int m, int nsample)
{
    long long g = (long long)b * m / 4096;
    const int g_min = nsample >= 64 ? 1 : (64 + nsample - 1) / nsample;
    if (g < g_min) g = g_min;
    if (g > 16) g = 16;
    if (g > m) g = m;
    return (int)(g < 1 ? 1 : g);
}

static int ball_query_grid(int b, int n, int m, float radius, int nsample, const float *new_xyz, const float *xyz, int *idx,
                           hipStream_t st, int zero_empty = 0)
{
    const size_t ncs = (size_t)BQG_CELLS + 1;
    const size_t off_cs = ((size_t)b * sizeof(BqGridHeader) + 255) / 256 * 256;
    const size_t off_sp = (off_cs + (size_t)b * ncs * sizeof(int) + 255) / 256 * 256;
    // The grid lives in the per-stream library scratch between the two launches.  The scales of an encoder call are enqueued
    // from several host threads, two of them onto the SAME stream: the pair of launches must be adjacent in the stream, or
    // another chain's build overwrites the grid before this query has read it.
    // (the lease keeps the stream's slot locked until both are enqueued).
    const CmfScratchLease lease = cmf_stream_scratch(st, 1, off_sp + (size_t)b * n * sizeof(float4));
    char *scratch = (char *)lease.ptr;
    if (!scratch) return (int)hipErrorOutOfMemory;
    BqGridHeader *hdr = (BqGridHeader *)scratch;
    int *cs = (int *)(scratch + off_cs);
    float4 *sp = (float4 *)(scratch + off_sp);
    hipLaunchKernelGGL(bq_grid_build_kernel, dim3(b), dim3(BQG_THREADS), 0, st, n, radius, xyz, hdr, cs, sp);
    hipLaunchKernelGGL(bq_grid_query_kernel, dim3(cmf_divup(m, 4), b), dim3(256), 0, st, n, m, radius * radius, nsample, new_xyz,
                       hdr, cs, sp, idx, zero_empty);
    return cmf_launch_status();
}
// large clouds: cell grid + index-ordered read-back (CMF_BALL_QUERY_GRID=0 keeps the scan: diagnostics).  A radius that is not a
// positive finite number has no grid; the scan handles it like the reference.
static bool ball_query_grid_takes(int n, float radius)
{
    static const bool use_grid = !(getenv("CMF_BALL_QUERY_GRID") && getenv("CMF_BALL_QUERY_GRID")[0] == '0');
    return use_grid && n >= 4096 && n <= BQG_MAX_N && radius > 0.f && radius < 3.0e38f;
}

static int ball_query_ballot(int b, int n, int m, float radius, int nsample, int zero_empty, const float *new_xyz, const float *xyz,
                             int *idx, hipStream_t st)
{
    const int G = bqb_group(b, m, nsample);
    const dim3 grid(cmf_divup(m, G * BQB_WAVES), b), block(BQB_WAVES * CMF_WAVE);
    const size_t lds = (size_t)BQB_WAVES * G * nsample * sizeof(int);
    const float r2 = radius * radius;
    if (n <= 256) hipLaunchKernelGGL(ball_query_ballot_kernel<1>, grid, block, lds, st, n, m, r2, nsample, G, zero_empty, new_xyz, xyz, idx);
    else if (n <= 512) hipLaunchKernelGGL(ball_query_ballot_kernel<2>, grid, block, lds, st, n, m, r2, nsample, G, zero_empty, new_xyz, xyz, idx);
    else hipLaunchKernelGGL(ball_query_ballot_kernel<4>, grid, block, lds, st, n, m, r2, nsample, G, zero_empty, new_xyz, xyz, idx);
    return cmf_launch_status();
}

// cmf_ball_query with EVERY entry of idx defined: an empty ball's row reads 0 (the reference pre-zeroes idx,
// lib/pointnet2_utils.py:246).  Small clouds: the ballot kernel writes the zeros itself (no memset launch in front of it).
int cmf_ball_query_defined(int b, int n, int m, float radius, int nsample, const float *new_xyz, const float *xyz, int *idx, void *stream)
{
    CMF_CHECK_ARG(b >= 0 && n >= 0 && m >= 0 && nsample > 0);
    if (b == 0 || m == 0) return 0;
    CMF_CHECK_ARG(new_xyz && xyz && idx);
    static const bool use_ballot = !(getenv("CMF_BALL_QUERY_BALLOT") && getenv("CMF_BALL_QUERY_BALLOT")[0] == '0');
    if (use_ballot && n > 0 && n <= BQB_MAX_N && nsample <= 256)
        return ball_query_ballot(b, n, m, radius, nsample, 1, new_xyz, xyz, idx, (hipStream_t)stream);
    if (ball_query_grid_takes(n, radius)) return ball_query_grid(b, n, m, radius, nsample, new_xyz, xyz, idx, (hipStream_t)stream, 1);
    if (hipMemsetAsync(idx, 0, (size_t)b * m * nsample * sizeof(int), (hipStream_t)stream) != hipSuccess) return (int)hipGetLastError();
    return cmf_ball_query(b, n, m, radius, nsample, new_xyz, xyz, idx, stream);
}

extern "C" int cmf_ball_query(int b, int n, int m, float radius, int nsample,
                              const float *new_xyz, const float *xyz, int *idx, void *stream)
{
    CMF_CHECK_ARG(b >= 0 && n >= 0 && m >= 0 && nsample > 0);
    if (b == 0 || m == 0 || n == 0) return 0;
    CMF_CHECK_ARG(new_xyz && xyz && idx);
    if (ball_query_grid_takes(n, radius)) return ball_query_grid(b, n, m, radius, nsample, new_xyz, xyz, idx, (hipStream_t)stream);
    // small clouds: the ballot kernel (CMF_BALL_QUERY_BALLOT=0 keeps the scan kernels: diagnostics)
    static const bool use_ballot = !(getenv("CMF_BALL_QUERY_BALLOT") && getenv("CMF_BALL_QUERY_BALLOT")[0] == '0');
    if (use_ballot && n <= BQB_MAX_N && nsample <= 256)
        return ball_query_ballot(b, n, m, radius, nsample, 0, new_xyz, xyz, idx, (hipStream_t)stream);
    dim3 grid(cmf_divup(m, CMF_WAVE), b);
    if (nsample <= BQ_MAX_NS_LDS && n <= 65535 && n >= 64) {
        const int seg = (cmf_divup(n, BQM_NW) + 3) / 4 * 4;
        const long long wgs = (long long)grid.x * grid.y;
        if (wgs > 3 * 256) {                                     // enough workgroups for LDS capacity to limit occupancy
            const size_t bytes = (size_t)wgs * BQM_NW * BQ_MAX_NS_LDS * CMF_WAVE * sizeof(unsigned short);
            const CmfScratchLease lease = cmf_stream_scratch((hipStream_t)stream, 1, bytes);   // per-stream library scratch
            unsigned short *scratch = (unsigned short *)lease.ptr;
            if (!scratch) return (int)hipErrorOutOfMemory;
            hipLaunchKernelGGL(ball_query_multi_kernel<true>, grid, dim3(BQM_NW * CMF_WAVE), 0, (hipStream_t)stream,
                               n, m, radius * radius, nsample, seg, new_xyz, xyz, idx, scratch);
            return cmf_launch_status();
        }
        hipLaunchKernelGGL(ball_query_multi_kernel<false>, grid, dim3(BQM_NW * CMF_WAVE), 0, (hipStream_t)stream,
                           n, m, radius * radius, nsample, seg, new_xyz, xyz, idx, (unsigned short *)nullptr);
        return cmf_launch_status();
    }
    if (nsample <= BQ_MAX_NS_LDS && n <= 1024)      // small clouds: latency bound, coalesced write-out pays
        hipLaunchKernelGGL(ball_query_kernel<true>, grid, dim3(CMF_WAVE), 0, (hipStream_t)stream,
                           n, m, radius * radius, nsample, new_xyz, xyz, idx);
    else
        hipLaunchKernelGGL(ball_query_kernel<false>, grid, dim3(CMF_WAVE), 0, (hipStream_t)stream,
                           n, m, radius * radius, nsample, new_xyz, xyz, idx);
    return cmf_launch_status();
}

// whether cmf_ball_query_multi takes this set of scales (cloud size, list lengths, LDS of one wave's lists at G = 1)
bool cmf_ball_query_multi_takes(int n, int nq, const int *nsamples)
{
    if (n <= 0 || n > BQB_MAX_N || nq < 1 || nq > 4) return false;
    size_t tot = 0;
    for (int q = 0; q < nq; ++q) {
        if (nsamples[q] <= 0 || nsamples[q] > 256) return false;
        tot += (size_t)nsamples[q];
    }
    return (size_t)BQB_WAVES * tot * sizeof(int) <= BQB_LDS_LIMIT;
}

// nq <= 4 ball queries (radii[q], nsamples[q]) -> idx[c][q] (B, M, nsamples[q]) over the same centres and cloud, for nclouds <= 2
// (centres, cloud) pairs of equal geometry, in ONE launch; every list equals cmf_ball_query's for that scale.  zero_empty != 0: the
// rows of empty balls are written as zeros (a caller that does not pre-zero idx).  Clouds of up to 1024 points (the ballot kernel).
extern "C" int cmf_ball_query_multi(int b, int n, int m, int nq, const float *radii, const int *nsamples, int nclouds,
                                    const float *const *new_xyz, const float *const *xyz, int *const *idx, int zero_empty, void *stream)
{
    CMF_CHECK_ARG(b >= 0 && n > 0 && n <= BQB_MAX_N && m >= 0 && nq >= 1 && nq <= 4 && nclouds >= 1 && nclouds <= 2 && radii && nsamples &&
                  new_xyz && xyz && idx);
    if (b == 0 || m == 0) return 0;
    BqMultiArgs a{};
    a.n = n; a.m = m; a.nq = nq; a.zero_empty = zero_empty;
    int nsmax = 0, tot = 0;
    for (int q = 0; q < nq; ++q) {
        CMF_CHECK_ARG(nsamples[q] > 0 && nsamples[q] <= 256);
        a.r2[q] = radii[q] * radii[q]; a.ns[q] = nsamples[q]; a.lst_off[q] = tot;
        tot += nsamples[q]; nsmax = std::max(nsmax, nsamples[q]);
    }
    for (int q = nq; q <= 4; ++q) a.lst_off[q] = tot;
    for (int c = 0; c < nclouds; ++c) {
        CMF_CHECK_ARG(new_xyz[c] && xyz[c]);
        a.new_xyz[c] = new_xyz[c]; a.xyz[c] = xyz[c];
        for (int q = 0; q < nq; ++q) { CMF_CHECK_ARG(idx[c * nq + q]); a.idx[c][q] = idx[c * nq + q]; }
    }
    a.G = bqb_group(b * nclouds, m, nsmax);
    // a wave's lists of all nq scales for its G centres live in LDS: G shrinks until the workgroup fits the 64 KB a launch may ask for
    // without an attribute (4 scales of 256 entries fit at G = 4); callers with longer lists use the single-scale queries
    while (a.G > 1 && (size_t)BQB_WAVES * a.G * tot * sizeof(int) > BQB_LDS_LIMIT) --a.G;
    CMF_CHECK_ARG((size_t)BQB_WAVES * a.G * tot * sizeof(int) <= BQB_LDS_LIMIT);
    const dim3 grid(cmf_divup(m, a.G * BQB_WAVES), b, nclouds), block(BQB_WAVES * CMF_WAVE);
    const size_t lds = (size_t)BQB_WAVES * a.G * tot * sizeof(int);
    hipStream_t st = (hipStream_t)stream;
    if (n <= 256) hipLaunchKernelGGL(ball_query_multi_ballot_kernel<1>, grid, block, lds, st, a);
    else if (n <= 512) hipLaunchKernelGGL(ball_query_multi_ballot_kernel<2>, grid, block, lds, st, a);
    else hipLaunchKernelGGL(ball_query_multi_ballot_kernel<4>, grid, block, lds, st, a);
    return cmf_launch_status();
}

// internal (group_points.hip): the LDS-staged gather writing the feature planes AND the relative-coordinate planes of the
// fused op's (B, 3 + C, M, nsample) output in one launch
int cmf_group_points_xyz(int b, int c, int n, int npoints, int nsample, const float *points, const int *idx, float *out,
                         long long out_bstride, const float *xyz, const float *new_xyz, void *stream);

extern "C" int cmf_query_and_group(int b, int n, int m, float radius, int nsample, int c, int use_xyz,
                                   const float *new_xyz, const float *xyz, const float *features, int *idx, float *out, void *stream)
{
    CMF_CHECK_ARG(b >= 0 && n >= 0 && m >= 0 && nsample > 0 && c >= 0 && (use_xyz || c > 0));
    if (b == 0 || m == 0) return 0;
    CMF_CHECK_ARG(n > 0 && new_xyz && xyz && out && (c == 0 || features));
    hipStream_t st = (hipStream_t)stream;
    const int ctot = (use_xyz ? 3 : 0) + c;
    const long long plane = (long long)m * nsample;
    // ONE launch when the cloud fits the ballot kernel's registers and the gather is a few planes (the waves that ran the
    // query write them: 8.6 us against 14.5 us for three launches at (64,256,32) with C = 3); wider features go through the
    // LDS-staged gather, which also writes the relative-coordinate planes: two launches (measured: the single launch with
    // cache-served feature gathers is slower from C ~ 32 on -- 36.6 against 33.0 us at C = 64)
    if (n <= BQB_MAX_N && nsample <= 256 && ctot <= 24) {
        const int G = bqb_group(b, m, nsample);
        const dim3 grid(cmf_divup(m, G * BQB_WAVES), b), block(BQB_WAVES * CMF_WAVE);
        const size_t lds = (size_t)BQB_WAVES * G * nsample * sizeof(int);
        const float r2 = radius * radius;
        if (n <= 256) hipLaunchKernelGGL(query_and_group_kernel<1>, grid, block, lds, st, n, m, r2, nsample, G, c, use_xyz, new_xyz, xyz, features, idx, out);
        else if (n <= 512) hipLaunchKernelGGL(query_and_group_kernel<2>, grid, block, lds, st, n, m, r2, nsample, G, c, use_xyz, new_xyz, xyz, features, idx, out);
        else hipLaunchKernelGGL(query_and_group_kernel<4>, grid, block, lds, st, n, m, r2, nsample, G, c, use_xyz, new_xyz, xyz, features, idx, out);
        return cmf_launch_status();
    }
    // query (+ zero fill for empty balls: the reference pre-zeroes idx), then one gather launch for all planes
    int *ix = idx;
    CmfScratchLease lease;                                   // held until the gather that reads ix is enqueued
    if (!ix) {
        lease = cmf_stream_scratch(st, 2, (size_t)b * plane * sizeof(int));
        ix = (int *)lease.ptr;
        if (!ix) return (int)hipErrorOutOfMemory;
    }
    int err;
    if (n <= BQB_MAX_N && nsample <= 256) err = ball_query_ballot(b, n, m, radius, nsample, 1, new_xyz, xyz, ix, st);
    else if (ball_query_grid_takes(n, radius)) err = ball_query_grid(b, n, m, radius, nsample, new_xyz, xyz, ix, st, 1);     // (the query writes empty balls' zeros: no memset launch)
    else {
        if (hipMemsetAsync(ix, 0, (size_t)b * plane * sizeof(int), st) != hipSuccess) return (int)hipGetLastError();
        err = cmf_ball_query(b, n, m, radius, nsample, new_xyz, xyz, ix, stream);
    }
    if (err) return err;
    return cmf_group_points_xyz(b, c, n, m, nsample, features, ix, out, (long long)ctot * plane, use_xyz ? xyz : nullptr,
                                use_xyz ? new_xyz : nullptr, stream);
}

// ---------------------------------------------------------------------------------------------
// kNN.  One lane per query; database points + their squared norms staged in LDS (16 B/point);
// the K best (distance, index) pairs live in registers as a sorted list.  Strict '<' in both
// the admission test and the bubble-up keeps the earliest index ahead on ties (= the oracle).
// ---------------------------------------------------------------------------------------------
constexpr int KNN_TILE = 1024;      // 16 KiB

__device__ __forceinline__ float sqnorm3(float x, float y, float z)
{
    const float xx = x * x;
    const float yy = y * y;
    const float zz = z * z;
    const float s = xx + yy;
    return s + zz;
}

// PLAIN = false: the torch-level kNN of the model (matmul-form distance, radarflow_util.py:8-30).
// PLAIN = true : the extension's knn / three_nn kernels (lib/src/interpolate_gpu.cu:9-57,81-124), which use the
//                direct form d = (ux-x)^2 + (uy-y)^2 + (uz-z)^2 (non-contracted here) and keep the first-seen
//                point on ties -- identical bookkeeping, only the distance differs.
template <int K, bool PLAIN>
__global__ __launch_bounds__(CMF_WAVE) void knn_kernel(
    int n, int s, int nsample, const float *__restrict__ xyz, const float *__restrict__ new_xyz,
    int *__restrict__ idx, float *__restrict__ dist)
{
    __shared__ float4 tile[KNN_TILE];
    const int bs = blockIdx.y;
    const int q = blockIdx.x * CMF_WAVE + threadIdx.x;
    const bool live = q < s;
    const float *pts = xyz + (size_t)bs * n * 3;
    float qx = 0.f, qy = 0.f, qz = 0.f;
    if (live) {
        const float *c = new_xyz + ((size_t)bs * s + q) * 3;
        qx = c[0]; qy = c[1]; qz = c[2];
    }
    const float ss = sqnorm3(qx, qy, qz);
    float bd[K];
    int bi[K];
#pragma unroll
    for (int t = 0; t < K; ++t) { bd[t] = __builtin_inff(); bi[t] = 0; }

    for (int base = 0; base < n; base += KNN_TILE) {
        const int len = min(KNN_TILE, n - base);
        __syncthreads();
        for (int i = threadIdx.x; i < len; i += CMF_WAVE) {
            const float x = pts[(size_t)(base + i) * 3 + 0];
            const float y = pts[(size_t)(base + i) * 3 + 1];
            const float z = pts[(size_t)(base + i) * 3 + 2];
            tile[i] = make_float4(x, y, z, sqnorm3(x, y, z));
        }
        __syncthreads();
        for (int k = 0; k < len; ++k) {
            const float4 p = tile[k];
            float v;
            if (PLAIN) {
                const float dx = qx - p.x;
                const float dy = qy - p.y;
                const float dz = qz - p.z;
                const float xx = dx * dx;
                const float yy = dy * dy;
                const float zz = dz * dz;
                const float sxy = xx + yy;
                v = sxy + zz;
            } else {
                const float p0 = qx * p.x;
                const float p01 = __builtin_fmaf(qy, p.y, p0);
                const float dot = __builtin_fmaf(qz, p.z, p01);
                const float t = -2.0f * dot;
                const float u = t + ss;
                v = u + p.w;
                v = (v > 0.0f) ? v : 0.0f;
            }
            if (v < bd[K - 1]) {
                bd[K - 1] = v;
                bi[K - 1] = base + k;
#pragma unroll
                for (int j = K - 1; j > 0; --j) {
                    if (bd[j] < bd[j - 1]) {
                        const float td = bd[j]; bd[j] = bd[j - 1]; bd[j - 1] = td;
                        const int ti = bi[j]; bi[j] = bi[j - 1]; bi[j - 1] = ti;
                    }
                }
            }
        }
    }
    if (live) {
        int *o = idx + ((size_t)bs * s + q) * nsample;
#pragma unroll
        for (int t = 0; t < K; ++t)
            if (t < nsample) o[t] = (bd[t] == __builtin_inff()) ? 0 : bi[t];
        if (dist) {
            float *d = dist + ((size_t)bs * s + q) * nsample;
#pragma unroll
            for (int t = 0; t < K; ++t)
                if (t < nsample) d[t] = (bd[t] == __builtin_inff()) ? 0.0f : bd[t];
        }
    }
}

extern "C" int cmf_knn(int b, int n, int s, int nsample, const float *xyz, const float *new_xyz,
                       int *idx, float *dist, void *stream)
{
    CMF_CHECK_ARG(b >= 0 && n >= 0 && s >= 0 && nsample > 0 && nsample <= 32);
    if (b == 0 || s == 0) return 0;
    CMF_CHECK_ARG(xyz && new_xyz && idx);
    dim3 grid(cmf_divup(s, CMF_WAVE), b), block(CMF_WAVE);
    hipStream_t st = (hipStream_t)stream;
    // the list length is a compile-time constant so it stays in registers; the list is sorted,
    // so running a longer list and emitting its first nsample entries is exact.
    if (nsample <= 1)       hipLaunchKernelGGL((knn_kernel<1, false>),  grid, block, 0, st, n, s, nsample, xyz, new_xyz, idx, dist);
    else if (nsample <= 4)  hipLaunchKernelGGL((knn_kernel<4, false>),  grid, block, 0, st, n, s, nsample, xyz, new_xyz, idx, dist);
    else if (nsample <= 8)  hipLaunchKernelGGL((knn_kernel<8, false>),  grid, block, 0, st, n, s, nsample, xyz, new_xyz, idx, dist);
    else if (nsample <= 16) hipLaunchKernelGGL((knn_kernel<16, false>), grid, block, 0, st, n, s, nsample, xyz, new_xyz, idx, dist);
    else                    hipLaunchKernelGGL((knn_kernel<32, false>), grid, block, 0, st, n, s, nsample, xyz, new_xyz, idx, dist);
    return cmf_launch_status();
}

// ---- rest of the pointnet2_cuda neighbour surface (not called by CMFlow) -------------------

// knn_wrapper (lib/src/interpolate_gpu.cu:9-57): unknown (b,n,3) queries, known (b,m,3) -> dist2, idx (b,n,k).
extern "C" int cmf_knn_points(int b, int n, int m, int k, const float *unknown, const float *known,
                              float *dist2, int *idx, void *stream)
{
    CMF_CHECK_ARG(b >= 0 && n >= 0 && m >= 0 && k > 0 && k <= 64);      // the reference allows k <= 200
    if (b == 0 || n == 0) return 0;
    CMF_CHECK_ARG(unknown && known && dist2 && idx);
    dim3 grid(cmf_divup(n, CMF_WAVE), b), block(CMF_WAVE);
    hipStream_t st = (hipStream_t)stream;
    if (k <= 4)       hipLaunchKernelGGL((knn_kernel<4, true>),  grid, block, 0, st, m, n, k, known, unknown, idx, dist2);
    else if (k <= 8)  hipLaunchKernelGGL((knn_kernel<8, true>),  grid, block, 0, st, m, n, k, known, unknown, idx, dist2);
    else if (k <= 16) hipLaunchKernelGGL((knn_kernel<16, true>), grid, block, 0, st, m, n, k, known, unknown, idx, dist2);
    else if (k <= 32) hipLaunchKernelGGL((knn_kernel<32, true>), grid, block, 0, st, m, n, k, known, unknown, idx, dist2);
    else              hipLaunchKernelGGL((knn_kernel<64, true>), grid, block, 0, st, m, n, k, known, unknown, idx, dist2);
    return cmf_launch_status();
}

// three_nn_wrapper (lib/src/interpolate_gpu.cu:81-124): the 3 nearest known points, dist^2 returned.
extern "C" int cmf_three_nn(int b, int n, int m, const float *unknown, const float *known,
                            float *dist2, int *idx, void *stream)
{
    return cmf_knn_points(b, n, m, 3, unknown, known, dist2, idx, stream);
}

// furthest_point_sampling_wrapper (lib/src/sampling_gpu.cu:93-209): one workgroup per sample, m sequential
// rounds; temp (b,n) holds the running min distance (caller fills it with 1e10, lib/pointnet2_utils.py:26).
// Ties in the arg-max go to the LOWEST index (the reference's tree reduction prefers the lower thread).
constexpr int FPS_THREADS = 256;
__global__ __launch_bounds__(FPS_THREADS) void fps_kernel(int n, int m, const float *__restrict__ dataset,
                                                          float *__restrict__ temp, int *__restrict__ idxs)
{
    __shared__ float sv[FPS_THREADS / CMF_WAVE];
    __shared__ int si[FPS_THREADS / CMF_WAVE];
    __shared__ int s_old;
    const int bs = blockIdx.x, tid = threadIdx.x;
    const float *d = dataset + (size_t)bs * n * 3;
    float *t = temp + (size_t)bs * n;
    int *o = idxs + (size_t)bs * m;
    if (m <= 0) return;
    int old = 0;
    if (tid == 0) o[0] = 0;
    for (int j = 1; j < m; ++j) {
        const float x1 = d[old * 3 + 0], y1 = d[old * 3 + 1], z1 = d[old * 3 + 2];
        float best = -1.f;
        int besti = 0;
        for (int k = tid; k < n; k += FPS_THREADS) {
            const float dx = d[k * 3 + 0] - x1;
            const float dy = d[k * 3 + 1] - y1;
            const float dz = d[k * 3 + 2] - z1;
            const float xx = dx * dx;
            const float yy = dy * dy;
            const float zz = dz * dz;
            const float sxy = xx + yy;
            const float dd = sxy + zz;
            const float d2 = fminf(dd, t[k]);
            t[k] = d2;
            if (d2 > best) { best = d2; besti = k; }
        }
        // arg-max with lowest-index tie-break: wave shuffle tree, then across the 4 waves
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float ov = __shfl_xor(best, off, 64);
            const int oi = __shfl_xor(besti, off, 64);
            if (ov > best || (ov == best && oi < besti)) { best = ov; besti = oi; }
        }
        if ((tid & 63) == 0) { sv[tid >> 6] = best; si[tid >> 6] = besti; }
        __syncthreads();
        if (tid == 0) {
            float bv = sv[0]; int bi = si[0];
            for (int w = 1; w < FPS_THREADS / CMF_WAVE; ++w)
                if (sv[w] > bv || (sv[w] == bv && si[w] < bi)) { bv = sv[w]; bi = si[w]; }
            s_old = bi;
            o[j] = bi;
        }
        __syncthreads();
        old = s_old;
        __syncthreads();
    }
}

extern "C" int cmf_furthest_point_sampling(int b, int n, int m, const float *dataset, float *temp, int *idxs, void *stream)
{
    CMF_CHECK_ARG(b >= 0 && n > 0 && m >= 0);
    if (b == 0 || m == 0) return 0;
    CMF_CHECK_ARG(dataset && temp && idxs);
    hipLaunchKernelGGL(fps_kernel, dim3(b), dim3(FPS_THREADS), 0, (hipStream_t)stream, n, m, dataset, temp, idxs);
    return cmf_launch_status();
}
