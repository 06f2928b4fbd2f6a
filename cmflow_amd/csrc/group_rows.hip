// Point-major grouping: the layout the fused CMFlow path computes in.
//
// The reference groups (B,C,N) tensors channel by channel (lib/src/group_points_gpu.cu:47-66):
// an uncoalesced 4-byte gather per output float.  In point-major layout (B,N,C) a neighbour's
// feature vector is one contiguous row, so grouping is a ROW gather with fully coalesced 16-byte
// loads and stores:      out[b, e, :] = feat[b, idx[b, e], :]      e = p*nsample + s.
//
// Backward is a deterministic gather over an inverse index (CSR per sample) instead of the
// reference's 538 M fp32 atomics (group_points_gpu.cu:8-25): for every source point j the list
// of entries e with idx[e] == j, in ascending e, so the sum order is fixed (bit-reproducible)
// and equals the CPU oracle's scan order.
#include <algorithm>
#include "cmf_common.h"
#include "../../include/cmflow_hip.h"

int cmf_build_inverse_rows(int b, int n, int P, int S, const int *idx, int *offsets, int *inv, void *stream, int entries);

constexpr int GR_THREADS = 256;

// feat: (b, n, ldf) rows of c floats; out: (b, entries, c) dense.
template <int VEC>
__global__ __launch_bounds__(GR_THREADS) void group_rows_kernel(
    int n, int c, int ldf, int entries, long long total_vec,
    const float *__restrict__ feat, const int *__restrict__ idx, float *__restrict__ out)
{
    const int cv = c / VEC;
    for (long long i = (long long)blockIdx.x * GR_THREADS + threadIdx.x; i < total_vec;
         i += (long long)gridDim.x * GR_THREADS) {
        const long long row = i / cv;               // b*entries + e
        const int col = (int)(i - row * cv) * VEC;
        const int bs = (int)(row / entries);
        const int j = idx[row];
        const float *src = feat + ((size_t)bs * n + j) * ldf + col;
        float *dst = out + (size_t)row * c + col;
        if (VEC == 4) *reinterpret_cast<float4 *>(dst) = *reinterpret_cast<const float4 *>(src);
        else *dst = *src;
    }
}

extern "C" int cmf_group_rows(int b, int n, int c, int ldf, int entries,
                              const float *feat, const int *idx, float *out, void *stream)
{
    CMF_CHECK_ARG(b >= 0 && n > 0 && c > 0 && ldf >= c && entries >= 0);
    if (b == 0 || entries == 0) return 0;
    CMF_CHECK_ARG(feat && idx && out);
    const bool v4 = (c % 4 == 0) && (ldf % 4 == 0) && (((uintptr_t)feat | (uintptr_t)out) % 16 == 0);
    const long long total = (long long)b * entries * (v4 ? c / 4 : c);
    const int grid = (int)std::min<long long>((total + GR_THREADS - 1) / GR_THREADS, 256 * 32);
    if (v4) hipLaunchKernelGGL(group_rows_kernel<4>, dim3(grid), dim3(GR_THREADS), 0, (hipStream_t)stream,
                               n, c, ldf, entries, total, feat, idx, out);
    else hipLaunchKernelGGL(group_rows_kernel<1>, dim3(grid), dim3(GR_THREADS), 0, (hipStream_t)stream,
                            n, c, ldf, entries, total, feat, idx, out);
    return cmf_launch_status();
}

// ---- small layout helpers of the host side: each replaces two or three torch kernels (fill + copy [+ sub]) by one launch ----
// dst[r][c] = c < k ? src[r * ld_src + c] : 0   for c < ld_dst   (rows padded to 16-byte multiples for the GEMMs)
__global__ __launch_bounds__(256) void pad_rows_kernel(long long rows, int k, const float *__restrict__ src, long long ld_src,
                                                       float *__restrict__ dst, int ld_dst)
{
    const long long total = rows * ld_dst;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / ld_dst; const int c = (int)(i - r * ld_dst);
        dst[i] = c < k ? src[r * ld_src + c] : 0.f;
    }
}
extern "C" int cmf_pad_rows(long long rows, int k, const float *src, long long ld_src, float *dst, int ld_dst, void *stream)
{
    CMF_CHECK_ARG(rows >= 0 && k > 0 && ld_dst >= k && ld_src >= k);
    if (rows == 0) return 0;
    CMF_CHECK_ARG(src && dst);
    const int grid = (int)std::min<long long>((rows * ld_dst + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(pad_rows_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, rows, k, src, ld_src, dst, ld_dst);
    return cmf_launch_status();
}

// The model's inputs in point-major rows, all four in one launch: pc1, pc2 (b,3,n) -> x1, x2 (b,n,3);  ft1, ft2 (b,c,n) -> a1, a2 (b,n,cp)
// with zero columns c .. cp-1 (cmflow.py:59-64 keeps them channel-major; the fused path computes in rows)
__global__ __launch_bounds__(256) void inputs_point_major_kernel(int n, int c, int cp, const float *__restrict__ pc1, const float *__restrict__ pc2,
                                                                 const float *__restrict__ ft1, const float *__restrict__ ft2,
                                                                 float *__restrict__ x1, float *__restrict__ x2, float *__restrict__ a1, float *__restrict__ a2)
{
    const int bs = blockIdx.y, per = 3 + cp;
    const float *p = blockIdx.z ? pc2 : pc1, *f = blockIdx.z ? ft2 : ft1;
    float *x = blockIdx.z ? x2 : x1, *a = blockIdx.z ? a2 : a1;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n * per; i += gridDim.x * 256) {
        const int pt = i / per, ch = i - pt * per;
        if (ch < 3) x[((size_t)bs * n + pt) * 3 + ch] = p[((size_t)bs * 3 + ch) * n + pt];
        else a[((size_t)bs * n + pt) * cp + (ch - 3)] = (ch - 3) < c ? f[((size_t)bs * c + (ch - 3)) * n + pt] : 0.f;
    }
}
extern "C" int cmf_inputs_point_major(int b, int n, int c, int cp, const float *pc1, const float *pc2, const float *ft1, const float *ft2,
                                      float *x1, float *x2, float *a1, float *a2, void *stream)
{
    CMF_CHECK_ARG(b >= 0 && n > 0 && c > 0 && cp >= c);
    if (b == 0) return 0;
    CMF_CHECK_ARG(pc1 && pc2 && ft1 && ft2 && x1 && x2 && a1 && a2);
    hipLaunchKernelGGL(inputs_point_major_kernel, dim3(cmf_divup(n * (3 + cp), 256), b, 2), dim3(256), 0, (hipStream_t)stream,
                       n, c, cp, pc1, pc2, ft1, ft2, x1, x2, a1, a2);
    return cmf_launch_status();
}

// Relative coordinates of the neighbours as 4-float rows: out[b][p][s] = (xyz[b][idx[b][p][s]] - centre[b][p], 0)   (radarflow_util.py:207-208)
__global__ __launch_bounds__(256) void rel_xyz_kernel(int n, int m, int S, const float *__restrict__ xyz, const float *__restrict__ centre,
                                                      const int *__restrict__ idx, float4 *__restrict__ out)
{
    const int bs = blockIdx.y;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < m * S; e += gridDim.x * 256) {
        const int p = e / S;
        const float *q = xyz + ((size_t)bs * n + idx[(size_t)bs * m * S + e]) * 3, *cc = centre + ((size_t)bs * m + p) * 3;
        out[(size_t)bs * m * S + e] = make_float4(q[0] - cc[0], q[1] - cc[1], q[2] - cc[2], 0.f);
    }
}
extern "C" int cmf_rel_xyz(int b, int n, int m, int S, const float *xyz, const float *centre, const int *idx, float *out, void *stream)
{
    CMF_CHECK_ARG(b >= 0 && n > 0 && m > 0 && S > 0);
    if (b == 0) return 0;
    CMF_CHECK_ARG(xyz && centre && idx && out && ((uintptr_t)out % 16 == 0));
    hipLaunchKernelGGL(rel_xyz_kernel, dim3(cmf_divup(m * S, 256), b), dim3(256), 0, (hipStream_t)stream, n, m, S, xyz, centre, idx, (float4 *)out);
    return cmf_launch_status();
}

// ---------------------------------------------------------------------------------------------
// Inverse index: per sample, offsets[n+1] and the entry list sorted by (target point, entry).
// One workgroup per sample; thread j owns target point j and scans the sample's idx (staged in
// LDS, all lanes read the same word -> broadcast) twice: count, then fill.
// ---------------------------------------------------------------------------------------------
constexpr int INV_THREADS = 256;
constexpr int INV_TILE = 8192;          // idx entries per LDS tile (32 KiB)

// pass 1: offsets[b][0..n] = exclusive scan of the per-target counts (LDS integer histogram: the counts
// are order independent, so atomics are deterministic here)
__global__ __launch_bounds__(INV_THREADS) void inverse_count_kernel(
    int n, int entries, const int *__restrict__ idx, int *__restrict__ offsets)
{
    extern __shared__ int hist[];        // [n + 1]
    const int bs = blockIdx.x;
    const int *ix = idx + (size_t)bs * entries;
    for (int i = threadIdx.x; i <= n; i += INV_THREADS) hist[i] = 0;
    __syncthreads();
    for (int e = threadIdx.x; e < entries; e += INV_THREADS) atomicAdd(&hist[ix[e]], 1);
    __syncthreads();
    if (threadIdx.x == 0) {              // n is a few hundred to a few thousand: a serial scan is negligible
        int run = 0;
        for (int i = 0; i < n; ++i) { const int c = hist[i]; hist[i] = run; run += c; }
        hist[n] = run;
    }
    __syncthreads();
    int *off = offsets + (size_t)bs * (n + 1);
    for (int i = threadIdx.x; i <= n; i += INV_THREADS) off[i] = hist[i];
}

// pass 2: one lane per target point scans the sample's idx (LDS broadcast reads) and appends the entries
// that reference it in ascending order -> the list order (hence every later sum) is deterministic.
// grid (b, ceil(n/64)): 64 targets per wavefront so the scan spreads over the whole chip.
__global__ __launch_bounds__(CMF_WAVE) void inverse_fill_kernel(
    int n, int entries, const int *__restrict__ idx, const int *__restrict__ offsets, int *__restrict__ inv)
{
    __shared__ int tile[INV_TILE];
    const int bs = blockIdx.x;
    const int j = blockIdx.y * CMF_WAVE + threadIdx.x;
    const int *ix = idx + (size_t)bs * entries;
    int *lst = inv + (size_t)bs * entries;
    int pos = (j < n) ? offsets[(size_t)bs * (n + 1) + j] : 0;
    for (int e0 = 0; e0 < entries; e0 += INV_TILE) {
        const int len = min(INV_TILE, entries - e0);
        __syncthreads();
        for (int i = threadIdx.x; i < len; i += CMF_WAVE) tile[i] = ix[e0 + i];
        __syncthreads();
        if (j < n)
            for (int i = 0; i < len; ++i)
                if (tile[i] == j) lst[pos++] = e0 + i;
    }
}

// Fast path for the model's sizes (rows P <= 256, targets n with P*(n+1)*2 bytes of LDS): a P x n matrix
// of uint16 multiplicities cnt[p][j] in LDS gives every entry its rank without scanning:
//   A) thread p walks its row: rank_in_row = cnt[p][j]++            (one owner per row: no atomics)
//   B) thread j turns column j into an exclusive prefix over p; column totals -> offsets (scan over j)
//   C) thread p writes entry (p,s) to offsets[j] + prefix[p][j] + rank_in_row  => ascending entry order.
// O(P*S + P*n/threads) work per sample instead of O(n * P*S).
constexpr int INVM_THREADS = 256;

// inclusive prefix sum over the 256 threads of a workgroup; wsum: 4 ints of LDS
__device__ __forceinline__ int invm_block_scan(int v, int *wsum, int &block_total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int u = __shfl_up(v, d, 64);
        if (lane >= d) v += u;
    }
    __syncthreads();                                                // previous users of wsum are done
    if (lane == 63) wsum[wave] = v;
    __syncthreads();
    int add = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < INVM_THREADS / 64; ++w) {
        const int x = wsum[w];
        if (w < wave) add += x;
        tot += x;
    }
    block_total = tot;
    return v + add;
}

// Round 3: the three serial chains of the first version are gone -- (A) the per-entry ranks lived in a dynamically indexed
// local array (= scratch memory), now a byte array in LDS; (B) the column walk was a 256-step load -> add -> store chain the
// compiler could not overlap (the store may alias the next load), now 8 loads in flight per step; the column totals were
// scanned by ONE thread over 256 LDS words, now a block scan: 33 -> ~10 us per call at P = n = 256, S = 32 (14 calls per
// training step).
__device__ __forceinline__ void inverse_matrix_body(
    int n, int P, int S, const int *__restrict__ idx, int *__restrict__ offsets, int *__restrict__ inv)
{
    extern __shared__ unsigned short cnt[];             // [P][ld], ld = n | 1 (odd stride spreads the banks) | rk[P * S] bytes
    __shared__ int scan[INVM_THREADS / 64];
    const int ld = n | 1;
    const int bs = blockIdx.x, t = threadIdx.x;
    const int entries = P * S;
    unsigned char *rk = reinterpret_cast<unsigned char *>(cnt + (((size_t)P * ld + 7) / 8) * 8);
    const int *ix = idx + (size_t)bs * entries;
    int *off = offsets + (size_t)bs * (n + 1);
    int *lst = inv + (size_t)bs * entries;
    for (int i = t; i < (int)(((size_t)P * ld + 7) / 8); i += INVM_THREADS) reinterpret_cast<uint4 *>(cnt)[i] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    // A: within-row ranks (one owner per row: no atomics)
    if (t < P)
        for (int s = 0; s < S; ++s) {
            const int j = ix[t * S + s];
            const unsigned short c = cnt[t * ld + j];
            rk[t * S + s] = (unsigned char)c;
            cnt[t * ld + j] = (unsigned short)(c + 1);
        }
    __syncthreads();
    // B: thread j turns column j into an exclusive prefix over the rows; column totals -> offsets (block scan, running base)
    int base = 0;
    for (int j0 = 0; j0 < n; j0 += INVM_THREADS) {
        const int j = j0 + t;
        int run = 0;
        if (j < n) {
            int p0 = 0;
            for (; p0 + 8 <= P; p0 += 8) {
                unsigned short v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = cnt[(p0 + u) * ld + j];
#pragma unroll
                for (int u = 0; u < 8; ++u) { cnt[(p0 + u) * ld + j] = (unsigned short)run; run += v[u]; }
            }
            for (; p0 < P; ++p0) { const int c = cnt[p0 * ld + j]; cnt[p0 * ld + j] = (unsigned short)run; run += c; }
        }
        int tot;
        const int incl = invm_block_scan(run, scan, tot);
        if (j < n) off[j] = base + incl - run;
        base += tot;
    }
    if (t == 0) off[n] = base;
    __threadfence_block();
    __syncthreads();
    // C: scatter the entries to their slots
    if (t < P)
        for (int s = 0; s < S; ++s) {
            const int j = ix[t * S + s];
            lst[off[j] + cnt[t * ld + j] + rk[t * S + s]] = t * S + s;
        }
}

__global__ __launch_bounds__(INVM_THREADS) void inverse_matrix_kernel(
    int n, int P, int S, const int *__restrict__ idx, int *__restrict__ offsets, int *__restrict__ inv)
{
    inverse_matrix_body(n, P, S, idx, offsets, inv);
}

__global__ __launch_bounds__(INVM_THREADS) void inverse_matrix_batch_kernel(const CmfBatch<CmfInverseArgs> b)
{
    const CmfInverseArgs &p = b.a[blockIdx.y];
    inverse_matrix_body(p.n, p.P, p.S, p.idx, p.offsets, p.inv);
}

// the inverse indices of up to CMF_MAX_BATCH groupings (b samples each) in one launch: the matrix form only (P <= 256 centres, S <= 64)
int cmf_build_inverse_ps_batch(int n, int b, const CmfInverseArgs *a, hipStream_t st)
{
    CMF_CHECK_ARG(n >= 1 && n <= CMF_MAX_BATCH && a && b > 0);
    CmfBatch<CmfInverseArgs> bt;
    size_t mat = 0;
    for (int i = 0; i < n; ++i) {
        const CmfInverseArgs &q = a[i];
        const size_t m = (((size_t)q.P * (q.n | 1) + 7) / 8) * 16 + (size_t)q.P * q.S;
        CMF_CHECK_ARG(q.n > 0 && q.n < 40000 && q.P > 0 && q.P <= INVM_THREADS && q.S > 0 && q.S <= 64 && m <= 150 * 1024 && q.idx && q.offsets && q.inv);
        bt.a[i] = q;
        mat = std::max(mat, m);
    }
    static CmfPerDevice attr_set;
    int attr_dev;
    if (attr_set.need(attr_dev)) {
        (void)hipFuncSetAttribute((const void *)inverse_matrix_batch_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        attr_set.done(attr_dev);
    }
    hipLaunchKernelGGL(inverse_matrix_batch_kernel, dim3(b, n), dim3(INVM_THREADS), mat, st, bt);
    return cmf_launch_status();
}

extern "C" int cmf_build_inverse(int b, int n, int entries, const int *idx, int *offsets, int *inv, void *stream)
{
    CMF_CHECK_ARG(b >= 0 && n > 0 && entries >= 0 && n < 40000);
    if (b == 0) return 0;
    CMF_CHECK_ARG(idx && offsets && inv);
    return cmf_build_inverse_rows(b, n, 0, 0, idx, offsets, inv, stream, entries);
}

// rows/S known (idx is (b,P,S)): enables the matrix fast path
extern "C" int cmf_build_inverse_ps(int b, int n, int P, int S, const int *idx, int *offsets, int *inv, void *stream)
{
    CMF_CHECK_ARG(b >= 0 && n > 0 && P >= 0 && S >= 0 && n < 40000);
    if (b == 0) return 0;
    CMF_CHECK_ARG(idx && offsets && inv);
    return cmf_build_inverse_rows(b, n, P, S, idx, offsets, inv, stream, P * S);
}

int cmf_build_inverse_rows(int b, int n, int P, int S, const int *idx, int *offsets, int *inv, void *stream, int entries)
{
    const size_t mat = (((size_t)P * (n | 1) + 7) / 8) * 16 + (size_t)P * S;         // count matrix (16-byte granules) + rank bytes
    if (P > 0 && P <= INVM_THREADS && S <= 64 && mat <= 150 * 1024) {
        static CmfPerDevice attr_set;                   // the dynamic-LDS limit is per (function, device)
        int attr_dev;
        if (attr_set.need(attr_dev)) {
            (void)hipFuncSetAttribute((const void *)inverse_matrix_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
            attr_set.done(attr_dev);
        }
        hipLaunchKernelGGL(inverse_matrix_kernel, dim3(b), dim3(INVM_THREADS), mat, (hipStream_t)stream, n, P, S, idx, offsets, inv);
        return cmf_launch_status();
    }
    hipLaunchKernelGGL(inverse_count_kernel, dim3(b), dim3(INV_THREADS), (size_t)(n + 1) * sizeof(int), (hipStream_t)stream,
                       n, entries, idx, offsets);
    hipLaunchKernelGGL(inverse_fill_kernel, dim3(b, cmf_divup(n, CMF_WAVE)), dim3(CMF_WAVE), 0, (hipStream_t)stream,
                       n, entries, idx, offsets, inv);
    return cmf_launch_status();
}

// grad_feat[b, j, :] (ld = ldg) (+)= sum over entries e in inv(j) of grad_out[b, e, :]
template <int VEC>
__global__ __launch_bounds__(GR_THREADS) void group_rows_grad_kernel(
    int n, int c, int ldg, int entries, int accumulate, int total_waves,
    const float *__restrict__ grad_out, const int *__restrict__ offsets, const int *__restrict__ inv,
    float *__restrict__ grad_feat)
{
    // one wave per (sample, point): lanes stride over channels
    const int wave = (blockIdx.x * GR_THREADS + threadIdx.x) / CMF_WAVE;
    if (wave >= total_waves) return;
    const int lane = threadIdx.x % CMF_WAVE;
    const int bs = wave / n, j = wave - bs * n;
    const int *off = offsets + (size_t)bs * (n + 1);
    const int beg = off[j], end = off[j + 1];
    const int *lst = inv + (size_t)bs * entries;
    const float *g = grad_out + (size_t)bs * entries * c;
    float *dst = grad_feat + ((size_t)bs * n + j) * ldg;
    for (int col = lane * VEC; col < c; col += CMF_WAVE * VEC) {
        if (VEC == 4) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            int t = beg;
            for (; t + 8 <= end; t += 8) {          // 8 index loads, then 8 row loads in flight; sum order unchanged
                int e[8];
                float4 v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) e[q] = lst[t + q];
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = *reinterpret_cast<const float4 *>(g + (size_t)e[q] * c + col);
#pragma unroll
                for (int q = 0; q < 8; ++q) { acc.x += v[q].x; acc.y += v[q].y; acc.z += v[q].z; acc.w += v[q].w; }
            }
            for (; t + 4 <= end; t += 4) {          // 4 index loads, then 4 row loads in flight
                const int e0 = lst[t], e1 = lst[t + 1], e2 = lst[t + 2], e3 = lst[t + 3];
                const float4 v0 = *reinterpret_cast<const float4 *>(g + (size_t)e0 * c + col);
                const float4 v1 = *reinterpret_cast<const float4 *>(g + (size_t)e1 * c + col);
                const float4 v2 = *reinterpret_cast<const float4 *>(g + (size_t)e2 * c + col);
                const float4 v3 = *reinterpret_cast<const float4 *>(g + (size_t)e3 * c + col);
                acc.x += v0.x; acc.y += v0.y; acc.z += v0.z; acc.w += v0.w;
                acc.x += v1.x; acc.y += v1.y; acc.z += v1.z; acc.w += v1.w;
                acc.x += v2.x; acc.y += v2.y; acc.z += v2.z; acc.w += v2.w;
                acc.x += v3.x; acc.y += v3.y; acc.z += v3.z; acc.w += v3.w;
            }
            for (; t < end; ++t) {
                const float4 v = *reinterpret_cast<const float4 *>(g + (size_t)lst[t] * c + col);
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
            float4 *d = reinterpret_cast<float4 *>(dst + col);
            if (accumulate) { const float4 o = *d; acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w; }
            *d = acc;
        } else {
            float acc = 0.f;
            for (int t = beg; t < end; ++t) acc += g[(size_t)lst[t] * c + col];
            if (accumulate) acc += dst[col];
            dst[col] = acc;
        }
    }
}

extern "C" int cmf_group_rows_grad(int b, int n, int c, int ldg, int entries, int accumulate,
                                   const float *grad_out, const int *offsets, const int *inv,
                                   float *grad_feat, void *stream)
{
    CMF_CHECK_ARG(b >= 0 && n > 0 && c > 0 && ldg >= c && entries >= 0);
    if (b == 0) return 0;
    CMF_CHECK_ARG(grad_out && offsets && inv && grad_feat);
    const bool v4 = (c % 4 == 0) && (ldg % 4 == 0) && (((uintptr_t)grad_out | (uintptr_t)grad_feat) % 16 == 0);
    const long long waves = (long long)b * n;
    const int grid = (int)((waves * CMF_WAVE + GR_THREADS - 1) / GR_THREADS);
    if (v4) hipLaunchKernelGGL(group_rows_grad_kernel<4>, dim3(grid), dim3(GR_THREADS), 0, (hipStream_t)stream,
                               n, c, ldg, entries, accumulate, (int)waves, grad_out, offsets, inv, grad_feat);
    else hipLaunchKernelGGL(group_rows_grad_kernel<1>, dim3(grid), dim3(GR_THREADS), 0, (hipStream_t)stream,
                            n, c, ldg, entries, accumulate, (int)waves, grad_out, offsets, inv, grad_feat);
    return cmf_launch_status();
}

// ---------------------------------------------------------------------------------------------
// Set-conv first-layer backward: scatter of dZ with the BatchNorm backward fused in.
//   dZ[e,:] = a*(dU[e,:] - s1/M - zhat[e,:]*s2/M),  zhat = (z - mean)*invstd     (never written)
//   grad_feat[b,j,:] = sum_{e in inv(j)} dZ[e,:]      (ascending e: deterministic)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(GR_THREADS) void group_rows_grad_bn_kernel(
    int n, int c, int entries, int total_waves, const float *__restrict__ dU, const float *__restrict__ z,
    const float *__restrict__ a, const float *__restrict__ mean, const float *__restrict__ invstd,
    const float *__restrict__ sums, float inv_count,
    const int *__restrict__ offsets, const int *__restrict__ inv, float *__restrict__ grad_feat, int ldg)
{
    const int wave = (blockIdx.x * GR_THREADS + threadIdx.x) / CMF_WAVE;
    if (wave >= total_waves) return;
    const int lane = threadIdx.x % CMF_WAVE;
    const int bs = wave / n, j = wave - bs * n;
    const int *off = offsets + (size_t)bs * (n + 1);
    const int beg = off[j], end = off[j + 1];
    const int *lst = inv + (size_t)bs * entries;
    const float *gu = dU + (size_t)bs * entries * c;
    const float *gz = z + (size_t)bs * entries * c;
    float *dst = grad_feat + ((size_t)bs * n + j) * ldg;
    for (int col = lane * 4; col < c; col += CMF_WAVE * 4) {
        const float4 sa = *(const float4 *)(a + col);
        float4 k0 = make_float4(0.f, 0.f, 0.f, 0.f), k1 = k0, mu = k0;     // dZ = sa*(dU - k0 - (z-mu)*k1)
        if (sums) {
            const float4 t1 = *(const float4 *)(sums + col), t2 = *(const float4 *)(sums + c + col);
            const float4 is = *(const float4 *)(invstd + col);
            mu = *(const float4 *)(mean + col);
            k0 = make_float4(t1.x * inv_count, t1.y * inv_count, t1.z * inv_count, t1.w * inv_count);
            k1 = make_float4(is.x * t2.x * inv_count, is.y * t2.y * inv_count, is.z * t2.z * inv_count, is.w * t2.w * inv_count);
        }
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        int t = beg;
        for (; t + 2 <= end; t += 2) {
            const int e0 = lst[t], e1 = lst[t + 1];
            const float4 u0 = *(const float4 *)(gu + (size_t)e0 * c + col), u1 = *(const float4 *)(gu + (size_t)e1 * c + col);
            float4 z0 = mu, z1 = mu;
            if (sums) { z0 = *(const float4 *)(gz + (size_t)e0 * c + col); z1 = *(const float4 *)(gz + (size_t)e1 * c + col); }
            acc.x += sa.x * (u0.x - k0.x - (z0.x - mu.x) * k1.x); acc.y += sa.y * (u0.y - k0.y - (z0.y - mu.y) * k1.y);
            acc.z += sa.z * (u0.z - k0.z - (z0.z - mu.z) * k1.z); acc.w += sa.w * (u0.w - k0.w - (z0.w - mu.w) * k1.w);
            acc.x += sa.x * (u1.x - k0.x - (z1.x - mu.x) * k1.x); acc.y += sa.y * (u1.y - k0.y - (z1.y - mu.y) * k1.y);
            acc.z += sa.z * (u1.z - k0.z - (z1.z - mu.z) * k1.z); acc.w += sa.w * (u1.w - k0.w - (z1.w - mu.w) * k1.w);
        }
        for (; t < end; ++t) {
            const int e0 = lst[t];
            const float4 u0 = *(const float4 *)(gu + (size_t)e0 * c + col);
            float4 z0 = mu;
            if (sums) z0 = *(const float4 *)(gz + (size_t)e0 * c + col);
            acc.x += sa.x * (u0.x - k0.x - (z0.x - mu.x) * k1.x); acc.y += sa.y * (u0.y - k0.y - (z0.y - mu.y) * k1.y);
            acc.z += sa.z * (u0.z - k0.z - (z0.z - mu.z) * k1.z); acc.w += sa.w * (u0.w - k0.w - (z0.w - mu.w) * k1.w);
        }
        *(float4 *)(dst + col) = acc;
    }
}

// The same sums without reading z.  Every entry of inv(j) gathers the SAME source row, z[e,:] = y[j,:] + wx . d[e]
// with d[e] = xyz_src[j] - xyz_ctr[e / S] (cmf_group_affine), so the z-dependent part of the sum has a closed form:
//   sum_e (z[e,:] - mean) = cnt_j * (y[j,:] - mean) + wx . D_j,    D_j = sum_e d[e]   (3 floats per source point)
//   grad_feat[b,j,:] = a * (sum_e dU[e,:] - cnt_j*s1/M - (invstd*s2/M) * (cnt_j*(y[j,:] - mean) + wx . D_j)).
// Only dU is streamed (half the bytes of the kernel above); y, wx and the coordinates are per-point and cache-resident.
// DEEP: 16 rows of the inverse list in flight (neighbourhoods of 32: 296 -> 240 us at 524288 rows, 4.5 TB/s); the shorter
// lists of the small scales run 8 deep (16 costs them occupancy).  The sums keep ascending entry order either way.
template <bool DEEP>
__global__ __launch_bounds__(GR_THREADS) void group_rows_grad_bn_cf_kernel(
    int n, int c, int entries, int S, int total_waves, const float *__restrict__ dU, const float *__restrict__ y, long long ldy,
    const float *__restrict__ wx, long long ldw, const float *__restrict__ xyz_src, const float *__restrict__ xyz_ctr,
    const float *__restrict__ a, const float *__restrict__ mean, const float *__restrict__ invstd,
    const float *__restrict__ sums, float inv_count,
    const int *__restrict__ offsets, const int *__restrict__ inv, float *__restrict__ grad_feat, int ldg, const float *__restrict__ pieces)
{
    // a wave owns (source point, block of 256 columns): the inverse lists are skewed (first-hit padding: some are 10x the
    // mean), and one wave walking a long list once per column block set the pace of the launch's tail
    const int gwave = (blockIdx.x * GR_THREADS + threadIdx.x) / CMF_WAVE;
    if (gwave >= total_waves) return;
    const int lane = threadIdx.x % CMF_WAVE;
    const int cblocks = (c + CMF_WAVE * 4 - 1) / (CMF_WAVE * 4);
    const int wave = gwave / cblocks, cb = gwave - wave * cblocks;
    const int bs = wave / n, j = wave - bs * n;
    const int *off = offsets + (size_t)bs * (n + 1);
    const int beg = off[j], end = off[j + 1];
    const int *lst = inv + (size_t)bs * entries;
    const float *gu = dU + (size_t)bs * entries * c;
    float *dst = grad_feat + ((size_t)bs * n + j) * ldg;
    // D_j: lanes stride over the entries, then a fixed butterfly
    float Dx = 0.f, Dy = 0.f, Dz = 0.f;
    {
        const float *xs = xyz_src + ((size_t)bs * n + j) * 3;
        const float sx = xs[0], sy = xs[1], sz = xs[2];
        const float *xc = xyz_ctr + (size_t)bs * (entries / S) * 3;
        for (int t = beg + lane; t < end; t += CMF_WAVE) {
            const int p = lst[t] / S;
            Dx += sx - xc[p * 3]; Dy += sy - xc[p * 3 + 1]; Dz += sz - xc[p * 3 + 2];
        }
#pragma unroll
        for (int m = CMF_WAVE / 2; m > 0; m >>= 1) { Dx += __shfl_xor(Dx, m, CMF_WAVE); Dy += __shfl_xor(Dy, m, CMF_WAVE); Dz += __shfl_xor(Dz, m, CMF_WAVE); }
    }
    const float cnt = (float)(end - beg);
    for (int col = cb * CMF_WAVE * 4 + lane * 4; col < min(c, (cb + 1) * CMF_WAVE * 4); col += CMF_WAVE * 4) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        int t = beg;
        if (pieces) {
            // the sums over the point's slots were formed by the data-gradient GEMM itself (cmf_gemm_dx_gather_sum), one piece per 64-row
            // range of the inverse-ordered slots the list touches: rows (point + range) of `pieces`, added in range order
            if (end > beg) {
                const long long o0 = (long long)bs * entries + beg, o1 = (long long)bs * entries + end, pt = (long long)bs * n + j;
                for (long long R = o0 / 64; R <= (o1 - 1) / 64; ++R) {
                    const float4 u0 = *(const float4 *)(pieces + (size_t)(pt + R) * c + col);
                    acc.x += u0.x; acc.y += u0.y; acc.z += u0.z; acc.w += u0.w;
                }
            }
            t = end;
        }
        for (; DEEP && t + 16 <= end; t += 16) {            // 16 rows in flight; the sum keeps ascending entry order
            int e[16];
            float4 u[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) e[q] = lst[t + q];
#pragma unroll
            for (int q = 0; q < 16; ++q) u[q] = *(const float4 *)(gu + (size_t)e[q] * c + col);
#pragma unroll
            for (int q = 0; q < 16; ++q) { acc.x += u[q].x; acc.y += u[q].y; acc.z += u[q].z; acc.w += u[q].w; }
        }
        for (; t + 8 <= end; t += 8) {                      // 8 rows in flight
            int e[8];
            float4 u[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) e[q] = lst[t + q];
#pragma unroll
            for (int q = 0; q < 8; ++q) u[q] = *(const float4 *)(gu + (size_t)e[q] * c + col);
#pragma unroll
            for (int q = 0; q < 8; ++q) { acc.x += u[q].x; acc.y += u[q].y; acc.z += u[q].z; acc.w += u[q].w; }
        }
        for (; t + 4 <= end; t += 4) {                      // 4 rows in flight
            const int e0 = lst[t], e1 = lst[t + 1], e2 = lst[t + 2], e3 = lst[t + 3];
            const float4 u0 = *(const float4 *)(gu + (size_t)e0 * c + col), u1 = *(const float4 *)(gu + (size_t)e1 * c + col);
            const float4 u2 = *(const float4 *)(gu + (size_t)e2 * c + col), u3 = *(const float4 *)(gu + (size_t)e3 * c + col);
            acc.x += u0.x; acc.y += u0.y; acc.z += u0.z; acc.w += u0.w;
            acc.x += u1.x; acc.y += u1.y; acc.z += u1.z; acc.w += u1.w;
            acc.x += u2.x; acc.y += u2.y; acc.z += u2.z; acc.w += u2.w;
            acc.x += u3.x; acc.y += u3.y; acc.z += u3.z; acc.w += u3.w;
        }
        for (; t < end; ++t) {
            const float4 u0 = *(const float4 *)(gu + (size_t)lst[t] * c + col);
            acc.x += u0.x; acc.y += u0.y; acc.z += u0.z; acc.w += u0.w;
        }
        const float4 sa = *(const float4 *)(a + col);
        float r[4] = {acc.x, acc.y, acc.z, acc.w};
        if (sums) {
            const float4 t1 = *(const float4 *)(sums + col), t2 = *(const float4 *)(sums + c + col);
            const float4 is = *(const float4 *)(invstd + col), mu = *(const float4 *)(mean + col);
            const float4 yv = *(const float4 *)(y + ((size_t)bs * n + j) * ldy + col);
            const float s1[4] = {t1.x, t1.y, t1.z, t1.w}, s2[4] = {t2.x, t2.y, t2.z, t2.w}, iv[4] = {is.x, is.y, is.z, is.w};
            const float ym[4] = {yv.x - mu.x, yv.y - mu.y, yv.z - mu.z, yv.w - mu.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float *w = wx + (size_t)(col + i) * ldw;
                const float zsum = cnt * ym[i] + (w[0] * Dx + w[1] * Dy + w[2] * Dz);
                r[i] = r[i] - cnt * (s1[i] * inv_count) - (iv[i] * s2[i] * inv_count) * zsum;
            }
        }
        *(float4 *)(dst + col) = make_float4(sa.x * r[0], sa.y * r[1], sa.z * r[2], sa.w * r[3]);
    }
}

// Narrow rows (c = 16 .. 128: the first encoder's 32 channels): LPP = c / 4 lanes own a source point and a wave walks 64 / LPP lists at
// once -- the kernel above gives a wave one point and 256 columns, which left 56 of 64 lanes idle at c = 32 (0.5 TB/s).  Same order
// of the row sums (ascending entries); D_j is reduced over the LPP lanes of the point.
template <int LPP>
__device__ __forceinline__ void group_rows_grad_bn_cf_narrow_body(
    int n, int entries, int S, long long points, const float *__restrict__ dU, const float *__restrict__ y, long long ldy,
    const float *__restrict__ wx, long long ldw, const float *__restrict__ xyz_src, const float *__restrict__ xyz_ctr,
    const float *__restrict__ a, const float *__restrict__ mean, const float *__restrict__ invstd,
    const float *__restrict__ sums, float inv_count,
    const int *__restrict__ offsets, const int *__restrict__ inv, float *__restrict__ grad_feat, int ldg)
{
    constexpr int PPW = CMF_WAVE / LPP, c = LPP * 4;
    const long long gwave = ((long long)blockIdx.x * GR_THREADS + threadIdx.x) / CMF_WAVE;
    const int lane = threadIdx.x % CMF_WAVE, sub = lane / LPP, sl = lane % LPP;
    const long long pt = gwave * PPW + sub;
    const bool live = pt < points;
    const int bs = live ? (int)(pt / n) : 0, j = live ? (int)(pt - (long long)bs * n) : 0;
    const int *off = offsets + (size_t)bs * (n + 1);
    const int beg = live ? off[j] : 0, end = live ? off[j + 1] : 0;
    const int *lst = inv + (size_t)bs * entries;
    const float *gu = dU + (size_t)bs * entries * c;
    const int col = sl * 4;
    float Dx = 0.f, Dy = 0.f, Dz = 0.f;
    // LPP == 8 (32 channels: the first encoder): the coordinate sums D ride in the row loop -- a group of 8 entries is one entry per lane
    // of the point for D and 8 rows in flight per lane, all behind ONE index round trip (as a loop of its own in front, D cost as many
    // dependent round trips again).  Same entries per lane in the same order: bit-identical sums.
    constexpr bool MERGED = LPP == 8;
    const float *xs = xyz_src + ((size_t)bs * n + j) * 3;
    const float *xc = xyz_ctr + (size_t)bs * (entries / S) * 3;
    float sx = 0.f, sy = 0.f, sz = 0.f;
    if (sums) { sx = xs[0]; sy = xs[1]; sz = xs[2]; }
    if (sums && !MERGED) {
        for (int t = beg + sl; t < end; t += LPP) {
            const int p = lst[t] / S;
            Dx += sx - xc[p * 3]; Dy += sy - xc[p * 3 + 1]; Dz += sz - xc[p * 3 + 2];
        }
    }
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int t = beg;
    for (; t + 8 <= end; t += 8) {                          // 8 rows in flight, ascending entry order
        int e[8];
        float4 u[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) e[q] = lst[t + q];
        const int me = (MERGED && sums) ? lst[t + sl] : 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) u[q] = *(const float4 *)(gu + (size_t)e[q] * c + col);
        if (MERGED && sums) {
            const int p = me / S;
            Dx += sx - xc[p * 3]; Dy += sy - xc[p * 3 + 1]; Dz += sz - xc[p * 3 + 2];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) { acc.x += u[q].x; acc.y += u[q].y; acc.z += u[q].z; acc.w += u[q].w; }
    }
    if (MERGED && sums && t + sl < end) {                   // (fewer than 8 entries left: one per lane)
        const int p = lst[t + sl] / S;
        Dx += sx - xc[p * 3]; Dy += sy - xc[p * 3 + 1]; Dz += sz - xc[p * 3 + 2];
    }
    for (; t < end; ++t) {
        const float4 u0 = *(const float4 *)(gu + (size_t)lst[t] * c + col);
        acc.x += u0.x; acc.y += u0.y; acc.z += u0.z; acc.w += u0.w;
    }
    if (sums) {
#pragma unroll
        for (int m = LPP / 2; m > 0; m >>= 1) { Dx += __shfl_xor(Dx, m, CMF_WAVE); Dy += __shfl_xor(Dy, m, CMF_WAVE); Dz += __shfl_xor(Dz, m, CMF_WAVE); }
    }
    if (!live) return;
    const float cnt = (float)(end - beg);
    const float4 sa = *(const float4 *)(a + col);
    float r[4] = {acc.x, acc.y, acc.z, acc.w};
    if (sums) {
        const float4 t1 = *(const float4 *)(sums + col), t2 = *(const float4 *)(sums + c + col);
        const float4 is = *(const float4 *)(invstd + col), mu = *(const float4 *)(mean + col);
        const float4 yv = *(const float4 *)(y + ((size_t)bs * n + j) * ldy + col);
        const float s1[4] = {t1.x, t1.y, t1.z, t1.w}, s2[4] = {t2.x, t2.y, t2.z, t2.w}, iv[4] = {is.x, is.y, is.z, is.w};
        const float ym[4] = {yv.x - mu.x, yv.y - mu.y, yv.z - mu.z, yv.w - mu.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float *w = wx + (size_t)(col + i) * ldw;
            const float zsum = cnt * ym[i] + (w[0] * Dx + w[1] * Dy + w[2] * Dz);
            r[i] = r[i] - cnt * (s1[i] * inv_count) - (iv[i] * s2[i] * inv_count) * zsum;
        }
    }
    *(float4 *)(grad_feat + ((size_t)bs * n + j) * ldg + col) = make_float4(sa.x * r[0], sa.y * r[1], sa.z * r[2], sa.w * r[3]);
}

template <int LPP>
__global__ __launch_bounds__(GR_THREADS) void group_rows_grad_bn_cf_narrow_kernel(
    int n, int entries, int S, long long points, const float *__restrict__ dU, const float *__restrict__ y, long long ldy,
    const float *__restrict__ wx, long long ldw, const float *__restrict__ xyz_src, const float *__restrict__ xyz_ctr,
    const float *__restrict__ a, const float *__restrict__ mean, const float *__restrict__ invstd,
    const float *__restrict__ sums, float inv_count,
    const int *__restrict__ offsets, const int *__restrict__ inv, float *__restrict__ grad_feat, int ldg)
{
    group_rows_grad_bn_cf_narrow_body<LPP>(n, entries, S, points, dU, y, ldy, wx, ldw, xyz_src, xyz_ctr, a, mean, invstd, sums, inv_count, offsets,
                                           inv, grad_feat, ldg);
}

template <int LPP>
__global__ __launch_bounds__(GR_THREADS) void group_rows_grad_bn_cf_narrow_batch_kernel(const CmfBatch<CmfScatterArgs> b, long long points)
{
    const CmfScatterArgs &p = b.a[blockIdx.y];
    group_rows_grad_bn_cf_narrow_body<LPP>(p.n, p.entries, p.S, points, p.dU, p.y, p.ldy, p.wx, p.ldw, p.xyz_src, p.xyz_ctr, p.a, p.mean, p.invstd,
                                           p.sums, p.inv_count, p.offsets, p.inv, p.grad_feat, p.ldg);
}

// the scatters of up to CMF_MAX_BATCH narrow blocks over the same b x n source points (c = 16 .. 128 channels) in one launch
int cmf_group_rows_grad_bn_cf_batch(int n, int b, int c, const CmfScatterArgs *a, hipStream_t st)
{
    CMF_CHECK_ARG(n >= 1 && n <= CMF_MAX_BATCH && a && b > 0 && (c == 16 || c == 32 || c == 64 || c == 128));
    CmfBatch<CmfScatterArgs> bt;
    for (int i = 0; i < n; ++i) {
        const CmfScatterArgs &q = a[i];
        CMF_CHECK_ARG(q.n > 0 && q.n == a[0].n && q.S > 0 && q.entries % q.S == 0 && q.ldg >= c && q.ldg % 4 == 0 && q.dU && q.a && q.offsets && q.inv && q.grad_feat);
        CMF_CHECK_ARG(!q.sums || (q.y && q.wx && q.xyz_src && q.xyz_ctr && q.mean && q.invstd && q.ldy >= c && q.ldy % 4 == 0 && q.ldw >= 3));
        CMF_CHECK_ARG((((uintptr_t)q.dU | (uintptr_t)q.y | (uintptr_t)q.grad_feat | (uintptr_t)q.a) & 15) == 0);
        bt.a[i] = q;
    }
    const int lpp = c / 4;
    const long long points = (long long)b * a[0].n, nw = (points * lpp + CMF_WAVE - 1) / CMF_WAVE;
    const dim3 grid((unsigned)((nw * CMF_WAVE + GR_THREADS - 1) / GR_THREADS), n);
#define CMF_GRNB(L) hipLaunchKernelGGL(group_rows_grad_bn_cf_narrow_batch_kernel<L>, grid, dim3(GR_THREADS), 0, st, bt, points)
    if (lpp == 4) CMF_GRNB(4); else if (lpp == 8) CMF_GRNB(8); else if (lpp == 16) CMF_GRNB(16); else CMF_GRNB(32);
#undef CMF_GRNB
    return cmf_launch_status();
}

int cmf_group_rows_grad_bn_cf_impl(int b, int n, int c, int entries, int S, const float *dU, const float *y, long long ldy,
                                   const float *wx, long long ldw, const float *xyz_src, const float *xyz_ctr,
                                   const float *a, const float *mean, const float *invstd, const float *sums,
                                   float inv_count, const int *offsets, const int *inv, float *grad_feat, int ldg, const float *pieces, void *stream);

extern "C" int cmf_group_rows_grad_bn_cf(int b, int n, int c, int entries, int S, const float *dU, const float *y, long long ldy,
                                         const float *wx, long long ldw, const float *xyz_src, const float *xyz_ctr,
                                         const float *a, const float *mean, const float *invstd, const float *sums,
                                         float inv_count, const int *offsets, const int *inv, float *grad_feat, int ldg, void *stream)
{
    return cmf_group_rows_grad_bn_cf_impl(b, n, c, entries, S, dU, y, ldy, wx, ldw, xyz_src, xyz_ctr, a, mean, invstd, sums, inv_count, offsets, inv,
                                          grad_feat, ldg, nullptr, stream);
}

// ... with the per-point sums of dU already formed by cmf_gemm_dx_gather_sum (pieces, see there): dU is not read
extern "C" int cmf_group_rows_grad_bn_cf_pieces(int b, int n, int c, int entries, int S, const float *pieces, const float *y, long long ldy,
                                                const float *wx, long long ldw, const float *xyz_src, const float *xyz_ctr,
                                                const float *a, const float *mean, const float *invstd, const float *sums,
                                                float inv_count, const int *offsets, const int *inv, float *grad_feat, int ldg, void *stream)
{
    CMF_CHECK_ARG(pieces && ((uintptr_t)pieces & 15) == 0);
    return cmf_group_rows_grad_bn_cf_impl(b, n, c, entries, S, pieces, y, ldy, wx, ldw, xyz_src, xyz_ctr, a, mean, invstd, sums, inv_count, offsets, inv,
                                          grad_feat, ldg, pieces, stream);
}

int cmf_group_rows_grad_bn_cf_impl(int b, int n, int c, int entries, int S, const float *dU, const float *y, long long ldy,
                                   const float *wx, long long ldw, const float *xyz_src, const float *xyz_ctr,
                                   const float *a, const float *mean, const float *invstd, const float *sums,
                                   float inv_count, const int *offsets, const int *inv, float *grad_feat, int ldg, const float *pieces, void *stream)
{
    CMF_CHECK_ARG(b >= 0 && n > 0 && c > 0 && c % 4 == 0 && entries >= 0 && S > 0 && entries % S == 0 && ldg >= c && ldg % 4 == 0);
    if (b == 0) return 0;
    CMF_CHECK_ARG(dU && a && offsets && inv && grad_feat);
    CMF_CHECK_ARG(!sums || (y && wx && xyz_src && xyz_ctr && mean && invstd && ldy >= c && ldy % 4 == 0 && ldw >= 3));
    CMF_CHECK_ARG((((uintptr_t)dU | (uintptr_t)y | (uintptr_t)grad_feat | (uintptr_t)a) & 15) == 0);
    if (!pieces && (c == 16 || c == 32 || c == 64 || c == 128)) {
        const int lpp = c / 4;
        const long long points = (long long)b * n, nw = (points * lpp + CMF_WAVE - 1) / CMF_WAVE;
        const dim3 grid((unsigned)((nw * CMF_WAVE + GR_THREADS - 1) / GR_THREADS));
#define CMF_GRN(L) hipLaunchKernelGGL(group_rows_grad_bn_cf_narrow_kernel<L>, grid, dim3(GR_THREADS), 0, (hipStream_t)stream, n, entries, S, points, dU, y, \
                                      ldy, wx, ldw, xyz_src, xyz_ctr, a, mean, invstd, sums, inv_count, offsets, inv, grad_feat, ldg)
        if (lpp == 4) CMF_GRN(4); else if (lpp == 8) CMF_GRN(8); else if (lpp == 16) CMF_GRN(16); else CMF_GRN(32);
#undef CMF_GRN
        return cmf_launch_status();
    }
    const long long waves = (long long)b * n * ((c + CMF_WAVE * 4 - 1) / (CMF_WAVE * 4));
    CMF_CHECK_ARG(waves * CMF_WAVE < (1ll << 31));
    const int grid = (int)((waves * CMF_WAVE + GR_THREADS - 1) / GR_THREADS);
    if (S >= 32)
        hipLaunchKernelGGL(group_rows_grad_bn_cf_kernel<true>, dim3(grid), dim3(GR_THREADS), 0, (hipStream_t)stream,
                           n, c, entries, S, (int)waves, dU, y, ldy, wx, ldw, xyz_src, xyz_ctr, a, mean, invstd, sums, inv_count, offsets, inv,
                           grad_feat, ldg, pieces);
    else
        hipLaunchKernelGGL(group_rows_grad_bn_cf_kernel<false>, dim3(grid), dim3(GR_THREADS), 0, (hipStream_t)stream,
                           n, c, entries, S, (int)waves, dU, y, ldy, wx, ldw, xyz_src, xyz_ctr, a, mean, invstd, sums, inv_count, offsets, inv,
                           grad_feat, ldg, pieces);
    return cmf_launch_status();
}

extern "C" int cmf_group_rows_grad_bn(int b, int n, int c, int entries, const float *dU, const float *z,
                                      const float *a, const float *mean, const float *invstd, const float *sums,
                                      float inv_count, const int *offsets, const int *inv, float *grad_feat, int ldg, void *stream)
{
    CMF_CHECK_ARG(b >= 0 && n > 0 && c > 0 && c % 4 == 0 && entries >= 0 && ldg >= c && ldg % 4 == 0);
    if (b == 0) return 0;
    CMF_CHECK_ARG(dU && a && offsets && inv && grad_feat && (!sums || (z && mean && invstd)));
    const long long waves = (long long)b * n;
    const int grid = (int)((waves * CMF_WAVE + GR_THREADS - 1) / GR_THREADS);
    hipLaunchKernelGGL(group_rows_grad_bn_kernel, dim3(grid), dim3(GR_THREADS), 0, (hipStream_t)stream,
                       n, c, entries, (int)waves, dU, z, a, mean, invstd, sums, inv_count, offsets, inv, grad_feat, ldg);
    return cmf_launch_status();
}
