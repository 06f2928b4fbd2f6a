// Grouping gather / scatter in the reference's (B,C,N) layout:
//   group_points_kernel_fast       lib/src/group_points_gpu.cu:47-66
//   group_points_grad_kernel_fast  lib/src/group_points_gpu.cu:8-25
//
// HBM-bound byte movers: the forward writes b*c*npoints*nsample*4 bytes and reads
// only b*c*n*4 + b*npoints*nsample*4.  The reference launches one thread per output float and
// re-reads idx once per channel with an uncoalesced gather from global memory.  Here a
// workgroup owns (sample, tile of GP_TILE idx entries, chunk of GP_CH channels): the idx tile is
// read ONCE into registers, the GP_CH feature rows are staged in LDS (n floats each), the random
// gather is served by LDS, and every global store is a coalesced 16-byte store.
#include "cmf_common.h"
#include "../../include/cmflow_hip.h"

int cmf_group_points_strided(int b, int c, int n, int npoints, int nsample, const float *points, const int *idx, float *out,
                             long long out_bstride, void *stream);
int cmf_group_points_xyz(int b, int c, int n, int npoints, int nsample, const float *points, const int *idx, float *out,
                         long long out_bstride, const float *xyz, const float *new_xyz, void *stream);

constexpr int GP_THREADS = 256;
// channels per workgroup: 8 when the rows are short (8 rows of n floats in LDS), 2 for long rows so that
// several workgroups still fit on a CU (occupancy hides the LDS gather latency)
constexpr int GP_VEC = 4;             // idx entries per thread per step (one float4 store)
constexpr int GP_STEPS = 4;           // steps per thread -> GP_TILE = 256*4*4 = 4096 entries
constexpr int GP_TILE = GP_THREADS * GP_VEC * GP_STEPS;
constexpr int GP_MAX_N_LDS = 8192;    // rows staged in LDS up to this n

// XYZ planes (cmf_query_and_group): the last ceil(3 / GP_CH) channel chunks of the grid write the grouped coordinates relative
// to the centre, xyz[idx] - new_xyz (lib/pointnet2_utils.py:279-280), into planes 0..2 of `out`, and the feature planes start at
// plane 3 -- one gather launch for the whole (B, 3 + C, M, nsample) tensor.  xyz / ctr are point-major (b,n,3) / (b,m,3).
struct GpXyz { const float *xyz, *ctr; int nsample, m, chunks; };

template <bool ROWS_IN_LDS, int GP_CH, int TPB>
__global__ __launch_bounds__(TPB) void group_points_kernel(
    int c, int n, int total /* npoints*nsample */, int tiles_per_sample, int tiles_per_wg,
    const float *__restrict__ points, const int *__restrict__ idx, float *__restrict__ out, long long out_bstride, const GpXyz X)
{
    extern __shared__ __attribute__((aligned(16))) float rows[];   // [GP_CH][n] when ROWS_IN_LDS
    // a workgroup stages its feature rows once and walks tiles_per_wg consecutive idx tiles with them: with long rows
    // (n = 4096: 64 tiles per sample) one tile per workgroup re-staged the rows 64 times -- 2.7 GB of fetches
    // against 0.1 GB of algorithmic reads (rocprofv3 FETCH_SIZE), enough to pin the kernel at the HBM limit
    const int groups_per_sample = (tiles_per_sample + tiles_per_wg - 1) / tiles_per_wg;
    const int tile0 = (blockIdx.x % groups_per_sample) * tiles_per_wg;
    const int tile1 = min(tile0 + tiles_per_wg, tiles_per_sample);
    const int bs = blockIdx.x / groups_per_sample;
    const bool is_xyz = (int)blockIdx.y < X.chunks;                // the coordinate chunks come FIRST in the grid (they are the slower
                                                                   // ones per plane: behind the feature chunks they were a 150 us tail at n = 4096)
    const int c0 = is_xyz ? (int)blockIdx.y * GP_CH : ((int)blockIdx.y - X.chunks) * GP_CH;
    const int nch = min(GP_CH, (is_xyz ? 3 : c) - c0);
    const int *ix = idx + (size_t)bs * total;
    // a "row" = the n values of one plane: features[bs][c0 + ch][:] (unit stride) or coordinate c0 + ch of xyz[bs] (stride 3)
    const float *src = is_xyz ? X.xyz + (size_t)bs * n * 3 + c0 : points + ((size_t)bs * c + c0) * n;
    float *dst = out + (size_t)bs * out_bstride + (size_t)((is_xyz ? 0 : (X.chunks ? 3 : 0)) + c0) * total;
    const float *ctr = is_xyz ? X.ctr + (size_t)bs * X.m * 3 + c0 : nullptr;

    if (ROWS_IN_LDS) {
        if (!is_xyz) { for (int i = threadIdx.x; i < nch * n; i += TPB) rows[i] = src[i]; }
        else for (int k = threadIdx.x; k < n; k += TPB)
            for (int ch = 0; ch < nch; ++ch) rows[(size_t)ch * n + k] = src[(size_t)k * 3 + ch];
    }
    const bool vec_ok = (total % GP_VEC) == 0;
    // Workgroups of different channel chunks walk their tiles in rotated order.  With every workgroup at tile t the launch wrote
    // the same offset of up to 2048 planes that lie a power of two apart (n x nsample x 4 bytes = 1 MB at BASELINE config 5) at
    // the same time, and those streams pile up on few HBM channels: 516-630 us against 436-460 us rotated, (32,4096,64,C = 64),
    // four placements of the tensors each (tools/gp_probe.py).  [Also tried: rotation by sample, by sample and chunk, by a hash
    // of the block index: equal or worse.]
    const int rot = (int)blockIdx.y % (tile1 - tile0);
    for (int ti = tile0; ti < tile1; ++ti) {
    const int tile = tile0 + (ti - tile0 + rot) % (tile1 - tile0);
    const int e0 = tile * (TPB * GP_VEC * GP_STEPS);
    // this thread's idx entries, loaded once and reused for every channel
    int my[GP_STEPS][GP_VEC];
#pragma unroll
    for (int s = 0; s < GP_STEPS; ++s) {
        const int e = e0 + (s * TPB + threadIdx.x) * GP_VEC;
        if (vec_ok && e + GP_VEC <= total) {
            const int4 v = *reinterpret_cast<const int4 *>(ix + e);
            my[s][0] = v.x; my[s][1] = v.y; my[s][2] = v.z; my[s][3] = v.w;
        } else {
#pragma unroll
            for (int j = 0; j < GP_VEC; ++j) my[s][j] = (e + j < total) ? ix[e + j] : 0;
        }
    }
    if (ROWS_IN_LDS && ti == tile0) __syncthreads();

    if (!is_xyz) {                                      // feature planes: the plain gather
    for (int ch = 0; ch < nch; ++ch) {
        const float *row = ROWS_IN_LDS ? rows + (size_t)ch * n : src + (size_t)ch * n;
        float *o = dst + (size_t)ch * total;
#pragma unroll
        for (int s = 0; s < GP_STEPS; ++s) {
            const int e = e0 + (s * TPB + threadIdx.x) * GP_VEC;
            if (vec_ok && e + GP_VEC <= total) {
                float4 v;
                v.x = row[my[s][0]]; v.y = row[my[s][1]]; v.z = row[my[s][2]]; v.w = row[my[s][3]];
                *reinterpret_cast<float4 *>(o + e) = v;
            } else {
#pragma unroll
                for (int j = 0; j < GP_VEC; ++j)
                    if (e + j < total) o[e + j] = row[my[s][j]];
            }
        }
    }
    } else {                                            // coordinate planes: gather minus the entry's centre (ONE subtraction: bit-equal)
    for (int ch = 0; ch < nch; ++ch) {
        float *o = dst + (size_t)ch * total;
#pragma unroll
        for (int s = 0; s < GP_STEPS; ++s) {
            const int e = e0 + (s * TPB + threadIdx.x) * GP_VEC;
            if (e >= total) continue;
            const int pe = e / X.nsample, re = e - pe * X.nsample;      // one division per 4 entries: they mostly share the centre
            float v[GP_VEC];
#pragma unroll
            for (int j = 0; j < GP_VEC; ++j) {
                const float a = ROWS_IN_LDS ? rows[(size_t)ch * n + my[s][j]] : src[ch + (size_t)my[s][j] * 3];
                const int pj = re + j < X.nsample ? pe : (e + j) / X.nsample;
                v[j] = a - ((e + j < total) ? ctr[(size_t)pj * 3 + ch] : 0.f);
            }
            if (vec_ok && e + GP_VEC <= total) *reinterpret_cast<float4 *>(o + e) = make_float4(v[0], v[1], v[2], v[3]);
            else {
#pragma unroll
                for (int j = 0; j < GP_VEC; ++j) if (e + j < total) o[e + j] = v[j];
            }
        }
    }
    }
    }
}

extern "C" int cmf_group_points(int b, int c, int n, int npoints, int nsample,
                                const float *points, const int *idx, float *out, void *stream)
{
    CMF_CHECK_ARG(b >= 0 && c >= 0 && n >= 0 && npoints >= 0 && nsample >= 0);
    return cmf_group_points_strided(b, c, n, npoints, nsample, points, idx, out, (long long)c * npoints * nsample, stream);
}

// out_bstride: floats between consecutive samples of `out` (c * npoints * nsample for a dense output; larger when the
// feature planes are part of a wider tensor: cmf_query_and_group)
int cmf_group_points_strided(int b, int c, int n, int npoints, int nsample, const float *points, const int *idx, float *out,
                             long long out_bstride, void *stream)
{
    return cmf_group_points_xyz(b, c, n, npoints, nsample, points, idx, out, out_bstride, nullptr, nullptr, stream);
}

// + the relative-coordinate planes when xyz / new_xyz are given: out is then (b, 3 + c, npoints, nsample)
int cmf_group_points_xyz(int b, int c, int n, int npoints, int nsample, const float *points, const int *idx, float *out,
                         long long out_bstride, const float *xyz, const float *new_xyz, void *stream)
{
    CMF_CHECK_ARG(b >= 0 && c >= 0 && n >= 0 && npoints >= 0 && nsample >= 0);
    const long long total = (long long)npoints * nsample;
    const bool with_xyz = xyz != nullptr;
    if (b == 0 || (c == 0 && !with_xyz) || total == 0) return 0;
    CMF_CHECK_ARG((points || c == 0) && idx && out && n > 0 && total < (1LL << 31) && out_bstride >= (long long)(c + (with_xyz ? 3 : 0)) * total);
    CMF_CHECK_ARG(!with_xyz || new_xyz);
    GpXyz X{xyz, new_xyz, nsample, npoints, 0};
#define CMF_GP_CHUNKS(CH) (X.chunks = with_xyz ? (3 + (CH) - 1) / (CH) : 0, (unsigned)(cmf_divup(c, (CH)) + X.chunks))
    int tiles = cmf_divup(total, GP_TILE);
    hipStream_t st = (hipStream_t)stream;
    // tiles per workgroup: as many as possible while the launch still has >= ~1024 workgroups
    auto tiles_per_wg = [&](int chan_groups) {
        const long long base = (long long)b * chan_groups;
        int t = (int)((base * tiles) / 1024);
        return t < 1 ? 1 : (t > tiles ? tiles : t);
    };
    if (n <= 1024) {
        const int t = tiles_per_wg(cmf_divup(c, 8));
        dim3 grid((unsigned)(cmf_divup(tiles, t) * (long long)b), CMF_GP_CHUNKS(8));
        hipLaunchKernelGGL((group_points_kernel<true, 8, GP_THREADS>), grid, dim3(GP_THREADS), (size_t)8 * n * sizeof(float), st,
                           c, n, (int)total, tiles, t, points, idx, out, out_bstride, X);
    } else if (n <= GP_MAX_N_LDS) {
        // 4 rows when they fit in 64 KB (two workgroups per CU), else 2
        const bool four = (size_t)4 * n * sizeof(float) <= 64 * 1024;
        static CmfPerDevice attr_set;
        int attr_dev;
        if (attr_set.need(attr_dev)) {
            (void)hipFuncSetAttribute((const void *)group_points_kernel<true, 2, 512>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, 2 * GP_MAX_N_LDS * 4);
            (void)hipFuncSetAttribute((const void *)group_points_kernel<true, 4, 512>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
            attr_set.done(attr_dev);
        }
        tiles = cmf_divup(total, 512 * GP_VEC * GP_STEPS);                 // 512 threads: twice the waves per CU for the LDS gather
        if (four) {
            const int t = tiles_per_wg(cmf_divup(c, 4));
            dim3 grid((unsigned)(cmf_divup(tiles, t) * (long long)b), CMF_GP_CHUNKS(4));
            hipLaunchKernelGGL((group_points_kernel<true, 4, 512>), grid, dim3(512), (size_t)4 * n * sizeof(float), st,
                               c, n, (int)total, tiles, t, points, idx, out, out_bstride, X);
        } else {
            const int t = tiles_per_wg(cmf_divup(c, 2));
            dim3 grid((unsigned)(cmf_divup(tiles, t) * (long long)b), CMF_GP_CHUNKS(2));
            hipLaunchKernelGGL((group_points_kernel<true, 2, 512>), grid, dim3(512), (size_t)2 * n * sizeof(float), st,
                               c, n, (int)total, tiles, t, points, idx, out, out_bstride, X);
        }
    } else {
        dim3 grid((unsigned)(tiles * (long long)b), CMF_GP_CHUNKS(8));
        hipLaunchKernelGGL((group_points_kernel<false, 8, GP_THREADS>), grid, dim3(GP_THREADS), 0, st,
                           c, n, (int)total, tiles, 1, points, idx, out, out_bstride, X);
    }
#undef CMF_GP_CHUNKS
    return cmf_launch_status();
}

// ---------------------------------------------------------------------------------------------
// Backward: grad_points[b,c,idx[b,p,s]] += grad_out[b,c,p,s].
// The reference issues one global fp32 atomicAdd per element (538 M atomics for one scale of
// mse_layer2, order undefined).  grad_out is read exactly once, in full coalesced rows; the scatter
// happens in LDS.  Two kernels:
//
//  * rows of up to 8192 entries (every shape of the model, N = 256): "balanced" kernel.  The scatter
//    becomes a GATHER over the inverse index of idx (cmf_build_inverse: entries sorted by target).  A
//    workgroup owns (sample, a run of channels); thread t owns the E consecutive SORTED entries
//    t*E..t*E+E-1 -- their positions in the row and the segment boundaries among them live in
//    registers for the whole channel loop, so per channel a thread does E independent LDS reads of
//    the staged row and a short segmented sum; nothing depends on how skewed the index is (with
//    first-hit padding the lists of low-numbered points are 10x the mean).  Segments that span
//    chunks are stitched from per-chunk tails in fixed order: no atomics, bit-reproducible (the
//    association differs from a sequential scan, so equal to the oracle to rounding, not bitwise).
//    The next channel's row is fetched into registers while the current one is reduced.
//  * longer rows (N = 4096 x K = 64: 1 MB per row), per-centre lists of 16 / 32 / 64 slots: the pad-folded CSR gather
//    (gpg_csr_index_kernel + gpg_csr_gather_kernel below; the DEFAULT since round 3): deterministic, no atomics.
//    Other list lengths / n > 8192: workgroup per (sample, 4 channels), the row streamed with 16-byte loads, idx read
//    once per 4 channels, sums kept in LDS per target with ds_add_f32 (order undefined, like the reference; LDS atomics
//    instead of HBM atomics), or the tiled deterministic kernel (opt-in).
// ---------------------------------------------------------------------------------------------
int cmf_build_inverse_rows(int b, int n, int P, int S, const int *idx, int *offsets, int *inv, void *stream, int entries);

constexpr int GG_THREADS = 256;
constexpr int GG_CH = 16;                  // channels per workgroup (balanced kernel): amortises the index set-up
constexpr int GG_MAX_E = 32;               // sorted entries per thread  => rows of up to 8192 entries
constexpr int GG_MAX_N_BAL = 4096;         // targets per sample the balanced kernel keeps tables for

// inclusive prefix sum over the 256 threads of a workgroup (4 waves); wsum: 4 ints of LDS
__device__ __forceinline__ int gg_block_scan(int v, int *wsum, int &block_total)
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
    for (int w = 0; w < GG_THREADS / 64; ++w) {
        const int x = wsum[w];
        if (w < wave) add += x;
        tot += x;
    }
    block_total = tot;
    return v + add;
}

template <int E>
__global__ __launch_bounds__(GG_THREADS, 4) void group_points_grad_bal_kernel(
    int c, int n, int total, int ch_per_wg, const float *__restrict__ grad_out, const int *__restrict__ offsets,
    const int *__restrict__ inv, float *__restrict__ grad_points)
{
    constexpr int RS = GG_THREADS * E + 4;                          // row buffer: entries + one zero slot (padding reads)
    constexpr int JT = 2;                                           // targets per thread whose tables live in registers
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *row = sm;
    float *ranked = row + RS;                                       // [n]   sum up to each segment end inside its chunk
    float *tails = ranked + n;                                      // [256] chunk sum after its last boundary
    int *offs = reinterpret_cast<int *>(tails + GG_THREADS);        // [n+1]
    int *rank_of = offs + n + 1;                                    // [n]   rank among non-empty targets, -1 if empty
    int *scan = rank_of + n;                                        // [4]   scratch for the block scans
    const int tid = threadIdx.x;
    const int bs = blockIdx.x;
    const int c0 = blockIdx.y * ch_per_wg;
    const int nch = min(ch_per_wg, c - c0);
    const int *off = offsets + (size_t)bs * (n + 1);
    const int *lst = inv + (size_t)bs * total;
    const float *g = grad_out + ((size_t)bs * c + c0) * total;
    float *gp = grad_points + ((size_t)bs * c + c0) * n;

    // ---- per-sample set-up (once per workgroup) ----
    for (int j = tid; j <= n; j += GG_THREADS) offs[j] = off[j];
    unsigned char *flags = reinterpret_cast<unsigned char *>(row);  // aliases the row buffer until the first row arrives
    for (int i = tid; i < GG_THREADS * E / 4; i += GG_THREADS) reinterpret_cast<int *>(flags)[i] = 0;
    __syncthreads();
    for (int j = tid; j < n; j += GG_THREADS)
        if (offs[j + 1] > offs[j]) flags[offs[j + 1] - 1] = 1;
    __syncthreads();
    unsigned pos2[E / 2];                                           // two 16-bit row positions per register
    unsigned mask = 0;
#pragma unroll
    for (int k = 0; k < E; k += 2) {
        const int t = tid * E + k;
        const unsigned p0 = t < total ? (unsigned)lst[t] : GG_THREADS * E;        // padding reads the zero slot
        const unsigned p1 = t + 1 < total ? (unsigned)lst[t + 1] : GG_THREADS * E;
        pos2[k / 2] = p0 | (p1 << 16);
        mask |= ((unsigned)flags[t] << k) | ((unsigned)flags[t + 1] << (k + 1));
    }
    // rank of this chunk's first boundary = boundaries in earlier chunks (exclusive block scan)
    int tot;
    const int rank0 = gg_block_scan(__popc(mask), scan, tot) - __popc(mask);
    // rank_of[j] = non-empty targets before j (targets in blocks of 256, running base)
    int base = 0;
    for (int j0 = 0; j0 < n; j0 += GG_THREADS) {
        const int j = j0 + tid;
        const int ne = (j < n && offs[j + 1] > offs[j]) ? 1 : 0;
        const int inc = gg_block_scan(ne, scan, tot);
        if (j < n) rank_of[j] = ne ? base + inc - 1 : -1;
        base += tot;
    }
    __syncthreads();
    int rj[JT], sc[JT], ec[JT];                                     // this thread's first JT targets: tid, tid+256
#pragma unroll
    for (int q = 0; q < JT; ++q) {
        const int j = tid + q * GG_THREADS;
        rj[q] = j < n ? rank_of[j] : -1;
        sc[q] = rj[q] >= 0 ? offs[j] / E : 0;
        ec[q] = rj[q] >= 0 ? (offs[j + 1] - 1) / E : 0;
    }
    // ---- rows ----
    constexpr int V = E / 4;                                        // float4 per thread per row
    const bool vec = (total & 3) == 0 && ((uintptr_t)grad_out & 15) == 0;
    auto fetch = [&](const float *src, float4 (&r)[V]) {
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const int i = (v * GG_THREADS + tid) * 4;
            if (vec) r[v] = i < total ? *(const float4 *)(src + i) : make_float4(0.f, 0.f, 0.f, 0.f);
            else {
                r[v].x = i < total ? src[i] : 0.f;         r[v].y = i + 1 < total ? src[i + 1] : 0.f;
                r[v].z = i + 2 < total ? src[i + 2] : 0.f; r[v].w = i + 3 < total ? src[i + 3] : 0.f;
            }
        }
    };
    auto stash = [&](const float4 (&r)[V]) {
#pragma unroll
        for (int v = 0; v < V; ++v) *(float4 *)(row + (v * GG_THREADS + tid) * 4) = r[v];
        if (tid == 0) row[GG_THREADS * E] = 0.f;
    };
    float4 nxt[V];
    fetch(g, nxt);
    stash(nxt);
    __syncthreads();
    for (int ch = 0; ch < nch; ++ch) {
        if (ch + 1 < nch) fetch(g + (size_t)(ch + 1) * total, nxt);  // in flight during the reduction below
        float old[JT];                                              // grad_points accumulates: fetch the old value early
#pragma unroll
        for (int q = 0; q < JT; ++q) old[q] = rj[q] >= 0 ? gp[(size_t)ch * n + tid + q * GG_THREADS] : 0.f;
        // phase 1: segmented sums over this thread's E sorted entries (LDS reads issued 8 at a time)
        float run = 0.f;
        int rk = rank0;
#pragma unroll
        for (int k0 = 0; k0 < E; k0 += 8) {
            float val[8];
#pragma unroll
            for (int k = 0; k < 8 && k0 + k < E; ++k) {
                const unsigned pp = pos2[(k0 + k) / 2];
                val[k] = row[((k0 + k) & 1) ? (pp >> 16) : (pp & 0xffffu)];
            }
#pragma unroll
            for (int k = 0; k < 8 && k0 + k < E; ++k) {
                run += val[k];
                if ((mask >> (k0 + k)) & 1u) { ranked[rk++] = run; run = 0.f; }
            }
        }
        tails[tid] = run;
        __syncthreads();
        if (ch + 1 < nch) stash(nxt);                               // every read of the current row is done
        // phase 2: one target per thread; a segment that began in earlier chunks adds their tails in order
#pragma unroll
        for (int q = 0; q < JT; ++q)
            if (rj[q] >= 0) {
                float s = 0.f;
                for (int k = sc[q]; k < ec[q]; ++k) s += tails[k];
                s += ranked[rj[q]];
                gp[(size_t)ch * n + tid + q * GG_THREADS] = old[q] + s;
            }
        for (int j = tid + JT * GG_THREADS; j < n; j += GG_THREADS) {
            const int r = rank_of[j];
            if (r < 0) continue;
            const int s0 = offs[j] / E, e0 = (offs[j + 1] - 1) / E;
            float s = 0.f;
            for (int k = s0; k < e0; ++k) s += tails[k];
            s += ranked[r];
            gp[(size_t)ch * n + j] += s;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Plan form of the balanced kernel (rows of up to 8192 entries over up to 2048 targets: the model's ball-query / kNN lists).
// What made narrow features slow was everything AROUND the rows: the inverse index was a separate launch of one
// workgroup per sample with 256-step dependent LDS chains (21-59 us), and every scatter workgroup then rebuilt its
// tables from that index (flags, two block scans per 256 targets, 32 list entries per thread) -- ~67 us of fixed cost
// against 0.45 us per channel row, i.e. (64,256,32) with C = 3 / 64 ran at 0.016 / 0.18 of HBM.  Here ONE kernel per call
// (gpg_plan_kernel, a workgroup per sample) turns idx straight into the register image of the scatter workgroups: per
// thread the row positions of its E sorted entries (two 16-bit positions per word), the run-end mask, the rank of its
// first run end; per target (rank among non-empty targets | first chunk | last chunk).  A scatter workgroup loads that
// image (E / 2 + 2 words per thread, n words through LDS) and starts streaming rows.  Same order of additions as
// group_points_grad_bal_kernel: deterministic, bit-reproducible.
//
// The plan kernel is a stable counting sort of the row's entries by target, without atomics and without per-row serial loops
// (round 4; the first form gave a thread one idx row and walked it with dependent LDS read-modify-writes against a dense
// rows x targets count matrix: 17 us per call on 4 waves, 37 % of the C = 64 call of (64,256,32)).  16 waves; a wave owns
// E / 4 consecutive 64-entry chunks and keeps its entries in registers through all phases:
//   A  per chunk, the lanes holding the same target find each other with one ballot per key bit; a lane's rank among them is a
//      population count, the wave's running count per target (run[wave][target], LDS) gives its rank inside the wave;
//   B  a thread per target turns the 16 per-wave counts into exclusive prefixes, a block scan of the totals gives offs[];
//   C  every entry goes to offs[target] + run[wave][target] + rank, tagged (bit 15) if it is the last of its target;
//   D  the scatter threads' register image is read off the sorted row.
// ---------------------------------------------------------------------------------------------------------------
__host__ __device__ inline size_t gpg_plan_words(int E, int n) { return (size_t)GG_THREADS * (E / 2) + 2 * GG_THREADS + (size_t)n; }
constexpr int GPL_THREADS = 1024;
constexpr int GPL_WAVES = GPL_THREADS / 64;
constexpr int GPL_MAX_N = 2048;              // targets the plan kernel keeps per-wave counts for (16 x n x 2 bytes of LDS)
inline size_t gpg_plan_lds(int E, int n)
{
    return ((size_t)GPL_WAVES * n * 2 + 15) / 16 * 16 + (size_t)GG_THREADS * E * 2 + (size_t)(n + 1) * 4;
}

// inclusive prefix sum over the 1024 threads of the plan workgroup; wsum: 16 ints of LDS
__device__ __forceinline__ int gp_block_scan(int v, int *wsum, int &block_total)
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
    for (int w = 0; w < GPL_WAVES; ++w) {
        const int x = wsum[w];
        if (w < wave) add += x;
        tot += x;
    }
    block_total = tot;
    return v + add;
}

// phases A-C (see above): the row's entries sorted by target into `sorted` (position | last-of-target << 15), offs[n + 1]
template <int E>
__device__ __forceinline__ void gpg_sort_row(int n, int total, const int *__restrict__ ix, unsigned short *run, unsigned short *sorted, int *offs, int *scan)
{
    constexpr int CH = E / 4;                                         // 64-entry chunks per wave
    constexpr unsigned PAD = (unsigned)(GG_THREADS * E);              // padding reads the zero slot behind the row
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    int key[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int e = (wave * CH + c) * 64 + lane;
        key[c] = e < total ? ix[e] : -1;
    }
    for (int i = t; i < (int)(((size_t)GPL_WAVES * n * 2 + 15) / 16); i += GPL_THREADS) reinterpret_cast<uint4 *>(run)[i] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    // A
    const int kb = 32 - __clz(max(n - 1, 1));
    const unsigned long long lt = (1ull << lane) - 1ull;
    unsigned short *myrun = run + (size_t)wave * n;
    int rank[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const bool valid = key[c] >= 0;
        unsigned long long peers = __ballot(valid);
        for (int bit = 0; bit < kb; ++bit) {
            const bool one = (key[c] >> bit) & 1;
            const unsigned long long bm = __ballot(one);
            peers &= one ? bm : ~bm;
        }
        rank[c] = 0;
        if (valid) {
            const int seen = myrun[key[c]];
            rank[c] = seen + __popcll(peers & lt);
            if ((peers >> lane) == 1ull) myrun[key[c]] = (unsigned short)(seen + __popcll(peers));   // the highest lane of the group
        }
    }
    __syncthreads();
    // B
    int base = 0;
    for (int j0 = 0; j0 < n; j0 += GPL_THREADS) {
        const int j = j0 + t;
        int tot_j = 0;
        if (j < n) {
            unsigned short v[GPL_WAVES];
#pragma unroll
            for (int w = 0; w < GPL_WAVES; ++w) v[w] = run[(size_t)w * n + j];
#pragma unroll
            for (int w = 0; w < GPL_WAVES; ++w) { run[(size_t)w * n + j] = (unsigned short)tot_j; tot_j += v[w]; }
        }
        int tot;
        const int incl = gp_block_scan(tot_j, scan, tot);
        if (j < n) offs[j] = base + incl - tot_j;
        base += tot;
    }
    if (t == 0) offs[n] = base;
    __syncthreads();
    // C: ascending position inside a target's list (waves, chunks and lanes all ascend with the position)
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int e = (wave * CH + c) * 64 + lane;
        if (key[c] >= 0) {
            const int dst = offs[key[c]] + myrun[key[c]] + rank[c];
            sorted[dst] = (unsigned short)(e | (dst + 1 == offs[key[c] + 1] ? 0x8000 : 0));
        }
    }
    for (int i = total + t; i < GG_THREADS * E; i += GPL_THREADS) sorted[i] = (unsigned short)PAD;
    __syncthreads();
}

// [Round 6, measured and not kept: a one-launch form for C <= 4 -- this sort followed by a thread per (channel, target) summing the
//  target's entries, from global memory (24.7 us) or from rows parked in LDS (40 us walked one by one, 22.7 us with eight entries in
//  flight) -- against 18.2 us for the plan + scatter pair at (64,256,32,3): one 1024-thread workgroup per sample leaves 192 of 256 CUs
//  idle for the whole call and the per-target walk is a chain of dependent LDS reads as long as the sample's longest list, where the
//  scatter kernel's 4 x 64 workgroups spread both.  The launch floor of the box is 3.6 us (bench.py roofline_hbm.launch_floor_us): the pair
//  runs at 5 floors for 8.6 MB.]
template <int E>
__global__ __launch_bounds__(GPL_THREADS) void gpg_plan_kernel(int n, int total, const int *__restrict__ idx, unsigned *__restrict__ plan_all)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char psm[];
    __shared__ int scan[GPL_WAVES];
    unsigned short *run = reinterpret_cast<unsigned short *>(psm);    // [GPL_WAVES][n] per-wave counts, then exclusive prefixes over the waves
    size_t o = ((size_t)GPL_WAVES * n * 2 + 15) / 16 * 16;
    unsigned short *sorted = reinterpret_cast<unsigned short *>(psm + o); o += (size_t)GG_THREADS * E * 2;   // positions by target | last-of-target << 15
    int *offs = reinterpret_cast<int *>(psm + o);                     // [n + 1]
    const int t = threadIdx.x, bs = blockIdx.x;
    unsigned *plan = plan_all + (size_t)bs * gpg_plan_words(E, n);
    gpg_sort_row<E>(n, total, idx + (size_t)bs * total, run, sorted, offs, scan);
    int base = 0;
    // D: the scatter workgroup's register image
    const unsigned *sw = reinterpret_cast<const unsigned *>(sorted);
    for (int i = t; i < GG_THREADS * (E / 2); i += GPL_THREADS) plan[i] = sw[i] & 0x7FFF7FFFu;
    unsigned mask = 0;
    if (t < GG_THREADS) {
#pragma unroll
        for (int k = 0; k < E / 2; ++k) {
            const unsigned w = sw[t * (E / 2) + k];
            mask |= ((w >> 15) & 1u) << (2 * k) | ((w >> 31) & 1u) << (2 * k + 1);
        }
    }
    unsigned *pm = plan + (size_t)GG_THREADS * (E / 2);
    int tot;
    const int rank0 = gp_block_scan(__popc(mask), scan, tot) - __popc(mask);
    if (t < GG_THREADS) { pm[t] = mask; pm[GG_THREADS + t] = (unsigned)rank0; }
    unsigned *pt = pm + 2 * GG_THREADS;                               // per target: rank (16) | first chunk (8) | last chunk (8); 0xFFFF....: empty
    base = 0;
    for (int j0 = 0; j0 < n; j0 += GPL_THREADS) {
        const int j = j0 + t;
        const int ne = (j < n && offs[j + 1] > offs[j]) ? 1 : 0;
        const int inc = gp_block_scan(ne, scan, tot);
        if (j < n) pt[j] = ne ? ((unsigned)(base + inc - 1) | ((unsigned)(offs[j] / E) << 16) | ((unsigned)((offs[j + 1] - 1) / E) << 24)) : 0xFFFFFFFFu;
        base += tot;
    }
}

template <int E>
__global__ __launch_bounds__(GG_THREADS, 4) void group_points_grad_plan_kernel(
    int c, int n, int total, int ch_per_wg, const float *__restrict__ grad_out, const unsigned *__restrict__ plan_all,
    float *__restrict__ grad_points)
{
    constexpr int RS = GG_THREADS * E + 4;                          // row buffer: entries + one zero slot (padding reads)
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *row = sm;
    float *ranked = row + RS;                                       // [n]   sum up to each run end inside its chunk
    float *tails = ranked + n;                                      // [256] chunk sum after its last run end
    unsigned *tgt = reinterpret_cast<unsigned *>(tails + GG_THREADS);   // [n]
    const int tid = threadIdx.x, bs = blockIdx.x;
    const int c0 = blockIdx.y * ch_per_wg;
    const int nch = min(ch_per_wg, c - c0);
    const unsigned *plan = plan_all + (size_t)bs * gpg_plan_words(E, n);
    const float *g = grad_out + ((size_t)bs * c + c0) * total;
    float *gp = grad_points + ((size_t)bs * c + c0) * n;
    constexpr int V = E / 4;
    const bool vec = (total & 3) == 0 && ((uintptr_t)grad_out & 15) == 0;
    auto fetch = [&](const float *src, float4 (&r)[V]) {
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const int i = (v * GG_THREADS + tid) * 4;
            if (vec) r[v] = i < total ? *(const float4 *)(src + i) : make_float4(0.f, 0.f, 0.f, 0.f);
            else {
                r[v].x = i < total ? src[i] : 0.f;         r[v].y = i + 1 < total ? src[i + 1] : 0.f;
                r[v].z = i + 2 < total ? src[i + 2] : 0.f; r[v].w = i + 3 < total ? src[i + 3] : 0.f;
            }
        }
    };
    auto stash = [&](const float4 (&r)[V]) {
#pragma unroll
        for (int v = 0; v < V; ++v) *(float4 *)(row + (v * GG_THREADS + tid) * 4) = r[v];
        if (tid == 0) row[GG_THREADS * E] = 0.f;
    };
    float4 nxt[V];
    fetch(g, nxt);                                                  // the first row is requested before the tables
    unsigned pos2[E / 2];
#pragma unroll
    for (int k = 0; k < E / 2; ++k) pos2[k] = plan[(size_t)tid * (E / 2) + k];
    const unsigned *pm = plan + (size_t)GG_THREADS * (E / 2);
    const unsigned mask = pm[tid];
    const int rank0 = (int)pm[GG_THREADS + tid];
    for (int j = tid; j < n; j += GG_THREADS) tgt[j] = pm[2 * GG_THREADS + j];
    stash(nxt);
    __syncthreads();
    for (int ch = 0; ch < nch; ++ch) {
        if (ch + 1 < nch) fetch(g + (size_t)(ch + 1) * total, nxt);  // in flight during the reduction below
        float run = 0.f;
        int rk = rank0;
#pragma unroll
        for (int k0 = 0; k0 < E; k0 += 8) {
            float val[8];
#pragma unroll
            for (int k = 0; k < 8 && k0 + k < E; ++k) {
                const unsigned pp = pos2[(k0 + k) / 2];
                val[k] = row[((k0 + k) & 1) ? (pp >> 16) : (pp & 0xffffu)];
            }
#pragma unroll
            for (int k = 0; k < 8 && k0 + k < E; ++k) {
                run += val[k];
                if ((mask >> (k0 + k)) & 1u) { ranked[rk++] = run; run = 0.f; }
            }
        }
        tails[tid] = run;
        __syncthreads();
        if (ch + 1 < nch) stash(nxt);                               // every read of the current row is done
        for (int j = tid; j < n; j += GG_THREADS) {
            const unsigned w = tgt[j];
            if (w == 0xFFFFFFFFu) continue;
            float s = 0.f;
            for (int k = (int)((w >> 16) & 0xFFu); k < (int)(w >> 24); ++k) s += tails[k];
            s += ranked[w & 0xFFFFu];
            gp[(size_t)ch * n + j] += s;
        }
        __syncthreads();
    }
}

template <int E>
static int launch_plan(int b, int c, int n, int total, int ch_per_wg, const float *grad_out, const int *idx,
                       unsigned *plan, float *grad_points, hipStream_t st)
{
    static CmfPerDevice attr_set;
    int attr_dev;
    if (attr_set.need(attr_dev)) {
        (void)hipFuncSetAttribute((const void *)gpg_plan_kernel<E>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gpg_plan_lds(E, GPL_MAX_N));
        (void)hipFuncSetAttribute((const void *)group_points_grad_plan_kernel<E>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  ((GG_THREADS * E + 4) + 2 * GG_MAX_N_BAL + GG_THREADS) * 4);
        attr_set.done(attr_dev);
    }
    hipLaunchKernelGGL(gpg_plan_kernel<E>, dim3(b), dim3(GPL_THREADS), gpg_plan_lds(E, n), st, n, total, idx, plan);
    const size_t lds = (size_t)((GG_THREADS * E + 4) + 2 * n + GG_THREADS) * 4;
    hipLaunchKernelGGL(group_points_grad_plan_kernel<E>, dim3(b, cmf_divup(c, ch_per_wg)), dim3(GG_THREADS), lds, st, c, n, total, ch_per_wg,
                       grad_out, plan, grad_points);
    return cmf_launch_status();
}

constexpr int GA_CH = 4;                    // channels per workgroup (streamed kernel)
constexpr int GA_THREADS = 512;
constexpr int GA_RUN = 8;                   // consecutive entries per thread

__global__ __launch_bounds__(GA_THREADS) void group_points_grad_stream_kernel(
    int c, int n, int total, const float *__restrict__ grad_out, const int *__restrict__ idx,
    float *__restrict__ grad_points)
{
    extern __shared__ __attribute__((aligned(16))) float acc[];     // [GA_CH][n]
    const int tid = threadIdx.x;
    const int bs = blockIdx.x;
    const int c0 = blockIdx.y * GA_CH;
    const int nch = min(GA_CH, c - c0);
    const int *id = idx + (size_t)bs * total;
    const float *g = grad_out + ((size_t)bs * c + c0) * total;
    float *gp = grad_points + ((size_t)bs * c + c0) * n;
    for (int i = tid; i < GA_CH * n; i += GA_THREADS) acc[i] = 0.f;
    __syncthreads();
    if ((total & 3) == 0 && (((uintptr_t)grad_out | (uintptr_t)idx) & 15) == 0) {
        // a thread owns GA_RUN consecutive entries (neighbours of one centre: with first-hit padding mostly
        // the SAME target) and adds each run of equal targets once -- the same-address LDS conflicts that
        // serialise a naive per-entry ds_add_f32 disappear
        for (int i0 = tid * GA_RUN; i0 < total; i0 += GA_THREADS * GA_RUN) {
            int t[GA_RUN];
            float v[GA_CH][GA_RUN];
#pragma unroll
            for (int q = 0; q < GA_RUN / 4; ++q) {
                const int i = i0 + 4 * q;
                const bool ok = i < total;
                const int4 tt = ok ? *(const int4 *)(id + i) : make_int4(0, 0, 0, 0);
                t[4 * q] = tt.x; t[4 * q + 1] = tt.y; t[4 * q + 2] = tt.z; t[4 * q + 3] = tt.w;
#pragma unroll
                for (int ch = 0; ch < GA_CH; ++ch) {
                    const float4 x = (ok && ch < nch) ? *(const float4 *)(g + (size_t)ch * total + i) : make_float4(0.f, 0.f, 0.f, 0.f);
                    v[ch][4 * q] = x.x; v[ch][4 * q + 1] = x.y; v[ch][4 * q + 2] = x.z; v[ch][4 * q + 3] = x.w;
                }
            }
            int cur = t[0];
            float run[GA_CH];
#pragma unroll
            for (int ch = 0; ch < GA_CH; ++ch) run[ch] = v[ch][0];
#pragma unroll
            for (int k = 1; k <= GA_RUN; ++k) {
                const bool flush = k == GA_RUN || t[k < GA_RUN ? k : 0] != cur;
                if (flush) {
#pragma unroll
                    for (int ch = 0; ch < GA_CH; ++ch)
                        if (ch < nch) __hip_atomic_fetch_add(acc + ch * n + cur, run[ch], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (k < GA_RUN) {
                        cur = t[k];
#pragma unroll
                        for (int ch = 0; ch < GA_CH; ++ch) run[ch] = v[ch][k];
                    }
                } else {
#pragma unroll
                    for (int ch = 0; ch < GA_CH; ++ch) run[ch] += v[ch][k < GA_RUN ? k : 0];
                }
            }
        }
    } else {
        for (int i = tid; i < total; i += GA_THREADS) {
            const int t = id[i];
            for (int ch = 0; ch < nch; ++ch)
                __hip_atomic_fetch_add(acc + ch * n + t, g[(size_t)ch * total + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __syncthreads();
    for (int i = tid; i < nch * n; i += GA_THREADS) gp[i] += acc[i];
}

// ---------------------------------------------------------------------------------------------------------------
// Long rows made of per-centre lists, deterministic and without atomics: pad-folded CSR gather (round 3).
// LDS fp32 atomics retire ~0.6 lanes per clock whatever the addresses (measured with a kernel that gave the S lanes of a
// wave one list, summed the padded slots across the lanes and issued one atomic per DISTINCT target: 15 distinct targets
// per instruction cost 24 clocks, 1331 us at BASELINE config 5 against 1081 us for the streamed kernel above), so every
// form that ADDS into LDS is bound by its atomic lanes.  This form only READS LDS.  Per call, an index kernel cuts a sample's row into tiles of 32768 entries and builds, per tile,
// (a) a 64-bit mask per list marking the slots that repeat the list's first entry (the reference's padding, three quarters
// of all entries on LiDAR-like clouds) and (b) a CSR index target -> positions over the remaining entries (counting sort
// in LDS with integer atomics, every target's short segment then sorted by position: a fixed order).  [A stable counting sort
// without atomics -- per-wave counts, the lanes of a 64-entry chunk that hold the same target matched with one ballot per key
// bit, as gpg_plan_kernel does for 8-bit keys -- was measured slower here: 100 against 76 us at config 5; with 12-bit keys and
// 512 chunks per tile the per-lane 64-bit mask arithmetic of the matching dominates.]  The gather kernel
// owns (sample, a few channels): per channel and tile it stages the 128 KB of the row in LDS -- while a list passes
// through the registers its padded slots are summed across the lanes (DPP) and folded into the list's first entry -- and
// then a thread walks the segments of its targets and accumulates in registers.  No atomics, every sum in a fixed order:
// bit-reproducible.  n <= 8192, lists of 16 / 32 / 64 slots.
// ---------------------------------------------------------------------------------------------------------------
constexpr int GC_THREADS = 512;
constexpr int GC_TILE = 32768;                         // entries per tile
constexpr int GC_MAX_N = 8192;
constexpr int GC_UN = 8;                               // chunks of indices in flight per wave (index kernel)
constexpr int GC_SHORT = 32;                           // longest segment a single thread sorts

template <int CTRL>
__device__ __forceinline__ float gc_dpp(float x)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}
// sum over the S lanes of a list (S = 16, 32, 64; lists are S-aligned in the wave), in a fixed order; valid in every lane
template <int S>
__device__ __forceinline__ float gc_list_sum(float x)
{
    x += gc_dpp<0xB1>(x);                              // quad_perm [1,0,3,2]
    x += gc_dpp<0x4E>(x);                              // quad_perm [2,3,0,1]
    x += gc_dpp<0x141>(x);                             // row_half_mirror
    x += gc_dpp<0x140>(x);                             // row_mirror: every lane of a 16-lane row holds the row's sum
    if (S >= 32) x += __shfl_xor(x, 16, 64);
    if (S >= 64) x += __shfl_xor(x, 32, 64);
    return x;
}

template <int S>
__global__ __launch_bounds__(GC_THREADS) void gpg_csr_index_kernel(
    int n, int total, int tiles, const int *__restrict__ idx, unsigned long long *__restrict__ padmask,
    unsigned short *__restrict__ off, unsigned short *__restrict__ pos)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char gc_smem[];
    unsigned *start = (unsigned *)gc_smem;             // [n + 1] segment starts
    unsigned *cur = start + (n + 1);                   // [n] counts, then cursors
    unsigned short *spos = (unsigned short *)(cur + n);   // [GC_TILE]
    __shared__ unsigned wsum[GC_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = blockIdx.x, bs = blockIdx.y;
    const int base = tile * GC_TILE, len = min(GC_TILE, total - base);
    const int *id = idx + (size_t)bs * total + base;
    unsigned long long *pm = padmask + ((size_t)bs * total + base) / S;
    for (int i = tid; i < n; i += GC_THREADS) cur[i] = 0u;
    __syncthreads();
    const int seg0 = lane & ~(S - 1);
    // (the indices of GC_UN chunks are requested together: with one load per step both passes were 64 exposed global round trips)
    for (int e0 = wave * 64; e0 < len; e0 += GC_THREADS * GC_UN) {
        int tv[GC_UN];
#pragma unroll
        for (int u = 0; u < GC_UN; ++u) { const int e = e0 + u * GC_THREADS + lane; tv[u] = e < len ? id[e] : -1; }
#pragma unroll
        for (int u = 0; u < GC_UN; ++u) {
            const int e = e0 + u * GC_THREADS + lane;
            const bool valid = e < len;
            const int t = tv[u];
            const int t0 = __shfl(t, seg0, 64);
            const bool pad = valid && lane != seg0 && t == t0;
            const unsigned long long m = __ballot(pad);
            if (valid && lane == seg0) pm[e / S] = S == 64 ? m : ((m >> seg0) & ((1ull << (S & 63)) - 1ull));
            if (valid && !pad) atomicAdd(&cur[t], 1u);
        }
    }
    __syncthreads();
    // exclusive scan of the counts (thread t owns the targets [t*nt, t*nt + nt))
    const int nt = (n + GC_THREADS - 1) / GC_THREADS;
    unsigned loc = 0;
    for (int q = 0; q < nt; ++q) { const int j = tid * nt + q; if (j < n) loc += cur[j]; }
    unsigned incl = loc;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const unsigned o = __shfl_up(incl, d, 64); if (lane >= d) incl += o; }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    unsigned wbase = 0;
    for (int w = 0; w < wave; ++w) wbase += wsum[w];
    unsigned run = wbase + incl - loc;
    for (int q = 0; q < nt; ++q) {
        const int j = tid * nt + q;
        if (j < n) { const unsigned cnt = cur[j]; start[j] = run; cur[j] = run; run += cnt; }
    }
    if (tid == GC_THREADS - 1) start[n] = run;
    __syncthreads();
    // placement (any order inside a target's segment), then every segment sorted by position
    for (int e0 = wave * 64; e0 < len; e0 += GC_THREADS * GC_UN) {
        int tv[GC_UN];
#pragma unroll
        for (int u = 0; u < GC_UN; ++u) { const int e = e0 + u * GC_THREADS + lane; tv[u] = e < len ? id[e] : -1; }
#pragma unroll
        for (int u = 0; u < GC_UN; ++u) {
            const int e = e0 + u * GC_THREADS + lane;
            const bool valid = e < len;
            const int t = tv[u];
            const int t0 = __shfl(t, seg0, 64);
            const bool pad = valid && lane != seg0 && t == t0;
            if (valid && !pad) spos[atomicAdd(&cur[t], 1u)] = (unsigned short)e;
        }
    }
    __syncthreads();
    // short segments (the rule on neighbour lists): insertion sort by their thread; long ones (arbitrary idx can put a whole
    // tile on one target) are queued and rank-sorted by the whole workgroup below
    __shared__ unsigned nlong;
    __shared__ unsigned short longlist[GC_TILE / (GC_SHORT + 1) + 1];
    if (tid == 0) nlong = 0u;
    __syncthreads();
    for (int j = tid; j < n; j += GC_THREADS) {
        const int b0 = (int)start[j], b1 = (int)start[j + 1];
        if (b1 - b0 > GC_SHORT) { longlist[atomicAdd(&nlong, 1u)] = (unsigned short)j; continue; }
        for (int i = b0 + 1; i < b1; ++i) {
            const unsigned short x = spos[i];
            int k = i - 1;
            while (k >= b0 && spos[k] > x) { spos[k + 1] = spos[k]; --k; }
            spos[k + 1] = x;
        }
    }
    __syncthreads();
    unsigned short *o = off + ((size_t)bs * tiles + tile) * (n + 1);
    for (int i = tid; i <= n; i += GC_THREADS) o[i] = (unsigned short)start[i];
    unsigned short *ps = pos + ((size_t)bs * tiles + tile) * GC_TILE;
    const int nnz = (int)start[n];
    for (int i = tid; i < nnz; i += GC_THREADS) ps[i] = spos[i];
    const int nl = (int)nlong;
    if (nl == 0) return;
    __syncthreads();                                                    // the copy above has landed (same addresses below)
    for (int li = 0; li < nl; ++li) {
        const int j = longlist[li];
        const int b0 = (int)start[j], b1 = (int)start[j + 1];
        for (int i = b0 + tid; i < b1; i += GC_THREADS) {               // positions are distinct: the rank is the place
            const unsigned short x = spos[i];
            int rank = 0;
            for (int k = b0; k < b1; ++k) rank += spos[k] < x ? 1 : 0;
            ps[b0 + rank] = x;
        }
    }
}

// LDS of the gather: the tile's floats | as many of its sorted positions as fit (the rest, if any, is read from global memory).
// [16384-entry tiles with two 512-thread workgroups per CU were measured slower, in round 3 (1633 against 1230-1270 us at config 5)
//  and again in round 4 with the index prefetch below (708 against 626 us at C = 64): twice the per-(target, tile) segment
//  overhead, and 8 targets per thread spill.]
constexpr int GCG_THREADS = 1024;
constexpr int GCG_EPT = GC_TILE / GCG_THREADS;         // entries per thread and tile (32 = 8 x float4)
constexpr size_t GCG_LDS = 160 * 1024 - 256;
constexpr int GCG_CAP = (int)((GCG_LDS - (size_t)GC_TILE * 4) / 2);

typedef unsigned gc_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t gc_rsrc(const void *q, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc((void *)q, 0, (int)bytes, 0x00020000);   // raw buffer: loads past `bytes` return 0
}
constexpr int GCG_PCH = (GCG_CAP / 8 + GCG_THREADS - 1) / GCG_THREADS;   // 16-byte chunks of staged positions per thread

// Every stream of the tile loop is a buffer load (descriptor in SGPRs + one lane offset + a scalar offset): with plain pointers the
// compiler keeps one 64-bit address per unrolled load (24 register pairs at 128 registers per thread: spills), and the range
// check of the descriptor replaces the tail selects.  A tile's segment bounds and its sorted positions are requested together
// with the tile, one tile ahead [before round 4 they were fetched behind the barriers: two exposed global latencies per tile].
template <int S, int NT>                               // NT targets per thread: n <= NT * GCG_THREADS
__global__ __launch_bounds__(GCG_THREADS, 4) void gpg_csr_gather_kernel(
    int c, int n, int total, int tiles, int ch_per_wg, const float *__restrict__ grad_out,
    const unsigned long long *__restrict__ padmask, const unsigned short *__restrict__ off, const unsigned short *__restrict__ pos,
    float *__restrict__ grad_points, int diag_arg)
{
#ifdef CMF_GC_EXPERIMENT
    const int diag = diag_arg;                                       // timing ablations (experiment builds only): 1 no walk, 2 no index staging, 4 no tile staging
#else
    constexpr int diag = 0;
#endif
    extern __shared__ __attribute__((aligned(16))) float gc_buf[];   // [GC_TILE]
    unsigned short *p_s = (unsigned short *)(gc_buf + GC_TILE);      // [GCG_CAP]
    const int tid = threadIdx.x, lane = tid & 63;
    const int bs = blockIdx.x;
    const int c0 = blockIdx.y * ch_per_wg, c1 = min(c, c0 + ch_per_wg);
    constexpr int LPL = S / 4;                                       // lanes per list: a lane holds 4 consecutive slots
    const int l0 = lane & ~(LPL - 1);
    // the byte of a list's 64-bit mask that holds this lane's four slots (slot 4 * (lane - l0) of the list)
    const __amdgpu_buffer_rsrc_t r_pm = gc_rsrc(padmask + (size_t)bs * total / S, (unsigned)(total / S) * 8u);
    const int vo_pm = (4 * tid / S) * 8 + ((lane - l0) >> 1);
    const int pm_sh = 4 * ((lane - l0) & 1);
    auto lds_barrier = [&]() {                                       // __syncthreads() would drain the prefetches (vmcnt(0)) as well
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    for (int ch = c0; ch < c1; ++ch) {
        const __amdgpu_buffer_rsrc_t r_g = gc_rsrc(grad_out + ((size_t)bs * c + ch) * total, (unsigned)total * 4u);
        float acc[NT], acc2[NT];
#pragma unroll
        for (int q = 0; q < NT; ++q) acc[q] = acc2[q] = 0.f;
        gc_u32x4 v[GCG_EPT / 4], psr[GCG_PCH];
        unsigned pmv[GCG_EPT / 4];
        unsigned ob[NT], oe[NT];                                     // segment bounds of this thread's targets in the requested tile
        int nnz_req = 0;
        // a thread stages the entries 4 * (tid + i * GCG_THREADS) .. + 3 of a tile (total % 4 == 0: a float4 is in or out)
        auto load_tile = [&](int tile) {
            const int base = tile * GC_TILE;
#pragma unroll
            for (int i = 0; i < GCG_EPT / 4; ++i) {
                v[i] = __builtin_amdgcn_raw_buffer_load_b128(r_g, 16 * tid, (base + 4 * i * GCG_THREADS) * 4, 0);
                pmv[i] = (unsigned)__builtin_amdgcn_raw_buffer_load_b8(r_pm, vo_pm, ((base + 4 * i * GCG_THREADS) / S) * 8, 0);
            }
            const unsigned short *o = off + ((size_t)bs * tiles + tile) * (n + 1);
            const __amdgpu_buffer_rsrc_t r_o = gc_rsrc(o, (unsigned)(n + 1) * 2u);
#pragma unroll
            for (int q = 0; q < NT; ++q) {
                ob[q] = (unsigned)__builtin_amdgcn_raw_buffer_load_b16(r_o, 2 * tid, q * GCG_THREADS * 2, 0);
                oe[q] = (unsigned)__builtin_amdgcn_raw_buffer_load_b16(r_o, 2 * tid + 2, q * GCG_THREADS * 2, 0);
            }
            nnz_req = o[n];
        };
        auto load_positions = [&](int tile, int nnz) {               // behind load_tile of the same tile, once its nnz is known
            const __amdgpu_buffer_rsrc_t r_p = gc_rsrc(pos + ((size_t)bs * tiles + tile) * GC_TILE, (unsigned)GCG_CAP * 2u);
#pragma unroll
            for (int k = 0; k < GCG_PCH; ++k)
                if (8 * (tid + k * GCG_THREADS) < nnz && !(diag & 2))
                    psr[k] = __builtin_amdgcn_raw_buffer_load_b128(r_p, 16 * tid, k * GCG_THREADS * 16, 0);
        };
        // [a rotated tile order per (sample, channel), which helps the gather kernel of the forward op, changes nothing here]
        load_tile(0);
        load_positions(0, nnz_req);
        for (int tile = 0; tile < tiles; ++tile) {
            const int len = min(GC_TILE, total - tile * GC_TILE);
            const unsigned short *ps = pos + ((size_t)bs * tiles + tile) * GC_TILE;
            const int nnz = nnz_req;
            int b0[NT], b1[NT];
#pragma unroll
            for (int q = 0; q < NT; ++q) {
                const bool in = tid + q * GCG_THREADS < n;
                b0[q] = in ? (int)ob[q] : 0; b1[q] = in ? (int)oe[q] : 0;
            }
            // fold every list's padded slots into its first entry (the lanes of a list are neighbours inside a DPP row), store
#pragma unroll
            for (int i = 0; i < GCG_EPT / 4; ++i) {
                const int e = 4 * (tid + i * GCG_THREADS);
                const unsigned bits = (pmv[i] >> pm_sh) & 0xFu;
                float4 w = __builtin_bit_cast(float4, v[i]);
                float part = ((bits & 1u) ? w.x : 0.f) + ((bits & 2u) ? w.y : 0.f) + ((bits & 4u) ? w.z : 0.f) + ((bits & 8u) ? w.w : 0.f);
                part += gc_dpp<0xB1>(part);                                         // 2 lanes
                part += gc_dpp<0x4E>(part);                                         // 4 lanes = a list of 16 slots
                if (S >= 32) part += gc_dpp<0x141>(part);                           // 8 lanes
                if (S >= 64) part += gc_dpp<0x140>(part);                           // 16 lanes = a list of 64 slots
                if (lane == l0) w.x += part;
                if (e < len && !(diag & 4)) *(float4 *)(gc_buf + e) = w;
                if (diag & 4) acc[0] += w.x + w.y + w.z + w.w;
            }
#pragma unroll
            for (int k = 0; k < GCG_PCH; ++k)
                if (8 * (tid + k * GCG_THREADS) < min(nnz, GCG_CAP) && !(diag & 2))
                    *(gc_u32x4 *)(p_s + 8 * (tid + k * GCG_THREADS)) = psr[k];
            lds_barrier();
            if (tile + 1 < tiles) { load_tile(tile + 1); load_positions(tile + 1, nnz_req); }   // in flight under the walk below
            int maxl = 0;
#pragma unroll
            for (int q = 0; q < NT; ++q) maxl = max(maxl, b1[q] - b0[q]);
            if (diag & 1) maxl = 0;
            // two interleaved chains per target (even / odd entries of its segment; four were measured slower: registers): the walk is bound by the latency of
            // its dependent LDS reads (position, then value), not by their number; the order of every sum stays fixed
            if (nnz <= GCG_CAP) {
                for (int sI = 0; sI < maxl; sI += 2) {
#pragma unroll
                    for (int q = 0; q < NT; ++q) {
                        const int k = b0[q] + sI;
                        const float x0 = k < b1[q] ? gc_buf[p_s[k]] : 0.f;
                        const float x1 = k + 1 < b1[q] ? gc_buf[p_s[k + 1]] : 0.f;
                        acc[q] += x0; acc2[q] += x1;
                    }
                }
            } else {
                for (int sI = 0; sI < maxl; ++sI) {
#pragma unroll
                    for (int q = 0; q < NT; ++q) {
                        const int k = b0[q] + sI;
                        if (k < b1[q]) { const float x = gc_buf[k < GCG_CAP ? p_s[k] : ps[k]]; if (sI & 1) acc2[q] += x; else acc[q] += x; }
                    }
                }
            }
            lds_barrier();
        }
        float *gp = grad_points + ((size_t)bs * c + ch) * n;
#pragma unroll
        for (int q = 0; q < NT; ++q) { const int j = tid + q * GCG_THREADS; if (j < n) gp[j] += acc[q] + acc2[q]; }
    }
}

template <int S>
static int launch_csr(int b, int c, int n, int total, const float *grad_out, const int *idx, float *grad_points, hipStream_t st)
{
    const int tiles = cmf_divup(total, GC_TILE);
    const size_t n_mask = (size_t)b * total / S * sizeof(unsigned long long);
    const size_t n_off = ((size_t)b * tiles * (n + 1) * sizeof(unsigned short) + 15) / 16 * 16;
    const size_t n_pos = (size_t)b * tiles * GC_TILE * sizeof(unsigned short);
    const CmfScratchLease lease = cmf_stream_scratch(st, 0, n_mask + n_off + n_pos);
    if (!lease.ptr) return (int)hipErrorOutOfMemory;
    unsigned long long *padmask = (unsigned long long *)lease.ptr;
    unsigned short *off = (unsigned short *)((char *)lease.ptr + n_mask);
    unsigned short *pos = (unsigned short *)((char *)lease.ptr + n_mask + n_off);
    static CmfPerDevice attr_set;
    int attr_dev;
    if (attr_set.need(attr_dev)) {
        (void)hipFuncSetAttribute((const void *)gpg_csr_index_kernel<S>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        (void)hipFuncSetAttribute((const void *)gpg_csr_gather_kernel<S, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)GCG_LDS);
        (void)hipFuncSetAttribute((const void *)gpg_csr_gather_kernel<S, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)GCG_LDS);
        attr_set.done(attr_dev);
    }
    const size_t lds_i = (size_t)(2 * n + 1) * sizeof(unsigned) + (size_t)GC_TILE * sizeof(unsigned short);
    hipLaunchKernelGGL(gpg_csr_index_kernel<S>, dim3(tiles, b), dim3(GC_THREADS), lds_i, st, n, total, tiles, idx, padmask, off, pos);
#ifdef CMF_GC_EXPERIMENT
    const int gdiag = getenv("CMF_GC_DIAG") ? atoi(getenv("CMF_GC_DIAG")) : 0;   // timing ablations: 1 no walk, 2 no index staging, 4 no tile staging
#else
    const int gdiag = 0;
#endif
    int ch_per_wg = 8;                                  // >= 2 workgroups per CU over the launch
    while (ch_per_wg > 1 && (long long)b * cmf_divup(c, ch_per_wg) < 512) ch_per_wg /= 2;
    const dim3 grid(b, cmf_divup(c, ch_per_wg));
    if (n <= 4 * GCG_THREADS)
        hipLaunchKernelGGL((gpg_csr_gather_kernel<S, 4>), grid, dim3(GCG_THREADS), GCG_LDS, st, c, n, total, tiles, ch_per_wg, grad_out,
                           padmask, off, pos, grad_points, gdiag);
    else
        hipLaunchKernelGGL((gpg_csr_gather_kernel<S, 8>), grid, dim3(GCG_THREADS), GCG_LDS, st, c, n, total, tiles, ch_per_wg, grad_out,
                           padmask, off, pos, grad_points, gdiag);
    return cmf_launch_status();
}

// ---------------------------------------------------------------------------------------------------------------
// Long rows (N = 4096 x K = 64: 262 144 entries = 1 MB per (sample, channel) row), deterministic.
// The row is cut into TILES of 8192 consecutive entries and each tile gets its own small inverse index, built once per
// call and shared by all channels (gpg_tile_index_kernel): the tile's entries sorted by target with a stable 4-bit
// LSD radix sort in LDS (16-bit keys; equal targets keep position order), the sorted POSITIONS (uint16), a bit per
// sorted entry marking the end of a target's run, the target of every run, and for every 16-entry chunk the chunk in
// which the run that is open at its start began.  The scatter kernel (gpg_tiled_kernel) then works like the balanced
// kernel above, tile after tile: a workgroup owns (sample, 4 channels) and keeps acc[4][n] in LDS; per tile and channel
// it stages the 8192 floats of the row (coalesced 16-byte loads, the next ones already in flight), every thread gathers
// its 16 sorted entries from LDS, forms the segmented sums, runs that cross chunks are stitched from the chunk tails in
// fixed order, and the one thread that holds a run's end adds the run's sum to acc[target] -- a target has exactly one
// run per tile, so no atomics, and the order of every addition is fixed: bit-reproducible.  It is the default only where
// the LDS-atomic kernel below does not fit (n > 8192) and otherwise opt-in: see the dispatch in cmf_group_points_grad.
// ---------------------------------------------------------------------------------------------------------------
constexpr int GT_THREADS = 512;
constexpr int GT_E = 16;                               // sorted entries per thread
constexpr int GT_TILE = GT_THREADS * GT_E;             // 8192 entries
constexpr int GT_CH = 4;                               // channels per workgroup
// per-tile index record, in 16-bit words: pos[TILE] | seg_target[TILE] | mask[THREADS] | rank0[THREADS] | ostart[THREADS] | nseg, pad
constexpr int GT_REC = 2 * GT_TILE + 3 * GT_THREADS + 16;

__device__ __forceinline__ int gt_block_scan(int v, int *wsum, int &block_total)      // inclusive, 512 threads (8 waves)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int u = __shfl_up(v, d, 64);
        if (lane >= d) v += u;
    }
    __syncthreads();
    if (lane == 63) wsum[wave] = v;
    __syncthreads();
    int add = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < GT_THREADS / 64; ++w) {
        const int x = wsum[w];
        if (w < wave) add += x;
        tot += x;
    }
    block_total = tot;
    return v + add;
}

__global__ __launch_bounds__(GT_THREADS) void gpg_tile_index_kernel(int total, int tiles, int key_bits,
                                                                   const int *__restrict__ idx, unsigned short *__restrict__ rec_all)
{
    __shared__ unsigned short skey[GT_TILE + 1], spos[GT_TILE];
    __shared__ unsigned short cnt[16][GT_THREADS + 2];
    __shared__ int scan[GT_THREADS / 64];
    __shared__ unsigned short runstart[GT_THREADS];
    const int tid = threadIdx.x, tile = blockIdx.x, bs = blockIdx.y;
    const int *ix = idx + (size_t)bs * total + (size_t)tile * GT_TILE;
    const int len = min(GT_TILE, total - tile * GT_TILE);
    unsigned short *rec = rec_all + ((size_t)bs * tiles + tile) * GT_REC;
    unsigned key[GT_E], pos[GT_E];
#pragma unroll
    for (int i = 0; i < GT_E; ++i) {
        const int e = tid * GT_E + i;
        key[i] = e < len ? (unsigned)ix[e] : 0xFFFFu;            // entries past the row's end sort last
        pos[i] = e < len ? (unsigned)e : (unsigned)GT_TILE;      // ... and read the zero slot behind the staged row
    }
    for (int shift = 0; shift < key_bits; shift += 4) {          // stable LSD radix sort, 4 bits per pass
#pragma unroll
        for (int d = 0; d < 16; ++d) cnt[d][tid] = 0;
#pragma unroll
        for (int i = 0; i < GT_E; ++i) ++cnt[(key[i] >> shift) & 15][tid];       // own column: no conflicts between threads
        __syncthreads();
        // exclusive scan over the counters in (digit, thread) order: thread t takes the 16 consecutive counters t*16 ..
        int loc[16], run = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) { const int f = tid * 16 + q; loc[q] = run; run += cnt[f / GT_THREADS][f % GT_THREADS]; }
        int tot;
        const int base = gt_block_scan(run, scan, tot) - run;
#pragma unroll
        for (int q = 0; q < 16; ++q) { const int f = tid * 16 + q; cnt[f / GT_THREADS][f % GT_THREADS] = (unsigned short)(base + loc[q]); }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < GT_E; ++i) {
            const int d = (key[i] >> shift) & 15;
            const int dst = cnt[d][tid]++;
            skey[dst] = (unsigned short)key[i]; spos[dst] = (unsigned short)pos[i];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < GT_E; ++i) { key[i] = skey[tid * GT_E + i]; pos[i] = spos[tid * GT_E + i]; }
        __syncthreads();
    }
    // run ends: sorted entry i ends its target's run when the next sorted key differs (padding keys end nothing)
    if (tid == 0) skey[GT_TILE] = 0xFFFFu;
#pragma unroll
    for (int i = 0; i < GT_E; ++i) skey[tid * GT_E + i] = (unsigned short)key[i];
    __syncthreads();
    unsigned mask = 0;
    bool inner = false;                                            // a run starts inside this chunk
#pragma unroll
    for (int i = 0; i < GT_E; ++i) {
        const unsigned nxt = skey[tid * GT_E + i + 1];
        if (key[i] != 0xFFFFu && key[i] != nxt) mask |= 1u << i;
        if (i + 1 < GT_E && key[i] != key[i + 1]) inner = true;
    }
    int tot;
    const int rank0 = gt_block_scan(__popc(mask), scan, tot) - __popc(mask);
    // runstart[t]: the chunk in which the run holding chunk t's LAST entry began (serial only through all-equal chunks)
    runstart[tid] = (unsigned short)(inner ? tid : 0xFFFF);
    __syncthreads();
    if (tid == 0)
        for (int t = 0; t < GT_THREADS; ++t)
            if (runstart[t] == 0xFFFF)
                runstart[t] = (unsigned short)((t > 0 && skey[t * GT_E - 1] == skey[t * GT_E]) ? runstart[t - 1] : t);
    __syncthreads();
    const int ostart = (tid > 0 && skey[tid * GT_E - 1] == (unsigned short)key[0]) ? runstart[tid - 1] : tid;
    unsigned short *r_pos = rec, *r_seg = rec + GT_TILE, *r_mask = rec + 2 * GT_TILE;
    unsigned short *r_rank = r_mask + GT_THREADS, *r_ost = r_rank + GT_THREADS;
    int rk = rank0;
#pragma unroll
    for (int i = 0; i < GT_E; ++i) {
        r_pos[tid * GT_E + i] = (unsigned short)pos[i];
        if ((mask >> i) & 1u) r_seg[rk++] = (unsigned short)key[i];
    }
    r_mask[tid] = (unsigned short)mask; r_rank[tid] = (unsigned short)rank0; r_ost[tid] = (unsigned short)ostart;
    if (tid == 0) r_ost[GT_THREADS] = (unsigned short)tot;
}

__global__ __launch_bounds__(GT_THREADS) void gpg_tiled_kernel(int c, int n, int total, int tiles, int nch_wg,
                                                              const float *__restrict__ grad_out,
                                                              const unsigned short *__restrict__ rec_all, float *__restrict__ grad_points)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *acc = sm;                                        // [nch_wg][n]
    float *row = acc + (size_t)nch_wg * n;                  // [TILE + 4]: staged row tile + a zero slot
    float *tails = row + GT_TILE + 4;                       // [THREADS]
    unsigned short *seg = reinterpret_cast<unsigned short *>(tails + GT_THREADS);      // [TILE] targets of this tile's runs
    const int tid = threadIdx.x, bs = blockIdx.x;
    const int c0 = blockIdx.y * nch_wg, nch = min(nch_wg, c - c0);
    const float *g = grad_out + ((size_t)bs * c + c0) * total;
    float *gp = grad_points + ((size_t)bs * c + c0) * n;
    for (int i = tid; i < nch_wg * n; i += GT_THREADS) acc[i] = 0.f;
    if (tid == 0) row[GT_TILE] = 0.f;
    const bool vec = (total & 3) == 0 && ((uintptr_t)grad_out & 15) == 0;
    constexpr int V = GT_E / 4;
    auto fetch = [&](int tile, int ch, float4 (&r)[V]) {    // this thread's part of the tile's row for channel ch
        const float *src = g + (size_t)ch * total + (size_t)tile * GT_TILE;
        const int len = min(GT_TILE, total - tile * GT_TILE);
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const int i = (v * GT_THREADS + tid) * 4;
            if (vec) r[v] = i < len ? *(const float4 *)(src + i) : make_float4(0.f, 0.f, 0.f, 0.f);
            else {
                r[v].x = i < len ? src[i] : 0.f;         r[v].y = i + 1 < len ? src[i + 1] : 0.f;
                r[v].z = i + 2 < len ? src[i + 2] : 0.f; r[v].w = i + 3 < len ? src[i + 3] : 0.f;
            }
        }
    };
    float4 nxt[V];
    fetch(0, 0, nxt);
    for (int tile = 0; tile < tiles; ++tile) {
        const unsigned short *rec = rec_all + ((size_t)bs * tiles + tile) * GT_REC;
        unsigned pos2[GT_E / 2];
#pragma unroll
        for (int k = 0; k < GT_E / 2; ++k) pos2[k] = reinterpret_cast<const unsigned *>(rec)[tid * (GT_E / 2) + k];
        const unsigned mask = rec[2 * GT_TILE + tid];
        const int rank0 = rec[2 * GT_TILE + GT_THREADS + tid], ostart = rec[2 * GT_TILE + 2 * GT_THREADS + tid];
        const int nseg = rec[2 * GT_TILE + 3 * GT_THREADS];
        __syncthreads();                                    // previous tile's runs are finished with `seg`
        for (int i = tid; i < (nseg + 1) / 2; i += GT_THREADS)
            reinterpret_cast<unsigned *>(seg)[i] = reinterpret_cast<const unsigned *>(rec + GT_TILE)[i];
        for (int ch = 0; ch < nch; ++ch) {
            // stage the row tile (every read of the previous one is done: barrier at the end of the last iteration)
#pragma unroll
            for (int v = 0; v < V; ++v) *(float4 *)(row + (v * GT_THREADS + tid) * 4) = nxt[v];
            const bool last = (ch + 1 == nch);
            if (!(last && tile + 1 == tiles)) fetch(last ? tile + 1 : tile, last ? 0 : ch + 1, nxt);      // in flight during the reduction
            __syncthreads();
            float val[GT_E];
#pragma unroll
            for (int k = 0; k < GT_E; ++k) {
                const unsigned pp = pos2[k / 2];
                val[k] = row[(k & 1) ? (pp >> 16) : (pp & 0xffffu)];
            }
            // segmented sums of this chunk: `head` = up to the first run end, `run` = after the last one
            float run = 0.f, head = 0.f;
            bool first = true;
            float ends[GT_E];                               // sum of the run ending at sorted entry k (valid where mask bit k)
#pragma unroll
            for (int k = 0; k < GT_E; ++k) {
                run += val[k];
                ends[k] = run;
                if ((mask >> k) & 1u) { if (first) { head = run; first = false; } run = 0.f; }
            }
            tails[tid] = run;                               // (a chunk without any run end: its whole sum)
            __syncthreads();
            // every run end adds its run to acc: the first one of a chunk collects the tails of the chunks the run crossed
            float *a = acc + (size_t)ch * n;
            int rk = rank0;
            bool f2 = true;
#pragma unroll
            for (int k = 0; k < GT_E; ++k)
                if ((mask >> k) & 1u) {
                    float sum = ends[k];
                    if (f2) {
                        float carry = 0.f;
                        for (int q = ostart; q < tid; ++q) carry += tails[q];
                        sum = carry + head;
                        f2 = false;
                    }
                    a[seg[rk++]] += sum;
                }
            __syncthreads();                                // tails / row reusable
        }
    }
    for (int i = tid; i < nch * n; i += GT_THREADS) gp[i] += acc[i];
}

// generic fall-back (any n): thread per target over the inverse index, row read through the caches
__global__ __launch_bounds__(GG_THREADS) void group_points_grad_generic_kernel(
    int c, int n, int total, const float *__restrict__ grad_out, const int *__restrict__ offsets,
    const int *__restrict__ inv, float *__restrict__ grad_points)
{
    const int bs = blockIdx.x, ch = blockIdx.y;
    const int *off = offsets + (size_t)bs * (n + 1);
    const int *lst = inv + (size_t)bs * total;
    const float *src = grad_out + ((size_t)bs * c + ch) * total;
    float *gp = grad_points + ((size_t)bs * c + ch) * n;
    for (int j = threadIdx.x; j < n; j += GG_THREADS) {
        float a = 0.f;
        for (int t = off[j]; t < off[j + 1]; ++t) a += src[lst[t]];
        gp[j] += a;
    }
}

template <int E>
static void launch_bal(dim3 grid, size_t lds, hipStream_t st, int c, int n, int total, int ch_per_wg, const float *grad_out,
                       const int *offsets, const int *inv, float *grad_points)
{
    static CmfPerDevice attr_set;
    int attr_dev;
    if (attr_set.need(attr_dev)) {
        (void)hipFuncSetAttribute((const void *)group_points_grad_bal_kernel<E>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  ((GG_THREADS * E + 4) + 3 * GG_MAX_N_BAL + GG_THREADS + 16) * 4);
        attr_set.done(attr_dev);
    }
    hipLaunchKernelGGL(group_points_grad_bal_kernel<E>, grid, dim3(GG_THREADS), lds, st, c, n, total, ch_per_wg, grad_out, offsets,
                       inv, grad_points);
}

#include <cstdlib>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>
namespace {
struct ScratchRetired { void *p; hipEvent_t done; size_t cap; };
struct ScratchEntry { void *p = nullptr; size_t cap = 0; std::mutex in_use; std::vector<ScratchRetired> retired; };
std::mutex g_scratch_mu;
std::map<std::tuple<int, hipStream_t, int>, ScratchEntry> *g_scratch_table =
    new std::map<std::tuple<int, hipStream_t, int>, ScratchEntry>();   // leaked on purpose (runtime teardown order)
}  // namespace

extern "C" long long cmf_mem_stats(void)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
    long long total = 0;
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    for (auto &kv : *g_scratch_table) {
        if (std::get<0>(kv.first) != dev) continue;
        std::lock_guard<std::mutex> use(kv.second.in_use);
        total += (long long)kv.second.cap;
        for (const auto &r : kv.second.retired) total += (long long)r.cap;
    }
    return total;
}

CmfScratchLease cmf_stream_scratch(hipStream_t stream, int slot, size_t bytes)
{
    typedef ScratchEntry Entry;
    std::mutex &mu = g_scratch_mu;
    auto *table = g_scratch_table;
    CmfScratchLease lease;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return lease;
    Entry *e;
    {
        std::lock_guard<std::mutex> lock(mu);
        e = &(*table)[std::make_tuple(dev, stream, slot)];              // map nodes do not move
    }
    lease.hold = std::unique_lock<std::mutex>(e->in_use);
    // buffers retired by an earlier growth: free the ones whose last possible user (work queued before the retirement) is done
    for (size_t i = 0; i < e->retired.size();) {
        if (hipEventQuery(e->retired[i].done) == hipSuccess) {
            (void)hipFree(e->retired[i].p);
            (void)hipEventDestroy(e->retired[i].done);
            e->retired[i] = e->retired.back();
            e->retired.pop_back();
        } else { (void)hipGetLastError(); ++i; }
    }
    if (e->cap < bytes) {
        // the old buffer may still be in use by queued work of this stream: it is retired behind an event, not freed here
        size_t want = bytes > e->cap + e->cap / 2 ? bytes : e->cap + e->cap / 2;
        want = (want + ((size_t)1 << 20) - 1) >> 20 << 20;
        void *p = nullptr;
        if (hipMalloc(&p, want) != hipSuccess) { (void)hipGetLastError(); lease.hold.unlock(); return lease; }
        if (e->p) {
            hipEvent_t ev = nullptr;
            if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess && hipEventRecord(ev, stream) == hipSuccess) e->retired.push_back({e->p, ev, e->cap});
            else { (void)hipGetLastError(); if (ev) (void)hipEventDestroy(ev); }       // could not fence it: keep it allocated (the old behaviour)
        }
        e->p = p; e->cap = want;
    }
    lease.ptr = e->p;
    return lease;
}

extern "C" int cmf_group_points_grad(int b, int c, int n, int npoints, int nsample,
                                     const float *grad_out, const int *idx, float *grad_points, void *stream)
{
    CMF_CHECK_ARG(b >= 0 && c >= 0 && n >= 0 && npoints >= 0 && nsample >= 0);
    const long long total = (long long)npoints * nsample;
    if (b == 0 || c == 0 || total == 0) return 0;
    CMF_CHECK_ARG(grad_out && idx && grad_points && n > 0 && total < (1LL << 31));
    hipStream_t st = (hipStream_t)stream;
    const bool balanced = total <= GG_THREADS * GG_MAX_E && n <= GG_MAX_N_BAL;
    // Long rows: the deterministic tiled kernel is opt-in (CMF_GROUP_GRAD_DETERMINISTIC=1).  Measured at BASELINE config 5,
    // (32,4096,64,64 / 128): 1747 / 3330 us against 1079 / 2148 us for the LDS-atomic kernel below -- with ~2 entries per
    // target and tile nearly every second sorted entry ends a run, so the per-run work (target lookup + accumulate) and
    // the two barriers per (tile, channel) at one workgroup per CU cost more than the same-address conflicts they remove.
    static const bool deterministic = getenv("CMF_GROUP_GRAD_DETERMINISTIC") && getenv("CMF_GROUP_GRAD_DETERMINISTIC")[0] == '1';
    const bool stream_fits = (size_t)GA_CH * n * sizeof(float) <= 128 * 1024;
    // per-centre lists of 16 / 32 / 64 slots: the pad-folded CSR gather -- deterministic, no atomics, and faster than the
    // LDS-atomic kernel (BASELINE config 5, C = 64 / 128: 859 / 1622 us against 1081 / 2148 us; the tiled deterministic
    // kernel: 1747 / 3330 us).  CMF_GROUP_GRAD_CSR=0 falls through to the older kernels (A/B).
    static const bool use_csr = !(getenv("CMF_GROUP_GRAD_CSR") && getenv("CMF_GROUP_GRAD_CSR")[0] == '0');
    if (!balanced && use_csr && (nsample == 16 || nsample == 32 || nsample == 64) && n <= GC_MAX_N && total <= (1 << 30) &&
        (((uintptr_t)grad_out) & 15) == 0) {
        if (nsample == 64) return launch_csr<64>(b, c, n, (int)total, grad_out, idx, grad_points, st);
        if (nsample == 32) return launch_csr<32>(b, c, n, (int)total, grad_out, idx, grad_points, st);
        return launch_csr<16>(b, c, n, (int)total, grad_out, idx, grad_points, st);
    }
    if (!balanced && n <= 16384 && (deterministic || !stream_fits)) {
        // per-tile inverse index (library scratch) + tiled deterministic scatter
        const int tiles = cmf_divup(total, GT_TILE);
        int key_bits = 4;
        while ((1 << key_bits) < n) key_bits += 4;
        const CmfScratchLease lease = cmf_stream_scratch(st, 0, (size_t)b * tiles * GT_REC * sizeof(unsigned short));
        unsigned short *rec = (unsigned short *)lease.ptr;
        if (!rec) return (int)hipErrorOutOfMemory;
        hipLaunchKernelGGL(gpg_tile_index_kernel, dim3(tiles, b), dim3(GT_THREADS), 0, st, (int)total, tiles, key_bits, idx, rec);
        int nch_wg = GT_CH;
        while (nch_wg > 1 && (size_t)nch_wg * n * sizeof(float) > 64 * 1024) nch_wg /= 2;
        const size_t lds = ((size_t)nch_wg * n + GT_TILE + 4 + GT_THREADS) * sizeof(float) + GT_TILE * sizeof(unsigned short);
        static CmfPerDevice attr_t;
        int attr_dev;
        if (attr_t.need(attr_dev)) {
            (void)hipFuncSetAttribute((const void *)gpg_tiled_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
            attr_t.done(attr_dev);
        }
        hipLaunchKernelGGL(gpg_tiled_kernel, dim3(b, cmf_divup(c, nch_wg)), dim3(GT_THREADS), lds, st, c, n, (int)total, tiles, nch_wg,
                           grad_out, rec, grad_points);
        return cmf_launch_status();
    }
    if (!balanced && stream_fits) {
        static CmfPerDevice attr_set;
        int attr_dev;
        if (attr_set.need(attr_dev)) {
            (void)hipFuncSetAttribute((const void *)group_points_grad_stream_kernel,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
            attr_set.done(attr_dev);
        }
        hipLaunchKernelGGL(group_points_grad_stream_kernel, dim3(b, cmf_divup(c, GA_CH)), dim3(GA_THREADS),
                           (size_t)GA_CH * n * sizeof(float), st, c, n, (int)total, grad_out, idx, grad_points);
        return cmf_launch_status();
    }
    // rows of the model's sizes: one plan kernel + the plan form of the balanced kernel (CMF_GROUP_GRAD_PLAN=0: the inverse
    // index + balanced kernel below, diagnostics)
    static const bool use_plan = !(getenv("CMF_GROUP_GRAD_PLAN") && getenv("CMF_GROUP_GRAD_PLAN")[0] == '0');
    if (use_plan && balanced && n <= GPL_MAX_N) {
        int ch_per_wg = GG_CH;
        while (ch_per_wg > 2 && (long long)b * cmf_divup(c, ch_per_wg) < 1024) ch_per_wg /= 2;
        const int e_need = (int)cmf_divup(total, GG_THREADS);
        const int E = e_need <= 4 ? 4 : e_need <= 8 ? 8 : e_need <= 16 ? 16 : 32;
        const CmfScratchLease lease = cmf_stream_scratch(st, 0, (size_t)b * gpg_plan_words(E, n) * sizeof(unsigned));
        unsigned *plan = (unsigned *)lease.ptr;
        if (!plan) return (int)hipErrorOutOfMemory;
        if (E == 4) return launch_plan<4>(b, c, n, (int)total, ch_per_wg, grad_out, idx, plan, grad_points, st);
        if (E == 8) return launch_plan<8>(b, c, n, (int)total, ch_per_wg, grad_out, idx, plan, grad_points, st);
        if (E == 16) return launch_plan<16>(b, c, n, (int)total, ch_per_wg, grad_out, idx, plan, grad_points, st);
        return launch_plan<32>(b, c, n, (int)total, ch_per_wg, grad_out, idx, plan, grad_points, st);
    }
    // per-stream library scratch for the inverse index (cmf_common.h)
    const size_t n_off = (size_t)b * (n + 1), n_inv = (size_t)b * total;
    const CmfScratchLease lease = cmf_stream_scratch(st, 0, (n_off + n_inv) * sizeof(int));
    int *scratch = (int *)lease.ptr;
    if (!scratch) return (int)hipErrorOutOfMemory;
    int *offsets = scratch, *inv = scratch + n_off;
    int err = cmf_build_inverse_rows(b, n, npoints, nsample, idx, offsets, inv, stream, (int)total);
    if (!err) {
        if (balanced) {
            // channels per workgroup: the per-sample index set-up costs about two channel rows, so more channels amortise
            // it -- but (b, c / 16) workgroups leave the chip empty for narrow features (b = 64, c = 64: 256 workgroups,
            // one per CU, 145 us for 140 MB); aim at >= 1024 workgroups, at least 4 channels each
            int ch_per_wg = GG_CH;
            while (ch_per_wg > 4 && (long long)b * cmf_divup(c, ch_per_wg) < 1024) ch_per_wg /= 2;
            const dim3 grid(b, cmf_divup(c, ch_per_wg));
            const int e_need = (int)cmf_divup(total, GG_THREADS);
            const int E = e_need <= 4 ? 4 : e_need <= 8 ? 8 : e_need <= 16 ? 16 : 32;
            const size_t lds = (size_t)((GG_THREADS * E + 4) + 3 * n + GG_THREADS + 16) * 4;
            if (E == 4) launch_bal<4>(grid, lds, st, c, n, (int)total, ch_per_wg, grad_out, offsets, inv, grad_points);
            else if (E == 8) launch_bal<8>(grid, lds, st, c, n, (int)total, ch_per_wg, grad_out, offsets, inv, grad_points);
            else if (E == 16) launch_bal<16>(grid, lds, st, c, n, (int)total, ch_per_wg, grad_out, offsets, inv, grad_points);
            else launch_bal<32>(grid, lds, st, c, n, (int)total, ch_per_wg, grad_out, offsets, inv, grad_points);
        } else {
            hipLaunchKernelGGL(group_points_grad_generic_kernel, dim3(b, c), dim3(GG_THREADS), 0, st,
                               c, n, (int)total, grad_out, offsets, inv, grad_points);
        }
        err = cmf_launch_status();
    }
    return err;
}
