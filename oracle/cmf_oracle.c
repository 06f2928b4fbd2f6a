/*
 * oracle/cmf_oracle.c -- TEST INFRASTRUCTURE ONLY (never imported by the product path).
 *
 * Plain-C, single-threaded CPU restatement of the three native kernels the CMFlow hot
 * path uses, plus the canonical arithmetic for the torch-level kNN.  Each function cites
 * the reference lines it follows (paths relative to the upstream repo Toytiny/CMFlow).
 *
 * Canonical floating-point arithmetic (the contract the HIP kernels must reproduce
 * bit-for-bit; see DESIGN.md "Canonical arithmetic"):
 *   - every fp32 operation is individually rounded (no FMA contraction); this file must
 *     be compiled with -ffp-contract=off and without -ffast-math / -march=native.
 *   - ball query   d2 = ((dx*dx) + (dy*dy)) + (dz*dz),  dx = new_x - x, hit iff d2 < r*r
 *   - kNN distance d  = max(((-2*dot) + |src|^2) + |dst|^2, 0)
 *         dot   = fmaf(sz, dz, fmaf(sy, dy, sx*dx))   <- the k-ordered FMA chain of a GEMM
 *                 (bit-equal to torch-CPU/MKL matmul on these shapes: tests/test_oracle.py)
 *         |p|^2 = ((px*px) + (py*py)) + (pz*pz)
 *     selection: k smallest, ties broken by lowest point index, output ascending (d, index).
 *
 * Parity status: the reference ships no tests or golden vectors for these kernels
 * ("parity unpinned" by upstream); the oracle is pinned instead against outputs of the
 * reference's own Python modules imported in the build container (tests/golden/).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#if defined(__FAST_MATH__)
#error "cmf_oracle.c must not be built with -ffast-math"
#endif

/* lib/src/ball_query_gpu.cu:9-45 -- one scan per centre, strict '<', first hit pre-fills
 * every slot, hits written in scan order, stop at nsample.  idx is NOT touched for an
 * empty ball (the Python caller pre-zeroes it, lib/pointnet2_utils.py:246). */
void orc_ball_query(int b, int n, int m, float radius, int nsample,
                    const float *new_xyz, const float *xyz, int *idx)
{
    const float radius2 = radius * radius;
    for (int bs = 0; bs < b; ++bs) {
        const float *pts = xyz + (size_t)bs * n * 3;
        for (int pt = 0; pt < m; ++pt) {
            const float *c = new_xyz + ((size_t)bs * m + pt) * 3;
            int *out = idx + ((size_t)bs * m + pt) * nsample;
            const float cx = c[0], cy = c[1], cz = c[2];
            int cnt = 0;
            for (int k = 0; k < n; ++k) {
                const float dx = cx - pts[k * 3 + 0];
                const float dy = cy - pts[k * 3 + 1];
                const float dz = cz - pts[k * 3 + 2];
                const float xx = dx * dx;
                const float yy = dy * dy;
                const float zz = dz * dz;
                const float s = xx + yy;
                const float d2 = s + zz;
                if (d2 < radius2) {
                    if (cnt == 0)
                        for (int l = 0; l < nsample; ++l) out[l] = k;
                    out[cnt] = k;
                    ++cnt;
                    if (cnt >= nsample) break;
                }
            }
        }
    }
}

/* lib/src/group_points_gpu.cu:47-66 -- out[b,c,p,s] = points[b,c,idx[b,p,s]] */
void orc_group_points(int b, int c, int n, int npoints, int nsample,
                      const float *points, const int *idx, float *out)
{
    for (int bs = 0; bs < b; ++bs)
        for (int ch = 0; ch < c; ++ch) {
            const float *row = points + ((size_t)bs * c + ch) * n;
            float *o = out + ((size_t)bs * c + ch) * npoints * nsample;
            const int *ix = idx + (size_t)bs * npoints * nsample;
            for (int j = 0; j < npoints * nsample; ++j) o[j] = row[ix[j]];
        }
}

/* lib/src/group_points_gpu.cu:8-25 -- grad_points[b,c,idx[b,p,s]] += grad_out[b,c,p,s].
 * The reference uses atomicAdd (order undefined); the oracle sums in (p,s) scan order.
 * grad_points must be zero-initialised by the caller (lib/pointnet2_utils.py:218). */
void orc_group_points_grad(int b, int c, int n, int npoints, int nsample,
                           const float *grad_out, const int *idx, float *grad_points)
{
    for (int bs = 0; bs < b; ++bs)
        for (int ch = 0; ch < c; ++ch) {
            float *row = grad_points + ((size_t)bs * c + ch) * n;
            const float *g = grad_out + ((size_t)bs * c + ch) * npoints * nsample;
            const int *ix = idx + (size_t)bs * npoints * nsample;
            for (int j = 0; j < npoints * nsample; ++j) row[ix[j]] += g[j];
        }
}

static inline float sq3(const float *p)
{
    const float xx = p[0] * p[0];
    const float yy = p[1] * p[1];
    const float zz = p[2] * p[2];
    const float s = xx + yy;
    return s + zz;
}

/* utils/model_utils/radarflow_util.py:8-30 (square_distance): -2*src.dst^T, += |src|^2,
 * += |dst|^2, clamp >= 0.  src (B,N,3), dst (B,M,3) -> dist (B,N,M). */
void orc_square_distance(int b, int n, int m, const float *src, const float *dst, float *dist)
{
    for (int bs = 0; bs < b; ++bs)
        for (int i = 0; i < n; ++i) {
            const float *s = src + ((size_t)bs * n + i) * 3;
            const float ss = sq3(s);
            for (int j = 0; j < m; ++j) {
                const float *d = dst + ((size_t)bs * m + j) * 3;
                const float p0 = s[0] * d[0];
                const float p01 = fmaf(s[1], d[1], p0);
                const float dot = fmaf(s[2], d[2], p01);
                const float t = -2.0f * dot;
                const float u = t + ss;
                float v = u + sq3(d);
                if (!(v > 0.0f)) v = 0.0f;
                dist[((size_t)bs * n + i) * m + j] = v;
            }
        }
}

/* utils/model_utils/radarflow_util.py:88-99 (knn_point): for every query in new_xyz
 * (B,S,3) the nsample nearest points of xyz (B,N,3) under square_distance above.
 * torch.topk(sorted=False) leaves order and tie-breaking unspecified; the canonical
 * order here is ascending (distance, index).  Optionally returns the distances. */
void orc_knn(int b, int n, int s, int nsample, const float *xyz, const float *new_xyz,
             int *idx, float *dist_out)
{
    float *row = (float *)malloc(sizeof(float) * (size_t)n);
    float *bd = (float *)malloc(sizeof(float) * (size_t)nsample);
    int *bi = (int *)malloc(sizeof(int) * (size_t)nsample);
    for (int bs = 0; bs < b; ++bs)
        for (int q = 0; q < s; ++q) {
            orc_square_distance(1, 1, n, new_xyz + ((size_t)bs * s + q) * 3,
                                xyz + (size_t)bs * n * 3, row);
            int cnt = 0;
            for (int j = 0; j < n; ++j) {
                const float d = row[j];
                if (cnt == nsample && !(d < bd[cnt - 1])) continue;
                int pos = cnt < nsample ? cnt : nsample - 1;
                while (pos > 0 && d < bd[pos - 1]) {   /* strict: earlier index wins ties */
                    bd[pos] = bd[pos - 1];
                    bi[pos] = bi[pos - 1];
                    --pos;
                }
                bd[pos] = d;
                bi[pos] = j;
                if (cnt < nsample) ++cnt;
            }
            for (int l = 0; l < nsample; ++l) {
                idx[((size_t)bs * s + q) * nsample + l] = l < cnt ? bi[l] : 0;
                if (dist_out) dist_out[((size_t)bs * s + q) * nsample + l] = l < cnt ? bd[l] : 0.0f;
            }
        }
    free(row); free(bd); free(bi);
}

/* ---- remaining pointnet2_cuda surface (not on the CMFlow path) ---- */

/* lib/src/sampling_gpu.cu:8-24 -- out[b,c,j] = points[b,c,idx[b,j]] */
void orc_gather_points(int b, int c, int n, int npoints, const float *points, const int *idx, float *out)
{
    for (int bs = 0; bs < b; ++bs)
        for (int ch = 0; ch < c; ++ch)
            for (int j = 0; j < npoints; ++j)
                out[((size_t)bs * c + ch) * npoints + j] =
                    points[((size_t)bs * c + ch) * n + idx[(size_t)bs * npoints + j]];
}

/* lib/src/sampling_gpu.cu:46-63 */
void orc_gather_points_grad(int b, int c, int n, int npoints, const float *grad_out, const int *idx,
                            float *grad_points)
{
    for (int bs = 0; bs < b; ++bs)
        for (int ch = 0; ch < c; ++ch)
            for (int j = 0; j < npoints; ++j)
                grad_points[((size_t)bs * c + ch) * n + idx[(size_t)bs * npoints + j]] +=
                    grad_out[((size_t)bs * c + ch) * npoints + j];
}

/* lib/src/interpolate_gpu.cu:81-124 -- three nearest known points per unknown point;
 * plain (ux-x)^2 form, strict '<' cascade so the earliest index wins ties; returns dist^2. */
void orc_three_nn(int b, int n, int m, const float *unknown, const float *known, float *dist2, int *idx)
{
    for (int bs = 0; bs < b; ++bs)
        for (int i = 0; i < n; ++i) {
            const float *u = unknown + ((size_t)bs * n + i) * 3;
            double best1 = 1e40, best2 = 1e40, best3 = 1e40;
            int b1 = 0, b2 = 0, b3 = 0;
            for (int k = 0; k < m; ++k) {
                const float *p = known + ((size_t)bs * m + k) * 3;
                const float dx = u[0] - p[0], dy = u[1] - p[1], dz = u[2] - p[2];
                const float xx = dx * dx, yy = dy * dy, zz = dz * dz;
                const float s = xx + yy;
                const float d = s + zz;
                if (d < best1) { best3 = best2; b3 = b2; best2 = best1; b2 = b1; best1 = d; b1 = k; }
                else if (d < best2) { best3 = best2; b3 = b2; best2 = d; b2 = k; }
                else if (d < best3) { best3 = d; b3 = k; }
            }
            float *d2 = dist2 + ((size_t)bs * n + i) * 3;
            int *ix = idx + ((size_t)bs * n + i) * 3;
            d2[0] = (float)best1; d2[1] = (float)best2; d2[2] = (float)best3;
            ix[0] = b1; ix[1] = b2; ix[2] = b3;
        }
}

/* lib/src/interpolate_gpu.cu:9-57 (knn_kernel_fast) -- k nearest known points per unknown point, direct
 * (ux-x)^2 form, ascending, strict '<' so the first-seen point wins ties; returns dist^2 (k <= 200). */
void orc_knn_points(int b, int n, int m, int k, const float *unknown, const float *known, float *dist2, int *idx)
{
    double best[200];
    int besti[200];
    for (int bs = 0; bs < b; ++bs)
        for (int i = 0; i < n; ++i) {
            const float *u = unknown + ((size_t)bs * n + i) * 3;
            for (int t = 0; t < k; ++t) { best[t] = 1e40; besti[t] = 0; }
            for (int j = 0; j < m; ++j) {
                const float *p = known + ((size_t)bs * m + j) * 3;
                const float dx = u[0] - p[0], dy = u[1] - p[1], dz = u[2] - p[2];
                const float xx = dx * dx, yy = dy * dy, zz = dz * dz;
                const float s = xx + yy;
                const float d = s + zz;
                for (int t = 0; t < k; ++t)
                    if (d < best[t]) {
                        for (int l = k - 1; l > t; --l) { best[l] = best[l - 1]; besti[l] = besti[l - 1]; }
                        best[t] = d; besti[t] = j;
                        break;
                    }
            }
            for (int t = 0; t < k; ++t) {
                idx[((size_t)bs * n + i) * k + t] = besti[t];
                dist2[((size_t)bs * n + i) * k + t] = (float)best[t];
            }
        }
}

/* lib/src/interpolate_gpu.cu:192-214 -- grad_points[b,c,idx[b,i,j]] += grad_out[b,c,i] * w[b,i,j] */
void orc_three_interpolate_grad(int b, int c, int n, int m, const float *grad_out, const int *idx,
                                const float *weight, float *grad_points)
{
    for (int bs = 0; bs < b; ++bs)
        for (int ch = 0; ch < c; ++ch)
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < 3; ++j)
                    grad_points[((size_t)bs * c + ch) * m + idx[((size_t)bs * n + i) * 3 + j]] +=
                        grad_out[((size_t)bs * c + ch) * n + i] * weight[((size_t)bs * n + i) * 3 + j];
}

/* lib/src/interpolate_gpu.cu:149-169 -- out[b,c,i] = sum_j w[b,i,j] * points[b,c,idx[b,i,j]] */
void orc_three_interpolate(int b, int c, int m, int n, const float *points, const int *idx,
                           const float *weight, float *out)
{
    for (int bs = 0; bs < b; ++bs)
        for (int ch = 0; ch < c; ++ch)
            for (int i = 0; i < n; ++i) {
                const float *w = weight + ((size_t)bs * n + i) * 3;
                const int *ix = idx + ((size_t)bs * n + i) * 3;
                const float *row = points + ((size_t)bs * c + ch) * m;
                const float t0 = w[0] * row[ix[0]];
                const float t1 = w[1] * row[ix[1]];
                const float t2 = w[2] * row[ix[2]];
                const float s = t0 + t1;
                out[((size_t)bs * c + ch) * n + i] = s + t2;
            }
}

/* lib/src/sampling_gpu.cu:93-209 -- iterative furthest point sampling, start at index 0,
 * temp holds the running min distance (caller fills with 1e10, lib/pointnet2_utils.py:26);
 * the |p|^2 <= 1e-3 skip is commented out upstream (sampling_gpu.cu:129-131) and is not
 * applied; the lowest index wins an argmax tie (scan order). */
void orc_furthest_point_sampling(int b, int n, int m, const float *dataset, float *temp, int *idxs)
{
    for (int bs = 0; bs < b; ++bs) {
        const float *d = dataset + (size_t)bs * n * 3;
        float *t = temp + (size_t)bs * n;
        int *o = idxs + (size_t)bs * m;
        int old = 0;
        o[0] = 0;
        for (int j = 1; j < m; ++j) {
            int besti = 0;
            float best = -1.0f;
            const float x1 = d[old * 3 + 0], y1 = d[old * 3 + 1], z1 = d[old * 3 + 2];
            for (int k = 0; k < n; ++k) {
                const float x2 = d[k * 3 + 0], y2 = d[k * 3 + 1], z2 = d[k * 3 + 2];
                const float dx = x2 - x1, dy = y2 - y1, dz = z2 - z1;
                const float xx = dx * dx, yy = dy * dy, zz = dz * dz;
                const float s = xx + yy;
                const float dd = s + zz;
                const float d2 = dd < t[k] ? dd : t[k];
                t[k] = d2;
                if (d2 > best) { best = d2; besti = k; }
            }
            old = besti;
            o[j] = old;
        }
    }
}
