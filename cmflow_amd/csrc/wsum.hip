// Cost-volume weighting (radarflow_util.py:219-221,235-236): cost[m,c] = sum_k w[m,k,c] * x[m,k,c], the product of
// the WeightNet output with the patch features summed over the K neighbours -- in the reference (and as torch ops)
// a (B,N,K,512) product tensor plus a reduction, and in backward two more such products, a comparison mask and a
// select.  Here one streaming kernel per direction:
//   forward : reads w and x once (x optionally gathered on the fly from per-point rows: the patch-to-patch stage
//             sums w * p2p[idx], radarflow_util.py:234-236, so the grouped tensor is never written), writes cost;
//   backward: dw = dcost * x, dx = dcost * w (times leaky'(x) when x is a stored LeakyReLU(0.1) activation, so the
//             consumer receives the gradient w.r.t. the pre-activation), both written once.
// Rows m = sample*n1 + point; C % 4 == 0; one thread owns 4 channels of a row and loops over K with all loads of
// the row in flight.
#include "cmf_common.h"
#include "../../include/cmflow_hip.h"

constexpr int WS_THREADS = 256;

template <int KU>
__global__ __launch_bounds__(WS_THREADS) void wsum_fwd_kernel(
    long long M, int K, int C, int n1, int n_src, const float *__restrict__ w, const float *__restrict__ x,
    const int *__restrict__ idx, float *__restrict__ out)
{
    const int cv = C / 4;
    const long long total = M * cv;
    for (long long t = (long long)blockIdx.x * WS_THREADS + threadIdx.x; t < total; t += (long long)gridDim.x * WS_THREADS) {
        const long long m = t / cv;
        const int c = (int)(t - m * cv) * 4;
        const float *wp = w + (m * K) * C + c;
        const long long src_base = idx ? (m / n1) * n_src : 0;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k0 = 0; k0 < K; k0 += KU) {
            float4 wv[KU], xv[KU];
#pragma unroll
            for (int u = 0; u < KU; ++u) {
                const int k = k0 + u;
                if (k < K) {
                    wv[u] = *(const float4 *)(wp + (long long)k * C);
                    const long long row = idx ? src_base + idx[m * K + k] : m * K + k;
                    xv[u] = *(const float4 *)(x + row * C + c);
                } else { wv[u] = make_float4(0.f, 0.f, 0.f, 0.f); xv[u] = wv[u]; }
            }
#pragma unroll
            for (int u = 0; u < KU; ++u) {
                acc.x = fmaf(wv[u].x, xv[u].x, acc.x); acc.y = fmaf(wv[u].y, xv[u].y, acc.y);
                acc.z = fmaf(wv[u].z, xv[u].z, acc.z); acc.w = fmaf(wv[u].w, xv[u].w, acc.w);
            }
        }
        *(float4 *)(out + m * C + c) = acc;
    }
}

__global__ __launch_bounds__(WS_THREADS) void wsum_bwd_kernel(
    long long M, int K, int C, int n1, int n_src, int leaky, const float *__restrict__ dcost, const float *__restrict__ w,
    const float *__restrict__ x, const int *__restrict__ idx, float *__restrict__ dw, float *__restrict__ dx,
    float *__restrict__ dx_colsum)
{
    // dx_colsum != NULL: per-workgroup column sums of dx ([gridDim.x][C], summed by cmf_colsum) -- the bias gradient of
    // the layer that produced x.  The launcher then picks a grid whose thread stride is a multiple of C/4, so a thread
    // keeps the same 4 channels for all its rows.
    __shared__ float4 red[WS_THREADS];
    float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
    const int cv = C / 4;
    const long long total = M * K * cv;                   // one thread per (m, k, 4 channels): fully coalesced rows
    for (long long t = (long long)blockIdx.x * WS_THREADS + threadIdx.x; t < total; t += (long long)gridDim.x * WS_THREADS) {
        const long long mk = t / cv;
        const int c = (int)(t - mk * cv) * 4;
        const long long m = mk / K;
        const float4 g = *(const float4 *)(dcost + m * C + c);
        const float4 wv = *(const float4 *)(w + mk * C + c);
        const long long row = idx ? (m / n1) * n_src + idx[mk] : mk;
        const float4 xv = *(const float4 *)(x + row * C + c);
        if (dw) {
            float4 e = make_float4(g.x * xv.x, g.y * xv.y, g.z * xv.z, g.w * xv.w);
            if (leaky & 2) {                                  // w is a stored ReLU activation: gradient w.r.t. its pre-activation
                e.x = wv.x > 0.f ? e.x : 0.f; e.y = wv.y > 0.f ? e.y : 0.f;
                e.z = wv.z > 0.f ? e.z : 0.f; e.w = wv.w > 0.f ? e.w : 0.f;
            }
            *(float4 *)(dw + mk * C + c) = e;
        }
        float4 d = make_float4(g.x * wv.x, g.y * wv.y, g.z * wv.z, g.w * wv.w);
        if (leaky & 1) {
            d.x = xv.x > 0.f ? d.x : 0.1f * d.x; d.y = xv.y > 0.f ? d.y : 0.1f * d.y;
            d.z = xv.z > 0.f ? d.z : 0.1f * d.z; d.w = xv.w > 0.f ? d.w : 0.1f * d.w;
        }
        if (dx) *(float4 *)(dx + mk * C + c) = d;
        cs.x += d.x; cs.y += d.y; cs.z += d.z; cs.w += d.w;
    }
    if (dx_colsum) {
        // threads t and t + cv (+ 2cv ...) of the workgroup own the same channels: fold them in fixed order
        red[threadIdx.x] = cs;
        __syncthreads();
        if ((int)threadIdx.x < cv) {
            float4 s = red[threadIdx.x];
            for (int t = threadIdx.x + cv; t < WS_THREADS; t += cv) { const float4 v = red[t]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
            *(float4 *)(dx_colsum + (long long)blockIdx.x * C + 4 * threadIdx.x) = s;
        }
    }
}

extern "C" int cmf_weighted_ksum(long long M, int K, int C, int n1, int n_src, const float *w, const float *x, const int *idx,
                                 float *out, void *stream)
{
    CMF_CHECK_ARG(M >= 0 && K > 0 && C > 0 && C % 4 == 0);
    if (M == 0) return 0;
    CMF_CHECK_ARG(w && x && out && (!idx || (n1 > 0 && n_src > 0)));
    CMF_CHECK_ARG((((uintptr_t)w | (uintptr_t)x | (uintptr_t)out) & 15) == 0);
    const long long total = M * (C / 4);
    const int grid = (int)((total + WS_THREADS - 1) / WS_THREADS < 256 * 64 ? (total + WS_THREADS - 1) / WS_THREADS : 256 * 64);
    if (K <= 8)
        hipLaunchKernelGGL(wsum_fwd_kernel<8>, dim3(grid), dim3(WS_THREADS), 0, (hipStream_t)stream, M, K, C, n1, n_src, w, x, idx, out);
    else
        hipLaunchKernelGGL(wsum_fwd_kernel<4>, dim3(grid), dim3(WS_THREADS), 0, (hipStream_t)stream, M, K, C, n1, n_src, w, x, idx, out);
    return cmf_launch_status();
}

extern "C" int cmf_weighted_ksum_grad_tiles(int C)
{
    return (C % 4 == 0 && C / 4 <= WS_THREADS && WS_THREADS % (C / 4) == 0) ? 1024 : 0;      // 0: column sums not supported for this C
}

extern "C" int cmf_weighted_ksum_grad(long long M, int K, int C, int n1, int n_src, int leaky, const float *dcost, const float *w,
                                      const float *x, const int *idx, float *dw, float *dx, float *dx_colsum, void *stream)
{
    CMF_CHECK_ARG(M >= 0 && K > 0 && C > 0 && C % 4 == 0);
    if (M == 0) return 0;
    CMF_CHECK_ARG(dcost && w && x && (dw || dx) && (!idx || (n1 > 0 && n_src > 0)));
    CMF_CHECK_ARG((((uintptr_t)w | (uintptr_t)x | (uintptr_t)dcost | (uintptr_t)dw | (uintptr_t)dx) & 15) == 0);
    const long long total = M * K * (C / 4);
    int grid = (int)((total + WS_THREADS - 1) / WS_THREADS < 256 * 64 ? (total + WS_THREADS - 1) / WS_THREADS : 256 * 64);
    if (dx_colsum) {
        CMF_CHECK_ARG(dx && cmf_weighted_ksum_grad_tiles(C) > 0);
        grid = cmf_weighted_ksum_grad_tiles(C);               // every workgroup writes its [C] partial, also when it has no rows
    }
    hipLaunchKernelGGL(wsum_bwd_kernel, dim3(grid), dim3(WS_THREADS), 0, (hipStream_t)stream, M, K, C, n1, n_src, leaky, dcost, w, x,
                       idx, dw, dx, dx_colsum);
    return cmf_launch_status();
}

// ---- WeightNet tail fused into the weighting ----------------------------------------------------------------
// The weights of the two calls above are the output of WeightNet's last 1x1 conv + ReLU (radarflow_util.py:307-318,
// hidden width 8 -> C).  As a tensor they are (M,K,C) -- with C = 512 the largest activation of the cost volume --
// written by one kernel and read back by two; here they are recomputed where they are used,
//     w[m,k,c] = relu(bl[c] + sum_j h[m,k,j] * Wl[c,j]),     h (M*K, 8) the hidden activation,
// so forward reads x once and backward reads x and writes dx once; the (M,K,C) weights and their gradient never
// exist.  Backward also produces the last conv's parameter gradients and the gradient of h:
//     e[m,k,c] = dcost[m,c] * x[m,k,c] * (w > 0);  dWl[c,j] = sum_mk e*h[.,j];  dbl[c] = sum_mk e;  dh[mk,j] = sum_c e*Wl[c,j].
// One thread owns 4 channels (its 4 rows of Wl live in registers) and a group of C/4 threads owns one (m,k) slot, so
// every access to x / dx is a coalesced row; C/4 is a multiple of the wavefront (C = 256, 512, 1024) so the slot --
// and with it h, idx and dcost's row -- is uniform across a wavefront.
constexpr int WN_J = 8;
constexpr int WN_TILES = 768;          // 3 workgroups per CU: what the gradient kernel's registers allow resident

__device__ __forceinline__ void wn_load_w(const float *__restrict__ Wl, const float *__restrict__ bl, int c, float (&W)[4][WN_J], float (&b)[4])
{
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float4 lo = *(const float4 *)(Wl + (long long)(c + i) * WN_J), hi = *(const float4 *)(Wl + (long long)(c + i) * WN_J + 4);
        W[i][0] = lo.x; W[i][1] = lo.y; W[i][2] = lo.z; W[i][3] = lo.w; W[i][4] = hi.x; W[i][5] = hi.y; W[i][6] = hi.z; W[i][7] = hi.w;
        b[i] = bl[c + i];
    }
}

__device__ __forceinline__ void wn_pre(const float (&W)[4][WN_J], const float (&b)[4], const float (&hv)[WN_J], float (&pre)[4])
{
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float a = b[i];
#pragma unroll
        for (int j = 0; j < WN_J; ++j) a = fmaf(hv[j], W[i][j], a);
        pre[i] = a;
    }
}

__global__ __launch_bounds__(WS_THREADS) void wn_ksum_fwd_kernel(
    int M, int K, int C, int n1, int n_src, const float *__restrict__ h, const float *__restrict__ Wl, const float *__restrict__ bl,
    const float *__restrict__ x, const int *__restrict__ idx, float *__restrict__ out)
{
    const int cv = C / 4, R = WS_THREADS / cv;
    const int c = ((int)threadIdx.x % cv) * 4, r = (int)threadIdx.x / cv;
    float W[4][WN_J], b[4];
    wn_load_w(Wl, bl, c, W, b);
    for (int m0 = (int)blockIdx.x * R; m0 < M; m0 += (int)gridDim.x * R) {
        const int m = __builtin_amdgcn_readfirstlane(m0 + r);
        if (m >= M) continue;
        const long long src_base = idx ? (long long)(m / n1) * n_src : 0;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int k0 = 0; k0 < K; k0 += 4) {
            float4 xv[4];
            float hv[4][WN_J];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = k0 + u < K ? k0 + u : K - 1;
                const long long mk = (long long)m * K + k;
                const long long row = idx ? src_base + idx[mk] : mk;
                xv[u] = *(const float4 *)(x + row * C + c);
#pragma unroll
                for (int j = 0; j < WN_J; ++j) hv[u][j] = h[mk * WN_J + j];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (k0 + u < K) {
                    float pre[4];
                    wn_pre(W, b, hv[u], pre);
                    acc[0] = fmaf(fmaxf(pre[0], 0.f), xv[u].x, acc[0]); acc[1] = fmaf(fmaxf(pre[1], 0.f), xv[u].y, acc[1]);
                    acc[2] = fmaf(fmaxf(pre[2], 0.f), xv[u].z, acc[2]); acc[3] = fmaf(fmaxf(pre[3], 0.f), xv[u].w, acc[3]);
                }
            }
        }
        *(float4 *)(out + (long long)m * C + c) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
}

__global__ __launch_bounds__(WS_THREADS, 3) void wn_ksum_bwd_kernel(
    int M, int K, int C, int n1, int n_src, int leaky, const float *__restrict__ dcost, long long ldd, const float *__restrict__ h,
    const float *__restrict__ Wl, const float *__restrict__ bl, const float *__restrict__ x, const int *__restrict__ idx,
    float *__restrict__ dx, float *__restrict__ dh, float *__restrict__ part)
{
    __shared__ float sred[2][WS_THREADS / 64][WN_J];
    __shared__ float4 fold[WS_THREADS];
    const int cv = C / 4, R = WS_THREADS / cv, wpr = cv / 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = (tid % cv) * 4, r = tid / cv;
    float W[4][WN_J], b[4];
    wn_load_w(Wl, bl, c, W, b);
    float dW[4][WN_J];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < WN_J; ++j) dW[i][j] = 0.f;
    float db[4] = {0.f, 0.f, 0.f, 0.f}, cs[4] = {0.f, 0.f, 0.f, 0.f};
    float dbh = 0.f;                                    // threads tid < R*8: column sum of the masked dh they write
    const int MK = M * K;
    int it = 0;
    // the row's two streams (x, dcost) are requested one iteration ahead: an iteration is a dependent chain load -> ~110 FMAs -> wave
    // reduction -> barrier, and with three workgroups per CU nothing else covered the loads' latency (2 us per iteration for 0.65 us of
    // arithmetic: tools/fc_profile.py)
    auto request = [&](int row0, float4 &g4, float4 &x4) {
        const int mk = __builtin_amdgcn_readfirstlane(row0 + r);
        if (mk < MK) {
            const int m = mk / K;
            const long long row = idx ? (long long)(m / n1) * n_src + idx[mk] : mk;
            g4 = *(const float4 *)(dcost + (long long)m * ldd + c);
            x4 = *(const float4 *)(x + row * C + c);
        }
    };
    float4 g4n = make_float4(0.f, 0.f, 0.f, 0.f), x4n = g4n;
    request((int)blockIdx.x * R, g4n, x4n);
    for (int row0 = (int)blockIdx.x * R; row0 < MK; row0 += (int)gridDim.x * R, ++it) {
        const int mk = __builtin_amdgcn_readfirstlane(row0 + r);
        float p[WN_J];
#pragma unroll
        for (int j = 0; j < WN_J; ++j) p[j] = 0.f;
        const float4 g4 = g4n, x4 = x4n;
        if (row0 + (int)gridDim.x * R < MK) request(row0 + (int)gridDim.x * R, g4n, x4n);
        if (mk < MK) {
            float hv[WN_J];
#pragma unroll
            for (int j = 0; j < WN_J; ++j) hv[j] = h[(long long)mk * WN_J + j];
            const float g[4] = {g4.x, g4.y, g4.z, g4.w}, xv[4] = {x4.x, x4.y, x4.z, x4.w};
            float pre[4], d[4];
            wn_pre(W, b, hv, pre);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float e = pre[i] > 0.f ? g[i] * xv[i] : 0.f;
                float dd = g[i] * fmaxf(pre[i], 0.f);
                if (leaky & 1) dd = xv[i] > 0.f ? dd : 0.1f * dd;
                d[i] = dd;
                cs[i] += dd;
                db[i] += e;
#pragma unroll
                for (int j = 0; j < WN_J; ++j) {
                    dW[i][j] = fmaf(e, hv[j], dW[i][j]);
                    p[j] = fmaf(e, W[i][j], p[j]);
                }
            }
            *(float4 *)(dx + (long long)mk * C + c) = make_float4(d[0], d[1], d[2], d[3]);
        }
        // sum p[0..7] over the wavefront in a fixed tree: each exchange halves the values a lane carries
        float q4[4], q2[2], q1;
        const bool up5 = lane & 32, up4 = lane & 16, up3 = lane & 8;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float keep = up5 ? p[i + 4] : p[i], send = up5 ? p[i] : p[i + 4];
            q4[i] = keep + __shfl_xor(send, 32, 64);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float keep = up4 ? q4[i + 2] : q4[i], send = up4 ? q4[i] : q4[i + 2];
            q2[i] = keep + __shfl_xor(send, 16, 64);
        }
        {
            const float keep = up3 ? q2[1] : q2[0], send = up3 ? q2[0] : q2[1];
            q1 = keep + __shfl_xor(send, 8, 64);
        }
        q1 += __shfl_xor(q1, 4, 64);
        q1 += __shfl_xor(q1, 2, 64);
        q1 += __shfl_xor(q1, 1, 64);
        if ((lane & 7) == 0) sred[it & 1][wave][(lane >> 5) * 4 + ((lane >> 4) & 1) * 2 + ((lane >> 3) & 1)] = q1;
        __syncthreads();
        if (tid < R * WN_J) {
            const int rr = tid / WN_J, j = tid % WN_J;
            if (row0 + rr < MK) {
                float s = sred[it & 1][rr * wpr][j];
                for (int w = 1; w < wpr; ++w) s += sred[it & 1][rr * wpr + w][j];
                if (leaky & 4) {                        // h is a stored ReLU activation: gradient w.r.t. its pre-activation
                    s = h[(long long)(row0 + rr) * WN_J + j] > 0.f ? s : 0.f;
                    dbh += s;
                }
                dh[(long long)(row0 + rr) * WN_J + j] = s;
            }
        }
    }
    // fold the R thread groups that own the same channels, then one [C*8 | C | C | 8] partial row per workgroup
    float *prow = part + (long long)blockIdx.x * (C * (WN_J + 2) + WN_J);
    {
        __syncthreads();
        float *fl = (float *)fold;
        if (tid < R * WN_J) fl[tid] = dbh;
        __syncthreads();
        if (tid < WN_J) {
            float t = fl[tid];
            for (int rr = 1; rr < R; ++rr) t += fl[rr * WN_J + tid];
            prow[C * (WN_J + 2) + tid] = t;
        }
    }
#pragma unroll
    for (int q = 0; q < 10; ++q) {
        float4 v;
        if (q < 8) { const int i = q >> 1, j = (q & 1) * 4; v = make_float4(dW[i][j], dW[i][j + 1], dW[i][j + 2], dW[i][j + 3]); }
        else if (q == 8) v = make_float4(db[0], db[1], db[2], db[3]);
        else v = make_float4(cs[0], cs[1], cs[2], cs[3]);
        __syncthreads();
        fold[tid] = v;
        __syncthreads();
        if (r == 0) {
            for (int rr = 1; rr < R; ++rr) { const float4 o = fold[tid + rr * cv]; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
            if (q < 8) *(float4 *)(prow + (c + (q >> 1)) * WN_J + (q & 1) * 4) = v;
            else *(float4 *)(prow + C * WN_J + (q - 8) * C + c) = v;
        }
    }
}

extern "C" int cmf_weightnet_ksum_tiles(int C)
{
    return (C == 256 || C == 512 || C == 1024) ? WN_TILES : 0;
}

extern "C" int cmf_weightnet_ksum(long long M, int K, int C, int n1, int n_src, const float *h, const float *Wl, const float *bl,
                                  const float *x, const int *idx, float *out, void *stream)
{
    CMF_CHECK_ARG(M >= 0 && K > 0 && cmf_weightnet_ksum_tiles(C) > 0 && M * K < (1LL << 31));
    if (M == 0) return 0;
    CMF_CHECK_ARG(h && Wl && bl && x && out && (!idx || (n1 > 0 && n_src > 0)));
    CMF_CHECK_ARG((((uintptr_t)h | (uintptr_t)Wl | (uintptr_t)x | (uintptr_t)out) & 15) == 0);
    const int R = WS_THREADS / (C / 4);
    const long long want = (M + R - 1) / R;
    const int grid = (int)(want < 2048 ? want : 2048);
    hipLaunchKernelGGL(wn_ksum_fwd_kernel, dim3(grid), dim3(WS_THREADS), 0, (hipStream_t)stream, (int)M, K, C, n1, n_src, h, Wl, bl, x, idx, out);
    return cmf_launch_status();
}

extern "C" int cmf_weightnet_ksum_grad(long long M, int K, int C, int n1, int n_src, int leaky, const float *dcost, long long ldd, const float *h,
                                       const float *Wl, const float *bl, const float *x, const int *idx, float *dx, float *dh,
                                       float *part, void *stream)
{
    CMF_CHECK_ARG(M >= 0 && K > 0 && cmf_weightnet_ksum_tiles(C) > 0 && M * K < (1LL << 31));
    CMF_CHECK_ARG(dcost && h && Wl && bl && x && dx && dh && part && (!idx || (n1 > 0 && n_src > 0)) && ldd >= C && ldd % 4 == 0);
    CMF_CHECK_ARG((((uintptr_t)h | (uintptr_t)Wl | (uintptr_t)x | (uintptr_t)dcost | (uintptr_t)dx | (uintptr_t)dh | (uintptr_t)part) & 15) == 0);
    // M == 0 still launches: every workgroup writes its (zero) partial row
    hipLaunchKernelGGL(wn_ksum_bwd_kernel, dim3(WN_TILES), dim3(WS_THREADS), 0, (hipStream_t)stream, (int)M, K, C, n1, n_src, leaky, dcost, ldd, h,
                       Wl, bl, x, idx, dx, dh, part);
    return cmf_launch_status();
}
