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
