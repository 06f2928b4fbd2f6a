// Remaining byte movers of the reference's pointnet2_cuda extension (not on the CMFlow path):
//   gather_points(+grad)       lib/src/sampling_gpu.cu:8-24, 46-63
//   three_interpolate(+grad)   lib/src/interpolate_gpu.cu:149-169, 192-214
// Small (npoints / n outputs per row), so a plain one-thread-per-output mapping with coalesced writes is used;
// the backward passes accumulate with hardware fp32 atomics like the reference (order undefined).
#include "cmf_common.h"
#include "../../include/cmflow_hip.h"

constexpr int GI_THREADS = 256;

__global__ void gather_points_kernel(int c, int n, int npoints, const float *__restrict__ points,
                                     const int *__restrict__ idx, float *__restrict__ out)
{
    const int bs = blockIdx.z, ch = blockIdx.y, j = blockIdx.x * GI_THREADS + threadIdx.x;
    if (j >= npoints) return;
    out[((size_t)bs * c + ch) * npoints + j] = points[((size_t)bs * c + ch) * n + idx[(size_t)bs * npoints + j]];
}

__global__ void gather_points_grad_kernel(int c, int n, int npoints, const float *__restrict__ grad_out,
                                          const int *__restrict__ idx, float *__restrict__ grad_points)
{
    const int bs = blockIdx.z, ch = blockIdx.y, j = blockIdx.x * GI_THREADS + threadIdx.x;
    if (j >= npoints) return;
    atomicAdd(grad_points + ((size_t)bs * c + ch) * n + idx[(size_t)bs * npoints + j],
              grad_out[((size_t)bs * c + ch) * npoints + j]);
}

extern "C" int cmf_gather_points(int b, int c, int n, int npoints, const float *points, const int *idx, float *out, void *stream)
{
    CMF_CHECK_ARG(b >= 0 && c >= 0 && n > 0 && npoints >= 0);
    if (b == 0 || c == 0 || npoints == 0) return 0;
    CMF_CHECK_ARG(points && idx && out);
    hipLaunchKernelGGL(gather_points_kernel, dim3(cmf_divup(npoints, GI_THREADS), c, b), dim3(GI_THREADS), 0,
                       (hipStream_t)stream, c, n, npoints, points, idx, out);
    return cmf_launch_status();
}

extern "C" int cmf_gather_points_grad(int b, int c, int n, int npoints, const float *grad_out, const int *idx,
                                      float *grad_points, void *stream)
{
    CMF_CHECK_ARG(b >= 0 && c >= 0 && n > 0 && npoints >= 0);
    if (b == 0 || c == 0 || npoints == 0) return 0;
    CMF_CHECK_ARG(grad_out && idx && grad_points);
    hipLaunchKernelGGL(gather_points_grad_kernel, dim3(cmf_divup(npoints, GI_THREADS), c, b), dim3(GI_THREADS), 0,
                       (hipStream_t)stream, c, n, npoints, grad_out, idx, grad_points);
    return cmf_launch_status();
}

#pragma clang fp contract(off)
__global__ void three_interpolate_kernel(int c, int m, int n, const float *__restrict__ points, const int *__restrict__ idx,
                                         const float *__restrict__ weight, float *__restrict__ out)
{
    const int bs = blockIdx.z, ch = blockIdx.y, i = blockIdx.x * GI_THREADS + threadIdx.x;
    if (i >= n) return;
    const float *w = weight + ((size_t)bs * n + i) * 3;
    const int *ix = idx + ((size_t)bs * n + i) * 3;
    const float *row = points + ((size_t)bs * c + ch) * m;
    const float t0 = w[0] * row[ix[0]];
    const float t1 = w[1] * row[ix[1]];
    const float t2 = w[2] * row[ix[2]];
    const float s = t0 + t1;
    out[((size_t)bs * c + ch) * n + i] = s + t2;
}

__global__ void three_interpolate_grad_kernel(int c, int n, int m, const float *__restrict__ grad_out,
                                              const int *__restrict__ idx, const float *__restrict__ weight,
                                              float *__restrict__ grad_points)
{
    const int bs = blockIdx.z, ch = blockIdx.y, i = blockIdx.x * GI_THREADS + threadIdx.x;
    if (i >= n) return;
    const float g = grad_out[((size_t)bs * c + ch) * n + i];
    const float *w = weight + ((size_t)bs * n + i) * 3;
    const int *ix = idx + ((size_t)bs * n + i) * 3;
    float *row = grad_points + ((size_t)bs * c + ch) * m;
    atomicAdd(row + ix[0], g * w[0]);
    atomicAdd(row + ix[1], g * w[1]);
    atomicAdd(row + ix[2], g * w[2]);
}

extern "C" int cmf_three_interpolate(int b, int c, int m, int n, const float *points, const int *idx,
                                     const float *weight, float *out, void *stream)
{
    CMF_CHECK_ARG(b >= 0 && c >= 0 && m > 0 && n >= 0);
    if (b == 0 || c == 0 || n == 0) return 0;
    CMF_CHECK_ARG(points && idx && weight && out);
    hipLaunchKernelGGL(three_interpolate_kernel, dim3(cmf_divup(n, GI_THREADS), c, b), dim3(GI_THREADS), 0,
                       (hipStream_t)stream, c, m, n, points, idx, weight, out);
    return cmf_launch_status();
}

extern "C" int cmf_three_interpolate_grad(int b, int c, int n, int m, const float *grad_out, const int *idx,
                                          const float *weight, float *grad_points, void *stream)
{
    CMF_CHECK_ARG(b >= 0 && c >= 0 && m > 0 && n >= 0);
    if (b == 0 || c == 0 || n == 0) return 0;
    CMF_CHECK_ARG(grad_out && idx && weight && grad_points);
    hipLaunchKernelGGL(three_interpolate_grad_kernel, dim3(cmf_divup(n, GI_THREADS), c, b), dim3(GI_THREADS), 0,
                       (hipStream_t)stream, c, n, m, grad_out, idx, weight, grad_points);
    return cmf_launch_status();
}
