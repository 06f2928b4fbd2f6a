// Lab: sustained fp32 MFMA rate (and so the chip's clock under that load) of register-only loops, 32x32x2 vs 16x16x4,
// 1 / 2 / 3 waves per SIMD, random operands.  hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

template <int NACC>
__global__ __launch_bounds__(256) void k32(float *out, int iters, float seed)
{
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a[4], b[4];
    for (int t = 0; t < 4; ++t) { a[t] = seed * (threadIdx.x * 0.37f + t) - (int)(seed * (threadIdx.x * 0.37f + t)); b[t] = 1.f - a[t] * 1.9f; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[(t + i) & 3], acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k16(float *out, int iters, float seed)
{
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    float a[4], b[4];
    for (int t = 0; t < 4; ++t) { a[t] = seed * (threadIdx.x * 0.37f + t) - (int)(seed * (threadIdx.x * 0.37f + t)); b[t] = 1.f - a[t] * 1.9f; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b[(t + i) & 3], acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename F>
static void run(const char *name, F launch, double flop_per_iter_wave, int wgs_per_cu)
{
    float *out; CK(hipMalloc(&out, 256 * 8 * 256 * 4));
    const int iters = 20000;
    launch(256 * wgs_per_cu, out, 100); CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, 0)); launch(256 * wgs_per_cu, out, iters); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double flops = flop_per_iter_wave * iters * 4.0 * 256 * wgs_per_cu;
        printf("%-34s %d wg/cu  %8.2f ms  %6.1f TF\n", name, wgs_per_cu, ms, flops / (ms * 1e-3) / 1e12);
    }
    CK(hipFree(out));
}
int main()
{
    for (int w = 1; w <= 3; ++w) {
        run("32x32x2, 8 accumulators", [](int g, float *o, int it) { hipLaunchKernelGGL(k32<8>, dim3(g), dim3(256), 0, 0, o, it, 0.731f); }, 8 * 4 * 4096.0, w);
        run("32x32x2, 4 accumulators", [](int g, float *o, int it) { hipLaunchKernelGGL(k32<4>, dim3(g), dim3(256), 0, 0, o, it, 0.731f); }, 4 * 4 * 4096.0, w);
        run("16x16x4, 32 accumulators", [](int g, float *o, int it) { hipLaunchKernelGGL(k16<32>, dim3(g), dim3(256), 0, 0, o, it, 0.731f); }, 32 * 4 * 2048.0, w);
        run("16x16x4, 8 accumulators", [](int g, float *o, int it) { hipLaunchKernelGGL(k16<8>, dim3(g), dim3(256), 0, 0, o, it, 0.731f); }, 8 * 4 * 2048.0, w);
    }
    return 0;
}
