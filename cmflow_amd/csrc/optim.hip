// The optimizer step of the training loop (main.py:107: torch.optim.Adam, lr 1e-3, weight decay 1e-4) as ONE launch over the flat
// gradient bucket.  torch's fused Adam takes six multi-tensor launches for the model's 182 parameter tensors (0.16 ms at the very
// end of a step, where nothing overlaps it); here the gradients, first and second moments are three flat arrays in bucket order and
// only the parameters stay where the module holds them (a table of their addresses and bucket offsets).  Same update rule, in
// torch's order of operations (torch/optim/adam.py, `_fused_adam`; L2 weight decay added to the gradient, no amsgrad):
//   g' = g + wd p;  m += (1 - b1) (g' - m);  v = b2 v + (1 - b2) g' g';  p -= (lr / (1 - b1^t)) m / (sqrt(v) / sqrt(1 - b2^t) + eps)
#include <cstdint>
#include "cmf_common.h"
#include "../../include/cmflow_hip.h"

constexpr int AD_THREADS = 256;
constexpr int AD_CHUNK = 1024;                 // bucket elements per workgroup

__global__ __launch_bounds__(AD_THREADS) void adam_flat_kernel(int n_tensors, const long long *__restrict__ offsets /* [n + 1] */,
                                                               float *const *__restrict__ params, long long total,
                                                               const float *__restrict__ grad, float *__restrict__ m, float *__restrict__ v,
                                                               float wd, float b1w, float b2, float b2w, float step_size, float bc2_sqrt, float eps)
{
    const long long e0 = (long long)blockIdx.x * AD_CHUNK;
    // the tensor that holds the chunk's first element (binary search, one lane's worth of work), then a walk: most chunks lie in one tensor
    int t = 0;
    {
        int lo = 0, hi = n_tensors;            // offsets[lo] <= e0 < offsets[hi]
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (offsets[mid] <= e0) lo = mid; else hi = mid; }
        t = lo;
    }
    for (int i = threadIdx.x; i < AD_CHUNK; i += AD_THREADS) {
        const long long e = e0 + i;
        if (e >= total) return;
        int tt = t;
        while (offsets[tt + 1] <= e) ++tt;
        float *p = params[tt] + (e - offsets[tt]);
        const float pv = *p;
        const float g = fmaf(wd, pv, grad[e]);
        const float mv = m[e], vv = v[e];
        const float mn = mv + b1w * (g - mv);
        const float vn = b2 * vv + b2w * (g * g);
        m[e] = mn; v[e] = vn;
        const float denom = sqrtf(vn) / bc2_sqrt + eps;
        *p = pv - step_size * (mn / denom);
    }
}

// The hyper-parameters arrive as doubles (Python floats) and every derived scalar is formed in double before it is rounded to fp32
// once, as torch does: 1 - beta is (float)(1.0 - beta), not 1.f - (float)beta (those differ by 1e-5 relative for beta2 = 0.999).
extern "C" int cmf_adam_step(int n_tensors, const long long *offsets, float *const *params, long long total, const float *grad, float *m,
                             float *v, double lr, double beta1, double beta2, double eps, double weight_decay, long long step, void *stream)
{
    CMF_CHECK_ARG(n_tensors > 0 && offsets && params && total > 0 && grad && m && v && step >= 1 && lr >= 0.0);
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    hipLaunchKernelGGL(adam_flat_kernel, dim3((unsigned)cmf_divup(total, AD_CHUNK)), dim3(AD_THREADS), 0, (hipStream_t)stream, n_tensors, offsets,
                       params, total, grad, m, v, (float)weight_decay, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)(lr / bc1),
                       (float)sqrt(bc2), (float)eps);
    return cmf_launch_status();
}
