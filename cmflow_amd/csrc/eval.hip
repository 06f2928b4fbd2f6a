// Evaluation metrics of one batch -- utils/eval_util.py: eval_scene_flow :42-86 (EPE, AccS, AccR,
// RNE, moving/static RNE, SAS, RAS with the sensor-resolution model of get_carterian_res :4-40), eval_motion_seg
// :104-118 (acc, mIoU, sensitivity) and eval_trans_RPE :89-102 with utils/odometry_util.py (RTE, RAE).  The
// reference copies every tensor to the host and runs numpy; here one workgroup per sample reduces the points of
// that sample, a second launch adds the per-sample partials in fixed order, and only 14 doubles leave the device.
// Precision follows the reference's: per-point errors and the spherical angles in fp32 (numpy float32 arrays),
// the resolution model and every accumulation in fp64.
#include "cmf_common.h"
#include "../../include/cmflow_hip.h"

constexpr int EV_THREADS = 256;
constexpr int EV_PART = 16;   // err, accs, accr, rerr, mov_sum, mov_cnt, stat_sum, stat_cnt, sas, ras, tp, tn, fp, fn, rte, rae

__device__ __forceinline__ double ev_block_sum(double v, double *red)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < EV_THREADS / 64; ++w) s += red[w];
    return s;
}

// sum over the three cartesian axes of |d axis / d (r, theta, phi)| . res   (eval_util.py:27-38)
__device__ __forceinline__ double ev_resolution(float r, float st, float ct, float sp, float cp, double rr, double rt, double rp)
{
    const float jx0 = cp * ct, jx1 = (-r * st) * cp, jx2 = (-r * ct) * sp;
    const float jy0 = sp * ct, jy1 = (-r * sp) * st, jy2 = (r * ct) * cp;
    const float jz0 = st, jz1 = r * ct;
    const double x = (fabs((double)jx0) * rr + fabs((double)jx1) * rt) + fabs((double)jx2) * rp;
    const double y = (fabs((double)jy0) * rr + fabs((double)jy1) * rt) + fabs((double)jy2) * rp;
    const double z = (fabs((double)jz0) * rr + fabs((double)jz1) * rt) + 0.0 * rp;
    return (x + y) + z;
}

__global__ __launch_bounds__(EV_THREADS) void eval_sample_kernel(
    int n, const float *__restrict__ pc, const float *__restrict__ pred, const float *__restrict__ labels,
    const float *__restrict__ mask, const float *__restrict__ pred_m, const float *__restrict__ gt_trans,
    const float *__restrict__ pred_trans, double r_res, double th_res, double ph_res, double *__restrict__ partial)
{
    __shared__ double red[EV_THREADS / 64];
    const int bs = blockIdx.x, tid = threadIdx.x;
    const float *p = pc + (size_t)bs * 3 * n;
    double acc[14];
#pragma unroll
    for (int t = 0; t < 14; ++t) acc[t] = 0.0;
    const bool do_sf = pc != nullptr, do_seg = pred_m != nullptr;
    for (int i = tid; i < n; i += EV_THREADS) {
        const float m = mask ? mask[(size_t)bs * n + i] : 0.f;
        if (do_seg) {
            const float pm = pred_m[(size_t)bs * n + i];
            acc[10] += (pm == 1.f && m == 1.f) ? 1.0 : 0.0;
            acc[11] += (pm == 0.f && m == 0.f) ? 1.0 : 0.0;
            acc[12] += (pm == 1.f && m == 0.f) ? 1.0 : 0.0;
            acc[13] += (pm == 0.f && m == 1.f) ? 1.0 : 0.0;
        }
        if (!do_sf) continue;
        const size_t o = ((size_t)bs * n + i) * 3;
        const float dx = pred[o] - labels[o], dy = pred[o + 1] - labels[o + 1], dz = pred[o + 2] - labels[o + 2];
        const float err = sqrtf(((dx * dx + dy * dy) + dz * dz) + 1e-20f);
        const float lx = labels[o], ly = labels[o + 1], lz = labels[o + 2];
        const float glen = sqrtf(((lx * lx + ly * ly) + lz * lz) + 1e-20f);
        const float rel = err / glen;
        acc[0] += (double)err;
        acc[1] += (err <= 0.05f || rel <= 0.05f) ? 1.0 : 0.0;
        acc[2] += (err <= 0.10f || rel <= 0.10f) ? 1.0 : 0.0;
        // resolution-normalised error: radar vs lidar cartesian resolution at this point
        const float x = p[i], y = p[n + i], z = p[2 * n + i];
        const float r = sqrtf((x * x + y * y) + z * z);
        const float th = asinf(z / r), ph = atan2f(y, x);
        const float st = sinf(th), ct = cosf(th), sp = sinf(ph), cp = cosf(ph);
        const double res_r = sqrt(ev_resolution(r, st, ct, sp, cp, r_res, th_res, ph_res) + 1e-20);
        const double res_l = sqrt(ev_resolution(r, st, ct, sp, cp, 0.04, 0.4 * M_PI / 180.0, 0.08 * M_PI / 180.0) + 1e-20);
        const double rerr = (double)err / (res_r / res_l);
        const double rrel = rerr / (double)glen;
        acc[3] += rerr;
        if (m == 0.f) { acc[4] += rerr; acc[5] += 1.0; }
        if (m == 1.f) { acc[6] += rerr; acc[7] += 1.0; }
        acc[8] += (rerr <= 0.10 || rrel <= 0.10) ? 1.0 : 0.0;
        acc[9] += (rerr <= 0.20 || rrel <= 0.20) ? 1.0 : 0.0;
    }
#pragma unroll
    for (int t = 0; t < 14; ++t) {
        const double s = ev_block_sum(acc[t], red);
        if (tid == 0) partial[(size_t)bs * EV_PART + t] = s;
    }
    if (tid == 0 && gt_trans == nullptr) { partial[(size_t)bs * EV_PART + 14] = 0.0; partial[(size_t)bs * EV_PART + 15] = 0.0; }
    if (tid == 0 && gt_trans != nullptr) {
        // relative pose error gt^-1 . pred (odometry_util.py:61-118): the inverse's translation is rounded to fp32
        // like the reference's float32 arrays, the product is fp64
        const float *q = gt_trans + (size_t)bs * 16, *w = pred_trans + (size_t)bs * 16;
        float tinv[3];
        for (int r = 0; r < 3; ++r) tinv[r] = -((q[r] * q[3] + q[4 + r] * q[7]) + q[8 + r] * q[11]);
        double e[3][4];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 4; ++c) {
                double s = ((double)q[r] * w[c] + (double)q[4 + r] * w[4 + c]) + (double)q[8 + r] * w[8 + c];
                s += (double)tinv[r] * w[12 + c];
                e[r][c] = s;
            }
        partial[(size_t)bs * EV_PART + 14] = sqrt((e[0][3] * e[0][3] + e[1][3] * e[1][3]) + e[2][3] * e[2][3]);
        // rotation angle = |rotation vector|: atan2(|axial vector| / 2, (trace - 1) / 2)
        const double ax = e[2][1] - e[1][2], ay = e[0][2] - e[2][0], az = e[1][0] - e[0][1];
        const double s2 = sqrt((ax * ax + ay * ay) + az * az), c2 = (e[0][0] + e[1][1] + e[2][2]) - 1.0;
        partial[(size_t)bs * EV_PART + 15] = fabs(atan2(s2, c2)) * 180.0 / M_PI;
    }
}

// metrics: rne, 50-50 rne, mov_rne, stat_rne, sas, ras, epe, accs, accr | acc, miou, sen | RTE, RAE
__global__ __launch_bounds__(64) void eval_finalize_kernel(int b, int n, const double *__restrict__ partial,
                                                           double *__restrict__ metrics)
{
    __shared__ double s[EV_PART];
    const int t = threadIdx.x;
    if (t < EV_PART) {
        double a = 0.0;
        for (int i = 0; i < b; ++i) a += partial[(size_t)i * EV_PART + t];
        s[t] = a;
    }
    __syncthreads();
    if (t == 0) {
        const double cnt = (double)b * (double)n;
        const double mov = s[4] / (s[5] + 1e-6), stat = s[6] / s[7];          // np.mean of an empty selection is NaN
        metrics[0] = s[3] / cnt; metrics[1] = (mov + stat) / 2.0; metrics[2] = mov; metrics[3] = stat;
        metrics[4] = s[8] / cnt; metrics[5] = s[9] / cnt; metrics[6] = s[0] / cnt; metrics[7] = s[1] / cnt;
        metrics[8] = s[2] / cnt;
        const double tp = s[10], tn = s[11], fp = s[12], fn = s[13];
        metrics[9] = (tp + tn) / (((tp + tn) + fp) + fn);
        metrics[10] = 0.5 * (tp / (((tp + fp) + fn) + 1e-10) + tn / (((tn + fp) + fn) + 1e-10));
        metrics[11] = tp / ((tp + fn) + 1e-10);
        metrics[12] = s[14] / (double)b; metrics[13] = s[15] / (double)b;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Pseudo labels of the training step (main_util.py:63-67): dyn_mask = extract_dynamic_from_fg (:209-225) and the
// motion-segmentation label mseg_label_RRV (:253-265), merged as where(dyn_mask == 1, mseg, dyn_mask).
// One workgroup per sample (the RRV threshold is relative to the sample's mean residual).
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(EV_THREADS) void pseudo_label_kernel(
    int n, const float *__restrict__ pc1, const float *__restrict__ gt_trans, const float *__restrict__ vel1,
    const float *__restrict__ interval, const float *__restrict__ fg_mask, const float *__restrict__ flow_label,
    float vr_thres, float *__restrict__ dyn_mask, float *__restrict__ mseg_gt, float *__restrict__ residual_out)
{
    extern __shared__ float resid[];                 // [n]
    __shared__ float redf[EV_THREADS / 64];
    const int bs = blockIdx.x, tid = threadIdx.x;
    const float *T = gt_trans + (size_t)bs * 16;
    const float *p = pc1 + (size_t)bs * 3 * n;
    const float dt = interval[bs];
    float sum = 0.f;
    for (int i = tid; i < n; i += EV_THREADS) {
        const float x = p[i], y = p[n + i], z = p[2 * n + i];
        // rigid_to_flow (models/cmflow.py:51-55): (T [p;1])[:3] - p
        const float fx = (((T[0] * x + T[1] * y) + T[2] * z) + T[3]) - x;
        const float fy = (((T[4] * x + T[5] * y) + T[6] * z) + T[7]) - y;
        const float fz = (((T[8] * x + T[9] * y) + T[10] * z) + T[11]) - z;
        const float proj = ((fx * x + fy * y) + fz * z) / sqrtf((x * x + y * y) + z * z);
        const float res = fabsf(vel1[(size_t)bs * n + i] - proj / dt);
        resid[i] = res;
        sum += res;
        // foreground points whose labelled flow equals the rigid flow to 5 cm are static
        const size_t o = ((size_t)bs * n + i) * 3;
        const float m = fg_mask[(size_t)bs * n + i];
        const float fg = (m != 1.f) ? 1.f : 0.f;
        const float ex = (fx - flow_label[o]) * fg, ey = (fy - flow_label[o + 1]) * fg, ez = (fz - flow_label[o + 2]) * fg;
        const bool is_static = sqrtf((ex * ex + ey * ey) + ez * ez) < 0.05f;
        dyn_mask[(size_t)bs * n + i] = (m == 1.f || is_static) ? 1.f : 0.f;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
    if ((tid & 63) == 0) redf[tid >> 6] = sum;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < EV_THREADS / 64; ++w) tot += redf[w];
    const float mean = tot / (float)n;
    for (int i = tid; i < n; i += EV_THREADS) {
        const float rrv = ((resid[i] - mean) < vr_thres) ? 1.f : 0.f;
        const float d = dyn_mask[(size_t)bs * n + i];
        mseg_gt[(size_t)bs * n + i] = (d == 1.f) ? rrv : d;
        if (residual_out) residual_out[(size_t)bs * n + i] = resid[i];
    }
}

extern "C" int cmf_pseudo_labels(int b, int n, const float *pc1, const float *gt_trans, const float *vel1,
                                 const float *interval, const float *fg_mask, const float *flow_label, float vr_thres,
                                 float *dyn_mask, float *mseg_gt, float *residual, void *stream)
{
    CMF_CHECK_ARG(b >= 0 && n > 0 && n <= 36000);
    if (b == 0) return 0;
    CMF_CHECK_ARG(pc1 && gt_trans && vel1 && interval && fg_mask && flow_label && dyn_mask && mseg_gt);
    static CmfPerDevice attr_set;                       // the dynamic-LDS limit is per (function, device)
    int attr_dev;
    if (attr_set.need(attr_dev)) {
        (void)hipFuncSetAttribute((const void *)pseudo_label_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 36000 * 4);
        attr_set.done(attr_dev);
    }
    hipLaunchKernelGGL(pseudo_label_kernel, dim3(b), dim3(EV_THREADS), (size_t)n * sizeof(float), (hipStream_t)stream,
                       n, pc1, gt_trans, vel1, interval, fg_mask, flow_label, vr_thres, dyn_mask, mseg_gt, residual);
    return cmf_launch_status();
}

extern "C" int cmf_eval_metrics(int b, int n, const float *pc, const float *pred, const float *labels, const float *mask,
                                const float *pred_m, const float *gt_trans, const float *pred_trans,
                                float r_res, float theta_res, float phi_res, double *metrics, double *workspace, void *stream)
{
    CMF_CHECK_ARG(b > 0 && n > 0 && metrics && workspace);
    // each of the three groups may be left out (NULL): scene flow (pc, pred, labels, mask), segmentation (pred_m,
    // mask), pose (gt_trans, pred_trans); the metrics of a skipped group are not meaningful
    CMF_CHECK_ARG((!pc || (pred && labels && mask)) && (!pred_m || mask) && (!gt_trans || pred_trans));
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(eval_sample_kernel, dim3(b), dim3(EV_THREADS), 0, st, n, pc, pred, labels, mask, pred_m, gt_trans,
                       pred_trans, (double)r_res, (double)theta_res, (double)phi_res, workspace);
    hipLaunchKernelGGL(eval_finalize_kernel, dim3(1), dim3(64), 0, st, b, n, workspace, metrics);
    return cmf_launch_status();
}
