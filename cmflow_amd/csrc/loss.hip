// RadarFlowLoss (losses/radar_loss.py:260-292, model 'cmflow' / 'cmflow_t') -- the seven cross-modal loss terms
// and their gradients with respect to the network outputs, in three launches:
//
//   loss_count_kernel     : the four batch-wide normalisers (class counts of the pseudo motion-seg label,
//                           sum(1 - mseg_gt), sum(1 - dyn_mask))
//   loss_sample_kernel    : one workgroup per sample.  Everything of a sample lives in LDS (N <= 704 points):
//                           the N x N distance work of SoftChamfer (radar_loss.py:17-58) and SpatialSmoothness
//                           (:60-97) is evaluated on the fly per thread (no (B,N,N) tensors, no top-k pass, no
//                           grouping call), the per-point terms (RadialDisplacement :99-122, EgoMotion :162-183,
//                           MotionSeg :185-205, OpticalFlow :207-243 with utils/util.py:31-58, DynamicFlow :245-258)
//                           ride along; the gradient w.r.t. pred_f / pre_trans / mseg_pre is written directly
//                           (scatter terms as LDS gathers: no atomics, fixed summation order).
//   loss_finalize_kernel  : fixed-order sum of the per-sample partials -> the 8 reported items + the total.
//
// Arithmetic follows the reference's expressions: squared distances in the matmul form of
// radarflow_util.py:8-30, d = max(((-2*dot) + |a|^2) + |b|^2, 0) with dot = fma(az,bz, fma(ay,by, ax*bx)) -- the
// same canonical evaluation as cmf_knn (neighbor.hip); the file is compiled with -ffp-contract=off.
#include "cmf_common.h"
#include "../../include/cmflow_hip.h"

constexpr int LS_THREADS = 256;
constexpr int LS_NB = 8;                    // smoothness neighbours (radar_loss.py:66 num_nb)
constexpr int LS_MAX_N = 704;               // 48 words of LDS per point
constexpr int LS_WORDS_PER_POINT = 48;
constexpr int LS_INV_MAX_N = 512;           // + 10 words per point (inverse list of pass 3) while that fits next to the rest
constexpr int LS_PARTIALS = 8;              // sc, ss, rd, em, ms0, ms1, of, dyn  (per-sample un-normalised sums)

__device__ __forceinline__ float ls_sqnorm3(float x, float y, float z)
{
    const float xx = x * x;
    const float yy = y * y;
    const float zz = z * z;
    const float s = xx + yy;
    return s + zz;
}

// square_distance element (radarflow_util.py:24-29): a = source row, b = destination row
__device__ __forceinline__ float ls_sqdist(float ax, float ay, float az, float aa, float bx, float by, float bz, float bb)
{
    const float p0 = ax * bx;
    const float p01 = __builtin_fmaf(ay, by, p0);
    const float dot = __builtin_fmaf(az, bz, p01);
    const float t = -2.0f * dot;
    const float u = t + aa;
    const float v = u + bb;
    return v > 0.0f ? v : 0.0f;
}

// fixed-order workgroup reductions; red: LS_THREADS/64 floats of LDS.  All threads get the result.
__device__ __forceinline__ float ls_block_sum(float v, float *red)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < LS_THREADS / 64; ++w) s += red[w];
    return s;
}

__device__ __forceinline__ float ls_block_max(float v, float *red)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = red[0];
#pragma unroll
    for (int w = 1; w < LS_THREADS / 64; ++w) s = fmaxf(s, red[w]);
    return s;
}

__global__ __launch_bounds__(1024) void loss_count_kernel(long long total, const float *__restrict__ mseg_gt,
                                                          const float *__restrict__ dyn_mask, float *__restrict__ counts)
{
    __shared__ float red[4][16];
    float c0 = 0.f, c1 = 0.f, om = 0.f, od = 0.f;
    for (long long i = threadIdx.x; i < total; i += 1024) {
        const float m = mseg_gt[i], d = dyn_mask[i];
        c0 += (m == 0.f) ? 1.f : 0.f;
        c1 += (m == 1.f) ? 1.f : 0.f;
        om += 1.f - m;
        od += 1.f - d;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        c0 += __shfl_xor(c0, off, 64); c1 += __shfl_xor(c1, off, 64);
        om += __shfl_xor(om, off, 64); od += __shfl_xor(od, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        const int w = threadIdx.x >> 6;
        red[0][w] = c0; red[1][w] = c1; red[2][w] = om; red[3][w] = od;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        float s = 0.f;
        for (int w = 0; w < 16; ++w) s += red[threadIdx.x][w];
        counts[threadIdx.x] = s;
    }
}

struct LossArgs {
    int B, N;
    const float *pc1, *pc2, *pred_f, *gt_f, *vel1, *mseg_pre, *mseg_gt, *dyn_mask, *radar_u, *radar_v, *opt;
    const float *pre_trans, *gt_trans, *cam_inv, *t_cr;
    float w_self, w_em, w_ms, w_opt, w_dyn, zeta, alpha, lower_bound;
    int self_only;
    int use_inv;                // pass 3 through an inverse list of the pushed gradients (N <= LS_INV_MAX_N)
    const float *counts;
    float *partials, *d_pred_f, *d_pre_trans, *d_mseg_pre;
};

__global__ __launch_bounds__(LS_THREADS) void loss_sample_kernel(const LossArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int N = a.N, tid = threadIdx.x, bs = blockIdx.x;
    float *p1 = sm;                 // [3][N] pc1
    float *p2 = p1 + 3 * N;         // [3][N] pc2
    float *pw = p2 + 3 * N;         // [3][N] pc1 + pred_f
    float *fl = pw + 3 * N;         // [3][N] pred_f
    float *n1 = fl + 3 * N;         // |pc1|^2, |pc2|^2, |pc1w|^2
    float *n2 = n1 + N;
    float *nw = n2 + N;
    int *arg2 = reinterpret_cast<int *>(nw + N);        // [N] argmin over i of d(pc1w_i, pc2_j), -1 if that term is inactive
    int *nbr = arg2 + N;                                // [N][8] smoothness neighbours
    float *gv = reinterpret_cast<float *>(nbr + LS_NB * N);   // [N][8][3] pair gradients (pushed to the neighbour)
    __shared__ float red[LS_THREADS / 64];
    __shared__ float sT[32];                            // pre_trans, gt_trans
    __shared__ float sC[9 + 16];                        // camera_inverse, t_camera_radar

    const size_t o3 = (size_t)bs * 3 * N, o1 = (size_t)bs * N;
    for (int i = tid; i < 3 * N; i += LS_THREADS) {
        const float x = a.pc1[o3 + i], f = a.pred_f[o3 + i];
        p1[i] = x; p2[i] = a.pc2[o3 + i]; fl[i] = f; pw[i] = x + f;
    }
    if (!a.self_only) {
        if (tid < 16) { sT[tid] = a.pre_trans[(size_t)bs * 16 + tid]; sT[16 + tid] = a.gt_trans[(size_t)bs * 16 + tid]; }
        if (tid < 9) sC[tid] = a.cam_inv[tid];
        if (tid >= 32 && tid < 48) sC[9 + tid - 32] = a.t_cr[tid - 32];
    }
    __syncthreads();
    for (int i = tid; i < N; i += LS_THREADS) {
        n1[i] = ls_sqnorm3(p1[i], p1[N + i], p1[2 * N + i]);
        n2[i] = ls_sqnorm3(p2[i], p2[N + i], p2[2 * N + i]);
        nw[i] = ls_sqnorm3(pw[i], pw[N + i], pw[2 * N + i]);
    }
    __syncthreads();

    const float inv_bn = 1.0f / ((float)a.B * (float)N);
    const float cnt0 = a.self_only ? 1.f : a.counts[0], cnt1 = a.self_only ? 1.f : a.counts[1];
    const float den_of = a.self_only ? 1.f : fmaxf(a.counts[2], 1.0f), den_dyn = a.self_only ? 1.f : fmaxf(a.counts[3], 1.0f);
    float part[LS_PARTIALS];
#pragma unroll
    for (int t = 0; t < LS_PARTIALS; ++t) part[t] = 0.f;
    float emax_local = 0.f;                             // exp(-d/alpha) > 0
    float et[12];                                       // d total / d pre_trans[:3,:4] partial sums
#pragma unroll
    for (int t = 0; t < 12; ++t) et[t] = 0.f;

    // ---------------- pass 1: the N x N work of this thread's points ----------------
    for (int i = tid; i < N; i += LS_THREADS) {
        const float ax = p1[i], ay = p1[N + i], az = p1[2 * N + i], aa = n1[i];
        const float wx = pw[i], wy = pw[N + i], wz = pw[2 * N + i], ww = nw[i];
        const float bx = p2[i], by = p2[N + i], bz = p2[2 * N + i], bb = n2[i];
        float bd[LS_NB + 1];
        int bi[LS_NB + 1];
#pragma unroll
        for (int t = 0; t <= LS_NB; ++t) { bd[t] = __builtin_inff(); bi[t] = 0; }
        float dens1 = 0.f, dens2 = 0.f, min1 = __builtin_inff(), min2 = __builtin_inff();
        int am1 = 0, am2 = 0;
        for (int j = 0; j < N; ++j) {
            const float qx = p1[j], qy = p1[N + j], qz = p1[2 * N + j], qq = n1[j];
            const float rx = p2[j], ry = p2[N + j], rz = p2[2 * N + j], rr = n2[j];
            // smoothness: top-(8+1) of square_distance(pc1, pc1)[i, :], ascending, ties -> lowest index
            const float d11 = ls_sqdist(ax, ay, az, aa, qx, qy, qz, qq);
            if (d11 < bd[LS_NB]) {
                bd[LS_NB] = d11; bi[LS_NB] = j;
#pragma unroll
                for (int t = LS_NB; t > 0; --t)
                    if (bd[t] < bd[t - 1]) {
                        const float td = bd[t]; bd[t] = bd[t - 1]; bd[t - 1] = td;
                        const int ti = bi[t]; bi[t] = bi[t - 1]; bi[t - 1] = ti;
                    }
            }
            // chamfer, this thread as pc1 point i: density of pc1_i in pc2, nearest pc2 point of the warped pc1_i
            const float d12 = ls_sqdist(ax, ay, az, aa, rx, ry, rz, rr);
            dens1 += expf(-d12 / 2.0f) / 2.5f;
            const float dw = ls_sqdist(wx, wy, wz, ww, rx, ry, rz, rr);
            if (dw < min1) { min1 = dw; am1 = j; }
            // chamfer, this thread as pc2 point i: density of pc2_i in pc1, nearest warped pc1 point
            const float d21 = ls_sqdist(bx, by, bz, bb, qx, qy, qz, qq);
            dens2 += expf(-d21 / 2.0f) / 2.5f;
            const float dwt = ls_sqdist(pw[j], pw[N + j], pw[2 * N + j], nw[j], bx, by, bz, bb);
            if (dwt < min2) { min2 = dwt; am2 = j; }
        }
        const bool mask1 = dens1 / (float)N > a.zeta, mask2 = dens2 / (float)N > a.zeta;
        const float r1 = min1 - 0.01f, r2 = min2 - 0.01f;
        if (mask1 && r1 > 0.f) part[0] += r1;
        if (mask2 && r2 > 0.f) part[0] += r2;
        // reuse: bd[0]/bi[0] (the nearest = the point itself or a duplicate) is dropped (radar_loss.py:86-87)
        arg2[i] = (mask2 && r2 > 0.f) ? am2 : -1;
        // keep what pass 2 needs in LDS/regs: neighbour list, e = exp(-d/alpha)
#pragma unroll
        for (int t = 0; t < LS_NB; ++t) {
            nbr[i * LS_NB + t] = bi[t + 1];
            const float e = expf(-bd[t + 1] / a.alpha);
            gv[(i * LS_NB + t) * 3] = e;                // parked until the soft-max normaliser is known
            emax_local = fmaxf(emax_local, e);
        }
        // chamfer gradient of term 1 (own nearest neighbour) goes to this point: parked in gv? no: written below
        gv[(i * LS_NB) * 3 + 1] = (mask1 && r1 > 0.f) ? (float)am1 : -1.0f;
    }
    const float emax = ls_block_max(emax_local, red);
    float zsum_local = 0.f;
    for (int i = tid; i < N; i += LS_THREADS)
#pragma unroll
        for (int t = 0; t < LS_NB; ++t) zsum_local += expf(gv[(i * LS_NB + t) * 3] - emax);
    const float zsum = ls_block_sum(zsum_local, red);

    // ---------------- pass 2: per-point terms, own-point gradients, pair gradients ----------------
    // gradient accumulators of up to LS_MAX_N/LS_THREADS = 3 points per thread stay in registers
    constexpr int PT = (LS_MAX_N + LS_THREADS - 1) / LS_THREADS;
    float gx[PT], gy[PT], gz[PT];
#pragma unroll
    for (int q = 0; q < PT; ++q) { gx[q] = gy[q] = gz[q] = 0.f; }
    const float k_self = a.w_self * inv_bn;
#pragma unroll
    for (int q = 0; q < PT; ++q) {
        const int i = tid + q * LS_THREADS;
        if (i >= N) continue;
        const float ax = p1[i], ay = p1[N + i], az = p1[2 * N + i];
        const float fx = fl[i], fy = fl[N + i], fz = fl[2 * N + i];
        const float wx = pw[i], wy = pw[N + i], wz = pw[2 * N + i];
        // chamfer term 1
        const int am1 = (int)gv[(i * LS_NB) * 3 + 1];
        if (am1 >= 0) {
            gx[q] += k_self * 2.0f * (wx - p2[am1]);
            gy[q] += k_self * 2.0f * (wy - p2[N + am1]);
            gz[q] += k_self * 2.0f * (wz - p2[2 * N + am1]);
        }
        // smoothness (radar_loss.py:88-96): weights = softmax over the sample's N*8 values exp(-d/alpha)
        float ss_i = 0.f;
#pragma unroll
        for (int t = 0; t < LS_NB; ++t) {
            const int j = nbr[i * LS_NB + t];
            const float w = expf(gv[(i * LS_NB + t) * 3] - emax) / zsum;
            const float dx = fl[j] - fx, dy = fl[N + j] - fy, dz = fl[2 * N + j] - fz;
            const float nrm = sqrtf(ls_sqnorm3(dx, dy, dz));
            const float nw_ = (float)N * w;
            ss_i += nw_ * nrm;
            const float s = nrm > 0.f ? k_self * nw_ / nrm : 0.f;       // torch.norm backward: 0 at the origin
            const float vx = s * dx, vy = s * dy, vz = s * dz;           // d/d f_j ; d/d f_i is the negative
            gx[q] -= vx; gy[q] -= vy; gz[q] -= vz;
            gv[(i * LS_NB + t) * 3] = vx; gv[(i * LS_NB + t) * 3 + 1] = vy; gv[(i * LS_NB + t) * 3 + 2] = vz;
        }
        part[1] += ss_i;
        // radial displacement (radar_loss.py:99-122): |v_r * 0.1 - <f, p>/|p||
        const float pn = sqrtf(ls_sqnorm3(ax, ay, az));
        const float fr = ((fx * ax + fy * ay) + fz * az) / pn;
        const float rdv = a.vel1[o1 + i] * 0.1f - fr;
        part[2] += fabsf(rdv);
        const float sg = rdv > 0.f ? -1.f : (rdv < 0.f ? 1.f : 0.f);
        gx[q] += k_self * sg * ax / pn; gy[q] += k_self * sg * ay / pn; gz[q] += k_self * sg * az / pn;
        if (a.self_only) {                                   // RaFlow: the self-supervised terms are the whole loss
            if (a.d_mseg_pre) a.d_mseg_pre[o1 + i] = 0.f;
            continue;
        }
        // ego-motion (radar_loss.py:162-183): |(R p + t) - (R_gt p + t_gt)|
        float e3[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float pre = ((sT[4 * r] * ax + sT[4 * r + 1] * ay) + sT[4 * r + 2] * az) + sT[4 * r + 3];
            const float gt = ((sT[16 + 4 * r] * ax + sT[16 + 4 * r + 1] * ay) + sT[16 + 4 * r + 2] * az) + sT[16 + 4 * r + 3];
            e3[r] = pre - gt;
        }
        const float en = sqrtf(ls_sqnorm3(e3[0], e3[1], e3[2]));
        part[3] += en;
        if (en > 0.f) {
            const float ke = a.w_em * inv_bn / en;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                et[4 * r] += ke * e3[r] * ax; et[4 * r + 1] += ke * e3[r] * ay;
                et[4 * r + 2] += ke * e3[r] * az; et[4 * r + 3] += ke * e3[r];
            }
        }
        // motion segmentation (radar_loss.py:185-205): BCE averaged separately over the two classes
        const float p = a.mseg_pre[o1 + i], y = a.mseg_gt[o1 + i];
        const float bce = -(y * fmaxf(logf(p), -100.f) + (1.f - y) * fmaxf(logf(1.f - p), -100.f));
        float dms = 0.f;
        if (y == 0.f) { part[4] += bce; dms = 0.5f / cnt0; }
        else if (y == 1.f) { part[5] += bce; dms = 0.5f / cnt1; }
        if (a.d_mseg_pre) a.d_mseg_pre[o1 + i] = a.w_ms * dms * (p - y) / fmaxf((1.f - p) * p, 1e-12f);
        // optical flow (radar_loss.py:207-243, utils/util.py:31-58): distance of the warped point to the pixel ray
        const float u = a.radar_u[o1 + i] + a.opt[(o1 + i) * 2], v = a.radar_v[o1 + i] + a.opt[(o1 + i) * 2 + 1];
        float ray[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) ray[r] = (sC[3 * r] * u + sC[3 * r + 1] * v) + sC[3 * r + 2];
        const float rn = sqrtf(ls_sqnorm3(ray[0], ray[1], ray[2]));
        const float ux = ray[0] / rn, uy = ray[1] / rn, uz = ray[2] / rn;
        const float *T = sC + 9;
        float wc[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) wc[r] = ((T[4 * r] * wx + T[4 * r + 1] * wy) + T[4 * r + 2] * wz) + T[4 * r + 3];
        const float cx = uy * wc[2] - uz * wc[1], cy = uz * wc[0] - ux * wc[2], cz = ux * wc[1] - uy * wc[0];
        const float cn = sqrtf(ls_sqnorm3(cx, cy, cz));
        const float om = 1.f - y;
        const float div = cn - a.lower_bound;
        if (div > 0.f) {
            part[6] += om * div;
            if (cn > 0.f) {
                const float ko = a.w_opt * om / den_of / cn;
                // d|u x w| / dw = c_hat x u ; back to the radar frame through R_cr^T
                const float hx = cy * uz - cz * uy, hy = cz * ux - cx * uz, hz = cx * uy - cy * ux;
                gx[q] += ko * ((T[0] * hx + T[4] * hy) + T[8] * hz);
                gy[q] += ko * ((T[1] * hx + T[5] * hy) + T[9] * hz);
                gz[q] += ko * ((T[2] * hx + T[6] * hy) + T[10] * hz);
            }
        }
        // dynamic flow (radar_loss.py:245-258)
        const float od = 1.f - a.dyn_mask[o1 + i];
        const float ex = a.gt_f[o3 + i] - fx, ey = a.gt_f[o3 + N + i] - fy, ez = a.gt_f[o3 + 2 * N + i] - fz;
        const float dn = sqrtf(ls_sqnorm3(ex, ey, ez));
        part[7] += od * dn;
        if (dn > 0.f) {
            const float kd = a.w_dyn * od / den_dyn / dn;
            gx[q] -= kd * ex; gy[q] -= kd * ey; gz[q] -= kd * ez;
        }
    }
    __syncthreads();
    // ---------------- pass 3: gather the gradients other points push onto this one ----------------
    // Every point scanning all N + 8 N pushes for its own (2304 LDS reads per thread at N = 256) was half of the kernel's
    // 280 us.  Instead the pushes are binned by target -- id j < N: chamfer pair of pc2_j, id N + e: smoothness pair e --
    // with a counting sort whose bins are then sorted by id, so a point adds its ~9 pushes in exactly the order the scan
    // visited them (chamfer j ascending, then pairs e ascending): bit-identical sums.
    int *inv_lst = reinterpret_cast<int *>(gv + 3 * LS_NB * N);       // [9 N] ids binned by target
    int *inv_end = inv_lst + (LS_NB + 1) * N;                           // [N + 1] bin ends (cursor during the fill)
    if (a.use_inv) {
        for (int i = tid; i <= N; i += LS_THREADS) inv_end[i] = 0;
        __syncthreads();
        for (int j = tid; j < N; j += LS_THREADS) if (arg2[j] >= 0) atomicAdd(&inv_end[arg2[j] + 1], 1);
        for (int e = tid; e < N * LS_NB; e += LS_THREADS) atomicAdd(&inv_end[nbr[e] + 1], 1);
        __syncthreads();
        {   // inclusive scan of inv_end[1 .. N] (N <= 512: two entries per thread), wave scan + wave totals
            __shared__ int wsum[LS_THREADS / 64];
            const int i0 = 1 + 2 * tid;
            const int v0 = i0 <= N ? inv_end[i0] : 0, v1 = i0 + 1 <= N ? inv_end[i0 + 1] : 0;
            int incl = v0 + v1;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(incl, off, 64); if ((tid & 63) >= off) incl += t; }
            if ((tid & 63) == 63) wsum[tid >> 6] = incl;
            __syncthreads();
            int before = incl - (v0 + v1);
            for (int w = 0; w < (tid >> 6); ++w) before += wsum[w];
            // inv_end[i] becomes the START of bin i - 1 ... shifted: after this, inv_end[t + 1] = start of bin t's successor;
            // stored as cursor: inv_end[i0 - 1 + 1] -- keep it simple: write exclusive starts into place i (bin i - 1 starts at inv_end[i - 1])
            if (i0 <= N) inv_end[i0] = before + v0;
            if (i0 + 1 <= N) inv_end[i0 + 1] = before + v0 + v1;
        }
        __syncthreads();
        // now inv_end[t + 1] = end of bin t and inv_end[t] = its start; fill with a cursor per bin kept in the bin's START slot:
        // the cursor of bin t is inv_end[t] and finishes at inv_end[t + 1]'s value, so afterwards inv_end[t] == end of bin t
        for (int j = tid; j < N; j += LS_THREADS)
            if (arg2[j] >= 0) inv_lst[atomicAdd(&inv_end[arg2[j]], 1)] = j;
        for (int e = tid; e < N * LS_NB; e += LS_THREADS) inv_lst[atomicAdd(&inv_end[nbr[e]], 1)] = N + e;
        __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < PT; ++q) {
        const int i = tid + q * LS_THREADS;
        if (i >= N) continue;
        const float wx = pw[i], wy = pw[N + i], wz = pw[2 * N + i];
        if (a.use_inv) {
            const int s0 = i > 0 ? inv_end[i - 1] : 0, s1 = inv_end[i];       // bin i (ends double as the next bin's start)
            for (int u = s0 + 1; u < s1; ++u) {                               // insertion sort by id (bins hold ~9 entries)
                const int v = inv_lst[u];
                int w = u - 1;
                while (w >= s0 && inv_lst[w] > v) { inv_lst[w + 1] = inv_lst[w]; --w; }
                inv_lst[w + 1] = v;
            }
            for (int u = s0; u < s1; ++u) {
                const int id = inv_lst[u];
                if (id < N) {
                    gx[q] += k_self * 2.0f * (wx - p2[id]);
                    gy[q] += k_self * 2.0f * (wy - p2[N + id]);
                    gz[q] += k_self * 2.0f * (wz - p2[2 * N + id]);
                } else {
                    const int e = id - N;
                    gx[q] += gv[e * 3]; gy[q] += gv[e * 3 + 1]; gz[q] += gv[e * 3 + 2];
                }
            }
        } else {
            for (int j = 0; j < N; ++j)                       // chamfer term 2: pc2_j whose nearest warped point is i
                if (arg2[j] == i) {
                    gx[q] += k_self * 2.0f * (wx - p2[j]);
                    gy[q] += k_self * 2.0f * (wy - p2[N + j]);
                    gz[q] += k_self * 2.0f * (wz - p2[2 * N + j]);
                }
            for (int e = 0; e < N * LS_NB; ++e)               // smoothness pairs (i', k) whose neighbour is i
                if (nbr[e] == i) { gx[q] += gv[e * 3]; gy[q] += gv[e * 3 + 1]; gz[q] += gv[e * 3 + 2]; }
        }
        if (a.d_pred_f) { a.d_pred_f[o3 + i] = gx[q]; a.d_pred_f[o3 + N + i] = gy[q]; a.d_pred_f[o3 + 2 * N + i] = gz[q]; }
    }
    // ---------------- per-sample partial sums ----------------
#pragma unroll
    for (int t = 0; t < LS_PARTIALS; ++t) {
        const float s = ls_block_sum(part[t], red);
        if (tid == 0) a.partials[(size_t)bs * LS_PARTIALS + t] = s;
    }
    if (a.d_pre_trans) {
#pragma unroll
        for (int t = 0; t < 12; ++t) {
            const float s = ls_block_sum(et[t], red);
            if (tid == 0) a.d_pre_trans[(size_t)bs * 16 + t] = s;
        }
        if (tid < 4) a.d_pre_trans[(size_t)bs * 16 + 12 + tid] = 0.f;
    }
}

// items: [0] total, [1] Loss (self-supervised sum), [2] smoothnessLoss, [3] chamferLoss, [4] veloLoss, [5] egoLoss,
//        [6] maskLoss, [7] opticalLoss, [8] superviseLoss     (radar_loss.py:285-288)
__global__ __launch_bounds__(64) void loss_finalize_kernel(const LossArgs a, float *__restrict__ items)
{
    const int t = threadIdx.x;
    __shared__ float s[LS_PARTIALS];
    if (t < LS_PARTIALS) {
        float acc = 0.f;
        for (int b = 0; b < a.B; ++b) acc += a.partials[(size_t)b * LS_PARTIALS + t];
        s[t] = acc;
    }
    __syncthreads();
    if (t == 0) {
        const float inv_bn = 1.0f / ((float)a.B * (float)a.N);
        const float sc = s[0] * inv_bn, ss = s[1] * inv_bn, rd = s[2] * inv_bn, em = s[3] * inv_bn;
        const float self_sup = (sc + ss) + rd;
        if (a.self_only) {
            items[0] = a.w_self * self_sup;
            items[1] = self_sup; items[2] = ss; items[3] = sc; items[4] = rd;
            items[5] = items[6] = items[7] = items[8] = 0.f;
            return;
        }
        const float ms = (s[4] / a.counts[0] + s[5] / a.counts[1]) / 2.f;
        const float of = s[6] / fmaxf(a.counts[2], 1.0f), dyn = s[7] / fmaxf(a.counts[3], 1.0f);
        items[0] = (((a.w_self * self_sup + a.w_em * em) + a.w_ms * ms) + a.w_opt * of) + a.w_dyn * dyn;
        items[1] = self_sup; items[2] = ss; items[3] = sc; items[4] = rd; items[5] = em; items[6] = ms;
        items[7] = of; items[8] = dyn;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Tiled form for clouds that do not fit one workgroup's LDS (N > 704) or another neighbour count (num_nb in {4, 8, 16};
// losses/radar_loss.py has no size limit).  The same terms in the same arithmetic, a sample's working set in a global
// workspace instead of LDS:
//   loss_nn_kernel<NB>          : the N x N work of pass 1, 256 points per workgroup, the cloud streamed through LDS in
//                                 tiles of 256 points (neighbour lists, exp(-d/alpha), nearest-neighbour arguments ->
//                                 workspace; chamfer partial sum and max exp per workgroup)
//   loss_big_sample_kernel<NB>  : one 1024-thread workgroup per sample: soft-max normaliser, pass 2 (per-point terms, own
//                                 gradients, pair gradients), the inverse lists and pass 3 (pushed gradients in id order:
//                                 no atomics in any floating-point sum), the per-sample partial sums
// followed by loss_finalize_kernel as above.  Per-point gradients accumulate their terms in the order of
// loss_sample_kernel; the partial sums are folded in another (fixed) association.
// ---------------------------------------------------------------------------------------------------------------
constexpr int LG_THREADS = 1024;
constexpr int LG_TILE = 256;

struct LossBigWs { int *nbr; float *ev, *gv; int *am1, *arg2; float *g; int *inv_lst, *inv_end; float *wgp; };

__host__ __device__ inline size_t lg_sample_floats(int N, int NB)
{
    const size_t nwg = (size_t)(N + LG_TILE - 1) / LG_TILE;
    const size_t n = (size_t)N * (6 * NB + 7) + 4 + 2 * nwg;
    return (n + 3) / 4 * 4;
}

__host__ __device__ inline LossBigWs lg_carve(float *base, int N, int NB)
{
    LossBigWs w;
    float *p = base;
    w.nbr = reinterpret_cast<int *>(p); p += (size_t)N * NB;
    w.ev = p; p += (size_t)N * NB;
    w.gv = p; p += (size_t)N * NB * 3;
    w.am1 = reinterpret_cast<int *>(p); p += N;
    w.arg2 = reinterpret_cast<int *>(p); p += N;
    w.g = p; p += 3 * (size_t)N;
    w.inv_lst = reinterpret_cast<int *>(p); p += (size_t)N * (NB + 1);
    w.inv_end = reinterpret_cast<int *>(p); p += N + 4;
    w.wgp = p;
    return w;
}

template <int T>
__device__ __forceinline__ float lg_block_sum(float v, float *red)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < T / 64; ++w) s += red[w];
    return s;
}

template <int NB>
__global__ __launch_bounds__(LG_TILE) void loss_nn_kernel(const LossArgs a, float *__restrict__ ws_base, size_t ws_stride)
{
    __shared__ float t[12][LG_TILE];                    // pc1 xyz |.|^2, pc2 xyz |.|^2, pc1 + flow xyz |.|^2 of the tile
    __shared__ float red[LG_TILE / 64];
    const int N = a.N, tid = threadIdx.x, bs = blockIdx.y;
    const int i = blockIdx.x * LG_TILE + tid;
    const bool live = i < N;
    const int ic = live ? i : N - 1;
    const size_t o3 = (size_t)bs * 3 * N;
    const LossBigWs w = lg_carve(ws_base + (size_t)bs * ws_stride, N, NB);
    const float ax = a.pc1[o3 + ic], ay = a.pc1[o3 + N + ic], az = a.pc1[o3 + 2 * N + ic];
    const float wx = ax + a.pred_f[o3 + ic], wy = ay + a.pred_f[o3 + N + ic], wz = az + a.pred_f[o3 + 2 * N + ic];
    const float bx = a.pc2[o3 + ic], by = a.pc2[o3 + N + ic], bz = a.pc2[o3 + 2 * N + ic];
    const float aa = ls_sqnorm3(ax, ay, az), ww = ls_sqnorm3(wx, wy, wz), bb = ls_sqnorm3(bx, by, bz);
    float bd[NB + 1];
    int bi[NB + 1];
#pragma unroll
    for (int q = 0; q <= NB; ++q) { bd[q] = __builtin_inff(); bi[q] = 0; }
    float dens1 = 0.f, dens2 = 0.f, min1 = __builtin_inff(), min2 = __builtin_inff();
    int am1 = 0, am2 = 0;
    for (int j0 = 0; j0 < N; j0 += LG_TILE) {
        __syncthreads();
        const int jl = j0 + tid;
        if (jl < N) {
            const float x1 = a.pc1[o3 + jl], y1 = a.pc1[o3 + N + jl], z1 = a.pc1[o3 + 2 * N + jl];
            const float x2 = a.pc2[o3 + jl], y2 = a.pc2[o3 + N + jl], z2 = a.pc2[o3 + 2 * N + jl];
            const float xw = x1 + a.pred_f[o3 + jl], yw = y1 + a.pred_f[o3 + N + jl], zw = z1 + a.pred_f[o3 + 2 * N + jl];
            t[0][tid] = x1; t[1][tid] = y1; t[2][tid] = z1; t[3][tid] = ls_sqnorm3(x1, y1, z1);
            t[4][tid] = x2; t[5][tid] = y2; t[6][tid] = z2; t[7][tid] = ls_sqnorm3(x2, y2, z2);
            t[8][tid] = xw; t[9][tid] = yw; t[10][tid] = zw; t[11][tid] = ls_sqnorm3(xw, yw, zw);
        }
        __syncthreads();
        const int nj = min(LG_TILE, N - j0);
        if (live)
            for (int jj = 0; jj < nj; ++jj) {
                const int j = j0 + jj;
                const float qx = t[0][jj], qy = t[1][jj], qz = t[2][jj], qq = t[3][jj];
                const float rx = t[4][jj], ry = t[5][jj], rz = t[6][jj], rr = t[7][jj];
                const float d11 = ls_sqdist(ax, ay, az, aa, qx, qy, qz, qq);
                if (d11 < bd[NB]) {
                    bd[NB] = d11; bi[NB] = j;
#pragma unroll
                    for (int q = NB; q > 0; --q)
                        if (bd[q] < bd[q - 1]) {
                            const float td = bd[q]; bd[q] = bd[q - 1]; bd[q - 1] = td;
                            const int ti = bi[q]; bi[q] = bi[q - 1]; bi[q - 1] = ti;
                        }
                }
                const float d12 = ls_sqdist(ax, ay, az, aa, rx, ry, rz, rr);
                dens1 += expf(-d12 / 2.0f) / 2.5f;
                const float dw = ls_sqdist(wx, wy, wz, ww, rx, ry, rz, rr);
                if (dw < min1) { min1 = dw; am1 = j; }
                const float d21 = ls_sqdist(bx, by, bz, bb, qx, qy, qz, qq);
                dens2 += expf(-d21 / 2.0f) / 2.5f;
                const float dwt = ls_sqdist(t[8][jj], t[9][jj], t[10][jj], t[11][jj], bx, by, bz, bb);
                if (dwt < min2) { min2 = dwt; am2 = j; }
            }
    }
    float part0 = 0.f, emax_local = 0.f;
    if (live) {
        const bool mask1 = dens1 / (float)N > a.zeta, mask2 = dens2 / (float)N > a.zeta;
        const float r1 = min1 - 0.01f, r2 = min2 - 0.01f;
        if (mask1 && r1 > 0.f) part0 += r1;
        if (mask2 && r2 > 0.f) part0 += r2;
        w.arg2[i] = (mask2 && r2 > 0.f) ? am2 : -1;
        w.am1[i] = (mask1 && r1 > 0.f) ? am1 : -1;
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            w.nbr[(size_t)i * NB + q] = bi[q + 1];
            const float e = expf(-bd[q + 1] / a.alpha);
            w.ev[(size_t)i * NB + q] = e;
            emax_local = fmaxf(emax_local, e);
        }
    }
    const float ps = ls_block_sum(part0, red);
    const float em = ls_block_max(emax_local, red);
    if (tid == 0) { w.wgp[2 * blockIdx.x] = ps; w.wgp[2 * blockIdx.x + 1] = em; }
}

template <int NB>
__global__ __launch_bounds__(LG_THREADS) void loss_big_sample_kernel(const LossArgs a, float *__restrict__ ws_base, size_t ws_stride)
{
    __shared__ float red[LG_THREADS / 64];
    __shared__ int wsum[LG_THREADS / 64];
    __shared__ float sT[32];
    __shared__ float sC[9 + 16];
    const int N = a.N, tid = threadIdx.x, bs = blockIdx.x;
    const int nwg = (N + LG_TILE - 1) / LG_TILE;
    const LossBigWs w = lg_carve(ws_base + (size_t)bs * ws_stride, N, NB);
    const size_t o3 = (size_t)bs * 3 * N, o1 = (size_t)bs * N;
    const float *p1 = a.pc1 + o3, *p2 = a.pc2 + o3, *fl = a.pred_f + o3;
    if (!a.self_only) {
        if (tid < 16) { sT[tid] = a.pre_trans[(size_t)bs * 16 + tid]; sT[16 + tid] = a.gt_trans[(size_t)bs * 16 + tid]; }
        if (tid < 9) sC[tid] = a.cam_inv[tid];
        if (tid >= 32 && tid < 48) sC[9 + tid - 32] = a.t_cr[tid - 32];
    }
    __syncthreads();
    const float inv_bn = 1.0f / ((float)a.B * (float)N);
    const float cnt0 = a.self_only ? 1.f : a.counts[0], cnt1 = a.self_only ? 1.f : a.counts[1];
    const float den_of = a.self_only ? 1.f : fmaxf(a.counts[2], 1.0f), den_dyn = a.self_only ? 1.f : fmaxf(a.counts[3], 1.0f);
    float part[LS_PARTIALS];
#pragma unroll
    for (int q = 0; q < LS_PARTIALS; ++q) part[q] = 0.f;
    float et[12];
#pragma unroll
    for (int q = 0; q < 12; ++q) et[q] = 0.f;
    // soft-max normaliser of the sample's N * NB values exp(-d / alpha)
    float emax = 0.f;
    for (int g = 0; g < nwg; ++g) emax = fmaxf(emax, w.wgp[2 * g + 1]);
    float zl = 0.f;
    for (int e = tid; e < N * NB; e += LG_THREADS) zl += expf(w.ev[e] - emax);
    const float zsum = lg_block_sum<LG_THREADS>(zl, red);
    const float k_self = a.w_self * inv_bn;
    // ---------------- pass 2 ----------------
    for (int i = tid; i < N; i += LG_THREADS) {
        float gx = 0.f, gy = 0.f, gz = 0.f;
        const float ax = p1[i], ay = p1[N + i], az = p1[2 * N + i];
        const float fx = fl[i], fy = fl[N + i], fz = fl[2 * N + i];
        const float wx = ax + fx, wy = ay + fy, wz = az + fz;
        const int am1 = w.am1[i];
        if (am1 >= 0) {
            gx += k_self * 2.0f * (wx - p2[am1]);
            gy += k_self * 2.0f * (wy - p2[N + am1]);
            gz += k_self * 2.0f * (wz - p2[2 * N + am1]);
        }
        float ss_i = 0.f;
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            const size_t e = (size_t)i * NB + q;
            const int j = w.nbr[e];
            const float sw = expf(w.ev[e] - emax) / zsum;
            const float dx = fl[j] - fx, dy = fl[N + j] - fy, dz = fl[2 * N + j] - fz;
            const float nrm = sqrtf(ls_sqnorm3(dx, dy, dz));
            const float nw_ = (float)N * sw;
            ss_i += nw_ * nrm;
            const float s = nrm > 0.f ? k_self * nw_ / nrm : 0.f;
            const float vx = s * dx, vy = s * dy, vz = s * dz;
            gx -= vx; gy -= vy; gz -= vz;
            w.gv[e * 3] = vx; w.gv[e * 3 + 1] = vy; w.gv[e * 3 + 2] = vz;
        }
        part[1] += ss_i;
        const float pn = sqrtf(ls_sqnorm3(ax, ay, az));
        const float fr = ((fx * ax + fy * ay) + fz * az) / pn;
        const float rdv = a.vel1[o1 + i] * 0.1f - fr;
        part[2] += fabsf(rdv);
        const float sg = rdv > 0.f ? -1.f : (rdv < 0.f ? 1.f : 0.f);
        gx += k_self * sg * ax / pn; gy += k_self * sg * ay / pn; gz += k_self * sg * az / pn;
        if (a.self_only) {
            if (a.d_mseg_pre) a.d_mseg_pre[o1 + i] = 0.f;
        } else {
            float e3[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const float pre = ((sT[4 * r] * ax + sT[4 * r + 1] * ay) + sT[4 * r + 2] * az) + sT[4 * r + 3];
                const float gt = ((sT[16 + 4 * r] * ax + sT[16 + 4 * r + 1] * ay) + sT[16 + 4 * r + 2] * az) + sT[16 + 4 * r + 3];
                e3[r] = pre - gt;
            }
            const float en = sqrtf(ls_sqnorm3(e3[0], e3[1], e3[2]));
            part[3] += en;
            if (en > 0.f) {
                const float ke = a.w_em * inv_bn / en;
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    et[4 * r] += ke * e3[r] * ax; et[4 * r + 1] += ke * e3[r] * ay;
                    et[4 * r + 2] += ke * e3[r] * az; et[4 * r + 3] += ke * e3[r];
                }
            }
            const float p = a.mseg_pre[o1 + i], y = a.mseg_gt[o1 + i];
            const float bce = -(y * fmaxf(logf(p), -100.f) + (1.f - y) * fmaxf(logf(1.f - p), -100.f));
            float dms = 0.f;
            if (y == 0.f) { part[4] += bce; dms = 0.5f / cnt0; }
            else if (y == 1.f) { part[5] += bce; dms = 0.5f / cnt1; }
            if (a.d_mseg_pre) a.d_mseg_pre[o1 + i] = a.w_ms * dms * (p - y) / fmaxf((1.f - p) * p, 1e-12f);
            const float u = a.radar_u[o1 + i] + a.opt[(o1 + i) * 2], v = a.radar_v[o1 + i] + a.opt[(o1 + i) * 2 + 1];
            float ray[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) ray[r] = (sC[3 * r] * u + sC[3 * r + 1] * v) + sC[3 * r + 2];
            const float rn = sqrtf(ls_sqnorm3(ray[0], ray[1], ray[2]));
            const float ux = ray[0] / rn, uy = ray[1] / rn, uz = ray[2] / rn;
            const float *T = sC + 9;
            float wc[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) wc[r] = ((T[4 * r] * wx + T[4 * r + 1] * wy) + T[4 * r + 2] * wz) + T[4 * r + 3];
            const float cx = uy * wc[2] - uz * wc[1], cy = uz * wc[0] - ux * wc[2], cz = ux * wc[1] - uy * wc[0];
            const float cn = sqrtf(ls_sqnorm3(cx, cy, cz));
            const float om = 1.f - y;
            const float div = cn - a.lower_bound;
            if (div > 0.f) {
                part[6] += om * div;
                if (cn > 0.f) {
                    const float ko = a.w_opt * om / den_of / cn;
                    const float hx = cy * uz - cz * uy, hy = cz * ux - cx * uz, hz = cx * uy - cy * ux;
                    gx += ko * ((T[0] * hx + T[4] * hy) + T[8] * hz);
                    gy += ko * ((T[1] * hx + T[5] * hy) + T[9] * hz);
                    gz += ko * ((T[2] * hx + T[6] * hy) + T[10] * hz);
                }
            }
            const float od = 1.f - a.dyn_mask[o1 + i];
            const float ex = a.gt_f[o3 + i] - fx, ey = a.gt_f[o3 + N + i] - fy, ez = a.gt_f[o3 + 2 * N + i] - fz;
            const float dn = sqrtf(ls_sqnorm3(ex, ey, ez));
            part[7] += od * dn;
            if (dn > 0.f) {
                const float kd = a.w_dyn * od / den_dyn / dn;
                gx -= kd * ex; gy -= kd * ey; gz -= kd * ez;
            }
        }
        w.g[i] = gx; w.g[N + i] = gy; w.g[2 * N + i] = gz;
    }
    __syncthreads();
    // ---------------- pass 3: pushes binned by target, each bin sorted by id (the order a scan would visit them) ----------------
    if (a.d_pred_f) {
        for (int i = tid; i <= N; i += LG_THREADS) w.inv_end[i] = 0;
        __syncthreads();
        for (int j = tid; j < N; j += LG_THREADS) if (w.arg2[j] >= 0) atomicAdd(&w.inv_end[w.arg2[j] + 1], 1);
        for (int e = tid; e < N * NB; e += LG_THREADS) atomicAdd(&w.inv_end[w.nbr[e] + 1], 1);
        __syncthreads();
        {   // inclusive scan of inv_end[1 .. N]: a run of consecutive entries per thread, wave scan of the run sums, wave totals
            const int run = (N + LG_THREADS - 1) / LG_THREADS;
            const int i0 = 1 + tid * run, i1 = min(N, i0 + run - 1);
            int s = 0;
            for (int k = i0; k <= i1; ++k) s += w.inv_end[k];
            int incl = s;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const int u = __shfl_up(incl, off, 64); if ((tid & 63) >= off) incl += u; }
            if ((tid & 63) == 63) wsum[tid >> 6] = incl;
            __syncthreads();
            int before = incl - s;
            for (int q = 0; q < (tid >> 6); ++q) before += wsum[q];
            for (int k = i0; k <= i1; ++k) { before += w.inv_end[k]; w.inv_end[k] = before; }
        }
        __syncthreads();
        for (int j = tid; j < N; j += LG_THREADS)
            if (w.arg2[j] >= 0) w.inv_lst[atomicAdd(&w.inv_end[w.arg2[j]], 1)] = j;
        for (int e = tid; e < N * NB; e += LG_THREADS) w.inv_lst[atomicAdd(&w.inv_end[w.nbr[e]], 1)] = N + e;
        __syncthreads();
        for (int i = tid; i < N; i += LG_THREADS) {
            float gx = w.g[i], gy = w.g[N + i], gz = w.g[2 * N + i];
            const float wx = p1[i] + fl[i], wy = p1[N + i] + fl[N + i], wz = p1[2 * N + i] + fl[2 * N + i];
            const int s0 = i > 0 ? w.inv_end[i - 1] : 0, s1 = w.inv_end[i];
            for (int u = s0 + 1; u < s1; ++u) {
                const int v = w.inv_lst[u];
                int q = u - 1;
                while (q >= s0 && w.inv_lst[q] > v) { w.inv_lst[q + 1] = w.inv_lst[q]; --q; }
                w.inv_lst[q + 1] = v;
            }
            for (int u = s0; u < s1; ++u) {
                const int id = w.inv_lst[u];
                if (id < N) {
                    gx += k_self * 2.0f * (wx - p2[id]);
                    gy += k_self * 2.0f * (wy - p2[N + id]);
                    gz += k_self * 2.0f * (wz - p2[2 * N + id]);
                } else {
                    const size_t e = (size_t)(id - N);
                    gx += w.gv[e * 3]; gy += w.gv[e * 3 + 1]; gz += w.gv[e * 3 + 2];
                }
            }
            a.d_pred_f[o3 + i] = gx; a.d_pred_f[o3 + N + i] = gy; a.d_pred_f[o3 + 2 * N + i] = gz;
        }
    }
    // ---------------- per-sample partial sums ----------------
    if (tid == 0) {
        float s = 0.f;
        for (int g = 0; g < nwg; ++g) s += w.wgp[2 * g];
        a.partials[(size_t)bs * LS_PARTIALS] = s;
    }
#pragma unroll
    for (int q = 1; q < LS_PARTIALS; ++q) {
        const float s = lg_block_sum<LG_THREADS>(part[q], red);
        if (tid == 0) a.partials[(size_t)bs * LS_PARTIALS + q] = s;
    }
    if (a.d_pre_trans) {
#pragma unroll
        for (int q = 0; q < 12; ++q) {
            const float s = lg_block_sum<LG_THREADS>(et[q], red);
            if (tid == 0) a.d_pre_trans[(size_t)bs * 16 + q] = s;
        }
        if (tid < 4) a.d_pre_trans[(size_t)bs * 16 + 12 + tid] = 0.f;
    }
}

static bool lg_nb_ok(int nb) { return nb == 4 || nb == 8 || nb == 16; }

extern "C" long long cmf_radar_loss_workspace_nb(int b, int n, int num_nb)
{
    const long long head = 4 + (long long)b * LS_PARTIALS;
    if (n <= LS_MAX_N && num_nb == LS_NB) return head;
    return (head + 3) / 4 * 4 + (long long)b * (long long)lg_sample_floats(n, lg_nb_ok(num_nb) ? num_nb : 16);
}

extern "C" long long cmf_radar_loss_workspace(int b, int n) { return cmf_radar_loss_workspace_nb(b, n, LS_NB); }

extern "C" long long cmf_radar_loss_workspace_tiled(int b, int n, int num_nb)
{
    const long long head = 4 + (long long)b * LS_PARTIALS;
    return (head + 3) / 4 * 4 + (long long)b * (long long)lg_sample_floats(n, lg_nb_ok(num_nb) ? num_nb : 16);
}

static int radar_loss_impl(const cmf_radar_loss_desc *d, void *stream, bool force_tiled)
{
    CMF_CHECK_ARG(d && d->B >= 0 && lg_nb_ok(d->num_nb) && d->N > d->num_nb && d->N <= CMF_RADAR_LOSS_MAX_N);
    if (d->B == 0) return 0;
    CMF_CHECK_ARG(d->pc1 && d->pc2 && d->pred_f && d->vel1 && d->items && d->workspace && d->alpha > 0.f);
    CMF_CHECK_ARG(d->self_only || (d->gt_f && d->mseg_pre && d->mseg_gt && d->dyn_mask && d->radar_u && d->radar_v &&
                                   d->opt && d->pre_trans && d->gt_trans && d->camera_inverse && d->t_camera_radar));
    hipStream_t st = (hipStream_t)stream;
    LossArgs a;
    a.B = d->B; a.N = d->N;
    a.pc1 = d->pc1; a.pc2 = d->pc2; a.pred_f = d->pred_f; a.gt_f = d->gt_f; a.vel1 = d->vel1; a.mseg_pre = d->mseg_pre;
    a.mseg_gt = d->mseg_gt; a.dyn_mask = d->dyn_mask; a.radar_u = d->radar_u; a.radar_v = d->radar_v; a.opt = d->opt;
    a.pre_trans = d->pre_trans; a.gt_trans = d->gt_trans; a.cam_inv = d->camera_inverse; a.t_cr = d->t_camera_radar;
    a.w_self = d->w_self; a.w_em = d->w_em; a.w_ms = d->w_ms; a.w_opt = d->w_opt; a.w_dyn = d->w_dyn;
    a.zeta = d->zeta; a.alpha = d->alpha; a.lower_bound = d->lower_bound; a.self_only = d->self_only;
    a.counts = d->workspace; a.partials = d->workspace + 4;
    a.d_pred_f = d->d_pred_f; a.d_pre_trans = d->d_pre_trans; a.d_mseg_pre = d->d_mseg_pre;
    if (!d->self_only)
        hipLaunchKernelGGL(loss_count_kernel, dim3(1), dim3(1024), 0, st, (long long)d->B * d->N, d->mseg_gt, d->dyn_mask,
                           d->workspace);
    a.use_inv = d->N <= LS_INV_MAX_N ? 1 : 0;
    if (!force_tiled && d->N <= LS_MAX_N && d->num_nb == LS_NB) {
        // a sample in one workgroup's LDS (the reference's training size, N = 256)
        const size_t lds = ((size_t)(LS_WORDS_PER_POINT + (a.use_inv ? LS_NB + 2 : 0)) * d->N + 4) * sizeof(float);
        static CmfPerDevice attr_set;                   // the dynamic-LDS limit is per (function, device)
        int attr_dev;
        if (attr_set.need(attr_dev)) {
            (void)hipFuncSetAttribute((const void *)loss_sample_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      LS_WORDS_PER_POINT * LS_MAX_N * (int)sizeof(float));
            attr_set.done(attr_dev);
        }
        hipLaunchKernelGGL(loss_sample_kernel, dim3(d->B), dim3(LS_THREADS), lds, st, a);
    } else {
        // tiled form: the caller sized the workspace with cmf_radar_loss_workspace_nb(B, N, num_nb)
        const long long head = (4 + (long long)d->B * LS_PARTIALS + 3) / 4 * 4;
        float *ws = d->workspace + head;
        const size_t stride = lg_sample_floats(d->N, d->num_nb);
        const dim3 grid_nn((d->N + LG_TILE - 1) / LG_TILE, d->B);
        switch (d->num_nb) {
        case 4:
            hipLaunchKernelGGL(loss_nn_kernel<4>, grid_nn, dim3(LG_TILE), 0, st, a, ws, stride);
            hipLaunchKernelGGL(loss_big_sample_kernel<4>, dim3(d->B), dim3(LG_THREADS), 0, st, a, ws, stride);
            break;
        case 8:
            hipLaunchKernelGGL(loss_nn_kernel<8>, grid_nn, dim3(LG_TILE), 0, st, a, ws, stride);
            hipLaunchKernelGGL(loss_big_sample_kernel<8>, dim3(d->B), dim3(LG_THREADS), 0, st, a, ws, stride);
            break;
        default:
            hipLaunchKernelGGL(loss_nn_kernel<16>, grid_nn, dim3(LG_TILE), 0, st, a, ws, stride);
            hipLaunchKernelGGL(loss_big_sample_kernel<16>, dim3(d->B), dim3(LG_THREADS), 0, st, a, ws, stride);
            break;
        }
    }
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, st, a, d->items);
    return cmf_launch_status();
}

extern "C" int cmf_radar_loss(const cmf_radar_loss_desc *d, void *stream) { return radar_loss_impl(d, stream, false); }

extern "C" int cmf_radar_loss_tiled(const cmf_radar_loss_desc *d, void *stream) { return radar_loss_impl(d, stream, true); }
