// Weighted Kabsch ego-motion solve (models/cmflow.py:128-169) and its backward.
//
// One wavefront per sample: the weighted centroids and the 3x3 covariance H are reduced over
// the N points with wave shuffles (fp64 accumulators -- N is a few hundred, the solve is
// latency bound, and fp64 removes the summation-order sensitivity of the reference's fp32
// torch.sum/matmul), lane 0 runs a 3x3 one-sided Jacobi SVD in registers, and the transform is
// written as (4,4).  The reference calls cuSOLVER/MAGMA batched SVD plus a host sync for det().
//
// Reference quirk reproduced on purpose: in the reflection case (det(V U^T) < 0) the reference
// negates ROW 2 of V (cmflow.py:161-162), i.e. R = diag(1,1,-1) V U^T -- not the textbook
// "negate the last column of V".
#include "cmf_common.h"
#include "../../include/cmflow_hip.h"

// aux layout (doubles): U[9] S[3] V[9] cA[3] cB[3] D sW  (29 used of 32)
constexpr int KB_AUX = 32;

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// H = U diag(S) V^T by one-sided (Hestenes) Jacobi on the columns of H.  Row-major 3x3.
__device__ void svd3(const double *H, double *U, double *S, double *V)
{
    double A[9];
    for (int i = 0; i < 9; ++i) { A[i] = H[i]; V[i] = (i % 4 == 0) ? 1.0 : 0.0; }
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double alpha = 0, beta = 0, gamma = 0;
                for (int i = 0; i < 3; ++i) {
                    alpha += A[i * 3 + p] * A[i * 3 + p];
                    beta += A[i * 3 + q] * A[i * 3 + q];
                    gamma += A[i * 3 + p] * A[i * 3 + q];
                }
                const double lim = 1e-15 * sqrt(alpha * beta);
                if (fabs(gamma) <= lim || gamma == 0.0) continue;
                off = fmax(off, fabs(gamma) / (sqrt(alpha * beta) + 1e-300));
                const double zeta = (beta - alpha) / (2.0 * gamma);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
                for (int i = 0; i < 3; ++i) {
                    const double ap = A[i * 3 + p], aq = A[i * 3 + q];
                    A[i * 3 + p] = c * ap - s * aq;
                    A[i * 3 + q] = s * ap + c * aq;
                    const double vp = V[i * 3 + p], vq = V[i * 3 + q];
                    V[i * 3 + p] = c * vp - s * vq;
                    V[i * 3 + q] = s * vp + c * vq;
                }
            }
        if (off < 1e-15) break;
    }
    // singular values = column norms; U = normalised columns
    double smax = 0;
    for (int j = 0; j < 3; ++j) {
        S[j] = sqrt(A[j] * A[j] + A[3 + j] * A[3 + j] + A[6 + j] * A[6 + j]);
        smax = fmax(smax, S[j]);
    }
    int good[3];
    for (int j = 0; j < 3; ++j) {
        good[j] = S[j] > 1e-14 * smax && S[j] > 0.0;
        if (good[j]) for (int i = 0; i < 3; ++i) U[i * 3 + j] = A[i * 3 + j] / S[j];
    }
    // rank-deficient H: complete U to an orthonormal basis (the rotation is not unique there;
    // any completion is a valid SVD)
    const int ng = good[0] + good[1] + good[2];
    if (ng == 3) return;
    if (ng == 2) {
        int z = !good[0] ? 0 : (!good[1] ? 1 : 2);
        int a = (z + 1) % 3, b = (z + 2) % 3;
        U[0 * 3 + z] = U[1 * 3 + a] * U[2 * 3 + b] - U[2 * 3 + a] * U[1 * 3 + b];
        U[1 * 3 + z] = U[2 * 3 + a] * U[0 * 3 + b] - U[0 * 3 + a] * U[2 * 3 + b];
        U[2 * 3 + z] = U[0 * 3 + a] * U[1 * 3 + b] - U[1 * 3 + a] * U[0 * 3 + b];
        return;
    }
    if (ng == 0) { for (int i = 0; i < 9; ++i) U[i] = (i % 4 == 0) ? 1.0 : 0.0; return; }
    // ng == 1: build two vectors orthogonal to the good column
    int g = good[0] ? 0 : (good[1] ? 1 : 2);
    double u0 = U[g], u1 = U[3 + g], u2 = U[6 + g];
    double e0 = fabs(u0) < 0.9 ? 1.0 : 0.0, e1 = 1.0 - e0, e2 = 0.0;
    double d = e0 * u0 + e1 * u1 + e2 * u2;
    double v0 = e0 - d * u0, v1 = e1 - d * u1, v2 = e2 - d * u2;
    double nv = sqrt(v0 * v0 + v1 * v1 + v2 * v2);
    v0 /= nv; v1 /= nv; v2 /= nv;
    int a = (g + 1) % 3, b = (g + 2) % 3;
    U[a] = v0; U[3 + a] = v1; U[6 + a] = v2;
    U[b] = u1 * v2 - u2 * v1; U[3 + b] = u2 * v0 - u0 * v2; U[6 + b] = u0 * v1 - u1 * v0;
}

__device__ __forceinline__ double det3(const double *M)
{
    return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) +
           M[2] * (M[3] * M[7] - M[4] * M[6]);
}

// one sample (one wavefront): a, b (3,n), w (n) -> T (4,4) [lane 0 writes], x = the sample's aux record or null
__device__ __forceinline__ void kabsch_fwd_sample(int n, const float *a, const float *b, const float *w, float *T, double *x, int lane)
{
    // pass 1: weighted centroids  (cmflow.py:138-139)
    double sw = 0, ca[3] = {0, 0, 0}, cb[3] = {0, 0, 0};
    for (int i = lane; i < n; i += CMF_WAVE) {
        const double wi = w[i];
        sw += wi;
        for (int k = 0; k < 3; ++k) { ca[k] += wi * a[k * n + i]; cb[k] += wi * b[k * n + i]; }
    }
    sw = wave_sum(sw);
    for (int k = 0; k < 3; ++k) { ca[k] = wave_sum(ca[k]); cb[k] = wave_sum(cb[k]); }
    // pass 2: H = Am (Bm^T . W)  (cmflow.py:148-151)
    double H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = lane; i < n; i += CMF_WAVE) {
        const double wi = w[i];
        double da[3], db[3];
        for (int k = 0; k < 3; ++k) { da[k] = a[k * n + i] - ca[k]; db[k] = (b[k * n + i] - cb[k]) * wi; }
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) H[r * 3 + c] += da[r] * db[c];
    }
    for (int k = 0; k < 9; ++k) H[k] = wave_sum(H[k]);
    if (lane != 0) return;
    double U[9], S[3], V[9];
    svd3(H, U, S, V);
    double Z[9];                                   // Z = V U^T  (cmflow.py:155)
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
            Z[r * 3 + c] = V[r * 3 + 0] * U[c * 3 + 0] + V[r * 3 + 1] * U[c * 3 + 1] + V[r * 3 + 2] * U[c * 3 + 2];
    const double D = det3(Z) < 0 ? -1.0 : 1.0;     // reflection: row 2 of V (hence of Z) negated
    double R[9];
    for (int c = 0; c < 3; ++c) { R[c] = Z[c]; R[3 + c] = Z[3 + c]; R[6 + c] = D * Z[6 + c]; }
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) T[r * 4 + c] = (float)R[r * 3 + c];
        T[r * 4 + 3] = (float)(-(R[r * 3 + 0] * ca[0] + R[r * 3 + 1] * ca[1] + R[r * 3 + 2] * ca[2]) + cb[r]);
    }
    T[12] = 0.f; T[13] = 0.f; T[14] = 0.f; T[15] = 1.f;
    if (x) {
        for (int k = 0; k < 9; ++k) { x[k] = U[k]; x[12 + k] = V[k]; }
        for (int k = 0; k < 3; ++k) { x[9 + k] = S[k]; x[21 + k] = ca[k]; x[24 + k] = cb[k]; }
        x[27] = D; x[28] = sw;
    }
}

__global__ __launch_bounds__(CMF_WAVE) void kabsch_fwd_kernel(
    int n, const float *__restrict__ A, const float *__restrict__ Bm, const float *__restrict__ W,
    float *__restrict__ trans, double *__restrict__ aux)
{
    const int bs = blockIdx.x;
    kabsch_fwd_sample(n, A + (size_t)bs * 3 * n, Bm + (size_t)bs * 3 * n, W + (size_t)bs * n, trans + (size_t)bs * 16,
                      aux ? aux + (size_t)bs * KB_AUX : nullptr, threadIdx.x);
}

// ---- the ego-motion head and the rigid refinement around the solve (cmflow.py:96-125) as ONE kernel per direction ----
//   w = (score + eps) / sum(score + eps),  B = pc1 + flow,  T = kabsch(pc1, B, w),  mask = score > thres,
//   sf = mask ? (R pc1 + t - pc1) : flow
// One wavefront per sample as above.  w and B are written out (the backward pass reads them; a lane re-reads only what it wrote
// itself).  As torch ops this was ~10 small kernels forward and ~15 backward on the main stream with nothing beside them.
__global__ __launch_bounds__(CMF_WAVE) void ego_refine_fwd_kernel(
    int n, float eps, float thres, const float *__restrict__ pc1, const float *__restrict__ flow, const float *__restrict__ score,
    float *W, float *Bm, float *__restrict__ trans, double *__restrict__ aux, float *__restrict__ sf, unsigned char *__restrict__ mask)
{
    const int bs = blockIdx.x, lane = threadIdx.x;
    const float *a = pc1 + (size_t)bs * 3 * n, *f = flow + (size_t)bs * 3 * n, *sc = score + (size_t)bs * n;
    float *w = W + (size_t)bs * n, *b = Bm + (size_t)bs * 3 * n;
    double ssum = 0.0;
    for (int i = lane; i < n; i += CMF_WAVE) ssum += (double)(sc[i] + eps);
    const float tot = (float)wave_sum(ssum);
    for (int i = lane; i < n; i += CMF_WAVE) {
        w[i] = (sc[i] + eps) / tot;
        for (int k = 0; k < 3; ++k) b[k * n + i] = a[k * n + i] + f[k * n + i];
    }
    __shared__ float Tsh[16];                       // lane 0 solves; the wave reads the transform back from LDS
    kabsch_fwd_sample(n, a, b, w, Tsh, aux ? aux + (size_t)bs * KB_AUX : nullptr, lane);
    __syncthreads();
    if (lane < 16) trans[(size_t)bs * 16 + lane] = Tsh[lane];
    float Tm[12];
    for (int k = 0; k < 12; ++k) Tm[k] = Tsh[k];
    for (int i = lane; i < n; i += CMF_WAVE) {
        const float x = a[i], y = a[n + i], z = a[2 * n + i];
        const bool m = sc[i] > thres;
        mask[(size_t)bs * n + i] = m ? 1 : 0;
        for (int r = 0; r < 3; ++r) {
            const float rig = fmaf(Tm[r * 4 + 2], z, fmaf(Tm[r * 4 + 1], y, Tm[r * 4 + 0] * x)) + Tm[r * 4 + 3] - a[r * n + i];
            sf[(size_t)bs * 3 * n + r * n + i] = m ? rig : f[r * n + i];
        }
    }
}

// Backward.  With M = H^T = Z P (Z = V U^T orthogonal, P = U S U^T), the derivative of the
// polar factor gives  G_H = -U Y V^T,  Y_ij = (Q_ij - Q_ji) / (s_i + s_j),  Q = V^T G_Z U
// (DESIGN.md "Kabsch backward").  Then
//   H  = sum_n W_n (A_n - cA)(B_n - cB)^T,   cA = sum W_n A_n,   cB = sum W_n B_n,
//   t  = -R cA + cB.
// one sample: gt = the gradient w.r.t. its (4,4) transform as 12 doubles (rows of [R | t]); a, b (3,n), w (n); outputs per sample
__device__ __forceinline__ void kabsch_bwd_sample(int n, const float *a, const float *b, const float *w, const double *x, const double (&gt)[12],
                                                  float *gA, float *gB, float *gW, int lane)
{
    double U[9], S[3], V[9], ca[3], cb[3];
    for (int k = 0; k < 9; ++k) { U[k] = x[k]; V[k] = x[12 + k]; }
    for (int k = 0; k < 3; ++k) { S[k] = x[9 + k]; ca[k] = x[21 + k]; cb[k] = x[24 + k]; }
    const double D = x[27], sw = x[28];
    double GR[9], g_t[3];
    for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) GR[r * 3 + c] = gt[r * 4 + c]; g_t[r] = gt[r * 4 + 3]; }
    // R as in forward
    double Z[9], R[9];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
            Z[r * 3 + c] = V[r * 3 + 0] * U[c * 3 + 0] + V[r * 3 + 1] * U[c * 3 + 1] + V[r * 3 + 2] * U[c * 3 + 2];
    for (int c = 0; c < 3; ++c) { R[c] = Z[c]; R[3 + c] = Z[3 + c]; R[6 + c] = D * Z[6 + c]; }
    // t = -R cA + cB
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) GR[r * 3 + c] -= g_t[r] * ca[c];
    double gcA[3], gcB[3];
    for (int c = 0; c < 3; ++c) {
        gcA[c] = -(R[0 * 3 + c] * g_t[0] + R[1 * 3 + c] * g_t[1] + R[2 * 3 + c] * g_t[2]);
        gcB[c] = g_t[c];
    }
    // G_Z = D-row-scaled G_R
    double GZ[9];
    for (int c = 0; c < 3; ++c) { GZ[c] = GR[c]; GZ[3 + c] = GR[3 + c]; GZ[6 + c] = D * GR[6 + c]; }
    // Q = V^T G_Z U
    double T1[9], Q[9];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
            T1[r * 3 + c] = V[0 * 3 + r] * GZ[0 * 3 + c] + V[1 * 3 + r] * GZ[1 * 3 + c] + V[2 * 3 + r] * GZ[2 * 3 + c];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
            Q[r * 3 + c] = T1[r * 3 + 0] * U[0 * 3 + c] + T1[r * 3 + 1] * U[1 * 3 + c] + T1[r * 3 + 2] * U[2 * 3 + c];
    double Y[9];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            const double den = S[r] + S[c];
            Y[r * 3 + c] = (r == c || den <= 0.0) ? 0.0 : (Q[r * 3 + c] - Q[c * 3 + r]) / den;
        }
    // G_H = -U Y V^T
    double GH[9];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
            T1[r * 3 + c] = U[r * 3 + 0] * Y[0 * 3 + c] + U[r * 3 + 1] * Y[1 * 3 + c] + U[r * 3 + 2] * Y[2 * 3 + c];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
            GH[r * 3 + c] = -(T1[r * 3 + 0] * V[c * 3 + 0] + T1[r * 3 + 1] * V[c * 3 + 1] + T1[r * 3 + 2] * V[c * 3 + 2]);
    // centroid terms through H (vanish when sum(W) == 1, kept for exactness)
    const double k1 = 1.0 - sw;
    for (int r = 0; r < 3; ++r) {
        gcA[r] -= (GH[r * 3 + 0] * cb[0] + GH[r * 3 + 1] * cb[1] + GH[r * 3 + 2] * cb[2]) * k1;
        gcB[r] -= (GH[0 * 3 + r] * ca[0] + GH[1 * 3 + r] * ca[1] + GH[2 * 3 + r] * ca[2]) * k1;
    }
    for (int i = lane; i < n; i += CMF_WAVE) {
        const double wi = w[i];
        double da[3], db[3], av[3], bv[3];
        for (int k = 0; k < 3; ++k) { av[k] = a[k * n + i]; bv[k] = b[k * n + i]; da[k] = av[k] - ca[k]; db[k] = bv[k] - cb[k]; }
        double hb[3], ha[3];       // G_H^T da,  G_H db
        for (int k = 0; k < 3; ++k) {
            hb[k] = GH[0 * 3 + k] * da[0] + GH[1 * 3 + k] * da[1] + GH[2 * 3 + k] * da[2];
            ha[k] = GH[k * 3 + 0] * db[0] + GH[k * 3 + 1] * db[1] + GH[k * 3 + 2] * db[2];
        }
        if (gB) for (int k = 0; k < 3; ++k) gB[k * n + i] = (float)(wi * (hb[k] + gcB[k]));
        if (gA) for (int k = 0; k < 3; ++k) gA[k * n + i] = (float)(wi * (ha[k] + gcA[k]));
        if (gW) {
            double g = da[0] * ha[0] + da[1] * ha[1] + da[2] * ha[2];
            for (int k = 0; k < 3; ++k) g += gcA[k] * av[k] + gcB[k] * bv[k];
            gW[i] = (float)g;
        }
    }
}

__global__ __launch_bounds__(CMF_WAVE) void kabsch_bwd_kernel(
    int n, const float *__restrict__ A, const float *__restrict__ Bm, const float *__restrict__ W,
    const double *__restrict__ aux, const float *__restrict__ grad_trans,
    float *__restrict__ gA, float *__restrict__ gB, float *__restrict__ gW)
{
    const int bs = blockIdx.x;
    double gt[12];
    for (int k = 0; k < 12; ++k) gt[k] = grad_trans[(size_t)bs * 16 + k];
    kabsch_bwd_sample(n, A + (size_t)bs * 3 * n, Bm + (size_t)bs * 3 * n, W + (size_t)bs * n, aux + (size_t)bs * KB_AUX, gt,
                      gA ? gA + (size_t)bs * 3 * n : nullptr, gB ? gB + (size_t)bs * 3 * n : nullptr, gW ? gW + (size_t)bs * n : nullptr, threadIdx.x);
}

// Backward of ego_refine_fwd_kernel: g_sf (b,3,n), g_trans (b,4,4) or null -> g_flow (b,3,n), g_score (b,n) or null.
//   masked points: sf = R a + t - a  ->  G_R += g_sf a^T, g_t += g_sf;  the others hand g_sf to flow
//   kabsch backward with the summed transform gradient -> g_B (= g_flow's second part: B = pc1 + flow), g_w
//   w = s / sum(s)  ->  g_s = (g_w - sum(g_w w)) / sum(s)
__global__ __launch_bounds__(CMF_WAVE) void ego_refine_bwd_kernel(
    int n, float eps, const float *__restrict__ pc1, const float *__restrict__ score, const float *__restrict__ W, const float *__restrict__ Bm,
    const unsigned char *__restrict__ mask, const double *__restrict__ aux, const float *__restrict__ g_sf, const float *__restrict__ g_trans,
    float *__restrict__ g_flow, float *g_w, float *__restrict__ g_score)
{
    const int bs = blockIdx.x, lane = threadIdx.x;
    const float *a = pc1 + (size_t)bs * 3 * n, *gs = g_sf + (size_t)bs * 3 * n, *w = W + (size_t)bs * n;
    const unsigned char *mk = mask + (size_t)bs * n;
    double gt[12];
    for (int k = 0; k < 12; ++k) gt[k] = 0.0;
    for (int i = lane; i < n; i += CMF_WAVE) {
        if (!mk[i]) continue;
        const double av[3] = {a[i], a[n + i], a[2 * n + i]};
        for (int r = 0; r < 3; ++r) {
            const double g = gs[r * n + i];
            gt[r * 4 + 0] += g * av[0]; gt[r * 4 + 1] += g * av[1]; gt[r * 4 + 2] += g * av[2]; gt[r * 4 + 3] += g;
        }
    }
    for (int k = 0; k < 12; ++k) gt[k] = wave_sum(gt[k]) + (g_trans ? (double)g_trans[(size_t)bs * 16 + k] : 0.0);
    float *gf = g_flow + (size_t)bs * 3 * n, *gw = g_w + (size_t)bs * n;
    kabsch_bwd_sample(n, a, Bm + (size_t)bs * 3 * n, w, aux + (size_t)bs * KB_AUX, gt, nullptr, gf, gw, lane);
    // (a lane re-reads only the entries it wrote itself: no barrier)
    double dot = 0.0, ssum = 0.0;
    for (int i = lane; i < n; i += CMF_WAVE) {
        if (!mk[i]) for (int k = 0; k < 3; ++k) gf[k * n + i] += gs[k * n + i];
        dot += (double)gw[i] * (double)w[i];
        ssum += (double)(score[(size_t)bs * n + i] + eps);
    }
    if (!g_score) return;
    dot = wave_sum(dot);
    const float tot = (float)wave_sum(ssum);
    for (int i = lane; i < n; i += CMF_WAVE) g_score[(size_t)bs * n + i] = (float)(((double)gw[i] - dot) / (double)tot);
}

extern "C" int cmf_weighted_kabsch(int b, int n, const float *A, const float *Bm, const float *W,
                                   float *trans, double *aux, void *stream)
{
    CMF_CHECK_ARG(b >= 0 && n > 0);
    if (b == 0) return 0;
    CMF_CHECK_ARG(A && Bm && W && trans);
    hipLaunchKernelGGL(kabsch_fwd_kernel, dim3(b), dim3(CMF_WAVE), 0, (hipStream_t)stream, n, A, Bm, W, trans, aux);
    return cmf_launch_status();
}

extern "C" int cmf_weighted_kabsch_grad(int b, int n, const float *A, const float *Bm, const float *W,
                                        const double *aux, const float *grad_trans,
                                        float *grad_A, float *grad_B, float *grad_W, void *stream)
{
    CMF_CHECK_ARG(b >= 0 && n > 0);
    if (b == 0) return 0;
    CMF_CHECK_ARG(A && Bm && W && aux && grad_trans);
    hipLaunchKernelGGL(kabsch_bwd_kernel, dim3(b), dim3(CMF_WAVE), 0, (hipStream_t)stream,
                       n, A, Bm, W, aux, grad_trans, grad_A, grad_B, grad_W);
    return cmf_launch_status();
}

extern "C" int cmf_ego_refine(int b, int n, float eps, float thres, const float *pc1, const float *flow, const float *score,
                              float *W, float *Bm, float *trans, double *aux, float *sf, unsigned char *mask, void *stream)
{
    CMF_CHECK_ARG(b >= 0 && n > 0);
    if (b == 0) return 0;
    CMF_CHECK_ARG(pc1 && flow && score && W && Bm && trans && sf && mask);
    hipLaunchKernelGGL(ego_refine_fwd_kernel, dim3(b), dim3(CMF_WAVE), 0, (hipStream_t)stream, n, eps, thres, pc1, flow, score, W, Bm, trans,
                       aux, sf, mask);
    return cmf_launch_status();
}

extern "C" int cmf_ego_refine_grad(int b, int n, float eps, const float *pc1, const float *score, const float *W, const float *Bm,
                                   const unsigned char *mask, const double *aux, const float *g_sf, const float *g_trans,
                                   float *g_flow, float *g_w, float *g_score, void *stream)
{
    CMF_CHECK_ARG(b >= 0 && n > 0);
    if (b == 0) return 0;
    CMF_CHECK_ARG(pc1 && score && W && Bm && mask && aux && g_sf && g_flow && g_w);
    hipLaunchKernelGGL(ego_refine_bwd_kernel, dim3(b), dim3(CMF_WAVE), 0, (hipStream_t)stream, n, eps, pc1, score, W, Bm, mask, aux, g_sf,
                       g_trans, g_flow, g_w, g_score);
    return cmf_launch_status();
}

// test-only stream delay (cmflow_hip.h): one wave polling the constant-frequency clock
__global__ __launch_bounds__(64) void debug_spin_kernel(unsigned long long ticks)
{
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}

extern "C" int cmf_debug_spin(float microseconds, void *stream)
{
    CMF_CHECK_ARG(microseconds >= 0.f && microseconds <= 1.0e6f);
    if (microseconds == 0.f) return 0;
    hipLaunchKernelGGL(debug_spin_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned long long)(microseconds * 100.0f));
    return cmf_launch_status();
}

extern "C" const char *cmf_version(void) { return "cmflow_hip 0.1 (gfx950)"; }
