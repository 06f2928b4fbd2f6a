// Shared helpers for the gfx950 kernels (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <mutex>
#include <stdint.h>

#define CMF_WAVE 64

#define CMF_CHECK_ARG(cond) \
    do { if (!(cond)) return (int)hipErrorInvalidValue; } while (0)

static inline int cmf_launch_status() { return (int)hipGetLastError(); }

static inline int cmf_divup(long long a, long long b) { return (int)((a + b - 1) / b); }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is an attribute of (function, DEVICE): a process-wide `static bool` would
// leave a second device without it (launch failure above 64 KB).  One bit per device, atomics (entry points are called from
// several host threads):   static CmfPerDevice once; int dev;  if (once.need(dev)) { ...set attributes...; once.done(dev); }
struct CmfPerDevice {
    std::atomic<unsigned> mask[4] = {};
    bool need(int &dev) { dev = 0; (void)hipGetDevice(&dev); return dev >= 128 || !(mask[dev >> 5].load(std::memory_order_acquire) & (1u << (dev & 31))); }
    void done(int dev) { if (dev < 128) mask[dev >> 5].fetch_or(1u << (dev & 31), std::memory_order_release); }
};

// Library-owned scratch for the few entry points that need more working memory than their reference signature
// carries (cmf_ball_query: spilled hit lists at nsample > 32; cmf_group_points_grad: the inverse index).  One buffer per
// (device, stream, slot), grown on demand and kept for the life of the process: work on one stream is ordered, so the
// next call on that stream may reuse it, and calls on different streams get different buffers.  [The first version
// used hipMallocAsync / hipFreeAsync per call: normally a pool hit, but measured at 4.6 ms for one 64 MB request after
// the pool had been trimmed -- 25x the kernel it served.]  group_points.hip.
// The buffer comes as a LEASE: several host threads may enqueue onto one stream (two encoder scales share a side stream),
// so the (device, stream, slot) entry stays locked from the hand-out until the lease goes out of scope -- which callers
// arrange to be after the last launch that uses the pointer (build + use are then adjacent in stream order).  A buffer
// that has to grow is retired behind an event recorded on its stream and freed at a later lease once that event has
// completed (queued work of the stream may still use it until then); it grows by >= 1.5x.
// ptr == nullptr on allocation failure.
struct CmfScratchLease {
    void *ptr = nullptr;
    std::unique_lock<std::mutex> hold;
};
CmfScratchLease cmf_stream_scratch(hipStream_t stream, int slot, size_t bytes);

// Train-mode BatchNorm backward of its two streams, per column:
//   dZ = a * (dU - s1/M - zhat * s2/M),  zhat = (z - mean) * invstd      ==      dZ = al * dU + be * (z - mean) + ga
// with al = a, be = -a * invstd * s2/M, ga = -a * s1/M.  The CENTRED form: (z - mean) is formed first, so the zhat term has no
// cancellation (the uncentred  be * z + (ga + a*mean*k2)  adds two large nearly cancelling terms and loses eps*|mean|/std).
// One definition for every kernel that forms dZ (bn_bwd_apply_kernel, the weight-gradient GEMM that materialises dZ while
// staging its A operand): the same operations in the same order, so the fused and the stand-alone forms are bit-identical.
#ifdef __HIPCC__
__device__ __forceinline__ void cmf_bnb_coef(float sa, float mu, float is, float t1, float t2, float ic, float &al, float &be, float &ga)
{
    (void)mu;
    const float k1 = t1 * ic, k2 = is * (t2 * ic);
    al = sa;
    be = -(sa * k2);
    ga = -(sa * k1);
}
__device__ __forceinline__ float cmf_bnb_apply(float d, float v, float mu, float al, float be, float ga) { return fmaf(al, d, fmaf(be, v - mu, ga)); }
#endif

// neighbor.hip: cmf_ball_query that also defines the rows of empty balls (zeros), without a memset launch on small clouds
int cmf_ball_query_defined(int b, int n, int m, float radius, int nsample, const float *new_xyz, const float *xyz, int *idx, void *stream);

// ---- batched launches (round 3) -------------------------------------------------------------------------------------
// The narrow layers of the set-conv chains are latency, not work: the per-point tail of a block is three 64-channel
// layers over B*N rows (128 workgroups per kernel), and the eight chains of an encoder call each issued their own ~10
// tiny kernels per direction -- 1.5 ms of the 22 ms training step (measured by leaving the tails out).  A batch kernel
// takes up to CMF_MAX_BATCH argument blocks BY VALUE; blockIdx.y selects the problem and blocks past a problem's last tile
// leave at once.  Same device code, same arithmetic, same results as the single launches -- one launch per layer for all
// chains instead of one per chain.
constexpr int CMF_MAX_BATCH = 8;
template <typename A> struct CmfBatch { A a[CMF_MAX_BATCH]; };

struct GemmArgs;
// BatchNorm finalize (cmf_bn_finalize), column sums (cmf_colsum_finalize / cmf_colsum_store), dU = dY * mask with the BN
// sums (cmf_act_bwd_stats), y = relu(a z + c) (cmf_affine_relu), split-K slab sums (cmf_splitk_reduce)
struct CmfBnFinArgs { int tiles, C; double count; const float *partial, *gamma, *beta; float eps, momentum;
                      float *rmean, *rvar, *mean_out, *invstd_out, *a_out, *c_out; long long *nbt; };
struct CmfColsumArgs { int tiles, ncols; const float *partial; float *out; int C; float *acc0, *acc1; int store; };
struct CmfActBwdArgs { long long rows; int C; const float *dY; long long ldy; const float *z; long long ldz;
                       const float *a, *c, *mean, *invstd; float *dU, *partial; };
struct CmfAffineArgs { long long M; int C; const float *z; long long ldz; const float *a, *c; float *out; long long ldo; };
// the narrow set-conv blocks' first layer (cmf_group_affine without centre rows: z, dxyz, statistics, z * d_k sums) and the max over the ball
struct CmfGroupAffineArgs { int b, n_src, P, S, C; const float *ysrc; int ld_src; const float *xyz_src, *xyz_ctr, *Wx; int ldw; const int *idx;
                            float *z, *dxyz, *partial, *partial_x; };
struct CmfPoolArgs { long long P; int S, C; const float *z, *a, *c; float *out; long long ldo; unsigned char *argmax; int grid; float *zsel; };
// backward of the narrow blocks' bodies: max-pool backward per point, the inverse index, dW_xyz from column sums, the scatter with the
// first layer's BN backward in closed form (cmf_maxpool_bwd_point, cmf_build_inverse_ps, cmf_setconv_dwx, cmf_group_rows_grad_bn_cf)
struct CmfPoolBwdArgs { long long P; int S, C; const float *dout; long long ldd; const float *z, *a, *c, *mean, *invstd; const unsigned char *argmax;
                        float *g, *partial; int sel; };     // sel: z = the (P, C) pre-activations at the arg-max slots (bn_relu_maxpool's zsel)
struct CmfInverseArgs { int n, P, S; const int *idx; int *offsets, *inv; };
struct CmfDwxArgs { int C; float inv_count; int train; const float *bwd5, *fwd, *a, *mean, *invstd; float *dwx; int ld, accumulate; };
struct CmfScatterArgs { int n, entries, S; const float *dU, *y; long long ldy; const float *wx; long long ldw; const float *xyz_src, *xyz_ctr,
                        *a, *mean, *invstd, *sums; float inv_count; const int *offsets, *inv; float *grad_feat; int ldg; };
// one block's inference chain (setconv_chain.hip): the arguments of cmf_setconv_chain_infer
struct CmfChainInferArgs { long long M; int N, S; const int *idx; const float *xyz, *y; long long ldy; const float *wx; long long ldwx;
                           const float *bn0, *bn1, *bn2, *w2, *w3; float *out; long long ldo; };
int cmf_setconv_chain_infer_batch(int n, const CmfChainInferArgs *q, void *stream);
struct CmfSplitkArgs { int M, N, split_k; const float *workspace; float *C; long long ldc; int accumulate; };
// one fused backward layer (cmf_thin_bwd_layer); nslab is filled in by the batch call
struct CmfThinBwdCall { long long rows; int cout, cin; const float *dU; long long lddu; const float *z; long long ldz;
                        const float *a, *mean, *invstd, *sums; const float *w; long long ldw; const float *x; long long ldx; int in_mode;
                        const float *a_in, *c_in, *mean_in, *invstd_in; float *dx; long long lddx; float *stats;
                        float *dw; long long lddw; int accumulate; float *slabs; int nslab;
                        const float *dxyz; const float *pool_g; const unsigned char *pool_am; int pool_S; };   // (in_mode 1: dxyz sums; pooled dU: rows = P * pool_S)
int cmf_bn_finalize_batch(int n, const CmfBnFinArgs *a, hipStream_t st);                 // pointwise.hip
int cmf_colsum_batch(int n, const CmfColsumArgs *a, hipStream_t st);                     // pointwise.hip
int cmf_maxpool_bwd_point_batch(int n, const CmfPoolBwdArgs *a, hipStream_t st);          // pointwise.hip: same C for all
int cmf_setconv_dwx_batch(int n, const CmfDwxArgs *a, hipStream_t st);                    // pointwise.hip
int cmf_build_inverse_ps_batch(int n, int b, const CmfInverseArgs *a, hipStream_t st);    // group_rows.hip: b samples each, the matrix form
int cmf_group_rows_grad_bn_cf_batch(int n, int b, int c, const CmfScatterArgs *a, hipStream_t st);   // group_rows.hip: narrow rows (c = 16 .. 128), same n for all
int cmf_group_affine_batch(int n, const CmfGroupAffineArgs *a, hipStream_t st);           // pointwise.hip: same C for all
int cmf_bn_relu_maxpool_batch(int n, CmfPoolArgs *a, hipStream_t st);                    // pointwise.hip (fills in grid)
int cmf_act_bwd_stats_batch(int n, const CmfActBwdArgs *a, hipStream_t st);              // pointwise.hip
int cmf_affine_relu_batch(int n, const CmfAffineArgs *a, hipStream_t st);                // pointwise.hip
int cmf_splitk_reduce_batch(int n, const CmfSplitkArgs *a, hipStream_t st);              // gemm.hip
int cmf_thin_fwd_batch(int n, const GemmArgs *g, hipStream_t st);                        // thin_gemm.hip: A[M,K] W[N,K]^T, K, N <= 64, same (N, K) for all
int cmf_thin_bwd_layer_batch(int n, CmfThinBwdCall *c, hipStream_t st);                  // thin_gemm.hip: kernels only (no slab sum), same widths / mode
