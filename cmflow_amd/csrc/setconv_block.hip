// Host-side orchestration of one set-conv block (PointLocalFeature, utils/model_utils/radarflow_util.py:144-162)
// forward and backward as ONE C-ABI call each.
//
// The block is ~25 kernel launches forward and ~45 backward, most of them a few microseconds long at N = 256.
// Issued one by one from Python (ctypes call + tensor allocations + autograd bookkeeping, ~10-15 us each)
// the first encoder of the model was entirely host bound (1.3 ms per call for ~0.2 ms of GPU work).  Here the
// whole sequence is enqueued from C++ into caller-provided arenas; the kernels and their order are exactly
// those of cmflow_amd/fused_blocks.py (SetConvFn), which remains the readable specification.
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include "cmf_common.h"
#include "gemm_args.h"
#include "../../include/cmflow_hip.h"

bool cmf_setconv_chain_supported(int N, int S, int O1, int C2, int C3, long long M);
long long cmf_setconv_chain_waves(long long M, int backward);
bool cmf_ball_query_multi_takes(int n, int nq, const int *nsamples);      // csrc/neighbor.hip

namespace {

// training through the register chain (csrc/setconv_chain.hip): the narrow block's slot-level activations z1 / z2 / z3 are never stored
// -- every pass recomputes them from the gathered per-point rows; z3's slot of `saved` holds the selected pre-activations (P x C3).
// OPT-IN (CMF_CHAIN_TRAIN=1), measured round 5 on the training step's first encoder (both clouds, four scales, isolated kernel
// time): five chain passes 1.64 ms against 1.50 ms for the per-layer kernels they replace -- at 64 K .. 512 K rows per launch the
// passes' fixed costs (constants and weights into LDS per workgroup, a statistics row per wave, the re-gather per pass) eat what
// the 3x lower HBM traffic saves; the step is equal within noise (20.15 vs 20.20 ms).  Inference (one pass, nothing else to
// amortise) is where the chain pays: forward 5.91 -> 5.56 ms.
// With eval-mode BatchNorm and a backward pass to follow (the regime of every epoch after the first in the reference's loop: no
// statistics passes) the chain is three passes against seven per-layer kernels -- 1.25 against 1.50 ms -- and is the DEFAULT there
// (CMF_CHAIN_TRAIN=0: the per-layer kernels everywhere).
inline bool chain_train(const cmf_setconv_desc *d)
{
    static const int mode = getenv("CMF_CHAIN_TRAIN") ? atoi(getenv("CMF_CHAIN_TRAIN")) : -1;     // -1: eval-mode BN only, 0: never, 1: always
    const bool on = mode == 1 || (mode == -1 && !d->training);
    const long long M = (long long)d->B * d->N * d->S;
    return on && !d->inference && cmf_setconv_chain_supported(d->N, d->S, d->O1, d->C[0], d->C[1], M) && M % 128 == 0 && d->ldy % 4 == 0 &&
           ((uintptr_t)d->y & 15) == 0;
}

struct Bump {
    float *base;
    size_t off;
    explicit Bump(float *b) : base(b), off(0) {}
    float *take(size_t n)
    {
        float *p = base ? base + off : nullptr;
        off += (n + 63) / 64 * 64;                       // 256-byte granules: every buffer 16-byte aligned
        return p;
    }
};

inline int dw_split(long long M, int N, int K);
// the gathering weight gradient's split: the count its 256-row tiles go with, else the usual choice
inline int dwg_split(long long M, int N, int K)
{
    const int s = cmf_gemm_dw_gather_split(N, K, M);
    return s ? s : dw_split(M, N, K);
}

inline int tiles128(long long rows) { return (int)((rows + 127) / 128); }

// split-K choice of fused_blocks.gemm_dw (positions M contracted into an N x K weight gradient)
inline int dw_split(long long M, int N, int K)
{
    if (N <= 64 && K <= 64) return (int)std::max<long long>(2, std::min<long long>(M / 128, 1024));   // thin kernel: slab per workgroup, >= 128 rows each
    // tiled kernel: slabs are dealt to the 8 XCDs (split % 8), so the split is a multiple of 8; among the candidates
    // take the one with the smallest wave-quantised cost: waves of 768 resident workgroups x (chunks per workgroup +
    // a fixed per-workgroup overhead of ~12 chunks).  [A "fill 512 workgroups" rule gave the 2048 x 1028 gradient of the
    // stacked first conv a split of 4: half the XCDs idle, 1.55 ms instead of 0.73 ms.]
    const long long tiles = (long long)((N + 127) / 128) * ((K + 127) / 128);
    const long long chunks = (M + 15) / 16;
    if (chunks < 64) return 1;
    static const int cand[] = {8, 16, 24, 32, 48, 64, 96, 128, 256, 384};     // 256 / 384: one- and two-tile outputs (64 x 256 over 524288 rows: 242 -> 203 us)
    int best = 8;
    double best_cost = 1e30;
    for (int s : cand) {
        if (chunks / s < 8) break;
        const double cost = (double)((tiles * s + 767) / 768) * ((double)chunks / s + 12.0);
        if (cost < best_cost) { best_cost = cost; best = s; }
    }
    return best;
}

struct Layout {
    // saved for backward
    int *idx, *offsets, *inv;
    float *dxyz, *z1, *z2, *z3, *x, *z4, *z5, *z6, *fwd_sums;
    unsigned char *argmax;
    float *zsel;                  // (P, C3) pre-activations of layer 3 at the arg-max slots (per-layer path; the chain keeps them in z3's slot)
    float *bn[6];                 // per layer: mean | invstd | a | c   (4 * C_l)
    // scratch
    float *partial, *partial_x, *sums, *t6, *t5, *t4, *dx, *dU3, *dU2, *dU1, *splitk;
    float *dZ2;                   // dZ of BN layer 1, written by the weight-gradient GEMM that fuses its BN backward (or null)
    size_t saved_floats, scratch_floats;
};

inline int chan(const cmf_setconv_desc *d, int layer)       // out channels of BN layer 0..5
{
    return layer == 0 ? d->O1 : d->C[layer - 1];
}

// The wide backward layer (cout <- cin over `rows`) may fuse its BN backward into the weight-gradient GEMM
// (cmf_gemm_dw_bn_bwd): shapes the fused kernel takes, and long enough that the unfused cmf_gemm would run the same
// register-staged loop (K >= 32768, gemm.hip) -- the two forms are then bit-identical.  Opt-in (CMF_BNB_FUSED=1), measured
// in four same-box A/B pairs of the training step: the stand-alone pass (0.48 ms per step) goes, the four weight-gradient
// GEMMs that absorb it run 99.5 instead of 112 TF (+0.27 ms; two more operand loads per thread and chunk in the staging
// loop) -- 21.33-21.43 against 21.42-21.53 ms per step, i.e. 0.1 ms, for a tiled kernel that is 2 points further from its
// MFMA roof.  Not worth a second form of the layer as the default.
inline bool bnb_fusable(long long rows, int cout, int cin)
{
    static const bool on = getenv("CMF_BNB_FUSED") && getenv("CMF_BNB_FUSED")[0] == '1';
    return on && cout % 128 == 0 && cin % 128 == 0 && rows % 16 == 0 && rows >= 32768 && rows < (1ll << 31);
}

// The arenas are sized by PATH (VERDICT r5 item 7 / ADVICE r4).  A block that never materialises its grouped first-layer tensor
// (gather_layout: the second encoder's shapes) keeps in that tensor's slot of `saved` only what the gathering GEMMs read -- M row
// indices and the three coordinate planes of Wx -- instead of M x O1 floats (1 GB at the largest scale, B = 64); and the slot of
// the data gradient into that layer in the backward scratch holds, when the gradient is summed inside the GEMM
// (gather_sum_layout: an input gradient is wanted, d->dy set), the pieces matrix, the permutation, the permuted source points
// and relative coordinates.  Both predicates are pure functions of the descriptor's SHAPE fields (and of d->dy being set or
// not for the second): cmf_setconv_sizes, the forward and the backward call must agree on them, so the pointer-alignment
// conditions of the gathering kernels are NOT part of them -- a block with the compact layout and a misaligned y is refused
// (CMF_CHECK_ARG in the calls) instead of overrunning a slot that was never sized for the materialised tensor.
// CMF_TRAIN_GATHER=0 / CMF_TRAIN_GATHER_SUM=0 (A/B) bring the materialised sizes back.
inline bool gather_layout(const cmf_setconv_desc *d)
{
    static const bool on = !(getenv("CMF_TRAIN_GATHER") && getenv("CMF_TRAIN_GATHER")[0] == '0');
    const long long M = (long long)d->B * d->N * d->S;
    return on && M % 128 == 0 && M < (1ll << 31) && d->C[0] % 128 == 0 && d->O1 % 128 == 0 && d->ldy % 4 == 0;
}
inline bool gather_sum_layout(const cmf_setconv_desc *d)
{
    static const bool on = !(getenv("CMF_TRAIN_GATHER_SUM") && getenv("CMF_TRAIN_GATHER_SUM")[0] == '0');
    return on && gather_layout(d) && d->dy != nullptr && d->S >= 2;
}

// BN backward of the layer behind the gathered first layer inside its weight-gradient GEMM (cmf_gemm_dw_gather_bn_bwd): the stand-alone
// pass over (M, C2) -- read dU, read Z, write dZ: 3 GB per step at B = 64 -- goes, dZ lands in a buffer of its own for the data-gradient
// GEMM (other column tiles of the slab still read dU).  Bit-identical to the two-kernel form.  [measured, round 6, A/B/A/B on one box:
// 18.999 / 19.035 -> 18.809 / 18.844 ms per step; the four GEMMs that absorb the pass run 2 points further from their MFMA roof --
// isolated cmf_gemm 0.739 -> 0.719 -- which is what the round-3 form of this fusion (CMF_BNB_FUSED, non-gathering) was left opt-in for:
// the step is the figure of merit.]  CMF_BNB_GATHER=0: the stand-alone pass (A/B).
inline bool bnb_gather(const cmf_setconv_desc *d)
{
    static const bool on = !(getenv("CMF_BNB_GATHER") && getenv("CMF_BNB_GATHER")[0] == '0');
    return on && d->training && gather_layout(d) && ((long long)d->B * d->N * d->S) >= 32768;
}

Layout make_layout(const cmf_setconv_desc *d, float *saved, float *scratch, bool backward)
{
    Layout L;
    const long long P = (long long)d->B * d->N, M = P * d->S;
    const int O1 = d->O1, C2 = d->C[0], C3 = d->C[1], C4 = d->C[2], C5 = d->C[3], C6 = d->C[4];
    Bump s(saved);
    L.idx = (int *)s.take(M);
    L.offsets = (int *)s.take((size_t)d->B * (d->N + 1));
    L.inv = (int *)s.take(M);
    L.dxyz = s.take(M * 4);
    L.z1 = s.take(gather_layout(d) ? (size_t)((M + 3) / 4 * 4) + 3 * (size_t)O1 : (size_t)M * O1);
    L.z2 = s.take(M * C2);
    L.z3 = s.take(M * C3);
    L.argmax = (unsigned char *)s.take((P * C3 + 3) / 4);
    L.x = s.take(P * C3);
    L.z4 = s.take(P * C4);
    L.z5 = s.take(P * C5);
    L.z6 = s.take(P * C6);
    L.fwd_sums = s.take(3 * O1 + 4);
    for (int l = 0; l < 6; ++l) L.bn[l] = s.take(4 * (size_t)chan(d, l));
    L.zsel = s.take(P * C3);       // (behind everything else: the offsets of the older slots are part of the Python side's contract)
    L.saved_floats = s.off;

    Bump t(scratch);
    const int cmax = std::max({O1, C2, C3, C4, C5, C6});
    L.partial = t.take((size_t)tiles128(M) * 5 * cmax);
    L.partial_x = t.take((size_t)tiles128(M) * (3 * O1 + 4));
    L.sums = t.take(5 * (size_t)cmax);
    if (backward) {
        L.t6 = t.take(P * C6);
        L.t5 = t.take(P * C5);
        L.t4 = t.take(P * C4);
        L.dx = t.take(P * C3);
        L.dU3 = t.take(M * C3);
        L.dU2 = t.take(M * C2);
        // (summed inside the GEMM: pieces (P + M/64) x O1 | perm M | source points M | relative coordinates 4 M -- sum_slots below)
        L.dU1 = t.take(gather_sum_layout(d) ? (size_t)(P + M / 64) * O1 + 6 * (size_t)M : (size_t)M * O1);
        L.dZ2 = (d->training && (bnb_fusable(M, C2, O1) || bnb_gather(d))) ? t.take(M * C2) : nullptr;
        size_t sk = 0;
        sk = std::max(sk, (size_t)dw_split(P, C6, C5) * C6 * C5);
        sk = std::max(sk, (size_t)dw_split(P, C5, C4) * C5 * C4);
        sk = std::max(sk, (size_t)dw_split(P, C4, C3) * C4 * C3);
        sk = std::max(sk, (size_t)dw_split(M, C3, C2) * C3 * C2);
        sk = std::max(sk, (size_t)dw_split(M, C2, O1) * C2 * O1);
        sk = std::max(sk, (size_t)dwg_split(M, C2, O1) * C2 * O1);
        // the fused single-pass layers write their own slab counts (one slab per workgroup): size for those as well --
        // the wide form's tiles128(M) slabs of 64 x C2 exceed the dw_split-based sizes for 128 <= M < 1024 rows
        const struct { long long rows; int cout, cin; } lay[5] = {{P, C6, C5}, {P, C5, C4}, {P, C4, C3}, {M, C3, C2}, {M, C2, O1}};
        for (const auto &l : lay)
            if (cmf_thin_bwd_supported(l.cout, l.cin)) sk = std::max(sk, (size_t)cmf_thin_bwd_slabs(l.rows, nullptr) * l.cout * l.cin);
        if (cmf_thin_bwd_wide_supported(C3, C2)) sk = std::max(sk, (size_t)cmf_thin_bwd_wide_slabs(M, C2, nullptr) * 64 * C2);
        if (cmf_setconv_chain_supported(d->N, d->S, O1, C2, C3, M)) sk = std::max(sk, (size_t)(cmf_setconv_chain_waves(M, 1) / 4) * C3 * C2);
        L.splitk = t.take(sk);
    } else {
        L.t6 = L.t5 = L.t4 = L.dx = L.dU3 = L.dU2 = L.dU1 = L.splitk = L.dZ2 = nullptr;
    }
    L.scratch_floats = t.off;
    return L;
}

#define CMF_TRY(call) do { int e_ = (call); if (e_) return e_; } while (0)

}  // namespace
// csrc/setconv_chain.hip (internal): the narrow block's neighbour-slot layers as register-chain passes
// (mode: 0 inference, 1 / 2 statistics of z2 / z3, 3 max over the ball + argmax + selected pre-activations, 4 / 5 backward of layers 3 / 2)
bool cmf_setconv_chain_supported(int N, int S, int O1, int C2, int C3, long long M);
long long cmf_setconv_chain_waves(long long M, int backward);
int cmf_setconv_chain_pass(int mode, long long M, int N, int S, const int *idx, const float *xyz, const float *dxyz, const float *y, long long ldy, const float *wx,
                           long long ldwx, const float *bn0, const float *bn1, const float *bn2, const float *w2, const float *w3, float *out,
                           long long ldo, float *zsel, unsigned char *argmax, float *partial, const float *g, const float *sums,
                           const float *dU_in, float *dU_out, float *slabs, void *stream);
// csrc/pointwise.hip (internal)
int cmf_bn_relu_maxpool_sel(long long P, int S, int C, const float *z, const float *a, const float *c, float *out, long long ldo,
                            unsigned char *argmax, float *zsel, void *stream);
void cmf_gemm_dx_gather_sum_hint(long long points);
int cmf_maxpool_bwd_point_sel(long long P, int C, const float *dout, long long ldd, const float *zsel, const float *a, const float *c,
                              const float *mean, const float *invstd, float *g, float *partial, void *stream);
int cmf_setconv_chain_infer(long long M, int N, int S, const int *idx, const float *xyz, const float *y, long long ldy, const float *wx,
                            long long ldwx, const float *bn0, const float *bn1, const float *bn2, const float *w2, const float *w3, float *out,
                            long long ldo, void *stream);
// csrc/pointwise.hip (internal): cmf_colsum with the first 2*C columns stored to dst0 / dst1
int cmf_colsum_store(int tiles, int ncols, const float *partial, float *out, int C, float *dst0, float *dst1, void *stream);
namespace {

// Eval mode: the six folds depend on nothing the block computes (running statistics), so they are ONE launch at the head
// of the chain instead of six tiny kernels between its GEMMs.  Same arithmetic as bn_finalize_kernel's eval branch.
struct FoldAll { int C[6]; const float *gamma[6], *beta[6], *rmean[6], *rvar[6]; float eps[6]; float *out[6]; };

__global__ __launch_bounds__(256) void bn_fold_eval_kernel(const FoldAll f)
{
    const int l = blockIdx.x, C = f.C[l];
    float *b = f.out[l];
    for (int ch = threadIdx.x; ch < C; ch += 256) {
        const double mean = f.rmean[l][ch], var = f.rvar[l][ch];
        const double invstd = 1.0 / sqrt(var + (double)f.eps[l]);
        const double a = (f.gamma[l] ? (double)f.gamma[l][ch] : 1.0) * invstd;
        b[ch] = (float)mean; b[C + ch] = (float)invstd; b[2 * C + ch] = (float)a;
        b[3 * C + ch] = (float)((f.beta[l] ? (double)f.beta[l][ch] : 0.0) - mean * a);
    }
}

__global__ __launch_bounds__(256) void bn_fold_eval_batch_kernel(const CmfBatch<FoldAll> b)
{
    const FoldAll &f = b.a[blockIdx.y];
    const int l = blockIdx.x, C = f.C[l];
    float *o = f.out[l];
    for (int ch = threadIdx.x; ch < C; ch += 256) {
        const double mean = f.rmean[l][ch], var = f.rvar[l][ch];
        const double invstd = 1.0 / sqrt(var + (double)f.eps[l]);
        const double a = (f.gamma[l] ? (double)f.gamma[l][ch] : 1.0) * invstd;
        o[ch] = (float)mean; o[C + ch] = (float)invstd; o[2 * C + ch] = (float)a;
        o[3 * C + ch] = (float)((f.beta[l] ? (double)f.beta[l][ch] : 0.0) - mean * a);
    }
}

int fold_all_eval(const cmf_setconv_desc *d, const Layout &L, void *st)
{
    FoldAll f;
    for (int l = 0; l < 6; ++l) {
        CMF_CHECK_ARG(d->rmean[l] && d->rvar[l]);
        f.C[l] = chan(d, l); f.gamma[l] = d->gamma[l]; f.beta[l] = d->beta[l]; f.rmean[l] = d->rmean[l]; f.rvar[l] = d->rvar[l];
        f.eps[l] = d->eps[l]; f.out[l] = L.bn[l];
    }
    hipLaunchKernelGGL(bn_fold_eval_kernel, dim3(6), dim3(256), 0, (hipStream_t)st, f);
    return cmf_launch_status();
}

// BN fold of layer l from `partial` (train: batch statistics + running-stat update) into L.bn[l]; eval: done up front
int fold(const cmf_setconv_desc *d, const Layout &L, int l, long long rows, void *st)
{
    if (!d->training) return 0;
    const int C = chan(d, l);
    float *b = L.bn[l];
    return cmf_bn_finalize(tiles128(rows), C, (double)rows, L.partial, d->gamma[l], d->beta[l], d->eps[l], d->momentum[l],
                           d->rmean[l], d->rvar[l], b, b + C, b + 2 * C, b + 3 * C, d->nbt[l], st);
}

// Z_out = act_{l_in}(Z_in) @ W^T (+ statistics), Z_in activated by BN layer l_in (or already active if l_in < 0)
int fwd_gemm(const cmf_setconv_desc *d, const Layout &L, long long rows, int cin, int cout, const float *zin, int l_in,
             const float *w, float *zout, void *st)
{
    const float *pa = l_in >= 0 ? L.bn[l_in] + 2 * chan(d, l_in) : nullptr;
    const float *pc = l_in >= 0 ? L.bn[l_in] + 3 * chan(d, l_in) : nullptr;
    return cmf_gemm((int)rows, cout, cin, 0, 1, zin, cin, w, cin, zout, cout, pa, pc, nullptr, nullptr, nullptr, 0,
                    d->training ? L.partial : nullptr, 0, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, 1, nullptr, 0, st);
}

// dW[cout,cin] (+)= dZ^T @ act_{l_in}(X)
int dw_gemm(const cmf_setconv_desc *d, const Layout &L, long long rows, int cout, int cin, const float *dZ, const float *x, int l_in,
            float *dw, int accumulate, void *st)
{
    if (!dw) return 0;
    const float *pa = l_in >= 0 ? L.bn[l_in] + 2 * chan(d, l_in) : nullptr;
    const float *pc = l_in >= 0 ? L.bn[l_in] + 3 * chan(d, l_in) : nullptr;
    const int split = dw_split(rows, cout, cin);
    return cmf_gemm(cout, cin, (int)rows, 1, 0, dZ, cout, x, cin, dw, cin, nullptr, nullptr, pa, pc, nullptr, 0, nullptr, 0,
                    nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, split, split > 1 ? L.splitk : nullptr, accumulate, st);
}

// dU_x = (dZ @ W) masked by x's BN+ReLU (layer l_x), with the BN-backward partial sums (and the dxyz sums when q != 0)
int dx_gemm(const cmf_setconv_desc *d, const Layout &L, long long rows, int cout, int cin, const float *dZ, const float *w,
            const float *x, int l_x, float *dU, const float *dxyz, void *st)
{
    if (l_x < 0)
        return cmf_gemm((int)rows, cin, cout, 0, 0, dZ, cout, w, cin, dU, cin, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr,
                        0, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, 1, nullptr, 0, st);
    const int C = chan(d, l_x);
    const float *b = L.bn[l_x];
    return cmf_gemm((int)rows, cin, cout, 0, 0, dZ, cout, w, cin, dU, cin, nullptr, nullptr, nullptr, nullptr, nullptr, 0, L.partial,
                    1, x, cin, b + 2 * C, b + 3 * C, b, b + C, dxyz, 1, nullptr, 0, st);
}

// One backward layer: the BN-backward sums of layer l_out (dgamma / dbeta), then dZ -> dW (+ dU of the layer below).
// Narrow layers run the fused single-pass kernel (cmf_thin_bwd_layer: dZ is never stored); the wide ones of the second
// encoder the BN backward in place followed by the two tiled GEMMs.  CMF_THIN_FUSED=0 forces the three-kernel form.
int bwd_layer(const cmf_setconv_desc *d, const Layout &L, int l_out, long long rows, int cout, int cin, float *dU, const float *z_out,
              const float *w, const float *x, int l_in, float *dU_in, const float *dxyz, float *dw, int acc_w, void *st,
              float *dz_buf = nullptr)
{
    const float *b = L.bn[l_out];
    if (d->acc_bn[l_out]) CMF_TRY(cmf_colsum_finalize(tiles128(rows), cout, L.partial, L.sums, d->dbeta[l_out], d->dgamma[l_out], st));
    else CMF_TRY(cmf_colsum_store(tiles128(rows), 2 * cout, L.partial, L.sums, cout, d->dbeta[l_out], d->dgamma[l_out], st));
    static const bool fused = !(getenv("CMF_THIN_FUSED") && getenv("CMF_THIN_FUSED")[0] == '0');
    if (fused && cmf_thin_bwd_supported(cout, cin)) {
        const float *bi = l_in >= 0 ? L.bn[l_in] : nullptr;
        const int ci = l_in >= 0 ? chan(d, l_in) : 0;
        return cmf_thin_bwd_layer(rows, cout, cin, dU, cout, z_out, cout, b + 2 * cout, b, b + cout, d->training ? L.sums : nullptr,
                                  w, cin, x, cin, l_in >= 0 ? 1 : 0, bi ? bi + 2 * ci : nullptr, bi ? bi + 3 * ci : nullptr, bi,
                                  bi ? bi + ci : nullptr, dxyz, dU_in, cin, l_in >= 0 ? L.partial : nullptr, dw, cin, acc_w, L.splitk, st);
    }
    if (dz_buf && dw && d->training && l_in >= 0 && bnb_fusable(rows, cout, cin)) {
        // BN backward inside the weight-gradient GEMM's operand staging; dZ lands in dz_buf for the data-gradient GEMM
        const float *bi = L.bn[l_in];
        const int ci = chan(d, l_in), split = dw_split(rows, cout, cin);
        CMF_TRY(cmf_gemm_dw_bn_bwd(cout, cin, rows, dU, cout, z_out, cout, b + 2 * cout, b, b + cout, L.sums, dz_buf, cout, x, cin,
                                   bi + 2 * ci, bi + 3 * ci, dw, cin, split, split > 1 ? L.splitk : nullptr, acc_w, st));
        return dx_gemm(d, L, rows, cout, cin, dz_buf, w, x, l_in, dU_in, dxyz, st);
    }
    CMF_TRY(cmf_bn_bwd_apply(rows, cout, dU, z_out, cout, b + 2 * cout, b, b + cout, d->training ? L.sums : nullptr, st));
    CMF_TRY(dw_gemm(d, L, rows, cout, cin, dU, x, l_in, dw, acc_w, st));
    return dx_gemm(d, L, rows, cout, cin, dU, w, x, l_in, dU_in, dxyz, st);
}

// Training without the grouped first-layer tensor (second-encoder shapes; CMF_TRAIN_GATHER=0 keeps the tensor: A/B): the forward pass takes the layer's
// statistics from cmf_group_affine's statistics-only form and the three GEMMs that read the tensor form it from the per-point rows
// (cmf_gemm_gather_affine / cmf_gemm_dx_gather / cmf_gemm_dw_gather: each bit-identical to its materialised counterpart); the
// tensor's slot in `saved` holds the row indices and the coordinate planes of Wx.
bool train_gather(const cmf_setconv_desc *d)
{
    // (train-mode or eval-mode BatchNorm alike: with eval-mode statistics -- the regime of every epoch after the first in the reference's
    //  training loop -- the forward pass is the inference form and the backward pass the same two gathering GEMMs)
    // = the layout predicate: the slot of the tensor is sized for the gathering form, there is no other form to fall back to
    return gather_layout(d);
}
// the gathering kernels' 16-byte loads: with the compact layout a misaligned operand is an argument error, not a fallback
bool gather_operands_aligned(const cmf_setconv_desc *d) { return (((uintptr_t)d->y | (uintptr_t)d->w[0]) & 15) == 0; }
int *gather_rows(const Layout &L) { return reinterpret_cast<int *>(L.z1); }
float *gather_wx3(const Layout &L, long long M) { return L.z1 + (M + 3) / 4 * 4; }

// bwd_layer for the layer behind the gathered first layer: BN-backward sums, BN backward in place, weight gradient and masked data
// gradient with the first layer formed in their operand / epilogue paths
// ... and, when the input gradient is wanted, without the data gradient into the first layer either (M x O1 again): the GEMM walks the
// slots in inverse-index order and reduces runs of equal source points in its epilogue (cmf_gemm_dx_gather_sum); the scatter at the end
// of the block adds a point's pieces.  CMF_TRAIN_GATHER_SUM=0 keeps the stored gradient (A/B).  The slot of dU1 holds the pieces, the
// permutation, the permuted source points and relative coordinates.
bool train_gather_sum(const cmf_setconv_desc *d) { return gather_sum_layout(d); }
struct SumSlots { float *pieces; int *perm, *pts; float *dq2; };
SumSlots sum_slots(const cmf_setconv_desc *d, const Layout &L)
{
    const long long P = (long long)d->B * d->N, M = P * d->S;
    SumSlots s;
    s.pieces = L.dU1;
    s.perm = reinterpret_cast<int *>(L.dU1 + (P + M / 64) * d->O1);
    s.pts = s.perm + M;
    s.dq2 = reinterpret_cast<float *>(s.pts + M);
    return s;
}

int bwd_layer1_gather(const cmf_setconv_desc *d, const Layout &L, long long M, int C2, int O1, void *st)
{
    const float *b = L.bn[1], *b0 = L.bn[0];
    if (d->acc_bn[1]) CMF_TRY(cmf_colsum_finalize(tiles128(M), C2, L.partial, L.sums, d->dbeta[1], d->dgamma[1], st));
    else CMF_TRY(cmf_colsum_store(tiles128(M), 2 * C2, L.partial, L.sums, C2, d->dbeta[1], d->dgamma[1], st));
    const float *dZ2 = L.dU2;
    if (bnb_gather(d) && L.dZ2 && d->dw[0]) {
        const int split = dwg_split(M, C2, O1);
        CMF_TRY(cmf_gemm_dw_gather_bn_bwd(C2, O1, M, L.dU2, C2, L.z2, C2, b + 2 * C2, b, b + C2, L.sums, L.dZ2, C2, d->y, d->ldy, gather_rows(L),
                                          L.dxyz, gather_wx3(L, M), b0 + 2 * O1, b0 + 3 * O1, d->dw[0], O1, split, split > 1 ? L.splitk : nullptr,
                                          d->acc_w[0], st));
        dZ2 = L.dZ2;
    } else {
    CMF_TRY(cmf_bn_bwd_apply(M, C2, L.dU2, L.z2, C2, b + 2 * C2, b, b + C2, d->training ? L.sums : nullptr, st));
    if (d->dw[0]) {
        const int split = dwg_split(M, C2, O1);
        CMF_TRY(cmf_gemm_dw_gather(C2, O1, M, L.dU2, C2, d->y, d->ldy, gather_rows(L), L.dxyz, gather_wx3(L, M), b0 + 2 * O1, b0 + 3 * O1,
                                   d->dw[0], O1, split, split > 1 ? L.splitk : nullptr, d->acc_w[0], st));
    }
    }
    if (train_gather_sum(d)) {
        const SumSlots q = sum_slots(d, L);
        CMF_TRY(cmf_group_perm(d->B, d->N * d->S, L.inv, gather_rows(L), L.dxyz, q.perm, q.pts, q.dq2, st));
        cmf_gemm_dx_gather_sum_hint((long long)d->B * d->N);
        return cmf_gemm_dx_gather_sum((int)M, O1, C2, dZ2, C2, d->w[0], O1, d->y, d->ldy, q.perm, q.pts, q.dq2, gather_wx3(L, M),
                                      b0 + 2 * O1, b0 + 3 * O1, b0, b0 + O1, q.pieces, L.partial, st);
    }
    return cmf_gemm_dx_gather((int)M, O1, C2, dZ2, C2, d->w[0], O1, L.dU1, O1, d->y, d->ldy, gather_rows(L), L.dxyz, gather_wx3(L, M),
                              b0 + 2 * O1, b0 + 3 * O1, b0, b0 + O1, L.partial, st);
}

}  // namespace

extern "C" int cmf_setconv_sizes(const cmf_setconv_desc *d, long long *saved_floats, long long *scratch_fwd, long long *scratch_bwd)
{
    CMF_CHECK_ARG(d && d->B > 0 && d->N > 0 && d->S > 0 && d->S <= 64 && d->O1 % 4 == 0);
    for (int i = 0; i < 5; ++i) CMF_CHECK_ARG(d->C[i] > 0 && d->C[i] % 4 == 0);
    Layout f = make_layout(d, nullptr, nullptr, false), b = make_layout(d, nullptr, nullptr, true);
    if (saved_floats) *saved_floats = (long long)f.saved_floats;
    if (scratch_fwd) *scratch_fwd = (long long)f.scratch_floats;
    if (scratch_bwd) *scratch_bwd = (long long)b.scratch_floats;
    return 0;
}

// part: 0 whole block, 1 up to the max over the ball (the per-point tail is left to cmf_setconv_tail_forward)
static int setconv_forward_part(const cmf_setconv_desc *d, void *st, int part)
{
    CMF_CHECK_ARG(d && d->xyz && d->y && d->wx && d->saved && d->scratch && d->out);
    CMF_CHECK_ARG(!gather_layout(d) || gather_operands_aligned(d));
    const Layout L = make_layout(d, d->saved, d->scratch, false);
    const long long P = (long long)d->B * d->N, M = P * d->S;
    const int O1 = d->O1, C2 = d->C[0], C3 = d->C[1], C4 = d->C[2], C5 = d->C[3], C6 = d->C[4];
    // idx is pre-zeroed like the reference's BallQuery.forward (every point is its own neighbour here, but keep the contract)
    if (!d->training) CMF_TRY(fold_all_eval(d, L, st));
    if (!d->idx_ready) CMF_TRY(cmf_ball_query_defined(d->B, d->N, d->N, d->radius, d->S, d->xyz, d->xyz, L.idx, st));
    // inference (eval-mode BN and no backward call to follow): the grouped first-layer tensor z1 (M x O1: 1 GB at the largest scale of
    // the second encoder) is never written -- the second layer's GEMM gathers the per-point rows and forms the layer in its A-operand
    // path (cmf_gemm_gather_affine, bit-identical); the slot of z1 holds the M source-row indices and the coordinate planes of Wx
    // inference of the narrow block (first encoder): layers 1-3 and the max over the ball as ONE register-chain kernel
    // (csrc/setconv_chain.hip; every layer bit-identical to the per-layer kernels) -- nothing but the pooled rows is written
    if (!d->training && d->inference && cmf_setconv_chain_supported(d->N, d->S, O1, C2, C3, M) && d->ldy % 4 == 0 && ((uintptr_t)d->y & 15) == 0) {
        CMF_TRY(cmf_setconv_chain_infer(M, d->N, d->S, L.idx, d->xyz, d->y, d->ldy, d->wx, d->ldwx, L.bn[0], L.bn[1], L.bn[2], d->w[0], d->w[1],
                                        L.x, C3, st));
        if (part == 1) return 0;
        CMF_TRY(fwd_gemm(d, L, P, C3, C4, L.x, -1, d->w[2], L.z4, st));
        CMF_TRY(fwd_gemm(d, L, P, C4, C5, L.z4, 3, d->w[3], L.z5, st));
        CMF_TRY(fwd_gemm(d, L, P, C5, C6, L.z5, 4, d->w[4], L.z6, st));
        return cmf_affine_relu(P, C6, L.z6, C6, L.bn[5] + 2 * C6, L.bn[5] + 3 * C6, d->out, d->ldo, st);
    }
    if (chain_train(d)) {
        // training (batch statistics: a pass per BatchNorm layer, each recomputing the layers below it) or eval-mode BN with a backward
        // pass to follow (one pass): nothing but the pooled rows, their argmax slots and the selected pre-activations is stored
        const int wv = (int)cmf_setconv_chain_waves(M, 0);
        auto pass = [&](int mode) {
            return cmf_setconv_chain_pass(mode, M, d->N, d->S, L.idx, d->xyz, d->training ? L.dxyz : nullptr, d->y, d->ldy, d->wx, d->ldwx, L.bn[0], L.bn[1], L.bn[2], d->w[0],
                                          d->w[1], L.x, C3, L.z3, L.argmax, L.partial, nullptr, nullptr, nullptr, nullptr, nullptr, st);
        };
        if (d->training) {
            CMF_TRY(cmf_group_affine(d->B, d->N, d->N, d->S, O1, d->y, (int)d->ldy, nullptr, 0, d->xyz, d->xyz, d->wx, (int)d->ldwx, L.idx, 0,
                                     nullptr, L.dxyz, L.partial, L.partial_x, st));
            CMF_TRY(cmf_colsum(tiles128(M), 3 * O1 + 4, L.partial_x, L.fwd_sums, 0, nullptr, nullptr, st));
            CMF_TRY(fold(d, L, 0, M, st));
            for (int l = 1; l <= 2; ++l) {
                CMF_TRY(pass(l));
                const int C = chan(d, l);
                float *b = L.bn[l];
                CMF_TRY(cmf_bn_finalize(wv, C, (double)M, L.partial, d->gamma[l], d->beta[l], d->eps[l], d->momentum[l], d->rmean[l], d->rvar[l],
                                        b, b + C, b + 2 * C, b + 3 * C, d->nbt[l], st));
            }
        }
        CMF_TRY(pass(3));
        if (part == 1) return 0;
        CMF_TRY(fwd_gemm(d, L, P, C3, C4, L.x, -1, d->w[2], L.z4, st));
        CMF_TRY(fold(d, L, 3, P, st));
        CMF_TRY(fwd_gemm(d, L, P, C4, C5, L.z4, 3, d->w[3], L.z5, st));
        CMF_TRY(fold(d, L, 4, P, st));
        CMF_TRY(fwd_gemm(d, L, P, C5, C6, L.z5, 4, d->w[4], L.z6, st));
        CMF_TRY(fold(d, L, 5, P, st));
        return cmf_affine_relu(P, C6, L.z6, C6, L.bn[5] + 2 * C6, L.bn[5] + 3 * C6, d->out, d->ldo, st);
    }
    const bool gather = !d->training && ((d->inference && M % 128 == 0 && C2 % 128 == 0 && O1 % 16 == 0 && d->ldy % 4 == 0 &&
                                          gather_operands_aligned(d) && M < (1ll << 31)) || train_gather(d));
    if (d->training && train_gather(d)) {
        CMF_TRY(cmf_group_affine(d->B, d->N, d->N, d->S, O1, d->y, (int)d->ldy, nullptr, 0, d->xyz, d->xyz, d->wx, (int)d->ldwx, L.idx, 0,
                                 nullptr, L.dxyz, L.partial, L.partial_x, st));
        CMF_TRY(cmf_colsum(tiles128(M), 3 * O1 + 4, L.partial_x, L.fwd_sums, 0, nullptr, nullptr, st));
        CMF_TRY(fold(d, L, 0, M, st));
        CMF_TRY(cmf_group_prep(d->B, d->N, d->N, d->S, O1, d->xyz, d->xyz, d->wx, (int)d->ldwx, L.idx, gather_rows(L), L.dxyz,
                               gather_wx3(L, M), st));
        CMF_TRY(cmf_gemm_gather_affine((int)M, C2, O1, d->y, d->ldy, gather_rows(L), L.dxyz, gather_wx3(L, M), L.bn[0] + 2 * O1,
                                       L.bn[0] + 3 * O1, d->w[0], O1, L.z2, C2, L.partial, st));
    } else
    if (gather) {
        int *rows = reinterpret_cast<int *>(L.z1);
        float *wx3 = L.z1 + (M + 3) / 4 * 4;
        CMF_TRY(cmf_group_prep(d->B, d->N, d->N, d->S, O1, d->xyz, d->xyz, d->wx, (int)d->ldwx, L.idx, rows, L.dxyz, wx3, st));
        CMF_TRY(cmf_gemm_gather_affine((int)M, C2, O1, d->y, d->ldy, rows, L.dxyz, wx3, L.bn[0] + 2 * O1, L.bn[0] + 3 * O1, d->w[0], O1,
                                       L.z2, C2, nullptr, st));
    } else {
    CMF_TRY(cmf_group_affine(d->B, d->N, d->N, d->S, O1, d->y, (int)d->ldy, nullptr, 0, d->xyz, d->xyz, d->wx, (int)d->ldwx, L.idx, 0,
                             L.z1, L.dxyz, d->training ? L.partial : nullptr, d->training ? L.partial_x : nullptr, st));
    if (d->training) CMF_TRY(cmf_colsum(tiles128(M), 3 * O1 + 4, L.partial_x, L.fwd_sums, 0, nullptr, nullptr, st));
    CMF_TRY(fold(d, L, 0, M, st));
    CMF_TRY(fwd_gemm(d, L, M, O1, C2, L.z1, 0, d->w[0], L.z2, st));
    }
    CMF_TRY(fold(d, L, 1, M, st));
    CMF_TRY(fwd_gemm(d, L, M, C2, C3, L.z2, 1, d->w[1], L.z3, st));
    CMF_TRY(fold(d, L, 2, M, st));
    CMF_TRY(cmf_bn_relu_maxpool_sel(P, d->S, C3, L.z3, L.bn[2] + 2 * C3, L.bn[2] + 3 * C3, L.x, C3, L.argmax, L.zsel, st));
    if (part == 1) return 0;
    // timing diagnostic (results are garbage): the per-point tail left out -- an upper bound for what fusing it can save
    CMF_TRY(fwd_gemm(d, L, P, C3, C4, L.x, -1, d->w[2], L.z4, st));
    CMF_TRY(fold(d, L, 3, P, st));
    CMF_TRY(fwd_gemm(d, L, P, C4, C5, L.z4, 3, d->w[3], L.z5, st));
    CMF_TRY(fold(d, L, 4, P, st));
    CMF_TRY(fwd_gemm(d, L, P, C5, C6, L.z5, 4, d->w[4], L.z6, st));
    CMF_TRY(fold(d, L, 5, P, st));
    return cmf_affine_relu(P, C6, L.z6, C6, L.bn[5] + 2 * C6, L.bn[5] + 3 * C6, d->out, d->ldo, st);
}

extern "C" int cmf_setconv_forward(const cmf_setconv_desc *d, void *st) { return setconv_forward_part(d, st, 0); }

extern "C" int cmf_setconv_queries(int n, const cmf_setconv_desc *descs, void *stream)
{
    CMF_CHECK_ARG(n >= 0 && (n == 0 || descs));
    int i = 0;
    while (i < n) {
        // a run of blocks over the same cloud (the scales of one encoder call), then -- if the next run has the same shape -- a second cloud
        const cmf_setconv_desc &d0 = descs[i];
        CMF_CHECK_ARG(d0.xyz && d0.saved);
        int nq = 1;
        while (i + nq < n && nq < 4 && descs[i + nq].xyz == d0.xyz && descs[i + nq].B == d0.B && descs[i + nq].N == d0.N) ++nq;
        int nclouds = 1;
        if (i + 2 * nq <= n && descs[i + nq].xyz != d0.xyz && descs[i + nq].B == d0.B && descs[i + nq].N == d0.N) {
            bool same = true;
            for (int q = 0; q < nq && same; ++q) {
                const cmf_setconv_desc &u = descs[i + q], &v = descs[i + nq + q];
                same = v.xyz == descs[i + nq].xyz && v.radius == u.radius && v.S == u.S && v.B == u.B && v.N == u.N;
            }
            if (same) nclouds = 2;
        }
        int ns_chk[4];
        for (int q = 0; q < nq; ++q) ns_chk[q] = descs[i + q].S;
        if (!cmf_ball_query_multi_takes(d0.N, nq, ns_chk)) {  // large clouds (cell grid) / lists too long for one launch: the single-scale queries
            for (int q = 0; q < nq * nclouds; ++q) {
                const cmf_setconv_desc &d = descs[i + q];
                const Layout L = make_layout(&d, d.saved, nullptr, false);
                CMF_TRY(cmf_ball_query_defined(d.B, d.N, d.N, d.radius, d.S, d.xyz, d.xyz, L.idx, stream));
            }
        } else {
            float radii[4];
            int ns[4];
            const float *ctr[2], *cloud[2];
            int *idx[8];
            for (int c = 0; c < nclouds; ++c) {
                ctr[c] = cloud[c] = descs[i + c * nq].xyz;
                for (int q = 0; q < nq; ++q) {
                    const cmf_setconv_desc &d = descs[i + c * nq + q];
                    CMF_CHECK_ARG(d.saved);
                    idx[c * nq + q] = make_layout(&d, d.saved, nullptr, false).idx;
                    if (c == 0) { radii[q] = d.radius; ns[q] = d.S; }
                }
            }
            CMF_TRY(cmf_ball_query_multi(d0.B, d0.N, d0.N, nq, radii, ns, nclouds, ctr, cloud, idx, 1, stream));
        }
        i += nq * nclouds;
    }
    return 0;
}

// ---- the per-point tails of n blocks in batched launches (cmf_common.h "batched launches") ---------------------------
// Layers 4-6 of a block work on B*N rows of <= 64 channels: 128 workgroups of latency per kernel.  The blocks of an
// encoder call (4 scales, or 2 x 4 for the two clouds of the first encoder) run them as ONE launch per kernel of the
// sequence below -- the same kernels in the same order as cmf_setconv_forward / _backward, so results are bit-identical.
static bool tail_batchable(int n, const cmf_setconv_desc *descs)
{
    static const bool on = !(getenv("CMF_TAIL_BATCH") && getenv("CMF_TAIL_BATCH")[0] == '0');
    if (!on || n < 1 || n > CMF_MAX_BATCH) return false;
    for (int i = 0; i < n; ++i) {
        const cmf_setconv_desc &d = descs[i];
        for (int k = 1; k < 5; ++k)
            if (d.C[k] > 64 || d.C[k] % 32 || d.C[k] != descs[0].C[k]) return false;
        if (d.training != descs[0].training || ((long long)d.B * d.N) % 128) return false;
        for (int k = 2; k < 5; ++k) if (!d.dw[k] && descs[0].dw[k]) return false;
    }
    return true;
}

extern "C" int cmf_setconv_tail_forward(int n, const cmf_setconv_desc *descs, void *stream)
{
    CMF_CHECK_ARG(n >= 0 && (n == 0 || descs));
    if (n == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    if (!tail_batchable(n, descs)) {                    // one block after the other, unbatched kernels
        for (int i = 0; i < n; ++i) {
            const cmf_setconv_desc *d = &descs[i];
            const Layout L = make_layout(d, d->saved, d->scratch, false);
            const long long P = (long long)d->B * d->N;
            const int C3 = d->C[1], C4 = d->C[2], C5 = d->C[3], C6 = d->C[4];
            CMF_TRY(fwd_gemm(d, L, P, C3, C4, L.x, -1, d->w[2], L.z4, stream));
            CMF_TRY(fold(d, L, 3, P, stream));
            CMF_TRY(fwd_gemm(d, L, P, C4, C5, L.z4, 3, d->w[3], L.z5, stream));
            CMF_TRY(fold(d, L, 4, P, stream));
            CMF_TRY(fwd_gemm(d, L, P, C5, C6, L.z5, 4, d->w[4], L.z6, stream));
            CMF_TRY(fold(d, L, 5, P, stream));
            CMF_TRY(cmf_affine_relu(P, C6, L.z6, C6, L.bn[5] + 2 * C6, L.bn[5] + 3 * C6, d->out, d->ldo, stream));
        }
        return 0;
    }
    Layout L[CMF_MAX_BATCH];
    for (int i = 0; i < n; ++i) L[i] = make_layout(&descs[i], descs[i].saved, descs[i].scratch, false);
    const bool training = descs[0].training != 0;
    for (int layer = 0; layer < 3; ++layer) {                        // BN layer 3 + layer: (x | z4 | z5) -> (z4 | z5 | z6)
        GemmArgs g[CMF_MAX_BATCH];
        CmfBnFinArgs f[CMF_MAX_BATCH];
        for (int i = 0; i < n; ++i) {
            const cmf_setconv_desc &d = descs[i];
            const long long P = (long long)d.B * d.N;
            const int cin = d.C[1 + layer], cout = d.C[2 + layer], lb = 3 + layer;
            const float *zin = layer == 0 ? L[i].x : (layer == 1 ? L[i].z4 : L[i].z5);
            float *zout = layer == 0 ? L[i].z4 : (layer == 1 ? L[i].z5 : L[i].z6);
            GemmArgs &q = g[i];
            q = GemmArgs{};
            q.M = (int)P; q.N = cout; q.K = cin; q.A = zin; q.lda = cin; q.B = d.w[2 + layer]; q.ldb = cin; q.C = zout; q.ldc = cout;
            if (layer > 0) { q.pro_a = L[i].bn[lb - 1] + 2 * cin; q.pro_c = L[i].bn[lb - 1] + 3 * cin; }
            q.stats = training ? L[i].partial : nullptr;
            q.split_k = 1;
            float *b = L[i].bn[lb];
            f[i] = CmfBnFinArgs{tiles128(P), cout, (double)P, L[i].partial, d.gamma[lb], d.beta[lb], d.eps[lb], d.momentum[lb], d.rmean[lb],
                                d.rvar[lb], b, b + cout, b + 2 * cout, b + 3 * cout, d.nbt[lb]};
        }
        CMF_TRY(cmf_thin_fwd_batch(n, g, st));
        if (training) CMF_TRY(cmf_bn_finalize_batch(n, f, st));
    }
    CmfAffineArgs a[CMF_MAX_BATCH];
    for (int i = 0; i < n; ++i) {
        const cmf_setconv_desc &d = descs[i];
        const int C6 = d.C[4];
        a[i] = CmfAffineArgs{(long long)d.B * d.N, C6, L[i].z6, C6, L[i].bn[5] + 2 * C6, L[i].bn[5] + 3 * C6, d.out, d.ldo};
    }
    return cmf_affine_relu_batch(n, a, st);
}

// gradient of the tails: from dout to the gradient of the pooled features (Layout::dx), with the weight / BN gradients of
// layers 4-6; the rest of each block's backward is cmf_setconv_backward with part 2
extern "C" int cmf_setconv_tail_backward(int n, const cmf_setconv_desc *descs, void *stream)
{
    CMF_CHECK_ARG(n >= 0 && (n == 0 || descs));
    if (n == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    if (!tail_batchable(n, descs)) {
        for (int i = 0; i < n; ++i) {
            const cmf_setconv_desc *d = &descs[i];
            const Layout L = make_layout(d, d->saved, d->scratch, true);
            const long long P = (long long)d->B * d->N;
            const int C3 = d->C[1], C4 = d->C[2], C5 = d->C[3], C6 = d->C[4];
            const float *b5 = L.bn[5];
            CMF_TRY(cmf_act_bwd_stats(P, C6, d->dout, d->lddout, L.z6, C6, b5 + 2 * C6, b5 + 3 * C6, b5, b5 + C6, L.t6, L.partial, stream));
            CMF_TRY(bwd_layer(d, L, 5, P, C6, C5, L.t6, L.z6, d->w[4], L.z5, 4, L.t5, nullptr, d->dw[4], d->acc_w[4], stream));
            CMF_TRY(bwd_layer(d, L, 4, P, C5, C4, L.t5, L.z5, d->w[3], L.z4, 3, L.t4, nullptr, d->dw[3], d->acc_w[3], stream));
            CMF_TRY(bwd_layer(d, L, 3, P, C4, C3, L.t4, L.z4, d->w[2], L.x, -1, L.dx, nullptr, d->dw[2], d->acc_w[2], stream));
        }
        return 0;
    }
    Layout L[CMF_MAX_BATCH];
    for (int i = 0; i < n; ++i) L[i] = make_layout(&descs[i], descs[i].saved, descs[i].scratch, true);
    const bool training = descs[0].training != 0;
    {
        CmfActBwdArgs a[CMF_MAX_BATCH];
        for (int i = 0; i < n; ++i) {
            const cmf_setconv_desc &d = descs[i];
            const int C6 = d.C[4];
            const float *b5 = L[i].bn[5];
            a[i] = CmfActBwdArgs{(long long)d.B * d.N, C6, d.dout, d.lddout, L[i].z6, C6, b5 + 2 * C6, b5 + 3 * C6, b5, b5 + C6, L[i].t6, L[i].partial};
        }
        CMF_TRY(cmf_act_bwd_stats_batch(n, a, st));
    }
    for (int layer = 2; layer >= 0; --layer) {                       // BN layer 3 + layer, weights w[2 + layer]: cout <- cin
        CmfColsumArgs cs[CMF_MAX_BATCH];
        CmfThinBwdCall tb[CMF_MAX_BATCH];
        CmfSplitkArgs sk[CMF_MAX_BATCH];
        for (int i = 0; i < n; ++i) {
            const cmf_setconv_desc &d = descs[i];
            const long long P = (long long)d.B * d.N;
            const int cin = d.C[1 + layer], cout = d.C[2 + layer], lb = 3 + layer;
            float *dU = layer == 2 ? L[i].t6 : (layer == 1 ? L[i].t5 : L[i].t4);
            const float *zout = layer == 2 ? L[i].z6 : (layer == 1 ? L[i].z5 : L[i].z4);
            const float *x = layer == 2 ? L[i].z5 : (layer == 1 ? L[i].z4 : L[i].x);
            float *dUin = layer == 2 ? L[i].t5 : (layer == 1 ? L[i].t4 : L[i].dx);
            const float *b = L[i].bn[lb], *bi = layer > 0 ? L[i].bn[lb - 1] : nullptr;
            cs[i] = CmfColsumArgs{tiles128(P), 2 * cout, L[i].partial, L[i].sums, cout, d.dbeta[lb], d.dgamma[lb], d.acc_bn[lb] ? 0 : 1};
            CmfThinBwdCall &q = tb[i];
            q = CmfThinBwdCall{};
            q.rows = P; q.cout = cout; q.cin = cin; q.dU = dU; q.lddu = cout; q.z = zout; q.ldz = cout;
            q.a = b + 2 * cout; q.mean = b; q.invstd = b + cout; q.sums = training ? L[i].sums : nullptr;
            q.w = d.w[2 + layer]; q.ldw = cin; q.x = x; q.ldx = cin; q.in_mode = layer > 0 ? 1 : 0;
            if (bi) { q.a_in = bi + 2 * cin; q.c_in = bi + 3 * cin; q.mean_in = bi; q.invstd_in = bi + cin; }
            q.dx = dUin; q.lddx = cin; q.stats = layer > 0 ? L[i].partial : nullptr;
            q.dw = d.dw[2 + layer]; q.lddw = cin; q.accumulate = d.acc_w[2 + layer]; q.slabs = L[i].splitk;
        }
        CMF_TRY(cmf_colsum_batch(n, cs, st));
        CMF_TRY(cmf_thin_bwd_layer_batch(n, tb, st));
        for (int i = 0; i < n; ++i)
            sk[i] = CmfSplitkArgs{tb[i].cout, tb[i].cin, tb[i].nslab, tb[i].slabs, tb[i].dw, tb[i].lddw, tb[i].accumulate};
        CMF_TRY(cmf_splitk_reduce_batch(n, sk, st));
    }
    return 0;
}

// part: 0 whole block, 2 everything behind the per-point tail (cmf_setconv_tail_backward has produced Layout::dx)
static int setconv_backward_part(const cmf_setconv_desc *d, void *st, int part)
{
    CMF_CHECK_ARG(d && d->xyz && d->saved && d->scratch && d->dout);
    CMF_CHECK_ARG(!gather_layout(d) || (d->y && gather_operands_aligned(d)));
    const Layout L = make_layout(d, d->saved, d->scratch, true);
    const long long P = (long long)d->B * d->N, M = P * d->S;
    const int O1 = d->O1, C2 = d->C[0], C3 = d->C[1], C4 = d->C[2], C5 = d->C[3], C6 = d->C[4];
    const float *b5 = L.bn[5], *b2 = L.bn[2];
    // The inverse index of the grouping is only needed by the scatter at the very end, but its kernel wants most of a
    // CU's LDS (one workgroup per sample, 132 KB at N = 256): issued there it has to wait until a CU has drained the
    // GEMM workgroups of the other chains -- up to a tile's duration on the critical tail of the chain.  Issued first
    // it runs next to the small per-point kernels and is long done when the scatter needs it.
    if (d->dy) CMF_TRY(cmf_build_inverse_ps(d->B, d->N, d->N, d->S, L.idx, L.offsets, L.inv, st));
    // layer 6 .. 4 (per point)
    if (part != 2) {
    CMF_TRY(cmf_act_bwd_stats(P, C6, d->dout, d->lddout, L.z6, C6, b5 + 2 * C6, b5 + 3 * C6, b5, b5 + C6, L.t6, L.partial, st));
    CMF_TRY(bwd_layer(d, L, 5, P, C6, C5, L.t6, L.z6, d->w[4], L.z5, 4, L.t5, nullptr, d->dw[4], d->acc_w[4], st));
    CMF_TRY(bwd_layer(d, L, 4, P, C5, C4, L.t5, L.z5, d->w[3], L.z4, 3, L.t4, nullptr, d->dw[3], d->acc_w[3], st));
    CMF_TRY(bwd_layer(d, L, 3, P, C4, C3, L.t4, L.z4, d->w[2], L.x, -1, L.dx, nullptr, d->dw[2], d->acc_w[2], st));
    }
    // max over the ball, layers 3 .. 1 (per neighbour slot)
    static const bool fused = !(getenv("CMF_THIN_FUSED") && getenv("CMF_THIN_FUSED")[0] == '0');
    // the wide-input form of the fused layer (64 <- 256 channels, second encoder): CMF_THIN_WIDE=0 keeps max-pool backward, BN
    // backward and the two tiled GEMMs (A/B: 22.6 vs 22.9 ms per step)
    static const bool wide = !(getenv("CMF_THIN_WIDE") && getenv("CMF_THIN_WIDE")[0] == '0');
    if (chain_train(d)) {
        // the chain's backward: the pooled gradient per point with layer 3's BN-backward sums from the selected pre-activations, then two
        // passes that recompute the layers they differentiate (weight-gradient slabs per workgroup, statistics rows per wave)
        float *g = L.dU3;
        const int wv = (int)cmf_setconv_chain_waves(M, 1), nslab = wv / 4;
        CMF_TRY(cmf_maxpool_bwd_point_sel(P, C3, L.dx, C3, L.z3, b2 + 2 * C3, b2 + 3 * C3, b2, b2 + C3, g, L.partial, st));
        if (d->acc_bn[2]) CMF_TRY(cmf_colsum_finalize(tiles128(P), C3, L.partial, L.sums, d->dbeta[2], d->dgamma[2], st));
        else CMF_TRY(cmf_colsum_store(tiles128(P), 2 * C3, L.partial, L.sums, C3, d->dbeta[2], d->dgamma[2], st));
        CMF_TRY(cmf_setconv_chain_pass(4, M, d->N, d->S, L.idx, d->xyz, d->training ? L.dxyz : nullptr, d->y, d->ldy, d->wx, d->ldwx, L.bn[0], L.bn[1], L.bn[2], d->w[0], d->w[1],
                                       nullptr, 0, nullptr, L.argmax, L.partial, g, d->training ? L.sums : nullptr, nullptr, L.dU2, L.splitk, st));
        if (d->dw[1]) CMF_TRY(cmf_splitk_reduce(C3, C2, nslab, L.splitk, d->dw[1], C2, d->acc_w[1], (hipStream_t)st));
        if (d->acc_bn[1]) CMF_TRY(cmf_colsum_finalize(wv, C2, L.partial, L.sums, d->dbeta[1], d->dgamma[1], st));
        else CMF_TRY(cmf_colsum_store(wv, 2 * C2, L.partial, L.sums, C2, d->dbeta[1], d->dgamma[1], st));
        CMF_TRY(cmf_setconv_chain_pass(5, M, d->N, d->S, L.idx, d->xyz, d->training ? L.dxyz : nullptr, d->y, d->ldy, d->wx, d->ldwx, L.bn[0], L.bn[1], L.bn[2], d->w[0], d->w[1],
                                       nullptr, 0, nullptr, nullptr, L.partial, nullptr, d->training ? L.sums : nullptr, L.dU2, L.dU1, L.splitk, st));
        if (d->dw[0]) CMF_TRY(cmf_splitk_reduce(C2, O1, nslab, L.splitk, d->dw[0], O1, d->acc_w[0], (hipStream_t)st));
    } else
    if (fused && cmf_thin_bwd_supported(C3, C2) && M % 128 == 0 && C3 % 32 == 0 && C2 % 32 == 0 && d->dw[1]) {
        // narrow layers: the gradient of the pooled tensor is kept per POINT (g, in L.dx's neighbour L.dU3) and expanded
        // by the fused layer kernel on the fly -- the [M, C3] matrix is neither written nor read
        float *g = L.dU3;
        CMF_TRY(cmf_maxpool_bwd_point_sel(P, C3, L.dx, C3, L.zsel, b2 + 2 * C3, b2 + 3 * C3, b2, b2 + C3, g, L.partial, st));
        if (d->acc_bn[2]) CMF_TRY(cmf_colsum_finalize(tiles128(P), C3, L.partial, L.sums, d->dbeta[2], d->dgamma[2], st));
        else CMF_TRY(cmf_colsum_store(tiles128(P), 2 * C3, L.partial, L.sums, C3, d->dbeta[2], d->dgamma[2], st));
        const float *b1 = L.bn[1];
        CMF_TRY(cmf_thin_bwd_layer_pooled(P, d->S, C3, C2, g, L.argmax, L.z3, b2 + 2 * C3, b2, b2 + C3, d->training ? L.sums : nullptr,
                                          d->w[1], L.z2, b1 + 2 * C2, b1 + 3 * C2, b1, b1 + C2, L.dU2, L.partial, d->dw[1], d->acc_w[1],
                                          L.splitk, st));
    } else if (fused && wide && cmf_thin_bwd_wide_supported(C3, C2) && M % 128 == 0 && d->dw[1]) {
        // 64 <- 256 channels (second encoder): the same single pass, a workgroup per (row range, 128 input channels)
        float *g = L.dU3;
        CMF_TRY(cmf_maxpool_bwd_point_sel(P, C3, L.dx, C3, L.zsel, b2 + 2 * C3, b2 + 3 * C3, b2, b2 + C3, g, L.partial, st));
        if (d->acc_bn[2]) CMF_TRY(cmf_colsum_finalize(tiles128(P), C3, L.partial, L.sums, d->dbeta[2], d->dgamma[2], st));
        else CMF_TRY(cmf_colsum_store(tiles128(P), 2 * C3, L.partial, L.sums, C3, d->dbeta[2], d->dgamma[2], st));
        const float *b1 = L.bn[1];
        CMF_TRY(cmf_thin_bwd_wide_layer(M, C2, nullptr, C3, g, L.argmax, d->S, L.z3, C3, b2 + 2 * C3, b2, b2 + C3,
                                        d->training ? L.sums : nullptr, d->w[1], C2, L.z2, C2, b1 + 2 * C2, b1 + 3 * C2, b1, b1 + C2,
                                        L.dU2, C2, L.partial, d->dw[1], C2, d->acc_w[1], L.splitk, st));
    } else {
        CMF_TRY(cmf_maxpool_bwd(P, d->S, C3, L.dx, C3, L.z3, b2 + 2 * C3, b2 + 3 * C3, b2, b2 + C3, L.argmax, L.dU3, L.partial, st));
        CMF_TRY(bwd_layer(d, L, 2, M, C3, C2, L.dU3, L.z3, d->w[1], L.z2, 1, L.dU2, nullptr, d->dw[1], d->acc_w[1], st));
    }
    const bool chain = chain_train(d);
    if (chain) {}
    else if (train_gather(d)) CMF_TRY(bwd_layer1_gather(d, L, M, C2, O1, st));
    else
    CMF_TRY(bwd_layer(d, L, 1, M, C2, O1, L.dU2, L.z2, d->w[0], L.z1, 0, L.dU1, L.dxyz, d->dw[0], d->acc_w[0], st, L.dZ2));
    // first layer: sums {s1,s2,q0,q1,q2}; dgamma/dbeta; dW_xyz from sums; BN backward folded into the scatter
    const float *b0 = L.bn[0];
    const int rows0 = chain ? (int)cmf_setconv_chain_waves(M, 1) : tiles128(M);      // statistics rows of the pass that formed dU1
    if (d->acc_bn[0]) CMF_TRY(cmf_colsum(rows0, 5 * O1, L.partial, L.sums, O1, d->dbeta[0], d->dgamma[0], st));
    else CMF_TRY(cmf_colsum_store(rows0, 5 * O1, L.partial, L.sums, O1, d->dbeta[0], d->dgamma[0], st));
    if (d->dwx)
        CMF_TRY(cmf_setconv_dwx(O1, (float)(1.0 / (double)M), d->training, L.sums, L.fwd_sums, b0 + 2 * O1, b0, b0 + O1, d->dwx,
                                (int)d->lddwx, d->acc_wx, st));
    if (d->dy) {
        // z1 rows of one source point differ only by wx . dxyz: the BN-backward part of the scatter has a closed form,
        // so only dU1 is streamed (csrc/group_rows.hip)
        if (train_gather_sum(d))
            CMF_TRY(cmf_group_rows_grad_bn_cf_pieces(d->B, d->N, O1, d->N * d->S, d->S, sum_slots(d, L).pieces, d->y, d->ldy, d->wx, d->ldwx, d->xyz,
                                                     d->xyz, b0 + 2 * O1, b0, b0 + O1, d->training ? L.sums : nullptr, (float)(1.0 / (double)M), L.offsets, L.inv, d->dy,
                                                     d->lddy ? (int)d->lddy : O1, st));
        else
        CMF_TRY(cmf_group_rows_grad_bn_cf(d->B, d->N, O1, d->N * d->S, d->S, L.dU1, d->y, d->ldy, d->wx, d->ldwx, d->xyz, d->xyz,
                                          b0 + 2 * O1, b0, b0 + O1, d->training ? L.sums : nullptr, (float)(1.0 / (double)M),
                                          L.offsets, L.inv, d->dy, d->lddy ? (int)d->lddy : O1, st));
    }
    return 0;
}

extern "C" int cmf_setconv_backward(const cmf_setconv_desc *d, void *st) { return setconv_backward_part(d, st, 0); }

// ---------------------------------------------------------------------------------------------------------------
// The independent scales of a MultiScaleEncoder (radarflow_util.py:101-118) in ONE call: descs[i] is issued on
// streams[i] from its own host thread.  At N = 256 a scale's ~20 (forward) / ~45 (backward) kernels run about as long
// as they take to enqueue, so one host thread cannot keep four streams fed.  Stream ordering against the
// caller's stream is the caller's business (events before / after the call); nothing is synchronised here.
// ---------------------------------------------------------------------------------------------------------------
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace {
// Persistent host threads for the chains of a multi-scale call: a chain is ~20 / ~45 launches, i.e. a few hundred
// microseconds of enqueueing -- creating and joining a std::thread per chain and call (up to 7 spawns, ~6 calls per step)
// costs a comparable amount.  Workers are created on first use, sleep on a condition variable between calls and are
// never destroyed (the pool is leaked on purpose: joining threads during static destruction races with the HIP
// runtime's own teardown).
struct ChainWorker {
    std::mutex m;
    std::condition_variable cv;
    std::function<void()> job;
    bool pending = false, done = true;
    void loop()
    {
        for (;;) {
            std::function<void()> j;
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return pending; });
                j = std::move(job);
                pending = false;
            }
            j();
            {
                std::lock_guard<std::mutex> lk(m);
                done = true;
            }
            cv.notify_all();
        }
    }
    void submit(std::function<void()> j)
    {
        {
            std::lock_guard<std::mutex> lk(m);
            job = std::move(j);
            pending = true;
            done = false;
        }
        cv.notify_all();
    }
    void wait()
    {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return done; });
    }
};
std::mutex g_pool_mutex;                                 // one multi-scale call at a time uses the pool
std::vector<ChainWorker *> *g_pool = nullptr;

ChainWorker *chain_worker(size_t i)
{
    if (!g_pool) g_pool = new std::vector<ChainWorker *>();
    while (g_pool->size() <= i) {
        ChainWorker *w = new ChainWorker();
        std::thread(&ChainWorker::loop, w).detach();
        g_pool->push_back(w);
    }
    return (*g_pool)[i];
}
}  // namespace

static int setconv_multi(int n, const cmf_setconv_desc *descs, void *const *streams, bool backward, int part = 0)
{
    CMF_CHECK_ARG(n >= 0 && n <= 16 && (n == 0 || (descs && streams)));
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return (int)hipGetLastError();
    int err[16] = {0};
    auto run = [&](int i) {
        if (i > 0 && hipSetDevice(dev) != hipSuccess) { err[i] = (int)hipGetLastError(); return; }
        err[i] = backward ? setconv_backward_part(&descs[i], streams[i], part) : setconv_forward_part(&descs[i], streams[i], part);
    };
    bool one_stream = true;                                 // every chain on the same stream (serialised diagnostic runs):
    for (int i = 1; i < n; ++i) one_stream = one_stream && streams[i] == streams[0];    // nothing to feed in parallel
    if (one_stream) {
        for (int i = 0; i < n; ++i) run(i);
    } else {
        std::lock_guard<std::mutex> pool_lock(g_pool_mutex);
        for (int i = 1; i < n; ++i) chain_worker((size_t)i - 1)->submit([&run, i] { run(i); });
        if (n > 0) run(0);                                  // the calling thread takes the first scale
        for (int i = 1; i < n; ++i) chain_worker((size_t)i - 1)->wait();
    }
    for (int i = 0; i < n; ++i) if (err[i]) return err[i];
    return 0;
}

extern "C" int cmf_setconv_forward_multi(int n, const cmf_setconv_desc *descs, void *const *streams)
{
    return setconv_multi(n, descs, streams, false);
}

extern "C" int cmf_setconv_backward_multi(int n, const cmf_setconv_desc *descs, void *const *streams)
{
    return setconv_multi(n, descs, streams, true);
}

// ---------------------------------------------------------------------------------------------------------------
// The slot-level bodies of up to CMF_MAX_BATCH NARROW blocks (the first encoder: 2 clouds x 4 scales, 32 / 32 / 64 channels) in lock
// step: one batched launch per stage for all blocks instead of one kernel per block and stage on the stream pool.  At these widths
// every kernel of a body is an HBM stream of 15-70 us -- eight chains of eight launches each kept four hardware queues busy with
// launch gaps, and co-running bought nothing (the bodies' wall time equalled the sum of their kernels' isolated durations).  Same
// kernels' device code, same arithmetic, same results as the per-block calls.  Train-mode BatchNorm only (eval-mode BN takes the
// register chain); CMF_BODY_BATCH=0: the per-block chains (A/B).
// ---------------------------------------------------------------------------------------------------------------
static bool body_batchable(int n, const cmf_setconv_desc *descs)
{
    static const bool on = !(getenv("CMF_BODY_BATCH") && getenv("CMF_BODY_BATCH")[0] == '0');
    if (!on || n < 2 || n > CMF_MAX_BATCH) return false;
    for (int i = 0; i < n; ++i) {
        const cmf_setconv_desc &d = descs[i];
        const long long M = (long long)d.B * d.N * d.S;
        if (!d.training || d.inference || !d.idx_ready || chain_train(&d) || train_gather(&d)) return false;
        if (d.O1 != descs[0].O1 || d.C[0] != descs[0].C[0] || d.C[1] != descs[0].C[1]) return false;
        if (d.O1 > 64 || d.O1 % 32 || d.C[0] > 64 || d.C[0] % 32 || d.C[1] > 64 || d.C[1] % 32) return false;
        if (M % 128 || M >= (1ll << 31) || d.ldy % 4 || ((uintptr_t)d.y & 15)) return false;
    }
    return true;
}

// inference of the narrow blocks (eval-mode BN, no backward call to follow): their register-chain kernels as ONE launch
static bool infer_batchable(int n, const cmf_setconv_desc *descs)
{
    static const bool on = !(getenv("CMF_BODY_BATCH") && getenv("CMF_BODY_BATCH")[0] == '0');
    if (!on || n < 2 || n > CMF_MAX_BATCH) return false;
    for (int i = 0; i < n; ++i) {
        const cmf_setconv_desc &d = descs[i];
        const long long M = (long long)d.B * d.N * d.S;
        if (d.training || !d.inference || !d.idx_ready || !cmf_setconv_chain_supported(d.N, d.S, d.O1, d.C[0], d.C[1], M) || d.ldy % 4 ||
            ((uintptr_t)d.y & 15) || M >= (1ll << 31)) return false;
        for (int l = 0; l < 6; ++l) if (!d.rmean[l] || !d.rvar[l]) return false;
    }
    return true;
}

static int setconv_infer_bodies_batch(int n, const cmf_setconv_desc *descs, hipStream_t st)
{
    CmfBatch<FoldAll> fb;
    CmfChainInferArgs q[CMF_MAX_BATCH];
    for (int i = 0; i < n; ++i) {
        const cmf_setconv_desc &d = descs[i];
        CMF_CHECK_ARG(d.xyz && d.y && d.wx && d.saved && d.scratch && d.out);
        const Layout L = make_layout(&d, d.saved, d.scratch, false);
        FoldAll &f = fb.a[i];
        for (int l = 0; l < 6; ++l) {
            f.C[l] = chan(&d, l); f.gamma[l] = d.gamma[l]; f.beta[l] = d.beta[l]; f.rmean[l] = d.rmean[l]; f.rvar[l] = d.rvar[l];
            f.eps[l] = d.eps[l]; f.out[l] = L.bn[l];
        }
        q[i] = CmfChainInferArgs{(long long)d.B * d.N * d.S, d.N, d.S, L.idx, d.xyz, d.y, d.ldy, d.wx, d.ldwx, L.bn[0], L.bn[1], L.bn[2], d.w[0], d.w[1],
                                 L.x, d.C[1]};
    }
    hipLaunchKernelGGL(bn_fold_eval_batch_kernel, dim3(6, n), dim3(256), 0, st, fb);
    CMF_TRY(cmf_launch_status());
    return cmf_setconv_chain_infer_batch(n, q, st);
}

// largest block first: blockIdx.y = block and the dispatcher walks y slowest, so the blocks with the most rows start first and the small
// ones fill the tail of the launch
static void largest_first(int n, const cmf_setconv_desc *in, cmf_setconv_desc *out)
{
    int ord[CMF_MAX_BATCH];
    for (int i = 0; i < n; ++i) ord[i] = i;
    std::stable_sort(ord, ord + n, [&](int a, int b) { return (long long)in[a].B * in[a].N * in[a].S > (long long)in[b].B * in[b].N * in[b].S; });
    for (int i = 0; i < n; ++i) out[i] = in[ord[i]];
}

static int setconv_forward_bodies_batch(int n, const cmf_setconv_desc *descs_in, hipStream_t st)
{
    cmf_setconv_desc descs[CMF_MAX_BATCH];
    largest_first(n, descs_in, descs);
    Layout L[CMF_MAX_BATCH];
    for (int i = 0; i < n; ++i) {
        CMF_CHECK_ARG(descs[i].xyz && descs[i].y && descs[i].wx && descs[i].saved && descs[i].scratch && descs[i].out);
        L[i] = make_layout(&descs[i], descs[i].saved, descs[i].scratch, false);
    }
    const int O1 = descs[0].O1, C2 = descs[0].C[0], C3 = descs[0].C[1];
    {   // z1 = y[idx] + Wx . dxyz with its statistics; the z * d_k sums for dW_xyz
        CmfGroupAffineArgs ga[CMF_MAX_BATCH];
        CmfColsumArgs cs[CMF_MAX_BATCH];
        for (int i = 0; i < n; ++i) {
            const cmf_setconv_desc &d = descs[i];
            const long long M = (long long)d.B * d.N * d.S;
            ga[i] = CmfGroupAffineArgs{d.B, d.N, d.N, d.S, O1, d.y, (int)d.ldy, d.xyz, d.xyz, d.wx, (int)d.ldwx, L[i].idx, L[i].z1, L[i].dxyz,
                                       L[i].partial, L[i].partial_x};
            cs[i] = CmfColsumArgs{tiles128(M), 3 * O1 + 4, L[i].partial_x, L[i].fwd_sums, 0, nullptr, nullptr, 0};
        }
        CMF_TRY(cmf_group_affine_batch(n, ga, st));
        CMF_TRY(cmf_colsum_batch(n, cs, st));
    }
    for (int layer = 0; layer < 3; ++layer) {            // fold BN layer `layer`, then (layers 0, 1) the next 1x1 conv / (layer 2) the max over the ball
        CmfBnFinArgs f[CMF_MAX_BATCH];
        GemmArgs g[CMF_MAX_BATCH];
        CmfPoolArgs pl[CMF_MAX_BATCH];
        const int C = layer == 0 ? O1 : (layer == 1 ? C2 : C3), cout = layer == 0 ? C2 : C3;
        for (int i = 0; i < n; ++i) {
            const cmf_setconv_desc &d = descs[i];
            const long long P = (long long)d.B * d.N, M = P * d.S;
            float *b = L[i].bn[layer];
            f[i] = CmfBnFinArgs{tiles128(M), C, (double)M, L[i].partial, d.gamma[layer], d.beta[layer], d.eps[layer], d.momentum[layer],
                                d.rmean[layer], d.rvar[layer], b, b + C, b + 2 * C, b + 3 * C, d.nbt[layer]};
            if (layer < 2) {
                GemmArgs &q = g[i];
                q = GemmArgs{};
                q.M = (int)M; q.N = cout; q.K = C; q.A = layer == 0 ? L[i].z1 : L[i].z2; q.lda = C; q.B = d.w[layer]; q.ldb = C;
                q.C = layer == 0 ? L[i].z2 : L[i].z3; q.ldc = cout; q.pro_a = b + 2 * C; q.pro_c = b + 3 * C; q.stats = L[i].partial; q.split_k = 1;
            } else
                pl[i] = CmfPoolArgs{P, d.S, C3, L[i].z3, b + 2 * C3, b + 3 * C3, L[i].x, C3, L[i].argmax, 0, L[i].zsel};
        }
        CMF_TRY(cmf_bn_finalize_batch(n, f, st));
        if (layer < 2) CMF_TRY(cmf_thin_fwd_batch(n, g, st));
        else CMF_TRY(cmf_bn_relu_maxpool_batch(n, pl, st));
    }
    return 0;
}

// backward of the same bodies (behind cmf_setconv_tail_backward, which has produced Layout::dx): the kernels of setconv_backward_part's
// narrow path, one batched launch per stage
static bool body_batchable_bwd(int n, const cmf_setconv_desc *descs)
{
    if (!body_batchable(n, descs)) return false;
    static const bool fused = !(getenv("CMF_THIN_FUSED") && getenv("CMF_THIN_FUSED")[0] == '0');
    for (int i = 0; i < n; ++i) {
        const cmf_setconv_desc &d = descs[i];
        if (!fused || !d.dw[0] || !d.dw[1] || !d.dout || !cmf_thin_bwd_supported(d.C[1], d.C[0]) || !cmf_thin_bwd_supported(d.C[0], d.O1)) return false;
        if (d.C[1] != 64 || d.C[0] != 32 || d.O1 != 32) return false;                 // the batched instantiations of the fused layer
        if ((d.dy != nullptr) != (descs[0].dy != nullptr) || (d.dwx != nullptr) != (descs[0].dwx != nullptr)) return false;
        if (d.B != descs[0].B || d.N != descs[0].N || d.N > 256 || d.S > 64) return false;
    }
    return true;
}

static int splitk_batch_or_each(int n, const CmfSplitkArgs *sk, hipStream_t st)
{
    if (cmf_splitk_reduce_batch(n, sk, st) == 0) return 0;          // (refused without a launch when the problems want different kernels)
    (void)hipGetLastError();
    for (int i = 0; i < n; ++i) CMF_TRY(cmf_splitk_reduce(sk[i].M, sk[i].N, sk[i].split_k, sk[i].workspace, sk[i].C, sk[i].ldc, sk[i].accumulate, st));
    return 0;
}

static int setconv_backward_bodies_batch(int n, const cmf_setconv_desc *descs_in, hipStream_t st)
{
    cmf_setconv_desc descs[CMF_MAX_BATCH];
    largest_first(n, descs_in, descs);
    Layout L[CMF_MAX_BATCH];
    for (int i = 0; i < n; ++i) {
        CMF_CHECK_ARG(descs[i].xyz && descs[i].saved && descs[i].scratch && descs[i].dout);
        L[i] = make_layout(&descs[i], descs[i].saved, descs[i].scratch, true);
    }
    const int O1 = descs[0].O1, C2 = descs[0].C[0], C3 = descs[0].C[1];
    const bool want_dy = descs[0].dy != nullptr, want_dwx = descs[0].dwx != nullptr;
    if (want_dy) {
        CmfInverseArgs iv[CMF_MAX_BATCH];
        for (int i = 0; i < n; ++i) iv[i] = CmfInverseArgs{descs[i].N, descs[i].N, descs[i].S, L[i].idx, L[i].offsets, L[i].inv};
        CMF_TRY(cmf_build_inverse_ps_batch(n, descs[0].B, iv, st));
    }
    CmfColsumArgs cs[CMF_MAX_BATCH];
    CmfThinBwdCall tb[CMF_MAX_BATCH];
    CmfSplitkArgs sk[CMF_MAX_BATCH];
    {   // max over the ball -> the pooled gradient per point with layer 3's BN-backward sums; layer 3 (64 <- 32) from the pooled gradient
        CmfPoolBwdArgs pb[CMF_MAX_BATCH];
        for (int i = 0; i < n; ++i) {
            const cmf_setconv_desc &d = descs[i];
            const long long P = (long long)d.B * d.N;
            const float *b2 = L[i].bn[2], *b1 = L[i].bn[1];
            float *g = L[i].dU3;
            pb[i] = CmfPoolBwdArgs{P, d.S, C3, L[i].dx, C3, L[i].zsel, b2 + 2 * C3, b2 + 3 * C3, b2, b2 + C3, L[i].argmax, g, L[i].partial, 1};
            cs[i] = CmfColsumArgs{tiles128(P), 2 * C3, L[i].partial, L[i].sums, C3, d.dbeta[2], d.dgamma[2], d.acc_bn[2] ? 0 : 1};
            CmfThinBwdCall &q = tb[i];
            q = CmfThinBwdCall{};
            q.rows = P * d.S; q.cout = C3; q.cin = C2; q.dU = nullptr; q.lddu = C3; q.z = L[i].z3; q.ldz = C3;
            q.a = b2 + 2 * C3; q.mean = b2; q.invstd = b2 + C3; q.sums = L[i].sums;
            q.w = d.w[1]; q.ldw = C2; q.x = L[i].z2; q.ldx = C2; q.in_mode = 1;
            q.a_in = b1 + 2 * C2; q.c_in = b1 + 3 * C2; q.mean_in = b1; q.invstd_in = b1 + C2;
            q.dx = L[i].dU2; q.lddx = C2; q.stats = L[i].partial; q.dw = d.dw[1]; q.lddw = C2; q.accumulate = d.acc_w[1]; q.slabs = L[i].splitk;
            q.pool_g = g; q.pool_am = L[i].argmax; q.pool_S = d.S;
        }
        CMF_TRY(cmf_maxpool_bwd_point_batch(n, pb, st));
        CMF_TRY(cmf_colsum_batch(n, cs, st));
        CMF_TRY(cmf_thin_bwd_layer_batch(n, tb, st));
        for (int i = 0; i < n; ++i) sk[i] = CmfSplitkArgs{C3, C2, tb[i].nslab, tb[i].slabs, tb[i].dw, tb[i].lddw, tb[i].accumulate};
        CMF_TRY(splitk_batch_or_each(n, sk, st));
    }
    {   // layer 2 (32 <- 32) with the dxyz sums of the first layer's coordinate weights
        for (int i = 0; i < n; ++i) {
            const cmf_setconv_desc &d = descs[i];
            const long long M = (long long)d.B * d.N * d.S;
            const float *b1 = L[i].bn[1], *b0 = L[i].bn[0];
            cs[i] = CmfColsumArgs{tiles128(M), 2 * C2, L[i].partial, L[i].sums, C2, d.dbeta[1], d.dgamma[1], d.acc_bn[1] ? 0 : 1};
            CmfThinBwdCall &q = tb[i];
            q = CmfThinBwdCall{};
            q.rows = M; q.cout = C2; q.cin = O1; q.dU = L[i].dU2; q.lddu = C2; q.z = L[i].z2; q.ldz = C2;
            q.a = b1 + 2 * C2; q.mean = b1; q.invstd = b1 + C2; q.sums = L[i].sums;
            q.w = d.w[0]; q.ldw = O1; q.x = L[i].z1; q.ldx = O1; q.in_mode = 1;
            q.a_in = b0 + 2 * O1; q.c_in = b0 + 3 * O1; q.mean_in = b0; q.invstd_in = b0 + O1; q.dxyz = L[i].dxyz;
            q.dx = L[i].dU1; q.lddx = O1; q.stats = L[i].partial; q.dw = d.dw[0]; q.lddw = O1; q.accumulate = d.acc_w[0]; q.slabs = L[i].splitk;
        }
        CMF_TRY(cmf_colsum_batch(n, cs, st));
        CMF_TRY(cmf_thin_bwd_layer_batch(n, tb, st));
        for (int i = 0; i < n; ++i) sk[i] = CmfSplitkArgs{C2, O1, tb[i].nslab, tb[i].slabs, tb[i].dw, tb[i].lddw, tb[i].accumulate};
        CMF_TRY(splitk_batch_or_each(n, sk, st));
    }
    {   // first layer: sums {s1, s2, q0, q1, q2}; dgamma / dbeta; dW_xyz; BN backward folded into the scatter
        CmfDwxArgs dx[CMF_MAX_BATCH];
        CmfScatterArgs sc[CMF_MAX_BATCH];
        for (int i = 0; i < n; ++i) {
            const cmf_setconv_desc &d = descs[i];
            const long long M = (long long)d.B * d.N * d.S;
            const float *b0 = L[i].bn[0];
            cs[i] = CmfColsumArgs{tiles128(M), 5 * O1, L[i].partial, L[i].sums, O1, d.dbeta[0], d.dgamma[0], d.acc_bn[0] ? 0 : 1};
            dx[i] = CmfDwxArgs{O1, (float)(1.0 / (double)M), d.training, L[i].sums, L[i].fwd_sums, b0 + 2 * O1, b0, b0 + O1, d.dwx, (int)d.lddwx, d.acc_wx};
            sc[i] = CmfScatterArgs{d.N, d.N * d.S, d.S, L[i].dU1, d.y, d.ldy, d.wx, d.ldwx, d.xyz, d.xyz, b0 + 2 * O1, b0, b0 + O1, L[i].sums,
                                   (float)(1.0 / (double)M), L[i].offsets, L[i].inv, d.dy, d.lddy ? (int)d.lddy : O1};
        }
        CMF_TRY(cmf_colsum_batch(n, cs, st));
        if (want_dwx) CMF_TRY(cmf_setconv_dwx_batch(n, dx, st));
        if (want_dy) CMF_TRY(cmf_group_rows_grad_bn_cf_batch(n, descs[0].B, O1, sc, st));
    }
    return 0;
}

// 1 when cmf_setconv_backward_bodies_multi will run these blocks' bodies as batched launches on streams[0] alone, else 0
extern "C" int cmf_setconv_backward_bodies_batched(int n, const cmf_setconv_desc *descs)
{
    return (n >= 0 && n <= 16 && descs && body_batchable_bwd(n, descs)) ? 1 : 0;
}

// The same with the per-point tails taken out: forward stops behind the max over the ball, backward starts there.  The
// caller runs cmf_setconv_tail_forward behind the forward heads (after joining the streams) and cmf_setconv_tail_backward in
// front of the backward bodies (before forking): the tails of all blocks as batched launches on one stream.
// 1 when cmf_setconv_forward_heads_multi will run these blocks' bodies as batched launches on streams[0] alone (the caller may then pass
// its own stream for every entry and skip the fork / join around the call), else 0
extern "C" int cmf_setconv_forward_bodies_batched(int n, const cmf_setconv_desc *descs)
{
    return (n >= 0 && n <= 16 && descs && (body_batchable(n, descs) || infer_batchable(n, descs))) ? 1 : 0;
}

// The batched forms run every block on streams[0].  A caller that prepared block i's inputs on streams[i] (the contract of the
// per-chain form: INTEGRATION.md "fork / join") is still ordered: streams[0] waits for what is queued on every other distinct stream
// before the batched launches, and those streams wait for streams[0] behind them.  With all entries equal (the Python host, which asks
// cmf_setconv_*_bodies_batched first) nothing is recorded.
static int batched_order(int n, void *const *streams, bool before)
{
    hipStream_t s0 = (hipStream_t)streams[0];
    hipEvent_t back = nullptr;
    for (int i = 1; i < n; ++i) {
        hipStream_t si = (hipStream_t)streams[i];
        if (si == s0) continue;
        bool seen = false;
        for (int j = 1; j < i; ++j) seen = seen || streams[j] == streams[i];
        if (seen) continue;
        if (before) {
            hipEvent_t ev;
            if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return (int)hipGetLastError();
            if (hipEventRecord(ev, si) != hipSuccess || hipStreamWaitEvent(s0, ev, 0) != hipSuccess) { (void)hipEventDestroy(ev); return (int)hipGetLastError(); }
            (void)hipEventDestroy(ev);                  // (released by the runtime once the recorded work has completed)
        } else {
            if (!back) {
                if (hipEventCreateWithFlags(&back, hipEventDisableTiming) != hipSuccess) return (int)hipGetLastError();
                if (hipEventRecord(back, s0) != hipSuccess) { (void)hipEventDestroy(back); return (int)hipGetLastError(); }
            }
            if (hipStreamWaitEvent(si, back, 0) != hipSuccess) { (void)hipEventDestroy(back); return (int)hipGetLastError(); }
        }
    }
    if (back) (void)hipEventDestroy(back);
    return 0;
}

extern "C" int cmf_setconv_forward_heads_multi(int n, const cmf_setconv_desc *descs, void *const *streams)
{
    CMF_CHECK_ARG(n >= 0 && n <= 16 && (n == 0 || (descs && streams)));
    const bool body = body_batchable(n, descs);
    if (body || infer_batchable(n, descs)) {
        CMF_TRY(batched_order(n, streams, true));
        CMF_TRY(body ? setconv_forward_bodies_batch(n, descs, (hipStream_t)streams[0]) : setconv_infer_bodies_batch(n, descs, (hipStream_t)streams[0]));
        return batched_order(n, streams, false);
    }
    return setconv_multi(n, descs, streams, false, 1);
}

extern "C" int cmf_setconv_backward_bodies_multi(int n, const cmf_setconv_desc *descs, void *const *streams)
{
    CMF_CHECK_ARG(n >= 0 && n <= 16 && (n == 0 || (descs && streams)));
    if (body_batchable_bwd(n, descs)) {
        CMF_TRY(batched_order(n, streams, true));
        CMF_TRY(setconv_backward_bodies_batch(n, descs, (hipStream_t)streams[0]));
        return batched_order(n, streams, false);
    }
    return setconv_multi(n, descs, streams, true, 2);
}

// ---------------------------------------------------------------------------------------------------------------
// [conv + BN + ReLU] x L on a materialised input (the heads' stacks): cmf_mlp_forward / _backward.  The kernel sequence
// of fused_blocks.MLPChainFn, issued from here.
// ---------------------------------------------------------------------------------------------------------------
namespace {
struct MlpLayout {
    float *z[4], *bn[4];
    float *partial, *sums, *dU[4], *splitk;
    size_t saved_floats, scratch_floats;
};
MlpLayout mlp_layout(const cmf_mlp_desc *d, float *saved, float *scratch, bool backward)
{
    MlpLayout L;
    Bump s(saved);
    for (int l = 0; l < d->L; ++l) L.z[l] = s.take((size_t)d->M * d->C[l + 1]);
    for (int l = 0; l < d->L; ++l) L.bn[l] = s.take(4 * (size_t)d->C[l + 1]);
    L.saved_floats = s.off;
    Bump t(scratch);
    int cmax = 0;
    for (int l = 0; l <= d->L; ++l) cmax = std::max(cmax, d->C[l]);
    L.partial = t.take((size_t)tiles128(d->M) * 2 * cmax);
    L.sums = t.take(2 * (size_t)cmax);
    size_t sk = 0;
    if (backward) {
        for (int l = 0; l < d->L; ++l) L.dU[l] = t.take((size_t)d->M * d->C[l + 1]);
        for (int l = 0; l < d->L; ++l) sk = std::max(sk, (size_t)dw_split(d->M, d->C[l + 1], d->C[l]) * d->C[l + 1] * d->C[l]);
        L.splitk = t.take(sk);
    } else {
        for (int l = 0; l < 4; ++l) L.dU[l] = nullptr;
        L.splitk = nullptr;
    }
    L.scratch_floats = t.off;
    return L;
}
bool mlp_ok(const cmf_mlp_desc *d)
{
    if (!d || d->M <= 0 || d->M >= (1ll << 31) || d->L < 1 || d->L > 4) return false;
    for (int l = 0; l <= d->L; ++l) if (d->C[l] <= 0 || d->C[l] % 4) return false;
    return true;
}
}  // namespace

extern "C" int cmf_mlp_sizes(const cmf_mlp_desc *d, long long *saved_floats, long long *scratch_fwd, long long *scratch_bwd)
{
    CMF_CHECK_ARG(mlp_ok(d));
    const MlpLayout f = mlp_layout(d, nullptr, nullptr, false), b = mlp_layout(d, nullptr, nullptr, true);
    if (saved_floats) *saved_floats = (long long)f.saved_floats;
    if (scratch_fwd) *scratch_fwd = (long long)f.scratch_floats;
    if (scratch_bwd) *scratch_bwd = (long long)b.scratch_floats;
    return 0;
}

extern "C" int cmf_mlp_forward(const cmf_mlp_desc *d, void *st)
{
    CMF_CHECK_ARG(mlp_ok(d) && d->x && d->saved && d->scratch && d->out && d->ldx % 4 == 0 && d->ldo % 4 == 0);
    const MlpLayout L = mlp_layout(d, d->saved, d->scratch, false);
    for (int l = 0; l < d->L; ++l) {
        const int cin = d->C[l], cout = d->C[l + 1];
        const float *zin = l == 0 ? d->x : L.z[l - 1];
        const long long ldin = l == 0 ? d->ldx : cin;
        const float *pa = l > 0 ? L.bn[l - 1] + 2 * cin : nullptr, *pc = l > 0 ? L.bn[l - 1] + 3 * cin : nullptr;
        CMF_TRY(cmf_gemm((int)d->M, cout, cin, 0, 1, zin, ldin, d->w[l], cin, L.z[l], cout, pa, pc, nullptr, nullptr, nullptr, 0,
                         d->training ? L.partial : nullptr, 0, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, 1, nullptr, 0, st));
        float *b = L.bn[l];
        if (d->training)
            CMF_TRY(cmf_bn_finalize(tiles128(d->M), cout, (double)d->M, L.partial, d->gamma[l], d->beta[l], d->eps[l], d->momentum[l],
                                    d->rmean[l], d->rvar[l], b, b + cout, b + 2 * cout, b + 3 * cout, d->nbt[l], st));
        else
            CMF_TRY(cmf_bn_finalize(0, cout, 1.0, nullptr, d->gamma[l], d->beta[l], d->eps[l], 0.f, d->rmean[l], d->rvar[l], b, b + cout,
                                    b + 2 * cout, b + 3 * cout, nullptr, st));
    }
    const int cl = d->C[d->L];
    return cmf_affine_relu(d->M, cl, L.z[d->L - 1], cl, L.bn[d->L - 1] + 2 * cl, L.bn[d->L - 1] + 3 * cl, d->out, d->ldo, st);
}

extern "C" int cmf_mlp_backward(const cmf_mlp_desc *d, void *st)
{
    CMF_CHECK_ARG(mlp_ok(d) && d->x && d->saved && d->scratch && d->dout && d->lddout % 4 == 0 && (!d->dx || d->lddx % 4 == 0));
    const MlpLayout L = mlp_layout(d, d->saved, d->scratch, true);
    const long long M = d->M;
    {
        const int c = d->C[d->L];
        const float *b = L.bn[d->L - 1];
        CMF_TRY(cmf_act_bwd_stats(M, c, d->dout, d->lddout, L.z[d->L - 1], c, b + 2 * c, b + 3 * c, b, b + c, L.dU[d->L - 1], L.partial, st));
    }
    for (int l = d->L - 1; l >= 0; --l) {
        const int cin = d->C[l], cout = d->C[l + 1];
        const float *b = L.bn[l];
        CMF_CHECK_ARG(d->dw[l] && d->dgamma[l] && d->dbeta[l]);
        if (d->acc_bn[l]) CMF_TRY(cmf_colsum_finalize(tiles128(M), cout, L.partial, L.sums, d->dbeta[l], d->dgamma[l], st));
        else CMF_TRY(cmf_colsum_store(tiles128(M), 2 * cout, L.partial, L.sums, cout, d->dbeta[l], d->dgamma[l], st));
        CMF_TRY(cmf_bn_bwd_apply(M, cout, L.dU[l], L.z[l], cout, b + 2 * cout, b, b + cout, d->training ? L.sums : nullptr, st));
        // weight gradient: dZ^T act(x_in), deterministic split-K
        const float *xin = l == 0 ? d->x : L.z[l - 1];
        const long long ldin = l == 0 ? d->ldx : cin;
        const float *bi = l > 0 ? L.bn[l - 1] : nullptr;
        const int split = dw_split(M, cout, cin);
        CMF_TRY(cmf_gemm(cout, cin, (int)M, 1, 0, L.dU[l], cout, xin, ldin, d->dw[l], cin, nullptr, nullptr, bi ? bi + 2 * cin : nullptr,
                         bi ? bi + 3 * cin : nullptr, nullptr, 0, nullptr, 0, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, split,
                         split > 1 ? L.splitk : nullptr, d->acc_w[l], st));
        if (l > 0)                      // gradient w.r.t. the pre-activation of the layer below, masked, with its BN-backward sums
            CMF_TRY(cmf_gemm((int)M, cin, cout, 0, 0, L.dU[l], cout, d->w[l], cin, L.dU[l - 1], cin, nullptr, nullptr, nullptr, nullptr, nullptr, 0,
                             L.partial, 1, L.z[l - 1], cin, bi + 2 * cin, bi + 3 * cin, bi, bi + cin, nullptr, 1, nullptr, 0, st));
        else if (d->dx)
            CMF_TRY(cmf_gemm((int)M, cin, cout, 0, 0, L.dU[0], cout, d->w[0], cin, d->dx, d->lddx, nullptr, nullptr, nullptr, nullptr, nullptr, 0,
                             nullptr, 0, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, 1, nullptr, 0, st));
    }
    return 0;
}

// Float offsets of the per-layer BatchNorm blocks (mean | invstd | a | c, 4*C_l floats each) inside `saved`.
extern "C" int cmf_setconv_bn_offsets(const cmf_setconv_desc *d, long long *offsets6)
{
    CMF_CHECK_ARG(d && offsets6);
    // lay the arena out over a fake non-null base (Bump hands out NULL for a NULL base) and subtract it again
    float *const fake = reinterpret_cast<float *>(uintptr_t(1) << 40);
    const Layout L = make_layout(d, fake, nullptr, false);
    for (int l = 0; l < 6; ++l) offsets6[l] = (long long)(L.bn[l] - fake);
    return 0;
}

// Deferred running-statistics update for encoder calls that ran concurrently (two clouds through the same weight-shared
// encoder, radarflow_util.py:111-118 called twice per step, cmflow.py:72-73): the block calls were issued with
// rmean == NULL (batch statistics only) and the nn.BatchNorm2d momentum update is applied here once per call, in call
// order, from the saved batch mean / invstd of each call.  One workgroup per table entry (= one BN layer of one scale).
__global__ __launch_bounds__(64) void bn_running_update_kernel(const cmf_bn_update_entry *__restrict__ table, int n_calls,
                                                              const float *__restrict__ saved0, const float *__restrict__ saved1)
{
    const cmf_bn_update_entry e = table[blockIdx.x];
    const double mo = e.momentum, cnt = e.count;
    for (int c = threadIdx.x; c < e.C; c += 64) {
        double rm = e.rmean[c], rv = e.rvar[c];
        for (int k = 0; k < n_calls; ++k) {
            const float *s = (k == 0 ? saved0 : saved1) + e.offset;
            const double mean = s[c], invstd = s[e.C + c];
            double var = 1.0 / (invstd * invstd) - (double)e.eps;
            if (var < 0.0) var = 0.0;
            const double unbiased = cnt > 1.0 ? var * cnt / (cnt - 1.0) : var;
            rm = (1.0 - mo) * rm + mo * mean;
            rv = (1.0 - mo) * rv + mo * unbiased;
        }
        e.rmean[c] = (float)rm; e.rvar[c] = (float)rv;
    }
    if (threadIdx.x == 0 && e.nbt) *e.nbt += n_calls;
}

extern "C" int cmf_bn_running_update(int n_entries, const cmf_bn_update_entry *table, int n_calls, const float *saved0,
                                     const float *saved1, void *stream)
{
    CMF_CHECK_ARG(n_entries >= 0 && n_calls >= 1 && n_calls <= 2 && (n_entries == 0 || (table && saved0 && (n_calls == 1 || saved1))));
    if (n_entries == 0) return 0;
    hipLaunchKernelGGL(bn_running_update_kernel, dim3(n_entries), dim3(64), 0, (hipStream_t)stream, table, n_calls, saved0, saved1);
    return cmf_launch_status();
}
