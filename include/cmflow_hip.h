/*
 * cmflow_hip.h -- C ABI of libcmflow_hip.so, the MI355X (gfx950) drop-in for the native
 * layer of the CMFlow hot path.
 *
 * Conventions (all entry points):
 *   - plain pointers to DEVICE memory + sizes; no torch / ATen types cross this boundary.
 *   - `stream` is a hipStream_t passed as void* (NULL = the legacy default stream).  Work is
 *     enqueued asynchronously, nothing is synchronised, nothing is retained after return --
 *     same threading model as the reference wrappers, which enqueue on
 *     at::cuda::getCurrentCUDAStream() (lib/src/ball_query.cpp:22).
 *   - the caller allocates every buffer, outputs included (lib/pointnet2_utils.py:200-202,246-248).  Two entry points
 *     need working memory their reference signature has no argument for and take it from a library-owned scratch, one
 *     buffer per (device, stream), allocated with hipMalloc on first use, grown on demand and kept for the life of the
 *     process: cmf_ball_query (spilled hit lists when nsample > 32 and more than 768 workgroups: 32 KB per workgroup,
 *     64 MB at b = 32, m = 4096; the cell grid of clouds with 4096-8192 points: 16 * b * n + 16 KB * b bytes, used by two
 *     launches) and cmf_group_points_grad (inverse index or scatter plan: <= 4 * b * (n + 1 + npoints * nsample) bytes);
 *     cmf_query_and_group keeps the indices there when the caller passes idx == NULL.  The buffer of a (device, stream) is
 *     handed out under a lock that is held until the last launch using it has been enqueued, so calls from several host
 *     threads onto one stream are safe; a buffer that must grow is retired (kept allocated), never freed while work may
 *     be queued on it.  Nothing else is retained between calls.
 *   - return value: hipError_t as int (0 = hipSuccess).  The reference launchers print and
 *     exit(-1) on a launch failure (lib/src/ball_query_gpu.cu:62-66); this library reports the
 *     error to the caller instead.  Invalid arguments return hipErrorInvalidValue (1).
 *   - all floating point is fp32, all indices int32, tensors dense row-major ("contiguous").
 */
#ifndef CMFLOW_HIP_H
#define CMFLOW_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

/* ---- the three kernels CMFlow uses from the reference's `pointnet2_cuda` extension -------- */

/* Replaces ball_query_kernel_launcher_fast (lib/src/ball_query_gpu.cu:48-49; binding
 * lib/src/ball_query.cpp:14-25).  new_xyz (b,m,3), xyz (b,n,3) -> idx (b,m,nsample).
 * First `nsample` points with d2 < radius*radius in index order, padded with the first hit;
 * idx is left untouched for an empty ball (caller pre-zeroes it, lib/pointnet2_utils.py:246). */
int cmf_ball_query(int b, int n, int m, float radius, int nsample,
                   const float *new_xyz, const float *xyz, int *idx, void *stream);
/* The nq <= 4 ball queries of one multi-scale grouping call (utils/model_utils/radarflow_util.py:111-118; models/cmflow.py:21-22:
 * (r, nsample) = (2,4) (4,8) (8,16) (16,32)) over the same centres and cloud in ONE launch, optionally for nclouds <= 2 (centres,
 * cloud) pairs of equal geometry: idx[c * nq + q] is (B, M, nsamples[q]) and equals cmf_ball_query(radii[q], nsamples[q]) bit for
 * bit; zero_empty != 0 writes the rows of empty balls as zeros (callers that do not pre-zero idx).  n <= 1024.  radii / nsamples /
 * the pointer tables are HOST arrays. */
int cmf_ball_query_multi(int b, int n, int m, int nq, const float *radii, const int *nsamples, int nclouds,
                         const float *const *new_xyz, const float *const *xyz, int *const *idx, int zero_empty, void *stream);

/* Replaces group_points_kernel_launcher_fast (lib/src/group_points_gpu.cu:69-70).
 * points (b,c,n), idx (b,npoints,nsample) -> out (b,c,npoints,nsample). */
int cmf_group_points(int b, int c, int n, int npoints, int nsample,
                     const float *points, const int *idx, float *out, void *stream);

/* Replaces group_points_grad_kernel_launcher_fast (lib/src/group_points_gpu.cu:27-28).
 * grad_out (b,c,npoints,nsample), idx -> grad_points (b,c,n) += scatter (caller zero-fills,
 * lib/pointnet2_utils.py:218).  The reference adds with fp32 atomics (order varies run to run); here rows of <= 8192 entries
 * over <= 2048 targets (plan form) and per-centre lists of 16 / 32 / 64 slots over <= 8192 targets (pad-folded CSR gather)
 * are summed in a fixed order: bit-reproducible.  Other shapes use LDS atomics unless CMF_GROUP_GRAD_DETERMINISTIC=1. */
int cmf_group_points_grad(int b, int c, int n, int npoints, int nsample,
                          const float *grad_out, const int *idx, float *grad_points, void *stream);

/* QueryAndGroup.forward (lib/pointnet2_utils.py:269-292) as one call: ball query (cmf_ball_query semantics; an empty ball
 * groups point 0, the reference's pre-zeroed idx) + grouped xyz minus the centre (:279-280) + grouped features (:283),
 * concatenated as the reference does (:285): out (b, 3*use_xyz + c, m, nsample), relative xyz planes first.
 * new_xyz (b,m,3) centres, xyz (b,n,3), features (b,c,n) or NULL with c == 0 (then use_xyz must be set), idx (b,m,nsample)
 * optional output (what GroupingOperation.backward needs, :204).  n <= 1024 and 3*use_xyz + c <= 24: ONE launch (the
 * waves that find the neighbour lists gather them; measured faster up to that width); otherwise the query followed by one
 * gather launch that writes the relative-xyz planes and the LDS-staged feature planes.  idx == NULL: the indices live in the
 * library's per-stream scratch between the two launches. */
int cmf_query_and_group(int b, int n, int m, float radius, int nsample, int c, int use_xyz,
                        const float *new_xyz, const float *xyz, const float *features, int *idx, float *out, void *stream);

/* ---- torch-level hot loops of the reference, as kernels ------------------------------------ */

/* knn_point (utils/model_utils/radarflow_util.py:88-99 = square_distance :8-30 + topk).
 * xyz (b,n,3) database, new_xyz (b,s,3) queries -> idx (b,s,nsample) int32 in canonical order
 * (ascending distance, ties by lowest index); dist (b,s,nsample) optional (may be NULL).
 * nsample <= 32. */
int cmf_knn(int b, int n, int s, int nsample, const float *xyz, const float *new_xyz,
            int *idx, float *dist, void *stream);

/* WeightedKabsch (models/cmflow.py:128-169).  A, Bm (b,3,n) point sets, W (b,n) weights
 * (normalised by the caller, cmflow.py:105-106) -> trans (b,4,4).  aux (b,32) doubles receives
 * {U,S,V of H, cA, cB, D} for the backward pass (may be NULL). */
int cmf_weighted_kabsch(int b, int n, const float *A, const float *Bm, const float *W,
                        float *trans, double *aux, void *stream);

/* Backward of cmf_weighted_kabsch.  grad_trans (b,4,4) -> grad_A, grad_B (b,3,n), grad_W (b,n)
 * (any of the three may be NULL).  Uses the polar-factor derivative (well conditioned: divides
 * by s_i + s_j, not s_i^2 - s_j^2 as a generic SVD backward does). */
int cmf_weighted_kabsch_grad(int b, int n, const float *A, const float *Bm, const float *W,
                             const double *aux, const float *grad_trans,
                             float *grad_A, float *grad_B, float *grad_W, void *stream);

/* The ego-motion head and the rigid refinement around that solve (models/cmflow.py:96-125) as one call per direction:
 *   w = (score + eps) / sum(score + eps),  B = pc1 + flow,  trans = weighted_kabsch(pc1, B, w),  mask = score > thres,
 *   sf = mask ? (R pc1 + t - pc1) : flow
 * pc1, flow (b,3,n), score (b,n) -> W (b,n), Bm (b,3,n) [kept for the backward call], trans (b,4,4), aux (b,32) doubles,
 * sf (b,3,n), mask (b,n) bytes.  One wavefront per sample. */
int cmf_ego_refine(int b, int n, float eps, float thres, const float *pc1, const float *flow, const float *score,
                   float *W, float *Bm, float *trans, double *aux, float *sf, unsigned char *mask, void *stream);
/* Backward: g_sf (b,3,n), g_trans (b,4,4; may be NULL) -> g_flow (b,3,n), g_score (b,n; may be NULL); g_w (b,n) scratch. */
int cmf_ego_refine_grad(int b, int n, float eps, const float *pc1, const float *score, const float *W, const float *Bm,
                        const unsigned char *mask, const double *aux, const float *g_sf, const float *g_trans,
                        float *g_flow, float *g_w, float *g_score, void *stream);

/* ---- point-major grouping (the layout the fused path computes in) ------------------------------ */

/* Layout helpers of the host side (each one launch where torch takes a fill, a copy and sometimes a subtraction):
 * cmf_pad_rows: dst[r][c] = c < k ? src[r * ld_src + c] : 0 for c < ld_dst (rows padded to 16-byte multiples);
 * cmf_inputs_point_major: the model's four inputs as point-major rows -- pc1, pc2 (b,3,n) -> x1, x2 (b,n,3); ft1, ft2 (b,c,n) ->
 *   a1, a2 (b,n,cp) with zero columns c..cp-1 (models/cmflow.py:59-64 keeps them channel-major);
 * cmf_rel_xyz: out (b,m,S,4) = (xyz[b][idx[b][p][s]] - centre[b][p], 0), the neighbours' relative coordinates
 *   (utils/model_utils/radarflow_util.py:207-208). */
int cmf_pad_rows(long long rows, int k, const float *src, long long ld_src, float *dst, int ld_dst, void *stream);
int cmf_inputs_point_major(int b, int n, int c, int cp, const float *pc1, const float *pc2, const float *ft1, const float *ft2,
                           float *x1, float *x2, float *a1, float *a2, void *stream);
int cmf_rel_xyz(int b, int n, int m, int S, const float *xyz, const float *centre, const int *idx, float *out, void *stream);

/* Row gather: feat (b,n,ldf) rows of c floats, idx (b,entries) -> out (b,entries,c) with
 * out[b,e,:] = feat[b,idx[b,e],:].  Same operation as cmf_group_points on the transposed layout
 * (reference: lib/src/group_points_gpu.cu:47-66 via utils/model_utils/radarflow_util.py:52-63),
 * with coalesced 16-byte accesses instead of a 4-byte gather. */
int cmf_group_rows(int b, int n, int c, int ldf, int entries,
                   const float *feat, const int *idx, float *out, void *stream);

/* Inverse index of idx (b,entries) with values in [0,n): offsets (b,n+1), inv (b,entries) --
 * for each target point the entries that reference it, ascending.  Feeds cmf_group_rows_grad. */
int cmf_build_inverse(int b, int n, int entries, const int *idx, int *offsets, int *inv, void *stream);
/* Same, for idx of shape (b,P,S): small P / n take an O(P*S) LDS-matrix path instead of a scan. */
int cmf_build_inverse_ps(int b, int n, int P, int S, const int *idx, int *offsets, int *inv, void *stream);

/* Backward of cmf_group_rows as a deterministic segmented sum (replaces the fp32 atomics of
 * lib/src/group_points_gpu.cu:8-25): grad_feat[b,j,:] (row stride ldg) = (accumulate ? old : 0) +
 * sum_{e in inv(j)} grad_out[b,e,:], summed in ascending e. */
int cmf_group_rows_grad(int b, int n, int c, int ldg, int entries, int accumulate,
                        const float *grad_out, const int *offsets, const int *inv,
                        float *grad_feat, void *stream);

/* Weight gradient of a [1x1 conv + train-mode BN] layer with the BatchNorm backward of its output gradient fused into the
 * operand staging (replaces the torch-level BN backward + conv weight gradient of radarflow_util.py:151-153 in the backward
 * pass):  dZ = a * (dU - s1/rows - zhat * s2/rows), zhat = (Z - mean) * invstd, (s1 | s2) = sums[2][cout];
 * dW[cout][cin] (+)= dZ^T @ act(X), act = relu(prob_a * X + prob_c) per input channel or identity (both NULL).
 * dU, Z (rows, cout), X (rows, cin); dZ_out (rows, cout) receives dZ (optional, must not alias dU).  cout, cin multiples of
 * 128, rows a multiple of 16.  split_k > 1: deterministic split over the rows, workspace of split_k * cout * cin floats.
 * Bit-identical to cmf_bn_bwd_apply followed by cmf_gemm(a_t = 1, b_t = 0) for rows >= 32768. */
int cmf_gemm_dw_bn_bwd(int cout, int cin, long long rows, const float *dU, long long ldu, const float *Z, long long ldz,
                       const float *a, const float *mean, const float *invstd, const float *sums, float *dZ_out, long long ldo,
                       const float *X, long long ldx, const float *prob_a, const float *prob_c,
                       float *dW, long long lddw, int split_k, float *workspace, int accumulate, void *stream);

/* ---- fp32 MFMA GEMM with fused BatchNorm/activation prologue and epilogues ---------------------- *
 * Every 1x1 Conv2d of the reference (utils/model_utils/radarflow_util.py:132-139,174,246-251,
 * 299-305) is  C[M,N] = epi( pro(A)[M,K] * B[K,N] )  on point-major matrices.
 *   a_t: A is stored [K][M] (lda = row stride) instead of [M][K];  b_t: B is stored [N][K]
 *        (a conv weight (out,in)) instead of [K][N].
 *   pro_a/pro_c [K]  (a_t == 0 only): A' = relu(pro_a[k]*A + pro_c[k])   -- producer's BN+ReLU
 *   prob_a/prob_c [N] (b_t == 0 only): B' = relu(prob_a[n]*B + prob_c[n])
 *   bias [N], act: 0 none / 1 relu / 2 leaky(0.1) / 3 sigmoid
 *   stats: [ceil(M/128)][2][N] per-row-tile partial (sum, sum of squares) of the stored C
 *   bwd_mode 1: C = acc * [ea[n]*Z + ec[n] > 0], stats <- partial (sum C, sum C*(Z-emean)*einvstd)
 *   dxyz (bwd_mode 1, 2 or 3 with stats, optional): rows of (dx,dy,dz,0) per output row; stats becomes [tiles][5][N]
 *                with the three extra partials sum C*d_k -- the xyz-weight gradient of the set-conv's / cost volume's
 *                first conv comes out of the epilogue instead of a 4-column GEMM over the same rows
 *   bwd_mode 2: C = acc * (Z > 0 ? 1 : 0.1)        bwd_mode 3: C = acc * [Z > 0]     (stats: column sums of C in row 0)
 *   split_k > 1: contraction split over split_k slabs in `workspace` ([split_k][M][N] floats),
 *                summed in a fixed order by a second kernel (deterministic weight gradients)
 *   accumulate: C += result
 * Row strides (lda, ldb) must be multiples of 4 floats and A, B 16-byte aligned. */
int cmf_gemm(int M, int N, int K, int a_t, int b_t,
             const float *A, long long lda, const float *B, long long ldb, float *C, long long ldc,
             const float *pro_a, const float *pro_c, const float *prob_a, const float *prob_c,
             const float *bias, int act, float *stats,
             int bwd_mode, const float *Z, long long ldz,
             const float *ea, const float *ec, const float *emean, const float *einvstd,
             const float *dxyz, int split_k, float *workspace, int accumulate, void *stream);
int cmf_gemm_tiles_m(int M);
/* The persistent form of the tiled kernel (csrc/gemm_persist.hip: workgroups stay resident and write tile t out between the
 * MFMAs of tile t + 1) takes the data-gradient calls with a backward epilogue on interior shapes (M, N multiples of 128,
 * K a multiple of 16, K >= 192).  mode 0: never; 1: where a workgroup gets at least three tiles (default; also
 * CMF_GEMM_PERSIST=0|1|2); 2: wherever the shape allows.  grid: its workgroup count (multiple of 8), 0 = two per CU.
 * Outputs are bit-identical to the non-persistent kernel's; the column statistics are summed in a different (fixed) order.
 * Returns 0, -1 on an invalid argument.  Process-wide. */
int cmf_gemm_persist_config(int mode, int grid);
/* Diagnostics / tests: the <= 64-channel forward layers (cmf_gemm's per-wave kernels) have a full-tile body that issues all of a
 * wave's loads before its first wait; on = 1 sends full tiles through the general body as well (also CMF_THIN_GENERAL=1).  Outputs
 * and statistics of the two bodies are bit-identical.  Returns the previous setting.  Process-wide. */
int cmf_thin_general(int on);

/* Live timing of the tiled GEMM kernel for bench.py's `roofline` object.  Between _begin and _end every launch of the
 * tiled kernel with 2*M*N*K >= min_flops is bracketed by a HIP event pair on the stream it is launched on -- inside
 * the library, so launches issued by cmf_setconv_forward/_backward count like direct cmf_gemm calls.  _end synchronises
 * the device and returns the number of bracketed launches, the sum of their durations (ms) and of their FLOPs, and the
 * launch count / FLOPs of ALL cmf_gemm calls in the window (thin kernels included): the share the measurement covers.
 * Not re-entrant (one window at a time); every output pointer may be NULL. */
/* Diagnostics (tools/gemm_timeline.py): _arm makes the NEXT tiled launch record, per workgroup, {start, end of main loop,
 * end} on the 100 MHz wall clock, (xcc_id << 32 | HW_ID) and three epilogue stamps (first transposition visible, band-0 stores
 * issued, last band done); _read synchronises and copies the 8 x u64 records to host memory, returning the workgroup
 * count of that launch. */
int cmf_gemm_trace_arm(void);
long long cmf_gemm_trace_read(unsigned long long *host_out, long long max_workgroups);
typedef struct cmf_gemm_launch_record {
    int M, N, K;          /* as passed to cmf_gemm (C is M x N, contraction K) */
    int layout;           /* bit 1: a_t, bit 0: b_t, bit 2: persistent kernel */
    int split_k, kind;    /* epilogue kind: 0 raw store, 1 forward (bias/act/stats), 2 backward BN+ReLU, 3 backward (leaky) ReLU */
    int bm, bn;           /* block tile */
    float ms;             /* duration between the two events */
} cmf_gemm_launch_record;
int cmf_gemm_profile_begin(double min_flops);
/* Bracket only every n-th launch that reaches min_flops (default 1 = all; the event pairs themselves cost 0.13-0.18 ms per training step
 * when every large launch carries one); every eligible launch is still counted: cmf_gemm_profile_eligible (after cmf_gemm_profile_end). */
int cmf_gemm_profile_sampling(int every);
int cmf_gemm_profile_eligible(long long *launches, double *flops);
int cmf_gemm_profile_end(long long *launches_timed, double *ms_timed, double *flops_timed, long long *launches_all,
                         double *flops_all);
/* per-launch records of the last closed window (shape, layout, kind, duration); returns the number available */
long long cmf_gemm_profile_records(cmf_gemm_launch_record *out, long long max_records);

/* ---- BatchNorm / activation / pooling kernels around the GEMMs (point-major) ---------------------- *
 * Per-channel reductions use a partial buffer [ceil(rows/128)][2][C] (no atomics, fixed order). */

/* partial (sum, sumsq) over `count` rows -> mean, invstd (biased variance), running-stat update
 * (momentum, unbiased variance: torch BatchNorm2d train semantics, radarflow_util.py:133-139), and the
 * folded affine a = gamma*invstd, c = beta - mean*a.  tiles == 0: eval mode, fold the running stats.
 * num_batches_tracked (int64, optional) is incremented by one (nn.BatchNorm2d's counter). */
int cmf_bn_finalize(int tiles, int C, double count, const float *partial, const float *gamma,
                    const float *beta, float eps, float momentum, float *running_mean, float *running_var,
                    float *mean_out, float *invstd_out, float *a_out, float *c_out,
                    long long *num_batches_tracked, void *stream);
/* out[2][C] = sum over tiles of partial[t][2][C]; optionally acc0[C] += row 0 (dbeta), acc1[C] += row 1
 * (dgamma): the BN-backward sums land directly in the parameters' gradient buffers */
int cmf_colsum_finalize(int tiles, int C, const float *partial, float *out, float *acc0, float *acc1, void *stream);
/* out[ncols] = sum over tiles of partial[t][ncols] (fixed order); same optional accumulation of the first 2*C columns */
int cmf_colsum(int tiles, int ncols, const float *partial, float *out, int C, float *acc0, float *acc1, void *stream);

/* Set-conv / cost-volume grouping with the first 1x1 conv hoisted per point
 * (QueryAndGroup + first Conv2d, lib/pointnet2_utils.py:277-285 + radarflow_util.py:151;
 *  cost volume radarflow_util.py:207-216):
 *   z[b,p,s,:] = act( ysrc[b,idx[b,p,s],:] + (yctr ? yctr[b,p,:] : 0) + Wx (xyz_src[b,idx] - xyz_ctr[b,p]) )
 * ysrc (b,n_src,ld_src), yctr (b,P,ld_ctr) or NULL, Wx (C,3) with row stride ldw, idx (b,P,S);
 * act 0 none / 2 leaky(0.1); z (b,P,S,C); dxyz (b,P,S,4) relative coordinates (optional);
 * partial: BN statistics of z (optional); partial_x (optional, with partial): [tiles][3*C+4] extra sums
 * sum z*d_k (k=0..2, per channel) and sum d_k, used by cmf_setconv_dwx in the backward pass.
 * z == NULL (yctr == NULL, partial given): the statistics only, bit-identical to the writing form's. */
int cmf_group_affine(int b, int n_src, int P, int S, int C,
                     const float *ysrc, int ld_src, const float *yctr, int ld_ctr,
                     const float *xyz_src, const float *xyz_ctr, const float *Wx, int ldw,
                     const int *idx, int act, float *z, float *dxyz, float *partial, float *partial_x, void *stream);

/* The same first layer WITHOUT the (b,P,S,C) tensor, for inference (nothing of it is needed afterwards): cmf_group_prep writes what
 * depends on the neighbour lists only -- rows[b,p,s] = b*n_src + idx[b,p,s], dxyz (b,P,S,4) = xyz_src[idx] - xyz_ctr[p], wx3 (3,C) the
 * coordinate columns of Wx as planes -- and cmf_gemm_gather_affine is the NEXT layer's GEMM
 *   out[m,n] = sum_k relu( pro_a[k] * ( Y[rows[m],k] + wx3[0,k] dx_m + wx3[1,k] dy_m + wx3[2,k] dz_m ) + pro_c[k] ) * W[n,k]
 * with that layer formed in its A-operand path (rows gathered by the LDS-direct loads, the coordinate term + BN + ReLU applied to
 * the fragments).  Same operations in the same order as cmf_group_affine followed by cmf_gemm with the A prologue: bit-identical.
 * M, N multiples of 128, K a multiple of 16, pointers 16-byte aligned. */
int cmf_group_prep(int b, int n_src, int P, int S, int C, const float *xyz_src, const float *xyz_ctr, const float *Wx, int ldw,
                   const int *idx, int *rows, float *dxyz, float *wx3, void *stream);
int cmf_gemm_gather_affine(int M, int N, int K, const float *Y, long long ldy, const int *rows, const float *dxyz,
                           const float *wx3, const float *pro_a, const float *pro_c, const float *W, long long ldw,
                           float *C, long long ldc, float *stats /* [M/128][2][N] partial sums of the output, or NULL */, void *stream);
/* ... and the weight gradient of that next layer with the first layer formed in its B-operand staging (register-staged loop, the
 * rows' source indices requested one chunk ahead):
 *   dW[cout,cin] (+)= sum_r dZ[r,cout] * relu( prob_a[k] * ( Y[rows[r],k] + wx3[:,k] . dxyz[r] ) + prob_c[k] )
 * bit-identical to cmf_gemm(a_t = 1, b_t = 0, prob_a, prob_c) on the materialised tensor.  cout, cin multiples of 128, nrows of 16;
 * split_k > 1: deterministic slabs in `workspace` (split_k * cout * cin floats). */
/* ... and the data gradient INTO that first layer (cmf_gemm's backward kind with dxyz: mask by the first layer's BN + ReLU, BN-backward
 * and dxyz partial sums [tiles_m][5][cin]) with the first layer's pre-activations formed from the per-point rows in the epilogue:
 *   dU[m,k] = (dZ @ W)[m,k] * [ea[k] z + ec[k] > 0],   z = Y[rows[m],k] + wx3[:,k] . dxyz[m]
 * bit-identical to cmf_gemm(bwd_mode = 1, Z = the materialised tensor, dxyz) in the non-persistent kernel.  M, cin multiples of 128. */
int cmf_gemm_dx_gather(int M, int cin, int cout, const float *dZ, long long ldz, const float *W, long long ldw,
                       float *dU, long long ldu, const float *Y, long long ldy, const int *rows, const float *dxyz,
                       const float *wx3, const float *ea, const float *ec, const float *emean, const float *einvstd,
                       float *stats, void *stream);
/* ... and the same data gradient when only its sums per source point are needed (the grouping's backward pass): the rows walk the
 * slots in inverse-index order (cmf_group_perm: arows[m] the slot at position m, pts[m] its source point, dxyz2 its relative
 * coordinates) and the kernel stores no dU -- for every run of equal source points inside a 64-row range it writes the run's column
 * sums to pieces[(point + m / 64)][cin] (P + M / 64 rows), which cmf_group_rows_grad_bn_cf_pieces adds per point in range order. */
int cmf_group_perm(int b, int entries, const int *inv, const int *rows, const float *dxyz, int *perm, int *pts, float *dxyz2, void *stream);
int cmf_gemm_dx_gather_sum(int M, int cin, int cout, const float *dZ, long long ldz, const float *W, long long ldw,
                           const float *Y, long long ldy, const int *arows, const int *pts, const float *dxyz2,
                           const float *wx3, const float *ea, const float *ec, const float *emean, const float *einvstd,
                           float *pieces, float *stats, void *stream);
int cmf_gemm_dw_gather(int cout, int cin, long long nrows, const float *dZ, long long ldz, const float *Y, long long ldy,
                       const int *rows, const float *dxyz, const float *wx3, const float *prob_a, const float *prob_c,
                       float *dW, long long lddw, int split_k, float *workspace, int accumulate, void *stream);
/* cmf_gemm_dw_gather with the train-mode BatchNorm backward of its output gradient formed in the A-operand staging as in
 * cmf_gemm_dw_bn_bwd (dU, Z (nrows, cout); sums[2][cout]; dZ_out (nrows, cout) receives dZ and must not alias dU): bit-identical to
 * cmf_bn_bwd_apply followed by cmf_gemm_dw_gather. */
int cmf_gemm_dw_gather_bn_bwd(int cout, int cin, long long nrows, const float *dU, long long ldu, const float *Z, long long ldz,
                              const float *a, const float *mean, const float *invstd, const float *sums, float *dZ_out, long long ldo,
                              const float *Y, long long ldy, const int *rows, const float *dxyz, const float *wx3,
                              const float *prob_a, const float *prob_c, float *dW, long long lddw, int split_k, float *workspace,
                              int accumulate, void *stream);
/* The split count with which cmf_gemm_dw_gather takes its 256-row tiles (workspace: split * cout * cin floats); 0 = no preference. */
int cmf_gemm_dw_gather_split(int cout, int cin, long long nrows);

/* out[p,:] = max_s relu(a*z[p,s,:] + c): BN + ReLU + max over the ball (radarflow_util.py:151-155);
 * argmax (P,C) uint8 optional. */
int cmf_bn_relu_maxpool(long long P, int S, int C, const float *z, const float *a, const float *c,
                        float *out, long long ldo, unsigned char *argmax, void *stream);
int cmf_maxpool_bwd(long long P, int S, int C, const float *dout, long long ldd, const float *z,
                    const float *a, const float *c, const float *mean, const float *invstd,
                    const unsigned char *argmax, float *dU, float *partial, void *stream);

/* out = relu(a*z + c) (row strides ldz / ldo: out may be a column slice of a concat buffer) */
int cmf_affine_relu(long long M, int C, const float *z, long long ldz, const float *a, const float *c,
                    float *out, long long ldo, void *stream);
/* dU = dY * [a*z + c > 0] + partial (sum dU, sum dU*(z-mean)*invstd) */
int cmf_act_bwd_stats(long long M, int C, const float *dY, long long ldy, const float *z, long long ldz,
                      const float *a, const float *c, const float *mean, const float *invstd,
                      float *dU, float *partial, void *stream);
/* in place: dZ = a*(dU - s1/M - zhat*s2/M) with sums = {s1[C], s2[C]}; sums == NULL: dZ = a*dU (eval BN) */
int cmf_bn_bwd_apply(long long M, int C, float *dU, const float *z, long long ldz, const float *a,
                     const float *mean, const float *invstd, const float *sums, void *stream);

/* One backward layer of a NARROW [1x1 conv + BatchNorm + ReLU] stack (cout, cin <= 64, cout % 8 == 0; the reference's
 * mlp_convs / mlp2_convs of PointLocalFeature, radarflow_util.py:144-162, at its 32 / 64 channel widths) in one pass over
 * the rows -- what bn_bwd_apply + the weight-gradient cmf_gemm + the data-gradient cmf_gemm compute as three kernels
 * streaming eight [rows, C] matrices is done here streaming four:
 *   dZ    = a (dU - s1/rows - (z - mean) invstd s2/rows)          (sums = {s1[cout], s2[cout]}; NULL: dZ = a dU, eval BN)
 *   dx    = mask_in(dZ @ w),  mask_in = [a_in x + c_in > 0]       in_mode 1: x is the pre-BN output of the layer below;
 *           stats[tile][2 or 5][cin] get the per-128-row-tile sums of dx, dx (x - mean_in) invstd_in (and dx * dxyz_k when
 *           dxyz is given), as cmf_gemm's backward epilogue does.  in_mode 0: x is an activated input, dx = dZ @ w.
 *   dw (+)= dZ^T @ act_in(x),  act_in = relu(a_in x + c_in) (in_mode 1) or identity; summed over cmf_thin_bwd_slabs(rows)
 *           slabs [cout][cin] in `slabs` in fixed order (deterministic).
 * dx == NULL skips the data gradient (and stats), dw == NULL the weight gradient.  dU and z rows must be 16-byte aligned. */
int cmf_thin_bwd_supported(int cout, int cin);
int cmf_thin_bwd_slabs(long long rows, int *tiles_per_workgroup);
int cmf_thin_bwd_layer(long long rows, int cout, int cin, const float *dU, long long lddu, const float *z, long long ldz,
                       const float *a, const float *mean, const float *invstd, const float *sums,
                       const float *w, long long ldw, const float *x, long long ldx, int in_mode,
                       const float *a_in, const float *c_in, const float *mean_in, const float *invstd_in, const float *dxyz,
                       float *dx, long long lddx, float *stats, float *dw, long long lddw, int accumulate, float *slabs,
                       void *stream);

/* The same layer directly behind the max over the ball (radarflow_util.py:155-157 backward), dense: rows = P * S, row
 * (p, s) of dU is (s == argmax[p][:]) ? g[p][:] : 0 and is formed on the fly from the per-point arrays of
 * cmf_maxpool_bwd_point -- the [P*S, cout] gradient of the pooled tensor is never stored.  x is the pre-BN output of the
 * layer below (in_mode 1); whole 128-row tiles and multiples of 32 channels only. */
int cmf_maxpool_bwd_point(long long P, int S, int C, const float *dout, long long ldd, const float *z,
                          const float *a, const float *c, const float *mean, const float *invstd,
                          const unsigned char *argmax, float *g, float *partial, void *stream);
int cmf_thin_bwd_layer_pooled(long long P, int S, int cout, int cin, const float *g, const unsigned char *argmax,
                              const float *z, const float *a, const float *mean, const float *invstd, const float *sums,
                              const float *w, const float *x, const float *a_in, const float *c_in, const float *mean_in,
                              const float *invstd_in, float *dx, float *stats, float *dw, int accumulate, float *slabs,
                              void *stream);

/* The fused backward layer for a WIDE input: cout = 64, cin a multiple of 128 (<= 1024), whole 128-row tiles, input
 * through BN + ReLU (in_mode 1), both products -- the 256 -> 64 conv of a second-encoder block (radarflow_util.py:144-162),
 * whose two gradients are the worst GEMM shapes of the step (K = 64 / 64 output rows).  dU is either given
 * ([rows][64], row stride lddu) or pooled (pool_g != NULL: rows = P * pool_S, per-point arrays of cmf_maxpool_bwd_point).
 * slabs: cmf_thin_bwd_wide_slabs(rows, cin, NULL) x 64 x cin floats of workspace. */
int cmf_thin_bwd_wide_supported(int cout, int cin);
int cmf_thin_bwd_wide_slabs(long long rows, int cin, int *tiles_per_workgroup);
int cmf_thin_bwd_wide_layer(long long rows, int cin, const float *dU, long long lddu, const float *pool_g,
                            const unsigned char *pool_am, int pool_S, const float *z, long long ldz,
                            const float *a, const float *mean, const float *invstd, const float *sums,
                            const float *w, long long ldw, const float *x, long long ldx,
                            const float *a_in, const float *c_in, const float *mean_in, const float *invstd_in,
                            float *dx, long long lddx, float *stats, float *dw, long long lddw, int accumulate, float *slabs,
                            void *stream);

/* Backward of the set-conv's grouping with the BatchNorm backward of the first layer fused in
 * (radarflow_util.py:148-151 backward): dZ = a*(dU - s1/M - zhat*s2/M) is formed on the fly from dU and z,
 * summed over the inverse index into grad_feat (b,n,c) with row stride ldg, and never written.  sums = {s1[C], s2[C]} or NULL
 * (eval-mode BN: dZ = a*dU). */
int cmf_group_rows_grad_bn(int b, int n, int c, int entries, const float *dU, const float *z,
                           const float *a, const float *mean, const float *invstd, const float *sums,
                           float inv_count, const int *offsets, const int *inv, float *grad_feat, int ldg, void *stream);
/* The same result without reading z, for z produced by cmf_group_affine from per-point rows: every entry e of inv(j)
 * gathers the same source row, z[e,:] = y[j,:] + wx . (xyz_src[j] - xyz_ctr[e / S]), so
 *   sum_e (z[e,:] - mean) = cnt_j*(y[j,:] - mean) + wx . D_j,   D_j = sum_e (xyz_src[j] - xyz_ctr[e / S]),
 * and only dU is streamed.  y (b,n,c) rows with stride ldy, wx (c,3) rows with stride ldw, xyz_src (b,n,3),
 * xyz_ctr (b, entries/S, 3); entries = centres * S in (centre, slot) order. */
int cmf_group_rows_grad_bn_cf(int b, int n, int c, int entries, int S, const float *dU, const float *y, long long ldy,
                              const float *wx, long long ldw, const float *xyz_src, const float *xyz_ctr,
                              const float *a, const float *mean, const float *invstd, const float *sums,
                              float inv_count, const int *offsets, const int *inv, float *grad_feat, int ldg, void *stream);
/* ... with the sums of dU over every point's slots already formed by cmf_gemm_dx_gather_sum (`pieces`): dU is not read */
int cmf_group_rows_grad_bn_cf_pieces(int b, int n, int c, int entries, int S, const float *pieces, const float *y, long long ldy,
                                     const float *wx, long long ldw, const float *xyz_src, const float *xyz_ctr,
                                     const float *a, const float *mean, const float *invstd, const float *sums,
                                     float inv_count, const int *offsets, const int *inv, float *grad_feat, int ldg, void *stream);

/* dW_xyz of the set-conv's first conv from column sums only (no pass over the grouped tensor):
 *   dWx[c,k] = a_c*( q_k[c] - (s1_c/M)*u_k - (s2_c/M)*invstd_c*(tz_k[c] - mean_c*u_k) )
 * bwd5 = {s1,s2,q0,q1,q2}[C] (cmf_gemm dxyz partials, reduced), fwd = {tz0,tz1,tz2}[C] then {u0,u1,u2,.}
 * (cmf_group_affine extra partials, reduced); train == 0: dWx = a*q.  dwx rows have stride ld floats
 * (the xyz columns of the (out, 3+C) conv weight gradient); accumulate: += instead of =. */
int cmf_setconv_dwx(int C, float inv_count, int train, const float *bwd5, const float *fwd,
                    const float *a, const float *mean, const float *invstd, float *dwx, int ld, int accumulate,
                    void *stream);

/* ---- the rest of the reference's pointnet2_cuda extension (lib/src/pointnet2_api.cpp:10-25) ------------ *
 * Not called by CMFlow (only by the unused lib/pointnet2_modules.py); provided so the drop-in module is
 * complete.  Argument order = the reference launchers. */
/* gather_points_kernel_launcher_fast (sampling_gpu.cu:27-44): out[b,c,j] = points[b,c,idx[b,j]] */
int cmf_gather_points(int b, int c, int n, int npoints, const float *points, const int *idx, float *out, void *stream);
/* gather_points_grad_kernel_launcher_fast (sampling_gpu.cu:66-83): accumulates into grad_points */
int cmf_gather_points_grad(int b, int c, int n, int npoints, const float *grad_out, const int *idx,
                           float *grad_points, void *stream);
/* furthest_point_sampling_kernel_launcher (sampling_gpu.cu:212-250): temp (b,n) pre-filled with 1e10 */
int cmf_furthest_point_sampling(int b, int n, int m, const float *dataset, float *temp, int *idxs, void *stream);
/* knn_kernel_launcher_fast (interpolate_gpu.cu:60-78): k nearest `known` points per `unknown`, ascending,
 * first-seen wins ties, squared distances; k <= 64 here (reference: k <= 200) */
int cmf_knn_points(int b, int n, int m, int k, const float *unknown, const float *known,
                   float *dist2, int *idx, void *stream);
/* three_nn_kernel_launcher_fast (interpolate_gpu.cu:127-146) */
int cmf_three_nn(int b, int n, int m, const float *unknown, const float *known, float *dist2, int *idx, void *stream);
/* three_interpolate_kernel_launcher_fast (interpolate_gpu.cu:172-189) and its grad (:217-233) */
int cmf_three_interpolate(int b, int c, int m, int n, const float *points, const int *idx,
                          const float *weight, float *out, void *stream);
int cmf_three_interpolate_grad(int b, int c, int n, int m, const float *grad_out, const int *idx,
                               const float *weight, float *grad_points, void *stream);

/* ---- one set-conv block (PointLocalFeature, radarflow_util.py:121-162) per call ------------------------- *
 * Host-side sequencing of the kernels above for one (radius, nsample) scale: ball query -> gather with the
 * hoisted first conv -> (BN, ReLU, 1x1 conv) x 2 -> BN + ReLU + max over the ball -> (1x1 conv, BN, ReLU) x 3.
 * All device memory is caller provided: `saved` keeps what backward needs, `scratch` is transient; sizes come
 * from cmf_setconv_sizes.  The forward and backward calls must see the same descriptor geometry and `saved`. */
typedef struct cmf_setconv_desc {
    int B, N, S;                 /* samples, points per sample, nsample (<= 64) */
    int O1;                      /* channels of the hoisted first conv */
    int C[5];                    /* output channels of layers 2..6 */
    float radius;
    int training;                /* 1: batch statistics + running-stat update; 0: fold the running statistics */
    float eps[6], momentum[6];
    const float *xyz;            /* (B,N,3) */
    const float *y; long long ldy;      /* (B,N,O1) rows, stride ldy: feats @ W_f^T */
    const float *wx; long long ldwx;    /* (O1,3) xyz columns of the first conv weight, row stride ldwx */
    const float *w[5];           /* layers 2..6 weights (C[i], C_in) dense */
    const float *gamma[6], *beta[6];
    float *rmean[6], *rvar[6];
    long long *nbt[6];           /* num_batches_tracked counters (may be NULL) */
    float *saved, *scratch;
    float *out; long long ldo;   /* (B*N, C[4]) */
    /* backward */
    const float *dout; long long lddout;
    float *dy; long long lddy;   /* (B,N,O1) rows with stride lddy (0 = dense O1), or NULL */
    float *dwx; long long lddwx; int acc_wx;
    float *dw[5]; int acc_w[5];  /* weight gradients: written (0) or accumulated into (1) */
    float *dgamma[6], *dbeta[6]; int acc_bn[6];
    int inference;               /* 1 (with training == 0): no backward call will follow -- the forward call may skip what only the
                                    backward pass reads (the grouped first-layer tensor: cmf_gemm_gather_affine) */
    int idx_ready;               /* 1: this block's ball query has been issued already (cmf_setconv_queries wrote the indices into
                                    `saved`); the forward call does not launch its own */
} cmf_setconv_desc;
/* Arena sizes in floats (each pointer may be NULL).  The arenas are sized by PATH: a block whose shapes take the gathering GEMMs
 * (M = B*N*S and O1, C[0] multiples of 128, ldy a multiple of 4: the second encoder) keeps M row indices + 3*O1 floats where the
 * materialised first-layer tensor would take M x O1, and -- in scratch_bwd, when d->dy is set at the time of the sizes call, i.e. the
 * backward call will produce an input gradient -- (B*N + M/64) x O1 + 6 M floats where the data gradient into that layer would
 * take M x O1.  scratch_bwd asked with d->dy == NULL is never smaller than with it set, so it is valid for either backward call.
 * With the compact sizes d->y and d->w[0] must be 16-byte aligned (the forward / backward calls refuse them otherwise). */
int cmf_setconv_sizes(const cmf_setconv_desc *d, long long *saved_floats, long long *scratch_fwd, long long *scratch_bwd);
/* Device memory the library holds on the current device outside caller-provided arenas: the per-(stream, slot) scratch buffers of the
 * drop-in entry points (cmf_ball_query spill lists, cmf_group_points_grad plans / inverse indices), live + retired, in bytes. */
long long cmf_mem_stats(void);
/* The ball queries of n blocks, issued ahead of their forward calls on ONE stream: blocks that share centres and cloud (the scales of
 * a MultiScaleEncoder call, radarflow_util.py:111-118) are served by one cmf_ball_query_multi launch, two clouds of equal geometry
 * (the weight-shared first encoder, models/cmflow.py:72-73) by the same launch -- 12 launches per CMFlow forward become 2.  The
 * indices land where cmf_setconv_forward keeps them (head of `saved`); the caller then sets idx_ready in the descriptors. */
int cmf_setconv_queries(int n, const cmf_setconv_desc *descs, void *stream);
int cmf_setconv_forward(const cmf_setconv_desc *d, void *stream);
int cmf_setconv_backward(const cmf_setconv_desc *d, void *stream);
/* The optimizer step of the reference's loop (main.py:107: torch.optim.Adam with L2 weight decay) over a flat gradient bucket in ONE
 * launch: grad / m / v are flat arrays of `total` floats in bucket order, params[t] the address of tensor t and offsets[t] its first
 * element in the bucket (offsets[n_tensors] = total; both tables in device memory).  step >= 1 is the count INCLUDING this update.
 *   g' = g + wd p;  m += (1 - b1)(g' - m);  v = b2 v + (1 - b2) g'^2;  p -= lr / (1 - b1^step) * m / (sqrt(v) / sqrt(1 - b2^step) + eps) */
int cmf_adam_step(int n_tensors, const long long *offsets, float *const *params, long long total, const float *grad, float *m, float *v,
                  double lr, double beta1, double beta2, double eps, double weight_decay, long long step, void *stream);

/* Float offsets, inside `saved`, of the six per-layer BatchNorm blocks (mean | invstd | a | c, 4*C_l floats each). */
int cmf_setconv_bn_offsets(const cmf_setconv_desc *d, long long *offsets6);

/* [1x1 conv (no bias) + BatchNorm + ReLU] x L on a materialised input: the stacks of the flow / motion heads
 * (utils/model_utils/radarflow_util.py:253-261,277-285: 512 -> 256 -> 128 -> 64) sequenced by the library like a set-conv
 * block -- forward GEMM with train-mode statistics, fold; backward: BN-backward sums, BN backward in place, weight
 * gradient (deterministic split-K), masked data gradient.  Issued from Python one kernel at a time these chains were host
 * bound (25 launches of 5-45 us with 15-30 us between them: the two heads' backward passes took 1.0 ms of a 22 ms step with
 * the GPU idle).  C[0] = input channels, C[l] = output channels of layer l (1..L, L <= 4, all multiples of 4).
 * saved: z_1 .. z_L and the BN blocks (mean | invstd | a | c) ; scratch: partial sums, gradient buffers, split-K slabs. */
typedef struct cmf_mlp_desc {
    long long M; int L; int C[5]; int training;
    float eps[4], momentum[4];
    const float *x; long long ldx;                       /* (M, C[0]) activated input, row stride ldx */
    const float *w[4];                                   /* (C[l], C[l-1]) dense */
    const float *gamma[4], *beta[4];
    float *rmean[4], *rvar[4]; long long *nbt[4];        /* running statistics (nbt NULL: counter not touched) */
    float *saved, *scratch;
    float *out; long long ldo;                           /* forward: relu(bn_L(z_L)), (M, C[L]) */
    const float *dout; long long lddout;                 /* backward */
    float *dx; long long lddx;                           /* (M, C[0]) or NULL */
    float *dw[4]; int acc_w[4];
    float *dgamma[4], *dbeta[4]; int acc_bn[4];
} cmf_mlp_desc;
int cmf_mlp_sizes(const cmf_mlp_desc *d, long long *saved_floats, long long *scratch_fwd, long long *scratch_bwd);
int cmf_mlp_forward(const cmf_mlp_desc *d, void *stream);
int cmf_mlp_backward(const cmf_mlp_desc *d, void *stream);

/* Deferred nn.BatchNorm2d running-statistics update.  A weight-shared encoder is called twice per step
 * (cmflow.py:72-73); to run the two calls CONCURRENTLY they are issued with rmean/rvar/nbt == NULL (batch statistics
 * only) and the momentum updates are applied afterwards, in call order, from each call's saved batch mean / invstd.
 * table (device memory): one entry per BN layer; offset = float offset of that layer's BN block inside a call's `saved`
 * arena (saved0 = first call, saved1 = second call). */
typedef struct cmf_bn_update_entry {
    float *rmean, *rvar; long long *nbt;
    int C; float momentum, eps; double count;      /* channels; bn.momentum, bn.eps; rows the statistics were taken over */
    long long offset;
} cmf_bn_update_entry;
int cmf_bn_running_update(int n_entries, const cmf_bn_update_entry *table, int n_calls, const float *saved0,
                          const float *saved1, void *stream);

/* The independent scales of a MultiScaleEncoder (radarflow_util.py:101-118) in one call: descs[i] is issued on
 * streams[i] from its own host thread inside the library (n <= 16).  The caller orders the streams against its own
 * (events before and after); nothing is synchronised. */
int cmf_setconv_forward_multi(int n, const cmf_setconv_desc *descs, void *const *streams);
int cmf_setconv_backward_multi(int n, const cmf_setconv_desc *descs, void *const *streams);
/* The same with the per-point tails (layers 4-6: three <= 64-channel layers over the B*N points) taken out of the chains and
 * run for ALL blocks of the call as batched launches on one stream -- one launch per kernel of the tail instead of one per
 * block (at N = 256 these kernels are 128 workgroups of latency each; the eight chains of an encoder call spent 1.5 ms of
 * the 22 ms training step in them).  Forward: cmf_setconv_forward_heads_multi (up to the max over the ball, per stream),
 * join the streams, cmf_setconv_tail_forward(n, descs, stream).  Backward: cmf_setconv_tail_backward(n, descs, stream) (from
 * dout to the gradient of the pooled features + the gradients of layers 4-6), fork, cmf_setconv_backward_bodies_multi.
 * Same kernels, same order per block: results are bit-identical to cmf_setconv_forward / _backward.  CMF_TAIL_BATCH=0 (or
 * blocks whose tails differ in width / mode) runs the tails block by block. */
int cmf_setconv_forward_heads_multi(int n, const cmf_setconv_desc *descs, void *const *streams);
/* 1 when cmf_setconv_forward_heads_multi will run these blocks' slot-level bodies in lock step as batched launches on streams[0]
 * alone (narrow blocks with train-mode BatchNorm and indices ready: the first encoder) -- the caller may then pass its own stream for
 * every entry and needs no fork / join around the call; else 0. */
int cmf_setconv_forward_bodies_batched(int n, const cmf_setconv_desc *descs);
int cmf_setconv_backward_bodies_batched(int n, const cmf_setconv_desc *descs);      /* the same for cmf_setconv_backward_bodies_multi */
int cmf_setconv_tail_forward(int n, const cmf_setconv_desc *descs, void *stream);
int cmf_setconv_tail_backward(int n, const cmf_setconv_desc *descs, void *stream);
int cmf_setconv_backward_bodies_multi(int n, const cmf_setconv_desc *descs, void *const *streams);

/* Cost-volume weighting (radarflow_util.py:219-221,235-236): out[m,c] = sum_k w[m,k,c] * x[m,k,c] over rows
 * m = sample*n1 + point.  idx == NULL: x is (M,K,C) dense.  idx (M,K) int32: x is (samples*n_src, C) per-point rows
 * and the k-th operand of row m is x[sample*n_src + idx[m,k]] (the grouped tensor is never materialised).
 * The gradient call writes dw = dcost*x and dx = dcost*w, each (M,K,C) (either may be NULL).  leaky bit 0: x is a
 * stored LeakyReLU(0.1) activation and dx is multiplied by its derivative; bit 1: w is a stored ReLU activation
 * (WeightNet, radarflow_util.py:307-318) and dw is masked by w > 0 -- gradients w.r.t. the pre-activations.
 * C % 4 == 0, 16-byte aligned pointers. */
int cmf_weighted_ksum(long long M, int K, int C, int n1, int n_src, const float *w, const float *x, const int *idx,
                      float *out, void *stream);
/* dx_colsum (optional): [cmf_weighted_ksum_grad_tiles(C)][C] per-workgroup column sums of dx, to be reduced with
 * cmf_colsum -- the bias gradient of the layer that produced x; cmf_weighted_ksum_grad_tiles returns 0 when C does not
 * allow it. */
int cmf_weighted_ksum_grad_tiles(int C);
int cmf_weighted_ksum_grad(long long M, int K, int C, int n1, int n_src, int leaky, const float *dcost, const float *w,
                           const float *x, const int *idx, float *dw, float *dx, float *dx_colsum, void *stream);
/* The same weighting with WeightNet's last layer (radarflow_util.py:307-318: Conv2d(8, C, 1) + ReLU) folded in: the
 * weights are not an operand but recomputed where they are used, w[m,k,c] = relu(bl[c] + sum_j h[m,k,j] * Wl[c,j]) with
 * h (M*K, 8) the hidden activation, Wl (C, 8), bl (C) -- the (M,K,C) weight tensor and its gradient never exist.
 * C in {256, 512, 1024} (cmf_weightnet_ksum_tiles(C) > 0), M*K < 2^31, x / idx / leaky bit 0 as above.
 * The gradient call writes dx (M,K,C), dh (M*K, 8) and one partial row per workgroup,
 * part [cmf_weightnet_ksum_tiles(C)][C*8 + C + C + 8] = sums of dWl (C,8) | dbl (C) | column sums of dx (C) | column sums
 * of dh (8), to be reduced in fixed order with cmf_colsum.  leaky bit 2 (value 4): h is itself a stored ReLU activation
 * (WeightNet's second hidden layer) -- dh is then masked by h > 0, i.e. it is the gradient w.r.t. that layer's
 * pre-activation and its column sums are that layer's bias gradient.  dcost has row stride ldd floats (>= C: it may be a column block of a wider gradient). */
int cmf_weightnet_ksum_tiles(int C);
int cmf_weightnet_ksum(long long M, int K, int C, int n1, int n_src, const float *h, const float *Wl, const float *bl,
                       const float *x, const int *idx, float *out, void *stream);
int cmf_weightnet_ksum_grad(long long M, int K, int C, int n1, int n_src, int leaky, const float *dcost, long long ldd, const float *h,
                            const float *Wl, const float *bl, const float *x, const int *idx, float *dx, float *dh,
                            float *part, void *stream);

/* Global feature of Backbone (cmflow.py:76-81,89-91: torch.max over the points, expand, cat): out[b,n,0:C] = f[b,n,:],
 * out[b,n,C:2C] = max_n' f[b,n',:], point-major rows with strides ldf / ldo floats (multiples of 4, so out may be a
 * column block of a wider buffer); arg (B,C) int32 = first row attaining the maximum.  The gradient call returns
 * df[b,n,c] = dout[b,n,c] + (n == arg[b,c]) * sum_n' dout[b,n',C+c].  C % 4 == 0, 16-byte aligned pointers. */
int cmf_global_max_cat(int B, int N, int C, const float *f, long long ldf, float *out, long long ldo, int *arg, void *stream);
int cmf_global_max_cat_grad(int B, int N, int C, const float *dout, long long ldd, const int *arg, float *df, long long ldf,
                            void *stream);

/* Weight of the stacked first conv of a MultiScaleEncoder (radarflow_util.py:132-139: the feature half of every scale's
 * first 1x1 conv applied to the SAME input, here one GEMM): wf[(s*O1 + r)][c], Kp columns, from the n_w conv weights
 * w[s] = [O1][3 + cin] (xyz columns first): columns [0, cin - n_tail) = input channels n_tail.., then the first n_tail
 * input channels, then zeros up to Kp.  cmf_unstack_first_conv_grad adds the GEMM's weight gradient dwf [n_w*O1][Kp] back
 * into the conv weights' gradients g[s] (same layout as w[s]; the xyz columns are not touched).  n_w <= 8; w / g are
 * HOST arrays of device pointers. */
int cmf_stack_first_conv(int n_w, int O1, int cin, int n_tail, int Kp, const float *const *w, float *wf, void *stream);
int cmf_unstack_first_conv_grad(int n_w, int O1, int cin, int n_tail, int Kp, const float *dwf, float *const *g, void *stream);

/* ---- the training step's loss -------------------------------------------------------------
 * RadarFlowLoss of losses/radar_loss.py:260-292 for model 'cmflow' / 'cmflow_t': SoftChamfer (:17-58),
 * SpatialSmoothness (:60-97), RadialDisplacement (:99-122), EgoMotion (:162-183), MotionSeg (:185-205),
 * OpticalFlow (:207-243 with utils/util.py:31-58) and DynamicFlow (:245-258), forward AND the gradient of the
 * weighted total with respect to the three network outputs, in one call (3 launches, no host sync; the
 * reference reads 8 loss items back with .item(), :156-159,285-288).  Layouts are the reference's:
 * clouds/flows (B,3,N), per-point scalars (B,N), opt (B,N,2), transforms (B,4,4) row-major.
 * items[9] = total, Loss (self-supervised sum), smoothnessLoss, chamferLoss, veloLoss, egoLoss, maskLoss,
 * opticalLoss, superviseLoss.  Gradient outputs may be NULL (evaluation).
 * Cloud size: num_nb < N <= CMF_RADAR_LOSS_MAX_N, num_nb in {4, 8, 16} (the reference has no limit, radar_loss.py:60-97).
 * N <= 704 with num_nb == 8 (the reference's training size is 256) keeps a sample in one workgroup's LDS; everything else
 * takes the tiled form (the cloud streamed through LDS in tiles of 256 points, a sample's lists in `workspace`): same terms,
 * same per-point arithmetic, the per-sample sums folded in another fixed order.  cmf_radar_loss_tiled forces the tiled
 * form at any size (tests compare the two forms on the same input). */
#define CMF_RADAR_LOSS_MAX_N 65536
typedef struct cmf_radar_loss_desc {
    int B, N;
    const float *pc1, *pc2, *pred_f, *gt_f;                                    /* (B,3,N) */
    const float *vel1, *mseg_pre, *mseg_gt, *dyn_mask, *radar_u, *radar_v;     /* (B,N) */
    const float *opt;                                                          /* (B,N,2) */
    const float *pre_trans, *gt_trans;                                         /* (B,4,4) */
    const float *camera_inverse;                                               /* (3,3) inverse intrinsics */
    const float *t_camera_radar;                                               /* (4,4) */
    float w_self, w_em, w_ms, w_opt, w_dyn;                                    /* radar_loss.py:262: 1,1,1,0.1,1 */
    float zeta, alpha; int num_nb; float lower_bound;                          /* 0.005, 0.5, 8, 0.25 */
    int self_only;             /* 1: model 'raflow' (radar_loss.py:274-276) -- only the three self-supervised terms;
                                  the inputs of the other four may be NULL, their items are reported as 0 */
    float *items;                                                              /* [9] */
    float *d_pred_f, *d_pre_trans, *d_mseg_pre;                                /* (B,3,N), (B,4,4), (B,N) or NULL */
    float *workspace;                                                          /* cmf_radar_loss_workspace_nb floats */
} cmf_radar_loss_desc;
long long cmf_radar_loss_workspace(int b, int n);                  /* = cmf_radar_loss_workspace_nb(b, n, 8) */
long long cmf_radar_loss_workspace_nb(int b, int n, int num_nb);   /* floats; num_nb as in the descriptor */
int cmf_radar_loss(const cmf_radar_loss_desc *d, void *stream);
int cmf_radar_loss_tiled(const cmf_radar_loss_desc *d, void *stream);   /* workspace: cmf_radar_loss_workspace_tiled */
long long cmf_radar_loss_workspace_tiled(int b, int n, int num_nb);

/* ---- evaluation metrics of one batch ------------------------------------------------------
 * utils/eval_util.py: eval_scene_flow :42-86, eval_motion_seg :104-118, eval_trans_RPE :89-102 (with
 * utils/odometry_util.py:61-160), which copy every tensor to the host and run numpy.  pc (B,3,N) as
 * main_util.py:175 passes it; pred, labels (B,N,3); mask, pred_m (B,N) with 1 = static; transforms (B,4,4);
 * (r_res, theta_res, phi_res) the radar resolution of dataset/vod.py:21-23.
 * metrics[14] (fp64) = rne, 50-50 rne, mov_rne, stat_rne, sas, ras, epe, accs, accr, acc, miou, sen, RTE, RAE.
 * workspace: 16 * B doubles. */
int cmf_eval_metrics(int b, int n, const float *pc, const float *pred, const float *labels, const float *mask,
                     const float *pred_m, const float *gt_trans, const float *pred_trans,
                     float r_res, float theta_res, float phi_res, double *metrics, double *workspace, void *stream);

/* Pseudo labels of the training step (main_util.py:63-67): dyn_mask = extract_dynamic_from_fg (:209-225), mseg_gt =
 * where(dyn_mask == 1, mseg_label_RRV (:253-265), dyn_mask).  pc1 (B,3,N); gt_trans (B,4,4); vel1, fg_mask (B,N)
 * (fg_mask 1 = background); interval (B); flow_label (B,N,3).  Outputs (B,N) floats, 1 = static; residual may be NULL. */
int cmf_pseudo_labels(int b, int n, const float *pc1, const float *gt_trans, const float *vel1, const float *interval,
                      const float *fg_mask, const float *flow_label, float vr_thres,
                      float *dyn_mask, float *mseg_gt, float *residual, void *stream);

/* Test-only: occupies `stream` for about `microseconds` with a one-wave kernel that polls the constant 100 MHz clock
 * (s_sleep between polls: no measurable load on the chip).  tests/test_gpu_stress.py uses it to shift the relative timing of
 * the side-stream chains at every fork point (fused_blocks.stress_*): results must not depend on it. */
int cmf_debug_spin(float microseconds, void *stream);

/* Library / device identification: returns a static NUL-terminated string. */
const char *cmf_version(void);

#ifdef __cplusplus
}
#endif
#endif /* CMFLOW_HIP_H */
