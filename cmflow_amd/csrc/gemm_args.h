// Argument block shared by the tiled GEMM (gemm.hip) and the thin GEMMs (thin_gemm.hip); see cmf_gemm in
// include/cmflow_hip.h for the meaning of every field.
#pragma once
#include <hip/hip_runtime.h>

struct GemmArgs {
    int M, N, K;                // C is MxN, contraction length K (this split's range is [k_begin,k_end))
    const float *A; long long lda;
    const float *B; long long ldb;
    float *C; long long ldc;
    // prologue on A (A_MK only): per-contraction-index affine + relu
    const float *pro_a, *pro_c;
    // prologue on B by output column (used by the weight-gradient GEMM: B = activated layer input)
    const float *prob_a, *prob_c;
    // epilogue
    const float *bias;          // [N] or null
    int act;                    // 0 none, 1 relu, 2 leaky(0.1), 3 sigmoid
    float *stats;               // [tiles_m][2][N] partial (sum, sumsq) of the STORED values, or null
    // backward epilogue: multiply by the activation derivative of the producer layer
    //   mode 1 (BN+ReLU):  dU = acc * [ea[n]*Z[m,n] + ec[n] > 0];  partials s1 = sum dU,
    //                      s2 = sum dU * (Z - mean[n]) * invstd[n]  -> stats[tiles_m][2][N]
    //   mode 2 (leaky):    dZ = acc * (Z[m,n] > 0 ? 1 : 0.1)
    //   mode 3 (relu on stored activation Z>0)
    int bwd_mode;
    const float *Z; long long ldz;
    const float *ea, *ec, *emean, *einvstd;
    const float *dxyz;          // bwd_mode 1 only, optional: rows of (dx,dy,dz,0); adds partials q_k = sum dU * d_k (k=0..2)
                                // -> stats is then [tiles_m][5][N] (s1, s2, q0, q1, q2): the set-conv dW_xyz without a pass
    // A_KM (weight-gradient) layout only: the A operand is formed while staging as the train-mode BN backward of two streams,
    //   A[k][m] = al[m] * A[k][m] + be[m] * (bnb_z[k][m] - mean[m]) + ga[m]   (cmf_common.h cmf_bnb_coef from bnb_a / mean / invstd / sums)
    // and the workgroups of the first column tile write it to bnb_out (the dZ the data-gradient GEMM reads next)
    const float *bnb_z; long long ldbz;
    const float *bnb_a, *bnb_mean, *bnb_invstd, *bnb_sums;      // sums: [2][M] (s1 | s2)
    float bnb_ic;                                               // 1 / rows
    float *bnb_out; long long ldbo;
    // gathering A operand (cmf_gemm_gather_affine, A[M][K] layout only): row m of A is
    //   act( Y[ga_rows[m]][k] + wx0[k] dx_m + wx1[k] dy_m + wx2[k] dz_m )   -- the set-conv first layer, never materialised
    // A points at Y (row pitch lda), ga_dxyz holds (dx, dy, dz, 0) per row, ga_wx the three planes [3][K]; act = pro_a / pro_c
    const int *ga_rows; const float *ga_dxyz; const float *ga_wx;
    // GMODE 4 (cmf_gemm_dx_gather_sum): the rows of the data gradient walk the neighbour slots in inverse-index order (sorted by source
    // point): ga_arows[m] = the row of A for output row m, ga_rows[m] = its source point; nothing of C is stored -- the sums over runs of
    // equal source points inside a 64-row range go to ga_pieces[(point + range)][N]
    const int *ga_arows; float *ga_pieces;
    int ga_y32;                 // GMODE 4: every byte offset into the per-point matrix fits 32 bits (rows through a buffer descriptor + scalar offsets)
    int split_k;                // >1: C is [split][M][N] partial slabs (ldc = N), reduced by a second kernel
    int accumulate;             // C += result (beta = 1)
    int thin_general;           // diagnostics (env CMF_THIN_GENERAL=1): the narrow forward layers take their general body on full tiles too (A/B)
    int no_direct;              // diagnostics (env CMF_GEMM_NO_DIRECT=1): register-staged main loop everywhere
    int diag;                   // timing diagnostics, results invalid (env CMF_GEMM_DIAG_RT bits: 1 no loads in the loop, 2 no vmcnt waits, 4 no barrier,
                                // 8 no epilogue (nothing stored), 16 epilogue at raised wave priority)
    unsigned long long *trace;  // diagnostics (cmf_gemm_trace): per workgroup {t_start, t_mainloop_end, t_end (100 MHz wall clock), xcc_id << 32 | hw_id}
};

// thin_gemm.hip: returns -1 when the shape is not handled there
int cmf_thin_gemm(const GemmArgs &g, int a_t, int b_t, hipStream_t st);
// gemm.hip: C[M][N] (+)= sum of split_k slabs [M][N], fixed order
int cmf_splitk_reduce(int M, int N, int split_k, const float *workspace, float *C, long long ldc, int accumulate, hipStream_t st);
void cmf_gemm_count_flops(double flops);
// gemm_persist.hip: the persistent kernel (overlapped epilogue) for the calls it takes; grid = 0: not this call
int cmf_pgemm_grid(GemmArgs &g, int a_t, int b_t, int kind);      // may lower g.split_k (weight gradients)
int cmf_pgemm_launch(const GemmArgs &g, int a_t, int b_t, int kind, int grid, hipStream_t st);
