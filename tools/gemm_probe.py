"""Micro-probe for cmf_gemm (used under rocprofv3 --pmc): a few launches of the dominant shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmflow_amd.fused import gemm
from cmflow_amd.fused_blocks import gemm_dw
dev = torch.device("cuda:0")
M, N, K = 524288, 256, 512
A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev); dZ = torch.randn(M, N, device=dev)
for _ in range(3):
    gemm(A, W)                      # forward  (A[M][K], W[N][K])
    gemm(dZ, W, b_t=False)          # dX       (dZ[M][N] @ W[N][K])
    gemm_dw(dZ, A)                  # dW       (dZ^T @ A)
torch.cuda.synchronize()
