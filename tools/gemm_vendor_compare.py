"""cmf_gemm against the vendor fp32 GEMM behind torch.mm (rocBLAS / hipBLASLt, TF32 off) on the plain shapes of the training
step -- no prologue, no epilogue, the three operand layouts.  TFLOP/s of 2*M*N*K over the mean of 20 launches.

    python tools/gemm_vendor_compare.py
"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmflow_amd.fused import gemm
from cmflow_amd.fused_blocks import gemm_dw

torch.backends.cuda.matmul.allow_tf32 = False
dev = torch.device("cuda:0")


def rate(fn, flops, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return flops / (e0.elapsed_time(e1) / n * 1e-3) / 1e12


print("%-28s %10s %10s" % ("shape (M x N x K), layout", "cmf_gemm", "torch.mm"))
for M, N, K in [(524288, 256, 512), (524288, 512, 256), (131072, 512, 512), (16384, 2048, 1024)]:
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev); dZ = torch.randn(M, N, device=dev)
    out = torch.empty(M, N, device=dev); dx = torch.empty(M, K, device=dev); dw = torch.empty(N, K, device=dev)
    f = 2.0 * M * N * K
    Wt = W.t()
    print("%-28s %10.1f %10.1f" % ("%dx%dx%d  A W^T" % (M, N, K), rate(lambda: gemm(A, W, out=out), f), rate(lambda: torch.mm(A, Wt, out=out), f)))
    print("%-28s %10.1f %10.1f" % ("%dx%dx%d  dZ W" % (M, K, N), rate(lambda: gemm(dZ, W, b_t=False, out=dx), f), rate(lambda: torch.mm(dZ, W, out=dx), f)))
    dZt = dZ.t()
    print("%-28s %10.1f %10.1f" % ("%dx%dx%d  dZ^T A" % (N, K, M), rate(lambda: gemm_dw(dZ, A), f), rate(lambda: torch.mm(dZt, A, out=dw), f)))
    del A, W, dZ, out, dx, dw
