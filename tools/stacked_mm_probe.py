"""The three plain GEMMs of the stacked first conv of the second encoder (P = 16384 rows, 1040 -> 4 x 512 channels) through
cmf_gemm and through the vendor GEMM behind torch.mm, in the exact operand forms (strided views) the step uses."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmflow_amd.fused import gemm
from cmflow_amd.fused_blocks import gemm_dw
torch.backends.cuda.matmul.allow_tf32 = False
dev = torch.device("cuda:0")
M, Kp, N, ng = 16384, 1040, 2048, 1024

def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

feats = torch.randn(M, Kp, device=dev); wf = torch.randn(N, Kp, device=dev); dy = torch.randn(M, N, device=dev)
y = torch.empty(M, N, device=dev); dfeats = torch.empty(M, Kp, device=dev); dwf = torch.empty(N, Kp, device=dev)
rows = [("fwd  feats @ wf^T", 2.0 * M * N * Kp, lambda: gemm(feats, wf, out=y), lambda: torch.mm(feats, wf.t(), out=y)),
        ("dX   dy @ wf[:, :1024]", 2.0 * M * N * ng, lambda: gemm(dy, wf[:, :ng], b_t=False, out=dfeats[:, :ng]),
         lambda: torch.mm(dy, wf[:, :ng], out=dfeats[:, :ng])),
        ("dW   dy^T @ feats", 2.0 * M * N * Kp, lambda: gemm_dw(dy, feats), lambda: torch.mm(dy.t(), feats, out=dwf))]
for name, fl, a, b in rows:
    ta, tb = t(a), t(b)
    print("%-26s cmf_gemm %7.1f us %6.1f TF | torch.mm %7.1f us %6.1f TF" % (name, ta, fl / ta / 1e6, tb, fl / tb / 1e6))
# 512-wide plain GEMMs of the cost volume / heads
for (m, n, k) in [(16384, 512, 512), (16384, 512, 256), (16384, 256, 512)]:
    A = torch.randn(m, k, device=dev); W = torch.randn(n, k, device=dev); o = torch.empty(m, n, device=dev)
    ta, tb = t(lambda: gemm(A, W, out=o)), t(lambda: torch.mm(A, W.t(), out=o))
    print("%-26s cmf_gemm %7.1f us %6.1f TF | torch.mm %7.1f us %6.1f TF" % ("%dx%dx%d A W^T" % (m, n, k), ta, 2.0*m*n*k/ta/1e6, tb, 2.0*m*n*k/tb/1e6))
