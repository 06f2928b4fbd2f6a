"""Is the K=512/N=256 forward GEMM limited by HBM latency of the A panel?  Compare A streamed from HBM
with A served from cache (all rows alias one row: lda = 0)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmflow_amd.fused import gemm
dev = torch.device("cuda:0")
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for M, N, K in ((524288, 256, 512), (524288, 128, 512), (524288, 512, 512)):
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev)
    out = torch.empty(M, N, device=dev)
    A0 = torch.randn(1, K, device=dev).expand(M, K)
    a = t(lambda: gemm(A, W, out=out)); b = t(lambda: gemm(A0, W, out=out))
    print("M=%d N=%d K=%d  HBM A: %.3f ms %.1f TF   cached A: %.3f ms %.1f TF" % (M, N, K, a, 2e-9*M*N*K/a, b, 2e-9*M*N*K/b))
