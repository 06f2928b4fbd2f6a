import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmflow_amd.fused import gemm
from cmflow_amd.fused_blocks import gemm_dw, dw_split
dev = torch.device("cuda:0")
M, N, K = 16384, 2048, 1040
dy = torch.randn(M, N, device=dev); X = torch.randn(M, K, device=dev)
def rate(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
ref = gemm_dw(dy, X)
out = torch.empty(N, K, device=dev)
def split_form(s1, s2):
    gemm(dy, X[:, :1024], a_t=True, b_t=False, split_k=s1, out=out[:, :1024])
    gemm(dy, X[:, 1024:], a_t=True, b_t=False, split_k=s2, out=out[:, 1024:])
print("one call (split %d): %.1f us" % (dw_split(M, N, K), rate(lambda: gemm_dw(dy, X))))
for s1 in (16,):
    for s2 in (48, 64, 96, 128, 192, 256):
        t = rate(lambda: split_form(s1, s2))
        split_form(s1, s2); torch.cuda.synchronize()
        print("head split %2d + tail split %2d: %.1f us   maxdiff %.2e" % (s1, s2, t, float((out - ref).abs().max())))
