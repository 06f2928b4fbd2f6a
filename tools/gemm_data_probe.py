"""Does the operand distribution / placement move cmf_gemm's rate?  Same shapes, A and W filled N(0,1), U(-1,1), or zeros."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmflow_amd.fused import gemm
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


for M, N, K in [(524288, 256, 512), (131072, 512, 512), (16384, 2048, 1024)]:
    out = torch.empty(M, N, device=dev)
    res = []
    for name, mk in (("randn", lambda *s: torch.randn(*s, device=dev)), ("uniform", lambda *s: torch.rand(*s, device=dev) * 2 - 1),
                     ("zeros", lambda *s: torch.zeros(*s, device=dev)), ("randn", lambda *s: torch.randn(*s, device=dev))):
        A, W = mk(M, K), mk(N, K)
        res.append("%s %.1f" % (name, rate(lambda: gemm(A, W, out=out), 2.0 * M * N * K)))
        del A, W
    print("%dx%dx%d  " % (M, N, K) + "   ".join(res), flush=True)
