"""The 16384-row GEMMs of the heads / per-point tails / cost volume in isolation (forward layouts): TF per shape."""
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


for M, N, K in ((16384, 512, 512), (16384, 256, 512), (16384, 512, 256), (16384, 128, 256), (16384, 256, 128), (16384, 2048, 1040), (16384, 1024, 2048)):
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev); Wt = torch.randn(K, N, device=dev)
    pa, pc = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.3
    out = torch.empty(M, N, device=dev)
    f = 2.0 * M * N * K
    r1 = rate(lambda: gemm(A, W, out=out), f)
    r2 = rate(lambda: gemm(A, W, out=out, pro=(pa, pc), stats=True), f)
    r3 = rate(lambda: gemm(A, Wt, b_t=False, out=out), f)
    print("%6d x %4d x %4d   fwd plain %6.1f   fwd pro+stats %6.1f   dX plain %6.1f TF" % (M, N, K, r1, r2, r3), flush=True)
