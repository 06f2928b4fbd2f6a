"""Microbenchmark of the narrow forward layers (thin_fwd: BN+ReLU prologue, statistics epilogue) and the kernels around
them in the first encoder, at B = 64, N = 256: achieved GB/s of algorithmic bytes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmflow_amd import _lib, fused_blocks as FB
from cmflow_amd.fused import gemm
dev = torch.device("cuda:0")

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

print("%-34s %9s %8s %7s" % ("kernel", "rows", "us", "GB/s"))
for rows in (524288, 262144, 131072, 65536, 16384):
    for cin, cout in ((32, 32), (32, 64), (64, 64)):
        x = torch.randn(rows, cin, device=dev); w = torch.randn(cout, cin, device=dev)
        a, c = torch.rand(cin, device=dev) + 0.5, torch.randn(cin, device=dev)
        out = torch.empty(rows, cout, device=dev)
        t = timeit(lambda: gemm(x, w, pro=(a, c), stats=True, out=out))
        print("%-34s %9d %8.1f %7.0f" % ("thin_fwd %d->%d +pro +stats" % (cin, cout), rows, t, 4.0 * rows * (cin + cout) / t / 1e3))
    if rows >= 65536:
        S = rows // 16384
        z = torch.randn(16384, S, 64, device=dev)
        st = FB.BNState(); st.a, st.c = torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev)
        t = timeit(lambda: FB.bn_relu_maxpool(z, st))
        print("%-34s %9d %8.1f %7.0f" % ("bn_relu_maxpool C=64 S=%d" % S, rows, t, 4.0 * rows * 64 / t / 1e3))
