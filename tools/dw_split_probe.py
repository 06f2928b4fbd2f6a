"""Weight-gradient layout (A[K][M], B[K][N], split-K) of cmf_gemm: tile shape x split count on the model's dW shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmflow_amd.fused import gemm
dev = torch.device("cuda:0")


def rate(fn, flops, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return flops / (e0.elapsed_time(e1) / n * 1e-3) / 1e12


for (M, N, K), splits in (((256, 512, 524288), (64, 96, 128, 192)), ((512, 512, 131072), (32, 48, 64, 96)), ((2048, 1040, 16384), (4, 8, 16)),
                          ((256, 512, 131072), (64, 96, 128)), ((512, 512, 16384), (32, 48, 64))):
    dZ = torch.randn(K, M, device=dev); X = torch.randn(K, N, device=dev)
    res = []
    for sk in splits:
        res.append("split %3d: %6.1f" % (sk, rate(lambda: gemm(dZ, X, a_t=True, b_t=False, split_k=sk), 2.0 * M * N * K)))
    print("%4d x %4d x %6d  " % (M, N, K) + "   ".join(res), flush=True)
    del dZ, X
