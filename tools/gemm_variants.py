"""cmf_gemm on the model's dominant shapes, plain and with the fused prologue/epilogue the model uses (HIP-event
timed, 10 launches each).  Diagnostic for kernel work: prints TFLOP/s per variant."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmflow_amd.fused import gemm
dev = torch.device("cuda:0")
torch.manual_seed(0)


def timed(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


def report(name, M, N, K, fn):
    t = timed(fn)
    print("%-58s M=%7d N=%5d K=%5d  %7.1f us  %6.1f TF" % (name, M, N, K, t * 1e6, 2.0 * M * N * K / t / 1e12), flush=True)


only = sys.argv[1] if len(sys.argv) > 1 else ""
shapes = ((524288, 256, 512), (131072, 512, 512), (16384, 2048, 1028), (262144, 256, 512)) if 'stacked' not in only else ((16384, 2048, 1028), (16384, 2048, 8), (16384, 512, 512), (131072, 256, 512), (65536, 256, 512))
for (M, N, K) in shapes:
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev)
    pa, pc = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.1
    bias = torch.randn(N, device=dev)
    if "fwd" in only or not only:
        report("fwd plain", M, N, K, lambda: gemm(A, W))
        report("fwd +stats", M, N, K, lambda: gemm(A, W, stats=True))
        report("fwd +BN/ReLU prologue +stats", M, N, K, lambda: gemm(A, W, pro=(pa, pc), stats=True))
        report("fwd +bias +leaky", M, N, K, lambda: gemm(A, W, bias=bias, act=2))
    # dX: dZ[M][N] @ W[N][K] -> [M][K]
    dZ = torch.randn(M, N, device=dev); Zs = torch.randn(M, K, device=dev)
    ea, ec, em, ei = (torch.rand(K, device=dev) + 0.5 for _ in range(4))
    if "dx" in only or not only:
        report("dX plain", M, K, N, lambda: gemm(dZ, W, b_t=False))
        report("dX +relu mask +BN sums (mode 1)", M, K, N, lambda: gemm(dZ, W, b_t=False, bwd=(1, Zs, ea, ec, em, ei)))
        report("dX +leaky mask (mode 2)", M, K, N, lambda: gemm(dZ, W, b_t=False, bwd=(2, Zs)))
    if "dw" in only or not only:
        tiles = ((N + 127) // 128) * ((K + 127) // 128)
        for sk in sorted({8, 16, 24, 32, 48, 64, 96, 128}):
            report("dW split_k=%d" % sk, N, K, M, lambda: gemm(dZ, A, a_t=True, b_t=False, split_k=sk))
        report("dW split_k=64 +BN/ReLU on B", N, K, M, lambda: gemm(dZ, A, a_t=True, b_t=False, split_k=64, prob=(pa, pc)))
    del A, W, dZ, Zs
