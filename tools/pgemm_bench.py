"""Persistent cmf_gemm (csrc/gemm_persist.hip) against the non-persistent kernel on the model's data-gradient shapes:
HIP-event timed, 10 back-to-back launches each, alternating A/B/A/B in one process (same box, same clocks)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmflow_amd import _lib
from cmflow_amd.fused import gemm
dev = torch.device("cuda:0")
torch.manual_seed(0)
L = _lib.lib()


def timed(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


shapes = [(524288, 512, 256), (262144, 512, 256), (131072, 512, 256), (65536, 512, 256), (131072, 512, 512)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
GRID = int(os.environ.get("PGRID", "0"))            # workgroups of the persistent kernel (0: two per CU)
for (M, N, K) in shapes:
    dZ = torch.randn(M, K, device=dev); W = torch.randn(K, N, device=dev); Zs = torch.randn(M, N, device=dev)
    ea, ec, em, ei = (torch.rand(N, device=dev) + 0.5 for _ in range(4))
    d4 = torch.randn(M, 4, device=dev)
    out = torch.empty(M, N, device=dev)
    kinds = [("bwd BN+ReLU (2)", (1, Zs, ea, ec, em, ei), False), ("bwd BN+ReLU + dxyz (4)", (1, Zs, ea, ec, em, ei, d4), False),
             ("bwd leaky (3)", (2, Zs), False), ("bwd leaky + dxyz (5)", (2, Zs, None, None, None, None, d4), True)]
    for name, bwd, st in kinds:
        res = []
        for rep in range(2):
            for mode in (0, 2):
                L.cmf_gemm_persist_config(mode, GRID)
                t = timed(lambda: gemm(dZ, W, b_t=False, bwd=bwd, stats=st, out=out))
                res.append(2.0 * M * N * K / t / 1e12)
        L.cmf_gemm_persist_config(1, 0)
        print("%-26s M=%7d N=%4d K=%4d  tiled %6.1f %6.1f TF   persistent %6.1f %6.1f TF" % (name, M, N, K, res[0], res[2], res[1], res[3]), flush=True)
    del dZ, W, Zs, out
# forward (A[M][K] W[N][K]) and weight-gradient (A[K][M] B[K][N], split-K) forms on the model's shapes
fshapes = [(524288, 256, 512), (131072, 512, 512), (262144, 256, 512)] if len(sys.argv) <= 1 else []
for (M, N, K) in fshapes:
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev)
    pa, pc = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.1
    out = torch.empty(M, N, device=dev)
    for name, kw in (("fwd plain", {}), ("fwd +stats", dict(stats=True)), ("fwd +BN/ReLU prologue +stats", dict(pro=(pa, pc), stats=True))):
        res = []
        for rep in range(2):
            for mode in (0, 2):
                L.cmf_gemm_persist_config(mode, GRID)
                t = timed(lambda: gemm(A, W, out=out, **kw))
                res.append(2.0 * M * N * K / t / 1e12)
        L.cmf_gemm_persist_config(1, 0)
        print("%-30s M=%7d N=%4d K=%4d  tiled %6.1f %6.1f TF   persistent %6.1f %6.1f TF" % (name, M, N, K, res[0], res[2], res[1], res[3]), flush=True)
    dZ = torch.randn(M, N, device=dev)
    for name, kw in (("dW split 96", dict(split_k=96)), ("dW split 96 +BN/ReLU on B", dict(split_k=96, prob=(pa, pc)))):
        res = []
        for rep in range(2):
            for mode in (0, 2):
                L.cmf_gemm_persist_config(mode, GRID)
                t = timed(lambda: gemm(dZ, A, a_t=True, b_t=False, **kw))
                res.append(2.0 * M * N * K / t / 1e12)
        L.cmf_gemm_persist_config(1, 0)
        print("%-30s M=%7d N=%4d K=%4d  tiled %6.1f %6.1f TF   persistent %6.1f %6.1f TF" % (name, N, K, M, res[0], res[2], res[1], res[3]), flush=True)
    del A, W, out, dZ
