"""cmf_gemm on the step's four dominant forms under the run-time diagnostic switches of the kernel (CMF_GEMM_DIAG_RT, read
once per process: 1 no operand loads after the prefetch, 2 no wait for them, 4 no barrier, 8 no epilogue).  Timing only --
the switched-off variants compute garbage.  Usage: CMF_GEMM_DIAG_RT=8 python tools/gemm_diag.py [M]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmflow_amd.fused import gemm
dev = torch.device("cuda:0")
torch.manual_seed(0)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 524288


def timed(fn, n=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


def report(name, m, n, k, fn):
    t = timed(fn)
    print("diag=%-3s %-40s %7d x %4d x %6d  %7.1f us  %6.1f TF" % (os.environ.get("CMF_GEMM_DIAG_RT", "0"), name, m, n, k, t * 1e6,
                                                                  2.0 * m * n * k / t / 1e12), flush=True)


N, K = 256, 512
A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev)
pa, pc = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.1
dZ = torch.randn(M, N, device=dev); Zs = torch.randn(M, K, device=dev)
ea, ec, em, ei = (torch.rand(K, device=dev) + 0.5 for _ in range(4))
dxyz = torch.randn(M, 4, device=dev)
if os.environ.get("CMF_GEMM_DIAG_RT", "0") == "0":        # results are meaningful: check the plain forms against torch
    torch.backends.cuda.matmul.allow_tf32 = False
    for name, got, want in (("fwd", gemm(A[:4096], W), A[:4096] @ W.t()), ("dX", gemm(dZ[:4096], W, b_t=False), dZ[:4096] @ W),
                            ("dW", gemm(dZ, A, a_t=True, b_t=False, split_k=int(os.environ.get("DW_SPLIT", "96"))), dZ.t() @ A)):
        print("check %-4s max rel err %.2e" % (name, float((got - want).abs().max() / want.abs().max())), flush=True)
report("fwd plain", M, N, K, lambda: gemm(A, W))
report("fwd +prologue +stats", M, N, K, lambda: gemm(A, W, pro=(pa, pc), stats=True))
report("dX plain (K=256)", M, K, N, lambda: gemm(dZ, W, b_t=False))
report("dX BN+ReLU + dxyz sums (K=256)", M, K, N, lambda: gemm(dZ, W, b_t=False, bwd=(1, Zs, ea, ec, em, ei, dxyz)))
sk = int(os.environ.get("DW_SPLIT", "96"))
report("dW split %d" % sk, N, K, M, lambda: gemm(dZ, A, a_t=True, b_t=False, split_k=sk))
