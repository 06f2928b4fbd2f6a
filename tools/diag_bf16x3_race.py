"""Is cmf_gemm bit-reproducible under concurrent load, in both arithmetic modes and with every prologue / epilogue?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmflow_amd import _lib
from cmflow_amd.fused import gemm
dev = torch.device("cuda:0")
torch.manual_seed(0)
side = torch.cuda.Stream()
M, N, K = 131072, 256, 512
A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev)
pa, pc = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.1
dZ = torch.randn(M, N, device=dev); Zs = torch.randn(M, K, device=dev)
ea, ec, em, ei = (torch.rand(K, device=dev) + 0.5 for _ in range(4))
A2 = torch.randn(65536, 512, device=dev); W2 = torch.randn(512, 512, device=dev)
cases = {
    "plain": lambda: gemm(A, W),
    "stats": lambda: gemm(A, W, stats=True)[0],
    "pro+stats": lambda: gemm(A, W, pro=(pa, pc), stats=True)[0],
    "pro+stats (partials)": lambda: gemm(A, W, pro=(pa, pc), stats=True)[1],
    "dx mode1": lambda: gemm(dZ, W, b_t=False, bwd=(1, Zs, ea, ec, em, ei))[0],
    "dx mode2": lambda: gemm(dZ, W, b_t=False, bwd=(2, Zs)),
}
for mode in ("fp32", "bf16x3"):
    _lib.set_gemm_mode(mode)
    for name, fn in cases.items():
        ref = fn().clone()
        bad = 0
        for it in range(12):
            with torch.cuda.stream(side):
                for _ in range(3):
                    gemm(A2, W2)                      # load on another stream
            out = fn()
            torch.cuda.synchronize()
            if not torch.equal(out, ref):
                bad += 1
                d = (out - ref).abs().max().item()
        print(mode, name, "mismatching repeats:", bad, ("max diff %.3g" % d) if bad else "")
