"""Does the power-of-two row pitch of the GEMM operands matter?  cmf_gemm on the model's shapes with the rows of A / Z / C at their
natural pitch (K or N floats) and at pitch + PAD floats (views of wider tensors): HIP-event timed, 10 back-to-back launches, A/B/A/B."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmflow_amd import _lib
from cmflow_amd.fused import gemm
dev = torch.device("cuda:0")
torch.manual_seed(0)
PAD = int(os.environ.get("PAD", "16"))


def timed(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


def mat(M, C, pad):
    return torch.randn(M, C + pad, device=dev)[:, :C]


for (M, N, K) in [(524288, 256, 512), (131072, 512, 512)]:
    W = torch.randn(N, K, device=dev)
    res = {}
    for rep in range(2):
        for pad in (0, PAD):
            A = mat(M, K, pad); out = mat(M, N, pad)
            for name, kw in (("fwd plain", {}), ("fwd +stats", dict(stats=True))):
                res.setdefault((name, pad), []).append(2.0 * M * N * K / timed(lambda: gemm(A, W, out=out, **kw)) / 1e12)
            del A, out
    for name in ("fwd plain", "fwd +stats"):
        print("%-12s M=%7d N=%4d K=%4d  pitch natural %s TF   pitch +%d %s TF" % (name, M, N, K, " ".join("%.1f" % v for v in res[(name, 0)]), PAD,
                                                                             " ".join("%.1f" % v for v in res[(name, PAD)])), flush=True)
for (M, N, K) in [(524288, 512, 256), (131072, 512, 512)]:
    W = torch.randn(K, N, device=dev)
    ea, ec, em, ei = (torch.rand(N, device=dev) + 0.5 for _ in range(4))
    res = {}
    for rep in range(2):
        for pad in (0, PAD):
            dZ = mat(M, K, pad); Zs = mat(M, N, pad); out = mat(M, N, pad)
            res.setdefault(pad, []).append(2.0 * M * N * K / timed(lambda: gemm(dZ, W, b_t=False, bwd=(1, Zs, ea, ec, em, ei), out=out)) / 1e12)
            del dZ, Zs, out
    print("bwd BN+ReLU  M=%7d N=%4d K=%4d  pitch natural %s TF   pitch +%d %s TF" % (M, N, K, " ".join("%.1f" % v for v in res[0]), PAD,
                                                                               " ".join("%.1f" % v for v in res[PAD])), flush=True)
