"""Time the fused WeightNet-tail weighting against the materialised-weights path at the north-star shape."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmflow_amd.fused import Neighbors
from cmflow_amd.fused_blocks import WeightNetKSumFn, WeightedKSumFn, linear
dev = torch.device("cuda:0")
B, N, K, C = 64, 256, 8, 512
g = torch.Generator().manual_seed(0)
h = torch.relu(torch.randn(B, N, K, 8, generator=g)).to(dev).requires_grad_(True)
wl = (0.3 * torch.randn(C, 8, generator=g)).to(dev).requires_grad_(True)
bl = torch.randn(C, generator=g).to(dev).requires_grad_(True)
x = torch.randn(B, N, K, C, generator=g).to(dev).requires_grad_(True)
p = torch.randn(B, N, C, generator=g).to(dev).requires_grad_(True)
nbr = Neighbors(torch.randint(0, N, (B, N, K), generator=g, dtype=torch.int32).to(dev), N)
go = torch.randn(B, N, C, generator=g).to(dev)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def fused(xx, nb, leaky):
    def f():
        WeightNetKSumFn.apply(h, wl, bl, xx, nb, leaky).backward(go)
    return f


def unfused(xx, nb, leaky):
    def f():
        w = linear(h, wl, bl, act=1, preact_grad=True)
        WeightedKSumFn.apply(w, xx, nb, leaky, True).backward(go)
    return f


def fwd_only(fn):
    def f():
        with torch.no_grad():
            fn()
    return f


print("dense  fwd+bwd: fused %.0f us, materialised %.0f us" % (timeit(fused(x, None, True)), timeit(unfused(x, None, True))))
print("gather fwd+bwd: fused %.0f us, materialised %.0f us" % (timeit(fused(p, nbr, False)), timeit(unfused(p, nbr, False))))
print("dense  fwd    : fused %.0f us, materialised %.0f us" % (
    timeit(fwd_only(lambda: WeightNetKSumFn.apply(h, wl, bl, x, None, True))),
    timeit(fwd_only(lambda: WeightedKSumFn.apply(linear(h, wl, bl, act=1), x, None, True)))))
print("gather fwd    : fused %.0f us, materialised %.0f us" % (
    timeit(fwd_only(lambda: WeightNetKSumFn.apply(h, wl, bl, p, nbr, False))),
    timeit(fwd_only(lambda: WeightedKSumFn.apply(linear(h, wl, bl, act=1), p, nbr, False)))))
