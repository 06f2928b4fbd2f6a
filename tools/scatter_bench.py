"""cmf_group_rows_grad_bn (reads dU and z) vs cmf_group_rows_grad_bn_cf (closed form, reads dU only) at the second
encoder's shapes: B=64, N=256, C=512, S in 4/8/16/32 with the matching ball radii."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmflow_amd import _lib, synth
from cmflow_amd.fused import Neighbors
from cmflow_amd.pointnet2_utils import ball_query
dev = torch.device("cuda:0")
L = _lib.lib()
p = lambda t: _lib.dev_ptr(t, t.dtype)
B, N, C = 64, 256, 512
xyz = synth.make_batch(B, seed=1)["pc1"].to(dev).transpose(1, 2).contiguous()
g = torch.Generator().manual_seed(0)
y = torch.randn(B, N, C, generator=g).to(dev)
wx = torch.randn(C, 3, generator=g).to(dev)
a, mean, invstd = (torch.rand(C, generator=g).to(dev) + 0.5 for _ in range(3))
sums = torch.randn(2, C, generator=g).to(dev)


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


for r, S in ((2.0, 4), (4.0, 8), (8.0, 16), (16.0, 32)):
    idx = ball_query(r, S, xyz, xyz)
    off, inv = Neighbors(idx.int(), N).inverse()
    M = B * N * S
    dU = torch.randn(M, C, generator=g).to(dev)
    z = torch.randn(M, C, generator=g).to(dev)
    out = torch.empty(B, N, C, device=dev)
    st = _lib.stream_ptr()
    tz = timeit(lambda: L.cmf_group_rows_grad_bn(B, N, C, N * S, p(dU), p(z), p(a), p(mean), p(invstd), p(sums), 1.0 / M,
                                                 p(off), p(inv), p(out), C, st))
    tc = timeit(lambda: L.cmf_group_rows_grad_bn_cf(B, N, C, N * S, S, p(dU), p(y), C, p(wx), 3, p(xyz), p(xyz), p(a), p(mean),
                                                    p(invstd), p(sums), 1.0 / M, p(off), p(inv), p(out), C, st))
    gb = M * C * 4 / 1e9
    print("S=%2d rows %7d: z-reading %6.1f us (%.2f TB/s over dU+z), closed form %6.1f us (%.2f TB/s over dU)"
          % (S, M, tz, 2 * gb / tz * 1e3, tc, gb / tc * 1e3))
