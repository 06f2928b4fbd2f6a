"""Microbenchmark of cmf_thin_bwd_layer (one fused backward layer of a narrow conv + BN + ReLU stack) against the three
kernels it replaces, at the first encoder's shapes (B = 64, N = 256; neighbour rows M = 16384 * S).

    python tools/thin_bwd_probe.py
"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmflow_amd import _lib
from cmflow_amd.fused import gemm
from cmflow_amd.fused_blocks import gemm_dw

dev = torch.device("cuda:0")
L = _lib.lib()
p = lambda t: None if t is None else t.data_ptr()


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3                       # us


print("%9s %4s %4s %5s | %9s %7s | %9s %7s" % ("rows", "cout", "cin", "mode", "fused us", "GB/s", "3-kern us", "GB/s"))
for rows, cout, cin, in_mode, dq in [(524288, 64, 32, 1, 0), (262144, 64, 32, 1, 0), (131072, 64, 32, 1, 0), (65536, 64, 32, 1, 0),
                                     (524288, 32, 32, 1, 1), (262144, 32, 32, 1, 1), (131072, 32, 32, 1, 1), (65536, 32, 32, 1, 1),
                                     (16384, 64, 64, 1, 0), (16384, 64, 64, 0, 0)]:
    torch.manual_seed(0)
    dU, z, x = torch.randn(rows, cout, device=dev), torch.randn(rows, cout, device=dev), torch.randn(rows, cin, device=dev)
    w = torch.randn(cout, cin, device=dev) * 0.2
    a, mean, invstd = torch.randn(cout, device=dev), torch.randn(cout, device=dev), torch.rand(cout, device=dev) + 0.5
    sums = torch.randn(2, cout, device=dev)
    a_in, c_in, mean_in, invstd_in = (torch.randn(cin, device=dev) for _ in range(4))
    dxyz = torch.randn(rows, 4, device=dev) if dq else None
    tpw = ctypes.c_int()
    nslab = L.cmf_thin_bwd_slabs(rows, ctypes.addressof(tpw))
    tiles = (rows + 127) // 128
    dx = torch.empty(rows, cin, device=dev)
    stats = torch.empty(tiles, 5, cin, device=dev)
    dw = torch.zeros(cout, cin, device=dev)
    slabs = torch.empty(max(nslab, 2), cout, cin, device=dev)
    st = _lib.stream_ptr()

    def fused():
        _lib.check(L.cmf_thin_bwd_layer(rows, cout, cin, p(dU), cout, p(z), cout, p(a), p(mean), p(invstd), p(sums), p(w), cin, p(x), cin,
                                        in_mode, p(a_in), p(c_in), p(mean_in), p(invstd_in), p(dxyz), p(dx), cin, p(stats), p(dw), cin, 1,
                                        p(slabs), st), "fused")

    dZ = dU.clone()

    def three():
        _lib.check(L.cmf_bn_bwd_apply(rows, cout, p(dZ), p(z), cout, p(a), p(mean), p(invstd), p(sums), st), "apply")
        gemm_dw(dZ, x, prob=(a_in, c_in) if in_mode else None)
        if in_mode:
            gemm(dZ, w, b_t=False, bwd=(1, x, a_in, c_in, mean_in, invstd_in) + ((dxyz,) if dq else ()))
        else:
            gemm(dZ, w, b_t=False)

    tf, t3 = timeit(fused), timeit(three)
    bf = 4.0 * rows * (2 * cout + 2 * cin) + (16.0 * rows if dq else 0)
    b3 = 4.0 * rows * (3 * cout + (cout + cin) + (cout + 2 * cin)) + (16.0 * rows if dq else 0)
    print("%9d %4d %4d %5s | %9.1f %7.0f | %9.1f %7.0f" % (rows, cout, cin, "%d%s" % (in_mode, "q" if dq else ""), tf, bf / tf / 1e3, t3, b3 / t3 / 1e3))
