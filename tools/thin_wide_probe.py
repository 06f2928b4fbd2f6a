"""cmf_thin_bwd_wide_layer (64 <- 256 channels, pooled) against the kernels it replaces in a second-encoder block
(max-pool backward, BN backward in place, weight-gradient and data-gradient GEMMs), at B = 64, N = 256."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmflow_amd import _lib, fused_blocks as FB
from cmflow_amd.fused import gemm
dev = torch.device("cuda:0"); L = _lib.lib()
p = lambda t: None if t is None else t.data_ptr()

def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

P, cout, cin = 16384, 64, 256
print("%4s %9s | %9s | %9s" % ("S", "rows", "fused us", "4-kern us"))
for S in (32, 16, 8, 4):
    M = P * S
    torch.manual_seed(0)
    z3, z2, dout = torch.randn(M, cout, device=dev), torch.randn(M, cin, device=dev), torch.randn(P, cout, device=dev)
    w = torch.randn(cout, cin, device=dev) * 0.1
    st = FB.BNState(); st.a, st.c, st.mean, st.invstd = (torch.randn(cout, device=dev) for _ in range(4)); st.invstd = st.invstd.abs() + 0.5
    st.training, st.count = True, M
    a_in, c_in, mean_in, invstd_in = (torch.randn(cin, device=dev) for _ in range(4))
    _, am = FB.bn_relu_maxpool(z3.view(P, S, cout), st)
    nslab = L.cmf_thin_bwd_wide_slabs(M, cin, None)
    slabs = torch.empty(nslab, cout, cin, device=dev)
    dx = torch.empty(M, cin, device=dev); stats = torch.empty(M // 128, 2, cin, device=dev); dw = torch.zeros(cout, cin, device=dev)
    gp = torch.empty(P, cout, device=dev); part = torch.empty(P // 128, 2, cout, device=dev)
    sums = torch.randn(2, cout, device=dev)

    def fused():
        _lib.check(L.cmf_maxpool_bwd_point(P, S, cout, p(dout), cout, p(z3), p(st.a), p(st.c), p(st.mean), p(st.invstd), p(am), p(gp), p(part),
                                           _lib.stream_ptr()), "mp")
        _lib.check(L.cmf_thin_bwd_wide_layer(M, cin, None, cout, p(gp), p(am), S, p(z3), cout, p(st.a), p(st.mean), p(st.invstd), p(sums),
                                             p(w), cin, p(z2), cin, p(a_in), p(c_in), p(mean_in), p(invstd_in), p(dx), cin, p(stats), p(dw), cin, 1,
                                             p(slabs), _lib.stream_ptr()), "wide")

    def four():
        dU, _ = FB.maxpool_bwd(dout, z3.view(P, S, cout), st, am)
        _lib.check(L.cmf_bn_bwd_apply(M, cout, p(dU), p(z3), cout, p(st.a), p(st.mean), p(st.invstd), p(sums), _lib.stream_ptr()), "apply")
        FB.gemm_dw(dU, z2, prob=(a_in, c_in))
        gemm(dU, w, b_t=False, bwd=(1, z2, a_in, c_in, mean_in, invstd_in))

    print("%4d %9d | %9.1f | %9.1f" % (S, M, timeit(fused), timeit(four)))
