"""Do light streaming kernels run NEXT to a chip-filling cmf_gemm launch?  A forward GEMM (151 registers: three workgroups per CU
leave 56 registers per lane and 11 KB of LDS) on one stream, cmf_bn_bwd_apply (37 registers, no LDS) or the data-gradient GEMM
form (157 -> 160 registers: 32 free) on another: alone, and both at once."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmflow_amd import _lib
from cmflow_amd.fused import gemm
dev = torch.device("cuda:0")
L = _lib.lib()
M = 524288
A = torch.randn(M, 512, device=dev); W = torch.randn(256, 512, device=dev); out = torch.empty(M, 256, device=dev)
dZ = torch.randn(M, 256, device=dev); out2 = torch.empty(M, 512, device=dev)
dU = torch.randn(M, 256, device=dev); z = torch.randn(M, 256, device=dev)
a, mean, invstd = (torch.rand(256, device=dev) + 0.5 for _ in range(3))
sums = torch.randn(2, 256, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def g_fwd(): gemm(A, W, out=out)
def g_dx(): gemm(dZ, W, b_t=False, out=out2)
def apply_(): _lib.check(L.cmf_bn_bwd_apply(M, 256, dU.data_ptr(), z.data_ptr(), 256, a.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                                            sums.data_ptr(), _lib.stream_ptr()), "apply")


def timed(fa, fb, n=10):
    for _ in range(2):
        for f, s in ((fa, s1), (fb, s2)):
            if f:
                with torch.cuda.stream(s): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    s1.wait_stream(torch.cuda.current_stream()); s2.wait_stream(torch.cuda.current_stream())
    for _ in range(n):
        if fa:
            with torch.cuda.stream(s1): fa()
        if fb:
            with torch.cuda.stream(s2): fb()
    torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for _ in range(3): timed(g_fwd, apply_)                       # clocks up
for name, ga in (("fwd GEMM (151 regs)", g_fwd), ("dX GEMM (157 regs)", g_dx), ("fwd GEMM (151 regs)", g_fwd), ("dX GEMM (157 regs)", g_dx)):
    ta, tb, tab = timed(ga, None), timed(None, apply_), timed(ga, apply_)
    print("%-22s alone %.0f us | bn_bwd_apply alone %.0f us | both %.0f us (sum %.0f, max %.0f)" % (name, ta, tb, tab, ta + tb, max(ta, tb)))
