"""The two forms of the statistics fold kernels (4 columns per 256-thread workgroup / 16 columns per 1024-thread workgroup) over
partial-matrix sizes.  CMF_FIN_WIDE=0|1 forces the form only in an experiment build (make GROUP_DEFS=-DCMF_FIN_EXPERIMENT
OUT=../../tools/diag/libcmflow_fin.so, CMF_LIB=...); the product dispatches by size (pointwise.hip fin_wide)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmflow_amd import _lib
dev = torch.device("cuda:0"); L = _lib.lib(); st = _lib.stream_ptr()
def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for tiles in (4096, 2048, 512):
    for ncols in (64, 128, 160, 256, 320, 640, 1280, 2560):
        part = torch.randn(tiles, ncols, device=dev); out = torch.empty(ncols, device=dev)
        r = []
        for w in ("0", "1"):
            os.environ["CMF_FIN_WIDE"] = w
            r.append(timed(lambda: _lib.check(L.cmf_colsum(tiles, ncols, part.data_ptr(), out.data_ptr(), 0, None, None, st), "cs")))
        C = ncols // 2
        g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev); rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        o = [torch.empty(C, device=dev) for _ in range(4)]
        for w in ("0", "1"):
            os.environ["CMF_FIN_WIDE"] = w
            r.append(timed(lambda: _lib.check(L.cmf_bn_finalize(tiles, C, float(tiles * 128), part.data_ptr(), g.data_ptr(), b.data_ptr(), 1e-5, 0.1,
                     rm.data_ptr(), rv.data_ptr(), o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), None, st), "bn")))
        print("tiles %5d ncols %5d: colsum 4-col %6.1f us  16-col %6.1f us | bn_finalize (C = %4d) 4-col %6.1f us  16-col %6.1f us" % (tiles, ncols, r[0], r[1], C, r[2], r[3]), flush=True)
