"""Per-workgroup timeline of one launch of the persistent GEMM (cmf_gemm_trace_arm / _read): start, first tile done, drain
start, end per workgroup and the CU it ran on -- do the two workgroups of a CU advance at the same pace?

    python tools/pgemm_timeline.py [MxNxK] [mode] [dxyz]
"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmflow_amd import _lib
from cmflow_amd.fused import gemm
dev = torch.device("cuda:0")
M, N, K = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "524288x512x256").split("x"))
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dxyz = int(sys.argv[3]) if len(sys.argv) > 3 else 1
torch.manual_seed(0)
dZ = torch.randn(M, K, device=dev); W = torch.randn(K, N, device=dev); Zs = torch.randn(M, N, device=dev)
ea, ec, em, ei = (torch.rand(N, device=dev) + 0.5 for _ in range(4))
d4 = torch.randn(M, 4, device=dev)
bwd = (mode, Zs, ea, ec, em, ei) if mode == 1 else (mode, Zs, None, None, None, None)
if dxyz:
    bwd = bwd + (d4,)
out = torch.empty(M, N, device=dev)
L = _lib.lib()
L.cmf_gemm_persist_config(2, 0)
fn = lambda: gemm(dZ, W, b_t=False, bwd=bwd, stats=(mode == 1 or bool(dxyz)), out=out)
for _ in range(3):
    fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    fn()
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 10 * 1e-3
print("M=%d N=%d K=%d mode %d dxyz %d: %.1f us, %.1f TF (untraced, back to back)" % (M, N, K, mode, dxyz, t * 1e6, 2.0 * M * N * K / t / 1e12))
L.cmf_gemm_trace_arm()
fn()
n = L.cmf_gemm_trace_read(None, 0)
buf = np.zeros((n, 8), dtype=np.uint64)
L.cmf_gemm_trace_read(buf.ctypes.data_as(ctypes.c_void_p), n)
rec = buf[buf[:, 2] > 0]
t0 = rec[:, 0].min()
start, drain, end, first = ((rec[:, i] - t0).astype(np.float64) * 0.01 for i in (0, 1, 2, 4))     # us (100 MHz clock)
tiles = rec[:, 5].astype(np.int64)
hw = rec[:, 3]
xcc = (hw >> np.uint64(32)).astype(np.int64) & 0xF
hwid = (hw & np.uint64(0xFFFFFFFF)).astype(np.int64)
cu = (hwid >> 8) & 0xF; sh = (hwid >> 12) & 0x1; se = (hwid >> 13) & 0x7
cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
print("workgroups %d, CUs seen %d, tiles per workgroup %d..%d, span %.1f us (single traced launch, cold)" % (len(rec), len(np.unique(cuid)), tiles.min(), tiles.max(), end.max()))
print("start %.1f +- %.1f us | first tile %.1f +- %.1f | per later tile %.2f +- %.2f | drain %.1f +- %.1f | end %.1f +- %.1f (min %.1f max %.1f)" % (
    start.mean(), start.std(), (first - start).mean(), (first - start).std(),
    ((drain - first) / np.maximum(tiles - 1, 1)).mean(), ((drain - first) / np.maximum(tiles - 1, 1)).std(),
    (end - drain).mean(), (end - drain).std(), end.mean(), end.std(), end.min(), end.max()))
per = np.bincount(np.unique(cuid, return_inverse=True)[1])
print("workgroups per CU: " + ", ".join("%d CUs x %d" % ((per == k).sum(), k) for k in np.unique(per)))
# pairs on one CU: difference of end times
d = []
for c in np.unique(cuid):
    e = np.sort(end[cuid == c])
    if len(e) == 2:
        d.append(e[1] - e[0])
if d:
    d = np.array(d)
    print("two workgroups on one CU: |end difference| mean %.1f us, median %.1f, max %.1f; the later one ends at %.1f us on average" % (
        d.mean(), np.median(d), d.max(), np.mean([np.max(end[cuid == c]) for c in np.unique(cuid) if (cuid == c).sum() == 2])))
q = np.percentile(end, [0, 10, 50, 90, 100])
print("end-time percentiles (us): min %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f" % tuple(q))
L.cmf_gemm_persist_config(1, 0)
