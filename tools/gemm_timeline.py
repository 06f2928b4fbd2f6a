"""Per-workgroup timeline of one cmf_gemm launch (cmf_gemm_trace_arm / _read): when does each workgroup start, leave its
main loop and finish, and on which CU -- i.e. how much of a CU's time has NO workgroup in its MFMA main loop.

    python tools/gemm_timeline.py [fwd|dx|dxq|dw|plain] [M N K]
"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmflow_amd import _lib
from cmflow_amd.fused import gemm
from cmflow_amd.fused_blocks import gemm_dw

dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "fwd"
M, N, K = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (524288, 256, 512)
torch.manual_seed(0)
A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev); dZ = torch.randn(M, N, device=dev)
Zs = torch.randn(M, K, device=dev)
ea, ec, em, ei = (torch.rand(K, device=dev) + 0.5 for _ in range(4))
dxyz = torch.randn(M, 4, device=dev)
fn = {"fwd": lambda: gemm(A, W, stats=True), "dx": lambda: gemm(dZ, W, b_t=False, bwd=(1, Zs, ea, ec, em, ei)),
      "dxq": lambda: gemm(dZ, W, b_t=False, bwd=(1, Zs, ea, ec, em, ei, dxyz)),
      "dw": lambda: gemm_dw(dZ, A), "plain": lambda: gemm(A, W)}[which]
for _ in range(3):
    fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    fn()
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 10 * 1e-3
print("%s M=%d N=%d K=%d: %.1f us, %.1f TF (untraced)" % (which, M, N, K, t * 1e6, 2.0 * M * N * K / t / 1e12))

L = _lib.lib()
L.cmf_gemm_trace_arm()
fn()
n = L.cmf_gemm_trace_read(None, 0)
buf = np.zeros((n, 8), dtype=np.uint64)
L.cmf_gemm_trace_read(buf.ctypes.data_as(ctypes.c_void_p), n)
live = buf[:, 2] > 0
rec = buf[live]
t0 = rec[:, 0].min()
start, main, end = ((rec[:, i] - t0).astype(np.float64) * 0.01 for i in range(3))      # us (100 MHz clock)
ep = [(rec[:, i] - t0).astype(np.float64) * 0.01 for i in (4, 5, 6)]
if (rec[:, 4] > 0).all():
    print("epilogue stages (us, mean): transposition of band 0 visible %.2f | band 0 computed + stored %.2f | remaining bands %.2f | statistics + exit %.2f"
          % ((ep[0] - main).mean(), (ep[1] - ep[0]).mean(), (ep[2] - ep[1]).mean(), (end - ep[2]).mean()))
hw = rec[:, 3]
xcc = (hw >> np.uint64(32)).astype(np.int64) & 0xF
hwid = (hw & np.uint64(0xFFFFFFFF)).astype(np.int64)
cu = (hwid >> 8) & 0xF; sh = (hwid >> 12) & 0x1; se = (hwid >> 13) & 0x7
cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
print("workgroups %d (launched %d), CUs seen %d, span %.1f us" % (live.sum(), n, len(np.unique(cuid)), end.max()))
print("per workgroup: total %.1f +- %.1f us | main loop %.1f +- %.1f | epilogue %.1f +- %.1f (max %.1f)"
      % ((end - start).mean(), (end - start).std(), (main - start).mean(), (main - start).std(), (end - main).mean(),
         (end - main).std(), (end - main).max()))
# per CU: time with k workgroups inside their main loop
hist = np.zeros(8)
span_total = 0.0
for c in np.unique(cuid):
    m = cuid == c
    ev = sorted([(s, 1) for s in start[m]] + [(e, -1) for e in main[m]])
    lo, hi = start[m].min(), end[m].max()
    cur, last = 0, lo
    for tt, d in ev:
        hist[min(cur, 7)] += tt - last
        cur += d; last = tt
    hist[min(cur, 7)] += hi - last
    span_total += hi - lo
print("share of CU time with k workgroups in the main loop: " + "  ".join("k=%d %.1f%%" % (k, 100 * hist[k] / span_total) for k in range(5)))
# how synchronised are the epilogues?  histogram of main-loop end times in 5 us bins over the launch
bins = np.arange(0, end.max() + 5, 5.0)
h, _ = np.histogram(main, bins)
print("workgroups leaving the main loop per 5 us bin (first 60 bins):", h[:60].tolist())
first = np.sort(start)
print("start times: first %.1f, 768th %.1f, last %.1f us" % (first[0], first[min(767, len(first) - 1)], first[-1]))
