"""The second encoder's data gradient summed per source point (cmf_group_perm + cmf_gemm_dx_gather_sum) at its four scales: TF as a
GEMM (524288 .. 65536 rows x 512 x 256), beside the gathered-but-stored form (cmf_gemm_dx_gather) and the plain masked data gradient
(cmf_gemm, bwd_mode 1, on a materialised tensor).  With a diagnostics build (CMF_LIB=tools/diag/libcmflow_x.so CMF_GEMM_DIAG_RT=8: no
epilogue at all, results invalid) the same calls time the main loops alone."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmflow_amd import _lib, synth, pointnet2_utils as pu
from cmflow_amd.fused import Neighbors, gemm
dev = torch.device("cuda:0")
L = _lib.lib(); st = _lib.stream_ptr()
B, N, K, NO = 64, 256, 512, 256
xyz = synth.make_batch(B, seed=1234)["pc1"].to(dev).transpose(1, 2).contiguous()


def rate(fn, flops, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return flops / (e0.elapsed_time(e1) / n * 1e-3) / 1e12


for S, r in ((32, 16.0), (16, 8.0), (8, 4.0), (4, 2.0)):
    idx = pu.ball_query(r, S, xyz, xyz)
    off, inv = Neighbors(idx, N).inverse()
    M, P, E = B * N * S, B * N, N * S
    y = torch.randn(B, N, 4 * K, device=dev)[:, :, :K]
    wx = torch.randn(K, 3, device=dev)
    ea, ec, em, ei = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.3, torch.randn(K, device=dev), torch.rand(K, device=dev) + 0.5
    dZ = torch.randn(M, NO, device=dev); W = torch.randn(NO, K, device=dev)
    rows = torch.empty(M, dtype=torch.int32, device=dev); dq = torch.empty(M, 4, device=dev); wx3 = torch.empty(3, K, device=dev)
    _lib.check(L.cmf_group_prep(B, N, N, S, K, xyz.data_ptr(), xyz.data_ptr(), wx.data_ptr(), 3, idx.data_ptr(), rows.data_ptr(),
                                dq.data_ptr(), wx3.data_ptr(), st), "prep")
    perm = torch.empty(M, dtype=torch.int32, device=dev); pts = torch.empty(M, dtype=torch.int32, device=dev); dq2 = torch.empty(M, 4, device=dev)
    _lib.check(L.cmf_group_perm(B, E, inv.data_ptr(), rows.data_ptr(), dq.data_ptr(), perm.data_ptr(), pts.data_ptr(), dq2.data_ptr(), st), "perm")
    tiles = M // 128
    stats = torch.empty(tiles, 5, K, device=dev)
    pieces = torch.empty(P + M // 64, K, device=dev)
    dU = torch.empty(M, K, device=dev)
    f = 2.0 * M * K * NO
    r_sum = rate(lambda: _lib.check(L.cmf_gemm_dx_gather_sum(M, K, NO, dZ.data_ptr(), NO, W.data_ptr(), K, y.data_ptr(), y.stride(1), perm.data_ptr(),
                                                             pts.data_ptr(), dq2.data_ptr(), wx3.data_ptr(), ea.data_ptr(), ec.data_ptr(), em.data_ptr(),
                                                             ei.data_ptr(), pieces.data_ptr(), stats.data_ptr(), st), "dxs"), f)
    r_gat = rate(lambda: _lib.check(L.cmf_gemm_dx_gather(M, K, NO, dZ.data_ptr(), NO, W.data_ptr(), K, dU.data_ptr(), K, y.data_ptr(), y.stride(1),
                                                         rows.data_ptr(), dq.data_ptr(), wx3.data_ptr(), ea.data_ptr(), ec.data_ptr(), em.data_ptr(),
                                                         ei.data_ptr(), stats.data_ptr(), st), "dxg"), f)
    ident = torch.arange(M, dtype=torch.int32, device=dev)          # timing only: A rows in storage order (what a pre-permuted dZ would give)
    r_seq = rate(lambda: _lib.check(L.cmf_gemm_dx_gather_sum(M, K, NO, dZ.data_ptr(), NO, W.data_ptr(), K, y.data_ptr(), y.stride(1), ident.data_ptr(),
                                                             pts.data_ptr(), dq2.data_ptr(), wx3.data_ptr(), ea.data_ptr(), ec.data_ptr(), em.data_ptr(),
                                                             ei.data_ptr(), pieces.data_ptr(), stats.data_ptr(), st), "dxs"), f)
    print("   (summed with its A rows in storage order: %.1f TF)" % r_seq)
    L.cmf_gemm_persist_config(0, 0)
    r_pln = rate(lambda: gemm(dZ, W, b_t=False, out=dU, bwd=(1, dU, ea, ec, em, ei, dq)), f)
    L.cmf_gemm_persist_config(1, 0)
    r_raw = rate(lambda: gemm(dZ, W, b_t=False, out=dU), f)
    if os.environ.get("DXSUM_TIMELINE") == "1":
        import numpy as np, ctypes
        for name, fn in (("summed", lambda: L.cmf_gemm_dx_gather_sum(M, K, NO, dZ.data_ptr(), NO, W.data_ptr(), K, y.data_ptr(), y.stride(1), perm.data_ptr(),
                                                                    pts.data_ptr(), dq2.data_ptr(), wx3.data_ptr(), ea.data_ptr(), ec.data_ptr(), em.data_ptr(),
                                                                    ei.data_ptr(), pieces.data_ptr(), stats.data_ptr(), st)),
                         ("gathered + stored", lambda: L.cmf_gemm_dx_gather(M, K, NO, dZ.data_ptr(), NO, W.data_ptr(), K, dU.data_ptr(), K, y.data_ptr(), y.stride(1),
                                                                            rows.data_ptr(), dq.data_ptr(), wx3.data_ptr(), ea.data_ptr(), ec.data_ptr(), em.data_ptr(),
                                                                            ei.data_ptr(), stats.data_ptr(), st))):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            L.cmf_gemm_trace_arm()
            fn(); torch.cuda.synchronize()
            n = L.cmf_gemm_trace_read(None, 0)
            buf = np.zeros((n, 8), dtype=np.uint64)
            L.cmf_gemm_trace_read(buf.ctypes.data_as(ctypes.c_void_p), n)
            rec = buf[buf[:, 2] > 0]
            t0 = rec[:, 0].min()
            start, main, end = ((rec[:, i] - t0).astype(np.float64) * 0.01 for i in range(3))
            hw = rec[:, 3]
            xcc = (hw >> np.uint64(32)).astype(np.int64) & 0xF
            hwid = (hw & np.uint64(0xFFFFFFFF)).astype(np.int64)
            cuid = ((xcc * 8 + ((hwid >> 13) & 0x7)) * 2 + ((hwid >> 12) & 0x1)) * 16 + ((hwid >> 8) & 0xF)
            hist = np.zeros(8); tot = 0.0
            for c in np.unique(cuid):
                m = cuid == c
                ev = sorted([(s_, 1) for s_ in start[m]] + [(e_, -1) for e_ in main[m]])
                lo, hi = start[m].min(), end[m].max()
                cur, last = 0, lo
                for tt, d in ev:
                    hist[min(cur, 7)] += tt - last
                    cur += d; last = tt
                hist[min(cur, 7)] += hi - last
                tot += hi - lo
            if (rec[:, 4] > 0).all():
                ep = [(rec[:, i] - t0).astype(np.float64) * 0.01 for i in (4, 5, 6)]
                print("  %-18s epilogue stages (first wave, us): constants + indices + dxyz rows landed %.1f | first block row walked %.1f | second block row %.1f | statistics + waiting for the other waves %.1f"
                      % (name, (ep[0] - main).mean(), (ep[1] - ep[0]).mean(), (ep[2] - ep[1]).mean(), (end - ep[2]).mean()))
            print("  %-18s rows %7d: per workgroup total %.1f us | main loop %.1f +- %.1f | epilogue %.1f +- %.1f (max %.1f) | span %.0f us | k in main loop: %s"
                  % (name, M, (end - start).mean(), (main - start).mean(), (main - start).std(), (end - main).mean(), (end - main).std(), (end - main).max(),
                     end.max(), " ".join("%d:%.0f%%" % (k, 100 * hist[k] / tot) for k in range(4))), flush=True)
    print("rows %7d   summed %6.1f   gathered + stored %6.1f   materialised, masked (tiled) %6.1f   plain store %6.1f TF" % (M, r_sum, r_gat, r_pln, r_raw), flush=True)
    del dU, pieces, dZ
