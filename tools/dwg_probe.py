"""Weight gradient with the gathered B operand (cmf_gemm_dw_gather) at the second encoder's four scales: tile (env CMF_GEMM_DWG_TALL) x split count."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmflow_amd import _lib, synth, pointnet2_utils as pu
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


ref = {}
for S, r in ((32, 16.0), (16, 8.0), (8, 4.0), (4, 2.0)):
    idx = pu.ball_query(r, S, xyz, xyz)
    M = B * N * S
    y = torch.randn(B, N, 4 * K, device=dev)[:, :, :K]
    wx = torch.randn(K, 3, device=dev); pa, pc = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.3
    dZ = torch.randn(M, NO, device=dev)
    rows = torch.empty(M, dtype=torch.int32, device=dev); dq = torch.empty(M, 4, device=dev); wx3 = torch.empty(3, K, device=dev)
    _lib.check(L.cmf_group_prep(B, N, N, S, K, xyz.data_ptr(), xyz.data_ptr(), wx.data_ptr(), 3, idx.data_ptr(), rows.data_ptr(),
                                dq.data_ptr(), wx3.data_ptr(), st), "prep")
    res = []
    for sk in (64, 96, 128, 192, 256):
        if M // 16 // sk < 8:
            continue
        ws = torch.empty(sk * NO * K, device=dev); out = torch.zeros(NO, K, device=dev)
        f = lambda: _lib.check(L.cmf_gemm_dw_gather(NO, K, M, dZ.data_ptr(), NO, y.data_ptr(), y.stride(1), rows.data_ptr(), dq.data_ptr(), wx3.data_ptr(),
                                                    pa.data_ptr(), pc.data_ptr(), out.data_ptr(), K, sk, ws.data_ptr(), 0, st), "dwg")
        res.append("split %3d: %6.1f" % (sk, rate(f, 2.0 * M * NO * K)))
        torch.cuda.synchronize()
        print("  checksum rows %d split %d: %.6e" % (M, sk, float(out.double().abs().sum())))
    print("rows %7d  " % M + "   ".join(res), flush=True)
