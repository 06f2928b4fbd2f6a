"""The set-conv first layer formed in the A-operand path of the next layer's GEMM (cmf_group_prep + cmf_gemm_gather_affine) against the
materialised path (cmf_group_affine + cmf_gemm with the A prologue) at the second encoder's four scales (B = 64, N = 256, 512 -> 256
channels), inference form: 10 back-to-back repetitions between one event pair, A/B/A/B."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmflow_amd import _lib, synth, fused_blocks as FB, pointnet2_utils as pu
from cmflow_amd.fused import gemm
dev = torch.device("cuda:0")
torch.manual_seed(0)
L = _lib.lib()
st = _lib.stream_ptr()
B, N, K, NO = 64, 256, 512, 256
xyz = synth.make_batch(B, seed=1234)["pc1"].to(dev).transpose(1, 2).contiguous()


def timed(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for S, r in ((32, 16.0), (16, 8.0), (8, 4.0), (4, 2.0)):
    idx = pu.ball_query(r, S, xyz, xyz)
    M = B * N * S
    y = torch.randn(B, N, 4 * K, device=dev)[:, :, :K]
    wx = torch.randn(K, 3, device=dev); pa, pc = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.3
    W = torch.randn(NO, K, device=dev)
    out = torch.empty(M, NO, device=dev)
    z = torch.empty(B, N, S, K, device=dev); dq = torch.empty(M, 4, device=dev)
    rows = torch.empty(M, dtype=torch.int32, device=dev); wx3 = torch.empty(3, K, device=dev)

    def materialised():
        _lib.check(L.cmf_group_affine(B, N, N, S, K, y.data_ptr(), y.stride(1), None, 0, xyz.data_ptr(), xyz.data_ptr(), wx.data_ptr(), 3,
                                      idx.data_ptr(), 0, z.data_ptr(), dq.data_ptr(), None, None, st), "ga")
        gemm(z.view(M, K), W, pro=(pa, pc), out=out)

    def gathered():
        _lib.check(L.cmf_group_prep(B, N, N, S, K, xyz.data_ptr(), xyz.data_ptr(), wx.data_ptr(), 3, idx.data_ptr(), rows.data_ptr(),
                                    dq.data_ptr(), wx3.data_ptr(), st), "prep")
        _lib.check(L.cmf_gemm_gather_affine(M, NO, K, y.data_ptr(), y.stride(1), rows.data_ptr(), dq.data_ptr(), wx3.data_ptr(),
                                            pa.data_ptr(), pc.data_ptr(), W.data_ptr(), K, out.data_ptr(), NO, None, st), "gg")

    def gemm_only():
        gemm(z.view(M, K), W, pro=(pa, pc), out=out)

    t = [timed(materialised), timed(gathered), timed(materialised), timed(gathered)]
    tg = timed(gemm_only)
    fl = 2.0 * M * NO * K
    print("rows %7d: materialised %.1f %.1f us (its GEMM alone %.1f us = %.1f TF)   gathered %.1f %.1f us (%.1f TF as a GEMM)" % (
        M, t[0], t[2], tg, fl / tg / 1e6, t[1], t[3], fl / min(t[1], t[3]) / 1e6), flush=True)
    del z, out, y
