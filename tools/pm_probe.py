"""Probe for the HBM-bound kernels of the point-major path (used under rocprofv3 --kernel-trace / --pmc): each kernel is
launched 3 times in a fixed order at the shapes of the second encoder's largest scale (r = 16 m, 32 slots: M = 524288
neighbour rows, P = 16384 points at B = 64).  Prints, as JSON for tools/pm_table.py, the launch order and the algorithmic
bytes per launch (every tensor touched once)."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmflow_amd import _lib, synth, fused_blocks as FB, pointnet2_utils as pu
from cmflow_amd.fused import Neighbors
dev = torch.device("cuda:0")
torch.manual_seed(0)
L = _lib.lib()
B, N, S = 64, 256, 32
M, P = B * N * S, B * N
xyz = synth.make_batch(B, seed=1234)["pc1"].to(dev).transpose(1, 2).contiguous()
idx = pu.ball_query(16.0, S, xyz, xyz)
nbr = Neighbors(idx, N)
off, inv = nbr.inverse()
st = FB.BNState()
def bnstate(C):
    s = FB.BNState(); buf = torch.rand(4, C, device=dev) + 0.5
    s.mean, s.invstd, s.a, s.c = buf[0], buf[1], buf[2], buf[3]; s.training = True; s.count = M
    return s
seq = []
def rep(name, nbytes, fn):
    for _ in range(3):
        fn()
    seq.append((name, nbytes))
y = torch.randn(B, N, 512, device=dev); wx = torch.randn(512, 3, device=dev)
rep("group_affine_kernel", 4 * M * 512 + 16 * M + 4 * M + 4 * P * 512,
    lambda: FB.group_affine(y, None, xyz, xyz, wx, idx, act=0, stats=True, extra=True))
z3 = torch.randn(P, S, 64, device=dev); s64 = bnstate(64)
rep("bn_relu_maxpool_kernel", 4 * M * 64 + 4 * P * 64 + P * 64, lambda: FB.bn_relu_maxpool(z3, s64))
x, am = FB.bn_relu_maxpool(z3, s64); dx = torch.randn(P, 64, device=dev)
rep("maxpool_bwd_kernel", 8 * M * 64 + 4 * P * 64 + P * 64, lambda: FB.maxpool_bwd(dx, z3, s64, am))
for C in (256, 64):
    dU = torch.randn(M, C, device=dev); z = torch.randn(M, C, device=dev); s = bnstate(C)
    sums = torch.randn(2, C, device=dev)
    rep("bn_bwd_apply_kernel", 12 * M * C, lambda: _lib.check(L.cmf_bn_bwd_apply(M, C, dU.data_ptr(), z.data_ptr(), C, s.a.data_ptr(),
        s.mean.data_ptr(), s.invstd.data_ptr(), sums.data_ptr(), _lib.stream_ptr()), "bn_bwd_apply"))
    del dU, z
dU1 = torch.randn(M, 512, device=dev); s512 = bnstate(512); sums5 = torch.randn(5, 512, device=dev)
dy = torch.empty(B, N, 512, device=dev)
rep("group_rows_grad_bn_cf_kernel", 4 * M * 512 + 4 * M + 8 * P * 512,
    lambda: _lib.check(L.cmf_group_rows_grad_bn_cf(B, N, 512, N * S, S, dU1.data_ptr(), y.data_ptr(), 512, wx.data_ptr(), 3, xyz.data_ptr(),
        xyz.data_ptr(), s512.a.data_ptr(), s512.mean.data_ptr(), s512.invstd.data_ptr(), sums5.data_ptr(), 1.0 / M, off.data_ptr(),
        inv.data_ptr(), dy.data_ptr(), 512, _lib.stream_ptr()), "group_rows_grad_bn_cf"))
torch.cuda.synchronize()
print("SEQ " + json.dumps(seq))
