"""cmf_group_points_grad at BASELINE config 5's shape (32, 4096, 64, C) with ball-query indices of the LiDAR-like synthetic
cloud: time per call.  Usage: [CMF_GROUP_GRAD_WAVE=0|1] [CMF_GW_DIAG=bits] python tools/scatter_probe.py [C]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmflow_amd import _lib, synth
dev = torch.device("cuda:0")
C = int(sys.argv[1]) if len(sys.argv) > 1 else 64
B, N, K, r = 32, 4096, 64, 2.0
L = _lib.lib(); st = _lib.stream_ptr()
xyz_t = synth.make_batch(B, N=N, seed=1234, lidar=True)["pc1"].to(dev).transpose(1, 2).contiguous()
idx = torch.zeros(B, N, K, dtype=torch.int32, device=dev)
_lib.check(L.cmf_ball_query(B, N, N, r, K, xyz_t.data_ptr(), xyz_t.data_ptr(), idx.data_ptr(), st), "bq")
go = torch.randn(B, C, N, K, device=dev); gp = torch.zeros(B, C, N, device=dev)
fn = lambda: _lib.check(L.cmf_group_points_grad(B, C, N, N, K, go.data_ptr(), idx.data_ptr(), gp.data_ptr(), st), "gg")
for _ in range(3): fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): fn()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 10 * 1e3
nbytes = 4 * (B * C * N * K + B * N * K + B * C * N)
print("csr=%s wave=%s diag=%s C=%d: %.1f us, %.0f GB/s = %.3f of 8 TB/s" % (os.environ.get("CMF_GROUP_GRAD_CSR", "1"), os.environ.get("CMF_GROUP_GRAD_WAVE", "0"), os.environ.get("CMF_GW_DIAG", "0"), C, us,
                                                              nbytes / us / 1e3, nbytes / us / 1e3 / 8000))
