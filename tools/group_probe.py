"""Micro-probe for the drop-in grouping path (used under rocprofv3 --pmc): ball_query + group_points + group_points_grad
at the two op-level configurations of SURVEY 8d, 3 launches each."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmflow_amd import synth
from cmflow_amd.pointnet2_utils import ball_query, grouping_operation, pointnet2_cuda as ext
dev = torch.device("cuda:0")
for (B, N, K, C, lidar, r) in ((64, 256, 32, 1027, False, 2.0), (32, 4096, 64, 128, True, 2.0)):
    b = synth.make_batch(B, N, seed=1, lidar=lidar)
    xyz = b["pc1"].permute(0, 2, 1).contiguous().to(dev)
    feat = torch.randn(B, C, N, device=dev)
    for _ in range(3):
        idx = ball_query(r, K, xyz, xyz)
        out = grouping_operation(feat, idx)
        gp = torch.zeros(B, C, N, device=dev)
        ext.group_points_grad_wrapper(B, C, N, N, K, out, idx, gp)
    torch.cuda.synchronize()
    print("algorithmic bytes: group %.1f MB, group_grad %.1f MB" % ((B * C * N + B * N * K + B * C * N * K) * 4 / 1e6,
                                                                  (B * C * N * K + B * N * K + B * C * N) * 4 / 1e6))
