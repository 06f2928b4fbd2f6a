"""Test helper (run as a child process: the library reads its environment switches once): the drop-in neighbour-search / grouping calls at
a radar-sized and a LiDAR-sized shape through the C-ABI wrappers of cmflow_amd.pointnet2_utils -- ball-query indices, grouped tensors and
the grouping gradient are saved to the given file (tests/test_gpu_ops.py compares runs under different A/B switches)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from cmflow_amd import pointnet2_utils as pu, synth

out = sys.argv[1]
dev = torch.device("cuda:0")
res = {}
for tag, (B, N, K, r, C, lidar) in {"radar": (4, 256, 32, 2.0, 16, False), "lidar": (2, 4096, 64, 2.0, 8, True)}.items():
    xyz = synth.make_batch(B, N=N, seed=99, lidar=lidar)["pc1"].to(dev)               # (B,3,N)
    xyz_t = xyz.transpose(1, 2).contiguous()
    idx = pu.ball_query(r, K, xyz_t, xyz_t)
    g = torch.Generator().manual_seed(5)
    feats = torch.randn(B, C, N, generator=g).to(dev).requires_grad_(True)
    grouped = pu.grouping_operation(feats, idx)
    w = torch.randn(grouped.shape, generator=g).to(dev)
    (grouped * w).sum().backward()
    qg = pu.QueryAndGroup(r, K)(xyz_t, xyz_t, feats.detach())
    res[tag + ".idx"], res[tag + ".grouped"], res[tag + ".grad"], res[tag + ".qg"] = idx.cpu(), grouped.detach().cpu(), feats.grad.cpu(), qg.cpu()
torch.cuda.synchronize()
torch.save(res, out)
