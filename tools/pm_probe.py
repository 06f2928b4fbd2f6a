"""Probe for the HBM-bound kernels of the point-major path (used under rocprofv3 --kernel-trace / --pmc): the second
encoder's largest scale (r = 16 m, 32 slots: 524288 neighbour rows at B = 64) forward + backward through the Python-
sequenced block (same kernels as the C-ABI block call), 3 repetitions.  Prints the algorithmic bytes per launch of each
kernel (every tensor touched once) as JSON for tools/pm_table.py."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from cmflow_amd import synth, fused_blocks as FB
from cmflow_amd.cmflow import CMFlow
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = CMFlow(bench.Args()); net.load_state_dict(bench.load_weights("cmflow")); net = net.to(dev).train()
sa = net.mse_layer2.ms_ls[3]                       # r = 16, nsample = 32; channels 1030 -> 512 -> 256 -> 64 | 64 -> 64 -> 64
B, N, S = 64, 256, sa.nsample
xyz = synth.make_batch(B, seed=1234)["pc1"].to(dev).transpose(1, 2).contiguous()
y = torch.randn(B, N, 512, device=dev, requires_grad=True)
FB.USE_BLOCK_CALLS = False
for _ in range(3):
    out = FB.set_conv(sa, xyz, y)
    out.backward(torch.randn_like(out))
torch.cuda.synchronize()
M, P = B * N * S, B * N
alg = {
    # gather y[idx] (sources L2-resident: 33 MB), write z1 (M x 512) + dxyz (M x 4); idx (M) read
    "group_affine_kernel": 4 * M * 512 + 16 * M + 4 * M + 4 * P * 512,
    # read z3 (M x 64), write x (P x 64) + argmax (P x 64 bytes)
    "bn_relu_maxpool_kernel": 4 * M * 64 + 4 * P * 64 + P * 64,
    # read z3 (M x 64) + dx (P x 64) + argmax, write dU3 (M x 64)
    "maxpool_bwd_kernel": 8 * M * 64 + 4 * P * 64 + P * 64,
    # in place on dU (M x C) with z (M x C): read 2, write 1 -- C = 64 (layer 3) and 256 (layer 2)
    "bn_bwd_apply_kernel@64": 12 * M * 64,
    "bn_bwd_apply_kernel@256": 12 * M * 256,
    # read dU1 (M x 512) + inverse index (M), write dy (P x 512); y / xyz re-read per point
    "group_rows_grad_bn_cf_kernel": 4 * M * 512 + 4 * M + 8 * P * 512,
}
print("ALG " + json.dumps(alg))
