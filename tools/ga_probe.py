"""Timing of cmf_group_affine (gather + hoisted first conv + BN partial sums) at the second encoder's four scales (B = 64, N = 256,
C = 512), training form (statistics + dW_xyz sums) and inference form: 20 calls between one event pair, best of 3, 3 placements.
A checksum of z and of the partial sums is printed so that two builds can be compared for bit-identity."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmflow_amd import _lib, synth, fused_blocks as FB, pointnet2_utils as pu
dev = torch.device("cuda:0")
B, N = 64, 256
xyz = synth.make_batch(B, seed=1234)["pc1"].to(dev).transpose(1, 2).contiguous()
for S, r in ((32, 16.0), (16, 8.0), (8, 4.0), (4, 2.0)):
    idx = pu.ball_query(r, S, xyz, xyz)
    M = B * N * S
    for stats in (True, False):
        res = []
        for place in range(3):
            torch.manual_seed(S)
            pad = torch.empty((1 + place) * 1234567, device=dev)
            y = torch.randn(B, N, 512, device=dev); wx = torch.randn(512, 3, device=dev)
            fn = lambda: FB.group_affine(y, None, xyz, xyz, wx, idx, act=0, stats=stats, extra=stats)
            out = fn(); torch.cuda.synchronize()
            if place == 0:
                chk = [float(t.double().sum()) for t in out if torch.is_tensor(t) and t.is_floating_point()]
            best = 1e9
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    fn()
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) * 1e3 / 20)
            res.append(best); del y, wx, pad, out; torch.cuda.empty_cache()
        print("rows %7d %s: %s us  %.2f TB/s written  checksums %s" % (M, "train" if stats else "eval ", " ".join("%.1f" % t for t in res),
                                                                   4 * M * 512 / min(res) / 1e6, " ".join("%.10e" % c for c in chk)), flush=True)
