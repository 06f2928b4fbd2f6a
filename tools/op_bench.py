"""Op-level roofline of the drop-in grouping path: ball_query + group_points
(+ group_points_grad) through the reference-shaped wrappers, HIP-event timed, inputs resident in HBM.

algorithmic bytes = 2*B*N*12 (xyz, centres) + B*C*N*4 (features) + B*M*K*4 (idx) + B*(3+C)*M*K*4 (grouped out)
    python tools/op_bench.py            # prints a markdown table + one JSON line per row
"""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmflow_amd import synth
from cmflow_amd.pointnet2_utils import QueryAndGroup, ball_query, grouping_operation

dev = torch.device("cuda:0")


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


rows = []
for (B, N, K, r, lidar), Cs in ((((64, 256, 32, 2.0, False), (3, 64, 1027)), ((32, 4096, 64, 2.0, True), (64, 128)))):
    xyz = synth.make_batch(B, N=N, seed=1234, lidar=lidar)["pc1"].permute(0, 2, 1).contiguous().to(dev)
    for C in Cs:
        feats = torch.randn(B, C, N, device=dev)
        xyz_c = xyz.transpose(1, 2).contiguous()
        idx = ball_query(r, K, xyz, xyz)
        t_bq = timed(lambda: ball_query(r, K, xyz, xyz))
        t_gx = timed(lambda: grouping_operation(xyz_c, idx))
        t_gf = timed(lambda: grouping_operation(feats, idx))
        out = grouping_operation(feats, idx)
        go = torch.randn_like(out)
        from cmflow_amd.pointnet2_utils import group_points_grad_wrapper
        gp = torch.zeros(B, C, N, device=dev)
        t_gg = timed(lambda: group_points_grad_wrapper(B, C, N, N, K, go, idx, gp))
        nbytes = 2 * B * N * 12 + B * C * N * 4 + B * N * K * 4 + B * (3 + C) * N * K * 4
        t_all = t_bq + t_gx + t_gf
        gbytes = B * C * N * K * 4 + B * N * K * 4 + B * C * N * 4
        row = {"shape(B,N,K,C)": [B, N, K, C], "algorithmic_MB": round(nbytes / 1e6, 1),
               "ball_query_us": round(t_bq * 1e6, 1), "group_xyz_us": round(t_gx * 1e6, 1), "group_feat_us": round(t_gf * 1e6, 1),
               "bq+group_us": round(t_all * 1e6, 1), "bq+group_GBs": round(nbytes / t_all / 1e9, 1),
               "pct_of_8TBs": round(nbytes / t_all / 8e12 * 100, 1),
               "group_feat_only_GBs": round((B * C * N * 4 + B * N * K * 4 + B * C * N * K * 4) / t_gf / 1e9, 1),
               "group_grad_us": round(t_gg * 1e6, 1), "group_grad_GBs": round(gbytes / t_gg / 1e9, 1)}
        rows.append(row)
        print(json.dumps(row))
print("\n| shape (B,N,K,C) | algorithmic MB | ball_query us | group(xyz)+group(feat) us | bq+group GB/s | % of 8 TB/s | group(feat) alone GB/s | group_grad GB/s |")
print("|---|---|---|---|---|---|---|---|")
for r_ in rows:
    print("| %s | %s | %s | %s | %s | %s | %s | %s |" % (tuple(r_["shape(B,N,K,C)"]), r_["algorithmic_MB"], r_["ball_query_us"],
          round(r_["group_xyz_us"] + r_["group_feat_us"], 1), r_["bq+group_GBs"], r_["pct_of_8TBs"], r_["group_feat_only_GBs"], r_["group_grad_GBs"]))
