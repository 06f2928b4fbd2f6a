"""Probe for rocprofv3 (kernel trace / --pmc FETCH_SIZE / --pmc WRITE_SIZE, one pass each): the drop-in calls the bench line's
`roofline_hbm` object times -- cmf_ball_query, cmf_group_points, cmf_query_and_group, cmf_group_points_grad through the C-ABI --
at its five shapes, REPS calls each.  The counted calls of a phase sit between two marker kernels (torch sigmoid_ / cos_ on 7 elements) so that
tools/op_table.py can cut the dispatch stream of every pass at the same places; prints one PHASE line per phase."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmflow_amd import _lib, synth
dev = torch.device("cuda:0")
L = _lib.lib()
st = _lib.stream_ptr()
REPS = 3
marker = torch.zeros(7, device=dev)
phases = []


def phase(name, shape, nbytes, fn):
    fn(); torch.cuda.synchronize()                      # warm (scratch growth, attributes) outside the counted calls
    marker.sigmoid_()
    for _ in range(REPS):
        fn()
    marker.cos_()                                       # end of the counted calls (the next phase allocates and warms up behind it)
    torch.cuda.synchronize()
    phases.append({"op": name, "shape_BNKC": shape, "bytes": nbytes, "reps": REPS})


for (B, N, K, r, lidar), Cs in (((64, 256, 32, 2.0, False), (3, 64, 1027)), ((32, 4096, 64, 2.0, True), (64, 128))):
    xyz = synth.make_batch(B, N=N, seed=1234, lidar=lidar)["pc1"].to(dev)
    xyz_t = xyz.transpose(1, 2).contiguous()
    idx = torch.zeros(B, N, K, dtype=torch.int32, device=dev)
    phase("ball_query", [B, N, K, 0], 2 * B * N * 12 + B * N * K * 4,
          lambda: _lib.check(L.cmf_ball_query(B, N, N, r, K, xyz_t.data_ptr(), xyz_t.data_ptr(), idx.data_ptr(), st), "bq"))
    for C in Cs:
        feats = torch.randn(B, C, N, device=dev)
        out = torch.empty(B, C, N, K, device=dev)
        phase("group_points", [B, N, K, C], B * C * N * 4 + B * N * K * 4 + B * C * N * K * 4,
              lambda: _lib.check(L.cmf_group_points(B, C, N, N, K, feats.data_ptr(), idx.data_ptr(), out.data_ptr(), st), "gf"))
        fused = torch.empty(B, 3 + C, N, K, device=dev)
        idx2 = torch.empty(B, N, K, dtype=torch.int32, device=dev)
        phase("query_and_group", [B, N, K, C], 2 * B * N * 12 + B * C * N * 4 + B * N * K * 4 + B * (3 + C) * N * K * 4,
              lambda: _lib.check(L.cmf_query_and_group(B, N, N, r, K, C, 1, xyz_t.data_ptr(), xyz_t.data_ptr(), feats.data_ptr(),
                                                       idx2.data_ptr(), fused.data_ptr(), st), "qg"))
        del fused, idx2
        out.normal_()
        gp = torch.zeros(B, C, N, device=dev)
        phase("group_points_grad", [B, N, K, C], B * C * N * K * 4 + B * N * K * 4 + B * C * N * 4,
              lambda: _lib.check(L.cmf_group_points_grad(B, C, N, N, K, out.data_ptr(), idx.data_ptr(), gp.data_ptr(), st), "gg"))
        del feats, out, gp
torch.cuda.synchronize()
print("PHASES " + json.dumps(phases))
