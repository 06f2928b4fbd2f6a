"""Timing probe of cmf_group_points (the drop-in gather) through the C-ABI at the bench line's config-5 shapes and one model shape:
    python tools/gp_probe.py        (CMF_LIB=tools/diag/libcmflow_x.so for an experiment build)
Mean microseconds of one call (20 back-to-back calls between one event pair, best of 5), fraction of 8 TB/s at SURVEY 8d's bytes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmflow_amd import _lib, synth
dev = torch.device("cuda:0")
torch.manual_seed(0)
L = _lib.lib()
st = _lib.stream_ptr()
for (B, N, K, r, lidar), Cs in (((64, 256, 32, 2.0, False), (64,)), ((32, 4096, 64, 2.0, True), (64, 128))):
    xyz = synth.make_batch(B, N=N, seed=1234, lidar=lidar)["pc1"].to(dev)
    xyz_t = xyz.transpose(1, 2).contiguous()
    idx = torch.zeros(B, N, K, dtype=torch.int32, device=dev)
    _lib.check(L.cmf_ball_query(B, N, N, r, K, xyz_t.data_ptr(), xyz_t.data_ptr(), idx.data_ptr(), st), "bq")
    for C in Cs:
        res = []
        for place in range(4):                          # the rate depends on where the tensors land (HBM channel mapping): 4 placements
            pad = torch.empty((1 + place) * 1234567 + place * 333, device=dev)      # moves the next allocations
            feats = torch.randn(B, C, N, device=dev)
            out = torch.empty(B, C, N, K, device=dev)
            nbytes = B * C * N * 4 + B * N * K * 4 + B * C * N * K * 4
            call = lambda: _lib.check(L.cmf_group_points(B, C, N, N, K, feats.data_ptr(), idx.data_ptr(), out.data_ptr(), st), "gp")
            call(); torch.cuda.synchronize()
            if place == 0:
                want = torch.gather(feats, 2, idx.long().view(B, 1, N * K).expand(B, C, N * K)).view(B, C, N, K)
                ok = torch.equal(out, want)
                del want
            best = 1e9
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    call()
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) * 1e3 / 20)
            res.append(best)
            del feats, out, pad
            torch.cuda.empty_cache()
        print("(%d,%d,%d,C=%d): %s us  mean %.3f of 8 TB/s  exact %s" % (B, N, K, C, " ".join("%.1f" % r for r in res),
                                                                   nbytes / (sum(res) / len(res)) / 1e6 / 8.0, ok), flush=True)
