"""Timing probe of cmf_group_points_grad (the drop-in scatter) through the C-ABI at the bench line's shapes:
    python tools/gpg_probe.py [diag values ...]      (diag values only act on an experiment build: CMF_LIB=tools/diag/libcmflow_x.so)
Per shape and diag value: mean microseconds of one call (20 back-to-back calls between one event pair, best of 5) and the
fraction of 8 TB/s at SURVEY 8d's algorithmic bytes.  Checks the result against a float64 index_add of the same data."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmflow_amd import _lib, synth
dev = torch.device("cuda:0")
torch.manual_seed(0)
L = _lib.lib()
st = _lib.stream_ptr()
diags = sys.argv[1:] or ["0"]
for (B, N, K, r, lidar), Cs in (((64, 256, 32, 2.0, False), (64,)), ((32, 4096, 64, 2.0, True), (64, 128))):
    xyz = synth.make_batch(B, N=N, seed=1234, lidar=lidar)["pc1"].to(dev)
    xyz_t = xyz.transpose(1, 2).contiguous()
    idx = torch.zeros(B, N, K, dtype=torch.int32, device=dev)
    _lib.check(L.cmf_ball_query(B, N, N, r, K, xyz_t.data_ptr(), xyz_t.data_ptr(), idx.data_ptr(), st), "bq")
    for C in Cs:
        g = torch.randn(B, C, N, K, device=dev)
        gp = torch.zeros(B, C, N, device=dev)
        nbytes = B * C * N * K * 4 + B * N * K * 4 + B * C * N * 4
        call = lambda: _lib.check(L.cmf_group_points_grad(B, C, N, N, K, g.data_ptr(), idx.data_ptr(), gp.data_ptr(), st), "gg")
        for d in diags:
            os.environ["CMF_GC_DIAG"] = d
            gp.zero_(); call(); torch.cuda.synchronize()
            if d == "0":
                want = torch.zeros(B, C, N, dtype=torch.float64, device=dev)
                want.scatter_add_(2, idx.long().view(B, 1, N * K).expand(B, C, N * K), g.double().view(B, C, N * K))
                err = ((gp.double() - want).abs().max() / want.abs().max()).item()
                if os.environ.get("GPG_SAVE"):              # A/B of two builds: the results must be bit-identical
                    torch.save(gp.cpu(), "%s_%d_%d_%d_%d.pt" % (os.environ["GPG_SAVE"], B, N, K, C))
            best = 1e9
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    call()
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) * 1e3 / 20)
            print("(%d,%d,%d,C=%d) diag %s: %.1f us  %.3f of 8 TB/s%s" % (B, N, K, C, d, best, nbytes / best / 1e6 / 8.0,
                                                                      "  rel.err %.1e" % err if d == "0" else ""), flush=True)
        del g, gp
