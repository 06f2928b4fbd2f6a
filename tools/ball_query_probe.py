"""cmf_ball_query at config 5's shape (B = 32, N = 4096, nsample 64, synthetic lidar cloud): cell-grid kernel vs the scan
(CMF_BALL_QUERY_GRID=0 in a child process), results compared bit for bit."""
import os, subprocess, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmflow_amd import _lib, synth

def run():
    dev = torch.device("cuda:0")
    L = _lib.lib()
    out = {}
    for (B, N, K, r, lidar) in ((32, 4096, 64, 2.0, True), (32, 4096, 64, 0.5, True), (8, 8192, 32, 1.0, True), (32, 2048, 16, 4.0, False)):
        xyz = synth.make_batch(B, N=N, seed=1234, lidar=lidar)["pc1"].to(dev).transpose(1, 2).contiguous()
        idx = torch.zeros(B, N, K, dtype=torch.int32, device=dev)
        f = lambda: _lib.check(L.cmf_ball_query(B, N, N, r, K, xyz.data_ptr(), xyz.data_ptr(), idx.data_ptr(), _lib.stream_ptr()), "bq")
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        out[(B, N, K, r)] = (e0.elapsed_time(e1) / 10 * 1e3, idx.cpu())
    return out

if __name__ == "__main__":
    if len(sys.argv) > 1:
        res = run()
        torch.save({str(k): v for k, v in res.items()}, sys.argv[1])
    else:
        res = run()
        env = dict(os.environ, CMF_BALL_QUERY_GRID="0")
        subprocess.run([sys.executable, os.path.abspath(__file__), "/tmp/bq_scan.pt"], check=True, env=env)
        ref = torch.load("/tmp/bq_scan.pt")
        for k, (t, idx) in res.items():
            t0, idx0 = ref[str(k)]
            print("B=%d N=%d nsample=%d r=%.1f: grid %.1f us, scan %.1f us, identical: %s" % (*k, t, t0, bool(torch.equal(idx, idx0))))
