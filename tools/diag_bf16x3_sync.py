"""bf16x3 + side streams: does a device synchronisation between the stacked first-conv GEMM and the scale chains, or the
order of host-side launches, change the reproducibility?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from cmflow_amd import _lib, synth, fused_blocks as FB
from cmflow_amd.cmflow import CMFlow
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = CMFlow(bench.Args()); net.load_state_dict(bench.load_weights("cmflow")); net = net.to(dev).eval()
enc = net.mse_layer2
B, N = 64, 256
xyz = synth.make_batch(B, seed=77)["pc1"].to(dev).transpose(1, 2).contiguous()
emb = torch.randn(B, N, 1040, device=dev); emb[:, :, 1027:] = 0
FB.USE_BLOCK_CALLS = False
real_apply = FB.StackedFirstConvFn.apply
mode_sync = [None]
def patched(*a):
    y = real_apply(*a)
    if mode_sync[0] == "sync":
        torch.cuda.synchronize()
    elif mode_sync[0] == "sleep":
        torch.cuda._sleep(3_000_000)                  # ~1.5 ms of spinning on the main stream after the GEMM
    return y
FB.StackedFirstConvFn.apply = patched
def run():
    with torch.no_grad():
        out = enc.forward_pm(xyz, emb, n_tail=3, n_grad=1024)
    torch.cuda.synchronize()
    return out
for gm in ("fp32", "bf16x3"):
    _lib.set_gemm_mode(gm)
    for ms in (None, "sync", "sleep"):
        mode_sync[0] = ms
        ref = run()
        n_bad = sum(0 if torch.equal(run(), ref) else 1 for _ in range(12))
        print(gm, "after the stacked GEMM:", ms, "-> irreproducible repeats:", n_bad, "of 12")
