"""Where does the second encoder become irreproducible under bf16x3?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from cmflow_amd import _lib, synth, fused_blocks as FB
from cmflow_amd.cmflow import CMFlow
from cmflow_amd.fused import gemm
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = CMFlow(bench.Args()); net.load_state_dict(bench.load_weights("cmflow")); net = net.to(dev).eval()
enc = net.mse_layer2
B, N = 64, 256
xyz = synth.make_batch(B, seed=77)["pc1"].to(dev).transpose(1, 2).contiguous()
emb = torch.randn(B, N, 1040, device=dev)
emb[:, :, 1027:] = 0
def run(blocks, multi):
    FB.USE_BLOCK_CALLS = blocks
    enc.multi_stream = multi
    with torch.no_grad():
        out = enc.forward_pm(xyz, emb, n_tail=3, n_grad=1024)
    torch.cuda.synchronize()
    return out
for mode in ("fp32", "bf16x3"):
    _lib.set_gemm_mode(mode)
    for blocks, multi in ((True, True), (False, True), (False, False)):
        ref = run(blocks, multi)
        diffs = []
        for _ in range(6):
            o = run(blocks, multi)
            d = (o - ref).abs().amax(dim=(0, 1)).view(4, 64).amax(dim=1)      # per scale
            diffs.append([float(x) for x in d])
        print(mode, "blocks", blocks, "multi_stream", multi, "max diff per scale over 6 repeats:", [max(c) for c in zip(*diffs)])
# single kinds of GEMM in eval mode: prologue without statistics, 128x64 tiles, thin
_lib.set_gemm_mode("bf16x3")
M = 524288
z1 = torch.randn(M, 512, device=dev); w2 = torch.randn(256, 512, device=dev); w3 = torch.randn(64, 256, device=dev)
pa, pc = torch.rand(512, device=dev) + 0.5, torch.randn(512, device=dev) * 0.1
pa2, pc2 = torch.rand(256, device=dev) + 0.5, torch.randn(256, device=dev) * 0.1
r1 = gemm(z1, w2, pro=(pa, pc)); r2 = gemm(r1, w3, pro=(pa2, pc2))
for _ in range(5):
    a = gemm(z1, w2, pro=(pa, pc)); b = gemm(a, w3, pro=(pa2, pc2))
    print("pro-only GEMM repeat equal:", torch.equal(a, r1), " 128x64 next layer equal:", torch.equal(b, r2))
