"""Locate the first irreproducible tensor of the second encoder under bf16x3 + side streams (Python-sequenced path)."""
import os, sys, torch, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from cmflow_amd import _lib, synth, fused_blocks as FB, fused
from cmflow_amd.cmflow import CMFlow
dev = torch.device("cuda:0")
_lib.set_gemm_mode("bf16x3")
torch.manual_seed(0)
net = CMFlow(bench.Args()); net.load_state_dict(bench.load_weights("cmflow")); net = net.to(dev).eval()
enc = net.mse_layer2
B, N = 64, 256
xyz = synth.make_batch(B, seed=77)["pc1"].to(dev).transpose(1, 2).contiguous()
emb = torch.randn(B, N, 1040, device=dev); emb[:, :, 1027:] = 0
FB.USE_BLOCK_CALLS = False
log = None
real_gemm, real_ga, real_mp = fused.gemm, FB.group_affine, FB.bn_relu_maxpool
def rec(tag, t):
    if log is not None:
        c = t.detach().clone()                       # asynchronous, on the producing stream
        c.record_stream(torch.cuda.current_stream())
        log.append((tag + " " + str(tuple(t.shape)), c))
def gemm(A, Bm, **kw):
    out = real_gemm(A, Bm, **kw)
    o = out[0] if isinstance(out, tuple) else out
    rec("gemm %s pro=%s" % (tuple(A.shape), kw.get("pro") is not None), o)
    rec("   its input", A)
    return out
def ga(*a, **k):
    rec("group_affine IN ysrc", a[0].contiguous()); rec("group_affine IN idx", a[5].float())
    out = real_ga(*a, **k); rec("group_affine", out[0])
    rec("group_affine IN ysrc (after)", a[0].contiguous())
    return out
def mp(z, st, out=None):
    o = real_mp(z, st, out); rec("maxpool", o[0]); return o
FB.gemm = gemm; FB.group_affine = ga; FB.bn_relu_maxpool = mp
runs = []
for r in range(5):
    log = []
    with torch.no_grad():
        out = enc.forward_pm(xyz, emb, n_tail=3, n_grad=1024)
    torch.cuda.synchronize()
    d = collections.OrderedDict()
    for tag, c in log:
        k = tag
        while k in d:
            k += "'"
        d[k] = c
    runs.append((d, out.clone()))
ref = runs[0][0]
for r in range(1, 5):
    bad = [k for k in ref if not torch.equal(ref[k], runs[r][0][k])]
    print("run", r, "entries", len(ref), "differing", len(bad), "out equal", torch.equal(runs[r][1], runs[0][1]))
    for k in bad[:12]:
        a, b = ref[k], runs[r][0][k]
        nd = (a != b)
        rows = nd.view(a.shape[0], -1).any(dim=1).nonzero().flatten()
        print("   ", k, "differing elements", int(nd.sum()), "max diff %.3g" % (a - b).abs().max().item(),
              "rows %d..%d (%d rows)" % (int(rows.min()), int(rows.max()), rows.numel()))
