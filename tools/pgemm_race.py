"""Race screen for the persistent GEMM: repeat one call, compare every output element and statistic with the tiled kernel's,
report where mismatches sit (tile, wave quadrant, row / column inside the tile)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmflow_amd import _lib
from cmflow_amd.fused import gemm
dev = torch.device("cuda:0")
L = _lib.lib()
M, N, K = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "65536x512x256").split("x"))
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dxyz = int(sys.argv[3]) if len(sys.argv) > 3 else 1
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 30
grid = int(sys.argv[5]) if len(sys.argv) > 5 else 0
torch.manual_seed(0)
dZ = torch.randn(M, K, device=dev); W = torch.randn(K, N, device=dev); Zp = torch.randn(M, N, device=dev)
ea, ec, em, ei = (torch.rand(N, device=dev) + 0.5 for _ in range(4))
d4 = torch.randn(M, 4, device=dev)
bwd = (mode, Zp, ea, ec, em, ei) if mode == 1 else (mode, Zp, None, None, None, None)
if dxyz:
    bwd = bwd + (d4,)
st = mode == 1 or bool(dxyz)
L.cmf_gemm_persist_config(0, 0)
r = gemm(dZ, W, b_t=False, bwd=bwd, stats=st)
ref, ref_st = (r if st else (r, None))
torch.cuda.synchronize()
L.cmf_gemm_persist_config(2, grid)
bad = 0
first_st = None
for i in range(reps):
    r = gemm(dZ, W, b_t=False, bwd=bwd, stats=st)
    got, got_st = (r if st else (r, None))
    torch.cuda.synchronize()
    if st:
        if first_st is None:
            first_st = got_st.clone()
        elif not torch.equal(first_st, got_st):
            d = (first_st != got_st).nonzero()
            print("rep %d: %d statistics differ from rep 0; first at (tile, which, col) %s" % (i, d.shape[0], d[:4].tolist()), flush=True)
    if not torch.equal(got, ref):
        bad += 1
        d = (got != ref).nonzero()
        rows, cols = d[:, 0], d[:, 1]
        print("rep %d: %d elements differ; tiles (tm,tn) %s; rows in tile %s; cols in tile %s; sample got/ref %s" % (
            i, d.shape[0], sorted(set(zip((rows // 128).tolist(), (cols // 128).tolist())))[:6],
            sorted(set((rows % 128).tolist()))[:40], sorted(set((cols % 128).tolist()))[:40],
            [(float(got[a, b]), float(ref[a, b])) for a, b in d[:3].tolist()]), flush=True)
print("%d of %d repetitions differ" % (bad, reps))
L.cmf_gemm_persist_config(1, 0)
