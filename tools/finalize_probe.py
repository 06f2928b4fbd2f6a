"""What folding the per-tile statistics inside the persistent GEMM would buy: cmf_colsum (5 statistics of a kind-4 data gradient)
and cmf_bn_finalize (2 statistics of a forward layer) on the partial matrices the kernels write today ([tiles_m] rows, one per
128-row tile) against the [128]-row matrices a per-workgroup fold would leave (512 workgroups / 4 column tiles), 20 launches
each between one HIP event pair.  Plus the store side: the bytes a GEMM launch writes for its partials."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmflow_amd import _lib
dev = torch.device("cuda:0")
L = _lib.lib()
st = _lib.stream_ptr()


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for rows in (524288, 262144, 131072, 65536):
    for tiles in (rows // 128, 128):
        C = 512
        part = torch.randn(tiles, 5, C, device=dev)
        out = torch.empty(5, C, device=dev)
        t5 = timed(lambda: _lib.check(L.cmf_colsum(tiles, 5 * C, part.data_ptr(), out.data_ptr(), C, None, None, st), "colsum"))
        C2 = 256
        part2 = torch.randn(tiles, 2, C2, device=dev)
        g, b = torch.ones(C2, device=dev), torch.zeros(C2, device=dev)
        rm, rv = torch.zeros(C2, device=dev), torch.ones(C2, device=dev)
        o = [torch.empty(C2, device=dev) for _ in range(4)]
        nbt = torch.zeros(1, dtype=torch.int64, device=dev)
        t2 = timed(lambda: _lib.check(L.cmf_bn_finalize(tiles, C2, float(rows), part2.data_ptr(), g.data_ptr(), b.data_ptr(), 1e-5, 0.1,
                                                        rm.data_ptr(), rv.data_ptr(), o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(),
                                                        o[3].data_ptr(), nbt.data_ptr(), st), "bnfin"))
        print("rows %7d  partial rows %5d: colsum [5 x 512] %6.1f us (%.1f MB)   bn_finalize [2 x 256] %6.1f us (%.1f MB)"
              % (rows, tiles, t5, tiles * 5 * C * 4 / 1e6, t2, tiles * 2 * C2 * 4 / 1e6), flush=True)
