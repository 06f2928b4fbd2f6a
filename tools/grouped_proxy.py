"""A grouped launch of one layer of the four scales, by proxy: the four data-gradient GEMMs of the second encoder's widest layer
(M = 65536 / 131072 / 262144 / 524288, N = 512, K = 256, backward through BN + ReLU with the dxyz sums)
  (a) as four launches on four streams (what the step does: the scales' chains run concurrently),
  (b) as four launches one after the other on one stream,
  (c) as ONE persistent launch over M = 983040 rows (the same tiles, flops and bytes as the four together; one weight matrix
      instead of four -- 0.5 MB each, L2-resident either way): what a segment table over the four problems would cost.
HIP events around 10 repetitions of each form, A/B/C/A/B/C in one process."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmflow_amd import _lib
from cmflow_amd.fused import gemm
from cmflow_amd.fused_blocks import side_stream
dev = torch.device("cuda:0")
torch.manual_seed(0)
Ms = (65536, 131072, 262144, 524288)
N, K = 512, 256
ea, ec, em, ei = (torch.rand(N, device=dev) + 0.5 for _ in range(4))
probs = []
for M in Ms + (sum(Ms),):
    probs.append(dict(dZ=torch.randn(M, K, device=dev), W=torch.randn(K, N, device=dev), Z=torch.randn(M, N, device=dev),
                      d4=torch.randn(M, 4, device=dev), out=torch.empty(M, N, device=dev),
                      st=torch.empty(M // 128, 5, N, device=dev)))
L = _lib.lib()


def call(p):
    # straight through the C-ABI (no allocation inside the timed region)
    _lib.check(L.cmf_gemm(p["dZ"].shape[0], N, K, 0, 0, p["dZ"].data_ptr(), K, p["W"].data_ptr(), N, p["out"].data_ptr(), N,
                          None, None, None, None, None, 0, p["st"].data_ptr(), 1, p["Z"].data_ptr(), N,
                          ea.data_ptr(), ec.data_ptr(), em.data_ptr(), ei.data_ptr(), p["d4"].data_ptr(), 1, None, 0, _lib.stream_ptr()), "gemm")


main = torch.cuda.current_stream()
streams = [main] + [side_stream(i) for i in range(3)]


def four_streams():
    for st in streams[1:]:
        st.wait_stream(main)
    for p, st in zip(probs[:4][::-1], streams):           # largest on the caller's stream
        with torch.cuda.stream(st):
            call(p)
    for st in streams[1:]:
        main.wait_stream(st)


def four_serial():
    for p in probs[:4]:
        call(p)


def one_launch():
    call(probs[4])


def timed(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


fl = 2.0 * sum(Ms) * N * K
for mode, name in ((0, "tiled kernel"), (1, "persistent kernel (product default)")):
    L.cmf_gemm_persist_config(mode, 0)
    for rep in range(2):
        ta, tb, tc = timed(four_streams), timed(four_serial), timed(one_launch)
        print("%-36s four streams %7.1f us (%5.1f TF) | one stream %7.1f us (%5.1f TF) | one launch %7.1f us (%5.1f TF)"
              % (name, ta, fl / ta / 1e6, tb, fl / tb / 1e6, tc, fl / tc / 1e6), flush=True)
L.cmf_gemm_persist_config(1, 0)
