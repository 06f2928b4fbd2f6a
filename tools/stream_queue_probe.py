"""Which streams share a hardware queue?  Two streams mapped onto the same HW queue serialise their kernels (fused_blocks.py "side
streams"); the runtime deals streams to its 4 queues as they are created, so a library that creates streams first (RCCL at
init_process_group) shifts the deal.  Prints, for the current stream and the first pool streams, whether two 400 us one-wave spin
kernels launched on a pair run side by side (.) or one after the other (X) -- without and with a world-1 nccl process group created first
(argv[1] == "pg")."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cmflow_amd import _lib

if len(sys.argv) > 1 and sys.argv[1] == "pg":
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    t = torch.zeros(1024, device="cuda"); dist.all_reduce(t); torch.cuda.synchronize()
L = _lib.lib()
main = torch.cuda.current_stream()
streams = [main] + [torch.cuda.Stream() for _ in range(9)]


def serial(a, b, us=400.0):
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        L.cmf_debug_spin(us, a.cuda_stream); L.cmf_debug_spin(us, b.cuda_stream)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best > 1.6 * us * 1e-6, best


print("     " + " ".join("%2d" % j for j in range(len(streams))), "  (0 = current stream, 1.. = torch.cuda.Stream() in creation order)")
for i, a in enumerate(streams):
    row = []
    for j, b in enumerate(streams):
        row.append(" -" if i == j else (" X" if serial(a, b)[0] else " ."))
    print("%2d: %s" % (i, " ".join(row)))
from cmflow_amd import fused_blocks as FB
pool = [FB.side_stream(i) for i in range(FB.N_SIDE)]
print("side-stream pool vs current stream:", [("X" if serial(main, s)[0] else ".") for s in pool],
      " among themselves:", [("X" if serial(pool[i], pool[j])[0] else ".") for i in range(len(pool)) for j in range(i + 1, len(pool))])
