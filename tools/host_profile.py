"""Host-side (python) cost of one training step: cProfile over 10 steps, top functions by own and cumulative time."""
import cProfile, os, pstats, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from cmflow_amd import synth
from cmflow_amd.cmflow import CMFlow
from cmflow_amd.train import TrainStep
dev = torch.device("cuda:0")
net = CMFlow(bench.Args()); net.load_state_dict(bench.load_weights("cmflow")); net = net.to(dev).train()
step = TrainStep(net)
batch = {k: v.to(dev) for k, v in synth.make_batch(64, seed=1, train_extras=True).items()}
for _ in range(5):
    step(batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    step(batch)
t_host = (time.perf_counter() - t0) / 10
torch.cuda.synchronize()
t_all = (time.perf_counter() - t0) / 10
print("host enqueue time per step %.2f ms; with the GPU drained %.2f ms" % (t_host * 1e3, t_all * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    step(batch)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(18)
st.sort_stats("cumulative").print_stats(30)
