"""How much of a training step is host (Python + launch) time?  Times the enqueue of K steps without a
sync against the synchronised wall time."""
import os, sys, time, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from cmflow_amd import synth
from cmflow_amd.cmflow import CMFlow
from cmflow_amd.train import TrainStep
dev = torch.device("cuda:0")
net = CMFlow(bench.Args()); net.load_state_dict(bench.load_weights("cmflow")); net = net.to(dev).train()
step = TrainStep(net)
batch = {k: v.to(dev) for k, v in synth.make_batch(int(os.environ.get("PROBE_B", "64")), seed=1234, train_extras=True).items()}
for _ in range(3): step(batch)
torch.cuda.synchronize()
K = 10
t0 = time.perf_counter()
for _ in range(K): step(batch)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("enqueue %.1f ms/step, wall %.1f ms/step" % ((t1 - t0) / K * 1e3, (t2 - t0) / K * 1e3))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(3): step(batch)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(45)
