"""Run a few training steps and print how many hipGraphs the encoder calls recorded vs replayed."""
import os, sys, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from cmflow_amd import synth, _lib
from cmflow_amd.cmflow import CMFlow
from cmflow_amd.train import TrainStep
dev = torch.device("cuda:0")
net = CMFlow(bench.Args()); net.load_state_dict(bench.load_weights("cmflow")); net = net.to(dev).train()
step = TrainStep(net)
batch = {k: v.to(dev) for k, v in synth.make_batch(64, seed=1, train_extras=True).items()}
c, r = ctypes.c_longlong(), ctypes.c_longlong()
for i in range(8):
    step(batch)
    torch.cuda.synchronize()
    ms = _lib.lib().cmf_graph_stats(ctypes.addressof(c), ctypes.addressof(r))
    print("step", i, "captures", c.value, "replays", r.value, "capture ms total", ms, flush=True)
    for name, enc in (("mse2", net.mse_layer2), ("mse1", net.mse_layer)):
        for key, plan in enc._plans.items():
            d = plan.descs[0]
            print("   ", name, key[-1], "xyz %x y %x saved %x scratch %x out %x dout %x dy %x" % tuple(int(v or 0) for v in (d.xyz, d.y, d.saved, d.scratch, d.out, d.dout, d.dy)))
