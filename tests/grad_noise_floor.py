"""Not a test -- the measurement behind the bounds of tests/test_gpu_model.py::_check_gradients: the CPU oracle's training
step at the benchmark's size (B = 64) evaluated in fp32 and in fp64, and the three per-tensor gradient metrics between the
two.  Two correct fp32 implementations cannot agree tighter than this.  Measured in the build container (8 cores, 77 s +
143 s):  norm  median 7.3e-5  p90 2.5e-4  max 5.9e-4;   1 - cos  median 3.6e-7  p90 6.5e-7  max 3.6e-6;
         largest element / largest entry  median 8.5e-4  p90 1.6e-3  max 1.31e-2 (mse_layer2.ms_ls.3.mlp_convs.1.weight).

    python tests/grad_noise_floor.py [B]"""
import os, sys, time, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from cmflow_amd import synth
from oracle import cmflow_oracle as O, train_oracle as TO, ops
torch.set_num_threads(8)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
# index ops always on fp32 coordinates; grouping dtype-generic
bq0, knn0, gp0, gpg0 = ops.ball_query, ops.knn, ops.group_points, ops.group_points_grad
ops.ball_query = lambda r, ns, xyz, new: bq0(r, ns, xyz.float(), new.float())
ops.knn = lambda ns, xyz, new, return_dist=False: knn0(ns, xyz.float(), new.float(), return_dist)
def gp(points, idx):
    if points.dtype == torch.float32: return gp0(points, idx)
    Bb, C, N = points.shape; _, P, S = idx.shape
    return torch.gather(points, 2, idx.long().view(Bb, 1, P * S).expand(-1, C, -1)).view(Bb, C, P, S)
def gpg(go, idx, N):
    if go.dtype == torch.float32: return gpg0(go, idx, N)
    Bb, C, P, S = go.shape
    out = torch.zeros(Bb, C, N, dtype=go.dtype)
    out.scatter_add_(2, idx.long().view(Bb, 1, P * S).expand(-1, C, -1), go.reshape(Bb, C, P * S))
    return out
ops.group_points, ops.group_points_grad = gp, gpg
sd = bench.load_weights("cmflow")
b = synth.make_batch(B, seed=1234, train_extras=True)
P, Tcr = torch.tensor(synth.CAMERA_PROJECTION), torch.tensor(synth.T_CAMERA_RADAR)
grads = {}
for dt in (torch.float32, torch.float64):
    net = O.CMFlow(bench.Args()); net.load_state_dict(sd); net = net.to(dt).train()
    class NoStep:
        def zero_grad(self): net.zero_grad()
        def step(self): pass
    bb = {k: (v.to(dt) if v.is_floating_point() else v) for k, v in b.items()}
    t0 = time.time()
    loss, items, out, _ = TO.train_step(net, NoStep(), bb, P.to(dt), Tcr.to(dt))
    print(dt, "loss", loss.item(), "%.1f s" % (time.time() - t0), flush=True)
    grads[dt] = {k: (p.grad.double().reshape(-1).clone() if p.grad is not None else None) for k, p in net.named_parameters()}
rows = []
for k, r in grads[torch.float64].items():
    a = grads[torch.float32][k]
    if r is None: continue
    na, nr = float(a.norm()), float(r.norm())
    rows.append((k, abs(na - nr) / max(nr, 1e-3), 1 - float(a @ r) / (na * nr) if nr > 1e-6 else 0.0,
                 float((a - r).abs().max()) / max(float(r.abs().max()), 1e-6)))
import numpy as np
arr = np.array([[x[1], x[2], x[3]] for x in rows])
for j, name in enumerate(("norm", "1-cos", "element")):
    i = int(arr[:, j].argmax())
    print(name, "median %.3g  p90 %.3g  max %.3g (%s)" % (np.median(arr[:, j]), np.percentile(arr[:, j], 90), arr[i, j], rows[i][0]))
