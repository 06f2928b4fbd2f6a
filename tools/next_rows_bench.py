"""Timings of the SURVEY 8f rows that are single ops (HIP-event timed, inputs resident in HBM, B=64 N=256):
fused loss (cmf_radar_loss) vs the torch-op terms on the same device, evaluation metrics (cmf_eval_metrics) vs the
reference's way (copy to host + numpy, oracle/eval_oracle.py), pseudo labels (cmf_pseudo_labels) vs torch ops."""
import os, sys, time, warnings
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmflow_amd import synth, eval_util as EU
from cmflow_amd.losses import RadarFlowLoss, make_labels
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from loss_torch import TorchRadarFlowLoss, make_labels_torch                       # test fixture: the torch-op terms
from oracle import eval_oracle as EO

dev = torch.device("cuda:0")
B, N = 64, 256
batch = {k: v.to(dev) for k, v in synth.make_batch(B, N, seed=1, train_extras=True).items()}
g = torch.Generator().manual_seed(0)
gt_f = batch["flow_label"].transpose(2, 1).contiguous()
pred_f = (gt_f + 0.3 * torch.randn(B, 3, N, generator=g).to(dev)).requires_grad_(True)
pre_trans = batch["gt_trans"].clone().requires_grad_(True)
mseg_pre = torch.sigmoid(torch.randn(B, 1, N, generator=g)).to(dev).requires_grad_(True)


def timed(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


rows = []
dyn, mseg = make_labels(batch, 0.3)
for native in (True, False):
    crit = (RadarFlowLoss if native else TorchRadarFlowLoss)(synth.CAMERA_PROJECTION, synth.T_CAMERA_RADAR).to(dev)

    def step():
        for t in (pred_f, pre_trans, mseg_pre):
            t.grad = None
        total, _ = crit(batch["pc1"], batch["pc2"], pred_f, batch["ft1"][:, 0], gt_f, pre_trans, mseg_pre, batch["gt_trans"],
                        mseg, dyn, batch["radar_u"], batch["radar_v"], batch["opt_flow"])
        total.backward()
    rows.append(("RadarFlowLoss fwd+bwd, " + ("cmf_radar_loss (3 launches)" if native else "torch-op terms on the GPU"), timed(step)))
rows.append(("pseudo labels, cmf_pseudo_labels (1 launch)", timed(lambda: make_labels(batch, 0.3))))
rows.append(("pseudo labels, torch ops on the GPU", timed(lambda: make_labels_torch(batch, 0.3))))

pred = batch["flow_label"] + 0.1 * torch.randn(B, N, 3, generator=g).to(dev)
mask, pm = batch["fg_mask"], (torch.rand(B, N, generator=g) < 0.5).float().to(dev)


class A:
    radar_res = EU.VOD_RADAR_RES


rows.append(("eval metrics (14), cmf_eval_metrics incl. reading the 14 doubles back",
             timed(lambda: [float(v) for d in EU.eval_batch(batch["pc1"], pred, batch["flow_label"], mask, pm, batch["gt_trans"],
                                                            batch["gt_trans"], A) for v in d.values()])))


def host_eval():
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        EO.scene_flow_metrics(batch["pc1"].cpu().numpy(), pred.cpu().numpy(), batch["flow_label"].cpu().numpy(),
                              mask.cpu().numpy(), EU.VOD_RADAR_RES)
        EO.motion_seg_metrics(pm.cpu().numpy(), mask.cpu().numpy())
        EO.pose_metrics(batch["gt_trans"].cpu().numpy(), batch["gt_trans"].cpu().numpy())


t0 = time.perf_counter()
for _ in range(5):
    host_eval()
rows.append(("eval metrics, the reference's way: copy to host + numpy/scipy (oracle port)", (time.perf_counter() - t0) / 5 * 1e3))
print("| op (B=64, N=256) | ms |\n|---|---|")
for name, ms in rows:
    print("| %s | %.3f |" % (name, ms))
