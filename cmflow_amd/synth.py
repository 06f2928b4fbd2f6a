"""Deterministic synthetic inputs and weights for parity tests and bench.py.

There is no network for datasets or checkpoints, so every test/bench input is generated
from a seed with the torch CPU generator (bit-reproducible across machines for a fixed
torch version).  Shapes and value ranges follow the reference's data contract:

* clouds: ``dataset/vod.py:49-124`` -- N=256 points, x in [1,99] m, y in [-42,45], z in
  [-3,3]; features ``[v_r, RCS, RCS]`` (``dataset/vod.py:62-63``).
* weights: key/shape manifest of ``models/cmflow.py:12-48`` / ``models/cmflow_t.py:44-47``
  (374 / 378 state tensors), committed as ``tests/golden/state_manifest_*.json``.
"""
import math

import torch

# dataset/vod_radar_calib.txt:3,6 (P2 intrinsics, Tr_velo_to_cam extrinsics) -- needed only
# by the optical-flow loss term (utils/util.py:31-58).
CAMERA_PROJECTION = [[1495.468642, 0.0, 961.272442, 0.0],
                     [0.0, 1495.468642, 624.89592, 0.0],
                     [0.0, 0.0, 1.0, 0.0]]
T_CAMERA_RADAR = [[-0.013857, -0.9997468, 0.01772762, 0.05283124],
                  [0.10934269, -0.01913807, -0.99381983, 0.98100483],
                  [0.99390751, -0.01183297, 0.1095802, 1.44445002],
                  [0.0, 0.0, 0.0, 1.0]]


def _mixture(g, B, N, scene, box_lo, box_hi, f_sub=0.5, f_reg=0.42):
    """Two-level mixture calibrated to real VoD radar occupancy (mean points within r = 2/4/8/16 m of a
    point: 6.9/16.4/40.7/91.5 measured on the reference's saved clouds; this generator gives
    6.7/16.3/43.0/85.8): 50 % tight object-like sub-clusters, 42 % broad regions, 8 % uniform clutter."""
    reg, sub, s_reg, s_sub = scene
    n_reg, n_sub = reg.shape[1], sub.shape[1] // reg.shape[1]
    wr = torch.randint(0, n_reg, (B, N), generator=g)
    ws = wr * n_sub + torch.randint(0, n_sub, (B, N), generator=g)
    pr = torch.gather(reg, 1, wr.unsqueeze(-1).expand(B, N, 3)) + torch.randn(B, N, 3, generator=g) * s_reg
    ps = torch.gather(sub, 1, ws.unsqueeze(-1).expand(B, N, 3)) + torch.randn(B, N, 3, generator=g) * s_sub
    pu = box_lo + (box_hi - box_lo) * torch.rand(B, N, 3, generator=g)
    u = torch.rand(B, N, 1, generator=g)
    pts = torch.where(u < f_sub, ps, torch.where(u < f_sub + f_reg, pr, pu))
    return torch.maximum(torch.minimum(pts, box_hi), box_lo)


def _scene(g, B, lo, hi, n_reg, n_sub, s_reg, s_off, s_sub):
    reg = lo + (hi - lo) * torch.rand(B, n_reg, 3, generator=g)
    sub = (reg[:, :, None, :] + torch.randn(B, n_reg, n_sub, 3, generator=g) * torch.tensor(s_off)).reshape(B, n_reg * n_sub, 3)
    return reg, sub, torch.tensor(s_reg), torch.tensor(s_sub)


def rigid_transform(yaw_deg=0.5, t=(-0.5, 0.0, 0.0), B=1):
    a = math.radians(yaw_deg)
    T = torch.eye(4).repeat(B, 1, 1)
    T[:, 0, 0], T[:, 0, 1], T[:, 1, 0], T[:, 1, 1] = math.cos(a), -math.sin(a), math.sin(a), math.cos(a)
    T[:, 0, 3], T[:, 1, 3], T[:, 2, 3] = t
    return T


def make_batch(B, N=256, seed=1234, lidar=False, train_extras=False):
    """Synthetic two-frame batch in the reference's model layout.

    Returns a dict with pc1, pc2 (B,3,N), ft1, ft2 (B,3,N) fp32 contiguous -- what
    ``extract_data_info`` (main_util.py:21-36) hands to ``net(...)``.  With
    ``train_extras`` also gt_trans (B,4,4), flow_label (B,N,3), fg_mask (B,N), interval
    (B,), radar_u/v (B,N), opt_flow (B,N,2) for the loss path (main_util.py:63-72).
    """
    g = torch.Generator().manual_seed(seed)
    if lidar:   # config 5: LiDAR-like box, many more structures
        lo, hi = torch.tensor([0.0, -40.0, -3.0]), torch.tensor([70.0, 40.0, 1.0])
        scene = _scene(g, B, lo, hi, 24, 8, (6.0, 5.0, 0.6), (5.0, 5.0, 0.4), (0.8, 0.8, 0.3))
    else:       # radar: calibrated to VoD occupancy
        lo, hi = torch.tensor([2.0, -25.0, -3.0]), torch.tensor([90.0, 25.0, 3.0])
        scene = _scene(g, B, lo, hi, 3, 6, (6.0, 5.0, 0.8), (5.0, 5.0, 0.6), (0.8, 0.8, 0.3))
    p1 = _mixture(g, B, N, scene, lo, hi)
    T = rigid_transform(B=B)
    p2 = _mixture(g, B, N, scene, lo, hi)
    p2 = torch.einsum("bij,bnj->bni", T[:, :3, :3], p2) + T[:, None, :3, 3]
    p2 = p2 + 0.05 * torch.randn(B, N, 3, generator=g)

    def feats():
        vr = 2.0 * torch.randn(B, N, 1, generator=g)
        rcs = -20.0 + 40.0 * torch.rand(B, N, 1, generator=g)
        return torch.cat([vr, rcs, rcs], dim=-1)

    f1, f2 = feats(), feats()
    out = {
        "pc1": p1.transpose(2, 1).contiguous(), "pc2": p2.transpose(2, 1).contiguous(),
        "ft1": f1.transpose(2, 1).contiguous(), "ft2": f2.transpose(2, 1).contiguous(),
    }
    if train_extras:
        pc1 = out["pc1"]
        h = torch.cat([pc1, torch.ones(B, 1, N)], dim=1)
        flow = (T @ h)[:, :3] - pc1                                   # utils/util.py:184-189
        # ~10 % of the points belong to moving objects: non-rigid flow component, foreground label
        moving = torch.rand(B, N, generator=g) < 0.1
        flow = flow + moving.unsqueeze(1) * (0.3 * torch.randn(B, 3, N, generator=g))
        out["gt_trans"] = T
        out["flow_label"] = flow.transpose(2, 1).contiguous()
        out["fg_mask"] = torch.logical_not(moving | (torch.rand(B, N, generator=g) < 0.1)).float()
        out["interval"] = torch.full((B,), 0.1)
        P = torch.tensor(CAMERA_PROJECTION)
        Tcr = torch.tensor(T_CAMERA_RADAR)
        uvz = P.unsqueeze(0) @ (Tcr.unsqueeze(0) @ h)                   # utils/util.py:16-29
        out["radar_u"] = (uvz[:, 0] / uvz[:, 2]).contiguous()
        out["radar_v"] = (uvz[:, 1] / uvz[:, 2]).contiguous()
        out["opt_flow"] = torch.randn(B, N, 2, generator=g)
    return out


def synth_state_dict(manifest, seed=1234, calib=None):
    """Seeded weights for a ``{key: shape}`` manifest (ordered list of [key, shape, dtype]).

    BN affine and running statistics are randomised so eval-mode BN is not the identity
    (an identity BN would hide channel-layout bugs).  ``calib`` (path to an .npz or a dict)
    overlays BN running statistics measured on the reference model in train mode
    (tests/golden/make_golden.py: "BN calibration") so that eval-mode activations are O(1)
    like a trained network's.
    """
    if isinstance(calib, str):
        import numpy as np
        with np.load(calib) as z:
            calib = {k: torch.from_numpy(z[k]) for k in z.files}
    sd = {}
    for i, (key, shape, dtype) in enumerate(manifest):
        g = torch.Generator().manual_seed(seed * 1000003 + i)
        shape = tuple(shape)
        leaf = key.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            t = torch.zeros(shape, dtype=torch.int64)
        elif leaf == "running_mean":
            t = 0.2 * torch.randn(shape, generator=g)
        elif leaf == "running_var":
            t = 0.5 + torch.rand(shape, generator=g)
        elif len(shape) == 4:                                   # conv weight (out,in,1,1)
            t = torch.randn(shape, generator=g) * math.sqrt(2.0 / shape[1])
        elif "gru" in key:
            t = (torch.rand(shape, generator=g) * 2 - 1) / 16.0
        elif leaf == "weight":                                  # BN gamma
            t = 0.75 + 0.5 * torch.rand(shape, generator=g)
        elif leaf == "bias":
            t = 0.1 * torch.randn(shape, generator=g)
        else:
            raise KeyError(key)
        sd[key] = t if leaf == "num_batches_tracked" else t.float()
    if calib:
        for k, v in calib.items():
            assert k in sd and tuple(sd[k].shape) == tuple(v.shape), k
            sd[k] = v.float().clone()
    return sd
