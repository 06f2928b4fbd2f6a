"""Test fixture: the oracle's reference-layout modules (oracle/cmflow_oracle.py -- the reference's own op sequence on
(B,C,N) / (B,C,N,ns) tensors) bound to the HIP DROP-IN kernels instead of the C oracle.

This is how the reference itself would use libcmflow_hip.so: its Python stays as it is and the three `pointnet2_cuda`
entry points it calls (ball_query / group_points / group_points_grad, lib/pointnet2_utils.py:184-253) plus the kNN go
to the C-ABI through cmflow_amd.pointnet2_utils.  `reference_layout_net(...)` returns such a network on the GPU; model
level tests then compare it with the goldens like the fused product path ("ref" in the path parametrisations).
"""
import torch

from cmflow_amd import pointnet2_utils as pu
from cmflow_amd import radarflow_util as ru
from oracle import cmflow_oracle as O


class HipOps:
    """The four functions oracle.cmflow_oracle takes from oracle.ops, over the HIP kernels (CUDA tensors)."""

    @staticmethod
    def ball_query(radius, nsample, xyz, new_xyz):
        return pu.ball_query(radius, nsample, xyz.contiguous(), new_xyz.contiguous())

    @staticmethod
    def group_points(points, idx):
        B, C, N = points.shape
        _, P, S = idx.shape
        out = torch.empty(B, C, P, S, dtype=torch.float32, device=points.device)
        pu.group_points_wrapper(B, C, N, P, S, points.contiguous(), idx.contiguous(), out)
        return out

    @staticmethod
    def group_points_grad(grad_out, idx, N):
        B, C, P, S = grad_out.shape
        g = torch.zeros(B, C, N, dtype=torch.float32, device=grad_out.device)      # caller zero-fills (pointnet2_utils.py:218)
        pu.group_points_grad_wrapper(B, C, N, P, S, grad_out.contiguous(), idx.contiguous(), g)
        return g

    @staticmethod
    def knn(nsample, xyz, new_xyz, return_dist=False):
        return ru.knn_point(nsample, xyz, new_xyz, return_dist)


def reference_layout_net(cls, args, state_dict, dev, monkeypatch):
    """An oracle model (cls = O.CMFlow / O.CMFlow_T / O.RaFlow) on `dev` whose native ops are the HIP drop-in kernels for
    the rest of the test (monkeypatch restores the C oracle afterwards; backward needs the binding too).  `.last` holds
    the intermediate features of the latest forward like the product model's, `.path` reads "ref"."""
    monkeypatch.setattr(O, "ops", HipOps)
    net = cls(args)
    net.load_state_dict(state_dict)
    for m in net.modules():
        if isinstance(m, O.WeightNet):                  # BN tensors the forward never touches: no gradient, optimizer skips
            for p in m.mlp_bns.parameters():
                p._cmf_unused = True
    net = net.to(dev)
    net.path = "ref"
    return net
