"""PointNet++ set-abstraction / feature-propagation modules -- mirror of ``lib/pointnet2_modules.py``
(_PointnetSAModuleBase :9-54, PointnetSAModuleMSG :57-91, PointnetSAModule :94-112, PointnetFPModule :115-155)
on top of the HIP ops of ``cmflow_amd.pointnet2_utils`` (furthest-point sampling, gather, ball query + grouping,
three_nn + three_interpolate).  CMFlow itself does not use them; they complete the drop-in
surface for PointNet++-style callers.  Same constructor arguments, forward contract and state_dict keys.
"""
from typing import List

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import pointnet2_utils
from . import pytorch_utils as pt_utils


class _PointnetSAModuleBase(nn.Module):

    def __init__(self):
        super().__init__()
        self.npoint = None
        self.groupers = None
        self.mlps = None
        self.pool_method = 'max_pool'

    def forward(self, xyz: torch.Tensor, features: torch.Tensor = None, new_xyz=None):
        """xyz (B,N,3), features (B,C,N) -> new_xyz (B,npoint,3), new_features (B, sum_k mlps[k][-1], npoint)"""
        if new_xyz is None and self.npoint is not None:
            centres = pointnet2_utils.furthest_point_sample(xyz, self.npoint)
            new_xyz = pointnet2_utils.gather_operation(xyz.transpose(1, 2).contiguous(), centres).transpose(1, 2).contiguous()
        pooled = []
        for grouper, mlp in zip(self.groupers, self.mlps):
            x = mlp(grouper(xyz, new_xyz, features))                    # (B, mlp[-1], npoint, nsample)
            if self.pool_method == 'max_pool':
                x = F.max_pool2d(x, kernel_size=[1, x.size(3)])
            elif self.pool_method == 'avg_pool':
                x = F.avg_pool2d(x, kernel_size=[1, x.size(3)])
            else:
                raise NotImplementedError
            pooled.append(x.squeeze(-1))
        return new_xyz, torch.cat(pooled, dim=1)


class PointnetSAModuleMSG(_PointnetSAModuleBase):
    """Set abstraction with multi-scale grouping (lib/pointnet2_modules.py:57-91).  As in the reference the first
    entry of every mlp spec is incremented in place by 3 when use_xyz is set."""

    def __init__(self, *, npoint: int, radii: List[float], nsamples: List[int], mlps: List[List[int]], bn: bool = True,
                 use_xyz: bool = True, pool_method='max_pool', instance_norm=False):
        super().__init__()
        assert len(radii) == len(nsamples) == len(mlps)
        self.npoint = npoint
        self.groupers = nn.ModuleList()
        self.mlps = nn.ModuleList()
        for radius, nsample, spec in zip(radii, nsamples, mlps):
            self.groupers.append(pointnet2_utils.QueryAndGroup(radius, nsample, use_xyz=use_xyz)
                                 if npoint is not None else pointnet2_utils.GroupAll(use_xyz))
            if use_xyz:
                spec[0] += 3
            self.mlps.append(pt_utils.SharedMLP(spec, bn=bn, instance_norm=instance_norm))
        self.pool_method = pool_method


class PointnetSAModule(PointnetSAModuleMSG):
    """Single-scale set abstraction (lib/pointnet2_modules.py:94-112)"""

    def __init__(self, *, mlp: List[int], npoint: int = None, radius: float = None, nsample: int = None,
                 bn: bool = True, use_xyz: bool = True, pool_method='max_pool', instance_norm=False):
        super().__init__(mlps=[mlp], npoint=npoint, radii=[radius], nsamples=[nsample], bn=bn, use_xyz=use_xyz,
                         pool_method=pool_method, instance_norm=instance_norm)


class PointnetFPModule(nn.Module):
    """Feature propagation (lib/pointnet2_modules.py:115-155): inverse-distance interpolation from the 3 nearest
    known points, concatenated with the skip features, through a shared MLP."""

    def __init__(self, *, mlp: List[int], bn: bool = True):
        super().__init__()
        self.mlp = pt_utils.SharedMLP(mlp, bn=bn)

    def forward(self, unknown, known, unknow_feats, known_feats):
        """unknown (B,n,3), known (B,m,3), unknow_feats (B,C1,n), known_feats (B,C2,m) -> (B, mlp[-1], n)"""
        if known is not None:
            dist, idx = pointnet2_utils.three_nn(unknown, known)
            recip = 1.0 / (dist + 1e-8)
            weight = recip / torch.sum(recip, dim=2, keepdim=True)
            interpolated = pointnet2_utils.three_interpolate(known_feats, idx, weight)
        else:
            interpolated = known_feats.expand(*known_feats.size()[0:2], unknown.size(1))
        x = interpolated if unknow_feats is None else torch.cat([interpolated, unknow_feats], dim=1)
        return self.mlp(x.unsqueeze(-1)).squeeze(-1)
