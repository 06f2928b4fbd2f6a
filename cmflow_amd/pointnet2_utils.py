"""Host-side mirror of the reference's op wrappers (lib/pointnet2_utils.py) over the HIP C-ABI.

Same names, argument order, dtypes, contiguity asserts and autograd behaviour as the
reference's ``BallQuery`` (:228-253), ``GroupingOperation`` (:184-225) and ``QueryAndGroup``
(:259-292); the native calls go to libcmflow_hip.so instead of ``pointnet2_cuda``.
``pointnet2_cuda`` (below) is a module-like object with the reference extension's
``*_wrapper`` entry points (lib/src/pointnet2_api.cpp:10-25) for callers that import the
extension directly.
"""
import types
from typing import Tuple

import torch
import torch.nn as nn
from torch.autograd import Function

from . import _lib

_f32, _i32 = torch.float32, torch.int32


# ---- pointnet2_cuda-compatible wrappers (tensor arguments, write in place, return 1) ---------
def ball_query_wrapper(b, n, m, radius, nsample, new_xyz, xyz, idx):
    """lib/src/ball_query.cpp:14-25 (note: new_xyz before xyz)."""
    err = _lib.lib().cmf_ball_query(b, n, m, radius, nsample, _lib.dev_ptr(new_xyz, _f32), _lib.dev_ptr(xyz, _f32),
                                    _lib.dev_ptr(idx, _i32), _lib.stream_ptr())
    _lib.check(err, "cmf_ball_query")
    return 1


def group_points_wrapper(b, c, n, npoints, nsample, points, idx, out):
    """lib/src/group_points.cpp (group_points_wrapper_fast)."""
    args = (b, c, n, npoints, nsample, _lib.dev_ptr(points, _f32), _lib.dev_ptr(idx, _i32), _lib.dev_ptr(out, _f32),
            _lib.stream_ptr())
    # algorithmic bytes: features once + idx once + grouped output once
    nbytes = 4.0 * (b * c * n + b * npoints * nsample + b * c * npoints * nsample)
    err = _lib.tracked("cmf_group_points", nbytes, lambda: _lib.lib().cmf_group_points(*args))
    _lib.check(err, "cmf_group_points")
    return 1


def group_points_grad_wrapper(b, c, n, npoints, nsample, grad_out, idx, grad_points):
    err = _lib.lib().cmf_group_points_grad(b, c, n, npoints, nsample, _lib.dev_ptr(grad_out, _f32),
                                           _lib.dev_ptr(idx, _i32), _lib.dev_ptr(grad_points, _f32), _lib.stream_ptr())
    _lib.check(err, "cmf_group_points_grad")
    return 1


pointnet2_cuda = types.SimpleNamespace(
    ball_query_wrapper=ball_query_wrapper,
    group_points_wrapper=group_points_wrapper,
    group_points_grad_wrapper=group_points_grad_wrapper,
)


# ---- autograd Functions ----------------------------------------------------------------------
class GroupingOperation(Function):
    """lib/pointnet2_utils.py:184-225"""

    @staticmethod
    def forward(ctx, features: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
        assert features.is_contiguous()
        assert idx.is_contiguous()
        idx = idx.int()
        B, nfeatures, nsample = idx.size()
        _, C, N = features.size()
        output = torch.empty(B, C, nfeatures, nsample, dtype=_f32, device=features.device)
        group_points_wrapper(B, C, N, nfeatures, nsample, features, idx, output)
        ctx.for_backwards = (idx, N)
        return output

    @staticmethod
    def backward(ctx, grad_out: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        idx, N = ctx.for_backwards
        B, C, npoint, nsample = grad_out.size()
        grad_features = torch.zeros(B, C, N, dtype=_f32, device=grad_out.device)
        group_points_grad_wrapper(B, C, N, npoint, nsample, grad_out.contiguous(), idx, grad_features)
        return grad_features, None


grouping_operation = GroupingOperation.apply


class BallQuery(Function):
    """lib/pointnet2_utils.py:228-253"""

    @staticmethod
    def forward(ctx, radius: float, nsample: int, xyz: torch.Tensor, new_xyz: torch.Tensor) -> torch.Tensor:
        assert new_xyz.is_contiguous()
        assert xyz.is_contiguous()
        B, N, _ = xyz.size()
        npoint = new_xyz.size(1)
        idx = torch.zeros(B, npoint, nsample, dtype=_i32, device=xyz.device)
        ball_query_wrapper(B, N, npoint, radius, nsample, new_xyz, xyz, idx)
        ctx.mark_non_differentiable(idx)
        return idx

    @staticmethod
    def backward(ctx, a=None):
        return None, None, None, None


ball_query = BallQuery.apply


class QueryAndGroup(nn.Module):
    """lib/pointnet2_utils.py:259-292: -> (B, 3 + C, npoint, nsample), relative xyz first."""

    def __init__(self, radius: float, nsample: int, use_xyz: bool = True):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz

    def forward(self, xyz: torch.Tensor, new_xyz: torch.Tensor, features: torch.Tensor = None):
        idx = ball_query(self.radius, self.nsample, xyz, new_xyz)
        xyz_trans = xyz.transpose(1, 2).contiguous()
        grouped_xyz = grouping_operation(xyz_trans, idx)
        grouped_xyz = grouped_xyz - new_xyz.transpose(1, 2).unsqueeze(-1)
        if features is not None:
            grouped_features = grouping_operation(features, idx)
            if self.use_xyz:
                return torch.cat([grouped_xyz, grouped_features], dim=1)
            return grouped_features
        assert self.use_xyz, "Cannot have not features and not use xyz as a feature!"
        return grouped_xyz
