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
    _lib.check(_lib.lib().cmf_group_points(*args), "cmf_group_points")
    return 1


def group_points_grad_wrapper(b, c, n, npoints, nsample, grad_out, idx, grad_points):
    err = _lib.lib().cmf_group_points_grad(b, c, n, npoints, nsample, _lib.dev_ptr(grad_out, _f32),
                                           _lib.dev_ptr(idx, _i32), _lib.dev_ptr(grad_points, _f32), _lib.stream_ptr())
    _lib.check(err, "cmf_group_points_grad")
    return 1


def _call(name, *args):
    _lib.check(getattr(_lib.lib(), name)(*args, _lib.stream_ptr()), name)
    return 1


def gather_points_wrapper(b, c, n, npoints, points, idx, out):
    """lib/src/sampling.cpp (gather_points_wrapper_fast)"""
    return _call("cmf_gather_points", b, c, n, npoints, _lib.dev_ptr(points, _f32), _lib.dev_ptr(idx, _i32), _lib.dev_ptr(out, _f32))


def gather_points_grad_wrapper(b, c, n, npoints, grad_out, idx, grad_points):
    return _call("cmf_gather_points_grad", b, c, n, npoints, _lib.dev_ptr(grad_out, _f32), _lib.dev_ptr(idx, _i32),
                 _lib.dev_ptr(grad_points, _f32))


def furthest_point_sampling_wrapper(b, n, m, points, temp, idx):
    return _call("cmf_furthest_point_sampling", b, n, m, _lib.dev_ptr(points, _f32), _lib.dev_ptr(temp, _f32), _lib.dev_ptr(idx, _i32))


def knn_wrapper(b, n, m, k, unknown, known, dist2, idx):
    """lib/src/interpolate.cpp (knn_wrapper_fast)"""
    return _call("cmf_knn_points", b, n, m, k, _lib.dev_ptr(unknown, _f32), _lib.dev_ptr(known, _f32),
                 _lib.dev_ptr(dist2, _f32), _lib.dev_ptr(idx, _i32))


def three_nn_wrapper(b, n, m, unknown, known, dist2, idx):
    return _call("cmf_three_nn", b, n, m, _lib.dev_ptr(unknown, _f32), _lib.dev_ptr(known, _f32),
                 _lib.dev_ptr(dist2, _f32), _lib.dev_ptr(idx, _i32))


def three_interpolate_wrapper(b, c, m, n, points, idx, weight, out):
    return _call("cmf_three_interpolate", b, c, m, n, _lib.dev_ptr(points, _f32), _lib.dev_ptr(idx, _i32),
                 _lib.dev_ptr(weight, _f32), _lib.dev_ptr(out, _f32))


def three_interpolate_grad_wrapper(b, c, n, m, grad_out, idx, weight, grad_points):
    return _call("cmf_three_interpolate_grad", b, c, n, m, _lib.dev_ptr(grad_out, _f32), _lib.dev_ptr(idx, _i32),
                 _lib.dev_ptr(weight, _f32), _lib.dev_ptr(grad_points, _f32))


# all ten entry points of lib/src/pointnet2_api.cpp:10-25
pointnet2_cuda = types.SimpleNamespace(
    ball_query_wrapper=ball_query_wrapper,
    group_points_wrapper=group_points_wrapper,
    group_points_grad_wrapper=group_points_grad_wrapper,
    gather_points_wrapper=gather_points_wrapper,
    gather_points_grad_wrapper=gather_points_grad_wrapper,
    furthest_point_sampling_wrapper=furthest_point_sampling_wrapper,
    knn_wrapper=knn_wrapper,
    three_nn_wrapper=three_nn_wrapper,
    three_interpolate_wrapper=three_interpolate_wrapper,
    three_interpolate_grad_wrapper=three_interpolate_grad_wrapper,
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


class QueryAndGroupFn(Function):
    """QueryAndGroup.forward (lib/pointnet2_utils.py:269-292) as ONE kernel launch -- cmf_query_and_group: ball query,
    grouped xyz relative to the centre, grouped features, concatenated in the reference's (B, 3 + C, npoint, nsample)
    layout.  Gradient w.r.t. the features only (GroupingOperation.backward, :208-222, over the feature planes of the
    incoming gradient); coordinates that require a gradient take the unfused path in QueryAndGroup.forward."""

    @staticmethod
    def forward(ctx, xyz, new_xyz, features, radius, nsample, use_xyz):
        assert xyz.is_contiguous() and new_xyz.is_contiguous()                     # :241-242
        B, N, _ = xyz.size()
        M = new_xyz.size(1)
        C = 0
        if features is not None:
            assert features.is_contiguous()                                        # :195
            C = features.size(1)
        ctot = (3 if (use_xyz or features is None) else 0) + C
        out = torch.empty(B, ctot, M, nsample, dtype=_f32, device=xyz.device)
        idx = torch.empty(B, M, nsample, dtype=_i32, device=xyz.device)
        _lib.check(_lib.lib().cmf_query_and_group(B, N, M, radius, nsample, C, int(ctot > C), _lib.dev_ptr(new_xyz, _f32),
                                                  _lib.dev_ptr(xyz, _f32), _lib.dev_ptr(features, _f32), _lib.dev_ptr(idx, _i32),
                                                  _lib.dev_ptr(out, _f32), _lib.stream_ptr()), "cmf_query_and_group")
        ctx.for_backwards = (idx, N, C, ctot)
        ctx.mark_non_differentiable(idx)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        idx, N, C, ctot = ctx.for_backwards
        if C == 0 or not ctx.needs_input_grad[2]:
            return None, None, None, None, None, None
        B, M, nsample = idx.size()
        g = grad_out[:, ctot - C:].contiguous()                                    # the feature planes
        grad_features = torch.zeros(B, C, N, dtype=_f32, device=grad_out.device)
        group_points_grad_wrapper(B, C, N, M, nsample, g, idx, grad_features)
        return None, None, grad_features, None, None, None


class QueryAndGroup(nn.Module):
    """lib/pointnet2_utils.py:259-292: -> (B, 3 + C, npoint, nsample), relative xyz first."""

    def __init__(self, radius: float, nsample: int, use_xyz: bool = True):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz

    fused = True            # one launch through cmf_query_and_group (False: the reference's op sequence over the three kernels)

    def forward(self, xyz: torch.Tensor, new_xyz: torch.Tensor, features: torch.Tensor = None):
        if features is None:
            assert self.use_xyz, "Cannot have not features and not use xyz as a feature!"
        if self.fused and not (xyz.requires_grad or new_xyz.requires_grad):
            return QueryAndGroupFn.apply(xyz, new_xyz, features, self.radius, self.nsample, self.use_xyz)
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


# ---- the reference's remaining Functions (lib/pointnet2_utils.py:10-181; unused by CMFlow) ----------------
class FurthestPointSampling(Function):
    """lib/pointnet2_utils.py:10-36"""

    @staticmethod
    def forward(ctx, xyz: torch.Tensor, npoint: int) -> torch.Tensor:
        assert xyz.is_contiguous()
        B, N, _ = xyz.size()
        output = torch.empty(B, npoint, dtype=_i32, device=xyz.device)
        temp = torch.full((B, N), 1e10, dtype=_f32, device=xyz.device)
        furthest_point_sampling_wrapper(B, N, npoint, xyz, temp, output)
        ctx.mark_non_differentiable(output)
        return output

    @staticmethod
    def backward(ctx, a=None):
        return None, None


furthest_point_sample = FurthestPointSampling.apply


class GatherOperation(Function):
    """lib/pointnet2_utils.py:39-72"""

    @staticmethod
    def forward(ctx, features: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
        assert features.is_contiguous()
        assert idx.is_contiguous()
        B, npoint = idx.size()
        _, C, N = features.size()
        output = torch.empty(B, C, npoint, dtype=_f32, device=features.device)
        gather_points_wrapper(B, C, N, npoint, features, idx, output)
        ctx.for_backwards = (idx, C, N)
        return output

    @staticmethod
    def backward(ctx, grad_out):
        idx, C, N = ctx.for_backwards
        B, npoint = idx.size()
        grad_features = torch.zeros(B, C, N, dtype=_f32, device=grad_out.device)
        gather_points_grad_wrapper(B, C, N, npoint, grad_out.contiguous(), idx, grad_features)
        return grad_features, None


gather_operation = GatherOperation.apply


class KNN(Function):
    """lib/pointnet2_utils.py:75-102 (returns sqrt of the squared distances, like the reference)"""

    @staticmethod
    def forward(ctx, k: int, unknown: torch.Tensor, known: torch.Tensor):
        assert unknown.is_contiguous()
        assert known.is_contiguous()
        B, N, _ = unknown.size()
        m = known.size(1)
        dist2 = torch.empty(B, N, k, dtype=_f32, device=unknown.device)
        idx = torch.empty(B, N, k, dtype=_i32, device=unknown.device)
        knn_wrapper(B, N, m, k, unknown, known, dist2, idx)
        ctx.mark_non_differentiable(idx)
        return torch.sqrt(dist2), idx

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None, None


knn = KNN.apply


class ThreeNN(Function):
    """lib/pointnet2_utils.py:104-133"""

    @staticmethod
    def forward(ctx, unknown: torch.Tensor, known: torch.Tensor):
        assert unknown.is_contiguous()
        assert known.is_contiguous()
        B, N, _ = unknown.size()
        m = known.size(1)
        dist2 = torch.empty(B, N, 3, dtype=_f32, device=unknown.device)
        idx = torch.empty(B, N, 3, dtype=_i32, device=unknown.device)
        three_nn_wrapper(B, N, m, unknown, known, dist2, idx)
        ctx.mark_non_differentiable(idx)
        return torch.sqrt(dist2), idx

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None


three_nn = ThreeNN.apply


class ThreeInterpolate(Function):
    """lib/pointnet2_utils.py:136-181"""

    @staticmethod
    def forward(ctx, features: torch.Tensor, idx: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
        assert features.is_contiguous()
        assert idx.is_contiguous()
        assert weight.is_contiguous()
        B, c, m = features.size()
        n = idx.size(1)
        ctx.three_interpolate_for_backward = (idx, weight, m)
        output = torch.empty(B, c, n, dtype=_f32, device=features.device)
        three_interpolate_wrapper(B, c, m, n, features, idx, weight, output)
        return output

    @staticmethod
    def backward(ctx, grad_out: torch.Tensor):
        idx, weight, m = ctx.three_interpolate_for_backward
        B, c, n = grad_out.size()
        grad_features = torch.zeros(B, c, m, dtype=_f32, device=grad_out.device)
        three_interpolate_grad_wrapper(B, c, n, m, grad_out.contiguous(), idx, weight, grad_features)
        return grad_features, None, None


three_interpolate = ThreeInterpolate.apply


class GroupAll(nn.Module):
    """lib/pointnet2_utils.py:295-318"""

    def __init__(self, use_xyz: bool = True):
        super().__init__()
        self.use_xyz = use_xyz

    def forward(self, xyz: torch.Tensor, new_xyz: torch.Tensor, features: torch.Tensor = None):
        grouped_xyz = xyz.transpose(1, 2).unsqueeze(2)
        if features is not None:
            grouped_features = features.unsqueeze(2)
            return torch.cat([grouped_xyz, grouped_features], dim=1) if self.use_xyz else grouped_features
        return grouped_xyz
