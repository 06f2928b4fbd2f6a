"""ctypes bindings of oracle/liboracle.so (cmf_oracle.c) on torch-CPU tensors.

TEST INFRASTRUCTURE ONLY -- the checker, never the thing measured or shipped.
Function names follow the reference's ``pointnet2_cuda`` export table
(lib/src/pointnet2_api.cpp:10-25) so the shimmed reference import in
tests/golden/make_golden.py can use this module as its fake ``pointnet2_cuda``.
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "cmf_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-B", "liboracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
    return _lib


def _p(t: torch.Tensor, dtype):
    assert t.device.type == "cpu" and t.dtype == dtype and t.is_contiguous(), (t.device, t.dtype)
    return ctypes.c_void_p(t.data_ptr())


_f, _i = torch.float32, torch.int32
_ci, _cf = ctypes.c_int, ctypes.c_float


# --- same argument order as the pybind wrappers (lib/src/*.cpp) ---------------------------
def ball_query_wrapper(b, n, m, radius, nsample, new_xyz, xyz, idx):
    lib().orc_ball_query(_ci(b), _ci(n), _ci(m), _cf(radius), _ci(nsample),
                         _p(new_xyz, _f), _p(xyz, _f), _p(idx, _i))
    return 1


def group_points_wrapper(b, c, n, npoints, nsample, points, idx, out):
    lib().orc_group_points(_ci(b), _ci(c), _ci(n), _ci(npoints), _ci(nsample),
                           _p(points, _f), _p(idx, _i), _p(out, _f))
    return 1


def group_points_grad_wrapper(b, c, n, npoints, nsample, grad_out, idx, grad_points):
    lib().orc_group_points_grad(_ci(b), _ci(c), _ci(n), _ci(npoints), _ci(nsample),
                                _p(grad_out, _f), _p(idx, _i), _p(grad_points, _f))
    return 1


def gather_points_wrapper(b, c, n, npoints, points, idx, out):
    lib().orc_gather_points(_ci(b), _ci(c), _ci(n), _ci(npoints), _p(points, _f), _p(idx, _i), _p(out, _f))
    return 1


def gather_points_grad_wrapper(b, c, n, npoints, grad_out, idx, grad_points):
    lib().orc_gather_points_grad(_ci(b), _ci(c), _ci(n), _ci(npoints), _p(grad_out, _f), _p(idx, _i),
                                 _p(grad_points, _f))
    return 1


def three_nn_wrapper(b, n, m, unknown, known, dist2, idx):
    lib().orc_three_nn(_ci(b), _ci(n), _ci(m), _p(unknown, _f), _p(known, _f), _p(dist2, _f), _p(idx, _i))
    return 1


def three_interpolate_wrapper(b, c, m, n, points, idx, weight, out):
    lib().orc_three_interpolate(_ci(b), _ci(c), _ci(m), _ci(n), _p(points, _f), _p(idx, _i),
                                _p(weight, _f), _p(out, _f))
    return 1


def knn_wrapper(b, n, m, k, unknown, known, dist2, idx):
    lib().orc_knn_points(_ci(b), _ci(n), _ci(m), _ci(k), _p(unknown, _f), _p(known, _f), _p(dist2, _f), _p(idx, _i))
    return 1


def three_interpolate_grad_wrapper(b, c, n, m, grad_out, idx, weight, grad_points):
    lib().orc_three_interpolate_grad(_ci(b), _ci(c), _ci(n), _ci(m), _p(grad_out, _f), _p(idx, _i), _p(weight, _f),
                                     _p(grad_points, _f))
    return 1


def furthest_point_sampling_wrapper(b, n, m, dataset, temp, idxs):
    lib().orc_furthest_point_sampling(_ci(b), _ci(n), _ci(m), _p(dataset, _f), _p(temp, _f), _p(idxs, _i))
    return 1


# --- tensor-level conveniences -------------------------------------------------------------
def ball_query(radius, nsample, xyz, new_xyz):
    """lib/pointnet2_utils.py:231-249: xyz (B,N,3), new_xyz (B,M,3) -> idx (B,M,nsample) int32."""
    B, N, _ = xyz.shape
    M = new_xyz.shape[1]
    idx = torch.zeros(B, M, nsample, dtype=_i)
    ball_query_wrapper(B, N, M, radius, nsample, new_xyz.contiguous(), xyz.contiguous(), idx)
    return idx


def group_points(points, idx):
    """lib/pointnet2_utils.py:187-205: points (B,C,N), idx (B,P,S) -> (B,C,P,S)."""
    B, C, N = points.shape
    _, P, S = idx.shape
    out = torch.empty(B, C, P, S, dtype=_f)
    group_points_wrapper(B, C, N, P, S, points.contiguous(), idx.contiguous(), out)
    return out


def group_points_grad(grad_out, idx, N):
    B, C, P, S = grad_out.shape
    g = torch.zeros(B, C, N, dtype=_f)
    group_points_grad_wrapper(B, C, N, P, S, grad_out.contiguous(), idx.contiguous(), g)
    return g


def square_distance(src, dst):
    """radarflow_util.py:8-30 in canonical arithmetic: (B,N,3),(B,M,3) -> (B,N,M)."""
    B, N, _ = src.shape
    M = dst.shape[1]
    out = torch.empty(B, N, M, dtype=_f)
    lib().orc_square_distance(_ci(B), _ci(N), _ci(M), _p(src.contiguous(), _f), _p(dst.contiguous(), _f),
                              _p(out, _f))
    return out


def knn(nsample, xyz, new_xyz, return_dist=False):
    """radarflow_util.py:88-99 (knn_point), canonical order: -> idx (B,S,nsample) int32."""
    B, N, _ = xyz.shape
    S = new_xyz.shape[1]
    idx = torch.empty(B, S, nsample, dtype=_i)
    dist = torch.empty(B, S, nsample, dtype=_f)
    lib().orc_knn(_ci(B), _ci(N), _ci(S), _ci(nsample), _p(xyz.contiguous(), _f),
                  _p(new_xyz.contiguous(), _f), _p(idx, _i), _p(dist, _f))
    return (idx, dist) if return_dist else idx
