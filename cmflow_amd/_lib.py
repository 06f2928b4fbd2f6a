"""ctypes loader for libcmflow_hip.so (the C-ABI declared in include/cmflow_hip.h).

The product path has NO fallback: if the shared library is missing or a kernel launch
fails, an exception is raised.  Nothing here imports ``oracle/``.
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# CMF_LIB: kernel-diagnostic builds of the same library (tools/gemm_variants.py); the product uses the in-tree path
SO_PATH = os.environ.get("CMF_LIB") or os.path.join(_HERE, "libcmflow_hip.so")

_vp, _ci, _cf, _ll = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_longlong

# name -> argtypes (restype is int = hipError_t unless noted); mirrors include/cmflow_hip.h
SIGNATURES = {
    "cmf_ball_query": [_ci, _ci, _ci, _cf, _ci, _vp, _vp, _vp, _vp],
    "cmf_ball_query_multi": [_ci, _ci, _ci, _ci, _vp, _vp, _ci, _vp, _vp, _vp, _ci, _vp],
    "cmf_setconv_queries": [_ci, _vp, _vp],
    "cmf_group_points": [_ci, _ci, _ci, _ci, _ci, _vp, _vp, _vp, _vp],
    "cmf_group_points_grad": [_ci, _ci, _ci, _ci, _ci, _vp, _vp, _vp, _vp],
    "cmf_query_and_group": [_ci, _ci, _ci, _cf, _ci, _ci, _ci, _vp, _vp, _vp, _vp, _vp, _vp],
    "cmf_knn": [_ci, _ci, _ci, _ci, _vp, _vp, _vp, _vp, _vp],
    "cmf_pad_rows": [_ll, _ci, _vp, _ll, _vp, _ci, _vp],
    "cmf_inputs_point_major": [_ci, _ci, _ci, _ci, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "cmf_rel_xyz": [_ci, _ci, _ci, _ci, _vp, _vp, _vp, _vp, _vp],
    "cmf_weighted_kabsch": [_ci, _ci, _vp, _vp, _vp, _vp, _vp, _vp],
    "cmf_weighted_kabsch_grad": [_ci, _ci, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "cmf_ego_refine": [_ci, _ci, _cf, _cf, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "cmf_ego_refine_grad": [_ci, _ci, _cf, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "cmf_group_rows": [_ci, _ci, _ci, _ci, _ci, _vp, _vp, _vp, _vp],
    "cmf_build_inverse": [_ci, _ci, _ci, _vp, _vp, _vp, _vp],
    "cmf_build_inverse_ps": [_ci, _ci, _ci, _ci, _vp, _vp, _vp, _vp],
    "cmf_group_rows_grad": [_ci, _ci, _ci, _ci, _ci, _ci, _vp, _vp, _vp, _vp, _vp],
    "cmf_gemm": [_ci, _ci, _ci, _ci, _ci, _vp, _ll, _vp, _ll, _vp, _ll, _vp, _vp, _vp, _vp, _vp, _ci, _vp,
                 _ci, _vp, _ll, _vp, _vp, _vp, _vp, _vp, _ci, _vp, _ci, _vp],
    "cmf_gemm_dw_bn_bwd": [_ci, _ci, _ll, _vp, _ll, _vp, _ll, _vp, _vp, _vp, _vp, _vp, _ll, _vp, _ll, _vp, _vp, _vp, _ll, _ci, _vp,
                           _ci, _vp],
    "cmf_gemm_tiles_m": [_ci],
    "cmf_gemm_persist_config": [_ci, _ci],
    "cmf_thin_general": [_ci],
    "cmf_gemm_trace_arm": [],
    "cmf_gemm_trace_read": [_vp, _ll],
    "cmf_gemm_profile_begin": [ctypes.c_double],
    "cmf_gemm_profile_sampling": [_ci],
    "cmf_gemm_profile_eligible": [_vp, _vp],
    "cmf_gemm_profile_end": [_vp, _vp, _vp, _vp, _vp],
    "cmf_gemm_profile_records": [_vp, _ll],
    "cmf_setconv_sizes": [_vp, _vp, _vp, _vp],
    "cmf_setconv_forward": [_vp, _vp],
    "cmf_setconv_backward": [_vp, _vp],
    "cmf_setconv_bn_offsets": [_vp, _vp],
    "cmf_mlp_sizes": [_vp, _vp, _vp, _vp],
    "cmf_mlp_forward": [_vp, _vp],
    "cmf_mlp_backward": [_vp, _vp],
    "cmf_bn_running_update": [_ci, _vp, _ci, _vp, _vp, _vp],
    "cmf_setconv_forward_multi": [_ci, _vp, _vp],
    "cmf_setconv_backward_multi": [_ci, _vp, _vp],
    "cmf_setconv_forward_heads_multi": [_ci, _vp, _vp],
    "cmf_setconv_forward_bodies_batched": [_ci, _vp],
    "cmf_setconv_backward_bodies_batched": [_ci, _vp],
    "cmf_setconv_tail_forward": [_ci, _vp, _vp],
    "cmf_setconv_tail_backward": [_ci, _vp, _vp],
    "cmf_setconv_backward_bodies_multi": [_ci, _vp, _vp],
    "cmf_gather_points": [_ci, _ci, _ci, _ci, _vp, _vp, _vp, _vp],
    "cmf_gather_points_grad": [_ci, _ci, _ci, _ci, _vp, _vp, _vp, _vp],
    "cmf_furthest_point_sampling": [_ci, _ci, _ci, _vp, _vp, _vp, _vp],
    "cmf_knn_points": [_ci, _ci, _ci, _ci, _vp, _vp, _vp, _vp, _vp],
    "cmf_three_nn": [_ci, _ci, _ci, _vp, _vp, _vp, _vp, _vp],
    "cmf_three_interpolate": [_ci, _ci, _ci, _ci, _vp, _vp, _vp, _vp, _vp],
    "cmf_three_interpolate_grad": [_ci, _ci, _ci, _ci, _vp, _vp, _vp, _vp, _vp],
    "cmf_bn_finalize": [_ci, _ci, ctypes.c_double, _vp, _vp, _vp, _cf, _cf, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "cmf_colsum_finalize": [_ci, _ci, _vp, _vp, _vp, _vp, _vp],
    "cmf_group_affine": [_ci, _ci, _ci, _ci, _ci, _vp, _ci, _vp, _ci, _vp, _vp, _vp, _ci, _vp, _ci, _vp, _vp, _vp, _vp, _vp],
    "cmf_adam_step": [_ci, _vp, _vp, _ll, _vp, _vp, _vp, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, _ll, _vp],
    "cmf_group_prep": [_ci, _ci, _ci, _ci, _ci, _vp, _vp, _vp, _ci, _vp, _vp, _vp, _vp, _vp],
    "cmf_gemm_gather_affine": [_ci, _ci, _ci, _vp, _ll, _vp, _vp, _vp, _vp, _vp, _vp, _ll, _vp, _ll, _vp, _vp],
    "cmf_gemm_dx_gather": [_ci, _ci, _ci, _vp, _ll, _vp, _ll, _vp, _ll, _vp, _ll, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "cmf_group_perm": [_ci, _ci, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "cmf_gemm_dx_gather_sum": [_ci, _ci, _ci, _vp, _ll, _vp, _ll, _vp, _ll, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "cmf_group_rows_grad_bn_cf_pieces": [_ci, _ci, _ci, _ci, _ci, _vp, _vp, _ll, _vp, _ll, _vp, _vp, _vp, _vp, _vp, _vp, _cf, _vp, _vp, _vp,
                                         _ci, _vp],
    "cmf_gemm_dw_gather_split": [_ci, _ci, _ll],
    "cmf_gemm_dw_gather_bn_bwd": [_ci, _ci, _ll, _vp, _ll, _vp, _ll, _vp, _vp, _vp, _vp, _vp, _ll, _vp, _ll, _vp, _vp, _vp, _vp, _vp, _vp, _ll,
                                  _ci, _vp, _ci, _vp],
    "cmf_gemm_dw_gather": [_ci, _ci, _ll, _vp, _ll, _vp, _ll, _vp, _vp, _vp, _vp, _vp, _vp, _ll, _ci, _vp, _ci, _vp],
    "cmf_colsum": [_ci, _ci, _vp, _vp, _ci, _vp, _vp, _vp],
    "cmf_setconv_dwx": [_ci, _cf, _ci, _vp, _vp, _vp, _vp, _vp, _vp, _ci, _ci, _vp],
    "cmf_group_rows_grad_bn": [_ci, _ci, _ci, _ci, _vp, _vp, _vp, _vp, _vp, _vp, _cf, _vp, _vp, _vp, _ci, _vp],
    "cmf_group_rows_grad_bn_cf": [_ci, _ci, _ci, _ci, _ci, _vp, _vp, _ll, _vp, _ll, _vp, _vp, _vp, _vp, _vp, _vp, _cf, _vp, _vp, _vp,
                                  _ci, _vp],
    "cmf_bn_relu_maxpool": [_ll, _ci, _ci, _vp, _vp, _vp, _vp, _ll, _vp, _vp],
    "cmf_maxpool_bwd": [_ll, _ci, _ci, _vp, _ll, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "cmf_affine_relu": [_ll, _ci, _vp, _ll, _vp, _vp, _vp, _ll, _vp],
    "cmf_act_bwd_stats": [_ll, _ci, _vp, _ll, _vp, _ll, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "cmf_bn_bwd_apply": [_ll, _ci, _vp, _vp, _ll, _vp, _vp, _vp, _vp, _vp],
    "cmf_maxpool_bwd_point": [_ll, _ci, _ci, _vp, _ll, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "cmf_thin_bwd_layer_pooled": [_ll, _ci, _ci, _ci, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                  _ci, _vp, _vp],
    "cmf_thin_bwd_wide_supported": [_ci, _ci],
    "cmf_thin_bwd_wide_slabs": [_ll, _ci, _vp],
    "cmf_thin_bwd_wide_layer": [_ll, _ci, _vp, _ll, _vp, _vp, _ci, _vp, _ll, _vp, _vp, _vp, _vp, _vp, _ll, _vp, _ll,
                                _vp, _vp, _vp, _vp, _vp, _ll, _vp, _vp, _ll, _ci, _vp, _vp],
    "cmf_thin_bwd_supported": [_ci, _ci],
    "cmf_thin_bwd_slabs": [_ll, _vp],
    "cmf_thin_bwd_layer": [_ll, _ci, _ci, _vp, _ll, _vp, _ll, _vp, _vp, _vp, _vp, _vp, _ll, _vp, _ll, _ci,
                           _vp, _vp, _vp, _vp, _vp, _vp, _ll, _vp, _vp, _ll, _ci, _vp, _vp],
    "cmf_weighted_ksum": [_ll, _ci, _ci, _ci, _ci, _vp, _vp, _vp, _vp, _vp],
    "cmf_weighted_ksum_grad_tiles": [_ci],
    "cmf_weighted_ksum_grad": [_ll, _ci, _ci, _ci, _ci, _ci, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "cmf_weightnet_ksum_tiles": [_ci],
    "cmf_weightnet_ksum": [_ll, _ci, _ci, _ci, _ci, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "cmf_weightnet_ksum_grad": [_ll, _ci, _ci, _ci, _ci, _ci, _vp, _ll, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "cmf_global_max_cat": [_ci, _ci, _ci, _vp, _ll, _vp, _ll, _vp, _vp],
    "cmf_stack_first_conv": [_ci, _ci, _ci, _ci, _ci, _vp, _vp, _vp],
    "cmf_unstack_first_conv_grad": [_ci, _ci, _ci, _ci, _ci, _vp, _vp, _vp],
    "cmf_global_max_cat_grad": [_ci, _ci, _ci, _vp, _ll, _vp, _vp, _ll, _vp],
    "cmf_mem_stats": [],
    "cmf_radar_loss_workspace": [_ci, _ci],
    "cmf_radar_loss_workspace_nb": [_ci, _ci, _ci],
    "cmf_radar_loss_workspace_tiled": [_ci, _ci, _ci],
    "cmf_radar_loss": [_vp, _vp],
    "cmf_radar_loss_tiled": [_vp, _vp],
    "cmf_pseudo_labels": [_ci, _ci, _vp, _vp, _vp, _vp, _vp, _vp, _cf, _vp, _vp, _vp, _vp],
    "cmf_debug_spin": [_cf, _vp],
    "cmf_eval_metrics": [_ci, _ci, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _cf, _cf, _cf, _vp, _vp, _vp],
}
RESTYPES = {"cmf_mem_stats": _ll, "cmf_radar_loss_workspace": _ll, "cmf_radar_loss_workspace_nb": _ll, "cmf_radar_loss_workspace_tiled": _ll, "cmf_gemm_trace_read": _ll, "cmf_gemm_profile_records": _ll}


class GemmLaunchRecord(ctypes.Structure):
    """cmf_gemm_launch_record of include/cmflow_hip.h"""
    _fields_ = [("M", _ci), ("N", _ci), ("K", _ci), ("layout", _ci), ("split_k", _ci), ("kind", _ci), ("bm", _ci), ("bn", _ci),
                ("ms", _cf)]


class SetConvDesc(ctypes.Structure):
    """cmf_setconv_desc of include/cmflow_hip.h"""
    _fields_ = [("B", _ci), ("N", _ci), ("S", _ci), ("O1", _ci), ("C", _ci * 5), ("radius", _cf), ("training", _ci),
                ("eps", _cf * 6), ("momentum", _cf * 6),
                ("xyz", _vp), ("y", _vp), ("ldy", _ll), ("wx", _vp), ("ldwx", _ll), ("w", _vp * 5),
                ("gamma", _vp * 6), ("beta", _vp * 6), ("rmean", _vp * 6), ("rvar", _vp * 6), ("nbt", _vp * 6),
                ("saved", _vp), ("scratch", _vp), ("out", _vp), ("ldo", _ll),
                ("dout", _vp), ("lddout", _ll), ("dy", _vp), ("lddy", _ll), ("dwx", _vp), ("lddwx", _ll), ("acc_wx", _ci),
                ("dw", _vp * 5), ("acc_w", _ci * 5), ("dgamma", _vp * 6), ("dbeta", _vp * 6), ("acc_bn", _ci * 6),
                ("inference", _ci), ("idx_ready", _ci)]


class MlpDesc(ctypes.Structure):
    """cmf_mlp_desc of include/cmflow_hip.h"""
    _fields_ = [("M", _ll), ("L", _ci), ("C", _ci * 5), ("training", _ci), ("eps", _cf * 4), ("momentum", _cf * 4),
                ("x", _vp), ("ldx", _ll), ("w", _vp * 4), ("gamma", _vp * 4), ("beta", _vp * 4),
                ("rmean", _vp * 4), ("rvar", _vp * 4), ("nbt", _vp * 4), ("saved", _vp), ("scratch", _vp),
                ("out", _vp), ("ldo", _ll), ("dout", _vp), ("lddout", _ll), ("dx", _vp), ("lddx", _ll),
                ("dw", _vp * 4), ("acc_w", _ci * 4), ("dgamma", _vp * 4), ("dbeta", _vp * 4), ("acc_bn", _ci * 4)]


class BnUpdateEntry(ctypes.Structure):
    """cmf_bn_update_entry of include/cmflow_hip.h"""
    _fields_ = [("rmean", _vp), ("rvar", _vp), ("nbt", _vp), ("C", _ci), ("momentum", _cf), ("eps", _cf),
                ("count", ctypes.c_double), ("offset", _ll)]


class RadarLossDesc(ctypes.Structure):
    """cmf_radar_loss_desc of include/cmflow_hip.h"""
    _fields_ = [("B", _ci), ("N", _ci),
                ("pc1", _vp), ("pc2", _vp), ("pred_f", _vp), ("gt_f", _vp),
                ("vel1", _vp), ("mseg_pre", _vp), ("mseg_gt", _vp), ("dyn_mask", _vp), ("radar_u", _vp), ("radar_v", _vp),
                ("opt", _vp), ("pre_trans", _vp), ("gt_trans", _vp), ("camera_inverse", _vp), ("t_camera_radar", _vp),
                ("w_self", _cf), ("w_em", _cf), ("w_ms", _cf), ("w_opt", _cf), ("w_dyn", _cf),
                ("zeta", _cf), ("alpha", _cf), ("num_nb", _ci), ("lower_bound", _cf), ("self_only", _ci),
                ("items", _vp), ("d_pred_f", _vp), ("d_pre_trans", _vp), ("d_mseg_pre", _vp), ("workspace", _vp)]


def build(force: bool = False) -> str:
    """Compile the HIP extension in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    args = ["make", "-C", os.path.join(_HERE, "csrc"), "-j4"]
    if force:
        args.append("-B")
    subprocess.run(args, check=True, stdout=subprocess.DEVNULL)
    return SO_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise RuntimeError(
                "cmflow_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU/PyTorch fallback for the HIP path)" % SO_PATH)
        _lib = ctypes.CDLL(SO_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(_lib, name)
            fn.argtypes = argtypes
            fn.restype = RESTYPES.get(name, _ci)
        _lib.cmf_version.restype = ctypes.c_char_p
    return _lib


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_ptr():
    """hipStream_t of torch's current stream (what at::cuda::getCurrentCUDAStream() is to the
    reference wrappers, lib/src/ball_query.cpp:22).  Uses torch's raw-stream accessor when present:
    torch.cuda.current_stream() builds a Python Stream object (~8 us), and this runs once per launch."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def dev_ptr(t, dtype):
    """Device pointer of a dense tensor; refuses CPU tensors loudly (no fallback)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("cmflow_amd HIP op got a %s tensor: the product path runs on the GPU only" % t.device)
    if t.dtype != dtype:
        raise TypeError("expected %s, got %s" % (dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError("tensor must be contiguous")          # lib/pointnet2_utils.py:195-196,241-242
    return t.data_ptr()


def check(err, what):
    if err != 0:
        raise RuntimeError("%s failed: hipError_t %d" % (what, err))


# ---- live timing of the dominant kernel for bench.py's `roofline` object ---------------------------------------
# cmf_gemm (bound "mfma", units = 2*M*N*K flop) is bracketed INSIDE the library (cmf_gemm_profile_begin/_end,
# csrc/gemm.hip) with HIP events on the stream each launch goes to, so the GEMMs issued by cmf_setconv_forward/_backward
# on the side streams are covered exactly like the ones issued from Python.
# Only launches of at least this many flops are bracketed: an event pair costs a few microseconds of stream time, which
# would distort the step if the ~150 thin / small GEMM launches of a step were bracketed too.  The share of the FLOPs
# the bracketed launches carry is reported next to the result (`flop_share`).
def source_id():
    """Identity of the kernel SOURCES the in-tree library was built from (sha256 over csrc/*.hip, *.h and include/*.h, first 16
    hex digits): profiles that carry PMC constants record it, and bench.py only reports such a constant next to a library of the
    same sources."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(_HERE, "csrc")
    inc = os.path.join(os.path.dirname(_HERE), "include")
    for d in (csrc, inc):
        for f in sorted(os.listdir(d)):
            if f.endswith((".hip", ".h")):
                h.update(f.encode())
                h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


TRACK_MIN_UNITS = {"cmf_gemm": 1.0e9}
_prof_open = False


SAMPLE_EVERY = 1                                  # bench.py sets 4: an event pair around EVERY large launch costs ~0.15 ms per training step


def profile_begin():
    global _prof_open
    check(lib().cmf_gemm_profile_sampling(int(SAMPLE_EVERY)), "cmf_gemm_profile_sampling")
    check(lib().cmf_gemm_profile_begin(TRACK_MIN_UNITS["cmf_gemm"]), "cmf_gemm_profile_begin")
    _prof_open = True


def profile_end():
    global _prof_open
    if not _prof_open:
        return None
    _prof_open = False
    n, n_all = ctypes.c_longlong(), ctypes.c_longlong()
    ms, fl, fl_all = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
    check(lib().cmf_gemm_profile_end(ctypes.addressof(n), ctypes.addressof(ms), ctypes.addressof(fl), ctypes.addressof(n_all),
                                     ctypes.addressof(fl_all)), "cmf_gemm_profile_end")
    recs = (GemmLaunchRecord * max(1, n.value))()
    lib().cmf_gemm_profile_records(ctypes.addressof(recs), n.value)
    shapes = {}
    for r in recs[:n.value]:
        key = (r.M, r.N, r.K, r.layout, r.kind, r.split_k, r.bm, r.bn)
        e = shapes.setdefault(key, [0, 0.0])
        e[0] += 1
        e[1] += r.ms
    n_el, fl_el = ctypes.c_longlong(), ctypes.c_double()
    check(lib().cmf_gemm_profile_eligible(ctypes.addressof(n_el), ctypes.addressof(fl_el)), "cmf_gemm_profile_eligible")
    return {"kernel": "cmf_gemm", "bound": "mfma", "launches": n.value, "ms": ms.value, "units": fl.value,
            "launches_all": n_all.value, "units_all": fl_all.value, "shapes": shapes,
            "launches_eligible": n_el.value, "units_eligible": fl_el.value, "sample_every": int(SAMPLE_EVERY)}
