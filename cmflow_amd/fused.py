"""Point-major ("channels-last") primitives of the fused CMFlow path.

The reference keeps features as (B,C,N) and materialises (B,3+C,N,ns) grouped tensors for 1x1
convolutions (utils/model_utils/radarflow_util.py:144-162).  On MI355X the same arithmetic is
laid out as row-major matrices [positions, channels]:

* grouping a neighbour is a contiguous row copy (cmf_group_rows), its backward a deterministic
  segmented sum over an inverse index (cmf_build_inverse + cmf_group_rows_grad);
* every 1x1 conv is a GEMM  X[M,K] @ W[out,K]^T  on the conv weight viewed as (out,in);
* the first conv of a set-conv / cost-volume block is linear, so it is applied ONCE per point
  before grouping (W_f f)[idx] + W_xyz (x_j - x_i) instead of once per neighbour
  (exact algebra; only fp32 summation order differs) -- 63 % fewer model FLOPs.
"""
import torch
import torch.nn.functional as F
from torch.autograd import Function

from . import _lib

_f32, _i32 = torch.float32, torch.int32


class Neighbors:
    """idx (B,P,S) int32 into N points per sample, plus its lazily built inverse index."""

    def __init__(self, idx: torch.Tensor, n: int):
        assert idx.dtype == _i32 and idx.is_contiguous()
        self.idx, self.n = idx, n
        self.B, self.P, self.S = idx.shape
        self._inv = None

    def inverse(self):
        if self._inv is None:
            entries = self.P * self.S
            off = torch.empty(self.B, self.n + 1, dtype=_i32, device=self.idx.device)
            inv = torch.empty(self.B, entries, dtype=_i32, device=self.idx.device)
            err = _lib.lib().cmf_build_inverse_ps(self.B, self.n, self.P, self.S, _lib.dev_ptr(self.idx, _i32),
                                                  _lib.dev_ptr(off, _i32), _lib.dev_ptr(inv, _i32), _lib.stream_ptr())
            _lib.check(err, "cmf_build_inverse_ps")
            self._inv = (off, inv)
        return self._inv


def _rows_view(t):
    """(B,N,C) tensor whose rows are contiguous; returns (ptr-tensor, ld)."""
    B, N, C = t.shape
    if t.stride(2) != 1 or t.stride(0) != N * t.stride(1) or t.stride(1) < C:
        t = t.contiguous()
    return t, t.stride(1)


class _GroupRows(Function):
    @staticmethod
    def forward(ctx, feat, nbr: Neighbors):
        feat, ld = _rows_view(feat)
        B, N, C = feat.shape
        assert N == nbr.n and B == nbr.B
        out = torch.empty(B, nbr.P, nbr.S, C, dtype=_f32, device=feat.device)
        if not feat.is_cuda:
            raise RuntimeError("cmflow_amd HIP op got a %s tensor: the product path runs on the GPU only" % feat.device)
        err = _lib.lib().cmf_group_rows(B, N, C, ld, nbr.P * nbr.S, feat.data_ptr(), _lib.dev_ptr(nbr.idx, _i32),
                                        _lib.dev_ptr(out, _f32), _lib.stream_ptr())
        _lib.check(err, "cmf_group_rows")
        ctx.nbr, ctx.shape = nbr, (B, N, C)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        nbr = ctx.nbr
        B, N, C = ctx.shape
        off, inv = nbr.inverse()
        grad_out = grad_out.contiguous()
        g = torch.empty(B, N, C, dtype=_f32, device=grad_out.device)
        err = _lib.lib().cmf_group_rows_grad(B, N, C, C, nbr.P * nbr.S, 0, _lib.dev_ptr(grad_out, _f32),
                                             _lib.dev_ptr(off, _i32), _lib.dev_ptr(inv, _i32),
                                             _lib.dev_ptr(g, _f32), _lib.stream_ptr())
        _lib.check(err, "cmf_group_rows_grad")
        return g, None


def group_rows(feat, nbr: Neighbors):
    """feat (B,N,C) -> (B,P,S,C): out[b,p,s,:] = feat[b, idx[b,p,s], :]."""
    return _GroupRows.apply(feat, nbr)


def w2d(conv):
    """1x1 conv weight (out,in,1,1) as the (out,in) matrix of the equivalent GEMM."""
    return conv.weight.view(conv.weight.shape[0], conv.weight.shape[1])


# ---- fp32 MFMA GEMM (cmf_gemm) ---------------------------------------------------------------
def _p(t):
    return None if t is None else t.data_ptr()


def gemm(A, B, *, a_t=False, b_t=True, out=None, pro=None, prob=None, bias=None, act=0, stats=False,
         bwd=None, split_k=1, accumulate=False):
    """C = epi(pro(A) @ B) through cmf_gemm.  A: (M,K) [or (K,M) if a_t]; B: (K,N) [or (N,K) if b_t]
    -- both may be row-strided 2-D views.  Returns C, or (C, stats_partials) with stats=True."""
    assert A.dim() == 2 and B.dim() == 2 and A.stride(1) == 1 and B.stride(1) == 1
    M, K = (A.shape[1], A.shape[0]) if a_t else A.shape
    N = B.shape[0] if b_t else B.shape[1]
    assert (B.shape[1] if b_t else B.shape[0]) == K, (A.shape, B.shape, a_t, b_t)
    dev = A.device
    if out is None:
        out = torch.empty(M, N, dtype=_f32, device=dev)
    assert out.stride(1) == 1 and out.shape == (M, N)
    st = None
    dxyz = bwd[6] if (bwd is not None and len(bwd) > 6) else None
    if stats or (bwd is not None and bwd[0] == 1):
        st = torch.empty(_lib.lib().cmf_gemm_tiles_m(M), 5 if dxyz is not None else 2, N, dtype=_f32, device=dev)
    ws = torch.empty(split_k, M, N, dtype=_f32, device=dev) if split_k > 1 else None
    mode, Z, ea, ec, em, ei = 0, None, None, None, None, None
    if bwd is not None:
        mode, Z = bwd[0], bwd[1]
        if mode == 1:
            ea, ec, em, ei = bwd[2:6]
    err = _lib.lib().cmf_gemm(
        M, N, K, int(a_t), int(b_t), A.data_ptr(), A.stride(0), B.data_ptr(), B.stride(0), out.data_ptr(), out.stride(0),
        _p(pro[0]) if pro else None, _p(pro[1]) if pro else None, _p(prob[0]) if prob else None,
        _p(prob[1]) if prob else None, _p(bias), act, _p(st), mode, _p(Z), Z.stride(0) if Z is not None else 0,
        _p(ea), _p(ec), _p(em), _p(ei), _p(dxyz), split_k, _p(ws), int(accumulate), _lib.stream_ptr())
    _lib.check(err, "cmf_gemm")
    return (out, st) if st is not None else out
