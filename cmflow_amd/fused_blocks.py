"""Fused blocks of the point-major CMFlow path: hand-written forward AND backward over the HIP
kernels (cmf_gemm, cmf_group_affine, cmf_bn_*, cmf_*maxpool*), wrapped as autograd Functions.

Design (DESIGN.md "Fused path"):
* activations between layers stay as the PRE-BatchNorm GEMM output Z; the BatchNorm affine and
  the ReLU of layer l are applied by the consumer (GEMM prologue / max-pool / materialise kernel),
  so no normalised or activated tensor of the big grouped shape is ever written;
* train-mode BatchNorm statistics come out of the producing kernel's epilogue as per-tile partials
  (deterministic, no atomics) and are folded by cmf_bn_finalize into (a, c) = (gamma*invstd,
  beta - mean*a), which also updates running_mean / running_var exactly like nn.BatchNorm2d;
* backward: dX-GEMMs apply the producer's ReLU mask and emit the two BN-backward column sums in
  their epilogue; weight-gradient GEMMs contract over the positions with deterministic split-K and
  re-apply the producer's BN+ReLU to the saved Z in their B-operand prologue.
"""
import os

import torch
from torch.autograd import Function

from . import _lib
from . import pointnet2_utils as pointutils
from .fused import Neighbors, gemm, group_rows  # noqa: F401

_f32, _i32 = torch.float32, torch.int32
L = _lib.lib


def _p(t):
    return None if t is None else t.data_ptr()


def _tiles(M):
    return (M + 127) // 128


# ---- thin wrappers ---------------------------------------------------------------------------
class BNState:
    """Folded BatchNorm of one layer: y = relu(a*z + c); mean/invstd for the backward pass."""
    __slots__ = ("a", "c", "mean", "invstd", "training", "count")


def bn_fold(bn, partial, count):
    """partial [tiles][2][C] or None (eval).  Updates bn.running_* / num_batches_tracked in train mode."""
    C = bn.num_features
    dev = bn.weight.device
    st = BNState()
    buf = torch.empty(4, C, dtype=_f32, device=dev)
    st.mean, st.invstd, st.a, st.c = buf[0], buf[1], buf[2], buf[3]
    st.training = partial is not None
    st.count = count
    if st.training:
        track = bn.track_running_stats
        if bn.momentum is None:                              # cumulative moving average needs the count on the host
            bn.num_batches_tracked.add_(1)
            mom, nbt = 1.0 / float(bn.num_batches_tracked), None
        else:
            mom, nbt = bn.momentum, (_p(bn.num_batches_tracked) if track else None)   # counter bumped by the kernel
        err = L().cmf_bn_finalize(partial.shape[0], C, float(count), _p(partial), _p(bn.weight), _p(bn.bias),
                                  bn.eps, mom, _p(bn.running_mean) if track else None,
                                  _p(bn.running_var) if track else None,
                                  _p(st.mean), _p(st.invstd), _p(st.a), _p(st.c), nbt, _lib.stream_ptr())
    else:
        err = L().cmf_bn_finalize(0, C, 1.0, None, _p(bn.weight), _p(bn.bias), bn.eps, 0.0, _p(bn.running_mean),
                                  _p(bn.running_var), _p(st.mean), _p(st.invstd), _p(st.a), _p(st.c), None, _lib.stream_ptr())
    _lib.check(err, "cmf_bn_finalize")
    return st


# ---- side streams ---------------------------------------------------------------------------------
# Independent chains (the scales of an encoder, the two clouds, the two heads, the neighbourhood branch of the cost
# volume) are issued on side streams so that their small kernels run next to other chains' large ones.  The HIP runtime
# multiplexes all streams of a process onto a few hardware queues (4 by default); streams that land on the same queue
# serialise against each other in creation order, and measured on MI355X more than 4 busy queues is far slower
# (GPU_MAX_HW_QUEUES=5: 36 ms per step instead of 27) while 3 is faster than 4.  So the whole package shares ONE pool
# of N_SIDE side streams (+ the caller's stream), and chains are dealt to them by size instead of each module
# creating its own.
N_SIDE = max(1, int(os.environ.get("CMF_SIDE_STREAMS", "3")))
_side_pool = {}


SERIAL = False


def set_serial(net, on):
    """Diagnostic (bench.py --serial and its isolated-roofline pass): every chain of the product path on the caller's
    stream -- the same launches with nothing running next to them.  Clears the streams the modules of `net` cached."""
    global SERIAL
    SERIAL = bool(on)
    for m in net.modules():
        for attr in ("_streams", "_streams2", "_side", "_head_stream"):
            m.__dict__.pop(attr, None)


# WHICH streams make up the pool matters: the runtime deals streams to its hardware queues as they are created, balancing by how many
# streams each queue already carries -- so a library that created streams earlier shifts the deal.  Measured (tools/stream_queue_probe.py,
# profiles/r06_stream_queue_probe.txt): with a `nccl` process group initialised first (RCCL + c10d create their streams at init -- the
# normal order of a multi-GPU run), the FIRST torch.cuda.Stream() created afterwards shares the hardware queue of the default stream, i.e.
# side stream 0 serialised against the caller's stream: + 0.45 ms per training step with no collective running at all.  The pool is
# therefore picked by measurement: candidates are created one by one and a candidate is taken only if a 300 us one-wave spin kernel on
# it runs SIDE BY SIDE with the same kernel on the caller's stream and on every stream already taken (two streams of one hardware queue
# run them one after the other).  ~10 ms once per device; CMF_STREAM_PROBE=0 takes the first N_SIDE streams as they come (A/B).
_PROBE = os.environ.get("CMF_STREAM_PROBE", "1") != "0"


def _share_a_queue(a, b, us=300.0):
    import time
    best = 1e9
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        L().cmf_debug_spin(us, a.cuda_stream)
        L().cmf_debug_spin(us, b.cuda_stream)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best > 1.6e-6 * us


def _build_pool(key):
    main = torch.cuda.current_stream(key)
    if not _PROBE or torch.cuda.is_current_stream_capturing():
        return [torch.cuda.Stream(device=key) for _ in range(N_SIDE)]
    taken, spare = [], []
    with torch.cuda.device(key):
        for _ in range(N_SIDE + 6):                       # (4 hardware queues: a conflict-free set exists among the first few)
            if len(taken) == N_SIDE:
                break
            c = torch.cuda.Stream(device=key)
            if any(_share_a_queue(c, o) for o in [main] + taken):
                spare.append(c)
            else:
                taken.append(c)
    return taken + spare[:N_SIDE - len(taken)]            # (fewer free queues than N_SIDE: the rest share, as before)


def side_stream(slot, device=None):
    if SERIAL:
        return torch.cuda.current_stream(device)
    dev = torch.device(device if device is not None else torch.cuda.current_device())
    key = (dev.index if dev.index is not None else torch.cuda.current_device())
    pool = _side_pool.get(key)
    if pool is None:
        pool = _side_pool[key] = _build_pool(key)
    return pool[slot % N_SIDE]


def join_side_streams(main=None):
    """Make `main` (default: the current stream) wait for everything queued on the pool's side streams.  The backward
    kernels of chains that ran on a side stream accumulate parameter gradients straight into the flat bucket (grad_sink)
    without passing through autograd's AccumulateGrad, so autograd does not know those streams wrote the leaves: whoever
    reads the bucket next (all-reduce, optimizer) on `main` must be ordered behind the pool explicitly."""
    main = main or torch.cuda.current_stream()
    stress_point([st for pool in _side_pool.values() for st in pool])
    for pool in _side_pool.values():
        for st in pool:
            main.wait_stream(st)


# ---- timing perturbation at the fork points (tests/test_gpu_stress.py; CMF_STRESS=seed) -------------------------
# Every place where independent chains leave the caller's stream (the block calls of an encoder, the two heads, the
# cost-volume branches, the final join) can enqueue a random 0-300 us delay (cmf_debug_spin: one sleeping wave) ahead of
# the chain's first kernel.  Results must not depend on it: a missing stream dependency, an arena shared by two chains
# or a buffer handed back to the allocator too early shows up as a difference against the single-stream run.
_stress_rng = None


def stress_seed(seed):
    """seed: int to switch the perturbation on (deterministic sequence of delays), None to switch it off."""
    global _stress_rng
    import random
    _stress_rng = random.Random(seed) if seed is not None else None


if os.environ.get("CMF_STRESS"):
    stress_seed(int(os.environ["CMF_STRESS"]))


def stress_point(streams=()):
    """Fork point: maybe delay each of `streams` (the chains about to start) and one random stream of the pool."""
    r = _stress_rng
    if r is None or not torch.cuda.is_available():
        return
    for st in list(streams) + [side_stream(r.randrange(N_SIDE))]:
        if r.random() < 0.6:
            _lib.check(L().cmf_debug_spin(r.uniform(0.0, 300.0), st.cuda_stream), "cmf_debug_spin")


class _StressMarkFn(Function):
    """Identity whose backward is a fork point on the stream autograd replays it on (= the forward's stream): delays the
    backward of a side-stream branch.  Only inserted while the perturbation is on."""

    @staticmethod
    def forward(ctx, t):
        return t.view_as(t)

    @staticmethod
    def backward(ctx, g):
        stress_point([torch.cuda.current_stream()])
        return g


def stress_mark(t):
    return _StressMarkFn.apply(t) if (_stress_rng is not None and t.requires_grad) else t


def scale_streams(n_scales, cloud=0):
    """The side stream of each scale of a MultiScaleEncoder call (scales ordered by growing neighbourhood).  With four
    scales on three streams the two middle scales share a stream, the smallest and the largest (half of the rows) have
    one each; the second cloud's call is rotated by one stream.  Re-measured after the narrow layers were fused
    (tools/r02_session33.sh, three runs per deal): 23.1 ms per step against 23.5 for "largest shares with smallest"
    (the round-1 choice), 23.2-23.6 for five other deals, 24.0 with three scales on one stream; re-checked in round 3 (nine
    deals): 21.6-22.0 ms for all of them -- with the tails batched the deal no longer matters."""
    if n_scales == 4:
        slots = ([0, 1, 1, 2], [1, 2, 2, 0])[cloud % 2]
    else:
        slots = [(i + cloud) % N_SIDE for i in range(n_scales)]
    return [side_stream(s) for s in slots]


def enable_grad_sinks(params, on=True):
    """Opt the given parameters into in-place gradient accumulation (grad_sink).  dp.FlatGradBucket does this for the
    parameters it owns; a caller with its own backward() / optimizer loop who opts in must call join_side_streams() after
    backward() and before anything reads the .grad tensors: the kernels write them on the pool's side streams, bypassing
    autograd's AccumulateGrad (no hooks, no leaf-stream synchronisation, invisible to torch.autograd.grad)."""
    for p in params:
        p._cmf_sink = bool(on)


_COL_SINKS = os.environ.get("CMF_COL_SINKS", "1") != "0"       # 0: column blocks of a weight hand their gradients to autograd (A/B)


def grad_sink(t):
    """The gradient buffer a parameter-gradient can be accumulated into directly, or None (= hand the gradient to autograd).
    Only parameters opted in with enable_grad_sinks (the flat bucket's) have one.

    `t` is a Function input that is a leaf Parameter or the (out,in) view of a 1x1 conv weight.  When the
    parameter already owns a dense .grad (TrainStep keeps all of them as views into one flat bucket, zeroed
    once per step), the backward kernels write / accumulate straight into it and the Function returns None
    for that input -- otherwise autograd launches one tiny `grad += g` kernel per parameter (~300 per step,
    each ~25 us of host time on the autograd thread)."""
    base = t._base if t._base is not None else t
    if not getattr(base, "_cmf_sink", False) or not base.is_leaf or base.grad is None or not base.grad.is_contiguous():
        return None
    if t is base:
        return base.grad
    if t.dim() == 2 and base.dim() == 4 and t.is_contiguous() and t.numel() == base.numel():
        return base.grad.view(t.shape)
    # a block of columns of the (out, in) matrix (the cost volume's first conv is used as three column blocks): the same columns of
    # the gradient buffer -- as separate gradients autograd builds each block's full-size zero-padded copy and adds the three
    if _COL_SINKS and t.dim() == 2 and base.dim() in (2, 4) and base.is_contiguous():
        rows = base.shape[0]
        K = base.numel() // rows
        col = t.storage_offset() - base.storage_offset()
        if t.shape[0] == rows and t.stride() == (K, 1) and 0 <= col and col + t.shape[1] <= K:
            return base.grad.view(rows, K)[:, col:col + t.shape[1]]
    return None


def colsum(partial, acc_beta=None, acc_gamma=None):
    C = partial.shape[2]
    out = torch.empty(2, C, dtype=_f32, device=partial.device)
    _lib.check(L().cmf_colsum_finalize(partial.shape[0], C, _p(partial), _p(out), _p(acc_beta), _p(acc_gamma),
                                       _lib.stream_ptr()), "cmf_colsum_finalize")
    return out


def group_affine(ysrc, yctr, xyz_src, xyz_ctr, wx, idx, act=0, stats=True, want_dxyz=True, extra=False, write_z=True):
    """-> z (B,P,S,C), dxyz (B,P,S,4) or None, partial or None[, partial_x].  ysrc (B,n_src,C) row-strided view.
    write_z=False (with stats): the statistics only, z is None."""
    B, n_src, C = ysrc.shape
    _, P, S = idx.shape
    assert ysrc.stride(2) == 1 and ysrc.stride(0) == n_src * ysrc.stride(1)
    dev = ysrc.device
    z = torch.empty(B, P, S, C, dtype=_f32, device=dev) if write_z else None
    dxyz = torch.empty(B, P, S, 4, dtype=_f32, device=dev) if want_dxyz else None
    part = torch.empty(_tiles(B * P * S), 2, C, dtype=_f32, device=dev) if stats else None
    part_x = torch.empty(_tiles(B * P * S), 3 * C + 4, dtype=_f32, device=dev) if (stats and extra) else None
    wx = wx.contiguous()
    err = L().cmf_group_affine(B, n_src, P, S, C, ysrc.data_ptr(), ysrc.stride(1),
                               _p(yctr), yctr.stride(1) if yctr is not None else 0,
                               _p(xyz_src), _p(xyz_ctr), _p(wx), wx.stride(0), _p(idx), act,
                               _p(z), _p(dxyz), _p(part), _p(part_x), _lib.stream_ptr())
    _lib.check(err, "cmf_group_affine")
    if extra:
        return z, dxyz, part, part_x
    return z, dxyz, part


def bn_relu_maxpool(z, st, out=None):
    """z (P,S,C) -> out (P,C), argmax (P,C) uint8"""
    P, S, C = z.shape
    if out is None:
        out = torch.empty(P, C, dtype=_f32, device=z.device)
    am = torch.empty(P, C, dtype=torch.uint8, device=z.device)
    err = L().cmf_bn_relu_maxpool(P, S, C, _p(z), _p(st.a), _p(st.c), out.data_ptr(), out.stride(0), _p(am), _lib.stream_ptr())
    _lib.check(err, "cmf_bn_relu_maxpool")
    return out, am


def maxpool_bwd(dout, z, st, am):
    P, S, C = z.shape
    dU = torch.empty(P * S, C, dtype=_f32, device=z.device)
    part = torch.empty(_tiles(P * S), 2, C, dtype=_f32, device=z.device)
    err = L().cmf_maxpool_bwd(P, S, C, dout.data_ptr(), dout.stride(0), _p(z), _p(st.a), _p(st.c), _p(st.mean),
                              _p(st.invstd), _p(am), _p(dU), _p(part), _lib.stream_ptr())
    _lib.check(err, "cmf_maxpool_bwd")
    return dU, part


def affine_relu(z, st, out=None):
    M, C = z.shape
    if out is None:
        out = torch.empty(M, C, dtype=_f32, device=z.device)
    err = L().cmf_affine_relu(M, C, z.data_ptr(), z.stride(0), _p(st.a), _p(st.c), out.data_ptr(), out.stride(0), _lib.stream_ptr())
    _lib.check(err, "cmf_affine_relu")
    return out


def act_bwd_stats(dY, z, st):
    M, C = z.shape
    dU = torch.empty(M, C, dtype=_f32, device=z.device)
    part = torch.empty(_tiles(M), 2, C, dtype=_f32, device=z.device)
    err = L().cmf_act_bwd_stats(M, C, dY.data_ptr(), dY.stride(0), z.data_ptr(), z.stride(0), _p(st.a), _p(st.c),
                                _p(st.mean), _p(st.invstd), _p(dU), _p(part), _lib.stream_ptr())
    _lib.check(err, "cmf_act_bwd_stats")
    return dU, part


def colsum_n(partial, C=0, acc_beta=None, acc_gamma=None):
    """[tiles][...] -> [...] column sums in fixed order"""
    ncols = partial[0].numel()
    out = torch.empty(partial.shape[1:], dtype=_f32, device=partial.device)
    _lib.check(L().cmf_colsum(partial.shape[0], ncols, _p(partial), _p(out), C, _p(acc_beta), _p(acc_gamma),
                              _lib.stream_ptr()), "cmf_colsum")
    return out


def bn_backward(dU, part, z, st, gamma=None, beta=None):
    """dU (M,C) masked upstream gradient + its partial sums -> dZ (in place), dgamma, dbeta.
    gamma / beta: the BN parameters; if they own a gradient buffer the sums are accumulated into it by the
    reduction kernel and None is returned in their place."""
    sg = grad_sink(gamma) if gamma is not None else None
    sb = grad_sink(beta) if beta is not None else None
    sums = colsum(part, sb, sg)                           # [0] = sum dU = dbeta, [1] = sum dU*zhat = dgamma
    M, C = dU.shape
    err = L().cmf_bn_bwd_apply(M, C, _p(dU), z.data_ptr(), z.stride(0), _p(st.a), _p(st.mean), _p(st.invstd),
                               _p(sums) if st.training else None, _lib.stream_ptr())
    _lib.check(err, "cmf_bn_bwd_apply")
    return dU, (None if sg is not None else sums[1]), (None if sb is not None else sums[0])


def dw_split(M, N, K):
    """Split-K factor of a weight gradient (same rule as csrc/setconv_block.hip dw_split)."""
    if N <= 64 and K <= 64:
        return max(2, min(M // 128, 1024))                 # thin kernel: one slab per workgroup, >= 128 rows each
    tiles = ((N + 127) // 128) * ((K + 127) // 128)
    chunks = (M + 15) // 16
    if chunks < 64:
        return 1
    best, best_cost = 8, 1e30
    for s in (8, 16, 24, 32, 48, 64, 96, 128, 256, 384):   # multiples of 8: slabs are dealt to the 8 XCDs
        if chunks // s < 8:
            break
        cost = ((tiles * s + 767) // 768) * (chunks / s + 12.0)
        if cost < best_cost:
            best, best_cost = s, cost
    return best


def gemm_dw(dZ, X, prob=None, w=None):
    """Weight gradient dW[N,K] = dZ[M,N]^T @ act(X)[M,K]; contraction over the positions, split-K so that
    the launch fills the chip, slabs summed in fixed order (deterministic).  w: the weight input of the
    Function; when it owns a gradient buffer (grad_sink) dW is accumulated there and None is returned."""
    sink = grad_sink(w) if w is not None else None
    M, N = dZ.shape
    K = X.shape[1]
    split = dw_split(M, N, K)
    if sink is not None:
        gemm(dZ, X, a_t=True, b_t=False, prob=prob, split_k=split, out=sink, accumulate=True)
        return None
    tail = K % 128
    if prob is None and 0 < tail <= 64 and K >= 512 and N >= 512 and N % 128 == 0 and X.data_ptr() % 16 == 0 and M >= 4096:
        # a few columns past a multiple of 128 (the stacked first conv's 1040 = 1024 + 16): as one call the ninth column of tiles runs the
        # ragged-edge loop on 128-wide tiles for 16 columns -- 800 us for 2048 x 1040 x 16384; as a 1024-wide call plus a 16-wide one with
        # its own split count (one round of workgroups) 650 us (tools/dw1040_probe.py).  Same sums per element, another slab partition for the tail.
        out = torch.empty(N, K, dtype=_f32, device=dZ.device)
        K1 = K - tail
        gemm(dZ, X[:, :K1], a_t=True, b_t=False, split_k=dw_split(M, N, K1), out=out[:, :K1])
        ts = max(8, (768 // (N // 128)) // 8 * 8)
        while ts > 8 and M // 16 // ts < 8:
            ts -= 8
        gemm(dZ, X[:, K1:], a_t=True, b_t=False, split_k=ts, out=out[:, K1:])
        return out
    return gemm(dZ, X, a_t=True, b_t=False, prob=prob, split_k=split)


class LinBN:
    """One (1x1 conv, BatchNorm2d) pair viewed as GEMM weight + BN module."""

    def __init__(self, conv, bn):
        self.conv, self.bn = conv, bn

    @property
    def w(self):
        w = self.conv.weight
        return w.view(w.shape[0], w.shape[1])


def _fwd_layer(x, x_st, w, bn, training):
    """Z = act(x) @ w^T with train-mode statistics; x_st None => x is already activated."""
    pro = (x_st.a, x_st.c) if x_st is not None else None
    if training:
        z, part = gemm(x, w, pro=pro, stats=True)
        return z, bn_fold(bn, part, x.shape[0])
    return gemm(x, w, pro=pro), bn_fold(bn, None, x.shape[0])


def _bwd_layer(dZ, x, x_st, w, need_dx=True, w_in=None):
    """Given dZ of Z = act(x) @ w^T: -> dW, and (dU_x, partial) = gradient wrt x's PRE-activation, masked by
    x's ReLU with the BN-backward sums of layer x (x_st given), or the plain dX (x_st None)."""
    dW = gemm_dw(dZ, x, prob=(x_st.a, x_st.c) if x_st is not None else None, w=w_in)
    if not need_dx:
        return dW, None, None
    if x_st is None:
        return dW, gemm(dZ, w, b_t=False), None
    dU, part = gemm(dZ, w, b_t=False, bwd=(1, x, x_st.a, x_st.c, x_st.mean, x_st.invstd))
    return dW, dU, part


# ---- plain linear layer on own GEMM ------------------------------------------------------------
class LinearFn(Function):
    """y = act(x @ w^T + bias), x (M,K) row-strided, w (N,K).  act: 0 none / 1 relu / 2 leaky(0.1) / 3 sigmoid.
    Operands whose row stride is not a multiple of 4 floats are copied into padded buffers."""

    @staticmethod
    def forward(ctx, x, w, bias, act, preact_grad=False, in_bias=None):
        # preact_grad: the consumer hands back the gradient w.r.t. the pre-activation (WeightedKSumFn relu_w=True)
        # in_bias: x is the stored ReLU activation of a layer with this bias that was built with preact_grad=True and a
        # detached bias: the input gradient is then formed w.r.t. that layer's pre-activation (mask in the GEMM epilogue)
        # and its column sums -- the producer's bias gradient -- are returned for in_bias, instead of a compare, a
        # multiply and a reduction as separate torch kernels in the producer's backward
        ctx.preact_grad = preact_grad
        ctx.in_bias = in_bias
        ctx.w_in = w
        ok = x.stride(1) == 1 and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0 and x.shape[1] % 4 == 0
        x2 = x if ok else _pad_cols(x)
        w2 = _pad_cols(w)
        y = gemm(x2, w2, bias=bias, act=act)
        ctx.save_for_backward(x2, w2, y if act else None)
        ctx.act, ctx.has_bias, ctx.shape = act, bias is not None, (w.shape[0], w.shape[1])
        return y

    @staticmethod
    def backward(ctx, dy):
        x2, w2, y = ctx.saved_tensors
        N, K = ctx.shape
        dy = dy.contiguous()
        if ctx.preact_grad:
            pass
        elif ctx.act == 1:
            dy = dy * (y > 0)
        elif ctx.act == 2:
            dy = torch.where(y > 0, dy, 0.1 * dy)
        elif ctx.act == 3:
            dy = dy * y * (1 - y)
        dx = dw = db = dib = None
        dyp = _pad_cols(dy)                                             # (M, N4)
        if ctx.needs_input_grad[0]:
            wp = w2 if dyp.shape[1] == N else torch.nn.functional.pad(w2, (0, 0, 0, dyp.shape[1] - N))
            if ctx.in_bias is not None:
                dx, part = gemm(dyp, wp, b_t=False, bwd=(3, x2), stats=True)
                sink = grad_sink(ctx.in_bias)
                sums = colsum_n(part, K, sink, None)
                dib = None if sink is not None else sums[0, :K]
                dx = dx[:, :K]
            else:
                dx = gemm(dyp, wp, b_t=False)[:, :K]
        if ctx.needs_input_grad[1]:
            if dyp.shape[1] == N and x2.shape[1] == K and grad_sink(ctx.w_in) is not None and grad_sink(ctx.w_in).shape == (N, K):
                gemm_dw(dyp, x2, w=ctx.w_in)                                    # accumulated in place: no AccumulateGrad add_
            else:
                dw = gemm_dw(dyp, x2)[:N, :K]
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy.sum(0)
        return dx, dw, db, None, None, dib




class StackedFirstConvFn(Function):
    """y_all = feats @ [W_f of every scale stacked]^T -- the feature half of the hoisted first convs of a
    MultiScaleEncoder (radarflow_util.py:132-139 by linearity) as one GEMM, with the weight bookkeeping done here
    instead of by autograd over cat / slice / pad:
      * feats (M, Kp) may carry the original feature channels in the order [tail | head] -> [head, tail, zero pad]
        (n_tail leading channels moved behind the rest and Kp a multiple of 16): the input gradient is then only
        computed for the first Kp_grad columns (the moved channels are raw inputs, e.g. v_r / RCS, and need none),
        which makes the data-gradient GEMM N = 1024 wide instead of 1028 (9 -> 8 column tiles);
      * the weight gradient (n*O1, Kp) is added straight into the conv weights' .grad when they exist."""

    @staticmethod
    def _uniform(weights):
        w0 = weights[0]
        return len(weights) <= 8 and all(w.shape == w0.shape and w.is_contiguous() and w.dtype == _f32 for w in weights)

    @staticmethod
    def forward(ctx, feats, n_tail, n_grad, *weights):
        import ctypes
        M, Kp = feats.shape
        if StackedFirstConvFn._uniform(weights):                                 # one launch (cmf_stack_first_conv)
            o1, cin = weights[0].shape[0], weights[0].shape[1] - 3
            wf = torch.empty(len(weights) * o1, Kp, dtype=_f32, device=feats.device)
            ptrs = (ctypes.c_void_p * len(weights))(*[w.data_ptr() for w in weights])
            _lib.check(L().cmf_stack_first_conv(len(weights), o1, cin, n_tail, Kp, ctypes.addressof(ptrs), wf.data_ptr(),
                                                _lib.stream_ptr()), "cmf_stack_first_conv")
        else:
            with torch.no_grad():
                parts = []
                for w in weights:
                    w2 = w.view(w.shape[0], w.shape[1])[:, 3:]                   # feature columns (xyz columns first)
                    parts.append(torch.cat((w2[:, n_tail:], w2[:, :n_tail]), dim=1) if n_tail else w2)
                wf = torch.cat(parts, dim=0)
                if wf.shape[1] < Kp:
                    wf = torch.nn.functional.pad(wf, (0, Kp - wf.shape[1]))
        y = gemm(feats, wf)
        ctx.save_for_backward(feats, wf)
        ctx.n_tail, ctx.n_grad, ctx.weights = n_tail, n_grad, weights
        return y

    @staticmethod
    def backward(ctx, dy):
        feats, wf = ctx.saved_tensors
        M, Kp = feats.shape
        dy = dy.contiguous()
        dfeats = None
        if ctx.needs_input_grad[0]:
            ng = ctx.n_grad if ctx.n_grad else Kp
            dfeats = torch.empty(M, Kp, dtype=_f32, device=dy.device)
            gemm(dy, wf[:, :ng], b_t=False, out=dfeats[:, :ng])
            if ng < Kp:
                # columns >= ng are declared gradient-free by the caller (raw input channels + zero pad); the model never
                # reads them, but a caller whose tail DOES require grad must see zeros, not uninitialised memory
                dfeats[:, ng:].zero_()
        dwf = gemm_dw(dy, feats)                                                 # (n*O1, Kp)
        ws = ctx.weights
        if StackedFirstConvFn._uniform(ws) and all(w.is_leaf and w.grad is not None and w.grad.is_contiguous() for w in ws):
            import ctypes                                                        # every scale owns a gradient buffer: one launch
            o1, cin = ws[0].shape[0], ws[0].shape[1] - 3
            ptrs = (ctypes.c_void_p * len(ws))(*[w.grad.data_ptr() for w in ws])
            _lib.check(L().cmf_unstack_first_conv_grad(len(ws), o1, cin, ctx.n_tail, Kp, dwf.data_ptr(), ctypes.addressof(ptrs),
                                                       _lib.stream_ptr()), "cmf_unstack_first_conv_grad")
            return (dfeats, None, None, *([None] * len(ws)))
        grads, r0 = [], 0
        for w in ctx.weights:
            o, cin = w.shape[0], w.shape[1] - 3
            blk = dwf[r0:r0 + o]
            r0 += o
            head, tail = blk[:, :cin - ctx.n_tail], blk[:, cin - ctx.n_tail:cin]
            g = w.grad if (w.is_leaf and w.grad is not None) else None
            if g is not None:
                g2 = g.view(o, cin + 3)
                g2[:, 3 + ctx.n_tail:] += head
                if ctx.n_tail:
                    g2[:, 3:3 + ctx.n_tail] += tail
                grads.append(None)
            else:
                full = torch.zeros(o, cin + 3, dtype=_f32, device=dy.device)
                full[:, 3 + ctx.n_tail:] = head
                if ctx.n_tail:
                    full[:, 3:3 + ctx.n_tail] = tail
                grads.append(full.view_as(w))
        return (dfeats, None, None, *grads)


def inputs_point_major(pc1, pc2, ft1, ft2):
    """pc (B,3,N), ft (B,C,N) of both clouds -> x1, x2 (B,N,3), a1, a2 (B,N,C') with C' = C rounded up to a multiple of 4 floats, zero
    columns behind the channels (cmf_inputs_point_major: one launch where torch takes two copies, two fills and two more copies)."""
    B, C, N = ft1.shape
    cp = (C + 4) // 4 * 4 if C % 4 else C
    dev = pc1.device
    same = pc2.shape == pc1.shape and ft2.shape == ft1.shape and pc1.is_cuda and all(t.dtype == _f32 for t in (pc1, pc2, ft1, ft2))
    if not same:
        f = lambda t: torch.nn.functional.pad(t.transpose(1, 2), (0, cp - t.shape[1]))
        return pc1.transpose(1, 2).contiguous(), pc2.transpose(1, 2).contiguous(), f(ft1), f(ft2)
    x1, x2 = torch.empty(B, N, 3, dtype=_f32, device=dev), torch.empty(B, N, 3, dtype=_f32, device=dev)
    a1, a2 = torch.empty(B, N, cp, dtype=_f32, device=dev), torch.empty(B, N, cp, dtype=_f32, device=dev)
    p = lambda t: _lib.dev_ptr(t.contiguous(), _f32)
    _lib.check(_lib.lib().cmf_inputs_point_major(B, N, C, cp, p(pc1), p(pc2), p(ft1), p(ft2), x1.data_ptr(), x2.data_ptr(), a1.data_ptr(),
                                                 a2.data_ptr(), _lib.stream_ptr()), "cmf_inputs_point_major")
    return x1, x2, a1, a2


def _pad_k(t):
    """Copy a 2-D tensor into a buffer whose row stride is a multiple of 4 floats (zero padded); returns the
    (rows, K) view of it."""
    return _pad_cols(t)[:, :t.shape[1]]


def _pad_cols(t):
    """Like _pad_k but returns the padded (rows, ld) tensor (extra zero columns participate harmlessly)."""
    r, k = t.shape
    ld = (k + 3) // 4 * 4
    if ld == k and t.is_contiguous():
        return t
    if not (t.is_cuda and t.dtype == _f32 and t.stride(1) == 1) or (torch.is_grad_enabled() and t.requires_grad):
        return torch.nn.functional.pad(t, (0, ld - k))          # (differentiable where a caller outside a Function needs it)
    buf = torch.empty(r, ld, dtype=_f32, device=t.device)            # one launch (torch: a fill and a copy)
    _lib.check(_lib.lib().cmf_pad_rows(r, k, t.data_ptr(), t.stride(0), buf.data_ptr(), ld, _lib.stream_ptr()), "cmf_pad_rows")
    return buf


def linear(x, w, bias=None, act=0, preact_grad=False, in_bias=None):
    """(..., K) -> (..., N) through cmf_gemm"""
    shp = x.shape
    y = LinearFn.apply(x.reshape(-1, shp[-1]), w, bias, act, preact_grad, in_bias)
    return y.view(*shp[:-1], w.shape[0])


# ---- [linear + BN + ReLU] x L on a materialised input -------------------------------------------
class MLPChainFn(Function):
    """x (M,K) activated -> relu(bn_L(... relu(bn_1(x w_1^T)) ...)) materialised (M, C_L).
    params: w_1, gamma_1, beta_1, ..., w_L, gamma_L, beta_L; bns: the BatchNorm modules (running stats)."""

    @staticmethod
    def forward(ctx, x, bns, training, *params):
        x = x if (x.stride(1) == 1 and x.stride(0) % 4 == 0) else x.contiguous()
        zs, sts = [], []
        cur, cur_st = x, None
        for i, bn in enumerate(bns):
            w = params[3 * i].contiguous()
            z, st = _fwd_layer(cur, cur_st, w, bn, training)
            zs.append(z)
            sts.append(st)
            cur, cur_st = z, st
        y = affine_relu(cur, cur_st)
        ctx.x, ctx.zs, ctx.sts, ctx.ws = x, zs, sts, [params[3 * i].contiguous() for i in range(len(bns))]
        ctx.params = params
        return y

    @staticmethod
    def backward(ctx, dy):
        x, zs, sts, ws, P = ctx.x, ctx.zs, ctx.sts, ctx.ws, ctx.params
        Lr = len(zs)
        grads = [None] * (3 * Lr)
        dU, part = act_bwd_stats(dy, zs[-1], sts[-1])
        dx = None
        for i in range(Lr - 1, -1, -1):
            dZ, dg, db = bn_backward(dU, part, zs[i], sts[i], P[3 * i + 1], P[3 * i + 2])
            grads[3 * i + 1], grads[3 * i + 2] = dg, db
            if i > 0:
                grads[3 * i], dU, part = _bwd_layer(dZ, zs[i - 1], sts[i - 1], ws[i], w_in=P[3 * i])
            else:
                grads[0], dx, _ = _bwd_layer(dZ, x, None, ws[0], need_dx=ctx.needs_input_grad[0], w_in=P[0])
        return (dx, None, None, *grads)


class MLPChainBlockFn(Function):
    """MLPChainFn with the kernel sequence issued by ONE C-ABI call per direction (cmf_mlp_forward / _backward,
    csrc/setconv_block.hip): same kernels, same order, same numerics.  Sequenced from Python the heads' chains were host
    bound -- ~25 launches of 5-45 us with 15-30 us of interpreter time between them, and the autograd thread enqueues the
    two heads one after the other: 1.0 ms at the start of every backward pass with the GPU idle."""

    @staticmethod
    def forward(ctx, x, bns, training, *params):
        import ctypes
        x = x if (x.stride(1) == 1 and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0) else x.contiguous()
        M, dev, nl = x.shape[0], x.device, len(bns)
        d = _lib.MlpDesc()
        d.M, d.L, d.training = M, nl, int(training)
        d.C[0] = x.shape[1]
        ws = []
        for l, bn in enumerate(bns):
            w = params[3 * l].contiguous()
            ws.append(w)
            assert bn.momentum is not None, "cumulative moving average BN is not supported by the block call"
            d.C[l + 1] = w.shape[0]
            d.eps[l], d.momentum[l] = bn.eps, bn.momentum
            d.w[l], d.gamma[l], d.beta[l] = w.data_ptr(), bn.weight.data_ptr(), bn.bias.data_ptr()
            d.rmean[l], d.rvar[l] = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
            d.nbt[l] = bn.num_batches_tracked.data_ptr() if (training and bn.track_running_stats) else None
        n_s, n_f, n_b = ctypes.c_longlong(), ctypes.c_longlong(), ctypes.c_longlong()
        _lib.check(L().cmf_mlp_sizes(ctypes.addressof(d), ctypes.addressof(n_s), ctypes.addressof(n_f), ctypes.addressof(n_b)), "cmf_mlp_sizes")
        saved = torch.empty(n_s.value, dtype=_f32, device=dev)
        scratch = torch.empty(n_f.value, dtype=_f32, device=dev)
        out = torch.empty(M, ws[-1].shape[0], dtype=_f32, device=dev)
        d.x, d.ldx, d.saved, d.scratch, d.out, d.ldo = x.data_ptr(), x.stride(0), saved.data_ptr(), scratch.data_ptr(), out.data_ptr(), out.stride(0)
        _lib.check(L().cmf_mlp_forward(ctypes.addressof(d), _lib.stream_ptr()), "cmf_mlp_forward")
        ctx.state = (d, x, ws, saved, n_b.value)
        ctx.params = params
        return out

    @staticmethod
    def backward(ctx, dy):
        import ctypes
        d, x, ws, saved, n_bwd = ctx.state
        P = ctx.params
        dev = dy.device
        if dy.stride(1) != 1 or dy.stride(0) % 4 or dy.data_ptr() % 16:
            dy = dy.contiguous()
        scratch = torch.empty(n_bwd, dtype=_f32, device=dev)
        d.scratch, d.dout, d.lddout = scratch.data_ptr(), dy.data_ptr(), dy.stride(0)
        grads = [None] * len(P)
        for l in range(d.L):
            sink = grad_sink(P[3 * l])
            if sink is not None:
                d.dw[l], d.acc_w[l] = sink.data_ptr(), 1
            else:
                grads[3 * l] = torch.empty_like(ws[l])
                d.dw[l], d.acc_w[l] = grads[3 * l].data_ptr(), 0
            sg, sb = grad_sink(P[3 * l + 1]), grad_sink(P[3 * l + 2])
            if sg is not None and sb is not None:
                d.dgamma[l], d.dbeta[l], d.acc_bn[l] = sg.data_ptr(), sb.data_ptr(), 1
            else:
                grads[3 * l + 1], grads[3 * l + 2] = torch.empty_like(P[3 * l + 1]), torch.empty_like(P[3 * l + 2])
                d.dgamma[l], d.dbeta[l], d.acc_bn[l] = grads[3 * l + 1].data_ptr(), grads[3 * l + 2].data_ptr(), 0
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(x.shape[0], x.shape[1], dtype=_f32, device=dev)
            d.dx, d.lddx = dx.data_ptr(), dx.stride(0)
        else:
            d.dx, d.lddx = None, 0
        _lib.check(L().cmf_mlp_backward(ctypes.addressof(d), _lib.stream_ptr()), "cmf_mlp_backward")
        return (dx, None, None, *grads)


def mlp_chain(x, layers, training):
    """layers: list of (conv, bn) modules.  x (..., K) -> (..., C_L)."""
    shp = x.shape
    params = []
    for conv, bn in layers:
        params += [conv.weight.view(conv.weight.shape[0], conv.weight.shape[1]), bn.weight, bn.bias]
    block = USE_BLOCK_CALLS and len(layers) <= 4 and shp[-1] % 4 == 0 and all(c.weight.shape[0] % 4 == 0 for c, _ in layers)
    fn = MLPChainBlockFn if block else MLPChainFn
    y = fn.apply(x.reshape(-1, shp[-1]), [bn for _, bn in layers], training, *params)
    return y.view(*shp[:-1], y.shape[-1])


def mlp_chain_w(x, layers, training):
    """mlp_chain with the weights given as (out, in) matrices: layers = [(weight2d, bn)]."""
    shp = x.shape
    params = []
    for w, bn in layers:
        params += [w, bn.weight, bn.bias]
    block = USE_BLOCK_CALLS and len(layers) <= 4 and shp[-1] % 4 == 0 and all(w.shape[0] % 4 == 0 for w, _ in layers)
    fn = MLPChainBlockFn if block else MLPChainFn
    y = fn.apply(x.reshape(-1, shp[-1]), [bn for _, bn in layers], training, *params)
    return y.view(*shp[:-1], y.shape[-1])


# ---- the set-conv block (PointLocalFeature) -------------------------------------------------------
class SetConvFn(Function):
    """utils/model_utils/radarflow_util.py:144-162 in one fused forward/backward.
    xyz_t (B,N,3); y (B,N,O1) = feats @ W_f^T (the feature half of the first conv, applied per point);
    params: wx (O1,3), g1,b1, w2,g2,b2, w3,g3,b3, w4,g4,b4, w5,g5,b5, w6,g6,b6.   -> (B,N,C6)."""

    @staticmethod
    def forward(ctx, xyz_t, y, radius, nsample, bns, training, *params):
        B, N, _ = xyz_t.shape
        wx, g1, b1, w2, g2, b2, w3, g3, b3, w4, g4, b4, w5, g5, b5, w6, g6, b6 = params
        idx = pointutils.ball_query(radius, nsample, xyz_t, xyz_t)
        nbr = Neighbors(idx, N)
        if training:
            z1, dxyz, part, part_x = group_affine(y, None, xyz_t, xyz_t, wx, idx, act=0, stats=True, extra=True)
            fwd_sums = colsum_n(part_x)                     # [3*O1 + 4]: sum z*d_k per channel, sum d_k
        else:
            z1, dxyz, part = group_affine(y, None, xyz_t, xyz_t, wx, idx, act=0, stats=False)
            fwd_sums = None
        M = B * N * nsample
        z1 = z1.view(M, -1)
        st1 = bn_fold(bns[0], part, M)
        w2, w3, w4, w5, w6 = (t.contiguous() for t in (w2, w3, w4, w5, w6))
        z2, st2 = _fwd_layer(z1, st1, w2, bns[1], training)
        z3, st3 = _fwd_layer(z2, st2, w3, bns[2], training)
        x, am = bn_relu_maxpool(z3.view(B * N, nsample, -1), st3)
        z4, st4 = _fwd_layer(x, None, w4, bns[3], training)
        z5, st5 = _fwd_layer(z4, st4, w5, bns[4], training)
        z6, st6 = _fwd_layer(z5, st5, w6, bns[5], training)
        res = affine_relu(z6, st6)
        ctx.saved = (nbr, dxyz, z1, z2, z3, am, x, z4, z5, z6, (st1, st2, st3, st4, st5, st6), (w2, w3, w4, w5, w6),
                     (B, N, nsample), y.shape[2], fwd_sums)
        ctx.params = params
        return res.view(B, N, -1)

    @staticmethod
    def backward(ctx, dout):
        nbr, dxyz, z1, z2, z3, am, x, z4, z5, z6, sts, ws, (B, N, S), O1, fwd_sums = ctx.saved
        st1, st2, st3, st4, st5, st6 = sts
        w2, w3, w4, w5, w6 = ws
        dout = dout.reshape(B * N, -1)
        (pwx, pg1, pb1, pw2, pg2, pb2, pw3, pg3, pb3, pw4, pg4, pb4, pw5, pg5, pb5, pw6, pg6, pb6) = ctx.params
        g = {}
        dU, part = act_bwd_stats(dout, z6, st6)
        dZ6, g["g6"], g["b6"] = bn_backward(dU, part, z6, st6, pg6, pb6)
        g["w6"], dU, part = _bwd_layer(dZ6, z5, st5, w6, w_in=pw6)
        dZ5, g["g5"], g["b5"] = bn_backward(dU, part, z5, st5, pg5, pb5)
        g["w5"], dU, part = _bwd_layer(dZ5, z4, st4, w5, w_in=pw5)
        dZ4, g["g4"], g["b4"] = bn_backward(dU, part, z4, st4, pg4, pb4)
        g["w4"], dx, _ = _bwd_layer(dZ4, x, None, w4, w_in=pw4)
        dU, part = maxpool_bwd(dx, z3.view(B * N, S, -1), st3, am)
        dZ3, g["g3"], g["b3"] = bn_backward(dU, part, z3, st3, pg3, pb3)
        g["w3"], dU, part = _bwd_layer(dZ3, z2, st2, w3, w_in=pw3)
        dZ2, g["g2"], g["b2"] = bn_backward(dU, part, z2, st2, pg2, pb2)
        # first layer: z1 = y[idx] + wx . dxyz.  The dX GEMM masks by ReLU'(z1) and emits, next to the two BN sums,
        # the three column sums of dU*d_k; the BN backward is then folded into the scatter (dZ1 is never written)
        # and dW_xyz follows from column sums alone (no pass over the M x O1 tensor, no N=4 GEMM).
        g["w2"] = gemm_dw(dZ2, z1, prob=(st1.a, st1.c), w=pw2)
        dU, part = gemm(dZ2, w2, b_t=False, bwd=(1, z1, st1.a, st1.c, st1.mean, st1.invstd, dxyz.view(-1, 4)))
        sg1, sb1 = grad_sink(pg1), grad_sink(pb1)
        sums5 = colsum_n(part, O1, sb1, sg1)                 # [5][O1]: s1, s2, q0, q1, q2
        g["g1"], g["b1"] = (None if sg1 is not None else sums5[1]), (None if sb1 is not None else sums5[0])
        M = dU.shape[0]
        # wx is the [:, :3] slice of the first conv's (out, 3+C) weight: accumulate into the same slice of its grad
        wbase = pwx._base if pwx._base is not None else pwx
        direct = wbase.is_leaf and wbase.grad is not None and wbase.grad.is_contiguous() and wbase.dim() == 4 and \
            pwx.shape == (O1, 3) and pwx.storage_offset() == wbase.storage_offset()
        if direct:
            dwx, dst, ld, acc = None, wbase.grad, wbase.shape[1], 1
        else:
            dwx = torch.empty(O1, 3, dtype=_f32, device=dU.device)
            dst, ld, acc = dwx, 3, 0
        err = L().cmf_setconv_dwx(O1, 1.0 / M, int(st1.training), _p(sums5), _p(fwd_sums), _p(st1.a), _p(st1.mean),
                                  _p(st1.invstd), _p(dst), ld, acc, _lib.stream_ptr())
        _lib.check(err, "cmf_setconv_dwx")
        dy = None
        if ctx.needs_input_grad[1]:
            off, inv = nbr.inverse()
            dy = torch.empty(B, N, O1, dtype=_f32, device=dU.device)
            err = L().cmf_group_rows_grad_bn(B, N, O1, N * S, _p(dU), _p(z1), _p(st1.a), _p(st1.mean), _p(st1.invstd),
                                             _p(sums5) if st1.training else None, 1.0 / M, _p(off), _p(inv), _p(dy), O1,
                                             _lib.stream_ptr())
            _lib.check(err, "cmf_group_rows_grad_bn")
        return (None, dy, None, None, None, None, dwx, g["g1"], g["b1"], g["w2"], g["g2"], g["b2"], g["w3"], g["g3"],
                g["b3"], g["w4"], g["g4"], g["b4"], g["w5"], g["g5"], g["b5"], g["w6"], g["g6"], g["b6"])


def _block_forward(xyz_t, y, radius, nsample, bns, training, params, out=None, inference=False):
    """Fill a cmf_setconv_desc, allocate saved/scratch/out on the CURRENT stream and issue
    cmf_setconv_forward there.  out: optional (B*N, 64) strided view to write into."""
    import ctypes
    B, N, _ = xyz_t.shape
    dev = xyz_t.device
    wx = params[0]
    ws = [params[i].contiguous() for i in (3, 6, 9, 12, 15)]
    assert y.stride(2) == 1 and y.stride(0) == N * y.stride(1)
    xyz_t = xyz_t.contiguous()
    d = _lib.SetConvDesc()
    d.B, d.N, d.S, d.O1, d.radius, d.training = B, N, nsample, y.shape[2], radius, int(training)
    d.inference = int(bool(inference) and not training)     # no backward call will follow: nothing is kept for one
    for i, w in enumerate(ws):
        d.C[i] = w.shape[0]
        d.w[i] = w.data_ptr()
    for l, bn in enumerate(bns):
        d.eps[l] = bn.eps
        d.momentum[l] = bn.momentum if bn.momentum is not None else 0.1
        assert bn.momentum is not None, "cumulative moving average BN is not supported by the block call"
        d.gamma[l], d.beta[l] = bn.weight.data_ptr(), bn.bias.data_ptr()
        d.rmean[l], d.rvar[l] = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
        d.nbt[l] = bn.num_batches_tracked.data_ptr() if (training and bn.track_running_stats) else None
    d.xyz, d.y, d.ldy = xyz_t.data_ptr(), y.data_ptr(), y.stride(1)
    if wx.stride(1) != 1:
        wx = wx.contiguous()
    d.wx, d.ldwx = wx.data_ptr(), wx.stride(0)
    n_saved, n_f, n_b = ctypes.c_longlong(), ctypes.c_longlong(), ctypes.c_longlong()
    _lib.check(L().cmf_setconv_sizes(ctypes.addressof(d), ctypes.addressof(n_saved), ctypes.addressof(n_f),
                                     ctypes.addressof(n_b)), "cmf_setconv_sizes")
    saved = torch.empty(n_saved.value, dtype=_f32, device=dev)
    scratch = torch.empty(n_f.value, dtype=_f32, device=dev)
    if out is None:
        out = torch.empty(B * N, ws[4].shape[0], dtype=_f32, device=dev)
    d.saved, d.scratch, d.out, d.ldo = saved.data_ptr(), scratch.data_ptr(), out.data_ptr(), out.stride(0)
    _lib.check(L().cmf_setconv_forward(ctypes.addressof(d), _lib.stream_ptr()), "cmf_setconv_forward")
    return out, dict(desc=d, keep=(xyz_t, y, wx, ws, saved), n_bwd=n_b.value, params=params)


def _block_backward(state, dout, need_dy):
    """cmf_setconv_backward on the CURRENT stream.  dout: (B*N, 64) view (unit channel stride).
    -> (dy or None, 18 parameter gradients; None where the kernel accumulated into a grad sink)."""
    import ctypes
    d, (xyz_t, y, wx, ws, saved), params = state["desc"], state["keep"], state["params"]
    dev = dout.device
    B, N, O1 = d.B, d.N, d.O1
    if dout.stride(1) != 1 or dout.stride(0) % 4:
        dout = dout.contiguous()
    scratch = torch.empty(state["n_bwd"], dtype=_f32, device=dev)
    d.scratch, d.dout, d.lddout = scratch.data_ptr(), dout.data_ptr(), dout.stride(0)
    grads = [None] * 18
    # wx: the [:, :3] slice of the first conv's weight
    pwx = params[0]
    wbase = pwx._base if pwx._base is not None else pwx
    if wbase.is_leaf and wbase.grad is not None and wbase.grad.is_contiguous() and wbase.dim() == 4 and \
            pwx.shape == (O1, 3) and pwx.storage_offset() == wbase.storage_offset():
        d.dwx, d.lddwx, d.acc_wx = wbase.grad.data_ptr(), wbase.shape[1], 1
    else:
        g = torch.empty(O1, 3, dtype=_f32, device=dev)
        grads[0] = g
        d.dwx, d.lddwx, d.acc_wx = g.data_ptr(), 3, 0
    for i in range(5):                                   # layer weights 2..6 at params[3,6,9,12,15]
        sink = grad_sink(params[3 + 3 * i])
        if sink is not None:
            d.dw[i], d.acc_w[i] = sink.data_ptr(), 1
        else:
            g = torch.empty_like(ws[i])
            grads[3 + 3 * i] = g
            d.dw[i], d.acc_w[i] = g.data_ptr(), 0
    for l in range(6):                                   # (gamma, beta) of BN layer l at params[1+3l], params[2+3l]
        sg, sb = grad_sink(params[1 + 3 * l]), grad_sink(params[2 + 3 * l])
        if sg is not None and sb is not None:
            d.dgamma[l], d.dbeta[l], d.acc_bn[l] = sg.data_ptr(), sb.data_ptr(), 1
        else:
            gg, gb = torch.empty_like(params[1 + 3 * l]), torch.empty_like(params[2 + 3 * l])
            grads[1 + 3 * l], grads[2 + 3 * l] = gg, gb
            d.dgamma[l], d.dbeta[l], d.acc_bn[l] = gg.data_ptr(), gb.data_ptr(), 0
    dy = None
    if need_dy:
        dy = torch.empty(B, N, O1, dtype=_f32, device=dev)
        d.dy = dy.data_ptr()
    else:
        d.dy = None
    _lib.check(L().cmf_setconv_backward(ctypes.addressof(d), _lib.stream_ptr()), "cmf_setconv_backward")
    return dy, grads


class SetConvBlockFn(Function):
    """SetConvFn with the kernel sequence issued by ONE C-ABI call per direction (csrc/setconv_block.hip):
    same kernels, same order, same numerics -- only the ~70 Python-level launches per block are gone."""

    @staticmethod
    def forward(ctx, xyz_t, y, radius, nsample, bns, training, *params):
        B, N, _ = xyz_t.shape
        out, ctx.state = _block_forward(xyz_t, y, radius, nsample, bns, training, params, inference=not any(ctx.needs_input_grad))
        return out.view(B, N, -1)

    @staticmethod
    def backward(ctx, dout):
        st = ctx.state
        dy, grads = _block_backward(st, dout.reshape(st["desc"].B * st["desc"].N, -1), ctx.needs_input_grad[1])
        return (None, dy, None, None, None, None, *grads)


import os as _os


class EncoderPlan:
    """Everything about a MultiScaleEncoder call that does not change from step to step, built once per
    (batch, points, training, device): the four cmf_setconv_desc structs with geometry, hyper-parameters and the
    parameter / BN-buffer pointers filled in (optimizers update parameters in place), and the arena sizes.  A call
    then only sets the per-call pointers (xyz, y, saved, scratch, out) -- the Python work per encoder call drops
    from ~0.5 ms to a few tens of microseconds, which at N = 256 is what the GPU was waiting for."""

    def __init__(self, modules, B, N, O1, training, device, clouds=1):
        """clouds == 2: two calls of the (weight-shared) encoder issued CONCURRENTLY -- descriptors [0, ns) are the
        first call, [ns, 2 ns) the second.  Their BN running-statistics updates are deferred (cmf_bn_running_update,
        in call order) and the second call writes its parameter gradients to scratch tensors that are added to the
        sinks afterwards (two kernels accumulating into one buffer from different streams would race)."""
        import ctypes
        ns = len(modules)
        self.ns, self.clouds = ns, clouds
        self.n = n = ns * clouds
        self.key = (B, N, O1, bool(training), str(device), clouds)
        self.descs = (_lib.SetConvDesc * n)()
        self.keep, self.params, self.bns = [], [], []
        self.n_saved, self.n_fwd, self.n_bwd, self.n_bwd_dy = [], [], [], []
        defer = clouds > 1 and training
        for i in range(n):
            m = modules[i % ns]
            params, bns = set_conv_params(m)
            d = self.descs[i]
            wx = params[0]
            ws = [params[j] for j in (3, 6, 9, 12, 15)]
            assert all(w.is_contiguous() for w in ws) and wx.stride(1) == 1
            d.B, d.N, d.S, d.O1, d.radius, d.training = B, N, m.nsample, O1, m.radius, int(training)
            for j, w in enumerate(ws):
                d.C[j] = w.shape[0]
                d.w[j] = w.data_ptr()
            for l, bn in enumerate(bns):
                assert bn.momentum is not None, "cumulative moving average BN is not supported by the block call"
                d.eps[l], d.momentum[l] = bn.eps, bn.momentum
                d.gamma[l], d.beta[l] = bn.weight.data_ptr(), bn.bias.data_ptr()
                d.rmean[l], d.rvar[l] = (None, None) if defer else (bn.running_mean.data_ptr(), bn.running_var.data_ptr())
                d.nbt[l] = bn.num_batches_tracked.data_ptr() if (training and bn.track_running_stats and not defer) else None
            d.wx, d.ldwx = wx.data_ptr(), wx.stride(0)
            d.ldy = ns * O1                                         # y is a column slice of the stacked (B,N,ns*O1) GEMM output
            c_s, c_f, c_b, c_bd = ctypes.c_longlong(), ctypes.c_longlong(), ctypes.c_longlong(), ctypes.c_longlong()
            _lib.check(L().cmf_setconv_sizes(ctypes.addressof(d), ctypes.addressof(c_s), ctypes.addressof(c_f),
                                             ctypes.addressof(c_b)), "cmf_setconv_sizes")
            # the backward scratch is sized by path (cmflow_hip.h cmf_setconv_sizes): with an input gradient to produce (d.dy set) the
            # second encoder's blocks sum the first layer's data gradient inside the GEMM and need no M x O1 slot for it
            d.dy = 8                                                # (any non-null value: never dereferenced by the sizes call)
            _lib.check(L().cmf_setconv_sizes(ctypes.addressof(d), None, None, ctypes.addressof(c_bd)), "cmf_setconv_sizes")
            d.dy = None
            self.n_saved.append(c_s.value); self.n_fwd.append(c_f.value); self.n_bwd.append(c_b.value); self.n_bwd_dy.append(c_bd.value)
            self.params.append(params); self.bns.append(bns)
        self.co = self.params[0][15].shape[0]
        self.modules = list(modules)
        self.update_table, self.temps = None, None
        if defer:                                                   # one table entry per (scale, BN layer)
            entries = (_lib.BnUpdateEntry * (ns * 6))()
            offs = (ctypes.c_longlong * 6)()
            base = 0
            for i in range(ns):
                _lib.check(L().cmf_setconv_bn_offsets(ctypes.addressof(self.descs[i]), ctypes.addressof(offs)), "cmf_setconv_bn_offsets")
                for l, bn in enumerate(self.bns[i]):
                    e = entries[i * 6 + l]
                    e.rmean, e.rvar = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
                    e.nbt = bn.num_batches_tracked.data_ptr() if bn.track_running_stats else None
                    e.C, e.momentum, e.eps = bn.num_features, bn.momentum, bn.eps
                    e.count = float(B * N * (modules[i].nsample if l < 3 else 1))
                    e.offset = base + offs[l]
                base += (self.n_saved[i] + 63) // 64 * 64          # same rounding as off_saved below
            raw = bytes(entries)
            self.update_table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device)
            self.n_update = ns * 6
            # gradient scratch of the second call, one tensor per parameter (views for the conv weights)
            # (the xyz columns of the first conv are written into a full-size, zero-initialised weight-shaped tensor)
            self.temps = [[torch.zeros_like(t._base) if j == 0 else torch.empty_like(t) for j, t in enumerate(p)]
                          for p in self.params[:ns]]
        self.check_ptrs = self._live_ptrs()
        al = lambda v: (v + 63) // 64 * 64                          # keep every arena 256-byte aligned
        self.off_saved = [0]
        for v in self.n_saved:
            self.off_saved.append(self.off_saved[-1] + al(v))
        self.off_fwd, self.off_bwd, self.off_bwd_dy = [0], [0], [0]
        for v in self.n_fwd:
            self.off_fwd.append(self.off_fwd[-1] + al(v))
        for v in self.n_bwd:
            self.off_bwd.append(self.off_bwd[-1] + al(v))
        for v in self.n_bwd_dy:
            self.off_bwd_dy.append(self.off_bwd_dy[-1] + al(v))
        self.saved_per_cloud = self.off_saved[ns]                   # floats: the second call's arena starts here

    def _live_ptrs(self):
        """Storage of EVERY tensor the descriptors point at, as the MODULES hold them now (the views kept in self.params
        would keep a replaced storage alive and never change), plus the hyper-parameters baked into the descriptors:
        ~30 pointers per scale, compared as one list per call."""
        out = []
        for m in self.modules:
            for conv in list(m.mlp_convs) + list(m.mlp2_convs):
                out.append(conv.weight.data_ptr())
            for bn in list(m.mlp_bns) + list(m.mlp2_bns):
                out += [bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(), bn.running_var.data_ptr(),
                        bn.num_batches_tracked.data_ptr(), bn.momentum, bn.eps]
            out += [m.radius, m.nsample]
        return out

    def valid(self):
        """Parameters are updated in place by optimizers and load_state_dict; .to()/.float()/.data = ... re-allocate them."""
        return self.check_ptrs == self._live_ptrs()

    def sinks_ready(self):
        """All parameter gradients can be accumulated in place (TrainStep's flat bucket).  The sink pointers are
        written into the descriptors once; afterwards two representative .grad pointers per scale are re-checked
        (a flat bucket either keeps all of them or none)."""
        probe = []
        for params in self.params:
            for t in (params[1], params[17]):
                g = t.grad
                if g is None:
                    self._sink_probe = None
                    return False
                probe.append(g.data_ptr())
        if getattr(self, "_sink_probe", None) == probe:
            return True
        self.sink_list, self.temp_list = [], []
        for i, (d, params) in enumerate(zip(self.descs, self.params)):
            wsink = self._wx_sink(params[0])
            sinks = [grad_sink(t) for t in params[1:]]
            if wsink is None or any(x is None for x in sinks):
                self._sink_probe = None
                return False
            acc = 1
            if i >= self.ns:                                        # second call: write to scratch, added to the sinks afterwards
                tmp = self.temps[i - self.ns]
                self.sink_list += [wsink] + sinks
                self.temp_list += tmp
                wsink, sinks, acc = tmp[0], tmp[1:], 0
            d.dwx, d.lddwx, d.acc_wx = wsink.data_ptr(), params[0].stride(0), acc
            for j in range(5):
                d.dw[j], d.acc_w[j] = sinks[2 + 3 * j].data_ptr(), acc
            for l in range(6):
                d.dgamma[l], d.dbeta[l], d.acc_bn[l] = sinks[3 * l].data_ptr(), sinks[3 * l + 1].data_ptr(), acc
        self._sink_probe = probe
        return True

    @staticmethod
    def _wx_sink(pwx):
        base = pwx._base if pwx._base is not None else pwx
        if base.is_leaf and base.grad is not None and base.grad.is_contiguous() and base.dim() == 4 and \
                pwx.shape[1] == 3 and pwx.storage_offset() == base.storage_offset():
            return base.grad
        return None


def _multi_call(backward, n, plan, sp, streams, main):
    """Issue the n chains of a plan on their streams, forked from and joined back into `main` with torch stream waits around
    cmf_setconv_*_multi."""
    import ctypes
    # The per-point tails of the n blocks (three <= 64-channel layers over the B*N points: latency, not work) run as batched
    # launches on the caller's stream -- behind the joined chains in forward, in front of the fork in backward; the chains on
    # the side streams carry the neighbourhood layers only (cmflow_hip.h, cmf_setconv_tail_*).
    descs = ctypes.addressof(plan.descs)
    if backward:
        _lib.check(L().cmf_setconv_tail_backward(n, descs, main.cuda_stream), "cmf_setconv_tail_backward")
        stress_point([main])                    # (tests) between the tails and the bodies that consume their gradients
    elif NESTED_QUERIES:
        # the ball queries of all n blocks as ONE launch per encoder call (nested radii over the same centres; both clouds of the first
        # encoder), on the caller's stream in front of the fork -- the chains then start at their first layer
        _lib.check(L().cmf_setconv_queries(n, descs, main.cuda_stream), "cmf_setconv_queries")
        for i in range(n):
            plan.descs[i].idx_ready = 1
    if not backward and L().cmf_setconv_forward_bodies_batched(n, descs):
        # the bodies run in lock step as batched launches on ONE stream (narrow blocks: the first encoder): the caller's own -- no fork,
        # no join, no cross-queue dependency in front of the tails
        mp = (ctypes.c_void_p * n)(*([main.cuda_stream] * n))
        stress_point([main])
        _lib.check(L().cmf_setconv_forward_heads_multi(n, descs, ctypes.addressof(mp)), "cmf_setconv_forward_heads_multi")
        stress_point([main])
        _lib.check(L().cmf_setconv_tail_forward(n, descs, main.cuda_stream), "cmf_setconv_tail_forward")
        return
    if backward and L().cmf_setconv_backward_bodies_batched(n, descs):
        # ... and their backward bodies likewise: batched launches on the caller's stream
        mp = (ctypes.c_void_p * n)(*([main.cuda_stream] * n))
        stress_point([main])
        _lib.check(L().cmf_setconv_backward_bodies_multi(n, descs, ctypes.addressof(mp)), "cmf_setconv_backward_bodies_multi")
        stress_point([main])
        return
    for st in streams:
        st.wait_stream(main)
    stress_point(streams)
    fn = L().cmf_setconv_backward_bodies_multi if backward else L().cmf_setconv_forward_heads_multi
    _lib.check(fn(n, descs, ctypes.addressof(sp)), "cmf_setconv_*_multi")
    stress_point(streams)                       # (tests) at the END of each chain: between a block's body and its tail / the join
    for st in streams:
        main.wait_stream(st)
    if not backward:
        _lib.check(L().cmf_setconv_tail_forward(n, descs, main.cuda_stream), "cmf_setconv_tail_forward")


# Debug tap for tests: when a list, every multi-scale / dual-cloud forward call appends, per block, the ball-query indices the LIBRARY
# issued (cmf_setconv_forward's first kernel; they sit at the head of the block's `saved` arena as int32 (B, N, S)):
# (radius, nsample, idx clone).  None (default): nothing is copied.
IDX_TAP = None
NESTED_QUERIES = os.environ.get("CMF_NESTED_QUERIES", "1") != "0"       # 0: every block issues its own ball query (A/B)


def _tap_indices(plan, saved, blocks):
    if IDX_TAP is None:
        return
    for i in blocks:
        d = plan.descs[i]
        m = d.B * d.N * d.S
        idx = saved[plan.off_saved[i]:plan.off_saved[i] + m].view(torch.int32).view(d.B, d.N, d.S)
        IDX_TAP.append((float(d.radius), int(d.S), idx.clone()))


class MultiScaleBlockFn(Function):
    """The four set-conv scales of a MultiScaleEncoder (radarflow_util.py:101-118) as one autograd node and ONE
    C-ABI call per direction (cmf_setconv_forward_multi / _backward_multi: each scale is issued on its own HIP
    stream from its own host thread inside the library -- with N = 256 the ~20 / ~45 kernels of a scale run about
    as long as they take to enqueue, so one host thread cannot keep four streams fed).  The scales write disjoint
    channel slices of one (B,N,4*64) output and the input gradient comes back as one (B,N,4*O1) tensor, so the
    concat / slice-gradient kernels are gone too.  With every parameter gradient accumulated in place (grad sinks)
    the parameters are not even autograd inputs of the node.  Numerics: same kernels per scale as SetConvBlockFn."""

    @staticmethod
    def forward(ctx, xyz_t, y_all, plan, streams, sink_mode, *params):
        import ctypes
        B, N, _ = xyz_t.shape
        n, co, dev = plan.n, plan.co, xyz_t.device
        xyz_t = xyz_t.contiguous()
        assert y_all.is_contiguous() and y_all.shape[2] == n * plan.descs[0].O1
        main = torch.cuda.current_stream()
        out_all = torch.empty(B * N, n * co, dtype=_f32, device=dev)
        saved = torch.empty(plan.off_saved[-1], dtype=_f32, device=dev)
        scratch = torch.empty(plan.off_fwd[-1], dtype=_f32, device=dev)
        o1 = plan.descs[0].O1
        inference = int(not any(ctx.needs_input_grad) and not plan.descs[0].training)   # no backward call will follow
        for i in range(n):
            d = plan.descs[i]
            d.xyz, d.y = xyz_t.data_ptr(), y_all.data_ptr() + 4 * i * o1
            d.saved, d.scratch = saved.data_ptr() + 4 * plan.off_saved[i], scratch.data_ptr() + 4 * plan.off_fwd[i]
            d.out, d.ldo = out_all.data_ptr() + 4 * i * co, n * co
            d.inference = inference
        sp = (ctypes.c_void_p * n)(*[st.cuda_stream for st in streams])
        _multi_call(False, n, plan, sp, streams, main)
        _tap_indices(plan, saved, range(n))
        ctx.plan, ctx.keep, ctx.streams, ctx.sink_mode = plan, (xyz_t, y_all, saved), streams, sink_mode
        return out_all.view(B, N, n * co)

    @staticmethod
    def backward(ctx, dout):
        import ctypes
        plan, (xyz_t, y_all, saved), streams = ctx.plan, ctx.keep, ctx.streams
        n, co = plan.n, plan.co
        B, N, o1 = plan.descs[0].B, plan.descs[0].N, plan.descs[0].O1
        dev = dout.device
        dout = dout.reshape(B * N, n * co)
        if dout.stride(1) != 1 or dout.stride(0) % 4:
            dout = dout.contiguous()
        main = torch.cuda.current_stream()
        need_dy = ctx.needs_input_grad[1]
        off_bwd = plan.off_bwd_dy if need_dy else plan.off_bwd
        scratch = torch.empty(off_bwd[-1], dtype=_f32, device=dev)
        dy_all = torch.empty(B, N, n * o1, dtype=_f32, device=dev) if need_dy else None
        grads = []
        if ctx.sink_mode and not plan.sinks_ready():
            raise RuntimeError("parameter .grad buffers disappeared between forward and backward")
        for i in range(n):
            d, params = plan.descs[i], plan.params[i]
            d.xyz, d.y = xyz_t.data_ptr(), y_all.data_ptr() + 4 * i * o1
            d.saved, d.scratch = saved.data_ptr() + 4 * plan.off_saved[i], scratch.data_ptr() + 4 * off_bwd[i]
            d.dout, d.lddout = dout.data_ptr() + 4 * i * co, dout.stride(0)
            d.dy, d.lddy = (dy_all.data_ptr() + 4 * i * o1, n * o1) if need_dy else (None, 0)
            if ctx.sink_mode:                                       # sink pointers already sit in the descriptors
                continue
            plan._sink_probe = None                                 # the descriptors get per-call gradient buffers below
            g = [None] * 18
            wsink = plan._wx_sink(params[0])
            if wsink is not None:
                d.dwx, d.lddwx, d.acc_wx = wsink.data_ptr(), params[0].stride(0), 1
            else:
                if ctx.sink_mode:
                    raise RuntimeError("parameter .grad buffers disappeared between forward and backward")
                g[0] = torch.empty(o1, 3, dtype=_f32, device=dev)
                d.dwx, d.lddwx, d.acc_wx = g[0].data_ptr(), 3, 0
            for j in range(5):
                sink = grad_sink(params[3 + 3 * j])
                if sink is not None:
                    d.dw[j], d.acc_w[j] = sink.data_ptr(), 1
                else:
                    g[3 + 3 * j] = torch.empty_like(params[3 + 3 * j])
                    d.dw[j], d.acc_w[j] = g[3 + 3 * j].data_ptr(), 0
            for l in range(6):
                sg, sb = grad_sink(params[1 + 3 * l]), grad_sink(params[2 + 3 * l])
                if sg is not None and sb is not None:
                    d.dgamma[l], d.dbeta[l], d.acc_bn[l] = sg.data_ptr(), sb.data_ptr(), 1
                else:
                    g[1 + 3 * l], g[2 + 3 * l] = torch.empty_like(params[1 + 3 * l]), torch.empty_like(params[2 + 3 * l])
                    d.dgamma[l], d.dbeta[l], d.acc_bn[l] = g[1 + 3 * l].data_ptr(), g[2 + 3 * l].data_ptr(), 0
            grads += g
        if ctx.sink_mode and any(t is not None for t in grads):
            raise RuntimeError("parameter .grad buffers disappeared between forward and backward")
        sp = (ctypes.c_void_p * n)(*[st.cuda_stream for st in streams])
        _multi_call(True, n, plan, sp, streams, main)
        return (None, dy_all, None, None, None, *(() if ctx.sink_mode else grads))


class DualCloudBlockFn(Function):
    """Both calls of a weight-shared encoder (cmflow.py:72-73: mse_layer(pc1), mse_layer(pc2)) as ONE autograd node:
    2 x 4 scale chains on eight streams / host threads at once.  At N = 256 these chains are made of 5-40 us kernels on
    a fraction of the chip; doubling the number of concurrent chains is free.  What made the two calls order-dependent
    is handled outside the kernels: BN running statistics are updated afterwards in call order
    (cmf_bn_running_update), and the second call's parameter gradients go to scratch tensors that are added to the
    sinks in one multi-tensor add.  Training with in-place gradient sinks only (EncoderPlan clouds=2)."""

    @staticmethod
    def forward(ctx, xyz1, y1, xyz2, y2, plan, streams):
        import ctypes
        B, N, _ = xyz1.shape
        n, ns, co, dev = plan.n, plan.ns, plan.co, xyz1.device
        xyz = (xyz1.contiguous(), xyz2.contiguous())
        ys = (y1, y2)
        o1 = plan.descs[0].O1
        assert y1.is_contiguous() and y2.is_contiguous() and y1.shape[2] == ns * o1
        main = torch.cuda.current_stream()
        outs = [torch.empty(B * N, ns * co, dtype=_f32, device=dev) for _ in range(2)]
        saved = torch.empty(2 * plan.saved_per_cloud, dtype=_f32, device=dev)
        scratch = torch.empty(plan.off_fwd[-1], dtype=_f32, device=dev)
        for i in range(n):
            c, sc = divmod(i, ns)
            d = plan.descs[i]
            d.xyz, d.y = xyz[c].data_ptr(), ys[c].data_ptr() + 4 * sc * o1
            d.saved, d.scratch = saved.data_ptr() + 4 * plan.off_saved[i], scratch.data_ptr() + 4 * plan.off_fwd[i]
            d.out, d.ldo = outs[c].data_ptr() + 4 * sc * co, ns * co
        sp = (ctypes.c_void_p * n)(*[st.cuda_stream for st in streams])
        _multi_call(False, n, plan, sp, streams, main)
        _tap_indices(plan, saved, range(n))
        if plan.update_table is not None:                           # running statistics: first call, then second call
            _lib.check(L().cmf_bn_running_update(plan.n_update, plan.update_table.data_ptr(), 2, saved.data_ptr(),
                                                 saved.data_ptr() + 4 * plan.saved_per_cloud, _lib.stream_ptr()),
                       "cmf_bn_running_update")
        ctx.plan, ctx.keep, ctx.streams = plan, (xyz, ys, saved), streams
        return outs[0].view(B, N, ns * co), outs[1].view(B, N, ns * co)

    @staticmethod
    def backward(ctx, dout1, dout2):
        import ctypes
        plan, (xyz, ys, saved), streams = ctx.plan, ctx.keep, ctx.streams
        n, ns, co = plan.n, plan.ns, plan.co
        B, N, o1 = plan.descs[0].B, plan.descs[0].N, plan.descs[0].O1
        dev = dout1.device
        douts = []
        for g in (dout1, dout2):
            g = g.reshape(B * N, ns * co)
            douts.append(g if (g.stride(1) == 1 and g.stride(0) % 4 == 0) else g.contiguous())
        if not plan.sinks_ready():
            raise RuntimeError("parameter .grad buffers disappeared between forward and backward")
        main = torch.cuda.current_stream()
        scratch = torch.empty(plan.off_bwd[-1], dtype=_f32, device=dev)      # (sized without d.dy: never smaller than with it)
        need = (ctx.needs_input_grad[1], ctx.needs_input_grad[3])
        dys = [torch.empty(B, N, ns * o1, dtype=_f32, device=dev) if need[c] else None for c in range(2)]
        for i in range(n):
            c, sc = divmod(i, ns)
            d = plan.descs[i]
            d.xyz, d.y = xyz[c].data_ptr(), ys[c].data_ptr() + 4 * sc * o1
            d.saved, d.scratch = saved.data_ptr() + 4 * plan.off_saved[i], scratch.data_ptr() + 4 * plan.off_bwd[i]
            d.dout, d.lddout = douts[c].data_ptr() + 4 * sc * co, douts[c].stride(0)
            d.dy, d.lddy = (dys[c].data_ptr() + 4 * sc * o1, ns * o1) if need[c] else (None, 0)
        sp = (ctypes.c_void_p * n)(*[st.cuda_stream for st in streams])
        _multi_call(True, n, plan, sp, streams, main)
        torch._foreach_add_(plan.sink_list, plan.temp_list)         # the second call's parameter gradients
        return None, dys[0], None, dys[1], None, None


def dual_cloud_set_conv(encoder, modules, streams, xyz1, y1, xyz2, y2):
    """Both clouds through the weight-shared encoder concurrently; None when the preconditions (training, in-place
    gradient sinks, equal shapes) do not hold and the caller should issue the two calls one after the other."""
    if not torch.is_grad_enabled() or xyz1.shape != xyz2.shape or y1.shape != y2.shape:
        return None
    B, N, _ = xyz1.shape
    o1 = y1.shape[2] // len(modules)
    training = modules[0].mlp_bns[0].training
    if not training:
        return None
    key = (B, N, o1, True, str(xyz1.device), 2)
    plans = encoder.__dict__.setdefault("_plans", {})
    plan = plans.get(key)
    if plan is None or not plan.valid():
        plan = plans[key] = EncoderPlan(modules, B, N, o1, True, xyz1.device, clouds=2)
    if not plan.sinks_ready():
        return None
    return DualCloudBlockFn.apply(xyz1, y1.contiguous(), xyz2, y2.contiguous(), plan, streams)


def set_conv_params(module):
    w2d = lambda conv: conv.weight.view(conv.weight.shape[0], conv.weight.shape[1])
    c, b = module.mlp_convs, module.mlp_bns
    c2, b2 = module.mlp2_convs, module.mlp2_bns
    params = [w2d(c[0])[:, :3], b[0].weight, b[0].bias, w2d(c[1]), b[1].weight, b[1].bias, w2d(c[2]), b[2].weight, b[2].bias,
              w2d(c2[0]), b2[0].weight, b2[0].bias, w2d(c2[1]), b2[1].weight, b2[1].bias, w2d(c2[2]), b2[2].weight, b2[2].bias]
    return params, [b[0], b[1], b[2], b2[0], b2[1], b2[2]]


def multi_scale_set_conv(encoder, modules, streams, xyz_t, y_all):
    """modules: the PointLocalFeature scales; y_all (B,N,len*O1) the stacked hoisted first-conv features."""
    B, N, _ = xyz_t.shape
    o1 = y_all.shape[2] // len(modules)
    training = modules[0].mlp_bns[0].training
    key = (B, N, o1, bool(training), str(xyz_t.device), 1)
    plans = encoder.__dict__.setdefault("_plans", {})
    plan = plans.get(key)
    if plan is None or not plan.valid():
        plan = plans[key] = EncoderPlan(modules, B, N, o1, training, xyz_t.device)
    y_all = y_all.contiguous()
    sink_mode = torch.is_grad_enabled() and plan.sinks_ready()
    params = () if (sink_mode or not torch.is_grad_enabled()) else tuple(t for p in plan.params for t in p)
    if sink_mode and not y_all.requires_grad:
        # the node must exist for the parameter gradients even when the input features carry none (first encoder)
        y_all = y_all.detach().requires_grad_(True)
    return MultiScaleBlockFn.apply(xyz_t, y_all, plan, streams, sink_mode, *params)


def set_conv(module, xyz_t, y):
    """module: PointLocalFeature; y: (B,N,O1) view of the hoisted first-conv features."""
    w2d = lambda conv: conv.weight.view(conv.weight.shape[0], conv.weight.shape[1])
    c, b = module.mlp_convs, module.mlp_bns
    c2, b2 = module.mlp2_convs, module.mlp2_bns
    params = [w2d(c[0])[:, :3], b[0].weight, b[0].bias, w2d(c[1]), b[1].weight, b[1].bias, w2d(c[2]), b[2].weight, b[2].bias,
              w2d(c2[0]), b2[0].weight, b2[0].bias, w2d(c2[1]), b2[1].weight, b2[1].bias, w2d(c2[2]), b2[2].weight, b2[2].bias]
    bns = [b[0], b[1], b[2], b2[0], b2[1], b2[2]]
    fn = SetConvBlockFn if USE_BLOCK_CALLS else SetConvFn
    return fn.apply(xyz_t, y, module.radius, module.nsample, bns, b[0].training, *params)


USE_BLOCK_CALLS = True


# ---- cost volume (FeatureCorrelator) --------------------------------------------------------------
class CostVolumeMLPFn(Function):
    """Point-to-patch MLP of radarflow_util.py:207-221 with the first conv hoisted:
        x1 = leaky(p1[n] + p2[idx] + w3 . dxyz)   (p1 carries the bias),  x2 = leaky(x1 W2^T + b2),
        x3 = leaky(x2 W3^T + b3)                  -> x3 (B,N,K,C), dxyz (B,N,K,4)
    p1 (B,N1,C), p2 (B,N2,C) per-point GEMM outputs; wd (C,3) the direction columns of the first conv."""

    @staticmethod
    def forward(ctx, xyz1_t, xyz2_t, p1, p2, nbr, wd, w2, b2, w3, b3, preact_grad=False):
        # preact_grad: the consumer of x3 (WeightedKSumFn with leaky=True) already applies leaky'(x3), i.e. the
        # incoming gradient is w.r.t. the pre-activation z3
        ctx.preact_grad = preact_grad
        ctx.wd_in = wd
        B, N1, C = p1.shape
        K = nbr.S
        p1c, p2c = p1.contiguous(), p2.contiguous()
        x1, dxyz, _ = group_affine(p2c, p1c, xyz2_t, xyz1_t, wd, nbr.idx, act=2, stats=False)
        M = B * N1 * K
        x1 = x1.view(M, C)
        w2, w3 = w2.contiguous(), w3.contiguous()
        x2 = gemm(x1, w2, bias=b2, act=2)
        x3 = gemm(x2, w3, bias=b3, act=2)
        ctx.saved = (nbr, dxyz, x1, x2, x3, w2, w3, (B, N1, K, C), p2.shape[1])
        ctx.params = (w2, w3, b2)
        ctx.mark_non_differentiable(dxyz)
        return x3.view(B, N1, K, -1), dxyz

    @staticmethod
    def backward(ctx, dx3, _ddxyz):
        nbr, dxyz, x1, x2, x3, w2, w3, (B, N1, K, C), N2 = ctx.saved
        M = B * N1 * K
        off, inv = nbr.inverse()            # wants most of a CU's LDS: issued ahead of the GEMMs, not behind them
        dx3 = dx3.reshape(M, -1)
        # leaky'(z) has the sign of the stored activation
        dz3 = dx3.contiguous() if ctx.preact_grad else torch.where(x3 > 0, dx3, 0.1 * dx3)
        db3 = None if ctx.preact_grad else dz3.sum(0)             # preact_grad: the consumer returns it (WeightedKSumFn x_bias)
        dw3 = gemm_dw(dz3, x2, w=ctx.params[1])
        dz2, part2 = gemm(dz3, w3, b_t=False, bwd=(2, x2), stats=True)    # column sums of dz2 = db2 from the epilogue
        sink2 = grad_sink(ctx.params[2])
        db2 = colsum(part2, sink2, None)[0]
        if sink2 is not None:
            db2 = None
        dw2 = gemm_dw(dz2, x1, w=ctx.params[0])
        # the direction columns' gradient dwd[c,k] = sum_rows dz1[row,c] * dxyz[row,k] comes out of the same epilogue
        # (as a 512 x 4 GEMM over the 131072 rows it re-read dz1: 154 us)
        dz1, part1 = gemm(dz2, w2, b_t=False, bwd=(2, x1, None, None, None, None, dxyz.view(-1, 4)), stats=True)
        dwd = colsum_n(part1)[2:5].t()
        sink_d = grad_sink(ctx.wd_in)
        if sink_d is not None and sink_d.shape == dwd.shape:
            sink_d.add_(dwd)                # (a column block of the first conv's gradient buffer: one small add instead of zeros + copy + add)
            dwd = None
        dp1 = dz1.view(B, N1, K, C).sum(dim=2)
        dp2 = torch.empty(B, N2, C, dtype=_f32, device=dz1.device)
        err = L().cmf_group_rows_grad(B, N2, C, C, N1 * K, 0, _p(dz1), _p(off), _p(inv), _p(dp2), _lib.stream_ptr())
        _lib.check(err, "cmf_group_rows_grad")
        return None, None, dp1, dp2, None, dwd, dw2, db2, dw3, db3, None


class WeightedKSumFn(Function):
    """cost[b,n,:] = sum_k weights[b,n,k,:] * x[b,n,k,:] (radarflow_util.py:219-221) or, with nbr given,
    sum_k weights[b,n,k,:] * x[b, nbr[b,n,k], :] (:234-236, x = per-point rows, the grouped tensor is never
    written) -- cmf_weighted_ksum / cmf_weighted_ksum_grad (csrc/wsum.hip).  leaky=True: x is a stored
    LeakyReLU(0.1) activation and the gradient returned for it is the one w.r.t. its pre-activation."""

    @staticmethod
    def forward(ctx, weights, x, nbr, leaky, relu_w=False, x_bias=None):
        # relu_w: weights is a stored ReLU activation whose producer (LinearFn preact_grad=True) expects the gradient
        # w.r.t. its pre-activation.  x_bias: the bias of the layer that produced x; its gradient (column sums of the
        # pre-activation gradient) comes out of the same kernel instead of a separate reduction over (M*K, C)
        ctx.x_bias = x_bias
        B, N1, K, C = weights.shape
        weights, x = weights.contiguous(), x.contiguous()
        out = torch.empty(B, N1, C, dtype=_f32, device=weights.device)
        idx = _lib.dev_ptr(nbr.idx, torch.int32) if nbr is not None else None
        n_src = nbr.n if nbr is not None else 0
        _lib.check(L().cmf_weighted_ksum(B * N1, K, C, N1, n_src, _lib.dev_ptr(weights, _f32), _lib.dev_ptr(x, _f32), idx,
                                         _p(out), _lib.stream_ptr()), "cmf_weighted_ksum")
        ctx.saved = (weights, x, nbr, int(bool(leaky)) | (2 if relu_w else 0))
        return out

    @staticmethod
    def backward(ctx, dcost):
        weights, x, nbr, leaky = ctx.saved
        B, N1, K, C = weights.shape
        dcost = dcost.contiguous()
        need_w, need_x = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        dw = torch.empty_like(weights) if need_w else None
        dx = torch.empty(B, N1, K, C, dtype=_f32, device=weights.device) if need_x else None
        idx = _lib.dev_ptr(nbr.idx, torch.int32) if nbr is not None else None
        n_src = nbr.n if nbr is not None else 0
        part, db = None, None
        want_db = ctx.x_bias is not None and len(ctx.needs_input_grad) > 5 and ctx.needs_input_grad[5] and need_x
        tiles = L().cmf_weighted_ksum_grad_tiles(C) if want_db else 0
        if tiles > 0:
            part = torch.empty(tiles, 1, C, dtype=_f32, device=weights.device)
        _lib.check(L().cmf_weighted_ksum_grad(B * N1, K, C, N1, n_src, int(leaky), _p(dcost), _p(weights), _p(x), idx,
                                              _p(dw), _p(dx), _p(part), _lib.stream_ptr()), "cmf_weighted_ksum_grad")
        if want_db:
            if part is not None:
                sink = grad_sink(ctx.x_bias)
                sums = colsum_n(part, C, sink, None)                 # fixed-order sum of the per-workgroup partials
                db = None if sink is not None else sums.view(-1)
            else:
                db = dx.view(-1, C).sum(0)
        if nbr is not None and need_x:                       # scatter the per-slot gradients back to the points
            off, inv = nbr.inverse()
            dp = torch.empty(B, nbr.n, C, dtype=_f32, device=weights.device)
            _lib.check(L().cmf_group_rows_grad(B, nbr.n, C, C, N1 * K, 0, _p(dx), _p(off), _p(inv), _p(dp), _lib.stream_ptr()),
                       "cmf_group_rows_grad")
            dx = dp
        return dw, dx, None, None, None, db


class WeightNetKSumFn(Function):
    """cost[b,n,:] = sum_k relu(h[b,n,k,:] @ wl^T + bl) * x[...]: WeightedKSumFn with the last WeightNet layer
    (radarflow_util.py:307-318, Conv2d(8, C, 1) + ReLU) recomputed inside the kernels, so the (B,N,K,C) weights and their
    gradient are never materialised -- cmf_weightnet_ksum / cmf_weightnet_ksum_grad (csrc/wsum.hip).
    h (B,N,K,8) hidden activation; x (B,N,K,C) dense, or (B,n_src,C) per-point rows with nbr; leaky / x_bias as in
    WeightedKSumFn."""

    @staticmethod
    def supported(C, J):
        return J == 8 and L().cmf_weightnet_ksum_tiles(int(C)) > 0

    @staticmethod
    def forward(ctx, h, wl, bl, x, nbr, leaky, x_bias=None, h_bias=None):
        # h_bias: h is the stored ReLU activation of a layer with this bias (built with preact_grad=True and a detached
        # bias): the gradient returned for h is w.r.t. that layer's pre-activation and its column sums go to h_bias
        B, N1, K, J = h.shape
        C = wl.shape[0]
        h, wl, bl, x = h.contiguous(), wl.contiguous(), bl.contiguous(), x.contiguous()
        out = torch.empty(B, N1, C, dtype=_f32, device=h.device)
        idx = _lib.dev_ptr(nbr.idx, torch.int32) if nbr is not None else None
        n_src = nbr.n if nbr is not None else 0
        _lib.check(L().cmf_weightnet_ksum(B * N1, K, C, N1, n_src, _p(h), _p(wl), _p(bl), _p(x), idx, _p(out), _lib.stream_ptr()),
                   "cmf_weightnet_ksum")
        ctx.saved = (h, wl, bl, x, nbr, int(bool(leaky)) | (4 if h_bias is not None else 0))
        ctx.params = (wl, bl, x_bias, h_bias)
        return out

    @staticmethod
    def backward(ctx, dcost):
        h, wl, bl, x, nbr, leaky = ctx.saved
        B, N1, K, J = h.shape
        C = wl.shape[0]
        dev = h.device
        if not (dcost.stride(2) == 1 and dcost.stride(0) == N1 * dcost.stride(1) and dcost.stride(1) % 4 == 0
                and dcost.data_ptr() % 16 == 0):
            dcost = dcost.contiguous()                                  # a column block of a wider gradient is read in place
        dx = torch.empty(B, N1, K, C, dtype=_f32, device=dev)
        dh = torch.empty_like(h)
        part = torch.empty(L().cmf_weightnet_ksum_tiles(C), C * (J + 2) + J, dtype=_f32, device=dev)
        idx = _lib.dev_ptr(nbr.idx, torch.int32) if nbr is not None else None
        n_src = nbr.n if nbr is not None else 0
        _lib.check(L().cmf_weightnet_ksum_grad(B * N1, K, C, N1, n_src, leaky, dcost.data_ptr(), dcost.stride(1), _p(h), _p(wl), _p(bl), _p(x), idx,
                                               _p(dx), _p(dh), _p(part), _lib.stream_ptr()), "cmf_weightnet_ksum_grad")
        sums = colsum_n(part)                                            # fixed-order sum of the per-workgroup partials
        grads = [sums[:C * J].view(C, J), sums[C * J:C * J + C], sums[C * J + C:C * J + 2 * C] if ctx.params[2] is not None else None,
                 sums[C * (J + 2):] if ctx.params[3] is not None else None]
        sinks = [grad_sink(p) if p is not None else None for p in ctx.params]
        pairs = [(s.view(-1), g.reshape(-1)) for s, g in zip(sinks, grads) if s is not None and g is not None]
        if pairs:                                                        # straight into the parameters' .grad, one launch
            torch._foreach_add_([s for s, _ in pairs], [g for _, g in pairs])
        dwl, dbl, dxb, dhb = [None if (s is not None or g is None) else g for s, g in zip(sinks, grads)]
        if nbr is not None:                                              # scatter the per-slot gradients back to the points
            off, inv = nbr.inverse()
            dp = torch.empty(B, nbr.n, C, dtype=_f32, device=dev)
            _lib.check(L().cmf_group_rows_grad(B, nbr.n, C, C, N1 * K, 0, _p(dx), _p(off), _p(inv), _p(dp), _lib.stream_ptr()),
                       "cmf_group_rows_grad")
            dx = dp
        return dh, dwl, dbl, dx, None, None, dxb, dhb


class GlobalMaxCatFn(Function):
    """(B,N,C) -> (B,N,2C) = cat(f, max over the points broadcast to every point): Backbone's global feature
    (cmflow.py:76-81,89-91) in one kernel per direction -- cmf_global_max_cat(_grad), csrc/pointwise.hip.  The incoming
    gradient may be a column block of a wider tensor (row-strided view); it is read in place."""

    @staticmethod
    def forward(ctx, f):
        B, N, C = f.shape
        f = f.contiguous()
        out = torch.empty(B, N, 2 * C, dtype=_f32, device=f.device)
        arg = torch.empty(B, C, dtype=torch.int32, device=f.device)
        _lib.check(L().cmf_global_max_cat(B, N, C, _p(f), C, _p(out), 2 * C, _p(arg), _lib.stream_ptr()), "cmf_global_max_cat")
        ctx.arg = arg
        return out

    @staticmethod
    def backward(ctx, dout):
        B, C = ctx.arg.shape
        N = dout.shape[1]
        if not (dout.stride(2) == 1 and dout.stride(0) == N * dout.stride(1) and dout.stride(1) % 4 == 0 and dout.data_ptr() % 16 == 0):
            dout = dout.contiguous()
        df = torch.empty(B, N, C, dtype=_f32, device=dout.device)
        _lib.check(L().cmf_global_max_cat_grad(B, N, C, dout.data_ptr(), dout.stride(1), _p(ctx.arg), _p(df), C, _lib.stream_ptr()),
                   "cmf_global_max_cat_grad")
        return df


def global_max_cat(f):
    return GlobalMaxCatFn.apply(f)
