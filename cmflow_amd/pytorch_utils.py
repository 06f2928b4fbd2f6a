"""``lib/pytorch_utils.py`` of the reference: ``SharedMLP`` (:5-33), ``Conv1d`` / ``Conv2d`` (:36-100,126-198), ``BatchNorm1d`` /
``BatchNorm2d`` (:104-123) and ``FC`` (:200-236) -- conv / linear -> BatchNorm (wrapped once more, so parameter names read
``layer0.bn.bn.weight``) -> activation, or in the pre-activation order.  Same constructor arguments and the same state_dict keys,
so checkpoints of reference-built modules load unchanged.

On device tensors every 1x1 convolution / linear layer of these modules runs through the library's GEMM: the ``(B, C, ...)`` input
becomes a row-major ``[positions, channels]`` matrix and
  * the default configuration, [1x1 conv + BatchNorm + ReLU] stacks, goes through ``cmf_mlp_forward / _backward``
    (fused_blocks.mlp_chain: GEMM with train-mode statistics, fold with the running-stat update, hand-written backward), up to four
    layers per call; layers without BatchNorm (conv + bias + ReLU / no activation) through ``cmf_gemm`` with its bias / activation epilogue;
  * the other configurations (``preact=True``: norm and activation IN FRONT of the conv; ``first``: a plain first conv; instance norm;
    activations other than ReLU; BatchNorm widths that are not a power of two) run their 1x1 conv through ``cmf_gemm`` and the
    normalisation / activation around it as the torch modules of the same Sequential, in the reference's order.
Kernel sizes other than 1 (unused by the reference's models) keep ``nn.Conv1d`` / ``nn.Conv2d``.
"""
from typing import List

import torch
import torch.nn as nn
import torch.nn.functional as F


class _BNBase(nn.Sequential):
    """lib/pytorch_utils.py:104-111: a Sequential around the torch BatchNorm (hence the doubled ``bn.bn`` key)."""

    def __init__(self, in_size, batch_norm=None, name=""):
        super().__init__()
        self.add_module(name + "bn", batch_norm(in_size))
        nn.init.constant_(self[0].weight, 1.0)
        nn.init.constant_(self[0].bias, 0)


class BatchNorm1d(_BNBase):
    """lib/pytorch_utils.py:114-117"""

    def __init__(self, in_size: int, *, name: str = ""):
        super().__init__(in_size, batch_norm=nn.BatchNorm1d, name=name)


class BatchNorm2d(_BNBase):
    """lib/pytorch_utils.py:120-123"""

    def __init__(self, in_size: int, name: str = ""):
        super().__init__(in_size, batch_norm=nn.BatchNorm2d, name=name)


def _pow2(c):
    return 4 <= c <= 1024 and (c & (c - 1)) == 0


def _rows_of(x):
    """(B, C, ...) -> [positions, channels] rows (16-byte rows: zero columns behind C when C % 4) and the shape to restore."""
    from . import fused_blocks as FB
    C = x.shape[1]
    perm = (0,) + tuple(range(2, x.dim())) + (1,)
    rows = x.permute(*perm).reshape(-1, C)
    return (FB._pad_cols(rows) if C % 4 else rows.contiguous()), x.permute(*perm).shape[:-1]


def _from_rows(rows, lead):
    """[positions, C'] -> (B, C', ...)"""
    y = rows.view(*lead, rows.shape[1])
    back = (0, y.dim() - 1) + tuple(range(1, y.dim() - 1))
    return y.permute(*back).contiguous()


def _gemm_conv(rows, conv, act):
    """rows [positions, cin (+ zero pad)] through a 1x1 conv / linear layer's weight on cmf_gemm; act: 0 none, 1 ReLU."""
    from . import fused_blocks as FB
    w = conv.weight.view(conv.weight.shape[0], -1)
    if w.shape[1] != rows.shape[1]:                                   # zero columns for the padded input channels
        w = F.pad(w, (0, rows.shape[1] - w.shape[1]))
    return FB.linear(rows, w, conv.bias, act=act)


class _ConvBase(nn.Sequential):
    """lib/pytorch_utils.py:35-101 (conv -> BN -> activation -> instance norm, or with ``preact`` the norm / activation first)."""

    def __init__(self, in_size, out_size, kernel_size, stride, padding, activation, bn, init, conv=None, batch_norm=None, bias=True,
                 preact=False, name="", instance_norm=False, instance_norm_func=None):
        super().__init__()
        bias = bias and (not bn)
        conv_unit = conv(in_size, out_size, kernel_size=kernel_size, stride=stride, padding=padding, bias=bias)
        init(conv_unit.weight)
        if bias:
            nn.init.constant_(conv_unit.bias, 0)
        norm_size = in_size if preact else out_size

        def tail_or_head():
            if bn:
                self.add_module(name + 'bn', batch_norm(norm_size))
            if activation is not None:
                self.add_module(name + 'activation', activation)
            if not bn and instance_norm:
                self.add_module(name + 'in', instance_norm_func(norm_size, affine=False, track_running_stats=False))

        if preact:
            tail_or_head()
        self.add_module(name + 'conv', conv_unit)
        if not preact:
            tail_or_head()
        self._preact = preact

    def _parts(self):
        conv = bn = act = inorm = None
        for m in self.children():
            if isinstance(m, (nn.Conv1d, nn.Conv2d)):
                conv = m
            elif isinstance(m, _BNBase):
                bn = m[0]
            elif isinstance(m, (nn.InstanceNorm1d, nn.InstanceNorm2d)):
                inorm = m
            else:
                act = m
        return conv, bn, act, inorm

    def _pointwise(self, conv):
        one = lambda v: all(int(t) == 1 for t in (v if isinstance(v, tuple) else (v,)))
        zero = lambda v: all(int(t) == 0 for t in (v if isinstance(v, tuple) else (v,)))
        return one(conv.kernel_size) and one(conv.stride) and zero(conv.padding) and one(conv.dilation) and conv.groups == 1

    def forward(self, x):
        conv, bn, act, inorm = self._parts()
        if not (x.is_cuda and x.dtype == torch.float32 and self._pointwise(conv)):
            return super().forward(x)
        from . import fused_blocks as FB
        if not self._preact and bn is not None and inorm is None and isinstance(act, nn.ReLU) and _pow2(conv.weight.shape[0]):
            rows, lead = _rows_of(x)                                  # the default stack: conv + BN + ReLU in one library call
            w = conv.weight.view(conv.weight.shape[0], -1)
            if w.shape[1] != rows.shape[1]:
                w = F.pad(w, (0, rows.shape[1] - w.shape[1]))
            return _from_rows(FB.mlp_chain_w(rows, [(w, bn)], bn.training), lead)
        if not self._preact and bn is None and inorm is None and (act is None or isinstance(act, nn.ReLU)):
            rows, lead = _rows_of(x)                                  # conv + bias (+ ReLU) in the GEMM's epilogue
            return _from_rows(_gemm_conv(rows, conv, 1 if act is not None else 0), lead)
        y = x                                                         # everything else: the Sequential's own order, the conv on cmf_gemm
        for m in self.children():
            if m is conv:
                rows, lead = _rows_of(y)
                y = _from_rows(_gemm_conv(rows, conv, 0), lead)
            else:
                y = m(y)
        return y


class Conv1d(_ConvBase):
    """lib/pytorch_utils.py:126-160"""

    def __init__(self, in_size: int, out_size: int, *, kernel_size: int = 1, stride: int = 1, padding: int = 0,
                 activation=nn.ReLU(inplace=True), bn: bool = False, init=nn.init.kaiming_normal_, bias: bool = True,
                 preact: bool = False, name: str = "", instance_norm=False):
        super().__init__(in_size, out_size, kernel_size, stride, padding, activation, bn, init, conv=nn.Conv1d, batch_norm=BatchNorm1d,
                         bias=bias, preact=preact, name=name, instance_norm=instance_norm, instance_norm_func=nn.InstanceNorm1d)


class Conv2d(_ConvBase):
    """lib/pytorch_utils.py:163-197 (kernel (1,1), stride 1, no padding, kaiming-normal init by default)."""

    def __init__(self, in_size: int, out_size: int, *, kernel_size=(1, 1), stride=(1, 1), padding=(0, 0),
                 activation=nn.ReLU(inplace=True), bn: bool = False, init=nn.init.kaiming_normal_, bias: bool = True,
                 preact: bool = False, name: str = "", instance_norm: bool = False):
        super().__init__(in_size, out_size, kernel_size, stride, padding, activation, bn, init, conv=nn.Conv2d, batch_norm=BatchNorm2d,
                         bias=bias, preact=preact, name=name, instance_norm=instance_norm, instance_norm_func=nn.InstanceNorm2d)


class FC(nn.Sequential):
    """lib/pytorch_utils.py:200-236: Linear -> BatchNorm1d -> activation (or the pre-activation order)."""

    def __init__(self, in_size: int, out_size: int, *, activation=nn.ReLU(inplace=True), bn: bool = False, init=None,
                 preact: bool = False, name: str = ""):
        super().__init__()
        fc = nn.Linear(in_size, out_size, bias=not bn)
        if init is not None:
            init(fc.weight)
        if not bn:
            nn.init.constant_(fc.bias, 0)
        norm_size = in_size if preact else out_size

        def tail_or_head():
            if bn:
                self.add_module(name + 'bn', BatchNorm1d(norm_size))
            if activation is not None:
                self.add_module(name + 'activation', activation)

        if preact:
            tail_or_head()
        self.add_module(name + 'fc', fc)
        if not preact:
            tail_or_head()
        self._preact = preact

    def forward(self, x):
        fc = next(m for m in self.children() if isinstance(m, nn.Linear))
        bn = next((m[0] for m in self.children() if isinstance(m, _BNBase)), None)
        act = next((m for m in self.children() if not isinstance(m, (nn.Linear, _BNBase))), None)
        if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2):
            return super().forward(x)
        from . import fused_blocks as FB
        pad = lambda t: FB._pad_cols(t) if t.shape[1] % 4 else t.contiguous()
        if not self._preact and bn is not None and isinstance(act, nn.ReLU) and _pow2(fc.weight.shape[0]):
            rows = pad(x)
            w = fc.weight if fc.weight.shape[1] == rows.shape[1] else F.pad(fc.weight, (0, rows.shape[1] - fc.weight.shape[1]))
            return FB.mlp_chain_w(rows, [(w, bn)], bn.training)
        if not self._preact and bn is None and (act is None or isinstance(act, nn.ReLU)):
            return _gemm_conv(pad(x), fc, 1 if act is not None else 0)
        y = x
        for m in self.children():
            y = _gemm_conv(pad(y), fc, 0) if m is fc else m(y)
        return y


class SharedMLP(nn.Sequential):
    """lib/pytorch_utils.py:5-33"""

    def __init__(self, args: List[int], *, bn: bool = False, activation=nn.ReLU(inplace=True), preact: bool = False,
                 first: bool = False, name: str = "", instance_norm: bool = False):
        super().__init__()
        for i in range(len(args) - 1):
            plain = first and preact and i == 0
            self.add_module(name + 'layer{}'.format(i),
                            Conv2d(args[i], args[i + 1], bn=(not plain) and bn, activation=None if plain else activation,
                                   preact=preact, instance_norm=instance_norm))
        # the library's BN-backward passes tile rows x channels with C / 4 a power of two (csrc/pointwise.hip tile_ok): other widths
        # (e.g. 48, 96) take the per-layer path (conv on cmf_gemm, BatchNorm as the torch module)
        pow2 = all(_pow2(c) for c in args[1:]) if bn else True
        self._native = (not preact) and (not instance_norm) and isinstance(activation, nn.ReLU) and pow2

    @staticmethod
    def _parts(layer):
        conv, bn, _, _ = layer._parts()
        return conv, bn

    def forward(self, x):
        if not (x.is_cuda and x.dim() == 4):
            return super().forward(x)
        if not self._native:
            # pre-activation / `first` / instance-norm / other activations / odd widths: layer by layer, every 1x1 conv on cmf_gemm
            # (Conv2d.forward above) with its norm and activation in the reference's order
            for layer in self.children():
                x = layer(x)
            return x
        from . import fused_blocks as FB
        B, C, P, S = x.shape
        rows = x.permute(0, 2, 3, 1).reshape(B * P * S, C)             # [positions, channels]
        rows = FB._pad_cols(rows) if C % 4 else rows.contiguous()
        layers = [self._parts(l) for l in self.children()]
        i = 0
        while i < len(layers):
            conv, bn = layers[i]
            w = conv.weight.view(conv.weight.shape[0], conv.weight.shape[1])
            if w.shape[1] != rows.shape[1]:                               # zero columns for the padded input channels
                w = F.pad(w, (0, rows.shape[1] - w.shape[1]))
            if bn is None:                                                # conv + bias + ReLU
                rows = FB.linear(rows, w, conv.bias, act=1)
                i += 1
                continue
            group = [(conv, bn, w)]                                       # up to four [conv + BN + ReLU] layers per library call
            while len(group) < 4 and i + len(group) < len(layers) and layers[i + len(group)][1] is not None:
                c2, b2 = layers[i + len(group)]
                group.append((c2, b2, c2.weight.view(c2.weight.shape[0], c2.weight.shape[1])))
            rows = FB.mlp_chain_w(rows, [(wt, b) for _, b, wt in group], group[0][1].training)
            i += len(group)
        return rows.view(B, P, S, rows.shape[1]).permute(0, 3, 1, 2).contiguous()
