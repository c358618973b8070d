"""`Layers` factory with the reference's names and state_dict keys, running on the gfx950 kernels.

Mirrors utilities/layers.py of the reference (factory names, argument meaning, parameter names / Sequential
indices, default initialisation and RNG consumption order) but every block executes as fused HIP stages:

    conv (implicit GEMM, MFMA)  ->  batch statistics  ->  normalise + GLU / LeakyReLU / residual add

`nn.Upsample` never runs: an upBlock's nearest-x2 + conv3x3 is folded into four parity-class 2x2 convolutions on
the low-resolution input (include/agan.h), so the 4x larger tensor is never written or read.
"""
from __future__ import annotations

from math import floor
from typing import List, Optional, Sequence

import torch
from torch import Tensor, nn

from ..backend import functional as HF
from ..backend import lib as L


class HipConv2d(nn.Conv2d):
    """Parameter holder with nn.Conv2d's init; forward is the HIP implicit-GEMM conv."""

    def __init__(self, cin: int, cout: int, k: int, stride: int, pad: int, bias: bool, kind: Optional[str] = None):
        super().__init__(cin, cout, kernel_size=k, stride=stride, padding=pad, bias=bias)
        if kind is None:
            if stride == 1 and pad == (k - 1) // 2 and k % 2 == 1:
                kind = "same"
            elif stride == 2 and k == 4 and pad == 1:
                kind = "down"
            else:
                raise NotImplementedError(f"conv k={k} s={stride} p={pad} is not on the AttnGAN hot path")
        self.kind = kind
        self._packed = {}        # packed-weight cache (layout copies for the kernels; not part of state_dict)

    def forward(self, x: Tensor, act: int = L.ACT_NONE, handoff_out=None, handoff_in=None) -> Tensor:
        """`act` (only where HF.conv_fuses_activation allows it) applies the following activation in the conv epilogue;
        the handoffs link such a conv to its single consumer (HF.ActHandoff)."""
        return HF.conv2d(x, self.weight, self.bias, self.kind, self._packed, HF.grad_dst(self.weight),
                         HF.grad_dst(self.bias) if self.bias is not None else None, act, handoff_out, handoff_in)


class HipLinear(nn.Linear):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._packed = {}

    def forward(self, x: Tensor) -> Tensor:
        return HF.linear(x, self.weight, self.bias, self._packed, HF.grad_dst(self.weight),
                         HF.grad_dst(self.bias) if self.bias is not None else None)


class _BNState:
    """mixin: fused normalise+activation using this module's affine parameters and running statistics."""

    def fused(self, x: Tensor, act: int, residual: Optional[Tensor] = None) -> Tensor:
        if self.momentum is None or not self.track_running_stats or not self.affine:
            raise NotImplementedError("only the default BatchNorm configuration is on the hot path")
        return HF.bn_act(x, self.weight, self.bias, self.running_mean, self.running_var, self.num_batches_tracked,
                         self.training, act, residual, self.eps, self.momentum)


class HipBatchNorm2d(nn.BatchNorm2d, _BNState):
    def forward(self, x: Tensor) -> Tensor:
        return self.fused(x, L.ACT_NONE)


class HipBatchNorm1d(nn.BatchNorm1d, _BNState):
    def forward(self, x: Tensor) -> Tensor:
        return self.fused(x, L.ACT_NONE)


class GLU(nn.Module):
    """Gated linear unit over channel halves (reference layers.py:13-26)."""

    def forward(self, x: Tensor) -> Tensor:
        assert x.size(1) % 2 == 0, 'channels dont divide 2!'
        return HF.glu(x)


class _Stage:
    __slots__ = ("conv", "bn", "act")

    def __init__(self, conv: str, bn: Optional[str], act: int):
        self.conv, self.bn, self.act = conv, bn, act


class FusedChain(nn.Module):
    """A reference `nn.Sequential` of conv/BN/activation layers, executed stage by stage on the HIP kernels.

    Children are registered under the reference Sequential's integer names so `state_dict()` keys match
    (e.g. upBlock -> '1.weight', '2.weight', '2.running_mean', ...).
    """

    def __init__(self):
        super().__init__()
        self._stages: List[_Stage] = []

    def add_stage(self, conv_idx: int, conv: HipConv2d, bn_idx: Optional[int], act: int) -> "FusedChain":
        self.add_module(str(conv_idx), conv)
        if bn_idx is not None:
            self.add_module(str(bn_idx), HipBatchNorm2d(conv.out_channels))
        self._stages.append(_Stage(str(conv_idx), None if bn_idx is None else str(bn_idx), act))
        return self

    def run(self, x: Tensor, residual: Optional[Tensor] = None) -> Tensor:
        last = len(self._stages) - 1
        handoff = None
        for i, st in enumerate(self._stages):
            conv = getattr(self, st.conv)
            if st.bn is None and HF.conv_fuses_activation(st.act, conv.out_channels):
                # conv + LeakyReLU in one kernel (no BatchNorm in between); if a conv of this chain is its only consumer, that
                # conv's dgrad epilogue does the activation backward as well
                handoff = HF.ActHandoff() if i < last else None
                x = conv(x, st.act, handoff, None)
                continue
            y = conv(x, L.ACT_NONE, None, handoff)
            handoff = None
            if st.bn is not None:
                x = getattr(self, st.bn).fused(y, st.act, residual if i == last else None)
            elif st.act != L.ACT_NONE:
                x = HF.activation(y, st.act)
            else:
                x = y
        return x

    def forward(self, x: Tensor) -> Tensor:
        return self.run(x)


class _ReluChain(FusedChain):
    def forward(self, x: Tensor) -> Tensor:
        return torch.relu(self.run(x))


class ResBlock(nn.Module):
    """conv3x3 C->2C, BN, GLU, conv3x3 C->C, BN, += x   (reference layers.py:156-176; keys block.{0,1,3,4}.*)."""

    def __init__(self, channel_num: int):
        super().__init__()
        c = channel_num
        self.block = (FusedChain()
                      .add_stage(0, HipConv2d(c, c * 2, 3, 1, 1, False), 1, L.ACT_GLU)
                      .add_stage(3, HipConv2d(c, c, 3, 1, 1, False), 4, L.ACT_NONE))

    def forward(self, x: Tensor) -> Tensor:
        return self.block.run(x, residual=x)      # the residual add rides in the last normalise kernel


class Layers:
    """Static factories, same names/arguments as the reference's `Layers`."""

    @staticmethod
    def GLU() -> GLU:
        return GLU()

    @staticmethod
    def calculate_out_hw(hw: int, k: int, s: int, p=0) -> int:
        return floor(((hw + 2 * p - k) / s) + 1)

    @staticmethod
    def conv(in_channels: int, out_channels: int, in_hw: int, out_hw: int, max_kern=4, max_stride=3, max_pad=3) -> nn.Conv2d:
        cands = [(k, s, p) for k in range(1, max_kern + 1) for s in range(1, max_stride + 1) for p in range(max_pad + 1)
                 if Layers.calculate_out_hw(in_hw, k, s, p) == out_hw]
        if not cands:
            raise Exception('Could not find valid parameters to produce output hw')
        k, s, p = max(cands, key=lambda t: (t[0], t[2], t[1]))
        return HipConv2d(in_channels, out_channels, k, s, p, False)

    @staticmethod
    def conv1x1(in_planes: int, out_planes: int, bias=False) -> nn.Conv2d:
        return HipConv2d(in_planes, out_planes, 1, 1, 0, bias)

    @staticmethod
    def conv3x3(in_planes: int, out_planes: int) -> nn.Conv2d:
        return HipConv2d(in_planes, out_planes, 3, 1, 1, False)

    @staticmethod
    def conv4x4DownSpatial(in_planes: int, out_planes: int, bias=True) -> nn.Conv2d:
        return HipConv2d(in_planes, out_planes, 4, 2, 1, bias)

    @staticmethod
    def upBlock(in_planes: int, out_planes: int) -> FusedChain:
        """Upsample(x2) -> conv3x3(in, 2*out) -> BN -> GLU; keys '1.weight', '2.*' (index 0/3 hold no state)."""
        return FusedChain().add_stage(1, HipConv2d(in_planes, out_planes * 2, 3, 1, 1, False, kind="up"), 2, L.ACT_GLU)

    @staticmethod
    def upBlockReLU(in_planes: int, out_planes: int) -> FusedChain:
        """Upsample(x2) -> conv3x3(in, out) -> BN -> ReLU (reference layers.py:71-80; unused by the training path): the folded
        upsample conv and the fused BatchNorm of upBlock, followed by a stock ReLU."""
        return _ReluChain().add_stage(1, HipConv2d(in_planes, out_planes, 3, 1, 1, False, kind="up"), 2, L.ACT_NONE)

    @staticmethod
    def downBlockLeakyReLU(in_planes: int, out_planes: int) -> FusedChain:
        return FusedChain().add_stage(0, HipConv2d(in_planes, out_planes, 4, 2, 1, True), 1, L.ACT_LRELU)

    @staticmethod
    def Block3x3_relu(in_planes: int, out_planes: int) -> FusedChain:
        return FusedChain().add_stage(0, HipConv2d(in_planes, out_planes * 2, 3, 1, 1, False), 1, L.ACT_GLU)

    @staticmethod
    def Block3x3_leakRelu(in_planes: int, out_planes: int) -> FusedChain:
        return FusedChain().add_stage(0, HipConv2d(in_planes, out_planes, 3, 1, 1, False), 1, L.ACT_LRELU)

    @staticmethod
    def downBlock(in_planes: int, out_planes: int) -> FusedChain:
        return FusedChain().add_stage(0, HipConv2d(in_planes, out_planes, 4, 2, 1, False), 1, L.ACT_LRELU)

    @staticmethod
    def encode_image_by_16times(df_dims: int) -> FusedChain:
        """Four stride-2 conv4x4 stages 3->df->2df->4df->8df; the first has no BN (keys 0,2,3,5,6,8,9)."""
        d = df_dims
        chain = FusedChain().add_stage(0, HipConv2d(3, d, 4, 2, 1, False), None, L.ACT_LRELU)
        for idx, (ci, co) in zip((2, 5, 8), ((d, 2 * d), (2 * d, 4 * d), (4 * d, 8 * d))):
            chain.add_stage(idx, HipConv2d(ci, co, 4, 2, 1, False), idx + 1, L.ACT_LRELU)
        return chain

    @staticmethod
    def ResBlock(channel_num: int) -> ResBlock:
        return ResBlock(channel_num)
