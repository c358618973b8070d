"""Unconditional sigmoid discriminators with the reference's interface (networks/discriminators.py:10-70)."""
from __future__ import annotations

from torch import Tensor, nn

from ..backend import functional as HF
from ..backend import lib as L
from ..utilities.layers import HipConv2d, Layers


class _LogitHead(nn.Module):
    """Conv2d(8df, 1, k=4, s=4) + Sigmoid on the 4x4 code: a dot product per image (keys '0.weight', '0.bias')."""

    def __init__(self, cin: int):
        super().__init__()
        self.add_module("0", nn.Conv2d(cin, 1, kernel_size=4, stride=4))   # parameter holder only
        self._packed = {}

    def forward(self, x: Tensor) -> Tensor:
        conv = getattr(self, "0")
        b, c, h, w = x.shape
        if (h, w) != (4, 4):
            raise ValueError(f"discriminator head expects a 4x4 code, got {h}x{w}")
        logit = HF.linear(x.reshape(b, c * 16), conv.weight.view(1, c * 16), conv.bias, self._packed,
                          HF.grad_dst(conv.weight), HF.grad_dst(conv.bias))
        return HF.activation(logit, L.ACT_SIGMOID)


class _Disc(nn.Module):
    # every BatchNorm inside is a HipBatchNorm2d and nothing else couples samples of a batch, so a [real; fake] batch may go
    # through in one pass under functional.bn_groups(2) (losses/disc_loss.py)
    supports_batch_groups = True

    def _tail(self, x: Tensor) -> Tensor:
        return self.outlogits(x).view(-1)


class Disc64(_Disc):
    def __init__(self, df_dim: int):
        super().__init__()
        self.img_code_s16 = Layers.encode_image_by_16times(df_dim)
        self.outlogits = _LogitHead(df_dim * 8)

    def forward(self, X: Tensor) -> Tensor:
        return self._tail(self.img_code_s16(X))


class Disc128(_Disc):
    def __init__(self, df_dim: int):
        super().__init__()
        self.img_code_s16 = Layers.encode_image_by_16times(df_dim)
        self.img_code_s32 = Layers.downBlock(df_dim * 8, df_dim * 16)
        self.img_code_s32_1 = Layers.Block3x3_leakRelu(df_dim * 16, df_dim * 8)
        self.outlogits = _LogitHead(df_dim * 8)

    def forward(self, X: Tensor) -> Tensor:
        return self._tail(self.img_code_s32_1(self.img_code_s32(self.img_code_s16(X))))


class Disc256(_Disc):
    def __init__(self, df_dim: int):
        super().__init__()
        self.img_code_s16 = Layers.encode_image_by_16times(df_dim)
        self.img_code_s32 = Layers.downBlock(df_dim * 8, df_dim * 16)
        self.img_code_s64 = Layers.downBlock(df_dim * 16, df_dim * 32)
        self.img_code_s64_1 = Layers.Block3x3_leakRelu(df_dim * 32, df_dim * 16)
        self.img_code_s64_2 = Layers.Block3x3_leakRelu(df_dim * 16, df_dim * 8)
        self.outlogits = _LogitHead(df_dim * 8)

    def forward(self, X: Tensor) -> Tensor:
        x = self.img_code_s64(self.img_code_s32(self.img_code_s16(X)))
        return self._tail(self.img_code_s64_2(self.img_code_s64_1(x)))
