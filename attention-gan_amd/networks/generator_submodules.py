"""Generator stages with the reference's constructors, forward signatures and parameter names
(networks/generator_submodules.py), executed on the HIP kernels."""
from __future__ import annotations

from typing import Optional, Tuple

import torch
from torch import Tensor, nn

from ..backend import functional as HF
from ..backend import lib as L
from ..utilities.layers import GLU, HipBatchNorm1d, HipConv2d, HipLinear, Layers
from .attention import AttentionModule


class _FcBnGlu(nn.Module):
    """Linear(no bias) -> BatchNorm1d -> GLU with the reference Sequential's keys '0.weight', '1.*' (:36-40)."""

    def __init__(self, fin: int, fout: int):
        super().__init__()
        self.add_module("0", HipLinear(fin, fout, bias=False))
        self.add_module("1", HipBatchNorm1d(fout))

    def forward(self, x: Tensor) -> Tensor:
        return getattr(self, "1").fused(getattr(self, "0")(x), L.ACT_GLU)


class GenInitialStage(nn.Module):
    """(noise, condition) -> [B, gf_dim/16, 64, 64]: fc+BN1d+GLU, view [B,gf_dim,4,4], four upBlocks (:13-66)."""

    def __init__(self, gf_dim: int, z_dim: int, cond_dim: int):
        super().__init__()
        self.gf_dim, self.z_dim, self.cond_dim = gf_dim, z_dim, cond_dim
        self.define_module()

    def define_module(self):
        ng = self.gf_dim
        self.fc = _FcBnGlu(self.z_dim + self.cond_dim, ng * 4 * 4 * 2)
        self.upsample1 = Layers.upBlock(ng, ng // 2)
        self.upsample2 = Layers.upBlock(ng // 2, ng // 4)
        self.upsample3 = Layers.upBlock(ng // 4, ng // 8)
        self.upsample4 = Layers.upBlock(ng // 8, ng // 16)

    def forward(self, noise: Tensor, condition: Tensor) -> Tensor:
        x = self.fc(torch.cat((noise, condition), 1)).view(-1, self.gf_dim, 4, 4)
        for up in (self.upsample1, self.upsample2, self.upsample3, self.upsample4):
            x = up(x)
        return x


class GenNextStage(nn.Module):
    """attention -> concat -> residual blocks -> upBlock (:69-120).  Returns (images [B,gf,2h,2w], attn [B,T,h,w])."""

    def __init__(self, gf_dim: int, emb_dim: int, num_residual_blocks: int):
        super().__init__()
        self.gf_dim, self.emb_dim, self.num_residual_blocks = gf_dim, emb_dim, num_residual_blocks
        self.define_module()

    def define_module(self):
        self.attention = AttentionModule(nc_in=self.gf_dim, emb_dim=self.emb_dim)
        self.residual = self._make_layer(Layers.ResBlock, self.gf_dim * 2)
        self.upsample = Layers.upBlock(self.gf_dim * 2, self.gf_dim)

    def _make_layer(self, block, channel_num: int):
        return nn.Sequential(*[block(channel_num) for _ in range(self.num_residual_blocks)])

    def forward(self, images: Tensor, word_embs: Tensor, mask: Tensor) -> Tuple[Tensor, Tensor]:
        self.attention.apply_mask(mask=mask)
        context, attn = self.attention(images, word_embs)
        x = self.residual(torch.cat((images, context), 1))
        return self.upsample(x), attn


class _ConvTanh(nn.Module):
    """conv3x3(gf,3) + tanh with the reference Sequential's key '0.weight' (:134-138)."""

    def __init__(self, gf_dim: int):
        super().__init__()
        self.add_module("0", HipConv2d(gf_dim, 3, 3, 1, 1, False))

    def forward(self, x: Tensor) -> Tensor:
        return HF.activation(getattr(self, "0")(x), L.ACT_TANH)


class GenMakeImage(nn.Module):
    def __init__(self, gf_dim: int):
        super().__init__()
        self.gf_dim = gf_dim
        self.img = _ConvTanh(gf_dim)

    def forward(self, images: Tensor) -> Tensor:
        return self.img(images)


class VarAutoEncoder(nn.Module):
    """Conditioning augmentation (:145-170): Linear(emb, 4*cond) -> GLU -> (mu, logvar); c = eps*exp(logvar/2)+mu.

    The reference draws eps with torch.cuda.FloatTensor(...).normal_() (:163); here it is torch.randn on the
    input's device.  `forward(text_embedding, eps=None)` accepts an explicit eps so tests can replay a recorded draw.
    """

    def __init__(self, emb_dim: int, cond_dim=100):
        super().__init__()
        self.emb_dim, self.cond_dim = emb_dim, cond_dim
        self.fc = HipLinear(emb_dim, cond_dim * 4, bias=True)
        self.relu = GLU()

    def encode(self, text_embedding: Tensor):
        x = self.relu(self.fc(text_embedding))
        return x[:, :self.cond_dim], x[:, self.cond_dim:]

    def reparametrize(self, mu: Tensor, logvar: Tensor, eps: Optional[Tensor] = None) -> Tensor:
        if eps is None:
            eps = torch.randn(mu.shape, dtype=mu.dtype, device=mu.device)
        return HF.reparametrize(mu, logvar, eps)

    def forward(self, text_embedding: Tensor, eps: Optional[Tensor] = None):
        mu, logvar = self.encode(text_embedding)
        return self.reparametrize(mu, logvar, eps), mu, logvar
