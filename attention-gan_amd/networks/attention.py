"""Word-context attention with the reference's interface (networks/attention.py), one HIP pass per direction."""
from __future__ import annotations

from typing import Tuple

from torch import Tensor, nn

from ..backend import functional as HF
from ..utilities.layers import Layers


class AttentionModule(nn.Module):
    """Drop-in for reference AttentionModule (attention.py:15-79): same ctor, `apply_mask`, `forward` and the
    single parameter `conv1.weight` [nc_in, emb_dim, 1, 1]."""

    def __init__(self, nc_in: int, emb_dim: int):
        super().__init__()
        self.nc_in = nc_in
        self.conv1 = Layers.conv1x1(in_planes=emb_dim, out_planes=nc_in)
        self.mask = None

    def apply_mask(self, mask: Tensor) -> None:
        self.mask = mask

    def forward(self, images: Tensor, words: Tensor, scaled=True) -> Tuple[Tensor, Tensor]:
        """images [B,nc_in,h,w], words [B,emb_dim,T], self.mask [B,T] (0 = ignore) -> (context [B,nc_in,h,w], attn [B,T,h,w])."""
        (batch, nc_in, h, w) = images.shape
        (batch_w, emb_dim, seq_len) = words.shape
        (batch_m, seq_len_m) = self.mask.shape       # raises like the reference if no mask was applied
        return HF.attention(images, words, self.conv1.weight, self.mask, scaled)


def func_attention(query: Tensor, context: Tensor, gamma1=4.0, scaled=True):
    """Parameter-free DAMSM attention (reference attention.py:82-120): query [B,D,L], context [B,D,ih,iw] ->
    (weightedContext [B,D,L], attn [B,L,ih,iw])."""
    return HF.func_attention(query, context, gamma1, scaled)
