"""512x512 fourth-stage extension (BASELINE.json configs[4]).  NOT in the reference: built only from its pinned primitives by
extrapolating its own pattern -- one more `GenNextStage` + `GenMakeImage` on the generator (generator.py:33-35) and, for the
discriminator, `encode_image_by_16times` + three `downBlock`s + three `Block3x3_leakRelu`s back to 8*df channels
(discriminators.py:48-60 continued by one level).  There is no reference oracle for the composition;
tests check it against the CPU oracle's primitives composed the same way."""
from __future__ import annotations

from typing import List, Optional

from torch import Tensor, nn

from ..utilities.layers import Layers
from .discriminators import _Disc, _LogitHead
from .generator import Generator
from .generator_submodules import GenMakeImage, GenNextStage


class Generator4(Generator):
    """Generator with a 4th stage: returns 4 images (64, 128, 256, 512) and 3 attention maps."""

    def __init__(self, gf_dim: int, emb_dim: int, z_dim: int, cond_dim: int):
        super().__init__(gf_dim, emb_dim, z_dim, cond_dim)
        self.gen4 = GenNextStage(gf_dim=gf_dim, emb_dim=emb_dim, num_residual_blocks=2)
        self.img_out4 = GenMakeImage(gf_dim=gf_dim)

    def forward(self, noise: Tensor, sent_emb: Tensor, word_embs: Tensor, mask: Tensor, eps: Optional[Tensor] = None):
        fake_imgs: List[Tensor] = []
        attn_maps: List[Tensor] = []
        condition, mu, logvar = self.vae(sent_emb, eps)
        images = self.gen1(noise, condition)
        fake_imgs.append(self.img_out1(images))
        for stage, head in ((self.gen2, self.img_out2), (self.gen3, self.img_out3), (self.gen4, self.img_out4)):
            images, attn = stage(images, word_embs, mask)
            fake_imgs.append(head(images))
            attn_maps.append(attn)
        return (fake_imgs, attn_maps, mu, logvar)


class Disc512(_Disc):
    def __init__(self, df_dim: int):
        super().__init__()
        self.img_code_s16 = Layers.encode_image_by_16times(df_dim)
        self.img_code_s32 = Layers.downBlock(df_dim * 8, df_dim * 16)
        self.img_code_s64 = Layers.downBlock(df_dim * 16, df_dim * 32)
        self.img_code_s128 = Layers.downBlock(df_dim * 32, df_dim * 64)
        self.img_code_s128_1 = Layers.Block3x3_leakRelu(df_dim * 64, df_dim * 32)
        self.img_code_s128_2 = Layers.Block3x3_leakRelu(df_dim * 32, df_dim * 16)
        self.img_code_s128_3 = Layers.Block3x3_leakRelu(df_dim * 16, df_dim * 8)
        self.outlogits = _LogitHead(df_dim * 8)

    def forward(self, X: Tensor) -> Tensor:
        x = self.img_code_s128(self.img_code_s64(self.img_code_s32(self.img_code_s16(X))))
        return self._tail(self.img_code_s128_3(self.img_code_s128_2(self.img_code_s128_1(x))))
