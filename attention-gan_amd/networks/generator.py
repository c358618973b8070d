"""Three-stage AttnGAN generator with the reference's interface (networks/generator.py:12-66)."""
from __future__ import annotations

from typing import List, Optional, Tuple

from torch import Tensor, nn

from .generator_submodules import GenInitialStage, GenMakeImage, GenNextStage, VarAutoEncoder


class Generator(nn.Module):
    def __init__(self, gf_dim: int, emb_dim: int, z_dim: int, cond_dim: int):
        super().__init__()
        self.gf_dim, self.emb_dim, self.z_dim, self.cond_dim = gf_dim, emb_dim, z_dim, cond_dim
        # creation order = the reference's, so the same torch seed yields the same initial weights
        self.vae = VarAutoEncoder(emb_dim=emb_dim, cond_dim=cond_dim)
        self.gen1 = GenInitialStage(gf_dim=gf_dim * 16, z_dim=z_dim, cond_dim=cond_dim)
        self.img_out1 = GenMakeImage(gf_dim=gf_dim)
        self.gen2 = GenNextStage(gf_dim=gf_dim, emb_dim=emb_dim, num_residual_blocks=2)
        self.img_out2 = GenMakeImage(gf_dim=gf_dim)
        self.gen3 = GenNextStage(gf_dim=gf_dim, emb_dim=emb_dim, num_residual_blocks=2)
        self.img_out3 = GenMakeImage(gf_dim=gf_dim)

    def forward(self, noise: Tensor, sent_emb: Tensor, word_embs: Tensor, mask: Tensor, eps: Optional[Tensor] = None):
        """(noise [B,z], sent_emb [B,emb], word_embs [B,emb,T], mask [B,T]) ->
        (fake_imgs [64,128,256], attn_maps [64x64, 128x128], mu, logvar) -- the reference returns 2 attention maps
        (generator.py:59,64) despite its docstring."""
        fake_imgs: List[Tensor] = []
        attn_maps: List[Tensor] = []
        condition, mu, logvar = self.vae(sent_emb, eps)
        images = self.gen1(noise, condition)
        fake_imgs.append(self.img_out1(images))
        for stage, head in ((self.gen2, self.img_out2), (self.gen3, self.img_out3)):
            images, attn = stage(images, word_embs, mask)
            fake_imgs.append(head(images))
            attn_maps.append(attn)
        return (fake_imgs, attn_maps, mu, logvar)


class Generator1(nn.Module):
    """Stage-1 only (BASELINE.json configs[1]: 64x64 G + D): the CA-net, the initial stage and its image head of the generator
    above (generator.py:26-30 of the reference), nothing of stages 2 / 3 -- same submodule names, so a full Generator's
    `vae.*`, `gen1.*`, `img_out1.*` entries load into it.  forward() keeps the Generator's signature and return shape
    (one image, no attention maps), so trainers.GanTrainStep drives it with a single discriminator."""

    def __init__(self, gf_dim: int, emb_dim: int, z_dim: int, cond_dim: int):
        super().__init__()
        self.gf_dim, self.emb_dim, self.z_dim, self.cond_dim = gf_dim, emb_dim, z_dim, cond_dim
        self.vae = VarAutoEncoder(emb_dim=emb_dim, cond_dim=cond_dim)
        self.gen1 = GenInitialStage(gf_dim=gf_dim * 16, z_dim=z_dim, cond_dim=cond_dim)
        self.img_out1 = GenMakeImage(gf_dim=gf_dim)

    def forward(self, noise: Tensor, sent_emb: Tensor, word_embs: Optional[Tensor] = None, mask: Optional[Tensor] = None,
                eps: Optional[Tensor] = None):
        condition, mu, logvar = self.vae(sent_emb, eps)
        return ([self.img_out1(self.gen1(noise, condition))], [], mu, logvar)
