"""Image-encoder plug-in boundary (reference networks/cnn_encoder.py:65-102).

The reference's CNNEncoder is a frozen torchvision Inception-v3 trunk with downloaded weights (cnn_encoder.py:26-27):
third-party, not constructible offline, and outside the hand-written-kernel scope.  What the
training step needs from it is only the output contract

    forward(images [B,3,H,W]) -> (region features [B, out_dim, 17, 17], global code [B, out_dim])

with gradients flowing back to the images.  `StandInImageEncoder` honours that contract with a deliberately small frozen
map (adaptive 17x17 average pool -> 1x1 projection; region mean -> linear) on stock PyTorch-ROCm ops so that the DAMSM
branch of the generator update is exercised end to end.  Its FLOPs are NOT the Inception trunk's; bench.py says so.
Any module with the same contract (e.g. a real Inception-v3 with locally supplied weights) can be passed instead.
"""
from __future__ import annotations

from typing import Tuple

import torch
import torch.nn.functional as F
from torch import Tensor, nn


class StandInImageEncoder(nn.Module):
    def __init__(self, out_dim: int = 256, seed: int = 1000):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.out_dim = out_dim
        self.emb_features = nn.Parameter(0.5 * torch.randn(out_dim, 3, generator=g))          # 1x1 projection
        self.emb_cnn_code = nn.Parameter(0.5 * torch.randn(out_dim, out_dim, generator=g))    # linear on the region mean

    def freeze_all_weights(self):
        for p in self.parameters():
            p.requires_grad = False

    def forward(self, x: Tensor) -> Tuple[Tensor, Tensor]:
        r = F.adaptive_avg_pool2d(x, 17)
        regions = torch.einsum("ec,bchw->behw", self.emb_features, r)
        code = regions.mean(dim=(2, 3)) @ self.emb_cnn_code.t()
        return regions, code
