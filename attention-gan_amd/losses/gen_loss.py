"""Generator adversarial losses with the reference's interface (losses/gen_loss.py)."""
from __future__ import annotations

import torch
from torch import Tensor
from torch.nn import Module

from ..backend import functional as HF


class GenLoss:
    def make_labels_for_real_imgs(self, num_labels: int, device="cuda") -> Tensor:
        return torch.ones(num_labels, device=device)

    def get_loss(self, discriminator: Module, fake_images: Tensor) -> Tensor:
        raise NotImplementedError


class NonSaturatingGenLoss(GenLoss):
    """-mean(log(D(G(z)) + 1e-8))  (gen_loss.py:42-46)."""

    def get_loss(self, discriminator: Module, fake_images: Tensor) -> Tensor:
        return HF.ns_gen_loss(discriminator(fake_images))


class StandardGenLoss(GenLoss):
    """BCE(D(G(z)), 1)  (gen_loss.py:21-35).  Not used by train.py:74; the discriminator runs on the HIP kernels, the
    cross-entropy over its [B] probabilities is a stock ATen elementwise op."""

    def get_loss(self, discriminator: Module, fake_images: Tensor) -> Tensor:
        p = discriminator(fake_images)
        return torch.nn.functional.binary_cross_entropy(p, self.make_labels_for_real_imgs(p.shape[0], device=p.device))
