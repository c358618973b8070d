"""Discriminator losses with the reference's interface (losses/disc_loss.py)."""
from __future__ import annotations

import torch
from torch import Tensor
from torch.nn import Module

from ..backend import functional as HF


class DiscLoss:
    def make_labels_for_real_imgs(self, num_labels: int, label_smooth=0.8, device="cuda") -> Tensor:
        return torch.empty(num_labels, device=device).uniform_(label_smooth, 1.0)

    def make_labels_for_fake_imgs(self, num_labels: int, device="cuda") -> Tensor:
        return torch.zeros(num_labels, device=device)

    def get_loss(self, discriminator: Module, fake_images: Tensor, real_images: Tensor) -> Tensor:
        raise NotImplementedError


class NonSaturatingDiscLoss(DiscLoss):
    """-mean(log(D(x)+1e-8) + log(1-D(G(z))+1e-8)); the real batch goes through D first (disc_loss.py:55-61).

    With a discriminator of this package the two passes are ONE pass over the concatenated [real; fake] batch inside
    `functional.bn_groups(2)`: every BatchNorm still normalises each half with its own statistics and updates its running
    statistics twice, real first -- the reference's results -- while every convolution runs once on twice the pixels (half
    the launches; the deep 4x4..16x16 layers, M = 384 pixels per batch, fill their tiles twice as well).  `batch_pairs = False`
    restores the two separate passes."""

    batch_pairs = True

    def get_loss(self, discriminator: Module, fake_images: Tensor, real_images: Tensor) -> Tensor:
        if (self.batch_pairs and getattr(discriminator, "supports_batch_groups", False) and discriminator.training
                and fake_images.is_cuda and fake_images.shape == real_images.shape):
            with HF.bn_groups(2):
                score = discriminator(torch.cat([real_images, fake_images], dim=0))
            return HF.ns_disc_loss_paired(score)
        dx_score = discriminator(real_images)
        dg_score = discriminator(fake_images)
        return HF.ns_disc_loss(dx_score, dg_score)


class StandardDiscLoss(DiscLoss):
    """BCE variant (disc_loss.py:26-47); not used by train.py:74-75.  The discriminator passes run on the HIP kernels, the
    cross-entropy over [B] probabilities is a stock ATen elementwise op."""

    def get_loss(self, discriminator: Module, fake_images: Tensor, real_images: Tensor) -> Tensor:
        """(BCE(D(G(z)), 0) + BCE(D(x), U(0.8, 1))) / 2 -- fake batch first, smoothed real labels (disc_loss.py:31-47)."""
        bce = torch.nn.functional.binary_cross_entropy
        p_fake = discriminator(fake_images)
        loss_fake = bce(p_fake, self.make_labels_for_fake_imgs(p_fake.shape[0], device=p_fake.device))
        p_real = discriminator(real_images)
        loss_real = bce(p_real, self.make_labels_for_real_imgs(p_real.shape[0], device=p_real.device))
        return (loss_fake + loss_real) / 2
