"""KL(N(mu, sigma) || N(0, I)) with the reference's mean reduction (losses/KL_loss.py:5-9)."""
from torch import Tensor

from ..backend import functional as HF


def KL_loss(mu: Tensor, logvar: Tensor) -> Tensor:
    return HF.kl_loss(mu, logvar)
