"""DAMSM sentence-level loss with the reference's interface (losses/sentence_loss.py:7-50)."""
from __future__ import annotations

import torch

from ..backend import functional as HF


class SentenceLoss:
    def __init__(self, device: torch.device, gamma3=10.0, slambda=5.0):
        self.device = device
        self.gamma3, self.slambda = gamma3, slambda

    def get_loss(self, cnn_code, rnn_code, labels, class_ids, eps=1e-8):
        """cnn_code, rnn_code [B,nef]; labels [B] int64 CE targets (sentence_loss.py:46-47; arange in train.py:104);
        class_ids array or None -> scalar loss."""
        if cnn_code.dim() != 2 or cnn_code.shape[0] < 2:
            # the reference's squeeze() (sentence_loss.py:41) breaks at B == 1 as well
            raise ValueError("SentenceLoss needs [B, nef] codes with B >= 2")
        return HF.sentence_loss(cnn_code, rnn_code, class_ids, self.gamma3, self.slambda, eps, labels)
