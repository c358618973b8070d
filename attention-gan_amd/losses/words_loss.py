"""DAMSM word-level loss with the reference's interface (losses/words_loss.py), one fused HIP kernel pair."""
from __future__ import annotations

import torch

from ..backend import functional as HF


class WordsLoss:
    """Same constructor and `get_loss` contract as the reference (words_loss.py:13-18, 29-102).

    The reference loops `func_attention` over the B captions in Python (~25 tiny kernels each); here all
    B x B (image, caption) pairs are evaluated by one kernel and back-propagated by another.
    """

    def __init__(self, device: torch.device, gamma1=4.0, gamma2=5.0, gamma3=10.0, wlambda=5.0):
        self.device = device
        self.gamma1, self.gamma2, self.gamma3, self.wlambda = gamma1, gamma2, gamma3, wlambda

    def cosine_similarity(self, x1, x2, dim=1, eps=1e-8):
        w12 = torch.sum(x1 * x2, dim)
        return (w12 / (torch.norm(x1, 2, dim) * torch.norm(x2, 2, dim)).clamp(min=eps)).squeeze()

    def get_loss(self, img_features, words_emb, labels, cap_lens, class_ids):
        """img_features [B,nef,17,17], words_emb [B,nef,T], labels [B] int64 (the CE targets of words_loss.py:98-99; train.py:104
        builds arange, which `_make_match_labels` tags so the kernel's built-in default is used without a copy; any other vector
        is honoured as given), cap_lens [B], class_ids [B] array or None -> (loss, [attention map [1,L_i,17,17] per caption])."""
        b = img_features.shape[0]
        ih, iw = img_features.shape[2], img_features.shape[3]
        if isinstance(cap_lens, torch.Tensor) and cap_lens.is_cuda:
            # lengths already on the device (HIP-graph capture / no host sync): the per-caption list cannot be cut without
            # reading the lengths back, so the maps come back as one zero-padded tensor [B, T, ih, iw]
            loss, maps, _ = HF.words_loss(img_features, words_emb, cap_lens.to(torch.int64), class_ids, self.gamma1, self.gamma2,
                                          self.gamma3, self.wlambda, labels)
            return (loss, maps.view(b, -1, ih, iw))
        lens = [int(v) for v in (cap_lens.tolist() if hasattr(cap_lens, "tolist") else cap_lens)]
        loss, maps, _ = HF.words_loss(img_features, words_emb, lens, class_ids, self.gamma1, self.gamma2, self.gamma3,
                                      self.wlambda, labels)
        att_maps = [maps[i:i + 1, :lens[i]].reshape(1, lens[i], ih, iw) for i in range(b)]
        return (loss, att_maps)
