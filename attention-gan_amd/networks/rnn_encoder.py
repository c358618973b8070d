"""Caption encoder with the reference's interface (networks/rnn_encoder.py:12-96).

Frozen during GAN training (train.py:89) and < 0.1 % of the step's FLOPs, so it is deliberately NOT a hand-written kernel:
embedding -> dropout -> packed bidirectional LSTM run on stock PyTorch-ROCm (MIOpen) on the same device and stream as the
HIP path.  Constructor arguments, `forward(captions, caption_lengths) -> (word_embs [B,nhidden,T], sent_embs [B,nhidden])`,
parameter names (`embedding.weight`, `rnn.weight_ih_l0`, ...) and the uniform(-0.1, 0.1) embedding init follow the reference.
"""
from __future__ import annotations

from typing import Tuple

import torch
from torch import Tensor, nn
from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence


class RNNEncoder(nn.Module):
    def __init__(self, vocabsize: int, embdim=300, dropprob=0.5, nhidden=128, nlayers=1, bidirectional=True):
        super().__init__()
        self.vocabsize, self.embdim, self.dropprob, self.nlayers = vocabsize, embdim, dropprob, nlayers
        self.bidirectional = bidirectional
        self.ndirections = 2 if bidirectional else 1
        self.nhidden = nhidden // self.ndirections
        self.embedding = nn.Embedding(vocabsize, embdim)
        self.dropout = nn.Dropout(dropprob)
        self.rnn = nn.LSTM(input_size=embdim, hidden_size=self.nhidden, num_layers=nlayers, batch_first=True,
                           dropout=dropprob if nlayers > 1 else 0.0, bidirectional=bidirectional)
        self.embedding.weight.data.uniform_(-0.1, 0.1)

    def init_hidden_cell_states(self, batch_size: int) -> Tuple[Tensor, Tensor]:
        w = next(self.parameters())
        shape = (self.nlayers * self.ndirections, batch_size, self.nhidden)
        return w.new_zeros(shape), w.new_zeros(shape)

    def freeze_all_weights(self):
        for p in self.parameters():
            p.requires_grad = False

    def forward(self, captions: Tensor, caption_lengths) -> Tuple[Tensor, Tensor]:
        lens = [int(v) for v in (caption_lengths.tolist() if hasattr(caption_lengths, "tolist") else caption_lengths)]
        x = self.dropout(self.embedding(captions))
        packed = pack_padded_sequence(x, lengths=lens, batch_first=True, enforce_sorted=False)
        out, (hidden, _cell) = self.rnn(packed, self.init_hidden_cell_states(len(lens)))
        out = pad_packed_sequence(out, batch_first=True)[0]                    # [B, T_max, nhidden]
        word_embs = out.transpose(1, 2)                                        # [B, nhidden, T_max]
        sent_embs = hidden.transpose(0, 1).contiguous().view(-1, self.ndirections * self.nhidden)
        return word_embs, sent_embs
