"""Image-encoder plug-in boundary (reference networks/cnn_encoder.py:65-102).

What the training step needs from the encoder is only its output contract

    forward(images [B,3,H,W]) -> (region features [B, out_dim, 17, 17], global code [B, out_dim])

with gradients flowing back to the images.  The reference's CNNEncoder is a frozen torchvision Inception-v3 trunk with
weights downloaded at construction (cnn_encoder.py:26-27): third-party, not constructible offline, and outside the
hand-written-kernel scope.  Two implementations of the contract live here, both on stock
PyTorch-ROCm ops (MIOpen convolutions) on the same device/stream as the HIP path:

* `CNNEncoder(out_dim)` -- an Inception-v3-shaped trunk written from the published architecture (Szegedy et al. 2016),
  with torchvision's parameter names (`Conv2d_1a_3x3.conv.weight`, `Mixed_5b.branch1x1.bn.running_mean`, ...) so a locally
  supplied `inception_v3_google-*.pth` can be loaded with `load_trunk_state_dict()`; nothing is ever fetched.  Heads
  `emb_features` (conv1x1 768->out_dim, no bias) and `emb_cnn_code` (Linear 2048->out_dim) with uniform(-0.1, 0.1) init
  follow the reference (:56-63).  Parity of the trunk against torchvision is UNPINNED here (torchvision is not installed).
* `StandInImageEncoder(out_dim)` -- a deliberately tiny frozen map honouring only the contract (adaptive 17x17 average pool
  -> 1x1 projection; region mean -> linear).  bench.py's default: its FLOPs are negligible, i.e. the timed step is the hot
  path of SURVEY.md §8d (which prices the trunk separately); `bench.py --image-encoder inception` times the full trunk too.
"""
from __future__ import annotations

from typing import Dict, Tuple

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

    @staticmethod
    def _pool_matrix(n_in: int, n_out: int, device) -> Tensor:
        """adaptive_avg_pool windows (start = floor(i*n_in/n_out), end = ceil((i+1)*n_in/n_out)) as an [n_out, n_in] matrix"""
        P = torch.zeros(n_out, n_in)
        for i in range(n_out):
            lo, hi = (i * n_in) // n_out, -((-(i + 1) * n_in) // n_out)
            P[i, lo:hi] = 1.0 / (hi - lo)
        return P.to(device)

    def forward(self, x: Tensor) -> Tuple[Tensor, Tensor]:
        # adaptive 17x17 average pooling as two small matrix products: same values as F.adaptive_avg_pool2d, but the backward
        # is a pair of GEMMs instead of torch's float-atomic scatter kernel (slow at 256x256, and not run-to-run reproducible)
        H, W = x.shape[-2:]
        key = (H, W, x.device)
        if getattr(self, "_pool_key", None) != key:
            self._pool_key, self._pool = key, (self._pool_matrix(H, 17, x.device), self._pool_matrix(W, 17, x.device))
        Ph, Pw = self._pool
        r = torch.matmul(torch.matmul(Ph, x), Pw.t())
        regions = torch.einsum("ec,bchw->behw", self.emb_features, r)
        code = regions.mean(dim=(2, 3)) @ self.emb_cnn_code.t()
        return regions, code


# ------------------------------------------------------------------------------------------------------------------
# Inception-v3-shaped trunk
# ------------------------------------------------------------------------------------------------------------------
class _ConvBNReLU(nn.Module):
    """conv (no bias) -> BatchNorm(eps=1e-3) -> ReLU; children named `conv`, `bn` as in torchvision's BasicConv2d."""

    def __init__(self, cin: int, cout: int, k, stride=1, pad=0):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, kernel_size=k, stride=stride, padding=pad, bias=False)
        self.bn = nn.BatchNorm2d(cout, eps=0.001)
        nn.init.kaiming_normal_(self.conv.weight, mode="fan_in", nonlinearity="relu")   # variance-preserving random init

    def forward(self, x: Tensor) -> Tensor:
        return F.relu(self.bn(self.conv(x)), inplace=True)


def _chain(parent: nn.Module, prefix: str, cin: int, spec) -> None:
    """register `prefix_1`, `prefix_2`, ... (or the given suffixes) as a serial chain of conv-bn-relu layers"""
    for name, cout, k, stride, pad in spec:
        parent.add_module(prefix + name, _ConvBNReLU(cin, cout, k, stride, pad))
        cin = cout


def _run(parent: nn.Module, prefix: str, names, x: Tensor) -> Tensor:
    for n in names:
        x = getattr(parent, prefix + n)(x)
    return x


class _MixedA(nn.Module):        # 35x35 block: 1x1 | 1x1-5x5 | 1x1-3x3-3x3 | avgpool-1x1
    def __init__(self, cin: int, pool: int):
        super().__init__()
        self.branch1x1 = _ConvBNReLU(cin, 64, 1)
        _chain(self, "branch5x5", cin, [("_1", 48, 1, 1, 0), ("_2", 64, 5, 1, 2)])
        _chain(self, "branch3x3dbl", cin, [("_1", 64, 1, 1, 0), ("_2", 96, 3, 1, 1), ("_3", 96, 3, 1, 1)])
        self.branch_pool = _ConvBNReLU(cin, pool, 1)

    def forward(self, x):
        return torch.cat([self.branch1x1(x), _run(self, "branch5x5", ("_1", "_2"), x), _run(self, "branch3x3dbl", ("_1", "_2", "_3"), x),
                          self.branch_pool(F.avg_pool2d(x, 3, 1, 1))], 1)


class _MixedB(nn.Module):        # 35 -> 17 reduction
    def __init__(self, cin: int):
        super().__init__()
        self.branch3x3 = _ConvBNReLU(cin, 384, 3, 2)
        _chain(self, "branch3x3dbl", cin, [("_1", 64, 1, 1, 0), ("_2", 96, 3, 1, 1), ("_3", 96, 3, 2, 0)])

    def forward(self, x):
        return torch.cat([self.branch3x3(x), _run(self, "branch3x3dbl", ("_1", "_2", "_3"), x), F.max_pool2d(x, 3, 2)], 1)


class _MixedC(nn.Module):        # 17x17 block with factorised 7x7
    def __init__(self, cin: int, c7: int):
        super().__init__()
        self.branch1x1 = _ConvBNReLU(cin, 192, 1)
        _chain(self, "branch7x7", cin, [("_1", c7, 1, 1, 0), ("_2", c7, (1, 7), 1, (0, 3)), ("_3", 192, (7, 1), 1, (3, 0))])
        _chain(self, "branch7x7dbl", cin, [("_1", c7, 1, 1, 0), ("_2", c7, (7, 1), 1, (3, 0)), ("_3", c7, (1, 7), 1, (0, 3)),
                                           ("_4", c7, (7, 1), 1, (3, 0)), ("_5", 192, (1, 7), 1, (0, 3))])
        self.branch_pool = _ConvBNReLU(cin, 192, 1)

    def forward(self, x):
        return torch.cat([self.branch1x1(x), _run(self, "branch7x7", ("_1", "_2", "_3"), x),
                          _run(self, "branch7x7dbl", ("_1", "_2", "_3", "_4", "_5"), x), self.branch_pool(F.avg_pool2d(x, 3, 1, 1))], 1)


class _MixedD(nn.Module):        # 17 -> 8 reduction
    def __init__(self, cin: int):
        super().__init__()
        _chain(self, "branch3x3", cin, [("_1", 192, 1, 1, 0), ("_2", 320, 3, 2, 0)])
        _chain(self, "branch7x7x3", cin, [("_1", 192, 1, 1, 0), ("_2", 192, (1, 7), 1, (0, 3)), ("_3", 192, (7, 1), 1, (3, 0)), ("_4", 192, 3, 2, 0)])

    def forward(self, x):
        return torch.cat([_run(self, "branch3x3", ("_1", "_2"), x), _run(self, "branch7x7x3", ("_1", "_2", "_3", "_4"), x), F.max_pool2d(x, 3, 2)], 1)


class _MixedE(nn.Module):        # 8x8 block with split 3x3
    def __init__(self, cin: int):
        super().__init__()
        self.branch1x1 = _ConvBNReLU(cin, 320, 1)
        self.branch3x3_1 = _ConvBNReLU(cin, 384, 1)
        self.branch3x3_2a = _ConvBNReLU(384, 384, (1, 3), 1, (0, 1))
        self.branch3x3_2b = _ConvBNReLU(384, 384, (3, 1), 1, (1, 0))
        _chain(self, "branch3x3dbl", cin, [("_1", 448, 1, 1, 0), ("_2", 384, 3, 1, 1)])
        self.branch3x3dbl_3a = _ConvBNReLU(384, 384, (1, 3), 1, (0, 1))
        self.branch3x3dbl_3b = _ConvBNReLU(384, 384, (3, 1), 1, (1, 0))
        self.branch_pool = _ConvBNReLU(cin, 192, 1)

    def forward(self, x):
        a = self.branch3x3_1(x)
        b = _run(self, "branch3x3dbl", ("_1", "_2"), x)
        return torch.cat([self.branch1x1(x), self.branch3x3_2a(a), self.branch3x3_2b(a), self.branch3x3dbl_3a(b), self.branch3x3dbl_3b(b),
                          self.branch_pool(F.avg_pool2d(x, 3, 1, 1))], 1)


class CNNEncoder(nn.Module):
    """Drop-in for the reference CNNEncoder: same attribute names (`Conv2d_1a_3x3` ... `Mixed_7c`, `emb_features`,
    `emb_cnn_code`), `freeze_all_weights()`, and `forward(x) -> (features [B,out_dim,17,17], cnn_code [B,out_dim])`."""

    def __init__(self, out_dim: int = 256):
        super().__init__()
        self.out_dim = out_dim
        self.Conv2d_1a_3x3 = _ConvBNReLU(3, 32, 3, 2)
        self.Conv2d_2a_3x3 = _ConvBNReLU(32, 32, 3)
        self.Conv2d_2b_3x3 = _ConvBNReLU(32, 64, 3, 1, 1)
        self.Conv2d_3b_1x1 = _ConvBNReLU(64, 80, 1)
        self.Conv2d_4a_3x3 = _ConvBNReLU(80, 192, 3)
        self.Mixed_5b, self.Mixed_5c, self.Mixed_5d = _MixedA(192, 32), _MixedA(256, 64), _MixedA(288, 64)
        self.Mixed_6a = _MixedB(288)
        self.Mixed_6b, self.Mixed_6c, self.Mixed_6d, self.Mixed_6e = _MixedC(768, 128), _MixedC(768, 160), _MixedC(768, 160), _MixedC(768, 192)
        self.Mixed_7a = _MixedD(768)
        self.Mixed_7b, self.Mixed_7c = _MixedE(1280), _MixedE(2048)
        for p in self.parameters():                      # the trunk is frozen (reference :28-30); only the two heads train
            p.requires_grad = False
        self.emb_features = nn.Conv2d(768, out_dim, kernel_size=1, bias=False)
        self.emb_cnn_code = nn.Linear(2048, out_dim)
        self.emb_features.weight.data.uniform_(-0.1, 0.1)
        self.emb_cnn_code.weight.data.uniform_(-0.1, 0.1)

    def freeze_all_weights(self):
        for p in self.parameters():
            p.requires_grad = False

    def load_trunk_state_dict(self, state: Dict[str, Tensor]) -> None:
        """Load a torchvision inception_v3 state_dict supplied by the user (weights_only file); AuxLogits / fc are ignored."""
        own = self.state_dict()
        self.load_state_dict({k: v for k, v in state.items() if k in own and not k.startswith("emb_")}, strict=False)

    def forward(self, x: Tensor) -> Tuple[Tensor, Tensor]:
        x = F.interpolate(x, size=(299, 299), mode="bilinear", align_corners=False)      # :75
        x = self.Conv2d_2b_3x3(self.Conv2d_2a_3x3(self.Conv2d_1a_3x3(x)))
        x = F.max_pool2d(x, 3, 2)
        x = self.Conv2d_4a_3x3(self.Conv2d_3b_1x1(x))
        x = F.max_pool2d(x, 3, 2)
        x = self.Mixed_5d(self.Mixed_5c(self.Mixed_5b(x)))
        x = self.Mixed_6e(self.Mixed_6d(self.Mixed_6c(self.Mixed_6b(self.Mixed_6a(x)))))
        features = self.emb_features(x)                                                   # [B, out_dim, 17, 17]
        x = self.Mixed_7c(self.Mixed_7b(self.Mixed_7a(x)))
        x = F.avg_pool2d(x, 8).flatten(1)                                                 # [B, 2048]
        return features, self.emb_cnn_code(x)
