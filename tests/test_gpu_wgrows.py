"""Row-resident weight gradient (csrc/conv_wgrows.hip, round 3) against torch's CPU fp32/fp64 convolution gradient.

One case per code path of the kernel and its plan, each in every arithmetic mode that routes there:
  * conv3x3: 128-channel tiles, image rows wider than the 64-column tile (dy halo blocks LOADED), rows of one tile (shared zero halo),
    the two-chunk layout of <= 64 output channels, tiles spanning several images (8x8 maps), channel counts that do not fill the
    last chunk / cout tile, a pixel split over many workgroups;
  * conv4x4 stride 2: the de-interleaved [even | odd] columns, halo on 64-wide outputs, 8x8 outputs (tiles of two images);
  * the upsample conv taken as conv3x3 on the upsampled image built in LDS;
  * fp32 operands on v_mfma_f32_32x32x2 (AGAN_PREC_F32, 3x3 layers), the two-plane splits, 16-bit and fp32 activation storage.
The operands are exactly representable in the mode's operand type (16-bit modes), so the fp32 result differs from the reference only by
accumulation order.  `AGAN_WG_ROWS_OFF=1` (checked at library load) would send every case to the older kernels: the last test makes sure the
plan really takes these geometries.
"""
import ctypes
import importlib

import pytest
import torch

from helpers import assert_close, probe

pytestmark = pytest.mark.gpu
DEV = "cuda"

HF = importlib.import_module("attention-gan_amd.backend.functional")
L = importlib.import_module("attention-gan_amd.backend.lib")
from oracle import attngan_oracle as O   # noqa: E402  (checker only: upsample2)

# mode -> (precision, storage, operand rounding, tolerance of the fp32 weight gradient relative to its largest entry)
MODES = {
    "f32": (L.PREC_F32, None, None, 2e-5),
    "bf16x6": (L.PREC_BF16X6, None, None, 5e-5),
    "f16x3": (L.PREC_F16X3, None, None, 5e-5),
    "bf16": (L.PREC_BF16, None, torch.bfloat16, 2e-4),
    "f16": (L.PREC_F16, None, torch.float16, 2e-4),
    "bf16+s16": (L.PREC_BF16, "bf16", torch.bfloat16, 2e-4),
    "f16+s16": (L.PREC_F16, "f16", torch.float16, 2e-4),
}

CASES = [
    # kind, B, Cin, H, W, Cout
    ("same", 2, 64, 8, 128, 128),      # rows of two 64-column tiles: halo blocks loaded from the neighbouring tile
    ("same", 2, 32, 16, 64, 96),       # one tile per row (zero halo shared between rows), cout tile not full
    ("same", 2, 72, 16, 32, 64),       # <= 64 output channels: two chunks per workgroup, odd number of chunks (72 = 2 x 32 + 8)
    ("same", 5, 64, 8, 8, 160),        # 8x8 maps: a tile spans two images, ragged last tile, two cout tiles
    ("same", 1, 40, 32, 256, 32),      # four column tiles per row
    ("down", 2, 64, 16, 128, 128),     # 64-wide outputs: two 32-column tiles per row, halo loaded; de-interleaved columns
    ("down", 3, 48, 32, 32, 80),       # 16x16 outputs, partial chunk, cout tile not full
    ("down", 6, 128, 16, 16, 64),      # 8x8 outputs: tiles of two images
    ("up", 2, 64, 8, 32, 64),          # upsample conv -> 16 x 64 outputs, two-chunk layout
    ("up", 2, 32, 16, 64, 128),        # -> 32 x 128 outputs: halo, 128-channel tile
]


def _ref(kind, x, w):
    if kind == "same":
        return torch.nn.functional.conv2d(x, w, None, 1, 1)
    if kind == "down":
        return torch.nn.functional.conv2d(x, w, None, 2, 1)
    return torch.nn.functional.conv2d(O.upsample2(x), w, None, 1, 1)


@pytest.fixture(params=list(MODES))
def mode(request):
    prec, storage, rnd, tol = MODES[request.param]
    HF.set_precision(prec)
    HF.set_activation_storage(storage)
    try:
        yield request.param, storage, rnd, tol
    finally:
        HF.set_activation_storage(None)
        HF.set_precision(L.PREC_F32)


@pytest.mark.parametrize("kind,B,Cin,H,W,Cout", CASES)
def test_weight_gradient_vs_torch(mode, kind, B, Cin, H, W, Cout):
    name, storage, rnd, tol = mode
    k = 4 if kind == "down" else 3
    g = torch.Generator().manual_seed(7 + B + Cin + Cout)
    q = (lambda t: t.to(rnd).float()) if rnd is not None else (lambda t: t)
    x = q(torch.randn(B, Cin, H, W, generator=g))
    w = q(torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5)
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    yr = _ref(kind, xr, wr)
    gy = q(probe(tuple(yr.shape), 0.3))
    yr.backward(gy.double())
    tdt = rnd if storage else torch.float32
    xd = x.to(tdt).to(DEV).requires_grad_(True)
    wd = w.to(DEV).requires_grad_(True)
    y = HF.conv2d(xd, wd, None, kind)
    y.backward(gy.to(DEV).to(y.dtype))
    ref = wr.grad.float()
    err = float((wd.grad.cpu() - ref).abs().max() / ref.abs().max())
    assert err <= tol, f"{name} {kind} {B}x{Cin}x{H}x{W}->{Cout}: weight gradient max error / max |dw| = {err:.3e} > {tol:.1e}"


def test_second_backward_accumulates(mode):
    """two backward passes into the same .grad: twice the single gradient (the accumulate flag of the sum + unpack pass, or autograd's add)"""
    name, storage, rnd, tol = mode
    g = torch.Generator().manual_seed(3)
    q = (lambda t: t.to(rnd).float()) if rnd is not None else (lambda t: t)
    tdt = rnd if storage else torch.float32
    x = q(torch.randn(2, 32, 16, 16, generator=g)).to(tdt).to(DEV)
    w = q(torch.randn(64, 32, 3, 3, generator=g) / 17.0).to(DEV).requires_grad_(True)
    gy = None
    grads = []
    for rep in range(2):
        y = HF.conv2d(x, w, None, "same")
        if gy is None:
            gy = q(probe(tuple(y.shape), 0.5)).to(DEV).to(y.dtype)
        y.backward(gy)
        grads.append(w.grad.clone())
    assert_close(grads[1], 2.0 * grads[0], 1e-6, "accumulated weight gradient")


def test_the_plan_takes_these_geometries():
    """agan_conv_wgrad_effective_prec keeps the mode for the upsample conv only when the row-resident kernel takes it: a cheap probe that the
    library under test was not loaded with AGAN_WG_ROWS_OFF / _NOUP (which would make the cases above test the older kernels)"""
    lib = L.load()
    HF.set_precision(L.PREC_BF16)
    try:
        gf, pf = HF.conv_geoms("up", 2, 32, 16, 64, 128, 3)[:2]
        assert lib.agan_conv_wgrad_effective_prec(ctypes.byref(gf), pf, L.PREC_BF16) == L.PREC_BF16
        assert lib.agan_conv_wgrad_effective_prec(ctypes.byref(gf), pf, L.PREC_BF16X6) == L.PREC_F16X3
        assert lib.agan_conv_wgrad_effective_prec(ctypes.byref(gf), pf, L.PREC_F32) == L.PREC_F32
    finally:
        HF.set_precision(L.PREC_F32)
