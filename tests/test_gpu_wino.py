"""Winograd kernels of the fp32 mode (csrc/conv_wino.hip, round 3) against torch's CPU fp64 convolution.

AGAN_PREC_F32 routes the 3x3 stride-1 layers (forward, data gradient, weight gradient) and the large 4x4 stride-2 layers (forward: polyphase
F(2x2, 2x2); data gradient: class-wise F(2x2, 2x2)) through Winograd transforms when a layer has enough tiles to fill the chip.  Every case
below is sized to take one plan path (the comment says which); results must agree with the fp64 reference to a few fp32 ulps of the largest
entry -- the transforms add a handful of roundings, nothing more.  (Layers with too few tiles stay on the direct kernels: those are what the
fixed and random shapes of tests/test_gpu_parity.py exercise, against the same kind of reference.)
"""
import importlib

import pytest
import torch

from helpers import probe

pytestmark = pytest.mark.gpu
DEV = "cuda"

HF = importlib.import_module("attention-gan_amd.backend.functional")
L = importlib.import_module("attention-gan_amd.backend.lib")

TOL = 3e-5          # max |error| / max |reference| per tensor

CASES = [
    # kind, B, Cin, H, W, Cout, what
    ("same", 8, 16, 64, 64, 128, "3x3: 128-channel tiles, 4 x 8 tile blocks"),
    ("same", 4, 24, 128, 128, 64, "3x3: 64-channel tiles of 64 tiles (8 x 8 blocks); 24 input channels: weight gradient on the row-resident kernel"),
    ("same", 12, 8, 40, 96, 160, "3x3: ragged block rows (20 tile rows / 4), two cout tiles, the second half empty"),
    ("same", 4, 40, 128, 128, 72, "3x3: Winograd weight gradient (2048 tile octets) with an input-channel tail (40 of 64) and an output-channel tail (8 of 64 in the second tile): [tap][cout][cin] slabs, transposing sum"),
    ("same", 64, 32, 16, 16, 128, "3x3: 16 x 16 maps (two 4 x 8 blocks per image); too few tile octets for the Winograd weight gradient -> row-resident kernel"),
    ("down", 16, 16, 128, 128, 128, "4x4 s2 forward: polyphase, 512 workgroups; data gradient: 64 output channels -> direct kernel"),
    ("down", 8, 128, 128, 128, 128, "4x4 s2 forward: 256 workgroups -> input-channel split + slab sum; data gradient: class-wise Winograd"),
    ("down", 16, 128, 64, 64, 32, "4x4 s2: data gradient class-wise Winograd (128 output channels, 16-channel chunks), forward direct (32 channels)"),
    ("down", 8, 96, 96, 48, 144, "4x4 s2: non-square, ragged blocks, cout tail, both Winograd paths"),
]


def _ref(kind, x, w):
    if kind == "same":
        return torch.nn.functional.conv2d(x, w, None, 1, 1)
    return torch.nn.functional.conv2d(x, w, None, 2, 1)


@pytest.mark.parametrize("kind,B,Cin,H,W,Cout,what", CASES)
def test_fp32_layers_vs_fp64_reference(kind, B, Cin, H, W, Cout, what):
    HF.set_precision(L.PREC_F32)
    k = 4 if kind == "down" else 3
    g = torch.Generator().manual_seed(11 + B + Cin + Cout)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    yr = _ref(kind, xr, wr)
    gy = probe(tuple(yr.shape), 0.3)
    yr.backward(gy.double())
    xd, wd = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)
    y = HF.conv2d(xd, wd, None, kind)
    y.backward(gy.to(DEV))
    for name, got, ref in (("forward", y, yr), ("data gradient", xd.grad, xr.grad), ("weight gradient", wd.grad, wr.grad)):
        ref = ref.detach().float()
        err = float((got.detach().cpu() - ref).abs().max() / ref.abs().max())
        assert err <= TOL, f"{kind} {B}x{Cin}x{H}x{W}->{Cout} ({what}): {name} max error / max |ref| = {err:.3e} > {TOL:.0e}"


def test_padding_rows_and_columns_are_zero_not_neighbours():
    """a one-hot input at an image corner: the halo rows / columns of the Winograd patches must read zeros, not the neighbouring row, image or
    channel (the buffer offsets of out-of-image pixels alias valid memory: they are masked, not range-checked)"""
    HF.set_precision(L.PREC_F32)
    B, Cin, H, W, Cout = 8, 8, 64, 64, 128
    x = torch.zeros(B, Cin, H, W)
    x[:, :, 0, 0] = 1.0
    x[:, :, H - 1, W - 1] = 2.0
    x[:, :, 0, W - 1] = 3.0
    w = torch.randn(Cout, Cin, 3, 3, generator=torch.Generator().manual_seed(5))
    y = HF.conv2d(x.to(DEV), w.to(DEV), None, "same").cpu()
    yr = torch.nn.functional.conv2d(x.double(), w.double(), None, 1, 1).float()
    assert float((y - yr).abs().max()) <= 1e-5 * float(yr.abs().max())


def test_winograd_weight_gradient_accumulates_in_place():
    """agan_conv_wgrad with accumulate = 1 on a geometry the Winograd weight gradient takes (2048 tile octets): the transposing slab sum adds
    into dw instead of overwriting it (the optimiser's flat gradient buffers are written that way); accumulate = 0 ignores what was there."""
    import ctypes
    HF.set_precision(L.PREC_F32)
    lib = L.load()
    B, Cin, H, W, Cout = 4, 40, 128, 128, 72
    g = torch.Generator().manual_seed(21)
    x = torch.randn(B, Cin, H, W, generator=g)
    dy = probe((B, Cout, H, W), 0.3)
    xr = x.double()
    wr = torch.zeros(Cout, Cin, 3, 3, dtype=torch.float64, requires_grad=True)
    torch.nn.functional.conv2d(xr, wr, None, 1, 1).backward(dy.double())
    ref = wr.grad.float()
    gf, pf = HF.conv_geoms("same", B, Cin, H, W, Cout, 3)[:2]
    xd, dyd = x.to(DEV), dy.to(DEV)
    nbytes = lib.agan_conv_wgrad_ws_bytes(ctypes.byref(gf))
    ws = torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=DEV)
    kt = HF.ktable(gf, xd.device)
    dw = torch.full((Cout, Cin, 3, 3), 7.0, device=DEV)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def call(acc):
        L.call("agan_conv_wgrad", ctypes.c_void_p(xd.data_ptr()), ctypes.c_void_p(dyd.data_ptr()), ctypes.c_void_p(dw.data_ptr()), ctypes.byref(gf),
               ctypes.c_void_p(kt.data_ptr()), pf, 3, 3, L.PREC_F32, acc, ctypes.c_void_p(ws.data_ptr()), int(nbytes), stream, None, None)

    call(0)
    first = dw.cpu().clone()
    assert float((first - ref).abs().max() / ref.abs().max()) <= TOL
    call(1)
    second = dw.cpu()
    assert float((second - 2.0 * first).abs().max() / ref.abs().max()) <= 1e-6
