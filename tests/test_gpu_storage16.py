"""16-bit ACTIVATION STORAGE of the one-plane matrix-core modes (round 3; include/agan.h: AGAN_DT_*, csrc/conv_p16.hip).

`HF.set_activation_storage("bf16")` with `set_precision(PREC_BF16)` (or "f16" / PREC_F16) keeps conv outputs, BatchNorm inputs /
outputs and their gradients in the MFMA operand type.  What is checked here, through the C ABI on MI355X:

  * per layer: a conv whose operands ARE exactly representable in the 16-bit type (inputs, weights and the upstream gradient are
    rounded first, on the host) differs from torch's fp32 CPU convolution of those same values only by the ONE rounding of each
    stored result -- 2^-8 (bf16) / 2^-11 (f16) relative -- for the forward, the data gradient and the (fp32) weight gradient, for
    every geometry kind of the row-block gather (3x3, 4x4 stride 2 and its four-class data gradient, the folded upsample conv), with
    fp32 or 16-bit input, K-split launches, the fused LeakyReLU epilogue and the LeakyReLU'-mask data-gradient epilogue;
  * BatchNorm + activation with typed storage against the oracle on the widened values;
  * layers the row-block kernel does not take (rows shorter than a 16-byte block) fall back through fp32 and still compose;
  * the whole configs[1] stage in storage mode is tests/test_gpu_next_rows.py::test_config1_stage1_batch64_bf16[bf16-storage].
"""
import importlib

import pytest
import torch

from helpers import assert_close, probe

pytestmark = pytest.mark.gpu
DEV = "cuda"

HF = importlib.import_module("attention-gan_amd.backend.functional")
L = importlib.import_module("attention-gan_amd.backend.lib")
from oracle import attngan_oracle as O   # noqa: E402  (checker only)

MODES = {"bf16": (L.PREC_BF16, torch.bfloat16, 2.0 ** -8), "f16": (L.PREC_F16, torch.float16, 2.0 ** -11)}


@pytest.fixture(params=["bf16", "f16"])
def mode(request):
    prec, tdt, ulp = MODES[request.param]
    HF.set_precision(prec)
    HF.set_activation_storage(request.param)
    try:
        yield request.param, tdt, ulp
    finally:
        HF.set_activation_storage(None)
        HF.set_precision(L.PREC_F32)


def _ref_conv(kind, x, w, b):
    k = w.shape[-1]
    if kind == "same":
        return torch.nn.functional.conv2d(x, w, b, 1, (k - 1) // 2)
    if kind == "down":
        return torch.nn.functional.conv2d(x, w, b, 2, 1)
    return torch.nn.functional.conv2d(O.upsample2(x), w, b, 1, 1)


CASES = [
    # kind, B, Cin, H, W, Cout, k, 16-bit input?
    ("same", 2, 64, 32, 32, 128, 3, True),       # GK 0, 4x32 tiles
    ("same", 3, 32, 16, 16, 48, 3, True),        # 64-wide N tile, 8x16 tiles, ragged batch tile
    ("same", 2, 64, 32, 32, 64, 3, False),       # fp32 in -> 16-bit out (the first conv after an fp32 producer)
    ("same", 2, 256, 8, 8, 40, 3, True),         # small image, K split over workgroups (slab sum rounds once)
    ("down", 2, 64, 64, 64, 128, 4, True),       # GK 2 forward, GK 1 (four classes) data gradient
    ("down", 2, 64, 64, 64, 128, 4, False),
    ("down", 3, 128, 16, 16, 96, 4, True),       # two staging items per thread, split launches
    ("down", 2, 24, 32, 48, 48, 4, True),        # non-square, channel count not a multiple of the 16 / 64-channel stages
    ("up", 2, 64, 16, 16, 64, 3, True),          # folded upsample conv: GK 1 forward (four classes), GK 2 data gradient
    ("up", 2, 16, 8, 8, 16, 3, True),
    ("down", 6, 256, 8, 8, 128, 4, True),        # 4x4 outputs: tiles of 8 images, K split
    ("same", 10, 128, 8, 8, 64, 3, True),        # 8x8 images, tiles of 2 images, ragged last tile
    ("same", 2, 64, 4, 4, 96, 3, True),          # rows shorter than a 16-byte block: falls back through fp32
]


@pytest.mark.parametrize("kind,B,Cin,H,W,Cout", [("same", 2, 32, 64, 64, 3), ("same", 1, 8, 16, 512, 3), ("down", 2, 3, 64, 64, 16), ("down", 1, 3, 16, 48, 8)])
def test_small_channel_strip_kernels_read_16bit_storage(mode, kind, B, Cin, H, W, Cout):
    """The <= 4-channel row-strip kernels (csrc/conv_small.hip) gather a 16-bit-stored tensor directly: the RGB head reads the 16-bit h_code
    (fp32 image out), the image gradient of a discriminator's first conv reads the 16-bit dY."""
    name, tdt, ulp = mode
    g = torch.Generator().manual_seed(B * 100 + Cin)
    q = lambda t: t.to(tdt).float()
    k = 3 if kind == "same" else 4
    x = q(torch.randn(B, Cin, H, W, generator=g))
    w = q(torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = _ref_conv(kind, xr, wr, None)
    gy = q(probe(yr.shape, 0.3))
    yr.backward(gy)
    in16 = kind == "same"                      # (the first discriminator conv reads the fp32 image)
    xd = (x.to(tdt) if in16 else x).to(DEV).requires_grad_(True)
    wd = w.to(DEV).requires_grad_(True)
    y = HF.conv2d(xd, wd, None, kind)
    y.backward(gy.to(DEV).to(y.dtype))
    tol = 1.5 * ulp
    assert_close(y.float(), yr, tol, "fwd")
    assert_close(xd.grad.float(), xr.grad, tol, "dgrad")
    assert_close(wd.grad, wr.grad, max(tol / 4, 2e-4), "wgrad")


@pytest.mark.parametrize("kind,B,Cin,H,W,Cout,k,in16", CASES)
def test_conv_with_16bit_storage_vs_torch(mode, kind, B, Cin, H, W, Cout, k, in16):
    name, tdt, ulp = mode
    g = torch.Generator().manual_seed(hash((kind, B, Cin, H, Cout)) % 1000)
    q = lambda t: t.to(tdt).float()                                  # values exactly representable in the storage type
    x = q(torch.randn(B, Cin, H, W, generator=g))
    w = q(torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = _ref_conv(kind, xr, wr, None)
    gy = q(probe(yr.shape, 0.3))
    yr.backward(gy)
    xd = (x.to(tdt) if in16 else x).to(DEV).requires_grad_(True)
    wd = w.to(DEV).requires_grad_(True)
    y = HF.conv2d(xd, wd, None, kind)
    if W >= 8:
        assert y.dtype == tdt, "the row-block gather should take this layer"
    else:
        assert y.dtype == torch.float32          # fell back through the fp32-storage kernel
    y.backward(gy.to(DEV).to(y.dtype))
    tol = 1.5 * ulp
    assert_close(y.float(), yr, tol, "fwd")
    assert xd.grad.dtype == xd.dtype
    assert_close(xd.grad.float(), xr.grad, tol, "dgrad")
    assert_close(wd.grad, wr.grad, max(tol / 4, 2e-4), "wgrad (fp32 result of exact operands)")


def test_fused_leaky_relu_and_mask_epilogues_with_16bit_storage(mode):
    """conv + LeakyReLU in the epilogue (no BatchNorm in between) feeding a second conv whose data-gradient epilogue applies
    LeakyReLU'(x): the chain of encode_image_by_16times' first two stages, with the activation stored in 16 bits."""
    name, tdt, ulp = mode
    LAY = importlib.import_module("attention-gan_amd.utilities.layers")
    torch.manual_seed(4)
    chain = LAY.FusedChain().add_stage(0, LAY.HipConv2d(16, 64, 4, 2, 1, False), None, L.ACT_LRELU) \
                            .add_stage(2, LAY.HipConv2d(64, 96, 4, 2, 1, False), None, L.ACT_NONE).to(DEV)
    q = lambda t: t.to(tdt).float()
    with torch.no_grad():
        for p in chain.parameters():
            p.copy_(q(p.cpu()).to(DEV))
    g = torch.Generator().manual_seed(4)
    x = q(torch.randn(2, 16, 64, 64, generator=g))
    w0, w1 = (p.detach().cpu().clone().requires_grad_(True) for p in chain.parameters())
    xr = x.clone().requires_grad_(True)
    hr = q(torch.nn.functional.leaky_relu(torch.nn.functional.conv2d(xr, w0, None, 2, 1), 0.2).detach())   # what the 16-bit tensor holds
    hr_f = torch.nn.functional.leaky_relu(torch.nn.functional.conv2d(xr, w0, None, 2, 1), 0.2)
    h_in = (hr - hr_f.detach()) + hr_f                                    # rounded values, fp32 graph
    yr = torch.nn.functional.conv2d(h_in, w1, None, 2, 1)
    gy = q(probe(yr.shape, 0.7))
    yr.backward(gy)
    xd = x.to(tdt).to(DEV).requires_grad_(True)
    y = chain(xd)
    assert y.dtype == tdt
    y.backward(gy.to(DEV).to(tdt))
    tol = 3 * ulp                                                          # two stored tensors on the way
    assert_close(y.float(), yr, tol, "y")
    assert_close(xd.grad.float(), xr.grad, tol, "dx through the masked epilogue")
    ps = list(chain.parameters())
    assert_close(ps[1].grad, w1.grad, tol, "dw1")
    assert_close(ps[0].grad, w0.grad, tol, "dw0")


@pytest.mark.parametrize("act,B,C,H,res", [(L.ACT_GLU, 4, 64, 32, False), (L.ACT_LRELU, 6, 48, 16, False), (L.ACT_NONE, 3, 32, 64, True),
                                           (L.ACT_LRELU, 2, 512, 4, False), (L.ACT_GLU, 2, 128, 8, False)])
@pytest.mark.parametrize("x16", [True, False])
def test_batchnorm_with_16bit_storage_vs_oracle(mode, act, B, C, H, res, x16):
    """train-mode BatchNorm + GLU / LeakyReLU / residual reading 16-bit (or fp32) conv outputs and writing 16-bit activations:
    forward vs the oracle on the same (widened) values within one output rounding, running statistics at fp32 tolerance, backward
    (dx stored in x's type) within two roundings; large tensors (three-pass kernels) and small ones (one-launch kernels)."""
    name, tdt, ulp = mode
    g = torch.Generator().manual_seed(C + H)
    q = lambda t: t.to(tdt).float()
    x = torch.randn(B, C, H, H, generator=g) * 1.5 + 0.3
    x = q(x) if x16 else x
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1
    co = C // 2 if act == L.ACT_GLU else C
    r = q(torch.randn(B, co, H, H, generator=g)) if res else None
    p = {"bn.weight": gamma.clone().requires_grad_(True), "bn.bias": beta.clone().requires_grad_(True), "bn.running_mean": torch.zeros(C),
         "bn.running_var": torch.ones(C), "bn.num_batches_tracked": torch.tensor(0)}
    xr = x.clone().requires_grad_(True)
    z = O.batchnorm_train(xr, p, "bn")
    yr = O.glu(z) if act == L.ACT_GLU else (O.leaky(z) if act == L.ACT_LRELU else z)
    if res:
        yr = yr + r
    gy = q(probe(yr.shape, 0.9))
    yr.backward(gy)
    xd = (x.to(tdt) if x16 else x).to(DEV).requires_grad_(True)
    gd, bd = gamma.to(DEV).requires_grad_(True), beta.to(DEV).requires_grad_(True)
    rm, rv, nbt = torch.zeros(C, device=DEV), torch.ones(C, device=DEV), torch.zeros((), dtype=torch.int64, device=DEV)
    y = HF.bn_act(xd, gd, bd, rm, rv, nbt, True, act, r.to(tdt).to(DEV) if res else None)
    assert y.dtype == tdt
    y.backward(gy.to(DEV).to(tdt))
    assert_close(y.float(), yr, 1.5 * ulp, "y")
    assert_close(rm, p["bn.running_mean"], 1e-5, "running_mean")
    assert_close(rv, p["bn.running_var"], 1e-5, "running_var")
    assert xd.grad.dtype == xd.dtype
    assert_close(xd.grad.float(), xr.grad, 3 * ulp if x16 else 1e-4, "dx")
    assert_close(gd.grad, p["bn.weight"].grad, 1e-4, "dgamma")
    assert_close(bd.grad, p["bn.bias"].grad, 1e-4, "dbeta")


def test_attention_with_16bit_storage_vs_oracle(mode):
    """word-context attention reading 16-bit image features and writing a 16-bit context (agan_attn_*_dt): forward against the
    oracle on the widened values within one output rounding (the attention map stays fp32: tight), backward -- d(images) stored in
    16 bits, d(words) and d(conv1) fp32 -- within two roundings."""
    name, tdt, ulp = mode
    g = torch.Generator().manual_seed(11)
    B, C, E, T, H = 3, 32, 48, 10, 24
    q = lambda t: t.to(tdt).float()
    images, words = q(torch.randn(B, C, H, H, generator=g)), torch.randn(B, E, T, generator=g)
    w = torch.randn(C, E, 1, 1, generator=g) / E ** 0.5
    mask = O.make_mask([10, 4, 7])
    ir, wr, cr = images.clone().requires_grad_(True), words.clone().requires_grad_(True), w.clone().requires_grad_(True)
    ctx_r, attn_r = O.attention_module(ir, wr, cr, mask)
    gy = q(probe(ctx_r.shape, 0.4))
    (ctx_r * gy).sum().backward()
    idv = images.to(tdt).to(DEV).requires_grad_(True)
    wd, cd = words.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)
    ctx_d, attn_d = HF.attention(idv, wd, cd, mask.to(DEV))
    assert ctx_d.dtype == tdt and attn_d.dtype == torch.float32
    ctx_d.backward(gy.to(DEV).to(tdt))
    assert_close(attn_d, attn_r, 1e-4, "attention map")
    assert_close(ctx_d.float(), ctx_r, 1.5 * ulp, "context")
    assert idv.grad.dtype == tdt
    assert_close(idv.grad.float(), ir.grad, 3 * ulp, "d images")
    assert_close(wd.grad, wr.grad, 1e-4, "d words")
    assert_close(cd.grad, cr.grad, 1e-4, "d conv1")
