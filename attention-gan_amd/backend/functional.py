"""torch.autograd bridges onto the C ABI of libagan_hip.so.

PyTorch is plumbing here: it owns device memory (caching allocator), the current HIP stream and the autograd
tape.  Every numerical op on the hot path is a hand-written gfx950 kernel reached through `lib.call`.
All tensors are NCHW fp32 like the reference's.
"""
from __future__ import annotations

import ctypes
import math
from ctypes import byref, c_void_p
from typing import Dict, Optional, Tuple

import contextlib
import os
import weakref

import torch
from torch import Tensor
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import lib as L

# arithmetic mode used by the MFMA contractions; PREC_F32 = exact fp32 products (parity mode, the default)
_PRECISION = [L.PREC_F32]


def set_precision(p: int) -> None:
    if int(p) != _PRECISION[0]:
        # the batched re-pack (packed_weight / _repack_group) refreshes EVERY registered pack of a layout family: forget the ones the
        # old mode registered (they re-register on their next use), or a later mode keeps re-packing layouts nothing reads
        for rng in _FLAT_RANGES:
            rng[3], rng[4] = [], None
    _PRECISION[0] = int(p)


def get_precision() -> int:
    return _PRECISION[0]


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream() -> c_void_p:
    """raw handle of torch's current HIP stream (what every library call enqueues on)"""
    if _RAW_STREAM is not None:              # ~10x cheaper than building a torch.cuda.Stream object per launch
        return c_void_p(_RAW_STREAM(torch.cuda.current_device()))
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t: Optional[Tensor]) -> Optional[c_void_p]:
    return None if t is None else c_void_p(t.data_ptr())


def _dev(t: Tensor, what: str) -> Tensor:
    if not t.is_cuda:
        raise L.AganError(f"{what}: tensor is on {t.device}; the AttnGAN HIP path runs on an MI355X only (no CPU fallback)")
    if t.dtype != torch.float32:
        if t.dtype in _DT_OF:            # a 16-bit activation arriving at a kernel without typed storage: widen at the boundary
            return t.float().contiguous()
        raise L.AganError(f"{what}: expected float32, got {t.dtype}")
    return t.contiguous()


# ---- 16-bit activation storage (include/agan.h: AGAN_DT_*) ------------------------------------------------------------------
# Off by default: every tensor in HBM is fp32 like the reference's.  set_activation_storage("bf16" | "f16") -- valid with
# set_precision(PREC_BF16 / PREC_F16) -- lets conv outputs, BatchNorm inputs / outputs and their gradients live in the operand
# type of the matrix cores: rounded ONCE when stored, loaded without conversion, half the bytes.  Kernels without typed storage
# (attention, the <= 4-channel convs, losses) widen their inputs at the boundary (_dev).
_DT_OF = {torch.bfloat16: L.DT_BF16, torch.float16: L.DT_F16}
_TORCH_OF = {L.DT_F32: torch.float32, L.DT_BF16: torch.bfloat16, L.DT_F16: torch.float16}
_STORAGE = [L.DT_F32]


def set_activation_storage(kind) -> None:
    dt = {None: L.DT_F32, "f32": L.DT_F32, "bf16": L.DT_BF16, "f16": L.DT_F16}[kind]
    _STORAGE[0] = dt


def get_activation_storage() -> int:
    """the storage type 16-bit-capable kernels produce right now: AGAN_DT_F32 unless the precision mode matches the request"""
    dt = _STORAGE[0]
    if dt == L.DT_BF16 and _PRECISION[0] == L.PREC_BF16:
        return dt
    if dt == L.DT_F16 and _PRECISION[0] == L.PREC_F16:
        return dt
    return L.DT_F32


def _dt(t: Tensor) -> int:
    return L.DT_F32 if t.dtype == torch.float32 else _DT_OF[t.dtype]


def _act(t: Tensor, what: str) -> Tensor:
    """an activation tensor for a kernel WITH typed storage: fp32 or a 16-bit type, contiguous, on the device"""
    if not t.is_cuda:
        raise L.AganError(f"{what}: tensor is on {t.device}; the AttnGAN HIP path runs on an MI355X only (no CPU fallback)")
    if t.dtype != torch.float32 and t.dtype not in _DT_OF:
        raise L.AganError(f"{what}: expected float32 / bfloat16 / float16, got {t.dtype}")
    return t.contiguous()


_DT_SUPPORT: dict = {}


def _gather_types(g: "L.ConvGeom", pe: int, x: Tensor, want_out: int):
    """-> (x, storage type of the output) for a gather of geometry g in arithmetic mode pe: the 16-bit storage the caller would like
    where the kernel for this geometry offers it (agan_conv_gather_dt_supported), else fp32 -- widening a 16-bit x when the layer
    cannot read it as it stands (the deep 4x4 layers, whose rows are shorter than one 16-byte block)."""
    def ok(xin, out):
        if xin == L.DT_F32 and out == L.DT_F32:
            return True
        key = (g.B, g.Cin, g.IH, g.IW, g.Cout, g.OH, g.OW, g.R, g.S, g.OS, g.SY, g.DY, g.OY[0], g.OY[1], pe, xin, out)
        hit = _DT_SUPPORT.get(key)
        if hit is None:
            hit = _DT_SUPPORT[key] = bool(L.load().agan_conv_gather_dt_supported(byref(g), pe, xin, out))
        return hit
    outs = [want_out, L.DT_F32] if want_out != L.DT_F32 else [L.DT_F32]
    for out in outs:
        if ok(_dt(x), out):
            return x, out
    x = x.float()
    for out in outs:
        if ok(L.DT_F32, out):
            return x, out
    return x, L.DT_F32


def _ws(nbytes: int, like: Tensor) -> Tuple[Optional[Tensor], Optional[c_void_p]]:
    if nbytes <= 0:
        return None, None
    t = torch.empty(nbytes, dtype=torch.uint8, device=like.device)
    return t, c_void_p(t.data_ptr())


# --------------------------------------------------------------------------------------------------------------
# gradient destinations: optim.FlatAdam tags every parameter with `_agan_grad_dst = (flat_grad, offset, numel)`.  Backward
# kernels then write weight gradients straight into the flat buffer and hand autograd a fresh view of it, so AccumulateGrad
# adopts the view (p.grad is None at that point) instead of launching one `grad += new` kernel per parameter.
# --------------------------------------------------------------------------------------------------------------
def grad_dst(param):
    return getattr(param, "_agan_grad_dst", None)


def _grad_out(dst, shape, like: Tensor):
    """-> (buffer the backward kernel writes, accumulate flag for the kernel, value handed back to autograd).

    Without a flat destination: a scratch tensor that autograd accumulates as usual.  With one, the FIRST contribution of a
    backward pass overwrites the parameter's slice of the flat buffer and autograd adopts a view of it as p.grad; a parameter
    used again in the same backward (a discriminator sees the real and the fake batch) has the kernel ADD into the slice and
    autograd gets None for that edge -- no `grad += new` launches, no copies."""
    if dst is None:
        t = torch.empty(shape, dtype=torch.float32, device=like.device)
        return t, 0, t
    if dst.locked:
        raise L.AganError("a backward kernel is about to write into a flat gradient slice whose bucket is already being all-reduced: "
                          "call zero_grad() (and GradBuckets.arm()) before running another backward under data parallelism")
    view = dst.flat[dst.offset:dst.offset + dst.numel].view(shape)
    dst.edges += 1
    if dst.written:
        return view, 1, None
    dst.written = True
    return view, 0, view


class GradDst:
    """Slice of a FlatAdam gradient buffer that belongs to one parameter.  `written`: a kernel has produced the slice in this
    backward; `edges`: how many kernel contributions went into it (FlatAdam._rebind tells a harmless autograd clone from a summed
    stock edge with it); `locked`: the slice's bucket has been handed to the gradient exchange (GradBuckets._launch)."""
    __slots__ = ("flat", "offset", "numel", "written", "edges", "locked")

    def __init__(self, flat: Tensor, offset: int, numel: int):
        self.flat, self.offset, self.numel, self.written, self.edges, self.locked = flat, offset, numel, False, 0, False

    def reset(self) -> None:
        self.written, self.edges, self.locked = False, 0, False


# --------------------------------------------------------------------------------------------------------------
# convolution geometry (see include/agan.h: agan_conv_geom)
# --------------------------------------------------------------------------------------------------------------
def _geom(B, Cin, IH, IW, Cout, OH, OW, R, S, OS, SY, DY, OY) -> L.ConvGeom:
    g = L.ConvGeom()
    g.B, g.Cin, g.IH, g.IW, g.Cout, g.OH, g.OW = B, Cin, IH, IW, Cout, OH, OW
    g.R, g.S, g.OS, g.SY, g.DY = R, S, OS, SY, DY
    g.OY[0], g.OY[1] = OY
    return g


_GEOM_CACHE = {}


def conv_geoms(kind: str, B: int, Cin: int, H: int, W: int, Cout: int, k: int):
    """-> (fwd geom, fwd pack mode, dgrad geom, dgrad pack mode, (OH, OW)) for the conv kinds on the path.

    same : k x k, stride 1, pad (k-1)/2          utilities/layers.py:45-53 (and nn.Linear as 1x1 on a 1x1 image)
    down : 4 x 4, stride 2, pad 1                utilities/layers.py:122,139-150
    up   : nearest x2 upsample then 3x3 pad 1    utilities/layers.py:64-65, folded into 4 parity classes of 2x2 taps
    """
    key = (kind, B, Cin, H, W, Cout, k)
    hit = _GEOM_CACHE.get(key)
    if hit is not None:
        return hit
    if kind == "same":
        p = (k - 1) // 2
        out = (_geom(B, Cin, H, W, Cout, H, W, k, k, 1, 1, 1, (-p, -p)), L.PACK_FWD,
               _geom(B, Cout, H, W, Cin, H, W, k, k, 1, 1, 1, (-p, -p)), L.PACK_DGRAD_S1, (H, W))
    elif kind == "down":
        if k != 4 or H % 2 or W % 2:
            raise L.AganError(f"down conv needs k=4 and even H,W (got k={k}, {H}x{W})")
        out = (_geom(B, Cin, H, W, Cout, H // 2, W // 2, 4, 4, 1, 2, 1, (-1, -1)), L.PACK_FWD,
               _geom(B, Cout, H // 2, W // 2, Cin, H, W, 2, 2, 2, 1, -1, (0, 1)), L.PACK_DGRAD_4x4S2, (H // 2, W // 2))
    elif kind == "up":
        if k != 3:
            raise L.AganError("upsample conv needs k=3")
        out = (_geom(B, Cin, H, W, Cout, 2 * H, 2 * W, 2, 2, 2, 1, 1, (-1, 0)), L.PACK_UP_FWD,
               _geom(B, Cout, 2 * H, 2 * W, Cin, H, W, 4, 4, 1, 2, 1, (-1, -1)), L.PACK_UP_DGRAD, (2 * H, 2 * W))
    else:
        raise L.AganError(f"unknown conv kind {kind!r}")
    _GEOM_CACHE[key] = out
    return out


# Packed weights are cached per owning module (a plain dict the module passes in): a parameter is re-packed only when
# it changed -- in-place version bump (load_state_dict), new storage (.to(), FlatAdam re-homing) or an optimiser step
# that wrote through the raw pointer.  Optimiser steps are tracked per flat buffer (register_flat): the generator's Adam
# step must not invalidate the discriminators' packed weights.  Writes through a raw pointer into memory that is not a
# registered flat buffer bump the global epoch instead.  Without a cache dict every call re-packs.
_WEIGHT_EPOCH = [0]
_FLAT_RANGES: list = []          # [start address, end address, epoch]


def bump_weight_epoch() -> None:
    _WEIGHT_EPOCH[0] += 1


def register_flat(flat: Tensor) -> None:
    """Declare a flat parameter buffer whose views are conv/linear weights and which agan_adam_step updates in place."""
    start = flat.data_ptr()
    end = start + flat.numel() * flat.element_size()
    # a buffer registered earlier that overlaps this one has been freed (the allocator handed its memory out again): drop it,
    # and start above every epoch seen so far so that no cached pack of recycled memory can look fresh
    top = max([r[2] for r in _FLAT_RANGES] + [0])
    _FLAT_RANGES[:] = [r for r in _FLAT_RANGES if r[1] <= start or r[0] >= end]
    _FLAT_RANGES.append([start, end, top + 1, [], None])     # [start, end, epoch, pack entries, device job table]


def unregister_flat(flat: Tensor) -> None:
    start = flat.data_ptr()
    _FLAT_RANGES[:] = [r for r in _FLAT_RANGES if r[0] != start]


def _flat_range(ptr: int):
    for r in _FLAT_RANGES:
        if r[0] <= ptr < r[1]:
            return r
    return None


_PACK_JOB_DTYPE = [("w", "<u8"), ("wk", "<u8"), ("mode", "<i4"), ("cout", "<i4"), ("cin", "<i4"), ("kh", "<i4"), ("kw", "<i4"),
                   ("first_block", "<i4")]          # include/agan.h: agan_pack_job


def _pack_table(rng, prec: int):
    """device job table of a flat buffer's pack entries of ONE layout family (`prec`: the fp32 k-table layouts, or one of the 16-bit
    patch layouts); built on the host, one H2D copy; rebuilt when an entry is added"""
    import numpy as np
    entries = [e for e in rng[3] if e["prec"] == prec]
    if rng[4] is None:
        rng[4] = {}
    hit = rng[4].get(prec)
    if hit is None or hit[1] != len(entries):
        if torch.cuda.is_current_stream_capturing():
            raise L.AganError("packed-weight job table missing during HIP-graph capture: run a warm-up step and "
                              "build_pack_tables() first (GanTrainStep.capture does)")
        lib = L.load()
        jobs = np.zeros(len(entries), dtype=_PACK_JOB_DTYPE)
        first = 0
        for i, e in enumerate(entries):
            cout, cin, kh, kw = e["dims"]
            jobs[i] = (e["ptr"], e["wk"].data_ptr(), e["mode"], cout, cin, kh, kw, first)
            first += lib.agan_pack_job_blocks_prec(e["mode"], cout, cin, kh, kw, prec)
        table = torch.from_numpy(jobs.view(np.uint8).copy()).to(entries[0]["wk"].device)
        hit = rng[4][prec] = (table, len(entries), first)
    return hit


def build_pack_tables() -> None:
    """Build every missing job table now (host work + H2D copies), e.g. before capturing a HIP graph."""
    for rng in _FLAT_RANGES:
        for prec in sorted({e["prec"] for e in rng[3]}):
            _pack_table(rng, prec)


def _repack_group(rng, prec: int) -> None:
    """Re-pack every registered packed weight of one flat buffer and one layout family in a single launch (agan_pack_weights).
    After an optimiser step all of them are stale at once, and most are a few KB: one launch instead of one ~10-17 us launch per
    tensor and mode (1.7 ms per step in the 16-bit modes before they took this path too)."""
    table, n, total = _pack_table(rng, prec)
    L.call("agan_pack_weights", _p(table), n, total, prec, _stream())
    for e in rng[3]:
        if e["prec"] == prec:
            e["cache"][e["key"]] = ((e["ptr"], e["tver"], _WEIGHT_EPOCH[0], rng[2]) + e["dims"], e["wk"], rng)


_EFF_PREC = {}


def effective_precision(g: L.ConvGeom) -> int:
    """The arithmetic mode a gather geometry really runs in under the current precision setting (include/agan.h:
    agan_conv_effective_prec): the 16-bit patch kernels take the 3x3 / 2x2-per-class / 4x4-s2 geometries, everything else is fp32."""
    prec = _PRECISION[0]
    if prec == L.PREC_F32:
        return prec
    key = (prec, g.B, g.Cin, g.IH, g.IW, g.Cout, g.OH, g.OW, g.R, g.S, g.OS, g.SY, g.DY, g.OY[0], g.OY[1])
    hit = _EFF_PREC.get(key)
    if hit is None:
        hit = _EFF_PREC[key] = int(L.load().agan_conv_effective_prec(byref(g), prec))
    return hit


_EFF_WPREC = {}


def wgrad_effective_precision(g: L.ConvGeom, pack_mode: int) -> int:
    """the arithmetic mode the weight gradient of forward geometry g runs in (include/agan.h: agan_conv_wgrad_effective_prec)"""
    prec = _PRECISION[0]
    if prec == L.PREC_F32:
        return prec
    key = (prec, pack_mode, g.B, g.Cin, g.IH, g.IW, g.Cout, g.OH, g.OW, g.R, g.S, g.OS, g.SY, g.DY, g.OY[0], g.OY[1])
    hit = _EFF_WPREC.get(key)
    if hit is None:
        hit = _EFF_WPREC[key] = int(L.load().agan_conv_wgrad_effective_prec(byref(g), pack_mode, prec))
    return hit


def packed_weight(w: Tensor, mode: int, cache: Optional[dict] = None, prec: Optional[int] = None) -> Tensor:
    cout, cin, kh, kw = w.shape
    if prec is None:
        prec = _PRECISION[0]
    key = (mode, prec)
    hit = cache.get(key) if cache is not None else None
    ptr = w.data_ptr()
    rng = hit[2] if (hit is not None and hit[0][0] == ptr) else _flat_range(ptr)
    ver = (ptr, w._version, _WEIGHT_EPOCH[0], rng[2] if rng is not None else -1, cout, cin, kh, kw)
    if hit is not None and hit[0] == ver:
        return hit[1]
    if hit is not None and rng is not None and hit[0][:2] == ver[:2] and hit[0][4:] == ver[4:] \
            and any(e["cache"] is cache and e["key"] == key for e in rng[3]):
        # same tensor, only the optimiser epoch moved: its whole group (same layout family) is stale -> one batched launch
        _repack_group(rng, prec)
        hit = cache[key]
        if hit[0] == ver:
            return hit[1]
    n = L.load().agan_packed_weight_bytes(mode, cout, cin, kh, kw, prec)
    if n == 0:
        raise L.AganError(f"pack mode {mode} / precision {prec} does not take a {kh}x{kw} kernel")
    reuse = hit is not None and hit[1].numel() == n and hit[1].device == w.device
    wk = hit[1] if reuse else torch.empty(n, dtype=torch.uint8, device=w.device)
    L.call("agan_pack_weight", _p(w), _p(wk), mode, cout, cin, kh, kw, prec, _stream())
    if cache is not None:
        cache[key] = (ver, wk, rng)
        if rng is not None:
            ent = next((e for e in rng[3] if e["cache"] is cache and e["key"] == key), None)
            if ent is None:
                ent = {"cache": cache, "key": key}
                rng[3].append(ent)
            ent.update(ptr=ptr, tver=w._version, dims=(cout, cin, kh, kw), mode=mode, wk=wk, prec=prec)
            rng[4] = None                       # job tables are rebuilt at the next batched re-pack
    return wk


_KTABLE_CACHE = {}


def ktable(g: L.ConvGeom, device) -> Tensor:
    """Reduction-index table of a geometry (include/agan.h: agan_conv_ktable), built once per (geometry, device)."""
    key = (g.Cin, g.IH, g.IW, g.R, g.S, g.DY, str(device))
    t = _KTABLE_CACHE.get(key)
    if t is None:
        n = L.load().agan_conv_ktable_elems(byref(g))
        t = torch.empty(n, dtype=torch.int32, device=device)
        L.call("agan_conv_ktable", byref(g), _p(t), _stream())
        _KTABLE_CACHE[key] = t
    return t


# optional launch observer (bench.py times the conv engine with HIP events through it): begin(kind, phase, geom) / end()
_OBSERVER = [None]


def set_launch_observer(obs) -> None:
    _OBSERVER[0] = obs


# --------------------------------------------------------------------------------------------------------------
# amax slots (AGAN_PREC_F16X3, include/agan.h).  The fp16 split mode needs max|x| of every tensor a conv gathers.  The kernel that
# PRODUCES the tensor (BatchNorm forward / backward apply, a conv epilogue) folds that maximum into a small device slot while
# it writes the tensor; the slot travels with the tensor OBJECT (autograd hands the same Python object from a producer Function
# to its consumer, forward and backward) through an identity-keyed registry.  A tensor that arrives without a slot (torch.cat,
# the image batch, a gradient autograd summed from two edges) gets one from agan_absmax: one extra read of that tensor.
# Slots come from a zeroed arena window that amax_begin_step() opens (ONE fill launch per train step, graph-capturable);
# outside a step every slot is its own torch.zeros.  Only LARGE tensors take the fused route (_AMAX_FUSE_MIN).
# --------------------------------------------------------------------------------------------------------------
_AMAX_REG: dict = {}
_AMAX_ARENA: dict = {}            # device -> [arena tensor, window index, next slot, limit]
_AMAX_WINDOW, _AMAX_WINDOWS = 1024, 4


def _needs_amax() -> bool:
    """modes whose kernels want max|x| of the tensors they gather: the fp16 split (every gather and weight gradient) and bf16x6 (its
    weight gradients run on the fp16-split kernel: include/agan.h, agan_conv_wgrad_effective_prec)"""
    return _PRECISION[0] in (L.PREC_F16X3, L.PREC_BF16X6)


def amax_begin_step(device) -> None:
    """Open a fresh zeroed window of amax slots for one train step (no-op unless a mode that needs them is on)."""
    if not _needs_amax():
        return
    ent = _AMAX_ARENA.get(device)
    if ent is None:
        ent = _AMAX_ARENA[device] = [torch.zeros(_AMAX_WINDOWS * _AMAX_WINDOW * L.AMAX_SLOT, dtype=torch.float32, device=device), -1, 0, 0]
    ent[1] = (ent[1] + 1) % _AMAX_WINDOWS
    lo = ent[1] * _AMAX_WINDOW * L.AMAX_SLOT
    ent[0][lo:lo + _AMAX_WINDOW * L.AMAX_SLOT].zero_()
    ent[2], ent[3] = ent[1] * _AMAX_WINDOW, (ent[1] + 1) * _AMAX_WINDOW


def _amax_new(device) -> Tensor:
    ent = _AMAX_ARENA.get(device)
    if ent is not None and ent[2] < ent[3]:
        i = ent[2]
        ent[2] += 1
        return ent[0][i * L.AMAX_SLOT:(i + 1) * L.AMAX_SLOT]
    return torch.zeros(L.AMAX_SLOT, dtype=torch.float32, device=device)


def _amax_put(t: Tensor, slot: Tensor) -> None:
    k = id(t)
    _AMAX_REG[k] = (weakref.ref(t, lambda _r, k=k: _AMAX_REG.pop(k, None)), slot)


def amax_of(t: Tensor) -> Tensor:
    """the amax slot of a tensor: the one its producer filled, else a fresh one filled by agan_absmax"""
    e = _AMAX_REG.get(id(t))
    if e is not None and e[0]() is t:
        return e[1]
    slot = _amax_new(t.device)
    L.call("agan_absmax", _p(t), t.numel(), _p(slot), _stream())
    _amax_put(t, slot)
    return slot


_AMAX_FUSE_MIN = int(os.environ.get("AGAN_AMAX_FUSE_MIN", 1 << 22))      # elements: below this a separate agan_absmax launch (~5 us) is as cheap as the fused commit


def _amax_out(t: Tensor) -> Optional[Tensor]:
    """slot for a kernel that is about to produce the LARGE tensor `t` (None unless the fp16 split mode is on): the producer folds
    max|t| into it while it streams the tensor, which saves the consumer conv a second pass over tens of megabytes"""
    if not _needs_amax() or t.numel() < _AMAX_FUSE_MIN:
        return None
    slot = _amax_new(t.device)
    _amax_put(t, slot)
    return slot


def _gather(x: Tensor, wk: Tensor, bias: Optional[Tensor], g: L.ConvGeom, out: Tensor, kind: str = "", phase: str = "",
            act: int = 0, lrelu_mask: Optional[Tensor] = None, prec: int = L.PREC_F32, in_amax: Optional[Tensor] = None) -> None:
    lib = L.load()
    nbytes = lib.agan_conv_gather_ws_bytes(byref(g), prec)
    ws, wsp = _ws(nbytes, x)
    kt = ktable(g, x.device)
    obs = _OBSERVER[0]
    if obs is not None:
        obs.begin(kind, phase, g, x.element_size(), out.element_size())
    out_amax = _amax_out(out) if prec != L.PREC_F32 else None       # the 16-bit kernels fold max|out| into a slot as they store
    L.call("agan_conv_gather_dt", _p(x), _p(wk), _p(bias), _p(out), byref(g), _p(kt), prec, act, _p(lrelu_mask), wsp, nbytes,
           _stream(), _p(in_amax), _p(out_amax), _dt(x), _dt(out))
    if obs is not None:
        obs.end()


# Weight gradients feed nothing but the optimiser, so a training step may fork them onto a side stream: they then overlap the
# HBM-bound kernels (BatchNorm / activation backward, slab sums) of the data-gradient chain instead of queueing between them.
_SIDE_STREAMS: Dict[int, "torch.cuda.Stream"] = {}
_SIDE_DIRTY: Dict[int, bool] = {}
_WGRAD_SIDE = [False]


def set_wgrad_side_stream(flag: bool) -> bool:
    """Fork weight/bias gradients that go straight into a flat gradient buffer (grad_dst) onto a per-stream side stream.
    Whoever reads those gradients must call join_side_stream() on the forking stream first (FlatAdam.step and GradBuckets do)."""
    old = _WGRAD_SIDE[0]
    _WGRAD_SIDE[0] = bool(flag)
    return old


def wgrad_side_stream_enabled() -> bool:
    return _WGRAD_SIDE[0]


def join_all_side_streams(target) -> None:
    """Make `target` (a stream about to read gradients, e.g. the collective stream) wait for every forked side stream."""
    for key, side in _SIDE_STREAMS.items():
        if _SIDE_DIRTY.get(key):
            target.wait_stream(side)


def set_side_stream_for(stream, side) -> None:
    """Use `side` for the weight gradients forked off `stream` (instead of a private one).  The train step maps each
    discriminator stream to the stream it was itself forked from: that stream idles during the discriminator updates, and a
    fork of a fork does not survive hipStreamEndCapture on ROCm 7.2."""
    _SIDE_STREAMS[stream.cuda_stream] = side


def _fork_side_stream():
    cur = torch.cuda.current_stream()
    key = cur.cuda_stream
    side = _SIDE_STREAMS.get(key)
    if side is None:
        side = _SIDE_STREAMS[key] = torch.cuda.Stream(device=cur.device)
    side.wait_stream(cur)
    _SIDE_DIRTY[key] = True
    return side


def join_side_stream() -> None:
    """Make the current stream wait for the weight-gradient work forked from it (no-op if there is none)."""
    if not torch.cuda.is_available():
        return
    cur = torch.cuda.current_stream()
    key = cur.cuda_stream
    if _SIDE_DIRTY.get(key):
        cur.wait_stream(_SIDE_STREAMS[key])
        _SIDE_DIRTY[key] = False


class ActHandoff:
    """Links a conv whose epilogue applied LeakyReLU (the producer) to the ONE conv that consumes its output: the consumer's
    data-gradient epilogue multiplies by LeakyReLU'(its input) -- which is the producer's activation backward -- and says so
    here, so the producer skips the separate activation-backward pass.  One object per forward call of the pair."""
    __slots__ = ("masked",)

    def __init__(self):
        self.masked = False


class _ConvFn(Function):
    @staticmethod
    def forward(ctx, x: Tensor, weight: Tensor, bias: Optional[Tensor], kind: str, cache: Optional[dict], wdst, bdst, act: int = 0,
                handoff_out: Optional[ActHandoff] = None, handoff_in: Optional[ActHandoff] = None):
        x = _act(x, "conv input")
        w = _dev(weight.detach(), "conv weight")
        B, Cin, H, W = x.shape
        Cout, Cin_w, kh, kw = w.shape
        if Cin != Cin_w or kh != kw:
            raise L.AganError(f"conv: input has {Cin} channels, weight {tuple(w.shape)}")
        gf, pf, gd, pd, (OH, OW) = conv_geoms(kind, B, Cin, H, W, Cout, kh)
        pe = effective_precision(gf)
        x, odt = _gather_types(gf, pe, x, get_activation_storage())       # 16-bit activation storage where the layer offers it
        out = torch.empty((B, Cout, OH, OW), dtype=_TORCH_OF[odt], device=x.device)
        b = _dev(bias.detach(), "conv bias") if bias is not None else None
        # the amax slot of x: for the fp16-split gather, and for a weight gradient that runs on the fp16-split kernel (f16x3, bf16x6) --
        # taken HERE, where x still is the object its producer registered a slot for
        need_xs = pe == L.PREC_F16X3 or (ctx.needs_input_grad[1] and _needs_amax() and wgrad_effective_precision(gf, pf) == L.PREC_F16X3)
        xs = amax_of(x) if need_xs else None
        _gather(x, packed_weight(w, pf, cache, pe), b, gf, out, kind, "fwd", act, None, pe, xs)
        ctx.x_scale = xs
        if act == L.ACT_NONE:
            ctx.save_for_backward(x, w)
        else:                       # fused LeakyReLU: the backward needs the sign of the output
            ctx.save_for_backward(x, w, out)
        ctx.kind, ctx.has_bias, ctx.cache, ctx.wdst, ctx.bdst, ctx.act = kind, bias is not None, cache, wdst, bdst, act
        ctx.handoff_out = handoff_out if act != L.ACT_NONE else None
        # the consumer can fold the producer's LeakyReLU backward into its dgrad epilogue on the fp32 MFMA path (> 4 channels)
        ctx.handoff_in = handoff_in if (handoff_in is not None and Cin > 4) else None
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dy: Tensor):
        x, w = ctx.saved_tensors[:2]
        dy = _act(dy, "conv grad")
        if ctx.act != L.ACT_NONE:
            if ctx.handoff_out is not None and ctx.handoff_out.masked:
                ctx.handoff_out.masked = False          # the consumer's dgrad epilogue already applied LeakyReLU'(out)
            else:
                out = _dev(ctx.saved_tensors[2], "conv output")
                dy = _dev(dy, "conv grad")
                dz = torch.empty_like(dy)
                L.call("agan_act_bwd", _p(out), _p(dy), _p(dz), out.numel(), ctx.act, _stream())
                dy = dz
        B, Cin, H, W = x.shape
        Cout, _, kh, kw = w.shape
        gf, pf, gd, pd, _ = conv_geoms(ctx.kind, B, Cin, H, W, Cout, kh)
        dx = dw = db = None
        want_w, want_b = ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
        pe_d = effective_precision(gd) if ctx.needs_input_grad[0] else L.PREC_F32
        # weight gradient: the mode the library will run it in (the fp16 split needs its operands' amax slots only then)
        pe_w = wgrad_effective_precision(gf, pf)
        scaled_w = want_w and pe_w == L.PREC_F16X3
        dys = amax_of(dy) if (pe_d == L.PREC_F16X3 or scaled_w) else None      # ONE slot of dy serves both gradients
        side = None
        if _WGRAD_SIDE[0] and want_w and ctx.wdst is not None and (not want_b or ctx.bdst is not None):
            kt = ktable(gf, x.device)                  # (tables are built on the forking stream)
            side = _fork_side_stream()                 # dy is ready on the current stream at this point
            x.record_stream(side)
            dy.record_stream(side)
        with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
            if want_w:
                dwbuf, wacc, dw = _grad_out(ctx.wdst, w.shape, x)
                lib = L.load()
                nbytes = lib.agan_conv_wgrad_ws_bytes(byref(gf))
                ws, wsp = _ws(nbytes, x)
                kt = ktable(gf, x.device)
                obs = _OBSERVER[0]
                if obs is not None:
                    obs.begin(ctx.kind, "wgrad", gf, x.element_size(), dy.element_size())
                xs = None
                if scaled_w:
                    xs = ctx.x_scale if ctx.x_scale is not None else amax_of(x)
                xw, dyw = x, dy
                if (xw.dtype != torch.float32 or dyw.dtype != torch.float32) and not lib.agan_conv_wgrad_dt_supported(
                        byref(gf), pf, _PRECISION[0], _dt(xw), _dt(dyw)):
                    xw, dyw = xw.float(), dyw.float()         # a weight gradient without typed storage: widen its operands
                L.call("agan_conv_wgrad_dt", _p(xw), _p(dyw), _p(dwbuf), byref(gf), _p(kt), pf, kh, kw, pe_w, wacc, wsp, nbytes,
                       _stream(), _p(xs), _p(dys if scaled_w else None), _dt(xw), _dt(dyw))
                if obs is not None:
                    obs.end()
            if want_b:
                dbbuf, bacc, db = _grad_out(ctx.bdst, (Cout,), x)
                L.call("agan_bias_grad", _p(_dev(dy, "conv grad")), _p(dbbuf), B, Cout, dy.shape[2] * dy.shape[3], bacc, _stream())
        if ctx.needs_input_grad[0]:
            # dx takes x's storage type where the data-gradient kernel offers it (autograd wants the gradient in the input's dtype)
            dyg, ddt = _gather_types(gd, pe_d, dy, _dt(x))
            dx = torch.empty(x.shape, dtype=_TORCH_OF[ddt], device=x.device)
            mask = x if ctx.handoff_in is not None else None       # x is the producer's LeakyReLU output
            if mask is not None and mask.dtype != dx.dtype:
                mask = mask.to(dx.dtype)
            _gather(dyg, packed_weight(w, pd, ctx.cache, pe_d), None, gd, dx, ctx.kind, "dgrad", L.ACT_NONE, mask, pe_d,
                    dys if pe_d == L.PREC_F16X3 else None)
            if mask is not None:
                ctx.handoff_in.masked = True
        return dx, dw, db, None, None, None, None, None, None, None


def conv_fuses_activation(act: int, cout: int) -> bool:
    """can conv2d apply `act` in its epilogue? (LeakyReLU on the MFMA paths; include/agan.h: agan_conv_gather)"""
    return act == L.ACT_LRELU and cout > 4


def conv2d(x: Tensor, weight: Tensor, bias: Optional[Tensor] = None, kind: str = "same", cache: Optional[dict] = None,
           wdst=None, bdst=None, act: int = 0, handoff_out: Optional[ActHandoff] = None,
           handoff_in: Optional[ActHandoff] = None) -> Tensor:
    """conv forward with autograd (dgrad + wgrad kernels).  kind: 'same' | 'down' | 'up' (conv_geoms).
    wdst / bdst: optional flat-gradient destinations of weight / bias (grad_dst(param)).
    act: ACT_LRELU applies the activation in the conv epilogue (see conv_fuses_activation).
    handoff_out / handoff_in: one ActHandoff shared by a fused conv+LeakyReLU and the single conv consuming its output."""
    return _ConvFn.apply(x, weight, bias, kind, cache, wdst, bdst, act, handoff_out, handoff_in)


def linear(x: Tensor, weight: Tensor, bias: Optional[Tensor] = None, cache: Optional[dict] = None, wdst=None, bdst=None) -> Tensor:
    """nn.Linear as a 1x1 conv on a 1x1 image (generator_submodules.py:36,152)."""
    B, Cin = x.shape
    y = _ConvFn.apply(x.reshape(B, Cin, 1, 1), weight.view(weight.shape[0], Cin, 1, 1), bias, "same", cache, wdst, bdst)
    return y.view(B, weight.shape[0])


# --------------------------------------------------------------------------------------------------------------
# BatchNorm (train mode) + activation
# --------------------------------------------------------------------------------------------------------------
# Batch groups: a discriminator update runs the real and the fake batch through the same weights.  The reference does two
# forward passes (disc_loss.py:55-61), so every BatchNorm sees each batch on its own: batch statistics per pass, two running-
# statistic updates in call order.  `with bn_groups(2)` lets one pass over the concatenated [real; fake] batch reproduce that:
# each BatchNorm call treats the batch axis as `groups` consecutive sub-batches (include/agan.h: agan_bn_train_fwd): statistics,
# running-stat update and normalisation per sub-batch in order; the backward sums gamma/beta gradients over the groups.  The
# convolutions in between simply see twice the pixels -- half the launches, better-filled tiles on the deep 4x4..16x16 layers.
_BN_GROUPS = [1]


@contextlib.contextmanager
def bn_groups(groups: int):
    old = _BN_GROUPS[0]
    _BN_GROUPS[0] = int(groups)
    try:
        yield
    finally:
        _BN_GROUPS[0] = old


class _BnActFn(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, residual, running_mean, running_var, nbt, training, act, eps, momentum, gdst, bdst, groups):
        x = _act(x, "bn input")
        shape = x.shape
        B, C = shape[0], shape[1]
        HW = x.numel() // (B * C)
        if not training:
            groups = 1
        if B % groups:
            raise L.AganError(f"BatchNorm: batch {B} does not split into {groups} groups")
        Bg = B // groups
        g, b = _dev(gamma.detach(), "bn weight"), _dev(beta.detach(), "bn bias")
        lib = L.load()
        co = C // 2 if act == L.ACT_GLU else C
        # storage types (include/agan.h: agan_bn_*_dt): the output takes the activation storage in force (4-D tensors only); a 16-bit
        # input is read as it stands when the output has the same type, else widened
        odt = get_activation_storage() if HW > 1 else L.DT_F32
        if x.dtype != torch.float32 and _dt(x) != odt:
            x = x.float()
        xdt = _dt(x)
        out = torch.empty((B, co) + tuple(shape[2:]), dtype=_TORCH_OF[odt], device=x.device)
        res = _act(residual, "bn residual") if residual is not None else None
        if res is not None and res.dtype != out.dtype:
            res = res.to(out.dtype)
        if training:
            mean = torch.empty((groups, C), dtype=torch.float32, device=x.device)
            invstd = torch.empty_like(mean)
            nbytes = lib.agan_bn_train_fwd_ws_bytes(Bg, C, HW)
            ws, wsp = _ws(nbytes, x)
            L.call("agan_bn_train_fwd_dt", _p(x), _p(g), _p(b), _p(res), _p(out), _p(mean), _p(invstd), _p(running_mean),
                   _p(running_var), _p(nbt), B, C, HW, float(eps), float(momentum), act, groups, wsp, nbytes, _stream(),
                   _p(_amax_out(out)), xdt, odt)
        else:
            mean = running_mean
            invstd = torch.rsqrt(running_var + eps)
            L.call("agan_bn_act_fwd_dt", _p(x), _p(mean), _p(invstd), _p(g), _p(b), _p(res), _p(out), B, C, HW, act, _stream(),
                   _p(_amax_out(out)), xdt, odt)
        ctx.save_for_backward(x, g, b, mean, invstd)
        ctx.odt = odt
        ctx.act, ctx.training, ctx.has_res, ctx.gdst, ctx.bdst, ctx.groups = act, training, residual is not None, gdst, bdst, groups
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        if not ctx.training:
            raise L.AganError("BatchNorm backward in eval mode is not on the training path")
        x, g, b, mean, invstd = ctx.saved_tensors
        dout = _act(dout, "bn grad")
        if _dt(dout) != ctx.odt:                     # (autograd delivers the output's dtype; a hand-made gradient may not)
            dout = dout.to(_TORCH_OF[ctx.odt])
        B, C = x.shape[0], x.shape[1]
        HW = x.numel() // (B * C)
        groups = ctx.groups
        Bg = B // groups
        dx = torch.empty_like(x)
        # frozen affine parameters (a discriminator inside the generator update): the sums are still needed for dx, but they go
        # to scratch, not into the owner's flat gradient buffer
        dgbuf, gacc, dg = _grad_out(ctx.gdst if ctx.needs_input_grad[1] else None, g.shape, x)
        dbbuf, bacc, db = _grad_out(ctx.bdst if ctx.needs_input_grad[2] else None, b.shape, x)
        if gacc != bacc:
            raise L.AganError("BatchNorm weight/bias gradient destinations out of step")
        nbytes = L.load().agan_bn_act_bwd_ws_bytes(Bg, C, HW)
        ws, wsp = _ws(nbytes, x)
        L.call("agan_bn_act_bwd_dt", _p(x), _p(dout), _p(mean), _p(invstd), _p(g), _p(b), _p(dx), _p(dgbuf), _p(dbbuf), B, C, HW,
               ctx.act, gacc, groups, wsp, nbytes, _stream(), _p(_amax_out(dx)), _dt(x), ctx.odt)
        dres = dout if ctx.has_res else None
        return dx, dg, db, dres, None, None, None, None, None, None, None, None, None, None


def bn_act(x, gamma, beta, running_mean, running_var, nbt, training: bool, act: int, residual=None,
           eps: float = 1e-5, momentum: float = 0.1) -> Tensor:
    """Train-mode BatchNorm + activation (+ residual).  Inside `with bn_groups(G)` the batch axis is G consecutive sub-batches,
    each normalised with its own statistics (see bn_groups above)."""
    return _BnActFn.apply(x, gamma, beta, residual, running_mean, running_var, nbt, training, act, eps, momentum,
                          grad_dst(gamma), grad_dst(beta), _BN_GROUPS[0])


class _ActFn(Function):
    @staticmethod
    def forward(ctx, x, act):
        x = _dev(x, "activation input")
        out = torch.empty_like(x)
        L.call("agan_act_fwd", _p(x), _p(out), x.numel(), act, _stream())
        ctx.save_for_backward(out)
        ctx.act = act
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        (out,) = ctx.saved_tensors
        dout = _dev(dout, "activation grad")
        dx = torch.empty_like(out)
        L.call("agan_act_bwd", _p(out), _p(dout), _p(dx), out.numel(), ctx.act, _stream())
        return dx, None


def activation(x: Tensor, act: int) -> Tensor:
    return _ActFn.apply(x, act)


class _GluFn(Function):
    @staticmethod
    def forward(ctx, x):
        x = _dev(x, "GLU input")
        B, C = x.shape[0], x.shape[1]
        if C % 2:
            raise AssertionError("channels dont divide 2!")      # utilities/layers.py:22
        HW = x.numel() // (B * C)
        out = torch.empty((B, C // 2) + tuple(x.shape[2:]), dtype=torch.float32, device=x.device)
        L.call("agan_glu_fwd", _p(x), _p(out), B, C, HW, _stream())
        ctx.save_for_backward(x)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        (x,) = ctx.saved_tensors
        dout = _dev(dout, "GLU grad")
        B, C = x.shape[0], x.shape[1]
        dx = torch.empty_like(x)
        L.call("agan_glu_bwd", _p(x), _p(dout), _p(dx), B, C, x.numel() // (B * C), _stream())
        return dx


def glu(x: Tensor) -> Tensor:
    return _GluFn.apply(x)


# --------------------------------------------------------------------------------------------------------------
# word-context attention (networks/attention.py:25-79)
# --------------------------------------------------------------------------------------------------------------
class _AttentionFn(Function):
    @staticmethod
    def forward(ctx, images, words, weight, mask, scale, wdst):
        # images / context / their gradients keep the activation storage in force (typed kernels: agan_attn_*_dt); words, the
        # projection and the attention map stay fp32
        images, words = _act(images, "attention images"), _dev(words, "attention words")
        if images.dtype != torch.float32 and _dt(images) != get_activation_storage():
            images = images.float()
        w = _dev(weight.detach(), "attention conv1 weight")
        B, C, H, W = images.shape
        Bw, E, T = words.shape
        if Bw != B or tuple(mask.shape) != (B, T):
            raise ValueError(f"attention: images {tuple(images.shape)}, words {tuple(words.shape)}, mask {tuple(mask.shape)}")
        m = mask.to(device=images.device, dtype=torch.int64).contiguous()
        proj = torch.empty((B, C, T), dtype=torch.float32, device=images.device)
        ctxt = torch.empty_like(images)
        attn = torch.empty((B, T, H, W), dtype=torch.float32, device=images.device)
        L.call("agan_attn_fwd_dt", _p(images), _p(words), _p(w), _p(m), float(scale), _p(proj), _p(ctxt), _p(attn),
               B, C, E, T, H * W, _stream(), _dt(images))
        ctx.save_for_backward(images, words, w, proj, attn)
        ctx.scale, ctx.wdst = float(scale), wdst
        ctx.set_materialize_grads(False)     # the attention map usually feeds nothing: no zero-filled 16 MB gradient for it
        return ctxt, attn

    @staticmethod
    @once_differentiable
    def backward(ctx, dctx, dattn):
        images, words, w, proj, attn = ctx.saved_tensors
        B, C, H, W = images.shape
        _, E, T = words.shape
        if dctx is not None:
            dctx = _act(dctx, "attention dctx")
            if dctx.dtype != images.dtype:
                dctx = dctx.to(images.dtype)
        dattn = _dev(dattn, "attention dattn") if dattn is not None else None
        dimages, dwords = torch.empty_like(images), torch.empty_like(words)
        # a frozen projection (requires_grad False) still needs d(proj) for nothing: its gradient goes to scratch, not into the
        # owner's flat buffer (as the conv and BatchNorm paths do)
        dwbuf, wacc, dw = _grad_out(ctx.wdst if ctx.needs_input_grad[2] else None, w.shape, images)
        if not ctx.needs_input_grad[2]:
            dw = None
        nbytes = L.load().agan_attn_bwd_ws_bytes(B, C, T, H * W)
        ws, wsp = _ws(nbytes, images)
        L.call("agan_attn_bwd_dt", _p(images), _p(words), _p(w), _p(proj), _p(attn), _p(dctx), _p(dattn), ctx.scale,
               _p(dimages), _p(dwords), _p(dwbuf), B, C, E, T, H * W, wacc, wsp, nbytes, _stream(), _dt(images))
        return dimages, dwords, dw, None, None, None


def attention(images: Tensor, words: Tensor, conv1_weight: Tensor, mask: Tensor, scaled: bool = True):
    C = images.shape[1]
    scale = 1.0 / math.sqrt(C) if scaled else 1.0
    return _AttentionFn.apply(images, words, conv1_weight, mask, scale, grad_dst(conv1_weight))


# --------------------------------------------------------------------------------------------------------------
# small heads
# --------------------------------------------------------------------------------------------------------------
class _ReparamFn(Function):
    @staticmethod
    def forward(ctx, mu, logvar, eps):
        mu, logvar, eps = _dev(mu, "mu"), _dev(logvar, "logvar"), _dev(eps, "eps")
        c = torch.empty_like(mu)
        L.call("agan_reparam_fwd", _p(mu), _p(logvar), _p(eps), _p(c), mu.numel(), _stream())
        ctx.save_for_backward(logvar, eps)
        return c

    @staticmethod
    @once_differentiable
    def backward(ctx, dc):
        logvar, eps = ctx.saved_tensors
        dc = _dev(dc, "dc")
        dmu, dlv = torch.empty_like(dc), torch.empty_like(dc)
        L.call("agan_reparam_bwd", _p(logvar), _p(eps), _p(dc), _p(dmu), _p(dlv), dc.numel(), _stream())
        return dmu, dlv, None


def reparametrize(mu, logvar, eps):
    return _ReparamFn.apply(mu, logvar, eps)


class _DiscLossFn(Function):
    @staticmethod
    def forward(ctx, p_real, p_fake):
        p_real, p_fake = _dev(p_real, "D(x)"), _dev(p_fake, "D(G(z))")
        loss = torch.empty((), dtype=torch.float32, device=p_real.device)
        dr, df = torch.empty_like(p_real), torch.empty_like(p_fake)
        L.call("agan_disc_loss", _p(p_real), _p(p_fake), _p(loss), _p(dr), _p(df), p_real.numel(), _stream())
        ctx.save_for_backward(dr, df)
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        dr, df = ctx.saved_tensors
        return dr * g, df * g


def ns_disc_loss(p_real, p_fake):
    return _DiscLossFn.apply(p_real, p_fake)


class _DiscLossPairedFn(Function):
    """the same loss on one [real; fake] score vector (the paired discriminator pass): no slice views, so autograd does not
    zero-fill and scatter two half gradients"""

    @staticmethod
    def forward(ctx, p):
        p = _dev(p, "D([x; G(z)])")
        n = p.numel() // 2
        loss = torch.empty((), dtype=torch.float32, device=p.device)
        dp = torch.empty_like(p)
        L.call("agan_disc_loss", _p(p[:n]), _p(p[n:]), _p(loss), _p(dp[:n]), _p(dp[n:]), n, _stream())
        ctx.save_for_backward(dp)
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (dp,) = ctx.saved_tensors
        return dp * g


def ns_disc_loss_paired(p: Tensor) -> Tensor:
    """NonSaturatingDiscLoss on a score vector whose first half is D(x) and second half D(G(z))"""
    if p.dim() != 1 or p.numel() % 2:
        raise L.AganError("ns_disc_loss_paired: expected a [2B] score vector")
    return _DiscLossPairedFn.apply(p)


class _GenLossFn(Function):
    @staticmethod
    def forward(ctx, p_fake):
        p_fake = _dev(p_fake, "D(G(z))")
        loss = torch.empty((), dtype=torch.float32, device=p_fake.device)
        df = torch.empty_like(p_fake)
        L.call("agan_gen_loss", _p(p_fake), _p(loss), _p(df), p_fake.numel(), _stream())
        ctx.save_for_backward(df)
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (df,) = ctx.saved_tensors
        return df * g


def ns_gen_loss(p_fake):
    return _GenLossFn.apply(p_fake)


class _KLFn(Function):
    @staticmethod
    def forward(ctx, mu, logvar):
        mu, logvar = _dev(mu, "mu"), _dev(logvar, "logvar")
        loss = torch.empty((), dtype=torch.float32, device=mu.device)
        dmu, dlv = torch.empty_like(mu), torch.empty_like(logvar)
        L.call("agan_kl_loss", _p(mu), _p(logvar), _p(loss), _p(dmu), _p(dlv), mu.numel(), _stream())
        ctx.save_for_backward(dmu, dlv)
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        dmu, dlv = ctx.saved_tensors
        return dmu * g, dlv * g


def kl_loss(mu, logvar):
    return _KLFn.apply(mu, logvar)


# --------------------------------------------------------------------------------------------------------------
# DAMSM losses (losses/words_loss.py:29-102, losses/sentence_loss.py:12-50)
# --------------------------------------------------------------------------------------------------------------
def _ids(class_ids, device) -> Optional[Tensor]:
    if class_ids is None:
        return None
    return torch.as_tensor(class_ids).to(device=device, dtype=torch.int64).contiguous()


def _labels(labels, batch: int, device) -> Optional[Tensor]:
    """CE targets for the kernels (words_loss.py:98-99, sentence_loss.py:46-47): an int64 device vector, or None for arange(batch)
    (the kernels' built-in default -- what train.py:104 / _make_match_labels builds, recognised by the `_agan_arange` tag = (batch, tensor version at tagging) so that
    the hot path neither copies nor reads the labels back)."""
    if labels is None:
        return None
    tag = getattr(labels, "_agan_arange", None)
    if tag is not None and tag == (batch, labels._version):      # still the untouched arange _make_match_labels built
        return None
    lab = torch.as_tensor(labels).reshape(-1)
    if lab.numel() != batch:
        raise ValueError(f"labels must have one entry per sample (got {lab.numel()} for a batch of {batch})")
    return lab.to(device=device, dtype=torch.int64).contiguous()


class _FuncAttentionFn(Function):
    @staticmethod
    def forward(ctx, query, context, gamma1, scale):
        q, c = _dev(query, "query"), _dev(context, "context")
        B, D, Lq = q.shape
        ih, iw = c.shape[2], c.shape[3]
        wctx = torch.empty((B, D, Lq), dtype=torch.float32, device=q.device)
        attn = torch.empty((B, Lq, ih, iw), dtype=torch.float32, device=q.device)
        L.call("agan_func_attention_fwd", _p(q), _p(c), gamma1, scale, _p(wctx), _p(attn), B, D, Lq, ih * iw, _stream())
        ctx.save_for_backward(q, c)
        ctx.hp = (gamma1, scale)
        ctx.set_materialize_grads(False)
        return wctx, attn

    @staticmethod
    @once_differentiable
    def backward(ctx, dwctx, dattn):
        q, c = ctx.saved_tensors
        B, D, Lq = q.shape
        S = c.shape[2] * c.shape[3]
        dq, dc = torch.empty_like(q), torch.empty_like(c)
        if dwctx is None and dattn is None:
            return dq.zero_(), dc.zero_(), None, None
        dwctx = _dev(dwctx, "func_attention d(weightedContext)") if dwctx is not None else None
        dattn = _dev(dattn, "func_attention d(attn)") if dattn is not None else None
        L.call("agan_func_attention_bwd", _p(q), _p(c), _p(dwctx), _p(dattn), *ctx.hp, _p(dq), _p(dc), B, D, Lq, S, _stream())
        return dq, dc, None, None


def func_attention(query: Tensor, context: Tensor, gamma1: float = 4.0, scaled: bool = True):
    """The parameter-free DAMSM attention (attention.py:82-120), differentiable like the reference's plain-autograd version.
    Inside the training step it runs fused into the words-loss kernels; this is the standalone entry point."""
    D = query.shape[1]
    return _FuncAttentionFn.apply(query, context, float(gamma1), (1.0 / math.sqrt(D)) if scaled else 1.0)


class _WordsLossFn(Function):
    @staticmethod
    def forward(ctx, feat, wemb, lens, cids, labels, g1, g2, g3, lam):
        feat, wemb = _dev(feat, "img_features"), _dev(wemb, "words_emb")
        B, D = feat.shape[0], feat.shape[1]
        S = feat.numel() // (B * D)
        T = wemb.shape[2]
        loss = torch.empty((), dtype=torch.float32, device=feat.device)
        sim = torch.empty((B, B), dtype=torch.float32, device=feat.device)
        maps = torch.zeros((B, T, S), dtype=torch.float32, device=feat.device)
        save = torch.empty(L.load().agan_words_loss_save_elems(B, D, T, S), dtype=torch.float32, device=feat.device)
        L.call("agan_words_loss_fwd", _p(feat), _p(wemb), _p(lens), _p(cids), _p(labels), g1, g2, g3, lam, _p(loss), _p(sim), _p(maps),
               _p(save), B, D, T, S, _stream())
        ctx.save_for_backward(feat, wemb, lens, save)
        ctx.hp = (g1, g2, g3, lam)
        ctx.mark_non_differentiable(maps, sim)
        return loss, maps, sim

    @staticmethod
    @once_differentiable
    def backward(ctx, dloss, _dmaps, _dsim):
        feat, wemb, lens, save = ctx.saved_tensors
        B, D = feat.shape[0], feat.shape[1]
        S = feat.numel() // (B * D)
        T = wemb.shape[2]
        dfeat, dwemb = torch.empty_like(feat), torch.empty_like(wemb)
        dl = dloss.to(torch.float32).contiguous()
        nbytes = L.load().agan_words_loss_bwd_ws_bytes(B, D, T, S)
        ws, wsp = _ws(nbytes, feat)
        L.call("agan_words_loss_bwd", _p(feat), _p(wemb), _p(lens), _p(save), _p(dl), *ctx.hp, _p(dfeat), _p(dwemb),
               B, D, T, S, wsp, nbytes, _stream())
        return dfeat, dwemb, None, None, None, None, None, None, None


def words_loss(feat, wemb, cap_lens, class_ids, gamma1, gamma2, gamma3, wlambda, labels=None):
    if isinstance(cap_lens, Tensor) and cap_lens.is_cuda and cap_lens.dtype == torch.int64:
        lens = cap_lens.contiguous()                       # already resident: no host round trip (HIP-graph capturable)
    else:
        lens = torch.as_tensor(cap_lens).to(device=feat.device, dtype=torch.int64).contiguous()
    return _WordsLossFn.apply(feat, wemb, lens, _ids(class_ids, feat.device), _labels(labels, feat.shape[0], feat.device),
                              float(gamma1), float(gamma2), float(gamma3), float(wlambda))


class _SentLossFn(Function):
    @staticmethod
    def forward(ctx, cnn, rnn, cids, labels, g3, lam, eps):
        cnn, rnn = _dev(cnn, "cnn_code"), _dev(rnn, "rnn_code")
        B, D = cnn.shape
        loss = torch.empty((), dtype=torch.float32, device=cnn.device)
        save = torch.empty(2 * B * B + 2 * B, dtype=torch.float32, device=cnn.device)
        L.call("agan_sent_loss_fwd", _p(cnn), _p(rnn), _p(cids), _p(labels), g3, lam, eps, _p(loss), _p(save), B, D, _stream())
        ctx.save_for_backward(cnn, rnn, save)
        ctx.hp = (g3, lam, eps)
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, dloss):
        cnn, rnn, save = ctx.saved_tensors
        B, D = cnn.shape
        dc, dr = torch.empty_like(cnn), torch.empty_like(rnn)
        dl = dloss.to(torch.float32).contiguous()
        L.call("agan_sent_loss_bwd", _p(cnn), _p(rnn), _p(save), _p(dl), *ctx.hp, _p(dc), _p(dr), B, D, _stream())
        return dc, dr, None, None, None, None, None


def sentence_loss(cnn_code, rnn_code, class_ids, gamma3, slambda, eps=1e-8, labels=None):
    return _SentLossFn.apply(cnn_code, rnn_code, _ids(class_ids, cnn_code.device), _labels(labels, cnn_code.shape[0], cnn_code.device),
                             float(gamma3), float(slambda), float(eps))


# --------------------------------------------------------------------------------------------------------------
# fused Adam on a flat buffer
# --------------------------------------------------------------------------------------------------------------
def adam_step_(param: Tensor, grad: Tensor, exp_avg: Tensor, exp_avg_sq: Tensor, step_state: Tensor, lr: float, beta1: float,
               beta2: float, eps: float, grad_scale: float = 1.0) -> None:
    """One fused Adam step over a flat buffer.  `step_state` is the optimiser's 4-element int32 device tensor: the kernel
    advances the step count in it and derives the bias corrections on the device (graph-replay safe)."""
    for t in (param, grad, exp_avg, exp_avg_sq):
        if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
            raise L.AganError("adam_step_: flat contiguous float32 device buffers required")
    if not step_state.is_cuda or step_state.dtype != torch.int32 or step_state.numel() < 4:
        raise L.AganError("adam_step_: step_state must be an int32 device tensor of 4 elements")
    L.call("agan_adam_step", _p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), param.numel(), _p(step_state), float(lr),
           float(beta1), float(beta2), float(eps), float(grad_scale), _stream())
    rng = _flat_range(param.data_ptr())
    if rng is not None:
        rng[2] += 1
    else:
        bump_weight_epoch()
