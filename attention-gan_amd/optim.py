"""Flat-buffer Adam: torch.optim.Adam(lr, betas=(0.5, 0.999)) semantics (train.py:78-80) as ONE fused HIP launch.

All trainable parameters of a module are re-homed as views into a single contiguous fp32 buffer (and their .grad
into a second one).  One optimiser step is then one kernel over the flat buffer, and the data-parallel gradient
exchange (dataparallel.py) all-reduces large contiguous slices instead of hundreds of small tensors -- sized for
288 GB HBM3E and point-to-point xGMI links, not for per-tensor NCCL calls.
"""
from __future__ import annotations

from typing import Dict, Iterable, List

import torch
from torch import Tensor, nn

from .backend import functional as HF


class FlatAdam:
    def __init__(self, params: Iterable[nn.Parameter], lr: float = 2e-4, betas=(0.5, 0.999), eps: float = 1e-8):
        self.params: List[nn.Parameter] = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("FlatAdam: no trainable parameters")
        dev, dt = self.params[0].device, self.params[0].dtype
        if dt != torch.float32:
            raise ValueError("FlatAdam: float32 parameters only")
        self.lr, self.betas, self.eps = lr, tuple(betas), eps
        self.offsets, total = [], 0
        for p in self.params:
            self.offsets.append(total)
            total += (p.numel() + 3) // 4 * 4           # keep every parameter 16-byte aligned inside the buffer
        self.numel = total
        self.flat = torch.zeros(total, dtype=dt, device=dev)
        self.grad = torch.zeros(total, dtype=dt, device=dev)
        self.exp_avg = torch.zeros(total, dtype=dt, device=dev)
        self.exp_avg_sq = torch.zeros(total, dtype=dt, device=dev)
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                self.flat[o:o + p.numel()].copy_(p.detach().reshape(-1))
                p.data = self.flat[o:o + p.numel()].view(p.shape)
                p.grad = self.grad[o:o + p.numel()].view(p.shape)
                p._agan_grad_dst = HF.GradDst(self.grad, o, p.numel())   # backward kernels write here directly
        if self.flat.is_cuda:
            HF.register_flat(self.flat)        # packed-weight caches follow THIS buffer's Adam steps only
        self.step_count = 0
        # device-resident step counter + bias-correction coefficients (see agan_adam_step): advanced by the kernel itself, so a
        # captured HIP graph keeps counting on replay; step_count mirrors it on the host for state_dict()
        self.step_state = torch.zeros(4, dtype=torch.int32, device=dev)

    def __del__(self):
        try:
            HF.unregister_flat(self.flat)
        except Exception:      # interpreter shutdown
            pass

    # -- torch.optim-like surface ---------------------------------------------------------------------------
    def zero_grad(self, set_to_none: bool = False) -> None:
        """Start a new backward: drop the per-parameter .grad references and re-arm the first-writer flags.  The flat gradient
        buffer itself is NOT filled with zeros here (376 MB per step at the metric config): the first backward kernel that
        produces a parameter's gradient overwrites its slice, and step() zeroes the slice of any parameter that received no
        gradient at all in this backward (torch.optim.Adam skips such a parameter; a zero gradient keeps the fused update
        branch-free and only decays its moments)."""
        for p in self.params:
            p.grad = None
            p._agan_grad_dst.reset()

    def _rebind(self, indices=None) -> int:
        """Make p.grad the view into the flat buffer (for all parameters, or the given indices); gradients that landed
        elsewhere (a kernel without a flat destination, or autograd cloning instead of adopting) are copied in.
        Returns how many copies were needed."""
        copies = 0
        base = self.grad.data_ptr()
        self.rebind_calls = getattr(self, "rebind_calls", 0) + 1
        it = zip(self.params, self.offsets) if indices is None else ((self.params[i], self.offsets[i]) for i in indices)
        for p, o in it:
            g = p.grad
            if g is not None and g.data_ptr() == base + 4 * o:
                continue
            view = self.grad[o:o + p.numel()].view(p.shape)
            if g is not None:
                dst = p._agan_grad_dst
                if not dst.written:
                    view.copy_(g)             # no kernel wrote the slice: a stock autograd gradient
                    copies += 1
                elif HF.wgrad_side_stream_enabled():
                    # the kernel wrote the slice on a side stream autograd knows nothing about: its copy may have been taken too early
                    raise RuntimeError("FlatAdam: autograd copied a gradient that was written on the weight-gradient side stream")
                elif dst.edges <= 1:
                    # ONE kernel contribution, and autograd holds a tensor of its own: either a clone of that contribution (it clones
                    # instead of adopting the view when something else still references it) or its sum with stock autograd edges
                    # of the same parameter (a tied weight, an op without a flat destination).  In both cases autograd's tensor is
                    # the complete gradient.
                    view.copy_(g)
                    copies += 1
                else:
                    # Several kernel contributions accumulated in the slice.  A clone autograd took once every edge had arrived
                    # equals the slice and is dropped; anything else is a stock autograd edge summed with only PART of the kernel
                    # contributions, which cannot be reconstructed -- refuse rather than train on a wrong gradient.
                    if torch.cuda.is_available() and p.is_cuda and torch.cuda.is_current_stream_capturing():
                        raise RuntimeError("FlatAdam: autograd replaced a gradient with several in-kernel contributions during graph capture")
                    if g.shape != view.shape or not torch.equal(g, view):
                        raise RuntimeError("FlatAdam: a parameter received several in-kernel gradient contributions AND a stock autograd "
                                           "gradient in one backward (tied weight?): unsupported, the sum cannot be recovered")
            elif not p._agan_grad_dst.written:
                view.zero_()                  # no gradient reached this parameter in this backward
            p.grad = view
        self.rebind_copies = getattr(self, "rebind_copies", 0) + copies
        return copies

    def join_and_rebind(self) -> None:
        """Make the flat gradient buffer complete and current on this stream (what step() does first)."""
        HF.join_side_stream()          # weight gradients forked off the current stream (functional.set_wgrad_side_stream)
        self._rebind()

    def named_gradients(self, module) -> Dict[str, Tensor]:
        """{parameter name: view of its slice of the flat gradient buffer} for the trainable parameters of `module`"""
        names = [k for k, p in module.named_parameters() if p.requires_grad]
        return {k: self.grad[o:o + p.numel()].view(p.shape) for k, p, o in zip(names, self.params, self.offsets)}

    def step(self, grad_scale: float = 1.0) -> None:
        self.join_and_rebind()
        self.step_count += 1
        if self.flat.is_cuda:
            HF.adam_step_(self.flat, self.grad, self.exp_avg, self.exp_avg_sq, self.step_state, self.lr,
                          self.betas[0], self.betas[1], self.eps, grad_scale)
        else:
            raise RuntimeError("FlatAdam.step: parameters are not on an MI355X (no CPU fallback)")

    # -- checkpoint interchange with torch.optim.Adam.state_dict() (reference _save_weights, trainer.py:109-115) --
    def sync_step_count(self) -> int:
        """The device counter is the truth (a replayed HIP graph advances only it): mirror it on the host.  One host sync."""
        if self.step_state.is_cuda:
            self.step_count = int(self.step_state[0].item())
        return self.step_count

    def state_dict(self) -> Dict:
        self.sync_step_count()
        state = {}
        for i, (p, o) in enumerate(zip(self.params, self.offsets)):
            n = p.numel()
            state[i] = {"step": torch.tensor(float(self.step_count)),
                        "exp_avg": self.exp_avg[o:o + n].view(p.shape).clone(),
                        "exp_avg_sq": self.exp_avg_sq[o:o + n].view(p.shape).clone()}
        group = {"lr": self.lr, "betas": self.betas, "eps": self.eps, "weight_decay": 0, "amsgrad": False,
                 "params": list(range(len(self.params)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd: Dict) -> None:
        g = sd["param_groups"][0]
        self.lr, self.betas, self.eps = g["lr"], tuple(g["betas"]), g["eps"]
        with torch.no_grad():
            for i, (p, o) in enumerate(zip(self.params, self.offsets)):
                st = sd["state"].get(i)
                if st is None:
                    continue
                n = p.numel()
                self.exp_avg[o:o + n].copy_(st["exp_avg"].reshape(-1))
                self.exp_avg_sq[o:o + n].copy_(st["exp_avg_sq"].reshape(-1))
                self.step_count = int(float(st["step"]))
        with torch.no_grad():
            self.step_state.zero_()
            self.step_state[0] = self.step_count
