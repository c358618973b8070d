"""Shared test helpers: golden-fixture loading, the closed-form gradient probe, tolerances."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# north_star: outputs match the reference CPU forward/backward within 1e-3 relative.
RTOL = 1e-3


def probe(shape, phase):
    """Same closed form as tests/golden/make_golden.py::probe."""
    n = int(np.prod(shape))
    return torch.cos(torch.arange(n, dtype=torch.float64) * 0.37 + phase).float().view(*shape)


def load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


def sub(d, prefix, as_torch=True, clone=True):
    out = {}
    for k, v in d.items():
        if k.startswith(prefix):
            t = torch.from_numpy(np.array(v)) if as_torch else v
            out[k[len(prefix):]] = t
    return out


def T(a):
    return torch.from_numpy(np.array(a))


def rel_err(got, want):
    """max |got-want| / max(|want|) -- the 'relative fp tolerance' used throughout (scale of the tensor)."""
    got = torch.as_tensor(got).detach().double().cpu()
    want = torch.as_tensor(want).detach().double().cpu()
    scale = want.abs().max().clamp(min=1e-30)
    return float((got - want).abs().max() / scale)


def assert_close(got, want, tol=RTOL, what=""):
    e = rel_err(got, want)
    assert e <= tol, f"{what}: rel err {e:.3e} > {tol:.1e}"


def free_port() -> int:
    """a TCP port nobody listens on right now (rendezvous of the multi-process tests; a pid-derived number can collide)"""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]
