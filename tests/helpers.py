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


def collect_from_children(q, procs, n=None, timeout=300.0, poll=0.5):
    """`n` results (default: one per process) from the queue the child processes report into -- WATCHING the children: a child that
    died without reporting (a crash in native code, an assertion in the worker) fails the test within `poll` seconds with its exit
    code, instead of after the whole queue timeout (a dead child used to cost five minutes of the GPU-test budget)."""
    import queue as _queue
    import time
    want = len(procs) if n is None else n
    got, t0 = [], time.monotonic()
    while len(got) < want:
        try:
            got.append(q.get(timeout=poll))
            continue
        except _queue.Empty:
            pass
        dead = [p for p in procs if p.exitcode not in (None, 0)]
        if dead:
            for p in procs:
                if p.is_alive():
                    p.terminate()
            raise AssertionError(f"child process died with exit code {dead[0].exitcode} before reporting ({len(got)} of {want} results in)")
        if all(p.exitcode == 0 for p in procs) and q.empty():
            raise AssertionError(f"every child exited cleanly but only {len(got)} of {want} results arrived")
        if time.monotonic() - t0 > timeout:
            for p in procs:
                if p.is_alive():
                    p.terminate()
            raise AssertionError(f"timed out after {timeout:.0f} s waiting for the child processes ({len(got)} of {want} results in)")
    return got


class LeakyMaskRecorder:
    """Records, in call order, the branch every LeakyReLU of the HIP path took (sign of its output), per discriminator call.

    `with LeakyMaskRecorder(discriminators) as rec:` ... run the HIP step ...; `rec.calls` is then a list of
    (discriminator index, batch size, [bool mask per LeakyReLU layer, CPU]) in the order the discriminators were called.
    Feed them to the oracle through oracle.LEAKY_MASKS (see oracle.leaky) so that both sides differentiate the same branch."""

    def __init__(self, discriminators):
        import importlib
        self.LAY = importlib.import_module("attention-gan_amd.utilities.layers")
        self.L = importlib.import_module("attention-gan_amd.backend.lib")
        self.Ds = list(discriminators)
        self.calls = []
        self._cur = None

    def __enter__(self):
        LAY, L = self.LAY, self.L
        self._conv, self._fused = LAY.HipConv2d.forward, LAY._BNState.fused
        self._fwd = [type(d).forward for d in self.Ds]
        rec = self

        def conv_fwd(mod, x, act=L.ACT_NONE, *a, **k):
            y = rec._conv(mod, x, act, *a, **k)
            if act == L.ACT_LRELU and rec._cur is not None:
                rec._cur.append((y.detach() >= 0).cpu())
            return y

        def bn_fused(mod, x, act, residual=None):
            y = rec._fused(mod, x, act, residual)
            if act == L.ACT_LRELU and rec._cur is not None:
                rec._cur.append((y.detach() >= 0).cpu())
            return y

        LAY.HipConv2d.forward, LAY._BNState.fused = conv_fwd, bn_fused
        self._patched = []
        for i, d in enumerate(self.Ds):
            orig = d.forward

            def fwd(X, _orig=orig, _i=i):
                rec._cur = []
                out = _orig(X)
                rec.calls.append((_i, int(X.shape[0]), rec._cur))
                rec._cur = None
                return out
            d.forward = fwd
            self._patched.append(d)
        return self

    def __exit__(self, *exc):
        self.LAY.HipConv2d.forward, self.LAY._BNState.fused = self._conv, self._fused
        for d in self._patched:
            del d.forward
        return False
