"""Per-workgroup phase times of the row-block gather (csrc/conv_p16.hip built with -DAGAN_P16_TIMING: profiles/build_variant.sh):
    AGAN_LIB=attention-gan_amd/csrc/variants/libagan_timing.so [AGAN_P16_TILE=t] python profiles/p16_timing.py [layer ...]
s_memtime stamps (shader cycles) at kernel entry / after the prologue barrier / after the K loop / at exit, per workgroup."""
import ctypes
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "profiles"))
HF = importlib.import_module("attention-gan_amd.backend.functional")
L = importlib.import_module("attention-gan_amd.backend.lib")
from conv_micro import LAYERS  # noqa: E402


def main():
    names = sys.argv[1:] or ["d_down_64_128", "d_down_128_256", "d_down_256_512", "g_same_64_128"]
    HF.set_precision(L.PRECISIONS["bf16"])
    HF.set_activation_storage("bf16")
    lib = L.load()
    lib.agan_debug_p16_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
    dev = "cuda"
    for name in names:
        kind, B, Cin, H, Cout, k = LAYERS[name]
        g = torch.Generator().manual_seed(1)
        x = torch.randn(B, Cin, H, H, generator=g).to(dev).to(torch.bfloat16)
        w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).to(dev)
        cache = {}
        for phase in ("fwd", "dgrad"):
            xr = x.clone().requires_grad_(phase == "dgrad")
            y = HF.conv2d(xr, w, None, kind, cache)
            gy = torch.randn(y.shape, device=dev).to(y.dtype)

            def run():
                if phase == "fwd":
                    HF.conv2d(xr, w, None, kind, cache)
                else:
                    torch.autograd.grad(y, xr, gy, retain_graph=True)
            n = 16384
            buf = np.zeros((n, 4), dtype=np.uint64)
            for _ in range(5):
                run()
            torch.cuda.synchronize()
            assert lib.agan_debug_p16_stamps(buf.ctypes.data_as(ctypes.c_void_p), n) == 0        # (reading clears the stamps)
            run()
            torch.cuda.synchronize()
            assert lib.agan_debug_p16_stamps(buf.ctypes.data_as(ctypes.c_void_p), n) == 0
            t = buf.astype(np.int64)
            t = t[(t[:, 3] > t[:, 0]) & (t[:, 0] > 0)]
            pro, loop, epi = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]
            total = t[:, 3].max() - t[:, 0].min()
            conc = float((t[:, 3] - t[:, 0]).sum()) / float(total) / 256.0
            mfma = 2.0 * B * (H // 2 if kind == "down" else (2 * H if kind == "up" else H)) ** 2 * Cout * Cin * (k * k if kind != "up" else 4) / 32768.0 / 1024.0 * 32
            print(f"{name:18s} {phase:5s} WGs {len(t):5d} kernel {total:8d} cyc (MFMA-bound {mfma:7.0f}) | per WG: prologue {np.median(pro):7.0f}  K loop {np.median(loop):7.0f}  "
                  f"epilogue {np.median(epi):6.0f}  WGs/CU {conc:4.2f}  life {np.median(t[:, 3] - t[:, 0]):7.0f} (p90 {np.percentile(t[:, 3] - t[:, 0], 90):7.0f})")


if __name__ == "__main__":
    main()
