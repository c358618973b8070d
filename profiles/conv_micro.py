"""Micro-benchmark of single conv layers through the C ABI (fwd / dgrad / wgrad), any arithmetic mode and activation storage.

    python profiles/conv_micro.py --precision bf16 --storage bf16 [--layers d256] [--iters 20]

Prints per layer and phase: microseconds (torch events on the launch stream, best of 3 bursts), algorithmic TFLOP/s, algorithmic GB/s.
The burst goes through autograd + ctypes: below ~85 us per call it measures the HOST (an empty kernel reads 83 us) -- for the 16-bit
layers and for ablations take the kernel durations of `rocprofv3 --kernel-trace -- python3 profiles/conv_micro.py ... --iters 10` instead
(profiles/r03_rows_ablation.txt was made that way).  Under `rocprofv3 --pmc ...` use --iters 3."""
import argparse
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
HF = importlib.import_module("attention-gan_amd.backend.functional")
L = importlib.import_module("attention-gan_amd.backend.lib")

LAYERS = {
    # name: (kind, B, Cin, H, Cout, k)
    "d_down_64_128": ("down", 48, 64, 128, 128, 4),
    "d_down_128_256": ("down", 48, 128, 64, 256, 4),
    "d_down_256_512": ("down", 48, 256, 32, 512, 4),
    "d_down_512_1024": ("down", 48, 512, 16, 1024, 4),
    "d_down_1024_2048": ("down", 48, 1024, 8, 2048, 4),
    "g_same_64_128": ("same", 24, 64, 128, 128, 3),
    "g_same_64_64": ("same", 24, 64, 128, 64, 3),
    "g_up_64_64": ("up", 24, 64, 128, 64, 3),
    "syn_same_256_128": ("same", 24, 256, 64, 128, 3),      # synthetic: a long K loop (32 Winograd chunks) on 64^2 maps
    "rgb_head_256": ("same", 24, 32, 256, 3, 3),      # GET_IMAGE_G of stage 3 (<= 4 output channels: conv_small.hip)
    "rgb_head_128": ("same", 24, 32, 128, 3, 3),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--storage", default="f32")
    ap.add_argument("--layers", default="all")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--phases", default="fwd,dgrad,wgrad")
    a = ap.parse_args()
    HF.set_precision(L.PRECISIONS[a.precision])
    HF.set_activation_storage(None if a.storage == "f32" else a.storage)
    tdt = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}[a.storage]
    names = list(LAYERS) if a.layers == "all" else [n for n in LAYERS if any(n.startswith(p) for p in a.layers.split(","))]
    dev = "cuda"
    print(f"{'layer':18s} {'phase':6s} {'us':>8s} {'TF/s':>8s} {'GB/s':>8s}")
    for name in names:
        kind, B, Cin, H, Cout, k = LAYERS[name]
        g = torch.Generator().manual_seed(1)
        x = torch.randn(B, Cin, H, H, generator=g).to(dev).to(tdt)
        w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).to(dev)
        cache = {}
        OH = H // 2 if kind == "down" else (2 * H if kind == "up" else H)
        flops = 2.0 * B * OH * OH * Cout * Cin * (k * k if kind != "up" else 9)
        esz = x.element_size()
        nbytes = esz * (x.numel() + B * Cout * OH * OH)
        for phase in a.phases.split(","):
            xr = x.clone().requires_grad_(phase == "dgrad")
            wr = w.clone().requires_grad_(phase == "wgrad")
            y = HF.conv2d(xr, wr, None, kind, cache)
            gy = torch.randn(y.shape, generator=torch.Generator(device=dev).manual_seed(2), device=dev).to(y.dtype)

            def run():
                if phase == "fwd":
                    HF.conv2d(xr, wr, None, kind, cache)
                else:
                    torch.autograd.grad(y, xr if phase == "dgrad" else wr, gy, retain_graph=True)
            for _ in range(3):
                run()
            best = 1e9
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(a.iters):
                    run()
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / a.iters * 1e3)
            print(f"{name:18s} {phase:6s} {best:8.1f} {flops / best / 1e6:8.1f} {nbytes / best / 1e3:8.1f}", flush=True)


if __name__ == "__main__":
    main()
