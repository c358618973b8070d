"""Micro-benchmark of agan_pack_weight (OIHW fp32 -> packed 16-bit operand layout) on the largest weight tensors of the metric config.

    python profiles/pack_micro.py [--precision bf16|f16x3|bf16x6]

Prints per tensor and pack mode: microseconds (torch events, best of 3 bursts of 10), GB/s over (fp32 bytes read + packed bytes written)."""
import argparse
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
HF = importlib.import_module("attention-gan_amd.backend.functional")
L = importlib.import_module("attention-gan_amd.backend.lib")

TENSORS = [  # (cout, cin, k, [pack modes])
    (2048, 1024, 4, ("fwd", "dgrad4x4")),
    (1024, 512, 4, ("fwd", "dgrad4x4")),
    (1024, 2048, 3, ("fwd", "dgrad_s1")),
    (512, 256, 4, ("fwd", "dgrad4x4")),
    (128, 64, 3, ("fwd", "dgrad_s1")),
]
MODES = {"fwd": L.PACK_FWD, "dgrad4x4": L.PACK_DGRAD_4x4S2, "dgrad_s1": L.PACK_DGRAD_S1}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default="bf16")
    a = ap.parse_args()
    prec = L.PRECISIONS[a.precision]
    lib = L.load()
    print(f"{'tensor':22s} {'mode':9s} {'us':>8s} {'GB/s':>8s}")
    for cout, cin, k, modes in TENSORS:
        w = torch.randn(cout, cin, k, k, device="cuda")
        for m in modes:
            n = lib.agan_packed_weight_bytes(MODES[m], cout, cin, k, k, prec)
            wk = torch.empty(n, dtype=torch.uint8, device="cuda")
            best = 1e9
            for burst in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    L.call("agan_pack_weight", HF._p(w), HF._p(wk), MODES[m], cout, cin, k, k, prec, HF._stream())
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) * 100.0)
            print(f"{cout}x{cin}x{k}x{k:<10d} {m:9s} {best:8.1f} {(w.numel() * 4 + n) / best / 1e3:8.0f}")


if __name__ == "__main__":
    main()
