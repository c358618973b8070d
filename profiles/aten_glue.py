"""Which ATen (non-library) launches does one train step make, and from where?  

    python profiles/aten_glue.py [out.txt]

Runs the metric-config step eagerly under torch.profiler (with Python stacks) and lists every device kernel / memcpy that is NOT one
of libagan_hip.so's, grouped by (op, innermost repo frame), with launches per step and device microseconds per step."""
import collections
import importlib
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

HF = importlib.import_module("attention-gan_amd.backend.functional")
LIB = importlib.import_module("attention-gan_amd.backend.lib")
if os.environ.get("AGAN_GLUE_MODE"):          # e.g. AGAN_GLUE_MODE=bf16 [AGAN_GLUE_STORAGE=bf16]
    HF.set_precision(LIB.PRECISIONS[os.environ["AGAN_GLUE_MODE"]])
    HF.set_activation_storage(os.environ.get("AGAN_GLUE_STORAGE"))
dev = torch.device("cuda", 0)
step = bench.build(dev, 24, HF)
words, sent, lens, reals = bench.synthetic_batch(dev, 24, seed=1234)
lens_dev = torch.tensor(lens, dtype=torch.int64, device=dev)
for _ in range(3):
    step.step(words, sent, lens_dev, None, reals)
torch.cuda.synchronize()
STEPS = 2
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    for _ in range(STEPS):
        step.step(words, sent, lens_dev, None, reals)
    torch.cuda.synchronize()

by = collections.defaultdict(lambda: [0, 0.0, set()])
for ev in prof.events():
    if not ev.name.startswith("aten::") or ev.device_time_total <= 0 or ev.cpu_children and any(c.name.startswith("aten::") and c.device_time_total > 0 for c in ev.cpu_children):
        continue
    frame = "?"
    for fr in ev.stack or []:
        if "/attention-gan_amd/" in fr or "bench.py" in fr:
            frame = fr.split("/root/repo/")[-1] if "/root/repo/" in fr else fr
            frame = frame[-110:]
            break
    key = (ev.name, frame) if ev.name != "aten::copy_" else (ev.name, str(ev.input_shapes)[:60])
    by[key][0] += 1
    by[key][1] += ev.device_time_total
    by[key][2].add(str(ev.input_shapes)[:80])
rows = sorted(by.items(), key=lambda kv: -kv[1][1])
out = [f"{'op':28s} {'n/step':>6s} {'us/step':>8s}  where (innermost repo frame) | shapes"]
tot_n = tot_us = 0
for (name, frame), (n, us, shapes) in rows:
    out.append(f"{name:28s} {n / STEPS:6.1f} {us / STEPS:8.1f}  {frame} | {sorted(shapes)[:2]}")
    tot_n += n
    tot_us += us
out.append(f"{'total':28s} {tot_n / STEPS:6.1f} {tot_us / STEPS:8.1f}")
text = "\n".join(out)
print(text)
if len(sys.argv) > 1:
    with open(sys.argv[1], "w") as f:
        f.write(text + "\n")
