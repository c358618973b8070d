"""One-rank `nccl` rehearsal timings of the eager step under variations (AGAN_DP_FORCE=1 python profiles/dp_rehearsal.py)."""
import importlib
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29545")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
os.environ.setdefault("TORCH_NCCL_HIGH_PRIORITY", "1")
torch.cuda.set_device(0)
if os.environ.get("AGAN_DP_FORCE") == "1" or os.environ.get("AGAN_PG_ONLY") == "1":
    dist.init_process_group(os.environ.get("AGAN_PG_BACKEND", "nccl"), **({"device_id": torch.device("cuda", 0)} if os.environ.get("AGAN_PG_BACKEND", "nccl") == "nccl" else {}))
if os.environ.get("AGAN_EXTRA_STREAMS"):
    _extra = [torch.cuda.Stream(priority=-1) for _ in range(int(os.environ["AGAN_EXTRA_STREAMS"]))]
    for _s in _extra:
        with torch.cuda.stream(_s):
            torch.zeros(1, device="cuda")
HF = importlib.import_module("attention-gan_amd.backend.functional")
dev = torch.device("cuda", 0)
step = bench.build(dev, 24, HF)
words, sent, lens, reals = bench.synthetic_batch(dev, 24, seed=1234)
lens_dev = torch.tensor(lens, dtype=torch.int64, device=dev)


import collections
ACC = collections.Counter()
DP = importlib.import_module("attention-gan_amd.dataparallel")
OPT = importlib.import_module("attention-gan_amd.optim")


def timed(cls, name):
    orig = getattr(cls, name)

    def wrap(*a, **k):
        t0 = time.perf_counter()
        try:
            return orig(*a, **k)
        finally:
            ACC[f"{cls.__name__}.{name}"] += time.perf_counter() - t0
            ACC[f"{cls.__name__}.{name} calls"] += 1
    setattr(cls, name, wrap)


for nm in ("_launch", "finish", "arm"):
    timed(DP.GradBuckets, nm)
for nm in ("step", "zero_grad", "_rebind"):
    timed(OPT.FlatAdam, nm)
timed(torch.Tensor, "backward")


def rate(tag, n=30):
    for _ in range(4):
        step.step(words, sent, lens_dev, None, reals)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step.step(words, sent, lens_dev, None, reals)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    print(f"{tag:40s} {t / n * 1e3:7.2f} ms/step (host enqueue {t_host / n * 1e3:6.2f})", flush=True)
    print("   per step ms:", {k: round(v / (n + 4) * 1e3, 3) if not k.endswith("calls") else v // (n + 4) for k, v in sorted(ACC.items())}, flush=True)
    ACC.clear()


rate("default (largest D first)")
print("rebind calls / copies since start:", [(o.rebind_calls, getattr(o, "rebind_copies", 0)) for o in [step.g_opt] + step.d_opts], flush=True)
if os.environ.get("AGAN_LONG"):
    step.d_order = [0, 1, 2]
    rate("D64, D128, D256 order")
    step.d_order = [2, 1, 0]
    step.overlap_discriminators = False
    rate("one stream")
    step.overlap_discriminators = True
    for bk in [step.g_buckets] + step.d_buckets:
        bk.active_saved, bk.active = bk.active, False
    rate("exchange switched off (hooks still registered)")
