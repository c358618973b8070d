"""cProfile of the eager train step's HOST side (where does the Python enqueue time go?): python profiles/host_profile.py [steps]"""
import cProfile
import importlib
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

if os.environ.get("AGAN_DP_FORCE") == "1":
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
    os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
HF = importlib.import_module("attention-gan_amd.backend.functional")
dev = torch.device("cuda", 0)
step = bench.build(dev, 24, HF)
words, sent, lens, reals = bench.synthetic_batch(dev, 24, seed=1234)
lens_dev = torch.tensor(lens, dtype=torch.int64, device=dev)
for _ in range(3):
    step.step(words, sent, lens_dev, None, reals)
torch.cuda.synchronize()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    step.step(words, sent, lens_dev, None, reals)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
