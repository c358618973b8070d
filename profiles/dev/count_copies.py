"""Where do the device-to-device copies of a train step come from?  (development tool: torch dispatch trace of aten::copy_/clone/contiguous)"""
import os, sys, importlib, collections, traceback, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from torch.utils._python_dispatch import TorchDispatchMode
HF = importlib.import_module("attention-gan_amd.backend.functional")
DEV = torch.device("cuda:0")
step = bench.build(DEV, 24, HF)
words, sent, lens, reals = bench.synthetic_batch(DEV, 24, 1)
def run():
    return step.step(words, sent, lens, None, reals)
for _ in range(2): run()
torch.cuda.synchronize()
cnt = collections.Counter()
class Mode(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if any(k in name for k in ("copy_", "clone", "_to_copy", "contiguous", "cat", "add", "mul", "fill", "zero")):
            fr = [f for f in traceback.extract_stack() if "attention-gan_amd" in f.filename or "bench.py" in f.filename]
            where = f"{os.path.basename(fr[-1].filename)}:{fr[-1].lineno}" if fr else "?"
            n = args[0].numel() if args and torch.is_tensor(args[0]) else -1
            cnt[(name, where)] += 1
        return func(*args, **(kwargs or {}))
with Mode():
    run()
torch.cuda.synchronize()
for (name, where), c in sorted(cnt.items(), key=lambda kv: -kv[1])[:60]:
    print(f"{c:4d}  {name:40s} {where}")
