"""Which weights are packed by single agan_pack_weight launches (instead of the per-optimiser batch) in a steady-state step?  (development tool)"""
import os, sys, importlib, collections, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
HF = importlib.import_module("attention-gan_amd.backend.functional")
L = importlib.import_module("attention-gan_amd.backend.lib")
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
storage = sys.argv[2] if len(sys.argv) > 2 else None
HF.set_precision(L.PRECISIONS[prec]); HF.set_activation_storage(storage)
DEV = torch.device("cuda:0")
step = bench.build(DEV, 24, HF)
words, sent, lens, reals = bench.synthetic_batch(DEV, 24, 1)
for _ in range(3): step.step(words, sent, lens, None, reals)
torch.cuda.synchronize()
log = collections.Counter()
orig = L.call
def call(name, *args):
    if name.startswith("agan_pack"):
        log[(name, tuple(int(a) if isinstance(a, int) else type(a).__name__ for a in args[2:9]))] += 1
    return orig(name, *args)
L.call = call
HF.L.call = call
step.step(words, sent, lens, None, reals)
torch.cuda.synchronize()
for k, c in sorted(log.items(), key=lambda kv: -kv[1]): print(c, k)
print("total pack calls in one step:", sum(log.values()))
