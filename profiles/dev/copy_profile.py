"""torch.profiler view of one train step: which autograd nodes / python lines issue aten::copy_, aten::add, aten::mul, aten::cat (development tool)"""
import os, sys, importlib, collections, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from torch.profiler import profile, ProfilerActivity
HF = importlib.import_module("attention-gan_amd.backend.functional")
DEV = torch.device("cuda:0")
step = bench.build(DEV, 24, HF)
words, sent, lens, reals = bench.synthetic_batch(DEV, 24, 1)
for _ in range(2): step.step(words, sent, lens, None, reals)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    step.step(words, sent, lens, None, reals)
    torch.cuda.synchronize()
ev = prof.events()
want = ("aten::copy_", "aten::add", "aten::add_", "aten::mul", "aten::cat", "aten::clone", "aten::fill_", "aten::zero_", "aten::sum", "aten::to", "aten::_to_copy")
cnt = collections.Counter()
for e in ev:
    if e.name in want:
        p = e.cpu_parent
        chain = []
        while p is not None and len(chain) < 4:
            chain.append(p.name); p = p.cpu_parent
        st = [s for s in (e.stack or []) if "attention-gan_amd" in s or "bench.py" in s]
        cnt[(e.name, " < ".join(chain)[:110], st[0][-60:] if st else "")] += 1
for (n, chain, st), c in sorted(cnt.items(), key=lambda kv: -kv[1])[:50]:
    print(f"{c:4d} {n:14s} {chain}  {st}")
