"""Which parameters' gradients did autograd clone instead of adopting the flat-buffer view?  (development tool)"""
import os, sys, importlib, collections, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
HF = importlib.import_module("attention-gan_amd.backend.functional")
OPT = importlib.import_module("attention-gan_amd.optim")
DEV = torch.device("cuda:0")
step = bench.build(DEV, 24, HF)
words, sent, lens, reals = bench.synthetic_batch(DEV, 24, 1)
names = {}
for tag, mod in [("G", step.G)] + [(f"D{i}", d) for i, d in enumerate(step.Ds)]:
    for k, p in mod.named_parameters(): names[id(p)] = f"{tag}/{k}"
log = collections.Counter()
orig = OPT.FlatAdam._rebind
def patched(self, indices=None):
    base = self.grad.data_ptr()
    it = zip(self.params, self.offsets) if indices is None else ((self.params[i], self.offsets[i]) for i in indices)
    for p, o in it:
        g = p.grad
        if g is not None and g.data_ptr() != base + 4 * o:
            d = p._agan_grad_dst
            log[(names.get(id(p), "?"), f"written={d.written} edges={d.edges}")] += 1
    return orig(self, indices)
OPT.FlatAdam._rebind = patched
for _ in range(2): step.step(words, sent, lens, None, reals)
torch.cuda.synchronize()
log.clear()
step.step(words, sent, lens, None, reals)
torch.cuda.synchronize()
print("gradients not living in the flat buffer at rebind time:", sum(log.values()))
for (n, info), c in sorted(log.items()): print(f"  {c} {n}  {info}")
