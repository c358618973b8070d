"""Run the same first train step from the same weights several times and compare every captured tensor with run 0 (development tool).
usage: python profiles/dev/step_repeat.py [precision] [runs]"""
import os, sys, importlib, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import bench
HF = importlib.import_module("attention-gan_amd.backend.functional")
LIB = importlib.import_module("attention-gan_amd.backend.lib")
mode = sys.argv[1] if len(sys.argv) > 1 else "bf16x6"
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 6
HF.set_precision(LIB.PRECISIONS[mode])
DEV = torch.device("cuda:0")
B = 24
g = torch.Generator().manual_seed(2024)
lens = torch.randint(2, 11, (B,), generator=g).tolist(); lens[3] = bench.T
data = dict(words=torch.randn(B, bench.EMB, bench.T, generator=g), sent=torch.randn(B, bench.EMB, generator=g),
            reals=[torch.rand(B, 3, r, r, generator=g) * 2 - 1 for r in (64, 128, 256)],
            noise=torch.randn(B, bench.Z, generator=g), eps=torch.randn(B, bench.COND, generator=g))
to = lambda t: t.to(DEV)
base = None
for run in range(runs):
    step = bench.build(DEV, B, HF)
    cap = {}
    orig = step.gen_loss.get_loss
    def wrapped(d, fake, _orig=orig, _cap=cap):
        idx = len([k for k in _cap if k.startswith("seen")]); _cap[f"seen{idx}"] = True
        fake.register_hook(lambda gr, i=idx: _cap.__setitem__(f"dfake{i}", gr.detach().clone()))
        return _orig(d, fake)
    step.gen_loss.get_loss = wrapped
    def grab(tag, opt, _cap=cap):
        mod = step.G if tag == "G" else step.Ds[int(tag[1])]
        for k, v in opt.named_gradients(mod).items():
            if tag == "G" and ("img_out" in k or "upsample4" in k): _cap[f"g{tag}/{k}"] = v.detach().clone()
    step.on_gradients = grab
    out = step.step(to(data["words"]), to(data["sent"]), lens, None, [to(r) for r in data["reals"]], to(data["noise"]), to(data["eps"]))
    torch.cuda.synchronize()
    cur = {k: v.cpu() for k, v in cap.items() if torch.is_tensor(v)}
    for i in range(3): cur[f"fake{i}"] = out["fake_imgs"][i].cpu()
    if base is None:
        base = cur
        print("captured:", sorted(base.keys()))
    else:
        msgs = []
        for k in sorted(base):
            if not torch.equal(base[k], cur[k]):
                d = (base[k] != cur[k])
                rel = float((base[k] - cur[k]).abs().max() / base[k].abs().max())
                idx = d.nonzero()
                info = f"{k}: {int(d.sum())} of {d.numel()} differ, max rel {rel:.2e}"
                if idx.shape[1] == 4:
                    info += f" | batch {sorted(set(idx[:,0].tolist()))[:6]} ch {sorted(set(idx[:,1].tolist()))} rows {sorted(set(idx[:,2].tolist()))[:10]} cols {sorted(set(idx[:,3].tolist()))[:16]}"
                msgs.append(info)
        print(f"run {run}: " + ("identical" if not msgs else "\n   " + "\n   ".join(msgs)), flush=True)
    del step, out
    torch.cuda.empty_cache()
