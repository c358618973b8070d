"""Determinism of the 4-class image-gradient strip kernel under concurrent load (development tool)."""
import os, sys, importlib, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
HF = importlib.import_module("attention-gan_amd.backend.functional")
F = torch.nn.functional
DEV = "cuda:0"
g = torch.Generator().manual_seed(7)
side = torch.cuda.Stream()
side2 = torch.cuda.Stream()
a = torch.randn(4096, 4096, device=DEV)
xs = torch.randn(24, 64, 64, 64, device=DEV); ws = torch.randn(128, 64, 3, 3, device=DEV) / 24
for (B, H, Cout) in [(24, 64, 64), (24, 128, 64), (24, 256, 64)]:
    x = torch.randn(B, 3, H, H, generator=g)
    w = torch.randn(Cout, 3, 4, 4, generator=g) / 48 ** 0.5
    gy = torch.randn(B, Cout, H // 2, H // 2, generator=g)
    xr = x.clone().requires_grad_(True)
    F.conv2d(xr, w, None, stride=2, padding=1).backward(gy)
    ref = xr.grad
    xd0 = x.to(DEV); gyd = gy.to(DEV); wd0 = w.to(DEV)
    first = None; worst = 0.0; ndiff = 0; detail = ""
    for it in range(60):
        with torch.cuda.stream(side):
            for _ in range(3): a2 = a @ a
        with torch.cuda.stream(side2):
            ys = HF.conv2d(xs, ws, None, "same")
        xd = xd0.clone().requires_grad_(True); wd = wd0.clone().requires_grad_(True)
        y = HF.conv2d(xd, wd, None, "down"); y.backward(gyd)
        torch.cuda.synchronize()
        got = xd.grad.cpu()
        err = float((got - ref).abs().max() / ref.abs().max())
        worst = max(worst, err)
        if first is None: first = got
        elif not torch.equal(first, got):
            ndiff += 1
            if not detail:
                d = (first != got).nonzero()
                detail = f" first mismatch set: {d.shape[0]} elements, channels {sorted(set(d[:,1].tolist()))}, rows {sorted(set(d[:,2].tolist()))[:12]}, cols {sorted(set(d[:,3].tolist()))[:24]}, batch {sorted(set(d[:,0].tolist()))[:8]}"
    print(f"B{B} 3x{H} dx: worst rel err {worst:.2e}, runs differing from the first: {ndiff}/59{detail}", flush=True)
