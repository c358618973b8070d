"""Three image-gradient launches (64 / 128 / 256 images, batch 24) concurrently on three streams, repeated: determinism check (development tool)."""
import os, sys, importlib, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
HF = importlib.import_module("attention-gan_amd.backend.functional")
F = torch.nn.functional
DEV = "cuda:0"
g = torch.Generator().manual_seed(7)
streams = [torch.cuda.Stream() for _ in range(3)]
cases = []
for H in (64, 128, 256):
    x = torch.randn(24, 3, H, H, generator=g).to(DEV)
    w = (torch.randn(64, 3, 4, 4, generator=g) / 48 ** 0.5).to(DEV)
    gy = torch.randn(24, 64, H // 2, H // 2, generator=g).to(DEV)
    cases.append((x, w, gy))
first = [None] * 3
nd = [0] * 3
detail = [""] * 3
for it in range(40):
    grads = []
    for i, (x, w, gy) in enumerate(cases):
        streams[i].wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(streams[i]):
            xd = x.clone().requires_grad_(True)
            y = HF.conv2d(xd, w, None, "down")
            y.backward(gy)
            grads.append(xd)
    torch.cuda.synchronize()
    for i, xd in enumerate(grads):
        got = xd.grad.cpu()
        if first[i] is None: first[i] = got
        elif not torch.equal(first[i], got):
            nd[i] += 1
            if not detail[i]:
                d = (first[i] != got).nonzero()
                detail[i] = f" {d.shape[0]} elements, ch {sorted(set(d[:,1].tolist()))}, rows {sorted(set(d[:,2].tolist()))[:10]}, cols {sorted(set(d[:,3].tolist()))[:12]}"
for i, H in enumerate((64, 128, 256)):
    print(f"{H}x{H}: runs differing from the first {nd[i]}/39{detail[i]}", flush=True)
