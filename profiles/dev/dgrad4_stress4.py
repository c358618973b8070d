"""Image-gradient launches next to MFMA convolutions of another mode on side streams: does the packed-pair diagnostic build (AGAN_LIB=.../libagan_pkpairs.so)
lose bit-stability when its waves share compute units with bf16 / fp32 MFMA waves?  usage: dgrad4_stress4.py [side precision: bf16x6|f32|bf16|none]"""
import os, sys, importlib, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
HF = importlib.import_module("attention-gan_amd.backend.functional")
L = importlib.import_module("attention-gan_amd.backend.lib")
side_prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x6"
DEV = "cuda:0"
g = torch.Generator().manual_seed(7)
if side_prec != "none": HF.set_precision(L.PRECISIONS[side_prec])
streams = [torch.cuda.Stream() for _ in range(3)]
sides = [torch.cuda.Stream() for _ in range(2)]
xs = torch.randn(24, 64, 128, 128, device=DEV); ws = (torch.randn(128, 64, 3, 3, device=DEV) / 24)
xd2 = torch.randn(48, 64, 128, 128, device=DEV); wd2 = (torch.randn(128, 64, 4, 4, device=DEV) / 32)
cases = []
for H in (64, 128, 256):
    x = torch.randn(24, 3, H, H, generator=g).to(DEV)
    w = (torch.randn(64, 3, 4, 4, generator=g) / 48 ** 0.5).to(DEV)
    gy = torch.randn(24, 64, H // 2, H // 2, generator=g).to(DEV)
    cases.append((x, w, gy))
first = [None] * 3; nd = [0] * 3; detail = [""] * 3
N = 60
for it in range(N):
    if side_prec != "none":
        for s in sides:
            s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(sides[0]):
            for _ in range(3): ys = HF.conv2d(xs, ws, None, "same")
        with torch.cuda.stream(sides[1]):
            for _ in range(2): yd = HF.conv2d(xd2, wd2, None, "down")
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
                detail[i] = f" {d.shape[0]} elements, ch {sorted(set(d[:,1].tolist()))}, rows {sorted(set(d[:,2].tolist()))[:8]}, cols {sorted(set(d[:,3].tolist()))[:10]}"
for i, H in enumerate((64, 128, 256)):
    print(f"side convs {side_prec}: {H}x{H} image gradient: runs differing from the first {nd[i]}/{N - 1}{detail[i]}", flush=True)
