"""Does a tensor written by one kernel come back faster when the next kernel reads it soon after?  (Infinity Cache reuse probe: a read-read-write pass like
BatchNorm's backward pair over footprints of 12 .. 800 MB; development tool)"""
import torch
dev = "cuda:0"
def bench(n_floats, reps=20):
    x = torch.randn(n_floats, device=dev); d = torch.randn(n_floats, device=dev); out = torch.empty_like(x)
    # pass 1 (like bn_bwd_partial): reads x and d;  pass 2 (like bn_bwd_apply): reads x and d again, writes out
    for _ in range(3):
        s = (x * d).sum(); torch.add(x, d, out=out)
    torch.cuda.synchronize()
    e0, e1, e2 = torch.cuda.Event(True), torch.cuda.Event(True), torch.cuda.Event(True)
    t1 = t2 = 0.0
    for _ in range(reps):
        e0.record(); s = torch.dot(x, d); e1.record(); torch.add(x, d, out=out); e2.record()
        torch.cuda.synchronize()
        t1 += e0.elapsed_time(e1); t2 += e1.elapsed_time(e2)
    mb = n_floats * 4 / 1e6
    print(f"tensors of {mb:7.1f} MB each: read-read pass {2 * mb / (t1 / reps) / 1e3:6.2f} TB/s, read-read-write pass right after {3 * mb / (t2 / reps) / 1e3:6.2f} TB/s", flush=True)
for mb in (6, 12, 25, 50, 100, 200, 400):
    bench(int(mb * 1e6 / 4))
