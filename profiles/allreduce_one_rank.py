import os, time, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29546")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
x = torch.zeros(64 << 18, device="cuda")          # 64 MB
big = torch.randn(8192, 8192, device="cuda")
comm = torch.cuda.Stream(priority=-1)
for mode in ("async on side stream", "sync on current stream"):
    for busy in (False, True):
        torch.cuda.synchronize()
        if busy:
            for _ in range(20):
                big @ big                     # ~20 x 7 ms of queued GPU work
        t0 = time.perf_counter()
        hs = []
        for _ in range(8):
            if mode.startswith("async"):
                comm.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(comm):
                    hs.append(dist.all_reduce(x, async_op=True))
            else:
                dist.all_reduce(x)
        t1 = time.perf_counter()
        for h in hs:
            h.wait()
        t2 = time.perf_counter()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        print(f"{mode:24s} gpu busy={busy}: 8 calls issue {1e3 * (t1 - t0):7.2f} ms, waits {1e3 * (t2 - t1):6.2f} ms, drain {1e3 * (t3 - t2):7.2f} ms", flush=True)
dist.destroy_process_group()
