#!/usr/bin/env python3
"""Turn two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE) into the per-kernel summaries and the traffic JSON bench.py reads.

Collection (on the GPU box; counters in their own runs, kernel trace only -- see DESIGN.md "Measurement"):

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --graph off
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --graph off

    python profiles/make_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles r02

rocprofv3 reports both counters in KB.  gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE tallies a 128-B request as 64 B,
so reads = 2 * FETCH_SIZE; WRITE_SIZE is exact.  traffic = 2 * FETCH_SIZE + WRITE_SIZE, averaged per launch of the kernel.
"""
import csv
import glob
import json
import os
import sys


def per_kernel(directory, counter):
    files = glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no counter_collection.csv under {directory}")
    agg = {}
    with open(files[0], newline="") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            d = agg.setdefault(row["Kernel_Name"], [0, 0.0])
            d[0] += 1
            d[1] += float(row["Counter_Value"])
    return agg


def write_summary(path, agg):
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "launches", "total_KB", "avg_KB_per_launch"])
        for name, (n, kb) in rows:
            w.writerow([name, n, round(kb), round(kb / n, 1)])


def bench_name(kernel):
    """the name bench.py's ConvTimer gives the kernel's launches (None for kernels it does not time)"""
    if "conv_gather_f32_kernel<128, 128" in kernel:
        return "conv_gather_f32_n128"
    if "conv_gather_f32_kernel<128, 64" in kernel:
        return "conv_gather_f32_n64"
    if "conv_gather_f32_kernel<128, 32" in kernel:
        return "conv_gather_f32_n32"
    if "conv_wgrad_f32_kernel" in kernel:
        return "conv_wgrad_f32"
    return None


def main():
    fetch_dir, write_dir, out_dir, tag = sys.argv[1:5]
    fetch, write = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
    write_summary(os.path.join(out_dir, f"{tag}_pmc_fetch_size_by_kernel.csv"), fetch)
    write_summary(os.path.join(out_dir, f"{tag}_pmc_write_size_by_kernel.csv"), write)
    groups = {}
    for kernel in set(fetch) | set(write):
        b = bench_name(kernel)
        if b is None:
            continue
        g = groups.setdefault(b, {"launches": 0, "fetch_kb": 0.0, "write_kb": 0.0})
        n_f, kb_f = fetch.get(kernel, (0, 0.0))
        n_w, kb_w = write.get(kernel, (0, 0.0))
        g["launches"] += max(n_f, n_w)
        g["fetch_kb"] += kb_f
        g["write_kb"] += kb_w
    kernels = {}
    for b, g in sorted(groups.items()):
        n = max(1, g["launches"])
        fr, wr = g["fetch_kb"] * 1024 / n, g["write_kb"] * 1024 / n
        kernels[b] = {"launches": g["launches"], "fetch_size_raw": round(fr), "write_size": round(wr), "traffic": round(2 * fr + wr)}
    doc = {
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, kernel trace only) over "
                  "`python bench.py --steps 2 --warmup 1 --no-cpu-baseline --graph off`; summarised by profiles/make_traffic.py",
        "unit": "bytes per launch (average over all launches of the kernel in the run)",
        "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request, i.e. half of a streamed read (MI355X_MICROARCH.md, HBM): "
                      "traffic = 2*FETCH_SIZE + WRITE_SIZE.  The x2 is calibrated for 16-B/lane streams; the conv gathers issue "
                      "4-B/lane (256 B per wave) loads, so their read side is an upper-bound estimate.",
        "kernels": kernels,
    }
    with open(os.path.join(out_dir, f"{tag}_traffic.json"), "w") as f:
        json.dump(doc, f, indent=1)
    print(json.dumps(kernels, indent=1))


if __name__ == "__main__":
    main()
