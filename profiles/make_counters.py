#!/usr/bin/env python3
"""Per-kernel-family counter evidence from rocprofv3 passes over bench.py (round 2).

Collection (on the GPU box; counters in their own runs, kernel trace only; the program directly after `--`):

    cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; B="python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-variants --graph off --precision $MODE"
    rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ev_${MODE}_stats -o p -- $B
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 \   # (_F16 for the f16 modes)
              --kernel-trace --output-format csv -d $R/gpurun_out/ev_${MODE}_mfma -o p -- $B
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/ev_${MODE}_fetch -o p -- $B
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/ev_${MODE}_write -o p -- $B

    python profiles/make_counters.py gpurun_out/ev_${MODE} profiles r02_${MODE}

Outputs  profiles/<tag>_kernel_stats.csv          (copy of rocprofv3's per-kernel time summary)
         profiles/<tag>_kernel_family_counters.csv  one row per kernel family:
              launches, avg duration (us, from the dispatch timestamps of the MFMA pass), share of the summed kernel time,
              MFMA utilisation  = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 128)
                                  (busy cycles are summed over the chip's 1024 SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs:
                                   active cycles x 1024 / 8 SIMD-cycles were available -- MI355X_MICROARCH.md, DVFS / cycle constants),
              MFMA MOPS (512-flop units) per launch,
              HBM-side bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (gfx950: FETCH_SIZE tallies a 128-B request as 64 B;
                                   calibrated for 16-B/lane streams, an upper bound for the 4-B/lane gathers) and the GB/s they imply
         profiles/<tag>_traffic.json               bytes per launch of the conv kernel families: what bench.py reports as roofline.traffic
"""
import csv
import glob
import json
import os
import shutil
import sys

FAMILIES = [  # (family, substrings any of which selects the kernel)
    # row-block gather (conv_p16.hip), by geometry kind = 2nd template argument: <ET, GK, BN, IN16, OUT16, NI>
    ("conv_p16 4x4-s2 gather (D forward convs; upsample-conv dgrad)", ["conv_p16_kernel<0, 2,", "conv_p16_kernel<1, 2,"]),
    ("conv_p16 2x2-class gather (D conv dgrad; upsample-conv fwd)", ["conv_p16_kernel<0, 1,", "conv_p16_kernel<1, 1,"]),
    ("conv_p16 3x3 gather (G ResBlock / D Block3x3 fwd + dgrad)", ["conv_p16_kernel<0, 0,", "conv_p16_kernel<1, 0,"]),
    ("conv_gather_patch16 (fwd/dgrad, 16-bit MFMA)", ["conv_patch_kernel"]),
    ("conv_wgrad_rows (row-resident weight gradient, conv_wgrows.hip)", ["conv_wgrad_rows_kernel"]),
    ("conv_wgrad_patch16", ["conv_patch_wgrad_kernel"]),
    ("conv_wino gather (fp32 Winograd: conv3x3 fwd/dgrad, conv4x4-s2 fwd + class-wise dgrad)", ["conv_wino_f32_kernel", "conv_wino_h_f32_kernel", "conv_wino_s2_f32_kernel", "conv_wino_cls_f32_kernel"]),
    ("conv_wino weight gradient (fp32 Winograd F(3x3,2x2))", ["conv_wino_wgrad_f32_kernel"]),
    ("conv_gather_f32 (fwd/dgrad)", ["conv_gather_f32_kernel"]),
    ("conv_wgrad_f32", ["conv_wgrad_f32_kernel"]),
    ("conv_small_n (RGB heads, image gradient)", ["small_n_kernel", "small_strip_kernel"]),
    ("sum_slabs (split-K / split-pixel combine)", ["sum_slabs_kernel", "wino_wgrad_sum_kernel", "sum_unpack_wgrad_kernel"]),
    ("weight packs", ["pack_", "weight_transform_kernel"]),
    ("absmax (f16x3: max|x| of small tensors)", ["absmax_kernel"]),
    ("bn_stats_partial", ["bn_stats_partial"]),
    ("bn_act_fwd (normalise + GLU/LeakyReLU)", ["bn_act_fwd_kernel"]),
    ("bn_bwd_partial", ["bn_bwd_partial"]),
    ("bn_bwd_apply", ["bn_bwd_apply"]),
    ("bn_small (one-launch BatchNorm)", ["bn_small_"]),
    ("bn finals", ["bn_stats_final", "bn_bwd_final", "bn_stats_rows"]),
    ("attention fwd", ["attn_fwd_kernel", "attn_proj_kernel"]),
    ("attention bwd", ["attn_bwd", "attn_dproj"]),
    ("DAMSM losses", ["words_pair", "pair_slab_sum", "contrastive_ce", "sent_", "func_attn"]),
    ("fused Adam", ["adam_"]),
]


def family_of(kernel):
    for fam, pats in FAMILIES:
        if any(p in kernel for p in pats):
            return fam
    return "other (ATen glue, activations, heads)"


def bench_name(kernel, mode):
    """the name bench.py's ConvTimer gives the kernel's launches (None for kernels it does not time)"""
    for n in (128, 64, 32):
        if f"conv_gather_f32_kernel<128, {n}" in kernel:
            return f"conv_gather_f32_n{n}"
    if "conv_wgrad_f32_kernel" in kernel or "conv_wino_wgrad_f32_kernel" in kernel:
        return "conv_wgrad_f32"
    if "conv_wino_h_f32_kernel" in kernel:        # 64-channel tiles of layers with any channel count: booked with the 128-channel family
        return "conv_gather_f32_n128"
    for wk in ("conv_wino_f32_kernel<", "conv_wino_s2_f32_kernel<", "conv_wino_cls_f32_kernel<"):
        if wk in kernel:                      # <NFW>: cout fragments per workgroup = 128 / 64 output channels
            return "conv_gather_f32_n128" if wk + "4>" in kernel else "conv_gather_f32_n64"
    if "conv_patch_wgrad_kernel" in kernel or "conv_wgrad_rows_kernel" in kernel:
        return f"conv_wgrad_{mode}"
    if "conv_patch_kernel" in kernel:
        for n in (128, 64, 32):
            if f", {n}, 4>" in kernel or f", {n}, 9>" in kernel:
                return f"conv_gather_{mode}_n{n}"
    if "conv_p16_kernel<" in kernel:          # <ET, GK, BN, IN16, OUT16, NI[, PX2]>: the cout tile is the 3rd template argument
        args = kernel.split("conv_p16_kernel<")[1].split(">")[0].split(",")
        return f"conv_gather_{mode}_n{int(args[2])}"
    return None


def short_symbol(name):
    """same rule as bench.py::short_symbol: the kernel symbol without return type, anonymous namespace and parameter list"""
    s = name[5:] if name.startswith("void ") else name
    s = s.replace("(anonymous namespace)::", "")
    depth = 0
    for i, ch in enumerate(s):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            return s[:i]
    return s


def read_pass(directory):
    files = glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no counter_collection.csv under {directory}")
    per_dispatch = {}
    with open(files[0], newline="") as f:
        for row in csv.DictReader(f):
            d = per_dispatch.setdefault(row["Dispatch_Id"], {"kernel": row["Kernel_Name"], "t0": int(row["Start_Timestamp"]),
                                                             "t1": int(row["End_Timestamp"]), "c": {},
                                                             "grid": str(int(row.get("Grid_Size", "0") or 0) // 256)})
            d["c"][row["Counter_Name"]] = d["c"].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
    return per_dispatch


def main():
    prefix, out_dir, tag = sys.argv[1:4]
    parts = [p for p in tag.split("_") if p not in ("stage1", "b64", "stage4", "b8")]      # r04_bf16 / r04_bf16_s16 / r04_bf16_stage1_b64
    mode = parts[-2] if parts[-1] == "s16" else parts[-1]
    mfma, fetch, write = read_pass(prefix + "_mfma"), read_pass(prefix + "_fetch"), read_pass(prefix + "_write")
    stats = glob.glob(os.path.join(prefix + "_stats", "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        shutil.copy(stats[0], os.path.join(out_dir, f"{tag}_kernel_stats.csv"))
    fam = {}
    total_ns = 0
    for d in mfma.values():
        f = fam.setdefault(family_of(d["kernel"]), {"n": 0, "ns": 0, "busy": 0.0, "gui": 0.0, "mops": 0.0, "fetch": 0.0, "write": 0.0, "nf": 0, "nw": 0})
        f["n"] += 1
        f["ns"] += d["t1"] - d["t0"]
        total_ns += d["t1"] - d["t0"]
        f["busy"] += d["c"].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        f["gui"] += d["c"].get("GRBM_GUI_ACTIVE", 0.0)
        f["mops"] += sum(d["c"].get("SQ_INSTS_VALU_MFMA_MOPS_" + t, 0.0) for t in ("F32", "BF16", "F16"))
    # optional instruction-mix pass (collect_evidence.sh): VALU / MFMA / SALU / LDS instructions issued, per family
    insts = {}
    if glob.glob(os.path.join(prefix + "_insts", "**", "*counter_collection.csv"), recursive=True):
        for d in read_pass(prefix + "_insts").values():
            i = insts.setdefault(family_of(d["kernel"]), {"valu": 0.0, "mfma": 0.0, "salu": 0.0, "lds": 0.0})
            i["valu"] += d["c"].get("SQ_INSTS_VALU", 0.0)
            i["mfma"] += d["c"].get("SQ_INSTS_MFMA", 0.0)
            i["salu"] += d["c"].get("SQ_INSTS_SALU", 0.0)
            i["lds"] += d["c"].get("SQ_INSTS_LDS", 0.0)
    conv = {}
    for src, key, cnt in ((fetch, "fetch", "nf"), (write, "write", "nw")):
        name = "FETCH_SIZE" if key == "fetch" else "WRITE_SIZE"
        for d in src.values():
            kb = d["c"].get(name, 0.0)
            f = fam.setdefault(family_of(d["kernel"]), {"n": 0, "ns": 0, "busy": 0.0, "gui": 0.0, "mops": 0.0, "fetch": 0.0, "write": 0.0, "nf": 0, "nw": 0})
            f[key] += kb * 1024
            f[cnt] += 1
            b = bench_name(d["kernel"], mode)
            if b:
                c = conv.setdefault(b, {"fetch": 0.0, "write": 0.0, "nf": 0, "nw": 0})
                c[key] += kb * 1024
                c[cnt] += 1
    rows = []
    for name, f in sorted(fam.items(), key=lambda kv: -kv[1]["ns"]):
        n = max(1, f["n"])
        avg_us = f["ns"] / n / 1e3
        traffic = (2 * f["fetch"] / max(1, f["nf"]) + f["write"] / max(1, f["nw"])) if (f["nf"] or f["nw"]) else 0.0
        rows.append([name, f["n"], round(avg_us, 1), round(100.0 * f["ns"] / max(1, total_ns), 2),
                     round(100.0 * f["busy"] / (f["gui"] * 128.0), 1) if f["gui"] else "",
                     round(f["mops"] / n), round(traffic / 1e6, 2), round(traffic / (avg_us * 1e-6) / 1e9) if avg_us else "",
                     # effective shader clock while the family's kernels ran: GRBM_GUI_ACTIVE is summed over the 8 XCDs (MI355X_MICROARCH.md,
                     # DVFS give-back); reads high for dispatches much shorter than 0.3 ms
                     round(f["gui"] / 8.0 / f["ns"], 2) if f["ns"] and f["gui"] else ""])
        if insts:
            i = insts.get(name)
            # SQ_INSTS_VALU counts the MFMAs too: the other vector instructions, the scalar and the LDS instructions, per MFMA instruction
            rows[-1] += ([round((i["valu"] - i["mfma"]) / i["mfma"], 2), round(i["salu"] / i["mfma"], 2), round(i["lds"] / i["mfma"], 2)]
                         if i and i["mfma"] > 0 else ["", "", ""])
    with open(os.path.join(out_dir, f"{tag}_kernel_family_counters.csv"), "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel_family", "launches", "avg_us", "pct_of_kernel_time", "mfma_util_pct", "mfma_mops_per_launch",
                    "hbm_MB_per_launch(2*FETCH+WRITE)", "hbm_GB_per_s", "effective_clock_GHz(GRBM_GUI_ACTIVE/8/duration)"] +
                   (["other_valu_insts_per_mfma", "salu_insts_per_mfma", "lds_insts_per_mfma"] if insts else []))
        w.writerows(rows)
    # per LAYER rows for the row-block gathers: one row per (kernel instance, grid), i.e. per layer shape -- the family average above mixes
    # 0.4 ms layers with 15 us ones
    layers = {}
    for d in mfma.values():
        if "conv_p16_kernel<" not in d["kernel"]:
            continue
        inst = d["kernel"].split("conv_p16_kernel<")[1].split(">")[0]
        key = (inst, d.get("grid", ""))
        L = layers.setdefault(key, {"n": 0, "ns": 0, "busy": 0.0, "gui": 0.0})
        L["n"] += 1
        L["ns"] += d["t1"] - d["t0"]
        L["busy"] += d["c"].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        L["gui"] += d["c"].get("GRBM_GUI_ACTIVE", 0.0)
    if layers:
        with open(os.path.join(out_dir, f"{tag}_p16_layer_counters.csv"), "w", newline="") as fh:
            w = csv.writer(fh)
            w.writerow(["conv_p16_kernel<ET, GK (0 3x3 | 1 2x2-class | 2 4x4-s2), BN, IN16, OUT16, NI[, PX2]>", "grid (workgroups x 256 threads)", "launches",
                        "avg_us", "total_ms", "mfma_util_pct"])
            for (inst, grid), L in sorted(layers.items(), key=lambda kv: -kv[1]["ns"]):
                w.writerow([inst, grid, L["n"], round(L["ns"] / L["n"] / 1e3, 1), round(L["ns"] / 1e6, 3),
                            round(100.0 * L["busy"] / (L["gui"] * 128.0), 1) if L["gui"] else ""])
    # per kernel SYMBOL (round 4): what bench.py's roofline / roofline_by_kernel look up -- keyed like rocprofv3's kernel_stats.csv
    sym = {}
    for d in mfma.values():
        e = sym.setdefault(short_symbol(d["kernel"]), {"n": 0, "ns": 0, "busy": 0.0, "gui": 0.0, "fetch": 0.0, "write": 0.0, "nf": 0, "nw": 0})
        e["n"] += 1
        e["ns"] += d["t1"] - d["t0"]
        e["busy"] += d["c"].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        e["gui"] += d["c"].get("GRBM_GUI_ACTIVE", 0.0)
    for src, key, cnt, cname in ((fetch, "fetch", "nf", "FETCH_SIZE"), (write, "write", "nw", "WRITE_SIZE")):
        for d in src.values():
            e = sym.setdefault(short_symbol(d["kernel"]), {"n": 0, "ns": 0, "busy": 0.0, "gui": 0.0, "fetch": 0.0, "write": 0.0, "nf": 0, "nw": 0})
            e[key] += d["c"].get(cname, 0.0) * 1024
            e[cnt] += 1
    symbols = {k: {"launches": e["n"], "avg_ms": round(e["ns"] / max(1, e["n"]) / 1e6, 4),
                   "mfma_busy_pct": round(100.0 * e["busy"] / (e["gui"] * 128.0), 1) if e["gui"] else None,
                   "traffic": round(2 * e["fetch"] / max(1, e["nf"]) + e["write"] / max(1, e["nw"])) if (e["nf"] or e["nw"]) else None}
               for k, e in sorted(sym.items(), key=lambda kv: -kv[1]["ns"]) if e["ns"] >= 0.002 * max(1, total_ns)}
    kernels = {b: {"launches": max(c["nf"], c["nw"]), "fetch_size_raw": round(c["fetch"] / max(1, c["nf"])),
                   "write_size": round(c["write"] / max(1, c["nw"])),
                   "traffic": round(2 * c["fetch"] / max(1, c["nf"]) + c["write"] / max(1, c["nw"]))} for b, c in sorted(conv.items())}
    doc = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, kernel trace only) over `python bench.py --steps 2 "
                     f"--warmup 1 --no-cpu-baseline --no-variants --graph off --precision {mode}" + (f" --storage {mode}" if parts[-1] == "s16" else "") +
                     "`; summarised by profiles/make_counters.py",
           "unit": "bytes per launch (average over all launches of the kernel in the run)",
           "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request (MI355X_MICROARCH.md, HBM): traffic = 2*FETCH_SIZE + WRITE_SIZE; "
                         "the x2 is calibrated for 16-B/lane streams, so it is an upper bound for the 4-B/lane patch loads",
           "kernels": kernels,
           "symbols_note": "per kernel symbol (as rocprofv3 names it, without return type / namespace / parameters): launches and average "
                           "duration in the MFMA pass (--single-stream), MFMA-busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 128), "
                           "traffic = 2*FETCH_SIZE + WRITE_SIZE bytes per launch; kernels below 0.2 % of the kernel time omitted",
           "symbols": symbols}
    with open(os.path.join(out_dir, f"{tag}_traffic.json"), "w") as fh:
        json.dump(doc, fh, indent=1)
    for r in rows:
        print(r)


if __name__ == "__main__":
    main()
