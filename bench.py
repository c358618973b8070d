#!/usr/bin/env python3
"""AttnGAN training throughput on MI355X (BASELINE.json metric: train images/s at 256x256 stage-3, batch 24/GPU).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one full GanTrainer batch (train.py:109-151): G forward (3 stages + 2 word attentions), three discriminator
updates (real + fake batch each), the generator update through all three discriminators + DAMSM words/sentence loss + KL,
four fused Adam steps; under N > 1 every rank runs its own 24-image shard (weak scaling) and weight gradients are
all-reduced over RCCL.  Inputs are synthetic and already resident in HBM when the timed region starts.
Prints ONE JSON line on rank 0.

--workload picks the BASELINE.json configuration (default: configs[2], the one the metric is quoted on):
    full3        configs[2]  full 3-stage step, batch 24, f32 arithmetic (the parity-proven headline)
    stage1_b64   configs[1]  stage-1 only (CA-net + gen1 + img_out1 + Disc64), batch 64, bf16 arithmetic
    stage4_b8    configs[4]  4-stage 512x512 extension (Generator4 + Disc64/128/256/512), batch 8, fp16 arithmetic
"""
import argparse
import ctypes
import importlib
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# hyper-parameters of the metric configuration (train.py:34-49; BASELINE.json configs[2])
GF, DF, EMB, COND, Z, T = 32, 64, 256, 100, 100, 10
# dense MFMA peaks, /opt/skills/guides/MI355X_MICROARCH.md (spec): f32 = v_mfma_f32_32x32x2_f32; the split modes are priced against the
# 16-bit peak although they issue 3 / 6 MFMAs per algorithmic product (their algorithmic fraction can therefore not exceed 1/3, 1/6)
MFMA_PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0, "f16": 2500.0, "bf16x3": 2500.0, "bf16x6": 2500.0, "f16x3": 2500.0}
HBM_PEAK_GBS = 8000.0
# what the conv engine multiplies in, per --precision (fp32 accumulate in every mode)
MFMA_PRODUCTS = {"f32": 1, "bf16": 1, "f16": 1, "bf16x3": 3, "bf16x6": 6, "f16x3": 3}
MFMA_PLANES = {"f32": 1, "bf16": 1, "f16": 1, "bf16x3": 2, "bf16x6": 3, "f16x3": 2}      # 16-bit planes of a packed weight
DTYPE_NOTE = {"f32": "f32", "bf16": "bf16", "f16": "f16", "bf16x3": "bf16x3", "bf16x6": "bf16x6", "f16x3": "f16x3"}

WORKLOADS = {
    "full3": dict(batch=24, precision="f32", reals=(64, 128, 256), config=2,
                  desc="full 3-stage AttnGAN 64->128->256 train step: G + 3xD updates + word attention + DAMSM words/sentence "
                       "loss + KL + 4x fused Adam (BASELINE.json configs[2])"),
    "stage1_b64": dict(batch=64, precision="bf16", reals=(64,), config=1,
                       desc="stage-1 only (64x64) G+D train step: CA-net + gen1 + img_out1, Disc64 update (real + fake), generator update "
                            "through Disc64 + KL, 2x fused Adam (BASELINE.json configs[1]; DAMSM acts on the 256x256 image only, "
                            "train.py:138-143, so this stage has none)"),
    "stage4_b8": dict(batch=8, precision="f16", reals=(64, 128, 256, 512), config=4,
                      desc="4-stage 64->128->256->512 train step: Generator4 + Disc64/128/256/512 updates + 3 word attentions + DAMSM on "
                           "the 512x512 image + KL + 5x fused Adam (BASELINE.json configs[4]: the 512x512 extension, built from the "
                           "reference's primitives -- SURVEY.md section 8d C5)"),
}


def short_symbol(name):
    """'void (anonymous namespace)::conv_gather_f32_kernel<128, 128, 2, 2, 0>(float const*, ...)' -> 'conv_gather_f32_kernel<128, 128, 2, 2, 0>':
    the key shared by this file, profiles/make_counters.py and a reader of rocprofv3's kernel_stats.csv (a substring of its Name column)"""
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


class ConvTimer:
    """HIP-event pairs recorded by the library itself on the launch stream, immediately around the main kernel of every
    conv-engine call (include/agan.h: agan_timer_arm) -- the slab-sum pass a split launch appends is outside the pair, so the
    per-kernel averages here are the ones `rocprofv3 --kernel-trace --stats` reports for the same kernels.  The library also says
    WHICH kernel it ran (agan_timer_last_kernel: the demangled symbol rocprofv3 prints), so every record is keyed by a real kernel.
    Durations are read after the instrumented region has been synchronised."""

    def __init__(self, lib, mode="f32"):
        self.lib, self.mode = lib, mode
        self.records, self.keys, self.enabled, self._pool = [], [], False, []
        self._name = ctypes.create_string_buffer(1024)
        self._pending = None

    def _event(self):
        if self._pool:
            return self._pool.pop()
        e = ctypes.c_void_p()
        self.lib.call("agan_timer_create", ctypes.byref(e))
        return e

    def begin(self, kind, phase, g, in_esz=4, out_esz=4):
        if not self.enabled:
            return
        K = g.Cin * g.R * g.S
        direct = 2.0 * g.B * g.OH * g.OW * g.Cout * K            # MACs of the direct contraction (x2) ...
        executed = direct
        # algorithmic = the REFERENCE's contraction for this call: Upsample+conv3x3 is 9 taps on the 2x grid,
        # the folded kernels (fwd: 4 classes x 2x2 taps; dgrad: 4x4 s2; wgrad: 4 classes) execute 4/9 of that.
        algorithmic = direct * (9.0 / 4.0) if kind == "up" else direct
        if self.mode == "f32" and g.Cout > 4:
            # ... of which the fp32 mode's Winograd kernels issue 16/36 (conv3x3), 36/64 (conv4x4 s2 forward) or 9/16 (its data gradient).
            # (the conv calls on this path carry no bias / activation epilogue except the first discriminator conv, which has 3 input channels)
            executed = direct * float(self.lib.load().agan_conv_executed_fraction(ctypes.byref(g), 0, 1 if phase == "wgrad" else 0, 1))
        # algorithmic HBM bytes of this call: gathered tensor + produced tensor + weights, each moved once, in the storage types of
        # this call (fp32, or 16 bits under --storage; a weight gradient reads both activations and writes fp32 weights)
        if phase == "wgrad":
            nbytes = in_esz * g.B * g.Cin * g.IH * g.IW + out_esz * g.B * g.Cout * g.OH * g.OW + 4.0 * g.Cout * K * (g.OS * g.OS)
        else:
            wesz = 4.0 if self.mode == "f32" else 2.0 * MFMA_PLANES.get(self.mode, 1)
            nbytes = in_esz * g.B * g.Cin * g.IH * g.IW + out_esz * g.B * g.Cout * g.OH * g.OW + wesz * g.Cout * K * (g.OS * g.OS)
        e0, e1 = self._event(), self._event()
        self.lib.call("agan_timer_arm", e0, e1)
        self._pending = [None, algorithmic, executed, nbytes, e0, e1]
        self.keys.append(f"{phase:5s} {kind:4s} B{g.B} {g.Cin:4d}x{g.IH:<3d} -> {g.Cout:4d}x{g.OH:<3d} taps {g.R}x{g.S} cls {g.OS * g.OS}")

    def end(self):
        if self._pending is None:
            return
        self.lib.call("agan_timer_last_kernel", self._name, 1024)
        self._pending[0] = self._name.value.decode() or "unknown"
        self.records.append(tuple(self._pending))
        self._pending = None

    def _ms(self, e0, e1):
        ms = ctypes.c_float()
        self.lib.call("agan_timer_elapsed_ms", e0, e1, ctypes.byref(ms))
        return ms.value

    def clear(self):
        for rec in self.records:
            self._pool += [rec[4], rec[5]]
        self.records, self.keys = [], []

    def layer_table(self, steps):
        """per (phase, layer shape): launches and ms per step, executed TFLOP/s, the kernel -- the list the next optimisation is read from"""
        by = {}
        for key, (name, fa, fe, nb, e0, e1) in zip(self.keys, self.records):
            d = by.setdefault(key, [0, 0.0, 0.0, short_symbol(name)])
            d[0] += 1; d[1] += self._ms(e0, e1); d[2] += fe
        rows = sorted(by.items(), key=lambda kv: -kv[1][1])
        out = [f"{'phase kind shape':58s} {'n/step':>6s} {'ms/step':>8s} {'exec TF/s':>9s}  kernel"]
        for key, (n, ms, fe, sym) in rows:
            out.append(f"{key:58s} {n / steps:6.1f} {ms / steps:8.3f} {fe / (ms * 1e-3) / 1e12:9.1f}  {sym}")
        out.append(f"{'total':58s} {sum(v[0] for v in by.values()) / steps:6.1f} {sum(v[1] for v in by.values()) / steps:8.3f}")
        return "\n".join(out)

    def summary(self):
        """{rocprof kernel symbol: [launches, ms, algorithmic flops, executed flops, algorithmic bytes]}"""
        by = {}
        for name, fa, fe, nb, e0, e1 in self.records:
            d = by.setdefault(name, [0, 0.0, 0.0, 0.0, 0.0])
            d[0] += 1; d[1] += self._ms(e0, e1); d[2] += fa; d[3] += fe; d[4] += nb
        return by


def baseline_metric_name():
    """the headline metric exactly as BASELINE.json words it (the file ships with the repo)"""
    try:
        with open(os.path.join(ROOT, "BASELINE.json")) as f:
            return json.load(f)["metric"]
    except (OSError, ValueError, KeyError):
        return "train images/sec at 256x256 stage-3, batch 24/GPU"


_PMC_CACHE = {}


def committed_counters(symbol, mode, storage="f32", workload="full3"):
    """What the COMMITTED rocprofv3 PMC passes of this same command measured for kernel `symbol` (short form): HBM bytes per launch
    (2 x FETCH_SIZE + WRITE_SIZE, gfx950 correction) and MFMA-busy -- profiles/r04_<mode>[_s16][_<workload>]_traffic.json, section
    "symbols", written by profiles/make_counters.py.  rocprofv3 cannot attach to a running process, so these are not from this run;
    the source file is named beside them.  {} if that kernel was not measured."""
    s16 = "_s16" if storage != "f32" else ""
    wl = "" if workload == "full3" else f"_{workload}"
    for name in (f"r04_{mode}{s16}{wl}_traffic.json",):
        path = os.path.join(ROOT, "profiles", name)
        if path not in _PMC_CACHE:
            try:
                with open(path) as f:
                    _PMC_CACHE[path] = json.load(f).get("symbols", {})
            except (OSError, ValueError):
                _PMC_CACHE[path] = {}
        hit = _PMC_CACHE[path].get(symbol)
        if hit:
            return dict(hit, source=f"profiles/{name}")
    return {}


def build(dev, batch, HF, encoder="standin", workload="full3"):
    GEN = importlib.import_module("attention-gan_amd.networks.generator")
    DISC = importlib.import_module("attention-gan_amd.networks.discriminators")
    ENC = importlib.import_module("attention-gan_amd.networks.cnn_encoder")
    TR = importlib.import_module("attention-gan_amd.trainers.trainer")
    torch.manual_seed(0)      # identical initial weights on every rank (and broadcast from rank 0 on top)
    if workload == "stage1_b64":
        G = GEN.Generator1(GF, EMB, Z, COND).to(dev)
        Ds = [DISC.Disc64(DF).to(dev)]
        enc = None                # DAMSM acts on the 256x256 image only (train.py:138-143)
    elif workload == "stage4_b8":
        S4 = importlib.import_module("attention-gan_amd.networks.stage4")
        G = S4.Generator4(GF, EMB, Z, COND).to(dev)
        Ds = [DISC.Disc64(DF).to(dev), DISC.Disc128(DF).to(dev), DISC.Disc256(DF).to(dev), S4.Disc512(DF).to(dev)]
        enc = ENC.StandInImageEncoder(EMB).to(dev)
    else:
        G = GEN.Generator(GF, EMB, Z, COND).to(dev)
        Ds = [DISC.Disc64(DF).to(dev), DISC.Disc128(DF).to(dev), DISC.Disc256(DF).to(dev)]
        enc = (ENC.StandInImageEncoder(EMB) if encoder == "standin" else ENC.CNNEncoder(EMB)).to(dev)
    if enc is not None:
        enc.freeze_all_weights()
        enc.eval()                 # the reference loads it with _load_weights(), which puts it in eval mode (trainer.py:124)
    # (AGAN_BUCKET_MB: gradient all-reduce bucket size for scaling experiments; default = the trainer's 64 MB)
    mb = os.environ.get("AGAN_BUCKET_MB")
    step = TR.GanTrainStep(G, Ds, enc, bucket_bytes=int(mb) << 20) if mb else TR.GanTrainStep(G, Ds, enc)
    step.overlap_weight_gradients = os.environ.get("AGAN_WGRAD_SIDE_STREAM", "0") == "1"      # (A/B switch)
    return step


def synthetic_batch(dev, batch, seed, resolutions=(64, 128, 256)):
    g = torch.Generator(device="cpu").manual_seed(seed)
    words = torch.randn(batch, EMB, T, generator=g).to(dev)             # frozen RNN bypassed: N(0,1) embeddings
    sent = torch.randn(batch, EMB, generator=g).to(dev)
    lens = [T] * batch
    reals = [(torch.rand(batch, 3, r, r, generator=g) * 2 - 1).to(dev) for r in resolutions]
    return words, sent, lens, reals


def usable_cores():
    """host cores this process may really use: scheduler affinity, cut down to the cgroup CPU quota where one is set (a one-GPU
    box exposes every core of the host but grants a 16-core share; more threads than that only oversubscribe)"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(int(parts[0]) / int(parts[1]))))
            else:
                q = int(parts[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f2:
                        n = min(n, max(1, q // int(f2.read())))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def cpu_baseline(batch, repeats=3):
    """The CPU oracle (a port of the reference's step) timed on this box's host cores, as BASELINE.md section 4 prescribes: full
    train steps at the metric shapes on all host cores available to the process, median of `repeats` after one warm-up step."""
    from oracle import attngan_oracle as O
    cores = usable_cores()
    torch.set_num_threads(cores)
    print(f"[bench] cpu_baseline: timing the CPU oracle on {cores} threads (1 warm-up + {repeats} steps) ...", file=sys.stderr, flush=True)
    gp = O.make_generator_params(GF, EMB, Z, COND, seed=0)
    dps = [O.make_disc_params(DF, r, seed=0) for r in (64, 128, 256)]
    gopt, dopts = O.AdamState(gp), [O.AdamState(d) for d in dps]
    ep = O.standin_encoder_params(EMB)
    g = torch.Generator().manual_seed(0)
    words, sent = torch.randn(batch, EMB, T, generator=g), torch.randn(batch, EMB, generator=g)
    reals = [torch.rand(batch, 3, r, r, generator=g) * 2 - 1 for r in (64, 128, 256)]
    noise, eps = torch.randn(batch, Z, generator=g), torch.randn(batch, COND, generator=g)
    times = []
    for i in range(repeats + 1):
        t0 = time.perf_counter()
        O.train_step(gp, dps, gopt, dopts, words, sent, [T] * batch, None, reals, noise, eps, lambda im: O.standin_encoder(im, ep))
        times.append(time.perf_counter() - t0)
    med = sorted(times[1:])[len(times[1:]) // 2]
    return {"value": round(batch / med, 4), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"full train steps, batch {batch}, same shapes as the GPU workload: median of {repeats} after 1 warm-up "
                      f"({', '.join(f'{t:.1f}' for t in times)} s)"}


def _rate(fn, batch, n, w):
    for _ in range(w):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return round(batch * n / (time.perf_counter() - t0), 1)


def side_measurements(args, dev, HF, LIB, step, words, sent, reals):
    """Short eager measurements reported BESIDE the headline, never inside it or its roofline:
    random caption lengths 2..10 (the headline uses full-length captions), the other arithmetic modes of the conv engine, the
    end-to-end step with the Inception-v3-shaped trunk (stock MIOpen convs, random weights) as the DAMSM image encoder and the
    bi-LSTM text encoder run on the device every step, and the other two BASELINE configurations in their own arithmetic
    (configs[1]: stage-1, batch 64, bf16; configs[4]: 512x512 stage, batch 8, fp16) -- `bench.py --workload ...` gives each of
    those its own line with a roofline."""
    # the SAME step and warm-up counts as the headline: a `--precision X` headline and the `precision_X` variant of another run are
    # then the same measurement (8-step variants read 6 % low in round 2: clocks and allocator pools had not settled)
    n, w = args.steps, max(args.warmup, 2)
    out = {"note": f"{n} eager steps each after {w} warm-up steps (the headline's counts), same batch; images/s"}
    B = args.batch
    g = torch.Generator().manual_seed(99)
    lens_r = torch.randint(2, T + 1, (B,), generator=g)
    lens_r[0] = T
    lens_r = lens_r.to(dev)
    out["random_caption_lengths_2_10"] = _rate(lambda: step.step(words, sent, lens_r, None, reals), B, n, w)
    lens_full = torch.full((B,), T, dtype=torch.int64, device=dev)
    for other in ("f32", "bf16x6", "f16x3", "bf16"):      # fp32-grade split modes (parity-tested at this size) and plain bf16
        if other == args.precision:
            continue
        HF.set_precision(LIB.PRECISIONS[other])
        try:
            out[f"precision_{other}"] = _rate(lambda: step.step(words, sent, lens_full, None, reals), B, n, w)
            if other == "bf16" and args.storage == "f32":
                HF.set_activation_storage("bf16")
                out["precision_bf16_storage_bf16"] = _rate(lambda: step.step(words, sent, lens_full, None, reals), B, n, w)
        finally:
            HF.set_activation_storage(None if args.storage == "f32" else args.storage)
            HF.set_precision(LIB.PRECISIONS[args.precision])
    try:
        RNN = importlib.import_module("attention-gan_amd.networks.rnn_encoder")
        step2 = build(dev, B, HF, "inception")
        rnn = RNN.RNNEncoder(vocabsize=1000, nhidden=EMB).to(dev).eval()
        rnn.freeze_all_weights()
        caps = torch.randint(1, 1000, (B, T), generator=g).to(dev)
        lens_host = [T] * B

        def e2e():
            with torch.no_grad():
                w_e, s_e = rnn(caps, lens_host)
            return step2.step(w_e.contiguous(), s_e.contiguous(), lens_full, None, reals)
        out["end_to_end_inception_trunk_plus_lstm"] = _rate(e2e, B, n, w)
        del step2
        torch.cuda.empty_cache()
    except Exception as exc:          # the side line must never cost the headline
        out["end_to_end_inception_trunk_plus_lstm"] = f"failed: {type(exc).__name__}: {exc}"
    # the other two BASELINE configurations, each in the arithmetic its config names (fewer steps: they are side lines)
    for wl in ("stage1_b64", "stage4_b8"):
        spec = WORKLOADS[wl]
        try:
            HF.set_precision(LIB.PRECISIONS[spec["precision"]])
            st = build(dev, spec["batch"], HF, workload=wl)
            wd, sn, _, rl = synthetic_batch(dev, spec["batch"], seed=77, resolutions=spec["reals"])
            ln = torch.full((spec["batch"],), T, dtype=torch.int64, device=dev)
            key = f"{wl}_{spec['precision']}"
            out[key] = _rate(lambda: st.step(wd, sn, ln, None, rl), spec["batch"], max(10, n // 2), w)
            if wl == "stage1_b64":
                HF.set_activation_storage("bf16")
                out[key + "_storage_bf16"] = _rate(lambda: st.step(wd, sn, ln, None, rl), spec["batch"], max(10, n // 2), w)
            del st, wd, sn, rl
            torch.cuda.empty_cache()
        except Exception as exc:
            out[f"{wl}_{spec['precision']}"] = f"failed: {type(exc).__name__}: {exc}"
        finally:
            HF.set_activation_storage(None if args.storage == "f32" else args.storage)
            HF.set_precision(LIB.PRECISIONS[args.precision])
    return out


def roofline_objects(timer, args, elapsed_ms_per_step, roof_steps, timing_note):
    """`roofline` (the dominant rocprof kernel of the instrumented steps) and `roofline_by_kernel` (every conv-engine kernel that takes
    >= 1 % of them), each keyed by the kernel symbol rocprofv3 prints, so that a reader of profiles/r04_*_kernel_stats.csv can divide
    the same numbers: achieved = ALGORITHMIC flops (the reference's contraction) / summed launch time;
    executed_tflops = what the kernel really issues (Winograd kernels issue 16/36, 36/64 or 9/16 of the direct contraction; the folded
    upsample conv 4/9); frac = achieved / peak (algorithmic: it can exceed the matrix pipe's busy fraction, which `mfma_busy_pct` gives
    from the committed PMC pass); frac_executed = executed / peak."""
    summ = timer.summary()
    if not summ:
        return None, []
    peak = MFMA_PEAK_TFLOPS[args.precision]
    products = MFMA_PRODUCTS[args.precision]
    total_ms = sum(v[1] for v in summ.values())

    def obj(name, n, ms, falg, fexec, nbytes):
        sym = short_symbol(name)
        hbm = "small_n" in sym or "small_strip" in sym       # <= 4 channels on one side: vector-ALU kernels, HBM-bound
        pmc = committed_counters(sym, args.precision, args.storage, args.workload)
        o = {"kernel": sym, "rocprof_name": name, "launches_per_step": round(n / roof_steps, 2), "avg_launch_ms": round(ms / n, 4),
             "share_of_step_time": round((ms / roof_steps) / elapsed_ms_per_step, 4),
             "algorithmic_bytes_per_launch": round(nbytes / n),
             "traffic": pmc.get("traffic"), "mfma_busy_pct": pmc.get("mfma_busy_pct"), "pmc_avg_launch_ms": pmc.get("avg_ms"),
             "pmc_source": pmc.get("source")}
        if hbm:
            gbs = nbytes / (ms * 1e-3) / 1e9
            o.update({"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4)})
        else:
            alg, exe = falg / (ms * 1e-3) / 1e12, fexec / (ms * 1e-3) / 1e12
            o.update({"bound": "mfma", "achieved": round(alg, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(alg / peak, 4),
                      "executed_tflops": round(exe, 2), "frac_executed": round(exe / peak, 4)})
            if products > 1:      # a split-precision mode issues `products` MFMAs per algorithmic multiply-add: the matrix pipe's own rate
                o.update({"mfma_products_per_multiply": products, "mfma_issue_tflops": round(products * exe, 1),
                          "mfma_issue_frac": round(products * exe / peak, 4)})
        return o
    ranked = sorted(summ.items(), key=lambda kv: -kv[1][1])
    by_kernel = [obj(name, *v) for name, v in ranked if v[1] >= 0.01 * total_ms]
    dom = next((o for o in by_kernel if o["bound"] == "mfma"), by_kernel[0])
    roofline = dict(dom)
    roofline.update({"frac_basis": "algorithmic flops of the reference's contraction (SURVEY.md section 8d) / kernel time; frac_executed "
                                   "counts what the kernel issues; mfma_busy_pct is the matrix pipe's busy fraction from the committed PMC "
                                   "pass of this command (profiles/, same kernel symbol)",
                     "traffic_source": "profiles/ (committed rocprofv3 FETCH_SIZE / WRITE_SIZE passes of this command, per kernel symbol), not this run",
                     "timing": timing_note, "conv_engine_ms_per_step": round(total_ms / roof_steps, 3)})
    return roofline, by_kernel


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=120, help="timed steps (default: a >= 3 s timed region at the metric config)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=list(WORKLOADS), default="full3",
                    help="BASELINE.json configuration: full3 = configs[2] (the metric's, default); stage1_b64 = configs[1]; stage4_b8 = configs[4]")
    ap.add_argument("--batch", type=int, default=None, help="images per GPU (default: the workload's -- 24 / 64 / 8)")
    ap.add_argument("--precision", choices=["f32", "bf16x6", "f16x3", "bf16x3", "bf16", "f16"], default=None,
                    help="MFMA mode of the conv engine (default: the workload's -- f32 / bf16 / f16): f32 = exact fp32 products "
                         "(v_mfma_f32_32x32x2_f32); bf16x6 = three bf16 planes, 6 MFMAs per product (fp32-grade); bf16x3 = two planes, 3 MFMAs; "
                         "bf16 / f16 = operands rounded to 16 bits (BASELINE configs[1] / configs[4] arithmetic)")
    ap.add_argument("--storage", choices=["f32", "bf16", "f16"], default="f32",
                    help="storage type of activations in HBM: f32 (the reference's layout, default) or the operand type of the one-plane "
                         "16-bit modes (--precision bf16 --storage bf16 / --precision f16 --storage f16): conv outputs, BatchNorm "
                         "inputs / outputs and their gradients are rounded once when stored and gathered without conversion")
    ap.add_argument("--image-encoder", choices=["standin", "inception"], default="standin",
                    help="frozen DAMSM image encoder plug-in: 'standin' = contract-only stub (SURVEY §8d prices the hot path without "
                         "the third-party trunk); 'inception' = Inception-v3-shaped trunk on stock MIOpen convs, random weights")
    ap.add_argument("--graph", choices=["auto", "on", "off", "segments"], default="auto",
                    help="launch mode of the step: captured HIP graph replay or eager.  auto: probe in UNTIMED steps whether the host keeps "
                         "the GPU fed (enqueue time > 0.85 x step time = launch-bound); if not, at 1 GPU capture the whole step and keep the "
                         "faster, under torch.distributed switch to `segments` on EVERY rank if ANY rank is launch-bound.  segments = nine HIP "
                         "graphs with the gradient exchange launched between them (GanTrainStep.capture_segments)")
    ap.add_argument("--single-stream", action="store_true",
                    help="run the discriminator updates on ONE stream (profiling aid: per-kernel PMC counters of a rocprofv3 pass are "
                         "diluted when kernels of different streams share the chip; the headline run overlaps them)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true",
                    help="skip the short side measurements reported beside the headline (random caption lengths, the other arithmetic modes, "
                         "the end-to-end step with the Inception-shaped trunk and the LSTM text encoder, the other two BASELINE configs)")
    ap.add_argument("--cpu-baseline-batch", type=int, default=24)
    ap.add_argument("--layer-table", default=None, help="write the per-layer conv timing table of the instrumented steps to this file")
    args = ap.parse_args()
    spec = WORKLOADS[args.workload]
    if args.batch is None:
        args.batch = spec["batch"]
    if args.precision is None:
        args.precision = spec["precision"]

    # stdout carries exactly ONE line, the JSON report of rank 0.  Libraries write there too (RCCL prints its version banner to
    # stdout when a communicator is created, MIOpen its find-db notes): everything but the report goes to stderr.
    sys.stdout.flush()
    report_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal knobs for a one-GPU box (the real multi-GPU run uses neither): AGAN_BENCH_BACKEND=gloo and
    # AGAN_BENCH_ONE_DEVICE=1 put every rank on cuda:0 and exchange gradients over gloo, so the whole N>1 control flow
    # (bucket hooks, barriers, max-over-ranks timing, rank-0 report) runs on real kernels
    backend = os.environ.get("AGAN_BENCH_BACKEND", "nccl")
    if os.environ.get("AGAN_BENCH_ONE_DEVICE") == "1":
        local = 0
    if world > 1 or os.environ.get("AGAN_DP_FORCE") == "1":        # (AGAN_DP_FORCE: one-rank rehearsal of the nccl path, dataparallel.py)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # every compute kernel of the step fills the chip: give RCCL's own stream priority, or the exchange of a bucket only starts
        # when the backward that follows it has drained (the trainer's comm streams stay at normal priority: dataparallel.py)
        os.environ.setdefault("TORCH_NCCL_HIGH_PRIORITY", "1")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the HIP path has no CPU fallback)")
    dev = torch.device("cuda", local)
    dp = dist.is_available() and dist.is_initialized()       # (world > 1, or the forced one-rank rehearsal)

    HF = importlib.import_module("attention-gan_amd.backend.functional")
    LIB = importlib.import_module("attention-gan_amd.backend.lib")
    DP = importlib.import_module("attention-gan_amd.dataparallel")
    HF.set_precision(LIB.PRECISIONS[args.precision])
    if args.storage != "f32":
        if args.storage != args.precision:
            raise SystemExit(f"--storage {args.storage} needs --precision {args.storage} (the storage type is the MFMA operand type)")
        HF.set_activation_storage(args.storage)
    step = build(dev, args.batch, HF, args.image_encoder, args.workload)
    if args.single_stream:
        step.overlap_discriminators = False
    words, sent, lens, reals = synthetic_batch(dev, args.batch, seed=1234 + rank, resolutions=spec["reals"])
    timer = ConvTimer(LIB, args.precision)
    HF.set_launch_observer(timer)
    buckets = [step.g_buckets] + list(step.d_buckets)

    # caption lengths live on the device like the rest of the batch (inputs are resident in HBM when the timed region starts):
    # a host list would cost two small blocking H2D copies per step, which stall the eager launch pipeline by ~4 ms
    lens_dev = torch.tensor(lens, dtype=torch.int64, device=dev)

    def eager_step():
        return step.step(words, sent, lens_dev, None, reals)

    # Launch mode.  Eager: ~850 launches per step from Python (~12 ms of host time, hidden behind ~24 ms of GPU time on an
    # unloaded host).  Graph: the whole step captured once and replayed -- no host work, but hipGraphLaunch adds per-node cost
    # and a process that has captured runs ~1.5 % slower even eagerly (857 vs 877 images/s measured).  `auto` therefore first checks,
    # in UNTIMED steps, whether eager is launch-bound on this host (enqueue time vs step time).  At N=1 it then captures the whole-step
    # graph and keeps whichever replays / steps faster.  Under torch.distributed the RCCL exchange is not captured: the graph form is
    # `segments` (nine graphs, eager exchange between them), chosen when ANY rank is launch-bound -- 8 ranks share the host's cores, and
    # the ranks must agree (the two forms issue their collectives in different groupings), so the verdict is one MAX all-reduce.
    def probe(fn, n=6):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        t_host = (time.perf_counter() - t0) / n
        torch.cuda.synchronize()
        return t_host, (time.perf_counter() - t0) / n

    use_graph = args.graph == "on" and not dp
    graphed = None
    launch_note = ""
    dp_info = None
    if args.graph == "auto":
        for _ in range(2):
            eager_step()
        t_host, t_eager = probe(eager_step)
        host_bound = t_host > 0.85 * t_eager
        if dp:
            switch = DP.any_rank(host_bound, None, dev)
            per_rank = [None] * dist.get_world_size()
            dist.all_gather_object(per_rank, (round(t_host * 1e3, 2), round(t_eager * 1e3, 2)))
            dp_info = {"probe_host_enqueue_ms_per_rank": [p[0] for p in per_rank], "probe_eager_step_ms_per_rank": [p[1] for p in per_rank],
                       "launch_bound_on_some_rank": bool(switch)}
            if switch:
                graphed = step.capture_segments(words, sent, lens_dev, reals, warmup=2)
                use_graph = True
                launch_note = "segments"
            print(f"[bench] rank {rank}: eager host enqueue {t_host * 1e3:.1f} ms of a {t_eager * 1e3:.2f} ms step -> "
                  f"{'graph segments (some rank is launch-bound)' if switch else 'eager'}", file=sys.stderr, flush=True)
        elif host_bound:                                 # the host cannot keep the GPU fed: try the graph
            graphed = step.capture(words, sent, lens_dev, reals, warmup=2)
            t_graph = min(probe(graphed.replay)[1], probe(graphed.replay)[1])
            use_graph = t_graph < t_eager
            print(f"[bench] eager is launch-bound here (host {t_host * 1e3:.1f} of {t_eager * 1e3:.1f} ms/step); graph replay "
                  f"{t_graph * 1e3:.2f} ms/step -> {'graph' if use_graph else 'eager'}", file=sys.stderr, flush=True)
        else:
            print(f"[bench] eager: host enqueue {t_host * 1e3:.1f} ms of a {t_eager * 1e3:.2f} ms step -> eager (no graph capture)",
                  file=sys.stderr, flush=True)
    elif not dp and args.graph == "on":
        graphed = step.capture(words, sent, lens_dev, reals, warmup=2)
    elif args.graph == "segments":
        graphed = step.capture_segments(words, sent, lens_dev, reals, warmup=2)
        use_graph = True
        launch_note = "segments"
    one_step = graphed.replay if use_graph else eager_step

    for _ in range(args.warmup):
        one_step()
    torch.cuda.synchronize()
    if dp:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = one_step()
    t_enqueued = time.perf_counter() - t0                     # host time to enqueue the timed steps (before the final sync)
    torch.cuda.synchronize()
    if dp:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dp:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    finite = all(bool(torch.isfinite(out[k]).all()) for k in (f"d_loss{len(step.Ds) - 1}", "g_total"))
    if dp:
        # EXPOSED exchange: HIP events around every wait of a compute stream on the exchange (GradBuckets._join), over two more
        # steps in the launch mode of the timed region; max over ranks.  Host enqueue of the timed region per rank beside it.
        for bk in buckets:
            bk.measure = True
        for _ in range(2):
            one_step()
        exposed = sum(bk.exposed_ms() for bk in buckets) / 2.0
        for bk in buckets:
            bk.measure = False
        per_rank = [None] * dist.get_world_size()
        dist.all_gather_object(per_rank, (round(t_enqueued / args.steps * 1e3, 2), round(exposed, 3)))
        dp_info = dict(dp_info or {})
        dp_info.update({"exchange": step.g_buckets.mode, "bucket_mb": [round((e - s) * 4 / 2 ** 20, 1) for s, e in step.d_buckets[-1].bounds][:3],
                        "timed_host_enqueue_ms_per_step_per_rank": [p[0] for p in per_rank],
                        "exposed_exchange_ms_per_step_per_rank": [p[1] for p in per_rank],
                        "exposed_exchange_ms_per_step": max(p[1] for p in per_rank),
                        "exposed_exchange_note": "time the compute streams sat blocked on the gradient exchange (events around each wait), "
                                                 "2 untimed steps after the timed region, same launch mode"})
    # Per-launch kernel durations for the roofline object.  The timed region above overlaps the discriminator branches on
    # separate HIP streams (and, at N=1, may replay a captured graph with no host code between launches), so a launch's
    # event-to-event time there includes whatever runs beside it.  The same launches (same shapes, same kernels, same inputs) are
    # therefore timed in ROOF_STEPS instrumented eager steps on one stream, right after the timed region, in this process.
    ROOF_STEPS = 2
    overlap = step.overlap_discriminators, step.overlap_weight_gradients
    step.overlap_discriminators = step.overlap_weight_gradients = False
    timer.clear()
    timer.enabled = True
    for _ in range(ROOF_STEPS):
        step.step(words, sent, lens_dev, None, reals)
    torch.cuda.synchronize()
    timer.enabled = False
    step.overlap_discriminators, step.overlap_weight_gradients = overlap
    roofline_timing = ("HIP events recorded by the library on the launch stream immediately around each main conv kernel "
                       f"(agan_timer_arm), over {ROOF_STEPS} single-stream eager steps run right after the timed region; the timed "
                       "region itself overlaps streams / replays a HIP graph")
    if rank == 0 and args.layer_table:
        with open(args.layer_table, "w") as f:
            f.write(timer.layer_table(ROOF_STEPS) + "\n")
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        roofline, by_kernel = roofline_objects(timer, args, ms_per_step, ROOF_STEPS, roofline_timing)
        alg_gflop = sum(v[2] for v in timer.summary().values()) / ROOF_STEPS / 1e9
        launch = (("hip-graph segments + eager gradient exchange" if launch_note == "segments" else "hip-graph replay") if use_graph else "eager")
        line = {
            "metric": baseline_metric_name() if args.workload == "full3" else
                      f"train images/sec, {spec['desc'].split(':')[0]}, batch {args.batch}/GPU (BASELINE.json configs[{spec['config']}]; the headline metric is configs[2]'s)",
            "value": round(world * args.batch * args.steps / elapsed, 3),
            "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": DTYPE_NOTE[args.precision], "data": "synthetic",
            "activation_storage": args.storage,
            "config": {"workload": spec["desc"], "workload_key": args.workload,
                       "batch_per_gpu": args.batch, "global_batch": world * args.batch, "gf_dim": GF, "df_dim": DF, "emb_dim": EMB,
                       "seq_len": T,
                       "image_encoder": ("none (no DAMSM at this stage)" if step.image_encoder is None else
                                         "frozen stand-in plug-in (pool+projection): the timed step is the hot path of SURVEY §8d, which "
                                         "prices the third-party trunk separately" if args.image_encoder == "standin" else
                                         "frozen Inception-v3-shaped trunk (random weights) on stock MIOpen convs, fwd + dgrad in the timed step"),
                       "text_encoder": "bypassed (frozen; N(0,1) embeddings)", "parallelism": f"dp{world}",
                       "discriminator_streams": 1 if args.single_stream else len(step.Ds),
                       "launch": launch, "gradient_exchange": step.g_buckets.mode,
                       "algorithmic_conv_gflop_per_step": round(alg_gflop, 1),
                       "losses_finite": finite},
            "roofline": roofline,
            "roofline_by_kernel": by_kernel,
        }
        if dp_info:
            line["data_parallel"] = dp_info
        if world == 1 and not dp and not args.no_variants and args.workload == "full3":
            line["variants"] = side_measurements(args, dev, HF, LIB, step, words, sent, reals)
        if world == 1 and not args.no_cpu_baseline and args.workload == "full3":
            line["cpu_baseline"] = cpu_baseline(args.cpu_baseline_batch)
        sys.stdout.flush()
        os.write(report_fd, (json.dumps(line) + "\n").encode())
    if dp:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
