#!/usr/bin/env python3
"""AttnGAN stage-3 training throughput on MI355X (BASELINE.json metric: train images/s at 256x256, batch 24/GPU).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one full GanTrainer batch (train.py:109-151): G forward (3 stages + 2 word attentions), three discriminator
updates (real + fake batch each), the generator update through all three discriminators + DAMSM words/sentence loss + KL,
four fused Adam steps; under N > 1 every rank runs its own 24-image shard (weak scaling) and weight gradients are
all-reduced over RCCL.  Inputs are synthetic and already resident in HBM when the timed region starts.
Prints ONE JSON line on rank 0.
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
# dense MFMA peaks, /opt/skills/guides/MI355X_MICROARCH.md (spec): f32 = v_mfma_f32_32x32x2_f32; bf16x3 is priced against the
# bf16 peak although it issues three MFMAs per algorithmic product (its fraction can therefore not exceed 1/3)
MFMA_PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0, "f16": 2500.0, "bf16x3": 2500.0, "bf16x6": 2500.0, "f16x3": 2500.0}
# what the conv engine multiplies in, per --precision (fp32 storage and fp32 accumulate in every mode)
MFMA_PRODUCTS = {"f32": 1, "bf16": 1, "f16": 1, "bf16x3": 3, "bf16x6": 6, "f16x3": 3}
MFMA_PLANES = {"f32": 1, "bf16": 1, "f16": 1, "bf16x3": 2, "bf16x6": 3, "f16x3": 2}      # 16-bit planes of a packed weight
DTYPE_NOTE = {"f32": "f32", "bf16": "bf16", "f16": "f16", "bf16x3": "bf16x3", "bf16x6": "bf16x6", "f16x3": "f16x3"}


def algorithmic_conv_flops(kind, B, Cin, H, W, Cout, k):
    """2*MAC of the REFERENCE convolution: the upsample conv is priced at 9 taps on the 2x grid even
    though the folded kernel executes 4/9 of those MACs."""
    if kind == "same":
        return 2.0 * B * H * W * Cout * Cin * k * k
    if kind == "down":
        return 2.0 * B * (H // 2) * (W // 2) * Cout * Cin * 16
    if kind == "up":
        return 2.0 * B * (2 * H) * (2 * W) * Cout * Cin * 9
    raise ValueError(kind)


class ConvTimer:
    """HIP-event pairs recorded by the library itself on the launch stream, immediately around the main kernel of every
    conv-engine call (include/agan.h: agan_timer_arm) -- the slab-sum pass a split launch appends is outside the pair, so the
    per-kernel averages here are the ones `rocprofv3 --kernel-trace --stats` reports for the same kernels.  Durations are read
    after the instrumented region has been synchronised."""

    def __init__(self, lib, mode="f32"):
        self.lib, self.mode = lib, mode
        self.records, self.keys, self.enabled, self._pool = [], [], False, []

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
        executed = 2.0 * g.B * g.OH * g.OW * g.Cout * K          # MACs of the direct contraction (x2) ...
        direct = executed
        # algorithmic = the REFERENCE's contraction for this call: Upsample+conv3x3 is 9 taps on the 2x grid,
        # the folded kernels (fwd: 4 classes x 2x2 taps; dgrad: 4x4 s2; wgrad: 4 classes) execute 4/9 of that.
        algorithmic = direct * (9.0 / 4.0) if kind == "up" else direct
        if self.mode == "f32" and g.Cout > 4:
            # ... of which the fp32 mode's Winograd kernels issue 16/36 (conv3x3), 36/64 (conv4x4 s2 forward) or 9/16 (its data gradient).
            # (the conv calls on this path carry no bias / activation epilogue except the first discriminator conv, which has 3 input channels)
            executed = direct * float(self.lib.load().agan_conv_executed_fraction(ctypes.byref(g), 0, 1 if phase == "wgrad" else 0, 1))
        tile = "n128" if (g.Cout >= 96) else ("n64" if g.Cout >= 48 else "n32")
        if g.Cout <= 4:
            tile = "small_n"
        wmode = "f32" if self.mode == "bf16x6" else self.mode      # bf16x6 weight gradients run on the fp32 MFMA kernels
        name = f"conv_wgrad_{wmode}" if phase == "wgrad" else f"conv_gather_{self.mode}_{tile}"
        # algorithmic HBM bytes of this call: gathered tensor + produced tensor + weights, each moved once, in the storage types of
        # this call (fp32, or 16 bits under --storage; a weight gradient reads both activations and writes fp32 weights)
        if phase == "wgrad":
            nbytes = in_esz * g.B * g.Cin * g.IH * g.IW + out_esz * g.B * g.Cout * g.OH * g.OW + 4.0 * g.Cout * K * (g.OS * g.OS)
        else:
            wesz = 4.0 if self.mode == "f32" else 2.0 * MFMA_PLANES.get(self.mode, 1)
            nbytes = in_esz * g.B * g.Cin * g.IH * g.IW + out_esz * g.B * g.Cout * g.OH * g.OW + wesz * g.Cout * K * (g.OS * g.OS)
        e0, e1 = self._event(), self._event()
        self.lib.call("agan_timer_arm", e0, e1)
        self.records.append((name, algorithmic, executed, nbytes, e0, e1))
        self.keys.append(f"{phase:5s} {kind:4s} B{g.B} {g.Cin:4d}x{g.IH:<3d} -> {g.Cout:4d}x{g.OH:<3d} taps {g.R}x{g.S} cls {g.OS * g.OS}")

    def end(self):
        pass

    def _ms(self, e0, e1):
        ms = ctypes.c_float()
        self.lib.call("agan_timer_elapsed_ms", e0, e1, ctypes.byref(ms))
        return ms.value

    def clear(self):
        for rec in self.records:
            self._pool += [rec[4], rec[5]]
        self.records, self.keys = [], []

    def layer_table(self, steps):
        """per (phase, layer shape): launches and ms per step, executed TFLOP/s -- the list the next optimisation is read from"""
        by = {}
        for key, (name, fa, fe, nb, e0, e1) in zip(self.keys, self.records):
            d = by.setdefault(key, [0, 0.0, 0.0])
            d[0] += 1; d[1] += self._ms(e0, e1); d[2] += fe
        rows = sorted(by.items(), key=lambda kv: -kv[1][1])
        out = [f"{'phase kind shape':58s} {'n/step':>6s} {'ms/step':>8s} {'exec TF/s':>9s}"]
        for key, (n, ms, fe) in rows:
            out.append(f"{key:58s} {n / steps:6.1f} {ms / steps:8.3f} {fe / (ms * 1e-3) / 1e12:9.1f}")
        out.append(f"{'total':58s} {sum(v[0] for v in by.values()) / steps:6.1f} {sum(v[1] for v in by.values()) / steps:8.3f}")
        return "\n".join(out)

    def summary(self):
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


def measured_traffic(kernel, mode, storage="f32"):
    """HBM bytes per launch from the COMMITTED PMC passes, not from this run (profiles/r03_<mode>[_s16]_traffic.json, made by
    profiles/make_counters.py from separate FETCH_SIZE / WRITE_SIZE runs of this same command; gfx950 x2 read correction);
    None if that kernel was not measured."""
    s16 = "_s16" if storage != "f32" else ""
    for name in (f"r03_{mode}{s16}_traffic.json", f"r02_{mode}_traffic.json", "r01_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                t = json.load(f)["kernels"].get(kernel, {}).get("traffic")
            if t is not None:
                return t
        except (OSError, ValueError, KeyError):
            continue
    return None


def build(dev, batch, HF, encoder="standin"):
    GEN = importlib.import_module("attention-gan_amd.networks.generator")
    DISC = importlib.import_module("attention-gan_amd.networks.discriminators")
    ENC = importlib.import_module("attention-gan_amd.networks.cnn_encoder")
    TR = importlib.import_module("attention-gan_amd.trainers.trainer")
    torch.manual_seed(0)      # identical initial weights on every rank (and broadcast from rank 0 on top)
    G = GEN.Generator(GF, EMB, Z, COND).to(dev)
    Ds = [DISC.Disc64(DF).to(dev), DISC.Disc128(DF).to(dev), DISC.Disc256(DF).to(dev)]
    enc = (ENC.StandInImageEncoder(EMB) if encoder == "standin" else ENC.CNNEncoder(EMB)).to(dev)
    enc.freeze_all_weights()
    enc.eval()                     # the reference loads it with _load_weights(), which puts it in eval mode (trainer.py:124)
    # (AGAN_BUCKET_MB: gradient all-reduce bucket size for scaling experiments; default = the trainer's 64 MB)
    mb = os.environ.get("AGAN_BUCKET_MB")
    step = TR.GanTrainStep(G, Ds, enc, bucket_bytes=int(mb) << 20) if mb else TR.GanTrainStep(G, Ds, enc)
    step.overlap_weight_gradients = os.environ.get("AGAN_WGRAD_SIDE_STREAM", "0") == "1"      # (A/B switch)
    return step


def synthetic_batch(dev, batch, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    words = torch.randn(batch, EMB, T, generator=g).to(dev)             # frozen RNN bypassed: N(0,1) embeddings
    sent = torch.randn(batch, EMB, generator=g).to(dev)
    lens = [T] * batch
    reals = [(torch.rand(batch, 3, r, r, generator=g) * 2 - 1).to(dev) for r in (64, 128, 256)]
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


def side_measurements(args, dev, HF, LIB, step, words, sent, reals):
    """Short eager measurements reported BESIDE the headline, never inside it or its roofline:
    random caption lengths 2..10 (the headline uses full-length captions), the other arithmetic modes of the conv engine, and the
    end-to-end step with the Inception-v3-shaped trunk (stock MIOpen convs, random weights) as the DAMSM image encoder and the
    bi-LSTM text encoder run on the device every step."""
    # the SAME step and warm-up counts as the headline: a `--precision X` headline and the `precision_X` variant of another run are
    # then the same measurement (8-step variants read 6 % low in round 2: clocks and allocator pools had not settled)
    n, w = args.steps, max(args.warmup, 2)

    def rate(fn):
        for _ in range(w):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return round(args.batch * n / (time.perf_counter() - t0), 1)
    out = {"note": f"{n} eager steps each after {w} warm-up steps (the headline's counts), same batch; images/s"}
    B = args.batch
    g = torch.Generator().manual_seed(99)
    lens_r = torch.randint(2, T + 1, (B,), generator=g)
    lens_r[0] = T
    lens_r = lens_r.to(dev)
    out["random_caption_lengths_2_10"] = rate(lambda: step.step(words, sent, lens_r, None, reals))
    lens_full = torch.full((B,), T, dtype=torch.int64, device=dev)
    for other in ("f32", "bf16x6", "f16x3", "bf16"):      # fp32-grade split modes (parity-tested at this size) and plain bf16
        if other == args.precision:
            continue
        HF.set_precision(LIB.PRECISIONS[other])
        try:
            out[f"precision_{other}"] = rate(lambda: step.step(words, sent, lens_full, None, reals))
            if other == "bf16" and args.storage == "f32":
                HF.set_activation_storage("bf16")
                out["precision_bf16_storage_bf16"] = rate(lambda: step.step(words, sent, lens_full, None, reals))
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
        out["end_to_end_inception_trunk_plus_lstm"] = rate(e2e)
    except Exception as exc:          # the side line must never cost the headline
        out["end_to_end_inception_trunk_plus_lstm"] = f"failed: {type(exc).__name__}: {exc}"
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=120, help="timed steps (default: a >= 3 s timed region at the metric config)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=24, help="images per GPU (metric: 24)")
    ap.add_argument("--precision", choices=["f32", "bf16x6", "f16x3", "bf16x3", "bf16", "f16"], default="f32",
                    help="MFMA mode of the conv engine: f32 = exact fp32 products (v_mfma_f32_32x32x2_f32); bf16x6 = three bf16 planes, "
                         "6 MFMAs per product (fp32-grade); bf16x3 = two planes, 3 MFMAs; bf16 / f16 = operands rounded to 16 bits "
                         "(BASELINE configs[1] / configs[4] arithmetic)")
    ap.add_argument("--storage", choices=["f32", "bf16", "f16"], default="f32",
                    help="storage type of activations in HBM: f32 (the reference's layout, default) or the operand type of the one-plane "
                         "16-bit modes (--precision bf16 --storage bf16 / --precision f16 --storage f16): conv outputs, BatchNorm "
                         "inputs / outputs and their gradients are rounded once when stored and gathered without conversion")
    ap.add_argument("--image-encoder", choices=["standin", "inception"], default="standin",
                    help="frozen DAMSM image encoder plug-in: 'standin' = contract-only stub (SURVEY §8d prices the hot path without "
                         "the third-party trunk); 'inception' = Inception-v3-shaped trunk on stock MIOpen convs, random weights")
    ap.add_argument("--graph", choices=["auto", "on", "off", "segments"], default="auto",
                    help="launch mode of the step: captured HIP graph replay or eager (auto: at 1 GPU probe both in the untimed warm-up and keep "
                         "the faster; eager under torch.distributed).  segments = nine HIP graphs with the gradient exchange launched "
                         "between them (GanTrainStep.capture_segments): the graph form that also works under torch.distributed")
    ap.add_argument("--single-stream", action="store_true",
                    help="run the three discriminator updates on ONE stream (profiling aid: per-kernel PMC counters of a rocprofv3 pass are "
                         "diluted when kernels of different streams share the chip; the headline run overlaps them)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true",
                    help="skip the short side measurements reported beside the headline (random caption lengths, the fp32-MFMA mode, "
                         "the end-to-end step with the Inception-shaped trunk and the LSTM text encoder)")
    ap.add_argument("--cpu-baseline-batch", type=int, default=24)
    ap.add_argument("--layer-table", default=None, help="write the per-layer conv timing table of the instrumented steps to this file")
    args = ap.parse_args()

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
    HF.set_precision(LIB.PRECISIONS[args.precision])
    if args.storage != "f32":
        if args.storage != args.precision:
            raise SystemExit(f"--storage {args.storage} needs --precision {args.storage} (the storage type is the MFMA operand type)")
        HF.set_activation_storage(args.storage)
    step = build(dev, args.batch, HF, args.image_encoder)
    if args.single_stream:
        step.overlap_discriminators = False
    words, sent, lens, reals = synthetic_batch(dev, args.batch, seed=1234 + rank)
    timer = ConvTimer(importlib.import_module("attention-gan_amd.backend.lib"), args.precision)
    HF.set_launch_observer(timer)

    # caption lengths live on the device like the rest of the batch (inputs are resident in HBM when the timed region starts):
    # a host list would cost two small blocking H2D copies per step, which stall the eager launch pipeline by ~4 ms
    lens_dev = torch.tensor(lens, dtype=torch.int64, device=dev)

    def eager_step():
        return step.step(words, sent, lens_dev, None, reals)

    # Launch mode.  Eager: ~850 launches per step from Python (~12 ms of host time, hidden behind ~27 ms of GPU time on an
    # unloaded host).  Graph: the whole step captured once and replayed -- no host work, but hipGraphLaunch adds per-node cost
    # and a process that has captured runs ~1.5 % slower even eagerly (857 vs 877 images/s measured).  `auto` at N=1 therefore
    # first checks, in UNTIMED steps, whether eager is launch-bound on this host (enqueue time vs step time); only then does it
    # capture the graph and keep whichever replays/steps faster.  Under torch.distributed the step stays eager (RCCL is not
    # captured).
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
    if not dp and args.graph == "auto":
        for _ in range(2):
            eager_step()
        t_host, t_eager = probe(eager_step)
        if t_host > 0.85 * t_eager:                      # the host cannot keep the GPU fed: try the graph
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
    torch.cuda.synchronize()
    if dp:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dp:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    finite = all(bool(torch.isfinite(out[k]).all()) for k in ("d_loss2", "g_total"))
    # Per-launch kernel durations for the roofline object.  The timed region above overlaps the three discriminator branches on
    # separate HIP streams (and, at N=1, replays a captured graph with no host code between launches), so a launch's
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
        summ = timer.summary()
        dom = max(summ.items(), key=lambda kv: kv[1][1]) if summ else None
        roofline = None
        if dom:
            name, (n, ms, falg, fexec, nbytes) = dom
            achieved = falg / (ms * 1e-3) / 1e12
            peak = MFMA_PEAK_TFLOPS[args.precision]
            roofline = {"bound": "mfma", "kernel": name, "achieved": round(achieved, 2), "peak": peak,
                        "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": measured_traffic(name, args.precision, args.storage),
                        "traffic_source": "profiles/ (committed rocprofv3 PMC passes of this command), not this run",
                        "algorithmic_bytes_per_launch": round(nbytes / n),
                        "launches": n, "avg_launch_ms": round(ms / n, 4), "timing": roofline_timing, "executed_tflops": round(fexec / (ms * 1e-3) / 1e12, 2),
                        "share_of_step_time": round((ms / ROOF_STEPS) / (elapsed / args.steps * 1e3), 3)}
            products = MFMA_PRODUCTS[args.precision]
            if products > 1:
                # a split-precision mode issues `products` MFMAs per algorithmic multiply-add: the matrix pipe's own rate
                roofline["mfma_products_per_multiply"] = products
                roofline["mfma_issue_tflops"] = round(products * fexec / (ms * 1e-3) / 1e12, 1)
                roofline["mfma_issue_frac"] = round(products * fexec / (ms * 1e-3) / 1e12 / peak, 4)
        line = {
            "metric": baseline_metric_name(),
            "value": round(world * args.batch * args.steps / elapsed, 3),
            "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": DTYPE_NOTE[args.precision], "data": "synthetic",
            "activation_storage": args.storage,
            "config": {"workload": "full 3-stage AttnGAN 64->128->256 train step: G + 3xD updates + word attention + DAMSM words/sentence "
                                   "loss + KL + 4x fused Adam (BASELINE.json configs[2])",
                       "batch_per_gpu": args.batch, "global_batch": world * args.batch, "gf_dim": GF, "df_dim": DF, "emb_dim": EMB,
                       "seq_len": T, "image_encoder": ("frozen stand-in plug-in (pool+projection): the timed step is the hot path of SURVEY §8d, which "
                                                       "prices the third-party trunk separately" if args.image_encoder == "standin" else
                                                       "frozen Inception-v3-shaped trunk (random weights) on stock MIOpen convs, fwd + dgrad in the timed step"),
                       "text_encoder": "bypassed (frozen; N(0,1) embeddings)", "parallelism": f"dp{world}", "discriminator_streams": 1 if args.single_stream else 3,
                       "launch": ("hip-graph segments + eager gradient exchange" if args.graph == "segments" else "hip-graph replay") if use_graph else "eager",
                       "losses_finite": finite},
            "roofline": roofline,
        }
        if world == 1 and not args.no_variants:
            line["variants"] = side_measurements(args, dev, HF, LIB, step, words, sent, reals)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.cpu_baseline_batch)
        sys.stdout.flush()
        os.write(report_fd, (json.dumps(line) + "\n").encode())
    if dp:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
