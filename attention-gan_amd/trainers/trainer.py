"""`ModelTrainer` helpers with the reference's names (trainers/trainer.py:13-127) and the GAN train step of
train.py:109-151 as a reusable object (`GanTrainStep`) with flat-buffer Adam and overlapped gradient all-reduce."""
from __future__ import annotations

import contextlib
import math
import os
from typing import Callable, Dict, List, Optional, Sequence

import torch
from torch import Tensor
from torch.nn import Module

from ..backend import functional as HF
from ..dataparallel import GradBuckets, any_rank, averaged_buffers, broadcast_module_, rank_of, world_size
from ..losses.disc_loss import NonSaturatingDiscLoss
from ..losses.gen_loss import NonSaturatingGenLoss
from ..losses.KL_loss import KL_loss
from ..losses.sentence_loss import SentenceLoss
from ..losses.words_loss import WordsLoss
from ..optim import FlatAdam


def _device() -> torch.device:
    return torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")


class ModelTrainer:
    """Label / noise / denormalise / checkpoint helpers (reference trainer.py:20-43,109-127)."""

    def __init__(self):
        pass

    def _make_match_labels(self, batch_size: int) -> Tensor:
        labels = torch.arange(batch_size, dtype=torch.int64, device=_device())
        # lets the DAMSM losses use the kernels' built-in arange targets (no copy, no readback); the tag carries the tensor's version
        # counter, so an in-place edit of the labels (labels[i] = j, labels.copy_(perm)) voids it and the values travel as given
        labels._agan_arange = (batch_size, labels._version)
        return labels

    def _count_parameters(self, model: Module):
        n = sum(p.numel() for p in model.parameters() if p.requires_grad)
        print(f"Model {model.__class__.__name__} has {n} parameters")
        return n

    def _make_noise(self, batch_size: int, z_dim: int) -> Tensor:
        return torch.randn(batch_size, z_dim, dtype=torch.float32, device=_device())

    def _make_mask(self, lengths, maxlen: Optional[int] = None) -> Tensor:
        """[[1]*len + [0]*(max-len)] int64 (train.py:96-100).  A device tensor of lengths is turned into the mask on the device
        (no host sync; `maxlen` = padded caption length, the reference's max(lengths))."""
        if isinstance(lengths, Tensor) and lengths.is_cuda:
            if maxlen is None:
                raise ValueError("_make_mask: maxlen is required for device-resident lengths")
            return (torch.arange(maxlen, device=lengths.device).view(1, -1) < lengths.view(-1, 1)).to(torch.int64)
        lens = [int(v) for v in (lengths.tolist() if hasattr(lengths, "tolist") else lengths)]
        mx = max(lens) if maxlen is None else int(maxlen)
        return torch.tensor([[1] * l + [0] * (mx - l) for l in lens], dtype=torch.int64, device=_device())

    def _denormalise_single(self, tensor: Tensor) -> Tensor:
        return tensor * 0.5 + 0.5

    def _denormalise_multiple(self, tensors: List[Tensor]) -> List[Tensor]:
        return [self._denormalise_single(t) for t in tensors]

    def _plot_history(self, *args, **kwargs) -> None:
        raise NotImplementedError("loss-curve plotting is outside the hot-path scope (SURVEY.md section 2 #7): the histories are "
                                  "plain lists of device scalars (g_losses, d_losses, damsm_losses)")

    def _image_grid(self, fake_images: List[Tensor]) -> List[Tensor]:
        """The reference's evaluation grid (trainers/trainer.py:68-98) as TENSORS: per resolution the first n*n images (n*n = the
        largest square number <= the batch, :76-78) tiled row-major into one [3, n*res, n*res] uint8 image (values clamped to [0, 1]
        like imshow does for float data).  No matplotlib: the caller decides where pixels go."""
        num = len(fake_images[0])
        square = next((i for i in range(num, 0, -1) if math.isqrt(i) ** 2 == i), 1)
        n = math.isqrt(square)
        grids = []
        for images in fake_images:
            c, h, w = images.shape[1:]
            g = images[:square].detach().float().clamp(0.0, 1.0).reshape(n, n, c, h, w).permute(2, 0, 3, 1, 4).reshape(c, n * h, n * w)
            grids.append((g * 255.0 + 0.5).to(torch.uint8))
        return grids

    def _plot_image_grid(self, fake_images: List[Tensor], epoch: Optional[int] = None, folder: str = 'generated_images') -> List[str]:
        """_image_grid written to {folder}/epoch_{e}-{res}x{res}.ppm (binary PPM: the reference writes PNGs through matplotlib,
        trainers/trainer.py:93-98; same file stems).  Returns the paths."""
        os.makedirs(folder, exist_ok=True)
        paths = []
        for g, images in zip(self._image_grid(fake_images), fake_images):
            res = images.shape[-1]
            stem = f"epoch_{epoch}-{res}x{res}" if epoch else f"_{res}x{res}"
            path = f"{folder}/{stem}.ppm"
            hwc = g.permute(1, 2, 0).contiguous().cpu().numpy()
            with open(path, "wb") as f:
                f.write(f"P6\n{hwc.shape[1]} {hwc.shape[0]}\n255\n".encode())
                f.write(hwc.tobytes())
            paths.append(path)
        return paths

    def _save_weights(self, modules: List, root_folder='saved_weights') -> None:
        """state_dict per module at {root}/{ClassName}.pkl.  Optimisers get an index suffix so that the reference's
        four `Adam` objects no longer overwrite one Adam.pkl (SURVEY.md §5 checkpoint note)."""
        os.makedirs(root_folder, exist_ok=True)
        seen: Dict[str, int] = {}
        for m in modules:
            name = m.__class__.__name__
            k = seen.get(name, 0)
            seen[name] = k + 1
            path = f"{root_folder}/{name}.pkl" if k == 0 else f"{root_folder}/{name}_{k}.pkl"
            torch.save(m.state_dict(), path)
            print(f'Module {name} weights saved to {path}')

    def _load_weights(self, modules: List[Module], root_folder='saved_weights') -> None:
        """Mirror of _save_weights: the k-th module of a class reads {ClassName}_{k}.pkl (k = 0: no suffix); nn.Modules are put
        in eval mode like the reference does (trainer.py:124), optimisers are not."""
        seen: Dict[str, int] = {}
        for m in modules:
            name = m.__class__.__name__
            k = seen.get(name, 0)
            seen[name] = k + 1
            path = f"{root_folder}/{name}.pkl" if k == 0 else f"{root_folder}/{name}_{k}.pkl"
            try:
                m.load_state_dict(torch.load(path, weights_only=True))
                if isinstance(m, Module):
                    m.eval()
                print(f'Module {name} weights loaded from {path}')
            except FileNotFoundError:
                print(f'FAILED: Module {name}... weights at path {path} were not found')


class GanTrainStep(ModelTrainer):
    """One batch of GanTrainer.train_gan (train.py:109-151): three discriminator updates, then one generator update.

    Results-identical departures from the script: fakes are detached in the D updates (the reference back-props
    through G there and discards the result at :132); D weight gradients are not produced in the G update.
    BatchNorm call order per D is kept: real batch, fake batch (D update), fake batch again (G update).
    """

    def __init__(self, generator: Module, discriminators: Sequence[Module], image_encoder: Optional[Callable] = None,
                 gen_lr: float = 2e-4, disc_lr: float = 2e-4, gamma1: float = 4.0, gamma2: float = 5.0, gamma3: float = 10.0,
                 wlambda: float = 5.0, slambda: float = 5.0, bucket_bytes: int = 64 << 20, group=None, seed: int = 0,
                 g_bucket_bytes: Optional[int] = None):
        super().__init__()
        self.G, self.Ds, self.image_encoder = generator, list(discriminators), image_encoder
        self.group = group
        for m in [self.G] + self.Ds:
            broadcast_module_(m, 0, group)
        # noise and the CA-net's eps come from a generator of this object, seeded per RANK: replicas share the initial weights
        # (broadcast above) but must draw different z / eps for their different captions
        dev0 = next(self.G.parameters()).device
        self.rank = rank_of(group)
        self.rng = torch.Generator(device=dev0)
        self.rng.manual_seed(int(seed) * 1000003 + self.rank)
        # sampling (generate_images) draws from a generator of its own: looking at samples mid-training must not shift the
        # training noise stream
        self.sample_rng = torch.Generator(device=dev0)
        self.sample_rng.manual_seed(int(seed) * 1000003 + 500009 + self.rank)
        self.g_opt = FlatAdam(self.G.parameters(), lr=gen_lr, betas=(0.5, 0.999))
        self.d_opts = [FlatAdam(d.parameters(), lr=disc_lr, betas=(0.5, 0.999)) for d in self.Ds]
        # the generator's 28 MB would be ONE 64 MB bucket, launched when its whole backward is done and fully exposed: it goes out in
        # 8 MB buckets instead, so all but the last travel under the ~10 ms of backward that remain
        self.g_buckets = GradBuckets(self.g_opt, min(bucket_bytes, 8 << 20) if g_bucket_bytes is None else g_bucket_bytes, group)
        self.d_buckets = [GradBuckets(o, bucket_bytes, group) for o in self.d_opts]
        # issue order of the three independent discriminator updates: LARGEST first (Disc256: 273 MB of gradients, the long pole of
        # the exchange), so that its buckets travel while the smaller discriminators still compute.  Results do not depend on it.
        self.d_order = sorted(range(len(self.Ds)), key=lambda i: -self.d_opts[i].numel)
        dev = next(self.G.parameters()).device
        self.words_loss = WordsLoss(dev, gamma1, gamma2, gamma3, wlambda)
        self.sent_loss = SentenceLoss(dev, gamma3, slambda)
        self.disc_loss, self.gen_loss = NonSaturatingDiscLoss(), NonSaturatingGenLoss()
        self.g_losses: List = []
        self.d_losses: List = []
        self.damsm_losses: List = []
        self.overlap_discriminators = True
        # optional: weight gradients on a side stream per compute stream (functional.set_wgrad_side_stream), joined by every
        # optimiser step.  Measured on MI355X at the metric config: 763 vs 778 images/s with it on -- the step is a sum of
        # kernels that each fill the chip, so the extra concurrency only adds cache pressure.  Off by default.
        self.overlap_weight_gradients = False
        self._streams: List = []
        # optional observer `f(tag, optimiser)` called right before each optimiser consumes its gradients (after the gradient
        # exchange), tag in {"D0", "D1", "D2", "G"}: tests read the flat gradient buffers through it
        self.on_gradients: Optional[Callable] = None

    def _d_streams(self, n: int, device) -> List:
        if len(self._streams) != n:
            self._streams = [torch.cuda.Stream(device=device) for _ in range(n)]
        return self._streams

    def step(self, word_embs: Tensor, sent_embs: Tensor, lengths, class_ids, real_imgs: Sequence[Tensor],
             noise: Optional[Tensor] = None, eps: Optional[Tensor] = None) -> Dict[str, Tensor]:
        if word_embs.is_cuda and not (isinstance(lengths, Tensor) and lengths.is_cuda):
            # host-side caption lengths (the reference's DataLoader hands over a list / CPU tensor): ONE pinned, non-blocking
            # copy up front -- a blocking H2D copy in the middle of the step drains the launch pipeline (~4 ms per step)
            host = torch.as_tensor(lengths, dtype=torch.int64).reshape(-1)
            lengths = host.pin_memory().to(word_embs.device, non_blocking=True)
        if word_embs.is_cuda:
            HF.amax_begin_step(word_embs.device)        # zeroed amax slots of this step (fp16 split mode only; one fill launch)
        prev_side = HF.set_wgrad_side_stream(self.overlap_weight_gradients and word_embs.is_cuda)
        try:
            return self._step(word_embs, sent_embs, lengths, class_ids, real_imgs, noise, eps)
        finally:
            HF.set_wgrad_side_stream(prev_side)

    def _step(self, word_embs, sent_embs, lengths, class_ids, real_imgs, noise, eps) -> Dict[str, Tensor]:
        b = word_embs.shape[0]
        labels = self._make_match_labels(b)
        mask = self._make_mask(lengths, word_embs.shape[2])
        if noise is None:
            noise = torch.randn(b, self.G.z_dim, dtype=torch.float32, device=word_embs.device, generator=self.rng)
        if eps is None:
            eps = torch.randn(b, self.G.cond_dim, dtype=torch.float32, device=word_embs.device, generator=self.rng)
        fakes, _attn, mu, logvar = self.G(noise, sent_embs, word_embs, mask, eps)
        out: Dict[str, Tensor] = {}
        # ---- discriminator updates (train.py:123-130) ----
        # The three updates are independent of each other (each D has its own weights, optimiser and images), so each runs on
        # its own HIP stream: the deep 4x4/8x8 layers of one discriminator (tiny grids) overlap the wide early layers of another.
        # Autograd replays each backward on the stream its forward ran on.
        main = torch.cuda.current_stream() if fakes[0].is_cuda else None
        streams = self._d_streams(len(self.Ds), fakes[0].device) if (main is not None and self.overlap_discriminators) else None
        if streams is not None:
            for st in streams:                      # discriminator weight gradients run on the (otherwise idle) main stream
                HF.set_side_stream_for(st, main)
        for i in self.d_order:
            d, opt, bk = self.Ds[i], self.d_opts[i], self.d_buckets[i]
            if streams is not None:
                streams[i].wait_stream(main)
                ctx = torch.cuda.stream(streams[i])
            else:
                ctx = contextlib.nullcontext()
            with ctx:
                opt.zero_grad()
                bk.arm()
                loss = self.disc_loss.get_loss(d, fakes[i].detach(), real_imgs[i])
                loss.backward()
                scale = bk.finish()
                if self.on_gradients is not None:
                    opt.join_and_rebind()
                    self.on_gradients(f"D{i}", opt)
                opt.step(scale)
                out[f"d_loss{i}"] = loss.detach()
        if streams is not None:
            for st in streams:
                main.wait_stream(st)
        # ---- generator update (train.py:132-151) ----
        self.g_opt.zero_grad()
        self.g_buckets.arm()
        for d in self.Ds:
            d.requires_grad_(False)
        # adversarial terms: one stream per discriminator again (forward here, autograd replays the dgrad chains on the same
        # streams); the DAMSM branch (image encoder + words/sentence loss) rides on the last discriminator's stream
        terms = []
        for i, d in enumerate(self.Ds):
            if streams is not None:
                streams[i].wait_stream(main)
                ctx = torch.cuda.stream(streams[i])
            else:
                ctx = contextlib.nullcontext()
            with ctx:
                gl = self.gen_loss.get_loss(d, fakes[i])
                terms.append(gl)
                out[f"g_loss{i}"] = gl.detach()
                if i == len(self.Ds) - 1 and self.image_encoder is not None:
                    regions, code = self.image_encoder(fakes[i])
                    wl, _ = self.words_loss.get_loss(regions, word_embs, labels, lengths, class_ids)
                    sl = self.sent_loss.get_loss(code, sent_embs, labels, class_ids)
                    terms += [wl, sl]
                    out["w_loss"], out["s_loss"] = wl.detach(), sl.detach()
        if streams is not None:
            for st in streams:
                main.wait_stream(st)
        total = terms[0]
        for t in terms[1:]:
            total = total + t
        kl = KL_loss(mu, logvar)
        total = total + kl
        out["kl"], out["g_total"] = kl.detach(), total.detach()
        total.backward()
        for d in self.Ds:
            d.requires_grad_(True)
        scale = self.g_buckets.finish()
        if self.on_gradients is not None:
            self.g_opt.join_and_rebind()
            self.on_gradients("G", self.g_opt)
        self.g_opt.step(scale)
        # loss histories stay device tensors: no .item() host sync inside the step (reference syncs at train.py:130,144-145)
        self.d_losses.append(out[f"d_loss{len(self.Ds) - 1}"])
        self.g_losses.append(out[f"g_loss{len(self.Ds) - 1}"])
        if "w_loss" in out:
            self.damsm_losses.append(out["w_loss"] + out["s_loss"])
        out["fake_imgs"] = [f.detach() for f in fakes]
        out["attn_maps"] = [a.detach() for a in _attn]
        out["mu"], out["logvar"] = mu.detach(), logvar.detach()
        return out

    # -- whole-step HIP graph ----------------------------------------------------------------------------------------
    def capture(self, word_embs: Tensor, sent_embs: Tensor, lengths: Tensor, real_imgs: Sequence[Tensor], warmup: int = 2,
                noise: Optional[Tensor] = None, eps: Optional[Tensor] = None) -> "GraphedStep":
        """Capture one full train step (all four optimiser updates, every stream) into a HIP graph.

        The arguments become the graph's STATIC input buffers: copy each new batch into them (`.copy_()`) and call `replay()`.
        `lengths` must be a device int64 tensor (no host reads inside the step); noise and the CA-net eps are drawn inside the
        graph (graph-safe Philox state) unless static `noise` / `eps` buffers are given; BatchNorm counters and the Adam step
        counters live in device memory, so replays keep advancing exactly like eager steps.  The warm-up steps are real
        training steps.  Single-process only: with world_size > 1 run the eager `step()`."""
        if self.g_buckets.active:
            raise RuntimeError("capture(): the all-reduce path is not captured; use step() under torch.distributed")
        if not (isinstance(lengths, Tensor) and lengths.is_cuda and lengths.dtype == torch.int64):
            raise ValueError("capture(): lengths must be an int64 device tensor")
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                      # warm-up off the default stream: tables, caches, allocator pools
            for _ in range(max(1, warmup)):
                self.step(word_embs, sent_embs, lengths, None, real_imgs, noise, eps)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        HF.build_pack_tables()                             # the batched re-pack's job tables: host work, not capturable
        graph = torch.cuda.CUDAGraph()
        if self.rng.device.type == "cuda":
            graph.register_generator_state(self.rng)       # noise / eps draws inside the graph advance this generator on replay
        with torch.cuda.graph(graph):
            out = self.step(word_embs, sent_embs, lengths, None, real_imgs, noise, eps)
        return GraphedStep(graph, out)

    # -- segment-wise HIP graphs for world > 1 --------------------------------------------------------------------------
    def capture_segments(self, word_embs: Tensor, sent_embs: Tensor, lengths: Tensor, real_imgs: Sequence[Tensor], warmup: int = 2,
                         noise: Optional[Tensor] = None, eps: Optional[Tensor] = None) -> "SegmentedStep":
        """The train step as NINE HIP graphs with the gradient exchange launched between them -- the form for world_size > 1, where
        the whole-step graph of capture() cannot be used (the RCCL collectives are not captured) and an eager step costs every rank
        ~11 ms of Python per 27 ms step on host cores the 8 ranks share:

            [G forward]  ->  per discriminator, largest first, each on its own stream:
                                 [D_i forward + backward]  -> all-reduce of D_i's flat gradient  ->  [Adam D_i]
                         ->  [G update: forward through the three D, DAMSM, KL, backward]  -> all-reduce of G's gradient  ->  [Adam G]

        A discriminator's exchange is issued when ITS backward graph has finished (not bucket by bucket inside it, as the eager
        step's hooks do), so it overlaps the backward graphs of the OTHER discriminators -- the largest (Disc256, 273 MB) goes
        first -- and nothing overlaps the generator's 28 MB.  The arguments become static input buffers exactly as for capture();
        the graphs share one memory pool and must be replayed in capture order (SegmentedStep.replay does).  Results are
        bit-identical to the eager step (tests/test_gpu_dataparallel.py, one-rank `nccl` rehearsal)."""
        if not (isinstance(lengths, Tensor) and lengths.is_cuda and lengths.dtype == torch.int64):
            raise ValueError("capture_segments(): lengths must be an int64 device tensor")
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                      # warm-up off the default stream: tables, caches, allocator pools
            for _ in range(max(1, warmup)):
                self.step(word_embs, sent_embs, lengths, None, real_imgs, noise, eps)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        HF.build_pack_tables()
        # memory pools: graphs that share a pool must never run concurrently (a later capture re-uses the temporaries an earlier one
        # freed).  The three discriminator updates replay CONCURRENTLY on their own streams, so each gets a private pool for its two
        # graphs; the generator's graphs, which run alone, share one.
        pool = torch.cuda.graph_pool_handle()
        d_pools = [torch.cuda.graph_pool_handle() for _ in self.Ds]
        # thread-local capture: the process group's watchdog thread polls its events (hipEventQuery) at any time, which a GLOBAL-mode
        # capture turns into "operation not permitted when stream is capturing" in THAT thread and kills the process (seen once in
        # the one-rank nccl rehearsal, round 3)
        mode = "thread_local"
        dev = word_embs.device
        n = len(self.Ds)
        seg = SegmentedStep(self)
        b = word_embs.shape[0]
        buckets = [self.g_buckets] + self.d_buckets
        armed = [bk._armed for bk in buckets]
        for bk in buckets:
            bk._armed = False                              # no hook-driven collectives inside a capture
        prev_side = HF.set_wgrad_side_stream(False)
        try:
            # ---- generator forward ----
            g0 = torch.cuda.CUDAGraph()
            if self.rng.device.type == "cuda":
                g0.register_generator_state(self.rng)
            with torch.cuda.graph(g0, pool=pool, capture_error_mode=mode):
                HF.amax_begin_step(dev)
                labels = self._make_match_labels(b)
                mask = self._make_mask(lengths, word_embs.shape[2])
                z = noise if noise is not None else torch.randn(b, self.G.z_dim, dtype=torch.float32, device=dev, generator=self.rng)
                e = eps if eps is not None else torch.randn(b, self.G.cond_dim, dtype=torch.float32, device=dev, generator=self.rng)
                fakes, attn, mu, logvar = self.G(z, sent_embs, word_embs, mask, e)
            seg.gen_forward = g0
            out: Dict[str, Tensor] = {}
            # ---- discriminator updates: backward graph, (exchange), Adam graph ----
            for i in self.d_order:
                d, opt = self.Ds[i], self.d_opts[i]
                gb = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gb, pool=d_pools[i], capture_error_mode=mode):
                    opt.zero_grad()
                    loss = self.disc_loss.get_loss(d, fakes[i].detach(), real_imgs[i])
                    loss.backward()
                    opt.join_and_rebind()
                    out[f"d_loss{i}"] = loss.detach()
                ga = torch.cuda.CUDAGraph()
                with torch.cuda.graph(ga, pool=d_pools[i], capture_error_mode=mode):
                    opt.step(1.0 / self.d_buckets[i].world if self.d_buckets[i].active else 1.0)
                seg.d_backward[i], seg.d_adam[i] = gb, ga
            # ---- generator update ----
            gg = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gg, pool=pool, capture_error_mode=mode):
                self.g_opt.zero_grad()
                for d in self.Ds:
                    d.requires_grad_(False)
                terms = []
                for i, d in enumerate(self.Ds):
                    gl = self.gen_loss.get_loss(d, fakes[i])
                    terms.append(gl)
                    out[f"g_loss{i}"] = gl.detach()
                    if i == n - 1 and self.image_encoder is not None:
                        regions, code = self.image_encoder(fakes[i])
                        wl, _ = self.words_loss.get_loss(regions, word_embs, labels, lengths, None)
                        sl = self.sent_loss.get_loss(code, sent_embs, labels, None)
                        terms += [wl, sl]
                        out["w_loss"], out["s_loss"] = wl.detach(), sl.detach()
                total = terms[0]
                for t in terms[1:]:
                    total = total + t
                kl = KL_loss(mu, logvar)
                total = total + kl
                out["kl"], out["g_total"] = kl.detach(), total.detach()
                total.backward()
                for d in self.Ds:
                    d.requires_grad_(True)
                self.g_opt.join_and_rebind()
            gadam = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gadam, pool=pool, capture_error_mode=mode):
                self.g_opt.step(1.0 / self.g_buckets.world if self.g_buckets.active else 1.0)
            seg.g_backward, seg.g_adam = gg, gadam
            out["fake_imgs"] = [f.detach() for f in fakes]
            out["attn_maps"] = [a.detach() for a in attn]
            out["mu"], out["logvar"] = mu.detach(), logvar.detach()
            seg.out = out
        finally:
            HF.set_wgrad_side_stream(prev_side)
            for bk, a in zip(buckets, armed):
                bk._armed = a
        return seg

    # -- checkpoint / resume (the reference only saves, and its four `Adam` objects share one Adam.pkl) --
    def state_dict(self, all_ranks: bool = True) -> Dict:
        """Checkpoint of the whole step: weights, BatchNorm buffers, the four optimisers and the noise generators' states (so a
        resumed run continues the z / eps sequence instead of replaying it from the seed).

        Under torch.distributed two things are PER REPLICA: the BatchNorm running statistics (local batches, as in the reference's
        single-process BN) and the noise generators (seeded seed * 1000003 + rank, so that replicas draw different z / eps).  With
        `all_ranks=True` (default) the checkpoint holds the MEAN of the running statistics over the ranks and the generator state of
        EVERY rank (`rng` / `sample_rng` = {rank: state}), so every rank writes the same file -- this is a COLLECTIVE: every rank of
        the group must call state_dict(), a lone `if rank == 0: save(step.state_dict())` would hang.  For that pattern pass
        `all_ranks=False`: no communication, the calling rank's own statistics and its own generator states ({rank: state} with one
        entry) are saved.  The live buffers and generators are never modified either way."""
        mine = {"rng": self.rng.get_state(), "sample_rng": self.sample_rng.get_state()}
        w = world_size(self.group)
        sd = {"generator": self.G.state_dict(), "discriminators": [d.state_dict() for d in self.Ds],
              "g_optim": self.g_opt.state_dict(), "d_optims": [o.state_dict() for o in self.d_opts],
              "rng": {self.rank: mine["rng"]}, "sample_rng": {self.rank: mine["sample_rng"]}, "rng_world": w}
        if all_ranks and w > 1:
            import torch.distributed as dist
            for m, msd in zip([self.G] + self.Ds, [sd["generator"]] + sd["discriminators"]):
                for k, v in averaged_buffers(m, self.group).items():
                    msd[k] = v
            gathered = [None] * w
            dist.all_gather_object(gathered, mine, group=self.group)
            for key in ("rng", "sample_rng"):
                sd[key] = {r: gathered[r][key].cpu() for r in range(w)}
        return sd

    @staticmethod
    def _restore_generator(gen: torch.Generator, saved, rank: int) -> None:
        """Put `gen` where rank `rank` of the checkpointed run left it.  `saved` is {rank: state} (or, from checkpoints written before
        the states were keyed by rank, one bare state = the saving rank's, by convention rank 0).  A rank the checkpoint has no state
        for (rank-0-only save, or a resume on more ranks) gets a state DERIVED from the saved one and its rank -- deterministic, and
        different on every rank, which is what the per-rank seeds are for (identical z / eps on all replicas would collapse the
        effective noise batch to one shard's)."""
        import hashlib
        if isinstance(saved, dict):
            if rank in saved:
                gen.set_state(saved[rank].cpu())
                return
            base = saved[min(saved)]
        else:
            if rank == 0:
                gen.set_state(saved.cpu())
                return
            base = saved
        digest = hashlib.sha256(base.cpu().numpy().tobytes() + int(rank).to_bytes(8, "little")).digest()
        gen.manual_seed(int.from_bytes(digest[:8], "little") & ((1 << 63) - 1))

    def load_state_dict(self, sd: Dict) -> None:
        self.G.load_state_dict(sd["generator"])
        self.g_opt.load_state_dict(sd["g_optim"])
        for d, o, ds, os_ in zip(self.Ds, self.d_opts, sd["discriminators"], sd["d_optims"]):
            d.load_state_dict(ds)
            o.load_state_dict(os_)
        # noise streams, per rank (checkpoints written before these were saved simply restart them from the seed)
        for key, gen in (("rng", self.rng), ("sample_rng", self.sample_rng)):
            if sd.get(key) is not None:
                self._restore_generator(gen, sd[key], self.rank)

    @torch.no_grad()
    def generate_images(self, word_embs: Tensor, sent_embs: Tensor, lengths, noise: Optional[Tensor] = None,
                        train_mode_bn: bool = False, eps: Optional[Tensor] = None) -> List[Tensor]:
        """Sampling path; images mapped from [-1,1] to [0,1].  Same kernels as training, forward only.

        train_mode_bn=False (default): test.py:77-87 -- eval-mode generator (running BatchNorm statistics).
        train_mode_bn=True: the EPOCH-END sample exactly as train.py:154-158 takes it -- under no_grad but WITHOUT .eval(): every
        BatchNorm normalises with the statistics of this batch and pushes them into its running statistics (one more update per
        epoch, which the next checkpoint carries), `noise` is the fixed input of train.py:105."""
        was_training = self.G.training
        self.G.train(bool(train_mode_bn))
        try:
            b = word_embs.shape[0]
            if noise is None:
                noise = torch.randn(b, self.G.z_dim, dtype=torch.float32, device=word_embs.device, generator=self.sample_rng)
            if eps is None:
                eps = torch.randn(b, self.G.cond_dim, dtype=torch.float32, device=word_embs.device, generator=self.sample_rng)
            fakes, _, _, _ = self.G(noise, sent_embs, word_embs, self._make_mask(lengths, word_embs.shape[2]), eps)
            return self._denormalise_multiple(fakes)
        finally:
            self.G.train(was_training)

    # -- the loop around the step (train.py:107-158) -------------------------------------------------------------------
    def skip_batch(self, lengths, batch_rows: int, batch_size: int) -> bool:
        """The reference's batch guard (train.py:112: `if min(lengths) < 2 or len(words) < BATCH_SIZE: continue`), decided for the
        whole data-parallel group: a rank that skipped on its own would leave the others waiting in the gradient all-reduce, so
        every rank contributes its local verdict to ONE small MAX all-reduce and all skip together if any shard fails the guard
        (dataparallel.any_rank).  A collective under torch.distributed: every rank calls it once per batch, in step."""
        lens = lengths.tolist() if hasattr(lengths, "tolist") else list(lengths)
        local = (min(int(v) for v in lens) < 2) if len(lens) else True
        local = local or int(batch_rows) < int(batch_size)
        return any_rank(local, self.group, next(self.G.parameters()).device)

    def train_epoch(self, batches, text_encoder: Callable, batch_size: int, max_batches: Optional[int] = None) -> int:
        """One pass over `batches` (tuples in the reference's wire format, data/bedrooms.py:229-236: words, lengths, class_ids,
        img64, img128, img256) as train.py:109-151 runs it: guard (skip_batch), frozen text encoder, step().  Under data parallelism
        every rank iterates ITS shard of the data with the same number of batches.  Returns the number of steps taken."""
        dev = next(self.G.parameters()).device
        steps = 0
        for n, batch in enumerate(batches):
            if max_batches is not None and n >= max_batches:
                break
            words, lengths, class_ids, img64, img128, img256 = batch
            if self.skip_batch(lengths, len(words), batch_size):
                continue
            with torch.no_grad():
                word_embs, sent_embs = text_encoder(words.to(dev), lengths)
            cids = class_ids.detach().cpu().numpy() if isinstance(class_ids, Tensor) else class_ids      # train.py:114
            self.step(word_embs.contiguous(), sent_embs.contiguous(), lengths, cids, [img64.to(dev), img128.to(dev), img256.to(dev)])
            steps += 1
        return steps


class GraphedStep:
    """A captured train step: `replay()` runs it; `out` holds the static result tensors (losses, fake images)."""

    def __init__(self, graph, out: Dict[str, Tensor]):
        self.graph, self.out = graph, out

    def replay(self) -> Dict[str, Tensor]:
        self.graph.replay()
        return self.out


class SegmentedStep:
    """The graphs of GanTrainStep.capture_segments and the eager gradient exchange between them; `replay()` runs one train step."""

    def __init__(self, trainer: "GanTrainStep"):
        self.t = trainer
        self.gen_forward = None
        self.d_backward: Dict[int, object] = {}
        self.d_adam: Dict[int, object] = {}
        self.g_backward = self.g_adam = None
        self.out: Dict[str, Tensor] = {}

    def replay(self) -> Dict[str, Tensor]:
        t = self.t
        main = torch.cuda.current_stream()
        self.gen_forward.replay()
        streams = t._d_streams(len(t.Ds), next(t.G.parameters()).device)
        for i in t.d_order:                                # largest discriminator first: its exchange travels under the others' graphs
            st = streams[i]
            st.wait_stream(main)
            with torch.cuda.stream(st):
                self.d_backward[i].replay()
                t.d_buckets[i].exchange_all()              # on the comm stream, behind this stream; this stream waits for it
                self.d_adam[i].replay()
        for st in streams:
            main.wait_stream(st)
        self.g_backward.replay()
        t.g_buckets.exchange_all()
        self.g_adam.replay()
        return self.out


class DAMSMTrainStep(ModelTrainer):
    """One batch of DAMSMTrainer.pretrain_damsm (pretrain_damsm.py:114-134): image encoder on the 256x256 image, text encoder
    on the captions, words + sentence loss (HIP kernels), clip_grad_norm_(RNN, 0.25), one Adam(lr 2e-3, betas (0.5, 0.999))
    over the RNN parameters and the image encoder's trainable heads.  The two encoders are stock PyTorch-ROCm modules."""

    def __init__(self, rnn: Module, cnn: Module, lr: float = 2e-3, rnn_grad_clip: float = 0.25, gamma1: float = 4.0,
                 gamma2: float = 5.0, gamma3: float = 10.0, wlambda: float = 5.0, slambda: float = 5.0):
        super().__init__()
        self.rnn, self.cnn, self.clip = rnn, cnn, rnn_grad_clip
        # ONE Adam over the RNN and the image encoder's trainable heads (pretrain_damsm.py:69-73), as one fused launch on a flat
        # buffer; the RNN's parameters come first so that clip_grad_norm_ (:133) sees them as one contiguous slice
        params = list(rnn.parameters()) + [p for p in cnn.parameters() if p.requires_grad]
        self.optim = FlatAdam(params, lr=lr, betas=(0.5, 0.999))
        self._rnn_count = sum(1 for p in rnn.parameters() if p.requires_grad)
        dev = next(rnn.parameters()).device
        self.words_loss = WordsLoss(dev, gamma1, gamma2, gamma3, wlambda)
        self.sent_loss = SentenceLoss(dev, gamma3, slambda)
        self.loss_history: List = []

    def step(self, captions: Tensor, lengths, class_ids, img256: Tensor) -> Dict[str, Tensor]:
        b = captions.shape[0]
        labels = self._make_match_labels(b)
        words_features, sent_code = self.cnn(img256)
        word_embs, sent_embs = self.rnn(captions, lengths)
        self.optim.zero_grad()
        wloss, _ = self.words_loss.get_loss(words_features, word_embs, labels, lengths, class_ids)
        sloss = self.sent_loss.get_loss(sent_code, sent_embs, labels, class_ids)
        loss = wloss + sloss
        loss.backward()
        self.optim.join_and_rebind()                       # every p.grad is now its slice of the flat gradient buffer
        torch.nn.utils.clip_grad_norm_(self.rnn.parameters(), self.clip)       # scales those slices in place (pretrain_damsm.py:133)
        self.optim.step()
        self.loss_history.append(loss.detach())
        return {"loss": loss.detach(), "w_loss": wloss.detach(), "s_loss": sloss.detach()}
