"""Data-parallel path on real kernels: two ranks (gloo rendezvous, both on the one GPU of the test box -- RCCL needs one
device per rank, the driver exercises that at N=2..8) run GanTrainStep on different shards; the result must equal the
single-process computation  mean over shards of (per-shard gradients with per-shard BatchNorm statistics) -> one Adam step."""
import importlib
import os

import pytest
import torch

from helpers import collect_from_children, free_port

pytestmark = pytest.mark.gpu

GEN = importlib.import_module("attention-gan_amd.networks.generator")
DISC = importlib.import_module("attention-gan_amd.networks.discriminators")
ENC = importlib.import_module("attention-gan_amd.networks.cnn_encoder")
TR = importlib.import_module("attention-gan_amd.trainers.trainer")
OPT = importlib.import_module("attention-gan_amd.optim")

DIMS = dict(gf=4, df=4, emb=16, z=8, cond=8, B=4, T=5)


def _nets(dev):
    torch.manual_seed(123)
    d = DIMS
    G = GEN.Generator(d["gf"], d["emb"], d["z"], d["cond"]).to(dev)
    Ds = [DISC.Disc64(d["df"]).to(dev), DISC.Disc128(d["df"]).to(dev), DISC.Disc256(d["df"]).to(dev)]
    enc = ENC.StandInImageEncoder(d["emb"]).to(dev)
    enc.freeze_all_weights()
    return G, Ds, enc


def _shard(rank, dev):
    d = DIMS
    g = torch.Generator().manual_seed(1000 + rank)
    words, sent = torch.randn(d["B"], d["emb"], d["T"], generator=g).to(dev), torch.randn(d["B"], d["emb"], generator=g).to(dev)
    noise, eps = torch.randn(d["B"], d["z"], generator=g).to(dev), torch.randn(d["B"], d["cond"], generator=g).to(dev)
    reals = [(torch.rand(d["B"], 3, r, r, generator=g) * 2 - 1).to(dev) for r in (64, 128, 256)]
    return words, sent, [5, 3, 2, 4], reals, noise, eps


def _worker(rank, world, port, q, bf16=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if bf16:
        os.environ["AGAN_DP_BF16"] = "1"        # buckets travel as bf16, fp32 accumulation on arrival (dataparallel.all_reduce_bf16_)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        G, Ds, enc = _nets(dev)
        step = TR.GanTrainStep(G, Ds, enc, bucket_bytes=64 << 10)          # small buckets -> several exchanges per optimiser
        words, sent, lens, reals, noise, eps = _shard(rank, dev)
        assert step.overlap_discriminators                                   # three D streams, each with its own exchange stream
        out = step.step(words, sent, lens, None, reals, noise, eps)
        torch.cuda.synchronize()
        g_sd = {k: v.detach().cpu().numpy() for k, v in G.state_dict().items()}      # BEFORE the checkpoint-time buffer sync
        d_sd = [{k: v.detach().cpu().numpy() for k, v in d.state_dict().items()} for d in Ds]
        draw = torch.randn(4, device=dev, generator=step.rng).cpu().numpy()          # the per-rank noise generator
        ckpt = step.state_dict()                                                    # averages BatchNorm statistics over the ranks
        bufs = {k: v.detach().cpu().numpy() for k, v in ckpt["generator"].items() if "running" in k}
        import numpy as np
        for k, v in G.state_dict().items():                                          # the checkpoint left the LIVE statistics alone
            if "running" in k:
                assert np.array_equal(v.detach().cpu().numpy(), g_sd[k]), k
        if rank == 0:                                                               # the rank-0-only pattern: no collective, no hang
            solo = step.state_dict(all_ranks=False)
            assert all(np.array_equal(v.detach().cpu().numpy(), g_sd[k]) for k, v in solo["generator"].items() if "running" in k)
        # numpy (pickled by value): torch tensors would travel as shared-memory handles that die with this process
        q.put((rank, g_sd, d_sd, len(step.g_buckets.bounds), draw, bufs))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("bf16", [False, True], ids=["fp32-exchange", "bf16-exchange"])
def test_two_rank_step_equals_mean_of_shard_gradients(bf16):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, bf16)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(collect_from_children(q, procs), key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    import numpy as np
    draws, ckpt_bufs = [t[4] for t in got], [t[5] for t in got]
    assert not np.array_equal(draws[0], draws[1])                     # replicas draw different z / eps (seed + rank)
    differ = sum(not np.array_equal(got[0][1][k], got[1][1][k]) for k in got[0][1] if "running" in k)
    assert differ > 0                                                 # per-replica BatchNorm statistics (local batches) ...
    for k in ckpt_bufs[0]:                                            # ... are averaged into ONE checkpoint, identical on every rank
        assert np.array_equal(ckpt_bufs[0][k], ckpt_bufs[1][k]), k
        assert np.allclose(ckpt_bufs[0][k], 0.5 * (got[0][1][k] + got[1][1][k]), rtol=1e-6, atol=1e-7), k
    got = [(r, {k: torch.from_numpy(v) for k, v in g.items()}, [{k: torch.from_numpy(v) for k, v in d.items()} for d in ds], nb)
           for r, g, ds, nb, _, _ in got]
    assert got[0][3] > 1                                              # more than one gradient bucket was exchanged
    for k, v in got[0][1].items():                                    # replicas end the step with identical weights
        if k.endswith((".weight", ".bias")):
            assert torch.equal(v, got[1][1][k]), k

    # single-process reference: per-shard gradients (per-shard BN statistics), averaged, one Adam step each
    dev = torch.device("cuda", 0)
    G, Ds, enc = _nets(dev)
    g_opt = OPT.FlatAdam(G.parameters(), lr=2e-4, betas=(0.5, 0.999))
    d_opts = [OPT.FlatAdam(d.parameters(), lr=2e-4, betas=(0.5, 0.999)) for d in Ds]
    ref = TR.GanTrainStep.__new__(TR.GanTrainStep)                     # borrow the loss helpers without re-homing twice
    TR.ModelTrainer.__init__(ref)
    WL = importlib.import_module("attention-gan_amd.losses.words_loss").WordsLoss(dev, 4.0, 5.0, 10.0, 5.0)
    SL = importlib.import_module("attention-gan_amd.losses.sentence_loss").SentenceLoss(dev, 10.0, 5.0)
    DL = importlib.import_module("attention-gan_amd.losses.disc_loss").NonSaturatingDiscLoss()
    GL = importlib.import_module("attention-gan_amd.losses.gen_loss").NonSaturatingGenLoss()
    KL = importlib.import_module("attention-gan_amd.losses.KL_loss").KL_loss
    shards = [_shard(r, dev) for r in range(2)]
    bn_state = [{k: v.clone() for k, v in m.state_dict().items() if "running" in k or "num_batches" in k} for m in [G] + Ds]
    fakes_all = []
    for words, sent, lens, reals, noise, eps in shards:               # generator forward per shard (its BN stats are per shard)
        G.load_state_dict(bn_state[0], strict=False)
        fakes_all.append(G(noise, sent, words, ref._make_mask(lens), eps))
    for i, (d, opt) in enumerate(zip(Ds, d_opts)):
        acc = torch.zeros_like(opt.grad)
        for s, (words, sent, lens, reals, noise, eps) in enumerate(shards):
            d.load_state_dict(bn_state[1 + i], strict=False)
            opt.zero_grad()
            DL.get_loss(d, fakes_all[s][0][i].detach(), reals[i]).backward()
            opt._rebind()
            acc += opt.grad
        opt.grad.copy_(acc)
        opt.step(0.5)
    g_acc = torch.zeros_like(g_opt.grad)
    for d in Ds:
        d.requires_grad_(False)
    d_bn_after = [{k: v.clone() for k, v in d.state_dict().items() if "running" in k or "num_batches" in k} for d in Ds]
    for s, (words, sent, lens, reals, noise, eps) in enumerate(shards):
        fakes, _, mu, logvar = fakes_all[s]
        g_opt.zero_grad()
        total = KL(mu, logvar)
        for i, d in enumerate(Ds):
            d.load_state_dict(d_bn_after[i], strict=False)
            total = total + GL.get_loss(d, fakes[i])
        regions, code = enc(fakes[2])
        labels = torch.arange(DIMS["B"], device=dev)
        total = total + WL.get_loss(regions, words, labels, lens, None)[0] + SL.get_loss(code, sent, labels, None)
        total.backward()
        g_opt._rebind()
        g_acc += g_opt.grad
    g_opt.grad.copy_(g_acc)
    g_opt.step(0.5)
    torch.cuda.synchronize()

    def close(a, b, what):
        scale = float(b.abs().max().clamp(min=1e-30))
        diff = (a.to(b.device) - b).abs()
        bad = int((diff > 1e-3 * scale).sum())
        # (Adam's first step is sign-like: a weight whose averaged gradient is within rounding of zero moves by 2 lr the other way.
        #  The bf16 exchange rounds the gradients twice at 2^-9: a few more such weights, same 2 lr bound)
        allowed = max(4, b.numel() // 200) if bf16 else max(2, b.numel() // 1000)
        assert float(diff.max()) <= 2.05 * 2e-4 and bad <= allowed, f"{what}: {bad} of {b.numel()} off, max {float(diff.max()):.2e}"

    for k, v in G.state_dict().items():
        if k.endswith((".weight", ".bias")):
            close(got[0][1][k], v.cpu(), f"G {k}")
    for i, d in enumerate(Ds):
        for k, v in d.state_dict().items():
            if k.endswith((".weight", ".bias")):
                close(got[0][2][i][k], v.cpu(), f"D{i} {k}")


def _nccl_one_rank_worker(port, q):
    """world of ONE rank on the `nccl` (= RCCL) backend with the exchange machinery forced on: hooks, three discriminator streams,
    comm streams, async handles and the bucket joins run exactly as at N > 1; a SUM all-reduce over one rank is the identity."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), AGAN_DP_FORCE="1")
    import torch.distributed as dist
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        G, Ds, enc = _nets(dev)
        step = TR.GanTrainStep(G, Ds, enc, bucket_bytes=64 << 10)
        assert step.g_buckets.active and len(step.g_buckets.bounds) > 1
        words, sent, lens, reals, noise, eps = _shard(0, dev)
        for _ in range(2):
            out = step.step(words, sent, lens, None, reals, noise, eps)
        torch.cuda.synchronize()
        q.put(({k: v.detach().cpu().numpy() for k, v in G.state_dict().items()}, float(out["g_total"])))
    finally:
        dist.destroy_process_group()


def test_one_rank_nccl_exchange_is_identity():
    import numpy as np
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_one_rank_worker, args=(free_port(), q))
    p.start()
    (g_sd, g_total), = collect_from_children(q, [p])
    p.join(60)
    assert p.exitcode == 0
    # the same two steps without torch.distributed: bitwise the same weights (the exchange added nothing and reordered nothing)
    dev = torch.device("cuda", 0)
    G, Ds, enc = _nets(dev)
    step = TR.GanTrainStep(G, Ds, enc)
    words, sent, lens, reals, noise, eps = _shard(0, dev)
    for _ in range(2):
        out = step.step(words, sent, lens, None, reals, noise, eps)
    torch.cuda.synchronize()
    assert float(out["g_total"]) == g_total
    for k, v in G.state_dict().items():
        assert np.array_equal(v.detach().cpu().numpy(), g_sd[k]), k


def _direct_rccl_worker(port, q):
    """the same one-rank rehearsal with the exchange driven through the C ABI (agan_comm_init / agan_allreduce_bucket: reduce-scatter +
    all-gather per bucket on the comm stream) instead of torch.distributed's all_reduce"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), AGAN_DP_FORCE="1", AGAN_RCCL_DIRECT="1")
    import torch.distributed as dist
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=0, world_size=1)          # only carries the communicator id
    try:
        G, Ds, enc = _nets(dev)
        step = TR.GanTrainStep(G, Ds, enc, bucket_bytes=64 << 10)
        assert step.g_buckets.direct is not None and len(step.g_buckets.bounds) > 1
        words, sent, lens, reals, noise, eps = _shard(0, dev)
        for _ in range(2):
            out = step.step(words, sent, lens, None, reals, noise, eps)
        torch.cuda.synchronize()
        q.put(({k: v.detach().cpu().numpy() for k, v in G.state_dict().items()}, float(out["g_total"])))
        step.g_buckets.direct.close()
    finally:
        dist.destroy_process_group()


def test_one_rank_direct_rccl_exchange_is_identity():
    import numpy as np
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_direct_rccl_worker, args=(free_port(), q))
    p.start()
    (g_sd, g_total), = collect_from_children(q, [p])
    p.join(60)
    assert p.exitcode == 0
    dev = torch.device("cuda", 0)
    G, Ds, enc = _nets(dev)
    step = TR.GanTrainStep(G, Ds, enc)
    words, sent, lens, reals, noise, eps = _shard(0, dev)
    for _ in range(2):
        out = step.step(words, sent, lens, None, reals, noise, eps)
    torch.cuda.synchronize()
    assert float(out["g_total"]) == g_total
    for k, v in G.state_dict().items():
        assert np.array_equal(v.detach().cpu().numpy(), g_sd[k]), k


def _segmented_worker(port, q, force_dp):
    """capture_segments(): nine HIP graphs with the gradient exchange launched eagerly between them (the launch form for
    world > 1), here in a world of ONE `nccl` rank with the exchange machinery forced on -- or with no process group at all."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    if force_dp:
        os.environ["AGAN_DP_FORCE"] = "1"
        import torch.distributed as dist
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        G, Ds, enc = _nets(dev)
        step = TR.GanTrainStep(G, Ds, enc, bucket_bytes=64 << 10)
        assert step.g_buckets.active == bool(force_dp)
        words, sent, lens, reals, noise, eps = _shard(0, dev)
        lens_dev = torch.tensor(lens, dtype=torch.int64, device=dev)
        seg = step.capture_segments(words, sent, lens_dev, reals, warmup=1, noise=noise, eps=eps)      # (the warm-up is step 1)
        assert len(seg.d_backward) == 3 and len(seg.d_adam) == 3
        for _ in range(2):
            out = seg.replay()
        torch.cuda.synchronize()
        q.put(({k: v.detach().cpu().numpy() for k, v in G.state_dict().items()},
               [{k: v.detach().cpu().numpy() for k, v in d.state_dict().items()} for d in Ds], float(out["g_total"]), float(out["d_loss2"])))
    finally:
        if force_dp:
            import torch.distributed as dist
            dist.destroy_process_group()


@pytest.mark.parametrize("force_dp", [True, False], ids=["one-rank-nccl", "no-process-group"])
def test_segmented_graphs_equal_eager_steps(force_dp):
    """Three train steps -- one eager (the capture's warm-up) + two replays of the segment graphs with the exchange between them --
    land bit for bit where three plain eager steps land: the segmentation reorders nothing and the exchange (a SUM over one rank)
    adds nothing.  With a process group this is the whole N > 1 launch path short of a second rank."""
    import numpy as np
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_segmented_worker, args=(free_port(), q, force_dp))
    p.start()
    (g_sd, d_sds, g_total, d_loss2), = collect_from_children(q, [p])
    p.join(60)
    assert p.exitcode == 0
    dev = torch.device("cuda", 0)
    G, Ds, enc = _nets(dev)
    step = TR.GanTrainStep(G, Ds, enc)
    words, sent, lens, reals, noise, eps = _shard(0, dev)
    for _ in range(3):
        out = step.step(words, sent, lens, None, reals, noise, eps)
    torch.cuda.synchronize()
    assert float(out["g_total"]) == g_total and float(out["d_loss2"]) == d_loss2
    for k, v in G.state_dict().items():
        assert np.array_equal(v.detach().cpu().numpy(), g_sd[k]), k
    for d, sd in zip(Ds, d_sds):
        for k, v in d.state_dict().items():
            assert np.array_equal(v.detach().cpu().numpy(), sd[k]), k


def test_bf16_exchange_kernels_match_the_formula():
    """The three element-wise passes of the 16-bit wire format through the C ABI (include/agan.h: agan_exchange_*):
    pack = bf16(g) zero-padded to `world` equal pieces; sum = bf16 of the rank-ordered fp32 sum of the pieces; unpack = widen --
    bit for bit the arithmetic dataparallel.all_reduce_bf16_ documents, for 1..8 ranks' worth of pieces."""
    import ctypes
    L = importlib.import_module("attention-gan_amd.backend.lib")
    lib = L.load()
    dev = torch.device("cuda", 0)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    g = torch.Generator().manual_seed(4)
    for world in (1, 2, 3, 8):
        for n in (4, 1000, 65536 + 12, 7_084_592):
            nw = lib.agan_exchange_wire_elems(n, world)
            per = nw // world
            assert nw >= n and per * world == nw and per % 8 == 0 and nw - n < 8 * world + world
            grads = [(torch.randn(n, generator=g) * 10 ** float(torch.randint(-6, 3, (1,), generator=g))).to(dev) for _ in range(world)]
            wires = []
            for gr in grads:
                w = torch.full((nw,), float("nan"), dtype=torch.bfloat16, device=dev)
                L.call("agan_exchange_pack_bf16", p(gr), p(w), n, nw, st)
                assert torch.equal(w[:n], gr.to(torch.bfloat16)) and float(w[n:].float().abs().sum()) == 0.0
                wires.append(w)
            # what rank j receives: piece j of every rank, stacked in rank order; all ranks' sums, gathered, are the whole bucket
            gathered = torch.empty(nw, dtype=torch.bfloat16, device=dev)
            for j in range(world):
                pieces = torch.stack([wires[r][j * per:(j + 1) * per] for r in range(world)]).contiguous()
                L.call("agan_exchange_sum_bf16", p(pieces), world, per, p(gathered[j * per:(j + 1) * per]), st)
            out = torch.empty(n, device=dev)
            L.call("agan_exchange_unpack_bf16", p(gathered), p(out), n, st)
            acc = grads[0].to(torch.bfloat16).float()
            for r in range(1, world):
                acc = acc + grads[r].to(torch.bfloat16).float()
            assert torch.equal(out, acc.to(torch.bfloat16).float()), (world, n)


def _direct_bf16_worker(port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), AGAN_DP_FORCE="1", AGAN_RCCL_DIRECT="1", AGAN_DP_BF16="1")
    import torch.distributed as dist
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        DP = importlib.import_module("attention-gan_amd.dataparallel")
        net = torch.nn.Sequential(torch.nn.Linear(64, 128), torch.nn.Linear(128, 32)).to(dev)
        opt = OPT.FlatAdam(net.parameters())
        bk = DP.GradBuckets(opt, bucket_bytes=16 << 10)
        assert bk.direct is not None and bk.bf16 and "bf16 wire" in bk.mode and len(bk.bounds) > 1
        opt.zero_grad()
        bk.arm()
        net(torch.randn(8, 64, device=dev, generator=torch.Generator(device=dev).manual_seed(1))).pow(2).sum().backward()
        before = None
        scale = bk.finish()
        torch.cuda.synchronize()
        exchanged = opt.grad.clone()
        # the same gradients without any exchange
        opt.zero_grad()
        net(torch.randn(8, 64, device=dev, generator=torch.Generator(device=dev).manual_seed(1))).pow(2).sum().backward()
        opt._rebind()
        torch.cuda.synchronize()
        q.put((scale, exchanged.cpu().numpy(), opt.grad.clone().cpu().numpy()))
    finally:
        dist.destroy_process_group()


def test_one_rank_direct_rccl_bf16_exchange_rounds_once():
    """AGAN_RCCL_DIRECT=1 + AGAN_DP_BF16=1 (the pair used to fall back to the fp32 exchange silently): the 16-bit wire
    format on the library's own communicator -- pack, grouped send/recv (to itself with one rank), rank-ordered sum, all-gather,
    unpack.  With one rank the result is the bucket rounded to bf16: fp32(bf16(g))."""
    import numpy as np
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_direct_bf16_worker, args=(free_port(), q))
    p.start()
    (scale, exchanged, plain), = collect_from_children(q, [p])
    p.join(60)
    assert p.exitcode == 0 and scale == 1.0
    want = torch.from_numpy(plain).to(torch.bfloat16).float().numpy()
    assert np.array_equal(exchanged, want) and not np.array_equal(exchanged, plain)
