#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE itself on CPU.

Run in the build container only (the GPU box has no /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference modules are imported read-only from /root/reference; nothing of
their source is copied.  Two harness-side shims are applied:
  * torch.cuda.FloatTensor -> CPU FloatTensor (generator_submodules.py:163 draws
    eps on CUDA); the drawn eps is captured and stored so it can be replayed.
  * torch.ByteTensor -> bool tensor (words_loss.py:90 / sentence_loss.py:24 build
    uint8 masks that torch>=2 refuses in masked_fill_).
Gradients are taken of  sum_i <out_i, probe_i>  where probe is the closed-form
cosine pattern of `probe()` below (recomputed, not stored, by the tests).
Every file records the seed that produced it.
"""
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, REF)

torch.set_num_threads(8)
torch.use_deterministic_algorithms(True)

from networks.attention import AttentionModule, func_attention            # noqa: E402
from networks.generator import Generator                                   # noqa: E402
from networks.generator_submodules import GenInitialStage, GenNextStage, GenMakeImage, VarAutoEncoder  # noqa: E402
from networks.discriminators import Disc64, Disc128, Disc256               # noqa: E402
from utilities.layers import Layers                                        # noqa: E402
from losses.words_loss import WordsLoss                                    # noqa: E402
from losses.sentence_loss import SentenceLoss                              # noqa: E402
from losses.disc_loss import NonSaturatingDiscLoss                         # noqa: E402
from losses.gen_loss import NonSaturatingGenLoss                           # noqa: E402
from losses.KL_loss import KL_loss                                         # noqa: E402
from networks.rnn_encoder import RNNEncoder                                # noqa: E402

CPU = torch.device("cpu")
_captured_eps = []


def _fake_cuda_float_tensor(*size):
    t = torch.FloatTensor(*size)
    _captured_eps.append(t)
    return t


torch.cuda.FloatTensor = _fake_cuda_float_tensor
torch.ByteTensor = lambda a: torch.from_numpy(np.asarray(a)).bool()


def probe(shape, phase):
    n = int(np.prod(shape))
    return torch.cos(torch.arange(n, dtype=torch.float64) * 0.37 + phase).float().view(*shape)


def npy(t):
    return t.detach().cpu().numpy().copy()


def randomise_bn(module, gen):
    """Default BN init (gamma=1, beta=0) hides affine bugs; perturb it deterministically."""
    for m in module.modules():
        if isinstance(m, (torch.nn.BatchNorm2d, torch.nn.BatchNorm1d)):
            with torch.no_grad():
                m.weight.copy_(1.0 + 0.2 * torch.randn(m.weight.shape, generator=gen))
                m.bias.copy_(0.2 * torch.randn(m.bias.shape, generator=gen))


def sd(module, prefix="param/"):
    return {prefix + k: npy(v) for k, v in module.state_dict().items()}


def param_grads(module, prefix="gparam/"):
    return {prefix + k: npy(p.grad) for k, p in module.named_parameters() if p.grad is not None}


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez(path, **arrs)
    print(f"{name}: {os.path.getsize(path) / 1e6:.2f} MB, {len(arrs)} arrays")


# --------------------------------------------------------------------------- a1
def gen_attention(name, B, C, E, T, hw, lens, seed):
    g = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    mod = AttentionModule(nc_in=C, emb_dim=E)
    images = torch.randn(B, C, hw, hw, generator=g, requires_grad=True)
    words = torch.randn(B, E, T, generator=g, requires_grad=True)
    mask = torch.tensor([[1] * l + [0] * (T - l) for l in lens], dtype=torch.int64)
    mod.apply_mask(mask)
    ctx, attn = mod(images, words)
    loss = (ctx * probe(ctx.shape, 0.1)).sum() + (attn * probe(attn.shape, 0.2)).sum()
    loss.backward()
    save(name, seed=seed, images=npy(images), words=npy(words), mask=npy(mask), ctx=npy(ctx), attn=npy(attn),
         g_images=npy(images.grad), g_words=npy(words.grad), **sd(mod), **param_grads(mod))


# --------------------------------------------------------------------------- a2
def gen_func_attention(seed=11):
    g = torch.Generator().manual_seed(seed)
    q = torch.randn(4, 32, 7, generator=g, requires_grad=True)
    c = torch.randn(4, 32, 17, 17, generator=g, requires_grad=True)
    w, a = func_attention(q, c, gamma1=4.0)
    ((w * probe(w.shape, 0.3)).sum() + (a * probe(a.shape, 0.4)).sum()).backward()
    save("a2_func_attention", seed=seed, query=npy(q), context=npy(c), wctx=npy(w), attn=npy(a),
         g_query=npy(q.grad), g_context=npy(c.grad))


# --------------------------------------------------------------------------- a3/a4/a6 blocks
def gen_block(name, module, x_shape, seed):
    g = torch.Generator().manual_seed(seed)
    randomise_bn(module, g)
    module.train()
    before = sd(module)
    x = torch.randn(*x_shape, generator=g, requires_grad=True)
    y = module(x)
    (y * probe(y.shape, 0.5)).sum().backward()
    after = sd(module, "after/")
    after = {k: v for k, v in after.items() if "running" in k or "num_batches" in k}
    save(name, seed=seed, x=npy(x), y=npy(y), g_x=npy(x.grad), **before, **after, **param_grads(module))


# --------------------------------------------------------------------------- a5
def gen_generator(seed=21):
    torch.manual_seed(seed)
    g = torch.Generator().manual_seed(seed)
    gf, emb, z, cond, B, T = 4, 16, 10, 10, 2, 6
    G = Generator(gf_dim=gf, emb_dim=emb, z_dim=z, cond_dim=cond)
    randomise_bn(G, g)
    G.train()
    before = sd(G)
    noise = torch.randn(B, z, generator=g)
    sent = torch.randn(B, emb, generator=g, requires_grad=True)
    words = torch.randn(B, emb, T, generator=g, requires_grad=True)
    mask = torch.tensor([[1] * 6, [1] * 3 + [0] * 3], dtype=torch.int64)
    _captured_eps.clear()
    fakes, attns, mu, logvar = G(noise, sent, words, mask)
    eps = _captured_eps[-1].clone()
    loss = sum((f * probe(f.shape, 0.6 + i)).sum() for i, f in enumerate(fakes))
    loss = loss + sum((a * probe(a.shape, 0.7 + i)).sum() for i, a in enumerate(attns))
    loss = loss + (mu * probe(mu.shape, 0.8)).sum() + (logvar * probe(logvar.shape, 0.9)).sum()
    loss.backward()
    after = {k: v for k, v in sd(G, "after/").items() if "running" in k or "num_batches" in k}
    save("a5_generator", seed=seed, dims=np.array([gf, emb, z, cond, B, T]), noise=npy(noise), sent=npy(sent),
         words=npy(words), mask=npy(mask), eps=npy(eps),
         fake0=npy(fakes[0]), fake1=npy(fakes[1]), fake2=npy(fakes[2]), attn0=npy(attns[0]), attn1=npy(attns[1]),
         mu=npy(mu), logvar=npy(logvar), g_sent=npy(sent.grad), g_words=npy(words.grad),
         **before, **after, **param_grads(G))


# --------------------------------------------------------------------------- a7
def gen_disc(cls, res, B, df, seed):
    torch.manual_seed(seed)
    g = torch.Generator().manual_seed(seed)
    D = cls(df)
    randomise_bn(D, g)
    D.train()
    before = sd(D)
    x = (torch.rand(B, 3, res, res, generator=g) * 2 - 1).requires_grad_(True)
    y = D(x)
    (y * probe(y.shape, 1.1)).sum().backward()
    after = {k: v for k, v in sd(D, "after/").items() if "running" in k or "num_batches" in k}
    save(f"a7_disc{res}", seed=seed, df=df, x=npy(x), y=npy(y), g_x=npy(x.grad), **before, **after, **param_grads(D))


# --------------------------------------------------------------------------- a8/a9
def gen_damsm(seed=41):
    g = torch.Generator().manual_seed(seed)
    B, nef, T = 4, 32, 10
    lens = torch.tensor([10, 7, 2, 10])
    labels = torch.arange(B)
    out = {}
    for tag, cids in (("none", None), ("cls", np.array([0, 1, 2, 0]))):
        feat = torch.randn(B, nef, 17, 17, generator=g, requires_grad=True)
        wemb = torch.randn(B, nef, T, generator=g, requires_grad=True)
        code = torch.randn(B, nef, generator=g, requires_grad=True)
        semb = torch.randn(B, nef, generator=g, requires_grad=True)
        wl, maps = WordsLoss(CPU).get_loss(feat, wemb, labels, lens, cids)
        sl = SentenceLoss(CPU).get_loss(code, semb, labels, cids)
        (wl + sl).backward()
        out.update({f"{tag}/feat": npy(feat), f"{tag}/wemb": npy(wemb), f"{tag}/code": npy(code), f"{tag}/semb": npy(semb),
                    f"{tag}/wloss": npy(wl), f"{tag}/sloss": npy(sl),
                    f"{tag}/g_feat": npy(feat.grad), f"{tag}/g_wemb": npy(wemb.grad),
                    f"{tag}/g_code": npy(code.grad), f"{tag}/g_semb": npy(semb.grad)})
        for i, m in enumerate(maps):
            out[f"{tag}/map{i}"] = npy(m)
    # single-word caption edge case: still defined
    feat = torch.randn(B, nef, 17, 17, generator=g)
    wemb = torch.randn(B, nef, T, generator=g)
    wl1, _ = WordsLoss(CPU).get_loss(feat, wemb, labels, torch.tensor([1, 4, 10, 3]), None)
    out.update({"one/feat": npy(feat), "one/wemb": npy(wemb), "one/wloss": npy(wl1)})
    save("a8_a9_damsm", seed=seed, lens=npy(lens), class_ids=np.array([0, 1, 2, 0]), **out)


# --------------------------------------------------------------------------- a10
def gen_small_losses(seed=51):
    g = torch.Generator().manual_seed(seed)

    class P(torch.nn.Module):            # a fixed "discriminator": returns stored probabilities
        def __init__(self, real, fake):
            super().__init__()
            self.r, self.f, self.n = real, fake, 0

        def forward(self, x):
            self.n += 1
            return self.r if x is REAL else self.f
    REAL, FAKE = torch.zeros(1), torch.ones(1)
    dr = torch.rand(6, generator=g).requires_grad_(True)
    df_ = torch.rand(6, generator=g).requires_grad_(True)
    dl = NonSaturatingDiscLoss().get_loss(P(dr, df_), FAKE, REAL)
    dl.backward()
    gd_r, gd_f = npy(dr.grad), npy(df_.grad)
    df2 = df_.detach().clone().requires_grad_(True)
    gl = NonSaturatingGenLoss().get_loss(P(dr, df2), FAKE)
    gl.backward()
    mu = torch.randn(4, 10, generator=g, requires_grad=True)
    lv = torch.randn(4, 10, generator=g, requires_grad=True)
    kl = KL_loss(mu, lv)
    kl.backward()
    save("a10_losses", seed=seed, d_real=npy(dr), d_fake=npy(df_), dloss=npy(dl), g_d_real=gd_r, g_d_fake=gd_f,
         gloss=npy(gl), g_gl_fake=npy(df2.grad), mu=npy(mu), logvar=npy(lv), kl=npy(kl), g_mu=npy(mu.grad), g_logvar=npy(lv.grad))


# --------------------------------------------------------------------------- a11
def standin_encoder(img, proj, code_w):
    """Same definition as oracle.attngan_oracle.standin_encoder (frozen plug-in encoder stand-in)."""
    r = torch.nn.functional.adaptive_avg_pool2d(img, 17)
    regions = torch.einsum("ec,bchw->behw", proj, r)
    return regions, regions.mean(dim=(2, 3)) @ code_w.t()


def gen_train_step(seed=61, steps=2):
    """train.py:109-151 composed from the reference's own modules/losses/Adam, two consecutive steps."""
    from torch.optim import Adam
    torch.manual_seed(seed)
    g = torch.Generator().manual_seed(seed)
    gf, df, emb, z, cond, B, T = 4, 4, 16, 10, 10, 4, 6
    G = Generator(gf, emb, z, cond)
    Ds = [Disc64(df), Disc128(df), Disc256(df)]
    for m in [G] + Ds:
        randomise_bn(m, g)
        m.train()
    gopt = Adam(G.parameters(), lr=2e-4, betas=(0.5, 0.999))
    dopts = [Adam(d.parameters(), lr=2e-4, betas=(0.5, 0.999)) for d in Ds]
    out = {"dims": np.array([gf, df, emb, z, cond, B, T, steps])}
    out.update(sd(G, "G0/"))
    for i, d in enumerate(Ds):
        out.update(sd(d, f"D{i}_0/"))
    proj = 0.5 * torch.randn(emb, 3, generator=g)
    code_w = 0.5 * torch.randn(emb, emb, generator=g)
    out["enc_proj"], out["enc_code"] = npy(proj), npy(code_w)
    lens = torch.tensor([6, 3, 2, 5])
    class_ids = np.array([0, 1, 0, 2])
    mask = torch.tensor([[1] * int(l) + [0] * (T - int(l)) for l in lens], dtype=torch.int64)
    labels = torch.arange(B)
    wl_fn, sl_fn = WordsLoss(CPU, 4.0, 5.0, 10.0, 5.0), SentenceLoss(CPU, 10.0, 5.0)
    for s in range(steps):
        words = torch.randn(B, emb, T, generator=g)
        sent = torch.randn(B, emb, generator=g)
        noise = torch.randn(B, z, generator=g)
        # real images stored exactly as int8/128 to keep the fixture small
        reals_q = [torch.randint(-128, 128, (B, 3, r, r), generator=g, dtype=torch.int16).to(torch.int8) for r in (64, 128, 256)]
        reals = [q.float() / 128.0 for q in reals_q]
        _captured_eps.clear()
        fakes, _, mu, logvar = G(noise, sent, words, mask)
        eps = _captured_eps[-1].clone()
        losses = {}
        for i, (d, opt) in enumerate(zip(Ds, dopts)):
            opt.zero_grad()
            loss = NonSaturatingDiscLoss().get_loss(d, fakes[i], reals[i])
            loss.backward(retain_graph=True)
            opt.step()
            losses[f"d_loss{i}"] = float(loss)
        gopt.zero_grad()
        total = 0
        for i, d in enumerate(Ds):
            gl = NonSaturatingGenLoss().get_loss(d, fakes[i])
            total = total + gl
            losses[f"g_loss{i}"] = float(gl)
            if i == 2:
                regions, code = standin_encoder(fakes[i], proj, code_w)
                wl, _ = wl_fn.get_loss(regions, words, labels, lens, class_ids)
                sl = sl_fn.get_loss(code, sent, labels, class_ids)
                total = total + wl + sl
                losses["w_loss"], losses["s_loss"] = float(wl), float(sl)
        kl = KL_loss(mu, logvar)
        total = total + kl
        losses["kl"], losses["g_total"] = float(kl), float(total)
        total.backward()
        gopt.step()
        out.update({f"s{s}/words": npy(words), f"s{s}/sent": npy(sent), f"s{s}/noise": npy(noise), f"s{s}/eps": npy(eps),
                    f"s{s}/real64_q": npy(reals_q[0]), f"s{s}/real128_q": npy(reals_q[1]), f"s{s}/real256_q": npy(reals_q[2]),
                    f"s{s}/fake64": npy(fakes[0])})
        for k, v in losses.items():
            out[f"s{s}/{k}"] = np.float64(v)
        out.update(sd(G, f"G{s + 1}/"))
        for i, d in enumerate(Ds):
            out.update(sd(d, f"D{i}_{s + 1}/"))
    save("a11_train_step", seed=seed, lens=npy(lens), class_ids=class_ids, **out)


# --------------------------------------------------------------------------- f2 (text encoder, rnn_encoder.py:68-96)
def gen_rnn_encoder(seed=51):
    """Reference RNNEncoder (dropout 0 so that the draw-free forward is reproducible), captions padded to T = 10 with the
    length pattern of the other fixtures; outputs, and gradients of the probe loss w.r.t. every parameter."""
    torch.manual_seed(seed)
    g = torch.Generator().manual_seed(seed)
    vocab, embdim, nhidden, B, T = 50, 24, 32, 4, 10
    m = RNNEncoder(vocabsize=vocab, embdim=embdim, dropprob=0.0, nhidden=nhidden)
    m.train()
    lens = torch.tensor([10, 7, 2, 10])
    caps = torch.randint(1, vocab, (B, T), generator=g)
    for i, l in enumerate(lens.tolist()):
        caps[i, l:] = 0
    before = sd(m)
    w, s_ = m(caps, lens)
    ((w * probe(w.shape, 0.7)).sum() + (s_ * probe(s_.shape, 0.8)).sum()).backward()
    save("f2_rnn_encoder", seed=seed, dims=np.array([vocab, embdim, nhidden, B, T]), captions=npy(caps), lengths=npy(lens),
         word_embs=npy(w), sent_embs=npy(s_), **before, **param_grads(m))


def main():
    only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else None
    if only == "f2_rnn_encoder":
        return gen_rnn_encoder()
    gen_attention("a1_attention_small", B=4, C=8, E=32, T=10, hw=16, lens=[10, 7, 2, 10], seed=1)
    gen_attention("a1_attention_gen2", B=2, C=32, E=64, T=10, hw=64, lens=[10, 4], seed=2)
    gen_func_attention()
    gen_block("a3_upblock", Layers.upBlock(16, 8), (4, 16, 8, 8), seed=31)
    gen_block("a4_resblock", Layers.ResBlock(16), (4, 16, 16, 16), seed=32)
    gen_block("a6_downblock", Layers.downBlock(8, 16), (4, 8, 16, 16), seed=33)
    gen_block("a6_block3x3_leak", Layers.Block3x3_leakRelu(16, 8), (4, 16, 4, 4), seed=34)
    gen_block("a6_encode16", Layers.encode_image_by_16times(8), (2, 3, 64, 64), seed=35)
    gen_block("a5_make_image", GenMakeImage(8), (2, 8, 32, 32), seed=36)
    gen_generator()
    gen_disc(Disc64, 64, 4, 8, seed=37)
    gen_disc(Disc128, 128, 2, 8, seed=38)
    gen_disc(Disc256, 256, 2, 4, seed=39)
    gen_damsm()
    gen_small_losses()
    gen_train_step()
    gen_rnn_encoder()


if __name__ == "__main__":
    main()
