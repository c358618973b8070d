/*
 * agan.h -- C ABI of libagan_hip.so, the MI355X (gfx950) kernels under the AttnGAN training hot path.
 *
 * The reference (ku222/Attention-GAN) is pure Python on PyTorch and has no FFI of its own
 * (SURVEY.md §8b); its boundary is the Python module surface (networks/, losses/, utilities/layers.py,
 * trainers/trainer.py).  This header is the layer directly underneath that surface: each entry point
 * names the reference call site whose ATen ops it replaces.  The Python host code in
 * attention-gan_amd/ binds these with ctypes from torch.autograd.Function.forward/backward
 * (see INTEGRATION.md for the stub a reference maintainer would add).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer into caller-owned, contiguous, 16-byte-aligned storage
 *     (NCHW fp32 unless stated); the library never allocates, frees, synchronises or keeps state;
 *   - every call only enqueues work on `stream` (a hipStream_t passed as void*), so it is graph-capturable;
 *   - return value 0 = ok, negative AGAN_E* otherwise; agan_last_error() gives the message (thread-local);
 *   - "ws" is scratch the caller provides; the *_ws_bytes() function of an op says how much it needs.
 */
#ifndef AGAN_H_
#define AGAN_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AGAN_VERSION 101 /* 0.1.1 */
/* 101 (round 4): agan_words_loss_fwd / agan_sent_loss_fwd carry `labels` after class_ids (added in place during round 3 without a
 * version change: a caller built against the 100 header must be rebuilt -- bind by version, backend/lib.py asserts it);
 * agan_allreduce_bucket_dt + agan_exchange_* (16-bit wire format of the gradient exchange); amax slots of agan_conv_wgrad are
 * needed whenever agan_conv_wgrad_effective_prec says AGAN_PREC_F16X3, which includes AGAN_PREC_BF16X6 callers. */

enum {
    AGAN_OK = 0,
    AGAN_EINVAL = -1,    /* bad argument / unsupported shape */
    AGAN_EWORKSPACE = -2,/* workspace too small */
    AGAN_ELAUNCH = -3    /* HIP launch failure */
};

/* arithmetic mode of the MFMA contractions (fp32 storage everywhere, fp32 accumulate always) */
enum {
    AGAN_PREC_F32 = 0,    /* v_mfma_f32_32x32x2_f32: exact fp32 products                                              */
    AGAN_PREC_BF16 = 1,   /* operands rounded to bf16, v_mfma_f32_32x32x16_bf16 (BASELINE configs[1])                 */
    AGAN_PREC_BF16X3 = 2, /* bf16 hi/lo split (16 mantissa bits), 3 MFMAs per product                                 */
    AGAN_PREC_F16 = 3,    /* operands rounded to fp16, v_mfma_f32_32x32x16_f16 (BASELINE configs[4])                  */
    AGAN_PREC_BF16X6 = 4, /* bf16 hi/mid/lo split (24 mantissa bits), 6 MFMAs per product: fp32-grade products at 2.67x
                             the fp32-MFMA rate                                                                       */
    AGAN_PREC_F16X3 = 5   /* fp16 hi/lo split (22 mantissa bits), 3 MFMAs per product, operands scaled by powers of two into
                             fp16's range: fp32-grade products at 5.3x the fp32-MFMA rate.  The gathered operand's scale comes
                             from an amax slot (see agan_absmax below), the weights are packed times 2^11                   */
};
/* The 16-bit modes run on the patch-resident kernels (csrc/conv_patch.hip), which take 3x3 / 2x2-per-class stride-1 and 4x4
 * stride-2 geometries with more than 4 input and output channels and images of at least 4x4; any other call (linear layers, the RGB
 * heads) runs in AGAN_PREC_F32 whatever mode is asked for.  agan_conv_effective_prec says which one a geometry gets -- pack the
 * weights for THAT precision (declared below, after agan_conv_geom). */

/* Storage type of an ACTIVATION tensor in HBM (round 3).  fp32 is the reference's layout and the default everywhere.  The
 * one-plane 16-bit modes can also keep activations in their operand type -- AGAN_DT_BF16 with AGAN_PREC_BF16, AGAN_DT_F16 with
 * AGAN_PREC_F16 -- through the `*_dt` entry points below: conv outputs, BatchNorm inputs/outputs and their gradients are then
 * rounded ONCE to 16 bits when they are stored, the gathers load 16-bit values without converting, and half the bytes move.
 * BatchNorm statistics, softmax, losses, weight gradients, master weights and the optimiser stay fp32/fp64 in every mode. */
enum { AGAN_DT_F32 = 0, AGAN_DT_BF16 = 1, AGAN_DT_F16 = 2 };

int agan_version(void);
const char* agan_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * Gather-convolution geometry.  One descriptor covers every convolution on the path:
 *
 *   out[b, n, y'*OS+py, x'*OS+px] = sum_{c<Cin, r<R, s<S}
 *         in[b, c, y'*SY + r*DY + OY[py], x'*SY + s*DY + OY[px]] * wk[py*OS+px][(c*R + r)*S + s][n]
 *
 * (taps outside [0,IH)x[0,IW) read zero).  OS = 1 is an ordinary strided conv; OS = 2 splits the output
 * into its 4 parity classes, each with its own RxS taps -- which is how a stride-2 transposed conv and a
 * conv over a nearest-x2-upsampled input are expressed WITHOUT zero-insertion / without materialising the
 * upsampled tensor (K3 in SURVEY.md §2.1 never exists in HBM, and costs 4/9 of the 3x3 MACs).
 *
 * wk is the packed weight  [OS*OS][K = Cin*R*S][Nld],  Nld = agan_round_up(Cout, 32), zero padded,
 * produced from the reference's OIHW parameter by agan_pack_weight().
 * ---------------------------------------------------------------------------------------------- */
typedef struct agan_conv_geom {
    int32_t B;
    int32_t Cin, IH, IW;   /* gathered tensor  [B, Cin, IH, IW]  */
    int32_t Cout, OH, OW;  /* produced tensor  [B, Cout, OH, OW] */
    int32_t R, S;          /* taps per class                      */
    int32_t OS;            /* 1 or 2                              */
    int32_t SY, DY;        /* input step per output-lattice step / per tap (same in y and x) */
    int32_t OY[2];         /* input offset for output parity 0 / 1 */
} agan_conv_geom;

/* how agan_pack_weight lays an OIHW tensor out for a given use */
enum {
    AGAN_PACK_FWD = 0,       /* conv KHxKW any stride (layers.py:50-53,122):  wk[(ci,r,s)][co] = w[co][ci][r][s]            */
    AGAN_PACK_DGRAD_S1 = 1,  /* dgrad of a stride-1 'same' conv: wk[(co,r,s)][ci] = w[co][ci][KH-1-r][KW-1-s]              */
    AGAN_PACK_DGRAD_4x4S2 = 2,/* dgrad of conv4x4 s2 p1 as 4 parity classes of 2x2 taps                                     */
    AGAN_PACK_UP_FWD = 3,    /* Upsample(x2 nearest)+conv3x3 (layers.py:64-65) folded into 4 parity classes of 2x2 taps    */
    AGAN_PACK_UP_DGRAD = 4   /* its dgrad folded into one 4x4 stride-2 conv over dY                                         */
};

/* Packed size in BYTES for a precision mode: AGAN_PREC_F32 -> fp32 [cls][K][Nld];  the 16-bit modes -> [cls][k-step][plane][Nld][16]
 * (k-step = 16 channels of one tap, ordered chunk of 32 channels > input phase > tap > half; planes = 1 / 2 / 3 for
 * BF16, F16 / BF16X3 / BF16X6): the 32 x 16 operand block of a wave is 1 KB contiguous.  0 = combination not supported. */
size_t agan_packed_weight_bytes(int mode, int cout, int cin, int kh, int kw, int prec);
/* w: OIHW [cout][cin][kh][kw] -> wk (see modes).  Replaces nothing in the reference: layout prep for the kernels below. */
int agan_pack_weight(const float* w, void* wk, int mode, int cout, int cin, int kh, int kw, int prec, void* stream);
/* Many packs in one launch (all jobs of one call share `prec`, i.e. one layout family): what a module re-packs after an optimiser step.  `jobs` is a DEVICE array; job i owns
 * workgroups [first_block, first_block + agan_pack_job_blocks(...)), first_block ascending from 0; total_blocks = their sum.
 * Same layouts as agan_pack_weight, including zeroed padding columns. */
typedef struct agan_pack_job {
    const float* w;
    void* wk;
    int32_t mode, cout, cin, kh, kw, first_block;
} agan_pack_job;
int agan_pack_job_blocks(int mode, int cout, int cin, int kh, int kw);
int agan_pack_job_blocks_prec(int mode, int cout, int cin, int kh, int kw, int prec);   /* ... for a 16-bit layout (prec != AGAN_PREC_F32) */
int agan_pack_weights(const agan_pack_job* jobs, int njobs, int total_blocks, int prec, void* stream);

/* conv forward / dgrad: replaces F.conv2d fwd+dgrad under Layers.conv3x3 / conv4x4 s2 / Upsample+conv3x3 /
 * nn.Linear (1x1 on a 1x1 image) -- utilities/layers.py:50-53,64-65,122,139-150; generator_submodules.py:36,152;
 * discriminators.py:15.   bias may be NULL.  act: AGAN_ACT_NONE, or AGAN_ACT_LRELU to apply the LeakyReLU(0.2) that follows a
 * conv without BatchNorm (layers.py:139-141, first stage of encode_image_by_16times) in the epilogue (fp32 MFMA path only). */
/* Reduction-index table of a geometry (one int32 pair per k = (ci,r,s), padded): build it once per geometry with
 * agan_conv_ktable() into agan_conv_ktable_elems() int32s and pass it to every conv call of that geometry.  It lets the
 * kernels fetch the im2col offsets with scalar loads instead of decoding k with integer divisions. */
size_t agan_conv_ktable_elems(const agan_conv_geom* g);
int agan_conv_ktable(const agan_conv_geom* g, int32_t* table, void* stream);

int agan_conv_effective_prec(const agan_conv_geom* g, int prec);
/* ... and the one its weight gradient gets (agan_conv_wgrad demotes by itself; a caller needs this to know whether the
 * AGAN_PREC_F16X3 scales are wanted).  g = the FORWARD geometry, pack_mode = AGAN_PACK_FWD or AGAN_PACK_UP_FWD. */
int agan_conv_wgrad_effective_prec(const agan_conv_geom* g, int pack_mode, int prec);
/* Fraction of the direct contraction's multiply-adds that the kernel chosen for this call really issues (measurement aid: bench.py's
 * executed_tflops).  1.0 except where AGAN_PREC_F32 takes a Winograd kernel (csrc/conv_wino.hip): 16/36 for conv3x3 stride 1 (forward,
 * data gradient, weight gradient), 36/64 for the conv4x4 stride-2 forward, 9/16 for its class-wise data gradient.  wgrad != 0: g is the
 * FORWARD geometry of a weight gradient; plain_epilogue = 0: the call carries a bias / activation / mask (the Winograd gathers have none). */
double agan_conv_executed_fraction(const agan_conv_geom* g, int prec, int wgrad, int plain_epilogue);
size_t agan_conv_gather_ws_bytes(const agan_conv_geom* g, int prec);
/* lrelu_mask (optional, data-gradient launches): a tensor of the OUTPUT's shape; out is multiplied by LeakyReLU'(mask) = 1 where
 * mask > 0, else 0.2 -- the backward of the LeakyReLU that produced this conv's forward input (the mask is that input), folded
 * into the epilogue so that the activation's backward needs no pass of its own. */
int agan_conv_gather(const float* in, const void* wk, const float* bias, float* out, const agan_conv_geom* g,
                     const int32_t* ktable, int prec, int act, const float* lrelu_mask, void* ws, size_t ws_bytes, void* stream,
                     const float* in_amax /* AGAN_PREC_F16X3: amax slot of `in`; NULL otherwise */,
                     float* out_amax /* optional: amax slot that receives max|out| (16-bit MFMA paths) */);

/* The same call with typed activation storage: `in` has in_dtype, `out` and `lrelu_mask` have out_dtype (AGAN_DT_*).  16-bit
 * tensors need prec == AGAN_PREC_BF16 (AGAN_DT_BF16) / AGAN_PREC_F16 (AGAN_DT_F16) and a geometry the row-block gather
 * (csrc/conv_p16.hip) takes: agan_conv_gather_dt_supported says so (1 / 0) -- a caller converts around an unsupported layer.
 * With both dtypes AGAN_DT_F32 this IS agan_conv_gather (workspace: agan_conv_gather_ws_bytes in every case). */
int agan_conv_gather_dt_supported(const agan_conv_geom* g, int prec, int in_dtype, int out_dtype);
int agan_conv_gather_dt(const void* in, const void* wk, const float* bias, void* out, const agan_conv_geom* g,
                        const int32_t* ktable, int prec, int act, const void* lrelu_mask, void* ws, size_t ws_bytes, void* stream,
                        const float* in_amax, float* out_amax, int in_dtype, int out_dtype);

/* amax slots (AGAN_PREC_F16X3).  The fp16 split mode scales each gathered operand by the power of two that lands its largest
 * magnitude in (2^12, 2^13]; the kernels derive that scale themselves from an AMAX SLOT: AGAN_AMAX_SLOT floats of device memory,
 * ZEROED by the caller, into which the kernel that PRODUCES a tensor folds max|value| (the `*_amax` arguments of the BatchNorm
 * and conv entry points: no extra pass over the tensor), or which agan_absmax fills for a tensor that has no such producer.
 * Everything stays on the device and on the stream: no host sync, HIP-graph capturable. */
#define AGAN_AMAX_SLOT 256    /* 8 running maxima, each on its own 128-byte line */
int agan_absmax(const float* x, size_t n, float* amax_slot, void* stream);

/* conv weight gradient: x is the forward input, dy the gradient of the forward output, g the FORWARD geometry.
 * Produces dw in OIHW [cout][cin][kh][kw] (pack mode AGAN_PACK_FWD or AGAN_PACK_UP_FWD says how g was built).
 * Replaces the wgrad half of conv2d backward at the same call sites. */
size_t agan_conv_wgrad_ws_bytes(const agan_conv_geom* g);
/* accumulate != 0: dw += gradient (a parameter used twice in one backward, e.g. D on the real and the fake batch). */
/* x_amax / dy_amax: the amax slots of x and dy, REQUIRED whenever agan_conv_wgrad_effective_prec(g, pack_mode, prec) returns
 * AGAN_PREC_F16X3 -- that is prec == AGAN_PREC_F16X3 and also prec == AGAN_PREC_BF16X6, whose weight gradients run as two-plane
 * fp16 splits on the row-resident kernel (csrc/conv_wgrows.hip); NULL in every other mode. */
int agan_conv_wgrad(const float* x, const float* dy, float* dw, const agan_conv_geom* g, const int32_t* ktable, int pack_mode,
                    int kh, int kw, int prec, int accumulate, void* ws, size_t ws_bytes, void* stream,
                    const float* x_amax, const float* dy_amax);

/* the same with typed activation storage (x: x_dtype, dy: dy_dtype; dw stays fp32): 16-bit tensors where
 * agan_conv_wgrad_dt_supported says 1 -- the one-plane patch weight gradient of the matching precision mode */
int agan_conv_wgrad_dt_supported(const agan_conv_geom* g, int pack_mode, int prec, int x_dtype, int dy_dtype);
int agan_conv_wgrad_dt(const void* x, const void* dy, float* dw, const agan_conv_geom* g, const int32_t* ktable, int pack_mode,
                       int kh, int kw, int prec, int accumulate, void* ws, size_t ws_bytes, void* stream,
                       const float* x_amax, const float* dy_amax, int x_dtype, int dy_dtype);

/* dbias[n] = sum_{b,y,x} dy[b,n,y,x]  (bias of nn.Linear / outlogits conv: generator_submodules.py:152, discriminators.py:15) */
int agan_bias_grad(const float* dy, float* dbias, int B, int C, int HW, int accumulate, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Train-mode BatchNorm + activation (utilities/layers.py:66-67,123-124,143-144,164-174; generator_submodules.py:38)
 * ---------------------------------------------------------------------------------------------- */
enum { AGAN_ACT_NONE = 0, AGAN_ACT_GLU = 1, AGAN_ACT_LRELU = 2, AGAN_ACT_TANH = 3, AGAN_ACT_SIGMOID = 4 };

/* batch statistics over (B, HW) per channel: mean[C], invstd[C] = 1/sqrt(biased var + eps).  If running_mean is
 * non-NULL also does running = (1-momentum)*running + momentum*stat (unbiased var) and ++*num_batches_tracked. */
size_t agan_bn_stats_ws_bytes(int B, int C, int HW);
int agan_bn_stats(const float* x, int B, int C, int HW, float eps, float* mean, float* invstd,
                  float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum,
                  void* ws, size_t ws_bytes, void* stream);
/* y = act(gamma*(x-mean)*invstd+beta) [+ residual].  GLU halves the channels (out has C/2).  residual may be NULL. */
int agan_bn_act_fwd(const float* x, const float* mean, const float* invstd, const float* gamma, const float* beta,
                    const float* residual, float* out, int B, int C, int HW, int act, void* stream, float* out_amax);
/* Training-mode forward in one call: batch statistics (written to mean/invstd for the backward, folded into the running
 * statistics) + normalise + activation.  Tensors with B*HW <= 8192 per channel run as ONE launch (a workgroup per channel
 * keeps the channel in registers between the two phases); larger ones as agan_bn_stats + agan_bn_act_fwd.
 * groups: the batch axis is `groups` consecutive sub-batches of B/groups samples, each normalised with ITS OWN statistics
 * (mean / invstd are [groups][C]); the running statistics receive one update per group, in batch order.  That is what
 * `groups` separate forward passes through the layer do -- the discriminator's [real; fake] pass (disc_loss.py:55-61) uses 2.
 * Workspace sizes are per group: call the *_ws_bytes functions with B/groups. */
size_t agan_bn_train_fwd_ws_bytes(int B, int C, int HW);
int agan_bn_train_fwd(const float* x, const float* gamma, const float* beta, const float* residual, float* out, float* mean,
                      float* invstd, float* running_mean, float* running_var, int64_t* num_batches_tracked, int B, int C, int HW,
                      float eps, float momentum, int act, int groups, void* ws, size_t ws_bytes, void* stream, float* out_amax);
/* backward: dx[B,C,HW], dgamma[C], dbeta[C] from dout (C/2 channels under GLU); gamma/beta gradients are summed over the groups. */
size_t agan_bn_act_bwd_ws_bytes(int B, int C, int HW);
int agan_bn_act_bwd(const float* x, const float* dout, const float* mean, const float* invstd, const float* gamma,
                    const float* beta, float* dx, float* dgamma, float* dbeta, int B, int C, int HW, int act, int accumulate,
                    int groups, void* ws, size_t ws_bytes, void* stream, float* dx_amax);
/* The same three calls with typed activation storage (AGAN_DT_*): x and dx have x_dtype; out, dout and the residual have
 * out_dtype.  Combinations: (F32, F32) = the calls above; (BF16, BF16), (F16, F16); (F32, BF16), (F32, F16) for a layer whose conv
 * ran on an fp32-storage kernel.  Statistics, gamma / beta and their gradients stay fp32 (sums in fp64).  [B, C] inputs
 * (BatchNorm1d) are fp32 only. */
int agan_bn_stats_dt(const void* x, int B, int C, int HW, float eps, float* mean, float* invstd, float* running_mean,
                     float* running_var, int64_t* num_batches_tracked, float momentum, void* ws, size_t ws_bytes, void* stream,
                     int x_dtype);
int agan_bn_act_fwd_dt(const void* x, const float* mean, const float* invstd, const float* gamma, const float* beta,
                       const void* residual, void* out, int B, int C, int HW, int act, void* stream, float* out_amax, int x_dtype,
                       int out_dtype);
int agan_bn_train_fwd_dt(const void* x, const float* gamma, const float* beta, const void* residual, void* out, float* mean,
                         float* invstd, float* running_mean, float* running_var, int64_t* num_batches_tracked, int B, int C, int HW,
                         float eps, float momentum, int act, int groups, void* ws, size_t ws_bytes, void* stream, float* out_amax,
                         int x_dtype, int out_dtype);
int agan_bn_act_bwd_dt(const void* x, const void* dout, const float* mean, const float* invstd, const float* gamma,
                       const float* beta, void* dx, float* dgamma, float* dbeta, int B, int C, int HW, int act, int accumulate,
                       int groups, void* ws, size_t ws_bytes, void* stream, float* dx_amax, int x_dtype, int out_dtype);
/* plain activations without BN (first D conv + LeakyReLU layers.py:139-140; tanh generator_submodules.py:137) */
int agan_act_fwd(const float* x, float* out, size_t n, int act, void* stream);
int agan_act_bwd(const float* out, const float* dout, float* dx, size_t n, int act, void* stream);
/* standalone GLU over channel halves of [B,C,HW] (utilities/layers.py:13-26; the CA-net's "relu", generator_submodules.py:153) */
int agan_glu_fwd(const float* x, float* out, int B, int C, int HW, void* stream);
int agan_glu_bwd(const float* x, const float* dout, float* dx, int B, int C, int HW, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Word-context attention: AttentionModule.forward, networks/attention.py:25-79
 *   proj[b,c,t] = sum_e w[c,e] words[b,e,t]                      (:50-52, conv1x1)
 *   attn[b,t,p] = softmax_t( scale * sum_c images[b,c,p] proj[b,c,t], mask[b,t]==0 -> -inf )   (:59-68)
 *   ctx[b,c,p]  = sum_t proj[b,c,t] attn[b,t,p]                 (:73)
 * T <= 64.  mask is the reference's int64 [B,T].
 * ---------------------------------------------------------------------------------------------- */
int agan_attn_fwd(const float* images, const float* words, const float* w, const int64_t* mask, float scale,
                  float* proj, float* ctx, float* attn, int B, int C, int E, int T, int HW, void* stream);
/* dctx / dattn may be NULL (treated as zero).  Outputs dimages[B,C,HW], dwords[B,E,T], dw[C,E].  No atomics: the projected-word
 * gradient is reduced through per-workgroup slabs in `ws`, so the result is bit-reproducible. */
size_t agan_attn_bwd_ws_bytes(int B, int C, int T, int HW);
int agan_attn_bwd(const float* images, const float* words, const float* w, const float* proj, const float* attn,
                  const float* dctx, const float* dattn, float scale, float* dimages, float* dwords, float* dw,
                  int B, int C, int E, int T, int HW, int accumulate, void* ws, size_t ws_bytes, void* stream);
/* the same two calls with typed activation storage: `images`, `ctx`, `dctx` and `dimages` have `dtype` (AGAN_DT_*); the attention
 * map, the projection and all arithmetic stay fp32 */
int agan_attn_fwd_dt(const void* images, const float* words, const float* w, const int64_t* mask, float scale, float* proj,
                     void* ctx, float* attn, int B, int C, int E, int T, int HW, void* stream, int dtype);
int agan_attn_bwd_dt(const void* images, const float* words, const float* w, const float* proj, const float* attn,
                     const void* dctx, const float* dattn, float scale, void* dimages, float* dwords, float* dw, int B, int C,
                     int E, int T, int HW, int accumulate, void* ws, size_t ws_bytes, void* stream, int dtype);

/* ------------------------------------------------------------------------------------------------
 * DAMSM losses: WordsLoss.get_loss (losses/words_loss.py:29-102, which loops func_attention
 * networks/attention.py:82-120 over captions) and SentenceLoss.get_loss (losses/sentence_loss.py:12-50).
 *   feat [B,D,S] image regions, wemb [B,D,T] words, lens[B] int64, class_ids[B] int64 or NULL.
 * Outputs: loss[1]; sim[B*B] (gamma3-scaled, masked similarity matrix: [image][caption]); attn maps
 * [B(caption)][T][S] of each caption against ITS OWN image (what the reference returns in att_maps).
 * fwd also leaves everything the backward needs in `save` (agan_words_loss_save_elems floats).
 * ---------------------------------------------------------------------------------------------- */
/* standalone func_attention forward (attention.py:82-120): query [B,D,L], context [B,D,S] -> wctx [B,D,L], attn [B,L,S] */
int agan_func_attention_fwd(const float* query, const float* context, float gamma1, float scale, float* wctx, float* attn,
                            int B, int D, int L, int S, void* stream);
/* its backward (plain autograd in the reference): dwctx [B,D,L] / dattn [B,L,S] are the upstream gradients (either may be NULL);
 * writes dquery [B,D,L] and dcontext [B,D,S].  Recomputes the forward of each batch element; no atomics. */
int agan_func_attention_bwd(const float* query, const float* context, const float* dwctx, const float* dattn, float gamma1,
                            float scale, float* dquery, float* dcontext, int B, int D, int L, int S, void* stream);
size_t agan_words_loss_save_elems(int B, int D, int T, int S);
/* labels [B] int64: the CE targets of words_loss.py:98-99 / sentence_loss.py:46-47 (NULL = arange(B), what train.py:104 builds;
 * a label outside [0,B) makes the loss NaN).  class_ids [B] int64 or NULL: same-class pairs masked to -inf. */
int agan_words_loss_fwd(const float* feat, const float* wemb, const int64_t* lens, const int64_t* class_ids, const int64_t* labels,
                        float gamma1, float gamma2, float gamma3, float lambda, float* loss, float* sim, float* attn_maps,
                        float* save, int B, int D, int T, int S, void* stream);
/* backward: dfeat / dwemb are overwritten.  No atomics (bit-reproducible): each (image, caption) pair writes its contribution to a
 * slab in `ws` (agan_words_loss_bwd_ws_bytes: B*B*D*(S+T) floats) and a second pass adds the B slabs of each row in index order. */
size_t agan_words_loss_bwd_ws_bytes(int B, int D, int T, int S);
int agan_words_loss_bwd(const float* feat, const float* wemb, const int64_t* lens, const float* save, const float* dloss,
                        float gamma1, float gamma2, float gamma3, float lambda, float* dfeat, float* dwemb,
                        int B, int D, int T, int S, void* ws, size_t ws_bytes, void* stream);
int agan_sent_loss_fwd(const float* cnn_code, const float* rnn_code, const int64_t* class_ids, const int64_t* labels, float gamma3, float lambda,
                       float eps, float* loss, float* save /* 2*B*B + 2*B floats */, int B, int D, void* stream);
int agan_sent_loss_bwd(const float* cnn_code, const float* rnn_code, const float* save, const float* dloss, float gamma3,
                       float lambda, float eps, float* dcnn, float* drnn, int B, int D, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Small fused heads: GAN / KL losses and the CA-net reparametrisation
 *   disc: -mean(log(pr+1e-8)+log(1-pf+1e-8))  losses/disc_loss.py:55-61   (pr = D(x), pf = D(G(z)): probabilities)
 *   gen : -mean(log(pf+1e-8))                 losses/gen_loss.py:45-46
 *   kl  : -0.5*mean(1+lv-mu^2-exp(lv))        losses/KL_loss.py:5-9
 *   reparam: c = eps*exp(0.5*lv)+mu           networks/generator_submodules.py:161-165
 * Each loss kernel also writes d(loss)/d(input) (pointers may be NULL).
 * ---------------------------------------------------------------------------------------------- */
int agan_disc_loss(const float* p_real, const float* p_fake, float* loss, float* dp_real, float* dp_fake, int B, void* stream);
int agan_gen_loss(const float* p_fake, float* loss, float* dp_fake, int B, void* stream);
int agan_kl_loss(const float* mu, const float* logvar, float* loss, float* dmu, float* dlogvar, int n, void* stream);
int agan_reparam_fwd(const float* mu, const float* logvar, const float* eps, float* c, int n, void* stream);
int agan_reparam_bwd(const float* logvar, const float* eps, const float* dc, float* dmu, float* dlogvar, int n, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused Adam over one flat parameter buffer (torch.optim.Adam as configured at train.py:78-80:
 * no weight decay, no amsgrad).  grad_scale multiplies the gradient first (1/world_size after a sum all-reduce).
 * step_state: 16 bytes of device memory owned by the optimiser ([0] = int32 step count, advanced by the call; [2],[3] = the
 * bias-correction coefficients it derives) -- device-resident so that a captured HIP graph advances the step on every replay. */
int agan_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, int32_t* step_state, double lr,
                   double beta1, double beta2, double eps, float grad_scale, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Measurement hook (bench.py `roofline`): agan_timer_arm(start, stop) makes the NEXT agan_conv_gather / agan_conv_wgrad call on
 * this thread record the two HIP events on its launch stream immediately around its main MFMA (or small-N) kernel -- not
 * around the slab-sum / unpack passes a split launch appends.  One-shot; events come from agan_timer_create.
 * ---------------------------------------------------------------------------------------------- */
int agan_timer_create(void** event);
int agan_timer_destroy(void* event);
int agan_timer_arm(void* start, void* stop);
int agan_timer_elapsed_ms(void* start, void* stop, float* ms);
/* The kernel the most recent TIMED call on this thread ran as its main kernel: the demangled symbol exactly as `rocprofv3
 * --kernel-trace --stats` names it (e.g. "void (anonymous namespace)::conv_gather_f32_kernel<128, 128, 2, 2, 0>(float const*, ...)"),
 * copied into name[capacity]; empty if nothing has been timed.  bench.py keys `roofline_by_kernel` with it. */
int agan_timer_last_kernel(char* name, size_t capacity);

static inline int agan_round_up(int v, int m) { return (v + m - 1) / m * m; }

/* ------------------------------------------------------------------------------------------------
 * Data-parallel gradient exchange over RCCL (new component: the reference is single-GPU).
 * One communicator per process (one process per GPU).  Rank 0 makes the id and hands its AGAN_COMM_ID_BYTES bytes to every rank out
 * of band (the Python host uses torch.distributed's store); agan_comm_init is collective over all ranks and binds the calling
 * thread's current device.  agan_allreduce_bucket sums `n` floats in place across the ranks on `stream` (reduce-scatter + all-gather
 * when n divides evenly, else one all-reduce call); the 1/world scale is applied by agan_adam_step's grad_scale.
 * RCCL is bound at run time: without a loadable librccl these calls return AGAN_EINVAL and nothing else is affected.
 * ---------------------------------------------------------------------------------------------- */
#define AGAN_COMM_ID_BYTES 128
int agan_comm_unique_id(void* id);
int agan_comm_init(void** comm, int rank, int world, const void* id);
int agan_comm_destroy(void* comm);
int agan_allreduce_bucket(void* comm, float* buf, size_t n, void* stream);
/* The same exchange with a WIRE dtype (SURVEY.md section 8b's `agan_allreduce_bucket(buf, n, dtype, comm, stream)`):
 * AGAN_DT_F32 = the call above; AGAN_DT_BF16 = the bucket travels as bf16 and is ACCUMULATED IN FP32 on arrival:
 * round (agan_exchange_pack_bf16) -> all-to-all of the `world` pieces (grouped send/recv) -> rank-ordered fp32 sum of the pieces a
 * rank owns, rounded once (agan_exchange_sum_bf16) -> all-gather of the sums -> widen (agan_exchange_unpack_bf16):
 *     buf = fp32(bf16(sum_r fp32(bf16(buf_r)))), identical on every rank, half the bytes of the fp32 exchange on every link.
 * n must be a multiple of 4; `scratch` = agan_allreduce_scratch_bytes(n, world, dtype) bytes of device memory (0 for AGAN_DT_F32). */
size_t agan_allreduce_scratch_bytes(size_t n, int world, int wire_dtype);
int agan_allreduce_bucket_dt(void* comm, float* buf, size_t n, int wire_dtype, void* scratch, size_t scratch_bytes, void* stream);
/* The three element-wise passes of the 16-bit wire format on their own (a host that moves the bytes itself -- torch.distributed's
 * all_to_all_single / all_gather_into_tensor in dataparallel.all_reduce_bf16_ -- uses them around its collectives).  16 bytes per
 * lane, HBM-bound.  agan_exchange_wire_elems(n, world) = elements of the wire image: `world` equal pieces, each a multiple of 8
 * elements (piece r = elements [r*per, (r+1)*per) of the zero-padded bucket).  n and per are multiples of 4. */
size_t agan_exchange_wire_elems(size_t n, int world);
int agan_exchange_pack_bf16(const float* g, void* wire, size_t n, size_t n_wire, void* stream);
int agan_exchange_sum_bf16(const void* pieces /* [world][per] bf16, piece r from rank r */, int world, size_t per,
                           void* sum /* [per] bf16 */, void* stream);
int agan_exchange_unpack_bf16(const void* wire, float* g, size_t n, void* stream);
/* the reduce-scatter / all-gather chunk of a bucket of n floats over `world` ranks (rank r owns elements [r*chunk, (r+1)*chunk));
 * 0 = the bucket does not split into 16-byte aligned equal chunks and goes out as one library all-reduce.  Pure host arithmetic. */
size_t agan_allreduce_chunk_elems(size_t n, int world);

#ifdef __cplusplus
}
#endif
#endif /* AGAN_H_ */
