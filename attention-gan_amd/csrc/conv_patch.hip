// Patch-resident implicit-GEMM convolution on the 16-bit matrix cores of gfx950 (v_mfma_f32_32x32x16_{bf16,f16}).
//
// Modes (include/agan.h):  AGAN_PREC_BF16 / AGAN_PREC_F16  one rounded plane, one MFMA per product;
//                          AGAN_PREC_BF16X3               two bf16 planes (hi, lo), 3 MFMAs per product (~2^-16 per product);
//                          AGAN_PREC_BF16X6               three bf16 planes (hi, mid, lo = 24 mantissa bits), 6 MFMAs per
//                                                         product: fp32-grade products at 2.67x the fp32-MFMA rate.
// HBM tensors stay NCHW fp32; fp32 accumulate always.
//
// Why a new structure instead of the k-table gather of conv.hip: the fp32 kernel fetches every im2col element from L1/L2 --
// a 3x3 layer reads each input value nine times -- which is what capped the first split-precision kernel (conv_bf16.hip)
// at 1.2x.  Here a workgroup owns a 2-D tile of 128 output pixels (32 wide when the image allows it) and stages the INPUT
// PATCH of that tile (tile + halo) for 32 channels in LDS ONCE per stage, already converted to 16-bit planes and laid out
// [position][channel]; the reduction index is reordered to (channel chunk, phase, tap, channel), so the MFMA operand of a
// pixel for tap (r, s) and 8 consecutive channels is ONE aligned ds_read_b128 at
//        patch[(y + r) * PW + (x + s)][8 channels]
// -- no im2col tile is ever built, no per-element address work, and a 3x3 layer loads 1.6 instead of 9 values per pixel
// and channel.  Position rows are 80 bytes (32 x 16 bit + 16 pad): an odd multiple of 16 B, so the 16-lane groups of
// ds_read_b128 (lanes = 32 consecutive pixels of a tile row) and the 8-lane groups of ds_write_b128 are conflict-free.
// Stride-2 convolutions (conv4x4 s2, and the 4x4 s2 data gradient of the upsample conv) run as FOUR PHASES -- the four parity
// sub-lattices of the input, each a 2x2-tap stride-1 problem on its own patch -- so LDS reads stay unit-stride.
//
// Weights never touch LDS: agan_pack_weight lays them out [class][k-step][plane][cout][16] in 16-bit, i.e. the 32 x 16
// operand block of a wave for one k-step is 1 KB contiguous and is fetched (L2-resident, shared by every workgroup) by
// one buffer_load_b128 per lane straight into the MFMA operand registers, two k-steps ahead of its use.  The K loop therefore
// synchronises only where the patch changes (every 8 or 18 k-steps), not per k-step.
//
// Accumulators are D[cout][pixel] like conv.hip (weights are the A operand): a register holds 32 consecutive pixels of one
// output channel, NCHW stores are 128-byte coalesced.  Each of the 4 waves owns 32 output channels x all 128 pixels (BN = 128),
// so pixel fragments are shared through LDS and weight fragments are not duplicated between waves.
#include "conv_common.h"
#include "split16.h"

using namespace agan;
using namespace agan::conv;

namespace {

constexpr int kCH = 32;             // channels per stage
constexpr int kPosBytes = 80;       // LDS bytes per patch position and plane
constexpr int kMaxPos = 320;        // positions a patch may have (host checks)
constexpr int kItems = (kMaxPos * (kCH / 8) + 255) / 256;     // (position, channel octet) staging items per thread

// ================================================================================================
// forward / data-gradient kernel
// ================================================================================================
// ------------------------------------------------------------------------------------------------
// The gather kernel (the first generation -- single-buffered patch, two barriers per stage -- was removed in round 3): the patch is DOUBLE-BUFFERED in LDS and the conversion + LDS store of the next stage's
// patch is issued item by item BETWEEN the MFMA k-steps of the current stage (an MFMA occupies the vector issue port for 8 of
// its 32 cycles, so the ~500 VALU instructions of a three-plane conversion disappear behind the matrix pipe instead of
// standing between two barriers), which leaves ONE barrier per stage.  The three-plane mode stages 16 channels at a time (48-byte
// positions: still an odd multiple of 16 B), so both buffers of a 4x32 tile take 59 KB and two workgroups share a CU.
// LDS is dynamic: 2 buffers x planes x positions x position bytes for the geometry's real patch size.
template <int ET, int NPL, int BN, int NT>
__global__ __launch_bounds__(256, 2) void conv_patch_kernel2(const float* __restrict__ in, const unsigned short* __restrict__ wk,
                                                             const float* __restrict__ bias, float* __restrict__ out, const Geom g,
                                                             const PatchPlan pp, const int ksplit, const int stages_per_split,
                                                             const size_t slab, const int act, const float* __restrict__ lrelu_mask,
                                                             const int plane_bytes, const float* __restrict__ in_amax,
                                                             float* __restrict__ out_amax) {
    constexpr bool SCALED = (ET == 1 && NPL == 2);               // AGAN_PREC_F16X3: operands scaled into fp16's range
    float xs = 1.f, unscale = 1.f;                                // powers of two: exact
    if (SCALED) {
        float inv;
        xs = amax_scale(amax_read(in_amax), &inv);
        unscale = inv * (1.f / kF16WeightScale);
    }
    constexpr int CHS = NPL >= 3 ? 16 : 32;                       // channels per stage
    constexpr int PB = CHS * 2 + 16;                              // LDS bytes per position and plane
    constexpr int HS = CHS / 16;                                  // 16-channel k-steps per tap
    constexpr int SPS = NT * HS;                                  // k-steps per stage
    constexpr int NI = (kMaxPos * (CHS / 8) + 255) / 256;         // staging items per thread
    constexpr int U0 = SPS - NI;                                  // first k-step that carries a staging item
    static_assert(U0 >= 0, "stage too short for its staging items");
    constexpr int WN = BN / 32, WM = 4 / WN, TM = 4 / WM;
    // depth of the weight-fragment register ring = how many k-steps ahead a fragment is fetched from L2.  A one-plane k-step is
    // only TM MFMAs (128 cycles): two steps ahead (the depth the split modes use, whose k-steps are 3-6x longer) left every k-step
    // waiting ~300 cycles for its L2 hit -- MFMA-busy 11-18 % in round 2.  The depth divides the k-steps per stage (8 or 18), so
    // slot indices stay compile-time constants across stages.
#ifndef AGAN_PATCH_WD1
#define AGAN_PATCH_WD1 0
#endif
#ifndef AGAN_PATCH_WDS
#define AGAN_PATCH_WDS 0
#endif
    constexpr int WD = (NPL == 1 && HS == 2) ? (AGAN_PATCH_WD1 ? AGAN_PATCH_WD1 : (NT == 9 ? 6 : 4))
                                             : ((AGAN_PATCH_WDS && NPL == 2) ? (NT == 9 ? AGAN_PATCH_WDS * 3 : AGAN_PATCH_WDS * 2)
                                                                             : ((AGAN_PATCH_WDS && NPL == 3 && NT == 9) ? 3 : 2));
    static_assert(WD == 2 || SPS % WD == 0, "ring depth must divide the k-steps per stage");
    extern __shared__ __attribute__((aligned(16))) unsigned char patch2[];
    const int buf_bytes = NPL * plane_bytes;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave % WN, wm = wave / WN;
    const int l31 = lane & 31, lh = lane >> 5;
    int mt, nt, cls, split;
    {
        // (multiply-high with host-made constants instead of runtime scalar divisions, ~25 instructions each: with the stage decode below
        //  there were nine of them among the ~560 instructions in front of the kernel's first load -- round 4)
        const int ncls = pp.g_ncls, mtiles = gridDim.x, ntiles = pp.g_ntiles;
        int F = xcd_contiguous(linear_block_id(), mtiles * ntiles * (int)gridDim.z);
        int q = pp.dNT.div(F);
        nt = F - q * ntiles; F = q;
        q = pp.dNCLS.div(F);
        cls = F - q * ncls; F = q;
        split = pp.dMT.div(F);
        mt = F - split * mtiles;
    }
    const int py = cls >= g.OS ? 1 : 0, px = cls - py * g.OS;           // (OS <= 2)
    const int n0 = nt * BN;
    const int mty = pp.dTX.div(mt), txi = mt - mty * pp.tiles_x, tbi = pp.dTY.div(mty), tyi = mty - tbi * pp.tiles_y;
    const int twl = pp.twl, thl = pp.thl;
    const int tb0 = tbi << (7 - twl - thl), ty0 = tyi << thl, tx0 = txi << twl;
    const int ihw = g.IH * g.IW;
    // stages of this kernel: (32-channel chunk, [16-channel half,] phase); the host counts stages in units of (chunk, phase)
    const int nst = pp.nstages * (2 / HS);
    const int stage_beg = split * stages_per_split * (2 / HS), stage_end = min(nst, stage_beg + stages_per_split * (2 / HS));

    const __amdgpu_buffer_rsrc_t rin = make_rsrc(in, (size_t)g.B * g.Cin * ihw * sizeof(float));
    const size_t wbytes_cls = (size_t)pp.nsteps * NPL * g.Nld * 32;
    const __amdgpu_buffer_rsrc_t rwk = make_rsrc(reinterpret_cast<const unsigned char*>(wk) + (size_t)cls * wbytes_cls, wbytes_cls);

    int it_iy[NI], it_ix[NI], it_cb[NI], it_oct[NI];
    unsigned it_lds[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int e = tid + i * 256;
        const int oct = pp.dPP.div(e), p = e - oct * pp.PP;
        const int pb = pp.dPHW.div(p), rem = p - pb * pp.PHW;
        const int j = pp.dPW.div(rem), ii = rem - j * pp.PW;
        const int b = tb0 + pb;
        const bool ok = (oct < CHS / 8) & (b < g.B);
        it_iy[i] = pp.IS * (ty0 + j);
        it_ix[i] = pp.IS * (tx0 + ii);
        it_cb[i] = ok ? (b * g.Cin + oct * 8) * ihw : -1;
        it_oct[i] = oct;
        it_lds[i] = (unsigned)(p * PB + oct * 16);
    }
    float sreg[NI][8];
    const bool cin8 = (g.Cin & 7) == 0;

    // stage -> (chunk32, half16, phase): decoded ONCE (an integer division by a runtime value costs ~25 instructions), then advanced
    // by counters; wq = k-step of the packed weights of (stage, tap 0, sub-step 0)
    struct Stage { int chunk, half, ph, wq; };
    auto set_wq = [&](Stage& S) { S.wq = (S.chunk * pp.NPH + S.ph) * NT * 2 + (HS == 2 ? 0 : S.half); };
    auto stage_of = [&](int st) {
        Stage S;
        if (HS == 2) { S.chunk = pp.dNPH.div(st); S.ph = st - S.chunk * pp.NPH; S.half = 0; }
        else { S.chunk = pp.dNPH2.div(st); const int rem = st - S.chunk * 2 * pp.NPH; S.half = rem >= pp.NPH ? 1 : 0; S.ph = rem - S.half * pp.NPH; }
        set_wq(S);
        return S;
    };
    auto advance = [&](Stage S) {
        if (++S.ph == pp.NPH) {
            S.ph = 0;
            if (HS == 2) ++S.chunk;
            else if (++S.half == 2) { S.half = 0; ++S.chunk; }
        }
        set_wq(S);
        return S;
    };
    auto load_patch = [&](const Stage& S) {
        const int by = pp.base_y[py][S.ph >> 1], bx = pp.base_x[px][S.ph & 1];
        const int c0 = S.chunk * 32 + S.half * 16;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int iy = it_iy[i] + by, ix = it_ix[i] + bx;
            const bool ok = (it_cb[i] >= 0) & ((unsigned)iy < (unsigned)g.IH) & ((unsigned)ix < (unsigned)g.IW);
            const int nrem = g.Cin - c0 - it_oct[i] * 8;
            const unsigned voff = (ok & (nrem > 0)) ? (unsigned)(it_cb[i] + c0 * ihw + iy * g.IW + ix) * 4u : kOOB;
            if (cin8) {
#pragma unroll
                for (int c = 0; c < 8; ++c) sreg[i][c] = buf_load_s(rin, voff, (unsigned)(c * ihw) * 4u);
            } else {
#pragma unroll
                for (int c = 0; c < 8; ++c) sreg[i][c] = buf_load_s(rin, c < nrem ? voff : kOOB, (unsigned)(c * ihw) * 4u);
            }
        }
    };
    auto store_item = [&](int i, unsigned char* dstbuf) {
        if (it_oct[i] < CHS / 8) {
            u32x4 v[NPL];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                unsigned pl[NPL];
                if (SCALED) split_pack2<ET, NPL>(sreg[i][2 * c] * xs, sreg[i][2 * c + 1] * xs, pl);
                else split_pack2<ET, NPL>(sreg[i][2 * c], sreg[i][2 * c + 1], pl);
#pragma unroll
                for (int q = 0; q < NPL; ++q) v[q][c] = pl[q];
            }
#pragma unroll
            for (int q = 0; q < NPL; ++q) *reinterpret_cast<u32x4*>(dstbuf + q * plane_bytes + it_lds[i]) = v[q];
        }
    };

    const unsigned wlane = (unsigned)(min(n0 + wn * 32 + l31, g.Nld - 1) * 32 + lh * 16);
    const unsigned wstep = (unsigned)(NPL * g.Nld * 32);
    // k-step of the packed weights for (stage, tap, 16-channel sub-step): taps are 2 k-steps apart, the sub-step is the low bit
    auto qof = [&](const Stage& S, int tap, int hh) { return S.wq + tap * 2 + (HS == 2 ? hh : 0); };
    u32x4 wf[WD][NPL];
    auto load_w = [&](int slot, int q) {
        const unsigned so = (unsigned)min(q, pp.nsteps - 1) * wstep;
#pragma unroll
        for (int p = 0; p < NPL; ++p) wf[slot][p] = buf_load_u4s(rwk, wlane, so + (unsigned)(p * g.Nld * 32));
    };
    // the first stage's patch and the first weight fragments are requested HERE, ahead of the fragment addressing below: the HBM latency
    // of a workgroup's first loads then runs under that index math instead of after it (conv_p16.hip, round 4: -6..-13 % per kernel)
    Stage cs = stage_of(stage_beg);
    if (stage_beg < stage_end) {
        load_patch(cs);
#pragma unroll
        for (int d = 0; d < WD; ++d) load_w(d, qof(cs, d / HS, d % HS));       // (WD <= k-steps per stage)
    }
    __builtin_amdgcn_sched_barrier(0);
    unsigned lbase[TM];
#pragma unroll
    for (int t = 0; t < TM; ++t) {
        const int l = (wm * TM + t) * 32 + l31;
        const int tx = l & ((1 << twl) - 1), ty = (l >> twl) & ((1 << thl) - 1), tb = l >> (twl + thl);
        lbase[t] = (unsigned)(((tb * pp.PH + ty) * pp.PW + tx) * PB + lh * 16);
    }

    f32x16 acc[TM];
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    if (stage_beg < stage_end) {
#pragma unroll
        for (int i = 0; i < NI; ++i) store_item(i, patch2 + (stage_beg & 1) * buf_bytes);
        lds_barrier();
        for (int stage = stage_beg; stage < stage_end; ++stage) {
            const bool more = stage + 1 < stage_end;
            const Stage ns = more ? advance(cs) : cs;
            const unsigned char* cur = patch2 + (stage & 1) * buf_bytes;
            unsigned char* nxt = patch2 + ((stage + 1) & 1) * buf_bytes;
#if !defined(AGAN_PATCH_ABLATE) || AGAN_PATCH_ABLATE != 2
            if (more) load_patch(ns);
#endif
            // pixel fragments are read ONE k-step ahead of the MFMAs that consume them (LDS latency and bank conflicts off the
            // critical path); sched_group_barrier keeps that order in the emitted code
            u32x4 af[2][TM][NPL];
#pragma unroll
            for (int m = 0; m < TM; ++m)
#pragma unroll
                for (int p = 0; p < NPL; ++p)
                    af[0][m][p] = *reinterpret_cast<const u32x4*>(cur + p * plane_bytes + lbase[m] + (unsigned)(pp.tappos[0] * PB));
#pragma unroll
            for (int u = 0; u < SPS; ++u) {
                u32x4 w[NPL];
#pragma unroll
                for (int p = 0; p < NPL; ++p) w[p] = wf[u % WD][p];
                {   // weights WD k-steps ahead (possibly in the next stage; past the last stage: a harmless repeat)
                    const int u2 = u + WD;
#if !defined(AGAN_PATCH_ABLATE) || AGAN_PATCH_ABLATE != 3
                    if (u2 < SPS) load_w(u % WD, qof(cs, u2 / HS, u2 % HS));
                    else load_w(u % WD, qof(ns, (u2 - SPS) / HS, (u2 - SPS) % HS));
#endif
                }
#if defined(AGAN_PATCH_ABLATE) && AGAN_PATCH_ABLATE == 5
                if (false) {
#else
                if (u + 1 < SPS) {
#endif
                    const int t1 = (u + 1) / HS, h1 = (u + 1) % HS;
#pragma unroll
                    for (int m = 0; m < TM; ++m)
#pragma unroll
                        for (int p = 0; p < NPL; ++p)
                            af[(u + 1) & 1][m][p] = *reinterpret_cast<const u32x4*>(cur + p * plane_bytes + lbase[m] + (unsigned)(pp.tappos[t1] * PB) + h1 * 32);
                }
#if defined(AGAN_PATCH_ABLATE) && AGAN_PATCH_ABLATE == 5
#pragma unroll
                for (int m = 0; m < TM; ++m) acc[m] = mfma_split<ET, NPL>(w, af[0][m], acc[m]);
#elif defined(AGAN_PATCH_ABLATE) && AGAN_PATCH_ABLATE == 6
                // interleave the tiles' product chains: consecutive MFMAs never depend on each other
                if (NPL == 3) {
#pragma unroll
                    for (int m = 0; m < TM; ++m) acc[m] = mfma16<0>(w[2], af[u & 1][m][0], acc[m]);
#pragma unroll
                    for (int m = 0; m < TM; ++m) acc[m] = mfma16<0>(w[0], af[u & 1][m][2], acc[m]);
#pragma unroll
                    for (int m = 0; m < TM; ++m) acc[m] = mfma16<0>(w[1], af[u & 1][m][1], acc[m]);
#pragma unroll
                    for (int m = 0; m < TM; ++m) acc[m] = mfma16<0>(w[1], af[u & 1][m][0], acc[m]);
#pragma unroll
                    for (int m = 0; m < TM; ++m) acc[m] = mfma16<0>(w[0], af[u & 1][m][1], acc[m]);
#pragma unroll
                    for (int m = 0; m < TM; ++m) acc[m] = mfma16<0>(w[0], af[u & 1][m][0], acc[m]);
                } else {
#pragma unroll
                    for (int m = 0; m < TM; ++m) acc[m] = mfma_split<ET, NPL>(w, af[u & 1][m], acc[m]);
                }
#else
#pragma unroll
                for (int m = 0; m < TM; ++m) acc[m] = mfma_split<ET, NPL>(w, af[u & 1][m], acc[m]);
#endif
                __builtin_amdgcn_sched_group_barrier(0x100, TM * NPL, 0);                                  // next step's LDS reads first ...
                __builtin_amdgcn_sched_group_barrier(0x008, TM * (NPL == 1 ? 1 : (NPL == 2 ? 3 : 6)), 0);   // ... then this step's MFMAs
#if !defined(AGAN_PATCH_ABLATE) || AGAN_PATCH_ABLATE != 1
                if (u >= U0 && more) store_item(u - U0, nxt);      // next stage's patch, one item per k-step, behind the MFMAs
#endif
            }
            if (WD == 2 && (SPS & 1)) {        // an odd number of k-steps per stage flips the parity of the two-slot weight ring: swap it back
#pragma unroll
                for (int p = 0; p < NPL; ++p) { const u32x4 tmp = wf[0][p]; wf[0][p] = wf[1][p]; wf[1][p] = tmp; }
            }
#if !defined(AGAN_PATCH_ABLATE) || AGAN_PATCH_ABLATE != 4
            lds_barrier();
#endif
            cs = ns;
        }
    }

    // ---- epilogue ----
    const size_t ohw = (size_t)g.OH * g.OW;
    float* dst = (ksplit > 1) ? out + (size_t)split * slab : out;
    const __amdgpu_buffer_rsrc_t rout = make_rsrc(dst, (size_t)g.B * g.Cout * ohw * sizeof(float));
    const bool add_bias = (bias != nullptr) && (ksplit == 1);
    const bool lrelu = (act == AGAN_ACT_LRELU) && (ksplit == 1);
    const bool masked = (lrelu_mask != nullptr) && (ksplit == 1);
    const __amdgpu_buffer_rsrc_t rmask = make_rsrc(masked ? lrelu_mask : dst, (size_t)g.B * g.Cout * ohw * sizeof(float));
    const int nw = n0 + wn * 32;
    const bool nfull = nw + 32 <= g.Cout;
    const __amdgpu_buffer_rsrc_t rbias = make_rsrc(bias ? bias : dst, (size_t)g.Cout * sizeof(float));
    float omax = 0.f;
#pragma unroll
    for (int t = 0; t < TM; ++t) {
        const int l = (wm * TM + t) * 32 + l31;
        const int tx = l & ((1 << twl) - 1), ty = (l >> twl) & ((1 << thl) - 1), tb = l >> (twl + thl);
        const int b = tb0 + tb, yq = ty0 + ty, xq = tx0 + tx;
        const bool pvalid = (b < g.B) & (yq < g.OHs) & (xq < g.OWs);
        const unsigned pixoff = (unsigned)(b * g.Cout + nw + 4 * lh) * (unsigned)ohw + (unsigned)((yq * g.OS + py) * g.OW + (xq * g.OS + px));
        const unsigned voff = pvalid ? pixoff * 4u : kOOB;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int nr = (r & 3) + 8 * (r >> 2);
            float v = acc[t][r];
            if (SCALED) v *= unscale;
            if (add_bias) v += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rbias, (unsigned)(nw + 4 * lh) * 4u, (unsigned)nr * 4u, 0));
            if (lrelu) v = v > 0.f ? v : 0.2f * v;
            const unsigned so = (unsigned)nr * (unsigned)ohw * 4u;
            const unsigned vo = (nfull || (nw + nr + 4 * lh < g.Cout)) ? voff : kOOB;
            if (masked) v = buf_load_s(rmask, vo, so) > 0.f ? v : 0.2f * v;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rout, vo, so, 0);
            omax = fmaxf(omax, vo == kOOB ? 0.f : fabsf(v));
        }
    }
    if (out_amax != nullptr && ksplit == 1) {       // (a split launch: the slab sum commits it)
        __syncthreads();                            // every wave is done with the patch: its LDS hosts the 4-float reduction
        amax_commit_lds(omax, out_amax, reinterpret_cast<float*>(patch2));
    }
}

// ================================================================================================
// weight gradient:  dw[cls][cout][(phase, tap, ci)] = sum over the class lattice  dy[pixel][cout] * x[pixel + tap][ci]
// ================================================================================================
// The contraction runs over pixels, 16 consecutive pixels of a tile row per MFMA.  Both operands are staged exactly like the
// gather kernel's patch -- [position][channel] 16-bit planes, written with ds_write_b128 from coalesced NCHW loads -- the input
// as the tile's patch (tile + halo, 32 channels), dy as the tile itself (BJ output channels); the MFMA wants them
// [channel][8 consecutive pixels], which is what gfx950's transposing LDS read delivers: ds_read_b64_tr_b16 hands lane i of a
// 16-lane group column i of a 4-pixel x 16-channel block, so two of them are one operand fragment and a tap is, again, just
// a constant offset into the patch (cdna_hip_programming.md T10).  One workgroup owns (cout tile, 32-channel chunk, parity
// class, input phase, pixel split) and keeps all taps of its chunk in accumulators: D[cout][channel] per tap, i.e. the result
// is written as dw[cout][tap][ci] rows of 32 consecutive channels and put into OIHW order by the unpack pass.
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4 ds_read_tr8(const unsigned char* lds, unsigned off0, unsigned off1) {
    // 8 pixels x (this lane's channel): the 4-pixel block at off0 and the next one at off1
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + off0));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + off1));
    const unsigned long long ua = __builtin_bit_cast(unsigned long long, a), ub = __builtin_bit_cast(unsigned long long, b);
    u32x4 r = {(unsigned)ua, (unsigned)(ua >> 32), (unsigned)ub, (unsigned)(ub >> 32)};
    return r;
}

// ---- staging loads of the weight-gradient kernel: straight into ACCUMULATOR registers ----------------------------------------
// The kernel holds 72-104 staged dwords per lane across its k loop next to the fragment double buffers; left to the register
// allocator they start in arch VGPRs and are copied to AGPRs to make room (v_accvgpr_write needs the DATA, so every load was waited
// for right after issue and nothing overlapped the k loop: ablation, round 2).  gfx90a+ VMEM can write AGPRs directly: the loads
// are issued by inline asm with an "a" destination; the compiler does not track them, so consumers go through staged_wait().
__device__ __forceinline__ u32x4 make_rsrc4(const void* p, size_t bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(p);
    u32x4 r = {(unsigned)a, (unsigned)(a >> 32) & 0xFFFFu, (unsigned)bytes, 0x00020000u};
    return r;
}
__device__ __forceinline__ float buf_load_acc(u32x4 rsrc, unsigned voff, unsigned soff) {
    float v;
    asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=a"(v) : "v"(voff), "s"(rsrc), "s"(soff));
    return v;
}
__device__ __forceinline__ float buf_load_acc16(u32x4 rsrc, unsigned voff, unsigned soff) {      // one 16-bit value, zero-extended
    float v;
    asm volatile("buffer_load_ushort %0, %1, %2, %3 offen" : "=a"(v) : "v"(voff), "s"(rsrc), "s"(soff));
    return v;
}
// a staged value as fp32: fp32 storage as it stands; 16-bit storage (type ET) widened exactly
template <int ET, bool S16>
__device__ __forceinline__ float staged_f32(float v) {
    if (!S16) return v;
    const unsigned u = __float_as_uint(v);
    if (ET == 0) return __uint_as_float(u << 16);
    return (float)__builtin_bit_cast(_Float16, (unsigned short)u);
}
__device__ __forceinline__ void staged_wait(float (&v)[8]) {       // all staged loads have landed; ties the 8 values to the wait
    asm volatile("s_waitcnt vmcnt(0)" : "+a"(v[0]), "+a"(v[1]), "+a"(v[2]), "+a"(v[3]), "+a"(v[4]), "+a"(v[5]), "+a"(v[6]), "+a"(v[7]));
}

// LDS row strides of the weight-gradient kernel.  ds_read_b64_tr_b16 serves 16 lanes per cycle: 4 consecutive pixels x 4 column groups
// of 8 bytes, so the four pixel rows must start 8 banks wide windows that do not overlap: stride/4 mod 64 in {24, 72 = 8, 36}.  The
// gather's 80-byte positions (20 banks: 0, 20, 40, 60 -> the last window wraps onto the first) and the first version's 336-byte dy rows
// cost 39 % of the LDS cycles in bank conflicts (round-2 PMC); 96 / 288 / 144 bytes are conflict-free for these reads and for the
// 8-lanes-per-cycle ds_write_b128 of the staging pass.
constexpr int kWgPos = 96;
constexpr int kWgPlane = kMaxPos * kWgPos;

template <int BJ> struct WgCfg {
    static constexpr int kYRow = BJ == 128 ? 288 : 144;       // bytes per pixel of the dy image (BJ x 16 bit + pad), see kWgPos
    static constexpr int kYPlane = 128 * kYRow;
    static constexpr int kYItems = 128 * (BJ / 8) / 256;      // (pixel, cout octet) staging items per thread
};

// X16 / Y16: x / dy are stored in the 16-bit type of ET instead of fp32 (include/agan.h: AGAN_DT_*; one-plane modes only)
// B16 (with X16 and Y16): ROW-BLOCK staging as in conv_p16.hip -- a lane loads 16 bytes = 8 consecutive pixels of one channel, for 8
// channels, transposes the block in registers and writes 8 [position][8 channels] items: 16 vector-memory instructions per thread
// and tile instead of 104 (issuing those was 31 % of this kernel's time, round-2 clock64 breakdown).
template <int ET, int NPL, int BJ, int NT, bool X16 = false, bool Y16 = false, bool B16 = false>
__global__ __launch_bounds__(256) void conv_patch_wgrad_kernel(const void* __restrict__ x, const void* __restrict__ dy, float* __restrict__ dst,
                                                               const Geom g, const PatchPlan pp, const int psplit, const int tiles_per_split,
                                                               const size_t slab, const float* __restrict__ x_scale,
                                                               const float* __restrict__ dy_scale) {
    constexpr bool SCALED = (ET == 1 && NPL == 2);               // AGAN_PREC_F16X3 (x_scale / dy_scale: amax slots)
    float xsc = 1.f, ysc = 1.f, unscale = 1.f;
    if (SCALED) {
        float ix, iy;
        xsc = amax_scale(amax_read(x_scale), &ix);
        ysc = amax_scale(amax_read(dy_scale), &iy);
        unscale = ix * iy;
    }
    using C = WgCfg<BJ>;
    constexpr int WJ = BJ / 32;                // waves along output channels
    constexpr int WT = 4 / WJ;                 // waves sharing the taps of a cout block
    constexpr int TPW = (NT + WT - 1) / WT;    // taps per wave
    __shared__ __attribute__((aligned(16))) unsigned char lds[NPL * (kWgPlane + C::kYPlane)];
    unsigned char* const xs = lds;
    unsigned char* const ys = lds + NPL * kWgPlane;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wj = wave % WJ, wt = wave / WJ;
    // workgroup -> (cout tile, chunk x phase, class, pixel split); F = (((split * classes + class) * stages + stage) * jtiles + cout tile:
    // the cout tiles of one (pixel range, stage) share the patch, the stages share dy -- neighbours in F run on one XCD
    int jt, stage, cls, split;
    {
        const int jtiles = gridDim.x, nst = gridDim.y, ncls = gridDim.z / psplit;
        int F = xcd_contiguous(linear_block_id(), jtiles * nst * (int)gridDim.z);
        jt = F % jtiles; F /= jtiles;
        stage = F % nst; F /= nst;
        cls = F % ncls;  split = F / ncls;
    }
    const int py = cls / g.OS, px = cls - py * g.OS;
    const int chunk = stage / pp.NPH, ph = stage - chunk * pp.NPH;
    const int j0 = jt * BJ, c0 = chunk * kCH;
    const int twl = pp.twl, thl = pp.thl;
    const int ihw = g.IH * g.IW, ohw = g.OH * g.OW;
    const int by = pp.base_y[py][ph >> 1], bx = pp.base_x[px][ph & 1];
    static_assert(!(X16 || Y16) || NPL == 1, "16-bit activation storage goes with the one-plane modes");
    constexpr unsigned XE = X16 ? 2u : 4u, YE = Y16 ? 2u : 4u;      // bytes per stored element
    const u32x4 rx = make_rsrc4(x, (size_t)g.B * g.Cin * ihw * XE);
    const u32x4 rdy = make_rsrc4(dy, (size_t)g.B * g.Cout * ohw * YE);
    const int tile_beg = split * tiles_per_split, tile_end = min(pp.mtiles, tile_beg + tiles_per_split);

    // ---- staging items (tile independent parts) ----
    int xi_j[kItems], xi_i[kItems], xi_b[kItems], xi_oct[kItems];
    unsigned xi_lds[kItems];
#pragma unroll
    for (int i = 0; i < kItems; ++i) {
        const int e = tid + i * 256;
        const int oct = pp.dPP.div(e), p = e - oct * pp.PP;
        const int pb = pp.dPHW.div(p), rem = p - pb * pp.PHW;
        xi_j[i] = pp.dPW.div(rem);
        xi_i[i] = rem - xi_j[i] * pp.PW;
        xi_b[i] = pb;
        xi_oct[i] = oct;
        xi_lds[i] = (unsigned)(p * kWgPos + oct * 16);
    }
    float xr[kItems][8], yr[C::kYItems][8];

    // A wave can have 63 loads outstanding (vmcnt is 6 bits).  The 72-104 dword loads of a tile are therefore issued in TWO bursts:
    // part 0 (the x patch and the first two dy items, 56 loads) before the k loop, part 1 (the rest of dy) in the middle of it, when
    // part 0 has landed.  Issued as ONE burst the wave stalls at the 63rd load for a full memory latency, and with one workgroup per
    // CU nothing else runs meanwhile (ablation, round 2: half of the kernel's time).  part -1 = everything (prologue).
    auto load_tile = [&](int mt, int part) {
        const int txi = mt % pp.tiles_x, tyi = (mt / pp.tiles_x) % pp.tiles_y, tbi = mt / (pp.tiles_x * pp.tiles_y);
        const int tb0 = tbi << (7 - twl - thl), ty0 = tyi << thl, tx0 = txi << twl;
        if (part <= 0) {
#pragma unroll
            for (int i = 0; i < kItems; ++i) {
                const int b = tb0 + xi_b[i];
                const int iy = pp.IS * (ty0 + xi_j[i]) + by, ix = pp.IS * (tx0 + xi_i[i]) + bx;
                const bool ok = (xi_oct[i] < kCH / 8) & (b < g.B) & ((unsigned)iy < (unsigned)g.IH) & ((unsigned)ix < (unsigned)g.IW);
                const unsigned voff = ok ? (unsigned)((b * g.Cin + c0 + xi_oct[i] * 8) * ihw + iy * g.IW + ix) * XE : kOOB;
                const int nrem = g.Cin - c0 - xi_oct[i] * 8;
#pragma unroll
                for (int c = 0; c < 8; ++c)
                    xr[i][c] = X16 ? buf_load_acc16(rx, c < nrem ? voff : kOOB, (unsigned)(c * ihw) * XE)
                                   : buf_load_acc(rx, c < nrem ? voff : kOOB, (unsigned)(c * ihw) * XE);
            }
        }
        constexpr int kYFirst = 2;                      // dy items that ride with part 0
#pragma unroll
        for (int i = 0; i < C::kYItems; ++i) {
            const bool mine = part < 0 || (part == 0 ? i < kYFirst : i >= kYFirst);
            if (mine) {
                const int e = tid + i * 256;
                const int l = e & 127, oct = e >> 7;
                const int tx = l & ((1 << twl) - 1), ty = (l >> twl) & ((1 << thl) - 1), tb = l >> (twl + thl);
                const int b = tb0 + tb, yq = ty0 + ty, xq = tx0 + tx;
                const bool ok = (b < g.B) & (yq < g.OHs) & (xq < g.OWs);
                const int n = j0 + oct * 8;
                const unsigned voff = ok ? (unsigned)((b * g.Cout + n) * ohw + (yq * g.OS + py) * g.OW + (xq * g.OS + px)) * YE : kOOB;
                const int nrem = g.Cout - n;
#pragma unroll
                for (int c = 0; c < 8; ++c)
                    yr[i][c] = Y16 ? buf_load_acc16(rdy, c < nrem ? voff : kOOB, (unsigned)(c * ohw) * YE)
                                   : buf_load_acc(rdy, c < nrem ? voff : kOOB, (unsigned)(c * ohw) * YE);
            }
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < kItems; ++i) staged_wait(xr[i]);
#pragma unroll
        for (int i = 0; i < C::kYItems; ++i) staged_wait(yr[i]);
#pragma unroll
        for (int i = 0; i < kItems; ++i) {
            if (xi_oct[i] < kCH / 8) {
                u32x4 v[NPL];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    unsigned pl[NPL];
                    if (SCALED) split_pack2<ET, NPL>(xr[i][2 * c] * xsc, xr[i][2 * c + 1] * xsc, pl);
                    else split_pack2<ET, NPL>(staged_f32<ET, X16>(xr[i][2 * c]), staged_f32<ET, X16>(xr[i][2 * c + 1]), pl);
#pragma unroll
                    for (int q = 0; q < NPL; ++q) v[q][c] = pl[q];
                }
#pragma unroll
                for (int q = 0; q < NPL; ++q) *reinterpret_cast<u32x4*>(xs + q * kWgPlane + xi_lds[i]) = v[q];
            }
        }
#pragma unroll
        for (int i = 0; i < C::kYItems; ++i) {
            const int e = tid + i * 256;
            const int l = e & 127, oct = e >> 7;
            u32x4 v[NPL];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                unsigned pl[NPL];
                if (SCALED) split_pack2<ET, NPL>(yr[i][2 * c] * ysc, yr[i][2 * c + 1] * ysc, pl);
                else split_pack2<ET, NPL>(staged_f32<ET, Y16>(yr[i][2 * c]), staged_f32<ET, Y16>(yr[i][2 * c + 1]), pl);
#pragma unroll
                for (int q = 0; q < NPL; ++q) v[q][c] = pl[q];
            }
#pragma unroll
            for (int q = 0; q < NPL; ++q) *reinterpret_cast<u32x4*>(ys + q * C::kYPlane + l * C::kYRow + oct * 16) = v[q];
        }
    };

    // ---- row-block staging (B16): one x item and one dy item per thread, geometry independent of the tile ----
    // x item = (channel octet, image of the tile, patch row, 16-byte block): the blocks are aligned in the TENSOR's columns and cover
    // the patch row's span; pixel q of a block is patch column bx_i0 + q (stride-1 convs) or, for a stride-2 phase, every second pixel
    // (those of the phase's parity).  dy item = (cout octet, 8-pixel chunk of the tile).
    constexpr int PXB = 8;
    const int spanx = pp.IS * (pp.PW - 1) + 1;                       // tensor columns a patch row spans
    const int nxb = (PXB - 1 + spanx + PXB - 1) / PXB;               // blocks that cover it whatever the alignment of its origin
    int bx_row = 0, bx_pb = 0, bx_oct = kCH / 8, bx_xb = 0;
    unsigned bx_lds0 = 0;
    u32x4 xb16[8], yb16[8];
    if (B16) {
        const int e = tid;
        const int t1 = e / nxb;
        bx_xb = e - t1 * nxb;
        const int t2 = t1 / pp.PH;
        bx_row = t1 - t2 * pp.PH;
        const int tb_n = 128 >> (twl + thl);
        const int o = t2 / tb_n;
        bx_pb = t2 - o * tb_n;
        bx_oct = o < kCH / 8 ? o : kCH / 8;                          // (>= kCH/8: this thread has no x item)
        bx_lds0 = (unsigned)(((bx_pb * pp.PH + bx_row) * pp.PW) * kWgPos + o * 16);
    }
    auto load_tile16 = [&](int mt) {
        const int txi = mt % pp.tiles_x, tyi = (mt / pp.tiles_x) % pp.tiles_y, tbi = mt / (pp.tiles_x * pp.tiles_y);
        const int tb0 = tbi << (7 - twl - thl), ty0 = tyi << thl, tx0 = txi << twl;
        {   // x
            const int x0 = pp.IS * tx0 + bx, gx0 = x0 & ~(PXB - 1);
            const int b = tb0 + bx_pb, iy = pp.IS * (ty0 + bx_row) + by, gx = gx0 + bx_xb * PXB;
            const int ch = c0 + bx_oct * 8;
            const bool ok = (bx_oct < kCH / 8) & (ch < g.Cin) & (b < g.B) & ((unsigned)iy < (unsigned)g.IH) & ((unsigned)gx < (unsigned)g.IW);
            const unsigned voff = ok ? (unsigned)(((b * g.Cin + ch) * g.IH + iy) * g.IW + gx) * 2u : kOOB;
            const __amdgpu_buffer_rsrc_t r = make_rsrc(x, (size_t)g.B * g.Cin * ihw * 2u);
#pragma unroll
            for (int c = 0; c < 8; ++c) xb16[c] = buf_load_u4s(r, voff, (unsigned)(c * ihw) * 2u);
        }
        {   // dy: thread = (cout octet, chunk); BJ/8 octets x 16 chunks
            const int oct = tid >> 4, pc = tid & 15;
            const int l = pc * 8;
            const int tx = l & ((1 << twl) - 1), ty = (l >> twl) & ((1 << thl) - 1), tb = l >> (twl + thl);
            const int b = tb0 + tb, yq = ty0 + ty, xq = tx0 + tx;
            const int n = j0 + oct * 8;
            const bool ok = (oct < BJ / 8) & (n < g.Cout) & (b < g.B) & (yq < g.OH) & (xq < g.OW);
            const unsigned voff = ok ? (unsigned)(((b * g.Cout + n) * g.OH + yq) * g.OW + xq) * 2u : kOOB;
            const __amdgpu_buffer_rsrc_t r = make_rsrc(dy, (size_t)g.B * g.Cout * ohw * 2u);
#pragma unroll
            for (int c = 0; c < 8; ++c) yb16[c] = buf_load_u4s(r, voff, (unsigned)(c * ohw) * 2u);
        }
    };
    auto store_tile16 = [&](int mt) {
        const int txi = mt % pp.tiles_x;
        const int tx0 = txi << twl;
        const int x0 = pp.IS * tx0 + bx, gx0 = x0 & ~(PXB - 1);
        const int i0 = gx0 + bx_xb * PXB - x0;                          // tensor-column distance of pixel 0 of the block from patch column 0
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int d = q >> 1;
            const unsigned sel = (q & 1) ? 0x07060302u : 0x05040100u;
            u32x4 vx, vy;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                vx[k] = __builtin_amdgcn_perm(xb16[2 * k + 1][d], xb16[2 * k][d], sel);
                vy[k] = __builtin_amdgcn_perm(yb16[2 * k + 1][d], yb16[2 * k][d], sel);
            }
            const int dcol = i0 + q;                                    // in tensor columns
            const int col = pp.IS == 1 ? dcol : (dcol >> 1);
            const bool okx = (bx_oct < kCH / 8) & (dcol >= 0) & (col < pp.PW) & (pp.IS == 1 || (dcol & 1) == 0);
            if (okx) *reinterpret_cast<u32x4*>(xs + bx_lds0 + (unsigned)(col * kWgPos)) = vx;
            if ((tid >> 4) < BJ / 8)
                *reinterpret_cast<u32x4*>(ys + ((tid & 15) * 8 + q) * C::kYRow + (tid >> 4) * 16) = vy;
        }
    };

    // ---- transposed-read lane addresses: 16-lane group gq reads pixels 8*(gq>>1) + 4*e + q (q = (lane&15)>>2) for the 16 channels
    //      16*(gq&1) + ..., lane (lane&3) supplying columns 4*(lane&3)..+3 of row q
    const int gq = lane >> 4, li = lane & 15, rq = li >> 2, cp = li & 3;
    const int lpix = 8 * (gq >> 1) + rq;                                         // this lane's pixel within the 16 of a step (first block)
    const unsigned xcol = (unsigned)((16 * (gq & 1) + 4 * cp) * 2);
    const unsigned ylane = (unsigned)(lpix * C::kYRow + (wj * 32 + 16 * (gq & 1) + 4 * cp) * 2);

    f32x16 acc[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    if (tile_beg < tile_end) {
        if (B16) { load_tile16(tile_beg); store_tile16(tile_beg); }
        else { load_tile(tile_beg, -1); store_tile(); }
        __syncthreads();
        const int TW = 1 << twl;
        constexpr int nsteps = 8;                                 // 128 pixels / 16
        // tap offsets of this wave in registers (pp lives in the kernarg segment: indexing it by a runtime tap is a scalar
        // load + s_waitcnt lgkmcnt(0) -- which also drains the LDS queue -- per tap and k-step)
        constexpr bool kAllTaps = (TPW * WT == NT);               // every wave owns TPW valid taps (else the last wave's tail is masked)
        unsigned toff[TPW];
        bool tok[TPW];
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            const int tap = wt * TPW + t;
            tok[t] = kAllTaps || tap < NT;
            toff[t] = (unsigned)(pp.tappos[tap < NT ? tap : NT - 1] * kWgPos);
        }
        // 16 consecutive lattice points of the tile per k-step: l = 16*kk .. 16*kk+15  (a tile row holds 1, 2, 4 ... of them; tiles
        // narrower than 16 wrap into the next row / image, which the position arithmetic follows per 4-pixel block: a block never
        // straddles a tile row, rows are >= 4 wide)
        unsigned xc_tile = xcol;
        auto xpos = [&](int kk, unsigned (&xo)[2]) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int l = kk * 16 + lpix + 4 * e;
                const int tx = l & (TW - 1), ty = (l >> twl) & ((1 << thl) - 1), tb = l >> (twl + thl);
                xo[e] = xc_tile + (unsigned)(((tb * pp.PH + ty) * pp.PW + tx) * kWgPos);
            }
        };
        auto read_a = [&](int kk, u32x4 (&a)[NPL]) {
            const unsigned yo = ylane + (unsigned)(kk * 16 * C::kYRow);
#pragma unroll
            for (int p = 0; p < NPL; ++p) a[p] = ds_read_tr8(ys + p * C::kYPlane, yo, yo + 4 * C::kYRow);
        };
        auto read_b = [&](const unsigned (&xo)[2], int t, u32x4 (&bb)[NPL]) {
#pragma unroll
            for (int p = 0; p < NPL; ++p) bb[p] = ds_read_tr8(xs + p * kWgPlane, xo[0] + toff[t], xo[1] + toff[t]);
        };
#ifdef AGAN_WG_TIMING
        long long tph[6] = {0, 0, 0, 0, 0, 0};
#define AGAN_TICK(k) do { const long long now_ = clock64(); tph[k] += now_ - tlast; tlast = now_; } while (0)
#else
#define AGAN_TICK(k) do { } while (0)
#endif
        for (int mt = tile_beg; mt < tile_end; ++mt) {
#ifdef AGAN_WG_TIMING
            long long tlast = clock64();
#endif
            const bool more = mt + 1 < tile_end;
#if !defined(AGAN_WG_ABLATE) || AGAN_WG_ABLATE != 1
            if (more) { if (B16) load_tile16(mt + 1); else load_tile(mt + 1, 0); }
#endif
            AGAN_TICK(0);
            // (opaque per tile: otherwise all 8 x TPW x 2 fragment addresses of the unrolled loop are hoisted out of the tile loop and
            //  held in VGPRs next to the 72-104 staging registers -- recomputing them costs a few dozen VALU adds per tile)
            asm volatile("" : "+v"(xc_tile));
            // software pipeline over (k-step, group of G taps) items, fully unrolled: the fragments of item i+1 are read while the
            // MFMAs of item i run -- with ONE wave per SIMD (LDS: one workgroup per CU) nothing else hides the ~130-cycle LDS
            // latency, and read-then-wait per tap left the matrix core idle two thirds of the loop (round-2 PMC: 17 % MFMA-busy).
            // Both fragment sets are live at once (distinct registers by construction); sched_group_barrier interleaves the reads
            // of the next item with the MFMAs of this one, tap by tap.
            constexpr int G = (NPL == 1 || TPW <= 5) ? TPW : 3, NG = TPW / G;
            static_assert(TPW % G == 0, "tap groups must tile the taps of a wave");
            u32x4 a[2][NPL], bf[2][G][NPL];
            unsigned xo[2][2];
            xpos(0, xo[0]);
            read_a(0, a[0]);
#pragma unroll
            for (int q = 0; q < G; ++q) read_b(xo[0], q, bf[0][q]);
#if defined(AGAN_WG_ABLATE) && AGAN_WG_ABLATE == 3
            if (mt == tile_beg)
#endif
#pragma unroll
            for (int kk = 0; kk < nsteps; ++kk) {
#pragma unroll
                for (int tg = 0; tg < NG; ++tg) {
#if !defined(AGAN_WG_ABLATE) || AGAN_WG_ABLATE != 1
                    if (!B16 && kk == nsteps / 2 && tg == 0 && more) load_tile(mt + 1, 1);
#endif
                    const int cur = (kk * NG + tg) & 1, nxt = cur ^ 1;
                    const bool last = (kk + 1 == nsteps) && (tg + 1 == NG);
                    const bool newk = (tg + 1 == NG);                         // the next item starts the next k-step
                    if (!last) {
                        if (newk) {
                            xpos(kk + 1, xo[(kk + 1) & 1]);
                            read_a(kk + 1, a[(kk + 1) & 1]);
                        }
#pragma unroll
                        for (int q = 0; q < G; ++q) read_b(xo[newk ? (kk + 1) & 1 : kk & 1], (newk ? 0 : (tg + 1) * G) + q, bf[nxt][q]);
                    }
                    __builtin_amdgcn_sched_barrier(0);      // the reads above stay ABOVE this item's MFMAs (the scheduler sinks LDS loads to their use)
#pragma unroll
                    for (int q = 0; q < G; ++q)
                        if (kAllTaps || tok[tg * G + q]) acc[tg * G + q] = mfma_split<ET, NPL>(a[kk & 1], bf[cur][q], acc[tg * G + q]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            AGAN_TICK(1);
            __syncthreads();
            AGAN_TICK(2);
            if (more) {
#ifdef AGAN_WG_TIMING
#pragma unroll
                for (int i = 0; i < kItems; ++i) staged_wait(xr[i]);
#pragma unroll
                for (int i = 0; i < C::kYItems; ++i) staged_wait(yr[i]);
                AGAN_TICK(3);
#endif
#if !defined(AGAN_WG_ABLATE) || AGAN_WG_ABLATE != 2
                if (B16) store_tile16(mt + 1); else store_tile();
#endif
                AGAN_TICK(4);
                __syncthreads();
                AGAN_TICK(5);
            }
        }
#ifdef AGAN_WG_TIMING
        if (blockIdx.x == 0 && blockIdx.y == 1 && blockIdx.z == 3 && tid == 64)
            printf("wg timing (cycles over %d tiles): issue %lld kloop %lld bar1 %lld loadwait %lld convert+store %lld bar2 %lld\n", tile_end - tile_beg,
                   tph[0], tph[1], tph[2], tph[3], tph[4], tph[5]);
#endif
    }

    // ---- D[cout][channel of the chunk] per tap -> dst[split][cls][cout][kprime], kprime = ((chunk*NPH + ph)*NT + tap)*32 + ci ----
    const int Kp = pp.nstages * NT * kCH;
    float* o = dst + (size_t)split * slab + (size_t)cls * g.Cout * Kp;
    const __amdgpu_buffer_rsrc_t ro = make_rsrc(o, (size_t)g.Cout * Kp * sizeof(float));
    const int l31 = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int tap = wt * TPW + t;
        if (tap >= NT) break;
        const int kcol = (stage * NT + tap) * kCH + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = j0 + wj * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            buf_store(ro, n < g.Cout ? (unsigned)(n * Kp + kcol) * 4u : kOOB, SCALED ? acc[t][r] * unscale : acc[t][r]);
        }
    }
}

// dw[cout][ci][kh][kw] (OIHW) from the reduced [cls][cout][K'] result; UP mode folds the 4 classes' 2x2 taps back into 3x3.
// K' index = ((chunk * NPH + phase) * NT + tap) * 32 + ci%32.
__global__ __launch_bounds__(256) void unpack_patch_wgrad_kernel(const float* __restrict__ red, float* __restrict__ dw, int cout, int cin,
                                                                 int kh, int kw, int up, int NPH, int NT, int Kp, int accumulate) {
    const size_t total = (size_t)cout * cin * kh * kw;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int s = (int)(e % kw);
        size_t t = e / kw;
        const int r = (int)(t % kh); t /= kh;
        const int ci = (int)(t % cin), co = (int)(t / cin);
        const int chunk = ci / kCH, cc = ci - chunk * kCH;
        float v = 0.f;
        if (!up) {
            int ph = 0, tap;
            if (NPH == 1) tap = r * kw + s;
            else { ph = (r & 1) * 2 + (s & 1); tap = (r >> 1) * (kw / 2) + (s >> 1); }
            v = red[(size_t)co * Kp + ((chunk * NPH + ph) * NT + tap) * kCH + cc];
        } else {
            // 3x3 tap (r, s) was folded into tap (rr, ss) of class (py, px) whenever up_fwd_taps(py, rr) covers r and (px, ss) covers s
            for (int py = 0; py < 2; ++py)
                for (int rr = 0; rr < 2; ++rr) {
                    int rl, rh;
                    up_fwd_taps(py, rr, rl, rh);
                    if (r < rl || r > rh) continue;
                    for (int px = 0; px < 2; ++px)
                        for (int ss = 0; ss < 2; ++ss) {
                            int sl, sh;
                            up_fwd_taps(px, ss, sl, sh);
                            if (s < sl || s > sh) continue;
                            v += red[((size_t)(py * 2 + px) * cout + co) * Kp + (chunk * NT + rr * 2 + ss) * kCH + cc];
                        }
                }
        }
        dw[e] = accumulate ? dw[e] + v : v;
    }
}

// The same for the direct (non-folded) convs, FUSED with the sum over the pixel-split slabs and coalesced on both sides: a wave owns one
// (output channel, 32-channel chunk) tile, reads its kh*kw tap rows (32 consecutive floats each, from every slab) into LDS and writes the
// 32 * kh*kw floats of dw[co][chunk*32 ..][..][..], which are contiguous in OIHW.  (The element-wise kernel above reads with a 128-byte
// stride between lanes -- 1 of 16 fetched floats used -- and ran behind a separate slab-sum launch: together ~20 us of launch-bound
// work per layer, 10 % of the 16-bit-storage step in the round-3 profile.)
__global__ __launch_bounds__(256) void sum_unpack_wgrad_kernel(const float* __restrict__ slabs, int nslabs, size_t slab, float* __restrict__ dw,
                                                               int cout, int cin, int kh, int kw, int NPH, int NT, int Kp, int accumulate) {
    __shared__ float tile[4][16 * 33];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, l31 = lane & 31, half = lane >> 5;
    const int kk = kh * kw, nchunks = (cin + kCH - 1) / kCH;
    const int ntiles = cout * nchunks;
    for (int t0 = blockIdx.x * 4; t0 < ntiles; t0 += gridDim.x * 4) {
        const int tl = t0 + w;
        if (tl < ntiles) {                                       // (wave-uniform)
            const int co = tl / nchunks, chunk = tl - co * nchunks;
            const float* base = slabs + (size_t)co * Kp;
            for (int t = half; t < kk; t += 2) {
                const int r = t / kw, s = t - r * kw;
                int ph = 0, tap = t;
                if (NPH != 1) { ph = (r & 1) * 2 + (s & 1); tap = (r >> 1) * (kw / 2) + (s >> 1); }
                const float* p = base + ((chunk * NPH + ph) * NT + tap) * kCH + l31;
                float v = p[0];
                if (nslabs > 1) {                              // (<= 4: the caller sums larger splits with the slab-sum kernel first)
                    const float v1 = p[slab], v2 = nslabs > 2 ? p[2 * slab] : 0.f, v3 = nslabs > 3 ? p[3 * slab] : 0.f;
                    v = (v + v1) + (v2 + v3);
                }
                tile[w][t * 33 + l31] = v;
            }
        }
        __syncthreads();
        if (tl < ntiles) {
            const int co = tl / nchunks, chunk = tl - co * nchunks;
            const int nci = min(kCH, cin - chunk * kCH);
            float* o = dw + ((size_t)co * cin + chunk * kCH) * kk;
            for (int j = lane; j < nci * kk; j += 64) {
                const int ci = j / kk, t = j - ci * kk;
                const float v = tile[w][t * 33 + ci];
                o[j] = accumulate ? o[j] + v : v;
            }
        }
        __syncthreads();
    }
}

// ================================================================================================
// weight packing: OIHW -> [class][k-step][plane][Nld][16] 16-bit; k-step = ((chunk * NPH + phase) * NT + tap) * 2 + half
// ================================================================================================
// One workgroup packs 16 output columns x one 32-channel chunk for every class, phase and tap: the OIHW block those touch
// ([16 co][32 ci][kh*kw] for the forward-type packs, [32 co][16 ci][kh*kw] for the data-gradient packs) is read with coalesced
// row segments into LDS once (tap-major, so the channel walk of the emit phase is conflict-free), and each (k-step, plane)
// leaves as 16 rows x 32 B = 512 contiguous bytes.  The (class, phase, tap) loops are plain counters: no per-element decode.
constexpr int kPackN = 16;
constexpr int kPackTile = kPackN * kCH + 1;        // floats per tap plane of the LDS tile (+1: the fill walks taps across lanes)
struct TileOIHW {
    const float* t;        // LDS tile [kk][A][Bsz]
    int co0, ci0, A, Bsz, kw;
    __device__ __forceinline__ float operator()(int co, int ci, int a, int b) const {
        const int ia = co - co0, ib = ci - ci0;
        return ((unsigned)ia < (unsigned)A && (unsigned)ib < (unsigned)Bsz) ? t[(a * kw + b) * kPackTile + ia * Bsz + ib] : 0.f;
    }
};
// element (gathered channel ch, tap (r, s) of the GATHER geometry, output column n) of a pack mode, read through W4
template <class W4>
__device__ __forceinline__ float packed_value_direct(W4 w4, int mode, int cls, int ch, int r, int s, int n, int kh, int kw) {
    switch (mode) {
        case AGAN_PACK_FWD: return w4(n, ch, r, s);
        case AGAN_PACK_DGRAD_S1: return w4(ch, n, kh - 1 - r, kw - 1 - s);
        case AGAN_PACK_DGRAD_4x4S2: {
            const int py = cls >> 1, px = cls & 1;
            return w4(ch, n, ((py + 1) & 1) + 2 * r, ((px + 1) & 1) + 2 * s);
        }
        case AGAN_PACK_UP_FWD: {
            int rl, rh, sl, sh;
            up_fwd_taps(cls >> 1, r, rl, rh);
            up_fwd_taps(cls & 1, s, sl, sh);
            float v = 0.f;
            for (int a = rl; a <= rh; ++a)
                for (int b = sl; b <= sh; ++b) v += w4(n, ch, a, b);
            return v;
        }
        default: {      // AGAN_PACK_UP_DGRAD
            int rl, rh, sl, sh;
            up_dgrad_taps(r, rl, rh);
            up_dgrad_taps(s, sl, sh);
            float v = 0.f;
            for (int a = rl; a <= rh; ++a)
                for (int b = sl; b <= sh; ++b) v += w4(ch, n, a, b);
            return v;
        }
    }
}
struct PackPlanLite { int IS, NPH, NT, nsteps; };
template <int ET, int NPL>
__device__ __forceinline__ void pack_patch_tile(const float* __restrict__ w, unsigned short* __restrict__ wk, int mode, int cout, int cin, int kh,
                                                int kw, int ncls, int Kin, int R, int S, int Nld, const PackPlanLite pp, const FastDiv drow,
                                                const FastDiv dkk, const int nblk, const int chunk, float* tile) {
    const int n0 = nblk * kPackN, c0 = chunk * kCH;
    const int kk = kh * kw;
    const bool ncout = pack_n_is_cout(mode);
    TileOIHW T;
    T.t = tile; T.kw = kw;
    if (ncout) { T.co0 = n0; T.A = kPackN; T.ci0 = c0; T.Bsz = kCH; }
    else       { T.co0 = c0; T.A = kCH; T.ci0 = n0; T.Bsz = kPackN; }
    // rows of the block are contiguous in OIHW: (Bsz * kk) floats starting at [co0 + a][ci0][0][0]
    const int rowlen = T.Bsz * kk;
    const int nelem = T.A * rowlen;
    if (((cin * kk) & 3) == 0) {
        // 16-byte loads: 4 consecutive floats of a row per lane (rows are 16-byte aligned when cin * kh * kw is a multiple of 4; the block's
        // row length Bsz * kk always is), a quarter of the load instructions and one row division per four elements.  The buffer range
        // zero-fills what a partial channel chunk would read past the tensor's end.
        const __amdgpu_buffer_rsrc_t rw = make_rsrc(w, (size_t)cout * cin * kk * sizeof(float));
        for (int i0 = threadIdx.x * 4; i0 < nelem; i0 += 256 * 4 * 4) {       // 4 independent 16-byte loads in flight per thread
            f32x4 v[4];
            int a_[4], rem_[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = i0 + j * 1024;
                const int a = drow.div(i), rem = i - a * rowlen;
                a_[j] = a; rem_[j] = rem;
                const int co = T.co0 + a;
                v[j] = buf_load4(rw, (i < nelem && co < cout) ? (unsigned)(((size_t)co * cin + T.ci0) * kk + rem) * 4u : kOOB);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (i0 + j * 1024 >= nelem) continue;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int rem = rem_[j] + e;
                    const int ib = dkk.div(rem), tap = rem - ib * kk;
                    tile[tap * kPackTile + a_[j] * T.Bsz + ib] = (T.ci0 + ib < cin) ? v[j][e] : 0.f;
                }
            }
        }
    } else
    for (int i0 = threadIdx.x; i0 < nelem; i0 += 256 * 8) {       // 8 independent loads in flight per thread
        float v[8];
        int dst[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = i0 + j * 256;
            const int a = drow.div(i), rem = i - a * rowlen;       // (multiply-high divisions: this loop is the kernel's ALU hot spot)
            const int ib = dkk.div(rem), tap = rem - ib * kk;
            const int co = T.co0 + a, ci = T.ci0 + ib;
            dst[j] = i < nelem ? tap * kPackTile + a * T.Bsz + ib : -1;
            v[j] = (i < nelem && co < cout && ci < cin) ? w[((size_t)co * cin + T.ci0) * kk + rem] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (dst[j] >= 0) tile[dst[j]] = v[j];
    }
    __syncthreads();
    const int nchan = Kin / (R * S);
    // emit: thread = (one of 4 consecutive (class, phase, tap) slots, row, channel octet): 8 channels -> one 16-byte store per plane
    const int sub = threadIdx.x >> 6, row = (threadIdx.x >> 2) & (kPackN - 1), oct = threadIdx.x & 3;
    const int half = oct >> 1;
    const int n = n0 + row;
    const int ch = c0 + oct * 8;
    const bool live = n < Nld;
    const bool nvalid = n < (ncout ? cout : cin);
    const int cols = pp.IS == 1 ? S : S / 2;                 // taps per phase row
    const int nslots = ncls * pp.NPH * pp.NT;
    for (int f0 = 0; f0 < nslots; f0 += 4) {
        const int f = f0 + sub;
        if (f >= nslots) break;                               // (wave-uniform: sub is the wave index)
        const int tap = f % pp.NT, g2 = f / pp.NT, ph = g2 % pp.NPH, cls = g2 / pp.NPH;
        const int tr = tap / cols, tc = tap - tr * cols;
        const int r = pp.IS == 1 ? tr : 2 * tr + (ph >> 1), sx = pp.IS == 1 ? tc : 2 * tc + (ph & 1);
        u32x4 o[NPL];
#pragma unroll
        for (int c2 = 0; c2 < 4; ++c2) {
            float v[2];
#pragma unroll
            for (int c = 0; c < 2; ++c)
                v[c] = (nvalid && ch + 2 * c2 + c < nchan) ? packed_value_direct(T, mode, cls, ch + 2 * c2 + c, r, sx, n, kh, kw) : 0.f;
            unsigned pl[NPL];
            // (f16x3: weights travel times 2^11; |w| >= 32 would leave fp16's range -- saturate at the largest finite fp16 instead of
            // packing an infinity that turns every output it touches into NaN)
            if (ET == 1 && NPL == 2) split_pack2<ET, NPL>(fminf(fmaxf(v[0] * kF16WeightScale, -65504.f), 65504.f),
                                                          fminf(fmaxf(v[1] * kF16WeightScale, -65504.f), 65504.f), pl);
            else split_pack2<ET, NPL>(v[0], v[1], pl);
#pragma unroll
            for (int p = 0; p < NPL; ++p) o[p][c2] = pl[p];
        }
        const int q = ((chunk * pp.NPH + ph) * pp.NT + tap) * 2 + half;
        if (live) {
#pragma unroll
            for (int p = 0; p < NPL; ++p)
                reinterpret_cast<u32x4*>(wk)[((((size_t)cls * pp.nsteps + q) * NPL + p) * Nld + n) * 2 + (oct & 1)] = o[p];
        }
    }
}

template <int ET, int NPL>
__global__ __launch_bounds__(256) void pack_patch_weight_kernel(const float* __restrict__ w, unsigned short* __restrict__ wk, int mode, int cout,
                                                                int cin, int kh, int kw, int ncls, int Kin, int R, int S, int Nld,
                                                                const PackPlanLite pp, const FastDiv drow, const FastDiv dkk) {
    __shared__ float tile[16 * kPackTile];
    pack_patch_tile<ET, NPL>(w, wk, mode, cout, cin, kh, kw, ncls, Kin, R, S, Nld, pp, drow, dkk, (int)blockIdx.x, (int)blockIdx.y, tile);
}

// Every stale pack of an optimiser in ONE launch (16-bit layouts): job i owns workgroups [first_block, first_block + blocks(i)), each a
// (16 output columns, 32-channel chunk) tile of pack_patch_tile.  A step re-packs ~100 weight tensors, most of them a few KB: launched one
// by one they cost 1.7 ms per step at the metric config (17 us each, launch-bound) -- 9 % of the fp16-split step.
template <int ET, int NPL>
__global__ __launch_bounds__(256) void pack_patch_jobs_kernel(const agan_pack_job* __restrict__ jobs, int njobs) {
    __shared__ float tile[16 * kPackTile];
    __shared__ int which;
    const int b = blockIdx.x;
    for (int t = threadIdx.x; t < njobs; t += 256) {
        const int lo = jobs[t].first_block, hi = t + 1 < njobs ? jobs[t + 1].first_block : 0x7fffffff;
        if (b >= lo && b < hi) which = t;
    }
    __syncthreads();
    const agan_pack_job j = jobs[which];
    const int local = b - j.first_block;
    int ncls, K, N, R, S, SY;
    pack_dims(j.mode, j.cout, j.cin, j.kh, j.kw, ncls, K, N);
    pack_taps(j.mode, j.kh, j.kw, R, S, SY);
    PackPlanLite pp;
    pp.IS = SY;
    pp.NPH = SY == 2 ? 4 : 1;
    pp.NT = SY == 2 ? (R / 2) * (S / 2) : R * S;
    const int Kin = K / (R * S);
    pp.nsteps = cdiv(Kin, kCH) * pp.NPH * pp.NT * 2;
    const int Nld = (N + 31) / 32 * 32;
    const int nbn = cdiv(Nld, kPackN);
    const FastDiv dkk = make_fastdiv((unsigned)(j.kh * j.kw));
    const FastDiv drow = make_fastdiv((unsigned)((pack_n_is_cout(j.mode) ? kCH : kPackN) * j.kh * j.kw));
    pack_patch_tile<ET, NPL>(j.w, static_cast<unsigned short*>(j.wk), j.mode, j.cout, j.cin, j.kh, j.kw, ncls, K, R, S, Nld, pp, drow, dkk,
                             local % nbn, local / nbn, tile);
}

template <int ET, int NPL, int BN>
void launch_nt(const float* in, const void* wk, const float* bias, float* dst, const Geom& g, const PatchPlan& pp_in, const PatchGather& p,
               int act, const float* mask, hipStream_t st, const float* in_scale, float* out_amax) {
    PatchPlan pp = pp_in;                 // + the multiply-high constants of this launch's tile decode
    pp.g_ntiles = p.ntiles; pp.g_ncls = p.ncls;
    pp.dNT = make_fastdiv((unsigned)p.ntiles); pp.dNCLS = make_fastdiv((unsigned)p.ncls); pp.dMT = make_fastdiv((unsigned)pp.mtiles);
    pp.dTX = make_fastdiv((unsigned)pp.tiles_x); pp.dTY = make_fastdiv((unsigned)pp.tiles_y);
    pp.dNPH = make_fastdiv((unsigned)pp.NPH); pp.dNPH2 = make_fastdiv((unsigned)(2 * pp.NPH));
    dim3 grid(pp.mtiles, p.ntiles, p.ncls * p.ksplit);
    const unsigned short* w = static_cast<const unsigned short*>(wk);
    constexpr int PB = (NPL >= 3 ? 16 : 32) * 2 + 16;
    const int plane = (pp.PP + 7) / 8 * 8 * PB;
    const size_t smem = (size_t)2 * NPL * plane;
    if (pp.NT == 9) {
        static const hipError_t a9 = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_patch_kernel2<ET, NPL, BN, 9>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)a9;
        AGAN_LAUNCH((conv_patch_kernel2<ET, NPL, BN, 9>), grid, dim3(256), smem, st, in, w, bias, dst, g, pp, p.ksplit, p.stages_per_split, p.slab, act, mask, plane, in_scale, out_amax);
    } else {
        static const hipError_t a4 = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_patch_kernel2<ET, NPL, BN, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)a4;
        AGAN_LAUNCH((conv_patch_kernel2<ET, NPL, BN, 4>), grid, dim3(256), smem, st, in, w, bias, dst, g, pp, p.ksplit, p.stages_per_split, p.slab, act, mask, plane, in_scale, out_amax);
    }
}
template <int ET, int NPL>
void launch_bn(const float* in, const void* wk, const float* bias, float* dst, const Geom& g, const PatchPlan& pp, const PatchGather& p,
               int act, const float* mask, hipStream_t st, const float* in_scale, float* out_amax) {
    if (p.bn == 128) launch_nt<ET, NPL, 128>(in, wk, bias, dst, g, pp, p, act, mask, st, in_scale, out_amax);
    else if (p.bn == 64) launch_nt<ET, NPL, 64>(in, wk, bias, dst, g, pp, p, act, mask, st, in_scale, out_amax);
    else launch_nt<ET, NPL, 32>(in, wk, bias, dst, g, pp, p, act, mask, st, in_scale, out_amax);
}

template <int ET, int NPL, int BJ, bool X16 = false, bool Y16 = false>
void launch_wg_nt(const void* x, const void* dy, float* part, const Geom& g, const PatchPlan& pp, const PatchWgrad& p, hipStream_t st,
                  const float* xs = nullptr, const float* ys = nullptr) {
    dim3 grid(p.jtiles, pp.nstages, p.ncls * p.psplit);
    if (pp.NT == 9)
        AGAN_LAUNCH((conv_patch_wgrad_kernel<ET, NPL, BJ, 9, X16, Y16>), grid, dim3(256), 0, st, x, dy, part, g, pp, p.psplit, p.tiles_per_split, p.slab, xs, ys);
    else
        AGAN_LAUNCH((conv_patch_wgrad_kernel<ET, NPL, BJ, 4, X16, Y16>), grid, dim3(256), 0, st, x, dy, part, g, pp, p.psplit, p.tiles_per_split, p.slab, xs, ys);
}
// one-plane modes with typed activation storage
// can the row-block staging take this weight gradient?  (8-pixel chunks inside tile rows, whole 16-byte blocks per tensor row, one x
// item per thread)
static bool wg_block16_ok(const Geom& g, const PatchPlan& pp) {
    if (pp.twl < 3 || g.OS != 1 || (g.OW & 7) || (g.IW & 7) || (g.Cin & 7) || (g.Cout & 7)) return false;
    const int spanx = pp.IS * (pp.PW - 1) + 1;
    const int nxb = (7 + spanx + 7) / 8;
    const int tb = 128 >> (pp.twl + pp.thl);
    static const bool off = getenv("AGAN_WG_B16_OFF") != nullptr;
    return !off && (kCH / 8) * tb * pp.PH * nxb <= 256;
}
template <int ET, int BJ>
void launch_wg_b16(const void* x, const void* dy, float* part, const Geom& g, const PatchPlan& pp, const PatchWgrad& p, hipStream_t st) {
    dim3 grid(p.jtiles, pp.nstages, p.ncls * p.psplit);
    if (pp.NT == 9)
        AGAN_LAUNCH((conv_patch_wgrad_kernel<ET, 1, BJ, 9, true, true, true>), grid, dim3(256), 0, st, x, dy, part, g, pp, p.psplit, p.tiles_per_split, p.slab, (const float*)nullptr, (const float*)nullptr);
    else
        AGAN_LAUNCH((conv_patch_wgrad_kernel<ET, 1, BJ, 4, true, true, true>), grid, dim3(256), 0, st, x, dy, part, g, pp, p.psplit, p.tiles_per_split, p.slab, (const float*)nullptr, (const float*)nullptr);
}
template <int ET, int BJ>
void launch_wg_dt(const void* x, const void* dy, float* part, const Geom& g, const PatchPlan& pp, const PatchWgrad& p, hipStream_t st, bool x16,
                  bool y16) {
    if (x16 && y16 && wg_block16_ok(g, pp)) launch_wg_b16<ET, BJ>(x, dy, part, g, pp, p, st);
    else if (x16 && y16) launch_wg_nt<ET, 1, BJ, true, true>(x, dy, part, g, pp, p, st);
    else if (x16) launch_wg_nt<ET, 1, BJ, true, false>(x, dy, part, g, pp, p, st);
    else if (y16) launch_wg_nt<ET, 1, BJ, false, true>(x, dy, part, g, pp, p, st);
    else launch_wg_nt<ET, 1, BJ>(x, dy, part, g, pp, p, st);
}

inline int pow2ceil_log(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

}  // namespace

namespace {
// max |x| folded into an amax slot (agan_common.h: amax_commit); the slot is zero on entry
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, size_t n, float* __restrict__ slot) {
    float m = 0.f;
    const size_t n4 = n / 4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) m = fmaxf(m, fabsf(x[n4 * 4 + threadIdx.x]));
    amax_commit(m, slot);
}
}  // namespace

extern "C" int agan_absmax(const float* x, size_t n, float* amax_slot, void* stream) {
    AGAN_REQUIRE(x && amax_slot && n > 0, "absmax: bad argument");
    AGAN_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0, "absmax: tensor must be 16-byte aligned");
    const unsigned blocks = (unsigned)std::min<size_t>(cdivz(n / 4 + 1, 256 * 8), 1024);
    AGAN_LAUNCH(absmax_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), x, n, amax_slot);
    return check_launch("absmax");
}

namespace agan {
namespace conv {

int prec_planes(int prec) {
    switch (prec) {
        case AGAN_PREC_BF16: case AGAN_PREC_F16: return 1;
        case AGAN_PREC_BF16X3: case AGAN_PREC_F16X3: return 2;
        case AGAN_PREC_BF16X6: return 3;
    }
    return 0;
}

// Can the patch kernels run this geometry?  (taps 2x2 / 3x3 per phase, images of at least 4x4 lattice points, tensors < 2^30)
bool patch_supported(const Geom& g) {
    if (g.Cout <= 4) return false;                                    // <= 4 output channels: the vector-ALU kernels (conv_small.hip)
    // <= 4 gathered channels (the image-input convs, the data gradient of the RGB heads): a stage would stage 3 of its 16-32
    // channel slots; measured 2.5-7x slower than the fp32 k-table kernels (profiles/r02_*_conv_layer_table.txt, round 2)
    if (g.Cin <= 4) return false;
    if (g.SY == 1) {
        if (!((g.R == 3 && g.S == 3) || (g.R == 2 && g.S == 2))) return false;
        if (g.DY != 1 && g.DY != -1) return false;
    } else if (g.SY == 2) {
        if (!(g.R == 4 && g.S == 4 && g.DY == 1)) return false;
    } else {
        return false;
    }
    const PatchPlan pp = make_patch_plan(g);
    return pp.PP <= kMaxPos && pp.twl >= 2;        // (the weight gradient reads 4-pixel blocks that must not straddle a tile row)
}

PatchPlan make_patch_plan(const Geom& g) {
    PatchPlan pp;
    memset(&pp, 0, sizeof(pp));
    // 2-D tile of 128 lattice points: as wide as the image up to 32, then rows, then images
    pp.twl = std::min(5, pow2ceil_log(g.OWs));
    pp.thl = std::min(7 - pp.twl, pow2ceil_log(g.OHs));
    const int TW = 1 << pp.twl, TH = 1 << pp.thl, TB = 128 >> (pp.twl + pp.thl);
    pp.tiles_x = cdiv(g.OWs, TW);
    pp.tiles_y = cdiv(g.OHs, TH);
    pp.tiles_b = cdiv(g.B, TB);
    pp.mtiles = pp.tiles_x * pp.tiles_y * pp.tiles_b;
    pp.IS = g.SY;
    int rows, cols;        // taps per phase
    if (g.SY == 1) { pp.NPH = 1; rows = g.R; cols = g.S; }
    else { pp.NPH = 4; rows = g.R / 2; cols = g.S / 2; }
    pp.NT = rows * cols;
    pp.PH = TH + rows - 1;
    pp.PW = TW + cols - 1;
    pp.PHW = pp.PH * pp.PW;
    pp.PP = TB * pp.PHW;
    // input coordinate of patch row j:  IS * (ty0 + j) + base_y[class parity][phase parity]
    const int dmin = g.SY == 1 ? std::min(0, (g.R - 1) * g.DY) : 0;
    for (int par = 0; par < 2; ++par)
        for (int ph = 0; ph < 2; ++ph) {
            const int oy = par ? g.OY1 : g.OY0;
            pp.base_y[par][ph] = pp.base_x[par][ph] = (g.SY == 1 ? dmin : ph) + oy;
        }
    for (int t = 0; t < pp.NT; ++t) {
        const int r = t / cols, s = t - r * cols;
        const int ro = g.SY == 1 ? r * g.DY - dmin : r, so = g.SY == 1 ? s * g.DY - dmin : s;
        pp.tapoff[t] = (ro * pp.PW + so) * kPosBytes;
        pp.tappos[t] = ro * pp.PW + so;
    }
    pp.nchunks = cdiv(g.Cin, kCH);
    pp.nstages = pp.nchunks * pp.NPH;
    pp.nsteps = pp.nstages * pp.NT * 2;
    pp.dPP = make_fastdiv((unsigned)pp.PP);
    pp.dPHW = make_fastdiv((unsigned)pp.PHW);
    pp.dPW = make_fastdiv((unsigned)pp.PW);
    return pp;
}

PatchGather plan_patch_gather(const Geom& g, const PatchPlan& pp) {
    PatchGather p;
    p.bn = g.Cout >= 96 ? 128 : (g.Cout >= 48 ? 64 : 32);
    p.ntiles = cdiv(g.Cout, p.bn);
    p.ncls = g.OS * g.OS;
    const int tiles = pp.mtiles * p.ntiles * p.ncls;
    // fewer tiles than resident workgroups (256 CUs x 2): split the stages, keeping at least 4 per workgroup
    int ks = 1;
    if (tiles < 512) ks = std::max(1, std::min({512 / tiles, pp.nstages / 4, 32}));
    p.stages_per_split = cdiv(pp.nstages, ks);
    p.ksplit = cdiv(pp.nstages, p.stages_per_split);
    p.slab = ((size_t)g.B * g.Cout * g.OH * g.OW + 3) / 4 * 4;
    p.ws_bytes = p.ksplit > 1 ? p.slab * p.ksplit * sizeof(float) : 0;
    return p;
}

size_t patch_packed_weight_bytes(int mode, int cout, int cin, int kh, int kw, int prec) {
    int ncls, K, N;
    if (pack_dims(mode, cout, cin, kh, kw, ncls, K, N)) return 0;
    int R, S, SY;
    pack_taps(mode, kh, kw, R, S, SY);
    const int Kin = K / (R * S);
    const int nph = SY == 2 ? 4 : 1, nt = SY == 2 ? (R / 2) * (S / 2) : R * S;
    const size_t nsteps = (size_t)cdiv(Kin, kCH) * nph * nt * 2;
    return (size_t)ncls * nsteps * prec_planes(prec) * agan_round_up(N, 32) * 32;
}

int pack_weight_patch(const float* w, void* wk, int mode, int cout, int cin, int kh, int kw, int prec, hipStream_t st) {
    int ncls, K, N;
    if (pack_dims(mode, cout, cin, kh, kw, ncls, K, N)) return AGAN_EINVAL;
    int R, S, SY;
    pack_taps(mode, kh, kw, R, S, SY);
    // only the fields the pack kernel reads
    PatchPlan pp;
    memset(&pp, 0, sizeof(pp));
    pp.IS = SY;
    pp.NPH = SY == 2 ? 4 : 1;
    pp.NT = SY == 2 ? (R / 2) * (S / 2) : R * S;
    const int Kin = K / (R * S);
    pp.nchunks = cdiv(Kin, kCH);
    pp.nstages = pp.nchunks * pp.NPH;
    pp.nsteps = pp.nstages * pp.NT * 2;
    const int Nld = agan_round_up(N, 32);
    const dim3 blocks(cdiv(Nld, kPackN), pp.nchunks);
    unsigned short* o = static_cast<unsigned short*>(wk);
    const FastDiv dkk = make_fastdiv((unsigned)(kh * kw));
    const FastDiv drow = make_fastdiv((unsigned)((pack_n_is_cout(mode) ? kCH : kPackN) * kh * kw));
    PackPlanLite pl;
    pl.IS = pp.IS; pl.NPH = pp.NPH; pl.NT = pp.NT; pl.nsteps = pp.nsteps;
#define AGAN_PK(ET, NPL) AGAN_LAUNCH((pack_patch_weight_kernel<ET, NPL>), blocks, dim3(256), 0, st, w, o, mode, cout, cin, kh, kw, ncls, K, R, S, Nld, pl, drow, dkk)
    switch (prec) {
        case AGAN_PREC_BF16: AGAN_PK(0, 1); break;
        case AGAN_PREC_F16: AGAN_PK(1, 1); break;
        case AGAN_PREC_BF16X3: AGAN_PK(0, 2); break;
        case AGAN_PREC_BF16X6: AGAN_PK(0, 3); break;
        case AGAN_PREC_F16X3: AGAN_PK(1, 2); break;
        default: return AGAN_EINVAL;
    }
#undef AGAN_PK
    return check_launch("pack_weight/patch");
}

int pack_job_blocks_patch(int mode, int cout, int cin, int kh, int kw) {
    int ncls, K, N;
    if (pack_dims(mode, cout, cin, kh, kw, ncls, K, N)) return 0;
    int R, S, SY;
    pack_taps(mode, kh, kw, R, S, SY);
    return cdiv(agan_round_up(N, 32), kPackN) * cdiv(K / (R * S), kCH);
}

int pack_weights_patch(const agan_pack_job* jobs, int njobs, int total_blocks, int prec, hipStream_t st) {
#define AGAN_PKJ(ET, NPL) AGAN_LAUNCH((pack_patch_jobs_kernel<ET, NPL>), dim3(total_blocks), dim3(256), 0, st, jobs, njobs)
    switch (prec) {
        case AGAN_PREC_BF16: AGAN_PKJ(0, 1); break;
        case AGAN_PREC_F16: AGAN_PKJ(1, 1); break;
        case AGAN_PREC_BF16X3: AGAN_PKJ(0, 2); break;
        case AGAN_PREC_BF16X6: AGAN_PKJ(0, 3); break;
        case AGAN_PREC_F16X3: AGAN_PKJ(1, 2); break;
        default: return AGAN_EINVAL;
    }
#undef AGAN_PKJ
    return check_launch("pack_weights/patch");
}

PatchWgrad plan_patch_wgrad(const Geom& g, const PatchPlan& pp, int prec) {
    PatchWgrad p;
    p.bj = prec_planes(prec) >= 3 ? 64 : 128;
    if (g.Cout <= 64) p.bj = 64;
    p.jtiles = cdiv(g.Cout, p.bj);
    p.ncls = g.OS * g.OS;
    const int wgs = p.jtiles * pp.nstages * p.ncls;
    // one workgroup per CU is resident (LDS): split the pixel tiles until the grid covers the chip about twice
    int ps = 1;
    if (wgs < 512) ps = std::max(1, std::min(512 / wgs, pp.mtiles));
    p.tiles_per_split = cdiv(pp.mtiles, ps);
    p.psplit = cdiv(pp.mtiles, p.tiles_per_split);
    p.Kp = pp.nstages * pp.NT * kCH;
    p.slab = ((size_t)p.ncls * g.Cout * p.Kp + 3) / 4 * 4;
    p.ws_bytes = p.slab * (p.psplit + (p.psplit > 1 ? 1 : 0)) * sizeof(float);      // partial slabs + the reduced one
    return p;
}

void launch_patch_wgrad(const void* x, const void* dy, float* part, const Geom& g, const PatchPlan& pp, const PatchWgrad& p, int prec,
                        hipStream_t st, const float* x_scale, const float* dy_scale, bool x16, bool y16) {
    if (x16 || y16) {          // (the caller has checked: one-plane mode, storage type = operand type)
        if (prec == AGAN_PREC_BF16) { if (p.bj == 128) launch_wg_dt<0, 128>(x, dy, part, g, pp, p, st, x16, y16); else launch_wg_dt<0, 64>(x, dy, part, g, pp, p, st, x16, y16); }
        else { if (p.bj == 128) launch_wg_dt<1, 128>(x, dy, part, g, pp, p, st, x16, y16); else launch_wg_dt<1, 64>(x, dy, part, g, pp, p, st, x16, y16); }
        return;
    }
    switch (prec) {
        case AGAN_PREC_F16X3: if (p.bj == 128) launch_wg_nt<1, 2, 128>(x, dy, part, g, pp, p, st, x_scale, dy_scale); else launch_wg_nt<1, 2, 64>(x, dy, part, g, pp, p, st, x_scale, dy_scale); break;
        case AGAN_PREC_BF16: if (p.bj == 128) launch_wg_nt<0, 1, 128>(x, dy, part, g, pp, p, st); else launch_wg_nt<0, 1, 64>(x, dy, part, g, pp, p, st); break;
        case AGAN_PREC_F16: if (p.bj == 128) launch_wg_nt<1, 1, 128>(x, dy, part, g, pp, p, st); else launch_wg_nt<1, 1, 64>(x, dy, part, g, pp, p, st); break;
        case AGAN_PREC_BF16X3: if (p.bj == 128) launch_wg_nt<0, 2, 128>(x, dy, part, g, pp, p, st); else launch_wg_nt<0, 2, 64>(x, dy, part, g, pp, p, st); break;
        default: launch_wg_nt<0, 3, 64>(x, dy, part, g, pp, p, st); break;
    }
}

void launch_patch_wgrad_unpack(const float* red, float* dw, int cout, int cin, int kh, int kw, bool up, int NPH, int NT, int Kp, int accumulate,
                               hipStream_t st) {
    const size_t total = (size_t)cout * cin * kh * kw;
    AGAN_LAUNCH(unpack_patch_wgrad_kernel, dim3((unsigned)std::min<size_t>(cdivz(total, 256), 4096)), dim3(256), 0, st, red, dw, cout, cin,
                       kh, kw, up ? 1 : 0, NPH, NT, Kp, accumulate);
}
void launch_wgrad_sum_unpack(const float* slabs, int nslabs, size_t slab, float* dw, int cout, int cin, int kh, int kw, int NPH, int NT, int Kp,
                             int accumulate, hipStream_t st) {
    const int ntiles = cout * cdiv(cin, kCH);
    AGAN_LAUNCH(sum_unpack_wgrad_kernel, dim3((unsigned)std::min(cdiv(ntiles, 4), 8192)), dim3(256), 0, st, slabs, nslabs, slab, dw, cout, cin, kh,
                       kw, NPH, NT, Kp, accumulate);
}
void launch_patch_wgrad_unpack(const float* red, float* dw, int cout, int cin, int kh, int kw, bool up, const PatchPlan& pp, const PatchWgrad& p,
                               int accumulate, hipStream_t st) {
    launch_patch_wgrad_unpack(red, dw, cout, cin, kh, kw, up, pp.NPH, pp.NT, p.Kp, accumulate, st);
}

void launch_patch_gather(const float* in, const void* wk, const float* bias, float* dst, const Geom& g, const PatchPlan& pp,
                         const PatchGather& p, int prec, int act, const float* lrelu_mask, hipStream_t st, const float* in_amax,
                         float* out_amax) {
    switch (prec) {
        case AGAN_PREC_BF16: launch_bn<0, 1>(in, wk, bias, dst, g, pp, p, act, lrelu_mask, st, nullptr, out_amax); break;
        case AGAN_PREC_F16: launch_bn<1, 1>(in, wk, bias, dst, g, pp, p, act, lrelu_mask, st, nullptr, out_amax); break;
        case AGAN_PREC_BF16X3: launch_bn<0, 2>(in, wk, bias, dst, g, pp, p, act, lrelu_mask, st, nullptr, out_amax); break;
        case AGAN_PREC_F16X3: launch_bn<1, 2>(in, wk, bias, dst, g, pp, p, act, lrelu_mask, st, in_amax, out_amax); break;
        default: launch_bn<0, 3>(in, wk, bias, dst, g, pp, p, act, lrelu_mask, st, nullptr, out_amax); break;
    }
}

}  // namespace conv
}  // namespace agan
