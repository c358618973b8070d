// Row-block staging gather for the one-plane 16-bit matrix-core modes (AGAN_PREC_BF16 / AGAN_PREC_F16), with optional 16-BIT
// ACTIVATION STORAGE in HBM (round 3; include/agan.h: AGAN_DT_*).
//
// What bounded the patch-resident kernel of conv_patch.hip in these modes (11-18 % MFMA-busy in round 2) was not the matrix pipe
// and not HBM: every staged value was ONE 4-byte-per-lane global load -- 40 vector-memory instructions per thread and stage, six
// per 128-cycle k-step and wave, more than the texture-address path of a CU issues -- and, loads returning in order, the first
// weight fragment fetched after that burst waited for all of it.  Here a lane loads 16 BYTES of one NCHW row -- 4 fp32 or 8
// 16-bit pixels of one channel -- for 8 channels (8 loads per thread and stage instead of 40), transposes the 8 x PXB block in
// registers (v_perm_b32 / v_cvt_pk) and writes PXB [position][8 channels] items with ds_write_b128.  The LDS image is the FULL
// input patch of the tile: stride-2 convolutions no longer run as four phase stages with their own gathers -- the patch rows are
// kept de-interleaved ([even x | odd x]), so a tap of a stride-2 conv is still a unit-stride ds_read_b128 across the 32 pixels
// of a tile row.  With 16-bit storage nothing is converted at all on the way in, half the bytes move, and the epilogue rounds
// once to the storage type.  The weight fragments (packed exactly as for conv_patch.hip: [class][k-step][cout][16]) come
// straight from L2 through a register ring 8-9 k-steps deep.
//
// Geometry kinds (GK):   0  3x3 stride 1          32 channels per stage, 18 k-steps
//                        1  2x2 stride 1          64 channels per stage, 16 k-steps  (parity classes of conv4x4-s2 dgrad / upsample conv)
//                        2  4x4 stride 2          16 channels per stage, 16 k-steps  (all four input phases in one stage)
// Accumulators are D[cout][pixel] as in the other conv kernels; BN x 128-pixel workgroup tiles, 4 waves.
#include "conv_common.h"

#include <cstring>

using namespace agan;
using namespace agan::conv;

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ u32x4 ld16(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
template <int ET>
__device__ __forceinline__ unsigned pack2(float a, float b) {          // two fp32 -> two 16-bit values, round to nearest even
    f32x2 v = {a, b};
    if (ET == 0) return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
}
template <int ET>
__device__ __forceinline__ float up16(unsigned short h) {              // 16-bit storage value -> fp32
    if (ET == 0) return __uint_as_float((unsigned)h << 16);
    return (float)__builtin_bit_cast(_Float16, h);
}
template <int ET>
__device__ __forceinline__ f32x16 mfma16(u32x4 a, u32x4 b, f32x16 c) {
    if (ET == 0) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

template <int GK> struct GKTraits;
template <> struct GKTraits<0> { static constexpr int R = 3, S = 3, IS = 1, CHS = 32; };
template <> struct GKTraits<1> { static constexpr int R = 2, S = 2, IS = 1, CHS = 64; };
template <> struct GKTraits<2> { static constexpr int R = 4, S = 4, IS = 2, CHS = 16; };

// ET: 0 bf16 / 1 f16 (MFMA operand type = 16-bit storage type).  IN16 / OUT16: the gathered / produced tensor is stored in that
// 16-bit type (else fp32).  NI: staging items per thread and stage.
// PX2 (GK 1 on a stride-2 output lattice -- the data gradient of conv4x4-s2, the folded upsample conv): one workgroup computes BOTH
// column-parity classes of 64 lattice points from ONE staged patch.  Its accumulator fragments come in pairs (class px = 0, 1 of the
// same 32 lattice points), so a lane holds two ADJACENT output pixels and the epilogue writes whole contiguous rows -- as separate
// class launches every store touched every second pixel (2-byte stores at a 4-byte stride) and the patch was staged twice.
#ifndef AGAN_P16_OCC
#define AGAN_P16_OCC 2
#endif
#ifndef AGAN_P16_WD
#define AGAN_P16_WD 0
#endif
// Diagnostic builds only (-DAGAN_P16_TIMING, never shipped: profiles/r04_p16_timing.txt): s_memtime stamps per workgroup at kernel entry,
// after the prologue's barrier, after the K loop and at exit, into a __device__ array that agan_debug_p16_stamps() copies out.
#ifdef AGAN_P16_TIMING
__device__ unsigned long long g_p16_stamps[16384 * 4];
#define AGAN_P16_STAMP(i) do { if (threadIdx.x == 0 && linear_block_id() < 16384) g_p16_stamps[linear_block_id() * 4 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define AGAN_P16_STAMP(i) do { } while (0)
#endif
// -DAGAN_P16_ABLATE=n (diagnostic builds): 1 no patch loads, 2 no weight loads, 3 item writes to the trash slot, 4 no epilogue,
// 5 no item transposes / writes in the K loop, 6 no fragment reads in the K loop, 7 = 5 + 6
#ifndef AGAN_P16_ABLATE
#define AGAN_P16_ABLATE 0
#endif
// s_setprio 1 around each k-step's MFMA burst (experiment switch)
#ifndef AGAN_P16_PRIO
#define AGAN_P16_PRIO 0
#endif
#ifndef AGAN_P16_STAGGER
#define AGAN_P16_STAGGER 0
#endif
#ifndef AGAN_P16_PIN
#define AGAN_P16_PIN 0
#endif
// WC / LPX (round 4): WC weight (cout) fragments per wave and 2^LPX lattice points per workgroup.  The round-3 tile (WC 1, LPX 7: each wave one
// 32-channel fragment x 128 pixels) issues 5 fragment loads per 4 MFMAs and re-reads the packed weights once per 128 pixels: it is the L2 -> L1 path
// that saturates (340 B per k of weights + patch per 32 K MACs = 96 FLOP/B against the chip's ~72 FLOP/B of L2 bandwidth at the dense MFMA rate).
// WC 2 gives a wave a 64-channel x 128-pixel register tile (128 accumulator registers, ONE wave per SIMD): 6 fragment loads per 8 MFMAs, half the LDS
// reads and half the vector-memory instructions per MFMA, and -- as 128 channels x 256 pixels (LPX 8) -- half the weight bytes per FLOP.
template <int ET, int GK, int BN, bool IN16, bool OUT16, int NI, bool PX2 = false, int WC = 1, int LPX = 7>
__global__ __launch_bounds__(256, (WC == 2 ? 1 : AGAN_P16_OCC)) void conv_p16_kernel(const void* __restrict__ in, const unsigned short* __restrict__ wk,
                                                          const float* __restrict__ bias, void* __restrict__ out, const Geom g,
                                                          const P16Plan pp, const int act, const void* __restrict__ lrelu_mask) {
    using T = GKTraits<GK>;
    constexpr int R = T::R, S = T::S, IS = T::IS, CHS = T::CHS;
    constexpr int NT = R * S;
    constexpr int HS = CHS / 16;                     // 16-channel k-steps per tap (GK 2: one)
    constexpr int SPS = GK == 2 ? NT : NT * HS;      // k-steps per stage: 18 / 16 / 16
    constexpr int PB = CHS * 2 + 16;                 // LDS bytes per position: an odd multiple of 16 B (conflict-free b128 rows)
    constexpr int PXB = IN16 ? 8 : 4;                // pixels per 16-byte load
    constexpr int ESZ = IN16 ? 2 : 4;
    // Weight ring depth (divides SPS).  Loads return IN ORDER, so a weight fragment issued after the patch loads of a stage cannot land
    // before them: the ring depth is therefore also the number of k-steps the patch loads get before anything waits on them.
    // (measured, round 3: a ring as deep as the whole stage -- 64-72 VGPRs -- was 0-25 % SLOWER than 8-9 k-steps: the kernel is bound
    // by the texture-address path, 57 % busy, of which the weight fragments are two thirds, not by one exposed latency)
#ifndef AGAN_P16_WD_FULL
#define AGAN_P16_WD_FULL 0
#endif
    constexpr int WD = AGAN_P16_WD ? (SPS == 18 ? (AGAN_P16_WD == 4 ? 3 : AGAN_P16_WD) : AGAN_P16_WD)
                                   : (WC == 2 ? (SPS == 18 ? 6 : 4)      // (WC 2: two fragments per slot and 256 MFMA cycles per k-step)
                                              : (PX2 ? 4 : ((AGAN_P16_WD_FULL && NI == 1) ? SPS : (SPS == 18 ? 9 : 8))));     // (PX2: two fragments per slot)
    constexpr int NFRAG = PX2 ? 4 : (1 << (LPX - 5));   // 32-pixel fragments per workgroup tile
    constexpr int WN = BN / (32 * WC), WM = 4 / WN, TM = NFRAG / WM;
    static_assert(WN >= 1 && WN <= 4 && WM * WN == 4 && TM >= 1 && TM * WM == NFRAG, "wave grid");
    static_assert(!PX2 || (GK == 1 && TM >= 2 && WC == 1 && LPX == 7), "PX2: 2x2-tap classes, at least two pixel fragments per wave");
    constexpr int NPX = PX2 ? 2 : 1;                 // column-parity classes per workgroup
    constexpr int NWF = NPX * WC;                    // weight fragments per wave and k-step
    constexpr int LP = PX2 ? 6 : LPX;                // log2 of the lattice points per workgroup tile
    constexpr int NWR = NI * PXB;                    // LDS item writes per thread and stage
    constexpr int U0 = SPS / 2;                      // first k-step that carries item writes: the loads get SPS/2 k-steps to land
    constexpr int WPS = (NWR + (SPS - U0) - 1) / (SPS - U0);
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int buf_bytes = pp.buf_bytes;

#if AGAN_P16_STAGGER
    // experiment: de-phase the two workgroups of a CU -- the second-round-robin half of the first 512 workgroups starts late
    if (linear_block_id() >= 256 && linear_block_id() < 512) {
#pragma unroll 1
        for (int i = 0; i < AGAN_P16_STAGGER; ++i) __builtin_amdgcn_s_sleep(127);
    }
#endif
    AGAN_P16_STAMP(0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave % WN, wm = wave / WN;
    const int l31 = lane & 31, lh = lane >> 5;
    const int ksplit = pp.ksplit;
    int mt, nt, cls, split;
    {
        // (divisions by the plan's tile counts as multiply-high with host-made constants: seven runtime scalar divisions were ~250 of the
        //  360 instructions in front of the kernel's first load)
        const int ncls = pp.ncls, mtiles = pp.mtiles, ntiles = pp.ntiles;
        int F = xcd_contiguous(linear_block_id(), mtiles * ntiles * ncls * ksplit);
        int q = pp.dNT.div(F);
        nt = F - q * ntiles; F = q;
        q = pp.dNCLS.div(F);
        cls = F - q * ncls; F = q;
        split = pp.dMT.div(F);
        mt = F - split * mtiles;
    }
    const int py = PX2 ? cls : cls / g.OS, px = PX2 ? 0 : cls - py * g.OS;       // (PX2: the grid's classes are the ROW parities)
    const int n0 = nt * BN;
    const int mty = pp.dTX.div(mt), txi = mt - mty * pp.tiles_x, tbi = pp.dTY.div(mty), tyi = mty - tbi * pp.tiles_y;
    const int twl = pp.twl, thl = pp.thl;
    const int tb0 = tbi << (LP - twl - thl), ty0 = tyi << thl, tx0 = txi << twl;
    const int ihw = g.IH * g.IW;
    const int stage_beg = split * pp.stages_per_split, stage_end = min(pp.nstages, stage_beg + pp.stages_per_split);

    // patch origin in the gathered tensor (row / column of LDS row 0 / of the first needed column), and the aligned column the
    // 16-byte blocks start at
    const int dmin = (IS == 1 && g.DY < 0) ? -(R - 1) : 0;
    // (PX2: the patch starts at the left class's first column -- OY0 <= OY1 -- and is OY1 - OY0 columns wider)
    const int y0 = IS * ty0 + dmin + (py ? g.OY1 : g.OY0), x0 = IS * tx0 + dmin + ((px && !PX2) ? g.OY1 : g.OY0);
    const int gx0 = x0 & ~(PXB - 1);
    const int dx0 = x0 - gx0;
    const int LW = pp.LW, LWH = LW >> 1;

    const __amdgpu_buffer_rsrc_t rin = make_rsrc(in, (size_t)g.B * g.Cin * ihw * ESZ);
    const size_t wbytes_cls = (size_t)pp.wsteps * g.Nld * 32;
    const int wcls = PX2 ? py * 2 : cls;              // packed-weight class (row parity x 2 + column parity); PX2 reads wcls and wcls + 1
    const __amdgpu_buffer_rsrc_t rwk = make_rsrc(reinterpret_cast<const unsigned char*>(wk) + (size_t)wcls * wbytes_cls, NPX * wbytes_cls);

    // ---- staging items: (channel octet o, image pb, patch row j, 16-byte block xb), xb fastest (lanes walk along a row) ----
    unsigned it_voff[NI], it_lds[NI];
    int it_oct[NI], it_x0[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int e = tid + i * 256;
        const int t1 = pp.dNXB.div(e), xb = e - t1 * pp.NXB;
        const int t2 = pp.dPH.div(t1), j = t1 - t2 * pp.PH;
        const int o = pp.dTB.div(t2), pb = t2 - o * pp.TB;
        const int b = tb0 + pb, iy = y0 + j, gx = gx0 + xb * PXB;
        const bool ok = (e < pp.nitems) & (b < g.B) & ((unsigned)iy < (unsigned)g.IH) & ((unsigned)gx < (unsigned)g.IW);
        it_oct[i] = (e < pp.nitems) ? o : CHS;          // (>= CHS/8: no channel of any stage)
        it_voff[i] = ok ? (unsigned)(((b * g.Cin + o * 8) * g.IH + iy) * g.IW + gx) * (unsigned)ESZ : kOOB;
        const int row = pb * pp.PH + j;
        it_lds[i] = (unsigned)(row * LW * PB + o * 16);  // the item's LDS row
        it_x0[i] = xb * PXB - dx0;                       // patch column of the block's first pixel (the blocks are aligned in the TENSOR)
    }
    u32x4 blk[NI][8];
    auto load_items = [&](int chunk) {
        const int c0 = chunk * CHS;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const unsigned voff = (AGAN_P16_ABLATE != 1 && c0 + it_oct[i] * 8 < g.Cin && it_oct[i] < CHS / 8) ? it_voff[i] : kOOB;
#pragma unroll
            for (int c = 0; c < 8; ++c) blk[i][c] = ld16(rin, voff, (unsigned)((c0 + c) * ihw) * (unsigned)ESZ);
        }
    };
    // item write w of a stage = pixel q of item i: transpose on the fly, one ds_write_b128
    auto store_piece = [&](int w, unsigned char* dstbuf) {
        const int i = w / PXB, q = w % PXB;
        if (i >= NI) return;
        u32x4 v;
        if (IN16) {
            const int d = q >> 1;
            const unsigned sel = (q & 1) ? 0x07060302u : 0x05040100u;
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = __builtin_amdgcn_perm(blk[i][2 * k + 1][d], blk[i][2 * k][d], sel);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                v[k] = pack2<ET>(__uint_as_float(blk[i][2 * k][q]), __uint_as_float(blk[i][2 * k + 1][q]));
        }
        // LDS rows hold exactly the patch's columns (IS == 2: [even | odd]); pixels of the aligned blocks that fall outside are dropped
        const int xl = it_x0[i] + q;
        const unsigned a = it_lds[i] + (unsigned)((IS == 1 ? xl : (xl & 1) * LWH + (xl >> 1)) * PB);
        if ((it_oct[i] < CHS / 8) & ((unsigned)xl < (unsigned)pp.PW)) *reinterpret_cast<u32x4*>(dstbuf + a) = v;
    };

    // packed weights (conv_patch.hip): q = ((chunk32 * NPH + phase) * NT' + tap') * 2 + half
    auto wq_of = [&](int stage, int u) {
        if (GK == 0) return stage * 18 + u;                                   // chunk32 = stage; u = tap * 2 + half
        if (GK == 1) return stage * 16 + u;                                   // two chunk32 per stage; u = c32 * 8 + tap * 2 + half
        const int r = u >> 2, s = u & 3;                                      // GK 2: stage = (chunk32, half); u = tap (r, s)
        const int ph = (r & 1) * 2 + (s & 1), tp = (r >> 1) * 2 + (s >> 1);
        return (((stage >> 1) * 4 + ph) * 4 + tp) * 2 + (stage & 1);
    };
    // weight fragment f of a wave: PX2 -> class f (same 32 channels); WC 2 -> channels n0 + (wn * WC + f) * 32 ..
    unsigned wlane[NWF];
#pragma unroll
    for (int f = 0; f < NWF; ++f)
        wlane[f] = PX2 ? (unsigned)(min(n0 + wn * 32 + l31, g.Nld - 1) * 32 + lh * 16) + (unsigned)(f * wbytes_cls)
                       : (unsigned)(min(n0 + (wn * WC + f) * 32 + l31, g.Nld - 1) * 32 + lh * 16);
    const unsigned wstep = (unsigned)(g.Nld * 32);
    u32x4 wf[WD][NWF];
    auto load_w = [&](int slot, int q) {
#pragma unroll
        for (int c = 0; c < NWF; ++c) wf[slot][c] = ld16(rwk, AGAN_P16_ABLATE == 2 ? kOOB : wlane[c], (unsigned)min(q, pp.wsteps - 1) * wstep);
    };

    // The first stage's patch and the first weight fragments are requested HERE, before the ~250 instructions of operand addressing below
    // (fragment bases, tap offsets): the prologue was index math, THEN the loads, THEN ~1 us of HBM latency with nothing to do
    // (first loads at instruction 444 of the kernel: round-4 ISA; the prologue is 17 % of a tile's life).
    if (stage_beg < stage_end) {
        load_items(stage_beg);
#pragma unroll
        for (int d = 0; d < WD; ++d) load_w(d, wq_of(stage_beg, d));
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- MFMA operand addressing ----
    // fragment t of a wave: lattice points lat(t) * 32 + l31 (PX2: fragments 2f, 2f + 1 = classes px 0, 1 of lattice fragment f)
    auto lat_of = [&](int t) { return PX2 ? (wm * TM + t) >> 1 : wm * TM + t; };
    unsigned lbase[TM];
#pragma unroll
    for (int t = 0; t < TM; ++t) {
        const int l = lat_of(t) * 32 + l31;
        const int tx = l & ((1 << twl) - 1), ty = (l >> twl) & ((1 << thl) - 1), tb = l >> (twl + thl);
        lbase[t] = (unsigned)(((tb * pp.PH + IS * ty) * LW + tx + ((PX2 && (t & 1)) ? g.OY1 - g.OY0 : 0)) * PB + lh * 16);
    }
    unsigned tapoff[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int r = t / S, s = t - r * S;
        if (IS == 1) {
            const int ro = r * g.DY - dmin, so = s * g.DY - dmin;
            tapoff[t] = (unsigned)((ro * LW + so) * PB);
        } else {
            tapoff[t] = (unsigned)((r * LW + (s & 1) * LWH + (s >> 1)) * PB);
        }
    }
    // k-step u of a stage -> (tap, byte offset of its 16 channels inside a position, k-step of the packed weights)
    auto tap_of = [&](int u) { return GK == 2 ? u : (GK == 0 ? u / 2 : (u & 7) / 2); };
    auto choff_of = [&](int u) { return GK == 2 ? 0 : (GK == 0 ? (u & 1) * 32 : (u >> 3) * 64 + (u & 1) * 32); };
    f32x16 acc[WC][TM];
#pragma unroll
    for (int c = 0; c < WC; ++c)
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][t][r] = 0.f;

    if (stage_beg < stage_end) {
#pragma unroll
        for (int w = 0; w < NWR; ++w) store_piece(w, lds + (stage_beg & 1) * buf_bytes);
        lds_barrier();
        AGAN_P16_STAMP(1);
        for (int stage = stage_beg; stage < stage_end; ++stage) {
            const bool more = stage + 1 < stage_end;
            const unsigned char* cur = lds + (stage & 1) * buf_bytes;
            unsigned char* nxt = lds + ((stage + 1) & 1) * buf_bytes;
            if (more) load_items(stage + 1);
            u32x4 af[2][TM];
#pragma unroll
            for (int m = 0; m < TM; ++m) af[0][m] = *reinterpret_cast<const u32x4*>(cur + lbase[m] + tapoff[tap_of(0)] + choff_of(0));
            // the groups below form ONE ordered pipeline over the unrolled stage: [reads of step u+1][MFMAs of step u] ...  Without a
            // group of its own, step 0's reads above were taken as the FIRST read group and every later group slid back by one: the
            // compiled loop read each fragment right in front of the MFMA that consumes it (lgkmcnt wait per MFMA, the whole LDS latency
            // exposed in every k-step; round-4 disassembly + s_memtime stamps: 410 cycles per pair of k-steps for 256 of MFMA)
#if AGAN_P16_PIN
            __builtin_amdgcn_sched_group_barrier(0x100, TM, 0);
#endif
            if (AGAN_P16_ABLATE == 6 || AGAN_P16_ABLATE == 7) {
#pragma unroll
                for (int m = 0; m < TM; ++m) af[1][m] = af[0][m];
            }
#pragma unroll
            for (int u = 0; u < SPS; ++u) {
                u32x4 w[NWF];
#pragma unroll
                for (int c = 0; c < NWF; ++c) w[c] = wf[u % WD][c];
                {   // weights WD k-steps ahead (possibly in the next stage; past the last stage: a harmless repeat)
                    const int u2 = u + WD;
                    if (u2 < SPS) load_w(u % WD, wq_of(stage, u2));
                    else load_w(u % WD, wq_of(more ? stage + 1 : stage, u2 - SPS));
                }
                if (u + 1 < SPS && AGAN_P16_ABLATE != 6 && AGAN_P16_ABLATE != 7) {
#pragma unroll
                    for (int m = 0; m < TM; ++m)
                        af[(u + 1) & 1][m] = *reinterpret_cast<const u32x4*>(cur + lbase[m] + tapoff[tap_of(u + 1)] + choff_of(u + 1));
                }
#if AGAN_P16_PRIO
                __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
                for (int m = 0; m < TM; ++m)
#pragma unroll
                    for (int c = 0; c < WC; ++c) acc[c][m] = mfma16<ET>(w[PX2 ? (m & 1) : c], af[u & 1][m], acc[c][m]);
#if AGAN_P16_PRIO
                __builtin_amdgcn_s_setprio(0);
#endif
                if (WC == 1) {
                    __builtin_amdgcn_sched_group_barrier(0x100, TM, 0);      // next step's LDS reads first ...
                    __builtin_amdgcn_sched_group_barrier(0x008, TM, 0);      // ... then this step's MFMAs
                } else {
#pragma unroll
                    for (int m = 0; m < TM; ++m) {                           // one wave per SIMD: a fragment read between every pair of MFMAs
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, WC, 0);
                    }
                }
                if (u >= U0 && more && AGAN_P16_ABLATE != 5 && AGAN_P16_ABLATE != 7) {
#pragma unroll
                    for (int k = 0; k < WPS; ++k)
                        if ((u - U0) * WPS + k < NWR) store_piece((u - U0) * WPS + k, nxt);
                }
            }
            lds_barrier();
        }
    }

    AGAN_P16_STAMP(2);
    if (AGAN_P16_ABLATE == 4) return;
    // ---- epilogue: D[cout][pixel]; a register holds 32 consecutive pixels of one output channel ----
    const size_t ohw = (size_t)g.OH * g.OW;
    const bool split_out = ksplit > 1;                  // partial sums go to fp32 slabs; the slab sum applies bias / activation / rounding
    const bool add_bias = (bias != nullptr) && !split_out;
    const bool lrelu = (act == AGAN_ACT_LRELU) && !split_out;
    const bool masked = (lrelu_mask != nullptr) && !split_out;
    const size_t nelem = (size_t)g.B * g.Cout * ohw;
    const __amdgpu_buffer_rsrc_t rslab = make_rsrc(split_out ? static_cast<float*>(out) + (size_t)split * pp.slab : static_cast<float*>(out),
                                                   nelem * sizeof(float));
    const __amdgpu_buffer_rsrc_t rout = make_rsrc(out, nelem * (OUT16 ? 2 : 4));
    const __amdgpu_buffer_rsrc_t rmask = make_rsrc(masked ? lrelu_mask : out, nelem * (OUT16 ? 2 : 4));
    const __amdgpu_buffer_rsrc_t rbias = make_rsrc(bias ? bias : static_cast<const float*>(out), (size_t)g.Cout * sizeof(float));
    if (OUT16 && pp.lds_epi && !split_out && !masked) {
        // 16-bit output, unit-stride lattice: the tile goes through LDS (free after the K loop's last barrier) as [cout][tile pixels] and
        // leaves as 16-BYTE stores of 8 consecutive pixels -- 8 store instructions per wave instead of 64 two-byte ones (the kernel is
        // bound by its vector-memory instruction count: texture-address path 57 % busy, round-3 PMC)
        constexpr int NPIX = 1 << LP;
        constexpr int ROWB = NPIX * 2 + 16;                // bytes per cout row (rows 4 apart land 16 banks apart)
        unsigned short* const l16 = reinterpret_cast<unsigned short*>(lds);
        if (!add_bias && !lrelu) {
            // the path every layer of the step takes (BatchNorm follows: no bias, no activation here): ONE conversion per two values and the
            // two 16-bit halves stored as they are (ds_write_b16 / ds_write_b16_d16_hi) -- 1.5 instructions per value instead of ~7 plus a
            // branch in the generic form below (the epilogue was 17 % of a workgroup's life: s_memtime stamps, profiles/r04_p16_timing.txt)
#pragma unroll
            for (int c = 0; c < WC; ++c) {
                const int nwl = (wn * WC + c) * 32;
#pragma unroll
                for (int t = 0; t < TM; ++t) {
                    unsigned short* const base = l16 + (nwl + 4 * lh) * (ROWB / 2) + (wm * TM + t) * 32 + l31;
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const int nr = (r & 3) + 8 * (r >> 2);
                        const unsigned u = pack2<ET>(acc[c][t][r], acc[c][t][r + 1]);
                        base[nr * (ROWB / 2)] = (unsigned short)(u & 0xFFFFu);
                        base[(nr + 1) * (ROWB / 2)] = (unsigned short)(u >> 16);
                    }
                }
            }
        } else {
#pragma unroll
            for (int c = 0; c < WC; ++c) {
                const int nwl = (wn * WC + c) * 32, nw = n0 + nwl;
#pragma unroll
                for (int t = 0; t < TM; ++t) {
                    const int pcol = (wm * TM + t) * 32 + l31;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int nr = (r & 3) + 8 * (r >> 2);
                        float v = acc[c][t][r];
                        if (add_bias) v += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rbias, (unsigned)(nw + 4 * lh) * 4u, (unsigned)nr * 4u, 0));
                        if (lrelu) v = v > 0.f ? v : 0.2f * v;
                        l16[(nwl + nr + 4 * lh) * (ROWB / 2) + pcol] = (unsigned short)(pack2<ET>(v, 0.f) & 0xFFFFu);
                    }
                }
            }
        }
        __syncthreads();
        constexpr int CPR = NPIX / 8;                      // 16-byte chunks per cout row
#pragma unroll
        for (int i = 0; i < BN * CPR / 256; ++i) {
            const int c = tid + i * 256;
            const int n = c / CPR, l = (c % CPR) * 8;
            const int tx = l & ((1 << twl) - 1), ty = (l >> twl) & ((1 << thl) - 1), tb = l >> (twl + thl);
            const int b = tb0 + tb, yq = ty0 + ty, xq = tx0 + tx;
            const bool ok = (b < g.B) & (yq < g.OH) & (xq < g.OW) & (n0 + n < g.Cout);
            const u32x4 v = *reinterpret_cast<const u32x4*>(lds + n * ROWB + (c % CPR) * 16);
            const unsigned vo = ok ? (unsigned)(((b * g.Cout + n0 + n) * g.OH + yq) * g.OW + xq) * 2u : kOOB;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), rout, vo, 0, 0);
        }
        AGAN_P16_STAMP(3);
        return;
    }
    if (PX2) {
        // fragments 2f / 2f + 1 hold output columns 2 xq / 2 xq + 1 of the same lattice point: one 4-byte (16-bit) or 8-byte (fp32)
        // store per lane and channel, 32 lanes = one contiguous 128 / 256-byte piece of an output row
        const int nw = n0 + wn * 32;
        const bool nfull = nw + 32 <= g.Cout;
#pragma unroll
        for (int t = 0; t < TM; t += 2) {
            const int l = lat_of(t) * 32 + l31;
            const int tx = l & ((1 << twl) - 1), ty = (l >> twl) & ((1 << thl) - 1), tb = l >> (twl + thl);
            const int b = tb0 + tb, yq = ty0 + ty, xq = tx0 + tx;
            const bool pvalid = (b < g.B) & (yq < g.OHs) & (xq < g.OWs);
            const unsigned pix = (unsigned)(b * g.Cout + nw + 4 * lh) * (unsigned)ohw + (unsigned)((yq * 2 + py) * g.OW + xq * 2);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int nr = (r & 3) + 8 * (r >> 2);
                float v0 = acc[0][t][r], v1 = acc[0][PX2 ? t + 1 : t][r];
                const bool ok = pvalid & (nfull || (nw + nr + 4 * lh < g.Cout));
                if (split_out) {
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(__attribute__((ext_vector_type(2))) unsigned, f32x2{v0, v1}), rslab,
                                                          ok ? pix * 4u : kOOB, (unsigned)nr * (unsigned)ohw * 4u, 0);
                    continue;
                }
                if (add_bias) {
                    const float bv = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rbias, (unsigned)(nw + 4 * lh) * 4u, (unsigned)nr * 4u, 0));
                    v0 += bv; v1 += bv;
                }
                if (lrelu) { v0 = v0 > 0.f ? v0 : 0.2f * v0; v1 = v1 > 0.f ? v1 : 0.2f * v1; }
                if (OUT16) {
                    const unsigned vo = ok ? pix * 2u : kOOB, so = (unsigned)nr * (unsigned)ohw * 2u;
                    if (masked) {
                        const unsigned m2 = __builtin_amdgcn_raw_buffer_load_b32(rmask, vo, so, 0);
                        v0 = up16<ET>((unsigned short)(m2 & 0xFFFFu)) > 0.f ? v0 : 0.2f * v0;
                        v1 = up16<ET>((unsigned short)(m2 >> 16)) > 0.f ? v1 : 0.2f * v1;
                    }
                    __builtin_amdgcn_raw_buffer_store_b32(pack2<ET>(v0, v1), rout, vo, so, 0);
                } else {
                    const unsigned vo = ok ? pix * 4u : kOOB, so = (unsigned)nr * (unsigned)ohw * 4u;
                    if (masked) {
                        const f32x2 m2 = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rmask, vo, so, 0));
                        v0 = m2[0] > 0.f ? v0 : 0.2f * v0;
                        v1 = m2[1] > 0.f ? v1 : 0.2f * v1;
                    }
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(__attribute__((ext_vector_type(2))) unsigned, f32x2{v0, v1}), rout, vo, so, 0);
                }
            }
        }
        AGAN_P16_STAMP(3);
        return;
    }
#pragma unroll
    for (int c = 0; c < WC; ++c) {
        const int nw = n0 + (wn * WC + c) * 32;
        const bool nfull = nw + 32 <= g.Cout;
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            const int l = (wm * TM + t) * 32 + l31;
            const int tx = l & ((1 << twl) - 1), ty = (l >> twl) & ((1 << thl) - 1), tb = l >> (twl + thl);
            const int b = tb0 + tb, yq = ty0 + ty, xq = tx0 + tx;
            const bool pvalid = (b < g.B) & (yq < g.OHs) & (xq < g.OWs);
            const unsigned pix = (unsigned)(b * g.Cout + nw + 4 * lh) * (unsigned)ohw + (unsigned)((yq * g.OS + py) * g.OW + (xq * g.OS + px));
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int nr = (r & 3) + 8 * (r >> 2);
                float v = acc[c][t][r];
                const bool ok = pvalid & (nfull || (nw + nr + 4 * lh < g.Cout));
                if (split_out) {
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rslab, ok ? pix * 4u : kOOB, (unsigned)nr * (unsigned)ohw * 4u, 0);
                    continue;
                }
                if (add_bias) v += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rbias, (unsigned)(nw + 4 * lh) * 4u, (unsigned)nr * 4u, 0));
                if (lrelu) v = v > 0.f ? v : 0.2f * v;
                if (OUT16) {
                    const unsigned vo = ok ? pix * 2u : kOOB, so = (unsigned)nr * (unsigned)ohw * 2u;
                    if (masked) v = up16<ET>((unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rmask, vo, so, 0)) > 0.f ? v : 0.2f * v;
                    __builtin_amdgcn_raw_buffer_store_b16((short)(pack2<ET>(v, 0.f) & 0xFFFFu), rout, vo, so, 0);
                } else {
                    const unsigned vo = ok ? pix * 4u : kOOB, so = (unsigned)nr * (unsigned)ohw * 4u;
                    if (masked) v = buf_load_s(rmask, vo, so) > 0.f ? v : 0.2f * v;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rout, vo, so, 0);
                }
            }
        }
    }
    AGAN_P16_STAMP(3);
}

template <int ET, int GK, int BN, bool IN16, bool OUT16, int WC = 1, int LPX = 7>
void launch_ni(const void* in, const void* wk, const float* bias, void* dst, const Geom& g, const P16Plan& p, int act, const void* mask,
               hipStream_t st) {
    dim3 grid(p.mtiles, p.ntiles, p.ncls * p.ksplit);
    const unsigned short* w = static_cast<const unsigned short*>(wk);
    const size_t smem = (size_t)p.smem_bytes;
#define AGAN_P16_LAUNCH(NI_)                                                                                                             \
    do {                                                                                                                                 \
        static const hipError_t attr_ = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_p16_kernel<ET, GK, BN, IN16, OUT16, NI_, false, WC, LPX>), \
                                                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                     \
        (void)attr_;                                                                                                                     \
        AGAN_LAUNCH((conv_p16_kernel<ET, GK, BN, IN16, OUT16, NI_, false, WC, LPX>), grid, dim3(256), smem, st, in, w, bias, dst, g, p, act, mask); \
    } while (0)
    if (p.NI == 1) AGAN_P16_LAUNCH(1);
    else AGAN_P16_LAUNCH(2);
#undef AGAN_P16_LAUNCH
}
// both column-parity classes in one workgroup (GK 1 on a stride-2 lattice, 64- or 128-channel tiles, one staging item per thread)
template <int ET, int BN, bool IN16, bool OUT16>
void launch_px2(const void* in, const void* wk, const float* bias, void* dst, const Geom& g, const P16Plan& p, int act, const void* mask,
                hipStream_t st) {
    dim3 grid(p.mtiles, p.ntiles, p.ncls * p.ksplit);
    const unsigned short* w = static_cast<const unsigned short*>(wk);
    const size_t smem = (size_t)p.smem_bytes;
    static const hipError_t attr_ = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_p16_kernel<ET, 1, BN, IN16, OUT16, 1, true>),
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)attr_;
    AGAN_LAUNCH((conv_p16_kernel<ET, 1, BN, IN16, OUT16, 1, true>), grid, dim3(256), smem, st, in, w, bias, dst, g, p, act, mask);
}
template <int ET, int GK, int BN, int WC, int LPX>
void launch_io(const void* in, const void* wk, const float* bias, void* dst, const Geom& g, const P16Plan& p, int act, const void* mask,
               hipStream_t st, bool in16, bool out16) {
#ifdef AGAN_P16_FAST_BUILD      // (experiment builds: bf16 with 16-bit storage both sides only -- 20 s instead of 150)
    if (ET == 0 && in16 && out16) launch_ni<0, GK, BN, true, true, WC, LPX>(in, wk, bias, dst, g, p, act, mask, st);
#else
    if (in16 && out16) launch_ni<ET, GK, BN, true, true, WC, LPX>(in, wk, bias, dst, g, p, act, mask, st);
    else if (in16) launch_ni<ET, GK, BN, true, false, WC, LPX>(in, wk, bias, dst, g, p, act, mask, st);
    else if (out16) launch_ni<ET, GK, BN, false, true, WC, LPX>(in, wk, bias, dst, g, p, act, mask, st);
    else launch_ni<ET, GK, BN, false, false, WC, LPX>(in, wk, bias, dst, g, p, act, mask, st);
#endif
}
template <int ET, int GK, int BN>
void launch_dt(const void* in, const void* wk, const float* bias, void* dst, const Geom& g, const P16Plan& p, int act, const void* mask,
               hipStream_t st, bool in16, bool out16) {
#ifndef AGAN_P16_FAST_BUILD
    if constexpr (GK == 1 && BN >= 64 && BN <= 128) {
        if (p.px2) {
            if (in16 && out16) launch_px2<ET, BN, true, true>(in, wk, bias, dst, g, p, act, mask, st);
            else if (in16) launch_px2<ET, BN, true, false>(in, wk, bias, dst, g, p, act, mask, st);
            else if (out16) launch_px2<ET, BN, false, true>(in, wk, bias, dst, g, p, act, mask, st);
            else launch_px2<ET, BN, false, false>(in, wk, bias, dst, g, p, act, mask, st);
            return;
        }
    }
#else
    if constexpr (GK == 1 && BN >= 64 && BN <= 128) {
        if (p.px2 && ET == 0) { launch_px2<0, BN, true, true>(in, wk, bias, dst, g, p, act, mask, st); return; }
    }
#endif
    // the register-tile forms (round 4): 64 channels x 128 pixels per wave
    if constexpr (BN == 128) {
        if (p.wc == 2) { launch_io<ET, GK, 128, 2, 8>(in, wk, bias, dst, g, p, act, mask, st, in16, out16); return; }      // 128 channels x 256 pixels
    }
    if constexpr (BN == 256) {
        launch_io<ET, GK, 256, 2, 7>(in, wk, bias, dst, g, p, act, mask, st, in16, out16);                                 // 256 channels x 128 pixels
        return;
    } else {
        launch_io<ET, GK, BN, 1, 7>(in, wk, bias, dst, g, p, act, mask, st, in16, out16);
    }
}
template <int ET, int GK>
void launch_bn(const void* in, const void* wk, const float* bias, void* dst, const Geom& g, const P16Plan& p, int act, const void* mask,
               hipStream_t st, bool in16, bool out16) {
    if (p.bn == 256) launch_dt<ET, GK, 256>(in, wk, bias, dst, g, p, act, mask, st, in16, out16);
    else if (p.bn == 128) launch_dt<ET, GK, 128>(in, wk, bias, dst, g, p, act, mask, st, in16, out16);
    else if (p.bn == 64) launch_dt<ET, GK, 64>(in, wk, bias, dst, g, p, act, mask, st, in16, out16);
    else launch_dt<ET, GK, 32>(in, wk, bias, dst, g, p, act, mask, st, in16, out16);
}
template <int ET>
void launch_gk(const void* in, const void* wk, const float* bias, void* dst, const Geom& g, const P16Plan& p, int act, const void* mask,
               hipStream_t st, bool in16, bool out16) {
    if (p.gk == 0) launch_bn<ET, 0>(in, wk, bias, dst, g, p, act, mask, st, in16, out16);
    else if (p.gk == 1) launch_bn<ET, 1>(in, wk, bias, dst, g, p, act, mask, st, in16, out16);
    else launch_bn<ET, 2>(in, wk, bias, dst, g, p, act, mask, st, in16, out16);
}

int pow2ceil_log_(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

}  // namespace

namespace agan {
namespace conv {

// Plan of the row-block gather for geometry g with the gathered tensor stored as fp32 (in16 = false) or in 16 bits.  p.ok == 0:
// this kernel does not take the call (the caller falls back to conv_patch.hip's kernel on fp32 tensors).
// tile: 0 = the round-3 tile (32 channels x 128 pixels per wave), 1 = 128 channels x 256 pixels per workgroup, 2 = 256 channels x 128 pixels
// (both with 64 x 128 per wave, one workgroup per CU); a form the geometry cannot take falls back to 0
static P16Plan plan_impl(const Geom& g, bool in16, bool allow_px2, int tile = 0) {
    P16Plan p;
    memset(&p, 0, sizeof(p));
    if (g.Cout <= 4 || g.Cin <= 4 || (g.Cin & 7)) return p;
    int R, IS, CHS;
    if (g.SY == 1 && g.R == 3 && g.S == 3 && g.DY == 1) { p.gk = 0; R = 3; IS = 1; CHS = 32; }
    else if (g.SY == 1 && g.R == 2 && g.S == 2 && (g.DY == 1 || g.DY == -1)) { p.gk = 1; R = 2; IS = 1; CHS = 64; }
    else if (g.SY == 2 && g.R == 4 && g.S == 4 && g.DY == 1) { p.gk = 2; R = 4; IS = 2; CHS = 16; }
    else return p;
    p.PXB = in16 ? 8 : 4;
    if (g.IW % p.PXB) return p;                       // rows of the gathered tensor must be whole 16-byte blocks
    p.bn = g.Cout >= 96 ? 128 : (g.Cout >= 48 ? 64 : 32);
    p.wc = 1;
    if (tile == 1 && g.Cout >= 96) { p.wc = 2; p.bn = 128; }
    if (tile == 2 && g.Cout >= 224) { p.wc = 2; p.bn = 256; }
    if (p.wc == 1 && tile != 0) return plan_impl(g, in16, allow_px2, 0);
    // both column-parity classes of a stride-2 lattice in one workgroup (64 lattice points x 2): see the kernel's PX2
    static const bool px2_off = getenv("AGAN_P16_PX2_OFF") != nullptr;
    p.px2 = (p.wc == 1 && allow_px2 && p.gk == 1 && g.OS == 2 && p.bn >= 64 && g.OY1 >= g.OY0 && !px2_off) ? 1 : 0;
    const int lp = p.px2 ? 6 : (tile == 1 ? 8 : 7);
    p.lp = lp;
    p.twl = std::min(5, pow2ceil_log_(g.OWs));
    p.thl = std::min(lp - p.twl, pow2ceil_log_(g.OHs));
    const int TW = 1 << p.twl, TH = 1 << p.thl;
    p.TB = (1 << lp) >> (p.twl + p.thl);
    p.tiles_x = cdiv(g.OWs, TW);
    p.tiles_y = cdiv(g.OHs, TH);
    p.tiles_b = cdiv(g.B, p.TB);
    p.mtiles = p.tiles_x * p.tiles_y * p.tiles_b;
    p.PH = IS * (TH - 1) + R;
    const int PW = IS * (TW - 1) + R + (p.px2 ? g.OY1 - g.OY0 : 0);
    p.NXB = cdiv(p.PXB - 1 + PW, p.PXB);              // 16-byte blocks that cover the patch columns whatever the alignment of its origin
    p.PW = PW;
    p.LW = (PW + 1) & ~1;                             // LDS rows hold the patch's own columns (an even count: stride-2 rows are [even | odd])
    p.CHS = CHS;
    p.nitems = (CHS / 8) * p.TB * p.PH * p.NXB;
    p.NI = cdiv(p.nitems, 256);
    if (p.NI > 2) return tile ? plan_impl(g, in16, allow_px2, 0) : p;
    if (p.px2 && p.NI > 1) return plan_impl(g, in16, false);
    p.buf_bytes = p.TB * p.PH * p.LW * (CHS * 2 + 16);
    // (two workgroups per CU up to 78 KB; the 8x8 / 16x16 layers, whose tiles span several images, take up to 150 KB and run one)
    if (2 * p.buf_bytes > 150 * 1024) return tile ? plan_impl(g, in16, allow_px2, 0) : p;
    p.nstages = cdiv(g.Cin, CHS);
    // k-steps of the packed weights (conv_patch.hip layout: 32-channel chunks x phases x taps x 2)
    p.wsteps = cdiv(g.Cin, 32) * (IS == 2 ? 4 : 1) * (IS == 2 ? 4 : R * R) * 2;
    p.ntiles = cdiv(g.Cout, p.bn);
    p.ncls = p.px2 ? 2 : g.OS * g.OS;
    const int tiles = p.mtiles * p.ntiles * p.ncls;
    int ks = 1;
    const int slots = p.wc == 2 ? 256 : 512;          // resident workgroups per round (the register-tile forms run one per CU)
    if (tiles < slots) ks = std::max(1, std::min({slots / tiles, p.nstages / 4, 32}));
    p.stages_per_split = cdiv(p.nstages, ks);
    p.ksplit = cdiv(p.nstages, p.stages_per_split);
    p.slab = ((size_t)g.B * g.Cout * g.OH * g.OW + 3) / 4 * 4;
    p.ws_bytes = p.ksplit > 1 ? p.slab * p.ksplit * sizeof(float) : 0;
    // LDS-staged 16-byte output stores (16-bit output only): unit-stride lattice, rows of whole 8-pixel chunks, room for [bn][136] x 2 B
    p.lds_epi = (g.OS == 1 && p.twl >= 3 && (g.OW & 7) == 0) ? 1 : 0;
    p.smem_bytes = std::max(2 * p.buf_bytes, p.lds_epi ? p.bn * ((2 << lp) + 16) : 0);
    if (p.smem_bytes > 160 * 1024) return tile ? plan_impl(g, in16, allow_px2, 0) : P16Plan{};
    p.dNXB = make_fastdiv((unsigned)p.NXB);
    p.dPH = make_fastdiv((unsigned)p.PH);
    p.dTB = make_fastdiv((unsigned)p.TB);
    p.dNT = make_fastdiv((unsigned)p.ntiles);
    p.dNCLS = make_fastdiv((unsigned)p.ncls);
    p.dMT = make_fastdiv((unsigned)p.mtiles);
    p.dTX = make_fastdiv((unsigned)p.tiles_x);
    p.dTY = make_fastdiv((unsigned)p.tiles_y);
    p.ok = 1;
    return p;
}

P16Plan plan_p16(const Geom& g, bool in16) {
    // AGAN_P16_TILE (experiments): 0 = the round-3 tile everywhere, 1 / 2 = force the 128 x 256 / 256 x 128 register-tile form where a
    // geometry can take it; unset = the measured choice below
    static const char* force = getenv("AGAN_P16_TILE");
    if (force) return plan_impl(g, in16, true, atoi(force));
    return plan_impl(g, in16, true, 0);
}

void launch_p16_gather(const void* in, const void* wk, const float* bias, void* dst, const Geom& g, const P16Plan& p, int prec, int act,
                       const void* lrelu_mask, hipStream_t st, bool in16, bool out16) {
    if (prec == AGAN_PREC_F16) launch_gk<1>(in, wk, bias, dst, g, p, act, lrelu_mask, st, in16, out16);
    else launch_gk<0>(in, wk, bias, dst, g, p, act, lrelu_mask, st, in16, out16);
}

}  // namespace conv
}  // namespace agan

#ifdef AGAN_P16_TIMING
extern "C" int agan_debug_p16_stamps(unsigned long long* host, int nblocks) {
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_p16_stamps), sizeof(unsigned long long) * 4 * (size_t)nblocks) != hipSuccess) return -1;
    void* dev = nullptr;
    if (hipGetSymbolAddress(&dev, HIP_SYMBOL(g_p16_stamps)) != hipSuccess) return -1;
    return hipMemset(dev, 0, sizeof(unsigned long long) * 4 * 16384) == hipSuccess ? 0 : -1;      // cleared for the next read
}
#endif
