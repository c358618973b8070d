// Shared pieces of the convolution engine (geometry, reduction-index table, raw-buffer access, tap folding, launch plans).
#pragma once
#include "agan_common.h"

#include <algorithm>
#include <type_traits>

namespace agan {
namespace conv {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Every kernel launch of the conv engine goes through AGAN_LAUNCH, which remembers the host stub of the kernel it launched (thread-local):
// the measurement hook (agan_timer_arm / agan_timer_last_kernel, conv.hip) can then NAME the kernel a timed call ran -- the demangled
// symbol, exactly as `rocprofv3 --kernel-trace --stats` prints it -- instead of the caller guessing a label.
extern thread_local const void* g_noted_kernel;
#define AGAN_LAUNCH(kernel, ...)                                                   \
    do {                                                                           \
        ::agan::conv::g_noted_kernel = reinterpret_cast<const void*>(kernel);      \
        hipLaunchKernelGGL(kernel, __VA_ARGS__);                                   \
    } while (0)


// n / d for n < 2^31 by multiply-high (Granlund-Montgomery round-up form): 2 VALU ops instead of a ~40-instruction sequence
struct FastDiv {
    unsigned mul, shift;
    __device__ __forceinline__ int div(int n) const { return (int)((__umulhi(mul, (unsigned)n) + (unsigned)n) >> shift); }
};
__host__ __device__ inline FastDiv make_fastdiv(unsigned d) {
    FastDiv f;
    unsigned l = 0;
    while ((1ull << l) < d) ++l;
    f.mul = (unsigned)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
    f.shift = l;
    return f;
}

struct Geom {
    int B, Cin, IH, IW, Cout, OH, OW, R, S, OS, SY, DY, OY0, OY1;
    int OHs, OWs, HWs, Mtot, K, Nld, RS;
    FastDiv dHWs, dOWs;
};

// Reduction-index table: entry k = (ci, r, s) -> { element offset ci*IH*IW + r*DY*IW + s*DY , packed (dy = r*DY, dx = s*DY) }.
// Built once per geometry (agan_conv_ktable) and read with scalar loads, so the kernels spend no ALU work on decoding k.
// Entries >= K are sentinels whose dy fails every range check.
constexpr int kTabPad = 128;
constexpr int kSentinelDy = -32768;
constexpr int kMaxTaps = 16;           // taps per class the gather kernels' LDS tap table holds (RS <= 16: up to 4x4)
constexpr int kTapRowBytes = 128 * 4;  // one tap row = 128 pixels of a workgroup tile
// weight-gradient kernels add a per-lane channel offset to the tabulated tap offset, so their out-of-range marker must
// survive that add without wrapping: tensors on that path stay below 2^31 bytes (checked on the host)
constexpr unsigned kWOOB = 0x80000000u;
__host__ __device__ inline int ktable_entries(int K) { return (K + kTabPad - 1) / kTabPad * kTabPad + kTabPad; }

inline Geom make_geom(const agan_conv_geom* g) {
    Geom d;
    d.B = g->B; d.Cin = g->Cin; d.IH = g->IH; d.IW = g->IW; d.Cout = g->Cout; d.OH = g->OH; d.OW = g->OW;
    d.R = g->R; d.S = g->S; d.OS = g->OS; d.SY = g->SY; d.DY = g->DY; d.OY0 = g->OY[0]; d.OY1 = g->OY[1];
    d.OHs = g->OH / g->OS; d.OWs = g->OW / g->OS; d.HWs = d.OHs * d.OWs; d.Mtot = g->B * d.HWs;
    d.RS = g->R * g->S; d.K = g->Cin * d.RS; d.Nld = agan_round_up(g->Cout, 32);
    d.dHWs = make_fastdiv((unsigned)d.HWs); d.dOWs = make_fastdiv((unsigned)d.OWs);
    return d;
}

inline int check_geom(const agan_conv_geom* g) {
    AGAN_REQUIRE(g != nullptr, "conv: null geometry");
    AGAN_REQUIRE(g->B > 0 && g->Cin > 0 && g->IH > 0 && g->IW > 0 && g->Cout > 0 && g->OH > 0 && g->OW > 0,
                 "conv: non-positive dimension");
    AGAN_REQUIRE(g->R > 0 && g->S > 0 && g->R * g->S <= 16, "conv: taps %dx%d unsupported", g->R, g->S);
    AGAN_REQUIRE(g->OS == 1 || g->OS == 2, "conv: OS must be 1 or 2");
    AGAN_REQUIRE(g->OH % g->OS == 0 && g->OW % g->OS == 0, "conv: OH/OW not divisible by OS");
    const long long in_elems = 1LL * g->B * g->Cin * g->IH * g->IW, out_elems = 1LL * g->B * g->Cout * g->OH * g->OW;
    AGAN_REQUIRE(in_elems < (1LL << 30) && out_elems < (1LL << 30), "conv: tensor exceeds 2^30 elements (32-bit buffer offsets)");
    return AGAN_OK;
}

// Raw buffer resources: the hardware range check turns every out-of-image / out-of-tile access into a zero load
// (or a dropped store) -- no exec-mask branches around the gathers (cdna_hip_programming.md T8).
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned kOOB = 0xFFFFFFFFu;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, size_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)(unsigned)bytes, 0x00020000);
}
__device__ __forceinline__ float buf_load(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}
__device__ __forceinline__ float buf_load_s(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {   // + wave-uniform byte offset
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}
__device__ __forceinline__ void buf_store(__amdgpu_buffer_rsrc_t r, unsigned off, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, off, 0, 0);
}

// which 3x3 taps fold into tap t' of parity class p for Upsample(x2)+conv3x3:  p=0: {0},{1,2}; p=1: {0,1},{2}
__device__ __host__ inline void up_fwd_taps(int p, int t, int& lo, int& hi) {
    if (p == 0) { lo = t == 0 ? 0 : 1; hi = t == 0 ? 0 : 2; }
    else        { lo = t == 0 ? 0 : 2; hi = t == 0 ? 1 : 2; }
}
// which 3x3 taps fold into tap t (0..3) of the 4x4 s2 dgrad kernel: {2},{1,2},{0,1},{0}
__device__ __host__ inline void up_dgrad_taps(int t, int& lo, int& hi) {
    lo = 2 - t > 0 ? 2 - t : 0;
    hi = 3 - t < 2 ? 3 - t : 2;
}


// Value of packed-weight element (class cls, reduction index k, output column n) for each agan_pack_weight mode (0 = padding).
// W4(co, ci, a, b) returns element [co][ci][a][b] of the OIHW tensor (the caller's accessor: global memory or an LDS tile).
template <class W4>
__device__ inline float packed_weight_value_f(W4 w4, int mode, int cls, int k, int n, int cout, int cin, int kh, int kw) {
    float v = 0.f;
    if (mode == AGAN_PACK_FWD) {
        // k = (ci, r, s), n = co
        const int khw = kh * kw, ci = k / khw, rs = k - ci * khw, r = rs / kw, s = rs - r * kw;
        if (n < cout) v = w4(n, ci, r, s);
    } else if (mode == AGAN_PACK_DGRAD_S1) {
        // k = (co, r, s), n = ci
        const int khw = kh * kw, co = k / khw, rs = k - co * khw, r = rs / kw, s = rs - r * kw;
        if (n < cin) v = w4(co, n, kh - 1 - r, kw - 1 - s);
    } else if (mode == AGAN_PACK_DGRAD_4x4S2) {
        // cls = (py,px); k = (co, r, s) with r,s in {0,1}; n = ci; tap kh = ((py+1)&1) + 2r
        const int py = cls >> 1, px = cls & 1, co = k >> 2, r = (k >> 1) & 1, s = k & 1;
        const int th = ((py + 1) & 1) + 2 * r, tw = ((px + 1) & 1) + 2 * s;
        if (n < cin) v = w4(co, n, th, tw);
    } else if (mode == AGAN_PACK_UP_FWD) {
        // cls = (py,px); k = (ci, r', s'); n = co
        const int py = cls >> 1, px = cls & 1, ci = k >> 2, r = (k >> 1) & 1, s = k & 1;
        if (n < cout) {
            int rl, rh, sl, sh;
            up_fwd_taps(py, r, rl, rh);
            up_fwd_taps(px, s, sl, sh);
            for (int a = rl; a <= rh; ++a)
                for (int b = sl; b <= sh; ++b) v += w4(n, ci, a, b);
        }
    } else if (mode == AGAN_PACK_UP_DGRAD) {
        // k = (co, t, u) with t,u in 0..3; n = ci
        const int co = k >> 4, t = (k >> 2) & 3, u = k & 3;
        if (n < cin) {
            int rl, rh, sl, sh;
            up_dgrad_taps(t, rl, rh);
            up_dgrad_taps(u, sl, sh);
            for (int a = rl; a <= rh; ++a)
                for (int b = sl; b <= sh; ++b) v += w4(co, n, a, b);
        }
    }
    return v;
}
struct GlobalOIHW {
    const float* w;
    int cin, kh, kw;
    __device__ __forceinline__ float operator()(int co, int ci, int a, int b) const { return w[(((size_t)co * cin + ci) * kh + a) * kw + b]; }
};
__device__ inline float packed_weight_value(const float* __restrict__ w, int mode, int cls, int k, int n, int cout, int cin, int kh, int kw) {
    return packed_weight_value_f(GlobalOIHW{w, cin, kh, kw}, mode, cls, k, n, cout, cin, kh, kw);
}
// does output column n run over the conv's OUTPUT channels (forward-type packs) or its input channels (data-gradient packs)?
__host__ __device__ inline bool pack_n_is_cout(int mode) { return mode == AGAN_PACK_FWD || mode == AGAN_PACK_UP_FWD; }

__host__ __device__ inline int pack_dims(int mode, int cout, int cin, int kh, int kw, int& ncls, int& K, int& N) {
    switch (mode) {
        case AGAN_PACK_FWD: ncls = 1; K = cin * kh * kw; N = cout; return 0;
        case AGAN_PACK_DGRAD_S1: ncls = 1; K = cout * kh * kw; N = cin; return 0;
        case AGAN_PACK_DGRAD_4x4S2: if (kh != 4 || kw != 4) return -1; ncls = 4; K = cout * 4; N = cin; return 0;
        case AGAN_PACK_UP_FWD: if (kh != 3 || kw != 3) return -1; ncls = 4; K = cin * 4; N = cout; return 0;
        case AGAN_PACK_UP_DGRAD: if (kh != 3 || kw != 3) return -1; ncls = 1; K = cout * 16; N = cin; return 0;
    }
    return -1;
}

// ---- XCD-aware workgroup order ---------------------------------------------------------------------------------------
// The dispatcher hands consecutive workgroup ids (x fastest, then y, then z) round-robin to the 8 XCDs, each with its own L2.
// Workgroups that share operand data (neighbouring pixel tiles: the conv halo; the (k, cout) tiles of one pixel chunk in the
// weight gradient) should therefore NOT have consecutive ids.  xcd_contiguous maps the linear id L in [0, T) to a position F in
// [0, T) such that every XCD walks a CONTIGUOUS range of F (a bijection): order the work so that data-sharing tiles are
// adjacent in F and they land on one XCD, close in time.
__device__ __forceinline__ int linear_block_id() { return blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z); }
__device__ __forceinline__ int xcd_contiguous(int L, int T) {
    const int xcd = L & 7, idx = L >> 3;
    int start = 0;
    for (int i = 0; i < xcd; ++i) start += (T - i + 7) >> 3;       // ids with id % 8 == i
    return start + idx;
}

// ---- launch plans ------------------------------------------------------------------------------------------------
struct GatherPlan {
    int bn, mtiles, ntiles, ncls, ksplit, kchunk;
    size_t slab, ws_bytes;
};

inline GatherPlan plan_gather(const Geom& g, int prec) {
    const int bk = prec == AGAN_PREC_F32 ? 16 : 32;              // K tile of the kernel family
    const int slots = prec == AGAN_PREC_F32 ? 768 : 512;         // resident workgroups per round: 256 CUs x 3 (f32) / x 2 (bf16x3)
    GatherPlan p;
    p.bn = g.Cout >= 96 ? 128 : (g.Cout >= 48 ? 64 : 32);
    p.mtiles = cdiv(g.Mtot, 128);
    p.ntiles = cdiv(g.Cout, p.bn);
    p.ncls = g.OS * g.OS;
    const int tiles = p.mtiles * p.ntiles * p.ncls;
    const int ktiles = cdiv(g.K, bk);
    // fewer tiles than one round of resident workgroups: split K so that the grid is one full round, never just over it
    // (one block over a multiple of the slot count costs a whole extra round)
    int ks = 1;
    // (and a workgroup keeps at least 16 K tiles: below that its prologue/epilogue and the slab traffic outweigh the extra
    // parallelism: measured sweep of the split factor on the 4x4..32x32 discriminator layers)
    if (tiles < slots) ks = std::max(1, std::min({slots / tiles, std::max(1, ktiles / 16), 32}));
    p.kchunk = cdiv(ktiles, ks) * bk;
    p.ksplit = cdiv(g.K, p.kchunk);
    p.slab = (size_t)g.B * g.Cout * g.OH * g.OW;
    p.slab = (p.slab + 3) / 4 * 4;   // keep every slab 16-B aligned
    p.ws_bytes = p.ksplit > 1 ? p.slab * p.ksplit * sizeof(float) : 0;
    return p;
}

struct WgradPlan {
    int bi, bj, itiles, jtiles, ncls, psplit, pchunk;
    size_t slab, ws_bytes;
};

inline WgradPlan plan_wgrad(const Geom& g, bool needs_combine) {
    WgradPlan p;
    // 128-row tiles unless the 64-row tiling pads K less (K = 576 for the 64-channel 3x3 layers: 9 x 64 instead of 5 x 128)
    p.bi = (g.K >= 96 && cdiv(g.K, 128) * 128 <= cdiv(g.K, 64) * 64) ? 128 : 64;
    p.bj = g.Cout >= 96 ? 128 : 64;
    p.itiles = cdiv(g.K, p.bi);
    p.jtiles = cdiv(g.Cout, p.bj);
    p.ncls = g.OS * g.OS;
    const int tiles = p.itiles * p.jtiles * p.ncls;
    const int ptiles = cdiv(g.Mtot, 32);
    // Split the pixel reduction so that the grid fills one or two rounds of resident workgroups (256 CUs x 2): whichever round
    // count the tile count divides better, one round on a tie (half the slab traffic); a workgroup keeps >= 4 pixel tiles.
    // Grids just over a round boundary (e.g. 576 workgroups) were up to 25% slower in the measured sweep.
    int ps = 1;
    if (tiles < 512) {
        const int cap = std::max(1, ptiles / 4);
        const int ps1 = std::max(1, std::min(512 / tiles, cap)), ps2 = std::max(1, std::min(1024 / tiles, cap));
        const double fill1 = (double)tiles * ps1 / 512.0, fill2 = (double)tiles * ps2 / 1024.0;
        ps = std::min(fill2 > fill1 + 1e-9 ? ps2 : ps1, 2048);
    }
    p.pchunk = cdiv(ptiles, ps) * 32;
    p.psplit = cdiv(g.Mtot, p.pchunk);
    p.slab = ((size_t)p.ncls * g.Cout * g.K + 3) / 4 * 4;
    // partial slabs (one even when unsplit: an accumulating call stages its result there) + one reduced slab when a
    // tap-combine pass follows
    const size_t nslabs = p.psplit + (needs_combine ? 1 : 0);
    p.ws_bytes = p.slab * nslabs * sizeof(float);
    return p;
}

// <= 4 output channels on the vector ALU (conv_small.hip); fp32 only
struct SmallWgradPlan {
    int pchunk, nchunk;
    size_t slab, ws_bytes;
};
bool small_n_gather_supported(const Geom& g);
void launch_gather_small_n(const void* in, const float* wk, const float* bias, float* out, const Geom& g, hipStream_t st, int in_dtype = AGAN_DT_F32);
bool small_n_wgrad_supported(const Geom& g);
SmallWgradPlan plan_wgrad_small_n(const Geom& g);
void launch_wgrad_small_n(const void* x, const float* dy, float* part, const Geom& g, const SmallWgradPlan& p, hipStream_t st, int x_dtype = AGAN_DT_F32);

// ---- patch-resident 16-bit MFMA kernels (conv_patch.hip): AGAN_PREC_BF16 / F16 / BF16X3 / BF16X6 ------------------------------
// taps of the GATHER geometry a pack mode is used with (R x S per class, input step SY): a 3x3 forward conv is the stride-1
// 'same' conv, a 4x4 one the stride-2 'down' conv (the only two the layer factory builds)
__host__ __device__ inline void pack_taps(int mode, int kh, int kw, int& R, int& S, int& SY) {
    switch (mode) {
        case AGAN_PACK_DGRAD_4x4S2: case AGAN_PACK_UP_FWD: R = S = 2; SY = 1; return;
        case AGAN_PACK_UP_DGRAD: R = S = 4; SY = 2; return;
        default: R = kh; S = kw; SY = (kh == 4) ? 2 : 1; return;
    }
}
struct PatchPlan {
    int twl, thl;                       // log2 of the tile's width / height in lattice points; images per tile = 128 >> (twl + thl)
    int tiles_x, tiles_y, tiles_b, mtiles;
    int PW, PH, PHW, PP;                // patch width, height, positions per image, positions per tile
    int NPH, NT, IS;                    // phases (input parity sub-lattices), taps per phase, input step
    int nchunks, nstages, nsteps;       // 32-channel chunks; stages = chunks x phases; k-steps = stages x taps x 2
    int tapoff[9];                      // LDS byte offset of each tap (80-byte positions)
    int tappos[9];                      // the same in positions
    int base_y[2][2], base_x[2][2];     // [output parity class][phase parity]: input coordinate of patch row/col 0 = IS * tile origin + base
    FastDiv dPP, dPHW, dPW;
    // tile decode of the gather kernel without runtime scalar divisions (round 4; set by launch_patch_gather for the launch's grid)
    FastDiv dNT, dNCLS, dMT, dTX, dTY, dNPH, dNPH2;
    int g_ntiles, g_ncls;
};
struct PatchGather {
    int bn, ntiles, ncls, ksplit, stages_per_split;
    size_t slab, ws_bytes;
};
struct PatchWgrad {
    int bj, jtiles, ncls, psplit, tiles_per_split, Kp;
    size_t slab, ws_bytes;
};
// ---- row-block staging gather for the one-plane 16-bit modes, fp32 or 16-bit activation storage (conv_p16.hip) --------------------
struct P16Plan {
    int ok;                             // 0: the kernel does not take this geometry
    int gk;                             // 0: 3x3 s1, 1: 2x2 s1 (parity classes), 2: 4x4 s2
    int twl, thl, TB;                   // tile = TB images x 2^thl rows x 2^twl columns of the output lattice (128 points)
    int tiles_x, tiles_y, tiles_b, mtiles;
    int PH, PW, LW, NXB, PXB, CHS;      // patch rows / columns per image, LDS row width (positions), 16-byte blocks per row, pixels per block, channels per stage
    int nitems, NI;                     // staging items per stage / per thread
    int nstages, wsteps;                // stages (CHS-channel chunks); k-steps of the packed weights per class
    int bn, ntiles, ncls, ksplit, stages_per_split;
    int buf_bytes, smem_bytes;          // one LDS patch buffer (the kernel holds two); dynamic LDS of the launch
    int lds_epi;                        // 16-bit output leaves through LDS as 16-byte stores
    int px2;                            // GK 1 on a stride-2 lattice: both column-parity classes in one workgroup (grid classes = row parities)
    int wc, lp;                         // cout fragments per wave (1, or 2: the 64 x 128 register tile); log2 of the lattice points per workgroup tile
    size_t slab, ws_bytes;
    FastDiv dNXB, dPH, dTB, dNT, dNCLS, dMT, dTX, dTY;
};
P16Plan plan_p16(const Geom& g, bool in16);
void launch_p16_gather(const void* in, const void* wk, const float* bias, void* dst, const Geom& g, const P16Plan& p, int prec, int act,
                       const void* lrelu_mask, hipStream_t st, bool in16, bool out16);
// ---- row-resident weight gradient (conv_wgrows.hip): conv3x3 s1 / conv4x4 s2, one- and two-plane modes, fp32 or 16-bit storage ------
struct RowsPlan {
    int ok, gk, ci2, ups;               // ups: the upsample conv, as conv3x3 on the upsampled image built in LDS;  gk 0: conv3x3 stride 1 (ci2: two 32-channel chunks per workgroup, 64 output channels), 2: conv4x4 stride 2
    int twl, thl, tbl;                  // log2 of the tile's columns (x positions), dy rows, images: 128 pixels
    int tiles_x, tiles_y, tiles_b, mtiles;
    int XR, XRT;                        // x rows per image of a tile (tile rows + halo), per tile
    int rowb, cpitch;                   // LDS bytes per x row / x channel
    int halo, NBY, drowb, dpitch;       // dy: halo blocks loaded (image rows wider than the tile); blocks loaded per row; LDS bytes per row / channel
    int plane_bytes, smem_bytes, pc;    // one operand plane = x part + dy part; pc: producer / consumer waves on two buffers
    int nxbl, nxitems, nyitems;         // log2 of the 8-pixel blocks per x row; staging items per tile
    int bj, jtiles, nchunks, ngroups, psplit, tiles_per_split;
    int NPH, NT, Kp;                    // K' layout of the result slabs: as PatchPlan / PatchWgrad
    size_t slab, ws_bytes;
    FastDiv dXRT, dXR, dNBY;
};
// ---- Winograd F(2x2, 3x3) for the fp32 conv3x3 stride-1 gathers (conv_wino.hip) ----------------------------------------------------
struct WinoPlan {
    int ok, s2, hmode;                  // hmode: conv3x3 as 32 tiles x 64 channels, half the positions per wave, two workgroups per CU;  s2: conv4x4 stride 2 (polyphase F(2x2, 2x2)) instead of conv3x3 stride 1 (F(2x2, 3x3))
    int txl, tyl;                       // log2 of the tile block's columns / rows (2x2-output tiles); images per block = 32 >> (txl + tyl)
    int blocks_x, blocks_y, blocks_b, mtiles;
    int nf, ntiles;                     // cout fragments per workgroup (4: 128 channels, 2: 64), cout tiles
    int pcp, raw_items, raw_bytes;      // raw patch: row pitch (floats), values per 8-channel chunk, LDS bytes
    int mpc, mpr;                       // ceil(65536 / patch columns), ceil(65536 / patch rows): the item -> (row, column) divisions of the staging map
    int smem_bytes;
    int ksplit, chunks_per_split;       // stride 2: input-channel split of layers that would fill the chip 1.5 times
    size_t u_bytes, slab, ws_bytes;     // transformed weights [planes][Cin / 8][Nld][8] fp32 at the head of the workspace; partial-sum slabs behind it
};
struct WinoWgradPlan {
    int ok, noct, jtiles, itiles, psplit, octs_per_split, smem_bytes;      // tile octets; 64-channel tiles; pixel split
    size_t slab, ws_bytes;
};
WinoWgradPlan plan_wino_wgrad(const Geom& g);
void launch_wino_wgrad(const float* x, const float* dy, float* part, const Geom& g, const WinoWgradPlan& p, hipStream_t st);
void launch_wino_wgrad_sum(const float* part, const Geom& g, const WinoWgradPlan& p, float* dw, int accumulate, hipStream_t st);   // slabs [tap][cout][cin] -> OIHW
WinoPlan plan_wino(const Geom& g);
void launch_wino(const float* in, const float* wk, float* out, const Geom& g, const WinoPlan& p, void* ws, hipStream_t st);
int prec_planes(int prec);
RowsPlan plan_rows_wgrad(const Geom& g, int prec, bool x16, bool y16, bool up = false);
void launch_rows_wgrad(const void* x, const void* dy, float* part, const Geom& g, const RowsPlan& p, int prec, hipStream_t st,
                       const float* x_scale, const float* dy_scale, bool x16, bool y16);
void launch_patch_wgrad_unpack(const float* red, float* dw, int cout, int cin, int kh, int kw, bool up, int NPH, int NT, int Kp, int accumulate,
                               hipStream_t st);
// direct convs: sum of the pixel-split slabs + OIHW unpack in one coalesced pass (kh * kw <= 16)
void launch_wgrad_sum_unpack(const float* slabs, int nslabs, size_t slab, float* dw, int cout, int cin, int kh, int kw, int NPH, int NT, int Kp,
                             int accumulate, hipStream_t st);
PatchWgrad plan_patch_wgrad(const Geom& g, const PatchPlan& pp, int prec);
void launch_patch_wgrad(const void* x, const void* dy, float* part, const Geom& g, const PatchPlan& pp, const PatchWgrad& p, int prec,
                        hipStream_t st, const float* x_scale = nullptr, const float* dy_scale = nullptr, bool x16 = false, bool y16 = false);
constexpr float kF16WeightScale = 2048.f;      // AGAN_PREC_F16X3 packs weights times 2^11 (|w| < 32 keeps fp16 finite)
void launch_patch_wgrad_unpack(const float* red, float* dw, int cout, int cin, int kh, int kw, bool up, const PatchPlan& pp, const PatchWgrad& p,
                               int accumulate, hipStream_t st);
bool patch_supported(const Geom& g);
PatchPlan make_patch_plan(const Geom& g);
PatchGather plan_patch_gather(const Geom& g, const PatchPlan& pp);
size_t patch_packed_weight_bytes(int mode, int cout, int cin, int kh, int kw, int prec);
int pack_weight_patch(const float* w, void* wk, int mode, int cout, int cin, int kh, int kw, int prec, hipStream_t st);
int pack_job_blocks_patch(int mode, int cout, int cin, int kh, int kw);
int pack_weights_patch(const agan_pack_job* jobs, int njobs, int total_blocks, int prec, hipStream_t st);
void launch_patch_gather(const float* in, const void* wk, const float* bias, float* dst, const Geom& g, const PatchPlan& pp,
                         const PatchGather& p, int prec, int act, const float* lrelu_mask, hipStream_t st, const float* in_amax = nullptr,
                         float* out_amax = nullptr);

}  // namespace conv
}  // namespace agan
