// Row-resident weight gradient on the 16-bit matrix cores (round 3): no transposition anywhere.
//
//     dw[cout][ci][r][s] = sum over (b, oy, ox)  dy[b][cout][oy][ox] * x[b][ci][IS*oy + r - 1][IS*ox + s - 1]
//
// The contraction index of a weight gradient is the PIXEL, and v_mfma_f32_32x32x16 wants 8 consecutive values of the contraction index
// per lane -- which is exactly what 16 bytes of an NCHW row are.  The patch-resident kernel of conv_patch.hip stages both operands
// [position][channel] (the forward kernels' layout) and reads them back through the transposing ds_read_b64_tr_b16: 20 LDS reads per
// 9 MFMAs, every x fragment read by all four waves, two barriers per 128 pixels -- 12 % (one plane) / 20 % (two planes) MFMA-busy in
// the round-3 profiles.  Here both operands live in LDS the way they live in HBM, [channel][row][column], converted to the operand
// type once on the way in (16-byte global loads along the rows, 16-byte LDS writes, nothing permuted):
//   * an x fragment (input channel l31, 8 positions) is ONE aligned ds_read_b128 whatever the tap: the row taps are row offsets, and
//     the column taps are not taken on this side.  Stride 2 keeps the columns de-interleaved [even | odd]: tap s reads parity (s-1)&1;
//   * the +-1-position column taps are taken on the dy side: next to its aligned 16-byte block a lane reads the dword before and after
//     it and builds the three shifted fragments dy[j-1 ..], dy[j ..], dy[j+1 ..] with four v_alignbit_b32 each (unaligned 16-byte LDS
//     reads serialise per lane on gfx950: profiles/micro/lds_unaligned.hip, 256 vs 33 cycles).  The dy rows carry one halo block on
//     either side; it is loaded when the image row continues there and stays zero otherwise;
//   * a wave keeps all its taps in accumulators: 9 MFMAs per 3 x reads + 3 dy reads (3x3), 8 per 4 + 3 (4x4 stride 2).
// (A first version fetched the dy fragments straight from global memory -- a lane per output channel, 16 bytes each: 64 cache lines
//  per wave-load, and the texture path, not the matrix core, set the pace: 1.25x over the old kernel instead of the 2x+ of this one.)
// Column tiles are tiles of the X positions j (a pair (dy[ox], x[j]) is counted in the tile that holds j), row tiles are tiles of the
// dy rows (x rows carry the halo); everything outside the image is a zero load.
//
// Geometry kinds: GK 0  conv3x3 stride 1 pad 1: 4 waves = 4 cout fragments x 9 taps on one 32-channel chunk (CI2: 2 cout fragments x
//                       2 chunks, for layers with <= 64 output channels);
//                 GK 2  conv4x4 stride 2 pad 1: 4 waves = 2 cout fragments x 2 kernel-row pairs x 8 taps.
// 128 pixels per tile, two LDS barriers per tile, the next tile's global loads in flight during the k loop.
// Output: the same [split][cout][K'] slabs as conv_patch.hip's kernel (K' = ((chunk * NPH + phase) * NT + tap) * 32 + ci), so the slab
// sum and the OIHW unpack are shared.
#include "conv_common.h"
#include "split16.h"

#include <cstdlib>
#include <cstring>

using namespace agan;
using namespace agan::conv;

namespace {

__device__ __forceinline__ unsigned alignbit16(unsigned hi, unsigned lo) { return __builtin_amdgcn_alignbit(hi, lo, 16); }   // (lo >> 16) | (hi << 16)

// 8 stored values (one or two 16-byte blocks) -> NPL planes of (p0 p1) (p2 p3) (p4 p5) (p6 p7)
template <int ET, int NPL, bool S16, bool SCALED>
__device__ __forceinline__ void to_planes(const u32x4 (&raw)[S16 ? 1 : 2], float scale, u32x4 (&pl)[NPL]) {
    if constexpr (S16) {
        pl[0] = raw[0];
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            unsigned e[NPL];
            const float a = __uint_as_float(raw[k >> 1][(2 * k) & 3]), b = __uint_as_float(raw[k >> 1][(2 * k + 1) & 3]);
            if (SCALED) split_pack2<ET, NPL>(a * scale, b * scale, e);
            else split_pack2<ET, NPL>(a, b, e);
#pragma unroll
            for (int p = 0; p < NPL; ++p) pl[p][k] = e[p];
        }
    }
}

// PC (one-plane modes, two LDS buffers): 8 waves, PRODUCER / CONSUMER split.  Waves 0-3 only read fragments and issue MFMAs; waves 4-7 only
// move the next tile (global loads, conversion, LDS writes) into the other buffer; one barrier per tile.  With all of it in the same 4
// waves (one per SIMD: 270-320 registers) the matrix core idled while a wave sat in the load issue (64 KB per tile through the texture
// path: ~1600 of 5800 cycles per tile by clock64), in the conversion + LDS writes (~1200) and at the second barrier -- 37 % of a tile's
// time was its k loop; interleaving the staging work INTO the k loop of the same waves did not help either (a wave blocked on a full
// vector-memory queue issues no MFMA).
template <int ET, int NPL, int GK, bool CI2, bool X16, bool Y16, int NIX, int NIY, bool PC, bool UPS>
__global__ __launch_bounds__(PC ? 512 : 256, 1) void conv_wgrad_rows_kernel(const void* __restrict__ x, const void* __restrict__ dy, float* __restrict__ dst,
                                                                 const Geom g, const RowsPlan rp, const float* __restrict__ x_scale,
                                                                 const float* __restrict__ dy_scale) {
    // ET 2: fp32 operands on v_mfma_f32_32x32x2_f32 (AGAN_PREC_F32).  A 16-byte block is then 4 pixels, a k-group (8 pixels: 4 per lane half)
    // is four MFMAs per tap with the block's registers as they are, and the column shifts are a choice of registers.
    constexpr bool F32 = ET == 2;
    constexpr bool SCALED = (ET == 1 && NPL == 2);               // AGAN_PREC_F16X3 (x_scale / dy_scale: amax slots)
    static_assert(!(X16 || Y16) || (NPL == 1 && !F32), "16-bit activation storage goes with the one-plane 16-bit modes");
    static_assert(!F32 || (NPL == 1 && !PC && !UPS), "fp32 operands: one plane, one buffer, direct convs");
    static_assert(!CI2 || GK == 0, "the two-chunk layout is the 3x3 kernel's");
    static_assert(!UPS || GK == 0, "the upsample conv is a 3x3 conv on the upsampled image");
    float xsc = 1.f, ysc = 1.f, unscale = 1.f;
    if (SCALED) {
        float ix, iy;
        xsc = amax_scale(amax_read(x_scale), &ix);
        ysc = amax_scale(amax_read(dy_scale), &iy);
        unscale = ix * iy;
    }
    constexpr int IS = GK == 2 ? 2 : 1;
    constexpr int NWJ = (GK == 2 || CI2) ? 2 : 4;    // cout fragments per workgroup
    constexpr int BJ = NWJ * 32;
    constexpr int NCI = CI2 ? 64 : 32;               // input channels per workgroup
    constexpr int NTW = GK == 2 ? 8 : 9;             // taps per wave
    constexpr int NB = GK == 2 ? 4 : 3;              // x fragments per k-step and wave
    constexpr int PXL = F32 ? 2 : 3;                 // log2 of the pixels per 16-byte LDS block
    constexpr int PXB = 1 << PXL;
    constexpr int NK = 128 / (2 * PXB);              // k-steps per tile (128 pixels; two blocks per k-step, one per lane half)
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = PC && wave >= 4;
    const int stid = tid & 255;                      // staging-item index of this thread (PC: of the producer threads)
    const int wj = (wave & 3) % NWJ, part = (wave & 3) / NWJ;    // GK 2: kernel rows 2 * part, 2 * part + 1;  CI2: chunk 2 * chunk2 + part
    const int l31 = lane & 31, lh = lane >> 5;
    int jt, chunk, split;
    {
        int F = xcd_contiguous(linear_block_id(), rp.jtiles * rp.ngroups * rp.psplit);
        jt = F % rp.jtiles; F /= rp.jtiles;
        chunk = F % rp.ngroups; split = F / rp.ngroups;           // (CI2: a group is two chunks)
    }
    const int j0 = jt * BJ, c0 = chunk * NCI;
    const int twl = rp.twl, thl = rp.thl, bxl = twl - PXL;
    const int ihw = g.IH * g.IW, ohw = g.OH * g.OW;
    constexpr unsigned XE = X16 ? 2u : 4u, YE = Y16 ? 2u : 4u;
    // stored elements per staging item: one 16-byte LDS block = PXB operand values; from fp32 storage in a 16-bit mode that is two loads
    constexpr int XRAW = (F32 || X16 || UPS) ? 1 : 2, YRAW = (F32 || Y16) ? 1 : 2;
    // UPS: x is the LOW-RES tensor (IH/2 x IW/2); g describes the conv on its nearest-neighbour upsampling, which is built on the way into LDS
    const int sw = UPS ? g.IW >> 1 : g.IW, shw = UPS ? ihw >> 2 : ihw;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(x, (size_t)g.B * g.Cin * shw * XE);
    const __amdgpu_buffer_rsrc_t rdy = make_rsrc(dy, (size_t)g.B * g.Cout * ohw * YE);
    const int tile_beg = split * rp.tiles_per_split, tile_end = min(rp.mtiles, tile_beg + rp.tiles_per_split);
    const int rowb = rp.rowb, cpitch = rp.cpitch, drowb = rp.drowb, dpitch = rp.dpitch, plane_bytes = rp.plane_bytes;
    // LDS: [buffer (PC: 2)][plane][x part: NCI channels x cpitch | dy part: BJ channels x dpitch]
    static_assert(!PC || NPL == 1, "the producer / consumer split is the one-plane modes'");
    const int buf_bytes = NPL * plane_bytes;

    struct Tile { int b0, y0, x0; bool ok; };
    auto tile_of = [&](int mt) {
        Tile t;
        t.ok = mt < tile_end;
        const int txi = mt % rp.tiles_x, tyi = (mt / rp.tiles_x) % rp.tiles_y, tbi = mt / (rp.tiles_x * rp.tiles_y);
        t.b0 = tbi << rp.tbl; t.y0 = tyi << thl; t.x0 = txi << twl;
        return t;
    };

    // ---- staging items, tile independent parts, ONE packed register each (decoded when used: a handful of VALU ops per item and tile).
    //      x: (channel, tile x row, 16-byte block of PXB input pixels); dy: (channel, tile row, block incl. the halo blocks when the image row
    //      continues beyond the tile); blocks fastest: a wave-load walks along rows.  A thread's x block index is the same for all its items
    //      (the blocks per row are a power of two <= 256). ----
    const int xi_xb = stid & ((1 << rp.nxbl) - 1);
    int xi_pack[NIX];                 // ci << 16 | tb << 8 | row of the image's patch;  -1: no item
#pragma unroll
    for (int i = 0; i < NIX; ++i) {
        const int e = stid + i * 256;
        const int t1 = e >> rp.nxbl;
        const int ci = rp.dXRT.div(t1), rowi = t1 - ci * rp.XRT;
        const int tb = rp.dXR.div(rowi), j = rowi - tb * rp.XR;
        xi_pack[i] = ((e < rp.nxitems) & (c0 + ci < g.Cin)) ? (ci << 16 | tb << 8 | j) : -1;
    }
    int yi_pack[NIY];                 // nn << 16 | tile row << 8 | block + 1;  -1: no item
#pragma unroll
    for (int i = 0; i < NIY; ++i) {
        const int e = stid + i * 256;
        const int t1 = rp.dNBY.div(e), blk = e - t1 * rp.NBY - rp.halo;      // -1 .. TW/PXB with the halo, else 0 .. TW/PXB - 1
        const int nn = t1 >> (thl + rp.tbl), rowi = t1 & ((1 << (thl + rp.tbl)) - 1);
        yi_pack[i] = ((e < rp.nyitems) & (j0 + nn < g.Cout)) ? (nn << 16 | rowi << 8 | (blk + 1)) : -1;
    }
    u32x4 xr[NIX][XRAW], yr[NIY][YRAW];
    auto load_tile = [&](const Tile& t) {
#pragma unroll
        for (int i = 0; i < NIX; ++i) {
            const int pk = xi_pack[i];
            const int ci = pk >> 16, tb = (pk >> 8) & 255, j = pk & 255;
            const int b = t.b0 + tb, iy = IS * t.y0 + j - 1, ix = IS * t.x0 + PXB * xi_xb;
            const bool ok = t.ok & (pk >= 0) & (b < g.B) & ((unsigned)iy < (unsigned)g.IH) & (ix < g.IW);
            if constexpr (UPS) {
                // 8 upsampled pixels = 4 source pixels of source row iy / 2
                const unsigned e0 = (unsigned)((b * g.Cin + c0 + ci) * shw + (iy >> 1) * sw + (ix >> 1));
                if constexpr (X16) {
                    const uint2 v = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rx, ok ? e0 * XE : kOOB, 0, 0));
                    xr[i][0] = u32x4{v.x, v.y, 0u, 0u};
                } else {
                    xr[i][0] = buf_load_u4s(rx, ok ? e0 * XE : kOOB, 0);
                }
            } else {
                const unsigned e0 = (unsigned)((b * g.Cin + c0 + ci) * ihw + iy * g.IW + ix);
                xr[i][0] = buf_load_u4s(rx, ok ? e0 * XE : kOOB, 0);
                if constexpr (XRAW == 2) xr[i][1] = buf_load_u4s(rx, ok ? (e0 + 4u) * XE : kOOB, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < NIY; ++i) {
            const int pk = yi_pack[i];
            const int nn = pk >> 16, rowi = (pk >> 8) & 255, blk = (pk & 255) - 1;
            const int b = t.b0 + (rowi >> thl), oy = t.y0 + (rowi & ((1 << thl) - 1)), ox = t.x0 + PXB * blk;
            const bool ok = t.ok & (pk >= 0) & (b < g.B) & (oy < g.OH) & ((unsigned)ox < (unsigned)g.OW);
            const unsigned e0 = (unsigned)((b * g.Cout + j0 + nn) * ohw + oy * g.OW + ox);
            yr[i][0] = buf_load_u4s(rdy, ok ? e0 * YE : kOOB, 0);
            if constexpr (YRAW == 2) yr[i][1] = buf_load_u4s(rdy, ok ? (e0 + 4u) * YE : kOOB, 0);
        }
    };
    auto store_tile = [&](unsigned char* buf) {
#pragma unroll
        for (int i = 0; i < NIX; ++i) {
            const int pk = xi_pack[i];
            if (stid + i * 256 >= rp.nxitems) continue;
            const int ci = pk >> 16, tb = (pk >> 8) & 255, j = pk & 255;       // (a channel past Cin: pk = -1 -> a zero block somewhere inside the buffer's x part)
            const unsigned xl = pk >= 0 ? (unsigned)(ci * cpitch + (tb * rp.XR + j) * rowb + xi_xb * (IS == 2 ? 8 : 16)) : 0u;
            u32x4 pl[NPL];
            if constexpr (F32) {
                pl[0] = xr[i][0];
            } else if constexpr (UPS) {
                // every source pixel twice
                if constexpr (X16) {
                    pl[0] = u32x4{__builtin_amdgcn_perm(xr[i][0][0], xr[i][0][0], 0x01000100u), __builtin_amdgcn_perm(xr[i][0][0], xr[i][0][0], 0x03020302u),
                                  __builtin_amdgcn_perm(xr[i][0][1], xr[i][0][1], 0x01000100u), __builtin_amdgcn_perm(xr[i][0][1], xr[i][0][1], 0x03020302u)};
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        unsigned e[NPL];
                        const float a = __uint_as_float(xr[i][0][k]);
                        if (SCALED) split_pack2<ET, NPL>(a * xsc, a * xsc, e);
                        else split_pack2<ET, NPL>(a, a, e);
#pragma unroll
                        for (int p = 0; p < NPL; ++p) pl[p][k] = e[p];
                    }
                }
            } else {
                to_planes<ET, NPL, X16, SCALED>(xr[i], xsc, pl);
            }
            if (pk < 0) continue;
#pragma unroll
            for (int p = 0; p < NPL; ++p) {
                unsigned char* d = buf + p * plane_bytes + xl;
                if (IS == 1) {
                    *reinterpret_cast<u32x4*>(d) = pl[p];
                } else if (F32) {
                    *reinterpret_cast<uint2*>(d) = make_uint2(pl[p][0], pl[p][2]);                   // even columns
                    *reinterpret_cast<uint2*>(d + (rowb >> 1)) = make_uint2(pl[p][1], pl[p][3]);     // odd columns
                } else {
                    const uint2 ev = make_uint2(__builtin_amdgcn_perm(pl[p][1], pl[p][0], 0x05040100u), __builtin_amdgcn_perm(pl[p][3], pl[p][2], 0x05040100u));
                    const uint2 od = make_uint2(__builtin_amdgcn_perm(pl[p][1], pl[p][0], 0x07060302u), __builtin_amdgcn_perm(pl[p][3], pl[p][2], 0x07060302u));
                    *reinterpret_cast<uint2*>(d) = ev;
                    *reinterpret_cast<uint2*>(d + (rowb >> 1)) = od;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NIY; ++i) {
            const int pk = yi_pack[i];
            if (stid + i * 256 >= rp.nyitems || pk < 0) continue;
            const unsigned yl = (unsigned)((pk >> 16) * dpitch + ((pk >> 8) & 255) * drowb + (pk & 255) * 16);
            u32x4 pl[NPL];
            if constexpr (F32) pl[0] = yr[i][0];
            else to_planes<ET, NPL, Y16, SCALED>(yr[i], ysc, pl);
#pragma unroll
            for (int p = 0; p < NPL; ++p) *reinterpret_cast<u32x4*>(buf + NCI * cpitch + p * plane_bytes + yl) = pl[p];
        }
    };

    // ---- fragments of k-step kk: block q = 2 * kk + lh of the tile.  x: channel l31 (of this wave's chunk); dy: output channel l31 of
    //      this wave's fragment, as its aligned block and the dwords on either side ----
    const unsigned blane = (unsigned)(((CI2 ? part * 32 : 0) + l31) * cpitch);
    const unsigned alane = (unsigned)((wj * 32 + l31) * dpitch);
    struct AF { u32x4 c; unsigned l, r; };
    auto read_frags = [&](const unsigned char* buf, int kk, u32x4 (&bf)[NB][NPL], AF (&af)[NPL]) {
        const int q = 2 * kk + lh;
        const int bx = q & ((1 << bxl) - 1), trow = q >> bxl;                    // trow = tb * TH + ty
        const int ty = trow & ((1 << thl) - 1), tb = trow >> thl;
        const unsigned off = blane + (unsigned)((tb * rp.XR + IS * ty) * rowb + bx * 16);
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            // GK 0: i = kernel row r.  GK 2: i = rr * 2 + column parity, kernel row r = 2 * part + rr
            const unsigned o = GK == 2 ? off + (unsigned)((2 * part + (i >> 1)) * rowb + (i & 1) * (rowb >> 1)) : off + (unsigned)(i * rowb);
#pragma unroll
            for (int p = 0; p < NPL; ++p) bf[i][p] = *reinterpret_cast<const u32x4*>(buf + p * plane_bytes + o);
        }
        const unsigned ao = alane + (unsigned)(trow * drowb + (bx + 1) * 16);
#pragma unroll
        for (int p = 0; p < NPL; ++p) {
            const unsigned char* a = buf + NCI * cpitch + p * plane_bytes + ao;
            af[p].c = *reinterpret_cast<const u32x4*>(a);
            af[p].l = *reinterpret_cast<const unsigned*>(a - 4);                // 16-bit: positions -2, -1;  fp32: position -1
            af[p].r = *reinterpret_cast<const unsigned*>(a + 16);               // 16-bit: positions 8, 9;   fp32: position 4
        }
    };

    f32x16 acc[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // one tap's products of a k-step: av = the dy fragment in the tap's column shift, b = the x fragment of the tap's row (and parity)
    auto tap_mfma = [&](f32x16 c, const u32x4 (&av)[NPL], const u32x4 (&b)[NPL]) {
        if constexpr (F32) {
#pragma unroll
            for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(av[0][j]), __uint_as_float(b[0][j]), c, 0, 0, 0);
            return c;
        } else {
            return mfma_split<ET, NPL>(av, b, c);
        }
    };
    // the k loop of one tile out of buffer `buf`
    auto k_loop = [&](const unsigned char* buf) {
        u32x4 bf[2][NB][NPL];
        AF af[2][NPL];
        read_frags(buf, 0, bf[0], af[0]);
#pragma unroll
        for (int kk = 0; kk < NK; ++kk) {
            if (kk + 1 < NK) read_frags(buf, kk + 1, bf[(kk + 1) & 1], af[(kk + 1) & 1]);
#ifndef AGAN_WG_NOPIN
            // keep the next k-step's LDS reads HERE, in front of this k-step's MFMAs: left alone, the scheduler sank them behind the MFMAs
            // and every k-step waited out the LDS latency of its own fragments.  (Consumer waves alone, producers idle: 56 -> 52 us on the
            // bf16 stride-2 layers; the full kernel, 74 us, did not move -- profiles/r03_rows_ablation.txt has where its time goes.)
            __builtin_amdgcn_sched_barrier(0);
#endif
            // the three shifted dy fragments av[0] = dy[j-1 ..], av[1] = dy[j ..], av[2] = dy[j+1 ..]
            u32x4 av[3][NPL];
#pragma unroll
            for (int p = 0; p < NPL; ++p) {
                const AF& a = af[kk & 1][p];
                const unsigned d[6] = {a.l, a.c[0], a.c[1], a.c[2], a.c[3], a.r};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    av[0][p][k] = F32 ? d[k] : alignbit16(d[k + 1], d[k]);
                    av[1][p][k] = d[k + 1];
                    av[2][p][k] = F32 ? d[k + 2] : alignbit16(d[k + 2], d[k + 1]);
                }
            }
            const auto& b = bf[kk & 1];
            if (GK == 0) {
                // tap (r, s): dy shifted by 1 - s against the x positions
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int s = 0; s < 3; ++s) acc[r * 3 + s] = tap_mfma(acc[r * 3 + s], av[2 - s], b[r]);
            } else {
                // tap (rr, s): s = 0 -> (dy[j+1], odd), 1 -> (dy[j], even), 2 -> (dy[j], odd), 3 -> (dy[j-1], even)
#pragma unroll
                for (int rr = 0; rr < 2; ++rr) {
                    acc[rr * 4 + 0] = tap_mfma(acc[rr * 4 + 0], av[2], b[rr * 2 + 1]);
                    acc[rr * 4 + 1] = tap_mfma(acc[rr * 4 + 1], av[1], b[rr * 2 + 0]);
                    acc[rr * 4 + 2] = tap_mfma(acc[rr * 4 + 2], av[1], b[rr * 2 + 1]);
                    acc[rr * 4 + 3] = tap_mfma(acc[rr * 4 + 3], av[0], b[rr * 2 + 0]);
                }
            }
        }
    };

    if (tile_beg < tile_end) {
        // the dy halo blocks are written only where the image row continues: zero the dy parts once
        for (int o = tid * 16; o < (PC ? 2 : 1) * buf_bytes; o += (PC ? 512 : 256) * 16) {
            const int r = o % plane_bytes;
            if (r >= NCI * cpitch) *reinterpret_cast<u32x4*>(lds + o) = u32x4{0u, 0u, 0u, 0u};
        }
        if (PC) {
            // two separate loops with the same barrier count: nothing of one role is live in the other (accumulators / staging registers)
            if (producer) {
                load_tile(tile_of(tile_beg));
                lds_barrier();
                store_tile(lds);
                load_tile(tile_of(tile_beg + 1));
                lds_barrier();
                for (int mt = tile_beg; mt < tile_end; ++mt) {
                    const int par = (mt - tile_beg) & 1;
                    store_tile(lds + (par ^ 1) * buf_bytes);       // tile mt + 1, loaded one iteration ago
                    load_tile(tile_of(mt + 2));                    // (past the last tile: all-zero range, nothing is fetched)
                    lds_barrier();
                }
                return;
            }
            lds_barrier();
            lds_barrier();
            for (int mt = tile_beg; mt < tile_end; ++mt) {
                k_loop(lds + ((mt - tile_beg) & 1) * buf_bytes);
                lds_barrier();
            }
        } else {
            load_tile(tile_of(tile_beg));
            lds_barrier();
            store_tile(lds);
            lds_barrier();
            for (int mt = tile_beg; mt < tile_end; ++mt) {
                load_tile(tile_of(mt + 1));              // (past the last tile: all-zero range, nothing is fetched)
                k_loop(lds);
                lds_barrier();                           // every wave has read the tile
                if (mt + 1 < tile_end) {
                    store_tile(lds);
                    lds_barrier();
                }
            }
        }
    }
    if (producer) return;

    // ---- D[cout][channel of the chunk] per tap -> dst[split][cout][kprime], kprime = ((chunk * NPH + ph) * NT + tap) * 32 + ci ----
    float* o = dst + (size_t)split * rp.slab;
    const __amdgpu_buffer_rsrc_t ro = make_rsrc(o, (size_t)g.Cout * rp.Kp * sizeof(float));
    const int ch = CI2 ? 2 * chunk + part : chunk;
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
        int kcol;
        if (GK == 0) kcol = (ch * 9 + t) * 32 + l31;
        else {
            const int r = 2 * part + (t >> 2), s = t & 3;
            const int ph = (r & 1) * 2 + (s & 1), tp = (r >> 1) * 2 + (s >> 1);
            kcol = ((ch * 4 + ph) * 4 + tp) * 32 + l31;
        }
        const bool cok = ch * 32 < g.Cin;                      // (CI2 with an odd number of chunks: the last group's second chunk does not exist)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int nn = j0 + wj * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            buf_store(ro, (cok && nn < g.Cout) ? (unsigned)(nn * rp.Kp + kcol) * 4u : kOOB, SCALED ? acc[t][r] * unscale : acc[t][r]);
        }
    }
}

int pow2ceil_log_w(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

template <int ET, int NPL, int GK, bool CI2, bool X16, bool Y16, bool UPS>
void launch_rows(const void* x, const void* dy, float* part, const Geom& g, const RowsPlan& p, hipStream_t st, const float* xs, const float* ys) {
    // staging items per thread (16-byte LDS blocks): the plan checks the geometry against these
    constexpr int NIX = ET == 2 ? (GK == 2 ? 20 : (CI2 ? 16 : 8)) : (GK == 2 ? 10 : (CI2 ? 8 : 4));
    constexpr int NIY = ET == 2 ? (GK == 2 ? 10 : (CI2 ? 9 : 18)) : (GK == 2 ? 6 : (CI2 ? 5 : 10));
    dim3 grid(p.jtiles, p.ngroups, p.psplit);
#define AGAN_ROWS_LAUNCH(PC_)                                                                                                                         \
    do {                                                                                                                                              \
        static const hipError_t attr_ = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_rows_kernel<ET, NPL, GK, CI2, X16, Y16, NIX, NIY, PC_, UPS>), \
                                                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                                  \
        (void)attr_;                                                                                                                                  \
        AGAN_LAUNCH((conv_wgrad_rows_kernel<ET, NPL, GK, CI2, X16, Y16, NIX, NIY, PC_, UPS>), grid, dim3(PC_ ? 512 : 256), (size_t)p.smem_bytes, st, x, dy, \
                           part, g, p, xs, ys);                                                                                                       \
    } while (0)
    if constexpr (NPL == 1 && ET != 2) {
        if (p.pc) { AGAN_ROWS_LAUNCH(true); return; }
    }
    AGAN_ROWS_LAUNCH(false);
#undef AGAN_ROWS_LAUNCH
}
template <int ET, int NPL, bool X16, bool Y16>
void launch_rows_gk(const void* x, const void* dy, float* part, const Geom& g, const RowsPlan& p, hipStream_t st, const float* xs, const float* ys) {
    if (p.gk == 2) launch_rows<ET, NPL, 2, false, X16, Y16, false>(x, dy, part, g, p, st, xs, ys);
    else if (p.ups) {
        if constexpr (ET != 2) {
            if (p.ci2) launch_rows<ET, NPL, 0, true, X16, Y16, true>(x, dy, part, g, p, st, xs, ys);
            else launch_rows<ET, NPL, 0, false, X16, Y16, true>(x, dy, part, g, p, st, xs, ys);
        }
    } else {
        if (p.ci2) launch_rows<ET, NPL, 0, true, X16, Y16, false>(x, dy, part, g, p, st, xs, ys);
        else launch_rows<ET, NPL, 0, false, X16, Y16, false>(x, dy, part, g, p, st, xs, ys);
    }
}
template <int ET>
void launch_rows_dt(const void* x, const void* dy, float* part, const Geom& g, const RowsPlan& p, hipStream_t st, bool x16, bool y16) {
    if (x16 && y16) launch_rows_gk<ET, 1, true, true>(x, dy, part, g, p, st, nullptr, nullptr);
    else if (x16) launch_rows_gk<ET, 1, true, false>(x, dy, part, g, p, st, nullptr, nullptr);
    else if (y16) launch_rows_gk<ET, 1, false, true>(x, dy, part, g, p, st, nullptr, nullptr);
    else launch_rows_gk<ET, 1, false, false>(x, dy, part, g, p, st, nullptr, nullptr);
}

}  // namespace

namespace agan {
namespace conv {

// Plan of the row-resident weight gradient for FORWARD geometry g in mode prec (p.ok == 0: the kernel does not take the call).
// the geometry the kernel works on: the forward geometry itself, or -- for the folded upsample conv (4 classes of 2x2 taps on the low-res
// input) -- the plain conv3x3 on the upsampled image that it is a refactoring of: 9 taps per output pixel instead of the folded 4, on the
// 16-bit matrix core instead of the fp32 one, and the result is the 3x3 OIHW gradient directly (no class unfolding)
Geom rows_geom(const Geom& g, bool up) {
    if (!up) return g;
    Geom u = g;
    u.IH = g.OH; u.IW = g.OW;
    u.R = u.S = 3; u.RS = 9; u.K = g.Cin * 9;
    u.OS = 1; u.SY = 1; u.DY = 1; u.OY0 = u.OY1 = -1;
    u.OHs = g.OH; u.OWs = g.OW; u.HWs = g.OH * g.OW; u.Mtot = g.B * u.HWs;
    u.dHWs = make_fastdiv((unsigned)u.HWs); u.dOWs = make_fastdiv((unsigned)u.OWs);
    return u;
}

RowsPlan plan_rows_wgrad(const Geom& gf, int prec, bool x16, bool y16, bool up) {
    RowsPlan p;
    memset(&p, 0, sizeof(p));
    static const bool noup = getenv("AGAN_WG_ROWS_NOUP") != nullptr;
    if (up && (noup || gf.R != 2 || gf.S != 2 || gf.OS != 2 || gf.OH != 2 * gf.IH || gf.OW != 2 * gf.IW)) return p;
    const Geom g = rows_geom(gf, up);
    p.ups = up ? 1 : 0;
    static const bool off = getenv("AGAN_WG_ROWS_OFF") != nullptr;
    if (off) return p;
    const bool f32 = prec == AGAN_PREC_F32;
    static const bool nof32 = getenv("AGAN_WG_ROWS_NOF32") != nullptr;
    if (f32 && (up || x16 || y16 || nof32)) return p;     // (fp32: the folded upsample kernel does 4 taps per pixel instead of 9 at the same rate)
    const int planes = f32 ? 1 : prec_planes(prec);
    if (planes < 1 || planes > 2) return p;
    if ((x16 || y16) && planes != 1) return p;
    const int pxl = f32 ? 2 : 3, PXB = 1 << pxl, ESZ = f32 ? 4 : 2;      // pixels per 16-byte LDS block
    if (g.OS != 1 || g.OY0 != -1 || g.DY != 1) return p;
    int IS, R;
    if (g.SY == 1 && g.R == 3 && g.S == 3) { p.gk = 0; IS = 1; R = 3; }
    else if (g.SY == 2 && g.R == 4 && g.S == 4) { p.gk = 2; IS = 2; R = 4; }
    else return p;
    if (g.OW < PXB || (g.OW & (PXB - 1)) || g.IW != g.OW * IS || g.IH != g.OH * IS || g.Cin < 8 || g.Cout < 32) return p;
    // fp32: measured against conv.hip's k-table kernel (profiles/r03_conv_micro_wgrad.txt): 10-20 % faster on the 3x3 layers of 64-128 channels
    // (their K = 576 pads to 640 there), the same on the stride-2 layers (both sit at the sustained fp32-MFMA rate, ~120 TF/s), slower on the
    // 4x4 maps -- so only the former come here
    static const bool f32all = getenv("AGAN_WG_ROWS_F32_ALL") != nullptr;
    if (f32 && !f32all && (p.gk != 0 || g.OW < 16)) return p;
    p.ci2 = (p.gk == 0 && g.Cout <= 64) ? 1 : 0;
    const int tpl = 7;
    p.twl = std::min(p.gk == 2 ? 5 : 6, pow2ceil_log_w(g.OW));
    p.thl = std::min(tpl - p.twl, pow2ceil_log_w(g.OH));
    p.tbl = tpl - p.twl - p.thl;
    const int TW = 1 << p.twl, TH = 1 << p.thl, TB = 1 << p.tbl;
    p.tiles_x = cdiv(g.OW, TW);
    p.tiles_y = cdiv(g.OH, TH);
    p.tiles_b = cdiv(g.B, TB);
    p.mtiles = p.tiles_x * p.tiles_y * p.tiles_b;
    p.XR = IS * (TH - 1) + R;
    p.XRT = TB * p.XR;
    p.rowb = IS * TW * ESZ;
    // channel pitches = odd multiples of 16 bytes: the 8 lanes of a ds_read_b128 cycle hit 8 different 16-byte bank groups
    auto odd16 = [](int bytes) { const int q = (bytes + 15) / 16; return ((q & 1) ? q : q + 1) * 16; };
    p.cpitch = odd16(p.XRT * p.rowb + 16);
    p.halo = p.tiles_x > 1 ? 1 : 0;
    // a dy row = [left halo block][TW / 8 blocks][right halo block]; rows that never load a halo share it: the right neighbour of a row's last
    // block is the (always zero) left halo slot of the next row, the last row's is the channel's 16-byte pad
    p.drowb = (TW / PXB + 1 + p.halo) * 16;
    p.dpitch = odd16(TB * TH * p.drowb + 16);
    p.bj = (p.gk == 2 || p.ci2) ? 64 : 128;
    const int nci = p.ci2 ? 64 : 32;
    p.plane_bytes = nci * p.cpitch + p.bj * p.dpitch;
    static const bool nopc = getenv("AGAN_WG_ROWS_NOPC") != nullptr;
    p.pc = (!nopc && !f32 && planes == 1 && 2 * p.plane_bytes <= 160 * 1024) ? 1 : 0;      // producer / consumer waves on two LDS buffers
    p.smem_bytes = (p.pc ? 2 : 1) * planes * p.plane_bytes;
    if (p.smem_bytes > 160 * 1024) return p;
    p.nxbl = pow2ceil_log_w(IS * TW / PXB);
    p.nxitems = nci * p.XRT << p.nxbl;
    p.NBY = TW / PXB + 2 * p.halo;
    p.nyitems = p.bj * TB * TH * p.NBY;
    // (items per thread the kernels are instantiated for: launch_rows; fp32 storage in a 16-bit mode loads a block as two 16-byte pieces)
    const int nix = f32 ? (p.gk == 2 ? 20 : (p.ci2 ? 16 : 8)) : (p.gk == 2 ? 10 : (p.ci2 ? 8 : 4));
    const int niy = f32 ? (p.gk == 2 ? 10 : (p.ci2 ? 9 : 18)) : (p.gk == 2 ? 6 : (p.ci2 ? 5 : 10));
    if (cdiv(p.nxitems, 256) > nix || cdiv(p.nyitems, 256) > niy) return p;
    p.jtiles = cdiv(g.Cout, p.bj);
    p.nchunks = cdiv(g.Cin, 32);
    p.ngroups = cdiv(g.Cin, nci);
    const int wgs = p.jtiles * p.ngroups;
    int ps = 1;
    // one workgroup per CU is resident: split the pixel tiles until the grid is ONE round of the 256 CUs (measured sweep, round 3: 256 slots
    // 86-96 us per layer, 512: 99-131, 1024: 126-154 -- every further split writes and re-reads one more [cout][K'] slab)
    static const int slots = getenv("AGAN_WG_ROWS_SLOTS") ? atoi(getenv("AGAN_WG_ROWS_SLOTS")) : 256;
    if (wgs < slots) ps = std::max(1, std::min(slots / wgs, p.mtiles));
    p.tiles_per_split = cdiv(p.mtiles, ps);
    p.psplit = cdiv(p.mtiles, p.tiles_per_split);
    p.NPH = IS == 2 ? 4 : 1;
    p.NT = IS == 2 ? 4 : 9;
    p.Kp = p.nchunks * p.NPH * p.NT * 32;
    p.slab = ((size_t)g.Cout * p.Kp + 3) / 4 * 4;
    p.ws_bytes = p.slab * (p.psplit + (p.psplit > 1 ? 1 : 0)) * sizeof(float);      // partial slabs + the reduced one
    p.dXRT = make_fastdiv((unsigned)p.XRT);
    p.dXR = make_fastdiv((unsigned)p.XR);
    p.dNBY = make_fastdiv((unsigned)p.NBY);
    p.ok = 1;
    return p;
}

void launch_rows_wgrad(const void* x, const void* dy, float* part, const Geom& gf, const RowsPlan& p, int prec, hipStream_t st,
                       const float* x_scale, const float* dy_scale, bool x16, bool y16) {
    const Geom g = rows_geom(gf, p.ups != 0);
    switch (prec) {
        case AGAN_PREC_F32: launch_rows_gk<2, 1, false, false>(x, dy, part, g, p, st, nullptr, nullptr); break;
        case AGAN_PREC_BF16: launch_rows_dt<0>(x, dy, part, g, p, st, x16, y16); break;
        case AGAN_PREC_F16: launch_rows_dt<1>(x, dy, part, g, p, st, x16, y16); break;
        case AGAN_PREC_BF16X3: launch_rows_gk<0, 2, false, false>(x, dy, part, g, p, st, nullptr, nullptr); break;
        default: launch_rows_gk<1, 2, false, false>(x, dy, part, g, p, st, x_scale, dy_scale); break;      // AGAN_PREC_F16X3
    }
}

}  // namespace conv
}  // namespace agan
