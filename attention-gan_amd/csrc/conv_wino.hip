// Winograd F(2x2, 3x3) for the fp32 3x3 stride-1 convolutions (round 3): 16 multiplications per 2x2 outputs and channel pair instead of 36.
//
//     Y = A^T [ (G g G^T) (.) (B^T d B) ] A            d: 4x4 input patch, g: 3x3 kernel, Y: 2x2 outputs   (Lavin & Gray, correlation form)
//
// The fp32 conv kernels of this path sit at the sustained rate of the fp32 matrix pipe: the only way to make the
// generator's 3x3 layers faster in exact fp32 arithmetic is to issue fewer MFMAs.  One workgroup owns 32 Winograd tiles (2x2 outputs each:
// a 4 x 8 block of tiles, or smaller blocks of several images) x 128 (64) output channels and walks the input channels 8 at a time:
//   * the raw 8-channel patch (tiles + 1-pixel halo) goes through LDS once; thread (tile, channel) transforms its 4x4 patch (32 adds) and
//     writes the 16 transformed values V[position][tile][channel] back to LDS;
//   * wave w multiplies the four positions (xi = 0..3, nu = w): D[cout][tile] += U[pos][cout][ci] V[pos][tile][ci] on
//     v_mfma_f32_32x32x2_f32, four MFMAs per 16-byte fragment pair (a lane's 4 floats are k-steps j = 0..3, channel 4 * lh + j) --
//     the transformed weights U come from a small transform pass over the mode's ordinary packed weights ([K][Nld]: forward and
//     data-gradient packs alike, the latter already hold the flipped taps), laid out [pos][channel octet][cout][8];
//   * the output transform runs along xi in registers (a lane holds all four xi of its tile and cout), along nu through LDS (the four
//     waves), and leaves 2 x 2 pixels per lane as two 8-byte stores.
// 16 accumulator fragments per wave (256 registers): one wave per SIMD.  No bias / activation epilogue: the layers that come here are
// followed by BatchNorm (utilities/layers.py:45-53); calls with one fall back to the direct kernel.
#include "conv_common.h"

#include <cstdlib>
#include <cstring>

using namespace agan;
using namespace agan::conv;

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void lds_barrier_w() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// U[pos][ch8][n][8] = (G g G^T)[pos] of g = packed[k = (c, r, s)][n]
__global__ __launch_bounds__(256) void wino_weight_transform_kernel(const float* __restrict__ wk, float* __restrict__ U, int Kin, int Nld, int N) {
    const int total = Kin * Nld;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
        const int n = e % Nld, c = e / Nld;
        float gk[3][3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int s = 0; s < 3; ++s) gk[r][s] = n < N ? wk[(size_t)(c * 9 + r * 3 + s) * Nld + n] : 0.f;
        // t = G g  (4 x 3),  u = t G^T (4 x 4);  G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
        float t[4][3];
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            t[0][s] = gk[0][s];
            t[1][s] = 0.5f * (gk[0][s] + gk[1][s] + gk[2][s]);
            t[2][s] = 0.5f * (gk[0][s] - gk[1][s] + gk[2][s]);
            t[3][s] = gk[2][s];
        }
        const int nch8 = Kin >> 3;
#pragma unroll
        for (int xi = 0; xi < 4; ++xi) {
            const float u0 = t[xi][0], u1 = 0.5f * (t[xi][0] + t[xi][1] + t[xi][2]), u2 = 0.5f * (t[xi][0] - t[xi][1] + t[xi][2]), u3 = t[xi][2];
            const float u[4] = {u0, u1, u2, u3};
#pragma unroll
            for (int nu = 0; nu < 4; ++nu) U[((size_t)((xi * 4 + nu) * nch8 + (c >> 3)) * Nld + n) * 8 + (c & 7)] = u[nu];
        }
    }
}

// NFW = cout fragments per workgroup: 4 (128 output channels x 32 tiles) or 2 (64 output channels x 64 tiles).  Wave w owns cout fragment
// w % NFW of tile group w / NFW and ALL 16 transform positions of it (16 accumulator fragments = 256 registers, one wave per SIMD): a lane
// then holds the whole 4 x 4 product of its (tile, output channel) and the output transform never leaves its registers.  (The first
// version gave every wave one nu of all cout fragments: the nu-direction of the output transform went through 64 KB of LDS and four
// barriers per workgroup -- with one workgroup per CU that epilogue and the prologue were a third of its time.)
template <int NFW>
__global__ __launch_bounds__(256, 1) void conv_wino_f32_kernel(const float* __restrict__ x, const float* __restrict__ U, float* __restrict__ out,
                                                               const Geom g, const WinoPlan wp) {
    constexpr int NTG = 4 / NFW;                                   // 32-tile groups per workgroup
    constexpr int NTL = 32 * NTG;                                  // tiles per workgroup
    constexpr int LT = NTG == 1 ? 5 : 6;                           // log2 of that
    constexpr int NI = NTG == 1 ? 7 : 11;                          // raw staging values per thread and chunk
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fw = w % NFW, tg = w / NFW;
    const int l31 = lane & 31, lh = lane >> 5;
    int mt, nt;
    {
        const int F = xcd_contiguous(linear_block_id(), wp.mtiles * wp.ntiles);
        nt = F % wp.ntiles; mt = F / wp.ntiles;
    }
    const int n0 = nt * (NFW * 32);
    const int txl = wp.txl, tyl = wp.tyl;                           // log2 of the tile block's columns / rows; images = NTL >> (txl + tyl)
    const int TX = 1 << txl, TY = 1 << tyl;
    const int bxi = mt % wp.blocks_x, byi = (mt / wp.blocks_x) % wp.blocks_y, bbi = mt / (wp.blocks_x * wp.blocks_y);
    const int tx0 = bxi << txl, ty0 = byi << tyl, tb0 = bbi << (LT - txl - tyl);
    const int ihw = g.IH * g.IW, ohw = g.OH * g.OW;
    const int PR = 2 * TY + 2, PC = 2 * TX + 2;                     // raw patch rows / columns per image
    const int PCP = wp.pcp;                                        // row pitch (floats)
    float* const raw = reinterpret_cast<float*>(lds);                               // [2 buffers][tb][8 ci][PR][PCP]
    const int RAWBUF = wp.raw_bytes >> 2;                                           // floats per raw buffer
    float* const V = reinterpret_cast<float*>(lds + 2 * wp.raw_bytes);              // [2 buffers][16 pos][NTL tiles][8 ci]
    constexpr int VBUF = 16 * NTL * 8;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(x, (size_t)g.B * g.Cin * ihw * sizeof(float));
    const __amdgpu_buffer_rsrc_t rU = make_rsrc(U, (size_t)16 * (g.Cin >> 3) * g.Nld * 8 * sizeof(float));
    const int nch = g.Cin >> 3;

    // ---- raw staging items: (image, channel, row, column), column fastest ----
    unsigned it_lds[NI];
    int it_off[NI];                       // element offset of (b, ci, row, col) relative to channel chunk 0, or -1
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int e = tid + i * 256;
        const int col = e % PC, t1 = e / PC;
        const int row = t1 % PR, t2 = t1 / PR;
        const int ci = t2 & 7, tb = t2 >> 3;
        const int b = tb0 + tb, iy = 2 * ty0 - 1 + row, ix = 2 * tx0 - 1 + col;
        const bool ok = (e < wp.raw_items) & (b < g.B) & ((unsigned)iy < (unsigned)g.IH) & ((unsigned)ix < (unsigned)g.IW);
        it_off[i] = ok ? (b * g.Cin + ci) * ihw + iy * g.IW + ix : -1;
        it_lds[i] = (unsigned)(((tb * 8 + ci) * PR + row) * PCP + col);
    }
    float rawv[NI];
    auto load_raw = [&](int ch, float (&rv)[NI]) {
#pragma unroll
        for (int i = 0; i < NI; ++i)
            rv[i] = buf_load_s(rx, it_off[i] >= 0 ? (unsigned)it_off[i] * 4u : kOOB, (unsigned)(ch * 8 * ihw) * 4u);
    };
    auto store_raw = [&](int ch, const float (&rv)[NI]) {      // chunk ch's patch -> raw buffer ch & 1
        float* const rb = raw + (ch & 1) * RAWBUF;
#pragma unroll
        for (int i = 0; i < NI; ++i)
            if (tid + i * 256 < wp.raw_items) rb[it_lds[i]] = rv[i];
    };
    // ---- input transform: thread (tile tt = (tid >> 3) + 32 k, channel ci = tid & 7) ----
    const int tci = tid & 7;
    auto transform = [&](float* Vb) {                          // (the prologue's: chunk 0, raw buffer 0)
#pragma unroll
        for (int k = 0; k < NTG; ++k) {
            const int tt = (tid >> 3) + 32 * k;
            const int ttx = tt & (TX - 1), tty = (tt >> txl) & (TY - 1), ttb = tt >> (txl + tyl);
            const float* const tsrc = raw + ((ttb * 8 + tci) * PR + 2 * tty) * PCP + 2 * ttx;
            float d[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float2 a = *reinterpret_cast<const float2*>(tsrc + i * PCP), b = *reinterpret_cast<const float2*>(tsrc + i * PCP + 2);
                d[i][0] = a.x; d[i][1] = a.y; d[i][2] = b.x; d[i][3] = b.y;
            }
            // t = B^T d: rows (d0 - d2, d1 + d2, d2 - d1, d1 - d3); v = t B: columns likewise
            float t[4][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                t[0][j] = d[0][j] - d[2][j];
                t[1][j] = d[1][j] + d[2][j];
                t[2][j] = d[2][j] - d[1][j];
                t[3][j] = d[1][j] - d[3][j];
            }
#pragma unroll
            for (int xi = 0; xi < 4; ++xi) {
                const float v0 = t[xi][0] - t[xi][2], v1 = t[xi][1] + t[xi][2], v2 = t[xi][2] - t[xi][1], v3 = t[xi][1] - t[xi][3];
                Vb[((xi * 4 + 0) * NTL + tt) * 8 + tci] = v0;
                Vb[((xi * 4 + 1) * NTL + tt) * 8 + tci] = v1;
                Vb[((xi * 4 + 2) * NTL + tt) * 8 + tci] = v2;
                Vb[((xi * 4 + 3) * NTL + tt) * 8 + tci] = v3;
            }
        }
    };
    // ---- transformed weights: this wave's cout fragment, 16 bytes (channels 4 * lh ..) per lane and position; half a chunk's positions
    //      (8) at a time, so that two register sets of 8 fragments suffice ----
    const unsigned ulane = (unsigned)(min(n0 + fw * 32 + l31, g.Nld - 1) * 32 + lh * 16);      // (rows past Cout are zero in U)
    auto load_u = [&](int ch, int half, u32x4 (&uf)[8]) {
#pragma unroll
        for (int q = 0; q < 8; ++q)
            uf[q] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                rU, ulane, (unsigned)(((half * 8 + q) * nch + min(ch, nch - 1)) * g.Nld) * 32u, 0));
    };

    f32x16 acc[16];
#pragma unroll
    for (int q = 0; q < 16; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;

    u32x4 ua[8], ub[8];                  // weights of positions 0..7 / 8..15: each set is reloaded while the other one's products run
    // transform of the next chunk, in 4 * NTG pieces (one output row xi of one tile group each) that ride between the MFMA groups
    float td[NTG][4][4];
    auto transform_piece = [&](const float* rb, float* Vb, int pi) {
        const int k = pi >> 2, xi = pi & 3;
        const int tt = (tid >> 3) + 32 * k;
        if (xi == 0) {
            const int ttx = tt & (TX - 1), tty = (tt >> txl) & (TY - 1), ttb = tt >> (txl + tyl);
            const float* const tsrc = rb + ((ttb * 8 + tci) * PR + 2 * tty) * PCP + 2 * ttx;
            float d[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float2 a = *reinterpret_cast<const float2*>(tsrc + i * PCP), b = *reinterpret_cast<const float2*>(tsrc + i * PCP + 2);
                d[i][0] = a.x; d[i][1] = a.y; d[i][2] = b.x; d[i][3] = b.y;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {                       // t = B^T d
                td[k][0][j] = d[0][j] - d[2][j];
                td[k][1][j] = d[1][j] + d[2][j];
                td[k][2][j] = d[2][j] - d[1][j];
                td[k][3][j] = d[1][j] - d[3][j];
            }
        }
        const float v0 = td[k][xi][0] - td[k][xi][2], v1 = td[k][xi][1] + td[k][xi][2], v2 = td[k][xi][2] - td[k][xi][1], v3 = td[k][xi][1] - td[k][xi][3];
        Vb[((xi * 4 + 0) * NTL + tt) * 8 + tci] = v0;
        Vb[((xi * 4 + 1) * NTL + tt) * 8 + tci] = v1;
        Vb[((xi * 4 + 2) * NTL + tt) * 8 + tci] = v2;
        Vb[((xi * 4 + 3) * NTL + tt) * 8 + tci] = v3;
    };
    {   // prologue: the patches of chunks 0 and 1 in flight together, chunk 0 transformed
        float rawv1[NI];
        load_raw(0, rawv);
        if (nch > 1) load_raw(1, rawv1);
        load_u(0, 0, ua);
        store_raw(0, rawv);
        if (nch > 1) store_raw(1, rawv1);
        lds_barrier_w();
        transform(V);
        lds_barrier_w();
    }
    for (int ch = 0; ch < nch; ++ch) {
        const float* Vc = V + (ch & 1) * VBUF;
        float* Vn = V + ((ch + 1) & 1) * VBUF;
        const float* rn = raw + ((ch + 1) & 1) * RAWBUF;      // the raw patch of chunk ch + 1: in LDS since the last barrier
        const bool more = ch + 1 < nch;
        // 16 stages, one per transform position: [V fragment of the NEXT stage] [a piece of the other work] [this stage's 4 MFMAs = 256
        // cycles of the matrix pipe].  The wave issues in order, so whatever must hide behind the MFMAs has to stand between them in
        // program order; sched_barrier pins the stages.  Other work of chunk ch: the input transform of chunk ch + 1 (stages 1 .. 4 NTG),
        // the global loads of chunk ch + 2's raw patch (stage 1) and its LDS stores (stage 14, into the buffer chunk ch's patch has left),
        // the weight fragments (stages 0 and 8).  ONE barrier per chunk.
        const unsigned vlane = (unsigned)((tg * 32 + l31) * 8 + lh * 4);
        u32x4 vf[2];
        vf[0] = *reinterpret_cast<const u32x4*>(Vc + vlane);
#pragma unroll
        for (int pos = 0; pos < 16; ++pos) {
            if (pos + 1 < 16) vf[(pos + 1) & 1] = *reinterpret_cast<const u32x4*>(Vc + (pos + 1) * NTL * 8 + vlane);
            if (pos == 0) load_u(ch, 1, ub);                  // second half of this chunk's weights
            if (pos == 1 && ch + 2 < nch) load_raw(ch + 2, rawv);
            if (pos == 8) load_u(ch + 1, 0, ua);              // first half of the next chunk's (past the end: a harmless repeat)
            if (more && pos >= 1 && pos <= 4 * NTG) transform_piece(rn, Vn, pos - 1);
            if (pos == 14 && ch + 2 < nch) store_raw(ch + 2, rawv);
            const u32x4 uf = pos < 8 ? ua[pos & 7] : ub[pos & 7];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[pos] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(uf[j]), __uint_as_float(vf[pos & 1][j]), acc[pos], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        lds_barrier_w();                                      // V of chunk ch + 1 and the raw patch of chunk ch + 2 are complete; every wave is done with chunk ch
    }

    // ---- output transform, in registers: Y = A^T M A with M[xi][nu] = acc[xi * 4 + nu], A^T = [[1, 1, 1, 0], [0, 1, -1, -1]] ----
    const int lt = tg * 32 + l31;
    const int tx = lt & (TX - 1), ty = (lt >> txl) & (TY - 1), tb = lt >> (txl + tyl);
    const int b = tb0 + tb, oy = 2 * (ty0 + ty), ox = 2 * (tx0 + tx);
    const bool pvalid = (b < g.B) & (oy < g.OH) & (ox < g.OW);
    const __amdgpu_buffer_rsrc_t ro = make_rsrc(out, (size_t)g.B * g.Cout * ohw * sizeof(float));
    const int nw = n0 + fw * 32;
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float c0[4], c1[4];
#pragma unroll
        for (int nu = 0; nu < 4; ++nu) {
            c0[nu] = acc[0 + nu][r] + acc[4 + nu][r] + acc[8 + nu][r];
            c1[nu] = acc[4 + nu][r] - acc[8 + nu][r] - acc[12 + nu][r];
        }
        const float y00 = c0[0] + c0[1] + c0[2], y01 = c0[1] - c0[2] - c0[3];
        const float y10 = c1[0] + c1[1] + c1[2], y11 = c1[1] - c1[2] - c1[3];
        const int n = nw + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const bool ok = pvalid & (n < g.Cout);
        const unsigned off = (unsigned)((b * g.Cout + n) * ohw + oy * g.OW + ox) * 4u;
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_, f32x2_{y00, y01}), ro, ok ? off : kOOB, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_, f32x2_{y10, y11}), ro, ok ? off + (unsigned)g.OW * 4u : kOOB, 0, 0);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The same F(2x2, 3x3) gather with TWO workgroups per CU: 32 tiles x 64 output channels per workgroup, a wave = one cout fragment x HALF
// the positions (rows xi = 2 h, 2 h + 1 of the 4 x 4 product: 128 accumulator registers).  One workgroup per CU left its fixed cost --
// first loads, last stores, every barrier -- uncovered (~5 us of 26); two co-resident workgroups cover each other's.  The price: every
// (tile, channel) patch is transformed by both cout halves, and the output transform's row direction crosses the two waves of a cout
// fragment: each wave reduces its two rows to the partial sums of the two output rows and hands ONE of them to its partner through LDS
// (32 floats per lane), then finishes the other.
__global__ __launch_bounds__(256, 2) void conv_wino_h_f32_kernel(const float* __restrict__ x, const float* __restrict__ U, float* __restrict__ out,
                                                                 const Geom g, const WinoPlan wp) {
    constexpr int NTL = 32, NI = 7;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fw = w & 1, hh = w >> 1;
    const int l31 = lane & 31, lh = lane >> 5;
    int mt, nt;
    {
        const int F = xcd_contiguous(linear_block_id(), wp.mtiles * wp.ntiles);
        nt = F % wp.ntiles; mt = F / wp.ntiles;
    }
    const int n0 = nt * 64;
    const int txl = wp.txl, tyl = wp.tyl;
    const int TX = 1 << txl, TY = 1 << tyl;
    const int bxi = mt % wp.blocks_x, byi = (mt / wp.blocks_x) % wp.blocks_y, bbi = mt / (wp.blocks_x * wp.blocks_y);
    const int tx0 = bxi << txl, ty0 = byi << tyl, tb0 = bbi << (5 - txl - tyl);
    const int ihw = g.IH * g.IW, ohw = g.OH * g.OW;
    const int PR = 2 * TY + 2, PC = 2 * TX + 2;
    const int PCP = wp.pcp;
    float* const raw = reinterpret_cast<float*>(lds);
    const int RAWBUF = wp.raw_bytes >> 2;
    float* const V = reinterpret_cast<float*>(lds + 2 * wp.raw_bytes);
    constexpr int VBUF = 16 * NTL * 8;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(x, (size_t)g.B * g.Cin * ihw * sizeof(float));
    const __amdgpu_buffer_rsrc_t rU = make_rsrc(U, (size_t)16 * (g.Cin >> 3) * g.Nld * 8 * sizeof(float));
    const int nch = g.Cin >> 3;

    unsigned it_lds[NI];
    int it_off[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int e = tid + i * 256;                               // < 2048: e / d = (e * ceil(65536 / d)) >> 16 exactly for d <= 18 (two VALU
        const int t1 = (int)(__umul24(e, wp.mpc) >> 16), col = e - (int)__umul24(t1, PC);       // instructions instead of the ~12 of a general
        const int t2 = (int)(__umul24(t1, wp.mpr) >> 16), row = t1 - (int)__umul24(t2, PR);     // division; 24-bit multiplies are full rate)
        const int ci = t2 & 7, tb = t2 >> 3;
        const int b = tb0 + tb, iy = 2 * ty0 - 1 + row, ix = 2 * tx0 - 1 + col;
        const bool ok = (e < wp.raw_items) & (b < g.B) & ((unsigned)iy < (unsigned)g.IH) & ((unsigned)ix < (unsigned)g.IW);
        it_off[i] = ok ? (int)__umul24(b * g.Cin + ci, ihw) + iy * g.IW + ix : -1;
        it_lds[i] = (unsigned)((int)__umul24(__umul24(t2, PR) + row, PCP) + col);
    }
    float rawv[NI];
    auto load_raw = [&](int ch, float (&rv)[NI]) {
#pragma unroll
        for (int i = 0; i < NI; ++i)
            rv[i] = buf_load_s(rx, it_off[i] >= 0 ? (unsigned)it_off[i] * 4u : kOOB, (unsigned)(ch * 8 * ihw) * 4u);
    };
    auto store_raw = [&](int ch, const float (&rv)[NI]) {
        float* const rb = raw + (ch & 1) * RAWBUF;
#pragma unroll
        for (int i = 0; i < NI; ++i)
            if (tid + i * 256 < wp.raw_items) rb[it_lds[i]] = rv[i];
    };
    const int tci = tid & 7, tt = tid >> 3;
    const int ttx = tt & (TX - 1), tty = (tt >> txl) & (TY - 1), ttb = tt >> (txl + tyl);
    const int tsrc_off = ((ttb * 8 + tci) * PR + 2 * tty) * PCP + 2 * ttx;
    float td[4][4];
    auto transform_piece = [&](const float* rb, float* Vb, int xi) {
        if (xi == 0) {
            const float* const tsrc = rb + tsrc_off;
            float d[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float2 a = *reinterpret_cast<const float2*>(tsrc + i * PCP), b = *reinterpret_cast<const float2*>(tsrc + i * PCP + 2);
                d[i][0] = a.x; d[i][1] = a.y; d[i][2] = b.x; d[i][3] = b.y;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                td[0][j] = d[0][j] - d[2][j];
                td[1][j] = d[1][j] + d[2][j];
                td[2][j] = d[2][j] - d[1][j];
                td[3][j] = d[1][j] - d[3][j];
            }
        }
        Vb[((xi * 4 + 0) * NTL + tt) * 8 + tci] = td[xi][0] - td[xi][2];
        Vb[((xi * 4 + 1) * NTL + tt) * 8 + tci] = td[xi][1] + td[xi][2];
        Vb[((xi * 4 + 2) * NTL + tt) * 8 + tci] = td[xi][2] - td[xi][1];
        Vb[((xi * 4 + 3) * NTL + tt) * 8 + tci] = td[xi][1] - td[xi][3];
    };
    const unsigned ulane = (unsigned)(min(n0 + fw * 32 + l31, g.Nld - 1) * 32 + lh * 16);
    auto load_u = [&](int ch, int half, u32x4 (&uf)[4]) {            // this wave's positions hh * 8 + half * 4 ..
#pragma unroll
        for (int q = 0; q < 4; ++q)
            uf[q] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                rU, ulane, (unsigned)(((hh * 8 + half * 4 + q) * nch + min(ch, nch - 1)) * g.Nld) * 32u, 0));
    };

    f32x16 acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    u32x4 ua[4], ub[4];
    {
        float rawv1[NI];
        load_raw(0, rawv);
        if (nch > 1) load_raw(1, rawv1);
        load_u(0, 0, ua);
        store_raw(0, rawv);
        if (nch > 1) store_raw(1, rawv1);
        lds_barrier_w();
#pragma unroll
        for (int xi = 0; xi < 4; ++xi) transform_piece(raw, V, xi);
        lds_barrier_w();
    }
    for (int ch = 0; ch < nch; ++ch) {
        const float* Vc = V + (ch & 1) * VBUF;
        float* Vn = V + ((ch + 1) & 1) * VBUF;
        const float* rn = raw + ((ch + 1) & 1) * RAWBUF;
        const bool more = ch + 1 < nch;
        const unsigned vlane = (unsigned)(hh * 8 * NTL * 8 + l31 * 8 + lh * 4);
        u32x4 vf[2];
        vf[0] = *reinterpret_cast<const u32x4*>(Vc + vlane);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (q + 1 < 8) vf[(q + 1) & 1] = *reinterpret_cast<const u32x4*>(Vc + (q + 1) * NTL * 8 + vlane);
            if (q == 0) load_u(ch, 1, ub);
            if (q == 1 && ch + 2 < nch) load_raw(ch + 2, rawv);
            if (q == 4) load_u(ch + 1, 0, ua);
            if (more && q >= 1 && q <= 4) transform_piece(rn, Vn, q - 1);
            if (q == 6 && ch + 2 < nch) store_raw(ch + 2, rawv);
            const u32x4 uf = q < 4 ? ua[q & 3] : ub[q & 3];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(uf[j]), __uint_as_float(vf[q & 1][j]), acc[q], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        lds_barrier_w();
    }

    // ---- output transform: columns in registers, rows across the wave pair (fw, hh = 0 / 1).  acc[x2 * 4 + nu] = M[2 hh + x2][nu].
    //      hh = 0 holds rows 0, 1: output row 0 gets M0 + M1, row 1 gets M1;   hh = 1 holds rows 2, 3: row 0 gets M2, row 1 gets -M2 - M3.
    //      Wave hh finishes output row hh and sends its contribution to the other row to its partner.
    //      The fp32 MFMA and the vector ALU do not overlap on gfx950 (profiles/micro/mfma_issue.hip) and this kernel runs only 256 MFMAs per
    //      wave: every VALU instruction here is ~0.04 % of the workgroup.  So: packed adds on (r, r + 1) register pairs (consecutive
    //      accumulator registers: no moves), one code path per hh (no selects), the register-dependent part of the store address in
    //      the scalar offset. ----
    float* const X = reinterpret_cast<float*>(lds);            // [wave 4][32 values][64 lanes]
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
    const int tx = l31 & (TX - 1), ty = (l31 >> txl) & (TY - 1), tb = l31 >> (txl + tyl);
    const int b = tb0 + tb, oy = 2 * (ty0 + ty) + hh, ox = 2 * (tx0 + tx);
    const bool pvalid = (b < g.B) & (oy < g.OH) & (ox < g.OW);
    const __amdgpu_buffer_rsrc_t ro = make_rsrc(out, (size_t)g.B * g.Cout * ohw * sizeof(float));
    const int nw = n0 + fw * 32, pw = w ^ 2;                    // partner wave: same cout fragment, other position half
    const unsigned vo = pvalid ? (unsigned)((b * g.Cout + nw + 4 * lh) * ohw + oy * g.OW + ox) * 4u : kOOB;
    auto finish = [&](auto hc) {
        constexpr int HH = decltype(hc)::value;
        f32x2_ keep[8][2];                                       // [register pair][output column]; HH = 1: the NEGATIVE of its share
#pragma unroll
        for (int rp = 0; rp < 8; ++rp) {
            f32x2_ t[2][2];                                      // [x2][output column]
#pragma unroll
            for (int x2 = 0; x2 < 2; ++x2) {
                const f32x2_ m0 = {acc[x2 * 4 + 0][2 * rp], acc[x2 * 4 + 0][2 * rp + 1]};
                const f32x2_ m1 = {acc[x2 * 4 + 1][2 * rp], acc[x2 * 4 + 1][2 * rp + 1]};
                const f32x2_ m2 = {acc[x2 * 4 + 2][2 * rp], acc[x2 * 4 + 2][2 * rp + 1]};
                const f32x2_ m3 = {acc[x2 * 4 + 3][2 * rp], acc[x2 * 4 + 3][2 * rp + 1]};
                t[x2][0] = m0 + m1 + m2;
                t[x2][1] = m1 - m2 - m3;
            }
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                keep[rp][c] = t[0][c] + t[1][c];
                const f32x2_ send = HH == 0 ? t[1][c] : t[0][c];
                X[(w * 32 + (2 * rp) * 2 + c) * 64 + lane] = send[0];
                X[(w * 32 + (2 * rp + 1) * 2 + c) * 64 + lane] = send[1];
            }
        }
        lds_barrier_w();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            // channels nw + (r & 3) + 8 (r >> 2) + 4 lh: a group of 8 is valid or not as a whole (Cout is a multiple of 8: plan_wino)
            if (nw + 8 * (r >> 2) < g.Cout) {
                const float x0 = X[(pw * 32 + r * 2 + 0) * 64 + lane], x1 = X[(pw * 32 + r * 2 + 1) * 64 + lane];
                const float k0 = keep[r >> 1][0][r & 1], k1 = keep[r >> 1][1][r & 1];
                const f32x2_ y = HH == 0 ? f32x2_{x0 + k0, x1 + k1} : f32x2_{x0 - k0, x1 - k1};
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_, y), ro, vo, (unsigned)(((r & 3) + 8 * (r >> 2)) * ohw) * 4u, 0);
            }
        }
    };
    if (hh == 0) finish(std::integral_constant<int, 0>{});
    else finish(std::integral_constant<int, 1>{});
}

// ---------------------------------------------------------------------------------------------------------------------------------
// conv4x4 stride 2 pad 1 (the discriminators' down-sampling convs), forward: polyphase + Winograd F(2x2, 2x2).
//     out[oy][ox] = sum_{r,s} w[r][s] x[2 oy + r - 1][2 ox + s - 1]:  the odd input rows O[i] = x[2i + 1] see the taps r = 0, 2 (at O[oy - 1],
//     O[oy]), the even rows E[i] = x[2i] the taps r = 1, 3 (at E[oy], E[oy + 1]) -- per dimension two 2-tap stride-1 filters on the two
//     phase images, in 2-D four phases (p, q) of 2x2 taps.  Each is F(2x2, 2x2): 2x2 outputs from a 3x3 phase patch with 9 products
//     instead of 16:   B^T = [[1,-1,0],[0,1,0],[0,-1,1]],  G = [[1,0],[1,1],[0,1]],  A^T = [[1,1,0],[0,1,1]]  (all +-1: exact transforms
//     up to the adds).  The output transform is the same for the four phases, so their products are summed in the SAME nine
//     accumulators: 36 products per 2x2 outputs and input channel instead of 64.
// Structure as conv_wino_f32_kernel: 32 tiles x 128 output channels per workgroup, a wave = one cout fragment x all 9 positions (144
// accumulator registers), input channels 8 at a time = 36 stages (phase, position) of 4 MFMAs; the raw patch of a tile is 6 x 6 pixels
// (rows / columns 4 t .. 4 t + 5 of the block's patch), a phase takes every second row / column of it.
template <int NFW>
__global__ __launch_bounds__(256, 1) void conv_wino_s2_f32_kernel(const float* __restrict__ x, const float* __restrict__ U, float* __restrict__ out,
                                                                  const Geom g, const WinoPlan wp) {
    constexpr int NTG = 4 / NFW, NTL = 32 * NTG, LT = NTG == 1 ? 5 : 6;
    constexpr int NI = NTG == 1 ? 21 : 36;                         // raw staging values per thread and chunk
    constexpr int NP = 36;                                         // (phase, position) planes
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fw = w % NFW, tg = w / NFW;
    const int l31 = lane & 31, lh = lane >> 5;
    int mt, nt, ks;
    {
        int F = xcd_contiguous(linear_block_id(), wp.mtiles * wp.ntiles * wp.ksplit);
        nt = F % wp.ntiles; F /= wp.ntiles;
        mt = F % wp.mtiles; ks = F / wp.mtiles;                     // ks: which part of the input channels (partial sums to slab ks)
    }
    const int n0 = nt * (NFW * 32);
    const int txl = wp.txl, tyl = wp.tyl;
    const int TX = 1 << txl, TY = 1 << tyl;
    const int bxi = mt % wp.blocks_x, byi = (mt / wp.blocks_x) % wp.blocks_y, bbi = mt / (wp.blocks_x * wp.blocks_y);
    const int tx0 = bxi << txl, ty0 = byi << tyl, tb0 = bbi << (LT - txl - tyl);
    const int ihw = g.IH * g.IW, ohw = g.OH * g.OW;
    const int PR = 4 * TY + 2, PC = 4 * TX + 2;                     // raw patch rows / columns per image
    const int PCP = wp.pcp;
    float* const raw = reinterpret_cast<float*>(lds);                               // [2 buffers][tb][8 ci][PR][PCP]
    const int RAWBUF = wp.raw_bytes >> 2;
    float* const V = reinterpret_cast<float*>(lds + 2 * wp.raw_bytes);              // [2 buffers][36 planes][NTL tiles][8 ci]
    constexpr int VBUF = NP * NTL * 8;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(x, (size_t)g.B * g.Cin * ihw * sizeof(float));
    const __amdgpu_buffer_rsrc_t rU = make_rsrc(U, (size_t)NP * (g.Cin >> 3) * g.Nld * 8 * sizeof(float));
    const int nch_all = g.Cin >> 3;
    const int c_beg = ks * wp.chunks_per_split, nch = min(nch_all, c_beg + wp.chunks_per_split) - c_beg;      // this workgroup's chunks: c_beg .. c_beg + nch

    unsigned it_lds[NI];
    int it_off[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int e = tid + i * 256;
        const int col = e % PC, t1 = e / PC;
        const int row = t1 % PR, t2 = t1 / PR;
        const int ci = t2 & 7, tb = t2 >> 3;
        const int b = tb0 + tb, iy = 4 * ty0 - 1 + row, ix = 4 * tx0 - 1 + col;
        const bool ok = (e < wp.raw_items) & (b < g.B) & ((unsigned)iy < (unsigned)g.IH) & ((unsigned)ix < (unsigned)g.IW);
        it_off[i] = ok ? (b * g.Cin + ci) * ihw + iy * g.IW + ix : -1;
        it_lds[i] = (unsigned)(((tb * 8 + ci) * PR + row) * PCP + col);
    }
    float rawv[NI];
    auto load_raw = [&](int ch, float (&rv)[NI]) {
#pragma unroll
        for (int i = 0; i < NI; ++i)
            rv[i] = buf_load_s(rx, it_off[i] >= 0 ? (unsigned)it_off[i] * 4u : kOOB, (unsigned)((c_beg + ch) * 8 * ihw) * 4u);
    };
    auto store_raw = [&](int ch, const float (&rv)[NI]) {
        float* const rb = raw + (ch & 1) * RAWBUF;
#pragma unroll
        for (int i = 0; i < NI; ++i)
            if (tid + i * 256 < wp.raw_items) rb[it_lds[i]] = rv[i];
    };
    // input transform, one (tile group k, phase) piece at a time: thread (tile (tid >> 3) + 32 k, channel tid & 7)
    const int tci = tid & 7;
    auto transform_piece = [&](const float* rb, float* Vb, int pi) {
        const int k = pi >> 2, ph = pi & 3, p = ph >> 1, q = ph & 1;
        const int tt = (tid >> 3) + 32 * k;
        const int ttx = tt & (TX - 1), tty = (tt >> txl) & (TY - 1), ttb = tt >> (txl + tyl);
        const float* const src = rb + ((ttb * 8 + tci) * PR + 4 * tty + p) * PCP + 4 * ttx + q;
        float t[3][3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float d0 = src[2 * j], d1 = src[2 * PCP + 2 * j], d2 = src[4 * PCP + 2 * j];
            t[0][j] = d0 - d1; t[1][j] = d1; t[2][j] = d2 - d1;
        }
#pragma unroll
        for (int xi = 0; xi < 3; ++xi) {
            float* o = Vb + ((ph * 9 + xi * 3) * NTL + tt) * 8 + tci;
            o[0] = t[xi][0] - t[xi][1]; o[NTL * 8] = t[xi][1]; o[2 * NTL * 8] = t[xi][2] - t[xi][1];
        }
    };
    const unsigned ulane = (unsigned)(min(n0 + fw * 32 + l31, g.Nld - 1) * 32 + lh * 16);
    auto load_u = [&](int ch, int half, u32x4 (&uf)[18]) {           // planes 18 * half ..
#pragma unroll
        for (int q = 0; q < 18; ++q)
            uf[q] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                rU, ulane, (unsigned)(((half * 18 + q) * nch_all + c_beg + min(ch, nch - 1)) * g.Nld) * 32u, 0));
    };

    f32x16 acc[9];
#pragma unroll
    for (int q = 0; q < 9; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    u32x4 ua[18], ub[18];
    {
        load_raw(0, rawv);
        load_u(0, 0, ua);
        store_raw(0, rawv);
        if (nch > 1) load_raw(1, rawv);
        lds_barrier_w();
#pragma unroll
        for (int pi = 0; pi < 4 * NTG; ++pi) transform_piece(raw, V, pi);
        if (nch > 1) store_raw(1, rawv);
        lds_barrier_w();
    }
    for (int ch = 0; ch < nch; ++ch) {
        const float* Vc = V + (ch & 1) * VBUF;
        float* Vn = V + ((ch + 1) & 1) * VBUF;
        const float* rn = raw + ((ch + 1) & 1) * RAWBUF;
        const bool more = ch + 1 < nch;
        const unsigned vlane = (unsigned)((tg * 32 + l31) * 8 + lh * 4);
        u32x4 vf[2];
        vf[0] = *reinterpret_cast<const u32x4*>(Vc + vlane);
#pragma unroll
        for (int st = 0; st < NP; ++st) {
            if (st + 1 < NP) vf[(st + 1) & 1] = *reinterpret_cast<const u32x4*>(Vc + (st + 1) * NTL * 8 + vlane);
            if (st == 0) load_u(ch, 1, ub);
            if (st == 1 && ch + 2 < nch) load_raw(ch + 2, rawv);
            if (st == 18) load_u(ch + 1, 0, ua);
            if (more && st >= 2 && st < 2 + 4 * NTG) transform_piece(rn, Vn, st - 2);
            if (st == 32 && ch + 2 < nch) store_raw(ch + 2, rawv);
            const u32x4 uf = st < 18 ? ua[st % 18] : ub[st % 18];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[st % 9] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(uf[j]), __uint_as_float(vf[st & 1][j]), acc[st % 9], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        lds_barrier_w();
    }

    // ---- output transform in registers: Y = A^T M A, M[xi][nu] = acc[xi * 3 + nu], A^T = [[1, 1, 0], [0, 1, 1]] ----
    const int lt = tg * 32 + l31;
    const int tx = lt & (TX - 1), ty = (lt >> txl) & (TY - 1), tb = lt >> (txl + tyl);
    const int b = tb0 + tb, oy = 2 * (ty0 + ty), ox = 2 * (tx0 + tx);
    const bool pvalid = (b < g.B) & (oy < g.OH) & (ox < g.OW);
    const __amdgpu_buffer_rsrc_t ro = make_rsrc(out + (size_t)ks * wp.slab, (size_t)g.B * g.Cout * ohw * sizeof(float));
    const int nw = n0 + fw * 32;
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float c0[3], c1[3];
#pragma unroll
        for (int nu = 0; nu < 3; ++nu) {
            c0[nu] = acc[nu][r] + acc[3 + nu][r];
            c1[nu] = acc[3 + nu][r] + acc[6 + nu][r];
        }
        const int n = nw + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const bool ok = pvalid & (n < g.Cout);
        const unsigned off = (unsigned)((b * g.Cout + n) * ohw + oy * g.OW + ox) * 4u;
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_, f32x2_{c0[0] + c0[1], c0[1] + c0[2]}), ro, ok ? off : kOOB, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_, f32x2_{c1[0] + c1[1], c1[1] + c1[2]}), ro, ok ? off + (unsigned)g.OW * 4u : kOOB, 0, 0);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Data gradient of conv4x4 stride 2: four parity classes of the output, each a 2x2-tap stride-1 gather over dy with taps running backwards
// (include/agan.h: DY = -1, offsets (0, 1)).  Per class Winograd F(2x2, 2x2) on the class lattice: 9 products per 2x2 lattice points instead of
// 16.  out[2 yq + py][2 xq + px] = sum_{r,s} Wc[r][s] dy[yq + py - r][xq + px - s] = correlation of the ascending window
// (dy[yq + py - 1], dy[yq + py]) with g = (Wc[1], Wc[0]).  Same structure as the kernels above; a channel chunk is 16 channels = 18 stages
// (position, 8-channel half) of 4 MFMAs; the four classes are grid entries; the 2x2 outputs of a tile lie two pixels apart (scalar stores).
template <int NFW>
__global__ __launch_bounds__(256, 1) void conv_wino_cls_f32_kernel(const float* __restrict__ x, const float* __restrict__ U, float* __restrict__ out,
                                                                   const Geom g, const WinoPlan wp) {
    constexpr int NTG = 4 / NFW, NTL = 32 * NTG, LT = NTG == 1 ? 5 : 6;
    constexpr int NI = NTG == 1 ? 11 : 19;                         // raw staging values per thread and chunk (16 channels)
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fw = w % NFW, tg = w / NFW;
    const int l31 = lane & 31, lh = lane >> 5;
    int mt, nt, cls;
    {
        int F = xcd_contiguous(linear_block_id(), wp.mtiles * wp.ntiles * 4);
        nt = F % wp.ntiles; F /= wp.ntiles;
        cls = F & 3; mt = F >> 2;                                   // (the four classes of a block read the same dy patch: neighbours)
    }
    const int py = cls >> 1, px = cls & 1;
    const int n0 = nt * (NFW * 32);
    const int txl = wp.txl, tyl = wp.tyl;
    const int TX = 1 << txl, TY = 1 << tyl;
    const int bxi = mt % wp.blocks_x, byi = (mt / wp.blocks_x) % wp.blocks_y, bbi = mt / (wp.blocks_x * wp.blocks_y);
    const int tx0 = bxi << txl, ty0 = byi << tyl, tb0 = bbi << (LT - txl - tyl);
    const int ihw = g.IH * g.IW, ohw = g.OH * g.OW;
    const int PR = 2 * TY + 1, PC = 2 * TX + 1;
    const int PCP = wp.pcp;
    float* const raw = reinterpret_cast<float*>(lds);                               // [2 buffers][tb][16 ci][PR][PCP]
    const int RAWBUF = wp.raw_bytes >> 2;
    float* const V = reinterpret_cast<float*>(lds + 2 * wp.raw_bytes);              // [2 buffers][9 positions][NTL tiles][16 ci]
    constexpr int VBUF = 9 * NTL * 16;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(x, (size_t)g.B * g.Cin * ihw * sizeof(float));
    const int nch8 = g.Cin >> 3, nch = g.Cin >> 4;
    const __amdgpu_buffer_rsrc_t rU = make_rsrc(U + (size_t)cls * 9 * nch8 * g.Nld * 8, (size_t)9 * nch8 * g.Nld * 8 * sizeof(float));

    unsigned it_lds[NI];
    int it_off[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int e = tid + i * 256;
        const int col = e % PC, t1 = e / PC;
        const int row = t1 % PR, t2 = t1 / PR;
        const int ci = t2 & 15, tb = t2 >> 4;
        const int b = tb0 + tb, iy = 2 * ty0 + py - 1 + row, ix = 2 * tx0 + px - 1 + col;
        const bool ok = (e < wp.raw_items) & (b < g.B) & ((unsigned)iy < (unsigned)g.IH) & ((unsigned)ix < (unsigned)g.IW);
        it_off[i] = ok ? (b * g.Cin + ci) * ihw + iy * g.IW + ix : -1;
        it_lds[i] = (unsigned)(((tb * 16 + ci) * PR + row) * PCP + col);
    }
    float rawv[NI];
    auto load_raw = [&](int ch, float (&rv)[NI]) {
#pragma unroll
        for (int i = 0; i < NI; ++i)
            rv[i] = buf_load_s(rx, it_off[i] >= 0 ? (unsigned)it_off[i] * 4u : kOOB, (unsigned)(ch * 16 * ihw) * 4u);
    };
    auto store_raw = [&](int ch, const float (&rv)[NI]) {
        float* const rb = raw + (ch & 1) * RAWBUF;
#pragma unroll
        for (int i = 0; i < NI; ++i)
            if (tid + i * 256 < wp.raw_items) rb[it_lds[i]] = rv[i];
    };
    // input transform, one (tile, channel) pair per piece: pair pi of this thread = (tile (tid >> 3) + 32 (pi >> 1), channel (tid & 7) + 8 (pi & 1))
    auto transform_piece = [&](const float* rb, float* Vb, int pi) {
        const int tt = (tid >> 3) + 32 * (pi >> 1), ci = (tid & 7) + 8 * (pi & 1);
        const int ttx = tt & (TX - 1), tty = (tt >> txl) & (TY - 1), ttb = tt >> (txl + tyl);
        const float* const src = rb + ((ttb * 16 + ci) * PR + 2 * tty) * PCP + 2 * ttx;
        float t[3][3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float d0 = src[j], d1 = src[PCP + j], d2 = src[2 * PCP + j];
            t[0][j] = d0 - d1; t[1][j] = d1; t[2][j] = d2 - d1;
        }
#pragma unroll
        for (int xi = 0; xi < 3; ++xi) {
            float* o = Vb + ((xi * 3) * NTL + tt) * 16 + ci;
            o[0] = t[xi][0] - t[xi][1]; o[NTL * 16] = t[xi][1]; o[2 * NTL * 16] = t[xi][2] - t[xi][1];
        }
    };
    const unsigned ulane = (unsigned)(min(n0 + fw * 32 + l31, g.Nld - 1) * 32 + lh * 16);
    auto load_u = [&](int ch, int half, u32x4 (&uf)[9]) {           // stages 9 * half ..: stage = position * 2 + 8-channel half
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            const int st = half * 9 + q;
            uf[q] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                rU, ulane, (unsigned)(((st >> 1) * nch8 + 2 * min(ch, nch - 1) + (st & 1)) * g.Nld) * 32u, 0));
        }
    };

    f32x16 acc[9];
#pragma unroll
    for (int q = 0; q < 9; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    u32x4 ua[9], ub[9];
    {
        load_raw(0, rawv);
        load_u(0, 0, ua);
        store_raw(0, rawv);
        if (nch > 1) load_raw(1, rawv);
        lds_barrier_w();
#pragma unroll
        for (int pi = 0; pi < 2 * NTG; ++pi) transform_piece(raw, V, pi);
        if (nch > 1) store_raw(1, rawv);
        lds_barrier_w();
    }
    for (int ch = 0; ch < nch; ++ch) {
        const float* Vc = V + (ch & 1) * VBUF;
        float* Vn = V + ((ch + 1) & 1) * VBUF;
        const float* rn = raw + ((ch + 1) & 1) * RAWBUF;
        const bool more = ch + 1 < nch;
        const unsigned vlane = (unsigned)((tg * 32 + l31) * 16 + lh * 4);
        u32x4 vf[2];
        vf[0] = *reinterpret_cast<const u32x4*>(Vc + vlane);
#pragma unroll
        for (int st = 0; st < 18; ++st) {
            if (st + 1 < 18) vf[(st + 1) & 1] = *reinterpret_cast<const u32x4*>(Vc + ((st + 1) >> 1) * NTL * 16 + ((st + 1) & 1) * 8 + vlane);
            if (st == 0) load_u(ch, 1, ub);
            if (st == 1 && ch + 2 < nch) load_raw(ch + 2, rawv);
            if (st == 9) load_u(ch + 1, 0, ua);
            if (more && st >= 2 && st < 2 + 2 * NTG) transform_piece(rn, Vn, st - 2);
            if (st == 15 && ch + 2 < nch) store_raw(ch + 2, rawv);
            const u32x4 uf = st < 9 ? ua[st % 9] : ub[st % 9];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[st >> 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(uf[j]), __uint_as_float(vf[st & 1][j]), acc[st >> 1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        lds_barrier_w();
    }

    // ---- output transform in registers: Y = A^T M A, A^T = [[1, 1, 0], [0, 1, 1]]; lattice point (yq, xq) -> pixel (2 yq + py, 2 xq + px) ----
    const int lt = tg * 32 + l31;
    const int tx = lt & (TX - 1), ty = (lt >> txl) & (TY - 1), tb = lt >> (txl + tyl);
    const int b = tb0 + tb, yq = 2 * (ty0 + ty), xq = 2 * (tx0 + tx);
    const bool pvalid = (b < g.B) & (yq < g.OHs) & (xq < g.OWs);
    const __amdgpu_buffer_rsrc_t ro = make_rsrc(out, (size_t)g.B * g.Cout * ohw * sizeof(float));
    const int nw = n0 + fw * 32;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float c0[3], c1[3];
#pragma unroll
        for (int nu = 0; nu < 3; ++nu) {
            c0[nu] = acc[nu][r] + acc[3 + nu][r];
            c1[nu] = acc[3 + nu][r] + acc[6 + nu][r];
        }
        const int n = nw + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const bool ok = pvalid & (n < g.Cout);
        const unsigned off = (unsigned)((b * g.Cout + n) * ohw + (2 * yq + py) * g.OW + 2 * xq + px) * 4u;
        buf_store(ro, ok ? off : kOOB, c0[0] + c0[1]);
        buf_store(ro, ok ? off + 8u : kOOB, c0[1] + c0[2]);
        buf_store(ro, ok ? off + (unsigned)(2 * g.OW) * 4u : kOOB, c1[0] + c1[1]);
        buf_store(ro, ok ? off + (unsigned)(2 * g.OW + 2) * 4u : kOOB, c1[1] + c1[2]);
    }
}

// U[cls][xi * 3 + nu][ch8][n][8] = (G g G^T)[xi][nu],  g[a][b] = packed_cls[(c, r = 1 - a, s = 1 - b)][n]   (taps run backwards: DY = -1)
__global__ __launch_bounds__(256) void wino_cls_weight_transform_kernel(const float* __restrict__ wk, float* __restrict__ U, int Kin, int Nld, int N) {
    const int total = 4 * Kin * Nld, nch8 = Kin >> 3;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
        const int n = e % Nld, t1 = e / Nld, c = t1 % Kin, cls = t1 / Kin;
        const float* wc = wk + (size_t)cls * Kin * 4 * Nld;
        float gk[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) gk[a][b] = n < N ? wc[(size_t)(c * 4 + (1 - a) * 2 + (1 - b)) * Nld + n] : 0.f;
        const float t[3][2] = {{gk[0][0], gk[0][1]}, {gk[0][0] + gk[1][0], gk[0][1] + gk[1][1]}, {gk[1][0], gk[1][1]}};
#pragma unroll
        for (int xi = 0; xi < 3; ++xi) {
            const float u[3] = {t[xi][0], t[xi][0] + t[xi][1], t[xi][1]};
#pragma unroll
            for (int nu = 0; nu < 3; ++nu)
                U[((size_t)((cls * 9 + xi * 3 + nu) * nch8 + (c >> 3)) * Nld + n) * 8 + (c & 7)] = u[nu];
        }
    }
}

// U[(p * 2 + q) * 9 + xi * 3 + nu][ch8][n][8] = (G g_pq G^T)[xi][nu],  g_pq[a][b] = packed[(c, r = 2a + p, s = 2b + q)][n]
__global__ __launch_bounds__(256) void wino_s2_weight_transform_kernel(const float* __restrict__ wk, float* __restrict__ U, int Kin, int Nld, int N) {
    const int total = Kin * Nld, nch8 = Kin >> 3;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
        const int n = e % Nld, c = e / Nld;
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                float gk[2][2];
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) gk[a][b] = n < N ? wk[(size_t)(c * 16 + (2 * a + p) * 4 + 2 * b + q) * Nld + n] : 0.f;
                const float t[3][2] = {{gk[0][0], gk[0][1]}, {gk[0][0] + gk[1][0], gk[0][1] + gk[1][1]}, {gk[1][0], gk[1][1]}};
#pragma unroll
                for (int xi = 0; xi < 3; ++xi) {
                    const float u[3] = {t[xi][0], t[xi][0] + t[xi][1], t[xi][1]};
#pragma unroll
                    for (int nu = 0; nu < 3; ++nu)
                        U[((size_t)(((p * 2 + q) * 9 + xi * 3 + nu) * nch8 + (c >> 3)) * Nld + n) * 8 + (c & 7)] = u[nu];
                }
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Weight gradient of the same layers, Winograd F(3x3, 2x2):  dw[r][s] = sum over 2x2 dy tiles e  of  corr(d, e)[r][s],  d = the tile's
// 4x4 x patch:   dw = A^T [ sum_tiles (G e G^T) (.) (B^T d B) ] A     A^T = [[1,1,1,0],[0,1,-1,0],[0,1,1,1]],
//                G = [[1,0],[.5,.5],[.5,-.5],[0,1]],  B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,-1,0,1]]      (16 products per tile instead of 36).
// The contraction index of the 16 products is the TILE: both operands are transformed activations, staged [position][channel][8 tiles]
// in LDS (a lane's 16 bytes = 4 k-steps of its channel), produced by the threads straight from global memory (a thread loads its tile's
// 4x4 x patch / 2x2 dy tile, transforms it and writes 16 values; no raw copy in LDS).  A workgroup owns 64 output x 64 input channels
// (wave = one 32 x 32 fragment pair, all 16 positions: 256 accumulator registers, output transform in registers) and a range of tile
// octets; per octet 64 MFMAs per wave, ONE barrier.  Partial results [cout][cin][3][3] per pixel split, summed by sum_slabs_kernel.
__global__ __launch_bounds__(256, 1) void conv_wino_wgrad_f32_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ part,
                                                                     const Geom g, const WinoWgradPlan wp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int ZV = 16 * 64 * 8;                               // floats of one operand image [pos][64 channels][8 tiles]
    float* const L = reinterpret_cast<float*>(lds);               // [2 buffers][Z | V]
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fj = w & 1, fi = w >> 1;                            // cout / cin fragment of this wave
    const int l31 = lane & 31, lh = lane >> 5;
    int jt, it, split;
    {
        int F = xcd_contiguous(linear_block_id(), wp.jtiles * wp.itiles * wp.psplit);
        jt = F % wp.jtiles; F /= wp.jtiles;
        it = F % wp.itiles; split = F / wp.itiles;
    }
    const int j0 = jt * 64, i0 = it * 64;
    const int ihw = g.IH * g.IW;
    const int o_beg = split * wp.octs_per_split, o_end = min(wp.noct, o_beg + wp.octs_per_split);
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(x, (size_t)g.B * g.Cin * ihw * sizeof(float));
    const __amdgpu_buffer_rsrc_t rd = make_rsrc(dy, (size_t)g.B * g.Cout * ihw * sizeof(float));
    const int tw8 = g.OW >> 4, th = g.OH >> 1;                    // tile octets per row, tile rows per image

    // this thread's two (tile, channel) pairs of each operand: tile = tid & 7, channel = (tid >> 3) + 32 k
    const int tj = tid & 7, tc = tid >> 3;
    float xr[2][16];
    float2 dr[2][2];
    // loads of octet o in 4 pieces (pair k = lp >> 1; x rows 2 * (lp & 1), +1; the dy tile with the first half)
    auto load_piece = [&](int o, int lp) {
        const bool ov = o < o_end;
        const int txo = o % tw8, t1 = o / tw8;
        const int ty = t1 % th, b = t1 / th;
        const int tx = txo * 8 + tj;
        const int k = lp >> 1, half = lp & 1;
        const int ci = i0 + tc + 32 * k, co = j0 + tc + 32 * k;
        const bool cv = ov & (ci < g.Cin);
        const int xbase = (b * g.Cin + ci) * ihw + 2 * tx - 1;
        const bool c0ok = tx > 0, c3ok = 2 * tx + 2 < g.IW;
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
            const int i = 2 * half + ii;
            const int iy = 2 * ty - 1 + i;
            const bool rv = cv & ((unsigned)iy < (unsigned)g.IH);
            const unsigned ro = (unsigned)(xbase + iy * g.IW) * 4u;
            xr[k][i * 4 + 0] = buf_load(rx, (rv & c0ok) ? ro : kOOB);
            xr[k][i * 4 + 1] = buf_load(rx, rv ? ro + 4u : kOOB);
            xr[k][i * 4 + 2] = buf_load(rx, rv ? ro + 8u : kOOB);
            xr[k][i * 4 + 3] = buf_load(rx, (rv & c3ok) ? ro + 12u : kOOB);
        }
        if (half == 0) {
            const bool dv = ov & (co < g.Cout);
            const int dbase = (b * g.Cout + co) * ihw + 2 * ty * g.OW + 2 * tx;
#pragma unroll
            for (int a = 0; a < 2; ++a)
                dr[k][a] = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rd, dv ? (unsigned)(dbase + a * g.OW) * 4u : kOOB, 0, 0));
        }
    };
    auto load_oct = [&](int o) {
#pragma unroll
        for (int lp = 0; lp < 4; ++lp) load_piece(o, lp);
    };
    // transform + LDS stores of the loaded octet in 8 pieces (pair k = tp >> 2; V rows 0-1, V rows 2-3, Z rows 0-1, Z rows 2-3)
    float tt_[2][4][4];                                          // t = B^T d of pair k, kept between its two V pieces
    auto transform_piece = [&](float* buf, int tp) {             // buf: Z at 0, V at ZV
        const int k = tp >> 2, sub = tp & 3;
        const int ch = tc + 32 * k;
        if (sub == 0) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float d0 = xr[k][c], d1 = xr[k][4 + c], d2 = xr[k][8 + c], d3 = xr[k][12 + c];
                tt_[k][0][c] = d0 - d2; tt_[k][1][c] = d1 + d2; tt_[k][2][c] = d2 - d1; tt_[k][3][c] = d3 - d1;
            }
        }
        if (sub < 2) {                                           // V = (B^T d) B, rows 2 * sub, + 1
#pragma unroll
            for (int x2 = 0; x2 < 2; ++x2) {
                const int xi = 2 * sub + x2;
                const float* t = tt_[k][xi];
                float* o = buf + ZV + ((xi * 4) * 64 + ch) * 8 + tj;
                o[0] = t[0] - t[2]; o[64 * 8] = t[1] + t[2]; o[2 * 64 * 8] = t[2] - t[1]; o[3 * 64 * 8] = t[3] - t[1];
            }
        } else {                                                 // Z = G e G^T, rows 2 * (sub - 2), + 1
            const float e00 = dr[k][0].x, e01 = dr[k][0].y, e10 = dr[k][1].x, e11 = dr[k][1].y;
#pragma unroll
            for (int x2 = 0; x2 < 2; ++x2) {
                const int xi = 2 * (sub - 2) + x2;
                const float u0 = xi == 0 ? e00 : (xi == 1 ? 0.5f * (e00 + e10) : (xi == 2 ? 0.5f * (e00 - e10) : e10));
                const float u1 = xi == 0 ? e01 : (xi == 1 ? 0.5f * (e01 + e11) : (xi == 2 ? 0.5f * (e01 - e11) : e11));
                float* o = buf + ((xi * 4) * 64 + ch) * 8 + tj;
                o[0] = u0; o[64 * 8] = 0.5f * (u0 + u1); o[2 * 64 * 8] = 0.5f * (u0 - u1); o[3 * 64 * 8] = u1;
            }
        }
    };
    auto transform_store = [&](float* buf) {
#pragma unroll
        for (int tp = 0; tp < 8; ++tp) transform_piece(buf, tp);
    };

    f32x16 acc[16];
#pragma unroll
    for (int q = 0; q < 16; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;

    if (o_beg < o_end) {
        load_oct(o_beg);
        transform_store(L);
        load_oct(o_beg + 1);
        lds_barrier_w();
        const unsigned zl = (unsigned)((fj * 32 + l31) * 8 + lh * 4), vl = (unsigned)(ZV + (fi * 32 + l31) * 8 + lh * 4);
        for (int o = o_beg; o < o_end; ++o) {
            const float* cur = L + ((o - o_beg) & 1) * (2 * ZV);
            float* nxt = L + (((o - o_beg) & 1) ^ 1) * (2 * ZV);
            // the next octet's operands (its patches were loaded one octet ago) are transformed and written, the octet after that is loaded,
            // all in the shadow of this octet's 64 MFMAs -- pinned stage by stage as in the gather kernel
            u32x4 zf[2], vf[2];
            zf[0] = *reinterpret_cast<const u32x4*>(cur + zl);
            vf[0] = *reinterpret_cast<const u32x4*>(cur + vl);
#pragma unroll
            for (int pos = 0; pos < 16; ++pos) {
                if (pos + 1 < 16) {
                    zf[(pos + 1) & 1] = *reinterpret_cast<const u32x4*>(cur + (pos + 1) * 64 * 8 + zl);
                    vf[(pos + 1) & 1] = *reinterpret_cast<const u32x4*>(cur + (pos + 1) * 64 * 8 + vl);
                }
                // no branch around the pieces (after the last octet they transform zeros into the buffer nobody reads): one basic block per
                // position, so that the pieces can be issued BETWEEN the four MFMAs of the position (below)
                if (pos >= 1 && pos <= 8) transform_piece(nxt, pos - 1);
                if (pos >= 9 && pos <= 12) load_piece(o + 2, pos - 9);          // (past the range: zero range, nothing is fetched)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[pos] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(zf[pos & 1][j]), __uint_as_float(vf[pos & 1][j]), acc[pos], 0, 0, 0);
                // a wave is alone on its SIMD and issues in order: whatever FOLLOWS four back-to-back MFMAs in program order waits for all
                // of them to issue, i.e. overlaps only the last one.  The scalar / LDS / global-load instructions of the piece are spread
                // over the four 64-cycle shadows instead (the VALU ones cost their ~6 cycles wherever they stand: on gfx950 the fp32 MFMA
                // and the vector ALU do not overlap -- profiles/micro/mfma_issue.hip).
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);          // MFMA
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);          // DS read (the next position's fragments)
                    __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);          // VALU
                    __builtin_amdgcn_sched_group_barrier(0x020, 3, 0);          // VMEM read
                    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);          // DS write
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            lds_barrier_w();
        }
    }

    // ---- dw = A^T M A in registers; D[cout][cin]: lane l31 = input channel, register -> output channel.  The slab is [tap][cout][cin]:
    //      a store instruction writes 128 contiguous bytes per half wave (OIHW would scatter its lanes 36 bytes apart: the 38 MB of slabs
    //      of a 128-way pixel split took 30 us that way); wino_wgrad_sum_kernel transposes while it sums.  Addresses: lane part in the
    //      vector offset once, (register, tap) part in the scalar offset. ----
    float* const o = part + (size_t)split * wp.slab;
    const __amdgpu_buffer_rsrc_t ro = make_rsrc(o, (size_t)9 * g.Cout * g.Cin * sizeof(float));
    const int ci = i0 + fi * 32 + l31;
    const int cob = j0 + fj * 32;                                // + (r & 3) + 8 (r >> 2) + 4 lh
    const unsigned vo = ci < g.Cin ? (unsigned)(4 * lh * g.Cin + ci) * 4u : kOOB;
    const unsigned tapb = (unsigned)(g.Cout * g.Cin) * 4u;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        if (cob + 8 * (r >> 2) >= g.Cout) continue;             // (a group of 8 channels is valid as a whole: Cout is a multiple of 8)
        float c[3][4];                                           // rows of A^T M
#pragma unroll
        for (int nu = 0; nu < 4; ++nu) {
            const float m0 = acc[nu][r], m1 = acc[4 + nu][r], m2 = acc[8 + nu][r], m3 = acc[12 + nu][r];
            c[0][nu] = m0 + m1 + m2;
            c[1][nu] = m1 - m2;
            c[2][nu] = m1 + m2 + m3;
        }
        const unsigned so = (unsigned)((cob + (r & 3) + 8 * (r >> 2)) * g.Cin) * 4u;
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) {
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, c[rr][0] + c[rr][1] + c[rr][2]), ro, vo, so + (unsigned)(rr * 3 + 0) * tapb, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, c[rr][1] - c[rr][2]), ro, vo, so + (unsigned)(rr * 3 + 1) * tapb, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, c[rr][1] + c[rr][2] + c[rr][3]), ro, vo, so + (unsigned)(rr * 3 + 2) * tapb, 0);
        }
    }
}

// dw[cout][cin][tap] (+)= sum over the pixel-split slabs [tap][cout][cin].  As conv.hip's sum_slabs_kernel: a workgroup covers 32 float4
// elements (4 consecutive input channels of one (tap, cout)) x 8 slab groups, a thread adds every 8th slab with 4 loads in flight, the 8
// partial sums meet in LDS; the first group transposes on the way out.  Fixed summation order: reproducible.
__global__ __launch_bounds__(256) void wino_wgrad_sum_kernel(const float* __restrict__ part, int nslabs, size_t slab, int cout, int cin,
                                                             float* __restrict__ dw, int accumulate) {
    __shared__ f32x4 red[8][32];
    const int q4 = cin >> 2, n4 = 9 * cout * q4;
    const int el = threadIdx.x & 31, grp = threadIdx.x >> 5;
    for (int base = blockIdx.x * 32; base < n4; base += gridDim.x * 32) {
        const int e = base + el;
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        if (e < n4) {
            const float* src = part + (size_t)e * 4;
            int sp = grp;
            for (; sp + 24 < nslabs; sp += 32) {
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(src + (size_t)sp * slab);
                const f32x4 v1 = *reinterpret_cast<const f32x4*>(src + (size_t)(sp + 8) * slab);
                const f32x4 v2 = *reinterpret_cast<const f32x4*>(src + (size_t)(sp + 16) * slab);
                const f32x4 v3 = *reinterpret_cast<const f32x4*>(src + (size_t)(sp + 24) * slab);
                a += (v0 + v1) + (v2 + v3);
            }
            for (; sp < nslabs; sp += 8) a += *reinterpret_cast<const f32x4*>(src + (size_t)sp * slab);
        }
        red[grp][el] = a;
        __syncthreads();
        if (grp == 0 && e < n4) {
#pragma unroll
            for (int k = 1; k < 8; ++k) a += red[k][el];
            const int c4 = e % q4, t1 = e / q4;
            const int co = t1 % cout, tap = t1 / cout;
            float* dst = dw + ((size_t)co * cin + 4 * c4) * 9 + tap;
#pragma unroll
            for (int k = 0; k < 4; ++k) dst[k * 9] = accumulate ? dst[k * 9] + a[k] : a[k];
        }
        __syncthreads();
    }
}

int pow2floor_log(int v) {
    int l = 0;
    while ((2 << l) <= v) ++l;
    return l;
}

}  // namespace

namespace agan {
namespace conv {

// Plan of the Winograd kernel for gather geometry g (forward or data gradient of a conv3x3 stride 1 pad 1); p.ok == 0: not taken.
WinoPlan plan_wino(const Geom& g) {
    WinoPlan p;
    memset(&p, 0, sizeof(p));
    static const bool off = getenv("AGAN_WINO_OFF") != nullptr;
    if (off) return p;
    static const bool s2off = getenv("AGAN_WINO_S2_OFF") != nullptr;
    if (g.SY == 1 && g.R == 3 && g.S == 3 && g.DY == 1 && g.OS == 1 && g.OY0 == -1 && g.IH == g.OH && g.IW == g.OW) p.s2 = 0;
    else if (!s2off && g.SY == 2 && g.R == 4 && g.S == 4 && g.DY == 1 && g.OS == 1 && g.OY0 == -1 && g.IH == 2 * g.OH && g.IW == 2 * g.OW) p.s2 = 1;
    else if (!s2off && g.SY == 1 && g.R == 2 && g.S == 2 && g.DY == -1 && g.OS == 2 && g.OY0 == 0 && g.OY1 == 1 && g.OH == 2 * g.IH && g.OW == 2 * g.IW) p.s2 = 2;
    else return p;
    if ((g.OHs & 1) || (g.OWs & 1) || (g.Cin & 7) || g.Cout < 32) return p;
    if (p.s2 == 1 && g.Cout < 96) return p;                   // (the stride-2 forward kernel is built for 128-channel tiles)
    // (the class kernel walks 16 channels at a time; with 64 output channels -- 64-tile blocks, short K loops, 64 scalar stores per lane -- it
    //  measured 15 % SLOWER than the direct kernel, with 128+ 15 % faster)
    if (p.s2 == 2 && ((g.Cin & 15) || g.Cout < 96)) return p;
    const int tw = g.OWs / 2, th = g.OHs / 2;                 // tiles per row / column of the output lattice
    if (tw < 4) return p;                                     // (4x4 maps: two tiles per row -- the direct kernel's K split serves them better)
    p.nf = g.Cout >= 96 ? 4 : 2;                             // cout fragments per workgroup; 4: 32 tiles, 2: 64 tiles
    static const bool hoff = getenv("AGAN_WINO_H_OFF") != nullptr;
    p.hmode = (p.s2 == 0 && !hoff && (g.Cout & 7) == 0) ? 1 : 0;   // conv3x3: 32 tiles x 64 channels, two workgroups per CU (conv_wino_h_f32_kernel)
    if (p.hmode) p.nf = 4;                                    // (tile geometry of the 32-tile blocks)
    const int lt = p.nf == 4 ? 5 : 6;
    p.txl = std::min(3, pow2floor_log(tw));
    p.tyl = std::min(lt - p.txl, pow2floor_log(th));
    const int TX = 1 << p.txl, TY = 1 << p.tyl, TB = (1 << lt) >> (p.txl + p.tyl);
    p.blocks_x = cdiv(tw, TX);
    p.blocks_y = cdiv(th, TY);
    p.blocks_b = cdiv(g.B, TB);
    p.mtiles = p.blocks_x * p.blocks_y * p.blocks_b;
    p.ntiles = p.hmode ? cdiv(g.Cout, 64) : cdiv(g.Cout, p.nf * 32);
    // the 32 x 32 x 2 product needs thousands of tiles to pay for a workgroup's 256 accumulators: small layers stay on the direct kernel
    // (stride 2, measured: 384 workgroups = 1.5 rounds of the chip gain nothing, 768+ do: such layers split their input channels over two
    //  workgroups and sum the halves -- their outputs are small)
    p.ksplit = 1;
    if (p.s2 == 1 && p.mtiles * p.ntiles < 512 && p.mtiles * p.ntiles >= 256 && g.Cin >= 128) p.ksplit = 2;
    if (p.mtiles * p.ntiles * p.ksplit * (p.s2 == 2 ? 4 : 1) < (p.s2 ? 512 : 256)) return p;
    p.chunks_per_split = cdiv(g.Cin / 8, p.ksplit);
    p.slab = ((size_t)g.B * g.Cout * g.OH * g.OW + 3) / 4 * 4;
    const int PR = p.s2 == 2 ? 2 * TY + 1 : (p.s2 ? 4 : 2) * TY + 2, PC = p.s2 == 2 ? 2 * TX + 1 : (p.s2 ? 4 : 2) * TX + 2;
    const int cch = p.s2 == 2 ? 16 : 8;                       // channels per chunk
    p.pcp = (PC + 1) & ~1;
    p.mpc = (65536 + PC - 1) / PC;
    p.mpr = (65536 + PR - 1) / PR;
    p.raw_items = TB * cch * PR * PC;
    if (p.raw_items > (p.s2 == 2 ? (p.nf == 4 ? 11 : 19) : (p.s2 ? 21 : (p.nf == 4 ? 7 : 11))) * 256) return p;
    p.raw_bytes = (TB * cch * PR * p.pcp * 4 + 15) & ~15;
    const int planes = p.s2 == 2 ? 9 * 4 : (p.s2 ? 36 : 16);  // (class kernel: 9 positions x 4 classes of weights; its V holds 9 x 16 channels)
    p.smem_bytes = 2 * p.raw_bytes + 2 * (p.s2 == 2 ? 9 * 2 : planes) * (1 << lt) * 8 * 4;      // two raw patches, two V buffers
    if (p.smem_bytes > 160 * 1024) return p;
    p.u_bytes = ((size_t)planes * (g.Cin / 8) * g.Nld * 8 * sizeof(float) + 255) / 256 * 256;
    p.ws_bytes = p.u_bytes + (p.ksplit > 1 ? p.slab * p.ksplit * sizeof(float) : 0);
    p.ok = 1;
    return p;
}

// wk: the mode's ordinary fp32 pack [K = (c, r, s)][Nld]; ws: >= p.u_bytes
void launch_wino(const float* in, const float* wk, float* out, const Geom& g, const WinoPlan& p, void* ws, hipStream_t st) {
    float* U = static_cast<float*>(ws);
    const int total = g.Cin * g.Nld;
    dim3 grid(p.mtiles * p.ntiles * p.ksplit);
    if (p.s2 == 2) {
        AGAN_LAUNCH(wino_cls_weight_transform_kernel, dim3(std::min(cdiv(4 * total, 256), 4096)), dim3(256), 0, st, wk, U, g.Cin, g.Nld, g.Cout);
        dim3 gridc(p.mtiles * p.ntiles * 4);
        if (p.nf == 4) {
            static const hipError_t a4 = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino_cls_f32_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)a4;
            AGAN_LAUNCH(conv_wino_cls_f32_kernel<4>, gridc, dim3(256), (size_t)p.smem_bytes, st, in, U, out, g, p);
        } else {
            static const hipError_t a2 = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino_cls_f32_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)a2;
            AGAN_LAUNCH(conv_wino_cls_f32_kernel<2>, gridc, dim3(256), (size_t)p.smem_bytes, st, in, U, out, g, p);
        }
        return;
    }
    if (p.s2) {
        AGAN_LAUNCH(wino_s2_weight_transform_kernel, dim3(std::min(cdiv(total, 256), 2048)), dim3(256), 0, st, wk, U, g.Cin, g.Nld, g.Cout);
        static const hipError_t a4 = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino_s2_f32_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)a4;
        // (a split launch writes its partial sums to the slabs behind U in the workspace: the caller sums them into `out`)
        float* dst = p.ksplit > 1 ? reinterpret_cast<float*>(static_cast<char*>(ws) + p.u_bytes) : out;
        AGAN_LAUNCH(conv_wino_s2_f32_kernel<4>, grid, dim3(256), (size_t)p.smem_bytes, st, in, U, dst, g, p);
        return;
    }
    AGAN_LAUNCH(wino_weight_transform_kernel, dim3(std::min(cdiv(total, 256), 2048)), dim3(256), 0, st, wk, U, g.Cin, g.Nld, g.Cout);
    if (p.hmode) {
        static const hipError_t ah = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino_h_f32_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        (void)ah;
        AGAN_LAUNCH(conv_wino_h_f32_kernel, grid, dim3(256), (size_t)p.smem_bytes, st, in, U, out, g, p);
        return;
    }
    if (p.nf == 4) {
        static const hipError_t a4 = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino_f32_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)a4;
        AGAN_LAUNCH(conv_wino_f32_kernel<4>, grid, dim3(256), (size_t)p.smem_bytes, st, in, U, out, g, p);
    } else {
        static const hipError_t a2 = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino_f32_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)a2;
        AGAN_LAUNCH(conv_wino_f32_kernel<2>, grid, dim3(256), (size_t)p.smem_bytes, st, in, U, out, g, p);
    }
}


// Winograd weight gradient of a conv3x3 stride 1 pad 1 (forward geometry g); p.ok == 0: not taken.
WinoWgradPlan plan_wino_wgrad(const Geom& g) {
    WinoWgradPlan p;
    memset(&p, 0, sizeof(p));
    static const bool off = getenv("AGAN_WINO_OFF") != nullptr || getenv("AGAN_WINO_WGRAD_OFF") != nullptr;
    if (off) return p;
    if (!(g.SY == 1 && g.R == 3 && g.S == 3 && g.DY == 1 && g.OS == 1 && g.OY0 == -1)) return p;
    if (g.IH != g.OH || g.IW != g.OW || (g.OH & 1) || (g.OW & 15) || g.Cin < 32 || g.Cout < 32 || (g.Cin & 3) || (g.Cout & 7)) return p;
    p.noct = g.B * (g.OH / 2) * (g.OW / 16);
    p.jtiles = cdiv(g.Cout, 64);
    p.itiles = cdiv(g.Cin, 64);
    const int wgs = p.jtiles * p.itiles;
    if (p.noct < 64 * 16 || wgs > 256) return p;               // enough pixels to split over the chip with long loops; (many-channel layers have few pixels here)
    const int ps = std::max(1, std::min(256 / wgs, p.noct / 16));
    p.octs_per_split = cdiv(p.noct, ps);
    p.psplit = cdiv(p.noct, p.octs_per_split);
    p.slab = ((size_t)g.Cout * g.Cin * 9 + 3) / 4 * 4;
    p.ws_bytes = p.slab * p.psplit * sizeof(float);
    p.smem_bytes = 2 * 2 * 16 * 64 * 8 * 4;
    p.ok = 1;
    return p;
}

void launch_wino_wgrad(const float* x, const float* dy, float* part, const Geom& g, const WinoWgradPlan& p, hipStream_t st) {
    static const hipError_t a = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino_wgrad_f32_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)a;
    AGAN_LAUNCH(conv_wino_wgrad_f32_kernel, dim3(p.jtiles * p.itiles * p.psplit), dim3(256), (size_t)p.smem_bytes, st, x, dy, part, g, p);
}

void launch_wino_wgrad_sum(const float* part, const Geom& g, const WinoWgradPlan& p, float* dw, int accumulate, hipStream_t st) {
    const int n4 = 9 * g.Cout * (g.Cin >> 2);
    AGAN_LAUNCH(wino_wgrad_sum_kernel, dim3(std::min(cdiv(n4, 32), 8192)), dim3(256), 0, st, part, p.psplit, p.slab, g.Cout, g.Cin, dw, accumulate);
}

}  // namespace conv
}  // namespace agan
