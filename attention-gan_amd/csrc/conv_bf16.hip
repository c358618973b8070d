// AGAN_PREC_BF16X3: the same implicit-GEMM geometry as conv.hip, on the bf16 matrix cores at ~fp32 accuracy.
//
// Every fp32 operand x is split on the fly into two bf16 values  hi = top16(x),  lo = top16(x - hi)  (x - hi is exact in
// fp32, so hi + lo carries 16 mantissa bits) and each product is issued as three v_mfma_f32_32x32x16_bf16:
//      w*a  ~=  w_hi*a_hi + w_hi*a_lo + w_lo*a_hi          (dropped term w_lo*a_lo ~ 2^-32 relative; fp32 accumulate)
// i.e. ~2^-16 relative error per product, far inside the 1e-3 parity budget, at one third of the bf16 MFMA rate
// (833 TFLOP/s-equivalent peak vs 157 for the f32 MFMA).  HBM tensors stay fp32 NCHW; nothing else in the step changes.
//
// Operand staging follows what the MFMA fragment wants (lane = row, 8 consecutive k per lane, one ds_read_b128):
//   gather kernel : each thread gathers 8 consecutive k of ONE pixel (the k-table makes k wave-uniform), so its 8 values are
//                   exactly one fragment row: packed to bf16 and written with one ds_write_b128 into [pixel][k] tiles;
//                   weights are pre-packed in HBM as bf16 hi/lo planes [cout][K] and copied 16 B at a time.
//   wgrad kernel  : the reduction runs over pixels, which are the contiguous axis of NCHW: lanes load pixel pairs of a row
//                   (coalesced), pack them and write [row][pixel] tiles; fragments are again plain 16-B reads.
// Rows are padded to 80 B: ds_read_b128 (16-lane groups) and ds_write_b128 (8-lane groups) are both conflict-free.
#include "conv_common.h"

using namespace agan;
using namespace agan::conv;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int kLdk = 40;   // bf16 elements per LDS row: 32 data + 8 pad = 80 bytes

__device__ __forceinline__ u32x4 buf_load_u4(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}

// hi/lo split of two floats, packed as two bf16 pairs (element order: a = low half)
__device__ __forceinline__ void split_pack2(float a, float b, unsigned& hi, unsigned& lo) {
    const unsigned ua = __float_as_uint(a) & 0xFFFF0000u, ub = __float_as_uint(b) & 0xFFFF0000u;
    const unsigned ra = __float_as_uint(a - __uint_as_float(ua)), rb = __float_as_uint(b - __uint_as_float(ub));
    hi = __builtin_amdgcn_perm(ub, ua, 0x07060302u);     // (ua >> 16) | (ub & 0xFFFF0000)
    lo = __builtin_amdgcn_perm(rb, ra, 0x07060302u);
}

// ================================================================================================
// forward / dgrad gather kernel
// ================================================================================================
template <int BN, int WM, int WN>
__global__ __launch_bounds__(256, 2) void conv_gather_bf16x3_kernel(const float* __restrict__ in, const unsigned short* __restrict__ wkb,
                                                                     const float* __restrict__ bias, float* __restrict__ out,
                                                                     const int2* __restrict__ ktab, const Geom g, const int ksplit,
                                                                     const int kchunk, const size_t slab, const int Kp) {
    constexpr int BM = 128, BK = 32;
    constexpr int NCH = BN * 4 * 2;                 // 16-byte weight chunks per tile (hi + lo planes)
    constexpr int BV = (NCH + 255) / 256;
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    static_assert(WM * WN == 4 && TM >= 1 && TN >= 1, "bad wave tiling");

    __shared__ __attribute__((aligned(16))) unsigned short Ah[2][BM][kLdk];
    __shared__ __attribute__((aligned(16))) unsigned short Al[2][BM][kLdk];
    __shared__ __attribute__((aligned(16))) unsigned short Wh[2][BN][kLdk];
    __shared__ __attribute__((aligned(16))) unsigned short Wl[2][BN][kLdk];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wn = wave / WM;
    const int cls = blockIdx.z / ksplit, split = blockIdx.z - cls * ksplit;
    const int py = cls / g.OS, px = cls - py * g.OS;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int kbeg = split * kchunk, kend = min(g.K, kbeg + kchunk);
    const int nkt = cdiv(kend - kbeg, BK);
    const int ihw = g.IH * g.IW;

    const __amdgpu_buffer_rsrc_t rin = make_rsrc(in, (size_t)g.B * g.Cin * ihw * sizeof(float));
    const size_t plane = (size_t)g.Nld * Kp;        // one bf16 plane [Nld][Kp]
    const __amdgpu_buffer_rsrc_t rwk = make_rsrc(wkb + (size_t)cls * 2 * plane, 2 * plane * sizeof(unsigned short));

    // ---- per-thread pixel for the A gather: thread owns pixel am and the k groups akg, akg+2 (8 consecutive k each) ----
    const int am = tid & (BM - 1);
    const int akg = __builtin_amdgcn_readfirstlane(tid >> 7);
    const int m = m0 + am;
    const bool mvalid = m < g.Mtot;
    int iy0, ix0, pix0;
    {
        const int mm = mvalid ? m : 0;
        const int b = g.dHWs.div(mm), rem = mm - b * g.HWs;
        const int yq = g.dOWs.div(rem), xq = rem - yq * g.OWs;
        iy0 = yq * g.SY + (py ? g.OY1 : g.OY0);
        ix0 = xq * g.SY + (px ? g.OY1 : g.OY0);
        pix0 = b * g.Cin * ihw + iy0 * g.IW + ix0;
        if (!mvalid) iy0 = -(1 << 20);
    }
    // weight chunk mapping: f -> (plane, row, 16-byte chunk)
    int wrow[BV], wc4[BV], wpl[BV];
#pragma unroll
    for (int j = 0; j < BV; ++j) {
        const int f = tid + j * 256;
        wpl[j] = f / (BN * 4);
        const int rem = f - wpl[j] * (BN * 4);
        wrow[j] = rem >> 2;
        wc4[j] = rem & 3;
    }

    float areg[2][2][8];
    u32x4 wreg[2][BV];

    auto load_tile = [&](auto set, int kt) {
        constexpr int P = decltype(set)::value;
        const int kb = kbeg + kt * BK;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int4* tk4 = reinterpret_cast<const int4*>(ktab + kb + (akg + 2 * q) * 8);      // wave-uniform -> one wide scalar load
            int te[16];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int4 v = tk4[i];
                te[4 * i] = v.x; te[4 * i + 1] = v.y; te[4 * i + 2] = v.z; te[4 * i + 3] = v.w;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int ex = te[2 * i], ey = te[2 * i + 1];
                const int dy = (int)(short)(ey & 0xFFFF), dx = ey >> 16;
                const bool ok = ((unsigned)(iy0 + dy) < (unsigned)g.IH) & ((unsigned)(ix0 + dx) < (unsigned)g.IW);
                const unsigned off = (unsigned)(pix0 + ex) * 4u;
                areg[P][q][i] = buf_load(rin, ok ? off : kOOB);
            }
        }
#pragma unroll
        for (int j = 0; j < BV; ++j) {
            const int n = n0 + wrow[j];
            const bool ok = (BV * 256 == NCH || tid + j * 256 < NCH) & (n < g.Nld);
            wreg[P][j] = buf_load_u4(rwk, ok ? (unsigned)(((size_t)wpl[j] * g.Nld + n) * Kp + kb + wc4[j] * 8) * 2u : kOOB);
        }
    };
    auto store_tile = [&](auto set, int buf) {
        constexpr int P = decltype(set)::value;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            u32x4 h, l;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                unsigned hh, ll;
                split_pack2(areg[P][q][2 * i], areg[P][q][2 * i + 1], hh, ll);
                h[i] = hh;
                l[i] = ll;
            }
            *reinterpret_cast<u32x4*>(&Ah[buf][am][(akg + 2 * q) * 8]) = h;
            *reinterpret_cast<u32x4*>(&Al[buf][am][(akg + 2 * q) * 8]) = l;
        }
#pragma unroll
        for (int j = 0; j < BV; ++j) {
            if (BV * 256 == NCH || tid + j * 256 < NCH) {
                unsigned short* dst = wpl[j] ? &Wl[buf][wrow[j]][wc4[j] * 8] : &Wh[buf][wrow[j]][wc4[j] * 8];
                *reinterpret_cast<u32x4*>(dst) = wreg[P][j];
            }
        }
    };

    f32x16 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int l31 = lane & 31, lh = lane >> 5;
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    // FULL = steady state (no conditionals, so the compiler's waitcnt bookkeeping stays exact across iterations)
    auto step = [&](auto full, auto cur, auto nxt, int kt, int buf) {
        constexpr bool FULL = decltype(full)::value;
        if (FULL || kt + 2 < nkt) load_tile(cur, kt + 2);
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            const int kc = ks * 16 + lh * 8;
            bf16x8 ah[TM], al[TM], wh[TN], wl[TN];
#pragma unroll
            for (int t = 0; t < TM; ++t) {
                ah[t] = *reinterpret_cast<const bf16x8*>(&Ah[buf][wm * WTM + t * 32 + l31][kc]);
                al[t] = *reinterpret_cast<const bf16x8*>(&Al[buf][wm * WTM + t * 32 + l31][kc]);
            }
#pragma unroll
            for (int t = 0; t < TN; ++t) {
                wh[t] = *reinterpret_cast<const bf16x8*>(&Wh[buf][wn * WTN + t * 32 + l31][kc]);
                wl[t] = *reinterpret_cast<const bf16x8*>(&Wl[buf][wn * WTN + t * 32 + l31][kc]);
            }
#pragma unroll
            for (int a = 0; a < TN; ++a)
#pragma unroll
                for (int b = 0; b < TM; ++b) {
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[a], ah[b], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[a], al[b], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[a], ah[b], acc[a][b], 0, 0, 0);
                }
        }
        if (FULL || kt + 1 < nkt) store_tile(nxt, buf ^ 1);
        __syncthreads();
    };

    if (nkt > 0) load_tile(S0{}, 0);
    if (nkt > 1) load_tile(S1{}, 1);
    if (nkt > 0) store_tile(S0{}, 0);
    __syncthreads();
    int kt = 0;
    for (; kt + 3 < nkt; kt += 2) {
        step(std::true_type{}, S0{}, S1{}, kt, 0);
        step(std::true_type{}, S1{}, S0{}, kt + 1, 1);
    }
    for (; kt < nkt; kt += 2) {
        step(std::false_type{}, S0{}, S1{}, kt, 0);
        if (kt + 1 < nkt) step(std::false_type{}, S1{}, S0{}, kt + 1, 1);
    }

    // ---- epilogue: D[cout][pixel] (same accumulator layout as the f32 kernel) ----
    const size_t ohw = (size_t)g.OH * g.OW;
    float* dst = (ksplit > 1) ? out + (size_t)split * slab : out;
    const __amdgpu_buffer_rsrc_t rout = make_rsrc(dst, (size_t)g.B * g.Cout * ohw * sizeof(float));
    const bool add_bias = (bias != nullptr) && (ksplit == 1);
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        const int mo = m0 + wm * WTM + tm * 32 + l31;
        const bool pvalid = mo < g.Mtot;
        const int mm = pvalid ? mo : 0;
        const int b = g.dHWs.div(mm), rem = mm - b * g.HWs;
        const int yq = g.dOWs.div(rem), xq = rem - yq * g.OWs;
        const unsigned pixoff = (unsigned)(b * g.Cout) * (unsigned)ohw + (unsigned)((yq * g.OS + py) * g.OW + (xq * g.OS + px));
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wn * WTN + tn * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                float v = acc[tn][tm][r];
                if (add_bias) v += bias[min(n, g.Cout - 1)];
                buf_store(rout, (pvalid & (n < g.Cout)) ? (pixoff + (unsigned)n * (unsigned)ohw) * 4u : kOOB, v);
            }
        }
    }
}

// ================================================================================================
// weight gradient
// ================================================================================================
template <int BI, int BJ>
__global__ __launch_bounds__(256, 2) void conv_wgrad_bf16x3_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                    float* __restrict__ dst, const int2* __restrict__ ktab, const Geom g,
                                                                    const int psplit, const int pchunk, const size_t slab) {
    constexpr int BP = 32;
    constexpr int XR = BI / 16, YR = BJ / 16;     // (row, pixel-pair) items per thread per tile
    constexpr int TI = BI / 64, TJ = BJ / 64;
    __shared__ __attribute__((aligned(16))) unsigned short Xh[2][BI][kLdk];
    __shared__ __attribute__((aligned(16))) unsigned short Xl[2][BI][kLdk];
    __shared__ __attribute__((aligned(16))) unsigned short Yh[2][BJ][kLdk];
    __shared__ __attribute__((aligned(16))) unsigned short Yl[2][BJ][kLdk];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wave & 1, wj = wave >> 1;
    const int cls = blockIdx.z / psplit, split = blockIdx.z - cls * psplit;
    const int py = cls / g.OS, px = cls - py * g.OS;
    const int i0 = blockIdx.x * BI, j0 = blockIdx.y * BJ;
    const int pbeg = split * pchunk, pend = min(g.Mtot, pbeg + pchunk);
    const int npt = cdiv(pend - pbeg, BP);
    const int pp2 = lane & 15, rsub = lane >> 4;          // pixel pair within the tile, row within a group of 4
    const int l31 = lane & 31, lh = lane >> 5;
    const int ihw = g.IH * g.IW, ohw = g.OH * g.OW;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(x, (size_t)g.B * g.Cin * ihw * sizeof(float));
    const __amdgpu_buffer_rsrc_t rdy = make_rsrc(dy, (size_t)g.B * g.Cout * ohw * sizeof(float));

    // the reduction rows this thread stages never change across the pixel loop: fetch their table entries once
    int ex[XR], ey[XR];
#pragma unroll
    for (int ii = 0; ii < XR; ++ii) {
        const int2 e = ktab[i0 + wave * (BI / 4) + 4 * ii + rsub];
        ex[ii] = e.x;
        ey[ii] = e.y;
    }
    const int nbase = j0 + wave * (BJ / 4) + rsub;

    float xreg[2][XR][2], yreg[2][YR][2];

    auto load_tile = [&](auto set, int pt) {
        constexpr int P = decltype(set)::value;
        int iy0[2], ix0[2], pix0[2], dyoff[2];
        bool pv[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int p = pbeg + pt * BP + 2 * pp2 + h;
            pv[h] = p < pend;
            const int pq = pv[h] ? p : 0;
            const int b = g.dHWs.div(pq), rem = pq - b * g.HWs;
            const int yq = g.dOWs.div(rem), xq = rem - yq * g.OWs;
            iy0[h] = pv[h] ? yq * g.SY + (py ? g.OY1 : g.OY0) : -(1 << 20);
            ix0[h] = xq * g.SY + (px ? g.OY1 : g.OY0);
            pix0[h] = b * g.Cin * ihw + iy0[h] * g.IW + ix0[h];
            dyoff[h] = b * g.Cout * ohw + (yq * g.OS + py) * g.OW + (xq * g.OS + px);
        }
#pragma unroll
        for (int ii = 0; ii < XR; ++ii) {
            const int ddy = (int)(short)(ey[ii] & 0xFFFF), ddx = ey[ii] >> 16;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const bool ok = ((unsigned)(iy0[h] + ddy) < (unsigned)g.IH) & ((unsigned)(ix0[h] + ddx) < (unsigned)g.IW);
                xreg[P][ii][h] = buf_load(rx, ok ? (unsigned)(pix0[h] + ex[ii]) * 4u : kOOB);
            }
        }
#pragma unroll
        for (int jj = 0; jj < YR; ++jj) {
            const int n = nbase + 4 * jj;
#pragma unroll
            for (int h = 0; h < 2; ++h)
                yreg[P][jj][h] = buf_load(rdy, (pv[h] & (n < g.Cout)) ? (unsigned)(dyoff[h] + n * ohw) * 4u : kOOB);
        }
    };
    auto store_tile = [&](auto set, int buf) {
        constexpr int P = decltype(set)::value;
#pragma unroll
        for (int ii = 0; ii < XR; ++ii) {
            unsigned hh, ll;
            split_pack2(xreg[P][ii][0], xreg[P][ii][1], hh, ll);
            const int row = wave * (BI / 4) + 4 * ii + rsub;
            *reinterpret_cast<unsigned*>(&Xh[buf][row][2 * pp2]) = hh;
            *reinterpret_cast<unsigned*>(&Xl[buf][row][2 * pp2]) = ll;
        }
#pragma unroll
        for (int jj = 0; jj < YR; ++jj) {
            unsigned hh, ll;
            split_pack2(yreg[P][jj][0], yreg[P][jj][1], hh, ll);
            const int row = wave * (BJ / 4) + 4 * jj + rsub;
            *reinterpret_cast<unsigned*>(&Yh[buf][row][2 * pp2]) = hh;
            *reinterpret_cast<unsigned*>(&Yl[buf][row][2 * pp2]) = ll;
        }
    };

    f32x16 acc[TJ][TI];
#pragma unroll
    for (int a = 0; a < TJ; ++a)
#pragma unroll
        for (int b = 0; b < TI; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    auto step = [&](auto full, auto cur, auto nxt, int pt, int buf) {
        constexpr bool FULL = decltype(full)::value;
        if (FULL || pt + 2 < npt) load_tile(cur, pt + 2);
#pragma unroll
        for (int ks = 0; ks < BP / 16; ++ks) {
            const int pc = ks * 16 + lh * 8;
            bf16x8 xh[TI], xl[TI], yh[TJ], yl[TJ];
#pragma unroll
            for (int t = 0; t < TI; ++t) {
                xh[t] = *reinterpret_cast<const bf16x8*>(&Xh[buf][wi * (BI / 2) + t * 32 + l31][pc]);
                xl[t] = *reinterpret_cast<const bf16x8*>(&Xl[buf][wi * (BI / 2) + t * 32 + l31][pc]);
            }
#pragma unroll
            for (int t = 0; t < TJ; ++t) {
                yh[t] = *reinterpret_cast<const bf16x8*>(&Yh[buf][wj * (BJ / 2) + t * 32 + l31][pc]);
                yl[t] = *reinterpret_cast<const bf16x8*>(&Yl[buf][wj * (BJ / 2) + t * 32 + l31][pc]);
            }
#pragma unroll
            for (int a = 0; a < TJ; ++a)
#pragma unroll
                for (int b = 0; b < TI; ++b) {
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(yl[a], xh[b], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(yh[a], xl[b], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(yh[a], xh[b], acc[a][b], 0, 0, 0);
                }
        }
        if (FULL || pt + 1 < npt) store_tile(nxt, buf ^ 1);
        __syncthreads();
    };

    if (npt > 0) load_tile(S0{}, 0);
    if (npt > 1) load_tile(S1{}, 1);
    if (npt > 0) store_tile(S0{}, 0);
    __syncthreads();
    int pt = 0;
    for (; pt + 3 < npt; pt += 2) {
        step(std::true_type{}, S0{}, S1{}, pt, 0);
        step(std::true_type{}, S1{}, S0{}, pt + 1, 1);
    }
    for (; pt < npt; pt += 2) {
        step(std::false_type{}, S0{}, S1{}, pt, 0);
        if (pt + 1 < npt) step(std::false_type{}, S1{}, S0{}, pt + 1, 1);
    }

    // D[cout][k index]
    float* o = dst + (size_t)split * slab + (size_t)cls * g.Cout * g.K;
    const __amdgpu_buffer_rsrc_t ro = make_rsrc(o, (size_t)g.Cout * g.K * sizeof(float));
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
        for (int ti = 0; ti < TI; ++ti) {
            const int i = i0 + wi * (BI / 2) + ti * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = j0 + wj * (BJ / 2) + tj * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                buf_store(ro, ((i < g.K) & (n < g.Cout)) ? (unsigned)(n * g.K + i) * 4u : kOOB, acc[tj][ti][r]);
            }
        }
}

// packed bf16 weights: [cls][plane hi|lo][Nld][Kp], k fastest
__global__ __launch_bounds__(256) void pack_weight_bf16_kernel(const float* __restrict__ w, unsigned short* __restrict__ wkb, int mode,
                                                               int cout, int cin, int kh, int kw, int K, int Kp, int Nld, int ncls) {
    const size_t plane = (size_t)Nld * Kp;
    const size_t total = (size_t)ncls * plane;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(e % Kp);
        const size_t t = e / Kp;
        const int n = (int)(t % Nld), cls = (int)(t / Nld);
        const float v = k < K ? packed_weight_value(w, mode, cls, k, n, cout, cin, kh, kw) : 0.f;
        const unsigned hi = __float_as_uint(v) & 0xFFFF0000u;
        const unsigned lo = __float_as_uint(v - __uint_as_float(hi));
        wkb[(size_t)cls * 2 * plane + (size_t)n * Kp + k] = (unsigned short)(hi >> 16);
        wkb[(size_t)cls * 2 * plane + plane + (size_t)n * Kp + k] = (unsigned short)(lo >> 16);
    }
}

}  // namespace

namespace agan {
namespace conv {

int pack_weight_bf16x3(const float* w, void* wk, int mode, int cout, int cin, int kh, int kw, hipStream_t st) {
    int ncls, K, N;
    if (pack_dims(mode, cout, cin, kh, kw, ncls, K, N)) return AGAN_EINVAL;
    const int Nld = agan_round_up(N, 32), Kp = agan_round_up(K, 32);
    const size_t total = (size_t)ncls * Nld * Kp;
    hipLaunchKernelGGL(pack_weight_bf16_kernel, dim3((unsigned)std::min<size_t>(cdivz(total, 256), 8192)), dim3(256), 0, st, w,
                       static_cast<unsigned short*>(wk), mode, cout, cin, kh, kw, K, Kp, Nld, ncls);
    return check_launch("pack_weight/bf16x3");
}

void launch_gather_bf16x3(const float* in, const void* wk, const float* bias, float* dst, const int2* ktab, const Geom& g,
                          const GatherPlan& p, hipStream_t st) {
    dim3 grid(p.mtiles, p.ntiles, p.ncls * p.ksplit);
    const int Kp = agan_round_up(g.K, 32);
    const unsigned short* wkb = static_cast<const unsigned short*>(wk);
    if (p.bn == 128)
        hipLaunchKernelGGL((conv_gather_bf16x3_kernel<128, 2, 2>), grid, dim3(256), 0, st, in, wkb, bias, dst, ktab, g, p.ksplit, p.kchunk, p.slab, Kp);
    else if (p.bn == 64)
        hipLaunchKernelGGL((conv_gather_bf16x3_kernel<64, 2, 2>), grid, dim3(256), 0, st, in, wkb, bias, dst, ktab, g, p.ksplit, p.kchunk, p.slab, Kp);
    else
        hipLaunchKernelGGL((conv_gather_bf16x3_kernel<32, 4, 1>), grid, dim3(256), 0, st, in, wkb, bias, dst, ktab, g, p.ksplit, p.kchunk, p.slab, Kp);
}

void launch_wgrad_bf16x3(const float* x, const float* dy, float* part, const int2* ktab, const Geom& g, const WgradPlan& p, hipStream_t st) {
    dim3 grid(p.itiles, p.jtiles, p.ncls * p.psplit);
#define AGAN_WG(BI, BJ) hipLaunchKernelGGL((conv_wgrad_bf16x3_kernel<BI, BJ>), grid, dim3(256), 0, st, x, dy, part, ktab, g, p.psplit, p.pchunk, p.slab)
    if (p.bi == 128 && p.bj == 128) AGAN_WG(128, 128);
    else if (p.bi == 128) AGAN_WG(128, 64);
    else if (p.bj == 128) AGAN_WG(64, 128);
    else AGAN_WG(64, 64);
#undef AGAN_WG
}

}  // namespace conv
}  // namespace agan
