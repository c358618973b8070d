// Implicit-GEMM convolution engine for gfx950 (MI355X).
//
// One gather kernel + one weight-gradient kernel cover every convolution / linear layer on the AttnGAN path
// (see agan_conv_geom in include/agan.h).  Storage is NCHW fp32 like the reference's tensors, so the pixel axis
// is the contiguous one: lanes always run along pixels for global loads/stores (256-B wave accesses) and the
// reduction index (ci,r,s) is wave-uniform, i.e. decoded on the scalar unit.
//
// AGAN_PREC_F32 uses v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulate): each lane feeds ONE f32 per
// operand, read from LDS tiles laid out [k][pixel] / [k][cout] with a plain ds_read_b32 -- conflict-free because
// lanes 0..31 read 32 consecutive floats and lanes 32..63 the next k row.
// The MFMA is issued as D[cout][pixel] (weights as the A operand) so that every accumulator register holds
// 32 consecutive pixels of one output channel: NCHW stores are 128-B coalesced.
#include "agan_common.h"

#include <algorithm>

using namespace agan;

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

struct Geom {
    int B, Cin, IH, IW, Cout, OH, OW, R, S, OS, SY, DY, OY0, OY1;
    int OHs, OWs, HWs, Mtot, K, Nld, RS;
};

Geom make_geom(const agan_conv_geom* g) {
    Geom d;
    d.B = g->B; d.Cin = g->Cin; d.IH = g->IH; d.IW = g->IW; d.Cout = g->Cout; d.OH = g->OH; d.OW = g->OW;
    d.R = g->R; d.S = g->S; d.OS = g->OS; d.SY = g->SY; d.DY = g->DY; d.OY0 = g->OY[0]; d.OY1 = g->OY[1];
    d.OHs = g->OH / g->OS; d.OWs = g->OW / g->OS; d.HWs = d.OHs * d.OWs; d.Mtot = g->B * d.HWs;
    d.RS = g->R * g->S; d.K = g->Cin * d.RS; d.Nld = agan_round_up(g->Cout, 32);
    return d;
}

int check_geom(const agan_conv_geom* g) {
    AGAN_REQUIRE(g != nullptr, "conv: null geometry");
    AGAN_REQUIRE(g->B > 0 && g->Cin > 0 && g->IH > 0 && g->IW > 0 && g->Cout > 0 && g->OH > 0 && g->OW > 0,
                 "conv: non-positive dimension");
    AGAN_REQUIRE(g->R > 0 && g->S > 0 && g->R <= 8 && g->S <= 8, "conv: taps %dx%d unsupported", g->R, g->S);
    AGAN_REQUIRE(g->OS == 1 || g->OS == 2, "conv: OS must be 1 or 2");
    AGAN_REQUIRE(g->OH % g->OS == 0 && g->OW % g->OS == 0, "conv: OH/OW not divisible by OS");
    const long long in_elems = 1LL * g->B * g->Cin * g->IH * g->IW, out_elems = 1LL * g->B * g->Cout * g->OH * g->OW;
    AGAN_REQUIRE(in_elems < (1LL << 31) && out_elems < (1LL << 31), "conv: tensor exceeds 2^31 elements");
    return AGAN_OK;
}

// (c, r, s) counter advanced on the scalar unit
struct KIdx {
    int c, r, s;
    __device__ __forceinline__ void set(int k, int RS, int S) {
        c = k / RS;
        const int rs = k - c * RS;
        r = rs / S;
        s = rs - r * S;
    }
    __device__ __forceinline__ void step(int R, int S) {
        if (++s == S) {
            s = 0;
            if (++r == R) { r = 0; ++c; }
        }
    }
};

// ================================================================================================
// forward / dgrad gather kernel
// ================================================================================================
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void conv_gather_f32_kernel(const float* __restrict__ in, const float* __restrict__ wk,
                                                               const float* __restrict__ bias, float* __restrict__ out,
                                                               const Geom g, const int ksplit, const int kchunk,
                                                               const size_t slab) {
    constexpr int BK = 16;
    constexpr int NG = 256 / BM;       // wave-uniform k groups for the pixel-major A loads
    constexpr int AK = BK / NG;        // k rows per thread per tile
    constexpr int NB4 = BK * BN / 4;   // float4s in a weight tile
    constexpr int BV = (NB4 + 255) / 256;
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    static_assert(WM * WN == 4 && TM >= 1 && TN >= 1, "bad wave tiling");

    __shared__ float As[2][BK][BM];
    __shared__ float Bs[2][BK][BN];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wn = wave / WM;
    const int cls = blockIdx.z / ksplit, split = blockIdx.z - cls * ksplit;
    const int py = cls / g.OS, px = cls - py * g.OS;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int kbeg = split * kchunk, kend = min(g.K, kbeg + kchunk);
    const int nkt = cdiv(kend - kbeg, BK);

    // ---- per-thread pixel for the A gather -------------------------------------------------------
    const int am = tid % BM;
    const int akg = __builtin_amdgcn_readfirstlane(tid / BM);
    const int m = m0 + am;
    const bool mvalid = m < g.Mtot;
    int iy0, ix0;
    const float* inb;
    {
        const int mm = mvalid ? m : 0;
        const int b = mm / g.HWs, rem = mm - b * g.HWs;
        const int yq = rem / g.OWs, xq = rem - yq * g.OWs;
        iy0 = yq * g.SY + (py ? g.OY1 : g.OY0);
        ix0 = xq * g.SY + (px ? g.OY1 : g.OY0);
        inb = in + (size_t)b * g.Cin * g.IH * g.IW;
    }
    const float* wkc = wk + (size_t)cls * g.K * g.Nld;

    float areg[AK];
    float4 breg[BV];

    auto load_tile = [&](int kt) {
        const int kb = kbeg + kt * BK;
        int k = kb + akg * AK;
        KIdx ki;
        ki.set(k, g.RS, g.S);
#pragma unroll
        for (int i = 0; i < AK; ++i) {
            const int iy = iy0 + ki.r * g.DY, ix = ix0 + ki.s * g.DY;
            const bool ok = mvalid && (k + i) < kend && (unsigned)iy < (unsigned)g.IH && (unsigned)ix < (unsigned)g.IW;
            areg[i] = ok ? inb[(ki.c * g.IH + iy) * g.IW + ix] : 0.f;
            ki.step(g.R, g.S);
        }
#pragma unroll
        for (int j = 0; j < BV; ++j) {
            const int f = tid + j * 256;
            const int kr = f / (BN / 4), nc = (f - kr * (BN / 4)) * 4;
            const int kk = kb + kr, n = n0 + nc;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (f < NB4 && kk < kend && n < g.Nld) v = *reinterpret_cast<const float4*>(wkc + (size_t)kk * g.Nld + n);
            breg[j] = v;
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < AK; ++i) As[buf][akg * AK + i][am] = areg[i];
#pragma unroll
        for (int j = 0; j < BV; ++j) {
            const int f = tid + j * 256;
            const int kr = f / (BN / 4), nc = (f - kr * (BN / 4)) * 4;
            if (f < NB4) *reinterpret_cast<float4*>(&Bs[buf][kr][nc]) = breg[j];
        }
    };

    f32x16 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    if (nkt > 0) {
        load_tile(0);
        store_tile(0);
    }
    __syncthreads();
    const int l31 = lane & 31, lh = lane >> 5;
    int buf = 0;
    for (int kt = 0; kt < nkt; ++kt) {
        const bool more = kt + 1 < nkt;
        if (more) load_tile(kt + 1);
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float av[TM], bv[TN];
#pragma unroll
            for (int t = 0; t < TM; ++t) av[t] = As[buf][kk + lh][wm * WTM + t * 32 + l31];
#pragma unroll
            for (int t = 0; t < TN; ++t) bv[t] = Bs[buf][kk + lh][wn * WTN + t * 32 + l31];
#pragma unroll
            for (int a = 0; a < TN; ++a)
#pragma unroll
                for (int b = 0; b < TM; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[a], av[b], acc[a][b], 0, 0, 0);
        }
        if (more) store_tile(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }

    // ---- epilogue: D[cout][pixel]; lane owns one pixel column, 16 registers = 16 output channels ---
    float* dst = (ksplit > 1) ? out + (size_t)split * slab : out;
    const bool add_bias = (bias != nullptr) && (ksplit == 1);
    const size_t ohw = (size_t)g.OH * g.OW;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        const int mo = m0 + wm * WTM + tm * 32 + l31;
        if (mo >= g.Mtot) continue;
        const int b = mo / g.HWs, rem = mo - b * g.HWs;
        const int yq = rem / g.OWs, xq = rem - yq * g.OWs;
        const size_t pixoff = (size_t)b * g.Cout * ohw + (size_t)(yq * g.OS + py) * g.OW + (xq * g.OS + px);
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wn * WTN + tn * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (n < g.Cout) {
                    float v = acc[tn][tm][r];
                    if (add_bias) v += bias[n];
                    dst[pixoff + (size_t)n * ohw] = v;
                }
            }
        }
    }
}

// out[i] = sum_s ws[s][i] (+ bias[channel])
__global__ __launch_bounds__(256) void sum_slabs_kernel(const float* __restrict__ ws, int nsplit, size_t n, size_t slab,
                                                        const float* __restrict__ bias, int C, int HW,
                                                        float* __restrict__ out) {
    const size_t n4 = n / 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 a = *reinterpret_cast<const float4*>(ws + i * 4);
        for (int s = 1; s < nsplit; ++s) {
            const float4 b = *reinterpret_cast<const float4*>(ws + (size_t)s * slab + i * 4);
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        if (bias) {
            const size_t e = i * 4;
            a.x += bias[(e / HW) % C]; a.y += bias[((e + 1) / HW) % C];
            a.z += bias[((e + 2) / HW) % C]; a.w += bias[((e + 3) / HW) % C];
        }
        *reinterpret_cast<float4*>(out + i * 4) = a;
    }
    // tail (n not a multiple of 4)
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const size_t e = n4 * 4 + threadIdx.x;
        float a = 0.f;
        for (int s = 0; s < nsplit; ++s) a += ws[(size_t)s * slab + e];
        if (bias) a += bias[(e / HW) % C];
        out[e] = a;
    }
}

struct GatherPlan {
    int bn, mtiles, ntiles, ncls, ksplit, kchunk;
    size_t slab, ws_bytes;
};

GatherPlan plan_gather(const Geom& g) {
    GatherPlan p;
    p.bn = g.Cout >= 96 ? 128 : (g.Cout >= 48 ? 64 : 32);
    p.mtiles = cdiv(g.Mtot, 128);
    p.ntiles = cdiv(g.Cout, p.bn);
    p.ncls = g.OS * g.OS;
    const int tiles = p.mtiles * p.ntiles * p.ncls;
    const int ktiles = cdiv(g.K, 16);
    int ks = 1;
    if (tiles < 384) {
        ks = std::min({cdiv(768, tiles), std::max(1, ktiles / 4), 32});
    }
    p.kchunk = cdiv(ktiles, ks) * 16;
    p.ksplit = cdiv(g.K, p.kchunk);
    p.slab = (size_t)g.B * g.Cout * g.OH * g.OW;
    p.slab = (p.slab + 3) / 4 * 4;   // keep every slab 16-B aligned
    p.ws_bytes = p.ksplit > 1 ? p.slab * p.ksplit * sizeof(float) : 0;
    return p;
}

template <int BN, int WM, int WN>
void launch_gather(const float* in, const float* wk, const float* bias, float* dst, const Geom& g, const GatherPlan& p,
                   hipStream_t st) {
    dim3 grid(p.mtiles, p.ntiles, p.ncls * p.ksplit);
    hipLaunchKernelGGL((conv_gather_f32_kernel<128, BN, WM, WN>), grid, dim3(256), 0, st, in, wk, bias, dst, g, p.ksplit,
                       p.kchunk, p.slab);
}

// ================================================================================================
// weight gradient:  dwk[cls][(ci,r,s)][co] = sum_pixels im2col(x)[pixel][(ci,r,s)] * dy[pixel][co]
// ================================================================================================
template <int BI, int BJ>
__global__ __launch_bounds__(256) void conv_wgrad_f32_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                              float* __restrict__ dst, const Geom g, const int psplit,
                                                              const int pchunk, const size_t slab) {
    constexpr int BP = 32, LDP = 33;   // +1 pad: MFMA operand reads walk the row index across lanes
    constexpr int XR = BI / 8, YR = BJ / 8;   // rows per thread per tile
    constexpr int TI = BI / 64, TJ = BJ / 64;
    __shared__ float Xs[2][BI][LDP];
    __shared__ float Ys[2][BJ][LDP];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wave & 1, wj = wave >> 1;
    const int cls = blockIdx.z / psplit, split = blockIdx.z - cls * psplit;
    const int py = cls / g.OS, px = cls - py * g.OS;
    const int i0 = blockIdx.x * BI, j0 = blockIdx.y * BJ;
    const int pbeg = split * pchunk, pend = min(g.Mtot, pbeg + pchunk);
    const int npt = cdiv(pend - pbeg, BP);
    const int pl = lane & 31, half = lane >> 5;
    const size_t ohw = (size_t)g.OH * g.OW;

    float xreg[XR], yreg[YR];

    auto load_tile = [&](int pt) {
        const int p = pbeg + pt * BP + pl;
        const bool pvalid = p < pend;
        const int pp = pvalid ? p : 0;
        const int b = pp / g.HWs, rem = pp - b * g.HWs;
        const int yq = rem / g.OWs, xq = rem - yq * g.OWs;
        const int iy0 = yq * g.SY + (py ? g.OY1 : g.OY0), ix0 = xq * g.SY + (px ? g.OY1 : g.OY0);
        const float* xb = x + (size_t)b * g.Cin * g.IH * g.IW;
        const float* dyb = dy + (size_t)b * g.Cout * ohw + (size_t)(yq * g.OS + py) * g.OW + (xq * g.OS + px);
        // rows owned by this wave: wave*(BI/4) + 2*ii + half   (two consecutive k indices per wave access)
        int ka = i0 + wave * (BI / 4);
        KIdx ia;
        ia.set(ka, g.RS, g.S);
#pragma unroll
        for (int ii = 0; ii < XR; ++ii) {
            KIdx ib = ia;
            ib.step(g.R, g.S);
            const int c = half ? ib.c : ia.c, r = half ? ib.r : ia.r, s = half ? ib.s : ia.s;
            const int iy = iy0 + r * g.DY, ix = ix0 + s * g.DY;
            const bool ok = pvalid && (ka + half) < g.K && (unsigned)iy < (unsigned)g.IH && (unsigned)ix < (unsigned)g.IW;
            xreg[ii] = ok ? xb[(c * g.IH + iy) * g.IW + ix] : 0.f;
            ia = ib;
            ia.step(g.R, g.S);
            ka += 2;
        }
        const int nb = j0 + wave * (BJ / 4) + half;
#pragma unroll
        for (int jj = 0; jj < YR; ++jj) {
            const int n = nb + 2 * jj;
            yreg[jj] = (pvalid && n < g.Cout) ? dyb[(size_t)n * ohw] : 0.f;
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int ii = 0; ii < XR; ++ii) Xs[buf][wave * (BI / 4) + 2 * ii + half][pl] = xreg[ii];
#pragma unroll
        for (int jj = 0; jj < YR; ++jj) Ys[buf][wave * (BJ / 4) + 2 * jj + half][pl] = yreg[jj];
    };

    f32x16 acc[TI][TJ];
#pragma unroll
    for (int a = 0; a < TI; ++a)
#pragma unroll
        for (int b = 0; b < TJ; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    if (npt > 0) {
        load_tile(0);
        store_tile(0);
    }
    __syncthreads();
    int buf = 0;
    for (int pt = 0; pt < npt; ++pt) {
        const bool more = pt + 1 < npt;
        if (more) load_tile(pt + 1);
#pragma unroll
        for (int pp = 0; pp < BP; pp += 2) {
            float av[TI], bv[TJ];
#pragma unroll
            for (int t = 0; t < TI; ++t) av[t] = Xs[buf][wi * (BI / 2) + t * 32 + pl][pp + half];
#pragma unroll
            for (int t = 0; t < TJ; ++t) bv[t] = Ys[buf][wj * (BJ / 2) + t * 32 + pl][pp + half];
#pragma unroll
            for (int a = 0; a < TI; ++a)
#pragma unroll
                for (int b = 0; b < TJ; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a], bv[b], acc[a][b], 0, 0, 0);
        }
        if (more) store_tile(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }

    // D[i = k index][j = cout]: lane owns cout column, registers walk k rows
    float* o = dst + (size_t)split * slab + (size_t)cls * g.K * g.Nld;
#pragma unroll
    for (int ti = 0; ti < TI; ++ti)
#pragma unroll
        for (int tj = 0; tj < TJ; ++tj) {
            const int n = j0 + wj * (BJ / 2) + tj * 32 + pl;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = i0 + wi * (BI / 2) + ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (i < g.K && n < g.Nld) o[(size_t)i * g.Nld + n] = acc[ti][tj][r];
            }
        }
}

struct WgradPlan {
    int bi, bj, itiles, jtiles, ncls, psplit, pchunk;
    size_t slab, ws_bytes;
};

WgradPlan plan_wgrad(const Geom& g) {
    WgradPlan p;
    p.bi = g.K >= 96 ? 128 : 64;
    p.bj = g.Cout >= 96 ? 128 : 64;
    p.itiles = cdiv(g.K, p.bi);
    p.jtiles = cdiv(g.Cout, p.bj);
    p.ncls = g.OS * g.OS;
    const int tiles = p.itiles * p.jtiles * p.ncls;
    const int ptiles = cdiv(g.Mtot, 32);
    int ps = 1;
    if (tiles < 512) ps = std::min({cdiv(1024, tiles), std::max(1, ptiles / 8), 512});
    p.pchunk = cdiv(ptiles, ps) * 32;
    p.psplit = cdiv(g.Mtot, p.pchunk);
    p.slab = (size_t)p.ncls * g.K * g.Nld;
    p.ws_bytes = p.slab * p.psplit * sizeof(float);
    return p;
}

// ================================================================================================
// weight packing (OIHW -> [cls][K][Nld]) and gradient unpacking ([split][cls][K][Nld] -> OIHW)
// ================================================================================================
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

__global__ __launch_bounds__(256) void pack_weight_kernel(const float* __restrict__ w, float* __restrict__ wk, int mode,
                                                          int cout, int cin, int kh, int kw, int K, int Nld, int ncls) {
    // one thread per packed element, n fastest (coalesced writes)
    const size_t total = (size_t)ncls * K * Nld;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int n = (int)(e % Nld);
        const size_t t = e / Nld;
        const int k = (int)(t % K), cls = (int)(t / K);
        float v = 0.f;
        if (mode == AGAN_PACK_FWD) {
            if (n < cout) v = w[(size_t)n * K + k];
        } else if (mode == AGAN_PACK_DGRAD_S1) {
            // k = (co, r, s), n = ci
            const int khw = kh * kw, co = k / khw, rs = k - co * khw, r = rs / kw, s = rs - r * kw;
            if (n < cin) v = w[(((size_t)co * cin + n) * kh + (kh - 1 - r)) * kw + (kw - 1 - s)];
        } else if (mode == AGAN_PACK_DGRAD_4x4S2) {
            // cls = (py,px); k = (co, r, s) with r,s in {0,1}; n = ci; tap kh = ((py+1)&1) + 2r
            const int py = cls >> 1, px = cls & 1, co = k >> 2, r = (k >> 1) & 1, s = k & 1;
            const int th = ((py + 1) & 1) + 2 * r, tw = ((px + 1) & 1) + 2 * s;
            if (n < cin) v = w[(((size_t)co * cin + n) * 4 + th) * 4 + tw];
        } else if (mode == AGAN_PACK_UP_FWD) {
            // cls = (py,px); k = (ci, r', s'); n = co
            const int py = cls >> 1, px = cls & 1, ci = k >> 2, r = (k >> 1) & 1, s = k & 1;
            if (n < cout) {
                int rl, rh, sl, sh;
                up_fwd_taps(py, r, rl, rh);
                up_fwd_taps(px, s, sl, sh);
                const float* wp = w + ((size_t)n * cin + ci) * 9;
                for (int a = rl; a <= rh; ++a)
                    for (int b = sl; b <= sh; ++b) v += wp[a * 3 + b];
            }
        } else if (mode == AGAN_PACK_UP_DGRAD) {
            // k = (co, t, u) with t,u in 0..3; n = ci
            const int co = k >> 4, t = (k >> 2) & 3, u = k & 3;
            if (n < cin) {
                int rl, rh, sl, sh;
                up_dgrad_taps(t, rl, rh);
                up_dgrad_taps(u, sl, sh);
                const float* wp = w + ((size_t)co * cin + n) * 9;
                for (int a = rl; a <= rh; ++a)
                    for (int b = sl; b <= sh; ++b) v += wp[a * 3 + b];
            }
        }
        wk[e] = v;
    }
}

// FWD mode: dw[co][k] = sum_split dwk[split][k][co]  -- tiled transpose, coalesced both ways
__global__ __launch_bounds__(256) void unpack_wgrad_fwd_kernel(const float* __restrict__ dwk, int nsplit, size_t slab,
                                                               float* __restrict__ dw, int cout, int K, int Nld) {
    __shared__ float tile[32][33];
    const int k0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = k0 + ty + j * 8, n = n0 + tx;
        float v = 0.f;
        if (k < K && n < Nld)
            for (int s = 0; s < nsplit; ++s) v += dwk[(size_t)s * slab + (size_t)k * Nld + n];
        tile[ty + j * 8][tx] = v;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + ty + j * 8, k = k0 + tx;
        if (n < cout && k < K) dw[(size_t)n * K + k] = tile[tx][ty + j * 8];
    }
}

// UP mode: dw[co][ci][a][b] = sum over the (class, tap) pairs that 3x3 tap (a,b) was folded into
__global__ __launch_bounds__(256) void unpack_wgrad_up_kernel(const float* __restrict__ dwk, int nsplit, size_t slab,
                                                              float* __restrict__ dw, int cout, int cin, int Nld) {
    const int K = cin * 4;
    const size_t total = (size_t)cout * cin * 9;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int ab = (int)(e % 9), a = ab / 3, b = ab - a * 3;
        const size_t t = e / 9;
        const int ci = (int)(t % cin), co = (int)(t / cin);
        float v = 0.f;
        for (int py = 0; py < 2; ++py)
            for (int r = 0; r < 2; ++r) {
                int rl, rh;
                up_fwd_taps(py, r, rl, rh);
                if (a < rl || a > rh) continue;
                for (int px = 0; px < 2; ++px)
                    for (int s = 0; s < 2; ++s) {
                        int sl, sh;
                        up_fwd_taps(px, s, sl, sh);
                        if (b < sl || b > sh) continue;
                        const size_t off = ((size_t)(py * 2 + px) * K + (ci * 4 + r * 2 + s)) * Nld + co;
                        for (int sp = 0; sp < nsplit; ++sp) v += dwk[(size_t)sp * slab + off];
                    }
            }
        dw[e] = v;
    }
}

__global__ __launch_bounds__(256) void bias_grad_kernel(const float* __restrict__ dy, float* __restrict__ db, int B, int C,
                                                        int HW) {
    __shared__ float red[4];
    const int c = blockIdx.x;
    float s = 0.f;
    for (int b = 0; b < B; ++b) {
        const float* p = dy + ((size_t)b * C + c) * HW;
        for (int i = threadIdx.x; i < HW; i += 256) s += p[i];
    }
    s = block_sum<256>(s, red);
    if (threadIdx.x == 0) db[c] = s;
}

int pack_dims(int mode, int cout, int cin, int kh, int kw, int& ncls, int& K, int& N) {
    switch (mode) {
        case AGAN_PACK_FWD: ncls = 1; K = cin * kh * kw; N = cout; return 0;
        case AGAN_PACK_DGRAD_S1: ncls = 1; K = cout * kh * kw; N = cin; return 0;
        case AGAN_PACK_DGRAD_4x4S2: if (kh != 4 || kw != 4) return -1; ncls = 4; K = cout * 4; N = cin; return 0;
        case AGAN_PACK_UP_FWD: if (kh != 3 || kw != 3) return -1; ncls = 4; K = cin * 4; N = cout; return 0;
        case AGAN_PACK_UP_DGRAD: if (kh != 3 || kw != 3) return -1; ncls = 1; K = cout * 16; N = cin; return 0;
    }
    return -1;
}

}  // namespace

// ================================================================================================
// C ABI
// ================================================================================================
extern "C" {

size_t agan_packed_weight_elems(int mode, int cout, int cin, int kh, int kw) {
    int ncls, K, N;
    if (pack_dims(mode, cout, cin, kh, kw, ncls, K, N)) return 0;
    return (size_t)ncls * K * agan_round_up(N, 32);
}

int agan_pack_weight(const float* w, float* wk, int mode, int cout, int cin, int kh, int kw, void* stream) {
    int ncls, K, N;
    AGAN_REQUIRE(w && wk, "pack_weight: null pointer");
    AGAN_REQUIRE(pack_dims(mode, cout, cin, kh, kw, ncls, K, N) == 0, "pack_weight: mode %d does not take %dx%d", mode, kh, kw);
    const int Nld = agan_round_up(N, 32);
    const size_t total = (size_t)ncls * K * Nld;
    const int blocks = (int)std::min<size_t>(cdivz(total, 256), 8192);
    hipLaunchKernelGGL(pack_weight_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), w, wk, mode, cout, cin, kh, kw, K,
                       Nld, ncls);
    return check_launch("pack_weight");
}

size_t agan_conv_gather_ws_bytes(const agan_conv_geom* g) {
    if (check_geom(g)) return 0;
    return plan_gather(make_geom(g)).ws_bytes;
}

int agan_conv_gather(const float* in, const float* wk, const float* bias, float* out, const agan_conv_geom* gg, int prec,
                     void* ws, size_t ws_bytes, void* stream) {
    if (int e = check_geom(gg)) return e;
    AGAN_REQUIRE(in && wk && out, "conv_gather: null pointer");
    AGAN_REQUIRE(prec == AGAN_PREC_F32, "conv_gather: precision mode %d not built in this version", prec);
    const Geom g = make_geom(gg);
    const GatherPlan p = plan_gather(g);
    if (p.ws_bytes > ws_bytes || (p.ws_bytes && !ws)) {
        set_error("conv_gather: workspace %zu < %zu", ws_bytes, p.ws_bytes);
        return AGAN_EWORKSPACE;
    }
    hipStream_t st = as_stream(stream);
    float* dst = p.ksplit > 1 ? static_cast<float*>(ws) : out;
    if (p.bn == 128) launch_gather<128, 2, 2>(in, wk, bias, dst, g, p, st);
    else if (p.bn == 64) launch_gather<64, 2, 2>(in, wk, bias, dst, g, p, st);
    else launch_gather<32, 4, 1>(in, wk, bias, dst, g, p, st);
    if (int e = check_launch("conv_gather")) return e;
    if (p.ksplit > 1) {
        const size_t n = (size_t)g.B * g.Cout * g.OH * g.OW;
        const int blocks = (int)std::min<size_t>(cdivz(n / 4 + 1, 256), 4096);
        hipLaunchKernelGGL(sum_slabs_kernel, dim3(blocks), dim3(256), 0, st, static_cast<const float*>(ws), p.ksplit, n, p.slab,
                           bias, g.Cout, g.OH * g.OW, out);
        return check_launch("conv_gather/sum_slabs");
    }
    return AGAN_OK;
}

size_t agan_conv_wgrad_ws_bytes(const agan_conv_geom* g) {
    if (check_geom(g)) return 0;
    return plan_wgrad(make_geom(g)).ws_bytes;
}

int agan_conv_wgrad(const float* x, const float* dy, float* dw, const agan_conv_geom* gg, int pack_mode, int kh, int kw,
                    int prec, void* ws, size_t ws_bytes, void* stream) {
    if (int e = check_geom(gg)) return e;
    AGAN_REQUIRE(x && dy && dw && ws, "conv_wgrad: null pointer");
    AGAN_REQUIRE(prec == AGAN_PREC_F32, "conv_wgrad: precision mode %d not built in this version", prec);
    AGAN_REQUIRE(pack_mode == AGAN_PACK_FWD || pack_mode == AGAN_PACK_UP_FWD, "conv_wgrad: pack mode %d is not a forward mode", pack_mode);
    const Geom g = make_geom(gg);
    if (pack_mode == AGAN_PACK_FWD) AGAN_REQUIRE(g.R == kh && g.S == kw && g.OS == 1, "conv_wgrad: geometry is not a direct %dx%d conv", kh, kw);
    else AGAN_REQUIRE(g.R == 2 && g.S == 2 && g.OS == 2 && kh == 3 && kw == 3, "conv_wgrad: geometry is not the folded upsample conv");
    const WgradPlan p = plan_wgrad(g);
    if (p.ws_bytes > ws_bytes) {
        set_error("conv_wgrad: workspace %zu < %zu", ws_bytes, p.ws_bytes);
        return AGAN_EWORKSPACE;
    }
    hipStream_t st = as_stream(stream);
    float* part = static_cast<float*>(ws);
    dim3 grid(p.itiles, p.jtiles, p.ncls * p.psplit);
#define AGAN_WG(BI, BJ) hipLaunchKernelGGL((conv_wgrad_f32_kernel<BI, BJ>), grid, dim3(256), 0, st, x, dy, part, g, p.psplit, p.pchunk, p.slab)
    if (p.bi == 128 && p.bj == 128) AGAN_WG(128, 128);
    else if (p.bi == 128) AGAN_WG(128, 64);
    else if (p.bj == 128) AGAN_WG(64, 128);
    else AGAN_WG(64, 64);
#undef AGAN_WG
    if (int e = check_launch("conv_wgrad")) return e;
    if (pack_mode == AGAN_PACK_FWD) {
        dim3 ug(cdiv(g.K, 32), cdiv(g.Cout, 32));
        hipLaunchKernelGGL(unpack_wgrad_fwd_kernel, ug, dim3(256), 0, st, part, p.psplit, p.slab, dw, g.Cout, g.K, g.Nld);
    } else {
        const size_t total = (size_t)g.Cout * g.Cin * 9;
        hipLaunchKernelGGL(unpack_wgrad_up_kernel, dim3((unsigned)std::min<size_t>(cdivz(total, 256), 4096)), dim3(256), 0, st,
                           part, p.psplit, p.slab, dw, g.Cout, g.Cin, g.Nld);
    }
    return check_launch("conv_wgrad/unpack");
}

int agan_bias_grad(const float* dy, float* dbias, int B, int C, int HW, void* stream) {
    AGAN_REQUIRE(dy && dbias && B > 0 && C > 0 && HW > 0, "bias_grad: bad argument");
    hipLaunchKernelGGL(bias_grad_kernel, dim3(C), dim3(256), 0, as_stream(stream), dy, dbias, B, C, HW);
    return check_launch("bias_grad");
}

}  // extern "C"
