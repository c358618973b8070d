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
#include "conv_common.h"

#include <cxxabi.h>
#include <cstdlib>


using namespace agan;

using namespace agan::conv;

namespace {

// table layout: n entries {element offset incl. tap, packed (dy,dx)} for the weight-gradient kernels, followed by n entries
// {channel byte offset ci*IH*IW*4, byte offset of tap row t in the per-workgroup LDS tap table} for the gather kernels
__global__ __launch_bounds__(256) void ktable_kernel(int2* __restrict__ tab, int n, int K, int RS, int S, int IHW, int IW, int DY) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    int2 a, b;
    if (k < K) {
        const int c = k / RS, rs = k - c * RS, r = rs / S, q = rs - r * S;
        const int dy = r * DY, dx = q * DY;
        a.x = c * IHW + dy * IW + dx;
        a.y = (dy & 0xFFFF) | (dx << 16);
        b.x = c * IHW * 4;
        b.y = rs * kTapRowBytes;
    } else {
        a.x = 0;
        a.y = kSentinelDy & 0xFFFF;
        b.x = 0;
        b.y = RS * kTapRowBytes;          // the all-out-of-range row of the tap table
    }
    tab[k] = a;
    tab[n + k] = b;
}

// ================================================================================================
// forward / dgrad gather kernel
// ================================================================================================
// Build-time tuning / ablation switches (defaults are the shipped configuration; measurements in DESIGN.md §4):
//   AGAN_GATHER_BK     K tile depth;   AGAN_GATHER_WAVES   occupancy target of __launch_bounds__;
//   AGAN_ABLATE=1|2|3  drop the global loads | + the LDS stores | + the barrier of the K loop (wrong results: timing only)
#ifndef AGAN_XCD_REMAP
#define AGAN_XCD_REMAP 1
#endif
#ifndef AGAN_GATHER_BK
#define AGAN_GATHER_BK 16
#endif
#ifndef AGAN_GATHER_WAVES
#define AGAN_GATHER_WAVES 3
#endif
// ODT: storage type of `out` and of `lrelu_mask` (AGAN_DT_*; fp32 arithmetic either way -- a 16-bit output is the data gradient or
// the first-layer output of a network that keeps its activations in 16 bits: rounded once here instead of by a cast kernel)
template <int BM, int BN, int WM, int WN, int ODT = AGAN_DT_F32>
__global__ __launch_bounds__(256, AGAN_GATHER_WAVES) void conv_gather_f32_kernel(const float* __restrict__ in, const float* __restrict__ wk,
                                                               const float* __restrict__ bias, void* __restrict__ out_v,
                                                               const int2* __restrict__ ktab, const Geom g, const int ksplit,
                                                               const int kchunk, const size_t slab, const int act,
                                                               const void* __restrict__ lrelu_mask) {
    constexpr int BK = AGAN_GATHER_BK;
    constexpr int NG = 256 / BM;       // wave-uniform k groups for the pixel-major A loads
    constexpr int AK = BK / NG;        // k rows per thread per tile
    constexpr int NB4 = BK * BN / 4;   // float4s in a weight tile
    constexpr int BV = (NB4 + 255) / 256;
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    static_assert(WM * WN == 4 && TM >= 1 && TN >= 1, "bad wave tiling");

    __shared__ float As[2][BK][BM];
    __shared__ float Bs[2][BK][BN];
    // per-pixel byte offset of every tap (padding test folded in as an out-of-range offset), row RS = "always out of range":
    // computed once per workgroup, so the K loop does no address arithmetic beyond one LDS read per gathered element
    __shared__ unsigned Toff[kMaxTaps + 1][BM];
    static_assert(BM * 4 == kTapRowBytes, "tap table row size");

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wn = wave / WM;
    // XCD-aware tile order (conv_common.h: xcd_contiguous).  Pixel tiles that are neighbours in the image share input rows (the
    // 3x3 / 4x4 halo) and the channel tiles of one pixel tile share all of them: position F = (z, pixel tile, channel tile), so
    // each XCD takes a contiguous run of pixel tiles with all their channel tiles.  Measured on the 64->128 3x3 layer at
    // 128x128: 318 -> 109 MB fetched per launch (algorithmic 101 MB).
    int mt, nt, cls, split;
#if AGAN_XCD_REMAP
    {   // F = (((K split) * mtiles + pixel tile) * classes + parity class) * ntiles + channel tile: the parity classes of a pixel
        // tile (upsample conv, stride-2 dgrad) read the same input window, K splits read different channels
        const int ncls = gridDim.z / ksplit, mtiles = gridDim.x, ntiles = gridDim.y;
        int F = xcd_contiguous(linear_block_id(), mtiles * ntiles * (int)gridDim.z);
        nt = F % ntiles; F /= ntiles;
        cls = F % ncls;  F /= ncls;
        mt = F % mtiles; split = F / mtiles;
    }
#else
    mt = blockIdx.x; nt = blockIdx.y;
    cls = blockIdx.z / ksplit; split = blockIdx.z - cls * ksplit;
#endif
    const int py = cls / g.OS, px = cls - py * g.OS;
    const int m0 = mt * BM, n0 = nt * BN;
    const int kbeg = split * kchunk, kend = min(g.K, kbeg + kchunk);
    const int nkt = cdiv(kend - kbeg, BK);
    const int ihw = g.IH * g.IW;

    const __amdgpu_buffer_rsrc_t rin = make_rsrc(in, (size_t)g.B * g.Cin * ihw * sizeof(float));
    const __amdgpu_buffer_rsrc_t rwk = make_rsrc(wk + (size_t)cls * g.K * g.Nld, (size_t)g.K * g.Nld * sizeof(float));

    // ---- per-thread pixel for the A gather -------------------------------------------------------
    const int am = tid % BM;
    const int akg = __builtin_amdgcn_readfirstlane(tid / BM);
    const int m = m0 + am;
    const bool mvalid = m < g.Mtot;
    {
        const int mm = mvalid ? m : 0;
        const int b = g.dHWs.div(mm), rem = mm - b * g.HWs;
        const int yq = g.dOWs.div(rem), xq = rem - yq * g.OWs;
        const int iy0 = yq * g.SY + (py ? g.OY1 : g.OY0), ix0 = xq * g.SY + (px ? g.OY1 : g.OY0);
        const int pix0 = b * g.Cin * ihw + iy0 * g.IW + ix0;
        int r = 0, q = akg;                       // this thread fills taps akg, akg + NG, ... (q may start beyond S: normalise)
        while (q >= g.S) { q -= g.S; ++r; }
        for (int t = akg; t < g.RS; t += NG) {
            const int dy = r * g.DY, dx = q * g.DY;
            const bool ok = mvalid & ((unsigned)(iy0 + dy) < (unsigned)g.IH) & ((unsigned)(ix0 + dx) < (unsigned)g.IW);
            Toff[t][am] = ok ? (unsigned)(pix0 + dy * g.IW + dx) * 4u : kOOB;
            q += NG;
            while (q >= g.S) { q -= g.S; ++r; }
        }
        if (akg == 0) Toff[g.RS][am] = kOOB;
    }
    __syncthreads();
    const char* toff_lane = reinterpret_cast<const char*>(&Toff[0][am]);
    const int2* ktb = ktab + ktable_entries(g.K);        // second half of the table: {channel byte offset, tap row byte offset}
    // B tile mapping
    int bkr[BV], bnc[BV];
#pragma unroll
    for (int j = 0; j < BV; ++j) {
        const int f = tid + j * 256;
        bkr[j] = f / (BN / 4);
        bnc[j] = (f - bkr[j] * (BN / 4)) * 4;
    }

    // Register staging is two tiles deep: while tile kt is multiplied out of LDS, tile kt+1 is on its way from registers to
    // the other LDS buffer and the gathers of tile kt+2 are in flight, so the LDS store never waits on HBM/L2 latency.
    float areg[2][AK];
    f32x4 breg[2][BV];

    auto load_tile = [&](auto set, int kt) {
        constexpr int P = decltype(set)::value;
        const int kb = kbeg + kt * BK;
        // this wave's AK table entries in one wide scalar load (wave-uniform address, 16-byte aligned)
        const int4* tk4 = reinterpret_cast<const int4*>(ktb + kb + akg * AK);
        int te[2 * AK];
#pragma unroll
        for (int i = 0; i < AK / 2; ++i) {
            const int4 q = tk4[i];
            te[4 * i] = q.x; te[4 * i + 1] = q.y; te[4 * i + 2] = q.z; te[4 * i + 3] = q.w;
        }
#pragma unroll
        for (int i = 0; i < AK; ++i) {
            const unsigned voff = *reinterpret_cast<const unsigned*>(toff_lane + te[2 * i + 1]);
            areg[P][i] = buf_load_s(rin, voff, (unsigned)te[2 * i]);
        }
#pragma unroll
        for (int j = 0; j < BV; ++j) {
            const int kk = kb + bkr[j], n = n0 + bnc[j];
            const bool ok = (BV * 256 == NB4 || tid + j * 256 < NB4) & (kk < kend) & (n < g.Nld);
            breg[P][j] = buf_load4(rwk, ok ? (unsigned)(kk * g.Nld + n) * 4u : kOOB);
        }
    };
    auto store_tile = [&](auto set, int buf) {
        constexpr int P = decltype(set)::value;
#pragma unroll
        for (int i = 0; i < AK; ++i) As[buf][akg * AK + i][am] = areg[P][i];
#pragma unroll
        for (int j = 0; j < BV; ++j)
            if (BV * 256 == NB4 || tid + j * 256 < NB4) *reinterpret_cast<f32x4*>(&Bs[buf][bkr[j]][bnc[j]]) = breg[P][j];
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
    // one K tile: `cur` = register set that is free (tile kt already sits in LDS buffer `buf`), `nxt` = set holding tile kt+1
    // FULL = steady state (no conditionals, so the compiler's waitcnt bookkeeping stays exact across iterations)
    auto step = [&](auto full, auto cur, auto nxt, int kt, int buf) {
        constexpr bool FULL = decltype(full)::value;
#if !defined(AGAN_ABLATE) || AGAN_ABLATE < 1
        if (FULL || kt + 2 < nkt) load_tile(cur, kt + 2);
#endif
        // operand fragments are read one k-step ahead of the MFMAs that consume them (LDS latency off the critical path)
        float av[2][TM], bv[2][TN];
#pragma unroll
        for (int t = 0; t < TM; ++t) av[0][t] = As[buf][lh][wm * WTM + t * 32 + l31];
#pragma unroll
        for (int t = 0; t < TN; ++t) bv[0][t] = Bs[buf][lh][wn * WTN + t * 32 + l31];
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const int c = (kk >> 1) & 1;
            if (kk + 2 < BK) {
#pragma unroll
                for (int t = 0; t < TM; ++t) av[c ^ 1][t] = As[buf][kk + 2 + lh][wm * WTM + t * 32 + l31];
#pragma unroll
                for (int t = 0; t < TN; ++t) bv[c ^ 1][t] = Bs[buf][kk + 2 + lh][wn * WTN + t * 32 + l31];
            }
#pragma unroll
            for (int a = 0; a < TN; ++a)
#pragma unroll
                for (int b = 0; b < TM; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[c][a], av[c][b], acc[a][b], 0, 0, 0);
            // keep the next step's LDS reads ahead of this step's MFMAs in the emitted order
            __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);
        }
#if !defined(AGAN_ABLATE) || AGAN_ABLATE < 2
        if (FULL || kt + 1 < nkt) store_tile(nxt, buf ^ 1);
#endif
#if !defined(AGAN_ABLATE) || AGAN_ABLATE < 3
        __syncthreads();
#endif
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

    // ---- epilogue: D[cout][pixel]; lane owns one pixel column, 16 registers = 16 output channels ---
    const size_t ohw = (size_t)g.OH * g.OW;
    // (a split launch writes fp32 partial slabs whatever ODT is; the slab sum rounds)
    const bool typed = (ODT != AGAN_DT_F32) && (ksplit == 1);
    const unsigned esz = typed ? 2u : 4u;
    float* dst = (ksplit > 1) ? static_cast<float*>(out_v) + (size_t)split * slab : static_cast<float*>(out_v);
    const __amdgpu_buffer_rsrc_t rout = make_rsrc(dst, (size_t)g.B * g.Cout * ohw * esz);
    const bool add_bias = (bias != nullptr) && (ksplit == 1);
    const bool lrelu = (act == AGAN_ACT_LRELU) && (ksplit == 1);          // fused LeakyReLU(0.2) (a split launch applies it in the slab sum)
    // data-gradient launches: out *= LeakyReLU'(mask) where mask is the tensor the output is the gradient OF (same shape) -- the
    // backward of the activation that produced this conv's input, folded into the epilogue instead of a separate pass
    const bool masked = (lrelu_mask != nullptr) && (ksplit == 1);
    const __amdgpu_buffer_rsrc_t rmask = make_rsrc(masked ? lrelu_mask : (const void*)dst, (size_t)g.B * g.Cout * ohw * esz);
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
                if (lrelu) v = v > 0.f ? v : 0.2f * v;
                const unsigned off = (pvalid & (n < g.Cout)) ? (pixoff + (unsigned)n * (unsigned)ohw) * esz : kOOB;
                if (ODT != AGAN_DT_F32 && typed) {
                    if (masked) {
                        const unsigned short mh = (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rmask, off, 0, 0);
                        const float mv = ODT == AGAN_DT_BF16 ? __uint_as_float((unsigned)mh << 16) : (float)__builtin_bit_cast(_Float16, mh);
                        v = mv > 0.f ? v : 0.2f * v;
                    }
                    unsigned short h;
                    if (ODT == AGAN_DT_BF16) h = __builtin_bit_cast(unsigned short, (__bf16)v);
                    else h = __builtin_bit_cast(unsigned short, (_Float16)v);
                    __builtin_amdgcn_raw_buffer_store_b16((short)h, rout, off, 0, 0);
                } else {
                    if (masked) v = buf_load(rmask, off) > 0.f ? v : 0.2f * v;
                    buf_store(rout, off, v);
                }
            }
        }
    }
}

// out[i] = sum_s ws[s][i] (+ bias[channel]).  A workgroup covers 32 float4 elements x 8 split groups: each thread adds every
// 8th slab (4 independent loads in flight), the 8 partial sums meet in LDS.  Deterministic (fixed summation order).
// ODT: storage type of out (and of lrelu_mask); accumulate needs AGAN_DT_F32.
template <int ODT>
__global__ __launch_bounds__(256) void sum_slabs_kernel_t(const float* __restrict__ ws, int nsplit, size_t n, size_t slab,
                                                          const float* __restrict__ bias, int C, int HW,
                                                          void* __restrict__ out, int accumulate, int act = AGAN_ACT_NONE,
                                                          const void* __restrict__ lrelu_mask = nullptr, float* __restrict__ amax = nullptr) {
    __shared__ float4 part[8][32];
    float mx = 0.f;
    const size_t n4 = n / 4;
    const int e = threadIdx.x & 31, grp = threadIdx.x >> 5;
    for (size_t base = (size_t)blockIdx.x * 32; base < n4; base += (size_t)gridDim.x * 32) {
        const size_t i = base + e;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < n4) {
            const float* p = ws + i * 4;
            int sp = grp;
            for (; sp + 24 < nsplit; sp += 32) {
                const float4 v0 = *reinterpret_cast<const float4*>(p + (size_t)sp * slab);
                const float4 v1 = *reinterpret_cast<const float4*>(p + (size_t)(sp + 8) * slab);
                const float4 v2 = *reinterpret_cast<const float4*>(p + (size_t)(sp + 16) * slab);
                const float4 v3 = *reinterpret_cast<const float4*>(p + (size_t)(sp + 24) * slab);
                a.x += (v0.x + v1.x) + (v2.x + v3.x); a.y += (v0.y + v1.y) + (v2.y + v3.y);
                a.z += (v0.z + v1.z) + (v2.z + v3.z); a.w += (v0.w + v1.w) + (v2.w + v3.w);
            }
            for (; sp < nsplit; sp += 8) {
                const float4 v = *reinterpret_cast<const float4*>(p + (size_t)sp * slab);
                a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
            }
        }
        part[grp][e] = a;
        __syncthreads();
        if (grp == 0 && i < n4) {
#pragma unroll
            for (int k = 1; k < 8; ++k) {
                const float4 v = part[k][e];
                a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
            }
            if (bias) {
                const size_t q = i * 4;
                a.x += bias[(q / HW) % C]; a.y += bias[((q + 1) / HW) % C];
                a.z += bias[((q + 2) / HW) % C]; a.w += bias[((q + 3) / HW) % C];
            }
            if (accumulate) {
                const float4 o = ld4<ODT>(out, i * 4);
                a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w;
            }
            if (act == AGAN_ACT_LRELU) {
                a.x = a.x > 0.f ? a.x : 0.2f * a.x; a.y = a.y > 0.f ? a.y : 0.2f * a.y;
                a.z = a.z > 0.f ? a.z : 0.2f * a.z; a.w = a.w > 0.f ? a.w : 0.2f * a.w;
            }
            if (lrelu_mask) {
                const float4 m = ld4<ODT>(lrelu_mask, i * 4);
                a.x = m.x > 0.f ? a.x : 0.2f * a.x; a.y = m.y > 0.f ? a.y : 0.2f * a.y;
                a.z = m.z > 0.f ? a.z : 0.2f * a.z; a.w = m.w > 0.f ? a.w : 0.2f * a.w;
            }
            st4<ODT>(out, i * 4, a);
            mx = fmaxf(mx, fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))));
        }
        __syncthreads();
    }
    // tail (n not a multiple of 4)
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const size_t q = n4 * 4 + threadIdx.x;
        float a = 0.f;
        for (int sp = 0; sp < nsplit; ++sp) a += ws[(size_t)sp * slab + q];
        if (bias) a += bias[(q / HW) % C];
        if (accumulate) a += ld1<ODT>(out, q);
        if (act == AGAN_ACT_LRELU) a = a > 0.f ? a : 0.2f * a;
        if (lrelu_mask) a = ld1<ODT>(lrelu_mask, q) > 0.f ? a : 0.2f * a;
        st1<ODT>(out, q, a);
        mx = fmaxf(mx, fabsf(a));
    }
    if (amax) amax_commit(mx, amax);
}
// The same sum for FEW slabs (nsplit <= 8: every K split of a gather, most pixel splits of the large weight gradients).  The kernel above
// gives each of its 8 thread groups every 8th slab: with 2-4 slabs three quarters of a workgroup load nothing and every 512 bytes of a slab
// cost two barriers -- 0.3-1.1 TB/s on the 6-75 MB sums that make up two thirds of the step's slab time (round-4 kernel trace).  Here a thread
// owns one float4 of the result, has all its slab loads in flight at once and adds them in slab order -- the order the grouped kernel
// produces for nsplit <= 8, so results are bit-identical.
template <int ODT>
__global__ __launch_bounds__(256) void sum_slabs_few_kernel_t(const float* __restrict__ ws, int nsplit, size_t n, size_t slab,
                                                              const float* __restrict__ bias, int C, int HW,
                                                              void* __restrict__ out, int accumulate, int act = AGAN_ACT_NONE,
                                                              const void* __restrict__ lrelu_mask = nullptr, float* __restrict__ amax = nullptr) {
    float mx = 0.f;
    const size_t n4 = n / 4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float* p = ws + i * 4;
        float4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (k < nsplit) {
                const f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + (size_t)k * slab));      // (read exactly once)
                v[k] = make_float4(t[0], t[1], t[2], t[3]);
            }
        float4 a = v[0];
#pragma unroll
        for (int k = 1; k < 8; ++k)
            if (k < nsplit) { a.x += v[k].x; a.y += v[k].y; a.z += v[k].z; a.w += v[k].w; }
        if (bias) {
            const size_t q = i * 4;
            a.x += bias[(q / HW) % C]; a.y += bias[((q + 1) / HW) % C];
            a.z += bias[((q + 2) / HW) % C]; a.w += bias[((q + 3) / HW) % C];
        }
        if (accumulate) {
            const float4 o = ld4<ODT>(out, i * 4);
            a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w;
        }
        if (act == AGAN_ACT_LRELU) {
            a.x = a.x > 0.f ? a.x : 0.2f * a.x; a.y = a.y > 0.f ? a.y : 0.2f * a.y;
            a.z = a.z > 0.f ? a.z : 0.2f * a.z; a.w = a.w > 0.f ? a.w : 0.2f * a.w;
        }
        if (lrelu_mask) {
            const float4 m = ld4<ODT>(lrelu_mask, i * 4);
            a.x = m.x > 0.f ? a.x : 0.2f * a.x; a.y = m.y > 0.f ? a.y : 0.2f * a.y;
            a.z = m.z > 0.f ? a.z : 0.2f * a.z; a.w = m.w > 0.f ? a.w : 0.2f * a.w;
        }
        st4<ODT>(out, i * 4, a);
        mx = fmaxf(mx, fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))));
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {      // tail (n not a multiple of 4)
        const size_t q = n4 * 4 + threadIdx.x;
        float a = 0.f;
        for (int sp = 0; sp < nsplit; ++sp) a += ws[(size_t)sp * slab + q];
        if (bias) a += bias[(q / HW) % C];
        if (accumulate) a += ld1<ODT>(out, q);
        if (act == AGAN_ACT_LRELU) a = a > 0.f ? a : 0.2f * a;
        if (lrelu_mask) a = ld1<ODT>(lrelu_mask, q) > 0.f ? a : 0.2f * a;
        st1<ODT>(out, q, a);
        mx = fmaxf(mx, fabsf(a));
    }
    if (amax) amax_commit(mx, amax);
}

// out = sum of nsplit slabs (+ bias, + out, activation, mask), ODT = storage type of out / lrelu_mask
template <int ODT>
void launch_sum_slabs_t(const float* ws, int nsplit, size_t n, size_t slab, const float* bias, int C, int HW, void* out, int accumulate,
                        int act, const void* lrelu_mask, float* amax, hipStream_t st) {
    static const bool few_off = getenv("AGAN_SUM_FEW_OFF") != nullptr;          // (A/B switch)
    if (nsplit <= 8 && !few_off) {
        const int blocks = (int)std::min<size_t>(cdivz(n / 4 + 1, 256), 32768);
        AGAN_LAUNCH(sum_slabs_few_kernel_t<ODT>, dim3(blocks), dim3(256), 0, st, ws, nsplit, n, slab, bias, C, HW, out, accumulate, act,
                    lrelu_mask, amax);
    } else {
        const int blocks = (int)std::min<size_t>(cdivz(n / 4 + 1, 32), 8192);
        AGAN_LAUNCH(sum_slabs_kernel_t<ODT>, dim3(blocks), dim3(256), 0, st, ws, nsplit, n, slab, bias, C, HW, out, accumulate, act,
                    lrelu_mask, amax);
    }
}
void launch_sum_slabs(const float* ws, int nsplit, size_t n, size_t slab, const float* bias, int C, int HW, void* out, int accumulate,
                      int act, const void* lrelu_mask, float* amax, hipStream_t st, int out_dtype = AGAN_DT_F32) {
    if (out_dtype == AGAN_DT_F32) launch_sum_slabs_t<AGAN_DT_F32>(ws, nsplit, n, slab, bias, C, HW, out, accumulate, act, lrelu_mask, amax, st);
    else if (out_dtype == AGAN_DT_BF16) launch_sum_slabs_t<AGAN_DT_BF16>(ws, nsplit, n, slab, bias, C, HW, out, accumulate, act, lrelu_mask, amax, st);
    else launch_sum_slabs_t<AGAN_DT_F16>(ws, nsplit, n, slab, bias, C, HW, out, accumulate, act, lrelu_mask, amax, st);
}

template <int BN, int WM, int WN>
void launch_gather(const float* in, const float* wk, const float* bias, void* dst, const int2* ktab, const Geom& g,
                   const GatherPlan& p, int act, const void* lrelu_mask, hipStream_t st, int out_dtype = AGAN_DT_F32) {
    dim3 grid(p.mtiles, p.ntiles, p.ncls * p.ksplit);
    if (out_dtype == AGAN_DT_BF16)
        AGAN_LAUNCH((conv_gather_f32_kernel<128, BN, WM, WN, AGAN_DT_BF16>), grid, dim3(256), 0, st, in, wk, bias, dst, ktab, g, p.ksplit,
                           p.kchunk, p.slab, act, lrelu_mask);
    else if (out_dtype == AGAN_DT_F16)
        AGAN_LAUNCH((conv_gather_f32_kernel<128, BN, WM, WN, AGAN_DT_F16>), grid, dim3(256), 0, st, in, wk, bias, dst, ktab, g, p.ksplit,
                           p.kchunk, p.slab, act, lrelu_mask);
    else
        AGAN_LAUNCH((conv_gather_f32_kernel<128, BN, WM, WN>), grid, dim3(256), 0, st, in, wk, bias, dst, ktab, g, p.ksplit,
                           p.kchunk, p.slab, act, lrelu_mask);
}

// ================================================================================================
// weight gradient:  dwk[cls][(ci,r,s)][co] = sum_pixels im2col(x)[pixel][(ci,r,s)] * dy[pixel][co]
// ================================================================================================
template <int BI, int BJ>
__global__ __launch_bounds__(256) void conv_wgrad_f32_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                              float* __restrict__ dst, const int2* __restrict__ ktab, const Geom g,
                                                              const int psplit, const int pchunk, const size_t slab, const int accumulate) {
    constexpr int BP = 32, LDP = 33;   // +1 pad: MFMA operand reads walk the row index across lanes
    constexpr int XR = BI / 8, YR = BJ / 8;   // rows per thread per tile
    constexpr int TI = BI / 64, TJ = BJ / 64;
    __shared__ float Xs[2][BI][LDP];
    __shared__ float Ys[2][BJ][LDP];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wave & 1, wj = wave >> 1;
    // XCD-aware order: the (k, cout) tiles of one pixel chunk read the same dY rows and overlapping X -- position
    // F = (pixel chunk z, cout tile, k tile), so all tiles of a chunk run on one XCD (the 64-channel 3x3 layers have 9 k tiles
    // per chunk: in plain id order they hit 8 different L2s and dY came over the fabric 5-6x)
    int it_, jt_, cls, split;
#if AGAN_XCD_REMAP
    {   // F = (((pixel split) * classes + parity class) * jtiles + cout tile) * itiles + k tile
        const int ncls = gridDim.z / psplit, itiles = gridDim.x, jtiles = gridDim.y;
        int F = xcd_contiguous(linear_block_id(), itiles * jtiles * (int)gridDim.z);
        it_ = F % itiles; F /= itiles;
        jt_ = F % jtiles; F /= jtiles;
        cls = F % ncls;   split = F / ncls;
    }
#else
    it_ = blockIdx.x; jt_ = blockIdx.y;
    cls = blockIdx.z / psplit; split = blockIdx.z - cls * psplit;
#endif
    const int py = cls / g.OS, px = cls - py * g.OS;
    const int i0 = it_ * BI, j0 = jt_ * BJ;
    const int pbeg = split * pchunk, pend = min(g.Mtot, pbeg + pchunk);
    const int npt = cdiv(pend - pbeg, BP);
    const int pl = lane & 31, half = lane >> 5;
    const int ihw = g.IH * g.IW, ohw = g.OH * g.OW;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(x, (size_t)g.B * g.Cin * ihw * sizeof(float));
    const __amdgpu_buffer_rsrc_t rdy = make_rsrc(dy, (size_t)g.B * g.Cout * ohw * sizeof(float));
    // Per pixel tile, the byte offset of every (pixel, tap) -- padding and the pixel tail folded in as kWOOB -- and of the
    // pixel's dY element are computed ONCE by the workgroup into a small LDS table, two tiles ahead of the loads that use them;
    // a staged element then costs one LDS read and one add instead of a division-free but still ~10-instruction address chain.
    __shared__ unsigned Toff[2][kMaxTaps + 1][BP];
    __shared__ unsigned Doff[2][BP];
    // the reduction rows this thread stages never change across the pixel loop: fetch their table entries once
    // (second table half: {channel byte offset, tap row * 512}); rows past K read the all-out-of-range tap row
    unsigned xcoff[XR];
    int xtrow[XR];
#pragma unroll
    for (int ii = 0; ii < XR; ++ii) {
        const int2 e = ktab[ktable_entries(g.K) + i0 + wave * (BI / 4) + 2 * ii + half];
        xcoff[ii] = (unsigned)e.x;
        xtrow[ii] = (e.y / kTapRowBytes) * BP + pl;        // word index into Toff[buf]
    }
    unsigned ynoff[YR];
#pragma unroll
    for (int jj = 0; jj < YR; ++jj)       // output channels past Cout only feed accumulator rows the epilogue never stores: clamp
        ynoff[jj] = (unsigned)(min(j0 + wave * (BJ / 4) + half + 2 * jj, g.Cout - 1) * ohw) * 4u;
    // table fill: thread -> pixel tid & 31, taps tid >> 5 and (tid >> 5) + 8
    const int ft0 = tid >> 5, ft1 = ft0 + 8;
    const int frow0 = min(ft0, g.RS), frow1 = min(ft1, g.RS);
    int fdy0, fdx0, fdy1, fdx1;
    {
        const int r0 = ft0 / g.S, r1 = ft1 / g.S;
        fdy0 = r0 * g.DY; fdx0 = (ft0 - r0 * g.S) * g.DY;
        fdy1 = r1 * g.DY; fdx1 = (ft1 - r1 * g.S) * g.DY;
    }
    auto fill_offsets = [&](int pt) {
        const int b2 = pt & 1;
        const int p = pbeg + pt * BP + pl;
        const bool pvalid = p < pend;
        const int pp = pvalid ? p : 0;
        const int b = g.dHWs.div(pp), rem = pp - b * g.HWs;
        const int yq = g.dOWs.div(rem), xq = rem - yq * g.OWs;
        const int iy0 = yq * g.SY + (py ? g.OY1 : g.OY0), ix0 = xq * g.SY + (px ? g.OY1 : g.OY0);
        const int pix0 = b * g.Cin * ihw + iy0 * g.IW + ix0;
        // branch-free (the step stays one basic block for the scheduler): a tap index past RS writes kWOOB into the spare row RS,
        // and all eight tap groups store the same dY offset
        const bool ok0 = pvalid & (ft0 < g.RS) & ((unsigned)(iy0 + fdy0) < (unsigned)g.IH) & ((unsigned)(ix0 + fdx0) < (unsigned)g.IW);
        const bool ok1 = pvalid & (ft1 < g.RS) & ((unsigned)(iy0 + fdy1) < (unsigned)g.IH) & ((unsigned)(ix0 + fdx1) < (unsigned)g.IW);
        Toff[b2][frow0][pl] = ok0 ? (unsigned)(pix0 + fdy0 * g.IW + fdx0) * 4u : kWOOB;
        Toff[b2][frow1][pl] = ok1 ? (unsigned)(pix0 + fdy1 * g.IW + fdx1) * 4u : kWOOB;
        // (or-ing the marker in keeps this a select of constants: a value select makes the compiler branch around the multiplies,
        // and a second basic block in the loop costs exact wait counts)
        Doff[b2][pl] = ((unsigned)(b * g.Cout * ohw + (yq * g.OS + py) * g.OW + (xq * g.OS + px)) * 4u) | (pvalid ? 0u : kWOOB);
    };
    if (tid < 2 * BP) Toff[tid >> 5][g.RS][pl] = kWOOB;

    float xreg[2][XR], yreg[2][YR];     // two pixel tiles in flight (see the gather kernel)

    auto load_tile = [&](auto set, int pt) {
        constexpr int P = decltype(set)::value;
        const unsigned* to = &Toff[pt & 1][0][0];
        const unsigned doff = Doff[pt & 1][pl];
#pragma unroll
        for (int ii = 0; ii < XR; ++ii) xreg[P][ii] = buf_load(rx, to[xtrow[ii]] + xcoff[ii]);
#pragma unroll
        for (int jj = 0; jj < YR; ++jj) yreg[P][jj] = buf_load(rdy, doff + ynoff[jj]);
    };
    auto store_tile = [&](auto set, int buf) {
        constexpr int P = decltype(set)::value;
#pragma unroll
        for (int ii = 0; ii < XR; ++ii) Xs[buf][wave * (BI / 4) + 2 * ii + half][pl] = xreg[P][ii];
#pragma unroll
        for (int jj = 0; jj < YR; ++jj) Ys[buf][wave * (BJ / 4) + 2 * jj + half][pl] = yreg[P][jj];
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
    // One pixel tile.  The staging work of the step -- gathers of tile pt+2 into the free register set, LDS stores of tile pt+1
    // from the other set -- is spread over the 16 MFMA k-steps instead of sitting in front of / behind them: with two waves per
    // SIMD there is not enough parallelism to hide a serial address+load prologue behind the other wave's MFMAs.
    auto step = [&](auto full, auto cur, auto nxt, int pt, int buf) {
        constexpr bool FULL = decltype(full)::value;
        constexpr int PC = decltype(cur)::value, PN = decltype(nxt)::value;
        constexpr int NS = BP / 2, L = XR + YR;
        const bool do_load = FULL || pt + 2 < npt, do_store = FULL || pt + 1 < npt;
        unsigned xo[XR];
        const unsigned* to = &Toff[pt & 1][0][0];            // table of tile pt+2 (same parity)
#pragma unroll
        for (int ii = 0; ii < XR; ++ii) xo[ii] = to[xtrow[ii]] + xcoff[ii];
        const unsigned doff = Doff[pt & 1][pl];
        fill_offsets(pt + 3);           // past the last tile every entry is kWOOB: harmless, so never conditional
        float av[2][TI], bv[2][TJ];
#pragma unroll
        for (int t = 0; t < TI; ++t) av[0][t] = Xs[buf][wi * (BI / 2) + t * 32 + pl][half];
#pragma unroll
        for (int t = 0; t < TJ; ++t) bv[0][t] = Ys[buf][wj * (BJ / 2) + t * 32 + pl][half];
#pragma unroll
        for (int pp = 0; pp < BP; pp += 2) {
            const int c = (pp >> 1) & 1, q = pp >> 1;
            if (pp + 2 < BP) {
#pragma unroll
                for (int t = 0; t < TI; ++t) av[c ^ 1][t] = Xs[buf][wi * (BI / 2) + t * 32 + pl][pp + 2 + half];
#pragma unroll
                for (int t = 0; t < TJ; ++t) bv[c ^ 1][t] = Ys[buf][wj * (BJ / 2) + t * 32 + pl][pp + 2 + half];
            }
#pragma unroll
            for (int a = 0; a < TJ; ++a)
#pragma unroll
                for (int b = 0; b < TI; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[c][a], av[c][b], acc[a][b], 0, 0, 0);
#pragma unroll
            for (int it = q * L / NS; it < (q + 1) * L / NS; ++it) {
                if (it < XR) {
                    if (do_store) Xs[buf ^ 1][wave * (BI / 4) + 2 * it + half][pl] = xreg[PN][it];
                    if (do_load) xreg[PC][it] = buf_load(rx, xo[it]);
                } else {
                    const int jj = it - XR;
                    if (do_store) Ys[buf ^ 1][wave * (BJ / 4) + 2 * jj + half][pl] = yreg[PN][jj];
                    if (do_load) yreg[PC][jj] = buf_load(rdy, doff + ynoff[jj]);
                }
            }
        }
        __syncthreads();
    };

    fill_offsets(0);
    fill_offsets(1);
    __syncthreads();
    if (npt > 0) load_tile(S0{}, 0);
    if (npt > 1) load_tile(S1{}, 1);
    __syncthreads();
    fill_offsets(2);
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

    // D[cout][k index]: the lane owns one k column, so each register stores 32 consecutive floats of an OIHW row
    float* o = dst + (size_t)split * slab + (size_t)cls * g.Cout * g.K;
    const __amdgpu_buffer_rsrc_t ro = make_rsrc(o, (size_t)g.Cout * g.K * sizeof(float));
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
        for (int ti = 0; ti < TI; ++ti) {
            const int i = i0 + wi * (BI / 2) + ti * 32 + pl;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = j0 + wj * (BJ / 2) + tj * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const unsigned off = ((i < g.K) & (n < g.Cout)) ? (unsigned)(n * g.K + i) * 4u : kOOB;
                float v = acc[tj][ti][r];
                if (accumulate) v += buf_load(ro, off);        // unsplit second use of a parameter: add into the flat gradient
                buf_store(ro, off, v);
            }
        }
}

// ================================================================================================
// weight packing (OIHW -> [cls][K][Nld]) and gradient unpacking ([split][cls][K][Nld] -> OIHW)
// ================================================================================================
// AGAN_PACK_FWD is a plain [cout][K] -> [K][Nld] transpose: 32x32 LDS tiles, coalesced on both sides.
// 64x64 tiles, 16 independent loads per thread in flight before the first LDS store (these passes are pure HBM streaming)
__device__ __forceinline__ void pack_fwd_tile(const float* __restrict__ w, float* __restrict__ wk, int cout, int K, int Nld, int bx, int by,
                                              float (*tile)[65]) {
    const int k0 = bx * 64, n0 = by * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    float v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int n = n0 + ty + j * 4, k = k0 + tx;
        v[j] = (n < cout && k < K) ? w[(size_t)n * K + k] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) tile[ty + j * 4][tx] = v[j];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int k = k0 + ty + j * 4, n = n0 + tx;
        if (k < K && n < Nld) wk[(size_t)k * Nld + n] = tile[tx][ty + j * 4];
    }
}
__global__ __launch_bounds__(256) void pack_fwd_tiled_kernel(const float* __restrict__ w, float* __restrict__ wk, int cout, int K, int Nld) {
    __shared__ float tile[64][65];
    pack_fwd_tile(w, wk, cout, K, Nld, blockIdx.x, blockIdx.y, tile);
}

// dgrad packs: wk[cls][(co, tap')][ci] = w[co][ci][tap(cls, tap')] -- per output channel co a [cin][T] -> [T'][cin] shuffle.
// One workgroup stages a 64-input-channel slab of one co (64*T contiguous floats) in LDS and writes rows of 64 contiguous ci.
constexpr int kPackCo = 4;                  // output channels per workgroup of the dgrad packs (loads of all four in flight)
constexpr int kPackSlab = 64 * 16 + 64;     // one (co, 64 ci) slab: 64 x T floats, rows padded by one
template <int MODE>
__device__ __forceinline__ void pack_dgrad_tile(const float* __restrict__ w, float* __restrict__ wk, int cout, int cin, int kh, int kw, int Nld,
                                                int cog, int cblk, float* sl /* kPackCo * kPackSlab floats */, bool zero_pad) {
    const int T = kh * kw;
    const int c0 = cblk * 64;
    const int nc = min(64, cin - c0);
#pragma unroll
    for (int q = 0; q < kPackCo; ++q) {
        const int co = cog * kPackCo + q;
        if (co < cout) {
            const float* src = w + ((size_t)co * cin + c0) * T;
            for (int i = threadIdx.x; i < nc * T; i += 256) sl[q * kPackSlab + (i / T) * (T + 1) + (i % T)] = src[i];   // +1 pad: column walks below
        }
    }
    __syncthreads();
    // columns [cin, Nld) of this output channel's rows are padding; the batched path clears them itself (last channel block)
    const int cend = (zero_pad && c0 + 64 >= cin) ? Nld - c0 : nc;
#pragma unroll
    for (int q = 0; q < kPackCo; ++q) {
        const int co = cog * kPackCo + q;
        if (co >= cout) break;
        const float* s1 = sl + q * kPackSlab;
        if (MODE == AGAN_PACK_DGRAD_S1) {
            // K index = (co, r, s); source tap = flipped
            for (int i = threadIdx.x; i < T * 64; i += 256) {
                const int t = i / 64, c = i - t * 64;
                if (c < cend) wk[((size_t)co * T + t) * Nld + c0 + c] = c < nc ? s1[c * (T + 1) + (T - 1 - t)] : 0.f;
            }
        } else {   // AGAN_PACK_DGRAD_4x4S2: 4 classes x (co, r, s in {0,1}); tap kh = ((py+1)&1) + 2r
            const int K = cout * 4;
            for (int i = threadIdx.x; i < 16 * 64; i += 256) {
                const int qq = i / 64, c = i - qq * 64;
                const int cls = qq >> 2, r = (qq >> 1) & 1, sx = qq & 1, py = cls >> 1, px = cls & 1;
                const int th = ((py + 1) & 1) + 2 * r, tw = ((px + 1) & 1) + 2 * sx;
                if (c < cend) wk[((size_t)cls * K + co * 4 + r * 2 + sx) * Nld + c0 + c] = c < nc ? s1[c * 17 + th * 4 + tw] : 0.f;
            }
        }
    }
}
template <int MODE>
__global__ __launch_bounds__(256) void pack_dgrad_tiled_kernel(const float* __restrict__ w, float* __restrict__ wk, int cout, int cin,
                                                               int kh, int kw, int Nld) {
    __shared__ float sl[kPackCo * kPackSlab];
    pack_dgrad_tile<MODE>(w, wk, cout, cin, kh, kw, Nld, blockIdx.x, blockIdx.y, sl, false);
}

__global__ __launch_bounds__(256) void pack_weight_kernel(const float* __restrict__ w, float* __restrict__ wk, int mode,
                                                          int cout, int cin, int kh, int kw, int K, int Nld, int ncls) {
    // one thread per packed element, n fastest (coalesced writes)
    const size_t total = (size_t)ncls * K * Nld;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int n = (int)(e % Nld);
        const size_t t = e / Nld;
        const int k = (int)(t % K), cls = (int)(t / K);
        wk[e] = packed_weight_value(w, mode, cls, k, n, cout, cin, kh, kw);
    }
}

// Every stale packed weight of one optimiser in ONE launch (agan_pack_weights): after an Adam step a module re-packs ~25-80
// tensors, most of them a few KB -- as separate launches they cost 10 us each whatever their size.
__host__ __device__ inline int pack_job_blocks(int mode, int cout, int cin, int kh, int kw) {
    int ncls, K, N;
    if (pack_dims(mode, cout, cin, kh, kw, ncls, K, N)) return 0;
    const int Nld = ((N + 31) / 32 * 32);
    if (mode == AGAN_PACK_FWD) return cdiv(K, 64) * cdiv(Nld, 64);
    if ((mode == AGAN_PACK_DGRAD_S1 && kh * kw <= 16) || mode == AGAN_PACK_DGRAD_4x4S2) return cdiv(cout, kPackCo) * cdiv(cin, 64);
    const size_t total = (size_t)ncls * K * Nld;
    return (int)(total / 1024 < 1 ? 1 : (total / 1024 > 2048 ? 2048 : total / 1024));
}

__global__ __launch_bounds__(256) void pack_jobs_kernel(const agan_pack_job* __restrict__ jobs, int njobs) {
    __shared__ float buf[kPackCo * kPackSlab > 64 * 65 ? kPackCo * kPackSlab : 64 * 65];
    __shared__ int which;
    const int b = blockIdx.x;
    for (int t = threadIdx.x; t < njobs; t += 256) {
        const int lo = jobs[t].first_block, hi = t + 1 < njobs ? jobs[t + 1].first_block : 0x7fffffff;
        if (b >= lo && b < hi) which = t;
    }
    __syncthreads();
    const agan_pack_job j = jobs[which];
    const int local = b - j.first_block;
    float* wk = static_cast<float*>(j.wk);
    int ncls, K, N;
    pack_dims(j.mode, j.cout, j.cin, j.kh, j.kw, ncls, K, N);
    const int Nld = ((N + 31) / 32 * 32);
    if (j.mode == AGAN_PACK_FWD) {
        const int kt = cdiv(K, 64);
        pack_fwd_tile(j.w, wk, j.cout, K, Nld, local % kt, local / kt, reinterpret_cast<float (*)[65]>(buf));
    } else if (j.mode == AGAN_PACK_DGRAD_S1 && j.kh * j.kw <= 16) {
        const int cg = cdiv(j.cout, kPackCo);
        pack_dgrad_tile<AGAN_PACK_DGRAD_S1>(j.w, wk, j.cout, j.cin, j.kh, j.kw, Nld, local % cg, local / cg, buf, true);
    } else if (j.mode == AGAN_PACK_DGRAD_4x4S2) {
        const int cg = cdiv(j.cout, kPackCo);
        pack_dgrad_tile<AGAN_PACK_DGRAD_4x4S2>(j.w, wk, j.cout, j.cin, j.kh, j.kw, Nld, local % cg, local / cg, buf, true);
    } else {
        const int nb = pack_job_blocks(j.mode, j.cout, j.cin, j.kh, j.kw);
        const size_t total = (size_t)ncls * K * Nld;
        for (size_t e = (size_t)local * 256 + threadIdx.x; e < total; e += (size_t)nb * 256) {
            const int n = (int)(e % Nld);
            const size_t t = e / Nld;
            wk[e] = packed_weight_value(j.w, j.mode, (int)(t / K), (int)(t % K), n, j.cout, j.cin, j.kh, j.kw);
        }
    }
}

// UP mode: dw[co][ci][a][b] = sum over the (class, tap) pairs that 3x3 tap (a,b) was folded into; dwk is [cls][co][(ci,r,s)]
__global__ __launch_bounds__(256) void unpack_wgrad_up_kernel(const float* __restrict__ dwk, float* __restrict__ dw, int cout, int cin,
                                                              int accumulate) {
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
                    for (int q = 0; q < 2; ++q) {
                        int sl, sh;
                        up_fwd_taps(px, q, sl, sh);
                        if (b < sl || b > sh) continue;
                        v += dwk[((size_t)(py * 2 + px) * cout + co) * K + (ci * 4 + r * 2 + q)];
                    }
            }
        dw[e] = accumulate ? dw[e] + v : v;
    }
}

__global__ __launch_bounds__(256) void bias_grad_kernel(const float* __restrict__ dy, float* __restrict__ db, int B, int C,
                                                        int HW, int accumulate) {
    __shared__ float red[4];
    const int c = blockIdx.x;
    float s = 0.f;
    for (int b = 0; b < B; ++b) {
        const float* p = dy + ((size_t)b * C + c) * HW;
        for (int i = threadIdx.x; i < HW; i += 256) s += p[i];
    }
    s = block_sum<256>(s, red);
    if (threadIdx.x == 0) db[c] = accumulate ? db[c] + s : s;
}

}  // namespace

// ---- measurement hook (bench.py roofline): one-shot HIP event pair recorded on the launch stream right around the NEXT
// main conv kernel (gather or weight gradient), excluding the slab-sum / unpack passes that follow a split launch
static thread_local hipEvent_t g_timer_start = nullptr, g_timer_stop = nullptr;
static thread_local const void* g_timed_kernel = nullptr;      // host stub of the main kernel of the last timed call (AGAN_LAUNCH)
namespace agan { namespace conv { thread_local const void* g_noted_kernel = nullptr; } }
static inline void timer_begin(hipStream_t st) {
    if (g_timer_start) (void)hipEventRecord(g_timer_start, st);
}
static inline void timer_end(hipStream_t st) {
    if (g_timer_stop) {
        (void)hipEventRecord(g_timer_stop, st);
        g_timed_kernel = agan::conv::g_noted_kernel;           // the LAST launch inside the bracket is the main kernel (a weight transform precedes it)
    }
    g_timer_start = g_timer_stop = nullptr;
}

// ================================================================================================
// C ABI
// ================================================================================================
extern "C" {

int agan_timer_last_kernel(char* name, size_t capacity) {
    AGAN_REQUIRE(name && capacity > 1, "timer_last_kernel: bad argument");
    name[0] = 0;
    if (!g_timed_kernel) return AGAN_OK;
    const char* mangled = hipKernelNameRefByPtr(g_timed_kernel, nullptr);
    if (!mangled) return AGAN_OK;
    int status = 0;
    char* dem = abi::__cxa_demangle(mangled, nullptr, nullptr, &status);
    const char* src = (status == 0 && dem) ? dem : mangled;
    strncpy(name, src, capacity - 1);
    name[capacity - 1] = 0;
    free(dem);
    return AGAN_OK;
}

int agan_timer_create(void** event) {
    AGAN_REQUIRE(event != nullptr, "timer_create: null pointer");
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) {
        set_error("timer_create: hipEventCreate failed");
        return AGAN_ELAUNCH;
    }
    *event = e;
    return AGAN_OK;
}
int agan_timer_destroy(void* event) { return (event && hipEventDestroy(static_cast<hipEvent_t>(event)) != hipSuccess) ? AGAN_ELAUNCH : AGAN_OK; }
int agan_timer_arm(void* start, void* stop) {
    g_timer_start = static_cast<hipEvent_t>(start);
    g_timer_stop = static_cast<hipEvent_t>(stop);
    return AGAN_OK;
}
int agan_timer_elapsed_ms(void* start, void* stop, float* ms) {
    AGAN_REQUIRE(start && stop && ms, "timer_elapsed_ms: null pointer");
    if (hipEventElapsedTime(ms, static_cast<hipEvent_t>(start), static_cast<hipEvent_t>(stop)) != hipSuccess) {
        set_error("timer_elapsed_ms: events not complete");
        return AGAN_ELAUNCH;
    }
    return AGAN_OK;
}

size_t agan_packed_weight_bytes(int mode, int cout, int cin, int kh, int kw, int prec) {
    int ncls, K, N;
    if (pack_dims(mode, cout, cin, kh, kw, ncls, K, N)) return 0;
    if (prec == AGAN_PREC_F32) return (size_t)ncls * K * ((N + 31) / 32 * 32) * sizeof(float);
    if (prec_planes(prec) > 0) return patch_packed_weight_bytes(mode, cout, cin, kh, kw, prec);
    return 0;
}

int agan_pack_weight(const float* w, void* wkv, int mode, int cout, int cin, int kh, int kw, int prec, void* stream) {
    int ncls, K, N;
    AGAN_REQUIRE(w && wkv, "pack_weight: null pointer");
    AGAN_REQUIRE(prec == AGAN_PREC_F32 || prec_planes(prec) > 0, "pack_weight: unknown precision mode %d", prec);
    if (prec != AGAN_PREC_F32) {
        AGAN_REQUIRE(pack_dims(mode, cout, cin, kh, kw, ncls, K, N) == 0, "pack_weight: mode %d does not take %dx%d", mode, kh, kw);
        return pack_weight_patch(w, wkv, mode, cout, cin, kh, kw, prec, as_stream(stream));
    }
    float* wk = static_cast<float*>(wkv);
    AGAN_REQUIRE(pack_dims(mode, cout, cin, kh, kw, ncls, K, N) == 0, "pack_weight: mode %d does not take %dx%d", mode, kh, kw);
    const int Nld = ((N + 31) / 32 * 32);
    const size_t total = (size_t)ncls * K * Nld;
    const int blocks = (int)std::min<size_t>(cdivz(total, 256), 8192);
    hipStream_t pst = as_stream(stream);
    if (mode == AGAN_PACK_FWD) {
        AGAN_LAUNCH(pack_fwd_tiled_kernel, dim3(cdiv(K, 64), cdiv(Nld, 64)), dim3(256), 0, pst, w, wk, cout, K, Nld);
        return check_launch("pack_weight/fwd");
    }
    if ((mode == AGAN_PACK_DGRAD_S1 && kh * kw <= 16) || mode == AGAN_PACK_DGRAD_4x4S2) {
        if (Nld != cin) (void)hipMemsetAsync(wk, 0, total * sizeof(float), pst);      // zero the N padding columns
        dim3 grid(cdiv(cout, kPackCo), cdiv(cin, 64));
        if (mode == AGAN_PACK_DGRAD_S1) AGAN_LAUNCH((pack_dgrad_tiled_kernel<AGAN_PACK_DGRAD_S1>), grid, dim3(256), 0, pst, w, wk, cout, cin, kh, kw, Nld);
        else AGAN_LAUNCH((pack_dgrad_tiled_kernel<AGAN_PACK_DGRAD_4x4S2>), grid, dim3(256), 0, pst, w, wk, cout, cin, kh, kw, Nld);
        return check_launch("pack_weight/dgrad");
    }
    AGAN_LAUNCH(pack_weight_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), w, wk, mode, cout, cin, kh, kw, K,
                       Nld, ncls);
    return check_launch("pack_weight");
}

int agan_pack_job_blocks(int mode, int cout, int cin, int kh, int kw) { return pack_job_blocks(mode, cout, cin, kh, kw); }
int agan_pack_job_blocks_prec(int mode, int cout, int cin, int kh, int kw, int prec) {
    return prec == AGAN_PREC_F32 ? pack_job_blocks(mode, cout, cin, kh, kw) : (prec_planes(prec) > 0 ? pack_job_blocks_patch(mode, cout, cin, kh, kw) : 0);
}

int agan_pack_weights(const agan_pack_job* jobs, int njobs, int total_blocks, int prec, void* stream) {
    AGAN_REQUIRE(jobs && njobs > 0 && total_blocks > 0, "pack_weights: empty job list");
    AGAN_REQUIRE(prec == AGAN_PREC_F32 || prec_planes(prec) > 0, "pack_weights: unknown precision mode %d", prec);
    if (prec != AGAN_PREC_F32) return pack_weights_patch(jobs, njobs, total_blocks, prec, as_stream(stream));
    AGAN_LAUNCH(pack_jobs_kernel, dim3(total_blocks), dim3(256), 0, as_stream(stream), jobs, njobs);
    return check_launch("pack_weights");
}

int agan_conv_effective_prec(const agan_conv_geom* g, int prec) {
    if (prec == AGAN_PREC_F32 || check_geom(g)) return AGAN_PREC_F32;
    if (prec_planes(prec) == 0) return AGAN_PREC_F32;
    return patch_supported(make_geom(g)) ? prec : AGAN_PREC_F32;
}

// the mode a weight gradient of forward geometry g runs in.  Besides the geometries the patch kernels do not take:
//   * AGAN_PREC_BF16X6 -- fp32-grade by definition, so it may use whichever fp32-grade kernel is fastest.  Its own six-product patch
//     kernel is not (three planes of two tensors: 10 vs 6.5 ms per step against the fp32 MFMA kernel; -DAGAN_BF16X6_PATCH_WGRAD=1
//     builds it in).  Round 3: the two-plane fp16 split (AGAN_PREC_F16X3: 22-bit products, the same full-size parity) is 1.35-2.2x
//     faster than the fp32 kernel on every layer, so bf16x6 weight gradients run there -- the caller supplies the amax slots of x
//     and dy exactly as in AGAN_PREC_F16X3 (AGAN_BF16X6_WGRAD_F32=1 in the environment restores the fp32 kernel);
//   * the folded upsample conv (4 classes of 2x2 taps): its patch weight gradient is slower than the fp32 kernel in EVERY mode
//     (0.61 / 0.52 vs 0.49 ms at 64x128 -> 64x256; 16 short reductions per tile) -- fp32 products are at least as accurate.
int agan_conv_wgrad_effective_prec(const agan_conv_geom* g, int pack_mode, int prec) {
    if (prec == AGAN_PREC_F32 || check_geom(g) || prec_planes(prec) == 0) return AGAN_PREC_F32;
    if (!patch_supported(make_geom(g))) return AGAN_PREC_F32;
#ifndef AGAN_BF16X6_PATCH_WGRAD
    if (prec == AGAN_PREC_BF16X6) {
        static const bool f32wg = getenv("AGAN_BF16X6_WGRAD_F32") != nullptr;
        prec = f32wg ? AGAN_PREC_F32 : AGAN_PREC_F16X3;
    }
#endif
    // (round 3: the row-resident kernel takes the upsample conv as the conv3x3 on the upsampled image it is -- conv_wgrows.hip)
    if (pack_mode == AGAN_PACK_UP_FWD && (prec == AGAN_PREC_F32 || !plan_rows_wgrad(make_geom(g), prec, false, false, true).ok)) return AGAN_PREC_F32;
    return prec;
}

double agan_conv_executed_fraction(const agan_conv_geom* g, int prec, int wgrad, int plain_epilogue) {
    if (check_geom(g) || prec != AGAN_PREC_F32) return 1.0;
    const Geom gg = make_geom(g);
    if (wgrad) return plan_wino_wgrad(gg).ok ? 16.0 / 36.0 : 1.0;
    if (!plain_epilogue) return 1.0;
    const WinoPlan wp = plan_wino(gg);
    if (!wp.ok) return 1.0;
    return wp.s2 == 0 ? 16.0 / 36.0 : (wp.s2 == 1 ? 36.0 / 64.0 : 9.0 / 16.0);
}

size_t agan_conv_gather_ws_bytes(const agan_conv_geom* g, int prec) {
    if (check_geom(g)) return 0;
    const Geom gg = make_geom(g);
    prec = agan_conv_effective_prec(g, prec);
    if (prec == AGAN_PREC_F32 && small_n_gather_supported(gg)) return 0;
    if (prec == AGAN_PREC_F32) {
        const WinoPlan wp = plan_wino(gg);
        if (wp.ok) return std::max(wp.ws_bytes, plan_gather(gg, prec).ws_bytes);      // (a call with a bias / activation epilogue takes the direct kernel)
    }
    if (prec != AGAN_PREC_F32) {
        size_t a = plan_patch_gather(gg, make_patch_plan(gg)).ws_bytes;
        if (prec == AGAN_PREC_BF16 || prec == AGAN_PREC_F16)          // the row-block gather may split differently (either storage type)
            for (int in16 = 0; in16 < 2; ++in16) {
                const P16Plan p = plan_p16(gg, in16 != 0);
                if (p.ok) a = std::max(a, p.ws_bytes);
            }
        return a;
    }
    return plan_gather(gg, prec).ws_bytes;
}

size_t agan_conv_ktable_elems(const agan_conv_geom* g) {
    if (check_geom(g)) return 0;
    return (size_t)ktable_entries(g->Cin * g->R * g->S) * 4;      // two int2 halves (see ktable_kernel)
}

int agan_conv_ktable(const agan_conv_geom* gg, int32_t* table, void* stream) {
    if (int e = check_geom(gg)) return e;
    AGAN_REQUIRE(table != nullptr, "conv_ktable: null pointer");
    const Geom g = make_geom(gg);
    const int n = ktable_entries(g.K);
    AGAN_LAUNCH(ktable_kernel, dim3(cdiv(n, 256)), dim3(256), 0, as_stream(stream), reinterpret_cast<int2*>(table), n, g.K, g.RS,
                       g.S, g.IH * g.IW, g.IW, g.DY);
    return check_launch("conv_ktable");
}

// does the row-block gather (conv_p16.hip) take this call?  (one-plane 16-bit modes; AGAN_P16_OFF=1 keeps fp32-storage calls on
// the patch-resident kernel for A/B measurements)
static bool p16_takes(const Geom& g, int prec, int in_dtype, int out_dtype, P16Plan* plan) {
    if (prec != AGAN_PREC_BF16 && prec != AGAN_PREC_F16) return false;
    const int want = prec == AGAN_PREC_BF16 ? AGAN_DT_BF16 : AGAN_DT_F16;
    if ((in_dtype != AGAN_DT_F32 && in_dtype != want) || (out_dtype != AGAN_DT_F32 && out_dtype != want)) return false;
    static const bool off = getenv("AGAN_P16_OFF") != nullptr;
    if (off && in_dtype == AGAN_DT_F32 && out_dtype == AGAN_DT_F32) return false;
    const P16Plan p = plan_p16(g, in_dtype != AGAN_DT_F32);
    if (plan) *plan = p;
    return p.ok != 0;
}

static bool dt16(int dt) { return dt == AGAN_DT_BF16 || dt == AGAN_DT_F16; }
// the fp32-arithmetic kernels with typed storage: the <= 4-output-channel kernels read a 16-bit input (fp32 output), the k-table
// MFMA kernels write a 16-bit output (fp32 input)
static bool f32_kernel_takes(const Geom& g, const agan_conv_geom* gg, int prec, int in_dtype, int out_dtype) {
    if (agan_conv_effective_prec(gg, prec) != AGAN_PREC_F32) return false;
    if (small_n_gather_supported(g)) return (in_dtype == AGAN_DT_F32 || dt16(in_dtype)) && out_dtype == AGAN_DT_F32 && !(g.RS == 1 && g.IH == 1 && g.IW == 1 && in_dtype != AGAN_DT_F32);
    return in_dtype == AGAN_DT_F32 && (out_dtype == AGAN_DT_F32 || dt16(out_dtype));
}

int agan_conv_gather_dt_supported(const agan_conv_geom* gg, int prec, int in_dtype, int out_dtype) {
    if (check_geom(gg)) return 0;
    if (in_dtype == AGAN_DT_F32 && out_dtype == AGAN_DT_F32) return 1;
    const Geom g = make_geom(gg);
    if (p16_takes(g, prec, in_dtype, out_dtype, nullptr)) return 1;
    return f32_kernel_takes(g, gg, agan_conv_effective_prec(gg, prec), in_dtype, out_dtype) ? 1 : 0;
}

int agan_conv_gather_dt(const void* in_v, const void* wkv, const float* bias, void* out_v, const agan_conv_geom* gg, const int32_t* ktable,
                        int prec, int act, const void* lrelu_mask_v, void* ws, size_t ws_bytes, void* stream, const float* in_scale,
                        float* out_amax, int in_dtype, int out_dtype) {
    const float* wk = static_cast<const float*>(wkv);
    if (int e = check_geom(gg)) return e;
    AGAN_REQUIRE(in_v && wk && out_v && ktable, "conv_gather: null pointer");
    AGAN_REQUIRE(act == AGAN_ACT_NONE || (act == AGAN_ACT_LRELU && gg->Cout > 4),
                 "conv_gather: fused activation %d not available for this call (MFMA paths, LeakyReLU only)", act);
    AGAN_REQUIRE(!lrelu_mask_v || (gg->Cout > 4 && act == AGAN_ACT_NONE),
                 "conv_gather: the LeakyReLU-derivative mask needs an MFMA path and no other activation");
    const int2* ktab = reinterpret_cast<const int2*>(ktable);
    AGAN_REQUIRE(prec == AGAN_PREC_F32 || prec_planes(prec) > 0, "conv_gather: unknown precision mode %d", prec);
    AGAN_REQUIRE(agan_conv_effective_prec(gg, prec) == prec,
                 "conv_gather: precision mode %d does not take this geometry (agan_conv_effective_prec says which one does)", prec);
    const Geom g = make_geom(gg);
    hipStream_t st = as_stream(stream);
    {
        P16Plan p16;
        if (p16_takes(g, prec, in_dtype, out_dtype, &p16)) {
            if (p16.ws_bytes > ws_bytes || (p16.ws_bytes && !ws)) {
                set_error("conv_gather: workspace %zu < %zu", ws_bytes, p16.ws_bytes);
                return AGAN_EWORKSPACE;
            }
            void* dst = p16.ksplit > 1 ? ws : out_v;
            timer_begin(st);
            launch_p16_gather(in_v, wkv, bias, dst, g, p16, prec, act, lrelu_mask_v, st, in_dtype != AGAN_DT_F32, out_dtype != AGAN_DT_F32);
            timer_end(st);
            if (int e = check_launch("conv_gather/p16")) return e;
            if (p16.ksplit > 1) {
                const size_t n = (size_t)g.B * g.Cout * g.OH * g.OW;
                launch_sum_slabs(static_cast<const float*>(ws), p16.ksplit, n, p16.slab, bias, g.Cout, g.OH * g.OW, out_v, 0, act, lrelu_mask_v,
                                 nullptr, st, out_dtype);
                return check_launch("conv_gather/p16/sum_slabs");
            }
            return AGAN_OK;
        }
    }
    AGAN_REQUIRE((in_dtype == AGAN_DT_F32 && out_dtype == AGAN_DT_F32) || f32_kernel_takes(g, gg, prec, in_dtype, out_dtype),
                 "conv_gather: 16-bit activation storage (%d -> %d) is not available for this geometry / precision %d "
                 "(agan_conv_gather_dt_supported)", in_dtype, out_dtype, prec);
    const float* in = static_cast<const float*>(in_v);
    float* out = static_cast<float*>(out_v);
    const float* lrelu_mask = static_cast<const float*>(lrelu_mask_v);
    if (prec != AGAN_PREC_F32) {
        const PatchPlan pp = make_patch_plan(g);
        const PatchGather p = plan_patch_gather(g, pp);
        if (p.ws_bytes > ws_bytes || (p.ws_bytes && !ws)) {
            set_error("conv_gather: workspace %zu < %zu", ws_bytes, p.ws_bytes);
            return AGAN_EWORKSPACE;
        }
        float* dst = p.ksplit > 1 ? static_cast<float*>(ws) : out;
        timer_begin(st);
        AGAN_REQUIRE(prec != AGAN_PREC_F16X3 || in_scale != nullptr, "conv_gather: AGAN_PREC_F16X3 needs the operand's agan_absmax_scale pair");
        launch_patch_gather(in, wkv, bias, dst, g, pp, p, prec, act, lrelu_mask, st, in_scale, out_amax);
        timer_end(st);
        if (int e = check_launch("conv_gather/patch")) return e;
        if (p.ksplit > 1) {
            const size_t n = (size_t)g.B * g.Cout * g.OH * g.OW;
            launch_sum_slabs(static_cast<const float*>(ws), p.ksplit, n, p.slab, bias, g.Cout, g.OH * g.OW, out, 0, act, lrelu_mask, out_amax, st);
            return check_launch("conv_gather/sum_slabs");
        }
        return AGAN_OK;
    }
    if (prec == AGAN_PREC_F32 && in_dtype == AGAN_DT_F32 && out_dtype == AGAN_DT_F32 && !bias && act == AGAN_ACT_NONE && !lrelu_mask_v) {
        // conv3x3 stride 1 (forward or data gradient) on enough pixels: Winograd F(2x2, 3x3), 2.25x fewer fp32 MFMAs (conv_wino.hip)
        const WinoPlan wp = plan_wino(g);
        if (wp.ok) {
            if (wp.ws_bytes > ws_bytes || !ws) {
                set_error("conv_gather: workspace %zu < %zu", ws_bytes, wp.ws_bytes);
                return AGAN_EWORKSPACE;
            }
            timer_begin(st);
            launch_wino(in, wk, out, g, wp, ws, st);
            timer_end(st);
            if (int e = check_launch("conv_gather/winograd")) return e;
            if (wp.ksplit > 1) {
                const size_t n = (size_t)g.B * g.Cout * g.OH * g.OW;
                launch_sum_slabs(reinterpret_cast<const float*>(static_cast<const char*>(ws) + wp.u_bytes), wp.ksplit, n, wp.slab, nullptr,
                                 g.Cout, g.OH * g.OW, out, 0, AGAN_ACT_NONE, nullptr, nullptr, st);
                return check_launch("conv_gather/winograd/sum_slabs");
            }
            return AGAN_OK;
        }
    }
    if (prec == AGAN_PREC_F32 && small_n_gather_supported(g)) {
        timer_begin(st);
        launch_gather_small_n(in_v, wk, bias, out, g, st, in_dtype);
        timer_end(st);
        return check_launch("conv_gather/small_n");
    }
    const GatherPlan p = plan_gather(g, prec);
    if (p.ws_bytes > ws_bytes || (p.ws_bytes && !ws)) {
        set_error("conv_gather: workspace %zu < %zu", ws_bytes, p.ws_bytes);
        return AGAN_EWORKSPACE;
    }
    void* dst = p.ksplit > 1 ? ws : out_v;
    timer_begin(st);
    if (p.bn == 128) launch_gather<128, 2, 2>(in, wk, bias, dst, ktab, g, p, act, lrelu_mask_v, st, out_dtype);
    else if (p.bn == 64) launch_gather<64, 2, 2>(in, wk, bias, dst, ktab, g, p, act, lrelu_mask_v, st, out_dtype);
    else launch_gather<32, 4, 1>(in, wk, bias, dst, ktab, g, p, act, lrelu_mask_v, st, out_dtype);
    timer_end(st);
    if (int e = check_launch("conv_gather")) return e;
    if (p.ksplit > 1) {
        const size_t n = (size_t)g.B * g.Cout * g.OH * g.OW;
        launch_sum_slabs(static_cast<const float*>(ws), p.ksplit, n, p.slab, bias, g.Cout, g.OH * g.OW, out_v, 0, act, lrelu_mask_v, nullptr, st,
                         out_dtype);
        return check_launch("conv_gather/sum_slabs");
    }
    return AGAN_OK;
}

int agan_conv_gather(const float* in, const void* wkv, const float* bias, float* out, const agan_conv_geom* gg, const int32_t* ktable,
                     int prec, int act, const float* lrelu_mask, void* ws, size_t ws_bytes, void* stream, const float* in_scale,
                     float* out_amax) {
    return agan_conv_gather_dt(in, wkv, bias, out, gg, ktable, prec, act, lrelu_mask, ws, ws_bytes, stream, in_scale, out_amax,
                               AGAN_DT_F32, AGAN_DT_F32);
}

size_t agan_conv_wgrad_ws_bytes(const agan_conv_geom* g) {
    if (check_geom(g)) return 0;
    const Geom gg = make_geom(g);
    // the larger of the two candidate paths (the small-N path is fp32-only; the caller does not pass the precision here)
    size_t a = plan_wgrad(gg, g->OS == 2).ws_bytes;
    if (small_n_wgrad_supported(gg)) a = std::max(a, plan_wgrad_small_n(gg).ws_bytes);
    if (patch_supported(gg)) {
        const PatchPlan pp = make_patch_plan(gg);
        a = std::max({a, plan_patch_wgrad(gg, pp, AGAN_PREC_BF16).ws_bytes, plan_patch_wgrad(gg, pp, AGAN_PREC_BF16X6).ws_bytes});
        for (int up = 0; up < 2; ++up) {     // (the plan's split does not depend on planes / storage; one plane: the most geometries)
            const RowsPlan rp = plan_rows_wgrad(gg, AGAN_PREC_BF16, false, false, up != 0);
            if (rp.ok) a = std::max(a, rp.ws_bytes);
        }
    }
    {
        const RowsPlan rp = plan_rows_wgrad(gg, AGAN_PREC_F32, false, false, false);
        if (rp.ok) a = std::max(a, rp.ws_bytes);
        const WinoWgradPlan wp = plan_wino_wgrad(gg);
        if (wp.ok) a = std::max(a, wp.ws_bytes);
    }
    return a;
}

// 1 if agan_conv_wgrad_dt takes x / dy in these storage types for forward geometry g (both fp32: always; 16-bit: the one-plane patch
// weight gradient in the matching mode)
int agan_conv_wgrad_dt_supported(const agan_conv_geom* gg, int pack_mode, int prec, int x_dtype, int dy_dtype) {
    if (check_geom(gg)) return 0;
    if (x_dtype == AGAN_DT_F32 && dy_dtype == AGAN_DT_F32) return 1;
    {   // the <= 4-output-channel weight gradient (RGB heads) reads a 16-bit x
        const Geom g = make_geom(gg);
        if (pack_mode == AGAN_PACK_FWD && small_n_wgrad_supported(g) && agan_conv_wgrad_effective_prec(gg, pack_mode, prec) == AGAN_PREC_F32)
            return (dt16(x_dtype) && dy_dtype == AGAN_DT_F32) ? 1 : 0;
    }
    if (prec != AGAN_PREC_BF16 && prec != AGAN_PREC_F16) return 0;
    const int want = prec == AGAN_PREC_BF16 ? AGAN_DT_BF16 : AGAN_DT_F16;
    if ((x_dtype != AGAN_DT_F32 && x_dtype != want) || (dy_dtype != AGAN_DT_F32 && dy_dtype != want)) return 0;
    return agan_conv_wgrad_effective_prec(gg, pack_mode, prec) == prec ? 1 : 0;
}

int agan_conv_wgrad(const float* x, const float* dy, float* dw, const agan_conv_geom* gg, const int32_t* ktable, int pack_mode,
                    int kh, int kw, int prec, int accumulate, void* ws, size_t ws_bytes, void* stream, const float* x_scale,
                    const float* dy_scale) {
    return agan_conv_wgrad_dt(x, dy, dw, gg, ktable, pack_mode, kh, kw, prec, accumulate, ws, ws_bytes, stream, x_scale, dy_scale,
                              AGAN_DT_F32, AGAN_DT_F32);
}

int agan_conv_wgrad_dt(const void* x_v, const void* dy_v, float* dw, const agan_conv_geom* gg, const int32_t* ktable, int pack_mode,
                       int kh, int kw, int prec, int accumulate, void* ws, size_t ws_bytes, void* stream, const float* x_scale,
                       const float* dy_scale, int x_dtype, int dy_dtype) {
    AGAN_REQUIRE(agan_conv_wgrad_dt_supported(gg, pack_mode, prec, x_dtype, dy_dtype),
                 "conv_wgrad: 16-bit activation storage (%d, %d) is not available for this geometry / precision %d", x_dtype, dy_dtype, prec);
    const float* x = static_cast<const float*>(x_v);
    const float* dy = static_cast<const float*>(dy_v);

    if (int e = check_geom(gg)) return e;
    AGAN_REQUIRE(x && dy && dw && ktable, "conv_wgrad: null pointer");
    AGAN_REQUIRE(prec == AGAN_PREC_F32 || prec_planes(prec) > 0, "conv_wgrad: unknown precision mode %d", prec);
    const int prec_asked = prec;
    prec = agan_conv_wgrad_effective_prec(gg, pack_mode, prec);     // (callers may pass the mode that is set: fp32 operands either way)
    AGAN_REQUIRE(pack_mode == AGAN_PACK_FWD || pack_mode == AGAN_PACK_UP_FWD, "conv_wgrad: pack mode %d is not a forward mode", pack_mode);
    const Geom g = make_geom(gg);
    const bool up = pack_mode == AGAN_PACK_UP_FWD;
    AGAN_REQUIRE(1LL * g.B * g.Cin * g.IH * g.IW < (1LL << 29) && 1LL * g.B * g.Cout * g.OH * g.OW < (1LL << 29),
                 "conv_wgrad: tensor exceeds 2^29 elements");
    if (!up) AGAN_REQUIRE(g.R == kh && g.S == kw && g.OS == 1, "conv_wgrad: geometry is not a direct %dx%d conv", kh, kw);
    else AGAN_REQUIRE(g.R == 2 && g.S == 2 && g.OS == 2 && kh == 3 && kw == 3, "conv_wgrad: geometry is not the folded upsample conv");
    hipStream_t st = as_stream(stream);
    if (prec == AGAN_PREC_F32 && !up && x_dtype == AGAN_DT_F32 && dy_dtype == AGAN_DT_F32) {
        // conv3x3 stride 1 on enough pixels: Winograd F(3x3, 2x2), 2.25x fewer fp32 MFMAs (conv_wino.hip)
        const WinoWgradPlan wp = plan_wino_wgrad(g);
        if (wp.ok) {
            if (wp.ws_bytes > ws_bytes || !ws) {
                set_error("conv_wgrad: workspace %zu < %zu", ws_bytes, wp.ws_bytes);
                return AGAN_EWORKSPACE;
            }
            float* wsf = static_cast<float*>(ws);
            timer_begin(st);
            launch_wino_wgrad(x, dy, wsf, g, wp, st);
            timer_end(st);
            if (int e = check_launch("conv_wgrad/winograd")) return e;
            launch_wino_wgrad_sum(wsf, g, wp, dw, accumulate ? 1 : 0, st);
            return check_launch("conv_wgrad/winograd/sum");
        }
    }
    if (!(prec == AGAN_PREC_F32 && small_n_wgrad_supported(g))) {
        // the row-resident kernel (conv_wgrows.hip) where it takes the geometry (any mode, fp32 included): [cout][K'] slabs, its own pixel split
        const RowsPlan rp = plan_rows_wgrad(g, prec, x_dtype != AGAN_DT_F32, dy_dtype != AGAN_DT_F32, up);
        if (rp.ok) {
            if (rp.ws_bytes > ws_bytes || !ws) {
                set_error("conv_wgrad: workspace %zu < %zu", ws_bytes, rp.ws_bytes);
                return AGAN_EWORKSPACE;
            }
            float* wsf = static_cast<float*>(ws);
            AGAN_REQUIRE(prec != AGAN_PREC_F16X3 || (x_scale && dy_scale),
                         "conv_wgrad: precision mode %d runs this weight gradient as AGAN_PREC_F16X3 (agan_conv_wgrad_effective_prec), which needs the "
                         "amax slots of x and dy (x_amax / dy_amax)", prec_asked);
            timer_begin(st);
            launch_rows_wgrad(x_v, dy_v, wsf, g, rp, prec, st, x_scale, dy_scale, x_dtype != AGAN_DT_F32, dy_dtype != AGAN_DT_F32);
            timer_end(st);
            if (int e = check_launch("conv_wgrad/rows")) return e;
            // few slabs: summed inside the unpack pass; many (the small layers, split 100-fold over their pixels): the bandwidth-bound slab sum first
            const float* src = wsf;
            int nsl = rp.psplit;
            if (rp.psplit > 4) {
                float* reduced = wsf + rp.slab * rp.psplit;
                const size_t n = (size_t)g.Cout * rp.Kp;
                launch_sum_slabs(wsf, rp.psplit, n, rp.slab, nullptr, 1, 1, reduced, 0, AGAN_ACT_NONE, nullptr, nullptr, st);
                if (int e = check_launch("conv_wgrad/sum_slabs")) return e;
                src = reduced;
                nsl = 1;
            }
            launch_wgrad_sum_unpack(src, nsl, rp.slab, dw, g.Cout, g.Cin, kh, kw, rp.NPH, rp.NT, rp.Kp, accumulate, st);      // (ups: a plain 3x3 result)
            return check_launch("conv_wgrad/sum_unpack");
        }
    }
    if (prec != AGAN_PREC_F32) {
        const PatchPlan pp = make_patch_plan(g);
        const PatchWgrad p = plan_patch_wgrad(g, pp, prec);
        if (p.ws_bytes > ws_bytes || !ws) {
            set_error("conv_wgrad: workspace %zu < %zu", ws_bytes, p.ws_bytes);
            return AGAN_EWORKSPACE;
        }
        float* wsf = static_cast<float*>(ws);
        float* reduced = p.psplit > 1 ? wsf + p.slab * p.psplit : wsf;
        timer_begin(st);
        AGAN_REQUIRE(prec != AGAN_PREC_F16X3 || (x_scale && dy_scale),
                         "conv_wgrad: precision mode %d runs this weight gradient as AGAN_PREC_F16X3 (agan_conv_wgrad_effective_prec), which needs the "
                         "amax slots of x and dy (x_amax / dy_amax)", prec_asked);
        launch_patch_wgrad(x_v, dy_v, wsf, g, pp, p, prec, st, x_scale, dy_scale, x_dtype != AGAN_DT_F32, dy_dtype != AGAN_DT_F32);
        timer_end(st);
        if (int e = check_launch("conv_wgrad/patch")) return e;
        if (!up && p.psplit <= 4) {
            launch_wgrad_sum_unpack(wsf, p.psplit, p.slab, dw, g.Cout, g.Cin, kh, kw, pp.NPH, pp.NT, p.Kp, accumulate, st);
            return check_launch("conv_wgrad/sum_unpack");
        }
        if (p.psplit > 1) {
            const size_t n = (size_t)p.ncls * g.Cout * p.Kp;
            launch_sum_slabs(wsf, p.psplit, n, p.slab, nullptr, 1, 1, reduced, 0, AGAN_ACT_NONE, nullptr, nullptr, st);
            if (int e = check_launch("conv_wgrad/sum_slabs")) return e;
        }
        if (!up) launch_wgrad_sum_unpack(reduced, 1, p.slab, dw, g.Cout, g.Cin, kh, kw, pp.NPH, pp.NT, p.Kp, accumulate, st);
        else launch_patch_wgrad_unpack(reduced, dw, g.Cout, g.Cin, kh, kw, up, pp, p, accumulate, st);
        return check_launch("conv_wgrad/unpack");
    }
    if (prec == AGAN_PREC_F32 && !up && small_n_wgrad_supported(g)) {
        const SmallWgradPlan sp = plan_wgrad_small_n(g);
        if (sp.ws_bytes > ws_bytes || !ws) {
            set_error("conv_wgrad: workspace %zu < %zu", ws_bytes, sp.ws_bytes);
            return AGAN_EWORKSPACE;
        }
        float* part = static_cast<float*>(ws);
        timer_begin(st);
        launch_wgrad_small_n(x_v, dy, part, g, sp, st, x_dtype);
        timer_end(st);
        const size_t n = (size_t)g.Cout * g.K;
        launch_sum_slabs(part, sp.nchunk, n, sp.slab, nullptr, 1, 1, dw, accumulate, AGAN_ACT_NONE, nullptr, nullptr, st);
        return check_launch("conv_wgrad/small_n");
    }
    const WgradPlan p = plan_wgrad(g, up);
    if (p.ws_bytes > ws_bytes || (p.ws_bytes && !ws)) {
        set_error("conv_wgrad: workspace %zu < %zu", ws_bytes, p.ws_bytes);
        return AGAN_EWORKSPACE;
    }
    const int2* ktab = reinterpret_cast<const int2*>(ktable);
    float* wsf = static_cast<float*>(ws);
    // where the (reduced) [cls][cout][K] result lands: dw itself for a direct conv, a scratch slab before the tap combine.
    // Under `accumulate` an unsplit direct conv also goes through one scratch slab so that the sum pass can add to dw.
    // an unsplit direct conv accumulates in its own epilogue (f32 kernels); every other case goes through the slab sum
    const int acc_in_kernel = (accumulate && !up && p.psplit == 1) ? 1 : 0;
    const bool via_sum = p.psplit > 1 || (accumulate && !up && !acc_in_kernel);
    float* reduced = up ? wsf + (p.psplit > 1 ? p.slab * p.psplit : 0) : dw;
    float* part = via_sum ? wsf : reduced;
    dim3 grid(p.itiles, p.jtiles, p.ncls * p.psplit);
#define AGAN_WG(BI, BJ) AGAN_LAUNCH((conv_wgrad_f32_kernel<BI, BJ>), grid, dim3(256), 0, st, x, dy, part, ktab, g, p.psplit, p.pchunk, p.slab, acc_in_kernel)
    timer_begin(st);
    if (p.bi == 128 && p.bj == 128) AGAN_WG(128, 128);
    else if (p.bi == 128) AGAN_WG(128, 64);
    else if (p.bj == 128) AGAN_WG(64, 128);
    else AGAN_WG(64, 64);
    timer_end(st);
#undef AGAN_WG
    if (int e = check_launch("conv_wgrad")) return e;
    const size_t n = (size_t)p.ncls * g.Cout * g.K;
    if (via_sum) {
        launch_sum_slabs(part, p.psplit, n, p.slab, nullptr, 1, 1, reduced, (accumulate && !up) ? 1 : 0, AGAN_ACT_NONE, nullptr, nullptr, st);
        if (int e = check_launch("conv_wgrad/sum_slabs")) return e;
    }
    if (up) {
        const size_t total = (size_t)g.Cout * g.Cin * 9;
        AGAN_LAUNCH(unpack_wgrad_up_kernel, dim3((unsigned)std::min<size_t>(cdivz(total, 256), 4096)), dim3(256), 0, st,
                           reduced, dw, g.Cout, g.Cin, accumulate);
        return check_launch("conv_wgrad/unpack");
    }
    return AGAN_OK;
}

int agan_bias_grad(const float* dy, float* dbias, int B, int C, int HW, int accumulate, void* stream) {
    AGAN_REQUIRE(dy && dbias && B > 0 && C > 0 && HW > 0, "bias_grad: bad argument");
    AGAN_LAUNCH(bias_grad_kernel, dim3(C), dim3(256), 0, as_stream(stream), dy, dbias, B, C, HW, accumulate);
    return check_launch("bias_grad");
}

}  // extern "C"
