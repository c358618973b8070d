// Convolutions with at most 4 output channels (the generator's RGB heads, the discriminators' gradient w.r.t. the image,
// the 1-logit discriminator head).  On the MFMA tiles 29 of 32 accumulator rows would be padding, so these HBM-bound layers
// run on the vector ALU instead: lanes along pixels (coalesced NCHW), the <= 4 output channels in registers, weights read
// with scalar loads (they are wave-uniform), and the per-tap image offsets -- with the padding test already folded in as an
// out-of-range buffer offset -- computed ONCE per pixel; the channel stride goes into the buffer instruction's scalar
// offset, so the inner loop is one buffer_load + Cout FMAs per (channel, tap).
#include "conv_common.h"

using namespace agan;
using namespace agan::conv;

namespace {

// forward / dgrad, Cout <= 4.  RS = taps per class (1, 4, 9, 16), S = tap columns.
// IDT: storage type of `in` (AGAN_DT_*)
template <int RS, int S, int IDT = AGAN_DT_F32>
__global__ __launch_bounds__(256) void conv_small_n_kernel(const void* __restrict__ in, const float* __restrict__ wk,
                                                           const float* __restrict__ bias, float* __restrict__ out, const Geom g) {
    // XCD-aware order (conv_common.h): neighbouring pixel blocks share halo rows and the parity classes of a block read the same
    // window -- position F = (pixel block, class), each XCD a contiguous run.  These layers are HBM-bound, so the re-fetches the
    // plain id order caused (2.6x / 4.5x the algorithmic bytes on the RGB heads / the image gradient) were paid in time.
    int F = xcd_contiguous(linear_block_id(), (int)(gridDim.x * gridDim.y));
    const int cls = F % (int)gridDim.y;
    F /= (int)gridDim.y;
    const int py = cls / g.OS, px = cls - py * g.OS;
    const int m = F * 256 + threadIdx.x;
    const bool valid = m < g.Mtot;
    const int mm = valid ? m : 0;
    const int b = g.dHWs.div(mm), rem = mm - b * g.HWs;
    const int yq = g.dOWs.div(rem), xq = rem - yq * g.OWs;
    const int iy0 = yq * g.SY + (py ? g.OY1 : g.OY0), ix0 = xq * g.SY + (px ? g.OY1 : g.OY0);
    const int ihw = g.IH * g.IW;
    const int pix0 = b * g.Cin * ihw + iy0 * g.IW + ix0;
    unsigned off[RS];
#pragma unroll
    for (int t = 0; t < RS; ++t) {
        const int dy = (t / S) * g.DY, dx = (t % S) * g.DY;
        const bool ok = valid & ((unsigned)(iy0 + dy) < (unsigned)g.IH) & ((unsigned)(ix0 + dx) < (unsigned)g.IW);
        off[t] = ok ? (unsigned)(pix0 + dy * g.IW + dx) * (IDT == AGAN_DT_F32 ? 4u : 2u) : kOOB;
    }
    constexpr unsigned IE = IDT == AGAN_DT_F32 ? 4u : 2u;
    const __amdgpu_buffer_rsrc_t rin = make_rsrc(in, (size_t)g.B * g.Cin * ihw * IE);
    const float* wc = wk + (size_t)cls * g.K * g.Nld;      // packed fp32 weights [K][Nld]: row k starts with the <= 4 live columns
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int c = 0; c < g.Cin; ++c) {
        const unsigned soff = (unsigned)(c * ihw) * IE;
        const float* wr = wc + (size_t)c * RS * g.Nld;
#pragma unroll
        for (int t = 0; t < RS; ++t) {
            float v;
            if (IDT == AGAN_DT_F32) v = buf_load_s(rin, off[t], soff);
            else {
                const unsigned short h = (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rin, off[t], soff, 0);
                v = IDT == AGAN_DT_BF16 ? __uint_as_float((unsigned)h << 16) : (float)__builtin_bit_cast(_Float16, h);
            }
            const float4 w = *reinterpret_cast<const float4*>(wr + (size_t)t * g.Nld);      // wave-uniform -> s_load_dwordx4
            a0 += v * w.x; a1 += v * w.y; a2 += v * w.z; a3 += v * w.w;
        }
    }
    if (!valid) return;
    const size_t ohw = (size_t)g.OH * g.OW;
    float* o = out + (size_t)b * g.Cout * ohw + (size_t)(yq * g.OS + py) * g.OW + (xq * g.OS + px);
    const float acc[4] = {a0, a1, a2, a3};
#pragma unroll
    for (int n = 0; n < 4; ++n)
        if (n < g.Cout) o[(size_t)n * ohw] = acc[n] + (bias ? bias[n] : 0.f);
}

// 4 consecutive activation elements / one element at a byte offset of a buffer resource, widened to fp32
template <int DT> __device__ __forceinline__ void buf_load4_dt(const __amdgpu_buffer_rsrc_t r, unsigned eoff, unsigned esoff, float* v) {
    if (DT == AGAN_DT_F32) {
        const f32x4 m = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, eoff == kOOB ? kOOB : eoff * 4u, esoff * 4u, 0));
        v[0] = m[0]; v[1] = m[1]; v[2] = m[2]; v[3] = m[3];
    } else {
        const uint2 raw = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(r, eoff == kOOB ? kOOB : eoff * 2u, esoff * 2u, 0));
        if (DT == AGAN_DT_BF16) {
            v[0] = __uint_as_float(raw.x << 16); v[1] = __uint_as_float(raw.x & 0xFFFF0000u);
            v[2] = __uint_as_float(raw.y << 16); v[3] = __uint_as_float(raw.y & 0xFFFF0000u);
        } else {
            const agan_f32x4 f = __builtin_convertvector(__builtin_bit_cast(agan_f16x4, raw), agan_f32x4);
            v[0] = f[0]; v[1] = f[1]; v[2] = f[2]; v[3] = f[3];
        }
    }
}
template <int DT> __device__ __forceinline__ float buf_load1_dt(const __amdgpu_buffer_rsrc_t r, unsigned eoff, unsigned esoff) {
    if (DT == AGAN_DT_F32) return buf_load_s(r, eoff == kOOB ? kOOB : eoff * 4u, esoff * 4u);
    const unsigned short h = (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(r, eoff == kOOB ? kOOB : eoff * 2u, esoff * 2u, 0);
    return DT == AGAN_DT_BF16 ? __uint_as_float((unsigned)h << 16) : (float)__builtin_bit_cast(_Float16, h);
}

// ---- row strips (round 4) ----------------------------------------------------------------------------------------------------
// A thread owns 4 consecutive pixels of one row; the strips of a row are consecutive thread ids, so a strip's left / right neighbour
// pixel is lane - 1's last / lane + 1's first value: ONE 16-byte load per input channel and patch row, the two halo pixels by DPP
// wave shifts (v_mov_b32_dpp wave_shr:1 / wave_shl:1) instead of two more load instructions -- 3 vector-memory instructions per
// channel instead of 9.  Rows of 16 .. 256 pixels (4 .. 64 strips) never straddle a wave; wider or odd rows take the halo of the
// wave's first / last lane from memory (`edge`, wave-uniform).  The NC <= 4 output channels are plain scalar accumulators -- one v_fmac_f32 per
// channel, none for a dead fourth channel (the round-3 kernels always computed four).
//
// NO PACKED FP32 MATH (this file is compiled with -fno-slp-vectorize on top of the library-wide -packed-fp32-ops, csrc/Makefile).  The first version of
// these kernels kept channel pairs in float2 accumulators; the compiler turned the updates into v_pk_fma_f32 with the pixel broadcast by op_sel, and the
// eight instructions of the image-gradient kernel that took it from the HIGH dword of a register pair (op_sel:[0,1,0]) gave run-to-run DIFFERENT low-half
// results in lanes 48-63 inside the bf16x6 train step (up to 7 % on single pixels of the image gradient, 3e-3 on an RGB head's weight gradient --
// tests/test_gpu_metric_parity.py[bf16x6] caught it) while every launch in isolation was bit-stable.  Cause, reproduced without this library in
// profiles/micro/pk_fma_coexec.hip: on this pool's MI355X exactly that instruction form returns wrong low halves in lanes 48-63 now and then while OTHER
// waves of the compute unit issue v_mfma_f32_32x32x16_bf16 (the other discriminators' convolutions); next to fp32 MFMAs or on an idle chip it never does.
// The scalar kernels below are also the fastest of the three builds measured at 256 x 256: they wait on memory, not on the vector ALU.
// profiles/r04_p16_analysis.txt section 7; -DAGAN_STRIP_PACKED_PAIRS rebuilds the packed form of the image-gradient kernel for profiles/dev/dgrad4_stress4.py.
constexpr int kStripCh = 64;        // channels whose weights sit in LDS at a time
__device__ __forceinline__ float dpp_prev_lane(float v) {      // lane i <- lane i - 1 (lane 0: 0)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_next_lane(float v) {      // lane i <- lane i + 1 (lane 63: 0)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}
// the 3 patch rows of one channel for a strip: v[r][0..5] = pixels x0 - 1 .. x0 + 4 of row y + r - 1
struct StripHalo {
    bool has_l, has_r;          // the strip has a left / right neighbour in its row
    bool edge;                  // (wave-uniform) some lane's neighbour strip lives in another wave
    unsigned offl[3], offr[3];  // element offsets of those lanes' halo pixels (kOOB elsewhere)
};
// The three row loads of a channel are ISSUED one channel (strip) ahead of their use (strip_issue into raw registers, strip_finish widens them and
// adds the halo): a thread's loop is a dependent chain load -> DPP -> FMAs, and with the next loads not in flight every iteration paid the full
// memory latency (first version of these kernels: the same ~55 / ~180 us for a 64 x 64 and a 256 x 256 image).
typedef unsigned u32x4s __attribute__((ext_vector_type(4)));
template <int DT>
__device__ __forceinline__ void strip_issue(const __amdgpu_buffer_rsrc_t r, const unsigned (&off4)[3], unsigned soff, u32x4s (&raw)[3]) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        if (DT == AGAN_DT_F32) {
            raw[j] = __builtin_bit_cast(u32x4s, __builtin_amdgcn_raw_buffer_load_b128(r, off4[j] == kOOB ? kOOB : off4[j] * 4u, soff * 4u, 0));
        } else {
            const uint2 t = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(r, off4[j] == kOOB ? kOOB : off4[j] * 2u, soff * 2u, 0));
            raw[j] = u32x4s{t.x, t.y, 0u, 0u};
        }
    }
}
template <int DT>
__device__ __forceinline__ void strip_finish(const __amdgpu_buffer_rsrc_t r, const u32x4s (&raw)[3], const StripHalo& h, unsigned soff, float (&v)[3][6]) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        if (DT == AGAN_DT_F32) {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[j][1 + k] = __uint_as_float(raw[j][k]);
        } else if (DT == AGAN_DT_BF16) {
            v[j][1] = __uint_as_float(raw[j][0] << 16); v[j][2] = __uint_as_float(raw[j][0] & 0xFFFF0000u);
            v[j][3] = __uint_as_float(raw[j][1] << 16); v[j][4] = __uint_as_float(raw[j][1] & 0xFFFF0000u);
        } else {
            const uint2 t = {raw[j][0], raw[j][1]};
            const agan_f32x4 f = __builtin_convertvector(__builtin_bit_cast(agan_f16x4, t), agan_f32x4);
            v[j][1] = f[0]; v[j][2] = f[1]; v[j][3] = f[2]; v[j][4] = f[3];
        }
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const float l = dpp_prev_lane(v[j][4]), q = dpp_next_lane(v[j][1]);
        v[j][0] = h.has_l ? l : 0.f;
        v[j][5] = h.has_r ? q : 0.f;
    }
    if (h.edge) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float l = buf_load1_dt<DT>(r, h.offl[j], soff), q = buf_load1_dt<DT>(r, h.offr[j], soff);
            if (h.offl[j] != kOOB) v[j][0] = l;
            if (h.offr[j] != kOOB) v[j][5] = q;
        }
    }
}
// strip `ss` of a [.., C, IH, IW] tensor (4-pixel strips, xs fastest): offsets of its three patch rows and its halo description
__device__ __forceinline__ void strip_setup(bool valid, int b, int y, int xs, int W4, int C, int IH, int IW, unsigned (&off4)[3], StripHalo& h) {
    const int lane = threadIdx.x & 63;
    const int x0 = xs * 4;
    h.has_l = xs > 0;
    h.has_r = xs + 1 < W4;
    const bool nl = valid & h.has_l & (lane == 0), nr = valid & h.has_r & (lane == 63);
    h.edge = __builtin_amdgcn_ballot_w64(nl | nr) != 0;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int iy = y + r - 1;
        const bool rok = valid & ((unsigned)iy < (unsigned)IH);
        const unsigned base = (unsigned)((b * C * IH + iy) * IW + x0);
        off4[r] = rok ? base : kOOB;
        h.offl[r] = (rok & nl) ? base - 1u : kOOB;
        h.offr[r] = (rok & nr) ? base + 4u : kOOB;
    }
}

// 3x3 / stride 1 / pad 1 with NC <= 4 output channels (the RGB heads, generator_submodules.py:135)
// CS (channel split, 1 or 4): on small images the grid is a few hundred waves, each a dependent chain over all Cin channels (the 64 x 64 head ran the
// same 22 us as the 128 x 128 one) -- with CS = 4 the four waves of a workgroup share 64 strips, wave w takes channels w, w + 4, ... and the partial sums
// meet in LDS in wave order (a fixed order: results do not depend on timing).
template <int IDT, int NC, int CS>
__global__ __launch_bounds__(256) void conv_small_strip_kernel(const void* __restrict__ in, const float* __restrict__ wk,
                                                               const float* __restrict__ bias, float* __restrict__ out, const Geom g) {
    const int W4 = g.IW >> 2;
    const int nstrips = g.B * g.IH * W4;
    const int wave = threadIdx.x >> 6;
    const int sid = CS == 1 ? blockIdx.x * 256 + threadIdx.x : blockIdx.x * 64 + (threadIdx.x & 63);
    const bool valid = sid < nstrips;
    const int ss = valid ? sid : 0;
    const int xs = ss % W4, t1 = ss / W4, y = t1 % g.IH, b = t1 / g.IH;
    const int ihw = g.IH * g.IW;
    const __amdgpu_buffer_rsrc_t rin = make_rsrc(in, (size_t)g.B * g.Cin * ihw * (IDT == AGAN_DT_F32 ? 4 : 2));
    unsigned off4[3];
    StripHalo h;
    strip_setup(valid, b, y, xs, W4, g.Cin, g.IH, g.IW, off4, h);
    float acc[4][NC];                                   // [pixel][channel]
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int n = 0; n < NC; ++n) acc[p][n] = 0.f;
    // weights of up to kStripCh channels in LDS ([channel][tap] x 4 floats; every lane reads the same 16 bytes: a broadcast ds_read_b128) -- as
    // wave-uniform s_load_dwordx4 they were fetched two at a time right in front of their use, a scalar-cache miss each
    __shared__ float4 wl[kStripCh * 9];
    u32x4s raw[3];
    const int cfirst = CS == 1 ? 0 : wave;              // this wave's channels: cfirst, cfirst + CS, ...  (kStripCh is a multiple of CS)
    if (cfirst < g.Cin) strip_issue<IDT>(rin, off4, (unsigned)(cfirst * ihw), raw);
    for (int c0 = 0; c0 < g.Cin; c0 += kStripCh) {
        const int nch = min(kStripCh, g.Cin - c0);
        __syncthreads();
        for (int i = threadIdx.x; i < nch * 9; i += 256) wl[i] = *reinterpret_cast<const float4*>(wk + (size_t)(c0 * 9 + i) * g.Nld);
        __syncthreads();
        for (int cc = cfirst; cc < nch; cc += CS) {
            const int c = c0 + cc;
            float v[3][6];
            strip_finish<IDT>(rin, raw, h, (unsigned)(c * ihw), v);
            if (c + CS < g.Cin) strip_issue<IDT>(rin, off4, (unsigned)((c + CS) * ihw), raw);
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    const float4 w4 = wl[cc * 9 + r * 3 + t];
                    const float w[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
                    for (int p = 0; p < 4; ++p)
#pragma unroll
                        for (int n = 0; n < NC; ++n) acc[p][n] += w[n] * v[r][p + t];
                }
        }
    }
    if (CS > 1) {
        __shared__ float red[CS - 1 > 0 ? CS - 1 : 1][4 * NC][64];
        const int lane = threadIdx.x & 63;
        __syncthreads();
        if (wave > 0) {
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int n = 0; n < NC; ++n) red[wave - 1][p * NC + n][lane] = acc[p][n];
        }
        __syncthreads();
        if (wave > 0) return;
#pragma unroll
        for (int w = 0; w < CS - 1; ++w)
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int n = 0; n < NC; ++n) acc[p][n] += red[w][p * NC + n][lane];
    }
    if (!valid) return;
    float* o = out + ((size_t)b * g.Cout * g.IH + y) * g.IW + xs * 4;
#pragma unroll
    for (int n = 0; n < NC; ++n) {
        if (n >= g.Cout) break;
        const float bv = bias ? bias[n] : 0.f;
        *reinterpret_cast<float4*>(o + (size_t)n * ihw) = make_float4(acc[0][n] + bv, acc[1][n] + bv, acc[2][n] + bv, acc[3][n] + bv);
    }
}

// Data gradient of conv4x4 s2 p1 with NC <= 4 INPUT channels (the discriminators' gradient w.r.t. the image, discriminators.py:48-70), as its four
// parity classes of 2x2 taps (include/agan.h: R = S = 2, OS = 2, SY = 1, DY = -1, OY = (0, 1)) on row strips: a thread owns 4 consecutive
// lattice points of one lattice row and ALL FOUR classes -- the 2 x 8 output pixels they produce.  The classes' taps cover rows y' - 1 .. y' + 1
// and columns x' - 1 .. x' + 1 of dY: the same three 6-pixel rows per channel as the 3x3 strip, loaded once instead of once per class and tap
// (the per-pixel kernel above issued one 4-byte load per (class, channel, tap): 17 % of HBM peak, bound by its vector-memory instructions).
template <int IDT, int NC, int CS>
__global__ __launch_bounds__(256) void conv_small_dgrad4_strip_kernel(const void* __restrict__ in, const float* __restrict__ wk,
                                                                      const float* __restrict__ bias, float* __restrict__ out, const Geom g) {
    const int W4 = g.IW >> 2;
    const int nstrips = g.B * g.IH * W4;
    const int wave = threadIdx.x >> 6;
    const int sid = CS == 1 ? blockIdx.x * 256 + threadIdx.x : blockIdx.x * 64 + (threadIdx.x & 63);
    const bool valid = sid < nstrips;
    const int ss = valid ? sid : 0;
    const int xs = ss % W4, t1 = ss / W4, y = t1 % g.IH, b = t1 / g.IH;
    const int ihw = g.IH * g.IW;
    const __amdgpu_buffer_rsrc_t rin = make_rsrc(in, (size_t)g.B * g.Cin * ihw * (IDT == AGAN_DT_F32 ? 4 : 2));
    unsigned off4[3];
    StripHalo h;
    strip_setup(valid, b, y, xs, W4, g.Cin, g.IH, g.IW, off4, h);
#ifdef AGAN_STRIP_PACKED_PAIRS      // (diagnostic build only: the float2 / v_pk_fma_f32 form of profiles/r04_p16_analysis.txt section 7)
    typedef float f32x2s __attribute__((ext_vector_type(2)));
    f32x2s a01[4][4], a23[4][4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int p = 0; p < 4; ++p) { a01[k][p] = f32x2s{0.f, 0.f}; a23[k][p] = f32x2s{0.f, 0.f}; }
#else
    float acc[4][4][NC];                                // [class][lattice point][channel]
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int n = 0; n < NC; ++n) acc[k][p][n] = 0.f;
#endif
    const size_t wcls = (size_t)g.K * g.Nld;            // K = Cin * 4
    __shared__ float4 wl[kStripCh * 16];                // [channel][class][r][t] x 4 floats
    u32x4s raw[3];
    const int cfirst = CS == 1 ? 0 : wave;
    if (cfirst < g.Cin) strip_issue<IDT>(rin, off4, (unsigned)(cfirst * ihw), raw);
    for (int c0 = 0; c0 < g.Cin; c0 += kStripCh) {
        const int nch = min(kStripCh, g.Cin - c0);
        __syncthreads();
        for (int i = threadIdx.x; i < nch * 16; i += 256) {
            const int cc = i >> 4, k = (i >> 2) & 3, rt = i & 3;
            wl[i] = *reinterpret_cast<const float4*>(wk + k * wcls + (size_t)((c0 + cc) * 4 + rt) * g.Nld);
        }
        __syncthreads();
        for (int cc = cfirst; cc < nch; cc += CS) {
            const int c = c0 + cc;
            float v[3][6];
            strip_finish<IDT>(rin, raw, h, (unsigned)(c * ihw), v);
            if (c + CS < g.Cin) strip_issue<IDT>(rin, off4, (unsigned)((c + CS) * ihw), raw);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int py = k >> 1, px = k & 1;
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const float4 w4 = wl[cc * 16 + k * 4 + r * 2 + t];
                        const float w[4] = {w4.x, w4.y, w4.z, w4.w};
                        const int row = py - r + 1, col = px - t + 1;          // dY row y' + OY[py] - r, column x' + OY[px] - t
#ifdef AGAN_STRIP_PACKED_PAIRS
                        const f32x2s w01 = {w4.x, w4.y}, w23 = {w4.z, w4.w};
#pragma unroll
                        for (int p = 0; p < 4; ++p) {
                            const float x = v[row][p + col];
                            a01[k][p] += w01 * x;
                            if (NC >= 3) a23[k][p] += w23 * x;
                        }
                        (void)w;
#else
#pragma unroll
                        for (int p = 0; p < 4; ++p)
#pragma unroll
                            for (int n = 0; n < NC; ++n) acc[k][p][n] += w[n] * v[row][p + col];
#endif
                    }
            }
        }
    }
#ifdef AGAN_STRIP_PACKED_PAIRS
    float acc[4][4][NC];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int n = 0; n < NC; ++n) acc[k][p][n] = n < 2 ? a01[k][p][n & 1] : a23[k][p][n & 1];
#endif
    if (CS > 1) {
        __shared__ float red[CS - 1 > 0 ? CS - 1 : 1][16 * NC][64];
        const int lane = threadIdx.x & 63;
        __syncthreads();
        if (wave > 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int p = 0; p < 4; ++p)
#pragma unroll
                    for (int n = 0; n < NC; ++n) red[wave - 1][(k * 4 + p) * NC + n][lane] = acc[k][p][n];
        }
        __syncthreads();
        if (wave > 0) return;
#pragma unroll
        for (int w = 0; w < CS - 1; ++w)
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int p = 0; p < 4; ++p)
#pragma unroll
                    for (int n = 0; n < NC; ++n) acc[k][p][n] += red[w][(k * 4 + p) * NC + n][lane];
    }
    if (!valid) return;
    // output rows 2 y' + py, columns 8 xs .. 8 xs + 7 = (x' = 4 xs + p, px) interleaved
#pragma unroll
    for (int n = 0; n < NC; ++n) {
        if (n >= g.Cout) break;
        const float bv = bias ? bias[n] : 0.f;
#pragma unroll
        for (int py = 0; py < 2; ++py) {
            float q[8];
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int px = 0; px < 2; ++px)
                    q[2 * p + px] = acc[py * 2 + px][p][n] + bv;
            float* o = out + (((size_t)b * g.Cout + n) * g.OH + (2 * y + py)) * g.OW + xs * 8;
            *reinterpret_cast<float4*>(o) = make_float4(q[0], q[1], q[2], q[3]);
            *reinterpret_cast<float4*>(o + 4) = make_float4(q[4], q[5], q[6], q[7]);
        }
    }
}

// one output row per workgroup: out[m][n] = sum_k in[m][k] * w[k][n], n < 4   (discriminator logit: K = 8192, M = batch)
__global__ __launch_bounds__(256) void linear_small_n_kernel(const float* __restrict__ in, const float* __restrict__ wk,
                                                             const float* __restrict__ bias, float* __restrict__ out, int K, int Nld, int Cout) {
    __shared__ float red[4][4];
    const int m = blockIdx.x;
    const float* x = in + (size_t)m * K;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = threadIdx.x; k < K; k += 256) {
        const float v = x[k];
        const float4 w = *reinterpret_cast<const float4*>(wk + (size_t)k * Nld);
        a[0] += v * w.x; a[1] += v * w.y; a[2] += v * w.z; a[3] += v * w.w;
    }
#pragma unroll
    for (int n = 0; n < 4; ++n) a[n] = wave_sum(a[n]);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0)
        for (int n = 0; n < 4; ++n) red[n][w] = a[n];
    __syncthreads();
    if (threadIdx.x < Cout) {
        const int n = threadIdx.x;
        out[(size_t)m * Cout + n] = red[n][0] + red[n][1] + red[n][2] + red[n][3] + (bias ? bias[n] : 0.f);
    }
}

// weight gradient of a direct (OS = 1) conv with Cout <= 4: one workgroup = one input channel x one chunk of pixels.
// Each lane walks its pixels with RS*Cout running sums in registers; they meet once per workgroup (wave shuffle + LDS).
template <int RS, int S, int XDT = AGAN_DT_F32>
__global__ __launch_bounds__(256) void wgrad_small_n_kernel(const void* __restrict__ x, const float* __restrict__ dy, float* __restrict__ part,
                                                            const Geom g, const int pchunk, const size_t slab) {
    __shared__ float red[4][RS * 4];
    // XCD-aware order: the Cin workgroups of one pixel chunk all read the chunk's dY -- keep them on one XCD
    const int F = xcd_contiguous(linear_block_id(), (int)(gridDim.x * gridDim.y));
    const int c = F % (int)gridDim.x, chunk = F / (int)gridDim.x;
    const int pbeg = chunk * pchunk, pend = min(g.Mtot, pbeg + pchunk);
    const int ihw = g.IH * g.IW, ohw = g.OH * g.OW;
    constexpr unsigned XE = XDT == AGAN_DT_F32 ? 4u : 2u;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(x, (size_t)g.B * g.Cin * ihw * XE);
    const unsigned soff = (unsigned)(c * ihw) * XE;
    float acc[RS][4];
#pragma unroll
    for (int t = 0; t < RS; ++t)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[t][n] = 0.f;
    for (int p = pbeg + threadIdx.x; p < pend; p += 256) {
        const int b = g.dHWs.div(p), rem = p - b * g.HWs;
        const int yq = g.dOWs.div(rem), xq = rem - yq * g.OWs;
        const int iy0 = yq * g.SY + g.OY0, ix0 = xq * g.SY + g.OY0;
        const int pix0 = b * g.Cin * ihw + iy0 * g.IW + ix0;
        const float* dp = dy + (size_t)b * g.Cout * ohw + (size_t)yq * g.OW + xq;
        float d[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) d[n] = n < g.Cout ? dp[(size_t)n * ohw] : 0.f;
#pragma unroll
        for (int t = 0; t < RS; ++t) {
            const int ddy = (t / S) * g.DY, ddx = (t % S) * g.DY;
            const bool ok = ((unsigned)(iy0 + ddy) < (unsigned)g.IH) & ((unsigned)(ix0 + ddx) < (unsigned)g.IW);
            float v;
            if (XDT == AGAN_DT_F32) v = buf_load_s(rx, ok ? (unsigned)(pix0 + ddy * g.IW + ddx) * 4u : kOOB, soff);
            else {
                const unsigned short h = (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rx, ok ? (unsigned)(pix0 + ddy * g.IW + ddx) * 2u : kOOB, soff, 0);
                v = XDT == AGAN_DT_BF16 ? __uint_as_float((unsigned)h << 16) : (float)__builtin_bit_cast(_Float16, h);
            }
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[t][n] += v * d[n];
        }
    }
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int t = 0; t < RS; ++t)
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const float s = wave_sum(acc[t][n]);
            if (lane == 0) red[w][t * 4 + n] = s;
        }
    __syncthreads();
    if (threadIdx.x < RS * 4) {
        const int t = threadIdx.x >> 2, n = threadIdx.x & 3;
        if (n < g.Cout)
            part[(size_t)chunk * slab + (size_t)n * g.K + c * RS + t] =
                red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    }
}

// weight gradient of the same layers, row strips: per 4 pixels a thread loads the NC dy blocks and, per patch row of x, one 16-byte block
// (halo pixels by DPP, see above) -- 6 loads instead of 48.  One workgroup = one input channel x one chunk of strips, as above; every lane stays
// in the loop for the whole chunk (a strip beyond the chunk loads nothing and adds zeros) so that the DPP neighbours are always live.
template <int XDT, int NC>
__global__ __launch_bounds__(256) void wgrad_small_strip_kernel(const void* __restrict__ x, const float* __restrict__ dy, float* __restrict__ part,
                                                                const Geom g, const int schunk, const size_t slab) {
    __shared__ float red[4][9 * 4];
    const int F = xcd_contiguous(linear_block_id(), (int)(gridDim.x * gridDim.y));
    const int c = F % (int)gridDim.x, chunk = F / (int)gridDim.x;
    const int W4 = g.IW >> 2;
    const int nstrips = g.B * g.IH * W4;
    const int sbeg = chunk * schunk, send = min(nstrips, sbeg + schunk);
    const int ihw = g.IH * g.IW;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(x, (size_t)g.B * g.Cin * ihw * (XDT == AGAN_DT_F32 ? 4 : 2));
    const __amdgpu_buffer_rsrc_t rd = make_rsrc(dy, (size_t)g.B * g.Cout * ihw * sizeof(float));
    const unsigned soff = (unsigned)(c * ihw);
    float acc[9][NC];                                   // [tap][output channel]
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int n = 0; n < NC; ++n) acc[t][n] = 0.f;
    // a strip's loads (three x rows, NC dy blocks) are issued one pass ahead of the FMAs that consume them (two register sets, the loop unrolled
    // by two so that the sets never have to be copied)
    struct Pass { StripHalo h; u32x4s raw[3]; f32x4 dyr[4]; };
    auto issue = [&](int s0, Pass& P) {
        const int sid = s0 + (int)threadIdx.x;
        const bool valid = sid < send;
        const int ss = valid ? sid : sbeg;
        const int xs = ss % W4, t1 = ss / W4, y = t1 % g.IH, b = t1 / g.IH;
        unsigned off4[3];
        strip_setup(valid, b, y, xs, W4, g.Cin, g.IH, g.IW, off4, P.h);
        strip_issue<XDT>(rx, off4, soff, P.raw);
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            P.dyr[n] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (n < NC) P.dyr[n] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                       rd, (valid & (n < g.Cout)) ? (unsigned)(((b * g.Cout + n) * g.IH + y) * g.IW + xs * 4) * 4u : kOOB, 0, 0));
        }
    };
    auto consume = [&](const Pass& P) {
        float v[3][6];
        strip_finish<XDT>(rx, P.raw, P.h, soff, v);
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const float xv = v[r][p + t];
#pragma unroll
                    for (int n = 0; n < NC; ++n) acc[r * 3 + t][n] += P.dyr[n][p] * xv;
                }
    };
    Pass A, B;
    if (sbeg < send) issue(sbeg, A);
    for (int s0 = sbeg; s0 < send; s0 += 512) {
        const bool more1 = s0 + 256 < send, more2 = s0 + 512 < send;
        if (more1) issue(s0 + 256, B);
        consume(A);
        if (more1) {
            if (more2) issue(s0 + 512, A);
            consume(B);
        }
    }
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int n = 0; n < NC; ++n) {
            const float sum = wave_sum(acc[t][n]);
            if (lane == 0) red[w][t * 4 + n] = sum;
        }
    __syncthreads();
    if (threadIdx.x < 36) {
        const int t = threadIdx.x >> 2, n = threadIdx.x & 3;
        if (n < NC && n < g.Cout)
            part[(size_t)chunk * slab + (size_t)n * g.K + c * 9 + t] =
                red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    }
}

// the row-strip kernels take a 3x3 / stride 1 / pad 1 layer on whole 4-pixel blocks
bool strip_ok(const Geom& g) {
    static const bool off = getenv("AGAN_SMALL_STRIP_OFF") != nullptr;
    return !off && g.RS == 9 && g.S == 3 && g.OS == 1 && g.SY == 1 && g.DY == 1 && g.OY0 == -1 && g.IH == g.OH && g.IW == g.OW && (g.IW & 3) == 0 && g.Cout <= 4;
}
// the 4-class strip kernel takes the data gradient of conv4x4 s2 p1 (functional.py conv_geoms "down") on whole 4-point blocks
bool dgrad4_strip_ok(const Geom& g) {
    static const bool off = getenv("AGAN_SMALL_STRIP_OFF") != nullptr || getenv("AGAN_DGRAD4_STRIP_OFF") != nullptr;
    return !off && g.R == 2 && g.S == 2 && g.OS == 2 && g.SY == 1 && g.DY == -1 && g.OY0 == 0 && g.OY1 == 1 && g.OH == 2 * g.IH && g.OW == 2 * g.IW &&
           (g.IW & 3) == 0 && g.Cout <= 4;
}

}  // namespace

namespace agan {
namespace conv {

bool small_n_gather_supported(const Geom& g) {
    if (g.Cout > 4 || g.R != g.S) return false;
    return g.RS == 1 || g.RS == 4 || g.RS == 9 || g.RS == 16;
}

void launch_gather_small_n(const void* in, const float* wk, const float* bias, float* out, const Geom& g, hipStream_t st, int in_dtype) {
    if (g.RS == 1 && g.IH == 1 && g.IW == 1 && g.OS == 1 && g.SY == 1 && g.OY0 == 0) {       // a linear layer: one row per workgroup (fp32 only)
        AGAN_LAUNCH(linear_small_n_kernel, dim3(g.Mtot), dim3(256), 0, st, static_cast<const float*>(in), wk, bias, out, g.K, g.Nld, g.Cout);
        return;
    }
    if (strip_ok(g)) {
        // few strips (small images): split the channels over the four waves of a workgroup (the kernel's CS)
        const int nstrips = g.B * g.IH * (g.IW >> 2);
        const bool split = nstrips <= 64 * 1024 && g.Cin >= 8;      // (measured: at 1536 waves the split is neutral to slower)
        const dim3 sg(split ? cdiv(nstrips, 64) : cdiv(nstrips, 256));
#define AGAN_STRIP2(NC_, CS_)                                                                                                             \
    do {                                                                                                                                  \
        if (in_dtype == AGAN_DT_BF16) AGAN_LAUNCH((conv_small_strip_kernel<AGAN_DT_BF16, NC_, CS_>), sg, dim3(256), 0, st, in, wk, bias, out, g); \
        else if (in_dtype == AGAN_DT_F16) AGAN_LAUNCH((conv_small_strip_kernel<AGAN_DT_F16, NC_, CS_>), sg, dim3(256), 0, st, in, wk, bias, out, g); \
        else AGAN_LAUNCH((conv_small_strip_kernel<AGAN_DT_F32, NC_, CS_>), sg, dim3(256), 0, st, in, wk, bias, out, g);                  \
    } while (0)
#define AGAN_STRIP(NC_) do { if (split) AGAN_STRIP2(NC_, 4); else AGAN_STRIP2(NC_, 1); } while (0)
        if (g.Cout <= 2) AGAN_STRIP(2);
        else if (g.Cout == 3) AGAN_STRIP(3);
        else AGAN_STRIP(4);
#undef AGAN_STRIP
#undef AGAN_STRIP2
        return;
    }
    if (dgrad4_strip_ok(g)) {
        const int nstrips = g.B * g.IH * (g.IW >> 2);
        const bool split = nstrips <= 64 * 1024 && g.Cin >= 8;      // (measured: at 1536 waves the split is neutral to slower)
        const dim3 sg(split ? cdiv(nstrips, 64) : cdiv(nstrips, 256));
#define AGAN_STRIP2(NC_, CS_)                                                                                                             \
    do {                                                                                                                                  \
        if (in_dtype == AGAN_DT_BF16) AGAN_LAUNCH((conv_small_dgrad4_strip_kernel<AGAN_DT_BF16, NC_, CS_>), sg, dim3(256), 0, st, in, wk, bias, out, g); \
        else if (in_dtype == AGAN_DT_F16) AGAN_LAUNCH((conv_small_dgrad4_strip_kernel<AGAN_DT_F16, NC_, CS_>), sg, dim3(256), 0, st, in, wk, bias, out, g); \
        else AGAN_LAUNCH((conv_small_dgrad4_strip_kernel<AGAN_DT_F32, NC_, CS_>), sg, dim3(256), 0, st, in, wk, bias, out, g);           \
    } while (0)
#define AGAN_STRIP(NC_) do { if (split) AGAN_STRIP2(NC_, 4); else AGAN_STRIP2(NC_, 1); } while (0)
        if (g.Cout <= 2) AGAN_STRIP(2);
        else if (g.Cout == 3) AGAN_STRIP(3);
        else AGAN_STRIP(4);
#undef AGAN_STRIP
#undef AGAN_STRIP2
        return;
    }
    dim3 grid(cdiv(g.Mtot, 256), g.OS * g.OS);
#define AGAN_SN(RS_, S_)                                                                                                                  \
    do {                                                                                                                                  \
        if (in_dtype == AGAN_DT_BF16) AGAN_LAUNCH((conv_small_n_kernel<RS_, S_, AGAN_DT_BF16>), grid, dim3(256), 0, st, in, wk, bias, out, g); \
        else if (in_dtype == AGAN_DT_F16) AGAN_LAUNCH((conv_small_n_kernel<RS_, S_, AGAN_DT_F16>), grid, dim3(256), 0, st, in, wk, bias, out, g); \
        else AGAN_LAUNCH((conv_small_n_kernel<RS_, S_>), grid, dim3(256), 0, st, in, wk, bias, out, g);                              \
    } while (0)
    switch (g.RS) {
        case 1: AGAN_SN(1, 1); break;
        case 4: AGAN_SN(4, 2); break;
        case 9: AGAN_SN(9, 3); break;
        default: AGAN_SN(16, 4); break;
    }
#undef AGAN_SN
}

bool small_n_wgrad_supported(const Geom& g) { return g.Cout <= 4 && g.OS == 1 && g.R == 3 && g.S == 3 && g.Mtot >= 4096; }

SmallWgradPlan plan_wgrad_small_n(const Geom& g) {
    SmallWgradPlan p;
    // ~2k workgroups in total, at least 8 passes of a workgroup over its chunk: the 27-36 running sums of a lane meet by wave shuffles + LDS once
    // per workgroup, which cost as much as two passes (round 3 ran 2 passes per workgroup on the 128 x 128 head: 70 us where the 256 x 256 one took 133)
    const int want = std::max(1, 2048 / std::max(1, g.Cin));
    p.pchunk = std::max(8192, cdiv(cdiv(g.Mtot, want), 256) * 256);
    p.nchunk = cdiv(g.Mtot, p.pchunk);
    p.slab = ((size_t)g.Cout * g.K + 3) / 4 * 4;
    p.ws_bytes = p.slab * p.nchunk * sizeof(float);
    return p;
}

void launch_wgrad_small_n(const void* x, const float* dy, float* part, const Geom& g, const SmallWgradPlan& p, hipStream_t st, int x_dtype) {
    static const bool wg_off = getenv("AGAN_WGRAD_STRIP_OFF") != nullptr;
    if (strip_ok(g) && (p.pchunk & 255) == 0 && !wg_off) {     // the plan's pixel chunks are whole waves of strips (multiples of 256 pixels)
        const dim3 sg(g.Cin, p.nchunk);
#define AGAN_STRIP(NC_)                                                                                                                   \
    do {                                                                                                                                  \
        if (x_dtype == AGAN_DT_BF16) AGAN_LAUNCH((wgrad_small_strip_kernel<AGAN_DT_BF16, NC_>), sg, dim3(256), 0, st, x, dy, part, g, p.pchunk >> 2, p.slab); \
        else if (x_dtype == AGAN_DT_F16) AGAN_LAUNCH((wgrad_small_strip_kernel<AGAN_DT_F16, NC_>), sg, dim3(256), 0, st, x, dy, part, g, p.pchunk >> 2, p.slab); \
        else AGAN_LAUNCH((wgrad_small_strip_kernel<AGAN_DT_F32, NC_>), sg, dim3(256), 0, st, x, dy, part, g, p.pchunk >> 2, p.slab);    \
    } while (0)
        if (g.Cout <= 2) AGAN_STRIP(2);
        else if (g.Cout == 3) AGAN_STRIP(3);
        else AGAN_STRIP(4);
#undef AGAN_STRIP
        return;
    }
    if (x_dtype == AGAN_DT_BF16) AGAN_LAUNCH((wgrad_small_n_kernel<9, 3, AGAN_DT_BF16>), dim3(g.Cin, p.nchunk), dim3(256), 0, st, x, dy, part, g, p.pchunk, p.slab);
    else if (x_dtype == AGAN_DT_F16) AGAN_LAUNCH((wgrad_small_n_kernel<9, 3, AGAN_DT_F16>), dim3(g.Cin, p.nchunk), dim3(256), 0, st, x, dy, part, g, p.pchunk, p.slab);
    else AGAN_LAUNCH((wgrad_small_n_kernel<9, 3>), dim3(g.Cin, p.nchunk), dim3(256), 0, st, x, dy, part, g, p.pchunk, p.slab);
}

}  // namespace conv
}  // namespace agan
