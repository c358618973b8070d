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

// 3x3 / stride 1 / pad 1 with <= 4 output channels (the RGB heads, generator_submodules.py:135), row strips: a thread owns 4
// consecutive output pixels of one row and loads, per input channel and patch row, ONE aligned 16-byte block plus its left and right
// neighbour pixel -- 9 load instructions and 72 bytes for 4 pixels x 9 taps instead of 36 and 144 (the per-pixel kernel above re-reads
// every input value nine times through the texture path and ran at 1.2 TB/s, round-3 profile).
template <int IDT>
__global__ __launch_bounds__(256) void conv_small_strip_kernel(const void* __restrict__ in, const float* __restrict__ wk,
                                                               const float* __restrict__ bias, float* __restrict__ out, const Geom g) {
    const int W4 = g.IW >> 2;
    const int nstrips = g.B * g.IH * W4;
    const int sid = blockIdx.x * 256 + threadIdx.x;
    const bool valid = sid < nstrips;
    const int ss = valid ? sid : 0;
    const int xs = ss % W4, t1 = ss / W4, y = t1 % g.IH, b = t1 / g.IH;
    const int x0 = xs * 4;
    const int ihw = g.IH * g.IW;
    const __amdgpu_buffer_rsrc_t rin = make_rsrc(in, (size_t)g.B * g.Cin * ihw * (IDT == AGAN_DT_F32 ? 4 : 2));
    unsigned off4[3], offl[3], offr[3];                 // element offsets
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int iy = y + r - 1;
        const bool rok = valid & ((unsigned)iy < (unsigned)g.IH);
        const unsigned base = (unsigned)(b * g.Cin * ihw + iy * g.IW + x0);
        off4[r] = rok ? base : kOOB;
        offl[r] = (rok & (x0 > 0)) ? base - 1u : kOOB;
        offr[r] = (rok & (x0 + 4 < g.IW)) ? base + 4u : kOOB;
    }
    float acc[4][4];                                    // [pixel][cout]
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[p][n] = 0.f;
    for (int c = 0; c < g.Cin; ++c) {
        const unsigned soff = (unsigned)(c * ihw);
        const float* wr = wk + (size_t)c * 9 * g.Nld;
        float v[3][6];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            buf_load4_dt<IDT>(rin, off4[r], soff, &v[r][1]);
            v[r][0] = buf_load1_dt<IDT>(rin, offl[r], soff);
            v[r][5] = buf_load1_dt<IDT>(rin, offr[r], soff);
        }
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const float4 w = *reinterpret_cast<const float4*>(wr + (size_t)(r * 3 + t) * g.Nld);      // wave-uniform
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const float x = v[r][p + t];
                    acc[p][0] += x * w.x; acc[p][1] += x * w.y; acc[p][2] += x * w.z; acc[p][3] += x * w.w;
                }
            }
    }
    if (!valid) return;
    float* o = out + ((size_t)b * g.Cout * g.IH + y) * g.IW + x0;
#pragma unroll
    for (int n = 0; n < 4; ++n)
        if (n < g.Cout) {
            const float bv = bias ? bias[n] : 0.f;
            *reinterpret_cast<float4*>(o + (size_t)n * ihw) = make_float4(acc[0][n] + bv, acc[1][n] + bv, acc[2][n] + bv, acc[3][n] + bv);
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

// weight gradient of the same layers, row strips: per 4 pixels a thread loads the 3 dy blocks and, per patch row of x, one
// 16-byte block + two neighbours -- 12 loads instead of 48.  One workgroup = one input channel x one chunk of strips, as above.
template <int XDT>
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
    float acc[9][4];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[t][n] = 0.f;
    for (int sid = sbeg + threadIdx.x; sid < send; sid += 256) {
        const int xs = sid % W4, t1 = sid / W4, y = t1 % g.IH, b = t1 / g.IH;
        const int x0 = xs * 4;
        float d[4][4];                                  // [cout][pixel]
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const f32x4 m = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                rd, n < g.Cout ? (unsigned)(((b * g.Cout + n) * g.IH + y) * g.IW + x0) * 4u : kOOB, 0, 0));
            d[n][0] = m[0]; d[n][1] = m[1]; d[n][2] = m[2]; d[n][3] = m[3];
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int iy = y + r - 1;
            const bool rok = (unsigned)iy < (unsigned)g.IH;
            const unsigned base = (unsigned)(b * g.Cin * ihw + iy * g.IW + x0);
            float v[6];
            buf_load4_dt<XDT>(rx, rok ? base : kOOB, soff, &v[1]);
            v[0] = buf_load1_dt<XDT>(rx, (rok & (x0 > 0)) ? base - 1u : kOOB, soff);
            v[5] = buf_load1_dt<XDT>(rx, (rok & (x0 + 4 < g.IW)) ? base + 4u : kOOB, soff);
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    float a = acc[r * 3 + t][n];
#pragma unroll
                    for (int p = 0; p < 4; ++p) a += v[p + t] * d[n][p];
                    acc[r * 3 + t][n] = a;
                }
        }
    }
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const float sum = wave_sum(acc[t][n]);
            if (lane == 0) red[w][t * 4 + n] = sum;
        }
    __syncthreads();
    if (threadIdx.x < 36) {
        const int t = threadIdx.x >> 2, n = threadIdx.x & 3;
        if (n < g.Cout)
            part[(size_t)chunk * slab + (size_t)n * g.K + c * 9 + t] =
                red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    }
}

// the row-strip kernels take a 3x3 / stride 1 / pad 1 layer on whole 4-pixel blocks
bool strip_ok(const Geom& g) {
    static const bool off = getenv("AGAN_SMALL_STRIP_OFF") != nullptr;
    return !off && g.RS == 9 && g.S == 3 && g.OS == 1 && g.SY == 1 && g.DY == 1 && g.OY0 == -1 && g.IH == g.OH && g.IW == g.OW && (g.IW & 3) == 0 && g.Cout <= 4;
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
        const dim3 sg(cdiv(g.B * g.IH * (g.IW >> 2), 256));
        if (in_dtype == AGAN_DT_BF16) AGAN_LAUNCH((conv_small_strip_kernel<AGAN_DT_BF16>), sg, dim3(256), 0, st, in, wk, bias, out, g);
        else if (in_dtype == AGAN_DT_F16) AGAN_LAUNCH((conv_small_strip_kernel<AGAN_DT_F16>), sg, dim3(256), 0, st, in, wk, bias, out, g);
        else AGAN_LAUNCH((conv_small_strip_kernel<AGAN_DT_F32>), sg, dim3(256), 0, st, in, wk, bias, out, g);
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
    const int want = std::max(1, 8192 / std::max(1, g.Cin));                 // ~8k workgroups in total
    p.pchunk = std::max(2048, cdiv(cdiv(g.Mtot, want), 256) * 256);
    p.nchunk = cdiv(g.Mtot, p.pchunk);
    p.slab = ((size_t)g.Cout * g.K + 3) / 4 * 4;
    p.ws_bytes = p.slab * p.nchunk * sizeof(float);
    return p;
}

void launch_wgrad_small_n(const void* x, const float* dy, float* part, const Geom& g, const SmallWgradPlan& p, hipStream_t st, int x_dtype) {
    if (strip_ok(g) && (p.pchunk & 3) == 0) {       // the plan's pixel chunks are whole strips (multiples of 256 pixels)
        const dim3 sg(g.Cin, p.nchunk);
        if (x_dtype == AGAN_DT_BF16) AGAN_LAUNCH((wgrad_small_strip_kernel<AGAN_DT_BF16>), sg, dim3(256), 0, st, x, dy, part, g, p.pchunk >> 2, p.slab);
        else if (x_dtype == AGAN_DT_F16) AGAN_LAUNCH((wgrad_small_strip_kernel<AGAN_DT_F16>), sg, dim3(256), 0, st, x, dy, part, g, p.pchunk >> 2, p.slab);
        else AGAN_LAUNCH((wgrad_small_strip_kernel<AGAN_DT_F32>), sg, dim3(256), 0, st, x, dy, part, g, p.pchunk >> 2, p.slab);
        return;
    }
    if (x_dtype == AGAN_DT_BF16) AGAN_LAUNCH((wgrad_small_n_kernel<9, 3, AGAN_DT_BF16>), dim3(g.Cin, p.nchunk), dim3(256), 0, st, x, dy, part, g, p.pchunk, p.slab);
    else if (x_dtype == AGAN_DT_F16) AGAN_LAUNCH((wgrad_small_n_kernel<9, 3, AGAN_DT_F16>), dim3(g.Cin, p.nchunk), dim3(256), 0, st, x, dy, part, g, p.pchunk, p.slab);
    else AGAN_LAUNCH((wgrad_small_n_kernel<9, 3>), dim3(g.Cin, p.nchunk), dim3(256), 0, st, x, dy, part, g, p.pchunk, p.slab);
}

}  // namespace conv
}  // namespace agan
