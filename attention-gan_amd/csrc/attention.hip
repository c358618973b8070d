// Word-context attention (AttentionModule.forward, networks/attention.py:25-79) as ONE streaming pass per direction.
//
// The reference materialises images^T (contiguous copy), the [B*hw, T] score matrix, a repeat_interleave'd int64 mask
// (31 MB at 128x128), the masked copy, the softmax, its transpose copy and the context: ~9 full-size HBM round trips.
// Here each lane owns one pixel: it streams the C image channels once (coalesced along the NCHW pixel axis), keeps the
// T scores in registers, and writes attn[T] + ctx[C] once.  The projected words (C x T, <= 8 KB) sit in LDS, zero-padded
// to the compile-time TMAX so the inner loops have no bounds checks; every LDS read is a broadcast.
// Algorithmic bytes/pixel: 4*(C read + C write + T write); HBM-bound (116 MB at gen3).
#include "agan_common.h"

using namespace agan;

namespace {

constexpr int kMaxC = 64;   // LDS budget for the projected words: kMaxC * TMAX floats

// proj[b,c,t] = sum_e w[c,e] * words[b,e,t]        (attention.py:50-52: the 1x1 conv on the word axis)
__global__ __launch_bounds__(256) void attn_proj_kernel(const float* __restrict__ words, const float* __restrict__ w,
                                                        float* __restrict__ proj, int B, int C, int E, int T) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * C * T) return;
    const int t = i % T, c = (i / T) % C, b = i / (T * C);
    const float* wr = w + (size_t)c * E;
    const float* wd = words + (size_t)b * E * T + t;
    float s = 0.f;
    for (int e = 0; e < E; ++e) s += wr[e] * wd[(size_t)e * T];
    proj[i] = s;
}

template <int TMAX>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const float* __restrict__ images, const float* __restrict__ proj,
                                                       const int64_t* __restrict__ mask, float scale, float* __restrict__ ctx,
                                                       float* __restrict__ attn, int C, int T, int HW) {
    __shared__ float pj[kMaxC][TMAX];
    __shared__ float neg[TMAX];   // 0 for a live word, -inf for masked / padded
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < C * TMAX; i += 256) {
        const int c = i / TMAX, t = i - c * TMAX;
        pj[c][t] = t < T ? proj[((size_t)b * C + c) * T + t] : 0.f;
    }
    for (int t = threadIdx.x; t < TMAX; t += 256) neg[t] = (t < T && mask[(size_t)b * T + t] != 0) ? 0.f : -INFINITY;
    __syncthreads();
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    const float* img = images + (size_t)b * C * HW + p;
    float sc[TMAX];
#pragma unroll
    for (int t = 0; t < TMAX; ++t) sc[t] = 0.f;
    for (int c = 0; c < C; ++c) {
        const float v = img[(size_t)c * HW];
#pragma unroll
        for (int t = 0; t < TMAX; ++t) sc[t] += v * pj[c][t];
    }
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < TMAX; ++t) {
        sc[t] = sc[t] * scale + neg[t];
        mx = fmaxf(mx, sc[t]);
    }
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < TMAX; ++t) {
        sc[t] = __expf(sc[t] - mx);   // all-masked row: exp(-inf - -inf) = NaN, like the reference
        sum += sc[t];
    }
    const float inv = 1.f / sum;
    float* ao = attn + (size_t)b * T * HW + p;
#pragma unroll
    for (int t = 0; t < TMAX; ++t) {
        sc[t] *= inv;
        if (t < T) ao[(size_t)t * HW] = sc[t];
    }
    float* co = ctx + (size_t)b * C * HW + p;
    for (int c = 0; c < C; ++c) {
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < TMAX; ++t) s += pj[c][t] * sc[t];
        co[(size_t)c * HW] = s;
    }
}

// backward of the streaming pass.  dproj is accumulated per block in LDS, then one global atomic per (c,t) per block.
template <int TMAX>
__global__ __launch_bounds__(256) void attn_bwd_kernel(const float* __restrict__ images, const float* __restrict__ proj,
                                                       const float* __restrict__ attn, const float* __restrict__ dctx,
                                                       const float* __restrict__ dattn, float scale, float* __restrict__ dimages,
                                                       float* __restrict__ dproj, int C, int T, int HW) {
    __shared__ float pj[kMaxC][TMAX];
    __shared__ float acc[kMaxC][TMAX];
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < C * TMAX; i += 256) {
        const int c = i / TMAX, t = i - c * TMAX;
        pj[c][t] = t < T ? proj[((size_t)b * C + c) * T + t] : 0.f;
        acc[c][t] = 0.f;
    }
    __syncthreads();
    const int p = blockIdx.x * 256 + threadIdx.x;
    const bool live = p < HW;
    const size_t pp = live ? p : 0;
    float a[TMAX], da[TMAX];
#pragma unroll
    for (int t = 0; t < TMAX; ++t) {
        a[t] = (live && t < T) ? attn[((size_t)b * T + t) * HW + pp] : 0.f;
        da[t] = (live && t < T && dattn) ? dattn[((size_t)b * T + t) * HW + pp] : 0.f;
    }
    const float* dc = dctx ? dctx + (size_t)b * C * HW + pp : nullptr;
    if (dc) {
        for (int c = 0; c < C; ++c) {
            const float v = live ? dc[(size_t)c * HW] : 0.f;
#pragma unroll
            for (int t = 0; t < TMAX; ++t) da[t] += v * pj[c][t];
        }
    }
    float dot = 0.f;
#pragma unroll
    for (int t = 0; t < TMAX; ++t) dot += a[t] * da[t];
    float ds[TMAX];   // gradient w.r.t. the raw (unscaled) score
#pragma unroll
    for (int t = 0; t < TMAX; ++t) ds[t] = a[t] * (da[t] - dot) * scale;
    const float* img = images + (size_t)b * C * HW + pp;
    float* di = dimages + (size_t)b * C * HW + pp;
    const int lane = threadIdx.x & 63;
    for (int c = 0; c < C; ++c) {
        const float iv = live ? img[(size_t)c * HW] : 0.f;
        const float dv = (live && dc) ? dc[(size_t)c * HW] : 0.f;
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < TMAX; ++t) s += ds[t] * pj[c][t];
        if (live) di[(size_t)c * HW] = s;
        float part[TMAX];
#pragma unroll
        for (int t = 0; t < TMAX; ++t) part[t] = dv * a[t] + iv * ds[t];
        int t_own;
        const float r = wave_sum_scatter<TMAX>(part, lane, t_own);       // TMAX + 1 shuffles per channel, not 6 * TMAX
        if ((lane & (64 / TMAX - 1)) == 0 && t_own < T) atomicAdd(&acc[c][t_own], r);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C * T; i += 256) {
        const int c = i / T, t = i - c * T;
        atomicAdd(&dproj[((size_t)b * C + c) * T + t], acc[c][t]);
    }
}

// dwords[b,e,t] = sum_c w[c,e] dproj[b,c,t];   dw[c,e] = sum_{b,t} dproj[b,c,t] words[b,e,t]
__global__ __launch_bounds__(256) void attn_bwd_words_kernel(const float* __restrict__ words, const float* __restrict__ w,
                                                             const float* __restrict__ dproj, float* __restrict__ dwords,
                                                             float* __restrict__ dw, int B, int C, int E, int T, int accumulate) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int nw = B * E * T;
    if (i < nw) {
        const int t = i % T, e = (i / T) % E, b = i / (T * E);
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += w[(size_t)c * E + e] * dproj[((size_t)b * C + c) * T + t];
        dwords[i] = s;
    } else if (i < nw + C * E) {
        const int j = i - nw, e = j % E, c = j / E;
        float s = 0.f;
        for (int b = 0; b < B; ++b)
            for (int t = 0; t < T; ++t) s += dproj[((size_t)b * C + c) * T + t] * words[((size_t)b * E + e) * T + t];
        dw[j] = accumulate ? dw[j] + s : s;
    }
}

}  // namespace

extern "C" {

int agan_attn_fwd(const float* images, const float* words, const float* w, const int64_t* mask, float scale, float* proj,
                  float* ctx, float* attn, int B, int C, int E, int T, int HW, void* stream) {
    AGAN_REQUIRE(images && words && w && mask && proj && ctx && attn, "attn_fwd: null pointer");
    AGAN_REQUIRE(B > 0 && C > 0 && E > 0 && T > 0 && HW > 0, "attn_fwd: non-positive dimension");
    AGAN_REQUIRE(C <= kMaxC, "attn_fwd: nc_in %d > %d", C, kMaxC);
    AGAN_REQUIRE(T <= 64, "attn_fwd: seq_len %d > 64", T);
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(attn_proj_kernel, dim3(cdiv(B * C * T, 256)), dim3(256), 0, st, words, w, proj, B, C, E, T);
    dim3 grid(cdiv(HW, 256), B);
    if (T <= 16) hipLaunchKernelGGL((attn_fwd_kernel<16>), grid, dim3(256), 0, st, images, proj, mask, scale, ctx, attn, C, T, HW);
    else if (T <= 32) hipLaunchKernelGGL((attn_fwd_kernel<32>), grid, dim3(256), 0, st, images, proj, mask, scale, ctx, attn, C, T, HW);
    else hipLaunchKernelGGL((attn_fwd_kernel<64>), grid, dim3(256), 0, st, images, proj, mask, scale, ctx, attn, C, T, HW);
    return check_launch("attn_fwd");
}

size_t agan_attn_bwd_ws_bytes(int B, int C, int T) { return (size_t)B * C * T * sizeof(float); }

int agan_attn_bwd(const float* images, const float* words, const float* w, const float* proj, const float* attn,
                  const float* dctx, const float* dattn, float scale, float* dimages, float* dwords, float* dw, int B, int C,
                  int E, int T, int HW, int accumulate, void* ws, size_t ws_bytes, void* stream) {
    AGAN_REQUIRE(images && words && w && proj && attn && dimages && dwords && dw && ws, "attn_bwd: null pointer");
    AGAN_REQUIRE(C <= kMaxC && T <= 64, "attn_bwd: nc_in %d / seq_len %d out of range", C, T);
    if (ws_bytes < agan_attn_bwd_ws_bytes(B, C, T)) {
        set_error("attn_bwd: workspace too small");
        return AGAN_EWORKSPACE;
    }
    hipStream_t st = as_stream(stream);
    float* dproj = static_cast<float*>(ws);
    if (hipMemsetAsync(dproj, 0, agan_attn_bwd_ws_bytes(B, C, T), st) != hipSuccess) {
        set_error("attn_bwd: memset failed");
        return AGAN_ELAUNCH;
    }
    dim3 grid(cdiv(HW, 256), B);
    if (T <= 16) hipLaunchKernelGGL((attn_bwd_kernel<16>), grid, dim3(256), 0, st, images, proj, attn, dctx, dattn, scale, dimages, dproj, C, T, HW);
    else if (T <= 32) hipLaunchKernelGGL((attn_bwd_kernel<32>), grid, dim3(256), 0, st, images, proj, attn, dctx, dattn, scale, dimages, dproj, C, T, HW);
    else hipLaunchKernelGGL((attn_bwd_kernel<64>), grid, dim3(256), 0, st, images, proj, attn, dctx, dattn, scale, dimages, dproj, C, T, HW);
    hipLaunchKernelGGL(attn_bwd_words_kernel, dim3(cdiv(B * E * T + C * E, 256)), dim3(256), 0, st, words, w, dproj, dwords, dw, B, C, E, T, accumulate);
    return check_launch("attn_bwd");
}

}  // extern "C"
