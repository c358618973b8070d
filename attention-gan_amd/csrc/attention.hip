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

#ifndef AGAN_ATTN_ABLATE
#define AGAN_ATTN_ABLATE 0
#endif

namespace {

constexpr int kMaxC = 64;   // LDS budget for the projected words: kMaxC * TMAX floats
constexpr int kAttLd = 68;  // row stride (floats) of the per-wave transposition tiles of the backward

typedef float f32x4 __attribute__((ext_vector_type(4)));

// LDS traffic between the lanes of ONE wave needs no s_barrier (a wave's LDS instructions execute in order); this only stops the
// compiler from moving LDS accesses across the hand-over.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// proj[b,c,t] = sum_e w[c,e] * words[b,e,t]        (attention.py:50-52: the 1x1 conv on the word axis)
// one wave per (b, c): lanes stride the embedding axis, the T partial sums meet in one transposing butterfly
template <int TMAX>
__global__ __launch_bounds__(256) void attn_proj_kernel(const float* __restrict__ words, const float* __restrict__ w,
                                                        float* __restrict__ proj, int B, int C, int E, int T) {
    const int lane = threadIdx.x & 63;
    const int bc = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (bc >= B * C) return;                                  // wave-uniform
    const int b = bc / C, c = bc - b * C;
    const float* wr = w + (size_t)c * E;
    const float* wd = words + (size_t)b * E * T;
    float part[TMAX];
#pragma unroll
    for (int t = 0; t < TMAX; ++t) part[t] = 0.f;
    for (int e = lane; e < E; e += 64) {
        const float wv = wr[e];
#pragma unroll
        for (int t = 0; t < TMAX; ++t)
            if (t < T) part[t] += wv * wd[(size_t)e * T + t];
    }
    int t_own;
    const float r = wave_sum_scatter<TMAX>(part, lane, t_own);
    if ((lane & (64 / TMAX - 1)) == 0 && t_own < T) proj[(size_t)bc * T + t_own] = r;
}

// DT: storage type of `images` and `ctx` (include/agan.h: AGAN_DT_*); the attention map, projections and all arithmetic stay fp32
template <int TMAX, int DT = AGAN_DT_F32>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const void* __restrict__ images, const float* __restrict__ proj,
                                                       const int64_t* __restrict__ mask, float scale, void* __restrict__ ctx,
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
    const size_t img0 = (size_t)b * C * HW + p;
    float sc[TMAX];
#pragma unroll
    for (int t = 0; t < TMAX; ++t) sc[t] = 0.f;
    for (int c = 0; c < C; ++c) {
        const float v = ld1<DT>(images, img0 + (size_t)c * HW);
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
    for (int c = 0; c < C; ++c) {
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < TMAX; ++t) s += pj[c][t] * sc[t];
        st1<DT>(ctx, img0 + (size_t)c * HW, s);
    }
}

// backward of the streaming pass.  The d(proj) contribution of a workgroup's 256 pixels is reduced WITHOUT atomics so that
// the result is bit-reproducible: a butterfly per channel inside each wave, one LDS row per wave, a fixed-order sum of the
// four rows, and one partial [C][T] slab per workgroup that attn_dproj_reduce_kernel sums in block order.
template <int TMAX, int DT = AGAN_DT_F32>
__global__ __launch_bounds__(256) void attn_bwd_kernel(const void* __restrict__ images, const float* __restrict__ proj,
                                                       const float* __restrict__ attn, const void* __restrict__ dctx,
                                                       const float* __restrict__ dattn, float scale, void* __restrict__ dimages,
                                                       float* __restrict__ dproj_part, int C, int T, int HW) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float (*pj)[TMAX] = reinterpret_cast<float (*)[TMAX]>(smem_raw);                               // [kMaxC][TMAX]
    float (*acc)[kMaxC][TMAX] = reinterpret_cast<float (*)[kMaxC][TMAX]>(smem_raw + sizeof(float) * kMaxC * TMAX);   // [4 waves]
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < C * TMAX; i += 256) {
        const int c = i / TMAX, t = i - c * TMAX;
        pj[c][t] = t < T ? proj[((size_t)b * C + c) * T + t] : 0.f;
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
    const bool dc = dctx != nullptr;
    const size_t base0 = (size_t)b * C * HW + pp;      // element index of (b, channel 0, this pixel) in images / dctx / dimages
    // Both channel loops are LATENCY bound as written one channel at a time (a dependent HBM round trip per iteration, ~6 waves per
    // SIMD to hide it): they run kCU channels per trip with the next trip's loads already in flight.
    constexpr int kCU = 4;
    if (dc) {
        float nx[kCU];
#pragma unroll
        for (int j = 0; j < kCU; ++j) nx[j] = (live && j < C) ? ld1<DT>(dctx, base0 + (size_t)(j) * HW) : 0.f;
        for (int c0 = 0; c0 < C; c0 += kCU) {
            float v[kCU];
#pragma unroll
            for (int j = 0; j < kCU; ++j) v[j] = nx[j];
#pragma unroll
            for (int j = 0; j < kCU; ++j) nx[j] = (live && c0 + kCU + j < C) ? ld1<DT>(dctx, base0 + (size_t)((c0 + kCU + j)) * HW) : 0.f;
#pragma unroll
            for (int j = 0; j < kCU; ++j) {
                if (c0 + j < C) {
#pragma unroll
                    for (int t = 0; t < TMAX; ++t) da[t] += v[j] * pj[c0 + j][t];
                }
            }
        }
    }
    float dot = 0.f;
#pragma unroll
    for (int t = 0; t < TMAX; ++t) dot += a[t] * da[t];
    float ds[TMAX];   // gradient w.r.t. the raw (unscaled) score
#pragma unroll
    for (int t = 0; t < TMAX; ++t) ds[t] = a[t] * (da[t] - dot) * scale;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if constexpr (TMAX == 16) {
        // d(proj)[c][t] = sum over the pixels of dctx[c][p] attn[t][p] + images[c][p] ds[t][p] is a (channels x words x pixels) contraction
        // whose reduction axis is the LANE axis.  As a butterfly it cost 17 cross-lane shuffles per channel and wave -- 270 of the
        // kernel's 375 us at 128x128 (ablation).  Here each wave transposes its 64 pixels through a private LDS
        // tile and lets v_mfma_f32_16x16x4_f32 reduce them: A[i = channel][k = pixel], B[k = pixel][j = word], 16 k-steps per term.
        // Row stride kAttLd = 68 floats: the operand read of lane (i = l & 15, k = l >> 4) hits bank 4 i + k -- conflict-free.
        float* Yw = reinterpret_cast<float*>(smem_raw + sizeof(float) * kMaxC * TMAX * 5) + wave * (48 * kAttLd);   // [2][16][kAttLd]: attn, ds
        float* Xw = Yw + 32 * kAttLd;                                                                            // [16][kAttLd]: one term of 16 channels
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            Yw[t * kAttLd + lane] = a[t];
            Yw[(16 + t) * kAttLd + lane] = ds[t];
        }
        wave_lds_sync();
        const int mi = lane & 15, kq = lane >> 4;
        float Ba[16], Bd[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            Ba[k] = Yw[mi * kAttLd + 4 * k + kq];
            Bd[k] = Yw[(16 + mi) * kAttLd + 4 * k + kq];
        }
        for (int c0 = 0; c0 < C; c0 += 16) {
            float ivr[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const bool on = live && c0 + j < C;
                ivr[j] = on ? ld1<DT>(images, base0 + (size_t)((c0 + j)) * HW) : 0.f;
                Xw[j * kAttLd + lane] = (on && dc) ? ld1<DT>(dctx, base0 + (size_t)((c0 + j)) * HW) : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int c = c0 + j;
                if (c < C) {                                                     // workgroup-uniform
                    float s = 0.f;
#pragma unroll
                    for (int t = 0; t < TMAX; ++t) s += ds[t] * pj[c][t];
                    if (live) st1<DT>(dimages, base0 + (size_t)(c) * HW, s);
                }
            }
            wave_lds_sync();
            f32x4 dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 16; ++k) dp = __builtin_amdgcn_mfma_f32_16x16x4f32(Xw[mi * kAttLd + 4 * k + kq], Ba[k], dp, 0, 0, 0);
            wave_lds_sync();
#pragma unroll
            for (int j = 0; j < 16; ++j) Xw[j * kAttLd + lane] = ivr[j];
            wave_lds_sync();
#pragma unroll
            for (int k = 0; k < 16; ++k) dp = __builtin_amdgcn_mfma_f32_16x16x4f32(Xw[mi * kAttLd + 4 * k + kq], Bd[k], dp, 0, 0, 0);
            wave_lds_sync();
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[wave][c0 + 4 * kq + r][mi] = dp[r];      // D[i = 4 (l >> 4) + r][j = l & 15]
        }
    } else {
    float ivn[kCU], dvn[kCU];
    #pragma unroll
        for (int j = 0; j < kCU; ++j) {
            ivn[j] = (live && j < C) ? ld1<DT>(images, base0 + (size_t)(j) * HW) : 0.f;
            dvn[j] = (live && dc && j < C) ? ld1<DT>(dctx, base0 + (size_t)(j) * HW) : 0.f;
        }
        for (int c0 = 0; c0 < C; c0 += kCU) {
            float ivc[kCU], dvc[kCU];
    #pragma unroll
            for (int j = 0; j < kCU; ++j) {
                ivc[j] = ivn[j];
                dvc[j] = dvn[j];
            }
    #pragma unroll
            for (int j = 0; j < kCU; ++j) {
                const bool more = live && c0 + kCU + j < C;
                ivn[j] = more ? ld1<DT>(images, base0 + (size_t)((c0 + kCU + j)) * HW) : 0.f;
                dvn[j] = (more && dc) ? ld1<DT>(dctx, base0 + (size_t)((c0 + kCU + j)) * HW) : 0.f;
            }
    #pragma unroll
            for (int j = 0; j < kCU; ++j) {
                const int c = c0 + j;
                if (c >= C) break;                                               // workgroup-uniform
                float s = 0.f;
    #pragma unroll
                for (int t = 0; t < TMAX; ++t) s += ds[t] * pj[c][t];
                if (live) st1<DT>(dimages, base0 + (size_t)(c) * HW, s);
    #if AGAN_ATTN_ABLATE != 1
                float part[TMAX];
    #pragma unroll
                for (int t = 0; t < TMAX; ++t) part[t] = dvc[j] * a[t] + ivc[j] * ds[t];
                int t_own;
    #if AGAN_ATTN_ABLATE == 2
                float r = 0.f; t_own = lane & (TMAX - 1);
    #pragma unroll
                for (int t = 0; t < TMAX; ++t) r += part[t];
    #else
                const float r = wave_sum_scatter<TMAX>(part, lane, t_own);       // TMAX + 1 shuffles per channel, not 6 * TMAX
    #endif
                if ((lane & (64 / TMAX - 1)) == 0) acc[wave][c][t_own] = r;
    #endif
            }
        }
}
    __syncthreads();
    float* dst = dproj_part + ((size_t)b * gridDim.x + blockIdx.x) * C * T;
    for (int i = threadIdx.x; i < C * T; i += 256) {
        const int c = i / T, t = i - c * T;
        dst[i] = ((acc[0][c][t] + acc[1][c][t]) + acc[2][c][t]) + acc[3][c][t];
    }
}

// dproj[b][c][t] = sum over the pixel workgroups of their partial slabs, in workgroup order
__global__ __launch_bounds__(256) void attn_dproj_reduce_kernel(const float* __restrict__ part, float* __restrict__ dproj, int B, int nblk, int CT) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * CT) return;
    const int b = i / CT, e = i - b * CT;
    const float* p = part + (size_t)b * nblk * CT + e;
    float s = 0.f;
    for (int k = 0; k < nblk; ++k) s += p[(size_t)k * CT];
    dproj[i] = s;
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
    return agan_attn_fwd_dt(images, words, w, mask, scale, proj, ctx, attn, B, C, E, T, HW, stream, AGAN_DT_F32);
}

int agan_attn_fwd_dt(const void* images, const float* words, const float* w, const int64_t* mask, float scale, float* proj,
                     void* ctx, float* attn, int B, int C, int E, int T, int HW, void* stream, int dtype) {
    AGAN_REQUIRE(images && words && w && mask && proj && ctx && attn, "attn_fwd: null pointer");
    AGAN_REQUIRE(dtype == AGAN_DT_F32 || dtype == AGAN_DT_BF16 || dtype == AGAN_DT_F16, "attn_fwd: storage type %d", dtype);
    AGAN_REQUIRE(B > 0 && C > 0 && E > 0 && T > 0 && HW > 0, "attn_fwd: non-positive dimension");
    AGAN_REQUIRE(C <= kMaxC, "attn_fwd: nc_in %d > %d", C, kMaxC);
    AGAN_REQUIRE(T <= 64, "attn_fwd: seq_len %d > 64", T);
    hipStream_t st = as_stream(stream);
    if (T <= 16) hipLaunchKernelGGL((attn_proj_kernel<16>), dim3(cdiv(B * C, 4)), dim3(256), 0, st, words, w, proj, B, C, E, T);
    else if (T <= 32) hipLaunchKernelGGL((attn_proj_kernel<32>), dim3(cdiv(B * C, 4)), dim3(256), 0, st, words, w, proj, B, C, E, T);
    else hipLaunchKernelGGL((attn_proj_kernel<64>), dim3(cdiv(B * C, 4)), dim3(256), 0, st, words, w, proj, B, C, E, T);
    dim3 grid(cdiv(HW, 256), B);
#define AGAN_ATTN_FWD(TM)                                                                                                                        \
    do {                                                                                                                                         \
        if (dtype == AGAN_DT_BF16) hipLaunchKernelGGL((attn_fwd_kernel<TM, AGAN_DT_BF16>), grid, dim3(256), 0, st, images, proj, mask, scale, ctx, attn, C, T, HW); \
        else if (dtype == AGAN_DT_F16) hipLaunchKernelGGL((attn_fwd_kernel<TM, AGAN_DT_F16>), grid, dim3(256), 0, st, images, proj, mask, scale, ctx, attn, C, T, HW); \
        else hipLaunchKernelGGL((attn_fwd_kernel<TM>), grid, dim3(256), 0, st, images, proj, mask, scale, ctx, attn, C, T, HW);                      \
    } while (0)
    if (T <= 16) AGAN_ATTN_FWD(16);
    else if (T <= 32) AGAN_ATTN_FWD(32);
    else AGAN_ATTN_FWD(64);
#undef AGAN_ATTN_FWD
    return check_launch("attn_fwd");
}

size_t agan_attn_bwd_ws_bytes(int B, int C, int T, int HW) {
    if (B <= 0 || C <= 0 || T <= 0 || HW <= 0) return 0;
    return ((size_t)B * C * T + (size_t)B * cdiv(HW, 256) * C * T) * sizeof(float);       // dproj + one partial slab per workgroup
}

int agan_attn_bwd(const float* images, const float* words, const float* w, const float* proj, const float* attn,
                  const float* dctx, const float* dattn, float scale, float* dimages, float* dwords, float* dw, int B, int C,
                  int E, int T, int HW, int accumulate, void* ws, size_t ws_bytes, void* stream) {
    return agan_attn_bwd_dt(images, words, w, proj, attn, dctx, dattn, scale, dimages, dwords, dw, B, C, E, T, HW, accumulate, ws, ws_bytes,
                            stream, AGAN_DT_F32);
}

int agan_attn_bwd_dt(const void* images, const float* words, const float* w, const float* proj, const float* attn,
                     const void* dctx, const float* dattn, float scale, void* dimages, float* dwords, float* dw, int B, int C,
                     int E, int T, int HW, int accumulate, void* ws, size_t ws_bytes, void* stream, int dtype) {
    AGAN_REQUIRE(images && words && w && proj && attn && dimages && dwords && dw && ws, "attn_bwd: null pointer");
    AGAN_REQUIRE(dtype == AGAN_DT_F32 || dtype == AGAN_DT_BF16 || dtype == AGAN_DT_F16, "attn_bwd: storage type %d", dtype);
    AGAN_REQUIRE(B > 0 && C > 0 && E > 0 && T > 0 && HW > 0, "attn_bwd: non-positive dimension");
    AGAN_REQUIRE(C <= kMaxC && T <= 64, "attn_bwd: nc_in %d / seq_len %d out of range", C, T);
    if (ws_bytes < agan_attn_bwd_ws_bytes(B, C, T, HW)) {
        set_error("attn_bwd: workspace too small");
        return AGAN_EWORKSPACE;
    }
    hipStream_t st = as_stream(stream);
    float* dproj = static_cast<float*>(ws);
    float* part = dproj + (size_t)B * C * T;
    const int nblk = cdiv(HW, 256);
    dim3 grid(nblk, B);
#define AGAN_ATTN_BWD_DT(TM, DT)                                                                                                     \
    do {                                                                                                                             \
        const size_t smem_ = sizeof(float) * (kMaxC * TM * 5 + (TM == 16 ? 4 * 48 * kAttLd : 0));                                    \
        static const hipError_t attr_ = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_kernel<TM, DT>),                  \
                                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_);                 \
        (void)attr_;                                                                                                                 \
        hipLaunchKernelGGL((attn_bwd_kernel<TM, DT>), grid, dim3(256), smem_, st, images, proj, attn, dctx, dattn, scale, dimages, part, \
                           C, T, HW);                                                                                                \
    } while (0)
#define AGAN_ATTN_BWD(TM)                                                       \
    do {                                                                        \
        if (dtype == AGAN_DT_BF16) AGAN_ATTN_BWD_DT(TM, AGAN_DT_BF16);          \
        else if (dtype == AGAN_DT_F16) AGAN_ATTN_BWD_DT(TM, AGAN_DT_F16);       \
        else AGAN_ATTN_BWD_DT(TM, AGAN_DT_F32);                                 \
    } while (0)
    if (T <= 16) AGAN_ATTN_BWD(16);
    else if (T <= 32) AGAN_ATTN_BWD(32);
    else AGAN_ATTN_BWD(64);
#undef AGAN_ATTN_BWD
#undef AGAN_ATTN_BWD_DT
    hipLaunchKernelGGL(attn_dproj_reduce_kernel, dim3(cdiv(B * C * T, 256)), dim3(256), 0, st, part, dproj, B, nblk, C * T);
    hipLaunchKernelGGL(attn_bwd_words_kernel, dim3(cdiv(B * E * T + C * E, 256)), dim3(256), 0, st, words, w, dproj, dwords, dw, B, C, E, T, accumulate);
    return check_launch("attn_bwd");
}

}  // extern "C"
