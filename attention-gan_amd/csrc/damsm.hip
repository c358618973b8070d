// DAMSM losses: WordsLoss.get_loss (losses/words_loss.py:29-102, which loops func_attention networks/attention.py:82-120
// over the B captions) and SentenceLoss.get_loss (losses/sentence_loss.py:12-50), each as one forward and one backward kernel.
//
// words loss: one 320-thread workgroup per (image j, caption i) pair -- B*B workgroups, enough to fill the chip at B=24.
// Thread r owns image region r (S = 17*17 = 289 <= 320) for everything indexed by region, thread d owns embedding
// channel d (D <= 320) for everything indexed by channel; the three small matrices they exchange (caption words e[D][T],
// region attention a2[S][T], weighted context c[D][T]) live in LDS with a +1 pad so both row and column walks are
// conflict-free.  The backward recomputes the forward of its pair instead of storing B*B*(S*T + D*T) intermediates.
// Region features are read coalesced along r (NCHW pixel axis) by the region-owning threads.
#include "agan_common.h"

using namespace agan;

#ifndef AGAN_PAIR_UNROLL_N
#define AGAN_PAIR_UNROLL_N 4
#endif
#define AGAN_PRAGMA_(x) _Pragma(#x)
#define AGAN_PRAGMA(x) AGAN_PRAGMA_(x)
#define AGAN_PAIR_UNROLL AGAN_PRAGMA(unroll AGAN_PAIR_UNROLL_N)
// (the d loops of the MFMA reductions below carry no unroll count: with a runtime trip count and a convergent instruction inside, the compiler refuses
//  a counted unroll -- the pragma they used to carry only produced three "loop not unrolled" warnings per build)
#ifndef AGAN_PAIR_MFMA_WAVES
#define AGAN_PAIR_MFMA_WAVES 4
#endif

namespace {

constexpr int kNT = 320;      // threads per pair workgroup (5 waves): >= S and >= D
constexpr int kMaxB = 128;

// Everything the forward of one (image j, caption i) pair produces, left in LDS / registers for the caller.
// DR = rows of the channel-indexed matrices (>= D).  The <12, 256> instance (seq_len <= 12, nef <= 256: the metric config) is
// 43 KB, so three pair workgroups share a CU and the B*B = 576 pairs of batch 24 run in ONE round of 768 slots; the 65 KB
// <16, 320> instance fits two per CU = 512 slots, i.e. a second round for the last 64 pairs.
template <int TMAX, int DR = kNT>
struct PairSmem {
    // Row stride: the hot loops read whole rows that every lane shares (broadcast), so rows that are a multiple of 16 bytes let
    // them go as ds_read_b128 (3 per row at TMAX = 12 instead of 12 ds_read_b32).  A 12-float stride also keeps the
    // one-row-per-lane accesses at the 8-cycle minimum of a 1 KB wave access; 16- and 32-float strides would not, hence the pad.
    static constexpr int LDW = (TMAX % 8 == 4) ? TMAX : TMAX + 1;
    alignas(16) float e[DR][LDW];     // caption words   e[d][w]  (0 for w >= L)
    alignas(16) float a2[kNT][LDW];   // attention       a2[r][w] (softmax over regions of gamma1 * softmax over words)
    alignas(16) float c[DR][LDW];     // weighted context c[d][w]
    float num[TMAX], n1[TMAX], n2[TMAX], cosv[TMAX];
    float colmax[TMAX], colsum[TMAX];
    float red[8];
};

// Forward of a pair.  On return (after the trailing barrier): sm.e, sm.a2, sm.c, sm.cosv/num/n1/n2 are valid;
// a1[] holds thread r's first-softmax row (valid for r < S).
template <int TMAX, int DR>
__device__ __forceinline__ void pair_forward(PairSmem<TMAX, DR>& sm, const float* __restrict__ fj, const float* __restrict__ fjT,
                                             const float* __restrict__ ei, int D, int T, int S, int L, float scale, float gamma1,
                                             float eps, float (&a1)[TMAX]) {
    // fjT: the same features transposed to [S][D] (or null).  A thread that owns a CHANNEL walks the regions; on the NCHW layout
    // its wave touches 64 different 128-byte lines per load and the 15 resident waves of a CU evict each other's lines from the
    // 32 KB L1 long before a lane has used the other 31 floats of its line (measured: the pair backward spent two thirds of its
    // 318 us in the two channel-owned loops).  On the transposed copy the same load is one 256-byte wave access.
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    // 1. caption words into LDS
    for (int i = tid; i < D * TMAX; i += kNT) {
        const int d = i / TMAX, w = i - d * TMAX;
        sm.e[d][w] = w < L ? ei[(size_t)d * T + w] : 0.f;
    }
    __syncthreads();
    // 2. scores for region r = tid, softmax over words (attention.py:99-104)
    const int r = tid;
    const bool rlive = r < S;
#pragma unroll
    for (int w = 0; w < TMAX; ++w) a1[w] = 0.f;
    if (rlive) {
AGAN_PAIR_UNROLL
        for (int d = 0; d < D; ++d) {
            const float fv = fj[(size_t)d * S + r];
#pragma unroll
            for (int w = 0; w < TMAX; ++w) a1[w] += fv * sm.e[d][w];
        }
        float mx = -INFINITY;
#pragma unroll
        for (int w = 0; w < TMAX; ++w) {
            a1[w] = w < L ? a1[w] * scale : -INFINITY;
            mx = fmaxf(mx, a1[w]);
        }
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < TMAX; ++w) {
            a1[w] = __expf(a1[w] - mx);
            sum += a1[w];
        }
        const float inv = 1.f / sum;
#pragma unroll
        for (int w = 0; w < TMAX; ++w) {
            a1[w] *= inv;
            sm.a2[r][w] = a1[w] * gamma1;          // staged for the region softmax (attention.py:111)
        }
    }
    __syncthreads();
    // 3. softmax over regions, one word column per wave (attention.py:112)
    for (int w = wave; w < L; w += kNT / 64) {
        float mx = -INFINITY;
        for (int q = lane; q < S; q += 64) mx = fmaxf(mx, sm.a2[q][w]);
        mx = wave_max(mx);
        float s = 0.f;
        for (int q = lane; q < S; q += 64) s += __expf(sm.a2[q][w] - mx);
        s = wave_sum(s);
        if (lane == 0) { sm.colmax[w] = mx; sm.colsum[w] = s; }
    }
    __syncthreads();
    if (rlive) {
#pragma unroll
        for (int w = 0; w < TMAX; ++w) sm.a2[r][w] = w < L ? __expf(sm.a2[r][w] - sm.colmax[w]) / sm.colsum[w] : 0.f;
    }
    __syncthreads();
    // 4. weighted context c[d][w] = sum_r f[d][r] a2[r][w]   (attention.py:119), thread d = tid
    if (tid < D) {
        float acc[TMAX];
#pragma unroll
        for (int w = 0; w < TMAX; ++w) acc[w] = 0.f;
        #ifdef AGAN_PAIR_NO_T
        const float* fr = fj + (size_t)tid * S;
        const int fs = 1;
#else
        const float* fr = fjT ? fjT + tid : fj + (size_t)tid * S;
        const int fs = fjT ? D : 1;
#endif
AGAN_PAIR_UNROLL
        for (int q = 0; q < S; ++q) {
            const float fv = fr[(size_t)q * fs];
#pragma unroll
            for (int w = 0; w < TMAX; ++w) acc[w] += fv * sm.a2[q][w];
        }
#pragma unroll
        for (int w = 0; w < TMAX; ++w) sm.c[tid][w] = acc[w];
    }
    __syncthreads();
    // 5. cosine per word (words_loss.py:20-27,72)
    for (int w = wave; w < L; w += kNT / 64) {
        float nu = 0.f, s1 = 0.f, s2 = 0.f;
        for (int d = lane; d < D; d += 64) {
            const float ev = sm.e[d][w], cv = sm.c[d][w];
            nu += ev * cv; s1 += ev * ev; s2 += cv * cv;
        }
        nu = wave_sum(nu); s1 = wave_sum(s1); s2 = wave_sum(s2);
        if (lane == 0) {
            const float a = sqrtf(s1), b = sqrtf(s2);
            sm.num[w] = nu; sm.n1[w] = a; sm.n2[w] = b;
            sm.cosv[w] = nu / fmaxf(a * b, eps);
        }
    }
    __syncthreads();
}

template <int TMAX, int DR>
__global__ __launch_bounds__(kNT) void words_pair_fwd_kernel(const float* __restrict__ feat, const float* __restrict__ featT,
                                                             const float* __restrict__ wemb,
                                                             const int64_t* __restrict__ lens, float gamma1, float gamma2, float gamma3,
                                                             float* __restrict__ sim, float* __restrict__ maps, int B, int D, int T, int S) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    PairSmem<TMAX, DR>& sm = *reinterpret_cast<PairSmem<TMAX, DR>*>(smem_raw);
    const int j = blockIdx.x, i = blockIdx.y;      // image j, caption i
    const int L = min((int)lens[i], T);
    float a1[TMAX];
    pair_forward<TMAX, DR>(sm, feat + (size_t)j * D * S, featT + (size_t)j * D * S, wemb + (size_t)i * D * T, D, T, S, L,
                           rsqrtf((float)D), gamma1, 1e-8f, a1);
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int w = 0; w < L; ++w) s += expf(gamma2 * sm.cosv[w]);     // words_loss.py:77-79
        sim[(size_t)j * B + i] = logf(s) * gamma3;                      // :93
    }
    if (i == j && threadIdx.x < S) {                                    // att_maps.append(attn[i]), :63
        for (int w = 0; w < L; ++w) maps[((size_t)i * T + w) * S + threadIdx.x] = sm.a2[threadIdx.x][w];
    }
}

// ---------------------------------------------------------------------------------------------------------
// MFMA variant of the pair kernels for the metric shapes: seq_len <= 12, nef a multiple of 16 up to 256, regions <= 292.
//
// The region x word contractions  scores[r][w] = sum_d f[d][r] e[d][w],  ctx[d][w] = sum_r f[d][r] a2[r][w]  and their three
// backward counterparts run on v_mfma_f32_16x16x4_f32 (exact fp32 products): the 16-wide j axis holds the <= 12 words, the feature
// operand comes straight from global memory (NCHW rows for a region-indexed result, the transposed copy for a channel-indexed
// one: either way a lane group reads 64 consecutive bytes) and the [.][word] operand from LDS.  Why MFMA although fp32 MFMA is only
// twice the vector ALU rate: in the thread-per-row form every thread re-reads the shared 12-float row (3 ds_read_b128 per 12 FMAs) and
// the five waves of a pair saturate the LDS pipe (round-2 analysis: ~190 K LDS cycles per pair); a 16x16x4 tile reads that row once
// per 16 rows.  12-float rows put lane (i, k) of an operand read on bank 12 i + k: all 64 banks, conflict-free.
// Reductions stay in a fixed order (k ascending inside a tile, tiles independent): bit-reproducible like the vector form.
// ---------------------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kMW = 12;        // LDS row width (words)
constexpr int kMD = 256;       // channel rows
constexpr int kMS = 292;       // region rows: 17 * 17 = 289 padded to a multiple of 4 (the k axis of the channel-indexed contractions)

template <bool BWD>
struct PairMfmaSmem {
    alignas(16) float e[kMD][kMW];      // caption words   e[d][w]  (0 for w >= L)
    alignas(16) float c[kMD][kMW];      // weighted context c[d][w]; backward: d c, then the region part of d e
    alignas(16) float a2[kMS][kMW];     // attention a2[r][w] (rows >= S zero)
    alignas(16) float ds[BWD ? kMS : 1][kMW];   // backward: d a2 (scratch), then d(raw score) (rows >= S zero)
    float num[kMW], n1[kMW], n2[kMW], cosv[kMW];
    float colmax[kMW], colsum[kMW];
    float dotw[kNT / 64][kMW];
};

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// Feature operand of a wave: FOUR 16-row tiles at once whose rows interleave -- tile q, row i <-> feature row base + 4 i + q -- so that
// the A values of lane (i, k) for the four tiles are ONE aligned 16-byte load (a dword per tile and MFMA cost four times the cache-line
// requests and made the first MFMA version 2.5x SLOWER than the vector kernels).
//
// out[r][w] = sum_d f[d][r] * bm[d][w]   (rows r < S written, columns < kMW); fjP = features [D][kMS], rows zero-padded and 16-B aligned;
// wave v owns regions 64 v .. 64 v + 63
__device__ __forceinline__ void mfma_region_rows(const float* __restrict__ fjP, const float (*bm)[kMW], float (*out)[kMW], int D, int S) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, mi = lane & 15, kq = lane >> 4;
    f32x4 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int col = min(wave * 64 + 4 * mi, kMS - 4);          // (regions beyond the padded row: clamped address, never stored)
    const float* fp = fjP + (size_t)kq * kMS + col;
    for (int d0 = 0; d0 < D; d0 += 4) {
        const float b = bm[d0 + kq][mi];                       // B[k = d][j = w]   (columns 12..15 read the next row: never stored)
        const f32x4 a = *reinterpret_cast<const f32x4*>(fp + (size_t)d0 * kMS);      // A[i][k = d] of the four tiles
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = mfma4(a[q], b, acc[q]);
    }
    if (mi < kMW) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wave * 64 + 4 * (4 * kq + r) + q;             // D[i = 4 (l >> 4) + r][j = l & 15] of tile q
                if (row < S) out[row][mi] = acc[q][r];
            }
    }
}

// out[d][w] = sum_r f[d][r] * bm[r][w]  over the kMS padded regions (rows >= S of bm are zero); fjT = features as [S][D];
// wave v < D / 64 owns channels 64 v .. 64 v + 63
__device__ __forceinline__ void mfma_channel_rows(const float* __restrict__ fjT, const float (*bm)[kMW], float (*out)[kMW], int D, int S) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, mi = lane & 15, kq = lane >> 4;
    if (wave * 64 >= D) return;                                  // (wave-uniform; no barrier inside)
    f32x4 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* ft = fjT + wave * 64 + 4 * mi;
    for (int r0 = 0; r0 < kMS; r0 += 4) {
        const float b = bm[r0 + kq][mi];                        // B[k = r][j = w]
        const f32x4 a = *reinterpret_cast<const f32x4*>(ft + (size_t)min(r0 + kq, S - 1) * D);   // A[i][k = r] of the four tiles
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = mfma4(a[q], b, acc[q]);
    }
    if (mi < kMW) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) out[wave * 64 + 4 * (4 * kq + r) + q][mi] = acc[q][r];
    }
}

// Forward of a pair (same contract as pair_forward): on return sm.e, sm.a2, sm.c, sm.cosv/num/n1/n2 are valid, a1[] is thread r's
// first-softmax row (valid for r < S).
template <bool BWD>
__device__ __forceinline__ void pair_forward_mfma(PairMfmaSmem<BWD>& sm, const float* __restrict__ fjP, const float* __restrict__ fjT,
                                                  const float* __restrict__ ei, int D, int T, int S, int L, float scale, float gamma1,
                                                  float eps, float (&a1)[kMW]) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < D * kMW; i += kNT) {
        const int d = i / kMW, w = i - d * kMW;
        sm.e[d][w] = w < L ? ei[(size_t)d * T + w] : 0.f;
    }
    __syncthreads();
    mfma_region_rows(fjP, sm.e, sm.a2, D, S);                   // raw scores (attention.py:99)
    __syncthreads();
    const int r = tid;
    const bool rlive = r < S;
#pragma unroll
    for (int w = 0; w < kMW; ++w) a1[w] = 0.f;
    if (rlive) {
        float mx = -INFINITY;
#pragma unroll
        for (int w = 0; w < kMW; ++w) {
            a1[w] = w < L ? sm.a2[r][w] * scale : -INFINITY;
            mx = fmaxf(mx, a1[w]);
        }
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < kMW; ++w) {
            a1[w] = __expf(a1[w] - mx);
            sum += a1[w];
        }
        const float inv = 1.f / sum;
#pragma unroll
        for (int w = 0; w < kMW; ++w) {
            a1[w] *= inv;
            sm.a2[r][w] = a1[w] * gamma1;
        }
    } else if (r < kMS) {
#pragma unroll
        for (int w = 0; w < kMW; ++w) sm.a2[r][w] = 0.f;        // the padded tail of the k axis
    }
    __syncthreads();
    for (int w = wave; w < L; w += kNT / 64) {                  // softmax over regions, one word column per wave (attention.py:112)
        float mx = -INFINITY;
        for (int q = lane; q < S; q += 64) mx = fmaxf(mx, sm.a2[q][w]);
        mx = wave_max(mx);
        float s = 0.f;
        for (int q = lane; q < S; q += 64) s += __expf(sm.a2[q][w] - mx);
        s = wave_sum(s);
        if (lane == 0) { sm.colmax[w] = mx; sm.colsum[w] = s; }
    }
    __syncthreads();
    if (rlive) {
#pragma unroll
        for (int w = 0; w < kMW; ++w) sm.a2[r][w] = w < L ? __expf(sm.a2[r][w] - sm.colmax[w]) / sm.colsum[w] : 0.f;
    }
    __syncthreads();
    mfma_channel_rows(fjT, sm.a2, sm.c, D, S);                  // weighted context (attention.py:119)
    __syncthreads();
    for (int w = wave; w < L; w += kNT / 64) {                  // cosine per word (words_loss.py:20-27,72)
        float nu = 0.f, s1 = 0.f, s2 = 0.f;
        for (int d = lane; d < D; d += 64) {
            const float ev = sm.e[d][w], cv = sm.c[d][w];
            nu += ev * cv; s1 += ev * ev; s2 += cv * cv;
        }
        nu = wave_sum(nu); s1 = wave_sum(s1); s2 = wave_sum(s2);
        if (lane == 0) {
            const float a = sqrtf(s1), b = sqrtf(s2);
            sm.num[w] = nu; sm.n1[w] = a; sm.n2[w] = b;
            sm.cosv[w] = nu / fmaxf(a * b, eps);
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(kNT) void words_pair_fwd_mfma_kernel(const float* __restrict__ feat, const float* __restrict__ featT,
                                                                  const float* __restrict__ wemb, const int64_t* __restrict__ lens,
                                                                  float gamma1, float gamma2, float gamma3, float* __restrict__ sim,
                                                                  float* __restrict__ maps, int B, int D, int T, int S) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    PairMfmaSmem<false>& sm = *reinterpret_cast<PairMfmaSmem<false>*>(smem_raw);
    const int j = blockIdx.x, i = blockIdx.y;      // image j, caption i
    const int L = min((int)lens[i], T);
    float a1[kMW];
    const float* featP = featT + (size_t)B * D * S;                 // the padded copy follows the transposed one in the workspace
    pair_forward_mfma<false>(sm, featP + (size_t)j * D * kMS, featT + (size_t)j * D * S, wemb + (size_t)i * D * T, D, T, S, L,
                             rsqrtf((float)D), gamma1, 1e-8f, a1);
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int w = 0; w < L; ++w) s += expf(gamma2 * sm.cosv[w]);     // words_loss.py:77-79
        sim[(size_t)j * B + i] = logf(s) * gamma3;                      // :93
    }
    if (i == j && threadIdx.x < S) {                                    // att_maps.append(attn[i]), :63
        for (int w = 0; w < L; ++w) maps[((size_t)i * T + w) * S + threadIdx.x] = sm.a2[threadIdx.x][w];
    }
}

__global__ __launch_bounds__(kNT) __attribute__((amdgpu_waves_per_eu(AGAN_PAIR_MFMA_WAVES))) void words_pair_bwd_mfma_kernel(
    const float* __restrict__ feat, const float* __restrict__ featT, const float* __restrict__ wemb, const int64_t* __restrict__ lens,
    const float* __restrict__ dS, const float* __restrict__ dloss, float gamma1, float gamma2, float gamma3, float* __restrict__ part_f,
    float* __restrict__ part_w, int B, int D, int T, int S) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    PairMfmaSmem<true>& sm = *reinterpret_cast<PairMfmaSmem<true>*>(smem_raw);
    const int j = blockIdx.x, i = blockIdx.y;
    const float g = dS[(size_t)j * B + i] * dloss[0] * gamma3;     // d loss / d log-sum-exp of this pair
    float* pf = part_f + ((size_t)j * B + i) * D * S;
    float* pw = part_w + ((size_t)i * B + j) * D * T;
    if (g == 0.f) {                                                // masked pair (uniform across the workgroup): zero slabs
        for (int e = threadIdx.x; e < D * S; e += kNT) pf[e] = 0.f;
        for (int e = threadIdx.x; e < D * T; e += kNT) pw[e] = 0.f;
        return;
    }
    const int L = min((int)lens[i], T);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, mi = lane & 15, kq = lane >> 4;
    const float scale = rsqrtf((float)D), eps = 1e-8f;
    const float* fjP = featT + (size_t)B * D * S + (size_t)j * D * kMS;      // padded [D][kMS] copy (after the transposed one)
    const float* fjT = featT + (size_t)j * D * S;
    float a1[kMW];
    pair_forward_mfma<true>(sm, fjP, fjT, wemb + (size_t)i * D * T, D, T, S, L, scale, gamma1, eps, a1);

    // --- d cos, then d num / d n1 / d n2 per word (uniform, recomputed by every thread from LDS) ---
    float dnum[kMW], dn1[kMW], dn2[kMW];
    {
        float mx = -INFINITY;
        for (int w = 0; w < L; ++w) mx = fmaxf(mx, gamma2 * sm.cosv[w]);
        float s = 0.f;
        for (int w = 0; w < L; ++w) s += expf(gamma2 * sm.cosv[w] - mx);
#pragma unroll
        for (int w = 0; w < kMW; ++w) {
            dnum[w] = dn1[w] = dn2[w] = 0.f;
            if (w < L) {
                const float dcos = g * gamma2 * expf(gamma2 * sm.cosv[w] - mx) / s;
                const float nn = sm.n1[w] * sm.n2[w];
                if (nn > eps) {
                    dnum[w] = dcos / nn;
                    const float k = -dcos * sm.num[w] / (nn * nn);
                    dn1[w] = k * sm.n2[w];
                    dn2[w] = k * sm.n1[w];
                } else {
                    dnum[w] = dcos / eps;      // clamp(min=eps) active: denominator is a constant
                }
            }
        }
    }
    // --- thread d: de (direct part) in registers, dc overwrites c in LDS ---
    float de[kMW];
#pragma unroll
    for (int w = 0; w < kMW; ++w) de[w] = 0.f;
    if (tid < D) {
#pragma unroll
        for (int w = 0; w < kMW; ++w) {
            const float ev = sm.e[tid][w], cv = sm.c[tid][w];
            de[w] = dnum[w] * cv + (sm.n1[w] > 0.f ? dn1[w] * ev / sm.n1[w] : 0.f);
            sm.c[tid][w] = dnum[w] * ev + (sm.n2[w] > 0.f ? dn2[w] * cv / sm.n2[w] : 0.f);
        }
    }
    __syncthreads();
    // --- da2[r][w] = sum_d f[d][r] dc[d][w]  (into the ds scratch) ---
    mfma_region_rows(fjP, sm.c, sm.ds, D, S);
    __syncthreads();
    const int r = tid;
    const bool rlive = r < S;
    float a2r[kMW], da2[kMW];
#pragma unroll
    for (int w = 0; w < kMW; ++w) {
        a2r[w] = rlive ? sm.a2[r][w] : 0.f;
        da2[w] = rlive ? sm.ds[r][w] : 0.f;
    }
#pragma unroll
    for (int w = 0; w < kMW; ++w) {
        const float v = wave_sum(a2r[w] * da2[w]);
        if (lane == 0) sm.dotw[wave][w] = v;
    }
    __syncthreads();                 // (also: every thread has read its da2 row, the scratch may be overwritten)
    float ds[kMW];
    {
        float inner = 0.f;
#pragma unroll
        for (int w = 0; w < kMW; ++w) {
            float dot = 0.f;
#pragma unroll
            for (int q = 0; q < kNT / 64; ++q) dot += sm.dotw[q][w];          // fixed order over the five waves
            ds[w] = gamma1 * a2r[w] * (da2[w] - (w < L ? dot : 0.f));         // softmax-over-regions backward = d a1[w]
            inner += a1[w] * ds[w];
        }
#pragma unroll
        for (int w = 0; w < kMW; ++w) ds[w] = (w < L && rlive) ? a1[w] * (ds[w] - inner) * scale : 0.f;   // d raw score
    }
    if (r < kMS) {
#pragma unroll
        for (int w = 0; w < kMW; ++w) sm.ds[r][w] = ds[w];      // (zero for the padded rows)
    }
    __syncthreads();
    // --- dfeat[j][d][r] = sum_w dc[d][w] a2[r][w] + e[d][w] ds[r][w]: 16 x 16 tiles, k = the 12 words of each term ---
    {
        const int nrt = (S + 15) >> 4, ndt = D >> 4;
        float ba[4][3], bd[4][3];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = min((wave + 5 * q) * 16 + mi, kMS - 1);           // B[k = w][j = r]
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                ba[q][t] = sm.a2[row][4 * t + kq];
                bd[q][t] = sm.ds[row][4 * t + kq];
            }
        }
        for (int dt = 0; dt < ndt; ++dt) {
            float ac[3], ae[3];
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                ac[t] = sm.c[dt * 16 + mi][4 * t + kq];                       // A[i = d][k = w]
                ae[t] = sm.e[dt * 16 + mi][4 * t + kq];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int rt = wave + 5 * q;
                if (rt < nrt) {
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int t = 0; t < 3; ++t) acc = mfma4(ac[t], ba[q][t], acc);
#pragma unroll
                    for (int t = 0; t < 3; ++t) acc = mfma4(ae[t], bd[q][t], acc);
                    const int rc = rt * 16 + mi;
                    if (rc < S) {
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) pf[(size_t)(dt * 16 + 4 * kq + rr) * S + rc] = acc[rr];
                    }
                }
            }
        }
    }
    __syncthreads();                 // everyone is done reading dc: its storage takes the region part of d e
    mfma_channel_rows(fjT, sm.ds, sm.c, D, S);
    __syncthreads();
    if (tid < D) {
        float* dwi = pw + (size_t)tid * T;
#pragma unroll
        for (int w = 0; w < kMW; ++w)
            if (w < T) dwi[w] = w < L ? de[w] + sm.c[tid][w] : 0.f;
    }
}

// Two cross-entropies over a [B][B] similarity matrix, CE(S, labels) + CE(S^T, labels) (words_loss.py:98-99, sentence_loss.py:46-47),
// optional same-class mask.  labels == nullptr means arange(B) (what train.py:104 passes); a label outside [0, B) poisons the loss
// with NaN (torch raises there).  Writes loss and dS = d loss / d S (before the upstream gradient).  One workgroup.
__global__ __launch_bounds__(256) void contrastive_ce_kernel(float* sim, const int64_t* __restrict__ cids, const int64_t* __restrict__ labels,
                                                             float lambda, float* __restrict__ loss, float* dS, int B) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* S = reinterpret_cast<float*>(smem_raw);      // [B][B]
    float* rlse = S + B * B;                            // [B] row logsumexp
    float* clse = rlse + B;                             // [B] column logsumexp
    __shared__ float red[4];
    for (int e = threadIdx.x; e < B * B; e += 256) {
        const int r = e / B, c = e - r * B;
        float v = sim[e];
        if (cids && r != c && cids[r] == cids[c]) v = -INFINITY;       // words_loss.py:44-47,95
        S[e] = v;
        sim[e] = v;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < 2 * B; k += 256) {
        const bool col = k >= B;
        const int q = col ? k - B : k;
        float mx = -INFINITY;
        for (int t = 0; t < B; ++t) mx = fmaxf(mx, col ? S[t * B + q] : S[q * B + t]);
        float s = 0.f;
        for (int t = 0; t < B; ++t) s += expf((col ? S[t * B + q] : S[q * B + t]) - mx);
        (col ? clse : rlse)[q] = mx + logf(s);
    }
    __syncthreads();
    float part = 0.f;
    for (int q = threadIdx.x; q < B; q += 256) {
        const long long lq = labels ? (long long)labels[q] : q;
        if (lq < 0 || lq >= B) part = NAN;
        else part += (rlse[q] - S[q * B + (int)lq]) + (clse[q] - S[(int)lq * B + q]);     // S^T[q][l] = S[l][q]
    }
    part = block_sum<256>(part, red);
    if (threadIdx.x == 0) loss[0] = part / B * lambda;
    const float sc = lambda / B;
    for (int e = threadIdx.x; e < B * B; e += 256) {
        const int r = e / B, c = e - r * B;
        const float v = S[e];
        float g = 0.f;
        if (v > -INFINITY) g = expf(v - rlse[r]) + expf(v - clse[c]);
        const long long lr = labels ? (long long)labels[r] : r, lc = labels ? (long long)labels[c] : c;
        if (c == lr) g -= 1.f;
        if (r == lc) g -= 1.f;
        dS[e] = g * sc;
    }
}

// (the <12, 256> instance: 43 KB of LDS = three pair workgroups per CU only if the 15 waves also fit the register file, i.e.
// four waves per SIMD -> at most 128 VGPRs; left alone the compiler takes 136 and the last 64 of 576 pairs wait for a second round)
template <int TMAX, int DR>
__global__ __launch_bounds__(kNT) __attribute__((amdgpu_waves_per_eu(TMAX <= 12 ? 4 : 2))) void words_pair_bwd_kernel(const float* __restrict__ feat, const float* __restrict__ featT,
                                                             const float* __restrict__ wemb,
                                                             const int64_t* __restrict__ lens, const float* __restrict__ dS,
                                                             const float* __restrict__ dloss, float gamma1, float gamma2, float gamma3,
                                                             float* __restrict__ part_f, float* __restrict__ part_w, int B, int D, int T, int S) {
    // No atomics anywhere (bit-reproducible gradients): the pair writes its contribution to dfeat[j] into slab part_f[j][i] and
    // its contribution to dwemb[i] into part_w[i][j]; pair_slab_sum_kernel adds the B slabs of each row in index order.
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    PairSmem<TMAX, DR>& sm = *reinterpret_cast<PairSmem<TMAX, DR>*>(smem_raw);
    __shared__ float dotw[kNT / 64][TMAX];
    const int j = blockIdx.x, i = blockIdx.y;
    const float g = dS[(size_t)j * B + i] * dloss[0] * gamma3;     // d loss / d log-sum-exp of this pair
    float* pf = part_f + ((size_t)j * B + i) * D * S;
    float* pw = part_w + ((size_t)i * B + j) * D * T;
    if (g == 0.f) {                                                // masked pair (uniform across the workgroup): zero slabs
        for (int e = threadIdx.x; e < D * S; e += kNT) pf[e] = 0.f;
        for (int e = threadIdx.x; e < D * T; e += kNT) pw[e] = 0.f;
        return;
    }
    const int L = min((int)lens[i], T);
    const int tid = threadIdx.x, lane = tid & 63;
    const float scale = rsqrtf((float)D), eps = 1e-8f;
    const float* fj = feat + (size_t)j * D * S;
    const float* fjT = featT + (size_t)j * D * S;
    float a1[TMAX];
    pair_forward<TMAX, DR>(sm, fj, fjT, wemb + (size_t)i * D * T, D, T, S, L, scale, gamma1, eps, a1);

    // --- d cos, then d num / d n1 / d n2 per word (uniform, recomputed by every thread from LDS) ---
    float dnum[TMAX], dn1[TMAX], dn2[TMAX];
    {
        float mx = -INFINITY;
        for (int w = 0; w < L; ++w) mx = fmaxf(mx, gamma2 * sm.cosv[w]);
        float s = 0.f;
        for (int w = 0; w < L; ++w) s += expf(gamma2 * sm.cosv[w] - mx);
#pragma unroll
        for (int w = 0; w < TMAX; ++w) {
            dnum[w] = dn1[w] = dn2[w] = 0.f;
            if (w < L) {
                const float dcos = g * gamma2 * expf(gamma2 * sm.cosv[w] - mx) / s;
                const float nn = sm.n1[w] * sm.n2[w];
                if (nn > eps) {
                    dnum[w] = dcos / nn;
                    const float k = -dcos * sm.num[w] / (nn * nn);
                    dn1[w] = k * sm.n2[w];
                    dn2[w] = k * sm.n1[w];
                } else {
                    dnum[w] = dcos / eps;      // clamp(min=eps) active: denominator is a constant
                }
            }
        }
    }
    // --- thread d: de (direct part) in registers, dc overwrites c in LDS ---
    float de[TMAX];
#pragma unroll
    for (int w = 0; w < TMAX; ++w) de[w] = 0.f;
    if (tid < D) {
#pragma unroll
        for (int w = 0; w < TMAX; ++w) {
            const float ev = sm.e[tid][w], cv = sm.c[tid][w];
            de[w] = dnum[w] * cv + (sm.n1[w] > 0.f ? dn1[w] * ev / sm.n1[w] : 0.f);
            sm.c[tid][w] = dnum[w] * ev + (sm.n2[w] > 0.f ? dn2[w] * cv / sm.n2[w] : 0.f);
        }
    }
    __syncthreads();
    // --- thread r: da2[w] = sum_d f[d][r] dc[d][w];  dot[w] = sum_r a2 da2 ---
    const int r = tid;
    const bool rlive = r < S;
    float a2r[TMAX], da2[TMAX];
#pragma unroll
    for (int w = 0; w < TMAX; ++w) { da2[w] = 0.f; a2r[w] = rlive ? sm.a2[r][w] : 0.f; }
    if (rlive) {
AGAN_PAIR_UNROLL
        for (int d = 0; d < D; ++d) {
            const float fv = fj[(size_t)d * S + r];
#pragma unroll
            for (int w = 0; w < TMAX; ++w) da2[w] += fv * sm.c[d][w];
        }
    }
#pragma unroll
    for (int w = 0; w < TMAX; ++w) {
        const float v = wave_sum(a2r[w] * da2[w]);
        if (lane == 0) dotw[tid >> 6][w] = v;
    }
    __syncthreads();
    float dot[TMAX];
#pragma unroll
    for (int w = 0; w < TMAX; ++w) {
        float v = 0.f;
#pragma unroll
        for (int q = 0; q < kNT / 64; ++q) v += dotw[q][w];          // fixed order over the five waves
        dot[w] = v;
    }
    // softmax-over-regions backward, then softmax-over-words backward -> ds (gradient of the scaled score)
    float ds[TMAX];
    {
        float inner = 0.f;
#pragma unroll
        for (int w = 0; w < TMAX; ++w) {
            ds[w] = gamma1 * a2r[w] * (da2[w] - (w < L ? dot[w] : 0.f));   // = d a1[w]
            inner += a1[w] * ds[w];
        }
#pragma unroll
        for (int w = 0; w < TMAX; ++w) ds[w] = (w < L && rlive) ? a1[w] * (ds[w] - inner) * scale : 0.f;   // d raw score
    }
    // --- dfeat[j][d][r] += sum_w dc[d][w] a2[r][w] + ds[r][w] e[d][w]   (coalesced along r) ---
    if (rlive) {
        float* dfj = pf + r;
        for (int d = 0; d < D; ++d) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < TMAX; ++w) v += sm.c[d][w] * a2r[w] + ds[w] * sm.e[d][w];
            dfj[(size_t)d * S] = v;
        }
    }
    __syncthreads();                 // everyone is done reading a2 (registers now) -> reuse its storage for ds
    if (rlive) {
#pragma unroll
        for (int w = 0; w < TMAX; ++w) sm.a2[r][w] = ds[w];
    }
    __syncthreads();
    // --- thread d: de[d][w] += sum_r ds[r][w] f[d][r];  dwemb[i][d][w] += de ---
    if (tid < D) {
#ifdef AGAN_PAIR_NO_T
        const float* fr = fj + (size_t)tid * S;
        const int fs = 1;
#else
        const float* fr = fjT + tid;
        const int fs = D;
#endif
AGAN_PAIR_UNROLL
        for (int q = 0; q < S; ++q) {
            const float fv = fr[(size_t)q * fs];
#pragma unroll
            for (int w = 0; w < TMAX; ++w) de[w] += fv * sm.a2[q][w];
        }
        float* dwi = pw + (size_t)tid * T;
#pragma unroll
        for (int w = 0; w < TMAX; ++w)
            if (w < T) dwi[w] = w < L ? de[w] : 0.f;
    }
}

// featT[b][s][d] = feat[b][d][s]: 32x32 tiles through LDS, both sides coalesced (7 MB at the metric shapes)
// outP (optional): the same features as [B][D][SP] with zero-padded, 16-byte aligned rows (the MFMA pair kernels' region operand)
__global__ __launch_bounds__(256) void feat_transpose_kernel(const float* __restrict__ in, float* __restrict__ out, float* __restrict__ outP,
                                                             int D, int S, int SP) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, d0 = blockIdx.y * 32, s0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float* src = in + (size_t)b * D * S;
    float* dst = out + (size_t)b * D * S;
    for (int k = ty; k < 32; k += 8)
        tile[k][tx] = (d0 + k < D && s0 + tx < S) ? src[(size_t)(d0 + k) * S + s0 + tx] : 0.f;
    __syncthreads();
    for (int k = ty; k < 32; k += 8)
        if (s0 + k < S && d0 + tx < D) dst[(size_t)(s0 + k) * D + d0 + tx] = tile[tx][k];
    if (outP) {
        float* dp = outP + (size_t)b * D * SP;
        for (int k = ty; k < 32; k += 8)
            if (d0 + k < D && s0 + tx < SP) dp[(size_t)(d0 + k) * SP + s0 + tx] = tile[k][tx];
    }
}

// out[row][e] = sum_k part[row][k][e], k = 0..B-1 in order (row = image for dfeat, caption for dwemb)
__global__ __launch_bounds__(256) void pair_slab_sum_kernel(const float* __restrict__ part, float* __restrict__ out, int B, int n) {
    const int row = blockIdx.y;
    const float* p = part + (size_t)row * B * n;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < n; e += gridDim.x * 256) {
        float s = 0.f;
        for (int k = 0; k < B; ++k) s += p[(size_t)k * n + e];
        out[(size_t)row * n + e] = s;
    }
}

// ---------------------------------------------------------------------------------------------------------
// sentence loss (one workgroup)
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sent_scores_kernel(const float* __restrict__ cnn, const float* __restrict__ rnn, float gamma3,
                                                          float eps, float* __restrict__ sim, float* __restrict__ dots,
                                                          float* __restrict__ norms, int B, int D) {
    // norms[0..B) = |cnn_j|, norms[B..2B) = |rnn_i|; dots[j][i] = <cnn_j, rnn_i>; sim = gamma3 * dots / max(nc*nr, eps)
    // one workgroup per image j, one wave per caption i (lanes stride the embedding); every workgroup needs all |rnn_i| and
    // recomputes them on the way (B*D reads, L2-resident)
    const int j = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* cj = cnn + (size_t)j * D;
    float s = 0.f;
    for (int d = lane; d < D; d += 64) s += cj[d] * cj[d];
    const float ncj = sqrtf(wave_sum(s));
    if (threadIdx.x == 0) norms[j] = ncj;
    for (int i = wave; i < B; i += 4) {
        const float* ri = rnn + (size_t)i * D;
        float dt = 0.f, rr = 0.f;
        for (int d = lane; d < D; d += 64) {
            const float r = ri[d];
            dt += cj[d] * r;
            rr += r * r;
        }
        dt = wave_sum(dt);
        const float nri = sqrtf(wave_sum(rr));
        if (lane == 0) {
            dots[(size_t)j * B + i] = dt;
            sim[(size_t)j * B + i] = dt / fmaxf(ncj * nri, eps) * gamma3;      // sentence_loss.py:33-38
            if (j == 0) norms[B + i] = nri;
        }
    }
}

__global__ __launch_bounds__(256) void sent_bwd_kernel(const float* __restrict__ cnn, const float* __restrict__ rnn, const float* __restrict__ dS,
                                                       const float* __restrict__ dots, const float* __restrict__ norms,
                                                       const float* __restrict__ dloss, float gamma3, float eps, float* __restrict__ dcnn,
                                                       float* __restrict__ drnn, int B, int D) {
    const float up = dloss[0] * gamma3;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < 2 * B * D; e += gridDim.x * 256) {
        const bool isr = e >= B * D;
        const int q = isr ? e - B * D : e;
        const int a = q / D, d = q - a * D;            // a = own row (image j for cnn, caption i for rnn)
        float acc = 0.f;
        for (int o = 0; o < B; ++o) {
            const int j = isr ? o : a, i = isr ? a : o;
            const float gji = dS[j * B + i] * up;
            const float nc = norms[j], nr = norms[B + i], nn = nc * nr;
            const float other = isr ? cnn[(size_t)j * D + d] : rnn[(size_t)i * D + d];
            const float self = isr ? rnn[(size_t)i * D + d] : cnn[(size_t)j * D + d];
            if (nn > eps) {
                const float ns = isr ? nr : nc, no = isr ? nc : nr;
                acc += gji * (other / nn - dots[j * B + i] / (nn * nn) * no * (ns > 0.f ? self / ns : 0.f));
            } else {
                acc += gji * other / eps;
            }
        }
        (isr ? drnn : dcnn)[q] = acc;
    }
}

// standalone func_attention forward (networks/attention.py:82-120): query [B,D,L], context [B,D,S] -> wctx [B,D,L], attn [B,L,S]
template <int TMAX>
__global__ __launch_bounds__(kNT) void func_attn_fwd_kernel(const float* __restrict__ query, const float* __restrict__ context, float gamma1,
                                                            float scale, float* __restrict__ wctx, float* __restrict__ attn, int D, int L, int S) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    PairSmem<TMAX>& sm = *reinterpret_cast<PairSmem<TMAX>*>(smem_raw);
    const int b = blockIdx.x;
    float a1[TMAX];
    pair_forward<TMAX, kNT>(sm, context + (size_t)b * D * S, nullptr, query + (size_t)b * D * L, D, L, S, L, scale, gamma1, 1e-8f, a1);
    if (threadIdx.x < D)
        for (int w = 0; w < L; ++w) wctx[((size_t)b * D + threadIdx.x) * L + w] = sm.c[threadIdx.x][w];
    if (threadIdx.x < S)
        for (int w = 0; w < L; ++w) attn[((size_t)b * L + w) * S + threadIdx.x] = sm.a2[threadIdx.x][w];
}


// standalone func_attention backward: (d wctx [B,D,L], d attn [B,L,S], either may be null) -> dquery [B,D,L], dcontext [B,D,S].
// Same chain as the pair backward without the cosine head: recompute the forward of batch element b, then
//   da2[r][w] = sum_d f[d][r] dc[d][w] + dattn[w][r];  region-softmax and word-softmax backward -> ds;
//   dcontext[d][r] = sum_w dc[d][w] a2[r][w] + ds[r][w] q[d][w];   dquery[d][w] = sum_r ds[r][w] f[d][r].
template <int TMAX>
__global__ __launch_bounds__(kNT) void func_attn_bwd_kernel(const float* __restrict__ query, const float* __restrict__ context,
                                                            const float* __restrict__ dwctx, const float* __restrict__ dattn, float gamma1,
                                                            float scale, float* __restrict__ dquery, float* __restrict__ dcontext, int D, int L,
                                                            int S) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    PairSmem<TMAX>& sm = *reinterpret_cast<PairSmem<TMAX>*>(smem_raw);
    __shared__ float dotw[kNT / 64][TMAX];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const float* fb = context + (size_t)b * D * S;
    float a1[TMAX];
    pair_forward<TMAX, kNT>(sm, fb, nullptr, query + (size_t)b * D * L, D, L, S, L, scale, gamma1, 1e-8f, a1);
    if (tid < D) {
#pragma unroll
        for (int w = 0; w < TMAX; ++w) sm.c[tid][w] = (dwctx && w < L) ? dwctx[((size_t)b * D + tid) * L + w] : 0.f;
    }
    __syncthreads();
    const int r = tid;
    const bool rlive = r < S;
    float a2r[TMAX], da2[TMAX];
#pragma unroll
    for (int w = 0; w < TMAX; ++w) {
        a2r[w] = rlive ? sm.a2[r][w] : 0.f;
        da2[w] = (rlive && dattn && w < L) ? dattn[((size_t)b * L + w) * S + r] : 0.f;
    }
    if (rlive) {
        for (int d = 0; d < D; ++d) {
            const float fv = fb[(size_t)d * S + r];
#pragma unroll
            for (int w = 0; w < TMAX; ++w) da2[w] += fv * sm.c[d][w];
        }
    }
#pragma unroll
    for (int w = 0; w < TMAX; ++w) {
        const float v = wave_sum(a2r[w] * da2[w]);
        if (lane == 0) dotw[tid >> 6][w] = v;
    }
    __syncthreads();
    float ds[TMAX];
    {
        float inner = 0.f;
#pragma unroll
        for (int w = 0; w < TMAX; ++w) {
            float dot = 0.f;
#pragma unroll
            for (int q = 0; q < kNT / 64; ++q) dot += dotw[q][w];
            ds[w] = gamma1 * a2r[w] * (da2[w] - (w < L ? dot : 0.f));
            inner += a1[w] * ds[w];
        }
#pragma unroll
        for (int w = 0; w < TMAX; ++w) ds[w] = (w < L && rlive) ? a1[w] * (ds[w] - inner) * scale : 0.f;
    }
    if (rlive) {
        float* dfb = dcontext + (size_t)b * D * S + r;
        for (int d = 0; d < D; ++d) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < TMAX; ++w) v += sm.c[d][w] * a2r[w] + ds[w] * sm.e[d][w];
            dfb[(size_t)d * S] = v;
        }
    }
    __syncthreads();
    if (rlive) {
#pragma unroll
        for (int w = 0; w < TMAX; ++w) sm.a2[r][w] = ds[w];
    }
    __syncthreads();
    if (tid < D) {
        float de[TMAX];
#pragma unroll
        for (int w = 0; w < TMAX; ++w) de[w] = 0.f;
        const float* fr = fb + (size_t)tid * S;
        for (int q = 0; q < S; ++q) {
            const float fv = fr[q];
#pragma unroll
            for (int w = 0; w < TMAX; ++w) de[w] += fv * sm.a2[q][w];
        }
#pragma unroll
        for (int w = 0; w < TMAX; ++w)
            if (w < L) dquery[((size_t)b * D + tid) * L + w] = de[w];
    }
}

template <int TMAX, int DR = kNT>
constexpr size_t pair_smem_bytes() { return sizeof(PairSmem<TMAX, DR>); }

// shapes the MFMA pair kernels take (everything else runs the thread-per-row kernels); AGAN_PAIR_VALU=1 in the environment forces
// the latter (A/B measurements)
inline bool pair_mfma_ok(int D, int T, int S) {
    static const bool off = [] { const char* v = getenv("AGAN_PAIR_VALU"); return v && v[0] == '1'; }();
    return !off && T <= kMW && D <= kMD && (D & 63) == 0 && S <= kMS - 3 && S >= 16;
}

// the [B][B] + 2[B] matrix of the contrastive kernel exceeds the default 64 KB dynamic-LDS limit at B >= 127 (kMaxB = 128)
inline void allow_ce_smem() {      // (once per process: one process drives one GPU)
    static const hipError_t attr_ = hipFuncSetAttribute(reinterpret_cast<const void*>(contrastive_ce_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                        (int)((kMaxB * kMaxB + 2 * kMaxB) * sizeof(float)));
    (void)attr_;
}

}  // namespace

extern "C" {

size_t agan_words_loss_save_elems(int B, int D, int T, int S) {      // dS, transposed features, zero-padded features
    return (size_t)B * B + (size_t)B * D * S + (pair_mfma_ok(D, T, S) ? (size_t)B * D * kMS : 0);
}

int agan_words_loss_fwd(const float* feat, const float* wemb, const int64_t* lens, const int64_t* class_ids, const int64_t* labels, float gamma1,
                        float gamma2, float gamma3, float lambda, float* loss, float* sim, float* attn_maps, float* save, int B,
                        int D, int T, int S, void* stream) {
    AGAN_REQUIRE(feat && wemb && lens && loss && sim && attn_maps && save, "words_loss_fwd: null pointer");
    AGAN_REQUIRE(B >= 1 && B <= kMaxB, "words_loss: batch %d out of range [1,%d]", B, kMaxB);
    AGAN_REQUIRE(D >= 1 && D <= kNT && S >= 1 && S <= kNT, "words_loss: nef %d / regions %d exceed %d", D, S, kNT);
    AGAN_REQUIRE(T >= 1 && T <= 32, "words_loss: seq_len %d > 32", T);
    hipStream_t st = as_stream(stream);
    dim3 grid(B, B);
    float* featT = save + (size_t)B * B;              // [B][S][D], kept for the backward
    float* featP = pair_mfma_ok(D, T, S) ? featT + (size_t)B * D * S : nullptr;
    hipLaunchKernelGGL(feat_transpose_kernel, dim3(cdiv(featP ? kMS : S, 32), cdiv(D, 32), B), dim3(256), 0, st, feat, featT, featP, D, S, kMS);
#define AGAN_PAIR_FWD(TM, DRR)                                                                                                          \
    do {                                                                                                                               \
        static const hipError_t attr_ = hipFuncSetAttribute(reinterpret_cast<const void*>(words_pair_fwd_kernel<TM, DRR>),             \
                                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)pair_smem_bytes<TM, DRR>()); \
        (void)attr_;                                                                                                                   \
        const size_t smem_ = pair_smem_bytes<TM, DRR>();                                                                               \
        hipLaunchKernelGGL((words_pair_fwd_kernel<TM, DRR>), grid, dim3(kNT), smem_, st, feat, featT, wemb, lens, gamma1,  \
                           gamma2, gamma3, sim, attn_maps, B, D, T, S);                                                                \
    } while (0)
    if (pair_mfma_ok(D, T, S)) {
        static const hipError_t attr_ = hipFuncSetAttribute(reinterpret_cast<const void*>(words_pair_fwd_mfma_kernel),
                                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(PairMfmaSmem<false>));
        (void)attr_;
        hipLaunchKernelGGL(words_pair_fwd_mfma_kernel, grid, dim3(kNT), sizeof(PairMfmaSmem<false>), st, feat, featT, wemb, lens, gamma1, gamma2,
                           gamma3, sim, attn_maps, B, D, T, S);
    } else if (T <= 12 && D <= 256) AGAN_PAIR_FWD(12, 256);
    else if (T <= 16) AGAN_PAIR_FWD(16, kNT);
    else AGAN_PAIR_FWD(32, kNT);
#undef AGAN_PAIR_FWD
    allow_ce_smem();
    hipLaunchKernelGGL(contrastive_ce_kernel, dim3(1), dim3(256), (size_t)(B * B + 2 * B) * sizeof(float), st, sim, class_ids, labels, lambda, loss, save, B);
    return check_launch("words_loss_fwd");
}

size_t agan_words_loss_bwd_ws_bytes(int B, int D, int T, int S) {
    if (B <= 0 || D <= 0 || T <= 0 || S <= 0) return 0;
    return (size_t)B * B * D * ((size_t)S + T) * sizeof(float);        // one dfeat slab and one dwemb slab per (image, caption) pair
}

int agan_words_loss_bwd(const float* feat, const float* wemb, const int64_t* lens, const float* save, const float* dloss, float gamma1,
                        float gamma2, float gamma3, float lambda, float* dfeat, float* dwemb, int B, int D, int T, int S, void* ws,
                        size_t ws_bytes, void* stream) {
    AGAN_REQUIRE(feat && wemb && lens && save && dloss && dfeat && dwemb && ws, "words_loss_bwd: null pointer");
    AGAN_REQUIRE(B >= 1 && B <= kMaxB && D <= kNT && S <= kNT && T <= 32, "words_loss_bwd: shape out of range");
    if (ws_bytes < agan_words_loss_bwd_ws_bytes(B, D, T, S)) {
        set_error("words_loss_bwd: workspace %zu < %zu", ws_bytes, agan_words_loss_bwd_ws_bytes(B, D, T, S));
        return AGAN_EWORKSPACE;
    }
    hipStream_t st = as_stream(stream);
    float* part_f = static_cast<float*>(ws);
    float* part_w = part_f + (size_t)B * B * D * S;
    dim3 grid(B, B);
#define AGAN_PAIR_BWD(TM, DRR)                                                                                                          \
    do {                                                                                                                               \
        static const hipError_t attr_ = hipFuncSetAttribute(reinterpret_cast<const void*>(words_pair_bwd_kernel<TM, DRR>),             \
                                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)pair_smem_bytes<TM, DRR>()); \
        (void)attr_;                                                                                                                   \
        const size_t smem_ = pair_smem_bytes<TM, DRR>();                                                                               \
        hipLaunchKernelGGL((words_pair_bwd_kernel<TM, DRR>), grid, dim3(kNT), smem_, st, feat, save + (size_t)B * B, wemb, lens, save,   \
                           dloss, gamma1, gamma2, gamma3, part_f, part_w, B, D, T, S);                                                 \
    } while (0)
    if (pair_mfma_ok(D, T, S)) {
        static const hipError_t attr_ = hipFuncSetAttribute(reinterpret_cast<const void*>(words_pair_bwd_mfma_kernel),
                                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(PairMfmaSmem<true>));
        (void)attr_;
        hipLaunchKernelGGL(words_pair_bwd_mfma_kernel, grid, dim3(kNT), sizeof(PairMfmaSmem<true>), st, feat, save + (size_t)B * B, wemb, lens,
                           save, dloss, gamma1, gamma2, gamma3, part_f, part_w, B, D, T, S);
    } else if (T <= 12 && D <= 256) AGAN_PAIR_BWD(12, 256);
    else if (T <= 16) AGAN_PAIR_BWD(16, kNT);
    else AGAN_PAIR_BWD(32, kNT);
#undef AGAN_PAIR_BWD
    hipLaunchKernelGGL(pair_slab_sum_kernel, dim3(std::min(cdiv(D * S, 256), 64), B), dim3(256), 0, st, part_f, dfeat, B, D * S);
    hipLaunchKernelGGL(pair_slab_sum_kernel, dim3(std::min(cdiv(D * T, 256), 64), B), dim3(256), 0, st, part_w, dwemb, B, D * T);
    return check_launch("words_loss_bwd");
}

int agan_sent_loss_fwd(const float* cnn_code, const float* rnn_code, const int64_t* class_ids, const int64_t* labels, float gamma3, float lambda, float eps,
                       float* loss, float* save, int B, int D, void* stream) {
    AGAN_REQUIRE(cnn_code && rnn_code && loss && save, "sent_loss_fwd: null pointer");
    AGAN_REQUIRE(B >= 1 && B <= kMaxB && D >= 1, "sent_loss: batch %d out of range", B);
    hipStream_t st = as_stream(stream);
    float* dS = save;                 // [B*B]  d loss / d sim
    float* dots = save + B * B;       // [B*B]
    float* norms = dots + B * B;      // [2B]
    // sim is staged in dS's storage, then overwritten by the CE kernel's gradient
    hipLaunchKernelGGL(sent_scores_kernel, dim3(B), dim3(256), 0, st, cnn_code, rnn_code, gamma3, eps, dS, dots, norms, B, D);
    allow_ce_smem();
    hipLaunchKernelGGL(contrastive_ce_kernel, dim3(1), dim3(256), (size_t)(B * B + 2 * B) * sizeof(float), st, dS, class_ids, labels, lambda, loss, dS, B);
    return check_launch("sent_loss_fwd");
}

int agan_sent_loss_bwd(const float* cnn_code, const float* rnn_code, const float* save, const float* dloss, float gamma3, float lambda,
                       float eps, float* dcnn, float* drnn, int B, int D, void* stream) {
    AGAN_REQUIRE(cnn_code && rnn_code && save && dloss && dcnn && drnn, "sent_loss_bwd: null pointer");
    const float* dS = save;
    const float* dots = save + B * B;
    const float* norms = dots + B * B;
    hipLaunchKernelGGL(sent_bwd_kernel, dim3(cdiv(2 * B * D, 256)), dim3(256), 0, as_stream(stream), cnn_code, rnn_code, dS, dots, norms,
                       dloss, gamma3, eps, dcnn, drnn, B, D);
    return check_launch("sent_loss_bwd");
}

int agan_func_attention_fwd(const float* query, const float* context, float gamma1, float scale, float* wctx, float* attn, int B,
                            int D, int L, int S, void* stream) {
    AGAN_REQUIRE(query && context && wctx && attn, "func_attention_fwd: null pointer");
    AGAN_REQUIRE(B >= 1 && D >= 1 && D <= kNT && S >= 1 && S <= kNT && L >= 1 && L <= 32, "func_attention: shape out of range");
    hipStream_t st = as_stream(stream);
    if (L <= 16) {
        static const hipError_t attr_16_ = hipFuncSetAttribute(reinterpret_cast<const void*>(func_attn_fwd_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pair_smem_bytes<16>()); (void)attr_16_;
        hipLaunchKernelGGL((func_attn_fwd_kernel<16>), dim3(B), dim3(kNT), pair_smem_bytes<16>(), st, query, context, gamma1, scale, wctx, attn, D, L, S);
    } else {
        static const hipError_t attr_32_ = hipFuncSetAttribute(reinterpret_cast<const void*>(func_attn_fwd_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pair_smem_bytes<32>()); (void)attr_32_;
        hipLaunchKernelGGL((func_attn_fwd_kernel<32>), dim3(B), dim3(kNT), pair_smem_bytes<32>(), st, query, context, gamma1, scale, wctx, attn, D, L, S);
    }
    return check_launch("func_attention_fwd");
}

int agan_func_attention_bwd(const float* query, const float* context, const float* dwctx, const float* dattn, float gamma1, float scale,
                            float* dquery, float* dcontext, int B, int D, int L, int S, void* stream) {
    AGAN_REQUIRE(query && context && dquery && dcontext, "func_attention_bwd: null pointer");
    AGAN_REQUIRE(B >= 1 && D >= 1 && D <= kNT && S >= 1 && S <= kNT && L >= 1 && L <= 32, "func_attention_bwd: shape out of range");
    hipStream_t st = as_stream(stream);
    if (L <= 16) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(func_attn_bwd_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pair_smem_bytes<16>());
        hipLaunchKernelGGL((func_attn_bwd_kernel<16>), dim3(B), dim3(kNT), pair_smem_bytes<16>(), st, query, context, dwctx, dattn, gamma1, scale, dquery, dcontext, D, L, S);
    } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(func_attn_bwd_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pair_smem_bytes<32>());
        hipLaunchKernelGGL((func_attn_bwd_kernel<32>), dim3(B), dim3(kNT), pair_smem_bytes<32>(), st, query, context, dwctx, dattn, gamma1, scale, dquery, dcontext, D, L, S);
    }
    return check_launch("func_attention_bwd");
}

}  // extern "C"
