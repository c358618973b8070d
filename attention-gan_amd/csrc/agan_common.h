// Internal helpers shared by the gfx950 kernels of libagan_hip.so.  CDNA4 only: wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/agan.h"

namespace agan {

constexpr int kWave = 64;

void set_error(const char* fmt, ...);

#define AGAN_REQUIRE(cond, ...)            \
    do {                                    \
        if (!(cond)) {                      \
            agan::set_error(__VA_ARGS__);   \
            return AGAN_EINVAL;             \
        }                                   \
    } while (0)

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return AGAN_ELAUNCH;
    }
    return AGAN_OK;
}

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

__host__ __device__ inline int cdiv(int a, int b) { return (a + b - 1) / b; }
__host__ __device__ inline size_t cdivz(size_t a, size_t b) { return (a + b - 1) / b; }

// ---- wave / block reductions (64-lane wave) --------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// Wave-wide sums of N per-lane values r[0..N) (N a power of two <= 64) by a transposing butterfly: each exchange halves the
// values a lane still carries, so the whole reduction costs N-1 + log2(64/N) shuffles instead of 6*N.  Returns the full sum of
// element `idx` (set per lane; the 64/N lanes that share an idx all hold the same sum).  r[] is clobbered.
template <int N>
__device__ __forceinline__ float wave_sum_scatter(float (&r)[N], int lane, int& idx) {
    static_assert(N >= 1 && N <= 64 && (N & (N - 1)) == 0, "N must be a power of two <= 64");
    idx = 0;
    int width = 32;
#pragma unroll
    for (int n = N / 2; n >= 1; n >>= 1, width >>= 1) {
        const bool up = (lane & width) != 0;
#pragma unroll
        for (int j = 0; j < n; ++j) {
            const float send = up ? r[j] : r[j + n];
            const float keep = up ? r[j + n] : r[j];
            r[j] = keep + __shfl_xor(send, width, 64);
        }
        if (up) idx += n;
    }
    float v = r[0];
#pragma unroll
    for (int w = 64 / N / 2; w >= 1; w >>= 1) v += __shfl_xor(v, w, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---- running |max| of a tensor for the fp16 split mode (include/agan.h: amax slots) ---------------------------------
// A slot holds kAmaxLanes running maxima (zeroed by the caller before the producer runs); producers fold their maximum into one
// of them (atomicMax on the bit pattern: non-negative floats order like unsigned integers), consumers take the maximum of all.
constexpr int kAmaxLanes = 8;          // independent maxima per slot ...
constexpr int kAmaxStride = 32;        // ... each on its own 128-byte line (floats): a slot is kAmaxLanes * kAmaxStride floats
// One commit per WORKGROUP (256 threads): memory-side atomics on one line serialise at ~13 ns each (MI355X_MICROARCH.md, Global
// float atomics), and even plain loads of one line queue at its L2 channel, so thousands of waves must not all touch the slot.
// The workgroup's maximum is first compared with the running maximum through a relaxed agent-scope load (served by L2): only
// a workgroup that RAISES it pays for the atomic -- after the first few hundred almost none does.
__device__ __forceinline__ void amax_commit_lds(float m, float* slot, float* amax_red) {   // amax_red: 4 floats of LDS
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) amax_red[(threadIdx.x >> 6) & 3] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nw = (blockDim.x + 63) >> 6;
        for (int i = 1; i < nw && i < 4; ++i) m = fmaxf(m, amax_red[i]);
        unsigned* p = reinterpret_cast<unsigned*>(slot) + (blockIdx.x & (kAmaxLanes - 1)) * kAmaxStride;
        const unsigned bits = __float_as_uint(m);
        if (bits > __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(p, bits);
    }
}
__device__ __forceinline__ void amax_commit(float m, float* slot) {
    __shared__ float amax_red[4];
    amax_commit_lds(m, slot, amax_red);
}
__device__ __forceinline__ float amax_read(const float* slot) {
    float m = 0.f;
#pragma unroll
    for (int i = 0; i < kAmaxLanes; ++i) m = fmaxf(m, slot[i * kAmaxStride]);
    return m;
}
// power-of-two scale that lands amax in (2^12, 2^13]  (1 for an all-zero or non-finite tensor); *inv = 1 / scale
__device__ __forceinline__ float amax_scale(float amax, float* inv) {
    float s = 1.f, r = 1.f;
    if (amax > 0.f && amax < INFINITY) {
        int e;
        frexpf(amax, &e);                                          // amax = f * 2^e, f in [0.5, 1)
        e = max(-100, min(100, 13 - e));
        s = ldexpf(1.f, e);
        r = ldexpf(1.f, -e);
    }
    *inv = r;
    return s;
}

// ---- typed access to activation tensors (include/agan.h: AGAN_DT_*): fp32, or 16-bit storage rounded to nearest even ------------
typedef __bf16 agan_bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 agan_f16x4 __attribute__((ext_vector_type(4)));
typedef float agan_f32x4 __attribute__((ext_vector_type(4)));
template <int DT> __device__ __forceinline__ float4 ld4(const void* p, size_t i) {          // elements i .. i+3 (i a multiple of 4)
    if (DT == AGAN_DT_F32) return *reinterpret_cast<const float4*>(static_cast<const float*>(p) + i);
    const uint2 raw = *reinterpret_cast<const uint2*>(static_cast<const unsigned short*>(p) + i);
    if (DT == AGAN_DT_BF16)
        return make_float4(__uint_as_float(raw.x << 16), __uint_as_float(raw.x & 0xFFFF0000u), __uint_as_float(raw.y << 16), __uint_as_float(raw.y & 0xFFFF0000u));
    const agan_f32x4 v = __builtin_convertvector(__builtin_bit_cast(agan_f16x4, raw), agan_f32x4);
    return make_float4(v[0], v[1], v[2], v[3]);
}
template <int DT> __device__ __forceinline__ void st4(void* p, size_t i, float4 v) {
    if (DT == AGAN_DT_F32) { *reinterpret_cast<float4*>(static_cast<float*>(p) + i) = v; return; }
    const agan_f32x4 f = {v.x, v.y, v.z, v.w};
    uint2 raw;
    if (DT == AGAN_DT_BF16) raw = __builtin_bit_cast(uint2, __builtin_convertvector(f, agan_bf16x4));
    else raw = __builtin_bit_cast(uint2, __builtin_convertvector(f, agan_f16x4));
    *reinterpret_cast<uint2*>(static_cast<unsigned short*>(p) + i) = raw;
}
// V consecutive elements (V = 1, 4 or 8; i a multiple of V): ONE 16-byte access for 8 16-bit values (round 4: the 16-bit storage kernels
// moved 8 bytes per lane and ran at 3.0-4.3 TB/s where their fp32 forms, at 16 bytes per lane, reach 4.6-5.8), two for 8 floats
typedef __bf16 agan_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 agan_f16x8 __attribute__((ext_vector_type(8)));
typedef float agan_f32x8 __attribute__((ext_vector_type(8)));
template <int DT, int V> __device__ __forceinline__ void ldv(const void* p, size_t i, float* v);
template <int DT, int V> __device__ __forceinline__ void stv(void* p, size_t i, const float* v);
template <int DT> __device__ __forceinline__ float ld1(const void* p, size_t i);
template <int DT> __device__ __forceinline__ void st1(void* p, size_t i, float v);
template <int DT, int V> __device__ __forceinline__ void ldv(const void* p, size_t i, float* v) {
    if (V == 1) { v[0] = ld1<DT>(p, i); return; }
    if (V == 4 || DT == AGAN_DT_F32) {
#pragma unroll
        for (int k = 0; k < V; k += 4) *reinterpret_cast<float4*>(v + k) = ld4<DT>(p, i + k);
        return;
    }
    const uint4 raw = *reinterpret_cast<const uint4*>(static_cast<const unsigned short*>(p) + i);
    if (DT == AGAN_DT_BF16) {
        const unsigned w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) { v[2 * k] = __uint_as_float(w[k] << 16); v[2 * k + 1] = __uint_as_float(w[k] & 0xFFFF0000u); }
    } else {
        const agan_f32x8 f = __builtin_convertvector(__builtin_bit_cast(agan_f16x8, raw), agan_f32x8);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = f[k];
    }
}
template <int DT, int V> __device__ __forceinline__ void stv(void* p, size_t i, const float* v) {
    if (V == 1) { st1<DT>(p, i, v[0]); return; }
    if (V == 4 || DT == AGAN_DT_F32) {
#pragma unroll
        for (int k = 0; k < V; k += 4) st4<DT>(p, i + k, make_float4(v[k], v[k + 1], v[k + 2], v[k + 3]));
        return;
    }
    const agan_f32x8 f = {v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]};
    uint4 raw;
    if (DT == AGAN_DT_BF16) raw = __builtin_bit_cast(uint4, __builtin_convertvector(f, agan_bf16x8));
    else raw = __builtin_bit_cast(uint4, __builtin_convertvector(f, agan_f16x8));
    *reinterpret_cast<uint4*>(static_cast<unsigned short*>(p) + i) = raw;
}
template <int DT> __device__ __forceinline__ float ld1(const void* p, size_t i) {
    if (DT == AGAN_DT_F32) return static_cast<const float*>(p)[i];
    const unsigned short h = static_cast<const unsigned short*>(p)[i];
    if (DT == AGAN_DT_BF16) return __uint_as_float((unsigned)h << 16);
    return (float)__builtin_bit_cast(_Float16, h);
}
template <int DT> __device__ __forceinline__ void st1(void* p, size_t i, float v) {
    if (DT == AGAN_DT_F32) { static_cast<float*>(p)[i] = v; return; }
    if (DT == AGAN_DT_BF16) static_cast<__bf16*>(p)[i] = (__bf16)v;
    else static_cast<_Float16*>(p)[i] = (_Float16)v;
}
__host__ __device__ inline int dt_size(int dt) { return dt == AGAN_DT_F32 ? 4 : 2; }
// rounds v to the storage type and back (what a consumer of the stored tensor will read)
template <int DT> __device__ __forceinline__ float round_dt(float v) {
    if (DT == AGAN_DT_F32) return v;
    if (DT == AGAN_DT_BF16) return (float)(__bf16)v;
    return (float)(_Float16)v;
}

// sum over a block of NT threads; result valid in every thread.  smem: NT/64 floats.
template <int NT>
__device__ __forceinline__ float block_sum(float v, float* smem) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) smem[w] = v;
    __syncthreads();
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) r += smem[i];
    return r;
}

}  // namespace agan
