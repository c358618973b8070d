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
