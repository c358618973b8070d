// Does v_pk_fma_f32 return wrong results when OTHER waves on the same compute unit run 16-bit MFMAs?   (round 4; profiles/r04_p16_analysis.txt section 7)
//
//   hipcc --offload-arch=gfx950 -O2 -o pk_fma_coexec profiles/micro/pk_fma_coexec.hip && ./pk_fma_coexec
//
// Victim kernel: every lane runs the SAME accumulation twice -- once as v_pk_fma_f32 in four operand-select forms (or v_pk_mul_f32 + v_pk_add_f32), once as plain v_fma_f32 on the
// same numbers -- and reports the lanes where the two disagree (both are fused multiply-adds of identical operands, so any difference is a wrong
// result, not rounding).  Aggressor kernels on a second stream: a loop over ONE MFMA instruction (the gfx950 16-bit shapes 32x32x16 bf16 / f16 and 16x16x32 bf16,
// the older 32x32x8 bf16, fp32 32x32x2), or nothing.
//
// Measured on this pool's MI355X (profiles/r04_pk_fma_coexec.txt): wrong lanes ONLY for the forms whose low result reads src1's HIGH dword (v_pk_fma_f32 op_sel:[0,1,0],
// v_pk_add_f32 op_sel:[0,1]) and ONLY next to the gfx950 16-bit shapes (176 / 16 / 14 128 of 84 M lanes next to 32x32x16 bf16 / 32x32x16 f16 / 16x16x32 bf16); always in lanes
// 48-63, always the low half.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// forms: 0 = default (lo*lo, hi*hi)   1 = op_sel_hi:[1,0,1] (src1 low dword to both halves)   2 = op_sel:[0,1,0] (src1 HIGH dword to both halves)
//        3 = op_sel:[1,0,0] (src0 high dword to both halves)
//        5 = v_pk_mul_f32 (no op_sel) then v_pk_add_f32 op_sel:[0,1] (the product's high dword added to both halves)
template <int FORM>
__device__ __forceinline__ void pk(f2& acc, f2 a, f2 b) {
    if (FORM == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
    if (FORM == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(a), "v"(b));
    if (FORM == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(acc) : "v"(a), "v"(b));
    if (FORM == 3) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0]" : "+v"(acc) : "v"(a), "v"(b));
    if (FORM == 5) { f2 t; asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(t) : "v"(a), "v"(b)); asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1]" : "+v"(acc) : "v"(t)); }
}
template <int FORM>
__device__ __forceinline__ void ref(f2& acc, f2 a, f2 b) {
    const float a0 = FORM == 3 ? a.y : a.x, a1 = a.y;
    const float b0 = FORM == 2 ? b.y : b.x, b1 = FORM == 1 ? b.x : b.y;
    if (FORM == 5) {          // t = a * b; low += t.y, high += t.y
        const float ty = __fmul_rn(a.y, b.y);
        acc.x = __fadd_rn(acc.x, ty);
        acc.y = __fadd_rn(acc.y, ty);
        return;
    }
    acc.x = __builtin_fmaf(a0, b0, acc.x);
    acc.y = __builtin_fmaf(a1, b1, acc.y);
}

template <int FORM>
__global__ __launch_bounds__(256) void victim(const float* __restrict__ in, unsigned* __restrict__ bad_lanes, unsigned* __restrict__ bad_count, int iters) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    f2 a = {in[(t * 4 + 0) & 65535], in[(t * 4 + 1) & 65535]};
    f2 b = {in[(t * 4 + 2) & 65535], in[(t * 4 + 3) & 65535]};
    f2 acc[8], rf[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { acc[k] = f2{0.f, 0.f}; rf[k] = f2{0.f, 0.f}; }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) pk<FORM>(acc[k], a, b);
#pragma unroll
        for (int k = 0; k < 8; ++k) ref<FORM>(rf[k], a, b);
        // keep the operands changing (and the compiler from folding the loop)
        a.x = a.x * 0.999f + 0.001f; b.y = b.y * 1.001f - 0.001f;
        const float tmp = a.y; a.y = b.x; b.x = tmp;
    }
    unsigned wrong = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (__float_as_uint(acc[k].x) != __float_as_uint(rf[k].x)) wrong |= 1u;
        if (__float_as_uint(acc[k].y) != __float_as_uint(rf[k].y)) wrong |= 2u;
    }
    if (wrong) {
        atomicAdd(bad_count, 1u);
        atomicOr(&bad_lanes[(threadIdx.x & 63) >> 4], wrong);      // which quarter of the wave (lanes 0-15, .., 48-63), which half
    }
}

__global__ __launch_bounds__(256) void aggressor_bf16(float* __restrict__ out, int iters) {
    bf8 a, b;
#pragma unroll
    for (int k = 0; k < 8; ++k) { a[k] = (__bf16)(0.001f * (threadIdx.x + k)); b[k] = (__bf16)(0.002f * (threadIdx.x - k)); }
    f16v c0 = {}, c1 = {};
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, c1, 0, 0, 0);
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[5];
}
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void aggressor_f16(float* __restrict__ out, int iters) {
    h8 a, b;
#pragma unroll
    for (int k = 0; k < 8; ++k) { a[k] = (_Float16)(0.001f * (threadIdx.x + k)); b[k] = (_Float16)(0.002f * (threadIdx.x - k)); }
    f16v c0 = {}, c1 = {};
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, c1, 0, 0, 0);
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[5];
}
__global__ __launch_bounds__(256) void aggressor_bf16_16x16x32(float* __restrict__ out, int iters) {
    bf8 a, b;
#pragma unroll
    for (int k = 0; k < 8; ++k) { a[k] = (__bf16)(0.001f * (threadIdx.x + k)); b[k] = (__bf16)(0.002f * (threadIdx.x - k)); }
    f4v c0 = {}, c1 = {};
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, c1, 0, 0, 0);
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[3];
}
__global__ __launch_bounds__(256) void aggressor_bf16_32x32x8(float* __restrict__ out, int iters) {      // the gfx90a / gfx942 shape (4 bf16 per lane)
    s4 a, b;
#pragma unroll
    for (int k = 0; k < 4; ++k) { a[k] = (short)(0x3c00 + threadIdx.x + k); b[k] = (short)(0x3d00 + threadIdx.x - k); }
    f16v c0 = {}, c1 = {};
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(b, a, c1, 0, 0, 0);
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[5];
}
__global__ __launch_bounds__(256) void aggressor_f32(float* __restrict__ out, int iters) {
    const float a = 0.001f * threadIdx.x, b = 0.002f * threadIdx.x;
    f16v c0 = {}, c1 = {};
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, c1, 0, 0, 0);
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[5];
}

template <int FORM>
void run(const char* form, int aggr, const float* in, unsigned* lanes, unsigned* count, float* sink, hipStream_t sv, hipStream_t sa) {
    CHECK(hipMemsetAsync(lanes, 0, 16, sv));
    CHECK(hipMemsetAsync(count, 0, 4, sv));
    CHECK(hipStreamSynchronize(sv));
    for (int rep = 0; rep < 20; ++rep) {
        if (aggr == 1) hipLaunchKernelGGL(aggressor_bf16, dim3(1024), dim3(256), 0, sa, sink, 40000);
        if (aggr == 2) hipLaunchKernelGGL(aggressor_f32, dim3(1024), dim3(256), 0, sa, sink, 20000);
        if (aggr == 3) hipLaunchKernelGGL(aggressor_f16, dim3(1024), dim3(256), 0, sa, sink, 40000);
        if (aggr == 4) hipLaunchKernelGGL(aggressor_bf16_16x16x32, dim3(1024), dim3(256), 0, sa, sink, 80000);
        if (aggr == 5) hipLaunchKernelGGL(aggressor_bf16_32x32x8, dim3(1024), dim3(256), 0, sa, sink, 40000);
        for (int k = 0; k < 8; ++k) hipLaunchKernelGGL(victim<FORM>, dim3(2048), dim3(256), 0, sv, in, lanes, count, 2000);
        CHECK(hipDeviceSynchronize());
    }
    unsigned h_l[4], h_c;
    CHECK(hipMemcpy(h_l, lanes, 16, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(&h_c, count, 4, hipMemcpyDeviceToHost));
    printf("%-38s next to %-27s: %9u wrong lanes of %u   (by wave quarter, bit 0 = low half, bit 1 = high half: %u %u %u %u)\n", form,
           aggr == 0 ? "nothing" : (aggr == 1 ? "v_mfma_f32_32x32x16_bf16" : (aggr == 2 ? "v_mfma_f32_32x32x2_f32" : (aggr == 3 ? "v_mfma_f32_32x32x16_f16" :
           (aggr == 4 ? "v_mfma_f32_16x16x32_bf16" : "v_mfma_f32_32x32x8_bf16_1k")))), h_c, 20u * 8u * 2048u * 256u, h_l[0], h_l[1], h_l[2], h_l[3]);
}

int main() {
    float* in; unsigned *lanes, *count; float* sink;
    std::vector<float> h(65536);
    for (int i = 0; i < 65536; ++i) h[i] = 0.5f + 0.37f * ((i * 2654435761u >> 8) & 0xFFFF) / 65536.f;
    CHECK(hipMalloc(&in, 65536 * 4)); CHECK(hipMalloc(&lanes, 16)); CHECK(hipMalloc(&count, 4)); CHECK(hipMalloc(&sink, 1024 * 256 * 4));
    CHECK(hipMemcpy(in, h.data(), 65536 * 4, hipMemcpyHostToDevice));
    hipStream_t sv, sa;
    CHECK(hipStreamCreate(&sv)); CHECK(hipStreamCreate(&sa));
    for (int aggr = 0; aggr < 6; ++aggr) {
        run<0>("v_pk_fma_f32 (no op_sel)", aggr, in, lanes, count, sink, sv, sa);
        run<1>("v_pk_fma_f32 op_sel_hi:[1,0,1]", aggr, in, lanes, count, sink, sv, sa);
        run<2>("v_pk_fma_f32 op_sel:[0,1,0]", aggr, in, lanes, count, sink, sv, sa);
        run<3>("v_pk_fma_f32 op_sel:[1,0,0]", aggr, in, lanes, count, sink, sv, sa);
        run<5>("v_pk_mul + v_pk_add_f32 op_sel:[0,1]", aggr, in, lanes, count, sink, sv, sa);
    }
    return 0;
}
