// fp32 -> 16-bit plane splits and the 16-bit MFMA product sequences shared by the patch-resident kernels (conv_patch.hip) and the
// row-resident weight gradient (conv_wgrows.hip).
#pragma once
#include "conv_common.h"

namespace agan {
namespace conv {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every outstanding global access (vmcnt(0)),
// i.e. it would drain the patch prefetch and the weight-fragment ring at every stage; what the barriers of the second-generation
// gather kernel protect is the LDS patch and nothing else.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ u32x4 buf_load_u4s(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

// ---- fp32 -> 16-bit planes ----------------------------------------------------------------------------------------
// ET 0 = bf16, 1 = f16.  NPL planes: plane 0 = round-to-nearest of x (one plane) or its leading bits, the further planes the
// exactly representable remainders (x - hi is exact in fp32), so hi (+ mid) + lo carries 16 / 24 mantissa bits.
template <int ET>
__device__ __forceinline__ unsigned pack2_rne(float a, float b) {
    if (ET == 0) {
        f32x2 v = {a, b};
        return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
    } else {
        f32x2 v = {a, b};
        return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
    }
}
__device__ __forceinline__ unsigned pack_top16(unsigned ua, unsigned ub) {       // (ua >> 16) | (ub & 0xFFFF0000)
    return __builtin_amdgcn_perm(ub, ua, 0x07060302u);
}
template <int ET, int NPL>
__device__ __forceinline__ void split_pack2(float a, float b, unsigned (&pl)[NPL]) {
    if (NPL == 1) {
        pl[0] = pack2_rne<ET>(a, b);
    } else {
        // truncating splits: every remainder is exact, |x - sum of planes| < 2^-(8*NPL) |x|
        if (ET == 1) {      // fp16 hi (round to nearest) + fp16 of the exact remainder: 22 mantissa bits
            f32x2 v = {a, b};
            const f16x2 h = __builtin_convertvector(v, f16x2);
            const f32x2 hb = __builtin_convertvector(h, f32x2);
            pl[0] = __builtin_bit_cast(unsigned, h);
            pl[NPL - 1] = pack2_rne<1>(a - hb[0], b - hb[1]);
            return;
        }
        const unsigned a0 = __float_as_uint(a) & 0xFFFF0000u, b0 = __float_as_uint(b) & 0xFFFF0000u;
        pl[0] = pack_top16(a0, b0);
        const float ra = a - __uint_as_float(a0), rb = b - __uint_as_float(b0);
        if (NPL == 2) {
            pl[1] = pack2_rne<0>(ra, rb);
        } else {
            const unsigned a1 = __float_as_uint(ra) & 0xFFFF0000u, b1 = __float_as_uint(rb) & 0xFFFF0000u;
            pl[1] = pack_top16(a1, b1);
            pl[NPL - 1] = pack2_rne<0>(ra - __uint_as_float(a1), rb - __uint_as_float(b1));
        }
    }
}

template <int ET>
__device__ __forceinline__ f32x16 mfma16(u32x4 a, u32x4 b, f32x16 c) {
    if (ET == 0) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// products of an NPL-plane split, smallest terms first:  1: hh   2: lh hl hh   3: lh hl mm mh hm hh  (w plane, a plane)
template <int ET, int NPL>
__device__ __forceinline__ f32x16 mfma_split(const u32x4 (&w)[NPL], const u32x4 (&a)[NPL], f32x16 c) {
    if (NPL == 1) return mfma16<ET>(w[0], a[0], c);
    if (NPL == 2) {
        c = mfma16<ET>(w[1], a[0], c);
        c = mfma16<ET>(w[0], a[1], c);
        return mfma16<ET>(w[0], a[0], c);
    }
    c = mfma16<0>(w[NPL - 1], a[0], c);
    c = mfma16<0>(w[0], a[NPL - 1], c);
    c = mfma16<0>(w[1], a[1], c);
    c = mfma16<0>(w[1], a[0], c);
    c = mfma16<0>(w[0], a[1], c);
    return mfma16<0>(w[0], a[0], c);
}


}  // namespace conv
}  // namespace agan
