// Micro-benchmark: what does a wave lose around v_mfma_f32_32x32x2_f32 / v_mfma_f32_32x32x16_bf16 on gfx950, and what does another wave get for free?
// Questions behind it (round 3, fp32 Winograd kernels at 40-55 % MFMA-busy with a clean inner loop):
//   * is a chain of 4 dependent MFMAs (same accumulator) slower than 4 independent ones?
//   * how much of a VALU / LDS block placed AFTER a chain of 4 is hidden, and how much when it is spread between the MFMAs?
//   * what is the shader clock under sustained MFMA load (s_memtime cycles vs the 100 MHz wall clock)?
//   * what do 2 waves per SIMD buy?
//       hipcc --offload-arch=gfx950 -O3 mfma_issue.hip -o mfma_issue && ./mfma_issue
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// MODE 0: chains of 4 dependent MFMAs, nothing else          MODE 1: round-robin over the 16 accumulators (independent neighbours)
// MODE 2: chain of 4, then NV VALU ops                        MODE 3: one MFMA, NV/4 VALU ops, four times (spread)
// MODE 4: chain of 4 + 2 ds_read_b128 feeding the NEXT chain   MODE 5: as 4, plus NV VALU spread
template <int MODE, int NV, int NACC, int TPB = 256>
__global__ __launch_bounds__(TPB) void k(float* out, long long* cyc, long long* wall, int iters, float a0, float b0) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = (float)i * 1e-6f;
    __syncthreads();
    f32x16 acc[NACC];
#pragma unroll
    for (int q = 0; q < NACC; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    float va[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) va[i] = a0 + i + threadIdx.x;
    f32x4 za = {a0, a0 + 1, a0 + 2, a0 + 3}, zb = {b0, b0 + 1, b0 + 2, b0 + 3};
    const int lane = threadIdx.x & 63;
    const long long w0 = wall_clock64();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < NACC; ++q) {
            f32x4 na = za, nb = zb;
            if (MODE >= 4) {
                na = *reinterpret_cast<const f32x4*>(&lds[((q * 64 + lane) * 4 + it * 16) & 8188 & ~3]);
                nb = *reinterpret_cast<const f32x4*>(&lds[((q * 64 + lane) * 4 + 4096 + it * 16) & 8188 & ~3]);
            }
            if (MODE == 1) {
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[(q + 4 * j) % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(za[j], zb[j], acc[(q + 4 * j) % NACC], 0, 0, 0);
            } else if (MODE == 3 || MODE == 5) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(za[j], zb[j], acc[q], 0, 0, 0);
#pragma unroll
                    for (int v = 0; v < NV / 4; ++v) va[(v + 2 * j) & 7] = va[(v + 2 * j) & 7] * 1.0001f + va[(v + 2 * j + 1) & 7];
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(za[j], zb[j], acc[q], 0, 0, 0);
                if (MODE == 2) {
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int v = 0; v < NV; ++v) va[v & 7] = va[v & 7] * 1.0001f + va[(v + 1) & 7];
                }
            }
            za = na; zb = nb;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long t1 = clock64();
    const long long w1 = wall_clock64();
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < NACC; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[q][r];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += va[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; wall[blockIdx.x] = w1 - w0; }
}

// wgs_per_cu = 2: ONE workgroup of 512 threads per CU (two waves per SIMD, certainly co-resident; two 256-thread workgroups per CU
// turned out to run one after the other on this dispatcher: their per-workgroup wall time was half the kernel time)
template <int MODE, int NV, int NACC>
void run(const char* what, int wgs_per_cu, float* out, long long* cyc, long long* wall) {
    const int iters = 4000, nwg = 256;
    long long hc[512], hw[512];
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        if (wgs_per_cu == 2) hipLaunchKernelGGL((k<MODE, NV, NACC, 512>), dim3(nwg), dim3(512), 0, 0, out, cyc, wall, iters, 1.f, 2.f);
        else hipLaunchKernelGGL((k<MODE, NV, NACC, 256>), dim3(nwg), dim3(256), 0, 0, out, cyc, wall, iters, 1.f, 2.f);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1);
    }
    hipMemcpy(hc, cyc, nwg * 8, hipMemcpyDeviceToHost);
    hipMemcpy(hw, wall, nwg * 8, hipMemcpyDeviceToHost);
    double c = 0, w = 0;
    for (int i = 0; i < nwg; ++i) { c += hc[i]; w += hw[i]; }
    c /= nwg; w /= nwg;
    const double nm = (double)iters * NACC * 4;                        // MFMAs per wave
    const double tf = nm * 4096.0 * 4 * wgs_per_cu * nwg / (ms * 1e-3) / 1e12;
    printf("%-58s %d waves/SIMD: %7.1f memtime ticks / MFMA / wave, kernel %.3f ms = %.1f TF/s, wall %.3f ms, memtime %.0f MHz\n", what, wgs_per_cu,
           c / nm, ms, tf, w / 100e6 * 1e3, c / (w / 100e6) / 1e6);
}


// Cost model of the memory instructions around a chain of 4 MFMAs (addresses precomputed: immediate offsets only, no VALU):
// NR ds_read_b128 (consumed by the NEXT chain), NW ds_write_b128, NG buffer-style global 16-byte loads (L2-resident, consumed 16 chains
// later through the LDS writes or just accumulated), BAR: one workgroup barrier per 16 chains.
template <int NR, int NW, int NG, int BAR>
__global__ __launch_bounds__(256) void km(float* out, long long* cyc, long long* wall, const f32x4* __restrict__ gsrc, int iters, float a0, float b0) {
    __shared__ __attribute__((aligned(16))) float lds[16384];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = (float)i * 1e-6f;
    __syncthreads();
    f32x16 acc[16];
#pragma unroll
    for (int q = 0; q < 16; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    f32x4 za = {a0, a0 + 1, a0 + 2, a0 + 3}, zb = {b0, b0 + 1, b0 + 2, b0 + 3};
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const f32x4* const rbase = reinterpret_cast<const f32x4*>(lds) + lane;            // 16 B per lane, 1 KB per wave-read: conflict-free
    f32x4* const wbase = reinterpret_cast<f32x4*>(lds) + 2048 + w * 64 + lane;
    const f32x4* gp = gsrc + (blockIdx.x & 7) * 4096 + threadIdx.x;
    f32x4 gacc = {0.f, 0.f, 0.f, 0.f};
    const long long w0 = wall_clock64();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            f32x4 nr[4] = {za, zb, za, zb};
#pragma unroll
            for (int r = 0; r < NR; ++r) nr[r] = rbase[(q * 4 + r) * 64 % 1024];
            f32x4 gl[4];
#pragma unroll
            for (int r = 0; r < NG; ++r) gl[r] = gp[((q * NG + r) * 256) % 4096];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(za[j], zb[j], acc[q], 0, 0, 0);
#pragma unroll
            for (int r = 0; r < NW; ++r) wbase[(q * 4 + r) * 256 % 2048] = gacc;
#pragma unroll
            for (int r = 0; r < NG; ++r) gacc += gl[r];
            if (NR > 0) { za = nr[0]; zb = nr[NR > 1 ? 1 : 0]; }
            if (NR > 2) { za += nr[2]; }
            if (NR > 3) { zb += nr[3]; }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (BAR) __syncthreads();
    }
    const long long t1 = clock64();
    const long long w1 = wall_clock64();
    float s = gacc[0] + gacc[1] + gacc[2] + gacc[3];
#pragma unroll
    for (int q = 0; q < 16; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[q][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; wall[blockIdx.x] = w1 - w0; }
}

template <int NR, int NW, int NG, int BAR>
void runm(const char* what, float* out, long long* cyc, long long* wall, const f32x4* gsrc) {
    const int iters = 4000, nwg = 256;
    long long hc[512];
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((km<NR, NW, NG, BAR>), dim3(nwg), dim3(256), 0, 0, out, cyc, wall, gsrc, iters, 1.f, 2.f);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1);
    }
    hipMemcpy(hc, cyc, nwg * 8, hipMemcpyDeviceToHost);
    double c = 0;
    for (int i = 0; i < nwg; ++i) c += hc[i];
    c /= nwg;
    const double nm = (double)iters * 16 * 4;
    printf("%-58s : %7.1f cycles / chain of 4 MFMAs (256 = MFMA-bound), %.1f TF/s\n", what, 4 * c / nm, nm * 4096.0 * 4 * nwg / (ms * 1e-3) / 1e12);
}

// The same question for the 16-bit matrix core: chains of 4 v_mfma_f32_32x32x16_bf16 (8 passes = 32 cycles each at full rate), NV fp32 VALU
// FMAs after each chain (MODE 2) or spread between its MFMAs (MODE 3); one or two workgroups per CU.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int MODE, int NV, int NACC, int TPB = 256>
__global__ __launch_bounds__(TPB) void kb(float* out, long long* cyc, int iters, float a0) {
    f32x16 acc[NACC];
#pragma unroll
    for (int q = 0; q < NACC; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    float va[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) va[i] = a0 + i + threadIdx.x;
    bf16x8 za, zb;
#pragma unroll
    for (int i = 0; i < 8; ++i) { za[i] = (__bf16)(a0 + i); zb[i] = (__bf16)(a0 - i); }
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < NACC; ++q) {
            if (MODE == 3) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(za, zb, acc[q], 0, 0, 0);
#pragma unroll
                    for (int v = 0; v < NV / 4; ++v) va[(v + 2 * j) & 7] = va[(v + 2 * j) & 7] * 1.0001f + va[(v + 2 * j + 1) & 7];
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(za, zb, acc[q], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int v = 0; v < NV; ++v) va[v & 7] = va[v & 7] * 1.0001f + va[(v + 1) & 7];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long t1 = clock64();
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < NACC; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[q][r];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += va[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, int NV, int NACC>
void runb(const char* what, int wgs_per_cu, float* out, long long* cyc) {
    const int iters = 4000, nwg = 256;
    long long hc[512];
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        if (wgs_per_cu == 2) hipLaunchKernelGGL((kb<MODE, NV, NACC, 512>), dim3(nwg), dim3(512), 0, 0, out, cyc, iters, 1.f);
        else hipLaunchKernelGGL((kb<MODE, NV, NACC, 256>), dim3(nwg), dim3(256), 0, 0, out, cyc, iters, 1.f);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1);
    }
    hipMemcpy(hc, cyc, nwg * 8, hipMemcpyDeviceToHost);
    double c = 0;
    for (int i = 0; i < nwg; ++i) c += hc[i];
    c /= nwg;
    const double nm = (double)iters * NACC * 4;
    printf("bf16 %-53s %d waves/SIMD: %7.1f cycles / chain of 4 per wave, %.0f TF/s\n", what, wgs_per_cu, 4 * c / nm, nm * 32768.0 * 4 * wgs_per_cu * nwg / (ms * 1e-3) / 1e12);
}

// Do the MFMAs of ONE wave overlap the VALU instructions of ANOTHER wave on the same SIMD?  (The symmetric runs above cannot tell: two
// waves running the same program stay in phase -- both want the matrix pipe, then both want the VALU.)  One workgroup of 8 waves per CU:
// waves 0-3 (one per SIMD) run only MFMA chains, waves 4-7 only fp32 FMAs; ROLES bit 0 / 1 enables them.  Each role reports its own cycles.
template <int F32, int ROLES, int NMW = 1, int NVW = 1>
__global__ __launch_bounds__(256 * (NMW + NVW)) void kx(float* out, long long* cyc, int iters, float a0) {
    const int w = threadIdx.x >> 6;
    f32x16 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    float va[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) va[i] = a0 + i + threadIdx.x;
    bf16x8 za, zb;
#pragma unroll
    for (int i = 0; i < 8; ++i) { za[i] = (__bf16)(a0 + i); zb[i] = (__bf16)(a0 - i); }
    const long long t0 = clock64();
    if (w < 4 * NMW) {
        if (ROLES & 1)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (F32) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(va[j], va[j + 4], acc[q], 0, 0, 0);
                        else acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(za, zb, acc[q], 0, 0, 0);
                    }
            }
    } else if (ROLES & 2) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int v = 0; v < 64; ++v) va[v & 7] = va[v & 7] * 1.0001f + va[(v + 1) & 7];
        }
    }
    const long long t1 = clock64();
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[q][r];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += va[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + w] = t1 - t0;
}

template <int F32, int ROLES, int NMW = 1, int NVW = 1>
void runx(const char* what, float* out, long long* cyc) {
    const int iters = 4000, nwg = 256;
    static long long hc[256 * 16];
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((kx<F32, ROLES, NMW, NVW>), dim3(nwg), dim3(256 * (NMW + NVW)), 0, 0, out, cyc, iters, 1.f);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1);
    }
    hipMemcpy(hc, cyc, nwg * 16 * 8, hipMemcpyDeviceToHost);
    double cm = 0, cv = 0;
    for (int i = 0; i < nwg; ++i)
        for (int w = 0; w < 4 * (NMW + NVW); ++w) (w < 4 * NMW ? cm : cv) += hc[i * 16 + w];
    cm /= nwg * 4 * NMW; cv /= nwg * 4 * NVW;
    const double fl = (ROLES & 1) ? (double)iters * 16 * (F32 ? 4096.0 : 32768.0) * 4 * NMW * nwg : 0.0;
    printf("%s %-44s: MFMA waves %6.1f cycles per MFMA, VALU waves %5.2f cycles per FMA; kernel %.3f ms = %.0f TF/s of MFMAs\n", F32 ? "f32 " : "bf16", what,
           (ROLES & 1) ? cm / (iters * 16.0) : 0.0, (ROLES & 2) ? cv / (iters * 64.0) : 0.0, ms, fl / (ms * 1e-3) / 1e12);
}

int main() {
    float* out; long long *cyc, *wall;
    hipMalloc(&out, 512 * 256 * 4); hipMalloc(&cyc, 512 * 8); hipMalloc(&wall, 512 * 8);
    run<0, 0, 16>("chains of 4 dependent MFMAs", 1, out, cyc, wall);
    run<1, 0, 16>("independent neighbours", 1, out, cyc, wall);
    run<2, 8, 16>("chain of 4 then 8 VALU", 1, out, cyc, wall);
    run<2, 16, 16>("chain of 4 then 16 VALU", 1, out, cyc, wall);
    run<2, 32, 16>("chain of 4 then 32 VALU", 1, out, cyc, wall);
    run<3, 16, 16>("16 VALU spread between the 4 MFMAs", 1, out, cyc, wall);
    run<3, 32, 16>("32 VALU spread between the 4 MFMAs", 1, out, cyc, wall);
    run<3, 48, 16>("48 VALU spread between the 4 MFMAs", 1, out, cyc, wall);
    run<4, 0, 16>("chain of 4 + 2 ds_read_b128 for the next chain", 1, out, cyc, wall);
    run<5, 32, 16>("the same + 32 VALU spread", 1, out, cyc, wall);
    run<0, 0, 8>("chains of 4, 8 accumulators", 2, out, cyc, wall);
    run<2, 32, 8>("chain of 4 then 32 VALU, 8 accumulators", 2, out, cyc, wall);
    run<5, 32, 8>("chain + ds_read + 32 VALU spread, 8 accumulators", 2, out, cyc, wall);
    f32x4* gsrc;
    hipMalloc(&gsrc, 8 * 4096 * 16);
    hipMemset(gsrc, 0, 8 * 4096 * 16);
    runm<0, 0, 0, 0>("chain only", out, cyc, wall, gsrc);
    runm<1, 0, 0, 0>("+ 1 ds_read_b128", out, cyc, wall, gsrc);
    runm<2, 0, 0, 0>("+ 2 ds_read_b128", out, cyc, wall, gsrc);
    runm<3, 0, 0, 0>("+ 3 ds_read_b128", out, cyc, wall, gsrc);
    runm<4, 0, 0, 0>("+ 4 ds_read_b128", out, cyc, wall, gsrc);
    runm<0, 1, 0, 0>("+ 1 ds_write_b128", out, cyc, wall, gsrc);
    runm<0, 2, 0, 0>("+ 2 ds_write_b128", out, cyc, wall, gsrc);
    runm<0, 0, 1, 0>("+ 1 global 16-byte load", out, cyc, wall, gsrc);
    runm<0, 0, 2, 0>("+ 2 global 16-byte loads", out, cyc, wall, gsrc);
    runm<0, 0, 0, 1>("+ a barrier per 16 chains", out, cyc, wall, gsrc);
    runm<2, 0, 0, 1>("+ 2 ds_read_b128 + a barrier per 16 chains", out, cyc, wall, gsrc);
    runm<3, 1, 1, 1>("3 reads, 1 write, 1 global load, barrier (direct gather)", out, cyc, wall, gsrc);
    runb<2, 0, 16>("chains of 4 x 32x32x16 bf16", 1, out, cyc);
    runb<2, 8, 16>("chain of 4 then 8 VALU", 1, out, cyc);
    runb<2, 16, 16>("chain of 4 then 16 VALU", 1, out, cyc);
    runb<3, 16, 16>("16 VALU spread between the 4 MFMAs", 1, out, cyc);
    runb<3, 32, 16>("32 VALU spread between the 4 MFMAs", 1, out, cyc);
    runb<2, 0, 8>("chains of 4, 8 accumulators", 2, out, cyc);
    runb<2, 16, 8>("chain of 4 then 16 VALU, 8 accumulators", 2, out, cyc);
    runb<2, 32, 8>("chain of 4 then 32 VALU, 8 accumulators", 2, out, cyc);
    runb<3, 32, 8>("32 VALU spread, 8 accumulators", 2, out, cyc);
    hipFree(cyc); hipMalloc(&cyc, 256 * 16 * 8);
    hipFree(out); hipMalloc(&out, 256 * 1024 * 4);
    runx<1, 1>("MFMA waves alone", out, cyc);
    runx<1, 2>("VALU waves alone", out, cyc);
    runx<1, 3>("both: one MFMA wave + one VALU wave per SIMD", out, cyc);
    runx<0, 1>("MFMA waves alone", out, cyc);
    runx<0, 2>("VALU waves alone", out, cyc);
    runx<0, 3>("both: one MFMA wave + one VALU wave per SIMD", out, cyc);
    // is the foreign-instruction budget per wave or per SIMD?  two VALU waves beside one MFMA wave (12 waves; assumes the hardware places
    // wave w of a workgroup on SIMD w % 4, which the one + one results above support)
    runx<1, 3, 1, 2>("one MFMA wave + TWO VALU waves per SIMD", out, cyc);
    runx<0, 3, 1, 2>("one MFMA wave + TWO VALU waves per SIMD", out, cyc);
    return 0;
}
