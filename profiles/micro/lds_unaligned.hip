// Micro-benchmark: throughput of ds_read_b128 at 2-byte-aligned (not 16-byte-aligned) LDS addresses on gfx950, and of the same for
// global (buffer) 16-byte loads.  Question behind it: can a weight-gradient kernel keep its x tile pixel-contiguous in LDS and read the
// +-1-pixel taps as byte-shifted 16-byte fragments?       hipcc --offload-arch=gfx950 -O3 lds_unaligned.hip -o lds_unaligned && ./lds_unaligned
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(2))) U16B { u32x4 v; };

template <int STRIDE>
__global__ __launch_bounds__(256) void k(unsigned* out, long long* cyc, int shift, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    for (int i = threadIdx.x; i < 32 * STRIDE / 4 + 1024; i += 256) reinterpret_cast<unsigned*>(lds)[i] = i * 2654435761u;
    __syncthreads();
    const int lane = threadIdx.x & 63, l31 = lane & 31, lh = lane >> 5;
    const unsigned base = l31 * STRIDE + lh * 16 + shift;
    u32x4 acc = {0, 0, 0, 0};
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const U16B* p = reinterpret_cast<const U16B*>(lds + base + ((it + j) & 15) * 32);
            acc += p->v;
        }
    }
    const long long t1 = clock64();
    out[blockIdx.x * 256 + threadIdx.x] = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    unsigned* out; long long* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
    long long h[256];
    const int iters = 2000;
    for (int stride : {880, 1040}) {
        for (int shift : {0, 2, 4, 6, 8, 14}) {
            for (int rep = 0; rep < 2; ++rep) {
                if (stride == 880) hipLaunchKernelGGL(k<880>, dim3(256), dim3(256), 60000, 0, out, cyc, shift, iters);
                else hipLaunchKernelGGL(k<1040>, dim3(256), dim3(256), 60000, 0, out, cyc, shift, iters);
                hipDeviceSynchronize();
            }
            hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
            double s = 0; for (int i = 0; i < 256; ++i) s += h[i];
            // 4 waves per CU share the LDS pipe: cycles per (wave-level) ds_read_b128, per wave
            printf("stride %4d B  shift %2d B : %.1f cycles per ds_read_b128 per wave (4 waves/CU)\n", stride, shift, s / 256 / (iters * 16.0));
        }
    }
    return 0;
}
