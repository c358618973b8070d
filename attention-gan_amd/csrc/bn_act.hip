// Train-mode BatchNorm (+GLU / LeakyReLU / residual) forward and backward, plus the bare activations.
// HBM-bound streaming kernels over NCHW fp32: lanes run along the contiguous pixel axis with 16-B accesses; the
// per-channel reductions accumulate in fp64 per thread (the kernels are bandwidth-bound, the DP adds are free)
// so that var = E[x^2] - mean^2 does not cancel.
#include "agan_common.h"

#include <algorithm>
#include <type_traits>

using namespace agan;

namespace {

constexpr float kSlope = 0.2f;   // LeakyReLU slope, utilities/layers.py:124

__device__ __forceinline__ float sigmoidf_(float v) { return 1.f / (1.f + __expf(-v)); }

// ---- chunked per-channel reduction driver -----------------------------------------------------------------
// Calls f(x_index, flat_pixel_index, vec) for every element of channel c inside [beg, end) of the (b, p) space.
struct Chunk { int beg, end; };

__device__ __forceinline__ Chunk chunk_of(int n, int nchunk, int chunk) {
    int len = (cdiv(n, nchunk) + 7) & ~7;          // (whole 8-element groups: the widest vector access of the kernels below)
    Chunk c;
    c.beg = min(n, chunk * len);
    c.end = min(n, c.beg + len);
    return c;
}

// ---------------------------------------------------------------------------------------------------------
// statistics
// ---------------------------------------------------------------------------------------------------------
// XDT: storage type of x (include/agan.h: AGAN_DT_*)
template <int XDT>
__global__ __launch_bounds__(256) void bn_stats_partial_kernel(const void* __restrict__ x, int B, int C, int HW, int nchunk,
                                                               double* __restrict__ part) {
    __shared__ double red[2][4];
    const int c = blockIdx.x, chunk = blockIdx.y;
    const int n = B * HW;
    const Chunk ch = chunk_of(n, nchunk, chunk);
    double s = 0.0, q = 0.0;
    if (XDT != AGAN_DT_F32 && (HW & 7) == 0) {           // 16-bit storage: 8 values = one 16-byte load (chunks are multiples of 8 then)
        for (int i = ch.beg + threadIdx.x * 8; i < ch.end; i += 2048) {
            const int b = i / HW, p = i - b * HW;
            float v[8];
            ldv<XDT, 8>(x, ((size_t)b * C + c) * HW + p, v);
            float s4 = 0.f;                              // (8 values of one bf16 / fp16 tensor: the fp32 partial is exact enough, the chain stays fp64)
            double q4 = 0.0;
#pragma unroll
            for (int k = 0; k < 8; ++k) { s4 += v[k]; q4 += (double)v[k] * v[k]; }
            s += (double)s4;
            q += q4;
        }
    } else if ((HW & 3) == 0) {
        for (int i = ch.beg + threadIdx.x * 4; i < ch.end; i += 1024) {
            const int b = i / HW, p = i - b * HW;
            const float4 v = ld4<XDT>(x, ((size_t)b * C + c) * HW + p);
            s += (double)v.x + (double)v.y + (double)v.z + (double)v.w;
            q += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
        }
    } else {
        for (int i = ch.beg + threadIdx.x; i < ch.end; i += 256) {
            const int b = i / HW, p = i - b * HW;
            const float v = ld1<XDT>(x, ((size_t)b * C + c) * HW + p);
            s += v;
            q += (double)v * v;
        }
    }
    s = wave_sum_d(s);
    q = wave_sum_d(q);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][w] = s; red[1][w] = q; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[((size_t)c * nchunk + chunk) * 2 + 0] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        part[((size_t)c * nchunk + chunk) * 2 + 1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    }
}

__global__ __launch_bounds__(256) void bn_stats_final_kernel(const double* __restrict__ part, int C, int nchunk, int n, float eps,
                                                             float* __restrict__ mean, float* __restrict__ invstd,
                                                             float* __restrict__ rmean, float* __restrict__ rvar,
                                                             int64_t* __restrict__ nbt, float momentum) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0 && nbt) *nbt += 1;
    if (c >= C) return;
    double s = 0.0, q = 0.0;
    for (int k = 0; k < nchunk; ++k) {
        s += part[((size_t)c * nchunk + k) * 2 + 0];
        q += part[((size_t)c * nchunk + k) * 2 + 1];
    }
    const double m = s / n;
    double var = q / n - m * m;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)m;
    invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (rmean) {
        const double unbiased = n > 1 ? var * ((double)n / (n - 1)) : var;
        rmean[c] = (float)((1.0 - momentum) * rmean[c] + momentum * m);
        rvar[c] = (float)((1.0 - momentum) * rvar[c] + momentum * unbiased);
    }
}

// HW == 1 (BatchNorm1d on [B, C], generator_submodules.py:38): one thread per channel, coalesced across channels
__global__ __launch_bounds__(256) void bn_stats_rows_kernel(const float* __restrict__ x, int B, int C, float eps,
                                                            float* __restrict__ mean, float* __restrict__ invstd,
                                                            float* __restrict__ rmean, float* __restrict__ rvar,
                                                            int64_t* __restrict__ nbt, float momentum) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0 && nbt) *nbt += 1;
    if (c >= C) return;
    double s = 0.0, q = 0.0;
    for (int b = 0; b < B; ++b) {
        const float v = x[(size_t)b * C + c];
        s += v;
        q += (double)v * v;
    }
    const double m = s / B;
    double var = q / B - m * m;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)m;
    invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (rmean) {
        const double unbiased = B > 1 ? var * ((double)B / (B - 1)) : var;
        rmean[c] = (float)((1.0 - momentum) * rmean[c] + momentum * m);
        rvar[c] = (float)((1.0 - momentum) * rvar[c] + momentum * unbiased);
    }
}

int stats_chunks(int B, int C, int HW) {
    const int n = B * HW;
    int nchunk = std::max(1, 2048 / std::max(C, 1));
    nchunk = std::min(nchunk, std::max(1, n / 2048));
    return nchunk;
}

// ---------------------------------------------------------------------------------------------------------
// forward apply
// ---------------------------------------------------------------------------------------------------------
struct Affine { float s, t; };
__device__ __forceinline__ Affine affine_of(const float* mean, const float* invstd, const float* gamma, const float* beta, int c) {
    Affine a;
    a.s = gamma[c] * invstd[c];
    a.t = beta[c] - mean[c] * a.s;
    return a;
}

// XDT: storage type of x; ODT: of out and the residual
// V: elements per thread and iteration (1, 4, or 8 where a 16-bit tensor is involved: 16-byte accesses on it)
template <int ACT, int V, int XDT, int ODT>
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const void* __restrict__ x, const float* __restrict__ mean,
                                                         const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, const void* __restrict__ res,
                                                         void* __restrict__ out, int B, int C, int HW, float* __restrict__ amax) {
    const int Co = (ACT == AGAN_ACT_GLU) ? C / 2 : C;
    const size_t total = (size_t)B * Co * HW / V;
    float mx = 0.f;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t i = e * V;
        const int p = (int)(i % HW);
        const size_t t = i / HW;
        const int c = (int)(t % Co), b = (int)(t / Co);
        const size_t xi = ((size_t)b * C + c) * HW + p;
        const Affine a = affine_of(mean, invstd, gamma, beta, c);
        float xv[V], yv[V];
        ldv<XDT, V>(x, xi, xv);
        if (ACT == AGAN_ACT_GLU) {
            const Affine g = affine_of(mean, invstd, gamma, beta, c + Co);
            float gv[V];
            ldv<XDT, V>(x, xi + (size_t)Co * HW, gv);
#pragma unroll
            for (int k = 0; k < V; ++k) yv[k] = (xv[k] * a.s + a.t) * sigmoidf_(gv[k] * g.s + g.t);
        } else {
#pragma unroll
            for (int k = 0; k < V; ++k) {
                float z = xv[k] * a.s + a.t;
                if (ACT == AGAN_ACT_LRELU) z = z >= 0.f ? z : z * kSlope;
                yv[k] = z;
            }
            if (res) {
                float rv[V];
                ldv<ODT, V>(res, i, rv);
#pragma unroll
                for (int k = 0; k < V; ++k) yv[k] += rv[k];
            }
        }
#pragma unroll
        for (int k = 0; k < V; ++k) mx = fmaxf(mx, fabsf(yv[k]));
        stv<ODT, V>(out, i, yv);
    }
    if (amax) amax_commit(mx, amax);
}

// ---------------------------------------------------------------------------------------------------------
// backward: pass 1 = per-channel sums of dz and dz*xhat, pass 2 = dx
// ---------------------------------------------------------------------------------------------------------
// dz for one element.  GLU: returns dz of the value half in dza and of the gate half in dzg.
template <int ACT>
__device__ __forceinline__ void dz_of(float xa, float xg, float d, const Affine& a, const Affine& g, float& dza, float& dzg) {
    if (ACT == AGAN_ACT_GLU) {
        const float za = xa * a.s + a.t;
        const float sg = sigmoidf_(xg * g.s + g.t);
        dza = d * sg;
        dzg = d * za * sg * (1.f - sg);
    } else if (ACT == AGAN_ACT_LRELU) {
        const float z = xa * a.s + a.t;
        dza = z >= 0.f ? d : d * kSlope;
        dzg = 0.f;
    } else {
        dza = d;
        dzg = 0.f;
    }
}

template <int ACT, int XDT, int ODT>
__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const void* __restrict__ x, const void* __restrict__ dout,
                                                             const float* __restrict__ mean, const float* __restrict__ invstd,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             int B, int C, int HW, int nchunk, double* __restrict__ part) {
    // GLU: block handles the channel pair (c, c+C/2).  part[c][chunk][4] = {sum dz_a, sum dz_a*xhat_a, sum dz_g, sum dz_g*xhat_g}
    __shared__ double red[4][4];
    const int Co = (ACT == AGAN_ACT_GLU) ? C / 2 : C;
    const int c = blockIdx.x, chunk = blockIdx.y;
    const int n = B * HW;
    const Chunk ch = chunk_of(n, nchunk, chunk);
    const Affine a = affine_of(mean, invstd, gamma, beta, c);
    const Affine g = (ACT == AGAN_ACT_GLU) ? affine_of(mean, invstd, gamma, beta, c + Co) : a;
    const float ma = mean[c], ia = invstd[c];
    const float mg = (ACT == AGAN_ACT_GLU) ? mean[c + Co] : 0.f, ig = (ACT == AGAN_ACT_GLU) ? invstd[c + Co] : 0.f;
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    // one loop per vector width (compile-time inside: no per-element width tests): 8 values where a 16-bit tensor is involved and the rows
    // are whole 8-groups (16-byte accesses on it), else 4, else 1
    auto sweep = [&](auto nv_tag) {
        constexpr int NV = decltype(nv_tag)::value;
        for (int i = ch.beg + threadIdx.x * NV; i < ch.end; i += 256 * NV) {
            const int b = i / HW, p = i - b * HW;
            const size_t xi = ((size_t)b * C + c) * HW + p, di = ((size_t)b * Co + c) * HW + p;
            float xa[NV], xg[NV], d[NV];
            ldv<XDT, NV>(x, xi, xa);
            ldv<ODT, NV>(dout, di, d);
            if (ACT == AGAN_ACT_GLU) ldv<XDT, NV>(x, xi + (size_t)Co * HW, xg);
            float t0 = 0.f, t2 = 0.f;                 // (sums of <= 8 values in fp32, the running chains in fp64)
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                float dza, dzg;
                dz_of<ACT>(xa[k], ACT == AGAN_ACT_GLU ? xg[k] : 0.f, d[k], a, g, dza, dzg);
                t0 += dza;
                s1 += (double)dza * ((xa[k] - ma) * ia);
                if (ACT == AGAN_ACT_GLU) {
                    t2 += dzg;
                    s3 += (double)dzg * ((xg[k] - mg) * ig);
                }
            }
            s0 += (double)t0;
            if (ACT == AGAN_ACT_GLU) s2 += (double)t2;
        }
    };
    constexpr bool any16 = XDT != AGAN_DT_F32 || ODT != AGAN_DT_F32;
    if (any16 && (HW & 7) == 0) sweep(std::integral_constant<int, 8>{});
    else if ((HW & 3) == 0) sweep(std::integral_constant<int, 4>{});
    else sweep(std::integral_constant<int, 1>{});
    s0 = wave_sum_d(s0); s1 = wave_sum_d(s1);
    if (ACT == AGAN_ACT_GLU) { s2 = wave_sum_d(s2); s3 = wave_sum_d(s3); }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][w] = s0; red[1][w] = s1; red[2][w] = s2; red[3][w] = s3; }
    __syncthreads();
    if (threadIdx.x < 4) {
        const int k = threadIdx.x;
        part[((size_t)c * nchunk + chunk) * 4 + k] = red[k][0] + red[k][1] + red[k][2] + red[k][3];
    }
}

// coef[c] = {mean dz, mean dz*xhat}; dgamma, dbeta
__global__ __launch_bounds__(256) void bn_bwd_final_kernel(const double* __restrict__ part, int C, int Co, int nchunk, int n, bool glu,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                           float* __restrict__ coef, int accumulate) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const int pc = (glu && c >= Co) ? c - Co : c;
    const int off = (glu && c >= Co) ? 2 : 0;
    double s = 0, q = 0;
    for (int k = 0; k < nchunk; ++k) {
        s += part[((size_t)pc * nchunk + k) * 4 + off];
        q += part[((size_t)pc * nchunk + k) * 4 + off + 1];
    }
    dbeta[c] = accumulate ? dbeta[c] + (float)s : (float)s;
    dgamma[c] = accumulate ? dgamma[c] + (float)q : (float)q;
    coef[2 * c] = (float)(s / n);
    coef[2 * c + 1] = (float)(q / n);
}

template <int ACT, int V, int XDT, int ODT>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const void* __restrict__ x, const void* __restrict__ dout,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ coef, void* __restrict__ dx, int B, int C, int HW,
                                                           float* __restrict__ amax) {
    const int Co = (ACT == AGAN_ACT_GLU) ? C / 2 : C;
    const size_t total = (size_t)B * Co * HW / V;
    float mx = 0.f;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t i = e * V;
        const int p = (int)(i % HW);
        const size_t t = i / HW;
        const int c = (int)(t % Co), b = (int)(t / Co);
        const size_t xi = ((size_t)b * C + c) * HW + p;
        const Affine a = affine_of(mean, invstd, gamma, beta, c);
        const Affine g = (ACT == AGAN_ACT_GLU) ? affine_of(mean, invstd, gamma, beta, c + Co) : a;
        const float ma = mean[c], ia = invstd[c], c0 = coef[2 * c], c1 = coef[2 * c + 1];
        float xa[V], xg[V], d[V], oa[V], og[V];
#pragma unroll
        for (int k = 0; k < V; ++k) xg[k] = 0.f;
        ldv<XDT, V>(x, xi, xa);
        ldv<ODT, V>(dout, i, d);
        if (ACT == AGAN_ACT_GLU) ldv<XDT, V>(x, xi + (size_t)Co * HW, xg);
        float mg = 0, ig = 0, g0 = 0, g1 = 0;
        if (ACT == AGAN_ACT_GLU) { mg = mean[c + Co]; ig = invstd[c + Co]; g0 = coef[2 * (c + Co)]; g1 = coef[2 * (c + Co) + 1]; }
#pragma unroll
        for (int k = 0; k < V; ++k) {
            float dza, dzg;
            dz_of<ACT>(xa[k], xg[k], d[k], a, g, dza, dzg);
            oa[k] = a.s * (dza - c0 - (xa[k] - ma) * ia * c1);
            mx = fmaxf(mx, fabsf(oa[k]));
            if (ACT == AGAN_ACT_GLU) { og[k] = g.s * (dzg - g0 - (xg[k] - mg) * ig * g1); mx = fmaxf(mx, fabsf(og[k])); }
        }
        stv<XDT, V>(dx, xi, oa);
        if (ACT == AGAN_ACT_GLU) stv<XDT, V>(dx, xi + (size_t)Co * HW, og);
    }
    if (amax) amax_commit(mx, amax);
}

// ---------------------------------------------------------------------------------------------------------
// bare activations
// ---------------------------------------------------------------------------------------------------------
template <int ACT>
__device__ __forceinline__ float act1(float v) {
    if (ACT == AGAN_ACT_LRELU) return v >= 0.f ? v : v * kSlope;
    if (ACT == AGAN_ACT_TANH) return tanhf(v);
    if (ACT == AGAN_ACT_SIGMOID) return 1.f / (1.f + expf(-v));
    return v;
}
// derivative expressed from the OUTPUT y (lrelu keeps the sign, so y works there too)
template <int ACT>
__device__ __forceinline__ float dact1(float y, float d) {
    if (ACT == AGAN_ACT_LRELU) return y >= 0.f ? d : d * kSlope;
    if (ACT == AGAN_ACT_TANH) return d * (1.f - y * y);
    if (ACT == AGAN_ACT_SIGMOID) return d * y * (1.f - y);
    return d;
}

template <int ACT>
__global__ __launch_bounds__(256) void act_fwd_kernel(const float* __restrict__ x, float* __restrict__ out, size_t n) {
    const size_t n4 = n / 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 v = reinterpret_cast<const float4*>(x)[i];
        v.x = act1<ACT>(v.x); v.y = act1<ACT>(v.y); v.z = act1<ACT>(v.z); v.w = act1<ACT>(v.w);
        reinterpret_cast<float4*>(out)[i] = v;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) out[n4 * 4 + threadIdx.x] = act1<ACT>(x[n4 * 4 + threadIdx.x]);
}
template <int ACT>
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ y, const float* __restrict__ d, float* __restrict__ dx, size_t n) {
    const size_t n4 = n / 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 yv = reinterpret_cast<const float4*>(y)[i], dv = reinterpret_cast<const float4*>(d)[i];
        float4 o;
        o.x = dact1<ACT>(yv.x, dv.x); o.y = dact1<ACT>(yv.y, dv.y); o.z = dact1<ACT>(yv.z, dv.z); o.w = dact1<ACT>(yv.w, dv.w);
        reinterpret_cast<float4*>(dx)[i] = o;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const size_t e = n4 * 4 + threadIdx.x;
        dx[e] = dact1<ACT>(y[e], d[e]);
    }
}

// standalone GLU (the CA-net "relu" is a GLU: generator_submodules.py:153; layers.py:20-24)
__global__ __launch_bounds__(256) void glu_fwd_kernel(const float* __restrict__ x, float* __restrict__ out, int B, int C, int HW) {
    const int Co = C / 2;
    const size_t total = (size_t)B * Co * HW;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t plane = (size_t)Co * HW, b = i / plane, r = i - b * plane;
        const size_t xi = b * 2 * plane + r;
        out[i] = x[xi] * sigmoidf_(x[xi + plane]);
    }
}
__global__ __launch_bounds__(256) void glu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dout, float* __restrict__ dx,
                                                      int B, int C, int HW) {
    const int Co = C / 2;
    const size_t total = (size_t)B * Co * HW;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t plane = (size_t)Co * HW, b = i / plane, r = i - b * plane;
        const size_t xi = b * 2 * plane + r;
        const float a = x[xi], sg = sigmoidf_(x[xi + plane]), d = dout[i];
        dx[xi] = d * sg;
        dx[xi + plane] = d * a * sg * (1.f - sg);
    }
}

// ---------------------------------------------------------------------------------------------------------
// small tensors (B*HW <= kSmallN per channel: the 4x4 / 8x8 / 16x16 layers): statistics + normalise + activation in ONE launch,
// one workgroup per (output) channel with the channel's values held in registers between the two phases
// ---------------------------------------------------------------------------------------------------------
constexpr int kSmallN = 8192;
constexpr int kSmallPer = kSmallN / 256;

__device__ __forceinline__ void block_sum2_d(double& a, double& b, double (*red)[4]) {
    a = wave_sum_d(a);
    b = wave_sum_d(b);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { red[0][w] = a; red[1][w] = b; }
    __syncthreads();
    a = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    b = red[1][0] + red[1][1] + red[1][2] + red[1][3];
}

// PER = values a thread holds per half (round 4): 32 covers kSmallN; the 4x4 / 8x8 maps (n <= 512 / 2048) take PER = 2 / 8 -- the 32-deep
// form spent most of its ~11 us on 30 dead, fully unrolled iterations (an integer division and four predicated loads each)
template <int ACT, int XDT, int ODT, int PER = kSmallPer>
__global__ __launch_bounds__(256) void bn_small_fwd_kernel(const void* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const void* __restrict__ res, void* __restrict__ out, float* __restrict__ mean,
                                                           float* __restrict__ invstd, float* __restrict__ rmean, float* __restrict__ rvar,
                                                           int64_t* __restrict__ nbt, int B, int C, int HW, float eps, float momentum,
                                                           float* __restrict__ amax) {
    __shared__ double red[2][4];
    constexpr bool GLU = ACT == AGAN_ACT_GLU;
    const int Co = GLU ? C / 2 : C;
    const int c = blockIdx.x, n = B * HW;
    if (c == 0 && threadIdx.x == 0 && nbt) *nbt += 1;
    float va[PER], vg[GLU ? PER : 1];
    double s = 0.0, q = 0.0, sg = 0.0, qg = 0.0;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int i = threadIdx.x + j * 256;
        va[j] = 0.f;
        if (GLU) vg[j] = 0.f;
        if (i < n) {
            const int b = i / HW, p = i - b * HW;
            const size_t xi = ((size_t)b * C + c) * HW + p;
            va[j] = ld1<XDT>(x, xi);
            s += va[j];
            q += (double)va[j] * va[j];
            if (GLU) {
                vg[j] = ld1<XDT>(x, xi + (size_t)Co * HW);
                sg += vg[j];
                qg += (double)vg[j] * vg[j];
            }
        }
    }
    block_sum2_d(s, q, red);
    const double m = s / n;
    double var = q / n - m * m;
    var = var < 0.0 ? 0.0 : var;
    const float is = (float)(1.0 / sqrt(var + (double)eps));
    float mgf = 0.f, isg = 0.f;
    double mg = 0.0, varg = 0.0;
    if (GLU) {
        block_sum2_d(sg, qg, red);
        mg = sg / n;
        varg = qg / n - mg * mg;
        varg = varg < 0.0 ? 0.0 : varg;
        mgf = (float)mg;
        isg = (float)(1.0 / sqrt(varg + (double)eps));
    }
    if (threadIdx.x == 0) {
        const double ub = n > 1 ? (double)n / (n - 1) : 1.0;
        mean[c] = (float)m;
        invstd[c] = is;
        if (rmean) {
            rmean[c] = (float)((1.0 - momentum) * rmean[c] + momentum * m);
            rvar[c] = (float)((1.0 - momentum) * rvar[c] + momentum * var * ub);
        }
        if (GLU) {
            mean[c + Co] = mgf;
            invstd[c + Co] = isg;
            if (rmean) {
                rmean[c + Co] = (float)((1.0 - momentum) * rmean[c + Co] + momentum * mg);
                rvar[c + Co] = (float)((1.0 - momentum) * rvar[c + Co] + momentum * varg * ub);
            }
        }
    }
    const float sa = gamma[c] * is, ta = beta[c] - (float)m * sa;
    float sgs = 0.f, tg = 0.f, mx = 0.f;
    if (GLU) { sgs = gamma[c + Co] * isg; tg = beta[c + Co] - mgf * sgs; }
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int i = threadIdx.x + j * 256;
        if (i < n) {
            const int b = i / HW, p = i - b * HW;
            const size_t oi = ((size_t)b * Co + c) * HW + p;
            float z = va[j] * sa + ta;
            if (GLU) z *= sigmoidf_(vg[j] * sgs + tg);
            else if (ACT == AGAN_ACT_LRELU) z = z >= 0.f ? z : z * kSlope;
            if (!GLU && ACT == AGAN_ACT_NONE && res) z += ld1<ODT>(res, oi);
            st1<ODT>(out, oi, z);
            mx = fmaxf(mx, fabsf(z));
        }
    }
    if (amax) amax_commit(mx, amax);
}

template <int ACT, int XDT, int ODT, int PER = kSmallPer>
__global__ __launch_bounds__(256) void bn_small_bwd_kernel(const void* __restrict__ x, const void* __restrict__ dout, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, void* __restrict__ dx, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta, int B, int C, int HW, int accumulate,
                                                           float* __restrict__ amax) {
    __shared__ double red[2][4];
    constexpr bool GLU = ACT == AGAN_ACT_GLU;
    const int Co = GLU ? C / 2 : C;
    const int c = blockIdx.x, n = B * HW;
    const Affine a = affine_of(mean, invstd, gamma, beta, c);
    const Affine g = GLU ? affine_of(mean, invstd, gamma, beta, c + Co) : a;
    const float ma = mean[c], ia = invstd[c];
    const float mg = GLU ? mean[c + Co] : 0.f, ig = GLU ? invstd[c + Co] : 0.f;
    float xa[PER], za[PER], xg[GLU ? PER : 1], zg[GLU ? PER : 1];
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int i = threadIdx.x + j * 256;
        xa[j] = 0.f; za[j] = 0.f;
        if (GLU) { xg[j] = 0.f; zg[j] = 0.f; }
        if (i < n) {
            const int b = i / HW, p = i - b * HW;
            const size_t xi = ((size_t)b * C + c) * HW + p;
            xa[j] = ld1<XDT>(x, xi);
            const float d = ld1<ODT>(dout, ((size_t)b * Co + c) * HW + p);
            float xgv = 0.f;
            if (GLU) { xgv = ld1<XDT>(x, xi + (size_t)Co * HW); xg[j] = xgv; }
            float dza, dzg;
            dz_of<ACT>(xa[j], xgv, d, a, g, dza, dzg);
            za[j] = dza;
            s0 += dza;
            s1 += (double)dza * ((xa[j] - ma) * ia);
            if (GLU) {
                zg[j] = dzg;
                s2 += dzg;
                s3 += (double)dzg * ((xgv - mg) * ig);
            }
        }
    }
    block_sum2_d(s0, s1, red);
    if (GLU) block_sum2_d(s2, s3, red);
    if (threadIdx.x == 0) {
        dbeta[c] = accumulate ? dbeta[c] + (float)s0 : (float)s0;
        dgamma[c] = accumulate ? dgamma[c] + (float)s1 : (float)s1;
        if (GLU) {
            dbeta[c + Co] = accumulate ? dbeta[c + Co] + (float)s2 : (float)s2;
            dgamma[c + Co] = accumulate ? dgamma[c + Co] + (float)s3 : (float)s3;
        }
    }
    const float c0 = (float)(s0 / n), c1 = (float)(s1 / n), g0 = (float)(s2 / n), g1 = (float)(s3 / n);
    float mx = 0.f;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int i = threadIdx.x + j * 256;
        if (i < n) {
            const int b = i / HW, p = i - b * HW;
            const size_t xi = ((size_t)b * C + c) * HW + p;
            const float va = a.s * (za[j] - c0 - (xa[j] - ma) * ia * c1);
            st1<XDT>(dx, xi, va);
            mx = fmaxf(mx, fabsf(va));
            if (GLU) {
                const float vg2 = g.s * (zg[j] - g0 - (xg[j] - mg) * ig * g1);
                st1<XDT>(dx, xi + (size_t)Co * HW, vg2);
                mx = fmaxf(mx, fabsf(vg2));
            }
        }
    }
    if (amax) amax_commit(mx, amax);
}

int ew_blocks(size_t work) { return (int)std::max<size_t>(1, std::min<size_t>(cdivz(work, 256), 256 * 8)); }

}  // namespace

// storage-type combinations the BatchNorm kernels are built for: (x / dx, out / dout / residual)
#define AGAN_BN_DT(XD, OD, CALL)                                                                                  \
    do {                                                                                                          \
        if ((XD) == AGAN_DT_F32 && (OD) == AGAN_DT_F32) { CALL(AGAN_DT_F32, AGAN_DT_F32); }                       \
        else if ((XD) == AGAN_DT_BF16 && (OD) == AGAN_DT_BF16) { CALL(AGAN_DT_BF16, AGAN_DT_BF16); }              \
        else if ((XD) == AGAN_DT_F32 && (OD) == AGAN_DT_BF16) { CALL(AGAN_DT_F32, AGAN_DT_BF16); }                \
        else if ((XD) == AGAN_DT_F16 && (OD) == AGAN_DT_F16) { CALL(AGAN_DT_F16, AGAN_DT_F16); }                  \
        else { CALL(AGAN_DT_F32, AGAN_DT_F16); }                                                                  \
    } while (0)
static bool bn_dt_ok(int xd, int od) {
    return (xd == AGAN_DT_F32 && (od == AGAN_DT_F32 || od == AGAN_DT_BF16 || od == AGAN_DT_F16)) || (xd == od && (xd == AGAN_DT_BF16 || xd == AGAN_DT_F16));
}
static const void* off(const void* p, size_t elems, int dt) { return static_cast<const char*>(p) + elems * dt_size(dt); }
static void* off(void* p, size_t elems, int dt) { return static_cast<char*>(p) + elems * dt_size(dt); }

extern "C" {

size_t agan_bn_stats_ws_bytes(int B, int C, int HW) {
    if (HW == 1) return 0;
    return (size_t)C * stats_chunks(B, C, HW) * 2 * sizeof(double);
}

int agan_bn_stats_dt(const void* x, int B, int C, int HW, float eps, float* mean, float* invstd, float* running_mean,
                     float* running_var, int64_t* nbt, float momentum, void* ws, size_t ws_bytes, void* stream, int x_dtype) {
    AGAN_REQUIRE(x && mean && invstd && B > 0 && C > 0 && HW > 0, "bn_stats: bad argument");
    AGAN_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_stats: running_mean/var must come together");
    AGAN_REQUIRE(x_dtype == AGAN_DT_F32 || x_dtype == AGAN_DT_BF16 || x_dtype == AGAN_DT_F16, "bn_stats: storage type %d", x_dtype);
    hipStream_t st = as_stream(stream);
    if (HW == 1) {
        AGAN_REQUIRE(x_dtype == AGAN_DT_F32, "bn_stats: [B, C] inputs (BatchNorm1d) are fp32 only");
        hipLaunchKernelGGL(bn_stats_rows_kernel, dim3(cdiv(C, 256)), dim3(256), 0, st, static_cast<const float*>(x), B, C, eps, mean, invstd,
                           running_mean, running_var, nbt, momentum);
        return check_launch("bn_stats_rows");
    }
    const int nchunk = stats_chunks(B, C, HW);
    if (ws_bytes < agan_bn_stats_ws_bytes(B, C, HW) || !ws) {
        set_error("bn_stats: workspace too small");
        return AGAN_EWORKSPACE;
    }
    double* part = static_cast<double*>(ws);
    if (x_dtype == AGAN_DT_F32) hipLaunchKernelGGL(bn_stats_partial_kernel<AGAN_DT_F32>, dim3(C, nchunk), dim3(256), 0, st, x, B, C, HW, nchunk, part);
    else if (x_dtype == AGAN_DT_BF16) hipLaunchKernelGGL(bn_stats_partial_kernel<AGAN_DT_BF16>, dim3(C, nchunk), dim3(256), 0, st, x, B, C, HW, nchunk, part);
    else hipLaunchKernelGGL(bn_stats_partial_kernel<AGAN_DT_F16>, dim3(C, nchunk), dim3(256), 0, st, x, B, C, HW, nchunk, part);
    hipLaunchKernelGGL(bn_stats_final_kernel, dim3(cdiv(C, 256)), dim3(256), 0, st, part, C, nchunk, B * HW, eps, mean, invstd,
                       running_mean, running_var, nbt, momentum);
    return check_launch("bn_stats");
}
int agan_bn_stats(const float* x, int B, int C, int HW, float eps, float* mean, float* invstd, float* running_mean,
                  float* running_var, int64_t* nbt, float momentum, void* ws, size_t ws_bytes, void* stream) {
    return agan_bn_stats_dt(x, B, C, HW, eps, mean, invstd, running_mean, running_var, nbt, momentum, ws, ws_bytes, stream, AGAN_DT_F32);
}

int agan_bn_act_fwd_dt(const void* x, const float* mean, const float* invstd, const float* gamma, const float* beta,
                       const void* residual, void* out, int B, int C, int HW, int act, void* stream, float* out_amax, int x_dtype,
                       int out_dtype) {
    AGAN_REQUIRE(x && mean && invstd && gamma && beta && out && B > 0 && C > 0 && HW > 0, "bn_act_fwd: bad argument");
    AGAN_REQUIRE(act == AGAN_ACT_NONE || act == AGAN_ACT_GLU || act == AGAN_ACT_LRELU, "bn_act_fwd: activation %d", act);
    AGAN_REQUIRE(act != AGAN_ACT_GLU || (C % 2 == 0), "channels dont divide 2!");
    AGAN_REQUIRE(!(residual && act != AGAN_ACT_NONE), "bn_act_fwd: residual only with ACT_NONE");
    AGAN_REQUIRE(bn_dt_ok(x_dtype, out_dtype), "bn_act_fwd: storage types %d -> %d", x_dtype, out_dtype);
    hipStream_t st = as_stream(stream);
    // elements per thread and iteration: 8 where a 16-bit tensor is involved and rows are whole 8-groups (16-byte accesses on it), else 4 / 1
    const int vw = ((x_dtype != AGAN_DT_F32 || out_dtype != AGAN_DT_F32) && (HW & 7) == 0) ? 8 : ((HW & 3) == 0 ? 4 : 1);
    const int Co = act == AGAN_ACT_GLU ? C / 2 : C;
    const int blocks = ew_blocks((size_t)B * Co * HW / vw);
#define AGAN_L(A, V, XD, OD) hipLaunchKernelGGL((bn_act_fwd_kernel<A, V, XD, OD>), dim3(blocks), dim3(256), 0, st, x, mean, invstd, gamma, beta, residual, out, B, C, HW, out_amax)
#define AGAN_V(A, XD, OD)                                                                                           \
    do {                                                                                                            \
        if (vw == 8) { if constexpr ((XD) != AGAN_DT_F32 || (OD) != AGAN_DT_F32) AGAN_L(A, 8, XD, OD); }             \
        else if (vw == 4) AGAN_L(A, 4, XD, OD);                                                                     \
        else AGAN_L(A, 1, XD, OD);                                                                                  \
    } while (0)
#define AGAN_C(XD, OD)                                                  \
    if (act == AGAN_ACT_GLU) AGAN_V(AGAN_ACT_GLU, XD, OD);              \
    else if (act == AGAN_ACT_LRELU) AGAN_V(AGAN_ACT_LRELU, XD, OD);     \
    else AGAN_V(AGAN_ACT_NONE, XD, OD);
    AGAN_BN_DT(x_dtype, out_dtype, AGAN_C);
#undef AGAN_C
#undef AGAN_V
#undef AGAN_L
    return check_launch("bn_act_fwd");
}
int agan_bn_act_fwd(const float* x, const float* mean, const float* invstd, const float* gamma, const float* beta,
                    const float* residual, float* out, int B, int C, int HW, int act, void* stream, float* out_amax) {
    return agan_bn_act_fwd_dt(x, mean, invstd, gamma, beta, residual, out, B, C, HW, act, stream, out_amax, AGAN_DT_F32, AGAN_DT_F32);
}

static int bwd_chunks(int B, int C, int HW) { return stats_chunks(B, C, HW); }

size_t agan_bn_act_bwd_ws_bytes(int B, int C, int HW) {
    return (size_t)C * bwd_chunks(B, C, HW) * 4 * sizeof(double) + (size_t)C * 2 * sizeof(float) + 64;
}

int agan_bn_act_bwd_dt(const void* x, const void* dout, const float* mean, const float* invstd, const float* gamma,
                       const float* beta, void* dx, float* dgamma, float* dbeta, int B, int C, int HW, int act, int accumulate,
                       int groups, void* ws, size_t ws_bytes, void* stream, float* dx_amax, int x_dtype, int out_dtype) {
    AGAN_REQUIRE(x && dout && mean && invstd && gamma && beta && dx && dgamma && dbeta && ws, "bn_act_bwd: null pointer");
    AGAN_REQUIRE(act == AGAN_ACT_NONE || act == AGAN_ACT_GLU || act == AGAN_ACT_LRELU, "bn_act_bwd: activation %d", act);
    AGAN_REQUIRE(act != AGAN_ACT_GLU || (C % 2 == 0), "channels dont divide 2!");
    AGAN_REQUIRE(groups >= 1 && B % groups == 0, "bn_act_bwd: batch %d does not split into %d groups", B, groups);
    AGAN_REQUIRE(bn_dt_ok(x_dtype, out_dtype), "bn_act_bwd: storage types %d -> %d", x_dtype, out_dtype);
    const int Bg = B / groups;
    if (ws_bytes < agan_bn_act_bwd_ws_bytes(Bg, C, HW)) {
        set_error("bn_act_bwd: workspace too small");
        return AGAN_EWORKSPACE;
    }
    hipStream_t st = as_stream(stream);
    const bool glu = act == AGAN_ACT_GLU;
    const int Co = glu ? C / 2 : C;
    if (HW > 1 && (long long)Bg * HW <= kSmallN) {
        for (int gi = 0; gi < groups; ++gi) {      // (one launch per group: see agan_bn_train_fwd)
            const void* xg = off(x, (size_t)gi * Bg * C * HW, x_dtype);
            const void* dg = off(dout, (size_t)gi * Bg * Co * HW, out_dtype);
            void* dxg = off(dx, (size_t)gi * Bg * C * HW, x_dtype);
            const int acc = gi == 0 ? accumulate : 1;
#define AGAN_LP(A, XD, OD, P) hipLaunchKernelGGL((bn_small_bwd_kernel<A, XD, OD, P>), dim3(Co), dim3(256), 0, st, xg, dg, mean + (size_t)gi * C, invstd + (size_t)gi * C, gamma, beta, dxg, dgamma, dbeta, Bg, C, HW, acc, dx_amax)
#define AGAN_L(A, XD, OD) do { if (Bg * HW <= 512) AGAN_LP(A, XD, OD, 2); else if (Bg * HW <= 2048) AGAN_LP(A, XD, OD, 8); else AGAN_LP(A, XD, OD, kSmallPer); } while (0)
#define AGAN_C(XD, OD)                                           \
    if (act == AGAN_ACT_GLU) AGAN_L(AGAN_ACT_GLU, XD, OD);       \
    else if (act == AGAN_ACT_LRELU) AGAN_L(AGAN_ACT_LRELU, XD, OD); \
    else AGAN_L(AGAN_ACT_NONE, XD, OD);
            AGAN_BN_DT(x_dtype, out_dtype, AGAN_C);
#undef AGAN_C
#undef AGAN_L
#undef AGAN_LP
        }
        return check_launch("bn_act_bwd/small");
    }
    const int nchunk = bwd_chunks(Bg, C, HW);
    double* part = static_cast<double*>(ws);
    float* coef = reinterpret_cast<float*>(part + (size_t)C * nchunk * 4);
    dim3 grid(Co, nchunk);
    const int vw = ((x_dtype != AGAN_DT_F32 || out_dtype != AGAN_DT_F32) && (HW & 7) == 0) ? 8 : ((HW & 3) == 0 ? 4 : 1);
    const int blocks = ew_blocks((size_t)Bg * Co * HW / vw);
    for (int gi = 0; gi < groups; ++gi) {      // group 0 writes (or adds to) the gamma/beta gradients, the others add
        const void* xg = off(x, (size_t)gi * Bg * C * HW, x_dtype);
        const void* dg = off(dout, (size_t)gi * Bg * Co * HW, out_dtype);
        const float* mg = mean + (size_t)gi * C;
        const float* ig = invstd + (size_t)gi * C;
        void* dxg = off(dx, (size_t)gi * Bg * C * HW, x_dtype);
        const int acc = gi == 0 ? accumulate : 1;
#define AGAN_P(A, XD, OD) hipLaunchKernelGGL((bn_bwd_partial_kernel<A, XD, OD>), grid, dim3(256), 0, st, xg, dg, mg, ig, gamma, beta, Bg, C, HW, nchunk, part)
#define AGAN_C(XD, OD)                                           \
    if (glu) AGAN_P(AGAN_ACT_GLU, XD, OD);                        \
    else if (act == AGAN_ACT_LRELU) AGAN_P(AGAN_ACT_LRELU, XD, OD); \
    else AGAN_P(AGAN_ACT_NONE, XD, OD);
        AGAN_BN_DT(x_dtype, out_dtype, AGAN_C);
#undef AGAN_C
#undef AGAN_P
        hipLaunchKernelGGL(bn_bwd_final_kernel, dim3(cdiv(C, 256)), dim3(256), 0, st, part, C, Co, nchunk, Bg * HW, glu, dgamma, dbeta, coef, acc);
#define AGAN_L(A, V, XD, OD) hipLaunchKernelGGL((bn_bwd_apply_kernel<A, V, XD, OD>), dim3(blocks), dim3(256), 0, st, xg, dg, mg, ig, gamma, beta, coef, dxg, Bg, C, HW, dx_amax)
#define AGAN_V(A, XD, OD)                                                                                           \
    do {                                                                                                            \
        if (vw == 8) { if constexpr ((XD) != AGAN_DT_F32 || (OD) != AGAN_DT_F32) AGAN_L(A, 8, XD, OD); }             \
        else if (vw == 4) AGAN_L(A, 4, XD, OD);                                                                     \
        else AGAN_L(A, 1, XD, OD);                                                                                  \
    } while (0)
#define AGAN_C(XD, OD)                                                  \
    if (glu) AGAN_V(AGAN_ACT_GLU, XD, OD);                              \
    else if (act == AGAN_ACT_LRELU) AGAN_V(AGAN_ACT_LRELU, XD, OD);     \
    else AGAN_V(AGAN_ACT_NONE, XD, OD);
        AGAN_BN_DT(x_dtype, out_dtype, AGAN_C);
#undef AGAN_C
#undef AGAN_V
#undef AGAN_L
    }
    return check_launch("bn_act_bwd");
}
int agan_bn_act_bwd(const float* x, const float* dout, const float* mean, const float* invstd, const float* gamma,
                    const float* beta, float* dx, float* dgamma, float* dbeta, int B, int C, int HW, int act, int accumulate,
                    int groups, void* ws, size_t ws_bytes, void* stream, float* dx_amax) {
    return agan_bn_act_bwd_dt(x, dout, mean, invstd, gamma, beta, dx, dgamma, dbeta, B, C, HW, act, accumulate, groups, ws, ws_bytes,
                              stream, dx_amax, AGAN_DT_F32, AGAN_DT_F32);
}

int agan_act_fwd(const float* x, float* out, size_t n, int act, void* stream) {
    AGAN_REQUIRE(x && out && n > 0, "act_fwd: bad argument");
    hipStream_t st = as_stream(stream);
    const int blocks = ew_blocks(n / 4 + 1);
    switch (act) {
        case AGAN_ACT_LRELU: hipLaunchKernelGGL((act_fwd_kernel<AGAN_ACT_LRELU>), dim3(blocks), dim3(256), 0, st, x, out, n); break;
        case AGAN_ACT_TANH: hipLaunchKernelGGL((act_fwd_kernel<AGAN_ACT_TANH>), dim3(blocks), dim3(256), 0, st, x, out, n); break;
        case AGAN_ACT_SIGMOID: hipLaunchKernelGGL((act_fwd_kernel<AGAN_ACT_SIGMOID>), dim3(blocks), dim3(256), 0, st, x, out, n); break;
        default: set_error("act_fwd: activation %d", act); return AGAN_EINVAL;
    }
    return check_launch("act_fwd");
}

int agan_act_bwd(const float* y, const float* dout, float* dx, size_t n, int act, void* stream) {
    AGAN_REQUIRE(y && dout && dx && n > 0, "act_bwd: bad argument");
    hipStream_t st = as_stream(stream);
    const int blocks = ew_blocks(n / 4 + 1);
    switch (act) {
        case AGAN_ACT_LRELU: hipLaunchKernelGGL((act_bwd_kernel<AGAN_ACT_LRELU>), dim3(blocks), dim3(256), 0, st, y, dout, dx, n); break;
        case AGAN_ACT_TANH: hipLaunchKernelGGL((act_bwd_kernel<AGAN_ACT_TANH>), dim3(blocks), dim3(256), 0, st, y, dout, dx, n); break;
        case AGAN_ACT_SIGMOID: hipLaunchKernelGGL((act_bwd_kernel<AGAN_ACT_SIGMOID>), dim3(blocks), dim3(256), 0, st, y, dout, dx, n); break;
        default: set_error("act_bwd: activation %d", act); return AGAN_EINVAL;
    }
    return check_launch("act_bwd");
}

int agan_glu_fwd(const float* x, float* out, int B, int C, int HW, void* stream) {
    AGAN_REQUIRE(x && out && B > 0 && C > 0 && HW > 0, "glu_fwd: bad argument");
    AGAN_REQUIRE(C % 2 == 0, "channels dont divide 2!");
    hipLaunchKernelGGL(glu_fwd_kernel, dim3(ew_blocks((size_t)B * (C / 2) * HW)), dim3(256), 0, as_stream(stream), x, out, B, C, HW);
    return check_launch("glu_fwd");
}

int agan_glu_bwd(const float* x, const float* dout, float* dx, int B, int C, int HW, void* stream) {
    AGAN_REQUIRE(x && dout && dx && B > 0 && C > 0 && HW > 0 && C % 2 == 0, "glu_bwd: bad argument");
    hipLaunchKernelGGL(glu_bwd_kernel, dim3(ew_blocks((size_t)B * (C / 2) * HW)), dim3(256), 0, as_stream(stream), x, dout, dx, B, C, HW);
    return check_launch("glu_bwd");
}

size_t agan_bn_train_fwd_ws_bytes(int B, int C, int HW) {
    if (HW > 1 && (long long)B * HW <= kSmallN) return 0;
    return agan_bn_stats_ws_bytes(B, C, HW);
}

int agan_bn_train_fwd_dt(const void* x, const float* gamma, const float* beta, const void* residual, void* out, float* mean,
                         float* invstd, float* running_mean, float* running_var, int64_t* nbt, int B, int C, int HW, float eps,
                         float momentum, int act, int groups, void* ws, size_t ws_bytes, void* stream, float* out_amax, int x_dtype,
                         int out_dtype) {
    AGAN_REQUIRE(x && gamma && beta && out && mean && invstd && B > 0 && C > 0 && HW > 0, "bn_train_fwd: bad argument");
    AGAN_REQUIRE(act == AGAN_ACT_NONE || act == AGAN_ACT_GLU || act == AGAN_ACT_LRELU, "bn_train_fwd: activation %d", act);
    AGAN_REQUIRE(act != AGAN_ACT_GLU || (C % 2 == 0), "channels dont divide 2!");
    AGAN_REQUIRE(!(residual && act != AGAN_ACT_NONE), "bn_train_fwd: residual only with ACT_NONE");
    AGAN_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_train_fwd: running_mean/var must come together");
    AGAN_REQUIRE(groups >= 1 && B % groups == 0, "bn_train_fwd: batch %d does not split into %d groups", B, groups);
    AGAN_REQUIRE(bn_dt_ok(x_dtype, out_dtype), "bn_train_fwd: storage types %d -> %d", x_dtype, out_dtype);
    const int Bg = B / groups;
    const int Co = act == AGAN_ACT_GLU ? C / 2 : C;
    if (HW > 1 && (long long)Bg * HW <= kSmallN) {
        // one launch per group: a workgroup walking both groups itself was measured 2 % slower on the whole step (these launches
        // are latency-bound; two short ones overlap tail to head)
        hipStream_t st = as_stream(stream);
        for (int gi = 0; gi < groups; ++gi) {
            const void* xg = off(x, (size_t)gi * Bg * C * HW, x_dtype);
            const void* rg = residual ? off(residual, (size_t)gi * Bg * Co * HW, out_dtype) : nullptr;
            void* og = off(out, (size_t)gi * Bg * Co * HW, out_dtype);
            float* mg = mean + (size_t)gi * C;
            float* ig = invstd + (size_t)gi * C;
#define AGAN_LP(A, XD, OD, P) hipLaunchKernelGGL((bn_small_fwd_kernel<A, XD, OD, P>), dim3(Co), dim3(256), 0, st, xg, gamma, beta, rg, og, mg, ig, running_mean, running_var, nbt, Bg, C, HW, eps, momentum, out_amax)
#define AGAN_L(A, XD, OD) do { if (Bg * HW <= 512) AGAN_LP(A, XD, OD, 2); else if (Bg * HW <= 2048) AGAN_LP(A, XD, OD, 8); else AGAN_LP(A, XD, OD, kSmallPer); } while (0)
#define AGAN_C(XD, OD)                                           \
    if (act == AGAN_ACT_GLU) AGAN_L(AGAN_ACT_GLU, XD, OD);       \
    else if (act == AGAN_ACT_LRELU) AGAN_L(AGAN_ACT_LRELU, XD, OD); \
    else AGAN_L(AGAN_ACT_NONE, XD, OD);
            AGAN_BN_DT(x_dtype, out_dtype, AGAN_C);
#undef AGAN_C
#undef AGAN_L
#undef AGAN_LP
        }
        return check_launch("bn_train_fwd/small");
    }
    for (int gi = 0; gi < groups; ++gi) {      // same stream: statistics, running-stat update and apply in group order
        const void* xg = off(x, (size_t)gi * Bg * C * HW, x_dtype);
        float* mg = mean + (size_t)gi * C;
        float* ig = invstd + (size_t)gi * C;
        if (int e = agan_bn_stats_dt(xg, Bg, C, HW, eps, mg, ig, running_mean, running_var, nbt, momentum, ws, ws_bytes, stream, x_dtype)) return e;
        if (int e = agan_bn_act_fwd_dt(xg, mg, ig, gamma, beta, residual ? off(residual, (size_t)gi * Bg * Co * HW, out_dtype) : nullptr,
                                       off(out, (size_t)gi * Bg * Co * HW, out_dtype), Bg, C, HW, act, stream, out_amax, x_dtype, out_dtype))
            return e;
    }
    return AGAN_OK;
}
int agan_bn_train_fwd(const float* x, const float* gamma, const float* beta, const float* residual, float* out, float* mean,
                      float* invstd, float* running_mean, float* running_var, int64_t* nbt, int B, int C, int HW, float eps,
                      float momentum, int act, int groups, void* ws, size_t ws_bytes, void* stream, float* out_amax) {
    return agan_bn_train_fwd_dt(x, gamma, beta, residual, out, mean, invstd, running_mean, running_var, nbt, B, C, HW, eps, momentum, act,
                                groups, ws, ws_bytes, stream, out_amax, AGAN_DT_F32, AGAN_DT_F32);
}

}  // extern "C"
