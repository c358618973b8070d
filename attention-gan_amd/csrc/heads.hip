// Small fused heads of the training step: GAN / KL losses with their gradients, CA-net reparametrisation, fused Adam,
// plus the library's version / error plumbing.
#include "agan_common.h"

#include <algorithm>
#include <stdarg.h>

using namespace agan;

namespace agan {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace agan

namespace {

// -mean(log(pr + 1e-8) + log(1 - pf + 1e-8))      losses/disc_loss.py:55-61  (pr = D(x), pf = D(G(z)) are probabilities)
__global__ __launch_bounds__(256) void disc_loss_kernel(const float* __restrict__ pr, const float* __restrict__ pf,
                                                        float* __restrict__ loss, float* __restrict__ dpr, float* __restrict__ dpf, int B) {
    __shared__ float red[4];
    float s = 0.f;
    const float inv = 1.f / B;
    for (int i = threadIdx.x; i < B; i += 256) {
        const float a = pr[i] + 1e-8f, b = 1.f - pf[i] + 1e-8f;
        s += logf(a) + logf(b);
        if (dpr) dpr[i] = -inv / a;
        if (dpf) dpf[i] = inv / b;
    }
    s = block_sum<256>(s, red);
    if (threadIdx.x == 0) loss[0] = -s * inv;
}

// -mean(log(pf + 1e-8))      losses/gen_loss.py:45-46
__global__ __launch_bounds__(256) void gen_loss_kernel(const float* __restrict__ pf, float* __restrict__ loss, float* __restrict__ dpf, int B) {
    __shared__ float red[4];
    float s = 0.f;
    const float inv = 1.f / B;
    for (int i = threadIdx.x; i < B; i += 256) {
        const float a = pf[i] + 1e-8f;
        s += logf(a);
        if (dpf) dpf[i] = -inv / a;
    }
    s = block_sum<256>(s, red);
    if (threadIdx.x == 0) loss[0] = -s * inv;
}

// -0.5 * mean(1 + lv - mu^2 - exp(lv))      losses/KL_loss.py:5-9
__global__ __launch_bounds__(256) void kl_loss_kernel(const float* __restrict__ mu, const float* __restrict__ lv, float* __restrict__ loss,
                                                      float* __restrict__ dmu, float* __restrict__ dlv, int n) {
    __shared__ float red[4];
    float s = 0.f;
    const float inv = 1.f / n;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float m = mu[i], l = lv[i], e = expf(l);
        s += 1.f + l - m * m - e;
        if (dmu) dmu[i] = m * inv;
        if (dlv) dlv[i] = -0.5f * (1.f - e) * inv;
    }
    s = block_sum<256>(s, red);
    if (threadIdx.x == 0) loss[0] = -0.5f * s * inv;
}

// c = eps * exp(0.5*lv) + mu      networks/generator_submodules.py:161-165
__global__ __launch_bounds__(256) void reparam_fwd_kernel(const float* __restrict__ mu, const float* __restrict__ lv,
                                                          const float* __restrict__ eps, float* __restrict__ c, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) c[i] = eps[i] * expf(0.5f * lv[i]) + mu[i];
}
__global__ __launch_bounds__(256) void reparam_bwd_kernel(const float* __restrict__ lv, const float* __restrict__ eps,
                                                          const float* __restrict__ dc, float* __restrict__ dmu, float* __restrict__ dlv, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        dmu[i] = dc[i];
        dlv[i] = dc[i] * eps[i] * 0.5f * expf(0.5f * lv[i]);
    }
}

// torch.optim.Adam single-tensor update, written for one flat buffer (train.py:78-80)
// step counter and bias corrections live in device memory so that a captured HIP graph advances them on every replay:
// state[0] = step (as float bits of an int), coef[0] = lr / (1 - beta1^t), coef[1] = 1 / sqrt(1 - beta2^t)
__global__ void adam_tick_kernel(int* __restrict__ step, float* __restrict__ coef, double lr, double b1, double b2) {
    const int t = *step + 1;
    *step = t;
    // beta^t by squaring (double): the libm pow() sequence took 28 us on this single lane, 4 x per step
    double p1 = 1.0, p2 = 1.0, q1 = b1, q2 = b2;
    for (int e = t; e > 0; e >>= 1) {
        if (e & 1) { p1 *= q1; p2 *= q2; }
        q1 *= q1; q2 *= q2;
    }
    const double bc1 = 1.0 - p1, bc2 = 1.0 - p2;
    coef[0] = (float)(lr / bc1);
    coef[1] = (float)(1.0 / sqrt(bc2));
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, size_t n, const float* __restrict__ coef, float b1, float b2,
                                                   float omb1, float omb2, float eps, float gscale) {
    const float lr_c = coef[0], inv_sqrt_bc2 = coef[1];
    const size_t n4 = n / 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 pv = reinterpret_cast<float4*>(p)[i], mv = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
        const float4 gv = reinterpret_cast<const float4*>(g)[i];
        float* pp = &pv.x; float* mp = &mv.x; float* vp = &vv.x; const float* gp = &gv.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float gg = gp[k] * gscale;
            mp[k] = b1 * mp[k] + omb1 * gg;
            vp[k] = b2 * vp[k] + omb2 * gg * gg;
            pp[k] -= lr_c * mp[k] / (sqrtf(vp[k]) * inv_sqrt_bc2 + eps);
        }
        reinterpret_cast<float4*>(p)[i] = pv;
        reinterpret_cast<float4*>(m)[i] = mv;
        reinterpret_cast<float4*>(v)[i] = vv;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const size_t e = n4 * 4 + threadIdx.x;
        const float gg = g[e] * gscale;
        m[e] = b1 * m[e] + omb1 * gg;
        v[e] = b2 * v[e] + omb2 * gg * gg;
        p[e] -= lr_c * m[e] / (sqrtf(v[e]) * inv_sqrt_bc2 + eps);
    }
}

}  // namespace

extern "C" {

int agan_version(void) { return AGAN_VERSION; }
const char* agan_last_error(void) { return agan::g_err; }

int agan_disc_loss(const float* p_real, const float* p_fake, float* loss, float* dp_real, float* dp_fake, int B, void* stream) {
    AGAN_REQUIRE(p_real && p_fake && loss && B > 0, "disc_loss: bad argument");
    hipLaunchKernelGGL(disc_loss_kernel, dim3(1), dim3(256), 0, as_stream(stream), p_real, p_fake, loss, dp_real, dp_fake, B);
    return check_launch("disc_loss");
}

int agan_gen_loss(const float* p_fake, float* loss, float* dp_fake, int B, void* stream) {
    AGAN_REQUIRE(p_fake && loss && B > 0, "gen_loss: bad argument");
    hipLaunchKernelGGL(gen_loss_kernel, dim3(1), dim3(256), 0, as_stream(stream), p_fake, loss, dp_fake, B);
    return check_launch("gen_loss");
}

int agan_kl_loss(const float* mu, const float* logvar, float* loss, float* dmu, float* dlogvar, int n, void* stream) {
    AGAN_REQUIRE(mu && logvar && loss && n > 0, "kl_loss: bad argument");
    hipLaunchKernelGGL(kl_loss_kernel, dim3(1), dim3(256), 0, as_stream(stream), mu, logvar, loss, dmu, dlogvar, n);
    return check_launch("kl_loss");
}

int agan_reparam_fwd(const float* mu, const float* logvar, const float* eps, float* c, int n, void* stream) {
    AGAN_REQUIRE(mu && logvar && eps && c && n > 0, "reparam_fwd: bad argument");
    hipLaunchKernelGGL(reparam_fwd_kernel, dim3(cdiv(n, 256)), dim3(256), 0, as_stream(stream), mu, logvar, eps, c, n);
    return check_launch("reparam_fwd");
}

int agan_reparam_bwd(const float* logvar, const float* eps, const float* dc, float* dmu, float* dlogvar, int n, void* stream) {
    AGAN_REQUIRE(logvar && eps && dc && dmu && dlogvar && n > 0, "reparam_bwd: bad argument");
    hipLaunchKernelGGL(reparam_bwd_kernel, dim3(cdiv(n, 256)), dim3(256), 0, as_stream(stream), logvar, eps, dc, dmu, dlogvar, n);
    return check_launch("reparam_bwd");
}

int agan_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, int32_t* step_state, double lr,
                   double beta1, double beta2, double eps, float grad_scale, void* stream) {
    AGAN_REQUIRE(param && grad && exp_avg && exp_avg_sq && step_state && n > 0, "adam_step: bad argument");
    // step_state: 4 x 4 bytes of device memory owned by the optimiser: [0] int step count (incremented here), [2],[3] float
    // bias-correction coefficients.  Scalars are rounded to fp32 exactly where torch.optim.Adam rounds them.
    hipStream_t st = as_stream(stream);
    float* coef = reinterpret_cast<float*>(step_state + 2);
    hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, st, step_state, coef, lr, beta1, beta2);
    const int blocks = (int)std::max<size_t>(1, std::min<size_t>(cdivz(n / 4 + 1, 256), 2048));
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, st, param, grad, exp_avg, exp_avg_sq, n, coef, (float)beta1,
                       (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps, grad_scale);
    return check_launch("adam_step");
}

}  // extern "C"
