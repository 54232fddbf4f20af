// Shared device helpers for libmvae_hip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/mvae_hip.h"

#define MVAE_EXPORT extern "C" __attribute__((visibility("default")))

static inline int mvae_launch_status() {
    return hipGetLastError() == hipSuccess ? MVAE_OK : MVAE_ERR_LAUNCH;
}

static inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline bool aligned8(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 7u) == 0; }
__device__ __forceinline__ bool aligned16_dev(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// sigmoid / swish as the reference composes them: x * (1 / (1 + exp(-x))) (mnist/model.py:166-169),
// on the hardware transcendental units: v_exp_f32 (2^t, 1 ulp) and v_rcp_f32 (1 ulp), ~3e-7 relative
// against the 1e-4 the step is graded at.  libm's expf + an IEEE divide are ~25 VALU instructions per
// value; with 16 values per thread that was 1.7 us of every GEMM epilogue that applies Swish or Swish'
// (tools/kstep_probe.py: 7.2 vs 5.5 us for a one-tile launch with / without the activation output).
__device__ __forceinline__ float sigmoidf_(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504088896340736f));
}
__device__ __forceinline__ float swishf_(float x) { return x * sigmoidf_(x); }
// d/dx [x s(x)] = s + x s (1 - s)
__device__ __forceinline__ float swish_grad_(float x) {
    float s = sigmoidf_(x);
    return s * (1.0f + x * (1.0f - s));
}

// 64-lane wavefront sum (CDNA wave = 64).
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Sum over the 32 lanes of a half wavefront (lanes 0-31 and 32-63 separately); every lane receives its half's total.
__device__ __forceinline__ float half_wave_sum(float v) {
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__device__ __forceinline__ float half_wave_max(float v) {
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// Bernoulli reconstruction term on a logit: clamp(x,0) - x*t + log(1 + exp(-|x|))   (mnist/train.py:73-74)
// exp / log through the hardware's base-2 instructions (v_exp_f32 / v_log_f32, ~1 ulp) like the sigmoid above: libm's expf +
// logf + an IEEE divide were ~85 vector instructions per element for value + gradient -- 13.6 us of pure ALU time in the
// 25.6-us image term of CelebA (512 x 12288 logits, profiles/r04_celeba_by_shape.txt), and 1.7 us inside the epilogue of
// MNIST's loss-carrying last Linear, which sits on the step's critical chain.  The argument of the exp is <= 0 and that of
// the log in (1, 2]: no range handling is needed.  Relative error of a term ~1e-7, against the 1e-4 the step is graded at.
__device__ __forceinline__ float bce_exp_(float x) { return __builtin_amdgcn_exp2f(fabsf(x) * -1.44269504088896340736f); }
__device__ __forceinline__ float bce_elem(float x, float t) {
    return fmaxf(x, 0.f) - x * t + __builtin_amdgcn_logf(1.0f + bce_exp_(x)) * 0.69314718055994530942f;
}
// autograd of the expression above, term by term: 1[x>=0] - t - sign(x) * e/(1+e), e = exp(-|x|)
// (at x == 0 the reference's sub-gradient: 1 - t, SURVEY Appendix B-3 -- the comparisons below keep it)
__device__ __forceinline__ float bce_grad(float x, float t) {
    const float e = bce_exp_(x);
    const float sgn = (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f);
    return ((x >= 0.f) ? 1.f : 0.f) - t - sgn * (e * __builtin_amdgcn_rcpf(1.0f + e));
}

// One element of torch.optim.Adam (defaults: no weight decay, no amsgrad; mnist/train.py:168,219), shared by the
// arena-wide launch (misc.hip) and by the weight-gradient kernels that update their own outputs (linear_direct.h),
// so that both round identically:  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;
// p -= step_size * m / (sqrt(v) * inv_sqrt_bc2 + eps),  step_size = lr / (1 - b1^t),  inv_sqrt_bc2 = 1 / sqrt(1 - b2^t)
struct AdamCoef { float b1, b2, omb1, omb2, eps, gscale, step_size, inv_sqrt_bc2; };
__device__ __forceinline__ void adam_one(float &p, float &m, float &v, float g, const AdamCoef &c) {
#pragma clang fp contract(off)      // only the fused multiply-adds written here, wherever this is inlined
    const float gg = g * c.gscale;
    m = __builtin_fmaf(c.b1, m, c.omb1 * gg);
    v = __builtin_fmaf(c.b2, v, c.omb2 * gg * gg);
    p -= c.step_size * (m / __builtin_fmaf(sqrtf(v), c.inv_sqrt_bc2, c.eps));
}
// hyper-parameters arrive as doubles and are rounded the way torch rounds python floats into fp32 tensor ops:
// beta and (1 - beta) separately (1 - 0.999 != 1 - float(0.999))
__device__ __forceinline__ AdamCoef adam_coef(double b1d, double b2d, double epsd, float gscale) {
    AdamCoef c;
    c.b1 = (float)b1d; c.b2 = (float)b2d; c.eps = (float)epsd; c.gscale = gscale;
    c.omb1 = (float)(1.0 - b1d); c.omb2 = (float)(1.0 - b2d);
    c.step_size = 0.f; c.inv_sqrt_bc2 = 0.f;
    return c;
}
__device__ __forceinline__ void adam_bias_corrections(double lr, double b1d, double b2d, double t, float *step_size,
                                                      float *inv_sqrt_bc2) {
    *step_size = (float)(lr / (1.0 - pow(b1d, t)));
    *inv_sqrt_bc2 = (float)(1.0 / sqrt(1.0 - pow(b2d, t)));
}

// Block-wide sum for blocks of up to 1024 threads; `red` is >= 16 floats of LDS.
// Every thread receives the total.
__device__ __forceinline__ float block_sum(float v, float *red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}
