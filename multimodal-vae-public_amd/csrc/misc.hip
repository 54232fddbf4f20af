// misc.hip -- the remaining HBM-bound pieces of the step: stand-alone Swish, Embedding(+Swish)
// gather / scatter, counter-based Philox noise, fused Adam over the flat parameter arena, fill.
#include "common.h"
#include "philox.h"

namespace {

inline int ew_blocks(size_t n, int per_block) {
    size_t b = (n + per_block - 1) / per_block;
    if (b > 8192) b = 8192;          // grid-stride the rest (256 CUs x 32 resident blocks)
    if (b < 1) b = 1;
    return (int)b;
}

__global__ __launch_bounds__(256) void swish_fwd_kernel(const float *x, float *y, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        y[i] = swishf_(x[i]);
}

__global__ __launch_bounds__(256) void sigmoid_fwd_kernel(const float *x, float *y, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        y[i] = sigmoidf_(x[i]);
}

// y[i] = x[i] * scale[i % period] + shift[i % period]
__global__ __launch_bounds__(256) void affine_fwd_kernel(const float *x, const float *scale, const float *shift, float *y,
                                                         size_t n, size_t period) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const size_t j = i % period;
        y[i] = x[i] * scale[j] + shift[j];
    }
}

__global__ __launch_bounds__(256) void swish_bwd_kernel(const float *dy, const float *x, float *dx, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        dx[i] = dy[i] * swish_grad_(x[i]);
}

__device__ __forceinline__ int read_index(const void *idx, int is_float, int r) {
    // is_float = 0: contiguous int64 labels; s > 0: fp32 {0,1} values with element stride s
    // (celeba19 feeds column i of attrs[B,18]: s = 18)
    return is_float ? (int)((const float *)idx)[(size_t)r * is_float] : (int)((const int64_t *)idx)[r];
}

// act[r,:] = swish(w[idx[r],:])   (nn.Embedding + Swish, mnist/model.py:116,123)
// blockIdx.y = group g (celeba19's 18 attribute encoders in one launch): index g reads element
// offset g * idx_gs of the index array, table g * w_gs, output g * act_gs.
__global__ __launch_bounds__(256) void embedding_swish_fwd_kernel(const void *idx, int is_float, const float *w,
                                                                  float *act, int R, int n_classes, int width,
                                                                  size_t idx_gs, size_t w_gs, size_t act_gs) {
    const int g = blockIdx.y;
    idx = is_float ? (const void *)((const float *)idx + g * idx_gs) : (const void *)((const int64_t *)idx + g * idx_gs);
    w += g * w_gs; act += g * act_gs;
    const size_t total = (size_t)R * width;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int r = (int)(i / width), j = (int)(i - (size_t)r * width);
        int c = read_index(idx, is_float, r);
        c = min(max(c, 0), n_classes - 1);
        act[i] = swishf_(w[(size_t)c * width + j]);
    }
}

// dw[c,j] (+)= swish'(w[c,j]) * sum_{r: idx[r]==c} dact[r,j].  Block = 64 columns x 16 row lanes of
// class c; each lane sums its rows in order and the 16 partials are added in a fixed order:
// deterministic (the reference's index_add_ backward is not).
constexpr int EMB_LANES = 16;
__global__ __launch_bounds__(64 * EMB_LANES) void embedding_swish_bwd_kernel(const void *idx, int is_float, const float *w,
                                                                  const float *dact, float *dw, int R,
                                                                  int n_classes, int width, int accumulate,
                                                                  size_t idx_gs, size_t w_gs, size_t dact_gs) {
    __shared__ float part[EMB_LANES][64];
    const int g = blockIdx.z;
    idx = is_float ? (const void *)((const float *)idx + g * idx_gs) : (const void *)((const int64_t *)idx + g * idx_gs);
    w += g * w_gs; dw += g * w_gs; dact += g * dact_gs;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + tx, c = blockIdx.y;
    float s = 0.f;
    if (j < width) {
        const int per = (R + EMB_LANES - 1) / EMB_LANES, r0 = ty * per, r1 = min(R, r0 + per);
        // unconditional loads, eight rows in flight, rows of other classes added as +0 in the same order (the
        // rolled loop with a load under `if (class matches)` waited a memory latency per row: 15 us for 512 rows)
#pragma unroll 8
        for (int r = r0; r < r1; ++r) {
            int cr = read_index(idx, is_float, r);
            cr = min(max(cr, 0), n_classes - 1);
            const float v = dact[(size_t)r * width + j];
            s += (cr == c) ? v : 0.f;
        }
    }
    part[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && j < width) {
        s = 0.f;
#pragma unroll
        for (int q = 0; q < EMB_LANES; q += 4) s += (part[q][tx] + part[q + 1][tx]) + (part[q + 2][tx] + part[q + 3][tx]);
        const size_t o = (size_t)c * width + j;
        s *= swish_grad_(w[o]);
        dw[o] = accumulate ? dw[o] + s : s;
    }
}

// Philox4x32-10 and the draw conventions: philox.h (shared with poe.hip, which draws its own noise)
// mode 0: standard normal (Box-Muller on pairs), mode 1: Bernoulli(keep) in {0,1}
__global__ __launch_bounds__(256) void philox_fill_kernel(float *out, size_t n, uint64_t seed,
                                                          const uint64_t *counter, int mode, float keep,
                                                          uint64_t counter_offset = 0) {
    const uint64_t launch = *counter + counter_offset;
    const size_t groups = (n + 3) / 4;
    for (size_t gidx = (size_t)blockIdx.x * 256 + threadIdx.x; gidx < groups; gidx += (size_t)gridDim.x * 256) {
        uint32_t r[4];
        philox4x32_10(gidx, launch, seed, r);
        float v[4];
        if (mode == 0) {
            box_muller(r[0], r[1], v[0], v[1]);
            box_muller(r[2], r[3], v[2], v[3]);
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = u01(r[q]) < keep ? 1.f : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (gidx * 4 + q < n) out[gidx * 4 + q] = v[q];
    }
}

__global__ void bump_u64_kernel(uint64_t *counter) { *counter += 1; }

// ---- Adam (torch.optim.Adam defaults: no weight decay, no amsgrad) ----
// step t = *step_dev + 1; m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
// p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// ADAM_GROUPS float4 groups per thread and array: all their 16-byte loads in flight before the first is used (the arenas are a few
// tens of MB -- Infinity-Cache resident between steps -- and the launch is the LAST of the step, so its latency is the
// step's: profiles/r05_mnist_by_shape.txt read 23.5 us = 3.1 TB/s for MNIST's 72 MB with one group per thread).
constexpr int ADAM_GROUPS = 2;      // float4 groups per thread and array: 8 x 16-byte loads in flight per thread (4 groups measured equal)
__device__ __forceinline__ void adam_body(float *p, const float *g, float *m, float *v, size_t n, const AdamCoef &c) {
    const size_t n4 = n / 4;
    const bool vec = aligned16_dev(p) && aligned16_dev(g) && aligned16_dev(m) && aligned16_dev(v);
    if (vec) {
        const size_t stride = (size_t)gridDim.x * 256 * ADAM_GROUPS;
        for (size_t i0 = (size_t)blockIdx.x * 256 * ADAM_GROUPS + threadIdx.x; i0 < n4; i0 += stride) {
            float4 pv[ADAM_GROUPS], gv[ADAM_GROUPS], mv[ADAM_GROUPS], vv[ADAM_GROUPS];
            // (a group beyond the end re-reads the thread's first one -- a legal address: the loads stay unconditional)
#pragma unroll
            for (int u = 0; u < ADAM_GROUPS; ++u) {
                const size_t i = i0 + 256 * u, j = i < n4 ? i : i0;
                pv[u] = reinterpret_cast<float4 *>(p)[j]; gv[u] = reinterpret_cast<const float4 *>(g)[j];
                mv[u] = reinterpret_cast<float4 *>(m)[j]; vv[u] = reinterpret_cast<float4 *>(v)[j];
            }
#pragma unroll
            for (int u = 0; u < ADAM_GROUPS; ++u) {
                const size_t i = i0 + 256 * u;
                if (i >= n4) break;
                adam_one(pv[u].x, mv[u].x, vv[u].x, gv[u].x, c); adam_one(pv[u].y, mv[u].y, vv[u].y, gv[u].y, c);
                adam_one(pv[u].z, mv[u].z, vv[u].z, gv[u].z, c); adam_one(pv[u].w, mv[u].w, vv[u].w, gv[u].w, c);
                reinterpret_cast<float4 *>(p)[i] = pv[u];
                reinterpret_cast<float4 *>(m)[i] = mv[u];
                reinterpret_cast<float4 *>(v)[i] = vv[u];
            }
        }
    }
    const size_t tail0 = vec ? n4 * 4 : 0;
    for (size_t i = tail0 + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        adam_one(p[i], m[i], v[i], g[i], c);
}

// The bias corrections need two double-precision pow() (torch computes them in python floats): ~300 fp64 instructions that
// every WAVE of the launch used to run before touching memory.  Now thread 0 of a block computes them (the other waves
// skip the branch) and the block reads them from LDS; mvae_adam_apply_coef below takes them ready-made.
__global__ __launch_bounds__(256) void adam_kernel(float *p, const float *g, float *m, float *v, size_t n,
                                                   double lr, double b1d, double b2d, double epsd, float gscale,
                                                   const int64_t *step_dev, int64_t step_add) {
    __shared__ float bc[2];
    if (threadIdx.x == 0) adam_bias_corrections(lr, b1d, b2d, (double)(*step_dev + step_add), &bc[0], &bc[1]);
    __syncthreads();
    AdamCoef c = adam_coef(b1d, b2d, epsd, gscale);
    c.step_size = bc[0]; c.inv_sqrt_bc2 = bc[1];
    adam_body(p, g, m, v, n, c);
}

// ... with the step's two factors already in memory (mvae_adam_prepare, launched early in the step off the critical chain)
__global__ __launch_bounds__(256) void adam_coef_kernel(float *p, const float *g, float *m, float *v, size_t n,
                                                        const float *coef2, double b1d, double b2d, double epsd, float gscale) {
    AdamCoef c = adam_coef(b1d, b2d, epsd, gscale);
    c.step_size = coef2[0]; c.inv_sqrt_bc2 = coef2[1];
    adam_body(p, g, m, v, n, c);
}

// The step counter advanced by `delta` and the two bias-correction factors of step t = the NEW counter value left in
// coef[0..1] (step_size, inv_sqrt_bc2): what a weight-gradient launch that applies Adam to its own outputs reads
// (mvae_linear_wgrad_batched_adam).  One thread, early in the step, off the critical chain.
__global__ void adam_prepare_kernel(int64_t *step, int64_t delta, double lr, double b1d, double b2d, float *coef) {
    *step += delta;
    adam_bias_corrections(lr, b1d, b2d, (double)*step, coef, coef + 1);
}

__global__ void bump_i64_kernel(int64_t *step) { *step += 1; }
__global__ void add_i64_kernel(int64_t *counter, int64_t delta) { *counter += delta; }

__global__ __launch_bounds__(256) void fill_kernel(float *out, size_t n, float value) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = value;
}

}  // namespace

MVAE_EXPORT int mvae_swish_fwd(const float *x, float *y, size_t n, mvae_stream_t stream) {
    if (!x || !y) return MVAE_ERR_ARG;
    if (n == 0) return MVAE_OK;
    hipLaunchKernelGGL(swish_fwd_kernel, dim3(ew_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, x, y, n);
    return mvae_launch_status();
}

MVAE_EXPORT int mvae_sigmoid_fwd(const float *x, float *y, size_t n, mvae_stream_t stream) {
    if (!x || !y) return MVAE_ERR_ARG;
    if (n == 0) return MVAE_OK;
    hipLaunchKernelGGL(sigmoid_fwd_kernel, dim3(ew_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, x, y, n);
    return mvae_launch_status();
}

MVAE_EXPORT int mvae_affine_fwd(const float *x, const float *scale, const float *shift, float *y, size_t n,
                                size_t period, mvae_stream_t stream) {
    if (!x || !scale || !shift || !y || period == 0) return MVAE_ERR_ARG;
    if (n == 0) return MVAE_OK;
    hipLaunchKernelGGL(affine_fwd_kernel, dim3(ew_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, x, scale, shift, y,
                       n, period);
    return mvae_launch_status();
}

MVAE_EXPORT int mvae_swish_bwd(const float *dy, const float *x, float *dx, size_t n, mvae_stream_t stream) {
    if (!dy || !x || !dx) return MVAE_ERR_ARG;
    if (n == 0) return MVAE_OK;
    hipLaunchKernelGGL(swish_bwd_kernel, dim3(ew_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, dy, x, dx, n);
    return mvae_launch_status();
}

static int embedding_fwd_launch(const void *idx, int idx_is_float, const float *w, float *act, int R, int n_classes,
                                int width, int G, size_t idx_gs, size_t w_gs, size_t act_gs, hipStream_t st) {
    unsigned blocks = ew_blocks((size_t)R * width, 256);
    if (G > 1 && blocks > 256) blocks = 256;
    hipLaunchKernelGGL(embedding_swish_fwd_kernel, dim3(blocks, G), dim3(256), 0, st, idx, idx_is_float, w, act, R,
                       n_classes, width, idx_gs, w_gs, act_gs);
    return mvae_launch_status();
}

MVAE_EXPORT int mvae_embedding_swish_fwd(const void *idx, int idx_is_float, const float *w, float *act, int R,
                                         int n_classes, int width, mvae_stream_t stream) {
    if (!idx || !w || !act || R <= 0 || n_classes <= 0 || width <= 0) return MVAE_ERR_ARG;
    return embedding_fwd_launch(idx, idx_is_float, w, act, R, n_classes, width, 1, 0, 0, 0, (hipStream_t)stream);
}

MVAE_EXPORT int mvae_embedding_swish_fwd_grouped(const void *idx, int idx_is_float, size_t idx_gs, const float *w,
                                                 size_t w_gs, float *act, size_t act_gs, int G, int R,
                                                 int n_classes, int width, mvae_stream_t stream) {
    if (!idx || !w || !act || G <= 0 || G > 65535 || R <= 0 || n_classes <= 0 || width <= 0) return MVAE_ERR_ARG;
    return embedding_fwd_launch(idx, idx_is_float, w, act, R, n_classes, width, G, idx_gs, w_gs, act_gs,
                                (hipStream_t)stream);
}

MVAE_EXPORT int mvae_embedding_swish_bwd(const void *idx, int idx_is_float, const float *w, const float *dact,
                                         float *dw, int R, int n_classes, int width, int flags,
                                         mvae_stream_t stream) {
    if (!idx || !w || !dact || !dw || R <= 0 || n_classes <= 0 || width <= 0) return MVAE_ERR_ARG;
    hipLaunchKernelGGL(embedding_swish_bwd_kernel, dim3((width + 63) / 64, n_classes, 1), dim3(64 * EMB_LANES), 0,
                       (hipStream_t)stream, idx, idx_is_float, w, dact, dw, R, n_classes, width,
                       (flags & MVAE_ACCUMULATE) ? 1 : 0, (size_t)0, (size_t)0, (size_t)0);
    return mvae_launch_status();
}

MVAE_EXPORT int mvae_embedding_swish_bwd_grouped(const void *idx, int idx_is_float, size_t idx_gs, const float *w,
                                                 size_t w_gs, const float *dact, size_t dact_gs, float *dw, int G,
                                                 int R, int n_classes, int width, int flags, mvae_stream_t stream) {
    if (!idx || !w || !dact || !dw || G <= 0 || G > 65535 || R <= 0 || n_classes <= 0 || width <= 0)
        return MVAE_ERR_ARG;
    hipLaunchKernelGGL(embedding_swish_bwd_kernel, dim3((width + 63) / 64, n_classes, G), dim3(64 * EMB_LANES), 0,
                       (hipStream_t)stream, idx, idx_is_float, w, dact, dw, R, n_classes, width,
                       (flags & MVAE_ACCUMULATE) ? 1 : 0, idx_gs, w_gs, dact_gs);
    return mvae_launch_status();
}

static int philox_launch(float *out, size_t n, uint64_t seed, uint64_t *counter_dev, int mode, float keep,
                         mvae_stream_t stream) {
    if (!out || !counter_dev) return MVAE_ERR_ARG;
    if (n == 0) return MVAE_OK;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(philox_fill_kernel, dim3(ew_blocks((n + 3) / 4, 256)), dim3(256), 0, st, out, n, seed,
                       (const uint64_t *)counter_dev, mode, keep, (uint64_t)0);
    hipLaunchKernelGGL(bump_u64_kernel, dim3(1), dim3(1), 0, st, counter_dev);
    return mvae_launch_status();
}

// The same draws without the counter bump: launch index = *counter_dev + counter_offset.  A fused step draws
// its noise tensors at offsets 0, 1, ... and advances the counter once (mvae_elbo_reduce), so a step costs
// no single-thread bump launches.
MVAE_EXPORT int mvae_philox_fill(float *out, size_t n, int bernoulli, float keep_prob, uint64_t seed,
                                 const uint64_t *counter_dev, uint64_t counter_offset, mvae_stream_t stream) {
    if (!out || !counter_dev) return MVAE_ERR_ARG;
    if (n == 0) return MVAE_OK;
    hipLaunchKernelGGL(philox_fill_kernel, dim3(ew_blocks((n + 3) / 4, 256)), dim3(256), 0, (hipStream_t)stream, out, n,
                       seed, counter_dev, bernoulli ? 1 : 0, keep_prob, counter_offset);
    return mvae_launch_status();
}

MVAE_EXPORT int mvae_randn(float *out, size_t n, uint64_t seed, uint64_t *counter_dev, mvae_stream_t stream) {
    return philox_launch(out, n, seed, counter_dev, 0, 0.f, stream);
}

MVAE_EXPORT int mvae_bernoulli(float *out, size_t n, float keep_prob, uint64_t seed, uint64_t *counter_dev,
                               mvae_stream_t stream) {
    return philox_launch(out, n, seed, counter_dev, 1, keep_prob, stream);
}

MVAE_EXPORT int mvae_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, size_t n,
                               double lr, double beta1, double beta2, double eps, float grad_scale,
                               int64_t *step_dev, mvae_stream_t stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || !step_dev) return MVAE_ERR_ARG;
    if (n == 0) return MVAE_OK;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(adam_kernel, dim3(ew_blocks(n / (4 * ADAM_GROUPS) + 1, 256)), dim3(256), 0, st, param, grad, exp_avg,
                       exp_avg_sq, n, lr, beta1, beta2, eps, grad_scale, (const int64_t *)step_dev, (int64_t)1);
    hipLaunchKernelGGL(bump_i64_kernel, dim3(1), dim3(1), 0, st, step_dev);
    return mvae_launch_status();
}

// Adam over a RANGE of the arena without advancing the step counter: data-parallel replicas update bucket
// k as soon as its all-reduce has landed (t = *step_dev + 1 for every range of the step), then advance the
// counter once with mvae_counter_add.
MVAE_EXPORT int mvae_adam_apply(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, size_t n,
                                double lr, double beta1, double beta2, double eps, float grad_scale,
                                const int64_t *step_dev, mvae_stream_t stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || !step_dev) return MVAE_ERR_ARG;
    if (n == 0) return MVAE_OK;
    hipLaunchKernelGGL(adam_kernel, dim3(ew_blocks(n / (4 * ADAM_GROUPS) + 1, 256)), dim3(256), 0, (hipStream_t)stream, param, grad,
                       exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, grad_scale, step_dev, (int64_t)1);
    return mvae_launch_status();
}

// The same at step t = *step_dev + step_add: a caller that advanced the counter earlier in the step (off the
// critical chain) passes step_add = 0 and needs no counter launch behind the update.
MVAE_EXPORT int mvae_adam_apply_at(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, size_t n,
                                   double lr, double beta1, double beta2, double eps, float grad_scale,
                                   const int64_t *step_dev, int64_t step_add, mvae_stream_t stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || !step_dev) return MVAE_ERR_ARG;
    if (n == 0) return MVAE_OK;
    hipLaunchKernelGGL(adam_kernel, dim3(ew_blocks(n / (4 * ADAM_GROUPS) + 1, 256)), dim3(256), 0, (hipStream_t)stream, param, grad,
                       exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, grad_scale, step_dev, step_add);
    return mvae_launch_status();
}

// The same with the two bias-correction factors of the step read from `coef2` (left there by mvae_adam_prepare).
MVAE_EXPORT int mvae_adam_apply_coef(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, size_t n,
                                     const float *coef2, double beta1, double beta2, double eps, float grad_scale,
                                     mvae_stream_t stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || !coef2) return MVAE_ERR_ARG;
    if (n == 0) return MVAE_OK;
    hipLaunchKernelGGL(adam_coef_kernel, dim3(ew_blocks(n / (4 * ADAM_GROUPS) + 1, 256)), dim3(256), 0, (hipStream_t)stream, param, grad,
                       exp_avg, exp_avg_sq, n, coef2, beta1, beta2, eps, grad_scale);
    return mvae_launch_status();
}

// An empty kernel a profiling host puts between two calls: in a rocprofv3 kernel trace of ONE stream the k-th
// `trace_marker_kernel` dispatch separates the kernels of call k - 1 from those of call k, which is how
// tools/step_by_shape.py attributes dispatches (GEMM + finish launch + ...) to the launcher call and its shape.
__global__ void trace_marker_kernel(int tag) { (void)tag; }

MVAE_EXPORT int mvae_trace_marker(int tag, mvae_stream_t stream) {
    hipLaunchKernelGGL(trace_marker_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, tag);
    return mvae_launch_status();
}

MVAE_EXPORT int mvae_adam_prepare(int64_t *step_dev, int64_t delta, double lr, double beta1, double beta2,
                                  float *coef2, mvae_stream_t stream) {
    if (!step_dev || !coef2) return MVAE_ERR_ARG;
    hipLaunchKernelGGL(adam_prepare_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step_dev, delta, lr, beta1, beta2,
                       coef2);
    return mvae_launch_status();
}

MVAE_EXPORT int mvae_counter_add(int64_t *counter_dev, int64_t delta, mvae_stream_t stream) {
    if (!counter_dev) return MVAE_ERR_ARG;
    hipLaunchKernelGGL(add_i64_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, counter_dev, delta);
    return mvae_launch_status();
}

// One launch that brings a step's inputs to the fixed addresses a captured graph reads: the image batch (float4
// copy), the label batch (32-bit words) and the per-step table block (loss coefficients, PoE masks, ...), whose
// source is PINNED HOST memory read by the kernel itself (zero-copy over the host link: a few dozen words).  Three
// runtime copy commands in front of every graph replay were ~15 us of a 325-us MNIST step.
__global__ __launch_bounds__(256) void ingest_kernel(const float4 *img_src, float4 *img_dst, size_t n4,
                                                     const uint32_t *lbl_src, uint32_t *lbl_dst, size_t lbl_words,
                                                     const uint32_t *tbl_src, uint32_t *tbl_dst, size_t tbl_words) {
    const size_t total = n4 + lbl_words + tbl_words;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        if (i < n4) img_dst[i] = img_src[i];
        else if (i < n4 + lbl_words) lbl_dst[i - n4] = lbl_src[i - n4];
        else tbl_dst[i - n4 - lbl_words] = tbl_src[i - n4 - lbl_words];
    }
}

MVAE_EXPORT int mvae_ingest(const float *image_src, float *image_dst, size_t image_floats, const void *label_src,
                            void *label_dst, size_t label_bytes, const void *table_src_host, void *table_dst,
                            size_t table_bytes, mvae_stream_t stream) {
    if ((image_floats && (!image_src || !image_dst)) || (label_bytes && (!label_src || !label_dst)) ||
        (table_bytes && (!table_src_host || !table_dst)))
        return MVAE_ERR_ARG;
    if (image_floats % 4 || label_bytes % 4 || table_bytes % 4 || !aligned16(image_src) || !aligned16(image_dst))
        return MVAE_ERR_ARG;
    const size_t total = image_floats / 4 + label_bytes / 4 + table_bytes / 4;
    if (total == 0) return MVAE_OK;
    hipLaunchKernelGGL(ingest_kernel, dim3(ew_blocks(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float4 *)image_src, (float4 *)image_dst, image_floats / 4, (const uint32_t *)label_src,
                       (uint32_t *)label_dst, label_bytes / 4, (const uint32_t *)table_src_host, (uint32_t *)table_dst,
                       table_bytes / 4);
    return mvae_launch_status();
}

MVAE_EXPORT int mvae_fill(float *out, size_t n, float value, mvae_stream_t stream) {
    if (!out) return MVAE_ERR_ARG;
    if (n == 0) return MVAE_OK;
    hipLaunchKernelGGL(fill_kernel, dim3(ew_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, out, n, value);
    return mvae_launch_status();
}

// ---- Dropout fan-out: ONE activation h[B,N] feeding G calls that differ only in their keep-mask
// (celeba/model.py:89-92 is run twice per step on the same batch: the conv trunk and the
// Linear(6400,512)+Swish are identical, only the Dropout(0.1) draw differs) ----
namespace {
__global__ __launch_bounds__(256) void dropout_fanout_kernel(const float *h, const float *masks, float *out,
                                                             float scale, int G, size_t bn) {
    const size_t total = (size_t)G * bn;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256)
        out[i] = h[i % bn] * (masks[i] * scale);
}
__global__ __launch_bounds__(256) void dropout_fanin_kernel(const float *dout, const float *masks, float *dh,
                                                            float scale, int G, size_t bn) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < bn; i += (size_t)gridDim.x * 256) {
        float s = 0.f;
        for (int g = 0; g < G; ++g) s += dout[(size_t)g * bn + i] * (masks[(size_t)g * bn + i] * scale);
        dh[i] = s;
    }
}
// elementwise BCE-with-logits for the reference-surface helper (mnist/train.py:62-74)
__global__ __launch_bounds__(256) void bce_elem_fwd_kernel(const float *x, const float *t, float *out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        out[i] = fmaxf(x[i], 0.f) - x[i] * t[i] + logf(1.0f + expf(-fabsf(x[i])));
}
__global__ __launch_bounds__(256) void bce_elem_bwd_kernel(const float *x, const float *t, const float *g,
                                                           float *dx, float *dt, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float xv = x[i];
        const float e = expf(-fabsf(xv));
        const float sgn = (xv > 0.f) ? 1.f : ((xv < 0.f) ? -1.f : 0.f);
        if (dx) dx[i] = g[i] * (((xv >= 0.f) ? 1.f : 0.f) - t[i] - sgn * (e / (1.0f + e)));
        if (dt) dt[i] = -g[i] * xv;
    }
}
}  // namespace

MVAE_EXPORT int mvae_dropout_fanout_fwd(const float *h, const float *masks, float *out, float scale, int G,
                                        int B, int N, mvae_stream_t stream) {
    if (!h || !masks || !out || G <= 0 || B <= 0 || N <= 0) return MVAE_ERR_ARG;
    const size_t bn = (size_t)B * N;
    hipLaunchKernelGGL(dropout_fanout_kernel, dim3(ew_blocks(bn * G, 256)), dim3(256), 0, (hipStream_t)stream, h,
                       masks, out, scale, G, bn);
    return mvae_launch_status();
}

MVAE_EXPORT int mvae_dropout_fanin_bwd(const float *dout, const float *masks, float *dh, float scale, int G,
                                       int B, int N, mvae_stream_t stream) {
    if (!dout || !masks || !dh || G <= 0 || B <= 0 || N <= 0) return MVAE_ERR_ARG;
    const size_t bn = (size_t)B * N;
    hipLaunchKernelGGL(dropout_fanin_kernel, dim3(ew_blocks(bn, 256)), dim3(256), 0, (hipStream_t)stream, dout,
                       masks, dh, scale, G, bn);
    return mvae_launch_status();
}

MVAE_EXPORT int mvae_bce_elem_fwd(const float *logits, const float *target, float *out, size_t n,
                                  mvae_stream_t stream) {
    if (!logits || !target || !out) return MVAE_ERR_ARG;
    if (n == 0) return MVAE_OK;
    hipLaunchKernelGGL(bce_elem_fwd_kernel, dim3(ew_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, logits,
                       target, out, n);
    return mvae_launch_status();
}

MVAE_EXPORT int mvae_bce_elem_bwd(const float *logits, const float *target, const float *g, float *dlogits,
                                  float *dtarget, size_t n, mvae_stream_t stream) {
    if (!logits || !target || !g || (!dlogits && !dtarget)) return MVAE_ERR_ARG;
    if (n == 0) return MVAE_OK;
    hipLaunchKernelGGL(bce_elem_bwd_kernel, dim3(ew_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, logits,
                       target, g, dlogits, dtarget, n);
    return mvae_launch_status();
}
