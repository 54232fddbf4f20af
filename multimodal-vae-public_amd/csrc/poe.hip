// poe.hip -- prior expert + product-of-experts fuse + reparameterised draw + analytic KL,
// forward and backward, for all the ELBO terms of a train step in one launch each.
//
// Reference: prior_expert mnist/model.py:172-185; stacking MVAE.infer mnist/model.py:46-64,
// celeba19/model.py:63-89; ProductOfExperts mnist/model.py:156-163 (variant A) and
// celeba/model.py:200-207 (variant B); reparametrize mnist/model.py:29-35; KL mnist/train.py:56.
//
// HBM-bound and tiny: one wave per batch row (lanes along the latent dim, so the KL row sum is a
// wavefront reduction), every term t of the step handled by the same thread so each expert's
// mu/logvar is fetched once from L1/L2.  The [M,B,D] expert stack of the reference (torch.cat
// per expert) is never materialised; the N(0,1) prior is a constant.
#include "common.h"
#include "philox.h"

namespace {

constexpr int POE_THREADS = 128;   // 2 waves = 2 batch rows per block
constexpr int POE_MAX_TERMS = 40;   // 3 * T * 128 floats of LDS in the backward <= 60 KiB
constexpr float POE_EPS = 1e-8f;

struct PoeArgs {
    mvae_experts_t ex;
    int ld, E, T, B, D, variant;      // variant: MVAE_POE_VARIANT_A / _B (the NO_PRIOR bit is split off into no_prior)
    int no_prior;
    // many-term steps (celeba19: 21 terms x 21 experts): blockIdx.y owns `chunk` consecutive TERMS of the forward /
    // EXPERTS of the backward, so the launch is (rows / 2) x (T / chunk) blocks instead of one wave per batch row
    // walking all of them in sequence (62 / 106 us for 2.5 MB of tensors: a latency chain, one wave per CU)
    int chunk;
    // draw mode (mvae_poe_fwd_draw): eps is generated here -- element o of the Philox stream (seed, *counter +
    // counter_offset), the values mvae_philox_fill would have written -- and stored to `noise` for the backward
    uint64_t seed; const uint64_t *counter; uint64_t counter_offset; int draw;
};

// precision of one expert: 1 / (exp(lv) + eps [+ eps])
__device__ __forceinline__ float poe_precision(float lv, int variant) {
    float var = expf(lv) + POE_EPS;
    if (variant == MVAE_POE_VARIANT_A) var = var + POE_EPS;
    return 1.0f / var;
}

__global__ __launch_bounds__(POE_THREADS) void poe_fwd_kernel(PoeArgs a, const uint32_t *masks,
                                                              float *noise, float *mu, float *logvar,
                                                              float *z, float *kl) {
    // per thread: the precision T_e and mu_e * T_e of every expert at this (row, latent) -- computed ONCE and
    // reused by all the terms that contain the expert (celeba19: 21 terms x 21 experts would otherwise
    // re-evaluate 441 exponentials per element)
    extern __shared__ float lds[];                     // [E][2][POE_THREADS] then [T][POE_THREADS]
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * (POE_THREADS / 64) + (threadIdx.x >> 6);
    if (b >= a.B) return;                              // whole waves exit together; no block barrier below
    float *mine = lds + threadIdx.x;
    float *klacc = lds + (size_t)a.E * 2 * POE_THREADS + threadIdx.x;      // this lane's KL partial per term of the chunk
    // the N(0,1) prior: mu = 0, logvar = 0 (MVAE_POE_NO_PRIOR: the caller's experts are the whole stack)
    const float t0 = a.no_prior ? 0.f : poe_precision(0.f, a.variant);
    const int t_lo = blockIdx.y * a.chunk, t_hi = min(a.T, t_lo + a.chunk);
    uint32_t used = 0;                                 // experts any term of this chunk contains (block-uniform)
    for (int t = t_lo; t < t_hi; ++t) { klacc[(t - t_lo) * POE_THREADS] = 0.f; used |= masks[t]; }
    const uint64_t launch = a.draw ? *a.counter + a.counter_offset : 0;
    for (int d = lane; d < a.D; d += 64) {
        const size_t oe = (size_t)b * a.ld + d;
        for (int e = 0; e < a.E; ++e) {
            if (!((used >> e) & 1u)) continue;
            const float te = poe_precision(a.ex.logvar[e][oe], a.variant);
            mine[(e * 2 + 0) * POE_THREADS] = te;
            mine[(e * 2 + 1) * POE_THREADS] = a.ex.mu[e][oe] * te;
        }
        for (int t = t_lo; t < t_hi; ++t) {
            uint32_t mask = masks[t];
            float sum_t = t0, sum_mt = 0.f * t0;
            for (int e = 0; mask; ++e, mask >>= 1) {  // wave-uniform walk over the experts of the term, in order
                if (!(mask & 1u)) continue;
                sum_mt += mine[(e * 2 + 1) * POE_THREADS];
                sum_t += mine[(e * 2 + 0) * POE_THREADS];
            }
            const float pmu = sum_mt / sum_t;
            const float pvar = 1.0f / sum_t;
            const float plv = (a.variant == MVAE_POE_VARIANT_A) ? logf(pvar + POE_EPS) : logf(pvar);
            const size_t o = ((size_t)t * a.B + b) * a.D + d;
            mu[o] = pmu;
            logvar[o] = plv;
            if (a.draw) {
                const float eps = philox_normal_at(o, launch, a.seed);
                noise[o] = eps;
                z[o] = eps * expf(0.5f * plv) + pmu;
            } else if (z) {
                z[o] = noise ? noise[o] * expf(0.5f * plv) + pmu : pmu;
            }
            klacc[(t - t_lo) * POE_THREADS] += 1.0f + plv - pmu * pmu - expf(plv);
        }
    }
    if (kl) {                                          // all 64 lanes are here again: wave sums per term
        for (int t = t_lo; t < t_hi; ++t) {
            const float s = wave_sum(klacc[(t - t_lo) * POE_THREADS]);
            if (lane == 0) kl[(size_t)t * a.B + b] = -0.5f * s;
        }
    }
}

// Backward.  Phase 1 (per term): total gradient reaching (mu_t, logvar_t) from z, from the
// KL row and from direct consumers, folded into A_t = dmu_t / S_t, B_t = dlv_t * dlv/dS and the
// fused mean; kept in LDS ([term][3][thread], conflict-free).  Phase 2 (per expert): T_e and
// exp(lv_e) once, then a sweep over the terms that contain the expert.
// dz of term t = dz[slot_a[t]] (+ dz_b[slot_b[t]]): the latent gradient may arrive in two buffers -- one per
// decoder, each holding only the terms that decoder saw -- instead of one accumulated [T,B,D] tensor.
struct PoeDzMap { signed char slot_a[MVAE_ELBO_MAX_TERMS], slot_b[MVAE_ELBO_MAX_TERMS]; };

__global__ __launch_bounds__(POE_THREADS) void poe_bwd_kernel(PoeArgs a, const uint32_t *masks,
                                                              const float *noise, const float *mu,
                                                              const float *logvar, const float *dz,
                                                              const float *dz_b, PoeDzMap dzm,
                                                              const float *dmu, const float *dlogvar,
                                                              const float *dkl, int dkl_stride,
                                                              mvae_expert_grads_t gr, int ldg) {
    extern __shared__ float lds[];   // [T][3][POE_THREADS]
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * (POE_THREADS / 64) + (threadIdx.x >> 6);
    if (b >= a.B) return;            // whole waves exit together; no block barrier is used below
    float *mine = lds + threadIdx.x;
    const int e_lo = blockIdx.y * a.chunk, e_hi = min(a.E, e_lo + a.chunk);
    // the experts of this block, as a mask: phase 1 is only needed for the terms that contain one of them
    const uint32_t own = (e_hi - e_lo >= 32 ? 0xffffffffu : ((1u << (e_hi - e_lo)) - 1u)) << e_lo;
    for (int d = lane; d < a.D; d += 64) {
        for (int t = 0; t < a.T; ++t) {
            if (!(masks[t] & own)) continue;
            const size_t o = ((size_t)t * a.B + b) * a.D + d;
            const float pmu = mu[o], plv = logvar[o];
            float gmu = dmu ? dmu[o] : 0.f;
            float glv = dlogvar ? dlogvar[o] : 0.f;
            if (dz) {
                float g;
                if (dz_b) {
                    const int sa = dzm.slot_a[t], sb = dzm.slot_b[t];       // block-uniform
                    g = sa >= 0 ? dz[((size_t)sa * a.B + b) * a.D + d] : 0.f;
                    if (sb >= 0) g += dz_b[((size_t)sb * a.B + b) * a.D + d];
                } else {
                    g = dz[o];
                }
                gmu += g;
                if (noise) glv += g * noise[o] * 0.5f * expf(0.5f * plv);
            }
            if (dkl) {
                // dkl_stride == B: per-row gradient [T,B]; 0: a per-term table [T]
                const float ks = dkl_stride ? dkl[(size_t)t * dkl_stride + b] : dkl[t];
                gmu += ks * pmu;
                glv += ks * (-0.5f) * (1.0f - expf(plv));
            }
            // recover S = sum of precisions from the fused log-variance
            float s, dlv_ds;
            if (a.variant == MVAE_POE_VARIANT_A) {
                const float v = expf(plv) - POE_EPS;        // 1/S
                s = 1.0f / v;
                dlv_ds = -(v * v) / (v + POE_EPS);          // d log(1/S + eps) / dS
            } else {
                s = expf(-plv);
                dlv_ds = -1.0f / s;
            }
            mine[(t * 3 + 0) * POE_THREADS] = gmu / s;
            mine[(t * 3 + 1) * POE_THREADS] = glv * dlv_ds;
            mine[(t * 3 + 2) * POE_THREADS] = pmu;
        }
        for (int e = e_lo; e < e_hi; ++e) {
            const size_t o = (size_t)b * a.ld + d;
            const float me = a.ex.mu[e][o], lve = a.ex.logvar[e][o];
            const float ex = expf(lve);
            const float te = poe_precision(lve, a.variant);
            float gm = 0.f, gt = 0.f;
            bool any = false;
            for (int t = 0; t < a.T; ++t) {
                if (!((masks[t] >> e) & 1u)) continue;
                any = true;
                const float at = mine[(t * 3 + 0) * POE_THREADS];
                gm += at * te;
                gt += at * (me - mine[(t * 3 + 2) * POE_THREADS]) + mine[(t * 3 + 1) * POE_THREADS];
            }
            const size_t og = (size_t)b * ldg + d;
            gr.dmu[e][og] = any ? gm : 0.f;
            gr.dlogvar[e][og] = any ? gt * (-(te * te) * ex) : 0.f;
        }
    }
}

// Stand-alone KL rows for the reference-surface elbo_loss(mu, logvar): mnist/train.py:56.
__global__ __launch_bounds__(POE_THREADS) void kl_rows_fwd_kernel(const float *mu, const float *logvar, float *kl,
                                                                  int B, int D) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * (POE_THREADS / 64) + (threadIdx.x >> 6);
    if (b >= B) return;
    float s = 0.f;
    for (int d = lane; d < D; d += 64) {
        const float m = mu[(size_t)b * D + d], lv = logvar[(size_t)b * D + d];
        s += 1.0f + lv - m * m - expf(lv);
    }
    s = wave_sum(s);
    if (lane == 0) kl[b] = -0.5f * s;
}

__global__ __launch_bounds__(256) void kl_rows_bwd_kernel(const float *mu, const float *logvar, const float *dkl,
                                                          float *dmu, float *dlogvar, int B, int D) {
    const size_t n = (size_t)B * D;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float g = dkl[i / D];
        dmu[i] = g * mu[i];
        dlogvar[i] = g * (-0.5f) * (1.0f - expf(logvar[i]));
    }
}

#ifndef MVAE_POE_CHUNK
#define MVAE_POE_CHUNK 3        // terms (forward) / experts (backward) per block of a many-term launch; 0: one block walks all
#endif
#ifndef MVAE_POE_CHUNK_SMALL
#define MVAE_POE_CHUNK_SMALL 1  // terms / experts per block when there are at most 4 of them (the bimodal steps: ONE each -- CelebA 2.408 -> 2.379 ms, MNIST 0.2985 -> 0.2970, profiles/r04_taps_poe_ab.txt); 0: all in one block
#endif
inline int poe_chunk(int n) {
    if (n <= 0) return 1;
    if (n <= 4) return MVAE_POE_CHUNK_SMALL > 0 ? MVAE_POE_CHUNK_SMALL : n;
    return MVAE_POE_CHUNK <= 0 ? n : MVAE_POE_CHUNK;
}

inline bool poe_args_ok(const mvae_experts_t *ex, int ld, int E, int T, int B, int D, int variant) {
    if (!ex || E < 0 || E > MVAE_MAX_EXPERTS || T <= 0 || T > POE_MAX_TERMS || B <= 0 || D <= 0 || ld < D)
        return false;
    const int base = variant & ~MVAE_POE_NO_PRIOR;
    if (base != MVAE_POE_VARIANT_A && base != MVAE_POE_VARIANT_B) return false;
    if ((variant & MVAE_POE_NO_PRIOR) && E < 1) return false;       // an empty product has no precision
    for (int e = 0; e < E; ++e)
        if (!ex->mu[e] || !ex->logvar[e]) return false;
    return true;
}

}  // namespace

MVAE_EXPORT int mvae_poe_fwd(const mvae_experts_t *experts, int ld, int E, const uint32_t *masks_dev, int T,
                             const float *noise, float *mu, float *logvar, float *z, float *kl, int B, int D,
                             int variant, mvae_stream_t stream) {
    if (!poe_args_ok(experts, ld, E, T, B, D, variant) || !masks_dev || !mu || !logvar) return MVAE_ERR_ARG;
    PoeArgs a;
    a.ex = *experts; a.ld = ld; a.E = E; a.T = T; a.B = B; a.D = D;
    a.variant = variant & ~MVAE_POE_NO_PRIOR; a.no_prior = (variant & MVAE_POE_NO_PRIOR) ? 1 : 0;
    a.seed = 0; a.counter = nullptr; a.counter_offset = 0; a.draw = 0;
    const int rows = POE_THREADS / 64;
    a.chunk = poe_chunk(T);
    const size_t lds_bytes = ((size_t)E * 2 + a.chunk) * POE_THREADS * sizeof(float);
    hipLaunchKernelGGL(poe_fwd_kernel, dim3((B + rows - 1) / rows, (T + a.chunk - 1) / a.chunk), dim3(POE_THREADS), lds_bytes, (hipStream_t)stream, a,
                       masks_dev, const_cast<float *>(noise), mu, logvar, z, kl);
    return mvae_launch_status();
}

// mvae_poe_fwd that DRAWS its reparameterisation noise: eps[T,B,D] = the standard normals mvae_philox_fill(seed,
// *counter_dev + counter_offset) would write, generated inside the launch and stored to `noise_out` (the backward
// reads it).  One launch less at the head of a fused step.
MVAE_EXPORT int mvae_poe_fwd_draw(const mvae_experts_t *experts, int ld, int E, const uint32_t *masks_dev, int T,
                                  float *noise_out, uint64_t seed, const uint64_t *counter_dev,
                                  uint64_t counter_offset, float *mu, float *logvar, float *z, float *kl, int B,
                                  int D, int variant, mvae_stream_t stream) {
    if (!poe_args_ok(experts, ld, E, T, B, D, variant) || !masks_dev || !mu || !logvar || !z || !noise_out ||
        !counter_dev)
        return MVAE_ERR_ARG;
    PoeArgs a;
    a.ex = *experts; a.ld = ld; a.E = E; a.T = T; a.B = B; a.D = D;
    a.variant = variant & ~MVAE_POE_NO_PRIOR; a.no_prior = (variant & MVAE_POE_NO_PRIOR) ? 1 : 0;
    a.seed = seed; a.counter = counter_dev; a.counter_offset = counter_offset; a.draw = 1;
    const int rows = POE_THREADS / 64;
    a.chunk = poe_chunk(T);
    const size_t lds_bytes = ((size_t)E * 2 + a.chunk) * POE_THREADS * sizeof(float);
    hipLaunchKernelGGL(poe_fwd_kernel, dim3((B + rows - 1) / rows, (T + a.chunk - 1) / a.chunk), dim3(POE_THREADS), lds_bytes, (hipStream_t)stream, a,
                       masks_dev, noise_out, mu, logvar, z, kl);
    return mvae_launch_status();
}

MVAE_EXPORT int mvae_poe_bwd(const mvae_experts_t *experts, int ld, int E, const uint32_t *masks_dev, int T,
                             const float *noise, const float *mu, const float *logvar, const float *dz,
                             const float *dmu, const float *dlogvar, const float *dkl, int dkl_per_term,
                             const mvae_expert_grads_t *grads, int ldg, int B, int D, int variant,
                             mvae_stream_t stream) {
    // dkl is [T,B] (per row) or, with dkl_per_term, a [T] table (beta/B of each ELBO term)
    const int dkl_stride = dkl_per_term ? 0 : B;
    if (!poe_args_ok(experts, ld, E, T, B, D, variant) || !masks_dev || !mu || !logvar || !grads || ldg < D)
        return MVAE_ERR_ARG;
    for (int e = 0; e < E; ++e)
        if (!grads->dmu[e] || !grads->dlogvar[e]) return MVAE_ERR_ARG;
    PoeArgs a;
    a.ex = *experts; a.ld = ld; a.E = E; a.T = T; a.B = B; a.D = D;
    a.variant = variant & ~MVAE_POE_NO_PRIOR; a.no_prior = (variant & MVAE_POE_NO_PRIOR) ? 1 : 0;
    a.seed = 0; a.counter = nullptr; a.counter_offset = 0; a.draw = 0;
    const int rows = POE_THREADS / 64;
    const size_t lds_bytes = (size_t)T * 3 * POE_THREADS * sizeof(float);
    PoeDzMap none = {};
    a.chunk = poe_chunk(E);
    hipLaunchKernelGGL(poe_bwd_kernel, dim3((B + rows - 1) / rows, E ? (E + a.chunk - 1) / a.chunk : 1), dim3(POE_THREADS), lds_bytes,
                       (hipStream_t)stream, a, masks_dev, noise, mu, logvar, dz, (const float *)nullptr, none, dmu,
                       dlogvar, dkl, dkl_stride, *grads, ldg);
    return mvae_launch_status();
}

MVAE_EXPORT int mvae_poe_bwd_split(const mvae_experts_t *experts, int ld, int E, const uint32_t *masks_dev, int T,
                                   const float *noise, const float *mu, const float *logvar, const float *dz_a,
                                   const int *slot_a, const float *dz_b, const int *slot_b, const float *dkl,
                                   int dkl_per_term, const mvae_expert_grads_t *grads, int ldg, int B, int D,
                                   int variant, mvae_stream_t stream) {
    const int dkl_stride = dkl_per_term ? 0 : B;
    if (!poe_args_ok(experts, ld, E, T, B, D, variant) || !masks_dev || !mu || !logvar || !grads || ldg < D ||
        !dz_a || !dz_b || !slot_a || !slot_b || T > MVAE_ELBO_MAX_TERMS)
        return MVAE_ERR_ARG;
    for (int e = 0; e < E; ++e)
        if (!grads->dmu[e] || !grads->dlogvar[e]) return MVAE_ERR_ARG;
    PoeDzMap m;
    for (int t = 0; t < T; ++t) {
        if (slot_a[t] < -1 || slot_a[t] >= T || slot_b[t] < -1 || slot_b[t] >= T) return MVAE_ERR_ARG;
        m.slot_a[t] = (signed char)slot_a[t]; m.slot_b[t] = (signed char)slot_b[t];
    }
    PoeArgs a;
    a.ex = *experts; a.ld = ld; a.E = E; a.T = T; a.B = B; a.D = D;
    a.variant = variant & ~MVAE_POE_NO_PRIOR; a.no_prior = (variant & MVAE_POE_NO_PRIOR) ? 1 : 0;
    a.seed = 0; a.counter = nullptr; a.counter_offset = 0; a.draw = 0;
    const int rows = POE_THREADS / 64;
    const size_t lds_bytes = (size_t)T * 3 * POE_THREADS * sizeof(float);
    a.chunk = poe_chunk(E);
    hipLaunchKernelGGL(poe_bwd_kernel, dim3((B + rows - 1) / rows, E ? (E + a.chunk - 1) / a.chunk : 1), dim3(POE_THREADS), lds_bytes,
                       (hipStream_t)stream, a, masks_dev, noise, mu, logvar, dz_a, dz_b, m, (const float *)nullptr,
                       (const float *)nullptr, dkl, dkl_stride, *grads, ldg);
    return mvae_launch_status();
}

MVAE_EXPORT int mvae_kl_rows_fwd(const float *mu, const float *logvar, float *kl, int B, int D,
                                 mvae_stream_t stream) {
    if (!mu || !logvar || !kl || B <= 0 || D <= 0) return MVAE_ERR_ARG;
    const int rows = POE_THREADS / 64;
    hipLaunchKernelGGL(kl_rows_fwd_kernel, dim3((B + rows - 1) / rows), dim3(POE_THREADS), 0, (hipStream_t)stream,
                       mu, logvar, kl, B, D);
    return mvae_launch_status();
}

MVAE_EXPORT int mvae_kl_rows_bwd(const float *mu, const float *logvar, const float *dkl, float *dmu,
                                 float *dlogvar, int B, int D, mvae_stream_t stream) {
    if (!mu || !logvar || !dkl || !dmu || !dlogvar || B <= 0 || D <= 0) return MVAE_ERR_ARG;
    size_t blocks = ((size_t)B * D + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(kl_rows_bwd_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, mu, logvar, dkl,
                       dmu, dlogvar, B, D);
    return mvae_launch_status();
}
