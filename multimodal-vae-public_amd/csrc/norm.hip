// norm.hip -- training-mode BatchNorm2d / BatchNorm1d (+ fused Swish), HBM-bound.
//
// x is [G*B, C, HW]; each of the G groups of B samples is normalised with its own batch
// statistics, so the G `model()` calls the reference issues per train step on the same layer
// (celeba/train.py:193-195: 3 decoder passes; celeba19/train.py:264-302: 21) run as one launch.
//
// Work split: block (s, c, g) owns a slice of batch rows of channel c in group g and sweeps each
// row's HW contiguous floats with float4 loads (lanes along the spatial axis; rows in parallel when
// HW/4 < 256), no integer division in the loop.  Statistics are (count, mean, M2) per slice
// from ONE sweep of the slice (sums of x - p and (x - p)^2 around a pivot p = the slice's first
// element, so there is no E[x^2]-E[x]^2 cancellation), merged with Chan's formula.  All cross-block reductions go through the
// caller's workspace in a fixed order: deterministic, no atomics.
#include "common.h"

namespace {

constexpr int BN_THREADS = 256;
constexpr int BN_SLICE_ELEMS = 8192;

struct BnShape {
    int G, B, C, HW, S, n;      // n = B * HW elements per (group, channel); S slices of `rows` batch rows
    int rows, vec, tx;          // vec: HW % 4 == 0 (float4 path); tx = threads along a row (power of two)
};

__host__ __device__ inline int bn_max_slices(int n_per_group) { return n_per_group / BN_SLICE_ELEMS + 2; }

// f4(offset) is called with the element offset of each aligned float4 of the slice (vec path),
// f1(offset) with each scalar element (HW not a multiple of 4, e.g. 5x5 maps and BatchNorm1d).
template <class F4, class F1>
__device__ __forceinline__ void bn_slice_loop(const BnShape &sh, int g, int c, int s, F4 f4, F1 f1) {
    const int b_lo = s * sh.rows, b_hi = min(sh.B, b_lo + sh.rows);
    if (sh.vec) {
        const int hw4 = sh.HW >> 2, tx = threadIdx.x & (sh.tx - 1), ty = threadIdx.x / sh.tx;
        const int ny = BN_THREADS / sh.tx;
        // four rows (or four float4 columns) per trip: independent loads in flight instead of one per iteration
        if (hw4 <= sh.tx) {
            const bool col_ok = tx < hw4;
            const size_t row_stride = (size_t)sh.C * sh.HW;
            int b = b_lo + ty;
            size_t base = ((size_t)(g * sh.B + b) * sh.C + c) * sh.HW + 4 * (size_t)tx;
            for (; b + 3 * ny < b_hi; b += 4 * ny, base += 4 * ny * row_stride) {
                if (col_ok) {
                    f4(base); f4(base + ny * row_stride); f4(base + 2 * ny * row_stride); f4(base + 3 * ny * row_stride);
                }
            }
            for (; b < b_hi; b += ny, base += ny * row_stride)
                if (col_ok) f4(base);
        } else {
            for (int b = b_lo + ty; b < b_hi; b += ny) {
                const size_t base = ((size_t)(g * sh.B + b) * sh.C + c) * sh.HW;
                int q = tx;
                for (; q + 3 * sh.tx < hw4; q += 4 * sh.tx) {
                    f4(base + 4 * (size_t)q); f4(base + 4 * (size_t)(q + sh.tx));
                    f4(base + 4 * (size_t)(q + 2 * sh.tx)); f4(base + 4 * (size_t)(q + 3 * sh.tx));
                }
                for (; q < hw4; q += sh.tx) f4(base + 4 * (size_t)q);
            }
        }
    } else {
        const int n_lo = b_lo * sh.HW, n_hi = b_hi * sh.HW;
        for (int n = n_lo + threadIdx.x; n < n_hi; n += BN_THREADS) {
            const int b = n / sh.HW, sp = n - b * sh.HW;
            f1(((size_t)(g * sh.B + b) * sh.C + c) * sh.HW + sp);
        }
    }
}

__device__ __forceinline__ float4 ld4(const float *p, size_t o) { return *reinterpret_cast<const float4 *>(p + o); }
__device__ __forceinline__ void st4(float *p, size_t o, float4 v) { *reinterpret_cast<float4 *>(p + o) = v; }

// ws[((g*C + c)*S + s)*3 + {0,1,2}] = (count, mean, M2) of the slice.  ONE pass over x: sums of
// (x - p) and (x - p)^2 with the pivot p = first element of the slice (a sample of the same
// distribution, so |p - mean| ~ std and M2 = S2 - S1^2/n loses at most a bit or two); slices are
// then merged with Chan's formula.  (A second pass for the centred sum cost 25 % of the forward.)
__global__ __launch_bounds__(BN_THREADS) void bn_partial_stats_kernel(const float *__restrict__ x, float *__restrict__ ws, BnShape sh) {
    __shared__ float red[16];
    const int s = blockIdx.x, c = blockIdx.y, g = blockIdx.z;
    const int b_lo = s * sh.rows, b_hi = min(sh.B, b_lo + sh.rows);
    const float cnt = (float)(max(b_hi - b_lo, 0) * sh.HW);
    const float pivot = cnt > 0.f ? x[((size_t)(g * sh.B + b_lo) * sh.C + c) * sh.HW] : 0.f;
    float s1 = 0.f, s2 = 0.f;
    bn_slice_loop(sh, g, c, s,
                  [&](size_t o) {
                      const float4 v = ld4(x, o);
                      const float a = v.x - pivot, b = v.y - pivot, cc = v.z - pivot, d = v.w - pivot;
                      s1 += (a + b) + (cc + d);
                      s2 += (a * a + b * b) + (cc * cc + d * d);
                  },
                  [&](size_t o) { const float d = x[o] - pivot; s1 += d; s2 += d * d; });
    s1 = block_sum(s1, red);
    s2 = block_sum(s2, red);
    if (threadIdx.x == 0) {
        float *o = ws + ((size_t)(g * sh.C + c) * sh.S + s) * 3;
        const float mean_rel = cnt > 0.f ? s1 / cnt : 0.f;
        o[0] = cnt; o[1] = pivot + mean_rel; o[2] = fmaxf(s2 - s1 * mean_rel, 0.f);
    }
}

// Chan merge of the S slice statistics of (g, c): returns mean and biased variance.
__device__ inline void bn_merge(const float *ws, const BnShape &sh, int g, int c, float *mean, float *var) {
    const float *p = ws + (size_t)(g * sh.C + c) * sh.S * 3;
    float n = 0.f, m = 0.f, m2 = 0.f;
    for (int s = 0; s < sh.S; ++s) {
        const float nb = p[s * 3], mb = p[s * 3 + 1], m2b = p[s * 3 + 2];
        if (nb <= 0.f) continue;
        const float nt = n + nb, d = mb - m;
        m += d * (nb / nt);
        m2 += m2b + d * d * (n * nb / nt);
        n = nt;
    }
    *mean = m;
    *var = n > 0.f ? m2 / n : 0.f;
}

// y = swish?(gamma * ((x - mean) * invstd) + beta); also saves mean/invstd and advances the
// running statistics (one block per channel does that, sequentially over groups).
__global__ __launch_bounds__(BN_THREADS) void bn_fwd_apply_kernel(const float *__restrict__ x, const float *gamma,
                                                                  const float *beta, float *__restrict__ y, const float *ws,
                                                                  float *save_mean, float *save_invstd,
                                                                  float *running_mean, float *running_var,
                                                                  BnShape sh, float eps, float momentum,
                                                                  int n_updates, const int *n_updates_dev,
                                                                  int swish) {
    const int s = blockIdx.x, c = blockIdx.y, g = blockIdx.z;
    if (n_updates_dev) n_updates = *n_updates_dev;   // device-side count: graph replays may vary it
    float mean, var;
    bn_merge(ws, sh, g, c, &mean, &var);
    const float invstd = rsqrtf(var + eps);
    if (s == 0 && threadIdx.x == 0) {
        save_mean[g * sh.C + c] = mean;
        save_invstd[g * sh.C + c] = invstd;
        if (g == 0 && running_mean) {
            float rm = running_mean[c], rv = running_var[c];
            const float unb = sh.n > 1 ? (float)sh.n / (float)(sh.n - 1) : 1.f;
            for (int gg = 0; gg < sh.G; ++gg) {
                float m, v;
                bn_merge(ws, sh, gg, c, &m, &v);
                for (int u = 0; u < n_updates; ++u) {
                    rm = (1.f - momentum) * rm + momentum * m;
                    rv = (1.f - momentum) * rv + momentum * (v * unb);
                }
            }
            running_mean[c] = rm;
            running_var[c] = rv;
        }
    }
    if (!y) return;                                        // statistics-only call
    const float ga = gamma[c], be = beta[c];
    auto one = [&](float v) {
        const float h = ga * ((v - mean) * invstd) + be;   // same expression as the backward's
        return swish ? swishf_(h) : h;
    };
    bn_slice_loop(sh, g, c, s,
                  [&](size_t o) {
                      const float4 v = ld4(x, o);
                      st4(y, o, make_float4(one(v.x), one(v.y), one(v.z), one(v.w)));
                  },
                  [&](size_t o) { y[o] = one(x[o]); });
}

// ws[((g*C + c)*S + s)*2 + {0,1}] = (sum dh, sum dh * xhat), dh = dy * swish'(h)
__global__ __launch_bounds__(BN_THREADS) void bn_bwd_partial_kernel(const float *__restrict__ dy, const float *__restrict__ x,
                                                                    const float *gamma, const float *beta,
                                                                    const float *save_mean,
                                                                    const float *save_invstd, float *ws,
                                                                    BnShape sh, int swish) {
    __shared__ float red[16];
    const int s = blockIdx.x, c = blockIdx.y, g = blockIdx.z;
    const float mean = save_mean[g * sh.C + c], invstd = save_invstd[g * sh.C + c];
    const float ga = gamma[c], be = beta[c];
    float s1 = 0.f, s2 = 0.f;
    auto one = [&](float xv, float d) {
        const float xh = (xv - mean) * invstd;
        if (swish) d *= swish_grad_(ga * xh + be);
        s1 += d;
        s2 += d * xh;
    };
    bn_slice_loop(sh, g, c, s,
                  [&](size_t o) {
                      const float4 xv = ld4(x, o), dv = ld4(dy, o);
                      one(xv.x, dv.x); one(xv.y, dv.y); one(xv.z, dv.z); one(xv.w, dv.w);
                  },
                  [&](size_t o) { one(x[o], dy[o]); });
    s1 = block_sum(s1, red);
    s2 = block_sum(s2, red);
    if (threadIdx.x == 0) {
        float *o = ws + ((size_t)(g * sh.C + c) * sh.S + s) * 2;
        o[0] = s1; o[1] = s2;
    }
}

__global__ __launch_bounds__(BN_THREADS) void bn_bwd_apply_kernel(const float *__restrict__ dy, const float *__restrict__ x,
                                                                  const float *gamma, const float *beta,
                                                                  const float *save_mean,
                                                                  const float *save_invstd, const float *ws,
                                                                  float *__restrict__ dx, float *dgamma, float *dbeta,
                                                                  BnShape sh, int swish, int accumulate) {
    const int s = blockIdx.x, c = blockIdx.y, g = blockIdx.z;
    const float *p = ws + (size_t)(g * sh.C + c) * sh.S * 2;
    float s1 = 0.f, s2 = 0.f;
    for (int k = 0; k < sh.S; ++k) { s1 += p[k * 2]; s2 += p[k * 2 + 1]; }
    if (s == 0 && g == 0 && threadIdx.x == 0) {
        float t1 = 0.f, t2 = 0.f;
        for (int gg = 0; gg < sh.G; ++gg) {
            const float *q = ws + (size_t)(gg * sh.C + c) * sh.S * 2;
            for (int k = 0; k < sh.S; ++k) { t1 += q[k * 2]; t2 += q[k * 2 + 1]; }
        }
        if (accumulate) { t1 += dbeta[c]; t2 += dgamma[c]; }
        dbeta[c] = t1;
        dgamma[c] = t2;
    }
    const float mean = save_mean[g * sh.C + c], invstd = save_invstd[g * sh.C + c];
    const float ga = gamma[c], be = beta[c];
    const float inv_n = 1.f / (float)sh.n;
    const float m1 = s1 * inv_n, m2 = s2 * inv_n, k = ga * invstd;
    auto one = [&](float xv, float d) {
        const float xh = (xv - mean) * invstd;
        if (swish) d *= swish_grad_(ga * xh + be);
        return k * (d - m1 - xh * m2);
    };
    bn_slice_loop(sh, g, c, s,
                  [&](size_t o) {
                      const float4 xv = ld4(x, o), dv = ld4(dy, o);
                      st4(dx, o, make_float4(one(xv.x, dv.x), one(xv.y, dv.y), one(xv.z, dv.z), one(xv.w, dv.w)));
                  },
                  [&](size_t o) { dx[o] = one(x[o], dy[o]); });
}

__global__ __launch_bounds__(256) void bn_eval_kernel(const float *x, const float *gamma, const float *beta,
                                                      float *y, const float *rm, const float *rv, size_t total,
                                                      int C, int HW, float eps, int swish) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)((i / HW) % C);
        const float h = (x[i] - rm[c]) * rsqrtf(rv[c] + eps) * gamma[c] + beta[c];
        y[i] = swish ? swishf_(h) : h;
    }
}

inline bool bn_shape(int G, int B, int C, int HW, const void *a, const void *b, const void *c, BnShape *sh) {
    if (G <= 0 || B <= 0 || C <= 0 || HW <= 0) return false;
    if ((long)G * B * C * HW >= (1L << 40) || (long)B * HW >= (1L << 31)) return false;
    sh->G = G; sh->B = B; sh->C = C; sh->HW = HW; sh->n = B * HW;
    sh->rows = BN_SLICE_ELEMS / HW;
    if (sh->rows < 1) sh->rows = 1;
    if (sh->rows > B) sh->rows = B;
    sh->S = (B + sh->rows - 1) / sh->rows;
    sh->vec = (HW % 4 == 0) && aligned16(a) && (!b || aligned16(b)) && (!c || aligned16(c));
    int tx = 1;
    while (tx * 2 <= (HW >> 2) && tx * 2 <= BN_THREADS) tx *= 2;
    sh->tx = tx;
    return sh->S <= bn_max_slices(sh->n);
}

}  // namespace

MVAE_EXPORT size_t mvae_bn_ws_bytes(int G, int C, int n_per_group) {
    if (G <= 0 || C <= 0 || n_per_group <= 0) return 0;
    return (size_t)G * C * bn_max_slices(n_per_group) * 3 * sizeof(float);
}

MVAE_EXPORT int mvae_bn_train_fwd(const float *x, const float *gamma, const float *beta, float *y,
                                  float *save_mean, float *save_invstd, float *running_mean,
                                  float *running_var, int G, int B, int C, int HW, float eps, float momentum,
                                  int n_updates, const int *n_updates_dev, int flags, void *ws,
                                  size_t ws_bytes, mvae_stream_t stream) {
    BnShape sh;
    if (!x || !gamma || !beta || !save_mean || !save_invstd || !bn_shape(G, B, C, HW, x, y, nullptr, &sh))
        return MVAE_ERR_ARG;
    if ((running_mean == nullptr) != (running_var == nullptr)) return MVAE_ERR_ARG;
    if (!ws || ws_bytes < mvae_bn_ws_bytes(G, C, sh.n)) return MVAE_ERR_WS;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(sh.S, C, G);
    hipLaunchKernelGGL(bn_partial_stats_kernel, grid, dim3(BN_THREADS), 0, st, x, (float *)ws, sh);
    // y == NULL: only the (s = 0) block of each (channel, group) has work -- saved + running statistics
    const dim3 agrid(y ? sh.S : 1, C, G);
    hipLaunchKernelGGL(bn_fwd_apply_kernel, agrid, dim3(BN_THREADS), 0, st, x, gamma, beta, y, (const float *)ws,
                       save_mean, save_invstd, running_mean, running_var, sh, eps, momentum, n_updates,
                       n_updates_dev, (flags & MVAE_ACT_SWISH) ? 1 : 0);
    return mvae_launch_status();
}

MVAE_EXPORT int mvae_bn_train_bwd(const float *dy, const float *x, const float *gamma, const float *beta,
                                  const float *save_mean, const float *save_invstd, float *dx, float *dgamma,
                                  float *dbeta, int G, int B, int C, int HW, int flags, void *ws,
                                  size_t ws_bytes, mvae_stream_t stream) {
    BnShape sh;
    if (!dy || !x || !gamma || !beta || !save_mean || !save_invstd || !dx || !dgamma || !dbeta ||
        !bn_shape(G, B, C, HW, x, dy, dx, &sh))
        return MVAE_ERR_ARG;
    if (!ws || ws_bytes < mvae_bn_ws_bytes(G, C, sh.n)) return MVAE_ERR_WS;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(sh.S, C, G);
    const int swish = (flags & MVAE_ACT_SWISH) ? 1 : 0;
    hipLaunchKernelGGL(bn_bwd_partial_kernel, grid, dim3(BN_THREADS), 0, st, dy, x, gamma, beta, save_mean,
                       save_invstd, (float *)ws, sh, swish);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, grid, dim3(BN_THREADS), 0, st, dy, x, gamma, beta, save_mean,
                       save_invstd, (const float *)ws, dx, dgamma, dbeta, sh, swish,
                       (flags & MVAE_ACCUMULATE) ? 1 : 0);
    return mvae_launch_status();
}

MVAE_EXPORT int mvae_bn_eval_fwd(const float *x, const float *gamma, const float *beta, float *y,
                                 const float *running_mean, const float *running_var, int N, int C, int HW,
                                 float eps, int flags, mvae_stream_t stream) {
    if (!x || !gamma || !beta || !y || !running_mean || !running_var || N <= 0 || C <= 0 || HW <= 0)
        return MVAE_ERR_ARG;
    const size_t total = (size_t)N * C * HW;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(bn_eval_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y,
                       running_mean, running_var, total, C, HW, eps, (flags & MVAE_ACT_SWISH) ? 1 : 0);
    return mvae_launch_status();
}
