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

// Chan merge of the S slice statistics of (g, c) by the WHOLE block: returns mean and biased variance to every thread.
// Thread t folds slices t, t + 256, ... (one for every batch the step runs), the wave merges its 64 partials in a
// butterfly, and every thread folds the four wave results (lane 0's) in wave order from LDS: a fixed tree --
// deterministic, the same bits in every block of the (group, channel).  (Round 5's form was one thread walking the
// S records in order, every thread of every block redundantly: a chain of S dependent loads -- the `continue` on an empty
// record keeps hipcc from hoisting them -- and 2 S divisions in front of the first byte a block streams: 8 us of the
// 38-us apply launch of CelebA's 32 x 32 maps at S = 32, which ran at 3.5 TB/s where its backward sibling does 5.9.)
struct BnMoments { float n, m, m2; };
__device__ __forceinline__ BnMoments bn_chan(BnMoments a, BnMoments b) {
    const float nt = a.n + b.n;
    if (nt <= 0.f) return a;
    const float d = b.m - a.m, r = b.n / nt;
    BnMoments o;
    o.n = nt; o.m = a.m + d * r; o.m2 = a.m2 + b.m2 + d * d * (a.n * r);
    return o;
}
__device__ __forceinline__ void bn_merge_block(const float *ws, const BnShape &sh, int g, int c, float *mean, float *var,
                                               float (*red)[3]) {
    const float *p = ws + (size_t)(g * sh.C + c) * sh.S * 3;
    BnMoments a = {0.f, 0.f, 0.f};
    for (int s = threadIdx.x; s < sh.S; s += BN_THREADS) {
        const BnMoments b = {p[s * 3], p[s * 3 + 1], p[s * 3 + 2]};
        a = bn_chan(a, b);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        BnMoments b;
        b.n = __shfl_xor(a.n, off, 64); b.m = __shfl_xor(a.m, off, 64); b.m2 = __shfl_xor(a.m2, off, 64);
        a = bn_chan(a, b);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();                                   // the previous call's readers are done with `red`
    if (lane == 0) { red[wave][0] = a.n; red[wave][1] = a.m; red[wave][2] = a.m2; }
    __syncthreads();
    BnMoments t = {red[0][0], red[0][1], red[0][2]};
#pragma unroll
    for (int w = 1; w < BN_THREADS / 64; ++w) {
        const BnMoments b = {red[w][0], red[w][1], red[w][2]};
        t = bn_chan(t, b);
    }
    *mean = t.m;
    *var = t.n > 0.f ? t.m2 / t.n : 0.f;
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
    __shared__ float red[BN_THREADS / 64][3];
    const int s = blockIdx.x, c = blockIdx.y, g = blockIdx.z;
    if (n_updates_dev) n_updates = *n_updates_dev;   // device-side count: graph replays may vary it
    float mean, var;
    bn_merge_block(ws, sh, g, c, &mean, &var, red);
    const float invstd = rsqrtf(var + eps);
    if (s == 0 && threadIdx.x == 0) {
        save_mean[g * sh.C + c] = mean;
        save_invstd[g * sh.C + c] = invstd;
    }
    if (s == 0 && g == 0 && running_mean) {            // block-uniform: the whole block merges every group, in order
        float rm = running_mean[c], rv = running_var[c];
        const float unb = sh.n > 1 ? (float)sh.n / (float)(sh.n - 1) : 1.f;
        for (int gg = 0; gg < sh.G; ++gg) {
            float m, v;
            bn_merge_block(ws, sh, gg, c, &m, &v, red);
            for (int u = 0; u < n_updates; ++u) {
                rm = (1.f - momentum) * rm + momentum * m;
                rv = (1.f - momentum) * rv + momentum * (v * unb);
            }
        }
        if (threadIdx.x == 0) {
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


// ------------------------------------------------------------------------------------------
// Single-launch forms (round 4): statistics + apply from ONE read of the slice.
//
// The two-launch path above costs a (group, channel) slice two reads and two launches per direction; for the layers
// whose slice is small that is all launch latency and cold prologues (CelebA: 8 of the 14 BatchNorm layers hold
// <= 16 K elements per slice -- the 8x8 and 5x5 maps and the five BatchNorm1d -- 32 of the step's 50 BatchNorm
// launches).  Here a block owns ONE channel (or 16 BatchNorm1d columns) of ALL groups, keeps the elements in
// registers between the statistics and the apply, and thread 0 advances the running statistics over the groups in
// order, as the two-launch path does.  Mean and variance are a true two-pass (sum, then centred squares) over the
// registers; reductions go wave -> LDS -> every thread in a fixed order: deterministic, no atomics, no hand-off
// between blocks.
// ------------------------------------------------------------------------------------------
constexpr int BNF_THREADS = 1024;
constexpr int BNF_WAVES = BNF_THREADS / 64;
constexpr int BNF_MAX_N = 16384;            // elements per (group, channel) slice a block keeps in registers

// totals[g] = sum over the block of part[g], g < GMAX; red: BNF_WAVES * GMAX floats of LDS.  Every thread gets them.
template <int GMAX>
__device__ __forceinline__ void bnf_block_sums(float (&part)[GMAX], float *red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int g = 0; g < GMAX; ++g) part[g] = wave_sum(part[g]);
    __syncthreads();                        // the previous round's readers are done with `red`
    if (lane == 0) {
#pragma unroll
        for (int g = 0; g < GMAX; ++g) red[wave * GMAX + g] = part[g];
    }
    __syncthreads();
#pragma unroll
    for (int g = 0; g < GMAX; ++g) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < BNF_WAVES; ++w) t += red[w * GMAX + g];
        part[g] = t;
    }
}

template <bool VEC> struct BnfVal { typedef float T; };
template <> struct BnfVal<true> { typedef float4 T; };

// thread t owns units t, t + 1024, ... of every group's slice (a unit = one aligned float4 of a row, or one element);
// off[k] = offset of unit k inside a group's [B, C, HW] slab, ok[k] = the unit exists
template <bool VEC, int KMAX>
__device__ __forceinline__ void bnf_units(const BnShape &sh, int c, int (&off)[KMAX], bool (&ok)[KMAX]) {
    const int units = VEC ? sh.n >> 2 : sh.n, per_row = VEC ? sh.HW >> 2 : sh.HW;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        const int u = threadIdx.x + k * BNF_THREADS;
        ok[k] = u < units;
        const int uu = ok[k] ? u : 0;       // loads are unconditional (an existing unit); ok[] masks the sums
        const int b = uu / per_row, q = uu - b * per_row;
        off[k] = (b * sh.C + c) * sh.HW + (VEC ? 4 * q : q);
    }
}

template <bool VEC, int GMAX, int KMAX>
__global__ __launch_bounds__(BNF_THREADS) void bn_fused_fwd_kernel(const float *__restrict__ x, const float *gamma,
                                                                   const float *beta, float *__restrict__ y,
                                                                   float *save_mean, float *save_invstd,
                                                                   float *running_mean, float *running_var, BnShape sh,
                                                                   float eps, float momentum, int n_updates,
                                                                   const int *n_updates_dev, int swish) {
    typedef typename BnfVal<VEC>::T V;
    __shared__ float red[BNF_WAVES * GMAX];
    const int c = blockIdx.x;
    const size_t gstride = (size_t)sh.B * sh.C * sh.HW;
    int off[KMAX];
    bool ok[KMAX];
    bnf_units<VEC, KMAX>(sh, c, off, ok);
    V v[GMAX][KMAX];
#pragma unroll
    for (int g = 0; g < GMAX; ++g)
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
            v[g][k] = *reinterpret_cast<const V *>(x + (g < sh.G ? g : 0) * gstride + off[k]);
    float s[GMAX], mean[GMAX], invstd[GMAX], var[GMAX];
    auto sum_of = [](const V &a) { if constexpr (VEC) return (a.x + a.y) + (a.z + a.w); else return a; };
#pragma unroll
    for (int g = 0; g < GMAX; ++g) {
        s[g] = 0.f;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) s[g] += ok[k] ? sum_of(v[g][k]) : 0.f;
    }
    bnf_block_sums<GMAX>(s, red);
    const float inv_n = 1.f / (float)sh.n;
#pragma unroll
    for (int g = 0; g < GMAX; ++g) {
        mean[g] = s[g] * inv_n;
        s[g] = 0.f;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            float q;
            if constexpr (VEC) {
                const float a = v[g][k].x - mean[g], b = v[g][k].y - mean[g], cc = v[g][k].z - mean[g], d = v[g][k].w - mean[g];
                q = (a * a + b * b) + (cc * cc + d * d);
            } else {
                const float a = v[g][k] - mean[g];
                q = a * a;
            }
            s[g] += ok[k] ? q : 0.f;
        }
    }
    bnf_block_sums<GMAX>(s, red);
#pragma unroll
    for (int g = 0; g < GMAX; ++g) {
        var[g] = s[g] * inv_n;
        invstd[g] = rsqrtf(var[g] + eps);
    }
    if (threadIdx.x == 0) {
        if (n_updates_dev) n_updates = *n_updates_dev;
        float rm = running_mean ? running_mean[c] : 0.f, rv = running_mean ? running_var[c] : 0.f;
        const float unb = sh.n > 1 ? (float)sh.n / (float)(sh.n - 1) : 1.f;
#pragma unroll
        for (int g = 0; g < GMAX; ++g) {
            if (g < sh.G) {
                save_mean[g * sh.C + c] = mean[g];
                save_invstd[g * sh.C + c] = invstd[g];
                for (int u = 0; u < n_updates; ++u) {
                    rm = (1.f - momentum) * rm + momentum * mean[g];
                    rv = (1.f - momentum) * rv + momentum * (var[g] * unb);
                }
            }
        }
        if (running_mean) { running_mean[c] = rm; running_var[c] = rv; }
    }
    if (!y) return;                                        // statistics-only call
    const float ga = gamma[c], be = beta[c];
#pragma unroll
    for (int g = 0; g < GMAX; ++g) {
        if (g >= sh.G) break;
        auto one = [&](float a) {
            const float h = ga * ((a - mean[g]) * invstd[g]) + be;     // same expression as the backward's
            return swish ? swishf_(h) : h;
        };
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            if (!ok[k]) continue;
            if constexpr (VEC) st4(y, g * gstride + off[k], make_float4(one(v[g][k].x), one(v[g][k].y), one(v[g][k].z), one(v[g][k].w)));
            else y[g * gstride + off[k]] = one(v[g][k]);
        }
    }
}

template <bool VEC, int GMAX, int KMAX>
__global__ __launch_bounds__(BNF_THREADS) void bn_fused_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ x,
                                                                   const float *gamma, const float *beta,
                                                                   const float *save_mean, const float *save_invstd,
                                                                   float *__restrict__ dx, float *dgamma, float *dbeta,
                                                                   BnShape sh, int swish, int accumulate) {
    typedef typename BnfVal<VEC>::T V;
    __shared__ float red[BNF_WAVES * 2 * GMAX];
    const int c = blockIdx.x;
    const size_t gstride = (size_t)sh.B * sh.C * sh.HW;
    int off[KMAX];
    bool ok[KMAX];
    bnf_units<VEC, KMAX>(sh, c, off, ok);
    V xh[GMAX][KMAX], dh[GMAX][KMAX];       // loaded as x / dy, turned into xhat / dy * swish'(h) in place
#pragma unroll
    for (int g = 0; g < GMAX; ++g)
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const size_t o = (g < sh.G ? g : 0) * gstride + off[k];
            xh[g][k] = *reinterpret_cast<const V *>(x + o);
            dh[g][k] = *reinterpret_cast<const V *>(dy + o);
        }
    const float ga = gamma[c], be = beta[c];
    float s[2 * GMAX];                      // (sum dh, sum dh * xhat) per group
#pragma unroll
    for (int g = 0; g < GMAX; ++g) {
        const int gg = g < sh.G ? g : 0;
        const float mean = save_mean[gg * sh.C + c], invstd = save_invstd[gg * sh.C + c];
        float s1 = 0.f, s2 = 0.f;
        auto one = [&](float &xv, float &d, bool live) {
            xv = (xv - mean) * invstd;
            if (swish) d *= swish_grad_(ga * xv + be);
            s1 += live ? d : 0.f;
            s2 += live ? d * xv : 0.f;
        };
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            if constexpr (VEC) {
                one(xh[g][k].x, dh[g][k].x, ok[k]); one(xh[g][k].y, dh[g][k].y, ok[k]);
                one(xh[g][k].z, dh[g][k].z, ok[k]); one(xh[g][k].w, dh[g][k].w, ok[k]);
            } else {
                one(xh[g][k], dh[g][k], ok[k]);
            }
        }
        s[2 * g] = s1; s[2 * g + 1] = s2;
    }
    bnf_block_sums<2 * GMAX>(s, red);
    if (threadIdx.x == 0) {
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int g = 0; g < GMAX; ++g)
            if (g < sh.G) { t1 += s[2 * g]; t2 += s[2 * g + 1]; }
        if (accumulate) { t1 += dbeta[c]; t2 += dgamma[c]; }
        dbeta[c] = t1;
        dgamma[c] = t2;
    }
    const float inv_n = 1.f / (float)sh.n;
#pragma unroll
    for (int g = 0; g < GMAX; ++g) {
        if (g >= sh.G) break;
        const float kf = ga * save_invstd[g * sh.C + c];
        const float m1 = s[2 * g] * inv_n, m2 = s[2 * g + 1] * inv_n;
        auto one = [&](float xv, float d) { return kf * (d - m1 - xv * m2); };
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            if (!ok[k]) continue;
            if constexpr (VEC) st4(dx, g * gstride + off[k], make_float4(one(xh[g][k].x, dh[g][k].x), one(xh[g][k].y, dh[g][k].y),
                                                                         one(xh[g][k].z, dh[g][k].z), one(xh[g][k].w, dh[g][k].w)));
            else dx[g * gstride + off[k]] = one(xh[g][k], dh[g][k]);
        }
    }
}

// ---- BatchNorm1d (HW = 1): x is [G*B, C] and a channel is a COLUMN.  A block owns 16 neighbouring columns of all
//      groups: thread (tx = column, ty = row class) reads rows ty, ty + 64, ... (64-byte segments per row; the
//      per-channel kernels read one float per 2-KB row and every block touched every line of x), column sums go
//      through LDS over the 64 row classes in a fixed order.
constexpr int BN1_COLS = 16, BN1_TY = BNF_THREADS / BN1_COLS, BN1_MAX_ROWS = 8;      // B <= 512 rows per group

// totals of `NV` values per column: part[v] summed over the 64 row classes; red: NV * 64 * 16 floats
template <int NV>
__device__ __forceinline__ void bn1_col_sums(float (&part)[NV], float *red, float *tot) {
    const int tx = threadIdx.x & (BN1_COLS - 1), ty = threadIdx.x / BN1_COLS;
    __syncthreads();
#pragma unroll
    for (int v = 0; v < NV; ++v) red[(v * BN1_TY + ty) * BN1_COLS + tx] = part[v];
    __syncthreads();
    if (ty < NV) {
        float t = 0.f;
        for (int r = 0; r < BN1_TY; ++r) t += red[(ty * BN1_TY + r) * BN1_COLS + tx];
        tot[ty * BN1_COLS + tx] = t;
    }
    __syncthreads();
#pragma unroll
    for (int v = 0; v < NV; ++v) part[v] = tot[v * BN1_COLS + tx];
}

template <int GMAX>
__global__ __launch_bounds__(BNF_THREADS) void bn1d_fused_fwd_kernel(const float *__restrict__ x, const float *gamma,
                                                                     const float *beta, float *__restrict__ y,
                                                                     float *save_mean, float *save_invstd,
                                                                     float *running_mean, float *running_var, BnShape sh,
                                                                     float eps, float momentum, int n_updates,
                                                                     const int *n_updates_dev, int swish) {
    __shared__ float red[GMAX * BN1_TY * BN1_COLS];
    __shared__ float tot[GMAX * BN1_COLS];
    const int tx = threadIdx.x & (BN1_COLS - 1), ty = threadIdx.x / BN1_COLS;
    const int c = blockIdx.x * BN1_COLS + tx;
    const bool cok = c < sh.C;
    const int cc = cok ? c : sh.C - 1;
    float v[GMAX][BN1_MAX_ROWS];
#pragma unroll
    for (int g = 0; g < GMAX; ++g)
#pragma unroll
        for (int k = 0; k < BN1_MAX_ROWS; ++k) {
            const int r = ty + k * BN1_TY;
            v[g][k] = x[((size_t)(g < sh.G ? g : 0) * sh.B + (r < sh.B ? r : 0)) * sh.C + cc];
        }
    float s[GMAX], mean[GMAX], var[GMAX], invstd[GMAX];
#pragma unroll
    for (int g = 0; g < GMAX; ++g) {
        s[g] = 0.f;
#pragma unroll
        for (int k = 0; k < BN1_MAX_ROWS; ++k) s[g] += (ty + k * BN1_TY < sh.B) ? v[g][k] : 0.f;
    }
    bn1_col_sums<GMAX>(s, red, tot);
    const float inv_n = 1.f / (float)sh.B;
#pragma unroll
    for (int g = 0; g < GMAX; ++g) {
        mean[g] = s[g] * inv_n;
        s[g] = 0.f;
#pragma unroll
        for (int k = 0; k < BN1_MAX_ROWS; ++k) {
            const float a = v[g][k] - mean[g];
            s[g] += (ty + k * BN1_TY < sh.B) ? a * a : 0.f;
        }
    }
    bn1_col_sums<GMAX>(s, red, tot);
#pragma unroll
    for (int g = 0; g < GMAX; ++g) {
        var[g] = s[g] * inv_n;
        invstd[g] = rsqrtf(var[g] + eps);
    }
    if (ty == 0 && cok) {
        if (n_updates_dev) n_updates = *n_updates_dev;
        float rm = running_mean ? running_mean[c] : 0.f, rv = running_mean ? running_var[c] : 0.f;
        const float unb = sh.B > 1 ? (float)sh.B / (float)(sh.B - 1) : 1.f;
#pragma unroll
        for (int g = 0; g < GMAX; ++g) {
            if (g < sh.G) {
                save_mean[g * sh.C + c] = mean[g];
                save_invstd[g * sh.C + c] = invstd[g];
                for (int u = 0; u < n_updates; ++u) {
                    rm = (1.f - momentum) * rm + momentum * mean[g];
                    rv = (1.f - momentum) * rv + momentum * (var[g] * unb);
                }
            }
        }
        if (running_mean) { running_mean[c] = rm; running_var[c] = rv; }
    }
    if (!y || !cok) return;
    const float ga = gamma[c], be = beta[c];
#pragma unroll
    for (int g = 0; g < GMAX; ++g) {
        if (g >= sh.G) break;
#pragma unroll
        for (int k = 0; k < BN1_MAX_ROWS; ++k) {
            const int r = ty + k * BN1_TY;
            if (r >= sh.B) continue;
            const float h = ga * ((v[g][k] - mean[g]) * invstd[g]) + be;
            y[((size_t)g * sh.B + r) * sh.C + c] = swish ? swishf_(h) : h;
        }
    }
}

template <int GMAX>
__global__ __launch_bounds__(BNF_THREADS) void bn1d_fused_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ x,
                                                                     const float *gamma, const float *beta,
                                                                     const float *save_mean, const float *save_invstd,
                                                                     float *__restrict__ dx, float *dgamma, float *dbeta,
                                                                     BnShape sh, int swish, int accumulate) {
    __shared__ float red[2 * GMAX * BN1_TY * BN1_COLS];
    __shared__ float tot[2 * GMAX * BN1_COLS];
    const int tx = threadIdx.x & (BN1_COLS - 1), ty = threadIdx.x / BN1_COLS;
    const int c = blockIdx.x * BN1_COLS + tx;
    const bool cok = c < sh.C;
    const int cc = cok ? c : sh.C - 1;
    float xh[GMAX][BN1_MAX_ROWS], dh[GMAX][BN1_MAX_ROWS];
#pragma unroll
    for (int g = 0; g < GMAX; ++g)
#pragma unroll
        for (int k = 0; k < BN1_MAX_ROWS; ++k) {
            const int r = ty + k * BN1_TY;
            const size_t o = ((size_t)(g < sh.G ? g : 0) * sh.B + (r < sh.B ? r : 0)) * sh.C + cc;
            xh[g][k] = x[o];
            dh[g][k] = dy[o];
        }
    const float ga = gamma[cc], be = beta[cc];
    float s[2 * GMAX];
#pragma unroll
    for (int g = 0; g < GMAX; ++g) {
        const int gg = g < sh.G ? g : 0;
        const float mean = save_mean[gg * sh.C + cc], invstd = save_invstd[gg * sh.C + cc];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < BN1_MAX_ROWS; ++k) {
            const bool live = ty + k * BN1_TY < sh.B;
            xh[g][k] = (xh[g][k] - mean) * invstd;
            if (swish) dh[g][k] *= swish_grad_(ga * xh[g][k] + be);
            s1 += live ? dh[g][k] : 0.f;
            s2 += live ? dh[g][k] * xh[g][k] : 0.f;
        }
        s[2 * g] = s1; s[2 * g + 1] = s2;
    }
    bn1_col_sums<2 * GMAX>(s, red, tot);
    if (ty == 0 && cok) {
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int g = 0; g < GMAX; ++g)
            if (g < sh.G) { t1 += s[2 * g]; t2 += s[2 * g + 1]; }
        if (accumulate) { t1 += dbeta[c]; t2 += dgamma[c]; }
        dbeta[c] = t1;
        dgamma[c] = t2;
    }
    if (!cok) return;
    const float inv_n = 1.f / (float)sh.B;
#pragma unroll
    for (int g = 0; g < GMAX; ++g) {
        if (g >= sh.G) break;
        const float kf = ga * save_invstd[g * sh.C + c];
        const float m1 = s[2 * g] * inv_n, m2 = s[2 * g + 1] * inv_n;
#pragma unroll
        for (int k = 0; k < BN1_MAX_ROWS; ++k) {
            const int r = ty + k * BN1_TY;
            if (r >= sh.B) continue;
            dx[((size_t)g * sh.B + r) * sh.C + c] = kf * (dh[g][k] - m1 - xh[g][k] * m2);
        }
    }
}

// ---- slices of 16 K .. 64 K elements in MANY groups (celeba19's 18-group statistics-only pass over the 16x16 maps):
//      grid (C, G), block (c, g) keeps ITS slice in registers between the statistics and the apply -- one launch, one
//      read -- and leaves (mean, variance) in the workspace; the running statistics (which couple the groups of a
//      channel, in group order) are a one-thread-per-channel launch behind it.
constexpr int BNS_KMAX = 16;                // float4 per thread: 1024 threads x 16 x 4 = 65,536 elements
constexpr int BNS_MAX_N = BNF_THREADS * BNS_KMAX * 4;

__global__ __launch_bounds__(BNF_THREADS) void bn_slice_fwd_kernel(const float *__restrict__ x, const float *gamma,
                                                                   const float *beta, float *__restrict__ y,
                                                                   float *save_mean, float *save_invstd, float *stats,
                                                                   BnShape sh, float eps, int swish) {
    __shared__ float red[BNF_WAVES];
    const int c = blockIdx.x, g = blockIdx.y;
    const size_t gbase = (size_t)g * sh.B * sh.C * sh.HW;
    // HW / 4 divides the block (host check): thread t owns float4 t % (HW/4) of rows t / (HW/4) + k * rows_per_pass --
    // offsets advance by a constant, nothing to keep per unit
    const int hw4 = sh.HW >> 2, rpp = BNF_THREADS / hw4, b0 = threadIdx.x / hw4;
    const int off0 = (b0 * sh.C + c) * sh.HW + 4 * (threadIdx.x - b0 * hw4), dk = rpp * sh.C * sh.HW;
    const int safe = c * sh.HW + 4 * (threadIdx.x - b0 * hw4);   // row 0 of the slice: where a unit past the batch reads (masked out of the sums)
#define BNS_OK(k) (b0 + (k) * rpp < sh.B)
#define BNS_OFF(k) (gbase + (BNS_OK(k) ? off0 + (k) * dk : safe))
    float4 v[BNS_KMAX];
#pragma unroll
    for (int k = 0; k < BNS_KMAX; ++k) v[k] = ld4(x, BNS_OFF(k));
    float s[1] = {0.f};
#pragma unroll
    for (int k = 0; k < BNS_KMAX; ++k) s[0] += BNS_OK(k) ? (v[k].x + v[k].y) + (v[k].z + v[k].w) : 0.f;
    bnf_block_sums<1>(s, red);
    const float inv_n = 1.f / (float)sh.n;
    const float mean = s[0] * inv_n;
    s[0] = 0.f;
#pragma unroll
    for (int k = 0; k < BNS_KMAX; ++k) {
        const float a = v[k].x - mean, b = v[k].y - mean, cc = v[k].z - mean, d = v[k].w - mean;
        s[0] += BNS_OK(k) ? (a * a + b * b) + (cc * cc + d * d) : 0.f;
    }
    bnf_block_sums<1>(s, red);
    const float var = s[0] * inv_n, invstd = rsqrtf(var + eps);
    if (threadIdx.x == 0) {
        save_mean[g * sh.C + c] = mean;
        save_invstd[g * sh.C + c] = invstd;
        stats[(g * sh.C + c) * 2] = mean;
        stats[(g * sh.C + c) * 2 + 1] = var;
    }
    if (!y) return;
    const float ga = gamma[c], be = beta[c];
    auto one = [&](float a) {
        const float h = ga * ((a - mean) * invstd) + be;
        return swish ? swishf_(h) : h;
    };
#pragma unroll
    for (int k = 0; k < BNS_KMAX; ++k)
        if (BNS_OK(k)) st4(y, BNS_OFF(k), make_float4(one(v[k].x), one(v[k].y), one(v[k].z), one(v[k].w)));
}

// running statistics of every channel from stats[G][C][2], groups in order, n_updates times each
__global__ __launch_bounds__(256) void bn_running_kernel(const float *stats, int G, int C, int n, float *running_mean,
                                                         float *running_var, float momentum, int n_updates,
                                                         const int *n_updates_dev) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    if (n_updates_dev) n_updates = *n_updates_dev;
    const float unb = n > 1 ? (float)n / (float)(n - 1) : 1.f;
    float rm = running_mean[c], rv = running_var[c];
    for (int g = 0; g < G; ++g) {
        const float mean = stats[(g * C + c) * 2], var = stats[(g * C + c) * 2 + 1];
        for (int u = 0; u < n_updates; ++u) {
            rm = (1.f - momentum) * rm + momentum * mean;
            rv = (1.f - momentum) * rv + momentum * (var * unb);
        }
    }
    running_mean[c] = rm;
    running_var[c] = rv;
}

#undef BNS_OK
#undef BNS_OFF

// ---- statistics from the records a statistics-only conv launch left (mvae_convT2d_k4_fwd_stats: (mean, M2) per
//      column tile and channel over `elems` elements each, tiles of a group contiguous): per (group, channel) the
//      equal-count merge  mean = avg(mean_i),  M2 = sum(M2_i) + elems * sum((mean_i - mean)^2);  then the running
//      statistics advance over the groups in order -- what mvae_bn_train_fwd(y = NULL) does from a sweep of the
//      activations that are no longer written.  One block per channel, a wave per group (in turns).
__global__ __launch_bounds__(BNF_THREADS) void bn_stats_merge_kernel(const float *__restrict__ part, int tiles_per_group,
                                                                     int elems, int G, int C, float *save_mean,
                                                                     float *save_invstd, float *running_mean,
                                                                     float *running_var, float eps, float momentum,
                                                                     int n_updates, const int *n_updates_dev) {
    extern __shared__ float gm[];                      // [G][2]: mean, biased variance
    const int c = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int g = wave; g < G; g += BNF_WAVES) {
        const float *p = part + ((size_t)g * tiles_per_group * C + c) * 2;
        float s = 0.f;
        for (int i = lane; i < tiles_per_group; i += 64) s += p[(size_t)i * C * 2];
        const float mean = wave_sum(s) / (float)tiles_per_group;
        float m2 = 0.f;
        for (int i = lane; i < tiles_per_group; i += 64) {
            const float d = p[(size_t)i * C * 2] - mean;
            m2 += p[(size_t)i * C * 2 + 1] + (float)elems * d * d;
        }
        m2 = wave_sum(m2);
        if (lane == 0) { gm[g * 2] = mean; gm[g * 2 + 1] = m2 / ((float)tiles_per_group * (float)elems); }
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    if (n_updates_dev) n_updates = *n_updates_dev;
    const float n = (float)tiles_per_group * (float)elems;
    const float unb = n > 1.f ? n / (n - 1.f) : 1.f;
    float rm = running_mean ? running_mean[c] : 0.f, rv = running_mean ? running_var[c] : 0.f;
    for (int g = 0; g < G; ++g) {
        const float mean = gm[g * 2], var = gm[g * 2 + 1];
        if (save_mean) save_mean[g * C + c] = mean;
        if (save_invstd) save_invstd[g * C + c] = rsqrtf(var + eps);
        for (int u = 0; u < n_updates; ++u) {
            rm = (1.f - momentum) * rm + momentum * mean;
            rv = (1.f - momentum) * rv + momentum * (var * unb);
        }
    }
    if (running_mean) { running_mean[c] = rm; running_var[c] = rv; }
}

#ifndef MVAE_BN_FUSED
#define MVAE_BN_FUSED 1         // 0: every layer on the two-launch path (A/B builds)
#endif
#ifndef MVAE_BN_SLICE
#define MVAE_BN_SLICE 1         // 0: slices of 16 K .. 64 K elements / many groups stay on the two-launch path (A/B builds)
#endif
constexpr int BNF_GMAX_FWD = 3, BNF_GMAX_BWD = 2, BN1_GMAX = 3;

// which single-launch form takes the shape: 0 none, 1 spatial, 2 BatchNorm1d
inline int bn_fused_kind(const BnShape &sh, bool bwd) {
    if (!MVAE_BN_FUSED || (long)sh.B * sh.C * sh.HW >= (1L << 30)) return 0;
    if (sh.HW == 1) return (sh.G <= BN1_GMAX && sh.B <= BN1_TY * BN1_MAX_ROWS) ? 2 : 0;
    // (unaligned / odd-width maps keep one element per register: half the slice when several groups share the block)
    const int max_n = (!sh.vec && sh.G > 1) ? BNF_MAX_N / 2 : BNF_MAX_N;
    if (sh.n <= max_n && sh.G <= (bwd ? BNF_GMAX_BWD : BNF_GMAX_FWD)) return 1;
    // 3 (forward only): one block per (channel, group) slice + a per-channel launch for the running statistics.  Only
    // where it measured faster than the two-launch path (profiles/r04_*_by_shape.txt, session 2 vs final): slices of
    // 16 K .. 64 K elements with >= 512 of them -- celeba19's 18-group pass over the 16x16 maps, 181 -> 129 us.  With
    // 64 .. 128 slices (CelebA's own 16x16 layers) the 1024-thread blocks leave most CUs idle (13.8 -> 19.4 us), at
    // 16 K elements the two-launch path is already at 6 TB/s (75 -> 86 us), and the backward form (dh kept, x re-read)
    // lost everywhere it was tried (18.4 -> 34.4 us at 64 slices) and was removed.
    return (!bwd && MVAE_BN_SLICE && sh.vec && sh.n > BNF_MAX_N && sh.n <= BNS_MAX_N && (long)sh.C * sh.G >= 512 &&
            sh.G <= 65535 && (sh.HW >> 2) <= BNF_THREADS && BNF_THREADS % (sh.HW >> 2) == 0) ? 3 : 0;
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
    hipStream_t st = (hipStream_t)stream;
    const int sw = (flags & MVAE_ACT_SWISH) ? 1 : 0;
    if (bn_fused_kind(sh, false) == 3) {
        if (!ws || ws_bytes < (size_t)G * C * 2 * sizeof(float)) return MVAE_ERR_WS;
        hipLaunchKernelGGL(bn_slice_fwd_kernel, dim3(C, G), dim3(BNF_THREADS), 0, st, x, gamma, beta, y, save_mean,
                           save_invstd, (float *)ws, sh, eps, sw);
        if (running_mean)
            hipLaunchKernelGGL(bn_running_kernel, dim3((C + 255) / 256), dim3(256), 0, st, (const float *)ws, G, C, sh.n,
                               running_mean, running_var, momentum, n_updates, n_updates_dev);
        return mvae_launch_status();
    }
    if (const int kind = bn_fused_kind(sh, false)) {
#define MVAE_BNF_FWD(KERN, GRID)                                                                                    \
        hipLaunchKernelGGL(KERN, dim3(GRID), dim3(BNF_THREADS), 0, st, x, gamma, beta, y, save_mean, save_invstd,   \
                           running_mean, running_var, sh, eps, momentum, n_updates, n_updates_dev, sw)
        if (kind == 2) {
            if (G == 1) MVAE_BNF_FWD((bn1d_fused_fwd_kernel<1>), (C + BN1_COLS - 1) / BN1_COLS);
            else MVAE_BNF_FWD((bn1d_fused_fwd_kernel<BN1_GMAX>), (C + BN1_COLS - 1) / BN1_COLS);
        } else if (sh.vec) {
            if (G == 1) MVAE_BNF_FWD((bn_fused_fwd_kernel<true, 1, 4>), C);
            else if (G == 2) MVAE_BNF_FWD((bn_fused_fwd_kernel<true, 2, 4>), C);
            else MVAE_BNF_FWD((bn_fused_fwd_kernel<true, BNF_GMAX_FWD, 4>), C);
        } else {
            if (G == 1) MVAE_BNF_FWD((bn_fused_fwd_kernel<false, 1, 16>), C);
            else MVAE_BNF_FWD((bn_fused_fwd_kernel<false, BNF_GMAX_FWD, 8>), C);
        }
#undef MVAE_BNF_FWD
        return mvae_launch_status();
    }
    if (!ws || ws_bytes < mvae_bn_ws_bytes(G, C, sh.n)) return MVAE_ERR_WS;
    dim3 grid(sh.S, C, G);
    hipLaunchKernelGGL(bn_partial_stats_kernel, grid, dim3(BN_THREADS), 0, st, x, (float *)ws, sh);
    // y == NULL: only the (s = 0) block of each (channel, group) has work -- saved + running statistics
    const dim3 agrid(y ? sh.S : 1, C, G);
    hipLaunchKernelGGL(bn_fwd_apply_kernel, agrid, dim3(BN_THREADS), 0, st, x, gamma, beta, y, (const float *)ws,
                       save_mean, save_invstd, running_mean, running_var, sh, eps, momentum, n_updates,
                       n_updates_dev, (flags & MVAE_ACT_SWISH) ? 1 : 0);
    return mvae_launch_status();
}

MVAE_EXPORT int mvae_bn_stats_merge(const float *part, int tiles, int elems_per_tile, int G, int C, float *save_mean,
                                    float *save_invstd, float *running_mean, float *running_var, float eps,
                                    float momentum, int n_updates, const int *n_updates_dev, mvae_stream_t stream) {
    if (!part || tiles <= 0 || elems_per_tile <= 0 || G <= 0 || C <= 0 || tiles % G != 0 || G > 4096) return MVAE_ERR_ARG;
    if ((running_mean == nullptr) != (running_var == nullptr)) return MVAE_ERR_ARG;
    hipLaunchKernelGGL(bn_stats_merge_kernel, dim3(C), dim3(BNF_THREADS), (size_t)G * 2 * sizeof(float), (hipStream_t)stream,
                       part, tiles / G, elems_per_tile, G, C, save_mean, save_invstd, running_mean, running_var, eps,
                       momentum, n_updates, n_updates_dev);
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
    hipStream_t st = (hipStream_t)stream;
    const int swish = (flags & MVAE_ACT_SWISH) ? 1 : 0;
    if (const int kind = bn_fused_kind(sh, true)) {
        const int acc = (flags & MVAE_ACCUMULATE) ? 1 : 0;
#define MVAE_BNF_BWD(KERN, GRID)                                                                                    \
        hipLaunchKernelGGL(KERN, dim3(GRID), dim3(BNF_THREADS), 0, st, dy, x, gamma, beta, save_mean, save_invstd,  \
                           dx, dgamma, dbeta, sh, swish, acc)
        if (kind == 2) {
            if (G == 1) MVAE_BNF_BWD((bn1d_fused_bwd_kernel<1>), (C + BN1_COLS - 1) / BN1_COLS);
            else MVAE_BNF_BWD((bn1d_fused_bwd_kernel<BN1_GMAX>), (C + BN1_COLS - 1) / BN1_COLS);
        } else if (sh.vec) {
            if (G == 1) MVAE_BNF_BWD((bn_fused_bwd_kernel<true, 1, 4>), C);
            else MVAE_BNF_BWD((bn_fused_bwd_kernel<true, BNF_GMAX_BWD, 4>), C);
        } else {
            if (G == 1) MVAE_BNF_BWD((bn_fused_bwd_kernel<false, 1, 16>), C);
            else MVAE_BNF_BWD((bn_fused_bwd_kernel<false, BNF_GMAX_BWD, 8>), C);
        }
#undef MVAE_BNF_BWD
        return mvae_launch_status();
    }
    if (!ws || ws_bytes < mvae_bn_ws_bytes(G, C, sh.n)) return MVAE_ERR_WS;
    dim3 grid(sh.S, C, G);
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
