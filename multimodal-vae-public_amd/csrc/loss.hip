// loss.hip -- the reconstruction terms of the ELBO and their gradients (HBM-bound).
//
//   BCE-with-logits row sums   mnist/train.py:47-49,62-74; celeba/train.py:50-58,68-80;
//                              celeba19/train.py:52-57,63-75
//   categorical cross-entropy  mnist/train.py:52,77-94
//   ELBO combine / batch mean  mnist/train.py:57-58,214; celeba19/train.py:59
//
// The reference evaluates the BCE as 7 elementwise ATen kernels + a row reduction and
// autograd replays ~10 more; here one pass reads logits+target once and emits the row sum
// and (optionally, since d loss / d rowsum is a known constant lambda/B) the gradient.
#include "common.h"

namespace {

// bce_elem / bce_grad: common.h (shared with the Linear epilogue that folds this term, gemm_core.h EpRowBce)

struct BceArgs {
    const float *logits, *target, *colw, *drow;
    float *rowsum, *dlogits;
    int R, P, rows_per_group, target_rows;
    int target_div, t_rs, t_cs;   // target row = (r / target_div) % target_rows; element (row, p) at row*t_rs + p*t_cs
};

// One block per row (wide rows: pixels).  Optionally writes the gradient in the same pass.
__global__ __launch_bounds__(256) void bce_row_block_kernel(BceArgs a) {
    __shared__ float red[16];
    const int r = blockIdx.x;
    const int g = r / a.rows_per_group;
    const float *x = a.logits + (size_t)r * a.P;
    const float *t = a.target + (size_t)((r / a.target_div) % a.target_rows) * a.t_rs;
    const float *w = a.colw ? a.colw + (size_t)g * a.P : nullptr;
    float *dx = a.dlogits ? a.dlogits + (size_t)r * a.P : nullptr;
    const float dr = (dx && a.drow) ? a.drow[g] : 0.f;
    float s = 0.f;
    const bool vec = (a.P % 4 == 0) && a.t_cs == 1 && aligned16_dev(x) && aligned16_dev(t) && (!w || aligned16_dev(w)) &&
                     (!dx || aligned16_dev(dx));
    if (vec) {
        const int p4 = a.P / 4;
        // FOUR float4 groups per trip, all their loads issued before the first is used: one block per row leaves 512 - 2048
        // blocks of 256 threads, and with one dependent load -> exp / log -> store chain per thread the launch moved 75 MB at
        // 2.95 TB/s (image BCE of CelebA, 512 x 12288: profiles/r04_celeba_by_shape.txt); the sums keep their order
        // (element i, i + 256, ... of a thread, as before)
        constexpr int U = 4;
        int i = threadIdx.x;
        // (only without column weights -- the image terms: a load under the block-uniform `if (w)` would make hipcc drain the
        //  memory queue behind it and undo the batching; weighted rows take the one-group loop below.  `1.f *` keeps the bits.)
        for (; !w && i + 256 * (U - 1) < p4; i += 256 * U) {
            float4 xv[U], tv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                xv[u] = reinterpret_cast<const float4 *>(x)[i + 256 * u];
                tv[u] = reinterpret_cast<const float4 *>(t)[i + 256 * u];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                s += 1.f * bce_elem(xv[u].x, tv[u].x) + 1.f * bce_elem(xv[u].y, tv[u].y) +
                     1.f * bce_elem(xv[u].z, tv[u].z) + 1.f * bce_elem(xv[u].w, tv[u].w);
                if (dx) {
                    float4 gv;
                    gv.x = dr * 1.f * bce_grad(xv[u].x, tv[u].x);
                    gv.y = dr * 1.f * bce_grad(xv[u].y, tv[u].y);
                    gv.z = dr * 1.f * bce_grad(xv[u].z, tv[u].z);
                    gv.w = dr * 1.f * bce_grad(xv[u].w, tv[u].w);
                    reinterpret_cast<float4 *>(dx)[i + 256 * u] = gv;
                }
            }
        }
        for (; i < p4; i += 256) {
            const float4 xv = reinterpret_cast<const float4 *>(x)[i];
            const float4 tv = reinterpret_cast<const float4 *>(t)[i];
            float4 wv = make_float4(1.f, 1.f, 1.f, 1.f);
            if (w) wv = reinterpret_cast<const float4 *>(w)[i];
            s += wv.x * bce_elem(xv.x, tv.x) + wv.y * bce_elem(xv.y, tv.y) + wv.z * bce_elem(xv.z, tv.z) +
                 wv.w * bce_elem(xv.w, tv.w);
            if (dx) {
                float4 gv;
                gv.x = dr * wv.x * bce_grad(xv.x, tv.x);
                gv.y = dr * wv.y * bce_grad(xv.y, tv.y);
                gv.z = dr * wv.z * bce_grad(xv.z, tv.z);
                gv.w = dr * wv.w * bce_grad(xv.w, tv.w);
                reinterpret_cast<float4 *>(dx)[i] = gv;
            }
        }
    } else {
        for (int i = threadIdx.x; i < a.P; i += 256) {
            const float wi = w ? w[i] : 1.f;
            const float ti = t[(size_t)i * a.t_cs];
            s += wi * bce_elem(x[i], ti);
            if (dx) dx[i] = dr * wi * bce_grad(x[i], ti);
        }
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0 && a.rowsum) a.rowsum[r] = s;
}

// One wave per row (narrow rows: the 18 attributes).
__global__ __launch_bounds__(256) void bce_row_wave_kernel(BceArgs a) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= a.R) return;
    const int g = r / a.rows_per_group;
    const float *x = a.logits + (size_t)r * a.P;
    const float *t = a.target + (size_t)((r / a.target_div) % a.target_rows) * a.t_rs;
    const float *w = a.colw ? a.colw + (size_t)g * a.P : nullptr;
    float *dx = a.dlogits ? a.dlogits + (size_t)r * a.P : nullptr;
    const float dr = (dx && a.drow) ? a.drow[g] : 0.f;
    float s = 0.f;
    for (int i = lane; i < a.P; i += 64) {
        const float wi = w ? w[i] : 1.f;
        const float ti = t[(size_t)i * a.t_cs];
        s += wi * bce_elem(x[i], ti);
        if (dx) dx[i] = dr * wi * bce_grad(x[i], ti);
    }
    s = wave_sum(s);
    if (lane == 0 && a.rowsum) a.rowsum[r] = s;
}

// -log_softmax(x + 1e-6)[label]  and its gradient  drow * (softmax(x + 1e-6) - onehot)
__global__ __launch_bounds__(256) void ce_kernel(const float *logits, const int64_t *label, const float *drow,
                                                 float *row, float *dlogits, int R, int K, int rows_per_group,
                                                 int label_rows) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= R) return;
    const float *x = logits + (size_t)r * K;
    const int64_t yraw = label[r % label_rows];
    // a label outside 0..K-1 (the reference fails in scatter_, mnist/train.py:86-88) must not index the
    // logits row: its loss row and gradient row become NaN, which the step's ELBO then shows
    const bool bad = yraw < 0 || yraw >= K;
    const int y = bad ? 0 : (int)yraw;
    float mx = -INFINITY;
    for (int k = 0; k < K; ++k) mx = fmaxf(mx, x[k] + 1e-6f);
    float se = 0.f;
    for (int k = 0; k < K; ++k) se += expf(x[k] + 1e-6f - mx);
    const float lse = logf(se) + mx;
    if (row) row[r] = bad ? NAN : -((x[y] + 1e-6f) - lse);
    if (dlogits) {
        const float dr = bad ? NAN : drow[r / rows_per_group];
        for (int k = 0; k < K; ++k) {
            const float p = expf(x[k] + 1e-6f - lse);
            dlogits[(size_t)r * K + k] = dr * (p - (k == y ? 1.f : 0.f));
        }
    }
}

// out[g] (+)= coef[g] * sum(rows of group g);  out[G] (+)= sum over g of the same
__global__ __launch_bounds__(1024) void group_sums_kernel(const float *rows, const float *coef, float *out, float *total_out, int G,
                                                          int rows_per_group, int accumulate) {
    __shared__ float red[16];
    float total = 0.f;
    for (int g = 0; g < G; ++g) {
        float s = 0.f;
        for (int i = threadIdx.x; i < rows_per_group; i += 1024) s += rows[(size_t)g * rows_per_group + i];
        s = block_sum(s, red) * (coef ? coef[g] : 1.f);
        total += s;
        if (threadIdx.x == 0 && out) out[g] = accumulate ? out[g] + s : s;
    }
    if (threadIdx.x == 0 && total_out) *total_out = accumulate ? *total_out + total : total;
}

// ---- the ELBO of a fused step in ONE launch (mnist/train.py:57-58,214; celeba19/train.py:59,265-302) ----
// elbo[t] = sum over the parts that feed term t of coef * (sum of the part's rows of that term); elbo[T] = the
// step's total, added up part by part in the order given (the order the separate group-sum launches had).
// Block 0 does the sums and advances the step's Philox counter; every block helps clear `zero` (the shared
// latent-gradient buffer the decoders' first layers accumulate into).
struct ElboParts { mvae_elbo_part p[MVAE_ELBO_MAX_PARTS]; int n; };
constexpr int ELBO_MAX_GROUPS = 1024;      // all parts' groups together: one thread of block 0 each

__global__ __launch_bounds__(1024) void elbo_reduce_kernel(ElboParts parts, float *elbo, int T, float *zero, size_t zero_n,
                                                           uint64_t *counter, uint64_t counter_inc) {
    const size_t zstride = (size_t)gridDim.x * 1024;
    if (zero) {
        const size_t z4 = aligned16_dev(zero) ? zero_n / 4 : 0;
        for (size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i < z4; i += zstride)
            reinterpret_cast<float4 *>(zero)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (size_t i = z4 * 4 + (size_t)blockIdx.x * 1024 + threadIdx.x; i < zero_n; i += zstride) zero[i] = 0.f;
    }
    if (blockIdx.x != 0) return;
    // every (part, group) sum is one wave's job (16 waves take them round-robin: lane-strided partial sums in a
    // fixed order, then the wave reduction), parked in LDS; thread 0 then adds them up part by part, group by
    // group -- one barrier instead of two per group (7 groups of 512 rows: 13 -> 4 us on the MNIST step's
    // critical path).  A LONG group (the per-32-column partials of a folded Bernoulli term: 512 rows x 25) is
    // spread over all 16 waves instead -- one wave would walk it as 200 dependent round trips.
    constexpr int LONG_GROUP = 2048;
    __shared__ float gsum[MVAE_ELBO_MAX_PARTS * MVAE_ELBO_MAX_TERMS];
    __shared__ float wsum[MVAE_ELBO_MAX_PARTS * MVAE_ELBO_MAX_TERMS][16];
    __shared__ float acc[MVAE_ELBO_MAX_TERMS + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // the weights of the group sums (coefficient, term index, ready-made sums) do not depend on the row sums: requested
    // FIRST, so that their round trip runs beside the row sums' instead of behind it
    // (unconditional loads from always-legal addresses: a load under a divergent branch is waited for at the branch's end)
    int my_slot = -1, my_first = 0;
    bool mine = false;
    const float *safe = parts.p[0].rows;
    const float *cp = nullptr, *rp = nullptr;
    const int *tp = nullptr;
    {
        int idx = 0, sl = 0;
        for (int q = 0; q < parts.n; ++q) {
            const mvae_elbo_part &p = parts.p[q];
            const int g = (int)threadIdx.x - idx;
            if (g >= 0 && g < p.groups) {
                mine = true;
                if (p.rows_per_group == 1) rp = p.rows + g; else my_slot = sl + g;
                cp = p.coef ? p.coef + g : nullptr;
                tp = p.term_of ? p.term_of + g : nullptr;
                my_first = p.first_term + g;
            }
            idx += p.groups;
            if (p.rows_per_group != 1) sl += p.groups;
        }
    }
    const float ld_coef = *(cp ? cp : safe), ld_ready = *(rp ? rp : safe);
    const int ld_term = *(tp ? tp : reinterpret_cast<const int *>(safe));
    const float my_coef = cp ? ld_coef : 1.f, my_ready = rp ? ld_ready : 0.f;
    const int my_term = tp ? ld_term : my_first;
    // Row sums.  A lane's elements are requested EIGHT at a time (clamped addresses, the surplus multiplied away) and
    // added in index order: one memory round trip per 512 rows of a wave instead of one per 64 -- the loop used to be a
    // chain of dependent loads, ~1 us each on this launch's single block (11.6 us on the MNIST step's critical chain).
    auto lane_sum = [&](const float *r, int lo, int hi) {       // sum over i = lo + lane, lo + lane + 64, ... < hi
        float s = 0.f;
        for (int i0 = lo + lane; i0 < hi; i0 += 512) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + 64 * u;
                v[u] = r[min(i, hi - 1)] * (i < hi ? 1.f : 0.f);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        return s;
    };
    int slot = 0, turn = 0;
    bool any_long = false;
    for (int q = 0; q < parts.n; ++q) {
        const mvae_elbo_part &p = parts.p[q];
        if (p.rows_per_group == 1) continue;         // a table of ready sums: read above
        const bool spread = p.rows_per_group > LONG_GROUP;
        any_long |= spread;
        for (int g = 0; g < p.groups; ++g, ++slot) {
            const float *r = p.rows + (size_t)g * p.rows_per_group;
            if (spread) {
                const int chunk = (((p.rows_per_group + 15) / 16) + 63) & ~63;
                const int lo = wave * chunk, hi = min(p.rows_per_group, lo + chunk);
                float s = lo < hi ? lane_sum(r, lo, hi) : 0.f;
                s = wave_sum(s);
                if (lane == 0) wsum[slot][wave] = s;
                continue;
            }
            if ((turn++ & 15) != wave) continue;
            float s = lane_sum(r, 0, p.rows_per_group);
            s = wave_sum(s);
            if (lane == 0) gsum[slot] = s;
        }
    }
    __syncthreads();
    if (any_long) {                                   // block-uniform
        slot = 0;
        for (int q = 0; q < parts.n; ++q) {
            const mvae_elbo_part &p = parts.p[q];
            if (p.rows_per_group == 1) continue;
            for (int g = 0; g < p.groups; ++g, ++slot)
                if (p.rows_per_group > LONG_GROUP && threadIdx.x == (slot & 1023)) {
                    float s = 0.f;
                    for (int w = 0; w < 16; ++w) s += wsum[slot][w];
                    gsum[slot] = s;
                }
        }
        __syncthreads();
    }
    // the weighted group sums, one thread per (part, group)
    __shared__ float val[ELBO_MAX_GROUPS];
    __shared__ int term[ELBO_MAX_GROUPS];
    if (mine) {
        val[threadIdx.x] = (my_slot >= 0 ? gsum[my_slot] : my_ready) * my_coef;
        term[threadIdx.x] = my_term;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int t = 0; t <= T; ++t) acc[t] = 0.f;
        int idx = 0;
        for (int q = 0; q < parts.n; ++q) {           // part by part, group by group: the order of the separate launches
            float part_total = 0.f;
            for (int g = 0; g < parts.p[q].groups; ++g, ++idx) {
                acc[term[idx]] += val[idx];
                part_total += val[idx];
            }
            acc[T] += part_total;
        }
        for (int t = 0; t <= T; ++t) elbo[t] = acc[t];
        if (counter) *counter += counter_inc;
    }
}

int bce_launch(BceArgs a, hipStream_t st) {
    if (!a.logits || !a.target || a.R <= 0 || a.P <= 0 || a.rows_per_group <= 0 || a.target_rows <= 0 ||
        a.target_div <= 0 || a.t_cs <= 0 || a.t_rs <= 0)
        return MVAE_ERR_ARG;
    if (!a.rowsum && !a.dlogits) return MVAE_ERR_ARG;
    if (a.dlogits && !a.drow) return MVAE_ERR_ARG;
    if (a.P >= 512)
        hipLaunchKernelGGL(bce_row_block_kernel, dim3(a.R), dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL(bce_row_wave_kernel, dim3((a.R + 3) / 4), dim3(256), 0, st, a);
    return mvae_launch_status();
}

}  // namespace

MVAE_EXPORT int mvae_bce_rowsum_fwd(const float *logits, const float *target, const float *colw, float *rowsum,
                                    const float *drow_dev, float *dlogits, int R, int P, int rows_per_group,
                                    int target_rows, int target_div, int target_row_stride,
                                    int target_col_stride, mvae_stream_t stream) {
    BceArgs a{logits, target, colw, drow_dev, rowsum, dlogits, R, P, rows_per_group, target_rows,
              target_div, target_row_stride, target_col_stride};
    if (!rowsum) return MVAE_ERR_ARG;
    return bce_launch(a, (hipStream_t)stream);
}

MVAE_EXPORT int mvae_bce_rowsum_bwd(const float *logits, const float *target, const float *colw,
                                    const float *drow_dev, float *dlogits, int R, int P, int rows_per_group,
                                    int target_rows, int target_div, int target_row_stride,
                                    int target_col_stride, mvae_stream_t stream) {
    BceArgs a{logits, target, colw, drow_dev, nullptr, dlogits, R, P, rows_per_group, target_rows,
              target_div, target_row_stride, target_col_stride};
    if (!dlogits) return MVAE_ERR_ARG;
    return bce_launch(a, (hipStream_t)stream);
}

MVAE_EXPORT int mvae_ce_fwd(const float *logits, const int64_t *label, float *row, const float *drow_dev,
                            float *dlogits, int R, int K, int rows_per_group, int label_rows,
                            mvae_stream_t stream) {
    if (!logits || !label || !row || R <= 0 || K <= 0 || K > 1024 || label_rows <= 0 || rows_per_group <= 0)
        return MVAE_ERR_ARG;
    if (dlogits && !drow_dev) return MVAE_ERR_ARG;
    hipLaunchKernelGGL(ce_kernel, dim3((R + 255) / 256), dim3(256), 0, (hipStream_t)stream, logits, label,
                       drow_dev, row, dlogits, R, K, rows_per_group, label_rows);
    return mvae_launch_status();
}

MVAE_EXPORT int mvae_ce_bwd(const float *logits, const int64_t *label, const float *drow_dev, float *dlogits,
                            int R, int K, int rows_per_group, int label_rows, mvae_stream_t stream) {
    if (!logits || !label || !drow_dev || !dlogits || R <= 0 || K <= 0 || K > 1024 || label_rows <= 0 ||
        rows_per_group <= 0)
        return MVAE_ERR_ARG;
    hipLaunchKernelGGL(ce_kernel, dim3((R + 255) / 256), dim3(256), 0, (hipStream_t)stream, logits, label,
                       drow_dev, (float *)nullptr, dlogits, R, K, rows_per_group, label_rows);
    return mvae_launch_status();
}

MVAE_EXPORT int mvae_elbo_reduce(const mvae_elbo_part *parts, int n_parts, float *elbo, int T, float *zero,
                                 size_t zero_n, uint64_t *counter_dev, uint64_t counter_inc, mvae_stream_t stream) {
    if (!parts || n_parts <= 0 || n_parts > MVAE_ELBO_MAX_PARTS || !elbo || T <= 0 || T > MVAE_ELBO_MAX_TERMS)
        return MVAE_ERR_ARG;
    ElboParts ps;
    ps.n = n_parts;
    long total_groups = 0;
    for (int q = 0; q < n_parts; ++q) {
        ps.p[q] = parts[q];
        if (!ps.p[q].rows || ps.p[q].groups <= 0 || ps.p[q].rows_per_group <= 0) return MVAE_ERR_ARG;
        if (!ps.p[q].term_of && (ps.p[q].first_term < 0 || ps.p[q].first_term + ps.p[q].groups > T)) return MVAE_ERR_ARG;
        if (ps.p[q].rows_per_group > 1 && ps.p[q].groups > MVAE_ELBO_MAX_TERMS) return MVAE_ERR_ARG;
        total_groups += ps.p[q].groups;
    }
    if (total_groups > ELBO_MAX_GROUPS) return MVAE_ERR_ARG;
    size_t blocks = zero ? (zero_n / 4 + 1023) / 1024 : 1;
    if (blocks < 1) blocks = 1;
    if (blocks > 256) blocks = 256;
    hipLaunchKernelGGL(elbo_reduce_kernel, dim3((unsigned)blocks), dim3(1024), 0, (hipStream_t)stream, ps, elbo, T, zero,
                       zero ? zero_n : 0, counter_dev, counter_inc);
    return mvae_launch_status();
}

MVAE_EXPORT int mvae_group_sums(const float *rows, const float *coef_dev, float *out, float *total_out, int G,
                                int rows_per_group,
                                int flags, mvae_stream_t stream) {
    if (!rows || (!out && !total_out) || G <= 0 || rows_per_group <= 0) return MVAE_ERR_ARG;
    hipLaunchKernelGGL(group_sums_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, rows, coef_dev, out, total_out, G,
                       rows_per_group, (flags & MVAE_ACCUMULATE) ? 1 : 0);
    return mvae_launch_status();
}
