// linear.hip -- Linear layers (forward, data gradient, weight gradient; single and grouped) on the GEMM
// kernel of gemm_core.h, and the library-wide entry points.
#include "gemm_core.h"
#include "gemm2.h"
#include "linear_direct.h"

#ifndef MVAE_XCD_MAP
#define MVAE_XCD_MAP 1          // Linear launches: XCD-local output sub-grids (gemm_core.h, igemm_kernel)
#endif


// ==========================================================================================
// C ABI
// ==========================================================================================
MVAE_EXPORT int mvae_abi_version(void) { return 6; }

#ifdef MVAE_TUNING
// tuning build only (libmvae_hip_tuning.so): force tile shapes / split counts for tools/gemm_bench.py
MvaeTune g_mvae_tune = {0, 0, 0, 0, 0, 0, 0, 0};
MVAE_EXPORT void mvae_debug_set_tiling(int wm, int wn, int splits) {
    g_mvae_tune.wm = wm; g_mvae_tune.wn = wn; g_mvae_tune.splits = splits;
}
MVAE_EXPORT void mvae_debug_set_kwaves(int kw) { g_mvae_tune.kw = kw; }
MVAE_EXPORT void mvae_debug_set_small(int off, int waves) { g_mvae_tune.small_off = off; g_mvae_tune.small_waves = waves; }
MVAE_EXPORT void mvae_debug_set_split_target(long blocks) { g_mvae_tune.split_target = blocks; }
MVAE_EXPORT void mvae_debug_set_knockout(int mode) { g_mvae_tune.knockout = mode; }
#endif

MVAE_EXPORT size_t mvae_gemm_ws_bytes(int rows_out, int cols_out, int reduce_len) {
    if (rows_out <= 0 || cols_out <= 0 || reduce_len <= 0) return 0;
    size_t n = split_ws_floats(rows_out, cols_out, reduce_len);
    const size_t repack = (size_t)rows_out * cols_out;      // dgrad-form weight repack: Cin x (Cout*16)
    if (MVAE_TUNE(splits) > 0) n = (size_t)MVAE_TUNE(splits) * ((size_t)rows_out * cols_out + rows_out);
    const size_t smallcin = (size_t)512 * rows_out * cols_out;      // per-block partials of wgrad_smallcin_kernel
    if (rows_out <= 64 && cols_out <= 64 && smallcin > n) n = smallcin;
    const size_t g2n = g2_ws_floats_max(rows_out, cols_out, reduce_len);    // slabs of the version-2 core's cut tiles
    if (g2n > n) n = g2n;
    return (n > repack ? n : repack) * sizeof(float);
}

// Linear layers.  G > 1: G independent problems of one shape in ONE launch (celeba19's 18 attribute
// experts, celeba19/model.py:173-196) -- operand g lives at base + g * group stride; the group index
// rides on the class slot of the grid, so a layer of all 18 experts is 18x the blocks instead of 18
// under-filled launches.  With scratch a grouped launch may split the reduction like a single one:
// class c keeps its partials in its own region of the scratch, the finish launch has one grid slice per class.
struct LinGroups { int G; size_t a, b, c, d; };     // meaning of a..d per entry point below

static int linear_fwd_impl(const float *x, int ldx, const float *w, const float *bias, float *pre, float *act,
                           int ldy, const float *mask, float mask_scale, int M, int N, int K, void *ws,
                           size_t ws_bytes, LinGroups gr, hipStream_t st) {
    // gr: a = x stride, b = w stride, c = bias stride, d = pre/act stride
    const bool vec = aligned16(x) && aligned16(w) && ldx % 4 == 0 && K % 4 == 0 && gr.a % 4 == 0 && gr.b % 4 == 0;
    Plan pl = make_plan(M, N, K, ws != nullptr, PLAN_FWD, gr.G, vec);
    pl.xcd = MVAE_XCD_MAP;
    SplitSink sink = make_sink(ws, M, N, false);
    sink.ncls = gr.G; sink.cls_region = (size_t)pl.splits * sink.stride;
    if (pl.splits > 1 && ws_bytes < gr.G * sink.cls_region * sizeof(float)) return MVAE_ERR_WS;
    EpRowMajor e;
    e.out = pre; e.act = act; e.ld = ldy; e.bias = bias; e.dpre = nullptr; e.ldp = 0;
    e.mask = mask; e.ldm = N; e.mask_scale = mask_scale; e.I = M; e.J = N; e.accumulate = 0;
    e.out_cs = gr.d; e.bias_cs = gr.c;
    auto mp = [&](auto &p) { p.src = x; p.ld = ldx; p.R = M; p.Klen = K; p.cls_stride = gr.a; };
    auto mq = [&](auto &q) { q.src = w; q.ld = K; q.R = N; q.Klen = K; q.cls_stride = gr.b; };
    if (vec) {
        G2Plan g2 = g2_plan_for(M, N, K, gr.G, false, ws, ws_bytes, (pre && act) ? G2_FWD_TWO_OUTPUTS : (act && !pre && !mask) ? G2_FWD_ACT_ONLY : G2_PLAIN);
        if (g2.ok) return launch_gemm2<G2RowsK, G2RowsK, EpRowMajor, false>(g2, mp, mq, e, st);
    }
    if (vec) {
        int rc2 = MVAE_OK;
        if (launch_gemm2s<EpRowMajor, true>(pl, x, ldx, gr.a, w, K, gr.b, e, M, N, K, gr.G, st, &rc2)) return rc2;
    }
    if (vec)
        return launch_igemm_small<LdRowsK, LdRowsK, LdRowsK64, LdRowsK64, EpRowMajor, false>(pl, mp, mq, e, M, N, K, sink, st);
    return launch_igemm<LdRowsKS, LdRowsKS, EpRowMajor, false>(pl, mp, mq, e, M, N, K, sink, st);
}

// Linear forward + reconstruction term in one launch (EpRowBce / EpRowCe, gemm_core.h): same loaders and plan as
// linear_fwd_impl, never a split reduction (the epilogue needs whole sums).
template <class E>
static int linear_loss_impl(const float *x, int ldx, const float *w, E e, int M, int N, int K, hipStream_t st) {
    const bool vec = aligned16(x) && aligned16(w) && ldx % 4 == 0 && K % 4 == 0;
    Plan pl = make_plan(M, N, K, false, PLAN_FWD, 1, vec);
    if (pl.splits != 1) return MVAE_ERR_ARG;
    pl.xcd = MVAE_XCD_MAP;
    SplitSink sink = make_sink(nullptr, M, N, false);
    sink.ncls = 1; sink.cls_region = 0;
    auto mp = [&](auto &p) { p.src = x; p.ld = ldx; p.R = M; p.Klen = K; p.cls_stride = 0; };
    auto mq = [&](auto &q) { q.src = w; q.ld = K; q.R = N; q.Klen = K; q.cls_stride = 0; };
    if (vec)
        return launch_igemm_small<LdRowsK, LdRowsK, LdRowsK64, LdRowsK64, E, false>(pl, mp, mq, e, M, N, K, sink, st);
    return launch_igemm<LdRowsKS, LdRowsKS, E, false>(pl, mp, mq, e, M, N, K, sink, st);
}

// Data gradient over a SHORT reduction (N <= 16 output features of the forward layer: the 10-class head of
// mnist/model.py:146, celeba19's one-logit attribute heads, celeba/model.py:170's 18): dx[i][j] = sum_n dy[i][n] * w[n][j] is
// N multiply-adds per output -- memory-side work.  The MFMA tile kernel ran it as a K = 10 GEMM at 0.8 TFLOP/s (12.8 us on
// MNIST's critical chain, profiles/r05_mnist_by_shape.txt; 34 us for celeba19's 18 groups with N = 1).  Here a thread owns
// one column j for ROWS rows: its N weights stay in registers, the dy values of a row are block-uniform (scalar loads),
// and the epilogue operands of all its rows are fetched before the first is used (EpRowMajor::fetch).
constexpr int DG_SMALLN_MAX = 16, DG_SMALLN_ROWS = 8;
__global__ __launch_bounds__(256) void dgrad_smalln_kernel(const float *__restrict__ dy, int lddy, size_t dy_cs,
                                                           const float *__restrict__ w, size_t w_cs, EpRowMajor e, int M,
                                                           int N, int K) {
    __shared__ float sdy[DG_SMALLN_ROWS][DG_SMALLN_MAX];
    const int cls = blockIdx.z;
    e.set_class(cls);
    dy += (size_t)cls * dy_cs; w += (size_t)cls * w_cs;
    const int t = threadIdx.x;
    const int j = blockIdx.x * 256 + t;
    const int jc = min(j, K - 1);
    const int i0 = blockIdx.y * DG_SMALLN_ROWS;
    // everything the block needs is requested at once: its dy rows (one value per thread, through LDS), the thread's N
    // weights, the epilogue operands of its ROWS outputs -- ONE memory round trip, then arithmetic and stores
    const int dr = t / DG_SMALLN_MAX, dn = t % DG_SMALLN_MAX;
    float dval = 0.f;
    if (t < DG_SMALLN_ROWS * DG_SMALLN_MAX) dval = dy[(size_t)min(i0 + dr, M - 1) * lddy + min(dn, N - 1)];
    float wv[DG_SMALLN_MAX];
#pragma unroll
    for (int n = 0; n < DG_SMALLN_MAX; ++n) wv[n] = w[(size_t)min(n, N - 1) * K + jc];
    EpRowMajor::Pre pre[DG_SMALLN_ROWS];
#pragma unroll
    for (int r = 0; r < DG_SMALLN_ROWS; ++r) pre[r] = e.fetch(i0 + r, j);
    if (t < DG_SMALLN_ROWS * DG_SMALLN_MAX) sdy[dr][dn] = dn < N ? dval : 0.f;      // columns beyond N multiply by zero
    __syncthreads();
#pragma unroll
    for (int r = 0; r < DG_SMALLN_ROWS; ++r) {
        const int i = i0 + r;
        float s = 0.f;
#pragma unroll
        for (int n = 0; n < DG_SMALLN_MAX; ++n) s += sdy[r][n] * wv[n];
        if (i < M && j < K) e.put_pre(i, j, s, pre[r]);
    }
}

static int linear_dgrad_impl(const float *dy, int lddy, const float *w, float *dx, int lddx, const float *pre_in,
                             const float *mask, float mask_scale, int M, int N, int K, int flags, void *ws,
                             size_t ws_bytes, LinGroups gr, hipStream_t st) {
    // D[i = m][j = k] = sum_n dy[m][n] * w[n][k];  gr: a = dy stride, b = w stride, c = pre_in stride, d = dx stride
    const bool vec = aligned16(dy) && aligned16(w) && lddy % 4 == 0 && N % 4 == 0 && K % 4 == 0 && gr.a % 4 == 0 &&
                     gr.b % 4 == 0;
    Plan pl = make_plan(M, K, N, ws != nullptr, PLAN_FWD, gr.G, vec);
    pl.xcd = MVAE_XCD_MAP;
    SplitSink sink = make_sink(ws, M, K, false);
    sink.ncls = gr.G; sink.cls_region = (size_t)pl.splits * sink.stride;
    if (pl.splits > 1 && ws_bytes < gr.G * sink.cls_region * sizeof(float)) return MVAE_ERR_WS;
    EpRowMajor e;
    e.out = dx; e.act = nullptr; e.ld = lddx; e.bias = nullptr; e.dpre = pre_in; e.ldp = K;
    e.mask = mask; e.ldm = K; e.mask_scale = mask_scale; e.I = M; e.J = K;
    e.accumulate = (flags & MVAE_ACCUMULATE) ? 1 : 0;
    e.out_cs = gr.d; e.dpre_cs = gr.c;
    if (N <= DG_SMALLN_MAX && !MVAE_TUNE(small_off)) {
        hipLaunchKernelGGL(dgrad_smalln_kernel, dim3((unsigned)cdiv(K, 256), (unsigned)cdiv(M, DG_SMALLN_ROWS), gr.G), dim3(256), 0,
                           st, dy, lddy, gr.a, w, gr.b, e, M, N, K);
        return mvae_launch_status();
    }
    auto mp = [&](auto &p) { p.src = dy; p.ld = lddy; p.R = M; p.Klen = N; p.cls_stride = gr.a; };
    auto mq = [&](auto &q) { q.src = w; q.ld = K; q.R = K; q.Klen = N; q.cls_stride = gr.b; };
    if (vec) {
        G2Plan g2 = g2_plan_for(M, K, N, gr.G, false, ws, ws_bytes);
        if (g2.ok) return launch_gemm2<G2RowsK, G2RowsMN, EpRowMajor, false>(g2, mp, mq, e, st);
    }
    if (vec) {
        int rc2 = MVAE_OK;
        if (launch_gemm2s<EpRowMajor, false>(pl, dy, lddy, gr.a, w, K, gr.b, e, M, K, N, gr.G, st, &rc2)) return rc2;
    }
    if (vec)
        return launch_igemm_small<LdRowsK, LdRowsMN, LdRowsK64, LdRowsMN64, EpRowMajor, false>(pl, mp, mq, e, M, K, N, sink, st);
    return launch_igemm<LdRowsKS, LdRowsMNS, EpRowMajor, false>(pl, mp, mq, e, M, K, N, sink, st);
}

static int linear_wgrad_impl(const float *dy, int lddy, const float *x, int ldx, float *dw, float *db, int M, int N,
                             int K, int flags, void *ws, size_t ws_bytes, LinGroups gr, hipStream_t st) {
    // D[i = n][j = k] = sum_m dy[m][n] * x[m][k];  gr: a = dy stride, b = x stride, c = db stride, d = dw stride
    const bool vec = aligned16(dy) && aligned16(x) && lddy % 4 == 0 && ldx % 4 == 0 && N % 4 == 0 && K % 4 == 0 &&
                     gr.a % 4 == 0 && gr.b % 4 == 0;
    Plan pl = make_plan(N, K, M, ws != nullptr, PLAN_LIN_WGRAD, gr.G, vec);
    pl.xcd = MVAE_XCD_MAP;
    SplitSink sink = make_sink(ws, N, K, db != nullptr);
    sink.ncls = gr.G; sink.cls_region = (size_t)pl.splits * sink.stride;
    if (pl.splits > 1 && ws_bytes < gr.G * sink.cls_region * sizeof(float)) return MVAE_ERR_WS;
    const int acc = (flags & MVAE_ACCUMULATE) ? 1 : 0;
    EpRowMajor e;
    e.out = dw; e.act = nullptr; e.ld = K; e.bias = nullptr; e.dpre = nullptr; e.ldp = 0;
    e.mask = nullptr; e.ldm = 0; e.mask_scale = 1.f; e.I = N; e.J = K; e.accumulate = acc;
    e.out_cs = gr.d;
    if (gr.G == 1 && wgrad_direct_ok(N, K, M))
        return wgrad_direct_launch(dy, lddy, x, ldx, e, N, K, M, db, acc, st);
    auto mp = [&](auto &p) { p.src = dy; p.ld = lddy; p.R = N; p.Klen = M; p.cls_stride = gr.a; };
    auto mq = [&](auto &q) { q.src = x; q.ld = ldx; q.R = K; q.Klen = M; q.cls_stride = gr.b; };
    if (vec) {
        G2Plan g2 = g2_plan_for(N, K, M, gr.G, db != nullptr, ws, ws_bytes);
        if (g2.ok) {
            g2.a.rowsum = db; g2.a.rowsum_cls_stride = gr.c; g2.a.rowsum_accumulate = acc;
            return db ? launch_gemm2<G2RowsMN, G2RowsMN, EpRowMajor, true>(g2, mp, mq, e, st)
                      : launch_gemm2<G2RowsMN, G2RowsMN, EpRowMajor, false>(g2, mp, mq, e, st);
        }
    }
    int rc;
    if (db) {
        // row sums of P = dy^T are the bias gradient; partials live right after each dw partial
        if (pl.splits == 1) {
            sink.rowsum = db; sink.rowsum_stride = 0; sink.rowsum_accumulate = acc; sink.rowsum_cls_stride = gr.c;
        } else {
            sink.rowsum = (float *)ws + (size_t)N * K; sink.rowsum_stride = sink.stride; sink.rowsum_accumulate = 0;
            sink.rowsum_cls_stride = sink.cls_region;
            sink.rowsum_final = db; sink.rowsum_final_accumulate = acc;     // summed by the finish launch
            sink.rowsum_final_cls_stride = gr.c;
        }
        rc = vec ? launch_igemm_small<LdRowsMN, LdRowsMN, LdRowsMN64, LdRowsMN64, EpRowMajor, true>(pl, mp, mq, e, N, K, M, sink, st)
                 : launch_igemm<LdRowsMNS, LdRowsMNS, EpRowMajor, true>(pl, mp, mq, e, N, K, M, sink, st);
        if (rc) return rc;
        return MVAE_OK;
    }
    return vec ? launch_igemm_small<LdRowsMN, LdRowsMN, LdRowsMN64, LdRowsMN64, EpRowMajor, false>(pl, mp, mq, e, N, K, M, sink, st)
               : launch_igemm<LdRowsMNS, LdRowsMNS, EpRowMajor, false>(pl, mp, mq, e, N, K, M, sink, st);
}

// items -> the kernel's table; with `adam`: no accumulation, and items without dy / x are finished gradients of J = K
// elements that only take the update
static int wgrad_batch_table(const mvae_wgrad_item *items, int n_items, bool adam, WgradBatchArgs &a) {
    if (n_items < 1 || n_items > WGRAD_BATCH_MAX || !items) return MVAE_ERR_ARG;
    a.n = n_items;
    int tiles = 0;
    for (int q = 0; q < n_items; ++q) {
        const mvae_wgrad_item &s = items[q];
        WgradBatchItem &w = a.it[q];
        for (int r = 0; r < q; ++r)
            if (items[r].dw == s.dw || (s.db && items[r].db == s.db)) return MVAE_ERR_ARG;   // one writer per gradient
        if (adam && !s.dy && !s.x) {
            if (!s.dw || s.db || s.K < 1 || s.flags) return MVAE_ERR_ARG;
            w.dy = w.x = nullptr; w.dw = s.dw; w.db = nullptr;
            w.lddy = w.ldx = w.M = w.I = 0; w.J = s.K; w.accumulate = 0; w.tiles_j = 1;
            tiles += cdiv(s.K, ADAM_ONLY_TILE);
            w.tile_end = tiles;
            continue;
        }
        if (!s.dy || !s.x || !s.dw || s.M < 1 || s.N < 1 || s.K < 1 || s.lddy < s.N || s.ldx < s.K) return MVAE_ERR_ARG;
        if (!wgrad_batch_item_ok(s.N, s.K, s.M, s.lddy, s.ldx)) return MVAE_ERR_ARG;
        if (adam && (s.flags & MVAE_ACCUMULATE)) return MVAE_ERR_ARG;      // the update needs the step's WHOLE gradient
        w.dy = s.dy; w.x = s.x; w.dw = s.dw; w.db = s.db;
        w.lddy = s.lddy; w.ldx = s.ldx; w.M = s.M; w.I = s.N; w.J = s.K;
        w.accumulate = (s.flags & MVAE_ACCUMULATE) ? 1 : 0;
        w.tiles_j = cdiv(s.K, 32);
        tiles += cdiv(s.N, 32) * w.tiles_j;
        w.tile_end = tiles;
    }
    return MVAE_OK;
}

MVAE_EXPORT int mvae_linear_wgrad_batched(const mvae_wgrad_item *items, int n_items, mvae_stream_t stream) {
    WgradBatchArgs a;
    const int rc = wgrad_batch_table(items, n_items, false, a);
    return rc != MVAE_OK ? rc : wgrad_batched_launch(a, (hipStream_t)stream);
}

MVAE_EXPORT int mvae_linear_wgrad_batched_adam(const mvae_wgrad_item *items, int n_items, const mvae_adam_fuse *adam,
                                               mvae_stream_t stream) {
    if (!adam || !adam->grad_base || !adam->param_base || !adam->exp_avg_base || !adam->exp_avg_sq_base || !adam->coef2)
        return MVAE_ERR_ARG;
    WgradBatchArgs a;
    const int rc = wgrad_batch_table(items, n_items, true, a);
    if (rc != MVAE_OK) return rc;
    for (int q = 0; q < n_items; ++q) {      // 32-bit element offsets from the arena bases inside the kernel
        const size_t n = items[q].dy ? (size_t)items[q].N * items[q].K : (size_t)items[q].K;
        if (items[q].dw < adam->grad_base || (size_t)(items[q].dw - adam->grad_base) + n >= ((size_t)1 << 32)) return MVAE_ERR_ARG;
        if (items[q].db && (items[q].db < adam->grad_base || (size_t)(items[q].db - adam->grad_base) + items[q].N >= ((size_t)1 << 32)))
            return MVAE_ERR_ARG;
    }
    AdamFuse f;
    f.grad_base = adam->grad_base; f.param = adam->param_base; f.m = adam->exp_avg_base; f.v = adam->exp_avg_sq_base;
    f.coef = adam->coef2;
    // rounded as mvae_adam_apply rounds them (beta and 1 - beta separately)
    f.b1 = (float)adam->beta1; f.b2 = (float)adam->beta2; f.eps = (float)adam->eps; f.gscale = adam->grad_scale;
    f.omb1 = (float)(1.0 - adam->beta1); f.omb2 = (float)(1.0 - adam->beta2);
    return wgrad_batched_launch(a, (hipStream_t)stream, &f);
}

static const LinGroups kOneGroup = {1, 0, 0, 0, 0};

MVAE_EXPORT int mvae_linear_fwd(const float *x, int ldx, const float *w, const float *bias,
                                float *pre, float *act, int ldy, const float *mask, float mask_scale,
                                int M, int N, int K, void *ws, size_t ws_bytes, mvae_stream_t stream) {
    if (!x || !w || (!pre && !act) || M <= 0 || N <= 0 || K <= 0 || ldx < K || ldy < N) return MVAE_ERR_ARG;
    return linear_fwd_impl(x, ldx, w, bias, pre, act, ldy, mask, mask_scale, M, N, K, ws, ws_bytes, kOneGroup,
                           (hipStream_t)stream);
}

MVAE_EXPORT int mvae_linear_bce_fwd(const float *x, int ldx, const float *w, const float *bias, const float *target,
                                    int target_rows, int target_row_stride, const float *drow_dev, int rows_per_group,
                                    float *dlogits, int ldy, float *logits, float *partial, int M, int N, int K,
                                    mvae_stream_t stream) {
    if (!x || !w || !target || !drow_dev || !dlogits || !partial || M <= 0 || N <= 0 || K <= 0 || ldx < K || ldy < N ||
        target_rows <= 0 || target_row_stride < N || rows_per_group <= 0)
        return MVAE_ERR_ARG;
    EpRowBce e;
    e.dlogits = dlogits; e.ld = ldy; e.logits = logits; e.bias = bias;
    e.target = target; e.t_rs = target_row_stride; e.target_rows = target_rows;
    e.drow = drow_dev; e.rows_per_group = rows_per_group;
    e.part = partial; e.nparts = (N + 31) / 32; e.I = M; e.J = N;
    return linear_loss_impl(x, ldx, w, e, M, N, K, (hipStream_t)stream);
}

MVAE_EXPORT int mvae_linear_ce_fwd(const float *x, int ldx, const float *w, const float *bias, const int64_t *label,
                                   int label_rows, const float *drow_dev, int rows_per_group, float *dlogits, int ldy,
                                   float *logits, float *row, int M, int N, int K, mvae_stream_t stream) {
    if (!x || !w || !label || !drow_dev || !dlogits || !row || M <= 0 || N <= 0 || N > 32 || K <= 0 || ldx < K ||
        ldy < N || label_rows <= 0 || rows_per_group <= 0)
        return MVAE_ERR_ARG;
    EpRowCe e;
    e.dlogits = dlogits; e.ld = ldy; e.logits = logits; e.bias = bias;
    e.label = label; e.label_rows = label_rows; e.drow = drow_dev; e.rows_per_group = rows_per_group;
    e.row = row; e.I = M; e.J = N;
    return linear_loss_impl(x, ldx, w, e, M, N, K, (hipStream_t)stream);
}

MVAE_EXPORT int mvae_linear_dgrad(const float *dy, int lddy, const float *w, float *dx, int lddx,
                                  const float *pre_in, const float *mask, float mask_scale,
                                  int M, int N, int K, int flags, void *ws, size_t ws_bytes,
                                  mvae_stream_t stream) {
    if (!dy || !w || !dx || M <= 0 || N <= 0 || K <= 0 || lddy < N || lddx < K) return MVAE_ERR_ARG;
    return linear_dgrad_impl(dy, lddy, w, dx, lddx, pre_in, mask, mask_scale, M, N, K, flags, ws, ws_bytes,
                             kOneGroup, (hipStream_t)stream);
}

MVAE_EXPORT int mvae_linear_wgrad(const float *dy, int lddy, const float *x, int ldx, float *dw, float *db,
                                  int M, int N, int K, int flags, void *ws, size_t ws_bytes,
                                  mvae_stream_t stream) {
    if (!dy || !x || !dw || M <= 0 || N <= 0 || K <= 0 || lddy < N || ldx < K) return MVAE_ERR_ARG;
    return linear_wgrad_impl(dy, lddy, x, ldx, dw, db, M, N, K, flags, ws, ws_bytes, kOneGroup, (hipStream_t)stream);
}

static inline bool groups_ok(int G) { return G >= 1 && G <= 4096; }

MVAE_EXPORT int mvae_linear_fwd_grouped(const float *x, int ldx, size_t x_gs, const float *w, size_t w_gs,
                                        const float *bias, size_t bias_gs, float *pre, float *act, int ldy,
                                        size_t y_gs, int G, int M, int N, int K, void *ws, size_t ws_bytes,
                                        mvae_stream_t stream) {
    if (!x || !w || (!pre && !act) || !groups_ok(G) || M <= 0 || N <= 0 || K <= 0 || ldx < K || ldy < N)
        return MVAE_ERR_ARG;
    const LinGroups gr = {G, x_gs, w_gs, bias_gs, y_gs};
    return linear_fwd_impl(x, ldx, w, bias, pre, act, ldy, nullptr, 1.f, M, N, K, ws, ws_bytes, gr, (hipStream_t)stream);
}

MVAE_EXPORT int mvae_linear_dgrad_grouped(const float *dy, int lddy, size_t dy_gs, const float *w, size_t w_gs,
                                          float *dx, int lddx, size_t dx_gs, const float *pre_in, size_t pre_gs,
                                          int G, int M, int N, int K, int flags, void *ws, size_t ws_bytes,
                                          mvae_stream_t stream) {
    if (!dy || !w || !dx || !groups_ok(G) || M <= 0 || N <= 0 || K <= 0 || lddy < N || lddx < K) return MVAE_ERR_ARG;
    const LinGroups gr = {G, dy_gs, w_gs, pre_gs, dx_gs};
    return linear_dgrad_impl(dy, lddy, w, dx, lddx, pre_in, nullptr, 1.f, M, N, K, flags, ws, ws_bytes, gr,
                             (hipStream_t)stream);
}

MVAE_EXPORT int mvae_linear_wgrad_grouped(const float *dy, int lddy, size_t dy_gs, const float *x, int ldx,
                                          size_t x_gs, float *dw, size_t dw_gs, float *db, size_t db_gs, int G,
                                          int M, int N, int K, int flags, void *ws, size_t ws_bytes,
                                          mvae_stream_t stream) {
    if (!dy || !x || !dw || !groups_ok(G) || M <= 0 || N <= 0 || K <= 0 || lddy < N || ldx < K) return MVAE_ERR_ARG;
    const LinGroups gr = {G, dy_gs, x_gs, db_gs, dw_gs};
    return linear_wgrad_impl(dy, lddy, x, ldx, dw, db, M, N, K, flags, ws, ws_bytes, gr, (hipStream_t)stream);
}
