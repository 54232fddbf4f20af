// linear_direct.h -- the Linear weight gradient without LDS tiles and without block barriers.
//
//   dw[n][k] (+)= sum_m dy[m][n] * x[m][k],   db[n] (+)= sum_m dy[m][n]        (mnist/model.py:75-78,95-98 ... backward)
//
// Both operands have the NON-reduced axis contiguous, which is exactly the layout an MFMA fragment wants:
// for the 32x32x2 fp32 MFMA lane (c, h) supplies A[i0 + c][m] and B[m][j0 + c] -- 32 consecutive floats of
// row m of dy / x per half-wave, a full 128-byte segment.  So every wave loads its own fragments straight from
// global memory (4 dword loads per operand per 8 rows of m, software-prefetched PD chunks ahead), owns one
// 32x32 output tile over every KW-th 8-row chunk of the batch and never meets another wave until the end,
// where the KW partial tiles are summed through LDS in a fixed order and all threads run the epilogue.
// The LDS-tiled kernel spent half its time in the per-k-step barrier on these shapes (4 MFMAs per wave between
// barriers); here the only synchronisation is the final reduction.
#pragma once
#include "gemm_core.h"

namespace {

#ifdef MVAE_TUNING
#define MVAE_KO1 1
#define MVAE_KO2 2
#else
#define MVAE_KO1 0
#define MVAE_KO2 0
#endif
constexpr int WD_PD = 4;      // chunks (of 8 reduction rows) in flight per wave: 2 x 16 VGPRs

// A weight-gradient launch that ALSO applies Adam to the parameters whose gradients it just produced
// (mvae_linear_wgrad_batched_adam): the gradient arena, the parameter arena and the two moment arenas are laid out
// alike, so the element behind gradient address g is at the same offset from each base.  `coef` = the two
// bias-correction factors of this step (mvae_adam_prepare).
struct AdamFuse {
    const float *grad_base; float *param, *m, *v; const float *coef;
    float b1, b2, omb1, omb2, eps, gscale;
    __device__ AdamCoef load() const {
        AdamCoef c;
        c.b1 = b1; c.b2 = b2; c.omb1 = omb1; c.omb2 = omb2; c.eps = eps; c.gscale = gscale;
        c.step_size = coef[0]; c.inv_sqrt_bc2 = coef[1];
        return c;
    }
    __device__ void update(const float *gptr, float g, const AdamCoef &c) const {
        const size_t off = (size_t)(gptr - grad_base);
        adam_one(param[off], m[off], v[off], g, c);
    }
};

// MODE (tuning build only: mvae_debug_set_knockout): 0 = the kernel; 1 = loads without the MFMAs; 2 = MFMAs
// without the loads -- where the time of a launch goes.
template <bool ROWSUM, int KW, int MODE = 0, bool ADAM = false>
__device__ __forceinline__ void wgrad_direct_tile(const float *dy, int lddy, const float *x, int ldx,
                                                  const EpRowMajor &e, int I, int J, int M, float *db,
                                                  int db_accumulate, int tile_i, int tile_j,
                                                  const AdamFuse *af = nullptr) {
    extern __shared__ __attribute__((aligned(16))) float wd_lds[];
    const int t = threadIdx.x, lane = t & 63, kg = t >> 6;
    const int lcol = lane & 31, lrow = lane >> 5;
    const int i0 = tile_i * 32, j0 = tile_j * 32;
    // 32-bit element offsets from the (wave-uniform) tensor bases: scalar base + vector offset addressing.
    // Columns beyond the operand are clamped (their products land in outputs the epilogue drops).
    const unsigned col_a = (unsigned)min(i0 + lcol, I - 1) * 4u, col_b = (unsigned)min(j0 + lcol, J - 1) * 4u;
    const unsigned ulda = (unsigned)lddy * 4u, uldb = (unsigned)ldx * 4u;       // BYTE strides / offsets (< 4 GiB)
    const char *dyb = reinterpret_cast<const char *>(dy), *xb = reinterpret_cast<const char *>(x);
    const int full = M >> 3;                        // chunks of 8 reduction rows entirely inside the batch

    // ADAM: the parameter and the two moments of the outputs this thread will finish in the epilogue are fetched NOW
    // and arrive behind the whole reduction -- fetched in the epilogue, every block ended with a dependent
    // load -> compute -> store round trip to HBM (MNIST step +3 % against the arena-wide launch; +1.9 % this way:
    // profiles/r04_fuse_adam_ab.txt -- the form stays an option, the engines do not use it by default).
    constexpr int NE = 1024 / (64 * KW);            // epilogue elements per thread
    float ap[NE], am[NE], av2[NE];
    unsigned aoff[NE];                              // arena element index (the arenas are far below 2^32 elements: host)
    if (ADAM) {
#pragma unroll
        for (int k = 0; k < NE; ++k) {
            const int el = t + k * 64 * KW, i = i0 + (el >> 5), j = j0 + (el & 31);
            const bool ok = i < I && j < J;
            aoff[k] = ok ? (unsigned)(e.out - af->grad_base) + (unsigned)i * (unsigned)e.ld + (unsigned)j : ~0u;
            ap[k] = ok ? af->param[aoff[k]] : 0.f;
            am[k] = ok ? af->m[aoff[k]] : 0.f;
            av2[k] = ok ? af->v[aoff[k]] : 0.f;
        }
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float rs = 0.f;
    // Two register sets of WD_PD chunks each: while one set is multiplied the other is in flight, and the
    // loop body handles both (A: refill set 1, consume set 0; B: refill set 0, consume set 1), so a set is
    // dead when its refill is issued and the loaded values stay in the registers the next trip reads --
    // a rotating single set made hipcc copy 32 in-flight registers at the back edge behind s_waitcnt vmcnt(0).
    float a0[WD_PD][4], b0[WD_PD][4], a1[WD_PD][4], b1[WD_PD][4];

    auto fetch = [&](int c, float (&av)[4], float (&bv)[4]) {
        const unsigned m0 = (unsigned)(c * 8 + 4 * lrow);
        unsigned oa = m0 * ulda + col_a, ob = m0 * uldb + col_b;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (MODE == 2) { av[q] = (float)oa; bv[q] = (float)ob; }
            else { av[q] = *reinterpret_cast<const float *>(dyb + oa); bv[q] = *reinterpret_cast<const float *>(xb + ob); }
            oa += ulda; ob += uldb;
        }
    };
    auto mfma4 = [&](const float (&av)[4], const float (&bv)[4], float live) {
        const float v0 = av[0] * live, v1 = av[1] * live, v2 = av[2] * live, v3 = av[3] * live;
        if (ROWSUM) rs += (v0 + v1) + (v2 + v3);
        if (MODE == 1) { acc[0] += (v0 * bv[0] + v1 * bv[1]) + (v2 * bv[2] + v3 * bv[3]); return; }
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v0, bv[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v1, bv[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v2, bv[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v3, bv[3], acc, 0, 0, 0);
    };
    // this wave's chunks: kg, kg + KW, ...  (all waves of the block stream through the same rows together).
    // Every load is UNCONDITIONAL (chunk index clamped to this wave's last chunk; a surplus chunk is
    // multiplied by zero): a load under a branch makes hipcc drain the whole memory queue.
    const int n_my = (full - kg + KW - 1) / KW;     // may be 0 when the batch is shorter than 8 * KW rows
    auto load_set = [&](int first, float (&as)[WD_PD][4], float (&bs)[WD_PD][4]) {
#pragma unroll
        for (int p = 0; p < WD_PD; ++p) fetch(kg + min(first + p, n_my - 1) * KW, as[p], bs[p]);
    };
    auto use_set = [&](int first, const float (&as)[WD_PD][4], const float (&bs)[WD_PD][4]) {
#pragma unroll
        for (int p = 0; p < WD_PD; ++p) mfma4(as[p], bs[p], (first + p < n_my) ? 1.f : 0.f);
    };
    if (n_my > 0) {
        load_set(0, a0, b0);
        for (int it = 0; it < n_my; it += 2 * WD_PD) {
            load_set(it + WD_PD, a1, b1);
            __builtin_amdgcn_sched_barrier(0);
            use_set(it, a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            load_set(it + 2 * WD_PD, a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            use_set(it + WD_PD, a1, b1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if ((M & 7) && kg == full % KW) {               // the ragged last chunk: rows past the batch are zero
        float av[4], bv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int m = full * 8 + 4 * lrow + q;
            const float ok = (m < M) ? 1.f : 0.f;
            const unsigned mc = (unsigned)min(m, M - 1);
            av[q] = *reinterpret_cast<const float *>(dyb + (mc * ulda + col_a)) * ok;
            bv[q] = *reinterpret_cast<const float *>(xb + (mc * uldb + col_b)) * ok;
        }
        mfma4(av, bv, 1.f);
    }
    // ---- reduce the KW partial tiles (fixed order) and run the epilogue with every thread
    constexpr int TP = 33;
    float *tile = wd_lds + kg * (32 * TP);
#pragma unroll
    for (int r = 0; r < 16; ++r) tile[(4 * lrow + (r & 3) + 8 * (r >> 2)) * TP + lcol] = acc[r];
    float *rsl = wd_lds + KW * (32 * TP);           // [KW][2][32] partial row sums
    if (ROWSUM) rsl[(kg * 2 + lrow) * 32 + lcol] = rs;
    __syncthreads();
    AdamCoef ac;
    if (ADAM) ac = af->load();
#pragma unroll
    for (int k = 0; k < NE; ++k) {
        const int el = t + k * 64 * KW;
        const int il = el >> 5, jl = el & 31;
        float v = 0.f;
#pragma unroll
        for (int g2 = 0; g2 < KW; ++g2) v += wd_lds[g2 * (32 * TP) + il * TP + jl];
        if (e.col(j0 + jl)) {
            e.put(i0 + il, j0 + jl, v);
            // ADAM launches never accumulate (host): v IS the gradient the put just stored
            if (ADAM && aoff[k] != ~0u) {
                adam_one(ap[k], am[k], av2[k], v, ac);
                af->param[aoff[k]] = ap[k]; af->m[aoff[k]] = am[k]; af->v[aoff[k]] = av2[k];
            }
        }
    }
    if (ROWSUM) {
        if (tile_j == 0 && t < 32 && i0 + t < I) {
            float s = 0.f;
#pragma unroll
            for (int g2 = 0; g2 < 2 * KW; ++g2) s += rsl[g2 * 32 + t];
            if (db_accumulate) s += db[i0 + t];
            db[i0 + t] = s;
            if (ADAM) af->update(db + i0 + t, s, ac);
        }
    }
}

template <bool ROWSUM, int KW, int MODE = 0>
__global__ __launch_bounds__(64 * KW) void wgrad_direct_kernel(const float *dy, int lddy, const float *x, int ldx,
                                                              EpRowMajor e, int I, int J, int M, float *db,
                                                              int db_accumulate) {
    wgrad_direct_tile<ROWSUM, KW, MODE>(dy, lddy, x, ldx, e, I, J, M, db, db_accumulate, blockIdx.y, blockIdx.x);
}

// ---- several Linear weight gradients in ONE launch (mvae_linear_wgrad_batched): the weight gradients of a
//      stack are off the data-gradient chain -- nothing but the optimizer reads them -- so the backward pass
//      queues them and issues them together: the 4 (decoder) + 3 (encoder) launches of an MNIST stack become
//      one grid that fills the chip, instead of seven 10-us launches on the critical path of a 380-us step.
//      A block looks its problem up in the table (block-uniform scan of <= 16 prefix counts) and runs the
//      same tile code as the single-problem kernel.
constexpr int WGRAD_BATCH_MAX = 16;
struct WgradBatchItem {
    const float *dy, *x; float *dw, *db;
    int lddy, ldx, M, I, J, accumulate;
    int tiles_j, tile_end;          // tiles along J; first tile index AFTER this problem in the launch
};
struct WgradBatchArgs { WgradBatchItem it[WGRAD_BATCH_MAX]; int n; };
// ... with Adam on the outputs (mvae_linear_wgrad_batched_adam).  An item with dy == nullptr is a gradient that is
// already final -- J elements at dw, written by an earlier launch of the stream (an Embedding's weight gradient):
// its tiles of 1024 elements only run the update, so that no parameter of the chain is left for a launch behind it.
struct WgradBatchAdamArgs { WgradBatchArgs w; AdamFuse adam; };
constexpr int ADAM_ONLY_TILE = 1024;

template <int KW, bool ADAM> __device__ __forceinline__ void wgrad_batched_body(const WgradBatchArgs &a, const AdamFuse *af);

template <int KW>
__global__ __launch_bounds__(64 * KW) void wgrad_batched_kernel(WgradBatchArgs a) {
    wgrad_batched_body<KW, false>(a, nullptr);
}
template <int KW>
__global__ __launch_bounds__(64 * KW) __attribute__((amdgpu_waves_per_eu(4, 4)))      // as resident as the plain batch
void wgrad_batched_adam_kernel(WgradBatchAdamArgs a) {
    wgrad_batched_body<KW, true>(a.w, &a.adam);
}

template <int KW, bool ADAM> __device__ __forceinline__ void wgrad_batched_body(const WgradBatchArgs &a, const AdamFuse *af) {
    int p = 0, first = 0;
    const int tile = blockIdx.x;
#pragma unroll 1
    for (int q = 0; q < a.n - 1; ++q) {
        if (tile >= a.it[q].tile_end) { p = q + 1; first = a.it[q].tile_end; }
    }
    const WgradBatchItem &w = a.it[p];
    const int local = tile - first;
    if (ADAM && w.dy == nullptr) {
        const AdamCoef ac = af->load();
        for (int el = threadIdx.x; el < ADAM_ONLY_TILE; el += 64 * KW) {
            const int idx = local * ADAM_ONLY_TILE + el;
            if (idx < w.J) af->update(w.dw + idx, w.dw[idx], ac);
        }
        return;
    }
    const int tile_i = local / w.tiles_j, tile_j = local - tile_i * w.tiles_j;
    EpRowMajor e;
    e.out = w.dw; e.act = nullptr; e.ld = w.J; e.bias = nullptr; e.dpre = nullptr; e.ldp = 0;
    e.mask = nullptr; e.ldm = 0; e.mask_scale = 1.f; e.I = w.I; e.J = w.J; e.accumulate = w.accumulate;
    if (w.db) wgrad_direct_tile<true, KW, 0, ADAM>(w.dy, w.lddy, w.x, w.ldx, e, w.I, w.J, w.M, w.db, w.accumulate, tile_i, tile_j, af);
    else wgrad_direct_tile<false, KW, 0, ADAM>(w.dy, w.lddy, w.x, w.ldx, e, w.I, w.J, w.M, nullptr, 0, tile_i, tile_j, af);
}

// ------------------------------------------------------------------------------------------
// Version 2 of the batched launch (round 5).  What was wrong with the one above, measured
// (profiles/r04_mnist_by_shape.txt, r04_traffic.json, r04_mnist_step_timeline.txt):
//   * 25-28 us per launch for 12.5 us of matrix time, while the launch moved 74.7 MB through the fabric for 19.9 MB of
//     operands -- the tiles of a layer went round-robin over the 8 XCDs, so every L2 pulled (nearly) every row band of
//     dy, and a 32 x 32 tile fed straight from global memory is two loads per MFMA (8 FLOP per byte at the cache);
//   * 936 blocks of 256 threads hold every CU slot for the whole launch: the 512^3 data gradient that starts 3 us
//     later on the other stream -- and IS on the critical chain -- waits for them (27.9 us instead of 7-9).
// Here
//   * a wave owns FM x FN MFMA tiles (64 x 64 or 64 x 32 outputs): an A fragment feeds FN MFMAs and a B fragment FM,
//     1 - 1.5 loads per MFMA instead of 2;
//   * fragments arrive through raw buffer loads whose descriptor covers exactly the operand: rows beyond the batch,
//     surplus chunks and the last row's columns beyond the matrix read as zero in hardware -- the main loop has no
//     clamp, no select and no multiply by a 0/1 flag left (a VALU instruction costs fp32-MFMA throughput);
//   * the tile list -- (layer, i band, j band), i-major -- is cut into 8 contiguous segments and XCD x (blocks
//     x, x + 8, ... of the launch order) walks segment x: a layer's tiles sit on as few XCDs as its share of the work,
//     an XCD's resident blocks share one or two row bands of dy and sweep x together;
//   * fewer, larger blocks (<= 2 per CU through the LDS footprint of the final reduction) leave wave slots, registers
//     and LDS on every CU for a chain kernel that arrives while the batch is running.
#ifndef MVAE_WGRAD2
#define MVAE_WGRAD2 1            // 0: the round-4 batch kernel (wgrad_batched_kernel) for every launch
#endif
#ifndef MVAE_WB2_KO
#define MVAE_WB2_KO 0
#endif
#ifndef MVAE_WB2_KW
#define MVAE_WB2_KW 0            // A/B builds: force the waves per tile (2 .. 16)
#endif
#ifndef MVAE_WB2_PD22
#define MVAE_WB2_PD22 1          // chunks of 8 rows per register set of the 64 x 64 wave tile.  2: ~150 registers; 1: 121 -- a chain kernel's
#endif                           // 512-thread block (116-122 registers) then FITS BESIDE a batch block on a CU.  The launch alone is
                                 // the same (24.5 vs 24.8 us), the MNIST step 2 % faster (0.2789-0.2802 vs 0.2844-0.2860 ms, x4
                                 // interleaved, profiles/r05_wgrad_ab.txt): sharing a CU is what the step's two streams need,
                                 // the same effect that makes four k-tiles in flight lose (gemm_core.h MVAE_PHASED_DEPTH)
#ifndef MVAE_WB2_PD11
#define MVAE_WB2_PD11 2          // chunks per register set of the 32 x 32 wave tile: 2 (48-63 registers) instead of 4 (82-98): the launch
                                 // alone 19.8 -> 18.9 us (label decoder), 12.9 -> 11.8 us (image encoder); the step unchanged (x4)
#endif
#ifndef MVAE_WB2_PRIO
#define MVAE_WB2_PRIO 0          // A/B builds: wave priority of the batch kernel (the batches sit on the side stream's chain)
#endif
#ifndef MVAE_WGRAD2_SHAPE
#define MVAE_WGRAD2_SHAPE 0      // 0: by tile count; 22 / 21 / 11: force the 64 x 64 / 64 x 32 / 32 x 32 wave tile (A/B builds)
#endif

constexpr int WB2_SEGS = 8;
struct WgradBatch2Args { WgradBatchItem it[WGRAD_BATCH_MAX]; int n; int seg[WB2_SEGS + 1]; };

// descriptor of `extent` bytes at b + off (off, extent: wave-uniform); nothing is readable when off >= extent
__device__ __forceinline__ i32x4_t wb2_rsrc(BufBase b, unsigned off, unsigned extent) {
    const unsigned long long a = (((unsigned long long)b.hi << 32) | b.lo) + off;
    i32x4_t r;
    r.x = (int)(unsigned)a; r.y = (int)((unsigned)(a >> 32) & 0xffffu);
    // signed (the host keeps extents and offsets below 2^31): `off < extent ? extent - off : 0` on unsigned values becomes a
    // saturating subtract, which only the VECTOR ALU has -- the descriptor then lives in vector registers and every load
    // in a waterfall loop
    const int left = (int)extent - (int)off;
    r.z = left > 0 ? left : 0; r.w = 0x00020000;
    return r;
}

template <int KW, int FM, int FN, int PD>
__device__ __forceinline__ void wgrad_tile2(const WgradBatchItem &w, int tile_i, int tile_j) {
    extern __shared__ __attribute__((aligned(16))) float wd_lds[];
    constexpr int TM = 32 * FM, TN = 32 * FN;
    const int t = threadIdx.x, lane = t & 63;
    const int kg = __builtin_amdgcn_readfirstlane(t >> 6);          // wave-uniform, and provably so
    const int lcol = lane & 31, lrow = lane >> 5;
    const int i0 = tile_i * TM, j0 = tile_j * TN;
    const BufBase ba = buf_base(w.dy), bb = buf_base(w.x);
    // the table entry was picked with a run-time index: the compiler holds its fields in vector registers and would wrap
    // every buffer load in a waterfall loop -- everything a descriptor is built from goes through readfirstlane once
    const int M = __builtin_amdgcn_readfirstlane(w.M), I = __builtin_amdgcn_readfirstlane(w.I), J = __builtin_amdgcn_readfirstlane(w.J);
    const unsigned ulda = (unsigned)__builtin_amdgcn_readfirstlane(w.lddy) * 4u;    // row strides in bytes
    const unsigned uldb = (unsigned)__builtin_amdgcn_readfirstlane(w.ldx) * 4u;
    // the operands end with the last row's last REAL column (a [M, ld] view of a wider tensor owns nothing behind it)
    const unsigned ext_a = (unsigned)(M - 1) * ulda + (unsigned)I * 4u;
    const unsigned ext_b = (unsigned)(M - 1) * uldb + (unsigned)J * 4u;
    // lane (c, h) supplies rows 4 h + q of a chunk of 8 to MFMA q, column c of fragment f (+ 128 f bytes: an immediate).
    // A column beyond the matrix is NOT clamped: it reads a neighbour (finite or not) and only reaches outputs the
    // epilogue drops -- column i of dy feeds row i of the product and nothing else.
    int va[4], vb[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        va[q] = (int)((unsigned)(4 * lrow + q) * ulda + (unsigned)(i0 + lcol) * 4u);
        vb[q] = (int)((unsigned)(4 * lrow + q) * uldb + (unsigned)(j0 + lcol) * 4u);
    }
    const bool rs_block = w.db != nullptr && tile_j == 0;           // block-uniform: this tile also sums dy's columns
    f32x16 acc[FM][FN];
#pragma unroll
    for (int a = 0; a < FM; ++a)
#pragma unroll
        for (int b = 0; b < FN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    float rs[FM];
#pragma unroll
    for (int a = 0; a < FM; ++a) rs[a] = 0.f;

    const int nchunks = (M + 7) >> 3;
    const int n_it = (nchunks + KW - 1) / KW;                       // per wave; chunks kg, kg + KW, ... (surplus ones read zeros)
    float a0[PD][FM][4], b0[PD][FN][4], a1[PD][FM][4], b1[PD][FN][4];
    auto load_set = [&](int first, float (&as)[PD][FM][4], float (&bs)[PD][FN][4]) {
#pragma unroll
        for (int p = 0; p < PD; ++p) {
            const unsigned c = (unsigned)(kg + (first + p) * KW);    // chunk index (scalar)
            const i32x4_t ra = wb2_rsrc(ba, c * 8u * ulda, ext_a), rb = wb2_rsrc(bb, c * 8u * uldb, ext_b);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#if MVAE_WB2_KO == 2        /* knock-out (results wrong): no loads -- the MFMA stream alone */
                (void)ra; (void)rb;
#pragma unroll
                for (int f = 0; f < FM; ++f) asm volatile("" : "=v"(as[p][f][q]));
#pragma unroll
                for (int f = 0; f < FN; ++f) asm volatile("" : "=v"(bs[p][f][q]));
#else
#pragma unroll
                for (int f = 0; f < FM; ++f) as[p][f][q] = buf_load1(ra, va[q] + 128 * f);
#pragma unroll
                for (int f = 0; f < FN; ++f) bs[p][f][q] = buf_load1(rb, vb[q] + 128 * f);
#endif
            }
        }
    };
    auto use_set = [&](auto with_rs, const float (&as)[PD][FM][4], const float (&bs)[PD][FN][4]) {
#pragma unroll
        for (int p = 0; p < PD; ++p) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int f = 0; f < FM; ++f)
#pragma unroll
                    for (int g = 0; g < FN; ++g) {
#if MVAE_WB2_KO == 1        /* knock-out (results wrong): no MFMAs -- the load stream alone (the values are 'used') */
                        asm volatile("" :: "v"(as[p][f][q]), "v"(bs[p][g][q]));
#else
                        acc[f][g] = __builtin_amdgcn_mfma_f32_32x32x2f32(as[p][f][q], bs[p][g][q], acc[f][g], 0, 0, 0);
#endif
                    }
            if constexpr (decltype(with_rs)::value) {
#pragma unroll
                for (int f = 0; f < FM; ++f) rs[f] += (as[p][f][0] + as[p][f][1]) + (as[p][f][2] + as[p][f][3]);
            }
        }
    };
    // two register sets: one is multiplied while the other is in flight (see wgrad_direct_tile on why not a rotating one).
    // The column sums of dy (the bias gradient) are the business of a layer's first j tile only: TWO copies of the loop
    // -- under one `if (rs_block)` inside it the adds were if-converted into every tile's loop (adds + selects: 0.6 - 1.2
    // vector instructions per MFMA).
    auto main_loop = [&](auto with_rs) {
        load_set(0, a0, b0);
        for (int it = 0; it < n_it; it += 2 * PD) {
            load_set(it + PD, a1, b1);
            __builtin_amdgcn_sched_barrier(0);
            use_set(with_rs, a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            load_set(it + 2 * PD, a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            use_set(with_rs, a1, b1);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    if (rs_block) main_loop(std::true_type{});
    else main_loop(std::false_type{});
    // ---- the KW partial tiles through LDS in halves (the upper half of the waves parks, the lower half adds: a fixed
    //      tree), the last sum back to LDS, epilogue with every thread
    constexpr int TP = TN + 1;
    constexpr int WTILE = TM * TP;                                  // floats per parked wave tile
    auto park = [&](float *dst) {
#pragma unroll
        for (int f = 0; f < FM; ++f)
#pragma unroll
            for (int g = 0; g < FN; ++g)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    dst[(32 * f + 4 * lrow + (r & 3) + 8 * (r >> 2)) * TP + 32 * g + lcol] = acc[f][g][r];
    };
    float *rsl = wd_lds + (KW > 1 ? KW / 2 : 1) * WTILE;           // [KW][2][TM] partial column sums of dy
    if (rs_block) {
#pragma unroll
        for (int f = 0; f < FM; ++f) rsl[(kg * 2 + lrow) * TM + 32 * f + lcol] = rs[f];
    }
#pragma unroll
    for (int half = KW / 2; half >= 1; half >>= 1) {
        if (kg >= half && kg < 2 * half) park(wd_lds + (kg - half) * WTILE);
        __syncthreads();
        if (kg < half) {
            const float *src = wd_lds + kg * WTILE;
#pragma unroll
            for (int f = 0; f < FM; ++f)
#pragma unroll
                for (int g = 0; g < FN; ++g)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        acc[f][g][r] += src[(32 * f + 4 * lrow + (r & 3) + 8 * (r >> 2)) * TP + 32 * g + lcol];
        }
        __syncthreads();
    }
    if (kg == 0) park(wd_lds);
    __syncthreads();
    float *dw = w.dw;
    const int acc_flag = w.accumulate;
    constexpr int NE = TM * TN / (64 * KW);
#pragma unroll
    for (int k = 0; k < NE; ++k) {
        const int el = t + k * 64 * KW;
        const int il = el / TN, jl = el % TN;
        const int i = i0 + il, j = j0 + jl;
        if (i < I && j < J) {
            float v = wd_lds[il * TP + jl];
            float *dst = dw + (size_t)i * J + j;
            if (acc_flag) v += *dst;
            *dst = v;
        }
    }
    if (rs_block && t < TM && i0 + t < I) {
        float s2 = 0.f;
#pragma unroll
        for (int g2 = 0; g2 < 2 * KW; ++g2) s2 += rsl[g2 * TM + t];
        if (acc_flag) s2 += w.db[i0 + t];
        w.db[i0 + t] = s2;
    }
}

template <int KW, int FM, int FN, int PD>
__global__ __launch_bounds__(64 * KW) void wgrad_batched2_kernel(WgradBatch2Args a) {
    // XCD x = blocks x, x + 8, ... of the launch order; it walks segment x of the tile list
    const int xcd = blockIdx.x & (WB2_SEGS - 1), slot = blockIdx.x >> 3;
    const int tile = a.seg[xcd] + slot;
    if (tile >= a.seg[xcd + 1]) return;
    if (MVAE_WB2_PRIO) __builtin_amdgcn_s_setprio(MVAE_WB2_PRIO);
    int p = 0, first = 0;
#pragma unroll 1
    for (int q = 0; q < a.n - 1; ++q) {
        if (tile >= a.it[q].tile_end) { p = q + 1; first = a.it[q].tile_end; }
    }
    const WgradBatchItem &w = a.it[p];
    const int local = tile - first;
    const int tile_i = local / w.tiles_j, tile_j = local - tile_i * w.tiles_j;
    wgrad_tile2<KW, FM, FN, PD>(w, tile_i, tile_j);
}

// Re-tile the table for FM x FN wave tiles, cut it into XCD segments, launch.  Returns false when the batch has an
// item the v2 tile code does not take (none today: kept for the Adam-fused form, which stays on the old kernel).
inline bool wgrad_batched2_launch(const WgradBatchArgs &a, hipStream_t st, int *status) {
    if (!MVAE_WGRAD2) return false;
    long t22 = 0, t21 = 0, t11 = 0;
    int max_m = 0;
    for (int q = 0; q < a.n; ++q) {
        const WgradBatchItem &w = a.it[q];
        if (!w.dy) return false;
        t22 += cdiv(w.I, 64) * cdiv(w.J, 64); t21 += cdiv(w.I, 64) * cdiv(w.J, 32); t11 += cdiv(w.I, 32) * cdiv(w.J, 32);
        if (w.M > max_m) max_m = w.M;
        // byte offsets of (surplus) chunks stay below 2^31 (signed in wb2_rsrc): with KW waves per tile and PD register sets
        // the prefetch runs up to 8 * KW * (3 * PD + 1) rows past the batch -- KW <= 16, PD <= 2 (8 * 16 * 5 was short)
        const long over = (long)w.M + 8 * 16 * 8;
        if (over * w.lddy * 4 >= (1L << 31) || over * w.ldx * 4 >= (1L << 31)) return false;
    }
    // Wave tile: every tile of a launch costs the same (M rows x tile area), blocks spread evenly over the 256 CUs, and the
    // launch lasts as long as its busiest CU -- ceil(tiles / 256) tiles of FM x FN MFMAs per 8 rows.  Pick the shape with
    // the least of that; between equals the larger tile (fewer loads per MFMA).  (MNIST, session 3 of round 5, the launch
    // alone, hot: image decoder 936 / 468 / 240 tiles -> 4 / 4 / 4 units: 64 x 64 wins, 24.6 us against 26.4 (32 x 32)
    // and 28.8 for the round-4 kernel; label decoder 560 / 288 / 144 tiles -> 3 / 4 / 4: 32 x 32 wins, 19.6 against 22.5.)
    const long busy22 = cdiv(t22, 256) * 4, busy21 = cdiv(t21, 256) * 2, busy11 = cdiv(t11, 256);
    int shape = 22;
    long busy = busy22;
    if (busy21 < busy) { shape = 21; busy = busy21; }
    if (busy11 < busy) { shape = 11; busy = busy11; }
    if (MVAE_WGRAD2_SHAPE) shape = MVAE_WGRAD2_SHAPE;
    const long tiles = shape == 22 ? t22 : (shape == 21 ? t21 : t11);
    const int fm = shape == 11 ? 1 : 2, fn = shape == 22 ? 2 : 1;
    // waves per tile (they split the batch rows): ~2 per SIMD over the launch, at least 4 chunks of 8 rows each
    int kw = tiles >= 1024 ? 2 : (tiles >= 400 ? 4 : (tiles >= 160 ? 8 : 16));
    if (shape == 22 && kw < 4) kw = 4;
    while (kw > 2 && max_m < 8 * 4 * kw) kw >>= 1;
    if (shape == 11 && kw < 4) kw = 4;
    if (MVAE_WB2_KW) kw = MVAE_WB2_KW;
    WgradBatch2Args b;
    b.n = a.n;
    int total = 0;
    for (int q = 0; q < a.n; ++q) {
        b.it[q] = a.it[q];
        b.it[q].tiles_j = (int)cdiv(a.it[q].J, 32 * fn);
        total += (int)cdiv(a.it[q].I, 32 * fm) * b.it[q].tiles_j;
        b.it[q].tile_end = total;
    }
    int longest = 0;
    for (int x = 0; x <= WB2_SEGS; ++x) b.seg[x] = (int)((long)total * x / WB2_SEGS);
    for (int x = 0; x < WB2_SEGS; ++x) if (b.seg[x + 1] - b.seg[x] > longest) longest = b.seg[x + 1] - b.seg[x];
    const dim3 grid((unsigned)(longest * WB2_SEGS));
#define MVAE_WB2(KWV, FMV, FNV, PDV)                                                                          \
    {                                                                                                         \
        constexpr size_t lds = ((size_t)(KWV > 1 ? KWV / 2 : 1) * (32 * FMV) * (32 * FNV + 1) + (size_t)KWV * 2 * 32 * FMV) * sizeof(float); \
        auto kern = wgrad_batched2_kernel<KWV, FMV, FNV, PDV>;                                                \
        static bool attr_done = false;                                                                        \
        if (!attr_done) {                                                                                     \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            attr_done = true;                                                                                 \
        }                                                                                                     \
        hipLaunchKernelGGL(kern, grid, dim3(64 * KWV), lds, st, b);                                           \
    }
    if (shape == 22) {
        if (kw >= 16) MVAE_WB2(16, 2, 2, 1) else if (kw == 8) MVAE_WB2(8, 2, 2, MVAE_WB2_PD22) else MVAE_WB2(4, 2, 2, MVAE_WB2_PD22)
    } else if (shape == 21) {
        if (kw >= 16) MVAE_WB2(16, 2, 1, 2) else if (kw == 8) MVAE_WB2(8, 2, 1, 2) else if (kw == 4) MVAE_WB2(4, 2, 1, 3) else MVAE_WB2(2, 2, 1, 3)
    } else {
        if (kw >= 16) MVAE_WB2(16, 1, 1, MVAE_WB2_PD11) else if (kw == 8) MVAE_WB2(8, 1, 1, MVAE_WB2_PD11) else MVAE_WB2(4, 1, 1, MVAE_WB2_PD11)
    }
#undef MVAE_WB2
    *status = mvae_launch_status();
    return true;
}

// shapes the direct tile code serves well on its own merits (32-bit byte offsets inside the operands included)
inline bool wgrad_batch_item_ok(int I, int J, int M, int lddy, int ldx) {
    const long tiles = cdiv(I, 32) * cdiv(J, 32);
    return tiles <= 2048 && M <= 4096 && (long)M * lddy * 4 < (1L << 32) && (long)M * ldx * 4 < (1L << 32);
}

inline int wgrad_batched_launch(WgradBatchArgs &a, hipStream_t st, const AdamFuse *adam = nullptr) {
    if (!adam) {
        int status = MVAE_OK;
        if (wgrad_batched2_launch(a, st, &status)) return status;
    }
    const int total = a.it[a.n - 1].tile_end;
    // waves per tile: enough blocks x waves to put ~4 waves on every SIMD, a reduction slice of >= 4 chunks each
    const int kw = total >= 768 ? 4 : (total >= 256 ? 8 : 16);
#define MVAE_WB(KWV)                                                                                          \
    {                                                                                                         \
        constexpr size_t lds = ((size_t)KWV * 32 * 33 + (size_t)KWV * 64) * sizeof(float);                    \
        static bool attr_done = false;                                                                        \
        if (!attr_done) {                                                                                     \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(wgrad_batched_kernel<KWV>),              \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                  \
            attr_done = true;                                                                                 \
        }                                                                                                     \
        static bool attr_done_adam = false;                                                                   \
        if (adam && !attr_done_adam) {                                                                        \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(wgrad_batched_adam_kernel<KWV>),         \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                  \
            attr_done_adam = true;                                                                            \
        }                                                                                                     \
        if (adam) {                                                                                           \
            WgradBatchAdamArgs aa;                                                                            \
            aa.w = a; aa.adam = *adam;                                                                        \
            hipLaunchKernelGGL(wgrad_batched_adam_kernel<KWV>, dim3(total), dim3(64 * KWV), lds, st, aa);     \
        } else                                                                                                \
            hipLaunchKernelGGL(wgrad_batched_kernel<KWV>, dim3(total), dim3(64 * KWV), lds, st, a);           \
    }
    if (kw == 4) MVAE_WB(4) else if (kw == 8) MVAE_WB(8) else MVAE_WB(16)
#undef MVAE_WB
    return mvae_launch_status();
}

// one problem (no groups); returns false when the shape is better served by the tiled kernel
inline bool wgrad_direct_ok(int I, int J, int M) {
    const long tiles = cdiv(I, 32) * cdiv(J, 32);
    if (MVAE_TUNE(small_off) || MVAE_TUNE(wm) || MVAE_TUNE(wn) || MVAE_TUNE(kw) || MVAE_TUNE(splits)) return false;
    // enough tiles to occupy the chip, a reduction short enough that KW <= 16 waves per tile cover it
    return tiles >= 16 && tiles <= 2048 && M <= 4096;
}

inline int wgrad_direct_launch(const float *dy, int lddy, const float *x, int ldx, EpRowMajor e, int I, int J, int M,
                               float *db, int db_accumulate, hipStream_t st) {
    const long tiles = cdiv(I, 32) * cdiv(J, 32);
    const dim3 grid((unsigned)cdiv(J, 32), (unsigned)cdiv(I, 32));
    int kw = tiles >= 192 ? 4 : (tiles >= 96 ? 8 : 16);
    if (MVAE_TUNE(small_waves)) kw = MVAE_TUNE(small_waves);
#define MVAE_WD(RS, KWV)                                                                                      \
    {                                                                                                         \
        constexpr size_t lds = ((size_t)KWV * 32 * 33 + (size_t)KWV * 64) * sizeof(float);                    \
        auto kern = MVAE_TUNE(knockout) == 1 ? wgrad_direct_kernel<RS, KWV, MVAE_KO1>                          \
                  : MVAE_TUNE(knockout) == 2 ? wgrad_direct_kernel<RS, KWV, MVAE_KO2> : wgrad_direct_kernel<RS, KWV, 0>; \
        static bool attr_done = false;                                                                        \
        if (!attr_done) {                                                                                     \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern),                                   \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                  \
            attr_done = true;                                                                                 \
        }                                                                                                     \
        hipLaunchKernelGGL(kern, grid, dim3(64 * KWV), lds, st, dy, lddy, x, ldx, e, I, J, M, db, db_accumulate); \
    }
    if (db) {
        if (kw == 4) MVAE_WD(true, 4) else if (kw == 8) MVAE_WD(true, 8) else MVAE_WD(true, 16)
    } else {
        if (kw == 4) MVAE_WD(false, 4) else if (kw == 8) MVAE_WD(false, 8) else MVAE_WD(false, 16)
    }
#undef MVAE_WD
    return mvae_launch_status();
}

}  // namespace
