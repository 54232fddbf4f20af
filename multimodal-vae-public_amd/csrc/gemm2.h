// gemm2.h -- GEMM core, version 2 (round 6): the large GEMM-shaped launches of the train step.
//
// Same contraction as gemm_core.h's igemm_kernel -- D[i][j] = sum_k P(i,k) * Q(k,j) on v_mfma_f32_32x32x2_f32, exact
// fp32, the same epilogue functors -- with a different machine underneath:
//
//   * BOTH operands go global -> LDS by LDS-DMA (`buffer_load_dword[x4] ... lds`): no register hop, no ds_write, no
//     staging registers.  A k-tile is 16 deep; the LDS ring holds 2 or 3 of them; ONE raw s_barrier per k-tile and a
//     counted `s_waitcnt vmcnt(N)` that leaves the younger tiles in flight across it.  (The DMA is issued from inline
//     asm: told about it through the builtin, hipcc waits vmcnt(0) before the next ds_read of the array -- it cannot
//     tell the ring stages apart -- and the prefetch drains every k-tile: tools/gemm2_probe.hip.)
//   * wave tiles of up to 64 x 64 (2 x 2 MFMA tiles: every fragment read from LDS feeds two matrix instructions),
//     block tiles 64 x 64 ... 128 x 128 on 4 waves.
//   * row operands with k contiguous land as [row][16] with the four float4 slots of a row XOR-swizzled on the SOURCE
//     address (the DMA destination is lane-linear), so that the ds_read_b128 fragment reads are conflict-free without
//     padding; operands whose tile axis is contiguous -- and every gather -- land k-major [16][tile], ds_read_b32.
//   * PERSISTENT blocks and a balanced schedule: the launch is `G` blocks (a multiple of the 256 CUs); a block first
//     walks its share of WHOLE tiles, then its share of the k-tiles of the leftover tiles ("stream-K" on the tail only):
//     a leftover tile's reduction is cut wherever a block's share ends, the pieces go to scratch slabs and
//     g2_finish_kernel sums them in a fixed order and runs the epilogue -- deterministic, no atomics.  The software
//     pipeline runs ACROSS tiles: the next tile's first k-tiles are in flight while the current tile finishes.
//     Every block does the same number of k-tiles (+-1): no block-count quantisation (profiles/r05_grid_report.txt:
//     0.77-0.81 balance on the 400-800-block launches of the 64 x 64 kernels).
#pragma once
#include "gemm_core.h"

namespace {

typedef __attribute__((address_space(3))) void g2_lds_void;

// 16 bytes per lane: global (descriptor + per-lane byte offset) -> LDS byte address m0 + 16 * lane
__device__ __forceinline__ void g2_dma16(i32x4_t rs, int voff, unsigned lds_byte) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rs), "s"(lds_byte) : "memory");
}
// 4 bytes per lane, with a scalar byte offset on top of the per-lane one: -> LDS byte address m0 + 4 * lane
__device__ __forceinline__ void g2_dma4(i32x4_t rs, int voff, int soff, unsigned lds_byte) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rs), "s"(soff), "s"(lds_byte) : "memory");
}
// (every word goes through readfirstlane: block-uniform values that come out of an integer division live in vector
//  registers, and the asm statements above take scalar operands only)
__device__ __forceinline__ int g2_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ i32x4_t g2_rsrc(BufBase b, long floats, int records) {
    const unsigned long long a = (((unsigned long long)b.hi << 32) | b.lo) + (unsigned long long)floats * 4ull;
    i32x4_t r;
    r.x = g2_uni((int)(unsigned)a); r.y = g2_uni((int)((unsigned)(a >> 32) & 0xffffu)); r.z = g2_uni(records); r.w = 0x00020000;
    return r;
}

constexpr int G2_BK = 16;

// swizzle of the [row][16] image: float4 slot f of row r sits at slot f ^ ((r >> 2) & 3).  A ds_read_b128 is served in lane
// groups of 16 ({0-3, 12-15, 20-27}, ... MI355X_MICROARCH.md): the rows of a group that share r % 4 (the 64-byte row
// pitch puts them on the same 16 banks) differ in (r >> 2) & 3, so the group covers all 64 banks exactly once.
__device__ __forceinline__ int g2_swz(int r) { return (r >> 2) & 3; }

// ---- loaders.  init(tile0, cls, lane, wave): point at a tile.  issue(lds_byte, k0, kend, wave): DMA of the k-tile
// [k0, k0 + 16) into the operand's LDS image (elements at or beyond kend read as zero).  NPW: DMA instructions per wave.
// RK: which image (see above).

// S[r * ld + k]: reduction axis contiguous (gemm_core.h LdRowsKT).  Needs 16-byte alignment, ld % 4 == 0, Klen % 4 == 0.
template <int TILE_>
struct G2RowsK {
    static constexpr int TILE = TILE_;
    static constexpr bool RK = true;
    static constexpr int NPW = TILE / 64;                   // TILE * 4 float4 slots / 64 lanes / 4 waves
    const float *src; int ld; int R; int Klen; size_t cls_stride = 0;
    BufBase blk; int voff[NPW]; int kcol;                    // kcol: first k of the lane's float4 inside a k-tile
    __device__ void init(int tile0, int cls, int lane, int wave) {
        blk = buf_base(src + (size_t)cls * cls_stride + (size_t)tile0 * ld);
#pragma unroll
        for (int u = 0; u < NPW; ++u) {
            const int slot = (wave * NPW + u) * 64 + lane;
            const int r = slot >> 2, f = (slot & 3) ^ g2_swz(r);
            voff[u] = (tile0 + r < R) ? (r * ld + f * 4) * 4 : BUF_OOB;
            if (u == 0) kcol = f * 4;                        // rows of one lane's slots differ by 16: same swizzle, same f
        }
    }
    __device__ __forceinline__ void issue(unsigned dst, int k0, int kend, int wave) const {
        const i32x4_t rs = g2_rsrc(blk, k0, 0x7fffffff);
        const bool cut = k0 + kcol >= kend;                  // partial last k-tile (Klen % 4 == 0: whole float4s)
#pragma unroll
        for (int u = 0; u < NPW; ++u) g2_dma16(rs, cut ? BUF_OOB : voff[u], g2_uni(dst + (wave * NPW + u) * 1024));
    }
};

// S[k * ld + r]: tile axis contiguous (gemm_core.h LdRowsMNT).  Needs 16-byte alignment, ld % 4 == 0, R % 4 == 0.
template <int TILE_>
struct G2RowsMN {
    static constexpr int TILE = TILE_;
    static constexpr bool RK = false;
    static constexpr int NPW = TILE / 64;
    static constexpr int V4 = TILE / 4;
    const float *src; int ld; int R; int Klen; size_t cls_stride = 0;
    BufBase blk; int voff[NPW];
    __device__ void init(int tile0, int cls, int lane, int wave) {
        blk = buf_base(src + (size_t)cls * cls_stride + tile0);
#pragma unroll
        for (int u = 0; u < NPW; ++u) {
            const int slot = (wave * NPW + u) * 64 + lane;
            const int k = slot / V4, n = (slot % V4) * 4;
            voff[u] = (tile0 + n < R) ? (k * ld + n) * 4 : BUF_OOB;
        }
    }
    __device__ __forceinline__ void issue(unsigned dst, int k0, int kend, int wave) const {
        // the descriptor ends with row kend - 1: rows beyond the reduction read as zero in hardware
        const long left = (long)(kend - k0) * ld * 4;
        const i32x4_t rs = g2_rsrc(blk, (long)k0 * ld, (int)(left < 0x7fffffffl ? left : 0x7fffffffl));
#pragma unroll
        for (int u = 0; u < NPW; ++u) g2_dma16(rs, voff[u], g2_uni(dst + (wave * NPW + u) * 1024));
    }
};

// ---- the launch's work list (see the header): all block-uniform integers
struct G2Args {
    int I, J, K;                  // output rows / columns, reduction length
    int nk;                       // k-tiles per tile
    int tiles_i, tiles_j, ncls;   // tile grid per class, classes (parity classes / groups)
    int cls_minor;                // 1: tile order (j tile, class, i tile), 0: (class, j tile, i tile); i fastest either way
    int G;                        // persistent blocks
    int q;                        // whole tiles per block (tiles 0 .. q * G - 1)
    int R;                        // leftover tiles (q * G .. q * G + R - 1), cut along k
    int upb;                      // leftover k-tiles per block
    int cmax;                     // slabs reserved per leftover tile
    float *ws;                    // slabs: [R][cmax][BM * BN (+ BM row sums)]
    float *rowsum; size_t rowsum_cls_stride; int rowsum_accumulate;   // ROWSUM: bias gradient (sum over k of P) per class
};
struct G2Seg { int tile, kb, ke, slab; };      // slab < 0: the whole reduction -> direct epilogue

struct G2Walk {
    int lb, r, u0, u1;
    __device__ void init(const G2Args &a, int lb_) {
        lb = lb_; r = 0;
        const long tu = (long)a.R * a.nk;
        long b0 = (long)lb * a.upb, b1 = b0 + a.upb;
        if (b0 > tu) b0 = tu;
        if (b1 > tu) b1 = tu;
        u0 = (int)b0; u1 = (int)b1;
    }
    __device__ int units(const G2Args &a) const { return a.q * a.nk + (u1 - u0); }
    __device__ bool next(const G2Args &a, G2Seg &s) {
        if (r < a.q) { s.tile = r * a.G + lb; s.kb = 0; s.ke = a.nk; s.slab = -1; ++r; return true; }
        if (u0 >= u1) return false;
        const int tt = u0 / a.nk, kb = u0 - tt * a.nk;
        const int ke = min(a.nk, kb + (u1 - u0));
        s.tile = a.q * a.G + tt; s.kb = kb; s.ke = ke;
        s.slab = (kb == 0 && ke == a.nk) ? -1 : tt * a.cmax + (lb - (tt * a.nk) / a.upb);
        u0 += ke - kb;
        return true;
    }
};
__device__ __forceinline__ void g2_tile_coords(const G2Args &a, int tile, int &cls, int &ti, int &tj) {
    ti = tile % a.tiles_i;
    const int rest = tile / a.tiles_i;
    if (a.cls_minor) { tj = rest / a.ncls; cls = rest - tj * a.ncls; }
    else { cls = rest / a.tiles_j; tj = rest - cls * a.tiles_j; }
}
// launch slot -> logical block: XCD x (slots x, x + 8, ...) owns the contiguous range [x * G / 8, (x + 1) * G / 8) of
// logical blocks, i.e. of every round's tiles and of the leftover k range: neighbours in the work list share operand
// rows in ONE L2, and the pieces of a cut tile mostly meet on one XCD
__device__ __forceinline__ int g2_logical_block(int b, int G) { return (G & 7) ? b : (b & 7) * (G >> 3) + (b >> 3); }

template <class P, class Q, class E, int WM, int WN, bool ROWSUM, int STAGES, int MINW>
__global__ __launch_bounds__(256, MINW)
void gemm2_kernel(P p, Q q, E e, G2Args a) {
    constexpr int BM = 64 * WM, BN = 64 * WN, BK = G2_BK;
    static_assert(P::TILE == BM && Q::TILE == BN, "loader tile mismatch");
    static_assert(!ROWSUM || !P::RK, "row sums read a k-major P image");
    constexpr int P_FLOATS = BM * BK, Q_FLOATS = BN * BK, STAGE_FLOATS = P_FLOATS + Q_FLOATS;
    constexpr int NPW = P::NPW + Q::NPW;
    constexpr int RS_PARTS = 256 / BM;
    constexpr int SLAB = BM * BN + (ROWSUM ? BM : 0);
    extern __shared__ __attribute__((aligned(16))) float lds[];      // ring [STAGES][P image | Q image] (+ ROWSUM scratch)
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wi = wave >> 1, wj = wave & 1;
    const int lrow = lane >> 5, lcol = lane & 31;
    const int lb = g2_logical_block(blockIdx.x, a.G);
    const unsigned lds0 = (unsigned)(unsigned long)(g2_lds_void *)lds;

    f32x16 acc[WM][WN];
#pragma unroll
    for (int x = 0; x < WM; ++x)
#pragma unroll
        for (int y = 0; y < WN; ++y)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.f;

    // fragment addresses (float index inside an operand image); both layouts: the j-th MFMA of the 8-k chunk c sums
    // k = 8c + j in lanes 0-31 and k = 8c + 4 + j in lanes 32-63
    const int pbase = P::RK ? (wi * 32 * WM + lcol) * BK : 4 * lrow * BM + wi * 32 * WM + lcol;
    const int qbase = Q::RK ? (wj * 32 * WN + lcol) * BK : 4 * lrow * BN + wj * 32 * WN + lcol;
    const int fswz = g2_swz(lcol);                       // row = 32-multiple + lcol: the swizzle is the lane's
    const int rs_row = t % BM, rs_part = t / BM;
    float rsum = 0.f;

    auto compute = [&](int stage, bool rs_tile) {
        const float *Ps = lds + stage * STAGE_FLOATS;
        const float *Qs = Ps + P_FLOATS;
        if (ROWSUM) {
            if (rs_tile) {
#pragma unroll
                for (int kk = 0; kk < BK / RS_PARTS; ++kk) rsum += Ps[(kk * RS_PARTS + rs_part) * BM + rs_row];
            }
        }
        float4 pa[WM], qb[WN], pa_n[WM], qb_n[WN];
        auto rd = [&](int c, float4 (&pa_)[WM], float4 (&qb_)[WN]) {
            if (P::RK) {
#pragma unroll
                for (int x = 0; x < WM; ++x) pa_[x] = *reinterpret_cast<const float4 *>(Ps + pbase + x * 32 * BK + 4 * ((2 * c + lrow) ^ fswz));
            }
            if (Q::RK) {
#pragma unroll
                for (int y = 0; y < WN; ++y) qb_[y] = *reinterpret_cast<const float4 *>(Qs + qbase + y * 32 * BK + 4 * ((2 * c + lrow) ^ fswz));
            }
        };
        rd(0, pa, qb);
#pragma unroll
        for (int c = 0; c < BK / 8; ++c) {
            if (c + 1 < BK / 8) rd(c + 1, pa_n, qb_n);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                float av[WM], bv[WN];
#pragma unroll
                for (int x = 0; x < WM; ++x)
                    av[x] = P::RK ? (s == 0 ? pa[x].x : s == 1 ? pa[x].y : s == 2 ? pa[x].z : pa[x].w)
                                  : Ps[pbase + (8 * c + s) * BM + x * 32];
#pragma unroll
                for (int y = 0; y < WN; ++y)
                    bv[y] = Q::RK ? (s == 0 ? qb[y].x : s == 1 ? qb[y].y : s == 2 ? qb[y].z : qb[y].w)
                                  : Qs[qbase + (8 * c + s) * BN + y * 32];
#pragma unroll
                for (int x = 0; x < WM; ++x)
#pragma unroll
                    for (int y = 0; y < WN; ++y) acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[x], bv[y], acc[x][y], 0, 0, 0);
            }
#pragma unroll
            for (int x = 0; x < WM; ++x) pa[x] = pa_n[x];
#pragma unroll
            for (int y = 0; y < WN; ++y) qb[y] = qb_n[y];
        }
    };

    // ---- what a finished segment does with its accumulators
    auto finish_segment = [&](const G2Seg &s) {
        int cls, ti, tj;
        g2_tile_coords(a, s.tile, cls, ti, tj);
        const int i0 = ti * BM, j0 = tj * BN;
        const bool rs_tile = ROWSUM && tj == 0;
        float rtot = 0.f;
        if (ROWSUM) {
            if (rs_tile) {                // block-uniform: RS_PARTS partial sums per row, combined in a fixed order
                float *red = lds + STAGES * STAGE_FLOATS;
                red[rs_part * BM + rs_row] = rsum;
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                if (t < BM) {
#pragma unroll
                    for (int pp = 0; pp < RS_PARTS; ++pp) rtot += red[pp * BM + t];
                }
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                rsum = 0.f;
            }
        }
        if (s.slab >= 0) {
            float *slab = a.ws + (size_t)s.slab * SLAB;
#pragma unroll
            for (int y = 0; y < WN; ++y)
#pragma unroll
                for (int x = 0; x < WM; ++x)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int il = (wi * WM + x) * 32 + 4 * lrow + (r & 3) + 8 * (r >> 2);
                        slab[il * BN + (wj * WN + y) * 32 + lcol] = acc[x][y][r];
                    }
            if (ROWSUM) {
                if (rs_tile && t < BM) slab[BM * BN + t] = rtot;
            }
        } else {
            E et = e;
            et.set_class(cls);
            if constexpr (ep_buffer<E>::value) {
                et.tile(j0);
#pragma unroll
                for (int y = 0; y < WN; ++y) {
                    (void)et.col(j0 + (wj * WN + y) * 32 + lcol);
#pragma unroll
                    for (int x = 0; x < WM; ++x) {
                        const int rb = __builtin_amdgcn_readfirstlane(i0 + (wi * WM + x) * 32);
#pragma unroll
                        for (int r = 0; r < 16; ++r) et.put_b(rb, r, acc[x][y][r]);
                    }
                }
            } else {
#pragma unroll
                for (int y = 0; y < WN; ++y) {
                    const int j = j0 + (wj * WN + y) * 32 + lcol;
                    if (!et.col(j)) continue;
#pragma unroll
                    for (int x = 0; x < WM; ++x) {
                        const int ib = i0 + (wi * WM + x) * 32 + 4 * lrow;
#pragma unroll
                        for (int r = 0; r < 16; ++r) et.put(ib + (r & 3) + 8 * (r >> 2), j, acc[x][y][r]);
                    }
                }
            }
            if (ROWSUM) {
                if (rs_tile && t < BM && i0 + t < a.I) {
                    float *dst = a.rowsum + (size_t)cls * a.rowsum_cls_stride + i0 + t;
                    if (a.rowsum_accumulate) rtot += *dst;
                    *dst = rtot;
                }
            }
        }
#pragma unroll
        for (int x = 0; x < WM; ++x)
#pragma unroll
            for (int y = 0; y < WN; ++y)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.f;
    };

    // ---- the two cursors over the block's k-tile stream: the DMA runs STAGES - 1 k-tiles ahead of the MFMAs
    G2Walk wis, wcs;
    wis.init(a, lb); wcs.init(a, lb);
    const int NU = wcs.units(a);
    if (NU <= 0) return;
    G2Seg si, sc;
    bool have_i = wis.next(a, si);
    int ki = si.kb;
    auto point_loaders = [&](const G2Seg &s) {
        int cls, ti, tj;
        g2_tile_coords(a, s.tile, cls, ti, tj);
        p.init(ti * BM, cls, lane, wave);
        q.init(tj * BN, cls, lane, wave);
    };
    point_loaders(si);
    auto issue_next = [&](int stage) {
        if (!have_i) return;
        asm volatile("s_nop 4" ::: "memory");       // descriptor words may come fresh from readfirstlane
        p.issue(lds0 + stage * STAGE_FLOATS * 4, ki * BK, a.K, wave);
        q.issue(lds0 + (stage * STAGE_FLOATS + P_FLOATS) * 4, ki * BK, a.K, wave);
        if (++ki == si.ke) {
            have_i = wis.next(a, si);
            if (have_i) { ki = si.kb; point_loaders(si); }
        }
    };
    (void)wcs.next(a, sc);
    int kc = sc.kb;
    int cls_c, ti_c, tj_c;
    g2_tile_coords(a, sc.tile, cls_c, ti_c, tj_c);
    bool rs_c = ROWSUM && tj_c == 0;

    int st_i = 0;                                    // ring stage of the next issue
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s) {
        if (s < NU) { issue_next(st_i); st_i = (st_i + 1 == STAGES) ? 0 : st_i + 1; }
    }
    int st_c = 0;
    for (int u = 0; u < NU; ++u) {
        // k-tile u must have landed; the STAGES - 2 tiles issued after it may stay in flight
        if (STAGES == 3 && u + 2 <= NU) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(NPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        // behind the barrier every wave has finished reading stage (u - 1) % STAGES: refill it
        if (u + STAGES - 1 < NU) { issue_next(st_i); st_i = (st_i + 1 == STAGES) ? 0 : st_i + 1; }
        compute(st_c, rs_c);
        st_c = (st_c + 1 == STAGES) ? 0 : st_c + 1;
        if (++kc == sc.ke) {
            finish_segment(sc);
            if (wcs.next(a, sc)) {
                kc = sc.kb;
                g2_tile_coords(a, sc.tile, cls_c, ti_c, tj_c);
                rs_c = ROWSUM && tj_c == 0;
            }
        }
    }
}

// Sum the pieces of the leftover tiles in block order and run the epilogue.  Grid (leftover tile, row chunk); a thread owns
// four consecutive columns of one row.
template <class E, int BM, int BN, bool ROWSUM>
__global__ __launch_bounds__(256) void g2_finish_kernel(E e, G2Args a) {
    constexpr int SLAB = BM * BN + (ROWSUM ? BM : 0);
    constexpr int V4 = BN / 4, ROWS = 256 / V4;      // rows per block pass
    const int tt = blockIdx.x;
    const int first = (tt * a.nk) / a.upb, last = ((tt + 1) * a.nk - 1) / a.upb;
    if (first == last) return;                       // one block held the whole reduction: it ran the epilogue itself
    int cls, ti, tj;
    g2_tile_coords(a, a.q * a.G + tt, cls, ti, tj);
    const int il = blockIdx.y * ROWS + threadIdx.x / V4, jl = (threadIdx.x % V4) * 4;
    const int i = ti * BM + il, j = tj * BN + jl;
    const float *src = a.ws + (size_t)tt * a.cmax * SLAB + il * BN + jl;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    const int n = last - first + 1;
#pragma unroll 4
    for (int c = 0; c < n; ++c) {
        const float4 v = *reinterpret_cast<const float4 *>(src + (size_t)c * SLAB);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    E et = e;
    et.set_class(cls);
    if (i < a.I) {
        if (j < a.J && et.col(j)) et.put(i, j, s.x);
        if (j + 1 < a.J && et.col(j + 1)) et.put(i, j + 1, s.y);
        if (j + 2 < a.J && et.col(j + 2)) et.put(i, j + 2, s.z);
        if (j + 3 < a.J && et.col(j + 3)) et.put(i, j + 3, s.w);
    }
    if (ROWSUM) {
        if (tj == 0 && blockIdx.y == 0 && threadIdx.x < BM && ti * BM + threadIdx.x < a.I) {
            float r = 0.f;
            for (int c = 0; c < n; ++c) r += a.ws[((size_t)tt * a.cmax + c) * SLAB + BM * BN + threadIdx.x];
            float *dst = a.rowsum + (size_t)cls * a.rowsum_cls_stride + ti * BM + threadIdx.x;
            if (a.rowsum_accumulate) r += *dst;
            *dst = r;
        }
    }
}

// ---- host side
struct G2Plan { int wm, wn, occ, stages; bool ok; size_t ws_floats; G2Args a; };

#ifndef MVAE_G2_ACT_ONLY
#define MVAE_G2_ACT_ONLY 1        // 0: A/B builds
#endif
#ifndef MVAE_G2
#define MVAE_G2 1                 // 0: every launch stays on the round-1-5 kernels (A/B builds)
#endif

// tile shape by a cost model: a block's time is its k-tiles x the MFMAs of a k-tile, and a launch lasts as long as the
// busiest CU; larger tiles read less per flop (LDS fragments, L2 -> LDS bytes) and pay more for ragged edges and slabs
inline G2Plan g2_plan(int I, int J, int K, int ncls, bool rowsum, int force_wm = 0, int force_wn = 0, int force_occ = 0) {
    G2Plan best; best.ok = false;
    double best_cost = 1e300;
    const int nk = (K + G2_BK - 1) / G2_BK;
    for (int wm = 1; wm <= 2; ++wm)
        for (int wn = 1; wn <= 2; ++wn) {
            if ((force_wm && wm != force_wm) || (force_wn && wn != force_wn)) continue;
            if (rowsum && false) continue;
            const int BM = 64 * wm, BN = 64 * wn;
            const long ti = cdiv(I, BM), tj = cdiv(J, BN);
            const long T = ti * tj * ncls;
            if (T > 0x3fffffff / (nk > 0 ? nk : 1)) continue;
            int occ = (wm * wn == 4) ? 2 : (wm * wn == 2) ? 3 : 4;
            if (force_occ) occ = force_occ;
            const bool np = force_occ < 0;       // one tile per block, as many blocks as tiles, the hardware's own dispatch order
            if (np) occ = (wm * wn == 4) ? 3 : (wm * wn == 2) ? 4 : 6;
            const int G = np ? (int)T : 256 * occ;
            const long q = T / G, R = T - q * G;
            const long upb = R ? cdiv(R * nk, G) : 1;
            // per-block MFMA time in units of one 32x32x2 instruction per wave, plus what the slabs cost
            const double kt_cost = (double)wm * wn * 8;                 // MFMAs per wave per k-tile
            double per_block = (double)(q * nk + (R ? upb : 0)) * kt_cost;
            per_block *= occ;                                            // occ blocks share a CU's matrix pipes
            // efficiency of the tile shape (measured on 4096^3, tools/gemm2_probe): 128x128 1.00, 128x64 / 64x128 0.98, 64x64 0.955
            const double eff = (wm * wn == 4) ? 1.0 : (wm * wn == 2) ? 0.98 : 0.955;
            // fixed cost per block: prologue + one epilogue per tile piece (~stores of BM x BN / 256 per thread)
            const double epi = (double)(q + (R ? 2 : 0)) * wm * wn * 16 * 1.5 * occ;
            const double cost = per_block / eff + epi;
            if (cost < best_cost) {
                best_cost = cost;
                best.ok = true; best.wm = wm; best.wn = wn; best.occ = np ? 0 : occ; best.stages = 3;
                G2Args &a = best.a;
                a.I = I; a.J = J; a.K = K; a.nk = nk; a.tiles_i = (int)ti; a.tiles_j = (int)tj; a.ncls = ncls; a.cls_minor = 0;
                a.G = G; a.q = (int)q; a.R = (int)R; a.upb = (int)upb;
                a.cmax = R ? (int)(cdiv(nk, upb) + 1) : 0;
                a.ws = nullptr; a.rowsum = nullptr; a.rowsum_cls_stride = 0; a.rowsum_accumulate = 0;
                best.ws_floats = (size_t)R * a.cmax * ((size_t)BM * BN + (rowsum ? BM : 0));
            }
        }
    return best;
}

// The plan of a launch, or !ok (the launch stays on gemm_core.h's kernels).  What tools/g2_bench.py measured on MI355X
// (profiles/r06_g2_bench_*.txt) decides where version 2 runs by default:
//   * Linear forward launches that store BOTH the pre-activation and its Swish (two outputs, ~20 vector instructions per
//     element) over a short reduction (K <= 640) and >= 1536 tiles: one 64 x 64 tile per block, hardware dispatch, 47
//     registers and 24 KiB of LDS per block -- SIX blocks per CU instead of four cover one another's epilogues:
//     FashionMNIST's 2048 x 6272 x 512: 81 -> 93 TFLOP/s, CelebA-19's 18-group 768 x 512 x 512: 78 -> 89, its
//     4608 x 6400 x 100: 49 -> 58.
//   * everything else measured within +-8 % of the round-5 kernels either way (the persistent balanced schedule wins 5-8 % on
//     the long-reduction weight gradients and loses as much on the short ones: the slabs of the cut tiles and the blocks
//     running in lock-step through their epilogues cost what the balance gains) and stays where it was; the persistent
//     modes remain reachable through MVAE_G2_FORCE in the tuning build.
enum G2Hint { G2_PLAIN = 0, G2_FWD_TWO_OUTPUTS = 1, G2_CONV_FWD = 2, G2_FWD_ACT_ONLY = 3 };
inline G2Plan g2_plan_for(int I, int J, int K, int ncls, bool rowsum, void *ws, size_t ws_bytes, G2Hint hint = G2_PLAIN) {
    G2Plan none; none.ok = false;
    if (!MVAE_G2) return none;
    int fwm = 0, fwn = 0, focc = 0;
#ifdef MVAE_TUNING
    if (getenv("MVAE_G2_OFF")) return none;
    if (const char *f = getenv("MVAE_G2_FORCE")) (void)sscanf(f, "%d,%d,%d", &fwm, &fwn, &focc);
#endif
    if (!fwm) {
        const long tiles64 = cdiv(I, 64) * cdiv(J, 64) * ncls;
        // ... and (MVAE_G2_ACT_ONLY) the forwards of a statistics-only pass, which keep the Swish output alone, over a very short
        // reduction: celeba19's 4608 x 6400 x 100 (7200 tiles of four k-steps: prologue and epilogue are the launch)
        const bool two = hint == G2_FWD_TWO_OUTPUTS && K <= 640 && tiles64 >= 1536;
        const bool one = MVAE_G2_ACT_ONLY && hint == G2_FWD_ACT_ONLY && K <= 128 && tiles64 >= 1536;
        if (!two && !one) return none;
        fwm = 1; fwn = 1; focc = -1;
    }
    G2Plan pl = g2_plan(I, J, K, ncls, rowsum, fwm, fwn, focc);
    if (!pl.ok) return none;
    if (pl.ws_floats && (!ws || ws_bytes < pl.ws_floats * sizeof(float) || !aligned16(ws))) return none;
    pl.a.ws = (float *)ws;
    return pl;
}
// upper bound of the slab scratch over the plans a (rows, columns, reduction) shape can get
// (the default plans cut no tile: only the tuning build's forced persistent modes need slabs)
inline size_t g2_ws_floats_max(int I, int J, int K) {
    size_t n = 0;
#ifdef MVAE_TUNING
    for (int wm = 1; wm <= 2; ++wm)
        for (int wn = 1; wn <= 2; ++wn) {
            G2Plan pl = g2_plan(I, J, K, 1, true, wm, wn, 0);
            if (pl.ok && pl.ws_floats > n) n = pl.ws_floats;
        }
#else
    (void)I; (void)J; (void)K;
#endif
    return n;
}

template <template <int> class PL, template <int> class QL, class E, bool ROWSUM, class PF, class QF>
int launch_gemm2(const G2Plan &pl, PF make_p, QF make_q, E e, hipStream_t st) {
#define MVAE_G2_LAUNCH(WM, WN, STG, MINW)                                                            \
    {                                                                                                \
        constexpr int BM = 64 * WM, BN = 64 * WN;                                                    \
        PL<BM> p; make_p(p);                                                                         \
        QL<BN> q; make_q(q);                                                                         \
        constexpr size_t lds_min = ((size_t)STG * (BM + BN) * G2_BK + (ROWSUM ? 256 : 0)) * sizeof(float); \
        /* exactly `occ` blocks per CU: the request is padded to a 1 / occ share of the 160 KiB */     \
        const size_t lds_share = pl.occ ? (size_t)(160 * 1024 / pl.occ) / 1024 * 1024 : 0;            \
        const size_t lds = lds_share > lds_min ? lds_share : lds_min;                                \
        auto kern = gemm2_kernel<PL<BM>, QL<BN>, E, WM, WN, ROWSUM, STG, MINW>;                      \
        static bool attr_done = false;                                                               \
        if (!attr_done) {                                                                            \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
            attr_done = true;                                                                        \
        }                                                                                            \
        hipLaunchKernelGGL(kern, dim3(pl.a.G), dim3(256), lds, st, p, q, e, pl.a);                   \
        if (pl.a.R > 0 && pl.a.upb < pl.a.nk) {                                                      \
            constexpr int ROWS = 256 / (BN / 4);                                                     \
            hipLaunchKernelGGL((g2_finish_kernel<E, BM, BN, ROWSUM>), dim3(pl.a.R, BM / ROWS), dim3(256), 0, st, e, pl.a); \
        }                                                                                            \
    }
    if (pl.wm == 2 && pl.wn == 2) MVAE_G2_LAUNCH(2, 2, 3, 2)
    else if (pl.wm == 2 && pl.wn == 1) MVAE_G2_LAUNCH(2, 1, 3, 3)
    else if (pl.wm == 1 && pl.wn == 2) MVAE_G2_LAUNCH(1, 2, 3, 3)
    else MVAE_G2_LAUNCH(1, 1, 3, 4)
#undef MVAE_G2_LAUNCH
    return mvae_launch_status();
}


// ==========================================================================================
// The latency layouts ("gemm2s"): the 512-wide MLP layers at batch 512-1024 -- 3.4 us of matrix time inside a 12-14 us
// launch (profiles/r05_mnist_by_shape.txt), a step that is a chain of ~26 of them.  Same k-grouped block shape as
// gemm_core.h's small layouts (tile 32 x 64 / 64 x 32 / 32 x 32, KW k-groups of waves that split every k-step, one block
// per CU or two, the k-groups' tiles summed through LDS in a cooperative epilogue), on the LDS-DMA machinery above:
//   * the operands of a k-step arrive by `buffer_load_dwordx4 ... lds` issued by the block's first four waves: nothing is
//     staged through registers, so the ring is FOUR steps deep at 116-122 -> ~60 registers (round 5 measured four tiles in
//     flight in REGISTERS: the launch 13.2 -> 12.4 us, the step +9 % -- two chain kernels no longer shared a CU);
//   * the fragments of step k + 1 are read from LDS BEFORE the matrix instructions of step k issue: with two waves per SIMD
//     that leave every barrier together, nothing else covers the LDS latency (~250 of a step's ~1300 cycles);
//   * tools/gemm2s_probe.hip, hot inside a hipGraph: 1024 x 512 x 512 with bias + Swish, two outputs: 9.2-9.7 us against
//     10.4 for the round-5 kernel.
// K % 4 == 0 (whole float4s; a partial last k-step reads as zero), ragged rows / columns through out-of-range offsets.
template <class E, int TMW, int TNW, int KW, int CH, int S, bool Q_RK>
__global__ __launch_bounds__(64 * TMW * TNW * KW)
void gemm2s_kernel(const float *__restrict__ P, int ldp, size_t p_cs, const float *__restrict__ Q, int ldq, size_t q_cs, E e,
                   int I, int J, int K) {
    constexpr int NIW = 4;                                         // issuing waves
    constexpr int BM = 32 * TMW, BN = 32 * TNW, BK = 8 * KW * CH, NT = 64 * TMW * TNW * KW;
    constexpr int F = BK / 4;
    constexpr int P_FLOATS = BM * BK, Q_FLOATS = BN * BK, STAGE_FLOATS = P_FLOATS + Q_FLOATS;
    constexpr int NA = P_FLOATS / 256, NB = Q_FLOATS / 256;
    static_assert(NA % NIW == 0 && NB % NIW == 0, "pieces must divide over the issuing waves");
    constexpr int NPW = (NA + NB) / NIW;
    static_assert(NPW * (S - 2) <= 63 && S >= 3, "ring depth");
    static_assert(F == 4 || F == 8 || F == 16, "k-step depth");
    static_assert((BM * BN) % NT == 0, "cooperative epilogue: every thread takes part in every round");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = g2_uni(t >> 6);
    const int kg = wave / (TMW * TNW), wq = wave % (TMW * TNW), wi = wq / TNW, wj = wq % TNW;
    const int lrow = lane >> 5, lcol = lane & 31;
    const int cls = blockIdx.y;
    const int tiles_j = (J + BN - 1) / BN, tiles_i = (I + BM - 1) / BM;
    int b = blockIdx.x;
    if (tiles_i % 8 == 0) b = (b & 7) * (tiles_i * tiles_j / 8) + (b >> 3);      // XCD x owns tiles_i / 8 row bands x all column tiles
    const int ti = b / tiles_j, tj = b - ti * tiles_j;
    const int i0 = ti * BM, j0 = tj * BN;
    const unsigned lds0 = (unsigned)(unsigned long)(g2_lds_void *)lds;
    auto swz = [](int r) { return F == 4 ? (r >> 2) & 3 : F == 8 ? (r >> 1) & 7 : r & 15; };
    E et = e;
    et.set_class(cls);
    // what this thread's outputs will need (bias, the producer's pre-activation, the dropout mask): requested first
    constexpr int CNE = (BM * BN) / NT;
    typename E::Pre cpre[CNE];
#pragma unroll
    for (int k = 0; k < CNE; ++k) {
        const int el = t + k * NT;
        cpre[k] = et.fetch(i0 + el / BN, j0 + el % BN);
    }
    // ---- DMA pieces of the issuing waves: piece q = wave + NIW * u (u < NPW); q < NA: a P piece, else Q piece q - NA
    int voff[NPW], kpos[NPW];
    if (wave < NIW) {
#pragma unroll
        for (int u = 0; u < NPW; ++u) {
            const int q = wave + NIW * u;
            if (NIW * u < NA) {
                const int slot = q * 64 + lane, r = slot / F, f = (slot % F) ^ swz(r);
                voff[u] = (i0 + r < I) ? (r * ldp + f * 4) * 4 : BUF_OOB;
                kpos[u] = f * 4;
            } else if (Q_RK) {
                const int slot = (q - NA) * 64 + lane, r = slot / F, f = (slot % F) ^ swz(r);
                voff[u] = (j0 + r < J) ? (r * ldq + f * 4) * 4 : BUF_OOB;
                kpos[u] = f * 4;
            } else {
                const int slot = (q - NA) * 64 + lane, k = slot / (BN / 4), n = (slot % (BN / 4)) * 4;
                voff[u] = (j0 + n < J) ? (k * ldq + n) * 4 : BUF_OOB;
                kpos[u] = 0;
            }
        }
    }
    const BufBase pb = buf_base(P + (size_t)cls * p_cs + (size_t)i0 * ldp);
    const BufBase qb0 = buf_base(Q_RK ? Q + (size_t)cls * q_cs + (size_t)j0 * ldq : Q + (size_t)cls * q_cs + j0);
    auto issue = [&](int kt, int stage) {
        if (wave >= NIW) return;
        const int k0 = kt * BK;
        const bool tail = k0 + BK > K;                             // block-uniform: the last, partial k-step
        const i32x4_t rp = g2_rsrc(pb, k0, 0x7fffffff);
        const long left = (long)(K - k0) * ldq * 4;
        const i32x4_t rq = Q_RK ? g2_rsrc(qb0, k0, 0x7fffffff)
                                : g2_rsrc(qb0, (long)k0 * ldq, (int)(left < 0x7fffffffl ? left : 0x7fffffffl));
        asm volatile("s_nop 4" ::: "memory");
#pragma unroll
        for (int u = 0; u < NPW; ++u) {
            const int q = wave + NIW * u;
            const bool isp = NIW * u < NA;
            const bool rk = isp || Q_RK;
            const int vo = (rk && tail && k0 + kpos[u] >= K) ? BUF_OOB : voff[u];
            g2_dma16(isp ? rp : rq, vo, g2_uni(lds0 + (stage * STAGE_FLOATS + (isp ? q * 256 : P_FLOATS + (q - NA) * 256)) * 4));
        }
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int pbase = (wi * 32 + lcol) * BK;
    const int qbase = Q_RK ? (wj * 32 + lcol) * BK : 4 * lrow * BN + wj * 32 + lcol;
    const int fsw = swz(lcol);
    const int nk = (K + BK - 1) / BK;
    // fragments of one k-step: CH chunks of 8 k's (this k-group's), 4 k-pairs each
    auto frags = [&](int stage, float4 (&pa)[CH], float4 (&qv)[CH]) {
        const float *Ps = lds + stage * STAGE_FLOATS;
        const float *Qs = Ps + P_FLOATS;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int ch = kg * CH + c;
            pa[c] = *reinterpret_cast<const float4 *>(Ps + pbase + 4 * ((2 * ch + lrow) ^ fsw));
            if (Q_RK) qv[c] = *reinterpret_cast<const float4 *>(Qs + qbase + 4 * ((2 * ch + lrow) ^ fsw));
            else qv[c] = make_float4(Qs[qbase + (8 * ch + 0) * BN], Qs[qbase + (8 * ch + 1) * BN], Qs[qbase + (8 * ch + 2) * BN],
                                     Qs[qbase + (8 * ch + 3) * BN]);
        }
    };
#pragma unroll
    for (int s2 = 0; s2 < S - 1; ++s2)
        if (s2 < nk) issue(s2, s2);
    // step 0 has landed when at most the younger S - 2 steps are outstanding
    if (nk >= S - 1) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(NPW * (S - 2)) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    float4 pa[CH], qv[CH], pa_n[CH], qv_n[CH];
    frags(0, pa, qv);
    int st_n = 1, st_i = S - 1;                                     // stage of step kt + 1, stage the next issue fills
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if (more) {
            // step kt + 1 must have landed: issued so far min(nk, kt + S - 1) steps, the ones after kt + 1 may stay in flight
            if (kt + S - 1 <= nk) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(NPW * (S - 3 > 0 ? S - 3 : 0)) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            // behind the barrier every wave holds step kt's fragments in registers: stage kt % S is free
            if (kt + S - 1 < nk) { issue(kt + S - 1, st_i); }
            st_i = st_i + 1 == S ? 0 : st_i + 1;
            frags(st_n, pa_n, qv_n);
            st_n = st_n + 1 == S ? 0 : st_n + 1;
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[c].x, qv[c].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[c].y, qv[c].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[c].z, qv[c].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[c].w, qv[c].w, acc, 0, 0, 0);
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) { pa[c] = pa_n[c]; qv[c] = qv_n[c]; }
    }
    // ---- cooperative epilogue: park the k-groups' tiles, sum them in group order, run the epilogue functor
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    constexpr int TP = BN + 1;
    float *tile = lds + kg * (BM * TP);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int il = wi * 32 + 4 * lrow + (r & 3) + 8 * (r >> 2);
        tile[il * TP + wj * 32 + lcol] = acc[r];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < CNE; ++k) {
        const int el = t + k * NT;
        const int il = el / BN, jl = el % BN;
        float v = 0.f;
#pragma unroll
        for (int g2 = 0; g2 < KW; ++g2) v += lds[g2 * (BM * TP) + il * TP + jl];
        const int i = i0 + il, j = j0 + jl;
        if (et.col(j)) et.put_pre(i, j, v, cpre[k]);
    }
}

#ifndef MVAE_G2S
#define MVAE_G2S 1                // 0: the k-grouped small layouts of gemm_core.h everywhere (A/B builds)
#endif
#ifndef MVAE_G2S_STAGES
#define MVAE_G2S_STAGES 4
#endif

// Launch for a plan of gemm_core.h's small layouts (pl.bk == 64): false when the shape has no instantiation here.
template <class E, bool Q_RK>
bool launch_gemm2s(const Plan &pl, const float *P, int ldp, size_t p_cs, const float *Q, int ldq, size_t q_cs, E e, int I,
                   int J, int K, int ncls, hipStream_t st, int *status) {
    if (!MVAE_G2S || pl.bk != 64 || pl.splits != 1 || (K & 3)) return false;
#ifdef MVAE_TUNING
    if (getenv("MVAE_G2S_OFF")) return false;
#endif
#define MVAE_G2S_LAUNCH(TMW, TNW, KW, CH)                                                                         \
    {                                                                                                             \
        constexpr int BM = 32 * TMW, BN = 32 * TNW, BK = 8 * KW * CH, NT = 64 * TMW * TNW * KW, STG = MVAE_G2S_STAGES; \
        constexpr size_t ring = (size_t)STG * (BM + BN) * BK * sizeof(float);                                     \
        constexpr size_t red = (size_t)KW * BM * (BN + 1) * sizeof(float);                                        \
        constexpr size_t ldsb = ring > red ? ring : red;                                                          \
        auto kern = gemm2s_kernel<E, TMW, TNW, KW, CH, STG, Q_RK>;                                                \
        static bool attr_done = false;                                                                            \
        if (!attr_done) {                                                                                         \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb); \
            attr_done = true;                                                                                     \
        }                                                                                                         \
        const dim3 grid((unsigned)(cdiv(I, BM) * cdiv(J, BN)), (unsigned)ncls);                                   \
        hipLaunchKernelGGL(kern, grid, dim3(NT), ldsb, st, P, ldp, p_cs, Q, ldq, q_cs, e, I, J, K);               \
        *status = mvae_launch_status();                                                                           \
        return true;                                                                                              \
    }
    if (pl.wgm == 1 && pl.wgn == 2 && pl.kw == 4) MVAE_G2S_LAUNCH(1, 2, 4, 1)
    if (pl.wgm == 2 && pl.wgn == 1 && pl.kw == 4) MVAE_G2S_LAUNCH(2, 1, 4, 1)
    if (pl.wgm == 1 && pl.wgn == 2 && pl.kw == 2) MVAE_G2S_LAUNCH(1, 2, 2, 2)
    if (pl.wgm == 2 && pl.wgn == 1 && pl.kw == 2) MVAE_G2S_LAUNCH(2, 1, 2, 2)
    if (pl.wgm == 1 && pl.wgn == 1 && pl.kw == 8) MVAE_G2S_LAUNCH(1, 1, 8, 1)
    if (pl.wgm == 1 && pl.wgn == 1 && pl.kw == 4) MVAE_G2S_LAUNCH(1, 1, 4, 2)
#undef MVAE_G2S_LAUNCH
    return false;
}

}  // namespace
