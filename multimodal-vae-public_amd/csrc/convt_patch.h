// convt_patch.h -- stride-2 transposed conv (dgrad form) with ONE LDS input patch for the four output parity classes
// (round 6).  ConvTranspose2d(128,64) / (64,32) forward of the decoders and the data gradient of
// Conv2d(32,64) / (64,128): celeba/model.py:79-83,119-124, fashionmnist/model.py:82,112.
//
// The dgrad-form launch of gemm_core.h treats each parity class as its own GEMM: per class and k-step every thread
// GATHERS 8 dwords of dy into registers and stages them through LDS -- the same 3 x 3 neighbourhood of every lattice
// position four times over, 512 dword loads + 512 ds_write_b32 per thread and tile at K = 512 (the knock-out table:
// global loads 8-12 % of those launches, staging + barriers 5-8 %; 3.5 fabric read requests per 64 B of input for
// FashionMNIST's 7 x 7 -> 14 x 14 layer, profiles/r05_convT_l2_counters.txt).  Here a block owns NPOS consecutive
// lattice positions j = (n, ih', iw') and ALL four classes of them:
//   * the input it can touch -- whole images (7 x 7, 8 x 8 maps) or a band of rows (16 x 16) of a PHASE of KPH input
//     channels -- comes in ONCE, in its natural layout [channel][position], by LDS-DMA (`buffer_load_dword[x4] ... lds`:
//     no registers, no ds_write); the B fragment of class (ph, pw), tap (a, b), channel c for the lane's position is the
//     patch value at (ih' + ph - a, iw' + pw - b): ONE ds_read_b32 at a per-lane neighbour address (a neighbour outside
//     the image points at a zero slot kept per channel), no im2col image anywhere;
//   * the repacked weights wr[class][k = (c, a, b)][ci] stream through a 3-deep LDS ring by LDS-DMA, 16 k's per step, the
//     two pw classes of the current ph side by side: a wave holds acc[ph][pw] -- FOUR accumulators over the whole launch,
//     two in use per step; the three patch values of a (channel, a) feed four matrix instructions;
//   * the epilogue is gemm_core.h's pair store (EpNCHW::put2_b: the two pw classes of an output row as float2) or the
//     statistics-only record (EpStats), unchanged.
#pragma once
#include "gemm2.h"

namespace {

#ifndef MVAE_CONVT_PATCH
#define MVAE_CONVT_PATCH 1          // 0: every stride-2 dgrad-form launch stays on igemm_kernel (A/B builds)
#endif
#ifndef MVAE_PATCH_MINBLOCKS
#define MVAE_PATCH_MINBLOCKS 512
#endif

struct PatchGeo {
    int B, Cout, Cin;               // images, channels of dy (the reduction), channels of dx (rows of the GEMM)
    int H2, W2, OHW;                // the lattice = dy's map
    int J, K;                       // B * OHW columns per class, Cout * 4
    int mode_a;                     // 1: whole images in the patch, 0: a band of rows of ONE image
    int nimg;                       // mode a: images per patch
    int ps_raw;                     // floats per channel before the zero slot
};

constexpr int CP_BK = 16, CP_STAGES = 3;

// E: EpNCHWPair (put2_b) or EpStats.  CI: rows (= Cin, 64 or 32).  NPOS: lattice positions per block (64 with CI = 64:
// 2 x 2 waves; 128 with CI = 32: 1 x 4).  PS: floats per channel in the patch incl. the zero slot (PS - 1).  X4: 16-byte
// DMA pieces.  KPH: input channels per phase.  NUI: DMA instructions per thread and phase (ceil(KPH * PS[/4] / 256)).
template <class E, int CI, int NPOS, int PS, bool X4, int KPH, int NUI>
__global__ __launch_bounds__(256) void convT_patch_kernel(const float *__restrict__ dy, const float *__restrict__ wr, E e,
                                                          PatchGeo g) {
    static_assert((CI == 64 && NPOS == 64) || (CI == 32 && NPOS == 128), "wave layouts");
    static_assert(!X4 || PS % 4 == 0, "16-byte pieces");
    constexpr int PSV = X4 ? PS / 4 : PS;                  // DMA units per channel
    constexpr int PATCH_FLOATS = NUI * 256 * (X4 ? 4 : 1);  // what the DMA covers (>= KPH * PS)
    constexpr int WT = 2 * CP_BK * CI;                      // floats per ring stage: [pw][16][CI]
    constexpr int SPP = KPH / 4;                            // k-steps per (phase, ph)
    constexpr int NPW = CI == 64 ? 2 : 1;                   // weight DMA instructions per wave and step
    extern __shared__ __attribute__((aligned(16))) float lds[];     // [patch | ring of 3 weight stages | stats scratch]
    const int t = threadIdx.x, lane = t & 63;
    const int wave = g2_uni(t >> 6);
    const int wi = CI == 64 ? wave >> 1 : 0, wj = CI == 64 ? wave & 1 : wave;
    const int lrow = lane >> 5, lcol = lane & 31;
    const int j0 = blockIdx.x * NPOS;
    const unsigned lds0 = (unsigned)(unsigned long)(g2_lds_void *)lds;
    const unsigned ring0 = lds0 + PATCH_FLOATS * 4;

    // ---- the lane's lattice position and its 3 x 3 neighbour places in a channel's patch slab
    const int j = j0 + wj * 32 + lcol;
    const bool jok = j < g.J;
    const int jj = jok ? j : 0;
    const int n = jj / g.OHW, rem = jj - n * g.OHW;
    const int ih2 = rem / g.W2, iw2 = rem - ih2 * g.W2;
    const int n0 = g2_uni(j0 / g.OHW);
    const int r0 = g2_uni((j0 - n0 * g.OHW) / g.W2);       // mode b: first lattice row of the block
    const int pidx = g.mode_a ? (n - n0) * g.OHW + rem : (ih2 - r0 + 1) * g.W2 + iw2;
    int nb[3][3];
#pragma unroll
    for (int dr = -1; dr <= 1; ++dr)
#pragma unroll
        for (int dc = -1; dc <= 1; ++dc) {
            const bool ok = jok && ih2 + dr >= 0 && ih2 + dr < g.H2 && iw2 + dc >= 0 && iw2 + dc < g.W2;
            nb[dr + 1][dc + 1] = ((ok ? pidx + dr * g.W2 + dc : PS - 1) + lrow * PS) * 4;      // bytes; the upper half wave takes the next channel
        }
    // ---- patch DMA: unit u = i * 256 + t -> (channel u / PSV, piece u % PSV) -> bytes from (image n0, first channel of the phase)
    int pvoff[NUI];
#pragma unroll
    for (int i = 0; i < NUI; ++i) {
        const int u = i * 256 + t, c = u / PSV, q = (u - c * PSV) * (X4 ? 4 : 1);
        int off = BUF_OOB;
        if (c < KPH && q < g.ps_raw) {
            if (g.mode_a) {
                const int img = q / g.OHW, pos = q - img * g.OHW;
                if (n0 + img < g.B) off = ((img * g.Cout + c) * g.OHW + pos) * 4;
            } else {
                const int row = q / g.W2, ih = r0 - 1 + row;
                if (ih >= 0 && ih < g.H2) off = (c * g.OHW + ih * g.W2 + (q - row * g.W2)) * 4;
            }
        }
        pvoff[i] = off;
    }
    const BufBase dyb = buf_base(dy + (size_t)n0 * g.Cout * g.OHW);
    auto issue_patch = [&](int phase) {
        const i32x4_t rs = g2_rsrc(dyb, (long)phase * KPH * g.OHW, 0x7fffffff);
        asm volatile("s_nop 4" ::: "memory");
#pragma unroll
        for (int i = 0; i < NUI; ++i) {
            const unsigned dst = g2_uni(lds0 + (i * 256 + wave * 64) * (X4 ? 16 : 4));
            if (X4) g2_dma16(rs, pvoff[i], dst);
            else g2_dma4(rs, pvoff[i], 0, dst);
        }
    };
    // ---- weight DMA: stage layout [pw][16][CI], 16 bytes per thread and class half
    const BufBase wb = buf_base(wr);
    const int tq = CI == 64 ? t : (t & 127);
    const int wvoff = ((tq / (CI / 4)) * g.Cin + (tq % (CI / 4)) * 4) * 4;
    const int steps_total = (g.Cout / KPH) * 2 * SPP;
    auto issue_w = [&](int u) {                                // flat step u = (phase, ph, ks)
        if (u >= steps_total) return;
        const int phase = u / (2 * SPP), r2 = u - phase * 2 * SPP, ph = r2 / SPP, ks = r2 - ph * SPP;
        const long k0 = (long)phase * KPH * 4 + ks * CP_BK;
        const unsigned dst = ring0 + (u % CP_STAGES) * WT * 4;
        asm volatile("s_nop 4" ::: "memory");
        if (CI == 64) {
#pragma unroll
            for (int pw = 0; pw < 2; ++pw) {
                const i32x4_t rs = g2_rsrc(wb, ((long)(ph * 2 + pw) * g.K + k0) * g.Cin, 0x7fffffff);
                g2_dma16(rs, wvoff, g2_uni(dst + (pw * CP_BK * CI + wave * 256) * 4));
            }
        } else {
            const int pw = wave >> 1;
            const i32x4_t rs = g2_rsrc(wb, ((long)(ph * 2 + pw) * g.K + k0) * g.Cin, 0x7fffffff);
            g2_dma16(rs, wvoff, g2_uni(dst + wave * 256 * 4));
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int abase = (4 * lrow * CI + wi * 32 + lcol) * 4;    // bytes inside a stage: row 4 * lrow + tap, column = the lane's GEMM row
    issue_w(0);
    issue_w(1);
    int u = 0;
    const int nphase = g.Cout / KPH;
    for (int phase = 0; phase < nphase; ++phase) {
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
            // neighbour rows of this ph: tap a = 0 -> row ih' + ph, a = 1 -> row ih' + ph - 1
            int pb[2][3];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int dc = 0; dc < 3; ++dc) pb[a][dc] = nb[ph - a + 1][dc];
            for (int ks = 0; ks < SPP; ++ks, ++u) {
                if (ph == 0 && ks == 0) {
                    // a new phase: everyone is done with the old patch and every weight piece has landed
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                    issue_patch(phase);
                    issue_w(u + 2);
                    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
                } else {
                    // step u's weights have landed when only the pieces of step u + 1 are still in flight
                    if (u + 1 < steps_total) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(NPW) : "memory");
                    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                    issue_w(u + 2);
                }
                const char *Ws = reinterpret_cast<const char *>(lds) + (PATCH_FLOATS + (u % CP_STAGES) * WT) * 4 + abase;
                const char *Pc = reinterpret_cast<const char *>(lds) + (size_t)ks * 4 * PS * 4;       // the step's first channel
#pragma unroll
                for (int c = 0; c < 2; ++c) {
#pragma unroll
                    for (int a = 0; a < 2; ++a) {
                        const float bm1 = *reinterpret_cast<const float *>(Pc + pb[a][0] + 2 * c * PS * 4);
                        const float b0 = *reinterpret_cast<const float *>(Pc + pb[a][1] + 2 * c * PS * 4);
                        const float bp1 = *reinterpret_cast<const float *>(Pc + pb[a][2] + 2 * c * PS * 4);
                        const int kr = 8 * c + 2 * a;                                               // + 4 * lrow inside abase
                        const float a00 = *reinterpret_cast<const float *>(Ws + (kr * CI) * 4);
                        const float a01 = *reinterpret_cast<const float *>(Ws + ((kr + 1) * CI) * 4);
                        const float a10 = *reinterpret_cast<const float *>(Ws + (CP_BK * CI + kr * CI) * 4);
                        const float a11 = *reinterpret_cast<const float *>(Ws + (CP_BK * CI + (kr + 1) * CI) * 4);
                        acc[ph][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a00, b0, acc[ph][0], 0, 0, 0);
                        acc[ph][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a10, bp1, acc[ph][1], 0, 0, 0);
                        acc[ph][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a01, bm1, acc[ph][0], 0, 0, 0);
                        acc[ph][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a11, b0, acc[ph][1], 0, 0, 0);
                    }
                }
            }
        }
    }

    // ---- epilogue
    if constexpr (ep_stats<E>::value) {
        // (mean, M2) of every row over the block's 4 * NPOS values, around a shift (gemm_core.h STATK): one record per block
        static_assert(CI == 32, "the statistics record is the 32-row layout's");
        f32x16 st1, st2, shf;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int bits = __float_as_int(acc[0][0][r]);
            const float lo = __int_as_float(__builtin_amdgcn_readlane(bits, 0));
            const float hi = __int_as_float(__builtin_amdgcn_readlane(bits, 32));
            shf[r] = lrow ? hi : lo;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const float d = acc[a][b][r] - shf[r];
                    s1 += d;
                    s2 = fmaf(d, d, s2);
                }
            st1[r] = half_wave_sum(s1);
            st2[r] = half_wave_sum(s2);
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        float *red = lds;                                   // [wave][32 rows][3]
        if (lcol == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lrow;
                red[(wave * 32 + row) * 3 + 0] = st1[r];
                red[(wave * 32 + row) * 3 + 1] = st2[r];
                red[(wave * 32 + row) * 3 + 2] = shf[r];
            }
        }
        __syncthreads();
        if (t < 32) {
            const float nw = 128.f;                         // values per (wave, row): 32 positions x 4 classes
            float mw[4], m2w[4], mean = 0.f;
#pragma unroll
            for (int w2 = 0; w2 < 4; ++w2) {
                const float *rr = red + (w2 * 32 + t) * 3;
                const float d = rr[0] / nw;
                mw[w2] = rr[2] + d;
                m2w[w2] = fmaxf(rr[1] - rr[0] * d, 0.f);
                mean += mw[w2];
            }
            mean /= 4.f;
            float m2 = 0.f;
#pragma unroll
            for (int w2 = 0; w2 < 4; ++w2) m2 += m2w[w2] + nw * (mw[w2] - mean) * (mw[w2] - mean);
            if (t < e.C) {
                float *dst = e.part + ((size_t)blockIdx.x * e.C + t) * 2;
                dst[0] = mean;
                dst[1] = m2;
            }
        }
    } else {
        E et = e;
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
            et.set_class(ph * 2);
            et.tile(j0);
            (void)et.col(j);
            const int rb = __builtin_amdgcn_readfirstlane(wi * 32);
#pragma unroll
            for (int r = 0; r < 16; ++r) et.put2_b(rb, r, acc[ph][0][r], acc[ph][1][r]);
        }
    }
}

// ---- version 2 of the storing form: 32 rows x 64 positions per block, wave = (position half, ph) with the two pw
// accumulators of ITS ph -- the same 16 matrix instructions per wave and step, half the accumulators (112 registers: four
// waves per SIMD), twice the blocks (the 256-image launches: 512 instead of 256), and the patch DOUBLE-BUFFERED in
// phases of 16 input channels: the next phase's pieces are issued behind the weights of the phase's first step and have
// three steps to land -- no drain, no exposed patch latency except the first (version 1 above drained the memory queue at
// every phase start: 7 x 7 maps 4 % SLOWER than the gather launch, 8 x 8 / 16 x 16 maps 4-9 % faster,
// profiles/r06_patch_bench.txt).  Rows: i0 = 32 * blockIdx.y (Cin = 64: two row halves re-read the patch).
constexpr int CP2_KPH = 16, CP2_SPP = CP2_KPH / 4;

// E = EpNCHWPair: one tile per block, pair stores.  E = EpStats: TILES = 2 consecutive tiles per block and one (mean, M2)
// record over their 4 x 128 values per row -- the record mvae_bn_stats_merge expects (gemm_core.h STATK).
template <class E, int TILES, int PS, bool X4, int NUI>
__global__ __launch_bounds__(256, 2) void convT_patch2_kernel(const float *__restrict__ dy, const float *__restrict__ wr,
                                                              E e, PatchGeo g) {
    static_assert(!X4 || PS % 4 == 0, "16-byte pieces");
    constexpr bool STATS = ep_stats<E>::value;
    constexpr int PSV = X4 ? PS / 4 : PS;
    constexpr int PATCH_FLOATS = NUI * 256 * (X4 ? 4 : 1);  // one buffer
    constexpr int WT = 4 * CP_BK * 32;                      // floats per ring stage: [ph][pw][16][32]
    constexpr int NPW = 2;
    static_assert(NPW + NUI <= 63, "vmcnt");
    extern __shared__ __attribute__((aligned(16))) float lds[];     // [2 patch buffers | ring of 3 weight stages]
    const int t = threadIdx.x, lane = t & 63;
    const int wave = g2_uni(t >> 6);
    const int wj = wave & 1, ph = wave >> 1;
    const int lrow = lane >> 5, lcol = lane & 31;
    const int i0 = blockIdx.y * 32;
    const unsigned lds0 = (unsigned)(unsigned long)(g2_lds_void *)lds;
    const unsigned ring0 = lds0 + 2 * PATCH_FLOATS * 4;
    // weights: wave w moves the 16 x 32 tile of class w (= ph * 2 + pw) in two 1-KiB pieces (rows 0-7, 8-15)
    const BufBase wb = buf_base(wr + i0);
    const int wvoff = ((lane >> 3) * g.Cin + (lane & 7) * 4) * 4;
    const int steps_total = (g.Cout / CP2_KPH) * CP2_SPP;
    const int nphase = g.Cout / CP2_KPH;
    auto issue_w = [&](int u) {
        const unsigned dst = ring0 + ((u % CP_STAGES) * WT + wave * 512) * 4;
        const i32x4_t rs = g2_rsrc(wb, ((long)wave * g.K + (long)u * CP_BK) * g.Cin, 0x7fffffff);
        const i32x4_t rs2 = g2_rsrc(wb, ((long)wave * g.K + (long)u * CP_BK + 8) * g.Cin, 0x7fffffff);
        asm volatile("s_nop 4" ::: "memory");
        g2_dma16(rs, wvoff, g2_uni(dst));
        g2_dma16(rs2, wvoff, g2_uni(dst + 1024));
    };
    const int abase = (ph * 2 * CP_BK * 32 + 4 * lrow * 32 + lcol) * 4;
    std::conditional_t<STATS, f32x16, char> st1, st2, shf;

    for (int tile = 0; tile < TILES; ++tile) {
        const int j0 = (blockIdx.x * TILES + tile) * 64;
        const int j = j0 + wj * 32 + lcol;
        const bool jok = j < g.J;
        const int jj = jok ? j : 0;
        const int n = jj / g.OHW, rem = jj - n * g.OHW;
        const int ih2 = rem / g.W2, iw2 = rem - ih2 * g.W2;
        const int n0 = g2_uni(j0 / g.OHW);
        const int n1 = g2_uni((min(j0 + 63, g.J - 1)) / g.OHW);      // last image of the tile
        const int r0 = g2_uni((j0 - n0 * g.OHW) / g.W2);
        const int pidx = g.mode_a ? (n - n0) * g.OHW + rem : (ih2 - r0 + 1) * g.W2 + iw2;
        // the wave's ph picks two of the three neighbour rows: tap a = 0 -> row ih' + ph, a = 1 -> row ih' + ph - 1
        int pb[2][3];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int dc = -1; dc <= 1; ++dc) {
                const int dr = ph - a;
                const bool ok = jok && ih2 + dr >= 0 && ih2 + dr < g.H2 && iw2 + dc >= 0 && iw2 + dc < g.W2;
                pb[a][dc + 1] = ((ok ? pidx + dr * g.W2 + dc : PS - 1) + lrow * PS) * 4;
            }
        int pvoff[NUI];
#pragma unroll
        for (int i = 0; i < NUI; ++i) {
            const int u = i * 256 + t, c = u / PSV, q = (u - c * PSV) * (X4 ? 4 : 1);
            int off = BUF_OOB;
            if (c < CP2_KPH && q < g.ps_raw) {
                if (g.mode_a) {
                    const int img = q / g.OHW, pos = q - img * g.OHW;
                    if (n0 + img <= n1) off = ((img * g.Cout + c) * g.OHW + pos) * 4;      // only the images the tile touches
                } else {
                    const int row = q / g.W2, ih = r0 - 1 + row;
                    if (ih >= 0 && ih < g.H2) off = (c * g.OHW + ih * g.W2 + (q - row * g.W2)) * 4;
                }
            }
            pvoff[i] = off;
        }
        const BufBase dyb = buf_base(dy + (size_t)n0 * g.Cout * g.OHW);
        auto issue_patch = [&](int phase) {
            const i32x4_t rs = g2_rsrc(dyb, (long)phase * CP2_KPH * g.OHW, 0x7fffffff);
            const unsigned base = lds0 + (phase & 1) * PATCH_FLOATS * 4;
            asm volatile("s_nop 4" ::: "memory");
#pragma unroll
            for (int i = 0; i < NUI; ++i) {
                const unsigned dst = g2_uni(base + (i * 256 + wave * 64) * (X4 ? 16 : 4));
                if (X4) g2_dma16(rs, pvoff[i], dst);
                else g2_dma4(rs, pvoff[i], 0, dst);
            }
        };

        f32x16 acc[2];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;

        if (tile > 0) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // the previous tile's readers are done
        issue_w(0);
        if (steps_total > 1) issue_w(1);
        issue_patch(0);
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        int u = 0;
        for (int phase = 0; phase < nphase; ++phase) {
            const bool more = phase + 1 < nphase;               // block-uniform
            const char *Pp = reinterpret_cast<const char *>(lds) + (phase & 1) * PATCH_FLOATS * 4;
#pragma unroll
            for (int ks = 0; ks < CP2_SPP; ++ks, ++u) {
                if (u > 0) {
                    // what may stay in flight behind step u's weights: step u + 1's, and (steps 1, 2 of a phase) the next patch
                    if (u + 1 >= steps_total) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                    else if ((ks == 1 || ks == 2) && more) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(NPW + NUI) : "memory");
                    else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(NPW) : "memory");
                }
                if (u + 2 < steps_total) issue_w(u + 2);
                if (ks == 0 && more) issue_patch(phase + 1);
                const char *Ws = reinterpret_cast<const char *>(lds) + (2 * PATCH_FLOATS + (u % CP_STAGES) * WT) * 4 + abase;
                const char *Pc = Pp + ks * 4 * PS * 4;
#pragma unroll
                for (int c = 0; c < 2; ++c) {
#pragma unroll
                    for (int a = 0; a < 2; ++a) {
                        const float bm1 = *reinterpret_cast<const float *>(Pc + pb[a][0] + 2 * c * PS * 4);
                        const float b0 = *reinterpret_cast<const float *>(Pc + pb[a][1] + 2 * c * PS * 4);
                        const float bp1 = *reinterpret_cast<const float *>(Pc + pb[a][2] + 2 * c * PS * 4);
                        const int kr = 8 * c + 2 * a;
                        const float a00 = *reinterpret_cast<const float *>(Ws + (kr * 32) * 4);
                        const float a01 = *reinterpret_cast<const float *>(Ws + ((kr + 1) * 32) * 4);
                        const float a10 = *reinterpret_cast<const float *>(Ws + (CP_BK * 32 + kr * 32) * 4);
                        const float a11 = *reinterpret_cast<const float *>(Ws + (CP_BK * 32 + (kr + 1) * 32) * 4);
                        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a00, b0, acc[0], 0, 0, 0);
                        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a10, bp1, acc[1], 0, 0, 0);
                        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a01, bm1, acc[0], 0, 0, 0);
                        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a11, b0, acc[1], 0, 0, 0);
                    }
                }
            }
        }
        if constexpr (STATS) {
            // sums around a shift (gemm_core.h STATK): the first value lane 0 of each half wave holds of every row
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (tile == 0) {
                    const int bits = __float_as_int(acc[0][r]);
                    const float lo = __int_as_float(__builtin_amdgcn_readlane(bits, 0));
                    const float hi = __int_as_float(__builtin_amdgcn_readlane(bits, 32));
                    shf[r] = lrow ? hi : lo;
                    st1[r] = 0.f; st2[r] = 0.f;
                }
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const float d = acc[b][r] - shf[r];
                    st1[r] += d;
                    st2[r] = fmaf(d, d, st2[r]);
                }
            }
        } else {
            E et = e;
            et.set_class(ph * 2);
            et.tile(j0);
            (void)et.col(j);
            const int rb = __builtin_amdgcn_readfirstlane(i0);
#pragma unroll
            for (int r = 0; r < 16; ++r) et.put2_b(rb, r, acc[0][r], acc[1][r]);
        }
    }
    if constexpr (STATS) {
        // per (wave, row): 2 tiles x 2 pw x 32 positions = 128 values; the four waves merged in wave order
#pragma unroll
        for (int r = 0; r < 16; ++r) { st1[r] = half_wave_sum(st1[r]); st2[r] = half_wave_sum(st2[r]); }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        float *red = lds;                                   // [wave][32 rows][3]
        if (lcol == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lrow;
                red[(wave * 32 + row) * 3 + 0] = st1[r];
                red[(wave * 32 + row) * 3 + 1] = st2[r];
                red[(wave * 32 + row) * 3 + 2] = shf[r];
            }
        }
        __syncthreads();
        if (t < 32) {
            const float nw = (float)(TILES * 64);
            float mw[4], m2w[4], mean = 0.f;
#pragma unroll
            for (int w2 = 0; w2 < 4; ++w2) {
                const float *rr = red + (w2 * 32 + t) * 3;
                const float d = rr[0] / nw;
                mw[w2] = rr[2] + d;
                m2w[w2] = fmaxf(rr[1] - rr[0] * d, 0.f);
                mean += mw[w2];
            }
            mean /= 4.f;
            float m2 = 0.f;
#pragma unroll
            for (int w2 = 0; w2 < 4; ++w2) m2 += m2w[w2] + nw * (mw[w2] - mean) * (mw[w2] - mean);
            if (i0 + t < e.C) {
                float *dst = e.part + ((size_t)blockIdx.x * e.C + i0 + t) * 2;
                dst[0] = mean;
                dst[1] = m2;
            }
        }
    }
}

// which instantiation covers a geometry (0: none)
struct PatchPlan { int kind; PatchGeo g; int blocks; size_t lds; };
inline PatchPlan convt_patch_plan(int B, int Cout, int Cin, int OH, int OW, bool stats) {
    PatchPlan pl; pl.kind = 0;
    if (!MVAE_CONVT_PATCH) return pl;
#ifdef MVAE_TUNING
    if (getenv("MVAE_PATCH_OFF")) return pl;
#endif
    PatchGeo &g = pl.g;
    g.B = B; g.Cout = Cout; g.Cin = Cin; g.H2 = OH; g.W2 = OW; g.OHW = OH * OW;
    const long J = (long)B * g.OHW;
    if (J * 4 >= (1L << 31) || (long)B * Cout * g.OHW >= (1L << 29)) return pl;
    g.J = (int)J; g.K = Cout * 4;
    if (stats) {
        if (Cin == 32 && OH == 16 && OW == 16 && Cout % CP2_KPH == 0 && J % 128 == 0) {     // two 64-position tiles per record
            pl.kind = 4; g.mode_a = 0; g.nimg = 1; g.ps_raw = 96;
            pl.blocks = (int)(J / 64);
        }
        return pl;
    }
    if ((Cin != 64 && Cin != 32) || Cout % CP2_KPH != 0) return pl;
    if (OH == 8 && OW == 8) { pl.kind = 1; g.mode_a = 1; g.nimg = 1; g.ps_raw = 64; }              // whole image per block
    else if (OH == 7 && OW == 7) { pl.kind = 2; g.mode_a = 1; g.nimg = 3; g.ps_raw = 147; }        // 64 positions touch <= 3 images of 49
    else if (OH == 16 && OW == 16) { pl.kind = 3; g.mode_a = 0; g.nimg = 1; g.ps_raw = 96; }       // 4 lattice rows + a halo row either side
    pl.blocks = (int)cdiv(J, 64);
    // few blocks: the gather launch's smaller per-block footprint wins (256 images of 8 x 8: 53 vs 65 us with version 1)
    if (pl.kind && (long)pl.blocks * (Cin / 32) < MVAE_PATCH_MINBLOCKS) pl.kind = 0;
    return pl;
}

template <class E, int CI, int NPOS, int PS, bool X4, int KPH>
int launch_convt_patch(const PatchPlan &pl, const float *dy, const float *wr, E e, hipStream_t st) {
    constexpr int PSV = X4 ? PS / 4 : PS;
    constexpr int NUI = (KPH * PSV + 255) / 256;
    constexpr size_t lds = ((size_t)NUI * 256 * (X4 ? 4 : 1) + (size_t)CP_STAGES * 2 * CP_BK * CI) * sizeof(float);
    auto kern = convT_patch_kernel<E, CI, NPOS, PS, X4, KPH, NUI>;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(pl.blocks), dim3(256), lds, st, dy, wr, e, pl.g);
    return mvae_launch_status();
}

template <class E, int TILES, int PS, bool X4>
int launch_convt_patch2(const PatchPlan &pl, const float *dy, const float *wr, const E &e, hipStream_t st) {
    constexpr int PSV = X4 ? PS / 4 : PS;
    constexpr int NUI = (CP2_KPH * PSV + 255) / 256;
    constexpr size_t lds = ((size_t)2 * NUI * 256 * (X4 ? 4 : 1) + (size_t)CP_STAGES * 4 * CP_BK * 32) * sizeof(float);
    auto kern = convT_patch2_kernel<E, TILES, PS, X4, NUI>;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(pl.blocks / TILES, pl.g.Cin / 32), dim3(256), lds, st, dy, wr, e, pl.g);
    return mvae_launch_status();
}

#ifndef MVAE_PATCH_KPH64
#define MVAE_PATCH_KPH64 64         // input channels per phase of the 64-row kernels (8 x 8 maps)
#endif
#ifndef MVAE_PATCH_KPH49
#define MVAE_PATCH_KPH49 32         // ... 7 x 7 maps (dword pieces: 148 floats per channel)
#endif
#ifndef MVAE_PATCH_KPH32
#define MVAE_PATCH_KPH32 32         // ... of the 32-row kernels (16 x 16 maps, 164 floats per channel)
#endif

}  // namespace
