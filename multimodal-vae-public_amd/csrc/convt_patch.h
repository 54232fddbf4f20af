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
//   * the repacked weights wr[class][k = (c, a, b)][ci] stream through a 3-deep LDS ring by LDS-DMA, 16 k's per step, all
//     four classes side by side; a block is 32 rows x 64 positions, a wave = (position half, ph) holds the two pw
//     accumulators of its ph: the three patch values of a (channel, a) feed four matrix instructions;
//   * the patch is DOUBLE-BUFFERED in phases of 16 input channels: the next phase's pieces are issued behind the weights
//     of the phase's first step and have three steps to land (a first version that drained the memory queue at every
//     phase start ran the 7 x 7 maps 4 % slower than the gather launch: profiles/r06_patch_bench_v1.txt);
//   * the epilogue is gemm_core.h's pair store (EpNCHW::put2_b: the two pw classes of an output row as float2), the
//     statistics-only record (EpStats) -- or, for maps whose output rows are shorter than a cache line (7 x 7 -> 14 x 14:
//     56-byte rows; a wave's pair stores are 56-byte pieces with 56-byte gaps that the OTHER ph's wave fills: 58 % partial
//     write requests, profiles/r05_convT_l2_counters.txt), a pass through LDS: the block's 4 x 64 outputs per channel are
//     laid out as they lie in memory and leave as 512 contiguous bytes per store instruction.
#pragma once
#include "gemm2.h"

namespace {

#ifndef MVAE_CONVT_PATCH
#define MVAE_CONVT_PATCH 1          // 0: every stride-2 dgrad-form launch stays on igemm_kernel (A/B builds)
#endif
#ifndef MVAE_PATCH_STAGE7
#define MVAE_PATCH_STAGE7 0         // 1: 7 x 7 lattices: outputs through LDS, whole lines per store.  Measured x3 interleaved (profiles/r06_patch_ab.txt): FashionMNIST 2.1103 ms against 2.0945 with pair stores (2.1032 on the gather launch) -- not the partial lines: off
#endif
#ifndef MVAE_PATCH_STAGE8
#define MVAE_PATCH_STAGE8 0         // 8 x 8 lattices: the same (64-byte output rows)
#endif
#ifndef MVAE_PATCH_CONSTGEO
#define MVAE_PATCH_CONSTGEO 1       // the lattice of an instantiation as compile-time constants (0: run-time divisors, A/B builds)
#endif
#ifndef MVAE_PATCH_MINBLK
#define MVAE_PATCH_MINBLK 4         // blocks per CU the register allocation aims at: the LDS footprint (40 KB) admits four; at 2 the
#endif                              // statistics form took 135 registers = three.  (At 4: 128, eight dwords spilled outside the loops.)
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

// Rows: i0 = 32 * blockIdx.y (Cin = 64: two row halves re-read the patch).  PS: floats per channel in the patch incl. the
// zero slot (PS - 1); X4: 16-byte DMA pieces; NUI: patch DMA instructions per thread and phase.
constexpr int CP2_KPH = 16, CP2_SPP = CP2_KPH / 4;

// E = EpNCHWPair: one tile per block, pair stores.  E = EpStats: TILES = 2 consecutive tiles per block and one (mean, M2)
// record over their 4 x 128 values per row -- the record mvae_bn_stats_merge expects (gemm_core.h STATK).
// SW2: 0 = pair stores straight from the accumulators; else the lattice width of a whole-image (mode a) geometry whose
// outputs pass through LDS (SROWS = lattice rows a 64-position tile can touch).
template <class E, int TILES, int PS, bool X4, int NUI, int SW2, int SROWS>
__global__ __launch_bounds__(256, MVAE_PATCH_MINBLK) void convT_patch2_kernel(const float *__restrict__ dy, const float *__restrict__ wr,
                                                              E e, PatchGeo g) {
    static_assert(!X4 || PS % 4 == 0, "16-byte pieces");
    static_assert(PS == 68 || PS == 100 || PS == 148, "the lattice is a compile-time constant of the instantiation");
    // the lattice this instantiation serves (convt_patch_plan picks it by exactly these geometries): every division of the
    // per-tile set-up below is by a constant -- shifts for the 8 x 8 / 16 x 16 maps.  With run-time divisors the set-up was 249
    // vector instructions per wave and tile (656 on the 7 x 7 maps) against 256 matrix instructions of a 64 -> 32-channel tile.
#if MVAE_PATCH_CONSTGEO
    constexpr int W2c = PS == 68 ? 8 : PS == 100 ? 16 : 7, H2c = W2c, OHWc = W2c * W2c;
#else
    const int W2c = g.W2, H2c = g.H2, OHWc = g.OHW;
#endif
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
        const int n = jj / OHWc, rem = jj - n * OHWc;
        const int ih2 = rem / W2c, iw2 = rem - ih2 * W2c;
        const int n0 = g2_uni(j0 / OHWc);
        const int n1 = g2_uni((min(j0 + 63, g.J - 1)) / OHWc);       // last image of the tile
        const int r0 = g2_uni((j0 - n0 * OHWc) / W2c);
        const int pidx = g.mode_a ? (n - n0) * OHWc + rem : (ih2 - r0 + 1) * W2c + iw2;
        // the wave's ph picks two of the three neighbour rows: tap a = 0 -> row ih' + ph, a = 1 -> row ih' + ph - 1
        int pb[2][3];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int dc = -1; dc <= 1; ++dc) {
                const int dr = ph - a;
                const bool ok = jok && ih2 + dr >= 0 && ih2 + dr < H2c && iw2 + dc >= 0 && iw2 + dc < W2c;
                pb[a][dc + 1] = ((ok ? pidx + dr * W2c + dc : PS - 1) + lrow * PS) * 4;
            }
        int pvoff[NUI];
#pragma unroll
        for (int i = 0; i < NUI; ++i) {
            const int u = i * 256 + t, c = u / PSV, q = (u - c * PSV) * (X4 ? 4 : 1);
            int off = BUF_OOB;
            if (c < CP2_KPH && q < g.ps_raw) {
                if (g.mode_a) {
                    const int img = q / OHWc, pos = q - img * OHWc;
                    if (n0 + img <= n1) off = ((img * g.Cout + c) * OHWc + pos) * 4;       // only the images the tile touches
                } else {
                    const int row = q / W2c, ih = r0 - 1 + row;
                    if (ih >= 0 && ih < H2c) off = (c * OHWc + ih * W2c + (q - row * W2c)) * 4;
                }
            }
            pvoff[i] = off;
        }
        const BufBase dyb = buf_base(dy + (size_t)n0 * g.Cout * OHWc);
        auto issue_patch = [&](int phase) {
            const i32x4_t rs = g2_rsrc(dyb, (long)phase * CP2_KPH * OHWc, 0x7fffffff);
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
        } else if constexpr (SW2 > 0) {
            // ---- through LDS: S[channel][lattice row of the tile][ph][2 b + pw] -- a channel's outputs in memory order
            // (image planes apart; the tile's lattice rows of one image are one contiguous run of 4 * SW2 floats each)
            constexpr int RW = 4 * SW2, SP = SROWS * RW;     // floats per lattice row of outputs / per channel
            static_assert(32 * SP <= 2 * PATCH_FLOATS + CP_STAGES * WT, "the staging image overlays the patch buffers and the ring");
            const int a0 = r0;                               // first lattice row of the tile in image n0
            const int jl = min(j0 + 63, g.J - 1);
            const int nl = g2_uni(jl / g.OHW), al = g2_uni((jl - nl * g.OHW) / SW2);
            const int rows = g2_uni((nl - n0) * g.H2 + al - a0 + 1);      // lattice rows the tile touches
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");          // every wave is done with the ring and the patch
            if (jok) {
                const int grow = (n - n0) * g.H2 + ih2 - a0;
                float *dstp = lds + grow * RW + ph * 2 * SW2 + 2 * iw2;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int cl = (r & 3) + 8 * (r >> 2) + 4 * lrow;
                    *reinterpret_cast<float2 *>(dstp + cl * SP) = make_float2(acc[0][r], acc[1][r]);
                }
            }
            __syncthreads();
            // wave w writes channels 8 w .. 8 w + 7: a lane owns pair p of the tile's run (64 lanes = 512 contiguous bytes
            // inside an image), geometry once per pair, reused for the eight channels
            const int np = rows * (RW / 2);
            const int HWo = e.HW, Wo = e.Wfull;
            for (int p2 = lane; p2 < np; p2 += 64) {
                const int grow = p2 / (RW / 2), within = p2 - grow * (RW / 2);
                const int pph = within / SW2, b = within - pph * SW2;
                const int gr = a0 + grow, img = gr / g.H2, a = gr - img * g.H2;
                const int jq = ((n0 + img) * g.H2 + a) * SW2 + b;
                const bool owned = jq >= j0 && jq <= jl;
                const size_t o0 = ((size_t)(n0 + img) * e.C + i0 + wave * 8) * HWo + (2 * a + pph) * Wo + 2 * b;
                const float *src = lds + (wave * 8) * SP + 2 * p2;
                if (owned) {
#pragma unroll
                    for (int ch = 0; ch < 8; ++ch) {
                        if (i0 + wave * 8 + ch >= e.C) break;
                        float2 v = *reinterpret_cast<const float2 *>(src + ch * SP);
                        const size_t o = o0 + (size_t)ch * HWo;
                        if (e.dpre) {
                            const float2 d = *reinterpret_cast<const float2 *>(e.dpre + o);
                            v.x *= swish_grad_(d.x); v.y *= swish_grad_(d.y);
                        }
                        if (e.out) *reinterpret_cast<float2 *>(e.out + o) = v;
                        if (e.act) *reinterpret_cast<float2 *>(e.act + o) = make_float2(swishf_(v.x), swishf_(v.y));
                    }
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

template <class E, int TILES, int PS, bool X4, int SW2 = 0, int SROWS = 0>
int launch_convt_patch2(const PatchPlan &pl, const float *dy, const float *wr, const E &e, hipStream_t st) {
    constexpr int PSV = X4 ? PS / 4 : PS;
    constexpr int NUI = (CP2_KPH * PSV + 255) / 256;
    constexpr size_t lds = ((size_t)2 * NUI * 256 * (X4 ? 4 : 1) + (size_t)CP_STAGES * 4 * CP_BK * 32) * sizeof(float);
    auto kern = convT_patch2_kernel<E, TILES, PS, X4, NUI, SW2, SROWS>;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(pl.blocks / TILES, pl.g.Cin / 32), dim3(256), lds, st, dy, wr, e, pl.g);
    return mvae_launch_status();
}

}  // namespace
