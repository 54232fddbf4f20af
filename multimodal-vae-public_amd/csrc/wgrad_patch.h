// wgrad_patch.h -- weight gradient of the stride-2 4x4 convs / transposed convs from operands in their NATURAL layout
// (round 6).  Conv2d(32,64) / (64,128) and ConvTranspose2d(128,64) / (64,32): celeba/model.py:78-83,119-124.
//
//   dw[sc][bc][kh][kw] = sum over (n, a, b) of  S[n][sc][a][b] * L[n][bc][2a - 1 + kh][2b - 1 + kw]
//
// S = the tensor on the SMALL map (dy of a conv, x of a transposed conv), L = the one on the map twice as large; a GEMM of
// SC rows x (BC * 16) columns over the B * OH * OW lattice positions.  The implicit-GEMM launch of gemm_core.h gathers BOTH
// operands position by position (lanes along the reduction): 32 dword loads + 32 ds_write_b32 per thread for every 64
// matrix instructions of a wave, every element of L sixteen times over.  Here a block (64 rows x 8 big channels = 128
// columns, 4 waves of 32 x 64) walks CHUNKS of 64 positions (an 8 x 8 image; four rows of a 16 x 16 one):
//   * S comes in as it lies in memory -- [row][64 positions], 256 contiguous bytes per row -- by 16-byte LDS-DMA pieces into
//     rows of 17 float4 (an odd pitch: the ds_read_b128 fragment reads are conflict-free without a swizzle);
//   * the rows of L the chunk touches come in as a zero-bordered image [8 channels][2 * rows + 2][W + 2] by 4-byte pieces
//     (a border or an absent row is a piece with an out-of-range source: the hardware writes 0); the B fragment of column
//     (bc, kh, kw) at position (a, b) is ONE ds_read_b32 at  lane part (bc, kh, kw, upper half wave) + immediate (a, b):
//     no im2col image, no address arithmetic in the loop;
//   * two such buffers: the DMA of chunk q + 1 flies while chunk q is multiplied; one barrier per 64 matrix instructions.
// (The stride-1 layers on 5 x 5 <-> 8 x 8 maps were tried the same way -- two images per chunk, S by 4-byte pieces -- and ran
// equal to 12 % slower than the implicit-GEMM launch: removed, profiles/r06_wgrad_patch_bench.txt.)
// The reduction is cut over blocks (grid.z); the partial tiles go to the caller's scratch in the layout of gemm_core.h's
// split launches and its finish kernels sum them in a fixed order -- deterministic, no atomics.
#pragma once
#include "gemm2.h"

namespace {

#ifndef MVAE_WGRAD_PATCH
#define MVAE_WGRAD_PATCH 1          // 0: these layers stay on igemm_kernel<LdWgradDy, LdWgradX> (A/B builds)
#endif
#ifndef MVAE_WGRAD_PATCH_BLOCKS
#define MVAE_WGRAD_PATCH_BLOCKS 512 // blocks a launch aims at (tiles x splits)
#endif

struct WgradPatchGeo {
    int B, SC, BC;                  // images, channels of S (rows), channels of L
    int OH, OW, H, W;               // small map, big map (H = 2 OH)
    int nch, cpi;                   // chunks in all, chunks per image
    int splits;
    float *ws; size_t stride;       // partial (split, i, j) at ws[split * stride + i * J + j], J = BC * 16
};

constexpr int WP_PA = 68;           // floats per row of the S tile: 16 float4 + 1 of padding
constexpr int WP_NUA = 5;           // 16-byte DMA pieces per thread and chunk: 64 rows x 17 = 1088 <= 1280
constexpr int WP_NUB = 11;          // 4-byte pieces: 8 channels x 324 (8 x 8) / 340 (16 x 16) floats <= 2816
constexpr int WP_A_FLOATS = WP_NUA * 256 * 4, WP_B_FLOATS = WP_NUB * 256, WP_BUF = WP_A_FLOATS + WP_B_FLOATS;

template <int W2>
__global__ __launch_bounds__(256, 2) void wgrad_patch_kernel(const float *__restrict__ S, const float *__restrict__ L,
                                                             WgradPatchGeo g) {
    static_assert(W2 == 8 || W2 == 16, "lattice width");
    constexpr int RPC = 64 / W2;                    // small rows per chunk
    constexpr int PWB = 2 * W2 + 2, ROWSB = 2 * RPC + 2, PSB = ROWSB * PWB;
    static_assert(8 * PSB <= WP_B_FLOATS, "the L image fits its pieces");
    extern __shared__ __attribute__((aligned(16))) float lds[];     // [2][S tile | L image]
    const int t = threadIdx.x, lane = t & 63;
    const int wave = g2_uni(t >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int lrow = lane >> 5, lcol = lane & 31;
    const int bc0 = blockIdx.x * 8, i0 = blockIdx.y * 64, z = blockIdx.z;
    const unsigned lds0 = (unsigned)(unsigned long)(g2_lds_void *)lds;
    const int OHW = g.OH * g.OW, HW = g.H * g.W;
    const int q_lo = (int)((long)z * g.nch / g.splits), q_hi = (int)((long)(z + 1) * g.nch / g.splits);

    // ---- DMA pieces.  S: unit u = i * 256 + t -> (row u / 17, float4 u % 17); L: (channel, image row, image column)
    int avoff[WP_NUA], bvoff[WP_NUB];
    unsigned top = 0, bot = 0;                      // L pieces in the first / last image row (absent at the map's edges)
#pragma unroll
    for (int i = 0; i < WP_NUA; ++i) {
        const int u = i * 256 + t, row = u / 17, f = u - row * 17;
        avoff[i] = (row < 64 && f < 16) ? (row * OHW + 4 * f) * 4 : BUF_OOB;
    }
#pragma unroll
    for (int i = 0; i < WP_NUB; ++i) {
        const int u = i * 256 + t, bc = u / PSB, rr = u - bc * PSB, r = rr / PWB, c = rr - r * PWB;
        const bool ok = bc < 8 && c >= 1 && c <= g.W;
        bvoff[i] = ok ? (bc * HW + r * g.W + c) * 4 : BUF_OOB;      // from (first channel, image row 2 a0 - 1, column -1)
        if (ok && r == 0) top |= 1u << i;
        if (ok && r == ROWSB - 1) bot |= 1u << i;
    }
    const BufBase sb = buf_base(S + (size_t)i0 * OHW), lb = buf_base(L + (size_t)bc0 * HW);
    // the 16 pieces of a chunk are issued TWO per group of eight matrix instructions of the chunk before it (all sixteen in
    // front of the first one stalled the wave's matrix stream for ~1600 cycles per 4096: 60-85 TFLOP/s)
    i32x4_t rsa, rsb;
    bool first = false, last = false;
    unsigned dbase = 0;
    auto point = [&](int q, int buf) {
        const int n = q / g.cpi, sub = q - n * g.cpi;
        rsa = g2_rsrc(sb, (long)n * g.SC * OHW + sub * 64, 0x7fffffff);
        // the image starts one row above and one column left of the chunk's first tap: the offset may be negative for the
        // pieces that are masked out below
        rsb = g2_rsrc(lb, (long)n * g.BC * HW + (long)(2 * sub * RPC - 1) * g.W - 1, 0x7fffffff);
        first = sub == 0; last = sub == g.cpi - 1;
        dbase = lds0 + buf * WP_BUF * 4;
    };
    auto piece = [&](int i) {                       // i: compile-time after unrolling
        if (i < WP_NUA) {
            g2_dma16(rsa, avoff[i], g2_uni(dbase + (i * 256 + wave * 64) * 16));
        } else {
            const int k = i - WP_NUA;
            const bool edge = (first && ((top >> k) & 1u)) || (last && ((bot >> k) & 1u));
            g2_dma4(rsb, edge ? BUF_OOB : bvoff[k], 0, g2_uni(dbase + WP_A_FLOATS * 4 + (k * 256 + wave * 64) * 4));
        }
    };
    static_assert(WP_NUA + WP_NUB == 16, "two pieces per group of eight matrix instructions");

    f32x16 acc[2];
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[y][r] = 0.f;

    // fragment addresses (bytes inside a buffer).  Position 8 c + j of the chunk sits in lanes 0-31, 8 c + 4 + j in lanes 32-63.
    const int aoff = ((wm * 32 + lcol) * WP_PA + 4 * lrow) * 4;
    const int tap = lcol & 15, kh = tap >> 2, kw = tap & 3;
    int boff[2];
#pragma unroll
    for (int y = 0; y < 2; ++y)
        boff[y] = (WP_A_FLOATS + ((wn * 2 + y) * 2 + (lcol >> 4)) * PSB + kh * PWB + kw + 8 * lrow) * 4;

    if (q_lo < q_hi) {
        point(q_lo, 0);
        asm volatile("s_nop 4" ::: "memory");
#pragma unroll
        for (int i = 0; i < 16; ++i) piece(i);
    }
    for (int q = q_lo; q < q_hi; ++q) {
        const int buf = (q - q_lo) & 1;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // chunk q landed; everyone is done with chunk q - 1
        const bool more = q + 1 < q_hi;             // block-uniform
        if (more) point(q + 1, buf ^ 1);
        const char *base = reinterpret_cast<const char *>(lds) + buf * WP_BUF * 4;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float4 a4 = *reinterpret_cast<const float4 *>(base + aoff + c * 32);
            // position 8 c + j: row (8 c) / W2 of the chunk, column (8 c) % W2 + j
            const int imm = (2 * ((8 * c) / W2) * PWB + 2 * ((8 * c) % W2)) * 4;
            float bv[4][2];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int y = 0; y < 2; ++y) bv[j][y] = *reinterpret_cast<const float *>(base + boff[y] + imm + 2 * j * 4);
            if (more) { piece(2 * c); piece(2 * c + 1); }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float av = j == 0 ? a4.x : j == 1 ? a4.y : j == 2 ? a4.z : a4.w;
#pragma unroll
                for (int y = 0; y < 2; ++y) acc[y] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[j][y], acc[y], 0, 0, 0);
            }
        }
    }
    // ---- the partial tile
    const int J = g.BC * 16;
    float *dst = g.ws + (size_t)z * g.stride + (size_t)(i0 + wm * 32 + 4 * lrow) * J + bc0 * 16 + wn * 64 + lcol;
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[(size_t)((r & 3) + 8 * (r >> 2)) * J + y * 32] = acc[y][r];
}

// the launch for a layer, or false (the caller keeps the implicit-GEMM launch)
inline bool wgrad_patch_plan(int B, int SC, int BC, int OH, int OW, const void *S, const void *L, void *ws, size_t ws_bytes,
                             WgradPatchGeo *g) {
    if (!MVAE_WGRAD_PATCH) return false;
#ifdef MVAE_TUNING
    if (getenv("MVAE_WGRAD_PATCH_OFF")) return false;
#endif
    if (!((OH == 8 && OW == 8) || (OH == 16 && OW == 16)) || SC % 64 != 0 || BC % 8 != 0 || !ws) return false;
    if (((size_t)S & 15) || ((size_t)L & 3)) return false;
    const long big = (long)B * BC * OH * OW * 4;
    if (big * 4 >= (1L << 31) || (long)B * SC * OH * OW * 4 >= (1L << 31)) return false;
    g->B = B; g->SC = SC; g->BC = BC; g->OH = OH; g->OW = OW; g->H = 2 * OH; g->W = 2 * OW;
    g->cpi = OH * OW / 64; g->nch = B * g->cpi;
    const size_t per_split = (size_t)SC * BC * 16 * sizeof(float);
    const long tiles = (long)(SC / 64) * (BC / 8);
    // two blocks per CU when a block then still walks >= 24 chunks, else one (fewer partial tiles to write and to sum:
    // profiles/r06_wgrad_patch_bench.txt -- 512 images of 8 x 8: 83 us either way; 256 images of 16 x 16: 59 vs 50 us)
    long target = ((long)B * (OH * OW / 64) * tiles >= 24L * MVAE_WGRAD_PATCH_BLOCKS) ? MVAE_WGRAD_PATCH_BLOCKS : MVAE_WGRAD_PATCH_BLOCKS / 2;
#ifdef MVAE_TUNING
    if (const char *tb = getenv("MVAE_WGRAD_PATCH_TARGET")) target = atol(tb);
#endif
    long s = target / tiles;
    if (s > g->nch) s = g->nch;
    if ((size_t)s > ws_bytes / per_split) s = (long)(ws_bytes / per_split);
    if (s < 1 || s > 1024 || tiles * s < 256) return false;
    g->splits = (int)s; g->ws = (float *)ws; g->stride = (size_t)SC * BC * 16;
    return true;
}

inline void launch_wgrad_patch(const WgradPatchGeo &g, const float *S, const float *L, hipStream_t st) {
    constexpr size_t lds = (size_t)2 * WP_BUF * sizeof(float);
    const dim3 grid(g.BC / 8, g.SC / 64, g.splits);
#define MVAE_WP(W2_)                                                                                              \
    {                                                                                                             \
        auto kern = wgrad_patch_kernel<W2_>;                                                                      \
        static bool attr_done = false;                                                                            \
        if (!attr_done) {                                                                                         \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            attr_done = true;                                                                                     \
        }                                                                                                         \
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, S, L, g);                                              \
    }
    if (g.OW == 8) MVAE_WP(8) else MVAE_WP(16)
#undef MVAE_WP
}

}  // namespace
