// conv.hip -- Conv2d / ConvTranspose2d 4x4 (forward, data gradient, weight gradient) on the implicit-GEMM
// kernel of gemm_core.h: the gather loaders, the three special-shape kernels and the C ABI.
#include "gemm_core.h"
#include "gemm2.h"
#include "convt_patch.h"
#include "wgrad_patch.h"
#include "conv_patch.h"

// k-tile depth of the gather-fed forward / dgrad forms (32 or 64; the weight-gradient loaders decode their
// tap fields for 32)
#ifndef MVAE_SC_BLOCKS
#define MVAE_SC_BLOCKS 256      // blocks (8 waves each) of the small-Cin conv weight gradient: one per CU
#endif
#ifndef MVAE_CONV_SMALL_FWD
#define MVAE_CONV_SMALL_FWD 1   // <= 4-input-channel stride-2 conv forward: direct VALU kernel instead of a K <= 48 GEMM
#endif
#ifndef MVAE_CONVT_SMALL2
#define MVAE_CONVT_SMALL2 1     // <= 4-channel transposed conv: two positions per thread, weights through scalar loads
#endif
#ifndef MVAE_CLS_MINOR
#define MVAE_CLS_MINOR 1        // parity classes of one tile adjacent in launch order (see igemm_kernel)
#endif
#ifndef MVAE_MULTI_ITEMS
#define MVAE_MULTI_ITEMS 4      // (class, j tile) items a block of the conv forward / dgrad forms walks (1: off)
#endif
#ifndef MVAE_MULTI_MAXK
#define MVAE_MULTI_MAXK 256     // ... when the reduction is at most this long (measured: K = 512 tiles lose 2-3 %)
#endif
#ifndef MVAE_CONV_BK
#define MVAE_CONV_BK 32
#endif
#ifndef MVAE_FAST_DIV
#define MVAE_FAST_DIV 1         // shifts instead of run-time divisions in the tile set-up / epilogue of power-of-two layers (0: A/B)
#endif
// XCD-aware launch-order re-mapping (round 4; each 0 = plain launch order, for A/B builds)
#ifndef MVAE_CONV_XCD
#define MVAE_CONV_XCD 1         // forward / dgrad forms: the channel bands of one column tile on one XCD (igemm_kernel, mode 3)
#endif
#ifndef MVAE_WGRAD_XCD
#define MVAE_WGRAD_XCD 1        // weight-gradient form: the tiles of one k range on one XCD (igemm_kernel, mode 4)
#endif
#ifndef MVAE_DY_KEEP
#define MVAE_DY_KEEP 1          // 32-row transposed-conv form: the column decode of a tile kept across its parity classes (0: A/B)
#endif
#ifndef MVAE_SMALL_EPI_BATCH
#define MVAE_SMALL_EPI_BATCH 1      // conv_small_fwd_kernel: the producer's pre-activations fetched eight channels at a time
#endif
#ifndef MVAE_CONVT_SMALL3
#define MVAE_CONVT_SMALL3 1         // <= 4-output-channel transposed conv: input rows staged through LDS (convT_small3_kernel)
#endif
#ifndef MVAE_SMALL2_DEPTH
#define MVAE_SMALL2_DEPTH 8
#endif
#ifndef MVAE_S1_XCD
#define MVAE_S1_XCD 1           // convT_s1_kernel: the channel groups of one image group on one XCD
#endif

typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ f32x2_t llvm_raw_buffer_load_f32x2(i32x4_t rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v2f32");

#ifndef MVAE_WIDE_BLOCKS
#define MVAE_WIDE_BLOCKS (1 << 30)   // conv-forward-form launches with >= 128 output channels took 128 x 64 tiles from 384 blocks on
                                     // (rounds 2-4).  With the gather loaders' one-offset-per-tap form and the buffer-store epilogue
                                     // the 64 x 64 kernels run four blocks of ~100 registers per CU and WIN: CelebA -1.6 %,
                                     // FashionMNIST -5.1 %, CelebA-19 -0.7 % (x3 interleaved, profiles/r05_conv_retune_ab.txt); 384: A/B
#endif
#ifndef MVAE_GATHER_COMPACT
#define MVAE_GATHER_COMPACT 1    // gather loaders: one per-lane offset per TAP + the channel on the scalar offset (see LdIm2colT)
#endif

namespace {

// Geometry of a 4x4 convolution y[B,Cout,OH,OW] = conv(x[B,Cin,H,W], w[Cout,Cin,4,4]).
struct ConvGeom {
    int B, Cin, H, W, Cout, OH, OW, stride, pad;
    // log2 of the sizes the loaders divide a tile's column index by, or -1 if not a power of two (make_geom): the output
    // map OH*OW / OW (im2col forms) and the class lattice (H/stride)*(W/stride) / (W/stride) (dgrad forms).  A 32-bit
    // division by a run-time value is ~30 vector instructions, and on fp32 MFMA the vector instructions of a tile's
    // set-up and epilogue are matrix time (profiles/r04_celeba_sq_counters.txt: 2-4 VALU per MFMA over whole conv
    // kernels whose main loops issue 0.2-0.8); every CelebA layer is a power of two (8x8 .. 32x32), the 5x5 / 7x7 ones not.
    int lg_ohw, lg_ow, lg_hw2, lg_w2;
};
__host__ __device__ inline int log2_or_neg(int v) {
    if (v <= 0 || (v & (v - 1))) return -1;
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}
// q = m / d, r = m % d for m >= 0: the shift form when lg >= 0 (block-uniform choice)
__device__ __forceinline__ void divmod_fast(int m, int d, int lg, int &q, int &r) {
    if (lg >= 0) { q = m >> lg; r = m & (d - 1); }
    else { q = m / d; r = m - q * d; }
}

// ---- gather loaders ----------------------------------------------------------------------
// The k index of an element a thread fetches is  k0 + kq + STEP*v  with k0 a multiple of BK = 32,
// kq = thread-constant (< STEP) and v the unrolled element counter.  STEP is a power of two, so the
// (channel, tap-row, tap-col) fields of k are the OR of compile-time fields of STEP*v and the
// thread-constant fields of kq: per element the address is ONE add of a wave-uniform offset, the
// bounds test a compile-time shift of a precomputed bit mask.  (The first version decoded k and
// re-tested the image bounds per element and was VALU-bound: 5 waves x ~250 VALU ops per k-step
// against 1024 MFMA cycles.)

// im2col of x for the forward conv: element (k = (ci,kh,kw), m = (b,oh,ow)); lanes along m.
template <int TILE_, int BKV_ = MVAE_CONV_BK>
struct LdIm2colT {
    static constexpr int TILE = TILE_, BKV = BKV_;
    static constexpr int NV = TILE * BKV / NTHREADS;  // elements per thread
    static constexpr int KSTEP = NTHREADS / TILE;     // 2 or 4: k rows covered per pass
    struct Regs { float v[NV]; unsigned ok; };        // raw data + validity bits (applied when staged)
    const float *x; ConvGeom g; int Mtot;
    int base, kq; unsigned vh, vwq;
    // full k-tiles: buffer base (first image of the tile) + byte offsets.  MVAE_GATHER_COMPACT: an element's offset is
    // (tap part, per lane) + (channel part, the same for every lane), and its validity depends on the tap alone -- so a
    // thread keeps ONE offset per distinct tap among its elements (4 - 8 instead of 16, out-of-range when the tap is outside
    // the image) and the channel part rides the instruction's SCALAR offset: per (re-)initialisation a handful of vector
    // instructions instead of ~6 per element, and 8 - 12 registers fewer (the multi-item blocks re-initialise per item)
    BufBase blk; int voff[MVAE_GATHER_COMPACT ? 16 : NV];
    static constexpr bool fast = true;
    __device__ void begin(int, int) {}
    __device__ void init(int tile0, int t, int) {
        const int m = tile0 + (t % TILE);
        kq = t / TILE;
        unsigned vw = 0;
        vh = 0; base = 0;
        const int ohw = g.OH * g.OW, hw = g.H * g.W;
        const int n0 = g.lg_ohw >= 0 ? tile0 >> g.lg_ohw : tile0 / ohw;      // block-uniform
        blk = buf_base(x + (size_t)n0 * g.Cin * hw);
        int rel = 0;
        if (m < Mtot) {
            int b, rem, oh, ow;
            divmod_fast(m, ohw, g.lg_ohw, b, rem);
            divmod_fast(rem, g.OW, g.lg_ow, oh, ow);
            const int ih0 = oh * g.stride - g.pad, iw0 = ow * g.stride - g.pad;
            base = (b * g.Cin * g.H + ih0) * g.W + iw0 + kq;      // kw = kq + (KSTEP*v & 3)
            rel = ((b - n0) * g.Cin * g.H + ih0) * g.W + iw0 + kq;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (ih0 + q >= 0 && ih0 + q < g.H) vh |= 1u << q;
                if (iw0 + q >= 0 && iw0 + q < g.W) vw |= 1u << q;
            }
        }
        vwq = vw >> kq;
        // element v of a full k-tile: channel (KSTEP*v >> 4) of the tile's channel group, tap (kh, kw); a tap
        // outside the image (or a lane without an output position) reads as zero through BUF_OOB
        if (MVAE_GATHER_COMPACT) {
#pragma unroll
            for (int p = 0; p < 16; ++p) {          // tap (kh, kwl) = (p >> 2, p & 3); the taps no element has are dead code
                const bool ok = ((vh >> (p >> 2)) & 1u) && ((vwq >> (p & 3)) & 1u);
                voff[p] = ok ? (rel + (p >> 2) * g.W + (p & 3)) * 4 : BUF_OOB;
            }
            return;
        }
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int c = KSTEP * v;
            const int kwl = c & 3, kh = (c >> 2) & 3, cil = c >> 4;
            const bool ok = ((vh >> kh) & 1u) && ((vwq >> kwl) & 1u);
            voff[v] = ok ? (rel + cil * hw + kh * g.W + kwl) * 4 : BUF_OOB;
        }
    }
    __device__ void load(int k0, int kend, int t, Regs &rg) const {
        if (k0 + BKV <= kend) {                       // block-uniform: the whole k-tile is inside the reduction
            load_part(k0, kend, t, rg, 0, 1);
            rg.ok = 0xffffffffu;
            return;
        }
        const int hw = g.H * g.W;
        const float *src = x + base + (k0 >> 4) * hw;
        const int safe = (int)(x - src);              // offset of x[0]: always a legal address
        const int krem = kend - k0 - kq;              // element valid iff KSTEP*v < krem
        unsigned okbits = 0;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int c = KSTEP * v;                  // compile-time after unrolling
            const int kwl = c & 3, kh = (c >> 2) & 3, cil = c >> 4;
            const bool ok = (c < krem) && ((vh >> kh) & 1u) && ((vwq >> kwl) & 1u);
            const float val = src[ok ? cil * hw + kh * g.W + kwl : safe];
            rg.v[v] = val;
            okbits |= (ok ? 1u : 0u) << v;
        }
        rg.ok = okbits;
    }
    static constexpr bool RMAJOR = false;
    static constexpr int ROWS = BKV, PITCH = TILE + LPAD;
    typedef float (*Tile)[PITCH];
    static __device__ __forceinline__ float4 frag(Tile L, int k0, int row) { return frag_kmajor(L, k0, row); }
    __device__ void store(Tile L, int t, const Regs &rg) const {
        const int m = t % TILE, kb = t / TILE;
#pragma unroll
        for (int v = 0; v < NV; ++v) L[kb + v * KSTEP][m] = rg.v[v] * mask0((rg.ok >> v) & 1u);
    }
    // slices of a FULL k-tile (gemm_core.h, buffer loads): no vector-ALU work per element
    static constexpr bool PARTS = true, TAIL = false;
    __device__ __forceinline__ void load_part(int k0, int, int, Regs &rg, int part, int nparts) const {
        const i32x4_t rs = buf_rsrc(blk, (size_t)(k0 >> 4) * (g.H * g.W));
        const int hw4 = g.H * g.W * 4;
#pragma unroll
        for (int v = 0; v < NV; ++v)
            if (MVAE_IN_PART(v, NV, part, nparts)) {
                if (MVAE_GATHER_COMPACT) {
                    const int c = KSTEP * v;        // tap = c & 15, channel of the tile's group = c >> 4 (scalar offset)
                    rg.v[v] = llvm_raw_buffer_load_f32(rs, voff[c & 15], (c >> 4) * hw4, 0);
                } else {
                    rg.v[v] = buf_load1(rs, voff[v]);
                }
            }
    }
    __device__ __forceinline__ void store_part(Tile L, int t, const Regs &rg, int part, int nparts) const {
        const int m = t % TILE, kb = t / TILE;
#pragma unroll
        for (int v = 0; v < NV; ++v)
            if (MVAE_IN_PART(v, NV, part, nparts)) L[kb + v * KSTEP][m] = rg.v[v];
    }
};

// The same gather for the version-2 core (gemm2.h): LDS-DMA, one dword per lane.  A k-tile is 16 deep = the 16 taps of ONE
// input channel; a DMA instruction fills the 64 consecutive columns (output positions) of one tap's row of the k-major image
// [16][TILE].  A lane keeps ONE byte offset per tap it issues (its column's window corner + the tap, out of range when the
// tap falls outside the image or the column does not exist); the channel rides the instruction's SCALAR offset.  Wave w
// issues taps 4w .. 4w + 3 (TILE 64) or half (w & 1), taps 8 (w >> 1) .. + 7 (TILE 128).
template <int TILE_>
struct G2Im2col {
    static constexpr int TILE = TILE_;
    static constexpr bool RK = false;
    static constexpr int NPW = TILE / 16;                   // 16 taps x TILE / 64 pieces over 4 waves
    const float *x; ConvGeom g; int Mtot;
    BufBase blk; int voff[NPW];
    __device__ void init(int tile0, int, int lane, int wave) {
        const int half = TILE == 128 ? (wave & 1) : 0, tap0 = TILE == 128 ? (wave >> 1) * 8 : wave * 4;
        const int m = tile0 + half * 64 + lane;
        const int ohw = g.OH * g.OW;
        const int n0 = g2_uni(g.lg_ohw >= 0 ? tile0 >> g.lg_ohw : tile0 / ohw);
        blk = buf_base(x + (size_t)n0 * g.Cin * g.H * g.W);
        int b, rem, oh, ow;
        divmod_fast(min(m, Mtot - 1), ohw, g.lg_ohw, b, rem);
        divmod_fast(rem, g.OW, g.lg_ow, oh, ow);
        const int ih0 = oh * g.stride - g.pad, iw0 = ow * g.stride - g.pad;
        const int rel = ((b - n0) * g.Cin * g.H + ih0) * g.W + iw0;
#pragma unroll
        for (int u = 0; u < NPW; ++u) {
            const int kh = (tap0 + u) >> 2, kw = (tap0 + u) & 3;
            const bool ok = m < Mtot && ih0 + kh >= 0 && ih0 + kh < g.H && iw0 + kw >= 0 && iw0 + kw < g.W;
            voff[u] = ok ? (rel + kh * g.W + kw) * 4 : BUF_OOB;
        }
    }
    __device__ __forceinline__ void issue(unsigned dst, int k0, int, int wave) const {
        const i32x4_t rs = g2_rsrc(blk, 0, 0x7fffffff);
        const int soff = g2_uni((k0 >> 4) * g.H * g.W * 4);
        const int half = TILE == 128 ? (wave & 1) : 0, tap0 = TILE == 128 ? (wave >> 1) * 8 : wave * 4;
#pragma unroll
        for (int u = 0; u < NPW; ++u) g2_dma4(rs, voff[u], soff, g2_uni(dst + ((tap0 + u) * TILE + half * 64) * 4));
    }
};

// The same gather staged ROW-major -- LDS image [m][k], k contiguous.  A thread owns one output position m
// and whole 4-k groups: k = (ci, kh, kw) with kw fastest, so a group is the 4 horizontal taps of one (ci, kh)
// -- 4 adjacent input floats -- and goes to LDS as ONE float4.  Fragments are then ds_read_b128 (4 MFMAs per
// read) like the weight operand's.  With the k-major image every MFMA of every wave cost two 256-byte
// ds_read_b32; at 16 waves per CU (64x64 tiles, one 32x32 accumulator per wave) that alone kept the LDS
// port ~75 % busy and held the conv kernels near 50 % of the MFMA rate.
template <int TILE_, int BKV_ = MVAE_CONV_BK>
struct LdIm2colR {
    static constexpr int TILE = TILE_, BKV = BKV_;
    static constexpr int GPT = NTHREADS / TILE;       // thread slots along the 4-k groups
    static constexpr int NG = BKV / 4;                // 4-k groups per k-tile
    static constexpr int NV4 = NG / GPT;              // groups per thread
    static_assert(NG % GPT == 0 && NV4 >= 1, "tile / k-depth combination not covered");
    struct Regs { float4 v[NV4]; unsigned ok; };      // raw data + 4 validity bits per group
    const float *x; ConvGeom g; int Mtot;
    int base, gq; unsigned vh, vw;
    BufBase blk; int voff[NV4][4];                    // full k-tiles: buffer base (first image of the tile) + byte offsets
    __device__ void init(int tile0, int t, int) {
        const int m = tile0 + (t % TILE);
        gq = t / TILE;
        vh = 0; vw = 0; base = 0;
        const int ohw = g.OH * g.OW, hw = g.H * g.W;
        const int n0 = tile0 / ohw;                   // block-uniform
        blk = buf_base(x + (size_t)n0 * g.Cin * hw);
        int rel = 0;
        if (m < Mtot) {
            const int b = m / ohw, rem = m - b * ohw;
            const int oh = rem / g.OW, ow = rem - oh * g.OW;
            const int ih0 = oh * g.stride - g.pad, iw0 = ow * g.stride - g.pad;
            base = (b * g.Cin * g.H + ih0) * g.W + iw0;
            rel = ((b - n0) * g.Cin * g.H + ih0) * g.W + iw0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (ih0 + q >= 0 && ih0 + q < g.H) vh |= 1u << q;
                if (iw0 + q >= 0 && iw0 + q < g.W) vw |= 1u << q;
            }
        }
#pragma unroll
        for (int v = 0; v < NV4; ++v) {
            const int gk = gq + GPT * v, kh = gk & 3, cil = gk >> 2;
#pragma unroll
            for (int kw = 0; kw < 4; ++kw) {
                const bool ok = ((vh >> kh) & 1u) && ((vw >> kw) & 1u);
                voff[v][kw] = ok ? (rel + cil * hw + kh * g.W + kw) * 4 : BUF_OOB;
            }
        }
    }
    __device__ void load(int k0, int kend, int t, Regs &rg) const {
        const int hw = g.H * g.W;
        const float *src = x + base + (k0 >> 4) * hw;
        const int safe = (int)(x - src);              // offset of x[0]: always a legal address
        const int grem = (kend - k0) >> 2;            // groups of this tile inside the reduction (K % 16 == 0)
        unsigned okbits = 0;
#pragma unroll
        for (int v = 0; v < NV4; ++v) {
            const int gk = gq + GPT * v;              // group = (ci_local, kh)
            const int kh = gk & 3, cil = gk >> 2;
            const bool okg = gk < grem && ((vh >> kh) & 1u);
            const int off = cil * hw + kh * g.W;
            float e[4];
#pragma unroll
            for (int kw = 0; kw < 4; ++kw) {
                const bool ok = okg && ((vw >> kw) & 1u);
                e[kw] = src[ok ? off + kw : safe];
                okbits |= (ok ? 1u : 0u) << (4 * v + kw);
            }
            rg.v[v] = make_float4(e[0], e[1], e[2], e[3]);
        }
        rg.ok = okbits;
    }
    static constexpr bool RMAJOR = true, PARTS = true, TAIL = false, fast = true;
    __device__ void begin(int, int) {}
    __device__ __forceinline__ void load_part(int k0, int, int, Regs &rg, int part, int nparts) const {
        const i32x4_t rs = buf_rsrc(blk, (size_t)(k0 >> 4) * (g.H * g.W));
#pragma unroll
        for (int v = 0; v < NV4; ++v)
            if (MVAE_IN_PART(v, NV4, part, nparts))
                rg.v[v] = make_float4(buf_load1(rs, voff[v][0]), buf_load1(rs, voff[v][1]), buf_load1(rs, voff[v][2]),
                                      buf_load1(rs, voff[v][3]));
    }
    __device__ __forceinline__ void store_part(float (*L)[BKV_ + LPAD], int t, const Regs &rg, int part, int nparts) const {
        const int m = t % TILE;
#pragma unroll
        for (int v = 0; v < NV4; ++v)
            if (MVAE_IN_PART(v, NV4, part, nparts)) *reinterpret_cast<float4 *>(&L[m][4 * (gq + GPT * v)]) = rg.v[v];
    }
    static constexpr int ROWS = TILE, PITCH = BKV + LPAD;
    typedef float (*Tile)[PITCH];
    static __device__ __forceinline__ float4 frag(Tile L, int k0, int row) {
        return *reinterpret_cast<const float4 *>(&L[row][k0]);
    }
    __device__ void store(Tile L, int t, const Regs &rg) const {
        const int m = t % TILE;
#pragma unroll
        for (int v = 0; v < NV4; ++v) {
            const unsigned b = rg.ok >> (4 * v);
            *reinterpret_cast<float4 *>(&L[m][4 * (gq + GPT * v)]) =
                make_float4(rg.v[v].x * mask0(b & 1u), rg.v[v].y * mask0(b & 2u), rg.v[v].z * mask0(b & 4u),
                            rg.v[v].w * mask0(b & 8u));
        }
    }
};
#ifndef MVAE_GATHER_ROWMAJOR
#define MVAE_GATHER_ROWMAJOR 0      // measured (r2): no gain over the k-major image, see DESIGN.md
#endif
#if MVAE_GATHER_ROWMAJOR
template <int T> using LdIm2col = LdIm2colR<T>;
#else
template <int T> using LdIm2col = LdIm2colT<T>;
#endif


// Transposed-conv (dgrad) gather of dy for the output parity class `cls` = (ph,pw) of dx:
// element (k = (co,a,b), m = (n, ih', iw')) with ih = ih'*s + ph, kh = kh0 + s*a,
// oh = (ih + pad - kh0)/s - a.  TPD = 4/s taps per dim (TLOG = log2 TPD); only the taps that can
// reach the class are enumerated, so stride 2 does no multiply-by-zero work.
template <int TILE_, int TLOG, int BKV_ = MVAE_CONV_BK>
struct LdDgradDyT {
    static constexpr int TILE = TILE_, BKV = BKV_;
    static constexpr bool PAIRABLE = (TLOG == 1);     // stride 2: classes = the 2 x 2 output parities (EpNCHW pair stores)
    static constexpr int NV = TILE * BKV / NTHREADS;
    static constexpr int KSTEP = NTHREADS / TILE;
    static constexpr int TMASK = (1 << TLOG) - 1;
    struct Regs { float v[NV]; unsigned ok; };        // raw data + validity bits (applied when staged)
    const float *dy; ConvGeom g; int Mtot; int H2, W2;
    int base, kq; unsigned vhq, vwq;
    static constexpr int NPAT = (TMASK + 1) * (TMASK + 1);      // taps (a, b) per class
    BufBase blk; int voff[MVAE_GATHER_COMPACT ? NPAT : NV];     // full k-tiles: buffer base (first image of the tile) + byte offsets (see LdIm2colT)
    // The column -> (image, row', col') decode of a tile (two run-time divisions per thread) is the same for every
    // parity class of that tile, and a multi-item block walks the classes of ONE tile back to back (class-minor
    // order): kept across init() calls.  The (stride, pad) pair is the template's: 4x4 convs here are (2, 1) or (1, 0).
    // Only the 32-row layout (128-column tiles, 8 k-steps per item) keeps it: there the set-up is as long as the loop;
    // the 64-row kernels sit at 127 registers and the three extra ones would cost them an occupancy step.
    static constexpr int S = (TLOG == 1) ? 2 : 1, PD = (TLOG == 1) ? 1 : 0;
    static constexpr bool KEEP = (TILE_ == 128) && MVAE_DY_KEEP;
    int c_tile0 = -1, c_n = 0, c_n0 = 0, c_ih2 = 0, c_iw2 = 0;
    static constexpr bool fast = true;
    __device__ void begin(int, int) {}
    __device__ void init(int tile0, int t, int cls) {
        const int ph = cls / S, pw = cls % S;
        const int kh0 = (ph + PD) % S, kw0 = (pw + PD) % S;
        const int m = tile0 + (t % TILE);
        kq = t / TILE;
        const int aq = (kq >> TLOG) & TMASK, bq = kq & TMASK;     // thread-constant tap fields
        unsigned vh = 0, vw = 0;
        base = 0;
        const int hw2 = H2 * W2, ohw = g.OH * g.OW;
        if (!KEEP || tile0 != c_tile0) {              // block-uniform
            c_tile0 = tile0;
            c_n0 = g.lg_hw2 >= 0 ? tile0 >> g.lg_hw2 : tile0 / hw2;
            const int mm = m < Mtot ? m : 0;
            int rem;
            divmod_fast(mm, hw2, g.lg_hw2, c_n, rem);
            divmod_fast(rem, W2, g.lg_w2, c_ih2, c_iw2);
        }
        const int n0 = c_n0;
        blk = buf_base(dy + (size_t)n0 * g.Cout * ohw);
        int rel = 0;
        if (m < Mtot) {
            const int n = c_n, ih2 = c_ih2, iw2 = c_iw2;
            const int ohb = (ih2 * S + ph + PD - kh0) / S;
            const int owb = (iw2 * S + pw + PD - kw0) / S;
            base = (n * g.Cout * g.OH + ohb - aq) * g.OW + owb - bq;
            rel = ((n - n0) * g.Cout * g.OH + ohb - aq) * g.OW + owb - bq;
#pragma unroll
            for (int a = 0; a <= TMASK; ++a) {
                if (ohb - a >= 0 && ohb - a < g.OH) vh |= 1u << a;
                if (owb - a >= 0 && owb - a < g.OW) vw |= 1u << a;
            }
        }
        vhq = vh >> aq; vwq = vw >> bq;
        if (MVAE_GATHER_COMPACT) {
#pragma unroll
            for (int p = 0; p < NPAT; ++p) {        // tap (al, bl) = (p >> TLOG, p & TMASK)
                const bool ok = ((vhq >> (p >> TLOG)) & 1u) && ((vwq >> (p & TMASK)) & 1u);
                voff[p] = ok ? (rel - (p >> TLOG) * g.OW - (p & TMASK)) * 4 : BUF_OOB;
            }
            return;
        }
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int c = KSTEP * v;
            const int bl = c & TMASK, al = (c >> TLOG) & TMASK, col = c >> (2 * TLOG);
            const bool ok = ((vhq >> al) & 1u) && ((vwq >> bl) & 1u);
            voff[v] = ok ? (rel + col * ohw - al * g.OW - bl) * 4 : BUF_OOB;
        }
    }
    __device__ void load(int k0, int kend, int t, Regs &rg) const {
        if (k0 + BKV <= kend) {                       // block-uniform: the whole k-tile is inside the reduction
            load_part(k0, kend, t, rg, 0, 1);
            rg.ok = 0xffffffffu;
            return;
        }
        const int ohw = g.OH * g.OW;
        const float *src = dy + base + (k0 >> (2 * TLOG)) * ohw;
        const int safe = (int)(dy - src);
        const int krem = kend - k0 - kq;
        unsigned okbits = 0;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int c = KSTEP * v;
            const int bl = c & TMASK, al = (c >> TLOG) & TMASK, col = c >> (2 * TLOG);
            const bool ok = (c < krem) && ((vhq >> al) & 1u) && ((vwq >> bl) & 1u);
            const float val = src[ok ? col * ohw - al * g.OW - bl : safe];
            rg.v[v] = val;
            okbits |= (ok ? 1u : 0u) << v;
        }
        rg.ok = okbits;
    }
    static constexpr bool RMAJOR = false;
    static constexpr int ROWS = BKV, PITCH = TILE + LPAD;
    typedef float (*Tile)[PITCH];
    static __device__ __forceinline__ float4 frag(Tile L, int k0, int row) { return frag_kmajor(L, k0, row); }
    __device__ void store(Tile L, int t, const Regs &rg) const {
        const int m = t % TILE, kb = t / TILE;
#pragma unroll
        for (int v = 0; v < NV; ++v) L[kb + v * KSTEP][m] = rg.v[v] * mask0((rg.ok >> v) & 1u);
    }
    static constexpr bool PARTS = true, TAIL = false;               // slices of a FULL k-tile: buffer loads, nothing on the vector ALU
    __device__ __forceinline__ void load_part(int k0, int, int, Regs &rg, int part, int nparts) const {
        const i32x4_t rs = buf_rsrc(blk, (size_t)(k0 >> (2 * TLOG)) * (g.OH * g.OW));
        const int ohw4 = g.OH * g.OW * 4;
#pragma unroll
        for (int v = 0; v < NV; ++v)
            if (MVAE_IN_PART(v, NV, part, nparts)) {
                if (MVAE_GATHER_COMPACT) {
                    const int c = KSTEP * v;        // tap = low 2 TLOG bits, output channel of the tile's group above them
                    rg.v[v] = llvm_raw_buffer_load_f32(rs, voff[c & (NPAT - 1)], (c >> (2 * TLOG)) * ohw4, 0);
                } else {
                    rg.v[v] = buf_load1(rs, voff[v]);
                }
            }
    }
    __device__ __forceinline__ void store_part(Tile L, int t, const Regs &rg, int part, int nparts) const {
        const int m = t % TILE, kb = t / TILE;
#pragma unroll
        for (int v = 0; v < NV; ++v)
            if (MVAE_IN_PART(v, NV, part, nparts)) L[kb + v * KSTEP][m] = rg.v[v];
    }
};
// Row-major staging of the same gather (see LdIm2colR): a 4-k group is the 2x2 taps of one output channel
// (stride 2) or the 4 horizontal taps of one (channel, vertical tap) (stride 1).
template <int TILE_, int TLOG, int BKV_ = MVAE_CONV_BK>
struct LdDgradDyR {
    static constexpr int TILE = TILE_, BKV = BKV_;
    static constexpr int GPT = NTHREADS / TILE;
    static constexpr int NG = BKV / 4;
    static constexpr int NV4 = NG / GPT;
    static_assert(NG % GPT == 0 && NV4 >= 1, "tile / k-depth combination not covered");
    struct Regs { float4 v[NV4]; unsigned ok; };
    const float *dy; ConvGeom g; int Mtot; int H2, W2;
    int base, gq; unsigned vh, vw;
    BufBase blk; int voff[NV4][4];                    // full k-tiles: buffer base (first image of the tile) + byte offsets
    __device__ void init(int tile0, int t, int cls) {
        const int ph = cls / g.stride, pw = cls % g.stride;
        const int kh0 = (ph + g.pad) % g.stride, kw0 = (pw + g.pad) % g.stride;
        const int m = tile0 + (t % TILE);
        gq = t / TILE;
        vh = 0; vw = 0; base = 0;
        const int hw2 = H2 * W2, ohw = g.OH * g.OW;
        const int n0 = tile0 / hw2;                   // block-uniform
        blk = buf_base(dy + (size_t)n0 * g.Cout * ohw);
        int rel = 0;
        if (m < Mtot) {
            const int n = m / hw2, rem = m - n * hw2;
            const int ih2 = rem / W2, iw2 = rem - ih2 * W2;
            const int ohb = (ih2 * g.stride + ph + g.pad - kh0) / g.stride;
            const int owb = (iw2 * g.stride + pw + g.pad - kw0) / g.stride;
            base = (n * g.Cout * g.OH + ohb) * g.OW + owb;
            rel = ((n - n0) * g.Cout * g.OH + ohb) * g.OW + owb;
#pragma unroll
            for (int a = 0; a < (1 << TLOG); ++a) {
                if (ohb - a >= 0 && ohb - a < g.OH) vh |= 1u << a;
                if (owb - a >= 0 && owb - a < g.OW) vw |= 1u << a;
            }
        }
#pragma unroll
        for (int v = 0; v < NV4; ++v) {
            const int gk = gq + GPT * v;
            const int col = TLOG == 1 ? gk : gk >> 2, a_g = TLOG == 1 ? 0 : gk & 3;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int a = TLOG == 1 ? (q >> 1) : a_g, b = TLOG == 1 ? (q & 1) : q;
                const bool ok = ((vh >> a) & 1u) && ((vw >> b) & 1u);
                voff[v][q] = ok ? (rel + col * ohw - a * g.OW - b) * 4 : BUF_OOB;
            }
        }
    }
    __device__ void load(int k0, int kend, int t, Regs &rg) const {
        const int ohw = g.OH * g.OW;
        const float *src = dy + base + (k0 >> (2 * TLOG)) * ohw;
        const int safe = (int)(dy - src);
        const int grem = (kend - k0) >> 2;
        unsigned okbits = 0;
#pragma unroll
        for (int v = 0; v < NV4; ++v) {
            const int gk = gq + GPT * v;
            // stride 2 (TLOG 1): group = channel, elements (a, b) = (e >> 1, e & 1);
            // stride 1 (TLOG 2): group = (channel, a), elements b = e
            const int col = TLOG == 1 ? gk : gk >> 2, a_g = TLOG == 1 ? 0 : gk & 3;
            float e[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int a = TLOG == 1 ? (q >> 1) : a_g, b = TLOG == 1 ? (q & 1) : q;
                const bool ok = gk < grem && ((vh >> a) & 1u) && ((vw >> b) & 1u);
                e[q] = src[ok ? col * ohw - a * g.OW - b : safe];
                okbits |= (ok ? 1u : 0u) << (4 * v + q);
            }
            rg.v[v] = make_float4(e[0], e[1], e[2], e[3]);
        }
        rg.ok = okbits;
    }
    static constexpr bool RMAJOR = true, PARTS = true, TAIL = false, fast = true;
    __device__ void begin(int, int) {}
    __device__ __forceinline__ void load_part(int k0, int, int, Regs &rg, int part, int nparts) const {
        const i32x4_t rs = buf_rsrc(blk, (size_t)(k0 >> (2 * TLOG)) * (g.OH * g.OW));
#pragma unroll
        for (int v = 0; v < NV4; ++v)
            if (MVAE_IN_PART(v, NV4, part, nparts))
                rg.v[v] = make_float4(buf_load1(rs, voff[v][0]), buf_load1(rs, voff[v][1]), buf_load1(rs, voff[v][2]),
                                      buf_load1(rs, voff[v][3]));
    }
    __device__ __forceinline__ void store_part(float (*L)[BKV_ + LPAD], int t, const Regs &rg, int part, int nparts) const {
        const int m = t % TILE;
#pragma unroll
        for (int v = 0; v < NV4; ++v)
            if (MVAE_IN_PART(v, NV4, part, nparts)) *reinterpret_cast<float4 *>(&L[m][4 * (gq + GPT * v)]) = rg.v[v];
    }
    static constexpr int ROWS = TILE, PITCH = BKV + LPAD;
    typedef float (*Tile)[PITCH];
    static __device__ __forceinline__ float4 frag(Tile L, int k0, int row) {
        return *reinterpret_cast<const float4 *>(&L[row][k0]);
    }
    __device__ void store(Tile L, int t, const Regs &rg) const {
        const int m = t % TILE;
#pragma unroll
        for (int v = 0; v < NV4; ++v) {
            const unsigned b = rg.ok >> (4 * v);
            *reinterpret_cast<float4 *>(&L[m][4 * (gq + GPT * v)]) =
                make_float4(rg.v[v].x * mask0(b & 1u), rg.v[v].y * mask0(b & 2u), rg.v[v].z * mask0(b & 4u),
                            rg.v[v].w * mask0(b & 8u));
        }
    }
};
#if MVAE_GATHER_ROWMAJOR
template <int TILE_> using LdDgradDyS2 = LdDgradDyR<TILE_, 1>;   // stride 2: 2x2 taps per class
template <int TILE_> using LdDgradDyS1 = LdDgradDyR<TILE_, 2>;   // stride 1: all 4x4 taps
#else
template <int TILE_> using LdDgradDyS2 = LdDgradDyT<TILE_, 1>;   // stride 2: 2x2 taps per class
template <int TILE_> using LdDgradDyS1 = LdDgradDyT<TILE_, 2>;   // stride 1: all 4x4 taps
#endif
// the weight-side loaders of those two forms at the same k-tile depth
template <int T> using LdRowsKC = LdRowsKT<T, true, MVAE_CONV_BK>;
template <int T> using LdRowsKSC = LdRowsKT<T, false, MVAE_CONV_BK>;
template <int T> using LdRowsMNC = LdRowsMNT<T, true, MVAE_CONV_BK>;
template <int T> using LdRowsMNSC = LdRowsMNT<T, false, MVAE_CONV_BK>;

// wgrad operands: the reduction runs over k = (b,oh,ow); lanes along k (spatially contiguous).
//
// Full k-tiles take the buffer path (gemm_core.h): the address of an element is a PER-LANE part that depends on
// the lane's reduction index k -- (image, output row, output column) -- plus a part that is the same for the
// whole wave (which channel / tap row the element v belongs to), so the lane part goes into voffset and the
// element part into the scalar soffset of the load: no per-element vector arithmetic.  The lane part is not
// recomputed from k each k-step (two integer divisions by run-time values: ~100 VALU instructions, on an fp32-MFMA
// kernel the whole story -- 6.6 VALU per MFMA, 60-80 TFLOP/s) but ADVANCED: k grows by 32 per k-step, i.e. by
// q32 = 32 / OHW whole images and r32 = 32 % OHW positions, with at most one column wrap and one image wrap.
struct KWalk {              // a lane's position on the k axis of a weight-gradient reduction
    int sp, oh, ow;         // position inside the image, its row and column
    __device__ void start(int k, int OH, int OW) {
        const int ohw = OH * OW;
        const int b = k / ohw;
        sp = k - b * ohw; oh = sp / OW; ow = sp - oh * OW;
    }
};

// P: element (i = co, k) = dy[b][co][oh][ow].
template <int TILE_>
struct LdWgradDy {
    static constexpr int TILE = TILE_, BKV = BK;
    static constexpr int NV = TILE * BK / NTHREADS;
    static constexpr int ISTEP = NTHREADS / BK;
    struct Regs { float v[NV]; unsigned ok; };        // raw data + validity bits (applied when staged)
    const float *dy; ConvGeom g;
    int ioff, nvalid;       // ioff = i * OHW of element 0; nvalid = how many of the NV rows are < Cout
    bool fast;              // buffer path usable: the tile is full along i and dy is below 2 GiB
    BufBase blk; int voff, sp, k_pos;                 // buffer path: lane offset (bytes) of element 0, walk state
    __device__ void init(int tile0, int t, int) {
        const int ib = tile0 + t / BK;
        ioff = ib * g.OH * g.OW;
        nvalid = (g.Cout - ib + ISTEP - 1) / ISTEP;     // rows ib + v*ISTEP < Cout  <=>  v < nvalid
        fast = tile0 + TILE <= g.Cout && (size_t)g.B * g.Cout * g.OH * g.OW * 4 < (1ull << 31);
        blk = buf_base(dy);
    }
    __device__ void begin(int kbeg, int t) {
        const int ohw = g.OH * g.OW;
        const int k = kbeg + (t % BK);
        const int b = k / ohw;
        sp = k - b * ohw;
        voff = (b * g.Cout * ohw + sp + ioff) * 4;
        k_pos = kbeg;
    }
    __device__ void load(int k0, int kend, int t, Regs &rg) const {
        const int k = k0 + (t % BK);
        const int ohw = g.OH * g.OW;
        const int b = k / ohw, spx = k - b * ohw;
        const float *src = dy + (size_t)b * g.Cout * ohw + spx + ioff;
        const int safe = (int)(dy - src);
        const int nv = (k < kend) ? nvalid : 0;
        rg.ok = nv >= 32 ? 0xffffffffu : ((1u << nv) - 1u);      // bit v set iff v < nv
#pragma unroll
        for (int v = 0; v < NV; ++v) rg.v[v] = src[(v < nv) ? v * ISTEP * ohw : safe];
    }
    static constexpr bool PARTS = true, TAIL = false;
    __device__ __forceinline__ void load_part(int k0, int, int, Regs &rg, int part, int nparts) {
        const int ohw = g.OH * g.OW;
        if (part == 0) {
            const bool adv = k0 != k_pos;             // block-uniform: the next k-tile (or the same one again)
            const int q32 = BK / ohw, r32 = BK - q32 * ohw;
            sp += adv ? r32 : 0;
            const bool wrap = sp >= ohw;
            sp -= wrap ? ohw : 0;
            voff += (adv ? (r32 + q32 * g.Cout * ohw) * 4 : 0) + (wrap ? (g.Cout - 1) * ohw * 4 : 0);
            k_pos = k0;
        }
        const i32x4_t rs = buf_rsrc(blk, 0);
#pragma unroll
        for (int v = 0; v < NV; ++v)
            if (MVAE_IN_PART(v, NV, part, nparts)) rg.v[v] = llvm_raw_buffer_load_f32(rs, voff, v * ISTEP * ohw * 4, 0);
    }
    static constexpr bool RMAJOR = false;
    static constexpr int ROWS = BK, PITCH = TILE + 1;       // odd pitch: the lanes of a store run along k (one row each)
    typedef float (*Tile)[PITCH];
    static __device__ __forceinline__ float4 frag(Tile L, int k0, int row) { return frag_kmajor(L, k0, row); }
    __device__ void store(Tile L, int t, const Regs &rg) const {
        const int kl = t % BK, ib = t / BK;
#pragma unroll
        for (int v = 0; v < NV; ++v) L[kl][ib + v * ISTEP] = rg.v[v] * mask0((rg.ok >> v) & 1u);
    }
    __device__ __forceinline__ void store_part(Tile L, int t, const Regs &rg, int part, int nparts) const {
        const int kl = t % BK, ib = t / BK;
#pragma unroll
        for (int v = 0; v < NV; ++v)
            if (MVAE_IN_PART(v, NV, part, nparts)) L[kl][ib + v * ISTEP] = rg.v[v];
    }
};

// Q: element (k, j = (ci,kh,kw)) = x[b][ci][oh*s-p+kh][ow*s-p+kw];  j = j0 + jq + 8*v, jq = t/32 < 8.
template <int TILE_>
struct LdWgradX {
    static constexpr int TILE = TILE_, BKV = BK;
    static constexpr int NV = TILE * BK / NTHREADS;
    static constexpr int JSTEP = NTHREADS / BK;       // 8
    struct Regs { float v[NV]; unsigned ok; };        // raw data + validity bits (applied when staged)
    const float *x; ConvGeom g; int J;
    int joff, jq, nvalid;
    bool fast;              // buffer path usable: the tile is full along j and x is below 2 GiB
    BufBase blk; int voff, k_pos; KWalk w;            // buffer path: lane offset (bytes) of tap (jq >> 2, jq & 3), walk state
    __device__ void init(int tile0, int t, int) {
        jq = t / BK;                                  // kw = jq & 3, kh = (jq >> 2) + 2*(v & 1), ci = j0/16 + (v >> 1)
        joff = (tile0 >> 4) * g.H * g.W + (jq >> 2) * g.W + (jq & 3);
        nvalid = (J - tile0 - jq + JSTEP - 1) / JSTEP;
        fast = tile0 + TILE <= J && ((size_t)g.B * g.Cin * g.H * g.W + g.pad * g.W + g.pad) * 4 < (1ull << 31);
        // voffset is UNSIGNED to the hardware: the base sits pad rows + pad columns before x, so that the offset
        // of a lane whose own tap (jq >> 2, jq & 3) is above / left of the image but whose tap two rows down
        // (soffset) is inside stays non-negative
        blk = buf_base(x - (g.pad * g.W + g.pad));
    }
    __device__ void begin(int kbeg, int t) {
        const int k = kbeg + (t % BK);
        w.start(k, g.OH, g.OW);
        const int b = k / (g.OH * g.OW);
        voff = (b * g.Cin * g.H * g.W + w.oh * g.stride * g.W + w.ow * g.stride + joff) * 4;    // relative to blk
        k_pos = kbeg;
    }
    __device__ void load(int k0, int kend, int t, Regs &rg) const {
        const int k = k0 + (t % BK);
        const int ohw = g.OH * g.OW;
        const int b = k / ohw, sp = k - b * ohw;
        const int oh = sp / g.OW, ow = sp - oh * g.OW;
        const int ih0 = oh * g.stride - g.pad, iw0 = ow * g.stride - g.pad;
        const int hw = g.H * g.W;
        const float *src = x + (size_t)b * g.Cin * hw + ih0 * g.W + iw0 + joff;
        const int safe = (int)(x - src);
        const int iw = iw0 + (jq & 3), ihq = ih0 + (jq >> 2);
        const bool okw = k < kend && iw >= 0 && iw < g.W;
        const bool ok0 = okw && ihq >= 0 && ihq < g.H;            // kh = jq>>2       (v even)
        const bool ok1 = okw && ihq + 2 >= 0 && ihq + 2 < g.H;    // kh = (jq>>2) + 2 (v odd)
        unsigned okbits = 0;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const bool ok = ((v & 1) ? ok1 : ok0) && v < nvalid;
            rg.v[v] = src[ok ? (v >> 1) * hw + 2 * (v & 1) * g.W : safe];
            okbits |= (ok ? 1u : 0u) << v;
        }
        rg.ok = okbits;
    }
    static constexpr bool PARTS = true, TAIL = false;
    int voff_a, voff_b;     // this k-step's offsets for the even / odd elements (tap rows jq>>2 and (jq>>2)+2), or BUF_OOB
    __device__ __forceinline__ void load_part(int k0, int, int, Regs &rg, int part, int nparts) {
        const int hw = g.H * g.W;
        if (part == 0) {
            const int ohw = g.OH * g.OW, s = g.stride;
            const bool adv = k0 != k_pos;             // block-uniform
            const int q32 = BK / ohw, r32 = BK - q32 * ohw;
            const int ra = r32 / g.OW, rc = r32 - ra * g.OW;            // r32 positions = ra rows + rc columns (uniform)
            w.ow += adv ? rc : 0; w.oh += adv ? ra : 0;
            const bool cw = w.ow >= g.OW;             // column wrap: next output row
            w.ow -= cw ? g.OW : 0; w.oh += cw ? 1 : 0;
            const bool iwrap = w.oh >= g.OH;          // image wrap
            w.oh -= iwrap ? g.OH : 0;
            voff += (adv ? (q32 * g.Cin * hw + ra * s * g.W + rc * s) * 4 : 0) + (cw ? (s * g.W - g.OW * s) * 4 : 0) +
                    (iwrap ? (g.Cin * hw - g.OH * s * g.W) * 4 : 0);
            k_pos = k0;
            const int iw = w.ow * s - g.pad + (jq & 3), ihq = w.oh * s - g.pad + (jq >> 2);
            const bool okw = iw >= 0 && iw < g.W;
            voff_a = (okw && ihq >= 0 && ihq < g.H) ? voff : BUF_OOB;
            voff_b = (okw && ihq + 2 >= 0 && ihq + 2 < g.H) ? voff : BUF_OOB;
        }
        const i32x4_t rs = buf_rsrc(blk, 0);
#pragma unroll
        for (int v = 0; v < NV; ++v)
            if (MVAE_IN_PART(v, NV, part, nparts))
                rg.v[v] = llvm_raw_buffer_load_f32(rs, (v & 1) ? voff_b : voff_a, ((v >> 1) * hw + 2 * (v & 1) * g.W) * 4, 0);
    }
    static constexpr bool RMAJOR = false;
    static constexpr int ROWS = BK, PITCH = TILE + 1;       // odd pitch: the lanes of a store run along k (one row each)
    typedef float (*Tile)[PITCH];
    static __device__ __forceinline__ float4 frag(Tile L, int k0, int row) { return frag_kmajor(L, k0, row); }
    __device__ void store(Tile L, int t, const Regs &rg) const {
        const int kl = t % BK, jb = t / BK;
#pragma unroll
        for (int v = 0; v < NV; ++v) L[kl][jb + v * JSTEP] = rg.v[v] * mask0((rg.ok >> v) & 1u);
    }
    __device__ __forceinline__ void store_part(Tile L, int t, const Regs &rg, int part, int nparts) const {
        const int kl = t % BK, jb = t / BK;
#pragma unroll
        for (int v = 0; v < NV; ++v)
            if (MVAE_IN_PART(v, NV, part, nparts)) L[kl][jb + v * JSTEP] = rg.v[v];
    }
};

// wr[cls][(co,a,b)][ci] = w[co][ci][kh0 + s*a][kw0 + s*b]: the weights of one output parity class of
// a transposed conv, reduction index major / input channel contiguous, so the dgrad-form GEMM
// fetches them with coalesced float4 loads instead of a 64-byte-stride gather.
__global__ __launch_bounds__(256) void repack_dgrad_weights_kernel(const float *w, float *wr, int Cout, int Cin,
                                                                   int stride, int pad) {
    const int tlog = (stride == 2) ? 1 : 2, tpd = 1 << tlog;
    const int kc = Cout * tpd * tpd;                 // reduction length per class
    const int total = stride * stride * kc * Cin;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
        const int ci = idx % Cin;
        int rest = idx / Cin;
        const int k = rest % kc, cls = rest / kc;
        const int ph = cls / stride, pw = cls % stride;
        const int kh0 = (ph + pad) % stride, kw0 = (pw + pad) % stride;
        const int co = k >> (2 * tlog), a = (k >> tlog) & (tpd - 1), b = k & (tpd - 1);
        wr[idx] = w[((co * Cin + ci) * 4 + kh0 + stride * a) * 4 + kw0 + stride * b];
    }
}

inline ConvGeom make_geom(int B, int Cin, int H, int W, int Cout, int stride, int pad) {
    ConvGeom g;
    g.B = B; g.Cin = Cin; g.H = H; g.W = W; g.Cout = Cout; g.stride = stride; g.pad = pad;
    g.OH = (H + 2 * pad - 4) / stride + 1;
    g.OW = (W + 2 * pad - 4) / stride + 1;
    g.lg_ohw = (log2_or_neg(g.OH) >= 0 && log2_or_neg(g.OW) >= 0) ? log2_or_neg(g.OH * g.OW) : -1;
    g.lg_ow = g.lg_ohw >= 0 ? log2_or_neg(g.OW) : -1;
    const int H2 = H / stride, W2 = W / stride;
    g.lg_hw2 = (log2_or_neg(H2) >= 0 && log2_or_neg(W2) >= 0) ? log2_or_neg(H2 * W2) : -1;
    g.lg_w2 = g.lg_hw2 >= 0 ? log2_or_neg(W2) : -1;
    if (!MVAE_FAST_DIV) g.lg_ohw = g.lg_ow = g.lg_hw2 = g.lg_w2 = -1;
    return g;
}

inline bool conv_args_ok(int B, int Cin, int H, int W, int Cout, int stride, int pad) {
    if (B <= 0 || Cin <= 0 || Cout <= 0 || H < 4 - 2 * pad || W < 4 - 2 * pad) return false;
    if (!((stride == 2 && pad == 1) || (stride == 1 && pad == 0))) return false;
    if (stride == 2 && ((H & 1) || (W & 1))) return false;
    // int32 offsets inside the gathers
    if ((long)B * Cin * H * W >= (1L << 31) || (long)B * Cout * H * W >= (1L << 31)) return false;
    return true;
}

// ---- direct forward conv for <= 4 INPUT channels, stride 2, pad 1 (Conv2d(3,32) of CelebA and the data gradient
//      of ConvTranspose2d(32,3), celeba/model.py:77,126; Conv2d(1,64) / ConvTranspose2d(64,1) of FashionMNIST).
//      As a GEMM the reduction is 16 .. 48 long -- one or two k-tiles, the second partial -- and the launch is all
//      prologue (28 TFLOP/s).  It is a streaming VALU kernel instead: a thread owns two adjacent output columns
//      of a 32-channel group (64 accumulators); per (input channel, tap row) it loads the 6 input columns both
//      outputs touch (two float2 + two scalars, clamped addresses and 0/1 factors: no load under a branch) and
//      per tap reads the 32 weights of the group from LDS (transposed there once per block: [ci][tap][co], eight
//      broadcast ds_read_b128 per 64 FMAs).  Outputs leave as float2 per channel. ----
// CG: output channels of a block's group (32; 16 for launches that would leave the chip half empty: conv_fwd_small)
template <int CIN, int CG>
__global__ __launch_bounds__(256) void conv_small_fwd_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                             float *__restrict__ out, float *__restrict__ act,
                                                             const float *__restrict__ dpre, ConvGeom g, int total) {
    __shared__ __attribute__((aligned(16))) float wl[CIN * 16 * CG];
    const int cg = blockIdx.y * CG;
    for (int j = threadIdx.x; j < CIN * 16 * CG; j += 256) {        // coalesced read of the group's rows, transposing store
        const int col = j / (CIN * 16), rem = j - col * (CIN * 16);  // rem = ci * 16 + tap
        wl[rem * CG + col] = (cg + col < g.Cout) ? w[(size_t)cg * CIN * 16 + j] : 0.f;
    }
    __syncthreads();
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int OW2 = g.OW >> 1;
    const int ow0 = (idx % OW2) * 2, oh = (idx / OW2) % g.OH, n = idx / (OW2 * g.OH);
    float acc[2][CG];
#pragma unroll
    for (int c = 0; c < CG; ++c) { acc[0][c] = 0.f; acc[1][c] = 0.f; }
    const float lm = ow0 > 0 ? 1.f : 0.f, rm = ow0 + 2 < g.OW ? 1.f : 0.f;     // columns 2*ow0-1 and 2*ow0+4 inside?
    const int lo = ow0 > 0 ? -1 : 0, ro = ow0 + 2 < g.OW ? 4 : 3;              // clamped (always legal) offsets
    const float *img = x + (size_t)n * CIN * g.H * g.W + 2 * ow0;
    // One rolled loop over (input channel, tap row) -- unrolled, hipcc hoists all 48 taps' loads and weight reads
    // and spills -- with the NEXT row's six values in flight while this one is multiplied.
    float2 n0, n1; float nl, nr;
    auto fetch = [&](int it) {
        const int ci = it >> 2, ih = 2 * oh - 1 + (it & 3);
        const float *row = img + ((size_t)ci * g.H + min(max(ih, 0), g.H - 1)) * g.W;
        n0 = *reinterpret_cast<const float2 *>(row); n1 = *reinterpret_cast<const float2 *>(row + 2);
        nl = row[lo]; nr = row[ro];
    };
    fetch(0);
#pragma unroll 1
    for (int it = 0; it < CIN * 4; ++it) {
        const int ih = 2 * oh - 1 + (it & 3);
        const float rok = (ih >= 0 && ih < g.H) ? 1.f : 0.f;
        float v[6];
        v[0] = nl * (rok * lm); v[1] = n0.x * rok; v[2] = n0.y * rok; v[3] = n1.x * rok; v[4] = n1.y * rok;
        v[5] = nr * (rok * rm);
        fetch(min(it + 1, CIN * 4 - 1));              // the last trip re-reads its own row: no load under a branch
#pragma unroll
        for (int kw = 0; kw < 4; ++kw) {
            const float4 *wp = reinterpret_cast<const float4 *>(wl + (it * 4 + kw) * CG);     // it * 4 = ci * 16 + kh * 4
            const float x0 = v[kw], x1 = v[kw + 2];
#pragma unroll
            for (int q = 0; q < CG / 4; ++q) {
                const float4 ww = wp[q];
                acc[0][4 * q + 0] = fmaf(ww.x, x0, acc[0][4 * q + 0]); acc[1][4 * q + 0] = fmaf(ww.x, x1, acc[1][4 * q + 0]);
                acc[0][4 * q + 1] = fmaf(ww.y, x0, acc[0][4 * q + 1]); acc[1][4 * q + 1] = fmaf(ww.y, x1, acc[1][4 * q + 1]);
                acc[0][4 * q + 2] = fmaf(ww.z, x0, acc[0][4 * q + 2]); acc[1][4 * q + 2] = fmaf(ww.z, x1, acc[1][4 * q + 2]);
                acc[0][4 * q + 3] = fmaf(ww.w, x0, acc[0][4 * q + 3]); acc[1][4 * q + 3] = fmaf(ww.w, x1, acc[1][4 * q + 3]);
            }
        }
    }
    const int ohw = g.OH * g.OW;
    const size_t o0 = ((size_t)n * g.Cout + cg) * ohw + (size_t)oh * g.OW + ow0;
    if (MVAE_SMALL_EPI_BATCH && dpre && cg + CG <= g.Cout) {
        // the data-gradient use (ConvTranspose2d(64, 1) / (32, 3) backward, times the producer's Swish'): the producer's
        // pre-activations of EIGHT channels are fetched together, one batch ahead of the batch being finished.  In the
        // loop below every channel's load sits under its own block-uniform branches, hipcc waits for the whole memory
        // queue behind it -- the previous channel's stores included -- and a thread walked 32 dependent round trips:
        // 83 us inside the FashionMNIST step for 206 MB.
        float2 pa[8], pb[8];
        auto load8 = [&](float2 (&p)[8], int c0) {
#pragma unroll
            for (int k = 0; k < 8; ++k) p[k] = *reinterpret_cast<const float2 *>(dpre + o0 + (size_t)(c0 + k) * ohw);
        };
        auto finish8 = [&](const float2 (&p)[8], int c0) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const size_t o = o0 + (size_t)(c0 + k) * ohw;
                const float v0 = acc[0][c0 + k] * swish_grad_(p[k].x), v1 = acc[1][c0 + k] * swish_grad_(p[k].y);
                if (out) *reinterpret_cast<float2 *>(out + o) = make_float2(v0, v1);
                if (act) *reinterpret_cast<float2 *>(act + o) = make_float2(swishf_(v0), swishf_(v1));
            }
        };
        load8(pa, 0);
        load8(pb, 8);
        __builtin_amdgcn_sched_barrier(0);
        finish8(pa, 0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (CG == 32) {
            load8(pa, 16);
            __builtin_amdgcn_sched_barrier(0);
            finish8(pb, 8);
            __builtin_amdgcn_sched_barrier(0);
            load8(pb, 24);
            __builtin_amdgcn_sched_barrier(0);
            finish8(pa, 16);
            finish8(pb, 24);
        } else {
            finish8(pb, 8);
        }
        return;
    }
#pragma unroll
    for (int c = 0; c < CG; ++c) {
        if (cg + c < g.Cout) {                      // (no break: the accumulators must stay statically indexed)
            const size_t o = o0 + (size_t)c * ohw;
            float v0 = acc[0][c], v1 = acc[1][c];
            if (dpre) { const float2 p2 = *reinterpret_cast<const float2 *>(dpre + o); v0 *= swish_grad_(p2.x); v1 *= swish_grad_(p2.y); }
            if (out) *reinterpret_cast<float2 *>(out + o) = make_float2(v0, v1);
            if (act) *reinterpret_cast<float2 *>(act + o) = make_float2(swishf_(v0), swishf_(v1));
        }
    }
}

inline bool conv_fwd_small_ok(const ConvGeom &g, const float *x, const float *pre, const float *act, const float *dpre) {
    return MVAE_CONV_SMALL_FWD && g.Cin <= 4 && g.stride == 2 && g.pad == 1 && g.H == 2 * g.OH && g.W == 2 * g.OW &&
           g.OW % 2 == 0 && aligned8(x) && (!pre || aligned8(pre)) && (!act || aligned8(act)) && (!dpre || aligned8(dpre));
}

#ifndef MVAE_SMALL_FWD_CG16
#define MVAE_SMALL_FWD_CG16 1     // 16-channel groups for launches with < 1024 blocks of 32 (0: A/B builds)
#endif
inline int conv_fwd_small(const float *x, const float *w, float *pre, float *act, const float *dpre, ConvGeom g,
                          hipStream_t st) {
    const int total = g.B * g.OH * (g.OW / 2);
    const dim3 blk(256);
    dim3 grid((total + 255) / 256, (g.Cout + 31) / 32);
    // A thread carries 64 accumulators and walks CIN * 4 dependent trips of six loads: with two blocks per CU (Conv2d(3, 32) at
    // 256 images: 512 blocks) nothing covers a trip's latency -- 22.9 us re-issued hot, 31.5 us in the step, where x comes
    // from HBM.  Half the channels per thread = twice the blocks.
    const bool half = MVAE_SMALL_FWD_CG16 && g.Cout % 16 == 0 && (long)grid.x * grid.y < 1024;
    if (half) grid.y = g.Cout / 16;
#define MVAE_CSF(CV)                                                                                          \
    if (half) hipLaunchKernelGGL((conv_small_fwd_kernel<CV, 16>), grid, blk, 0, st, x, w, pre, act, dpre, g, total); \
    else hipLaunchKernelGGL((conv_small_fwd_kernel<CV, 32>), grid, blk, 0, st, x, w, pre, act, dpre, g, total);
    switch (g.Cin) {
        case 1: MVAE_CSF(1) break;
        case 2: MVAE_CSF(2) break;
        case 3: MVAE_CSF(3) break;
        default: MVAE_CSF(4) break;
    }
#undef MVAE_CSF
    return mvae_launch_status();
}

// ---- conv forward form: y[n][co][oh][ow] = sum_k w[co][k] * im2col(x)[k][(n,oh,ow)] ----
int conv_fwd_impl(const float *x, const float *w, float *pre, float *act, const float *dpre,
                  ConvGeom g, hipStream_t st) {
    const int I = g.Cout, J = g.B * g.OH * g.OW, K = g.Cin * 16;
    if (conv_fwd_small_ok(g, x, pre, act, dpre) && !MVAE_TUNE(wm)) return conv_fwd_small(x, w, pre, act, dpre, g, st);
    Plan pl = make_plan(I, J, K, false);
    // >= 128 output channels and enough columns for >= 384 blocks of 128 x 64: two accumulators per wave share
    // every gathered fragment (dec2 / dec1 dgrad at 512 images: 91 -> 99 and 77 -> 80 TFLOP/s; at 256 images the
    // grid would be one block per CU and 64 x 64 tiles win)
    if (pl.wm == 1 && pl.wn == 1 && pl.kw == 1 && pl.wgn == 2 && I >= 128 && !MVAE_TUNE(wm) && !MVAE_TUNE(wn) &&
        cdiv(I, 128) * cdiv(J, 64) >= MVAE_WIDE_BLOCKS)
        pl.wm = 2;
    if (K <= MVAE_MULTI_MAXK) pl.items = MVAE_MULTI_ITEMS;      // short reductions: pipeline across consecutive tiles
    pl.xcd = MVAE_CONV_XCD ? 3 : 0;                             // the bands of one column tile on one XCD (igemm_kernel)
    EpNCHW e;
    e.out = pre; e.act = act; e.dpre = dpre;
    e.C = g.Cout; e.HW = g.OH * g.OW; e.Wfull = g.OW; e.H2 = g.OH; e.W2 = g.OW;
    e.sy = 1; e.py = 0; e.px = 0; e.J = J; e.off = 0;
    e.lg_hw2 = g.lg_ohw; e.lg_w2 = g.lg_ow;
    if (MVAE_CONV_PATCH && aligned16(w) && aligned16(x) && MVAE_EP_BUFFER && !MVAE_TUNE(wm)) {
        // the input as an LDS patch, the weights as they lie in memory (conv_patch.h)
        const ConvPatchPlan pp = conv_patch_plan(g.B, g.Cin, g.H, g.W, g.Cout, g.OH, g.OW, g.stride, g.pad);
        if (pp.kind == 1 || pp.kind == 4) return launch_conv_patch<260>(pp, x, w, e, st);
        if (pp.kind == 2) return launch_conv_patch<324>(pp, x, w, e, st);
        if (pp.kind == 3) return launch_conv_patch<592>(pp, x, w, e, st);
    }
    auto mp = [&](auto &p) { p.src = w; p.ld = K; p.R = I; p.Klen = K; };
    auto mq = [&](auto &q) { q.x = x; q.g = g; q.Mtot = J; };
    if (aligned16(w) && (size_t)g.Cin * g.H * g.W * 4 * (256 / (g.OH * g.OW) + 2) < ((size_t)1 << 31)) {
        void *g2ws = nullptr; size_t g2ws_bytes = 0;
#ifdef MVAE_TUNING
        {   // experiments only: the forward entry points carry no scratch argument (the persistent modes' slabs need one)
            static void *tune_ws = nullptr;
            if (!tune_ws && getenv("MVAE_G2_FORCE")) (void)hipMalloc(&tune_ws, (size_t)256 << 20);
            g2ws = tune_ws; g2ws_bytes = tune_ws ? (size_t)256 << 20 : 0;
        }
#endif
        G2Plan g2 = g2_plan_for(I, J, K, 1, false, g2ws, g2ws_bytes, G2_CONV_FWD);
        if (g2.ok) return launch_gemm2<G2RowsK, G2Im2col, EpNCHW, false>(g2, mp, mq, e, st);
    }
    if (aligned16(w))
        return launch_igemm<LdRowsKC, LdIm2col, EpNCHW, false>(pl, mp, mq, e, I, J, K, make_sink(nullptr, I, J, false), st);
    return launch_igemm<LdRowsKSC, LdIm2col, EpNCHW, false>(pl, mp, mq, e, I, J, K, make_sink(nullptr, I, J, false), st);
}

// ---- direct transposed conv for <= 4 OUTPUT channels (ConvTranspose2d(32,3) / (64,1), stride 2,
//      pad 1: celeba/model.py:126, fashionmnist/model.py:114).  As a GEMM these have a 3-row output
//      tile (5 % MFMA utilisation); they are HBM/L1-bound streaming ops instead: one thread owns the
//      2x2 output quad (2a..2a+1, 2b..2b+1) of every channel, which depends on the 3x3 dy
//      neighbourhood (a-1..a+1, b-1..b+1) of each of the Cout input maps; weights sit in LDS. ----
template <int C>
__global__ __launch_bounds__(256) void convT_small_kernel(const float *dy, const float *w, float *out, float *act,
                                                          const float *dpre, ConvGeom g, int total) {
    extern __shared__ __attribute__((aligned(16))) float wl[];      // [Cout][C][4][4]
    for (int i = threadIdx.x; i < g.Cout * C * 16; i += 256) wl[i] = w[i];
    __syncthreads();
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int OH = g.OH, OW = g.OW;                  // dy is [B][Cout][OH][OW]; out [B][C][2*OH][2*OW]
    const int b = idx % OW, a = (idx / OW) % OH, n = idx / (OW * OH);
    float acc[C][4];
#pragma unroll
    for (int c = 0; c < C; ++c) { acc[c][0] = acc[c][1] = acc[c][2] = acc[c][3] = 0.f; }
    const bool rm = a > 0, rp = a + 1 < OH, cm = b > 0, cp = b + 1 < OW;
    const float *src = dy + ((size_t)n * g.Cout * OH + a) * OW + b;
    for (int co = 0; co < g.Cout; ++co, src += OH * OW) {
        float d[3][3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const bool ok = (r == 1 || (r == 0 ? rm : rp)) && (q == 1 || (q == 0 ? cm : cp));
                d[r][q] = ok ? src[(r - 1) * OW + (q - 1)] : 0.f;
            }
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float4 *wp = reinterpret_cast<const float4 *>(wl + (co * C + c) * 16);
            const float4 w0 = wp[0], w1 = wp[1], w2 = wp[2], w3 = wp[3];    // rows kh = 0..3, fields kw
            // (ph,pw) = (0,0): kh in {1,3} <-> rows a, a-1 ; kw in {1,3} <-> cols b, b-1
            acc[c][0] += w1.y * d[1][1] + w1.w * d[1][0] + w3.y * d[0][1] + w3.w * d[0][0];
            // (0,1): kw in {0,2} <-> cols b+1, b
            acc[c][1] += w1.x * d[1][2] + w1.z * d[1][1] + w3.x * d[0][2] + w3.z * d[0][1];
            // (1,0): kh in {0,2} <-> rows a+1, a
            acc[c][2] += w0.y * d[2][1] + w0.w * d[2][0] + w2.y * d[1][1] + w2.w * d[1][0];
            acc[c][3] += w0.x * d[2][2] + w0.z * d[2][1] + w2.x * d[1][2] + w2.z * d[1][1];
        }
    }
    const int H = 2 * OH, W = 2 * OW;
#pragma unroll
    for (int c = 0; c < C; ++c) {
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
            const size_t o = (((size_t)n * C + c) * H + 2 * a + ph) * W + 2 * b;
            float v0 = acc[c][ph * 2], v1 = acc[c][ph * 2 + 1];
            if (dpre) { v0 *= swish_grad_(dpre[o]); v1 *= swish_grad_(dpre[o + 1]); }
            if (out) *reinterpret_cast<float2 *>(out + o) = make_float2(v0, v1);
            if (act) *reinterpret_cast<float2 *>(act + o) = make_float2(swishf_(v0), swishf_(v1));
        }
    }
}

// The same, two horizontally adjacent positions per thread (even OW): the 3 x 4 neighbourhood is 9 loads
// (a float2 and two scalars per row) for 96 FMAs per channel pair instead of 18, the outputs leave as float4 rows,
// and the weights are read at block-uniform addresses straight from global memory -- scalar loads into SGPRs that
// the FMAs take as operands -- instead of twelve LDS broadcasts per input channel (the LDS pipe was as busy as
// the vector ALU: 28 TFLOP/s on ConvTranspose2d(32, 3), celeba/model.py:126).
// U: input channels fetched per trip of the channel loop.  With ONE (rounds 2-3, the next channel in flight behind the
// current) every trip waited out a full memory latency -- 32 FMAs per output channel do not cover it, and these launches
// have only 3-5 waves per SIMD to hide it behind (ConvTranspose2d(64, 1) at 2048 rows: 74 us for 103 MB, 1.5 TB/s, with
// the ALU, the texture path and HBM each good for ~20 us) -- so a trip fetches U channels at once (host: Cout % U == 0).
template <int C, int U>
__global__ __launch_bounds__(256) void convT_small2_kernel(const float *__restrict__ dy, const float *__restrict__ w,
                                                           float *__restrict__ out, float *__restrict__ act,
                                                           const float *__restrict__ dpre, ConvGeom g, int total) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int OH = g.OH, OW = g.OW, OW2 = OW >> 1;  // dy is [B][Cout][OH][OW]; out [B][C][2*OH][2*OW]
    const int b0 = (idx % OW2) * 2, a = (idx / OW2) % OH, n = idx / (OW2 * OH);
    float acc[C][2][4];                             // [channel][output row parity][4 output columns 2*b0 .. 2*b0+3]
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[c][q >> 2][q & 3] = 0.f;
    const int plane = OH * OW;
    // Buffer loads: ONE scalar base (the block's first image), nine per-thread byte offsets that stay put for the
    // whole loop, and the channel rides the scalar offset -- with 64-bit pointers the U-deep ring kept a pointer
    // pair per load per stage (200 VGPRs, two waves per SIMD).  A neighbour outside the image carries the offset
    // BUF_OOB and reads as 0: the loaded registers go into the FMAs as they arrived (a 0/1 multiply made copies, and
    // the compiler parked the copies' waits at the loop's back edge -- behind every load of the trip).
    const int n_first = (blockIdx.x * 256) / (OW2 * OH);
    const i32x4_t rs = buf_rsrc(buf_base(dy), (size_t)n_first * g.Cout * plane);
    int vo[3][3];                                   // [row a-1, a, a+1][column b0-1 | b0, b0+1 | b0+2]
    {
        const int at = ((n - n_first) * g.Cout * plane + a * OW + b0) * 4;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const bool rok = r == 0 ? a > 0 : (r == 2 ? a + 1 < OH : true);
            const int ro = (r - 1) * OW;
            vo[r][0] = rok && b0 > 0 ? at + (ro - 1) * 4 : BUF_OOB;
            vo[r][1] = rok ? at + ro * 4 : BUF_OOB;
            vo[r][2] = rok && b0 + 2 < OW ? at + (ro + 2) * 4 : BUF_OOB;
        }
    }
    // U channels' neighbourhoods are fetched together at the top of a trip and multiplied as they land
    float nx[U][3][4];
    auto fetch = [&](int u, int ch) {
        const int so = ch * plane * 4;              // block-uniform
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const f32x2_t mid = llvm_raw_buffer_load_f32x2(rs, vo[r][1], so, 0);
            nx[u][r][0] = llvm_raw_buffer_load_f32(rs, vo[r][0], so, 0); nx[u][r][1] = mid.x; nx[u][r][2] = mid.y;
            nx[u][r][3] = llvm_raw_buffer_load_f32(rs, vo[r][2], so, 0);
        }
    };
    // (A ring -- stage u refilled for trip + 1 right after it is consumed -- does not survive hipcc 7.2: the loaded
    //  values are loop-carried, the compiler copies them into the registers the next trip reads at the BACK EDGE, and
    //  the copies wait for all but the last two loads of the trip.  Within one trip nothing is carried.)
#pragma unroll 1
    for (int co0 = 0; co0 < g.Cout; co0 += U) {
#pragma unroll
      for (int u = 0; u < U; ++u) fetch(u, co0 + u);
      __builtin_amdgcn_sched_barrier(0);            // all 9 U loads are issued before the first is waited for
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int co = co0 + u;
        const float *wc = w + (size_t)co * C * 16;  // block-uniform
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float *wp = wc + c * 16;          // [kh][kw]
#pragma unroll
            for (int pos = 0; pos < 2; ++pos) {     // input column b0 + pos -> output columns 2*(b0+pos), +1
                const float (*e)[4] = nx[u];        // rows a-1, a, a+1; columns b0-1 .. b0+2
                const int q = pos;                  // e[.][q] = column b-1, e[.][q+1] = b, e[.][q+2] = b+1
                // chained FMAs into the accumulator (a sum of four products added afterwards costs a fifth instruction)
                auto mac4 = [](float acc0, float w0, float x0, float w1, float x1, float w2, float x2, float w3, float x3) {
                    return fmaf(w3, x3, fmaf(w2, x2, fmaf(w1, x1, fmaf(w0, x0, acc0))));
                };
                // output (row parity 0, col parity 0): kh in {1,3} <-> rows a, a-1 ; kw in {1,3} <-> cols b, b-1
                acc[c][0][2 * pos] = mac4(acc[c][0][2 * pos], wp[5], e[1][q + 1], wp[7], e[1][q], wp[13], e[0][q + 1], wp[15], e[0][q]);
                // (0,1): kw in {0,2} <-> cols b+1, b
                acc[c][0][2 * pos + 1] = mac4(acc[c][0][2 * pos + 1], wp[4], e[1][q + 2], wp[6], e[1][q + 1], wp[12], e[0][q + 2], wp[14], e[0][q + 1]);
                // (1,0): kh in {0,2} <-> rows a+1, a
                acc[c][1][2 * pos] = mac4(acc[c][1][2 * pos], wp[1], e[2][q + 1], wp[3], e[2][q], wp[9], e[1][q + 1], wp[11], e[1][q]);
                acc[c][1][2 * pos + 1] = mac4(acc[c][1][2 * pos + 1], wp[0], e[2][q + 2], wp[2], e[2][q + 1], wp[8], e[1][q + 2], wp[10], e[1][q + 1]);
            }
        }
      }
    }
    const int H = 2 * OH, W = 2 * OW;
#pragma unroll
    for (int c = 0; c < C; ++c) {
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
            const size_t o = (((size_t)n * C + c) * H + 2 * a + ph) * W + 2 * b0;     // 16-byte aligned: b0 even, W % 4 == 0
            float4 v = make_float4(acc[c][ph][0], acc[c][ph][1], acc[c][ph][2], acc[c][ph][3]);
            if (dpre) {
                const float4 p4 = *reinterpret_cast<const float4 *>(dpre + o);
                v.x *= swish_grad_(p4.x); v.y *= swish_grad_(p4.y); v.z *= swish_grad_(p4.z); v.w *= swish_grad_(p4.w);
            }
            if (out) *reinterpret_cast<float4 *>(out + o) = v;
            if (act) *reinterpret_cast<float4 *>(act + o) = make_float4(swishf_(v.x), swishf_(v.y), swishf_(v.z), swishf_(v.w));
        }
    }
}

// ---- the same through LDS (round 4).  SQ / cache counters of convT_small2_kernel (profiles/r04_small_conv_counters.txt):
//      the texture addresser is the busy unit (57 % of the launch) -- the nine loads of a neighbourhood are 8- and 4-byte
//      accesses at a lane stride of 8 bytes, which the addresser takes a quad of lanes at a time (24 cache accesses per
//      load instruction: ~190 addresser cycles per channel per wave, 63 us of a 67-us launch on ConvTranspose2d(64, 1) at
//      2048 rows), and every input element is fetched six times.  Here a block owns NI whole images (or a band of R rows
//      of one) and stages each input channel's rows ONCE, with consecutive lanes on consecutive floats (4 addresser cycles
//      per load), into a zero-bordered LDS image [NI][R + 2][OW + 2]; a thread then reads its 3 x 4 neighbourhood as six
//      8-byte LDS reads with no bounds logic (the border and the rows outside the image stay zero from the fill at block
//      start: an element outside carries BUF_OOB, loads as 0 and is stored as 0).  D channels are staged per trip into one
//      half of a double buffer while the other half is multiplied: the staging registers live inside a trip, nothing
//      loaded is carried around the loop (see convT_small2_kernel's note on what hipcc does to that).
struct Small3Geo {
    int R, NI, bands;             // rows of a block's band, images per block, bands per image
    int E;                        // staged floats per channel per block: NI * (R + 2) * OW
    int ch_stride;                // LDS floats per staged channel: NI * (R + 2) * (OW + 2)
};
constexpr int S3_NE = 4;          // staged floats per thread per channel, at most

template <int C, int D>
__global__ __launch_bounds__(256) void convT_small3_kernel(const float *__restrict__ dy, const float *__restrict__ w,
                                                           float *__restrict__ out, float *__restrict__ act,
                                                           const float *__restrict__ dpre, ConvGeom g, Small3Geo sg) {
    extern __shared__ __attribute__((aligned(16))) float s3_lds[];      // [2][D][ch_stride] + a dummy float per thread
    const int t = threadIdx.x;
    const int OH = g.OH, OW = g.OW, OW2 = OW >> 1, PITCH = OW + 2, plane = OH * OW;
    const int n0 = blockIdx.x * sg.NI, a0 = blockIdx.y * sg.R;
    const int half = D * sg.ch_stride, dummy = 2 * half + t;
    for (int j = t; j < 2 * half; j += 256) s3_lds[j] = 0.f;
    // this thread's staged elements: (image, band row -1 .. R, column) -> global byte offset from the block's first image
    // (channel 0) and LDS offset inside a channel image; both fixed for the whole launch
    const i32x4_t rs = buf_rsrc(buf_base(dy), (size_t)n0 * g.Cout * plane);
    int gofs[S3_NE], lofs[S3_NE];
    const int per_img = (sg.R + 2) * OW;
#pragma unroll
    for (int k = 0; k < S3_NE; ++k) {
        const int e = t + k * 256;
        const int i = e / per_img, rem = e - i * per_img, rr = rem / OW, cc = rem - rr * OW;
        const int row = a0 - 1 + rr;
        const bool ok = e < sg.E && n0 + i < g.B && row >= 0 && row < OH;
        gofs[k] = ok ? ((i * g.Cout * OH + row) * OW + cc) * 4 : BUF_OOB;
        lofs[k] = e < sg.E ? (i * (sg.R + 2) + rr) * PITCH + cc + 1 : -1;
    }
    // (all S3_NE slots are always loaded and stored -- an unused one carries BUF_OOB and lands in the thread's dummy
    //  float: a block-uniform `if (k < ne)` around a load made hipcc drain the memory queue after every one of them)
    // this thread's outputs: image i, band row al, input columns b0, b0 + 1
    const int per_band = sg.R * OW2;
    const int ti = t / per_band, trem = t - ti * per_band, al = trem / OW2, b0 = (trem - al * OW2) * 2;
    const int n = n0 + ti, a = a0 + al;
    const bool live = ti < sg.NI && n < g.B && a < OH;
    const int rd = live ? (ti * (sg.R + 2) + al) * PITCH + b0 : 0;      // neighbourhood corner (row a - 1, column b0 - 1)
    float acc[C][2][4];
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[c][q >> 2][q & 3] = 0.f;
    float st[D][S3_NE];
    auto load_batch = [&](int co0) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int so = (co0 + d) * plane * 4;   // block-uniform
#pragma unroll
            for (int k = 0; k < S3_NE; ++k) st[d][k] = llvm_raw_buffer_load_f32(rs, gofs[k], so, 0);
        }
    };
    auto store_batch = [&](float *buf) {
#pragma unroll
        for (int d = 0; d < D; ++d)
#pragma unroll
            for (int k = 0; k < S3_NE; ++k) {
                const int o = lofs[k] < 0 ? dummy : (int)(buf - s3_lds) + d * sg.ch_stride + lofs[k];
                s3_lds[o] = st[d][k];
            }
    };
    const int nb = g.Cout / D;                      // host: Cout % D == 0
    load_batch(0);
    __syncthreads();                                // the zero fill is complete
    store_batch(s3_lds);
    __syncthreads();
#pragma unroll 1
    for (int kb = 0; kb < nb; ++kb) {
        const float *cur = s3_lds + (kb & 1) * half;
        // the next batch's rows are in flight while this one is multiplied (the last trip re-reads its own batch into
        // the buffer nobody reads any more: no load under a branch)
        load_batch(min(kb + 1, nb - 1) * D);
        __builtin_amdgcn_sched_barrier(0);
        // (rolled, two channels per trip: unrolled D deep the compiler issues all 6 D LDS reads first -- 200+ registers)
#pragma unroll 2
        for (int d = 0; d < D; ++d) {
            const float *img = cur + d * sg.ch_stride + rd;
            float e[3][4];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const float2 lo = *reinterpret_cast<const float2 *>(img + r * PITCH);
                const float2 hi = *reinterpret_cast<const float2 *>(img + r * PITCH + 2);
                e[r][0] = lo.x; e[r][1] = lo.y; e[r][2] = hi.x; e[r][3] = hi.y;
            }
            const float *wc = w + (size_t)(kb * D + d) * C * 16;      // block-uniform
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float *wp = wc + c * 16;      // [kh][kw]
#pragma unroll
                for (int pos = 0; pos < 2; ++pos) { // input column b0 + pos -> output columns 2*(b0+pos), +1
                    const int q = pos;              // e[.][q] = column b-1, e[.][q+1] = b, e[.][q+2] = b+1
                    auto mac4 = [](float acc0, float w0, float x0, float w1, float x1, float w2, float x2, float w3, float x3) {
                        return fmaf(w3, x3, fmaf(w2, x2, fmaf(w1, x1, fmaf(w0, x0, acc0))));
                    };
                    // the tap <-> neighbour table of convT_small2_kernel
                    acc[c][0][2 * pos] = mac4(acc[c][0][2 * pos], wp[5], e[1][q + 1], wp[7], e[1][q], wp[13], e[0][q + 1], wp[15], e[0][q]);
                    acc[c][0][2 * pos + 1] = mac4(acc[c][0][2 * pos + 1], wp[4], e[1][q + 2], wp[6], e[1][q + 1], wp[12], e[0][q + 2], wp[14], e[0][q + 1]);
                    acc[c][1][2 * pos] = mac4(acc[c][1][2 * pos], wp[1], e[2][q + 1], wp[3], e[2][q], wp[9], e[1][q + 1], wp[11], e[1][q]);
                    acc[c][1][2 * pos + 1] = mac4(acc[c][1][2 * pos + 1], wp[0], e[2][q + 2], wp[2], e[2][q + 1], wp[8], e[1][q + 2], wp[10], e[1][q + 1]);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        store_batch(s3_lds + ((kb + 1) & 1) * half);
        __syncthreads();
    }
    if (!live) return;
    const int H = 2 * OH, W = 2 * OW;
#pragma unroll
    for (int c = 0; c < C; ++c) {
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
            const size_t o = (((size_t)n * C + c) * H + 2 * a + ph) * W + 2 * b0;     // 16-byte aligned: b0 even, W % 4 == 0
            float4 v = make_float4(acc[c][ph][0], acc[c][ph][1], acc[c][ph][2], acc[c][ph][3]);
            if (dpre) {
                const float4 p4 = *reinterpret_cast<const float4 *>(dpre + o);
                v.x *= swish_grad_(p4.x); v.y *= swish_grad_(p4.y); v.z *= swish_grad_(p4.z); v.w *= swish_grad_(p4.w);
            }
            if (out) *reinterpret_cast<float4 *>(out + o) = v;
            if (act) *reinterpret_cast<float4 *>(act + o) = make_float4(swishf_(v.x), swishf_(v.y), swishf_(v.z), swishf_(v.w));
        }
    }
}

#ifndef MVAE_SMALL3_DMA
#define MVAE_SMALL3_DMA 1         // convT_small3's staging by LDS-DMA into a 3-deep ring (0: registers + ds_write, A/B builds)
#endif
// The same kernel with the zero-bordered channel images filled by LDS-DMA (gemm2.h's machinery): a channel image is NQ pieces of 64
// consecutive LDS floats, a piece = one `buffer_load_dword ... lds` whose lane l fetches the input element that belongs at float
// 64 q + l of the image -- or nothing: a border, a row outside the map, an image past the batch or the tail of the padded image is
// an out-of-range source, and the hardware writes the zero (no fill loop, no dummy slot logic).  Wave w owns pieces w, w + 4, ...
// (QPW of them: their source offsets are lane constants, the channel rides the scalar offset).  With one output channel the
// register form spent three quarters of its instructions on staging (32 loads + 32 ds_write + 96 address / select operations per
// trip against 64 packed multiply-adds); here a trip issues QPW x D pieces per wave and nothing else.  Three stages of D channels.
template <int C, int D, int QPW>
__global__ __launch_bounds__(256) void convT_small3d_kernel(const float *__restrict__ dy, const float *__restrict__ w,
                                                            float *__restrict__ out, float *__restrict__ act,
                                                            const float *__restrict__ dpre, ConvGeom g, Small3Geo sg) {
    extern __shared__ __attribute__((aligned(16))) float s3_lds[];      // [3][D][chs] + 4 x 64 floats nobody reads
    constexpr int ST = 3, QW = QPW * D;
    static_assert(2 * QW <= 63, "vmcnt");
    const int t = threadIdx.x, lane = t & 63;
    const int wv = g2_uni(t >> 6);
    const int OH = g.OH, OW = g.OW, OW2 = OW >> 1, PITCH = OW + 2, plane = OH * OW;
    const int n0 = blockIdx.x * sg.NI, a0 = blockIdx.y * sg.R;
    const int nq = (sg.ch_stride + 63) >> 6, chs = nq * 64, stage_fl = D * chs;
    const unsigned lds0 = (unsigned)(unsigned long)(g2_lds_void *)s3_lds;
    const unsigned sink = lds0 + (unsigned)((ST * stage_fl + wv * 64) * 4);
    int vo[QPW];
    const int per_img_l = (sg.R + 2) * PITCH;
#pragma unroll
    for (int u = 0; u < QPW; ++u) {
        const int p = (wv + 4 * u) * 64 + lane;
        const int i = p / per_img_l, rem = p - i * per_img_l, rr = rem / PITCH, cc = rem - rr * PITCH - 1;
        const int row = a0 - 1 + rr;
        const bool ok = p < sg.ch_stride && cc >= 0 && cc < OW && row >= 0 && row < OH && n0 + i < g.B;
        vo[u] = ok ? ((i * g.Cout * OH + row) * OW + cc) * 4 : BUF_OOB;
    }
    const BufBase bb = buf_base(dy + (size_t)n0 * g.Cout * plane);
    auto issue = [&](int kb, int stage) {
        const i32x4_t rs = g2_rsrc(bb, 0, 0x7fffffff);
        asm volatile("s_nop 4" ::: "memory");
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int so = g2_uni((kb * D + d) * plane * 4);
#pragma unroll
            for (int u = 0; u < QPW; ++u) {
                const int q = wv + 4 * u;                       // wave-uniform
                const bool real = q < nq;
                g2_dma4(rs, real ? vo[u] : BUF_OOB, so, g2_uni(real ? lds0 + (unsigned)((stage * stage_fl + d * chs + q * 64) * 4) : sink));
            }
        }
    };
    const int per_band = sg.R * OW2;
    const int ti = t / per_band, trem = t - ti * per_band, al = trem / OW2, b0 = (trem - al * OW2) * 2;
    const int n = n0 + ti, a = a0 + al;
    const bool live = ti < sg.NI && n < g.B && a < OH;
    const int rd = live ? (ti * (sg.R + 2) + al) * PITCH + b0 : 0;      // neighbourhood corner (row a - 1, column b0 - 1)
    float acc[C][2][4];
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[c][q >> 2][q & 3] = 0.f;
    const int nb = g.Cout / D;                      // host: Cout % D == 0
    issue(0, 0);
    if (nb > 1) issue(1, 1);
    int st_c = 0, st_i = 2;
#pragma unroll 1
    for (int kb = 0; kb < nb; ++kb) {
        if (kb + 1 < nb) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(QW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        if (kb + 2 < nb) issue(kb + 2, st_i);
        st_i = st_i == ST - 1 ? 0 : st_i + 1;
        const float *cur = s3_lds + st_c * stage_fl;
        st_c = st_c == ST - 1 ? 0 : st_c + 1;
#pragma unroll 2
        for (int d = 0; d < D; ++d) {
            const float *img = cur + d * chs + rd;
            float e[3][4];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const float2 lo = *reinterpret_cast<const float2 *>(img + r * PITCH);
                const float2 hi = *reinterpret_cast<const float2 *>(img + r * PITCH + 2);
                e[r][0] = lo.x; e[r][1] = lo.y; e[r][2] = hi.x; e[r][3] = hi.y;
            }
            const float *wc = w + (size_t)(kb * D + d) * C * 16;      // block-uniform
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float *wp = wc + c * 16;      // [kh][kw]
#pragma unroll
                for (int pos = 0; pos < 2; ++pos) {
                    const int q = pos;
                    auto mac4 = [](float acc0, float w0, float x0, float w1, float x1, float w2, float x2, float w3, float x3) {
                        return fmaf(w3, x3, fmaf(w2, x2, fmaf(w1, x1, fmaf(w0, x0, acc0))));
                    };
                    // the tap <-> neighbour table of convT_small3_kernel: the same products in the same order
                    acc[c][0][2 * pos] = mac4(acc[c][0][2 * pos], wp[5], e[1][q + 1], wp[7], e[1][q], wp[13], e[0][q + 1], wp[15], e[0][q]);
                    acc[c][0][2 * pos + 1] = mac4(acc[c][0][2 * pos + 1], wp[4], e[1][q + 2], wp[6], e[1][q + 1], wp[12], e[0][q + 2], wp[14], e[0][q + 1]);
                    acc[c][1][2 * pos] = mac4(acc[c][1][2 * pos], wp[1], e[2][q + 1], wp[3], e[2][q], wp[9], e[1][q + 1], wp[11], e[1][q]);
                    acc[c][1][2 * pos + 1] = mac4(acc[c][1][2 * pos + 1], wp[0], e[2][q + 2], wp[2], e[2][q + 1], wp[8], e[1][q + 2], wp[10], e[1][q + 1]);
                }
            }
        }
    }
    if (!live) return;
    const int H = 2 * OH, W = 2 * OW;
#pragma unroll
    for (int c = 0; c < C; ++c) {
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
            const size_t o = (((size_t)n * C + c) * H + 2 * a + ph) * W + 2 * b0;     // 16-byte aligned: b0 even, W % 4 == 0
            float4 v = make_float4(acc[c][ph][0], acc[c][ph][1], acc[c][ph][2], acc[c][ph][3]);
            if (dpre) {
                const float4 p4 = *reinterpret_cast<const float4 *>(dpre + o);
                v.x *= swish_grad_(p4.x); v.y *= swish_grad_(p4.y); v.z *= swish_grad_(p4.z); v.w *= swish_grad_(p4.w);
            }
            if (out) *reinterpret_cast<float4 *>(out + o) = v;
            if (act) *reinterpret_cast<float4 *>(act + o) = make_float4(swishf_(v.x), swishf_(v.y), swishf_(v.z), swishf_(v.w));
        }
    }
}

// block geometry and batch depth of convT_small3_kernel; false: the shape stays with convT_small2_kernel
inline bool conv_small3_plan(const ConvGeom &g, Small3Geo &sg, int &depth, size_t &lds) {
    if (!MVAE_CONVT_SMALL3 || (g.OW & 1) || g.OW / 2 > 256) return false;
    const int OW2 = g.OW / 2;
    sg.R = g.OH < 256 / OW2 ? g.OH : 256 / OW2;
    sg.NI = 256 / (sg.R * OW2);
    if (sg.NI < 1) return false;
    if (sg.R < g.OH) sg.NI = 1;                     // bands only of single images
    sg.bands = (g.OH + sg.R - 1) / sg.R;
    sg.E = sg.NI * (sg.R + 2) * g.OW;
    sg.ch_stride = sg.NI * (sg.R + 2) * (g.OW + 2);
    if (sg.E > S3_NE * 256) return false;
    // a block walks its channels in a few barrier-separated trips: it needs company on its CU (>= 4 blocks) to hide them.
    // Measured hot (profiles/r04_small_conv_ab.txt): 1024 blocks 71 -> 48 us (ConvTranspose2d(64, 1), 2048 rows) and
    // 42 -> 34 us (ConvTranspose2d(32, 3), 512 rows); 512 blocks 20.5 -> 28 us (the same at 256 rows)
    if ((long)((g.B + sg.NI - 1) / sg.NI) * sg.bands < 1024) return false;
    // byte offsets inside a block's images stay far below 2 GiB
    if ((size_t)sg.NI * g.Cout * g.OH * g.OW * 4 >= ((size_t)1 << 30)) return false;
    for (depth = 8; depth >= 1; depth >>= 1) {
        lds = ((size_t)2 * depth * sg.ch_stride + 256) * sizeof(float);
        if (g.Cout % depth == 0 && lds <= 48 * 1024) return true;
    }
    return false;
}

inline bool conv_dgrad_small_ok(const ConvGeom &g) {
    return g.stride == 2 && g.pad == 1 && g.Cin <= 4 && g.H == 2 * g.OH && g.W == 2 * g.OW &&
           (size_t)g.Cout * g.Cin * 16 * sizeof(float) <= 48 * 1024;
}

inline int conv_dgrad_small(const float *dy, const float *w, float *dx, float *act, const float *dpre,
                            ConvGeom g, hipStream_t st) {
    const int total = g.B * g.OH * g.OW;
    const size_t lds = (size_t)g.Cout * g.Cin * 16 * sizeof(float);
    const dim3 grid((total + 255) / 256), blk(256);
    {
        Small3Geo sg; int depth; size_t lds3;
        if (conv_small3_plan(g, sg, depth, lds3) && aligned16(dx ? dx : act) && (!dx || !act || aligned16(act)) &&
            (!dpre || aligned16(dpre)) && (2 * g.OW) % 4 == 0) {
            const dim3 grid3((g.B + sg.NI - 1) / sg.NI, sg.bands);
            {
                // the DMA form: pieces of 64 floats, at most 12 per channel image (three per wave), four channels per stage
                const int nq = (sg.ch_stride + 63) / 64, qpw = (nq + 3) / 4;
                const size_t ldsd = ((size_t)3 * 4 * nq * 64 + 256) * sizeof(float);
                if (MVAE_SMALL3_DMA && g.Cout % 4 == 0 && qpw >= 2 && qpw <= 3 && ldsd <= 48 * 1024) {
#define MVAE_S3DMA(CV)                                                                                                   \
                    if (qpw == 2) hipLaunchKernelGGL((convT_small3d_kernel<CV, 4, 2>), grid3, blk, ldsd, st, dy, w, dx, act, dpre, g, sg); \
                    else hipLaunchKernelGGL((convT_small3d_kernel<CV, 4, 3>), grid3, blk, ldsd, st, dy, w, dx, act, dpre, g, sg);
                    switch (g.Cin) {
                        case 1: MVAE_S3DMA(1) break;
                        case 2: MVAE_S3DMA(2) break;
                        case 3: MVAE_S3DMA(3) break;
                        default: MVAE_S3DMA(4) break;
                    }
#undef MVAE_S3DMA
                    return mvae_launch_status();
                }
            }
#define MVAE_S3(CV, DV) hipLaunchKernelGGL((convT_small3_kernel<CV, DV>), grid3, blk, lds3, st, dy, w, dx, act, dpre, g, sg)
#define MVAE_S3D(CV)                                                                                     \
            if (depth == 8) MVAE_S3(CV, 8); else if (depth == 4) MVAE_S3(CV, 4);                         \
            else if (depth == 2) MVAE_S3(CV, 2); else MVAE_S3(CV, 1);
            switch (g.Cin) {
                case 1: MVAE_S3D(1) break;
                case 2: MVAE_S3D(2) break;
                case 3: MVAE_S3D(3) break;
                default: MVAE_S3D(4) break;
            }
#undef MVAE_S3D
#undef MVAE_S3
            return mvae_launch_status();
        }
    }
    if (MVAE_CONVT_SMALL2 && g.OW % 2 == 0 && aligned16(dx ? dx : act) && (!dx || !act || aligned16(act)) && (!dpre || aligned16(dpre)) &&
        aligned8(dy)) {
        const int total2 = total / 2;
        const dim3 grid2((total2 + 255) / 256);
        // channels fetched per trip: MVAE_SMALL2_DEPTH, as far as it divides the channel count (8 only with one output
        // channel: 96 registers of neighbourhoods)
        const int want = (MVAE_SMALL2_DEPTH >= 8 && g.Cin > 1) ? 4 : MVAE_SMALL2_DEPTH;
        const int depth = (want >= 8 && g.Cout % 8 == 0) ? 8 : ((want >= 4 && g.Cout % 4 == 0) ? 4 : ((want >= 2 && g.Cout % 2 == 0) ? 2 : 1));
#define MVAE_S2(CV)                                                                                                         \
        if (depth == 4) hipLaunchKernelGGL((convT_small2_kernel<CV, 4>), grid2, blk, 0, st, dy, w, dx, act, dpre, g, total2);   \
        else if (depth == 2) hipLaunchKernelGGL((convT_small2_kernel<CV, 2>), grid2, blk, 0, st, dy, w, dx, act, dpre, g, total2); \
        else hipLaunchKernelGGL((convT_small2_kernel<CV, 1>), grid2, blk, 0, st, dy, w, dx, act, dpre, g, total2);
        if (depth == 8) {
            hipLaunchKernelGGL((convT_small2_kernel<1, 8>), grid2, blk, 0, st, dy, w, dx, act, dpre, g, total2);
            return mvae_launch_status();
        }
        switch (g.Cin) {
            case 1: MVAE_S2(1) break;
            case 2: MVAE_S2(2) break;
            case 3: MVAE_S2(3) break;
            default: MVAE_S2(4) break;
        }
#undef MVAE_S2
        return mvae_launch_status();
    }
    switch (g.Cin) {
        case 1: hipLaunchKernelGGL(convT_small_kernel<1>, grid, blk, lds, st, dy, w, dx, act, dpre, g, total); break;
        case 2: hipLaunchKernelGGL(convT_small_kernel<2>, grid, blk, lds, st, dy, w, dx, act, dpre, g, total); break;
        case 3: hipLaunchKernelGGL(convT_small_kernel<3>, grid, blk, lds, st, dy, w, dx, act, dpre, g, total); break;
        default: hipLaunchKernelGGL(convT_small_kernel<4>, grid, blk, lds, st, dy, w, dx, act, dpre, g, total); break;
    }
    return mvae_launch_status();
}

// ---- stride-1 transposed conv as a DENSE GEMM + col2im through LDS (ConvTranspose2d(256,128,4,1,0)
//      5x5 -> 8x8 and the dgrad of Conv2d(128,256,4,1,0): celeba/model.py:85,117).  In the gather
//      form only 39 % of the (output pixel, tap) pairs are inside the 5x5 input, so 61 % of the MFMA
//      work multiplies zeros.  Here the GEMM is  col[(n,oh,ow)][(ci,kh,kw)] = sum_co dy[n,co,oh,ow] *
//      w[co,ci,kh,kw]  -- every product is real -- and the scatter-add  dx[n,ci,oh+kh,ow+kw] += col  happens
//      inside the block, which owns whole images: NI = 128 / (OH*OW) images (5 for the 5x5 maps: 125 of the
//      128 tile rows are real; the first version padded each image to a 32-row MFMA tile, 78 %) x 4 input
//      channels x 16 taps.  4 waves, each 64 rows x 32 columns = two accumulators sharing the weight
//      fragments; k loop over Cout; the finished 128 x 64 tile is parked in LDS and every thread gathers the
//      <= 16 taps of its output pixels. ----
constexpr int S1_ROWS = 128, S1_COLS = 64;
#ifndef MVAE_S1_BK
#define MVAE_S1_BK 16           // k-tile depth of convT_s1_kernel: 16 = 33 KB of LDS (the col2im tile), FOUR blocks per CU; 32 = 51 KB, three (round 3).  More co-resident blocks hide the cold prologue + col2im epilogue of each: CelebA-19 6.67 -> 6.55 ms, CelebA 2.453 -> 2.425 (profiles/r04_s1bk_wgt_ab.txt)
#endif
constexpr int S1_BK = MVAE_S1_BK;
#ifndef MVAE_S1_TAPS
#define MVAE_S1_TAPS 1          // 8x8 outputs: the col2im tap table is computed once per thread, not once per image (0: A/B builds)
#endif
#ifndef MVAE_S1_EPI2
#define MVAE_S1_EPI2 1          // 5 x 5 -> 8 x 8: col2im with a zero slot, a row / column tap table and the image as an immediate (0: A/B builds)
#endif
#ifndef MVAE_S1_DMA
#define MVAE_S1_DMA 1           // operands by LDS-DMA into a 3-deep ring (dy rows as 4-byte pieces, weight rows as 16-byte pieces): no staging
#endif                          // registers, no ds_write, two k-tiles in flight behind the one being multiplied (0: register staging, A/B builds)
#ifndef MVAE_S1_WIDE_MIN
#define MVAE_S1_WIDE_MIN 6144   // blocks (of 128 columns) from which a launch takes the wide form of convT_s1_kernel; 0: never (A/B builds)
#endif
#ifndef MVAE_S1_KO
#define MVAE_S1_KO 0            // knock-out builds (tools/build_variants.sh; results are WRONG by construction): low 3 bits 1 = no global loads in
#endif                          // the main loop, 2 = + no LDS stores / barriers, 3 = + no fragment reads; bit 3 (8) = no col2im epilogue
// CW: 64-column blocks (4 channels x 16 taps) per block.  CW = 2 (DMA form only): 128 x 128 block tiles, 64 x 64 per wave -- every
// fragment read feeds two matrix instructions, a barrier every 32 of them, half the dy pieces per matrix instruction; the col2im
// runs once per column block through the same 33-KB tile.  48 KB of ring: three blocks per CU.  The host picks it for launches
// with enough blocks (conv_dgrad_s1).
template <int CW>
__global__ __launch_bounds__(256, 2) void convT_s1_kernel(const float *dy, const float *w, float *out, float *act,
                                                          const float *dpre, ConvGeom g, int NI) {
    static_assert(CW == 1 || (CW == 2 && MVAE_S1_DMA), "the wide form exists on the DMA ring only");
    constexpr int PP = S1_ROWS + LPAD, QP = S1_COLS + LPAD, TP = S1_COLS + 1;
    constexpr int P_FL = S1_BK * PP, Q_FL = S1_BK * QP;
    constexpr int COLS = S1_COLS * CW;
    // (col2im in two 32-column passes -- 17 KB of staging, six blocks per CU -- measured neutral against four: not kept)
    constexpr int S1_ST = 3, ST_FL = S1_BK * (S1_ROWS + COLS);          // DMA ring: stages, floats per stage (P [BK][128], then Q [BK][COLS])
    constexpr int RING_FL = MVAE_S1_DMA ? S1_ST * ST_FL : 2 * P_FL + 2 * Q_FL;
    __shared__ __attribute__((aligned(16))) float s1_lds[RING_FL > S1_ROWS * TP ? RING_FL : S1_ROWS * TP];
    auto Ps = [&](int b2) { return reinterpret_cast<float (*)[PP]>(s1_lds + b2 * P_FL); };
    auto Qs = [&](int b2) { return reinterpret_cast<float (*)[QP]>(s1_lds + 2 * P_FL + b2 * Q_FL); };
    (void)Ps; (void)Qs;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wi = wave >> 1, wj = wave & 1;
    const int P = g.OH * g.OW;                      // positions per image (<= 32)
    // XCD-local image groups.  Workgroups go to the 8 XCDs round-robin in launch order, each XCD has its own L2, and
    // the Cin / 4 channel-group blocks of one image group all read the same dy tile (NI * Cout * P floats: 128 KB for
    // 5 images of 256 x 25) against a 64 KB weight slab each.  In (channel group, image group) launch order the
    // blocks of an image group spread over all 8 XCDs and every L2 fetched every dy tile: FETCH 946 MB against
    // 118 MB of dy at 4608 rows (profiles/r03_traffic.json) -- the XCD count.  Re-map so that XCD x owns image groups
    // x, x + 8, ... with all their channel groups back to back: dy comes in once, the 2 MB of weights are resident
    // in every L2 (the host pads gridDim.y to a multiple of 8; the padding blocks leave here).
    int by = blockIdx.y, bx = blockIdx.x;
    if (MVAE_S1_XCD) {
        const unsigned lin = blockIdx.x + gridDim.x * blockIdx.y, xcd = lin & 7u, slot = lin >> 3;
        const unsigned grp = slot / gridDim.x;
        bx = (int)(slot - grp * gridDim.x);
        by = (int)(xcd + 8u * grp);
        if (by * NI >= g.B) return;
    }
    const int n0 = by * NI, ci0 = bx * 4 * CW;
    const int K = g.Cout, J = g.Cin * 16;
#if MVAE_S1_DMA
    // Operands by LDS-DMA (gemm2.h's machinery).  A stage holds k-tile [k0, k0 + 16): P as [k][128 packed rows] -- the 25
    // positions of an image at one channel are contiguous in dy but start at a multiple of 100 bytes, so its pieces are
    // 4-byte ones: one instruction fills 64 rows of one k (lane = row: image / position / validity are lane constants, the k
    // row rides the scalar offset; pad rows and images past the batch are out-of-range pieces, which the hardware zero-fills)
    // -- and Q as [k][64] by 16-byte pieces (a k row of the weight slab = 64 contiguous floats; one instruction = four rows).
    // Per wave and stage: 8 + 1 instructions, no vector registers, no ds_write.  Three stages: while k-tile s is multiplied,
    // s + 1 has landed or is landing and s + 2 is requested -- in the step the dy tile comes from HBM (118 MB at 4608 images:
    // 1139 us in situ against 991 re-issued hot), and one tile ahead in registers did not cover that.
    static_assert(S1_BK == 16, "four k rows of P and one 16-byte piece of Q per wave");
    const int wv = g2_uni(wave);
    const unsigned lds0 = (unsigned)(unsigned long)(g2_lds_void *)s1_lds;
    int pvo[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int r = h * 64 + lane, img = r / P, pos = r - img * P;
        pvo[h] = (img < NI && n0 + img < g.B) ? (img * K * P + pos) * 4 : BUF_OOB;
    }
    int qvo[CW];                                    // 16-byte piece f = wave * 64 + lane + 256 u of a stage's [16][COLS] weight image
#pragma unroll
    for (int u = 0; u < CW; ++u) {
        const int qf = wv * 64 + lane + 256 * u;
        qvo[u] = ((qf / (COLS / 4)) * J + (qf % (COLS / 4)) * 4) * 4;
    }
    const BufBase pblk = buf_base(dy + (size_t)n0 * K * P);
    const BufBase qblk = buf_base(w + (size_t)ci0 * 16);
    auto issue = [&](int s2, int stage) {
        const int k0 = s2 * S1_BK;
        const i32x4_t prs = g2_rsrc(pblk, 0, 0x7fffffff);
        const i32x4_t qrs = g2_rsrc(qblk, (long)k0 * J, 0x7fffffff);
        asm volatile("s_nop 4" ::: "memory");
        const unsigned base = lds0 + (unsigned)(stage * ST_FL) * 4u;
#pragma unroll
        for (int kr = 0; kr < 4; ++kr)
#pragma unroll
            for (int h = 0; h < 2; ++h)
                g2_dma4(prs, pvo[h], g2_uni((k0 + 4 * wv + kr) * P * 4), g2_uni(base + (unsigned)(((4 * wv + kr) * S1_ROWS + h * 64) * 4)));
#pragma unroll
        for (int u = 0; u < CW; ++u)
            g2_dma16(qrs, qvo[u], g2_uni(base + (unsigned)((S1_BK * S1_ROWS + wv * 256 + 1024 * u) * 4)));
    };
    constexpr int S1_NPW = 8 + CW;                  // DMA instructions per wave and stage
    f32x16 acc[CW][2];
#pragma unroll
    for (int y = 0; y < CW; ++y)
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[y][x][r] = 0.f;
    const int lrow = lane >> 5, lcol = lane & 31;
    const int nsteps = K / S1_BK;                   // K % S1_BK == 0: launch condition
    issue(0, 0);
    if (nsteps > 1) issue(1, 1);
    int st_c = 0, st_i = 2;
    for (int s = 0; s < nsteps; ++s) {
        // k-tile s has landed (the younger one may stay in flight), and every wave is done with the stage the next issue fills
        if (s + 1 < nsteps) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(S1_NPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        if (s + 2 < nsteps) issue(s + 2, st_i);
        st_i = st_i == S1_ST - 1 ? 0 : st_i + 1;
        const float *Pst = s1_lds + st_c * ST_FL, *Qst = Pst + S1_BK * S1_ROWS;
        st_c = st_c == S1_ST - 1 ? 0 : st_c + 1;
        float a0[2], b0[CW];
        // (a bank swizzle of the odd k rows -- the two half wavefronts of a fragment read on disjoint bank halves -- measured
        //  -0.8 % at 4608 images, +5 % on the 256-image data gradient, the steps equal: not kept)
        const int pc0 = lrow * S1_ROWS + wi * 64 + lcol, pc1 = pc0 + 32;
        const int qc = lrow * COLS + wj * 32 + lcol;        // column block y of this wave: + 64 y
        a0[0] = Pst[pc0]; a0[1] = Pst[pc1];
#pragma unroll
        for (int y = 0; y < CW; ++y) b0[y] = Qst[qc + 64 * y];
#pragma unroll
        for (int kk = 0; kk < S1_BK / 2; ++kk) {
            float a1[2] = {0.f, 0.f}, b1[CW];
#pragma unroll
            for (int y = 0; y < CW; ++y) b1[y] = 0.f;
            if (kk + 1 < S1_BK / 2) {
                a1[0] = Pst[(kk + 1) * 2 * S1_ROWS + pc0];
                a1[1] = Pst[(kk + 1) * 2 * S1_ROWS + pc1];
#pragma unroll
                for (int y = 0; y < CW; ++y) b1[y] = Qst[(kk + 1) * 2 * COLS + qc + 64 * y];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int y = 0; y < CW; ++y) {
                acc[y][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[0], b0[y], acc[y][0], 0, 0, 0);
                acc[y][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[1], b0[y], acc[y][1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            a0[0] = a1[0]; a0[1] = a1[1];
#pragma unroll
            for (int y = 0; y < CW; ++y) b0[y] = b1[y];
        }
    }
    __syncthreads();                                // every wave is done with the ring: the col2im tile takes its place
#else
    // P loader: lanes along the packed row axis r = image * P + position, 2 k rows per pass.  Buffer loads
    // (gemm_core.h): the lane part of the address -- image, position, k parity -- is a constant voffset (BUF_OOB
    // for the 3 pad rows / images past the batch: zero fill), the k-step part rides the scalar soffset; nothing
    // on the vector ALU (K % BK == 0 is a launch condition).  (Measured and not kept, round 4: lanes walking the
    // contiguous S1_BK * P run of each image -- 256 contiguous bytes per wave load instead of ~7 lines -- with the LDS
    // places as thread constants: 16 more registers, CelebA-19 6.63 -> 6.76 ms, profiles/r04_s1bk_wgt_ab.txt.)
    const int pr_ = t & 127, pkq = t >> 7;
    const int pimg = pr_ / P, ppos = pr_ - pimg * P;
    const bool pok = pimg < NI && n0 + pimg < g.B;
    const int pvoff = pok ? ((pimg * K + pkq) * P + ppos) * 4 : BUF_OOB;
    const BufBase pblk = buf_base(dy + (size_t)n0 * K * P);
    // Q loader: weight rows are contiguous in (ci, tap): 16 float4 per k row, 2 per thread
    const BufBase qblk = buf_base(w + (size_t)ci0 * 16);
    constexpr int S1_NP = S1_BK / 2, S1_NQ = S1_BK / 16;    // dwords of dy / float4 of w a thread moves per k-step
    int qvoff[S1_NQ];
#pragma unroll
    for (int v = 0; v < S1_NQ; ++v) {
        const int f = t + 256 * v;
        qvoff[v] = ((f >> 4) * J + (f & 15) * 4) * 4;
    }
    float pr[S1_NP];
    float4 qr[S1_NQ];
    auto load = [&](int k0) {
        const i32x4_t prs = buf_rsrc(pblk, 0), qrs = buf_rsrc(qblk, (size_t)k0 * J);
#pragma unroll
        for (int v = 0; v < S1_NP; ++v) pr[v] = llvm_raw_buffer_load_f32(prs, pvoff, (k0 + 2 * v) * P * 4, 0);
#pragma unroll
        for (int v = 0; v < S1_NQ; ++v) qr[v] = buf_load4(qrs, qvoff[v]);
    };
    auto store = [&](int buf) {
#pragma unroll
        for (int v = 0; v < S1_NP; ++v) Ps(buf)[pkq + 2 * v][pr_] = pr[v];
#pragma unroll
        for (int v = 0; v < S1_NQ; ++v) {
            const int f = t + 256 * v;
            *reinterpret_cast<float4 *>(&Qs(buf)[f >> 4][(f & 15) * 4]) = qr[v];
        }
    };
    f32x16 acc[1][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][x][r] = 0.f;
    const int lrow = lane >> 5, lcol = lane & 31;
    const int nsteps = (K + S1_BK - 1) / S1_BK;
    load(0);
    store(0);
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
        const int buf = s & 1;
        if ((MVAE_S1_KO & 7) < 1) load(min(s + 1, nsteps - 1) * S1_BK);    // unconditional (the last trip re-reads its own tile): no branch
        float a0[2], b0;
        a0[0] = Ps(buf)[lrow][wi * 64 + lcol]; a0[1] = Ps(buf)[lrow][wi * 64 + 32 + lcol];
        b0 = Qs(buf)[lrow][wj * 32 + lcol];
#pragma unroll
        for (int kk = 0; kk < S1_BK / 2; ++kk) {
            float a1[2] = {0.f, 0.f}, b1 = 0.f;
            if ((MVAE_S1_KO & 7) >= 3) { a1[0] = a0[0]; a1[1] = a0[1]; b1 = b0; }
            else if (kk + 1 < S1_BK / 2) {
                a1[0] = Ps(buf)[(kk + 1) * 2 + lrow][wi * 64 + lcol];
                a1[1] = Ps(buf)[(kk + 1) * 2 + lrow][wi * 64 + 32 + lcol];
                b1 = Qs(buf)[(kk + 1) * 2 + lrow][wj * 32 + lcol];
            }
            __builtin_amdgcn_sched_barrier(0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[0], b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[1], b0, acc[0][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            a0[0] = a1[0]; a0[1] = a1[1]; b0 = b1;
        }
        if ((MVAE_S1_KO & 7) < 2) {
            store(buf ^ 1);
            __syncthreads();
        }
    }
#endif
    if (MVAE_S1_KO & 8) {           // no col2im: one (never taken) store keeps the matrix instructions alive
        float sum = 0.f;
#pragma unroll
        for (int y = 0; y < CW; ++y)
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int r = 0; r < 16; ++r) sum += acc[y][x][r];
        if (sum == 12345.678f && out) out[t] = sum;
        return;
    }
    auto col2im = [&](const f32x16 (&accy)[2], const int cib) {
        // col2im: park the 128 (packed positions) x 64 (4 channels x 16 taps) tile in LDS and let every thread gather the
        // <= 16 taps of its output pixels (image, channel, pixel).  The reads are unconditional from clamped positions with
        // a 0/1 factor (16 LDS reads in flight; a branch per tap made every read wait for the one before it).
        float *sc = s1_lds;
    #pragma unroll
        for (int x = 0; x < 2; ++x)
    #pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wi * 64 + x * 32 + (r & 3) + 8 * (r >> 2) + 4 * lrow;
                sc[row * TP + wj * 32 + lcol] = accy[x][r];
            }
        if (MVAE_S1_EPI2 && t < 5) sc[t * 25 * TP + S1_COLS] = 0.f;     // the pad column of the first row of each 5 x 5 image: the zero its taps outside the image read
        __syncthreads();
        const int HW = g.H * g.W;
        const int per_img = 4 * HW;
        if (MVAE_S1_EPI2 && g.H == 8 && g.W == 8 && g.OH == 5 && g.OW == 5) {
            // 5 x 5 -> 8 x 8 (the two layers this kernel serves), NI = 5 whole images: as the path below, with what was still
            // re-derived per image or per tap taken out (profiles/r06_celeba_sq_counters.txt: 3.7 vector instructions per matrix
            // instruction over this kernel, 0.9 in its main loop -- the rest is here).  A tap outside the image reads the zero
            // slot of its image instead of a clamped position times a 0 / 1 factor (no mask registers, a plain add); the tap table is built from
            // 4 row parts + 4 column parts; the image loop is unrolled, so an image is an IMMEDIATE offset of the LDS read
            // (25 * 65 * 4 = 6500 bytes apart) and costs no address arithmetic.  Same taps in the same order: identical sums.
            constexpr int P5 = 25, IMG_FL = P5 * TP;
            const int cl = (t >> 6) & 3, ih = (t >> 3) & 7, iw = t & 7;
            int rowp[4], colp[4];
            bool rok[4], cok[4];
    #pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) {
                const int oh = ih - k4, ow = iw - k4;
                rok[k4] = (unsigned)oh < 5u; cok[k4] = (unsigned)ow < 5u;
                rowp[k4] = oh * (5 * TP) + cl * 16 + k4 * 4;
                colp[k4] = ow * TP + k4;
            }
            int toff[16];
    #pragma unroll
            for (int kh = 0; kh < 4; ++kh)
    #pragma unroll
                for (int kw = 0; kw < 4; ++kw) toff[kh * 4 + kw] = (rok[kh] && cok[kw]) ? rowp[kh] + colp[kw] : S1_COLS;
            const int ci = cib + cl;
            size_t o = ((size_t)n0 * g.Cin + ci) * 64 + (t & 63);
            const size_t ostep = (size_t)g.Cin * 64;
    #pragma unroll
            for (int img = 0; img < 5; ++img, o += ostep) {
                if (n0 + img >= g.B) break;                 // block-uniform
                float v = 0.f;
    #pragma unroll
                for (int k = 0; k < 16; ++k) v += sc[img * IMG_FL + toff[k]];
                if (dpre) v *= swish_grad_(dpre[o]);
                if (out) out[o] = v;
                if (act) act[o] = swishf_(v);
            }
            return;
        }
        if (MVAE_S1_TAPS && g.H == 8 && g.W == 8) {
            // 8 x 8 outputs (both layers that use this kernel): 4 channels x 64 pixels = the 256 threads, so a thread's
            // (channel, pixel) -- and with it the 16 tap positions and their validity -- is the same for every image of the
            // block: computed ONCE; per image a tap is one address add, one LDS read, one FMA.  On fp32 MFMA the vector
            // instructions of this epilogue are matrix time taken from the co-resident blocks (profiles/r04_celeba_sq_counters.txt:
            // 5.4 VALU per MFMA instruction over the whole kernel, about half of them here, re-derived per image).
            const int cl = (t >> 6) & 3, ih = (t >> 3) & 7, iw = t & 7;
            int toff[16];
            float tmask[16];
    #pragma unroll
            for (int kh = 0; kh < 4; ++kh) {
                const int oh = ih - kh;
                const bool okh = oh >= 0 && oh < g.OH;
                const int ohc = min(max(oh, 0), g.OH - 1);
    #pragma unroll
                for (int kw = 0; kw < 4; ++kw) {
                    const int ow = iw - kw;
                    const bool ok = okh && ow >= 0 && ow < g.OW;
                    const int owc = min(max(ow, 0), g.OW - 1);
                    toff[kh * 4 + kw] = (ohc * g.OW + owc) * TP + cl * 16 + kh * 4 + kw;
                    tmask[kh * 4 + kw] = ok ? 1.f : 0.f;
                }
            }
            const int ci = cib + cl, img_fl = P * TP;
            for (int img = 0; img < NI; ++img) {
                const int n = n0 + img;
                if (n >= g.B) break;                        // block-uniform
                const float *base = sc + img * img_fl;
                float v = 0.f;
    #pragma unroll
                for (int k = 0; k < 16; ++k) v += tmask[k] * base[toff[k]];
                const size_t o = ((size_t)n * g.Cin + ci) * HW + (t & 63);
                if (dpre) v *= swish_grad_(dpre[o]);
                if (out) out[o] = v;
                if (act) act[o] = swishf_(v);
            }
            return;
        }
        for (int idx = t; idx < NI * per_img; idx += 256) {
            const int img = idx / per_img;
            const int rem = idx - img * per_img;
            const int cl = rem / HW;
            const int px = rem - cl * HW;
            const int ih = px / g.W, iw = px - ih * g.W;
            const int n = n0 + img, ci = cib + cl;
            if (n >= g.B || ci >= g.Cin) continue;
            const float *base = sc + (img * P) * TP + cl * 16;
            float v = 0.f;
    #pragma unroll
            for (int kh = 0; kh < 4; ++kh) {
                const int oh = ih - kh;
                const bool okh = oh >= 0 && oh < g.OH;
                const int ohc = min(max(oh, 0), g.OH - 1);
    #pragma unroll
                for (int kw = 0; kw < 4; ++kw) {
                    const int ow = iw - kw;
                    const bool ok = okh && ow >= 0 && ow < g.OW;
                    const int owc = min(max(ow, 0), g.OW - 1);
                    v += (ok ? 1.f : 0.f) * base[(ohc * g.OW + owc) * TP + kh * 4 + kw];
                }
            }
            const size_t o = ((size_t)n * g.Cin + ci) * HW + ih * g.W + iw;
            if (dpre) v *= swish_grad_(dpre[o]);
            if (out) out[o] = v;
            if (act) act[o] = swishf_(v);
        }
    };
#pragma unroll
    for (int y = 0; y < CW; ++y) {
        if (y) __syncthreads();                     // the previous column block's readers are done with the tile
        col2im(acc[y], ci0 + 4 * y);
    }
}

inline bool conv_dgrad_s1_ok(const ConvGeom &g, const float *w) {
    return g.stride == 1 && g.pad == 0 && g.OH * g.OW <= 32 && g.Cin % 4 == 0 && g.Cout % S1_BK == 0 && aligned16(w) &&
           (size_t)128 * g.Cout * g.OH * g.OW * 4 < (1ull << 31);
}

inline int conv_dgrad_s1(const float *dy, const float *w, float *dx, float *act, const float *dpre, ConvGeom g,
                         hipStream_t st) {
    const int NI = S1_ROWS / (g.OH * g.OW);         // whole images per block
    dim3 grid(g.Cin / 4, (g.B + NI - 1) / NI);
    if (MVAE_S1_XCD) grid.y = (grid.y + 7) / 8 * 8;   // XCD-local image groups (see the kernel)
#if MVAE_S1_DMA
    // 128-column blocks (three per CU) where the launch still has MVAE_S1_WIDE_MIN of them: the 4608-image passes of celeba19
    if (MVAE_S1_WIDE_MIN > 0 && g.Cin % 8 == 0 && (long)(g.Cin / 8) * grid.y >= MVAE_S1_WIDE_MIN) {
        grid.x = g.Cin / 8;
        hipLaunchKernelGGL(convT_s1_kernel<2>, grid, dim3(256), 0, st, dy, w, dx, act, dpre, g, NI);
        return mvae_launch_status();
    }
#endif
    hipLaunchKernelGGL(convT_s1_kernel<1>, grid, dim3(256), 0, st, dy, w, dx, act, dpre, g, NI);
    return mvae_launch_status();
}

inline size_t dgrad_ws_floats(const ConvGeom &g) { return (size_t)g.Cout * g.Cin * 16; }

// ---- conv dgrad form: dx[n][ci][ih][iw] = sum_(co,kh,kw) w[co][ci][kh][kw] * dy[n][co][oh][ow],
//      one launch per output parity class (4 for stride 2, 1 for stride 1) on repacked weights ----
int conv_dgrad_impl(const float *dy, const float *w, float *dx, float *act, const float *dpre,
                    ConvGeom g, void *ws, size_t ws_bytes, hipStream_t st) {
    const int s = g.stride, tlog = (s == 2) ? 1 : 2;
    const int H2 = g.H / s, W2 = g.W / s;
    const int I = g.Cin, J = g.B * H2 * W2, K = g.Cout << (2 * tlog);
    // w == NULL: `ws` already holds the repacked weights (mvae_conv_k4_repack_batched) -- only valid for launches
    // that read them (mvae_conv_k4_repack_floats != 0)
    if (conv_dgrad_small_ok(g) && !MVAE_TUNE(wm)) return w ? conv_dgrad_small(dy, w, dx, act, dpre, g, st) : MVAE_ERR_ARG;
    if (conv_dgrad_s1_ok(g, w) && !MVAE_TUNE(wm)) return w ? conv_dgrad_s1(dy, w, dx, act, dpre, g, st) : MVAE_ERR_ARG;
    if (!ws || ws_bytes < dgrad_ws_floats(g) * sizeof(float)) return MVAE_ERR_WS;
    float *wr = (float *)ws;
    if (w) {
        const int total = s * s * K * g.Cin;
        int blocks = (total + 255) / 256;
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(repack_dgrad_weights_kernel, dim3(blocks), dim3(256), 0, st, w, wr, g.Cout, g.Cin, s, g.pad);
    }
    Plan pl = make_plan(I, J, K, false, PLAN_FWD, s * s);
    if (K <= MVAE_MULTI_MAXK) pl.items = MVAE_MULTI_ITEMS;      // short reductions: pipeline across consecutive tiles / classes
    pl.xcd = MVAE_CONV_XCD ? 3 : 0;
    const bool vec = (g.Cin % 4 == 0) && aligned16(wr);
    EpNCHW e;
    e.out = dx; e.act = act; e.dpre = dpre;
    e.C = g.Cin; e.HW = g.H * g.W; e.Wfull = g.W; e.H2 = H2; e.W2 = W2;
    e.sy = s; e.py = 0; e.px = 0; e.J = J; e.off = 0;
    e.lg_hw2 = g.lg_hw2; e.lg_w2 = g.lg_w2;
#ifndef MVAE_PAIR_STORE
#define MVAE_PAIR_STORE 1
#endif
#ifndef MVAE_PAIR_MAXK
#define MVAE_PAIR_MAXK 512
#endif
#ifndef MVAE_PAIR_MINBLOCKS
#define MVAE_PAIR_MINBLOCKS 2048
#endif
    // pair stores (gemm_core.h EpNCHW::PAIR): class-minor item order puts (py, 0), (py, 1) back to back in a block
    e.pair = (MVAE_PAIR_STORE && s == 2 && MVAE_CLS_MINOR && g.W % 2 == 0 && (!dx || aligned8(dx)) && (!act || aligned8(act)) &&
              (!dpre || aligned8(dpre))) ? 1 : 0;
    auto mp = [&](auto &p) {
        p.src = wr; p.ld = g.Cin; p.R = g.Cin; p.Klen = K; p.cls_stride = (size_t)K * g.Cin;
    };
    auto mq = [&](auto &q) { q.dy = dy; q.g = g; q.Mtot = J; q.H2 = H2; q.W2 = W2; };
    SplitSink sink = make_sink(nullptr, I, J, false);
    sink.ncls = s * s;      // all parity classes in ONE launch: s*s times the blocks
    sink.cls_minor = MVAE_CLS_MINOR;
    if (vec && s == 2 && e.pair && MVAE_EP_BUFFER && !MVAE_TUNE(wm) && aligned16(dy)) {
        // one LDS input patch for the four parity classes (convt_patch.h): the 64- and 32-row layers on 7 x 7 / 8 x 8 / 16 x 16 maps
        const PatchPlan pp = convt_patch_plan(g.B, g.Cout, g.Cin, g.OH, g.OW, false);
        if (pp.kind) {
            EpNCHWPair ep;
            static_cast<EpNCHW &>(ep) = e;
            if (pp.kind == 1) return launch_convt_patch2<EpNCHWPair, 1, 68, true, MVAE_PATCH_STAGE8 ? 8 : 0, 8>(pp, dy, wr, ep, st);
            if (pp.kind == 2) return launch_convt_patch2<EpNCHWPair, 1, 148, false, MVAE_PATCH_STAGE7 ? 7 : 0, 11>(pp, dy, wr, ep, st);
            if (pp.kind == 3) return launch_convt_patch2<EpNCHWPair, 1, 100, true>(pp, dy, wr, ep, st);
        }
    }
    if (vec) {
        // pair stores want the two px classes of a tile in ONE block.  Short reductions (K <= 256) run multi-item
        // blocks anyway; for K = 512 (the 64-channel layers) two items per block pay only when the launch still
        // has >= 8 blocks per CU afterwards (measured: FashionMNIST's 2048-row ConvTranspose2d(128, 64) -2.9 % of
        // the step, CelebA's 512-row one +0.9 %: profiles/r03_pair_store_ab.txt)
        const long pair_blocks = cdiv(J, pl.wgn == 4 ? 128 : 64) * 4 / 2;
        const bool pair_long = K > MVAE_MULTI_MAXK && K <= MVAE_PAIR_MAXK && pair_blocks >= MVAE_PAIR_MINBLOCKS;
        if (s == 2 && e.pair && pl.wm * pl.wn == 1 && pl.kw == 1 && (K <= MVAE_MULTI_MAXK || pair_long)) {
            if (pair_long) pl.items = 2;
            EpNCHWPair ep;
            static_cast<EpNCHW &>(ep) = e;
            return launch_igemm<LdRowsMNC, LdDgradDyS2, EpNCHWPair, false>(pl, mp, mq, ep, I, J, K, sink, st);
        }
        if (s == 2) return launch_igemm<LdRowsMNC, LdDgradDyS2, EpNCHW, false>(pl, mp, mq, e, I, J, K, sink, st);
        return launch_igemm<LdRowsMNC, LdDgradDyS1, EpNCHW, false>(pl, mp, mq, e, I, J, K, sink, st);
    }
    if (s == 2) return launch_igemm<LdRowsMNSC, LdDgradDyS2, EpNCHW, false>(pl, mp, mq, e, I, J, K, sink, st);
    return launch_igemm<LdRowsMNSC, LdDgradDyS1, EpNCHW, false>(pl, mp, mq, e, I, J, K, sink, st);
}

// ---- statistics-only form of the stride-2 dgrad-form launch (EpStats, gemm_core.h): the transposed conv in front of a
//      decoder pass's LAST BatchNorm when the pass exists only for the running statistics.  Supported: stride 2, <= 32
//      output channels (the 32-row layout), whole column tiles, a reduction of >= 2 full k-tiles; one block = the four
//      parity classes of one 128-column tile = 512 elements per channel.  Returns the number of records (column
//      tiles), 0 if the shape is not covered (the caller then runs the storing launch + the statistics sweep).
constexpr int STATS_TILE = 128;
inline long conv_dgrad_stats_tiles(const ConvGeom &g) {
    const int s = g.stride;
    if (s != 2 || g.pad != 1 || g.Cin > 32 || g.Cin % 4 != 0) return 0;
    const long J = (long)g.B * (g.H / s) * (g.W / s);
    const int K = g.Cout << 2;
    if (J % STATS_TILE != 0 || K % MVAE_CONV_BK != 0 || K < 2 * MVAE_CONV_BK || J * 4 >= (1L << 31)) return 0;
    return J / STATS_TILE;
}

int conv_dgrad_stats_impl(const float *dy, const float *w, float *part, ConvGeom g, void *ws, size_t ws_bytes,
                          hipStream_t st) {
    const int s = g.stride;
    const int H2 = g.H / s, W2 = g.W / s;
    const int I = g.Cin, J = g.B * H2 * W2, K = g.Cout << 2;
    if (!conv_dgrad_stats_tiles(g)) return MVAE_ERR_ARG;
    if (!ws || ws_bytes < dgrad_ws_floats(g) * sizeof(float)) return MVAE_ERR_WS;
    float *wr = (float *)ws;
    if (!aligned16(wr)) return MVAE_ERR_ARG;
    if (w) {
        const int total = s * s * K * g.Cin;
        int blocks = (total + 255) / 256;
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(repack_dgrad_weights_kernel, dim3(blocks), dim3(256), 0, st, w, wr, g.Cout, g.Cin, s, g.pad);
    }
    if (!MVAE_TUNE(wm) && aligned16(dy)) {
        const PatchPlan pp = convt_patch_plan(g.B, g.Cout, g.Cin, g.OH, g.OW, true);
        if (pp.kind == 4) {                                 // one record per 128-column tile, as below
            EpStats es;
            es.part = part; es.C = g.Cin; es.J = J;
            return launch_convt_patch2<EpStats, 2, 100, true>(pp, dy, wr, es, st);
        }
    }
    Plan pl = make_plan(I, J, K, false, PLAN_FWD, s * s);
    if (!(pl.wgn == 4 && pl.wm == 1 && pl.wn == 1 && pl.kw == 1 && pl.splits == 1)) return MVAE_ERR_ARG;   // the 32-row layout
    pl.items = s * s; pl.force_items = 1;               // a block = the four classes of one column tile
    EpStats e;
    e.part = part; e.C = g.Cin; e.J = J;
    auto mp = [&](auto &p) { p.src = wr; p.ld = g.Cin; p.R = g.Cin; p.Klen = K; p.cls_stride = (size_t)K * g.Cin; };
    auto mq = [&](auto &q) { q.dy = dy; q.g = g; q.Mtot = J; q.H2 = H2; q.W2 = W2; };
    SplitSink sink = make_sink(nullptr, I, J, false);
    sink.ncls = s * s;
    sink.cls_minor = 1;
    return launch_igemm<LdRowsMNC, LdDgradDyS2, EpStats, false>(pl, mp, mq, e, I, J, K, sink, st);
}

// ---- weight gradient of the <= 4-input-channel convs (Conv2d(3,32) / ConvTranspose2d(32,3) of CelebA,
//      Conv2d(1,64) / ConvTranspose2d(64,1) of FashionMNIST; stride 2, pad 1).  The output is 32..64 x
//      16..48 values over a reduction of B*OH*OW ~ 10^5..10^6: as an implicit GEMM that is ONE
//      under-filled tile split 512 ways, a third of whose gathered columns are padding (62 / 107 us on
//      CelebA B = 256).  It is an HBM-bound op (46 / 92 MB): here every wave streams whole output rows --
//      unit (b, oh): the CO x OW slab of dy and the CI x 4 input rows it touches, both staged through
//      the wave's own LDS with coalesced float4 loads -- and multiplies them with MFMAs (k = ow);
//      per-block partials go to scratch and the ordinary split finish sums them in a fixed order. ----
constexpr int SC_MAXW = 64;                 // input row length limit
constexpr int SC_XW = SC_MAXW + 8;          // staged input row: 4 zero floats, the row, 4 zero floats
constexpr int SC_DW = 33;                   // staged dy row (OW <= 32, odd pitch: conflict-free fragment reads)

constexpr int SC_WAVES = 8;                 // waves per block, each streaming its own units

template <int MT, int NT>
__global__ __launch_bounds__(64 * SC_WAVES) void wgrad_smallcin_kernel(const float *dy, const float *x, float *ws, ConvGeom g,
                                                             int units) {
    extern __shared__ __attribute__((aligned(16))) float sc_lds[];
    constexpr int CO = 32 * MT;
    constexpr int WAVE_FLOATS = CO * SC_DW + 16 * SC_XW;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    float *dys = sc_lds + wave * WAVE_FLOATS;          // [CO][SC_DW]
    float *xs = dys + CO * SC_DW;                      // [CI*4][SC_XW]
    const int J = g.Cin * 16, OW = g.OW, W = g.W;
    f32x16 acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    // zero the halos of the input rows once (the row bodies are rewritten per unit)
    for (int i = lane; i < g.Cin * 4 * 8; i += 64) {
        const int row = i >> 3, c = i & 7;
        xs[row * SC_XW + (c < 4 ? c : W + c)] = 0.f;
    }
    const int lr = lane & 31, lk = lane >> 5;
    // per-lane column j = (ci, kh, kw) of each column tile
    int xoff[NT]; float xmask[NT];
#pragma unroll
    for (int b = 0; b < NT; ++b) {
        const int j = b * 32 + lr;
        const bool ok = j < J;
        const int jj = ok ? j : 0;
        xoff[b] = ((jj >> 4) * 4 + ((jj >> 2) & 3)) * SC_XW + 4 - 1 + (jj & 3);      // + 2*k at use
        xmask[b] = ok ? 1.f : 0.f;
    }
    // registers of the NEXT unit: its global loads are in flight while this unit is multiplied
    constexpr int NDY = 8 * MT;                 // float2 per lane for CO x OW <= 32*MT x 32
    constexpr int NX = 4;                       // float4 per lane for <= 16 rows x 64
    float2 dyr[NDY]; float4 xr[NX];
    const int v2 = OW >> 1, v4 = W >> 2;
    auto fetch = [&](int u) {
        const int b = u / g.OH, oh = u - b * g.OH;
        const float *dyb = dy + ((size_t)b * CO * g.OH + oh) * OW;
#pragma unroll
        for (int i = 0; i < NDY; ++i) {
            const int e = lane + 64 * i;
            const int co = min(e / v2, CO - 1), c2 = e % v2;       // clamped: always a legal address
            dyr[i] = *reinterpret_cast<const float2 *>(dyb + (size_t)co * g.OH * OW + c2 * 2);
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int e = lane + 64 * i;
            const int row = min(e / v4, g.Cin * 4 - 1), c4 = e % v4;
            const int ci = row >> 2, ih = 2 * oh - 1 + (row & 3);
            const int ihc = min(max(ih, 0), g.H - 1);
            const float4 v = *reinterpret_cast<const float4 *>(x + (((size_t)b * g.Cin + ci) * g.H + ihc) * W + c4 * 4);
            const float m = (ih >= 0 && ih < g.H) ? 1.f : 0.f;      // rows outside the image are zero padding
            xr[i] = make_float4(v.x * m, v.y * m, v.z * m, v.w * m);
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int i = 0; i < NDY; ++i) {
            const int e = lane + 64 * i;
            if (e < CO * v2) {
                const int co = e / v2, c2 = e % v2;
                dys[co * SC_DW + c2 * 2] = dyr[i].x; dys[co * SC_DW + c2 * 2 + 1] = dyr[i].y;
            }
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int e = lane + 64 * i;
            if (e < g.Cin * 4 * v4) {
                const int row = e / v4, c4 = e % v4;
                *reinterpret_cast<float4 *>(xs + row * SC_XW + 4 + c4 * 4) = xr[i];
            }
        }
    };
    const int nwaves = gridDim.x * SC_WAVES;
    int u = blockIdx.x * SC_WAVES + wave;
    if (u < units) fetch(u);
    for (; u < units; u += nwaves) {
        stage();
        __builtin_amdgcn_wave_barrier();
        if (u + nwaves < units) fetch(u + nwaves);
        // ---- k = ow in pairs
#pragma unroll 4
        for (int q = 0; q < (OW >> 1); ++q) {
            const int k = 2 * q + lk;
            float af[MT], bf[NT];
#pragma unroll
            for (int a = 0; a < MT; ++a) af[a] = dys[(a * 32 + lr) * SC_DW + k];
#pragma unroll
            for (int b2 = 0; b2 < NT; ++b2) bf[b2] = xs[xoff[b2] + 2 * k] * xmask[b2];
#pragma unroll
            for (int a = 0; a < MT; ++a)
#pragma unroll
                for (int b2 = 0; b2 < NT; ++b2)
                    acc[a][b2] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a], bf[b2], acc[a][b2], 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
    }
    // ---- sum the waves' partials in a fixed order, write this block's [CO][J] partial
    __syncthreads();
    float *red = sc_lds;                               // (SC_WAVES - 1) x MT*NT tiles of 1024 floats
    if (wave > 0) {
        float *dst = red + (wave - 1) * MT * NT * 1024 + lane;
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
            for (int b = 0; b < NT; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) dst[((a * NT + b) * 16 + r) * 64] = acc[a][b][r];
    }
    __syncthreads();
    if (wave > 0) return;
    float *out = ws + (size_t)blockIdx.x * CO * J;
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[a][b][r];
#pragma unroll
                for (int w2 = 0; w2 < SC_WAVES - 1; ++w2) v += red[w2 * MT * NT * 1024 + ((a * NT + b) * 16 + r) * 64 + lane];
                const int co = a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk, j = b * 32 + lr;
                if (j < J) out[(size_t)co * J + j] = v;
            }
}

// ---- the same through LDS-DMA (round 6).  What the register-staged kernel above paid per unit: 12 vector loads of 8 / 16
//      bytes per lane (the 8-byte ones at a quarter of the addresser's rate), ~20 ds_write, ONE unit of prefetch (a second
//      register set does not fit beside 32-64 accumulators) -- and, per launch, a tail that does not shrink with the batch:
//      B = 256 -> 512 rows cost 11.6 us for 46 MB more (4 TB/s), the first 256 rows 23 us (profiles/r06_small_conv.txt).
//      Here a wave's units arrive by `buffer_load_dword ... lds` into a wave-private ring of S stages: no staging registers,
//      no ds_write, S - 1 units in flight, a counted vmcnt and NO barrier (the ring is the wave's own).  The dy slab lands
//      as [CO][P] (P = 16 or 32 columns) with the column index XOR-swizzled on the SOURCE side -- element (co, ow) sits at
//      column ow ^ s(co), s(co) = (co / (32 / P)) % P -- so that the A fragments (32 lanes = 32 channels, one column) read 32
//      different banks; the input rows land in zero-haloed rows of pitch XW = 4 x odd, as before.  A row outside the image is
//      fetched through an EMPTY descriptor (zeros), a dy column beyond OW through an out-of-range offset.
constexpr int SC2_WAVES = 8;
template <int MT, int NT, int CIN, int P, int S>
__global__ __launch_bounds__(64 * SC2_WAVES) void wgrad_smallcin2_kernel(const float *dy, const float *x, float *ws, ConvGeom g,
                                                                          int units, int XW) {
    extern __shared__ __attribute__((aligned(16))) float sc_lds[];
    constexpr int CO = 32 * MT;
    constexpr int NDY = CO * P / 64;               // DMA instructions of a dy slab
    constexpr int NXR = CIN * 4;                   // input rows of a unit: one DMA instruction each (W <= 64)
    constexpr int NPU = NDY + NXR;
    static_assert((S - 1) * NPU <= 63, "vmcnt is a 6-bit counter");
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int stage_floats = CO * P + NXR * XW;
    float *ring = sc_lds + wave * S * stage_floats;
    const unsigned ring_b = (unsigned)(unsigned long)(g2_lds_void *)ring;
    const int J = CIN * 16, OW = g.OW, W = g.W, OH = g.OH;
    f32x16 acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    // the halos of every stage's input rows: zero once, never rewritten (the DMA touches columns 4 .. 4 + W - 1 only)
    for (int i = lane; i < S * NXR * 8; i += 64) {
        const int st = i / (NXR * 8), rem = i - st * (NXR * 8), row = rem >> 3, c = rem & 7;
        ring[st * stage_floats + CO * P + row * XW + (c < 4 ? c : W + c)] = 0.f;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int lr = lane & 31, lk = lane >> 5;
    // dy: LDS dword `pos` of instruction i holds element (co, ow) = (pos / P, (pos % P) ^ s(co)), or zero
    int dvoff[NDY];
#pragma unroll
    for (int i = 0; i < NDY; ++i) {
        const int pos = i * 64 + lane, co = pos / P, ow = (pos % P) ^ ((co / (32 / P)) % P);
        dvoff[i] = ow < OW ? (co * OH * OW + ow) * 4 : BUF_OOB;
    }
    // fragment addresses: A = dy[co = 32 a + lr][ow = 2 q + lk], B = x tap column j = 32 b + lr at input column 2 ow - 1 + kw
    int aoff[MT], asw[MT], xoff[NT]; float xmask[NT];
#pragma unroll
    for (int a = 0; a < MT; ++a) { const int co = a * 32 + lr; aoff[a] = co * P; asw[a] = (co / (32 / P)) % P; }
#pragma unroll
    for (int b = 0; b < NT; ++b) {
        const int j = b * 32 + lr;
        const bool ok = j < J;
        const int jj = ok ? j : 0;
        xoff[b] = CO * P + ((jj >> 4) * 4 + ((jj >> 2) & 3)) * XW + 4 - 1 + (jj & 3);      // + 2 * ow at use
        xmask[b] = ok ? 1.f : 0.f;
    }
    const BufBase bdy = buf_base(dy), bx = buf_base(x);
    const int nwaves = gridDim.x * SC2_WAVES;
    const int u0 = blockIdx.x * SC2_WAVES + wave;
    auto issue = [&](int u, int stage) {
        const int b = u / OH, oh = u - b * OH;
        const unsigned dst = ring_b + (unsigned)(stage * stage_floats) * 4u;
        const i32x4_t rdy = g2_rsrc(bdy, ((long)b * CO * OH + oh) * OW, 0x7fffffff);
        asm volatile("s_nop 4" ::: "memory");
#pragma unroll
        for (int i = 0; i < NDY; ++i) g2_dma4(rdy, dvoff[i], 0, g2_uni(dst + i * 256));
        const long img = (long)b * CIN * g.H * W;
#pragma unroll
        for (int r = 0; r < NXR; ++r) {
            const int ci = r >> 2, ih = 2 * oh - 1 + (r & 3);
            const bool in = ih >= 0 && ih < g.H;                     // wave-uniform
            const i32x4_t rx = g2_rsrc(bx, img + ((long)ci * g.H + (in ? ih : 0)) * W, in ? 0x7fffffff : 0);
            asm volatile("s_nop 4" ::: "memory");
            // only the lanes of the row's W columns take part (EXEC): the piece must not run on into the halo and the next row
            if (lane < W) g2_dma4(rx, lane * 4, 0, g2_uni(dst + (CO * P + r * XW + 4) * 4));
        }
    };
    int nmine = 0;
    for (int u = u0; u < units; u += nwaves) ++nmine;
    int ui = u0;                                   // next unit to issue
#pragma unroll
    for (int s2 = 0; s2 < S - 1; ++s2) {
        if (s2 < nmine) { issue(ui, s2); ui += nwaves; }
    }
    int st_c = 0, st_i = (S - 1) % S;
    for (int n = 0; n < nmine; ++n) {
        if (n + S - 1 < nmine) {
            issue(ui, st_i); ui += nwaves;
            st_i = st_i + 1 == S ? 0 : st_i + 1;
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"((S - 1) * NPU) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const float *stg = ring + st_c * stage_floats;
#pragma unroll 4
        for (int q = 0; q < (OW >> 1); ++q) {
            const int k = 2 * q + lk;
            float af[MT], bf[NT];
#pragma unroll
            for (int a = 0; a < MT; ++a) af[a] = stg[aoff[a] + (k ^ asw[a])];
#pragma unroll
            for (int b2 = 0; b2 < NT; ++b2) bf[b2] = stg[xoff[b2] + 2 * k] * xmask[b2];
#pragma unroll
            for (int a = 0; a < MT; ++a)
#pragma unroll
                for (int b2 = 0; b2 < NT; ++b2)
                    acc[a][b2] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a], bf[b2], acc[a][b2], 0, 0, 0);
        }
        st_c = st_c + 1 == S ? 0 : st_c + 1;
    }
    // ---- sum the waves' partials in a fixed order, write this block's [CO][J] partial (as the kernel above)
    __syncthreads();
    float *red = sc_lds;
    if (wave > 0) {
        float *dst = red + (wave - 1) * MT * NT * 1024 + lane;
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
            for (int b = 0; b < NT; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) dst[((a * NT + b) * 16 + r) * 64] = acc[a][b][r];
    }
    __syncthreads();
    if (wave > 0) return;
    float *out = ws + (size_t)blockIdx.x * CO * J;
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[a][b][r];
#pragma unroll
                for (int w2 = 0; w2 < SC2_WAVES - 1; ++w2) v += red[w2 * MT * NT * 1024 + ((a * NT + b) * 16 + r) * 64 + lane];
                const int co = a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk, j = b * 32 + lr;
                if (j < J) out[(size_t)co * J + j] = v;
            }
}

// the partials of up to 1024 blocks summed with ONE memory round trip: 32 outputs x 32 groups per block, group q adds
// partials q, q + 32, ... (all its loads in flight), the 32 group sums combined through LDS in a fixed order
template <class E>
__global__ __launch_bounds__(1024) void finish_wide_kernel(SplitSink sink, int splits, E e) {
    __shared__ float part[32][33];
    const int o = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int j = blockIdx.x * 32 + o, i = blockIdx.y;
    float v[32];
    const int jc = min(j, sink.J - 1);
    const float *src = sink.ws + (size_t)i * sink.J + jc;
#pragma unroll
    for (int z = 0; z < 32; ++z) {
        const int zz = grp + 32 * z;
        v[z] = src[(size_t)min(zz, splits - 1) * sink.stride] * (zz < splits ? 1.f : 0.f);
    }
    float s0 = 0.f;
#pragma unroll
    for (int z = 0; z < 32; ++z) s0 += v[z];
    part[grp][o] = s0;
    __syncthreads();
    if (grp == 0 && j < sink.J && e.col(j)) {
        float tot = 0.f;
#pragma unroll
        for (int q = 0; q < 32; ++q) tot += part[q][o];
        e.put(i, j, tot);
    }
}

#ifndef MVAE_SC2
#define MVAE_SC2 1                // weight gradient of the <= 4-input-channel convs: the LDS-DMA kernel (0: the register-staged one)
#endif
#ifndef MVAE_SC2_STAGES
#define MVAE_SC2_STAGES 2
#endif

inline bool wgrad_smallcin_ok(const ConvGeom &g) {
    return g.stride == 2 && g.pad == 1 && g.Cin <= 4 && (g.Cout == 32 || g.Cout == 64) && g.OW <= 32 &&
           (g.OW & 1) == 0 && g.W <= SC_MAXW && (g.W & 3) == 0;
}
inline int wgrad_smallcin_blocks(const ConvGeom &g) {
    const int units = g.B * g.OH;
    int blocks = (units + SC_WAVES - 1) / SC_WAVES;
    return blocks > MVAE_SC_BLOCKS ? MVAE_SC_BLOCKS : blocks;
}

// ---- conv wgrad form: dw[co][(ci,kh,kw)] = sum_(n,oh,ow) dy[n][co][oh][ow] * x[n][ci][ih][iw] ----
int conv_wgrad_impl(const float *dy, const float *x, float *dw, ConvGeom g, int flags, void *ws,
                    size_t ws_bytes, hipStream_t st) {
    const int I = g.Cout, J = g.Cin * 16, K = g.B * g.OH * g.OW;
    EpRowMajor e;
    e.out = dw; e.act = nullptr; e.ld = J; e.bias = nullptr; e.dpre = nullptr; e.ldp = 0;
    e.mask = nullptr; e.ldm = 0; e.mask_scale = 1.f; e.I = I; e.J = J;
    e.accumulate = (flags & MVAE_ACCUMULATE) ? 1 : 0;
    if (wgrad_smallcin_ok(g) && !MVAE_TUNE(wm) && !MVAE_TUNE(splits) && ws) {
        const int blocks = wgrad_smallcin_blocks(g);
        if (ws_bytes >= (size_t)blocks * I * J * sizeof(float)) {
            const int mt = I / 32, nt = (J + 31) / 32;
            bool launched = false;
            if (MVAE_SC2) {
                int xw = g.W + 8;
                if (((xw / 4) & 1) == 0) xw += 4;                       // pitch = 4 x odd: the tap columns of a tile hit 32 banks
                const int p = g.OW <= 16 ? 16 : 32;
                const size_t stage_f = (size_t)I * p + (size_t)g.Cin * 4 * xw;
                const size_t red_f = (size_t)(SC2_WAVES - 1) * mt * nt * 1024;
                size_t lds2 = (size_t)SC2_WAVES * MVAE_SC2_STAGES * stage_f;
                if (red_f > lds2) lds2 = red_f;
                lds2 *= sizeof(float);
#define MVAE_SC2L(MT_, NT_, CI_, P_)                                                                        \
    {                                                                                                       \
        auto kern = wgrad_smallcin2_kernel<MT_, NT_, CI_, P_, MVAE_SC2_STAGES>;                             \
        static bool attr_done = false;                                                                      \
        if (!attr_done) {                                                                                   \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern),                                 \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);              \
            attr_done = true;                                                                               \
        }                                                                                                   \
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * SC2_WAVES), lds2, st, dy, x, (float *)ws, g, g.B * g.OH, xw); \
        launched = true;                                                                                    \
    }
                if (lds2 <= 160 * 1024 && (size_t)g.B * I * g.OH * g.OW * 4 < ((size_t)1 << 31) &&
                    (size_t)g.B * g.Cin * g.H * g.W * 4 < ((size_t)1 << 31)) {
                    if (I == 32 && g.Cin == 3 && p == 32) MVAE_SC2L(1, 2, 3, 32)
                    else if (I == 64 && g.Cin == 1 && p == 16) MVAE_SC2L(2, 1, 1, 16)
                    else if (I == 32 && g.Cin == 1 && p == 32) MVAE_SC2L(1, 1, 1, 32)
                    else if (I == 32 && g.Cin == 1 && p == 16) MVAE_SC2L(1, 1, 1, 16)
                }
#undef MVAE_SC2L
            }
            if (!launched) {
            const size_t wave_b = ((size_t)I * SC_DW + 16 * SC_XW) * sizeof(float);
            const size_t red_b = (size_t)(SC_WAVES - 1) * mt * nt * 1024 * sizeof(float);
            const size_t lds = SC_WAVES * wave_b > red_b ? SC_WAVES * wave_b : red_b;
#define MVAE_SC(MT_, NT_)                                                                                   \
    {                                                                                                       \
        auto kern = wgrad_smallcin_kernel<MT_, NT_>;                                                        \
        static bool attr_done = false;                                                                      \
        if (!attr_done) {                                                                                   \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern),                                 \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);              \
            attr_done = true;                                                                               \
        }                                                                                                   \
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * SC_WAVES), lds, st, dy, x, (float *)ws, g, g.B * g.OH); \
    }
            if (mt == 1 && nt == 1) MVAE_SC(1, 1)
            else if (mt == 1) MVAE_SC(1, 2)
            else if (nt == 1) MVAE_SC(2, 1)
            else MVAE_SC(2, 2)
#undef MVAE_SC
            }
            SplitSink fs = make_sink(ws, I, J, false);
            if (blocks > 16 && blocks <= 1024 && MVAE_SC2) {
                hipLaunchKernelGGL((finish_wide_kernel<EpRowMajor>), dim3((J + 31) / 32, I), dim3(1024), 0, st, fs, blocks, e);
            } else if (blocks > 16) {
                hipLaunchKernelGGL((finish_kernel<EpRowMajor>), dim3((J + 31) / 32, I), dim3(256), 0, st, fs, blocks, e);
            } else {
                hipLaunchKernelGGL((finish_few_kernel<EpRowMajor>), dim3((J + 255) / 256, I), dim3(256), 0, st, fs,
                                   blocks, e);
            }
            return mvae_launch_status();
        }
    }
    if (g.stride == 2 && g.pad == 1 && !MVAE_TUNE(wm) && !MVAE_TUNE(splits)) {
        // both operands in their natural layout through LDS-DMA (wgrad_patch.h): the 8 x 8 and 16 x 16 lattices
        WgradPatchGeo wg;
        if (wgrad_patch_plan(g.B, g.Cout, g.Cin, g.OH, g.OW, dy, x, ws, ws_bytes, &wg)) {
            launch_wgrad_patch(wg, dy, x, st);
            SplitSink fs = make_sink(ws, I, J, false);
            if (wg.splits > 64)
                hipLaunchKernelGGL((finish_wide_kernel<EpRowMajor>), dim3((J + 31) / 32, I), dim3(1024), 0, st, fs, wg.splits, e);
            else if (wg.splits > 16)         // (the 1024-thread form issues 32 loads per thread whatever the count)
                hipLaunchKernelGGL((finish_kernel<EpRowMajor>), dim3((J + 31) / 32, I), dim3(256), 0, st, fs, wg.splits, e);
            else
                hipLaunchKernelGGL((finish_few_kernel<EpRowMajor>), dim3((J + 255) / 256, I), dim3(256), 0, st, fs, wg.splits, e);
            return mvae_launch_status();
        }
    }
    Plan pl = make_plan(I, J, K, true, PLAN_CONV_WGRAD);
    pl.xcd = MVAE_WGRAD_XCD ? 4 : 0;                            // the tiles of one k range on one XCD (igemm_kernel)
    SplitSink sink = make_sink(ws, I, J, false);
    if (pl.splits > 1 && (!ws || ws_bytes < pl.splits * sink.stride * sizeof(float))) return MVAE_ERR_WS;
    auto mp = [&](auto &p) { p.dy = dy; p.g = g; };
    auto mq = [&](auto &q) { q.x = x; q.g = g; q.J = J; };
    return launch_igemm<LdWgradDy, LdWgradX, EpRowMajor, false>(pl, mp, mq, e, I, J, K, sink, st);
}

}  // namespace


MVAE_EXPORT int mvae_conv2d_k4_fwd(const float *x, const float *w, float *pre, float *act, int B, int Cin,
                                   int H, int W, int Cout, int stride, int pad, mvae_stream_t stream) {
    if (!x || !w || (!pre && !act) || !conv_args_ok(B, Cin, H, W, Cout, stride, pad)) return MVAE_ERR_ARG;
    return conv_fwd_impl(x, w, pre, act, nullptr, make_geom(B, Cin, H, W, Cout, stride, pad), (hipStream_t)stream);
}

MVAE_EXPORT int mvae_conv2d_k4_dgrad(const float *dy, const float *w, float *dx, const float *pre_in, int B,
                                     int Cin, int H, int W, int Cout, int stride, int pad, void *ws,
                                     size_t ws_bytes, mvae_stream_t stream) {
    if (!dy || (!w && !ws) || !dx || !conv_args_ok(B, Cin, H, W, Cout, stride, pad)) return MVAE_ERR_ARG;
    return conv_dgrad_impl(dy, w, dx, nullptr, pre_in, make_geom(B, Cin, H, W, Cout, stride, pad), ws, ws_bytes,
                           (hipStream_t)stream);
}

MVAE_EXPORT int mvae_conv2d_k4_wgrad(const float *dy, const float *x, float *dw, int B, int Cin, int H, int W,
                                     int Cout, int stride, int pad, int flags, void *ws, size_t ws_bytes,
                                     mvae_stream_t stream) {
    if (!dy || !x || !dw || !conv_args_ok(B, Cin, H, W, Cout, stride, pad)) return MVAE_ERR_ARG;
    return conv_wgrad_impl(dy, x, dw, make_geom(B, Cin, H, W, Cout, stride, pad), flags, ws, ws_bytes,
                           (hipStream_t)stream);
}

// ConvTranspose2d(Cin -> Cout), x[B,Cin,H,W] -> y[B,Cout,OH,OW], OH = (H-1)*s - 2p + 4, w[Cin,Cout,4,4]:
// the mirrored conv maps y-shaped tensors (its input, Cout channels) to x-shaped ones (its output).
static inline bool convT_geom(int B, int Cin, int H, int W, int Cout, int stride, int pad, ConvGeom *g) {
    const int OH = (H - 1) * stride - 2 * pad + 4, OW = (W - 1) * stride - 2 * pad + 4;
    if (!conv_args_ok(B, Cout, OH, OW, Cin, stride, pad)) return false;
    *g = make_geom(B, /*conv Cin*/ Cout, OH, OW, /*conv Cout*/ Cin, stride, pad);
    return g->OH == H && g->OW == W;
}

// ---- the repacked weight copies of several layers in ONE launch, ahead of the step: the dgrad-form launches
//      (Conv2d data gradient, ConvTranspose2d forward) read the weights class-major / channel-contiguous;
//      repacking inside every launch put a 5-us kernel on the chain in front of each of them ----
constexpr int REPACK_MAX = 16;
struct RepackArgs { const float *w[REPACK_MAX]; float *wr[REPACK_MAX]; int Cout[REPACK_MAX], Cin[REPACK_MAX], stride[REPACK_MAX], pad[REPACK_MAX], end[REPACK_MAX]; int n; };

__global__ __launch_bounds__(256) void repack_batched_kernel(RepackArgs a) {
    const int total = a.end[a.n - 1];
    for (int gidx = blockIdx.x * 256 + threadIdx.x; gidx < total; gidx += gridDim.x * 256) {
        int q = 0;
        while (gidx >= a.end[q]) ++q;
        const int idx = gidx - (q ? a.end[q - 1] : 0);
        const int Cout = a.Cout[q], Cin = a.Cin[q], stride = a.stride[q], pad = a.pad[q];
        const int tlog = (stride == 2) ? 1 : 2, tpd = 1 << tlog;
        const int kc = Cout * tpd * tpd;
        const int ci = idx % Cin;
        const int rest = idx / Cin;
        const int k = rest % kc, cls = rest / kc;
        const int ph = cls / stride, pw = cls % stride;
        const int kh0 = (ph + pad) % stride, kw0 = (pw + pad) % stride;
        const int co = k >> (2 * tlog), aa = (k >> tlog) & (tpd - 1), bb = k & (tpd - 1);
        a.wr[q][idx] = a.w[q][((co * Cin + ci) * 4 + kh0 + stride * aa) * 4 + kw0 + stride * bb];
    }
}

// geometry of the dgrad-form launch behind a Conv2d data gradient (transposed = 0) or a ConvTranspose2d forward (1)
static inline bool repack_geom(int transposed, int B, int Cin, int H, int W, int Cout, int stride, int pad, ConvGeom *g) {
    if (transposed) {
        const int OH = (H - 1) * stride - 2 * pad + 4, OW = (W - 1) * stride - 2 * pad + 4;
        if (!conv_args_ok(B, Cout, OH, OW, Cin, stride, pad)) return false;
        *g = make_geom(B, Cout, OH, OW, Cin, stride, pad);
        return g->OH == H && g->OW == W;
    }
    if (!conv_args_ok(B, Cin, H, W, Cout, stride, pad)) return false;
    *g = make_geom(B, Cin, H, W, Cout, stride, pad);
    return true;
}

MVAE_EXPORT size_t mvae_conv_k4_repack_floats(int transposed, const float *w, int B, int Cin, int H, int W, int Cout,
                                              int stride, int pad) {
    ConvGeom g;
    if (!w || B <= 0 || !repack_geom(transposed, B, Cin, H, W, Cout, stride, pad, &g)) return 0;
    if (conv_dgrad_small_ok(g) || conv_dgrad_s1_ok(g, w)) return 0;       // direct kernels read w itself
    return dgrad_ws_floats(g);
}

MVAE_EXPORT int mvae_conv_k4_repack_batched(const mvae_repack_item *items, int n_items, mvae_stream_t stream) {
    if (!items || n_items < 1 || n_items > REPACK_MAX) return MVAE_ERR_ARG;
    RepackArgs a;
    a.n = n_items;
    int total = 0;
    for (int q = 0; q < n_items; ++q) {
        const mvae_repack_item &s = items[q];
        if (!s.w || !s.wr || s.Cin < 1 || s.Cout < 1 || (s.stride != 1 && s.stride != 2) || s.pad < 0) return MVAE_ERR_ARG;
        a.w[q] = s.w; a.wr[q] = s.wr;
        a.Cout[q] = s.transposed ? s.Cin : s.Cout;      // of the underlying w[co][ci][4][4] view
        a.Cin[q] = s.transposed ? s.Cout : s.Cin;
        a.stride[q] = s.stride; a.pad[q] = s.pad;
        total += s.Cin * s.Cout * 16;
        a.end[q] = total;
    }
    int blocks = (total + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(repack_batched_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    return mvae_launch_status();
}

MVAE_EXPORT int mvae_convT2d_k4_fwd(const float *x, const float *w, float *pre, float *act, int B, int Cin,
                                    int H, int W, int Cout, int stride, int pad, void *ws, size_t ws_bytes,
                                    mvae_stream_t stream) {
    ConvGeom g;
    if (!x || (!w && !ws) || (!pre && !act) || B <= 0 || !convT_geom(B, Cin, H, W, Cout, stride, pad, &g)) return MVAE_ERR_ARG;
    return conv_dgrad_impl(x, w, pre, act, nullptr, g, ws, ws_bytes, (hipStream_t)stream);
}

MVAE_EXPORT size_t mvae_convT2d_k4_stats_tiles(int B, int Cin, int H, int W, int Cout, int stride, int pad) {
    ConvGeom g;
    if (B <= 0 || !convT_geom(B, Cin, H, W, Cout, stride, pad, &g)) return 0;
    return (size_t)conv_dgrad_stats_tiles(g);
}

MVAE_EXPORT int mvae_convT2d_k4_fwd_stats(const float *x, const float *w, float *part, size_t part_floats, int B, int Cin,
                                          int H, int W, int Cout, int stride, int pad, void *ws, size_t ws_bytes,
                                          mvae_stream_t stream) {
    ConvGeom g;
    if (!x || (!w && !ws) || !part || B <= 0 || !convT_geom(B, Cin, H, W, Cout, stride, pad, &g)) return MVAE_ERR_ARG;
    const long tiles = conv_dgrad_stats_tiles(g);
    if (tiles <= 0 || part_floats < (size_t)tiles * Cout * 2) return MVAE_ERR_ARG;
    return conv_dgrad_stats_impl(x, w, part, g, ws, ws_bytes, (hipStream_t)stream);
}

MVAE_EXPORT int mvae_convT2d_k4_dgrad(const float *dy, const float *w, float *dx, const float *pre_in, int B,
                                      int Cin, int H, int W, int Cout, int stride, int pad,
                                      mvae_stream_t stream) {
    ConvGeom g;
    if (!dy || !w || !dx || B <= 0 || !convT_geom(B, Cin, H, W, Cout, stride, pad, &g)) return MVAE_ERR_ARG;
    return conv_fwd_impl(dy, w, dx, nullptr, pre_in, g, (hipStream_t)stream);
}

MVAE_EXPORT int mvae_convT2d_k4_wgrad(const float *dy, const float *x, float *dw, int B, int Cin, int H, int W,
                                      int Cout, int stride, int pad, int flags, void *ws, size_t ws_bytes,
                                      mvae_stream_t stream) {
    ConvGeom g;
    if (!dy || !x || !dw || B <= 0 || !convT_geom(B, Cin, H, W, Cout, stride, pad, &g)) return MVAE_ERR_ARG;
    // mirrored conv: "dy" operand is the transpose's input x, "x" operand is the transpose's dy
    return conv_wgrad_impl(x, dy, dw, g, flags, ws, ws_bytes, (hipStream_t)stream);
}
