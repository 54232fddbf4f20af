// gemm_core.h -- the dense contractions of the MVAE train step on fp32 MFMA (gfx950).
//
// One LDS-tiled kernel template, `igemm_kernel`, computes D[i][j] = sum_k P(i,k) * Q(k,j)
// with v_mfma_f32_32x32x2_f32 (exact fp32, 157 TFLOP/s peak on MI355X).  What differs between
// Linear fwd/dgrad/wgrad, Conv2d 4x4 fwd/dgrad/wgrad and ConvTranspose2d is only
//   * how the P and Q tiles are fetched from HBM (loader functors: row-major vector loads,
//     implicit-im2col gathers, parity-decomposed transposed-conv gathers), and
//   * what the epilogue does with the accumulator tile (bias / swish / dropout mask / swish'
//     of the producer's pre-activation / accumulate / NCHW scatter).
// The j axis is the lane axis of the MFMA result (32 consecutive j per store instruction),
// so each op maps its memory-contiguous output axis to j.
//
// Tiling: see igemm_kernel (k-groups of WGM x WGN waves, WM x WN MFMA tiles of 32x32 each; BK = 32 or 64).
// Main loops issue MFMAs and memory instructions only: on fp32 MFMA a vector-ALU instruction costs matrix
// throughput (tools/mfma_peak.hip), so full k-tiles are fetched with raw buffer loads at per-thread CONSTANT
// offsets (an out-of-range offset zero-fills what must read as zero), the k-step moves the scalar base, and the
// next tile's loads / LDS stores are sliced into the shadows of the current tile's MFMA groups ("buffer loads for
// the main loops" and the interleaved loops below).  Register staging: one tile ahead for 2- and 4-tile waves,
// two for 1-tile waves; two LDS buffers; ONE barrier per k-step.  Row operands with k contiguous are staged
// row-major ([row][k + 4], ds_read_b128 fragments), the others k-major ([k][tile + 4], ds_read_b32).
//
// Long reductions with a small output (weight gradients over the batch; Linear layers with
// 6400 inputs or outputs) are split across blockIdx.z into a caller-provided workspace and
// finished by `finish_kernel`, which sums the splits in a fixed order and applies the same
// epilogue functor: deterministic, no atomics.  Grouped launches (G same-shaped problems, the group
// index on the grid's class slot) keep one partial region per class.
//
// Three shapes do not fit the template and have their own kernels below: the stride-1 transposed conv
// (convT_s1_kernel: dense GEMM + col2im), the <= 4-output-channel transposed conv (convT_small_kernel)
// and the weight gradient of the <= 4-input-channel conv (wgrad_smallcin_kernel).
#pragma once
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Tuning overrides exist only in the -DMVAE_TUNING build (libmvae_hip_tuning.so, used by tools/gemm_bench.py);
// the product library has no mutable global state: MVAE_TUNE(x) folds to 0.
#ifndef MVAE_STAGGER
#define MVAE_STAGGER 0          // > 0: s_sleep units (64 cycles) per CU slot at kernel start (experiment)
#endif
#ifndef MVAE_INTERLEAVE
#define MVAE_INTERLEAVE 1       // 1: next-tile loads / stores sliced into the MFMA groups' shadows (see igemm_kernel)
#endif
#ifndef MVAE_KO_EPI
#define MVAE_KO_EPI 0           // knock-out experiment (results are wrong): 1 = the conv / Linear epilogues store nothing
#endif
// ... unless the value equals this constant, i.e. never -- but the test keeps the accumulators (and with them every MFMA) alive:
// with an unconditional return the compiler deleted the matrix instructions and the build "ran at the matrix floor" (SQ_INSTS_MFMA = 0)
#define MVAE_KO_MAGIC 1.2345678e-31f
#ifndef MVAE_KO
#define MVAE_KO 0               // knock-out experiments on the interleaved loop (results are wrong): 1 no global loads,
#endif                          // 2 + no LDS stores, 3 + no barrier, 4 + no fragment reads (MFMAs only)
#ifndef MVAE_PAIR_NEIGH
#define MVAE_PAIR_NEIGH 1       // pair-storing transposed-conv launches: the two class PAIRS of a j tile on neighbouring blocks of one XCD
                                // (2 items per block) instead of four consecutive items of one block: the second pair re-reads the
                                // tile's input microseconds -- not ~45 us -- after the first, and the 128-byte output lines the pairs
                                // share meet in the L2.  profiles/r05_convT_class_ab.txt (x2 interleaved, outputs identical to the last
                                // digit): FashionMNIST's dominant launch -2.8 %, its step -0.5 / -0.7 %, CelebA-19 -0.2 %, CelebA -0.1 ... -0.4 %.
                                // (Non-temporal stores for the output, measured beside it, LOSE: that launch +4 %.)  0: A/B
#endif
#ifndef MVAE_MULTI_MINBLOCKS
#define MVAE_MULTI_MINBLOCKS 1024   // a multi-item launch keeps at least this many column blocks (4 per CU)
#endif
#ifndef MVAE_WGRAD_TARGET
#define MVAE_WGRAD_TARGET 512   // blocks a split conv weight gradient aims for (two per CU); fewer = fewer partial slabs (A/B builds)
#endif
#ifndef MVAE_WGRAD_TILE
#define MVAE_WGRAD_TILE 128
#endif
#ifndef MVAE_XCD_ROWS
#define MVAE_XCD_ROWS 0         // 1: Linear launches map XCDs to ROW BANDS of the output (experiment; see igemm_kernel)
#endif
#ifndef MVAE_SETPRIO
#define MVAE_SETPRIO 0          // 1: raise the wave priority around each MFMA group (measured: see DESIGN.md)
#endif
#ifdef MVAE_TUNING
struct MvaeTune { int wm, wn, splits, kw, small_off, small_waves; long split_target; int knockout; };
extern MvaeTune g_mvae_tune;      // defined in linear.hip
#define MVAE_TUNE(f) (g_mvae_tune.f)
#else
#define MVAE_TUNE(f) 0
#endif
#ifndef MVAE_FINISH_PREFETCH
#define MVAE_FINISH_PREFETCH 0
#endif
#ifndef MVAE_PHASED_PRELOAD
#define MVAE_PHASED_PRELOAD 2    // k-grouped blocks, two tiles really in flight (see the phased loop): 1 = every wave issues the tile loads
                                 // (out of range for the MFMA-only waves), 2 = those waves run a load-free copy of the loop, 0 = rounds 1-4.
                                 // Round 4 measured 1 once (+2.4 %) and never ran 2.  Round 5, x3 interleaved on one box
                                 // (profiles/r05_mnist_switches_ab.txt): 0: 0.2874 / 0.2897 / 0.2854 ms, 1: 0.2918 / 0.2928 / 0.2896,
                                 // 2: 0.2780 / 0.2837 / 0.2837 -- the movers' copy has no control-flow merge between its loads and its
                                 // stores, the compiler's wait counts are the two-tiles-in-flight ones, and the MFMA-only waves issue
                                 // no dummy loads: MNIST -2 %; 111 parity tests green on that build.  Adopted.
#endif
#ifndef MVAE_PHASED_DEPTH
#define MVAE_PHASED_DEPTH 2      // register sets (k-tiles in flight) of the movers in the k-grouped layouts' loop: 2, or 4 (A/B).  FOUR
                                 // tiles in flight make the launch itself faster (MNIST's 1024 x 512 x 512: 13.2 -> 12.4 us by rocprof,
                                 // single stream) and the STEP 9 % slower (0.3067-0.3077 vs 0.2808-0.2812 ms, x3 interleaved,
                                 // profiles/r05_mnist_switches_ab.txt): the movers then hold 171 registers, a 512-thread block fills
                                 // half of every SIMD's register file, and the two chain kernels the step's two streams run side
                                 // by side no longer share a CU.  Two tiles: 116-122 registers, two blocks per CU.
#endif
#ifndef MVAE_CHAIN_PRIO
#define MVAE_CHAIN_PRIO 0        // 1-3: the k-grouped (small-layout) GEMMs -- the launches of MNIST's data-gradient chains -- raise their
                                 // wave priority for their whole run.  Measured x3 (profiles/r05_wgrad_ab.txt): nothing -- beside a
                                 // weight-gradient batch the two launches share the MFMA pipes whatever their priority.  Off.
#endif
#ifndef MVAE_EP_BUFFER
#define MVAE_EP_BUFFER 1         // NCHW tile epilogues through buffer stores: per-lane column offset + SCALAR row offset, no 64-bit
                                 // address arithmetic per element (0: the round-4 pointer form)
#endif
#ifndef MVAE_EPI_BATCH
#define MVAE_EPI_BATCH 0         // tile epilogues: the operands of eight outputs fetched together.  Off: with 3-5 blocks per CU the other
                                 // blocks' matrix work already covers a block's epilogue -- CelebA +0.6 %, FashionMNIST +0.1 % (r04_epilogue_ab.txt)
#endif
#ifndef MVAE_EPI_PREFETCH_ROWRED
#define MVAE_EPI_PREFETCH_ROWRED 1     // ... also for the loss-folding epilogues (bias, target / label, row coefficient)
#endif
#ifndef MVAE_EPI_PREFETCH
#define MVAE_EPI_PREFETCH 1      // small layouts: the epilogue's operands fetched ahead of the main loop (0: rounds 1-3)
#endif

// raw buffer loads, declared on the LLVM intrinsics (see "buffer loads for the main loops" below)
typedef int i32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
__device__ f32x4_t llvm_raw_buffer_load_f32x4(i32x4_t rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4f32");
__device__ float llvm_raw_buffer_load_f32(i32x4_t rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.f32");
typedef float f32x2_ep_t __attribute__((ext_vector_type(2)));
__device__ f32x2_ep_t llvm_raw_buffer_load_f32x2_ep(i32x4_t rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v2f32");
__device__ void llvm_raw_buffer_store_f32(float v, i32x4_t rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.f32");
__device__ void llvm_raw_buffer_store_f32x2(f32x2_ep_t v, i32x4_t rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.v2f32");

namespace {

constexpr int BK = 32;
constexpr int LPAD = 4;
constexpr int NTHREADS = 256;

// ------------------------------------------------------------------------------------------
// loaders.  init(tile0, t) once; load(k0, kend, t, regs) global -> registers;
// store(lds, t, regs) registers -> the LDS image [BK][TILE + LPAD].
// ------------------------------------------------------------------------------------------

// All loads below are UNCONDITIONAL from clamped (always legal) addresses and the out-of-range lanes
// are zeroed by MULTIPLYING with a 0/1 mask.  A load under a branch -- and hipcc turns
// `ok ? load : 0` and even `load; if (!ok) x = 0` into one -- makes it drain the memory queue
// (s_waitcnt vmcnt(0)) after every load, which serialises the two-tile prefetch; the multiply keeps
// the load in straight-line code (x * 0.f cannot be folded without fast-math).
__device__ __forceinline__ float mask0(bool ok) { return ok ? 1.f : 0.f; }

// fragment of a k-major LDS image: elements (k0 .. k0+3, row) -- four ds_read_b32, lanes along `row`
template <int PITCH_>
__device__ __forceinline__ float4 frag_kmajor(float (*L)[PITCH_], int k0, int row) {
    return make_float4(L[k0][row], L[k0 + 1][row], L[k0 + 2][row], L[k0 + 3][row]);
}

// ---- buffer loads for the main loops ----
// In an fp32-MFMA kernel every VALU instruction costs matrix throughput (the fp32 matrix instruction runs on the
// same fp32 FMA lanes: tools/mfma_peak measures 145 -> 95 TFLOP/s with TWO VALU instructions per MFMA at one wave
// per SIMD, ~4.5 cycles per VALU instruction at four).  The loaders therefore fetch FULL k-tiles with raw buffer
// loads: the per-thread byte offsets are constants computed once in init(), the k-step moves the (scalar) base
// of the buffer resource, and an element that must read as zero (a row beyond the matrix, a tap outside the
// image) carries the offset BUF_OOB, which is beyond num_records -- the hardware returns 0.  No address
// arithmetic, no bounds test and no mask multiply is left on the vector ALU.
// (The loads are declared on the LLVM intrinsics: hipcc 7.2's __builtin_amdgcn_raw_buffer_load_b128 emits a
// ONE-dword load.  The block's base address is pinned to scalar registers with readfirstlane in init(); left to
// itself the compiler kept it in vector registers and wrapped every load in a waterfall loop.)
constexpr int BUF_OOB = (int)0x80000000u;
struct BufBase { unsigned lo, hi; };                  // a block-uniform address, held in scalar registers
__device__ __forceinline__ BufBase buf_base(const float *p) {
    const unsigned long long a = (unsigned long long)p;
    BufBase b;
    b.lo = __builtin_amdgcn_readfirstlane((unsigned)a);
    b.hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return b;
}
// raw buffer (stride 0) at base + `floats`, num_records 2 GiB - 1: offsets are bytes relative to that address,
// which every loader keeps within the current tile's neighbourhood
__device__ __forceinline__ i32x4_t buf_rsrc(BufBase b, size_t floats) {
    const unsigned long long a = (((unsigned long long)b.hi << 32) | b.lo) + (unsigned long long)floats * 4ull;
    i32x4_t r;
    r.x = (int)(unsigned)a; r.y = (int)((unsigned)(a >> 32) & 0xffffu); r.z = 0x7fffffff; r.w = 0x00020000;
    return r;
}
__device__ __forceinline__ float buf_load1(i32x4_t r, int voff) { return llvm_raw_buffer_load_f32(r, voff, 0, 0); }
__device__ __forceinline__ float4 buf_load4(i32x4_t r, int voff) {
    const f32x4_t u = llvm_raw_buffer_load_f32x4(r, voff, 0, 0);
    return make_float4(u.x, u.y, u.z, u.w);
}
// slice `part` of `nparts` of an NV-element per-thread transfer
#define MVAE_IN_PART(v, NV_, part, nparts) ((v) >= (part) * (NV_) / (nparts) && (v) < ((part) + 1) * (NV_) / (nparts))

// S[r * ld + k]: reduction axis contiguous (x and w of Linear fwd, dy of dgrad, conv weights).
// VEC: base 16-byte aligned, ld % 4 == 0 and Klen % 4 == 0 (float4 loads never straddle the end).
// BKV: k-tile depth (32 for the conv forms; the small Linear GEMMs use 64 -- half the barriers).
template <int TILE_, bool VEC, int BKV_ = BK>
struct LdRowsKT {
    static constexpr int TILE = TILE_, BKV = BKV_;
    static constexpr int NV = TILE * BKV / 4 / NTHREADS;
    static_assert(NV >= 1, "tile too small for 256 mover threads");
    struct Regs { float4 v[NV]; float m[VEC ? NV : 4 * NV]; };   // raw data + 0/1 masks (applied when staged)
    const float *src; int ld; int R; int Klen;
    size_t cls_stride = 0;                            // per-class (group) source offset
    size_t cls_off = 0;                               // = class * cls_stride, set by init()
    int r0;
    int voff[VEC ? NV : 1];                           // buffer path: byte offset of float4 v from row r0, k0 (or BUF_OOB)
    BufBase blk;                                      //              address of (row r0, k = 0)
    static constexpr bool fast = true;
    __device__ void begin(int, int) {}
    // a thread that fetches nothing (the MFMA-only waves of a k-grouped block): every buffer load reads out of range
    __device__ void disable() {
#pragma unroll
        for (int v = 0; v < (VEC ? NV : 1); ++v) voff[v] = BUF_OOB;
    }
    __device__ void init(int tile0, int t, int cls) {
        r0 = tile0; cls_off = (size_t)cls * cls_stride;
        if (VEC) {
            blk = buf_base(src + cls_off + (size_t)r0 * ld);
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int f = t + NTHREADS * v;
                const int r = f / (BKV / 4), kc = (f % (BKV / 4)) * 4;
                voff[v] = (r0 + r < R) ? (r * ld + kc) * 4 : BUF_OOB;
            }
        }
    }
    // general path (partial k-tiles, unaligned operands): clamped addresses + 0/1 masks
    __device__ void load(int k0, int kend, int t, Regs &rg) const {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int f = t + NTHREADS * v;
            const int r = r0 + f / (BKV / 4), k = k0 + (f % (BKV / 4)) * 4;
            const float *row = src + cls_off + (size_t)min(r, R - 1) * ld;
            float4 x;
            if (VEC) {
                x = *reinterpret_cast<const float4 *>(row + min(k, Klen - 4));
                rg.m[v] = mask0(r < R && k < kend);
            } else {
                x.x = row[min(k, Klen - 1)];     x.y = row[min(k + 1, Klen - 1)];
                x.z = row[min(k + 2, Klen - 1)]; x.w = row[min(k + 3, Klen - 1)];
                const bool rin = r < R;
                rg.m[4 * v + 0] = mask0(rin && k < kend);     rg.m[4 * v + 1] = mask0(rin && k + 1 < kend);
                rg.m[4 * v + 2] = mask0(rin && k + 2 < kend); rg.m[4 * v + 3] = mask0(rin && k + 3 < kend);
            }
            rg.v[v] = x;
        }
    }
    // full k-tile, slice `part` of `nparts` (the interleaved main loop issues one slice per MFMA group)
    // A partial LAST k-tile is covered too (TAIL): the float4s at or beyond kend take the out-of-range offset
    // (one compare + NV selects per k-step; a thread's k column is the same for all its float4s).
    static constexpr bool PARTS = VEC, TAIL = VEC;
    __device__ __forceinline__ void load_part(int k0, int kend, int t, Regs &rg, int part, int nparts) const {
        const i32x4_t rs = buf_rsrc(blk, (size_t)k0);
        const bool cut = (t % (BKV / 4)) * 4 >= kend - k0;
#pragma unroll
        for (int v = 0; v < NV; ++v)
            if (MVAE_IN_PART(v, NV, part, nparts)) rg.v[v] = buf_load4(rs, cut ? BUF_OOB : voff[VEC ? v : 0]);
    }
    // LDS image [TILE][BKV + 4]: the tile as it lies in memory (k contiguous), float4 stores, rows
    // 16-byte aligned.  The row pitch 4 * odd makes the ds_read_b128 fragment reads -- lane (row, 4 k's)
    // -- conflict-free (MI355X_MICROARCH.md, LDS: b128 lane groups of 16 rows on 64 banks).  The first
    // version stored this operand transposed ([k][row], four 4-way-conflicting ds_write_b32 per float4):
    // the LDS array, shared by the whole CU, was as busy as the MFMA pipe.
    static constexpr bool RMAJOR = true;
    static constexpr int ROWS = TILE, PITCH = BKV + LPAD;
    typedef float (*Tile)[PITCH];
    __device__ void store(Tile L, int t, const Regs &rg) const {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int f = t + NTHREADS * v;
            const int r = f / (BKV / 4), kc = (f % (BKV / 4)) * 4;
            const float m0 = rg.m[VEC ? v : 4 * v], m1 = rg.m[VEC ? v : 4 * v + 1];
            const float m2 = rg.m[VEC ? v : 4 * v + 2], m3 = rg.m[VEC ? v : 4 * v + 3];
            *reinterpret_cast<float4 *>(&L[r][kc]) =
                make_float4(rg.v[v].x * m0, rg.v[v].y * m1, rg.v[v].z * m2, rg.v[v].w * m3);
        }
    }
    __device__ __forceinline__ void store_part(Tile L, int t, const Regs &rg, int part, int nparts) const {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            if (!MVAE_IN_PART(v, NV, part, nparts)) continue;
            const int f = t + NTHREADS * v;
            *reinterpret_cast<float4 *>(&L[f / (BKV / 4)][(f % (BKV / 4)) * 4]) = rg.v[v];
        }
    }
    // the 4 consecutive k's starting at k0 of tile row `row`: one ds_read_b128
    static __device__ __forceinline__ float4 frag(Tile L, int k0, int row) {
        return *reinterpret_cast<const float4 *>(&L[row][k0]);
    }
};
template <int T> using LdRowsK = LdRowsKT<T, true>;
template <int T> using LdRowsKS = LdRowsKT<T, false>;
template <int T> using LdRowsK64 = LdRowsKT<T, true, 64>;
template <int T> using LdRowsKS64 = LdRowsKT<T, false, 64>;

// S[k * ld + r]: non-reduced axis contiguous (w of dgrad, dy and x of wgrad, repacked conv weights).
// VEC: base 16-byte aligned, ld % 4 == 0 and R % 4 == 0.
template <int TILE_, bool VEC, int BKV_ = BK>
struct LdRowsMNT {
    static constexpr int TILE = TILE_, BKV = BKV_;
    static constexpr int NV = TILE * BKV / 4 / NTHREADS;
    static constexpr int V4 = TILE / 4;     // float4 per k row
    static_assert(NV >= 1, "tile too small for 256 mover threads");
    struct Regs { float4 v[NV]; float m[VEC ? NV : 4 * NV]; };
    const float *src; int ld; int R; int Klen; size_t cls_stride;   // cls_stride: per-class source offset
    size_t cls_off = 0;                               // = class * cls_stride, set by init()
    int r0;
    int voff[VEC ? NV : 1];                           // buffer path: byte offset of float4 v from (k0, r0) (or BUF_OOB)
    BufBase blk;                                      //              address of (k = 0, r0)
    static constexpr bool fast = true;
    __device__ void begin(int, int) {}
    // a thread that fetches nothing (the MFMA-only waves of a k-grouped block): every buffer load reads out of range
    __device__ void disable() {
#pragma unroll
        for (int v = 0; v < (VEC ? NV : 1); ++v) voff[v] = BUF_OOB;
    }
    __device__ void init(int tile0, int t, int cls) {
        r0 = tile0; cls_off = (size_t)cls * cls_stride;
        if (VEC) {
            blk = buf_base(src + cls_off + r0);
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int f = t + NTHREADS * v;
                const int kl = f / V4, r = (f % V4) * 4;
                voff[v] = (r0 + r < R) ? (kl * ld + r) * 4 : BUF_OOB;
            }
        }
    }
    __device__ void load(int k0, int kend, int t, Regs &rg) const {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int f = t + NTHREADS * v;
            const int k = k0 + f / V4, r = r0 + (f % V4) * 4;
            const float *row = src + cls_off + (size_t)min(k, Klen - 1) * ld;
            float4 x;
            if (VEC) {
                x = *reinterpret_cast<const float4 *>(row + min(r, R - 4));
                rg.m[v] = mask0(k < kend && r < R);
            } else {
                x.x = row[min(r, R - 1)];     x.y = row[min(r + 1, R - 1)];
                x.z = row[min(r + 2, R - 1)]; x.w = row[min(r + 3, R - 1)];
                const bool kin = k < kend;
                rg.m[4 * v + 0] = mask0(kin && r < R);     rg.m[4 * v + 1] = mask0(kin && r + 1 < R);
                rg.m[4 * v + 2] = mask0(kin && r + 2 < R); rg.m[4 * v + 3] = mask0(kin && r + 3 < R);
            }
            rg.v[v] = x;
        }
    }
    static constexpr bool PARTS = VEC, TAIL = VEC;    // partial last k-tile: rows k >= kend read as zero
    __device__ __forceinline__ void load_part(int k0, int kend, int t, Regs &rg, int part, int nparts) const {
        const i32x4_t rs = buf_rsrc(blk, (size_t)k0 * ld);
        const int left = kend - k0 - t / V4;          // float4 v sits in tile row t / V4 + v * (NTHREADS / V4)
#pragma unroll
        for (int v = 0; v < NV; ++v)
            if (MVAE_IN_PART(v, NV, part, nparts))
                rg.v[v] = buf_load4(rs, (v * (NTHREADS / V4) >= left) ? BUF_OOB : voff[VEC ? v : 0]);
    }
    // LDS image [BKV][TILE + 4] (k-major: the non-reduced axis contiguous, as in memory)
    static constexpr bool RMAJOR = false;
    static constexpr int ROWS = BKV, PITCH = TILE + LPAD;
    typedef float (*Tile)[PITCH];
    __device__ void store(Tile L, int t, const Regs &rg) const {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int f = t + NTHREADS * v;
            const float4 x = rg.v[v];
            const float m0 = rg.m[VEC ? v : 4 * v], m1 = rg.m[VEC ? v : 4 * v + 1];
            const float m2 = rg.m[VEC ? v : 4 * v + 2], m3 = rg.m[VEC ? v : 4 * v + 3];
            *reinterpret_cast<float4 *>(&L[f / V4][(f % V4) * 4]) = make_float4(x.x * m0, x.y * m1, x.z * m2, x.w * m3);
        }
    }
    __device__ __forceinline__ void store_part(Tile L, int t, const Regs &rg, int part, int nparts) const {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            if (!MVAE_IN_PART(v, NV, part, nparts)) continue;
            const int f = t + NTHREADS * v;
            *reinterpret_cast<float4 *>(&L[f / V4][(f % V4) * 4]) = rg.v[v];
        }
    }
    static __device__ __forceinline__ float4 frag(Tile L, int k0, int row) { return frag_kmajor(L, k0, row); }
};
template <int T> using LdRowsMN = LdRowsMNT<T, true>;
template <int T> using LdRowsMNS = LdRowsMNT<T, false>;
template <int T> using LdRowsMN64 = LdRowsMNT<T, true, 64>;
template <int T> using LdRowsMNS64 = LdRowsMNT<T, false, 64>;

// ------------------------------------------------------------------------------------------
// epilogues:  col(j) prepares the lane's column, put(i, j, v) consumes one element.
// ------------------------------------------------------------------------------------------

// Row-major destination D[i * ld + j] with the Linear fusions.
struct EpRowMajor {
    static constexpr bool MULTI = false;      // multi-item blocks: conv forms only (set_class here is cumulative)
    static constexpr bool PAIR = false;
    static constexpr bool ROWRED = false;     // EpRowBce / EpRowCe below
    __device__ bool pair_ok() const { return false; }
    __device__ void put2(int, float, float) const {}
    float *out; float *act; int ld;           // out = raw / pre-activation result, act = swish(result)
    const float *bias;                        // per column j (Linear fwd)
    const float *dpre; int ldp;               // multiply by swish'(dpre[i][j])
    const float *mask; int ldm; float mask_scale;   // dropout keep-mask (fwd on act, bwd on the product)
    int I, J; int accumulate;
    size_t out_cs = 0, bias_cs = 0, dpre_cs = 0;    // grouped Linear: per-group offsets of out/act, bias, dpre
    __device__ void set_class(int cls) {
        if (out) out += (size_t)cls * out_cs;
        if (act) act += (size_t)cls * out_cs;
        if (bias) bias += (size_t)cls * bias_cs;
        if (dpre) dpre += (size_t)cls * dpre_cs;
    }
    __device__ bool col(int j) const { return j < J; }
    __device__ void put(int i, int j, float v) const {
        if (MVAE_KO_EPI && v != MVAE_KO_MAGIC) return;
        if (i >= I) return;
        if (bias) v += bias[j];
        float m = 1.f;
        if (mask) m = mask[(size_t)i * ldm + j] * mask_scale;
        if (dpre) v *= m * swish_grad_(dpre[(size_t)i * ldp + j]);
        const size_t idx = (size_t)i * ld + j;
        if (accumulate) v += out[idx];
        if (out) out[idx] = v;
        if (act) act[idx] = swishf_(v) * m;
    }
    // The same in two halves: fetch() what output (i, j) will need -- UNCONDITIONAL loads from always-legal addresses
    // (a null operand reads the destination instead; the value is not used) that a kernel issues long before its
    // epilogue -- and put_pre().  In put() every operand load sits under a block-uniform branch, hipcc waits for the
    // whole memory queue behind each, and an output costs one or two full memory latencies: four outputs per thread in
    // the small layouts' epilogue were 4-8 us of a 11-us launch (MNIST's 512-wide layers).
    struct Pre { float b, d, m; };
    static constexpr bool PREFETCH = true;
    __device__ Pre fetch(int i, int j) const {
        const int ic = min(i, I - 1), jc = min(j, J - 1);
        const float *safe = out ? out : act;
        const float *bp = bias ? bias + jc : safe;
        const float *dp = dpre ? dpre + (size_t)ic * ldp + jc : safe;
        const float *mp = mask ? mask + (size_t)ic * ldm + jc : safe;
        Pre p;
        p.b = *bp; p.d = *dp; p.m = *mp;
        return p;
    }
    __device__ void put_pre(int i, int j, float v, const Pre &p) const {
        if (MVAE_KO_EPI && v != MVAE_KO_MAGIC) return;
        if (i >= I) return;
        if (bias) v += p.b;
        float m = 1.f;
        if (mask) m = p.m * mask_scale;
        if (dpre) v *= m * swish_grad_(p.d);
        const size_t idx = (size_t)i * ld + j;
        if (accumulate) v += out[idx];
        if (out) out[idx] = v;
        if (act) act[idx] = swishf_(v) * m;
    }
};
template <class T, class = void> struct ld_can_disable : std::false_type {};
template <class T> struct ld_can_disable<T, std::void_t<decltype(std::declval<T &>().disable())>> : std::true_type {};
template <bool ON, class L> __device__ __forceinline__ void loader_disable(L &l) { if constexpr (ON) l.disable(); }
template <class T, class = void> struct ep_prefetch : std::false_type {};
template <class T> struct ep_prefetch<T, std::void_t<decltype(T::PREFETCH)>> : std::integral_constant<bool, T::PREFETCH> {};
template <class T, bool = ep_prefetch<T>::value> struct ep_pre { struct type {}; };
template <class T> struct ep_pre<T, true> { typedef typename T::Pre type; };

// Linear forward whose only consumer is a reconstruction term of the ELBO: the logits never reach memory, the
// epilogue emits d loss / d logits (the backward's input) and the loss itself.  ROWRED epilogues are driven through
// put_row(i, j, v), called TOGETHER by the 32 lanes of a half wavefront that hold 32 consecutive columns
// j = 32 * q + (lane & 31) of ONE row i (the kernel guarantees it; out-of-range rows / columns take part with
// i >= I or j >= J), so a row reduction is five cross-lane steps in a fixed order.  Split reductions never reach
// these (the host plans them without a split); put() exists for the finish kernels' instantiation only.
//
// Bernoulli term (mnist/train.py:47-49,62-74 on mnist/model.py:104's last Linear): part[i][j / 32] = the sum of the
// 32 columns' terms, summed over j / 32 by the ELBO launch (a group of B rows is B * nparts consecutive floats).
struct EpRowBce {
    static constexpr bool MULTI = false, PAIR = false, ROWRED = true;
    __device__ bool pair_ok() const { return false; }
    __device__ void put2(int, float, float) const {}
    float *dlogits; int ld;                          // d loss / d logits [I, ld]
    float *logits;                                   // optional: the logits as well (tests)
    const float *bias;
    const float *target; int t_rs, target_rows;      // target row of output row i: i % target_rows
    const float *drow; int rows_per_group;           // d loss / d rowsum of group i / rows_per_group
    float *part; int nparts;                         // [I, nparts], nparts = ceil(J / 32)
    int I, J;
    __device__ void set_class(int) const {}
    __device__ bool col(int j) const { return j < J; }
    __device__ void put(int, int, float) const {}
    __device__ void put_row(int i, int j, float v) const {
        float l = 0.f;
        if (i < I && j < J) {
            if (bias) v += bias[j];
            const float tg = target[(size_t)(i % target_rows) * t_rs + j];
            l = bce_elem(v, tg);
            const size_t idx = (size_t)i * ld + j;
            dlogits[idx] = drow[i / rows_per_group] * 1.f * bce_grad(v, tg);     // bce_row_block_kernel's product, bit for bit
            if (logits) logits[idx] = v;
        }
        l = half_wave_sum(l);
        if ((threadIdx.x & 31) == 0 && i < I && j < J) part[(size_t)i * nparts + (j >> 5)] = l;
    }
    // put_row in two halves (see EpRowMajor::fetch): bias, target and the row's coefficient fetched ahead of the main loop
    struct Pre { float b, tg, dr; };
    static constexpr bool PREFETCH = true;
    __device__ Pre fetch(int i, int j) const {
        const int ic = min(i, I - 1), jc = min(j, J - 1);
        Pre p;
        p.tg = target[(size_t)(ic % target_rows) * t_rs + jc];
        p.b = *(bias ? bias + jc : target);
        p.dr = drow[ic / rows_per_group];
        return p;
    }
    __device__ void put_row_pre(int i, int j, float v, const Pre &p) const {
        float l = 0.f;
        if (i < I && j < J) {
            if (bias) v += p.b;
            l = bce_elem(v, p.tg);
            const size_t idx = (size_t)i * ld + j;
            dlogits[idx] = p.dr * 1.f * bce_grad(v, p.tg);
            if (logits) logits[idx] = v;
        }
        l = half_wave_sum(l);
        if ((threadIdx.x & 31) == 0 && i < I && j < J) part[(size_t)i * nparts + (j >> 5)] = l;
    }
};

// Categorical term (mnist/train.py:52,77-94 on mnist/model.py:146's last Linear), J <= 32 classes: a half wavefront
// holds the whole row.  row[i] = -log_softmax(x + 1e-6)[label], dlogits = drow * (softmax(x + 1e-6) - onehot);
// a label outside 0..J-1 makes both NaN (ce_kernel's rule).
struct EpRowCe {
    static constexpr bool MULTI = false, PAIR = false, ROWRED = true;
    __device__ bool pair_ok() const { return false; }
    __device__ void put2(int, float, float) const {}
    float *dlogits; int ld;
    float *logits;
    const float *bias;
    const int64_t *label; int label_rows;            // label of output row i: label[i % label_rows]
    const float *drow; int rows_per_group;
    float *row;                                      // [I]
    int I, J;
    __device__ void set_class(int) const {}
    __device__ bool col(int j) const { return j < J; }
    __device__ void put(int, int, float) const {}
    __device__ void put_row(int i, int j, float v) const {
        const bool in = i < I && j < J;
        if (in && bias) v += bias[j];
        const float x = in ? v + 1e-6f : -INFINITY;
        const float mx = half_wave_max(x);
        const float se = half_wave_sum(in ? expf(x - mx) : 0.f);
        if (!in) return;
        const float lse = logf(se) + mx;
        const int64_t yraw = label[i % label_rows];
        const bool bad = yraw < 0 || yraw >= J;
        const int y = bad ? 0 : (int)yraw;
        if (j == y) row[i] = bad ? NAN : -(x - lse);
        const float dr = bad ? NAN : drow[i / rows_per_group];
        const size_t idx = (size_t)i * ld + j;
        dlogits[idx] = dr * (expf(x - lse) - (j == y ? 1.f : 0.f));
        if (logits) logits[idx] = v;
    }
    struct Pre { float b, dr; int64_t y; };
    static constexpr bool PREFETCH = true;
    __device__ Pre fetch(int i, int j) const {
        const int ic = min(i, I - 1), jc = min(j, J - 1);
        Pre p;
        p.y = label[ic % label_rows];
        p.b = *(bias ? bias + jc : drow);
        p.dr = drow[ic / rows_per_group];
        return p;
    }
    __device__ void put_row_pre(int i, int j, float v, const Pre &p) const {
        const bool in = i < I && j < J;
        if (in && bias) v += p.b;
        const float x = in ? v + 1e-6f : -INFINITY;
        const float mx = half_wave_max(x);
        const float se = half_wave_sum(in ? expf(x - mx) : 0.f);
        if (!in) return;
        const float lse = logf(se) + mx;
        const int64_t yraw = p.y;
        const bool bad = yraw < 0 || yraw >= J;
        const int y = bad ? 0 : (int)yraw;
        if (j == y) row[i] = bad ? NAN : -(x - lse);
        const float dr = bad ? NAN : p.dr;
        const size_t idx = (size_t)i * ld + j;
        dlogits[idx] = dr * (expf(x - lse) - (j == y ? 1.f : 0.f));
        if (logits) logits[idx] = v;
    }
};

// NCHW destination: i = channel, j = (n, row', col') of a (possibly strided) sub-lattice:
// address = (n * C + i) * HW + (row' * s + py) * Wfull + col' * s + px.
struct EpNCHW {
    static constexpr bool MULTI = true;
    static constexpr bool ROWRED = false;
    // PAIR: the two px classes of a stride-2 lattice row are neighbours in memory ((c*2 + 0), (c*2 + 1)) and the SAME
    // lane owns both (a lane is a column j = (n, r', c') of the class lattice): a multi-item block that walks the
    // classes (py, 0), (py, 1) back to back keeps the first one's accumulators and stores float2 -- full 256-byte
    // segments per half-wave instead of two passes of 4-byte stores at stride 8 (`pair` set by the host: stride 2,
    // even width, 8-byte aligned tensors, class-minor item order)
    static constexpr bool PAIR = false;       // EpNCHWPair below
    int pair = 0;                             // host-side choice between the two types
    __device__ bool pair_ok() const { return pair != 0; }
    __device__ void put2(int i, float v0, float v1) const {      // off was computed for px = 0
        if ((MVAE_KO_EPI && !(v0 == MVAE_KO_MAGIC && v1 == MVAE_KO_MAGIC)) || i >= C) return;
        const int idx = off + i * HW;
        if (dpre) {
            const float2 d = *reinterpret_cast<const float2 *>(dpre + idx);
            v0 *= swish_grad_(d.x); v1 *= swish_grad_(d.y);
        }
        if (out) *reinterpret_cast<float2 *>(out + idx) = make_float2(v0, v1);
        if (act) *reinterpret_cast<float2 *>(act + idx) = make_float2(swishf_(v0), swishf_(v1));
    }
    float *out; float *act; const float *dpre;
    int C, HW, Wfull, H2, W2, sy, py, px, J;
    int off;   // per-lane column offset, set by col()
    int lg_hw2 = -1, lg_w2 = -1;   // log2(H2 * W2), log2(W2) when both are powers of two (host), else -1: col() shifts instead of dividing
    __device__ void set_class(int cls) { if (sy > 1) { py = cls / sy; px = cls % sy; } }
    __device__ bool col(int j) {
        // no column: false for the pointer forms (`if (e.col(j)) e.put(...)` must not store through a stale `off`: ADVICE r5);
        // the buffer forms ignore the result -- the lane takes part and its accesses fall out of range
        if (j >= J) { voff = BUF_OOB; return false; }
        int n, rem, r, c;
        if (lg_w2 >= 0) {           // block-uniform
            n = j >> lg_hw2; rem = j & ((1 << lg_hw2) - 1);
            r = rem >> lg_w2; c = rem & ((1 << lg_w2) - 1);
        } else {
            const int hw2 = H2 * W2;
            n = j / hw2; rem = j - n * hw2;
            r = rem / W2; c = rem - r * W2;
        }
        const int in_img = (r * sy + py) * Wfull + c * sy + px;
        off = n * C * HW + in_img;
        // buffer form: bytes from the tile's first image (tile()); the rows of the upper half wavefront (4 further down:
        // the 32x32 MFMA result layout) ride the per-lane offset, so that a row's offset is the same for the whole wave
        voff = ((n - n0) * C * HW + in_img + ((threadIdx.x & 32) ? 4 * HW : 0)) * 4;
        return true;
    }
    __device__ void put(int i, int, float v) const {
        if ((MVAE_KO_EPI && v != MVAE_KO_MAGIC) || i >= C) return;
        const int idx = off + i * HW;
        if (dpre) v *= swish_grad_(dpre[idx]);
        if (out) out[idx] = v;
        if (act) act[idx] = swishf_(v);
    }
    // ---- the same through buffer instructions (tile epilogues of igemm_kernel).  put() forms a 64-bit address per stored
    //      element (v_lshl_add_u64 and friends: 3-4 vector instructions per access, 16-32 accesses per tile and wave --
    //      profiles/r04_celeba_sq_counters.txt read 2-4 VALU per MFMA over whole conv launches whose main loops issue 0.2-0.8,
    //      and on fp32 MFMA a vector instruction is matrix time).  Here an access is  descriptor(tile's first image) +
    //      per-lane byte offset (col(), once per column) + SCALAR row offset: no vector arithmetic at all.  A lane without
    //      a column carries BUF_OOB: its loads read 0, its stores are dropped -- no divergent branch around the tile either.
    static constexpr bool BUFFER = MVAE_EP_BUFFER != 0;
    int n0 = 0, voff = 0;
    i32x4_t r_out, r_act, r_dpre;
    __device__ void tile(int j0) {          // j0: the tile's first column (block-uniform)
        const int hw2 = H2 * W2;
        n0 = __builtin_amdgcn_readfirstlane(lg_w2 >= 0 ? j0 >> lg_hw2 : j0 / hw2);
        const size_t el = (size_t)n0 * C * HW;
        // (a null tensor borrows another's address: its branch is never taken)
        r_out = buf_rsrc(buf_base(out ? out : act), el);
        r_act = buf_rsrc(buf_base(act ? act : out), el);
        r_dpre = buf_rsrc(buf_base(dpre ? dpre : (out ? out : act)), el);
    }
    // rb: first row of the wave's 32-row fragment (wave-uniform); r: accumulator register 0 .. 15
    __device__ __forceinline__ void put_b(int rb, int r, float v) const {
        if (MVAE_KO_EPI && v != MVAE_KO_MAGIC) return;
        const int is = rb + (r & 3) + 8 * (r >> 2);             // row of the LOWER half wavefront
        int vo = voff;
        if (C & 7) {                                            // block-uniform; channel counts here are multiples of 8
            if (is + ((threadIdx.x & 32) ? 4 : 0) >= C) vo = BUF_OOB;
        } else if (is >= C) {
            return;                                             // wave-uniform: both halves are on the same side of C
        }
        const int so = is * HW * 4;
        if (dpre) v *= swish_grad_(llvm_raw_buffer_load_f32(r_dpre, vo, so, 0));
        if (out) llvm_raw_buffer_store_f32(v, r_out, vo, so, 0);
        if (act) llvm_raw_buffer_store_f32(swishf_(v), r_act, vo, so, 0);
    }
    __device__ __forceinline__ void put2_b(int rb, int r, float v0, float v1) const {      // col() ran for px = 0
        if (MVAE_KO_EPI && !(v0 == MVAE_KO_MAGIC && v1 == MVAE_KO_MAGIC)) return;
        const int is = rb + (r & 3) + 8 * (r >> 2);
        int vo = voff;
        if (C & 7) {
            if (is + ((threadIdx.x & 32) ? 4 : 0) >= C) vo = BUF_OOB;
        } else if (is >= C) {
            return;
        }
        const int so = is * HW * 4;
        if (dpre) {
            const f32x2_ep_t d = llvm_raw_buffer_load_f32x2_ep(r_dpre, vo, so, 0);
            v0 *= swish_grad_(d.x); v1 *= swish_grad_(d.y);
        }
        f32x2_ep_t o; o.x = v0; o.y = v1;
        if (out) llvm_raw_buffer_store_f32x2(o, r_out, vo, so, 0);
        if (act) { f32x2_ep_t a; a.x = swishf_(v0); a.y = swishf_(v1); llvm_raw_buffer_store_f32x2(a, r_act, vo, so, 0); }
    }
};

// EpNCHW whose multi-item blocks store the two px classes of a stride-2 lattice row TOGETHER (see EpNCHW::put2): a
// separate type, so that only launches the host found pairable carry the second accumulator set, and the kernel
// has no run-time choice between the two store forms (with one, hipcc spilled the loader state of the 32-row
// kernel to scratch: 630 scratch instructions, 466 of them among the MFMAs, and the launch ran 40 % longer).
struct EpNCHWPair : EpNCHW {
    static constexpr bool PAIR = true;
};

// Statistics-only destination: the last conv of a decoder pass that exists only for its BatchNorm running-statistics
// side effect (celeba19/train.py:278-283: 18 of the 21 model() calls decode an image nobody reads; SURVEY Appendix
// B-4).  NOTHING is stored: every lane adds its accumulators (v, v^2) up over the items of its multi-item block, and
// the block leaves one (mean, M2) record per output row over the n_items * BN columns it covered -- part[jt][C][2],
// jt = the block's column tile; mvae_bn_stats_merge combines the records of a group.  The 32-row transposed-conv
// launch it replaces spent as long on its 600 MB of stride-2 stores as on its matrix work, and the statistics sweep
// behind it read them all back (profiles/r04_celeba19_by_shape.txt: 825 + 240 us of a 7.1-ms step).  Two vector
// instructions per accumulator register per item; one cross-lane reduction per block.
struct EpStats {
    static constexpr bool MULTI = true, PAIR = false, ROWRED = false, STATS = true;
    __device__ bool pair_ok() const { return false; }
    __device__ void put2(int, float, float) const {}
    float *part; int C, J;
    __device__ void set_class(int) const {}
    __device__ bool col(int j) const { return j < J; }
    __device__ void put(int, int, float) const {}
};
template <class T, class = void> struct ep_buffer : std::false_type {};
template <class T> struct ep_buffer<T, std::void_t<decltype(T::BUFFER)>> : std::integral_constant<bool, T::BUFFER> {};
template <class T, class = void> struct ep_stats : std::false_type {};
template <class T> struct ep_stats<T, std::void_t<decltype(T::STATS)>> : std::integral_constant<bool, T::STATS> {};

// Where the raw partial tiles of a split reduction go (row-major [I][J] per split), plus the
// optional row sums of P (bias gradient of a Linear wgrad).
struct SplitSink {
    float *ws; size_t stride; int I, J;       // partial (split, i, j) at ws[split*stride + i*J + j]
    float *rowsum; size_t rowsum_stride; int rowsum_accumulate;   // (split, i) at rowsum[split*rowsum_stride + i]
    int ncls;                                 // parity classes (transposed conv) / groups (grouped Linear) folded into gridDim.x
    size_t rowsum_cls_stride;                 // grouped Linear wgrad: per-group offset of the bias gradient
    float *rowsum_final; int rowsum_final_accumulate;   // split launches: where the finish kernel puts the summed row sums
    size_t cls_region;                        // grouped + split: class c keeps its partials at ws + c * cls_region
    size_t rowsum_final_cls_stride;           //                  and its bias gradient at rowsum_final + c * this
    int xcd_map;                              // 1: re-map the launch order to XCD-local output sub-grids (see igemm_kernel)
    int tiles_j;                              // j tiles per class (set by the launcher)
    int items;                                // > 1: a block walks this many consecutive (class, j tile) items (see igemm_kernel)
    int cls_minor;                            // 1: launch order (j tile, class) instead of (class, j tile)
};

// finish kernels of a grouped launch: one grid slice per class
__device__ __forceinline__ void sink_select_class(SplitSink &sink, int cls) {
    sink.ws += (size_t)cls * sink.cls_region;
    if (sink.rowsum) sink.rowsum += (size_t)cls * sink.cls_region;
    if (sink.rowsum_final) sink.rowsum_final += (size_t)cls * sink.rowsum_final_cls_stride;
}

// the bias gradient of a split Linear wgrad: sum the per-split row sums in split order
__device__ __forceinline__ void finish_rowsum(const SplitSink &sink, int splits, int i) {
    float s = 0.f;
    for (int z = 0; z < splits; ++z) s += sink.rowsum[(size_t)z * sink.rowsum_stride + i];
    float *dst = sink.rowsum_final + i;
    if (sink.rowsum_final_accumulate) s += *dst;
    *dst = s;
}

// a Q loader says `static constexpr bool PAIRABLE = true` when its classes are the parity classes of a stride-2 lattice
template <class T, class = void> struct loader_pairable : std::false_type {};
template <class T> struct loader_pairable<T, std::void_t<decltype(T::PAIRABLE)>> : std::integral_constant<bool, T::PAIRABLE> {};

// ------------------------------------------------------------------------------------------
// the kernel
// ------------------------------------------------------------------------------------------
// Wave layout: a k-group is WGM x WGN waves, each owning WM x WN MFMA tiles of 32x32 -- block tile
// BM = 32*WM*WGM by BN = 32*WN*WGN -- and a block is KW k-groups: group kg takes every KW-th k-pair of
// each LDS tile, the partial accumulators are summed through LDS at the end (fixed order).  The block has
// 64*WGM*WGN*KW >= 256 threads; the first 256 fetch and stage the tiles, the rest only issue MFMAs.
//   2x2 waves, KW = 1         the conv forms and every large GEMM: 64x64 .. 128x128 tiles
//   2x2 waves, KW = 2 / 4     64x64 tile shared by 8 / 16 waves (long reductions with few tiles)
//   1x4 waves                 32x128 tile for outputs with <= 32 rows (32-channel convs, 32x48 wgrads)
//   2x1 / 1x2 / 1x1 waves with KW = 2 .. 8 (BK = 64)
//                             the 512-wide MLP layers at batch 512-1024: a 1024x512 output is only 128
//                             tiles of 64x64, half the CUs; 64x32 / 32x64 / 32x32 tiles give every CU a
//                             block, the k-groups give every SIMD one or two waves with a K/KW-long MFMA
//                             chain -- no split-K launch, no partials through HBM, no finish kernel.
template <class P, class Q, class E, int WM, int WN, bool ROWSUM, int KW, int WGM, int WGN>
__global__ __launch_bounds__(64 * WGM * WGN * KW, (64 * WGM * WGN * KW == 256 ? 2 : 1))
void igemm_kernel(P p, Q q, E e, int K, int klen, SplitSink sink) {
    constexpr int BKK = P::BKV;
    constexpr int WPG = WGM * WGN;                  // waves per k-group
    constexpr int BM = 32 * WM * WGM, BN = 32 * WN * WGN;
    constexpr int NT = 64 * WPG * KW;
    static_assert(P::TILE == BM && Q::TILE == BN, "loader tile mismatch");
    static_assert(P::BKV == Q::BKV, "loaders disagree on the k-tile depth");
    static_assert(NT >= NTHREADS && NT <= 1024, "a block needs 256 mover threads");
    static_assert(BKK / 8 % KW == 0, "the 8-k chunks of a tile must divide over the k-groups");
    static_assert(!ROWSUM || !P::RMAJOR, "row sums read a k-major P tile");
    // dynamic LDS (the 128x128 tile needs 66 KiB, above the 64 KiB static limit); the only LDS
    // object of the kernel, so its base is 16-byte aligned (cdna_hip_programming.md G17)
    extern __shared__ __attribute__((aligned(16))) float lds_raw[];
    typedef typename P::Tile PTile;
    typedef typename Q::Tile QTile;
    constexpr int P_FLOATS = P::ROWS * P::PITCH, Q_FLOATS = Q::ROWS * Q::PITCH;
    // buffer b of the two-stage LDS ring (computed, not tabulated: b is a run-time value in the 2 x 2 path)
    auto Ps = [&](int b) { return reinterpret_cast<PTile>(lds_raw + b * P_FLOATS); };
    auto Qs = [&](int b) { return reinterpret_cast<QTile>(lds_raw + 2 * P_FLOATS + b * Q_FLOATS); };

    if (MVAE_CHAIN_PRIO && KW > 1 && WGM * WGN < 4) __builtin_amdgcn_s_setprio(MVAE_CHAIN_PRIO);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int kg = wave / WPG, wq = wave % WPG;     // k-group of this wave, its slot inside the group
    const int wi = wq / WGN, wj = wq % WGN;
    const bool mover = (NT == NTHREADS) || t < NTHREADS;
    const int tiles_j = sink.tiles_j;
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (sink.xcd_map) {
        // Workgroups go to the 8 XCDs round-robin in launch order, and each XCD has its own L2: with the j tile on
        // blockIdx.x every XCD touches every row band of P (a 1024 x 512 x 512 Linear pulled 22 MB through the
        // fabric for 5 MB of operands).  Re-map the launch order so that XCD x owns a (tiles_i / 4) x (tiles_j / 2)
        // sub-grid of the output: P is fetched by 2 XCDs, Q by 4 (host checks divisibility, one class, no split).
        const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), xcd = lin & 7u, slot = lin >> 3;
        if (MVAE_PAIR_NEIGH && sink.xcd_map == 5) {
            // experiment: XCD x owns j tiles x, x + 8, ...; the two class pairs of a tile sit in consecutive slots
            bx = (int)((((slot >> 1) * 8u + xcd) << 1) | (slot & 1u));
            if ((long)bx * sink.items >= (long)sink.tiles_j * sink.ncls) return;
        } else if (sink.xcd_map == 4) {
            // split reductions (the conv weight gradients: 16 .. 128 k ranges x a few output tiles).  In launch order
            // the tiles of ONE k range -- which share its rows of both operands -- spread over all 8 XCDs, and every
            // L2 fetched every k range: 122.7 MB for 30 MB of operands on ConvTranspose2d(256, 128)'s weight gradient
            // (profiles/r03_traffic.json).  XCD x owns k ranges x, x + 8, ... with all their tiles back to back
            // (the host pads gridDim.z to a multiple of 8; the padding blocks leave here).
            const unsigned tiles = gridDim.x * gridDim.y, grp = slot / tiles, tile = slot - grp * tiles;
            bz = (int)(xcd + 8u * grp);
            by = (int)(tile / gridDim.x); bx = (int)(tile - (unsigned)by * gridDim.x);
            if ((long)bz * klen >= K) return;
        } else if (sink.xcd_map == 3) {
            // one reduction range, gather-fed conv forms: the i tiles (output-channel bands) of one (class, j tile) item
            // read the same gathered columns -- XCD x owns items x, x + 8, ..., their bands back to back (gridDim.x is
            // padded to a multiple of 8)
            const unsigned jl = slot / gridDim.y;
            by = (int)(slot - jl * gridDim.y); bx = (int)(jl * 8u + xcd);
            if ((long)bx * sink.items >= (long)sink.tiles_j * sink.ncls) return;
        } else
#if MVAE_XCD_ROWS
        if (sink.xcd_map == 2) {
            // row bands: XCD x owns rows [x * tiles_i / 8, (x + 1) * tiles_i / 8) x ALL column tiles -- the layer behind
            // this one (same rows, same bands) then finds its whole input in the L2 that produced it
            const unsigned sj = gridDim.x, si = gridDim.y >> 3;
            const unsigned ti = slot / sj, tj = slot - ti * sj;
            bx = (int)tj; by = (int)(xcd * si + ti);
        } else
#endif
        {
        const unsigned sj = gridDim.x >> 1, si = gridDim.y >> 2;
        const unsigned ti = slot / sj, tj = slot - ti * sj;
        bx = (int)((xcd & 1u) * sj + tj); by = (int)((xcd >> 1) * si + ti);
        }
    }
    // Multi-item blocks (sink.items > 1, conv forms with short reductions): the block owns `n_items` consecutive
    // (class, j tile) items, and the software pipeline runs ACROSS them -- the first k-tiles of item w+1 are
    // fetched and staged during the last k-steps of item w, its epilogue stores go out while the next item's
    // MFMAs already run.  With K = 256 .. 512 a tile is 8 - 16 k-steps; paying the cold start (loader set-up,
    // ~2 us of load latency, LDS staging) once per tile was 20-30 % of those kernels.
    const int first_item = bx * sink.items;
    const int n_items = min(sink.items, tiles_j * sink.ncls - first_item);
    if (sink.items > 1) bx = first_item;
    // class-minor order (transposed-conv parity classes): the s*s classes of one j tile are neighbours in launch
    // order (and the items of one multi-item block), so the interleaved output lattice of a region is written --
    // and the shared input neighbourhood read -- by blocks that run together, not a whole class sweep apart
    auto item_of = [&](int it, int &c, int &jt) {
        if (sink.cls_minor) { jt = it / sink.ncls; c = it - jt * sink.ncls; }
        else { c = it / tiles_j; jt = it - c * tiles_j; }
    };
    int cls, jt0;
    item_of(bx, cls, jt0);
    const int i0 = by * BM, split = bz;
    int j0 = jt0 * BN;
    const int kbeg = split * klen;
    const int kend = min(K, kbeg + klen);
    const int nsteps = (kend - kbeg + BKK - 1) / BKK;

    f32x16 acc[WM][WN];
#pragma unroll
    for (int a = 0; a < WM; ++a)
#pragma unroll
        for (int b = 0; b < WN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

#if MVAE_STAGGER
    // experiment: de-phase the blocks that share a CU (block b lands in slot (b / 256) % 4 of its CU when the
    // grid is dispatched in order), so that their non-MFMA phases do not coincide
    {
        const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        const unsigned slot = (lin >> 8) & 3u;
        if (slot == 1) __builtin_amdgcn_s_sleep(MVAE_STAGGER);
        else if (slot == 2) __builtin_amdgcn_s_sleep(2 * MVAE_STAGGER);
        else if (slot == 3) __builtin_amdgcn_s_sleep(3 * MVAE_STAGGER);
    }
#endif
    p.init(i0, t, cls);
    q.init(j0, t, cls);
    e.set_class(cls);
    // the small layouts' cooperative epilogue (below): what this thread's outputs will need -- bias, the producer's
    // pre-activation, the dropout mask -- is fetched NOW and lands behind the whole reduction (EpRowMajor::fetch)
    // (up to four outputs per thread: the 256-thread variants would hold eight -- 24 registers, an occupancy step)
    constexpr bool COOP_PRE = KW > 1 && WGM * WGN < 4 && ep_prefetch<E>::value && (BM * BN) % NT == 0 &&
                              (BM * BN) / NT <= (E::ROWRED ? 8 : 4) && MVAE_EPI_PREFETCH && (!E::ROWRED || MVAE_EPI_PREFETCH_ROWRED);
    constexpr int CNE = COOP_PRE ? (BM * BN) / NT : 1;
    typename ep_pre<E>::type cpre[CNE];
    if constexpr (COOP_PRE) {
#pragma unroll
        for (int k = 0; k < CNE; ++k) {
            const int el = t + k * NT;
            cpre[k] = e.fetch(i0 + el / BN, j0 + el % BN);
        }
    }
    typename P::Regs pr0, pr1;
    typename Q::Regs qr0, qr1;
    // db = sum over the reduction axis of P (dy^T): the bias gradient for free.  The 256 movers split the
    // tile: thread t owns row t % BM and every RS_PARTS-th k of it; parts are summed through LDS at the end.
    constexpr int RS_PARTS = NTHREADS / BM;
    const bool rs_block = ROWSUM && jt0 == 0;                         // block-uniform: the class's first j tile
    const int rs_row = t % BM, rs_part = t / BM;
    float rsum = 0.f;
    const int lrow = lane >> 5, lcol = lane & 31;

    auto compute = [&](int buf, auto &&hook) {
        if (ROWSUM) {
            if (rs_block && mover) {
#pragma unroll
                for (int kk = 0; kk < BKK / RS_PARTS; ++kk) rsum += Ps(buf)[kk * RS_PARTS + rs_part][rs_row];
            }
        }
        // A k-tile is cut into chunks of 8 k's; k-group kg takes chunks kg, kg + KW, ...  Within a chunk
        // lanes 0-31 hold k = 8c + j and lanes 32-63 k = 8c + 4 + j for the j-th of its 4 MFMAs (each
        // 32x32x2 MFMA sums two k's; which two is free as long as both operands agree), so a row-major
        // operand tile feeds 4 MFMAs with ONE ds_read_b128 per lane (fetched a whole chunk ahead); a k-major
        // tile is read one value per MFMA, one MFMA ahead (4 + 4 live registers at 2 x 2 tiles per wave --
        // whole-chunk prefetch of both operands spilled the 128 x 128 kernels).  Either way the LDS latency
        // hides behind the 64-cycle matrix instructions.
        constexpr int NCH = BKK / 8 / KW, NS = NCH * 4;
        auto krow = [&](int st) { return ((st >> 2) * KW + kg) * 8 + 4 * lrow + (st & 3); };   // tile row of step st
        float4 pa[WM], pa_n[WM], qb[WN], qb_n[WN];      // row-major operands: this chunk's / the next chunk's 4 k's
        float sa[WM], sa_n[WM], sb[WN], sb_n[WN];       // k-major operands: this step's / the next step's value
#pragma unroll
        for (int x = 0; x < WM; ++x) {
            if (P::RMAJOR) pa[x] = P::frag(Ps(buf), krow(0), (wi * WM + x) * 32 + lcol);
            else sa[x] = Ps(buf)[krow(0)][(wi * WM + x) * 32 + lcol];
        }
#pragma unroll
        for (int y = 0; y < WN; ++y) {
            if (Q::RMAJOR) qb[y] = Q::frag(Qs(buf), krow(0), (wj * WN + y) * 32 + lcol);
            else sb[y] = Qs(buf)[krow(0)][(wj * WN + y) * 32 + lcol];
        }
#if MVAE_KO >= 4
#pragma unroll
        for (int x = 0; x < WM; ++x) { pa_n[x] = pa[x]; sa_n[x] = sa[x]; }
#pragma unroll
        for (int y = 0; y < WN; ++y) { qb_n[y] = qb[y]; sb_n[y] = sb[y]; }
#endif
#pragma unroll
        for (int st = 0; st < NS; ++st) {
            const int j = st & 3;
            if (MVAE_KO < 4 && j == 0 && st + 4 < NS) { // row-major operands: the next chunk, a chunk ahead
#pragma unroll
                for (int x = 0; x < WM; ++x)
                    if (P::RMAJOR) pa_n[x] = P::frag(Ps(buf), krow(st + 4), (wi * WM + x) * 32 + lcol);
#pragma unroll
                for (int y = 0; y < WN; ++y)
                    if (Q::RMAJOR) qb_n[y] = Q::frag(Qs(buf), krow(st + 4), (wj * WN + y) * 32 + lcol);
            }
            if (MVAE_KO < 4 && st + 1 < NS) {           // k-major operands: the next step's values
#pragma unroll
                for (int x = 0; x < WM; ++x)
                    if (!P::RMAJOR) sa_n[x] = Ps(buf)[krow(st + 1)][(wi * WM + x) * 32 + lcol];
#pragma unroll
                for (int y = 0; y < WN; ++y)
                    if (!Q::RMAJOR) sb_n[y] = Qs(buf)[krow(st + 1)][(wj * WN + y) * 32 + lcol];
            }
            __builtin_amdgcn_sched_barrier(0);
#if MVAE_SETPRIO
            __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
            for (int x = 0; x < WM; ++x)
#pragma unroll
                for (int y = 0; y < WN; ++y) {
                    const float av = !P::RMAJOR ? sa[x] : j == 0 ? pa[x].x : j == 1 ? pa[x].y : j == 2 ? pa[x].z : pa[x].w;
                    const float bv = !Q::RMAJOR ? sb[y] : j == 0 ? qb[y].x : j == 1 ? qb[y].y : j == 2 ? qb[y].z : qb[y].w;
                    acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[x][y], 0, 0, 0);
                }
#if MVAE_SETPRIO
            __builtin_amdgcn_s_setprio(0);
#endif
            __builtin_amdgcn_sched_barrier(0);
            hook(st, NS);                               // a slice of the next tile's loads / stores, in this group's shadow
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int x = 0; x < WM; ++x) {
                if (!P::RMAJOR) sa[x] = sa_n[x];
                else if (j == 3) pa[x] = pa_n[x];
            }
#pragma unroll
            for (int y = 0; y < WN; ++y) {
                if (!Q::RMAJOR) sb[y] = sb_n[y];
                else if (j == 3) qb[y] = qb_n[y];
            }
        }
    };

    // 2 or 4 tiles per wave: 32 - 64 MFMAs (2048 - 4096 cycles) per k-step cover a global-load latency,
    // so ONE tile in flight in registers is enough -- the second register stage made the 128 x 128 kernels
    // spill (256 VGPRs)
    constexpr bool DEEP = WM * WN < 2;
    auto no_hook = [](int, int) {};
    // Interleaved main loop (every thread a mover, every k-tile full): the next tile's global loads are issued
    // slice by slice right after the first MFMA groups of the current tile and its LDS stores after the last
    // ones, so that address arithmetic, load issue, mask multiplies and ds_writes run in the shadow of the
    // wave's OWN matrix instructions.  With load(); compute(); store() as three phases a wave issues no MFMA
    // for a few hundred cycles per k-step and the pipe idles unless another block's wave happens to be in its
    // MFMA phase (tools/mfma_peak: the pipe itself sustains 154.6 TFLOP/s from one wave per SIMD).
    constexpr bool CAN_IL = (NT == NTHREADS) && P::PARTS && Q::PARTS && MVAE_INTERLEAVE;
    // block-uniform: the buffer path covers the block's k range (row loaders also take a partial last tile)
    const bool full = nsteps > 0 && ((kend - kbeg) % BKK == 0 || (P::TAIL && Q::TAIL)) && p.fast && q.fast;
    const bool il = CAN_IL && full;
    p.begin(kbeg, t); q.begin(kbeg, t);
    // ---- multi-item blocks: the pipeline of the interleaved loops, run across the block's items
    constexpr bool CAN_MULTI = CAN_IL && KW == 1 && !ROWSUM && E::MULTI;      // conv forms only (EpNCHW)
    if (CAN_MULTI && sink.items > 1) {
        if constexpr (CAN_MULTI) {
            if (!(il && nsteps >= 2 && gridDim.z == 1)) return;     // launch conditions (host): full k-tiles, >= 2 steps, no split
            const int G = n_items * nsteps;             // k-steps of the whole block
            auto item_tile = [&](int w, int &c, int &jt) { item_of(first_item + w, c, jt); jt *= BN; };
            auto loaders_to = [&](int w) {              // point the loaders at item w (the load stream runs ahead)
                int c, jt; item_tile(w, c, jt);
                p.init(i0, t, c); q.init(jt, t, c);
                p.begin(kbeg, t); q.begin(kbeg, t);
            };
            auto kof = [&](int g) { return kbeg + (g % nsteps) * BKK; };
            // PAIR (one-tile waves only: a second accumulator set costs the 2- and 4-tile kernels an occupancy step
            // or spills): the px = 0 class of a stride-2 row is kept until px = 1 is done
            // ... the 32-row (1 x 4 waves) and 64-row (2 x 2 waves) layouts: 146 and 127 VGPRs, same occupancy as without
            constexpr bool PAIRK = E::PAIR && WM * WN == 1 && loader_pairable<Q>::value;
            std::conditional_t<PAIRK, f32x16, char> hold;
            // (Round 4, profiles/r04_conv_knockout.txt: with stores, loads, LDS traffic and barriers ALL knocked out this kernel still
            //  needs 83.8 us for 54.6 us of matrix time at 512 rows -- 4.0 vector + 5.4 scalar instructions per MFMA, most of them in
            //  the per-item path below.  Issuing the stores outside the compiler's shared load / store wait counter, or spreading a
            //  finished pair's stores over the next item's first k-step (+16 registers), measured 0 and -3 %: not kept.)
            // statistics-only destination (EpStats; one-tile waves): per-lane sums of v and v^2 over the block's items
            constexpr bool STATK = ep_stats<E>::value && WM * WN == 1;
            // Sums are taken around a SHIFT -- the first value a half wavefront's lane 0 sees of each row -- so that
            // M2 = sum((v - s)^2) - (sum(v - s))^2 / n does not cancel when a channel's mean is large against its spread
            // (sum(v^2) - sum(v) * mean in fp32 loses every digit of the variance at |mean| / std ~ 1e3, and the
            // clamp at 0 hid it; the two-pass BatchNorm kernels this path replaces have no such limit).  One subtract more
            // per value; any s is exact in exact arithmetic, a sample of the row is close to its mean.
            std::conditional_t<STATK, f32x16, char> st1, st2, shf;
            bool have_shift = false;
            if constexpr (STATK) {
#pragma unroll
                for (int r = 0; r < 16; ++r) { st1[r] = 0.f; st2[r] = 0.f; shf[r] = 0.f; }
            }
            auto stats_flush = [&]() {
                if constexpr (STATK) {
                    // lanes of one half wave hold the 32 columns of rows (r & 3) + 8 * (r >> 2) + 4 * lrow, and share the shift
#pragma unroll
                    for (int r = 0; r < 16; ++r) { st1[r] = half_wave_sum(st1[r]); st2[r] = half_wave_sum(st2[r]); }
                    __syncthreads();                    // the tile buffers are free
                    float *red = lds_raw;               // [wave][32 rows][3]: sum(v - s), sum((v - s)^2), s
                    if (lcol == 0) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int row = (r & 3) + 8 * (r >> 2) + 4 * lrow;
                            red[(wq * 32 + row) * 3 + 0] = st1[r];
                            red[(wq * 32 + row) * 3 + 1] = st2[r];
                            red[(wq * 32 + row) * 3 + 2] = shf[r];
                        }
                    }
                    __syncthreads();
                    if (t < BM) {                       // row t of the block tile: its WGN column waves, merged in order
                        const int band = t >> 5, row = t & 31;
                        const float nw = (float)(n_items * 32);         // values per (wave, row)
                        float mw[WGN], m2w[WGN], mean = 0.f;
#pragma unroll
                        for (int w2 = 0; w2 < WGN; ++w2) {
                            const float *rr = red + ((band * WGN + w2) * 32 + row) * 3;
                            const float d = rr[0] / nw;
                            mw[w2] = rr[2] + d;
                            m2w[w2] = fmaxf(rr[1] - rr[0] * d, 0.f);
                            mean += mw[w2];
                        }
                        mean /= (float)WGN;
                        float m2 = 0.f;
#pragma unroll
                        for (int w2 = 0; w2 < WGN; ++w2) m2 += m2w[w2] + nw * (mw[w2] - mean) * (mw[w2] - mean);
                        if (i0 + t < e.C) {
                            float *dst = e.part + ((size_t)(first_item / sink.ncls) * e.C + i0 + t) * 2;
                            dst[0] = mean;
                            dst[1] = m2;
                        }
                    }
                }
            };
            auto finish_item = [&](int w) {             // epilogue of item w, accumulators cleared for the next
                int c, jt; item_tile(w, c, jt);
                if constexpr (STATK) {                  // every column is real (host: J % BN == 0)
                    if (!have_shift) {                  // block-uniform: the first item
                        have_shift = true;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int bits = __float_as_int(acc[0][0][r]);
                            const float lo = __int_as_float(__builtin_amdgcn_readlane(bits, 0));
                            const float hi = __int_as_float(__builtin_amdgcn_readlane(bits, 32));
                            shf[r] = lrow ? hi : lo;
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float d = acc[0][0][r] - shf[r];
                        st1[r] += d;
                        st2[r] = fmaf(d, d, st2[r]);
                        acc[0][0][r] = 0.f;
                    }
                    return;
                }
                if constexpr (PAIRK) {
                    // the host only launches this type with an even number of items per block in class-minor order:
                    // item w even = class (py, 0), item w + 1 = (py, 1) of the same j tile
                    if (!(w & 1)) {
                        hold = acc[0][0];
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;
                    } else {
                        e.set_class(c - 1);
                        const int j = jt + wj * 32 + lcol;
                        if constexpr (ep_buffer<E>::value) {
                            e.tile(jt);
                            (void)e.col(j);
                            const int rb = __builtin_amdgcn_readfirstlane(i0 + wi * 32);
#pragma unroll
                            for (int r = 0; r < 16; ++r) e.put2_b(rb, r, hold[r], acc[0][0][r]);
                        } else if (e.col(j)) {
                            const int ib = i0 + wi * 32 + 4 * lrow;
#pragma unroll
                            for (int r = 0; r < 16; ++r) e.put2(ib + (r & 3) + 8 * (r >> 2), hold[r], acc[0][0][r]);
                        }
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;
                    }
                    return;
                }
                e.set_class(c);
                if constexpr (ep_buffer<E>::value) e.tile(jt);
#pragma unroll
                for (int y = 0; y < WN; ++y) {
                    const int j = jt + (wj * WN + y) * 32 + lcol;
                    if constexpr (ep_buffer<E>::value) {
                        (void)e.col(j);
#pragma unroll
                        for (int x = 0; x < WM; ++x) {
                            const int rb = __builtin_amdgcn_readfirstlane(i0 + (wi * WM + x) * 32);
#pragma unroll
                            for (int r = 0; r < 16; ++r) e.put_b(rb, r, acc[x][y][r]);
                        }
                    } else if (e.col(j)) {
#pragma unroll
                        for (int x = 0; x < WM; ++x) {
                            const int ib = i0 + (wi * WM + x) * 32 + 4 * lrow;
#pragma unroll
                            for (int r = 0; r < 16; ++r) e.put(ib + (r & 3) + 8 * (r >> 2), j, acc[x][y][r]);
                        }
                    }
#pragma unroll
                    for (int x = 0; x < WM; ++x)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.f;
                }
            };
            if (!DEEP) {
                p.load_part(kbeg, kend, t, pr0, 0, 1); q.load_part(kbeg, kend, t, qr0, 0, 1);
                p.store_part(Ps(0), t, pr0, 0, 1); q.store_part(Qs(0), t, qr0, 0, 1);
                __syncthreads();
                for (int g = 0; g + 1 < G; ++g) {
                    const bool last = (g + 1) % nsteps == 0;      // step g ends its item; tile g+1 opens the next
                    if (last) loaders_to((g + 1) / nsteps);
                    const int kn = kof(g + 1), nb = (g + 1) & 1;
                    compute(g & 1, [&](int st, int ns) {
                        const int nl = (WM * WN >= 4) ? ns / 4 : ns / 8;
                        if (MVAE_KO < 1 && st < nl) { p.load_part(kn, kend, t, pr0, st, nl); q.load_part(kn, kend, t, qr0, st, nl); }
                        if (MVAE_KO < 2 && st >= ns - nl) { p.store_part(Ps(nb), t, pr0, st - (ns - nl), nl); q.store_part(Qs(nb), t, qr0, st - (ns - nl), nl); }
                    });
                    if (MVAE_KO < 3) __syncthreads();
                    if (last) finish_item(g / nsteps);
                }
                compute((G - 1) & 1, no_hook);
                finish_item(n_items - 1);
                stats_flush();
                return;
            }
            // one tile per wave: two tiles in flight in registers (see the single-item loop below)
            p.load_part(kbeg, kend, t, pr0, 0, 1); q.load_part(kbeg, kend, t, qr0, 0, 1);
            p.load_part(kbeg + BKK, kend, t, pr1, 0, 1); q.load_part(kbeg + BKK, kend, t, qr1, 0, 1);      // nsteps >= 2
            p.store_part(Ps(0), t, pr0, 0, 1); q.store_part(Qs(0), t, qr0, 0, 1);
            __syncthreads();
            int g = 0;
            for (; g + 2 < G; g += 2) {
                // even step g on buffer 0: fetch tile g+2 into set 0, stage tile g+1 (set 1) in buffer 1
                if ((g + 2) % nsteps == 0) loaders_to((g + 2) / nsteps);
                const int k2 = kof(g + 2);
                compute(0, [&](int st, int ns) {
                    const int nl = ns / 2;
                    if (st < nl) { if (MVAE_KO < 1) { p.load_part(k2, kend, t, pr0, st, nl); q.load_part(k2, kend, t, qr0, st, nl); } }
                    else if (MVAE_KO < 2) { p.store_part(Ps(1), t, pr1, st - nl, ns - nl); q.store_part(Qs(1), t, qr1, st - nl, ns - nl); }
                });
                if (MVAE_KO < 3) __syncthreads();
                if ((g + 1) % nsteps == 0) finish_item(g / nsteps);
                // odd step g+1 on buffer 1: fetch tile g+3 into set 1 (the last tile again when there is none), stage
                // tile g+2 (set 0) in buffer 0
                const bool more3 = g + 3 < G;
                if (more3 && (g + 3) % nsteps == 0) loaders_to((g + 3) / nsteps);
                const int k3 = kof(more3 ? g + 3 : g + 2);
                compute(1, [&](int st, int ns) {
                    const int nl = ns / 2;
                    if (st < nl) { if (MVAE_KO < 1) { p.load_part(k3, kend, t, pr1, st, nl); q.load_part(k3, kend, t, qr1, st, nl); } }
                    else if (MVAE_KO < 2) { p.store_part(Ps(0), t, pr0, st - nl, ns - nl); q.store_part(Qs(0), t, qr0, st - nl, ns - nl); }
                });
                if (MVAE_KO < 3) __syncthreads();
                if ((g + 2) % nsteps == 0) finish_item((g + 1) / nsteps);
            }
            // tail: tile g is staged in buffer 0; tile g+1 (if any) waits in register set 1
            if (g + 1 < G) {
                compute(0, [&](int st, int ns) {
                    if (st >= ns / 2) { p.store_part(Ps(1), t, pr1, st - ns / 2, ns - ns / 2); q.store_part(Qs(1), t, qr1, st - ns / 2, ns - ns / 2); }
                });
                __syncthreads();
                if ((g + 1) % nsteps == 0) finish_item(g / nsteps);
                compute(1, no_hook);
            } else {
                compute(0, no_hook);
            }
            finish_item(n_items - 1);
            stats_flush();
            return;
        }
    }
    if (CAN_IL && il && !DEEP) {
        p.load_part(kbeg, kend, t, pr0, 0, 1); q.load_part(kbeg, kend, t, qr0, 0, 1);
        p.store_part(Ps(0), t, pr0, 0, 1); q.store_part(Qs(0), t, qr0, 0, 1);
        __syncthreads();
        for (int s = 0; s + 1 < nsteps; ++s) {
            const int kn = kbeg + (s + 1) * BKK, nb = (s + 1) & 1;
            compute(s & 1, [&](int st, int ns) {
                const int nl = (WM * WN >= 4) ? ns / 4 : ns / 8;   // load slots at the head, store slots at the tail: >= 1500 MFMA cycles apart
#if MVAE_KO < 1
                if (st < nl) { p.load_part(kn, kend, t, pr0, st, nl); q.load_part(kn, kend, t, qr0, st, nl); }
#endif
#if MVAE_KO < 2
                if (st >= ns - nl) { p.store_part(Ps(nb), t, pr0, st - (ns - nl), nl); q.store_part(Qs(nb), t, qr0, st - (ns - nl), nl); }
#endif
            });
#if MVAE_KO < 3
            __syncthreads();
#endif
        }
        compute((nsteps - 1) & 1, no_hook);
        __syncthreads();
    } else if (CAN_IL && il && DEEP) {
        // one tile per wave: 16 MFMAs per k-step do not cover a global-load latency, so two tiles are in flight
        // in registers; tile s+2 is fetched in the first half of step s, tile s+1 staged in its second half
        p.load_part(kbeg, kend, t, pr0, 0, 1); q.load_part(kbeg, kend, t, qr0, 0, 1);
        if (nsteps > 1) { p.load_part(kbeg + BKK, kend, t, pr1, 0, 1); q.load_part(kbeg + BKK, kend, t, qr1, 0, 1); }
        p.store_part(Ps(0), t, pr0, 0, 1); q.store_part(Qs(0), t, qr0, 0, 1);
        __syncthreads();
        int s = 0;
        for (; s + 2 < nsteps; s += 2) {
            const int k2 = kbeg + (s + 2) * BKK, k3 = kbeg + min(s + 3, nsteps - 1) * BKK;
            compute(0, [&](int st, int ns) {
                const int nl = ns / 2;
                if (st < nl) { if (MVAE_KO < 1) { p.load_part(k2, kend, t, pr0, st, nl); q.load_part(k2, kend, t, qr0, st, nl); } }
                else if (MVAE_KO < 2) { p.store_part(Ps(1), t, pr1, st - nl, ns - nl); q.store_part(Qs(1), t, qr1, st - nl, ns - nl); }
            });
            if (MVAE_KO < 3) __syncthreads();
            compute(1, [&](int st, int ns) {
                const int nl = ns / 2;
                if (st < nl) { if (MVAE_KO < 1) { p.load_part(k3, kend, t, pr1, st, nl); q.load_part(k3, kend, t, qr1, st, nl); } }
                else if (MVAE_KO < 2) { p.store_part(Ps(0), t, pr0, st - nl, ns - nl); q.store_part(Qs(0), t, qr0, st - nl, ns - nl); }
            });
            if (MVAE_KO < 3) __syncthreads();
        }
        // tail: tile s is staged in buffer 0; tile s+1 (if any) waits in register set 1
        if (s + 1 < nsteps) {
            compute(0, [&](int st, int ns) {
                if (st >= ns / 2) { p.store_part(Ps(1), t, pr1, st - ns / 2, ns - ns / 2); q.store_part(Qs(1), t, qr1, st - ns / 2, ns - ns / 2); }
            });
            __syncthreads();
            compute(1, no_hook);
            __syncthreads();
        } else if (s < nsteps) {
            compute(0, no_hook);
            __syncthreads();
        }
    } else {
        // phased loops (k-grouped blocks, whose extra waves only issue MFMAs; partial k-tiles; unaligned operands).
        // With full k-tiles the movers still use the buffer loads / raw stores: `fullc` picks the version.
        auto LOAD = [&](auto fullc, int k0, typename P::Regs &pr, typename Q::Regs &qr) {
            if constexpr (decltype(fullc)::value) { p.load_part(k0, kend, t, pr, 0, 1); q.load_part(k0, kend, t, qr, 0, 1); }
            else { p.load(k0, kend, t, pr); q.load(k0, kend, t, qr); }
        };
        auto STORE = [&](auto fullc, int b, const typename P::Regs &pr, const typename Q::Regs &qr) {
            if constexpr (decltype(fullc)::value) { p.store_part(Ps(b), t, pr, 0, 1); q.store_part(Qs(b), t, qr, 0, 1); }
            else { p.store(Ps(b), t, pr); q.store(Qs(b), t, qr); }
        };
        auto phased = [&](auto fullc) {
            if (!DEEP) {
                if (mover && nsteps > 0) LOAD(fullc, kbeg, pr0, qr0);
                if (mover && nsteps > 0) STORE(fullc, 0, pr0, qr0);
                __syncthreads();
                for (int s = 0; s < nsteps; ++s) {
                    const bool more = s + 1 < nsteps;
                    if (mover && more) LOAD(fullc, kbeg + (s + 1) * BKK, pr0, qr0);
                    compute(s & 1, no_hook);
                    if (mover && more) STORE(fullc, (s + 1) & 1, pr0, qr0);
                    __syncthreads();
                }
                return;
            }
            // PRELOAD (full k-tiles through the buffer path): EVERY wave issues the tile loads, unconditionally -- the
            // MFMA-only waves with all offsets out of range, the last trips on the last tile again.  With
            // `if (mover && s + 2 < nsteps) LOAD(...)` the waves that skip the loads reach the stores' wait with fewer
            // operations outstanding, the compiler's wait counts are the minimum over both paths, and the movers wait
            // for the loads they have only just issued: vmcnt(5) .. vmcnt(0) where 11 .. 6 would do -- ONE tile in flight
            // instead of two, a full load latency per k-step (8 MFMAs per wave) in the 512-wide MLP layers.
            constexpr bool PRELOAD = decltype(fullc)::value && ld_can_disable<P>::value && ld_can_disable<Q>::value &&
                                     MVAE_PHASED_PRELOAD;
            if constexpr (PRELOAD) {
                if (nsteps <= 0) return;
                auto tile_k = [&](int s2) { return kbeg + min(s2, nsteps - 1) * BKK; };
                if (MVAE_PHASED_PRELOAD == 2 && !mover) {
                    // variant 2: the MFMA-only waves run their own copy of the loop -- the same barriers, no loads --
                    // so the movers' copy has no control-flow merge between its loads and its stores at all
                    __syncthreads();
                    int s = 0;
                    for (; s + 1 < nsteps; s += 2) {
                        compute(0, no_hook);
                        __syncthreads();
                        compute(1, no_hook);
                        __syncthreads();
                    }
                    if (s < nsteps) {
                        compute(0, no_hook);
                        __syncthreads();
                    }
                    return;
                }
                if constexpr (MVAE_PHASED_PRELOAD == 2 && MVAE_PHASED_DEPTH == 4 && P::TAIL && Q::TAIL) {
                    // FOUR tiles in flight in the movers' registers (only movers get here).  A k-step of these layouts is 8
                    // MFMAs per wave -- 0.2 us -- against ~1 us for an L2 / fabric round trip: with two tiles ahead a step
                    // still ends waiting for its successor's loads (MNIST's 1024 x 512 x 512 launches: 13.2 us for 3.4 us
                    // of matrix time, profiles/r05_mnist_by_shape.txt).  A 512-thread block runs two waves per SIMD, so a
                    // mover may hold 256 registers: four (P, Q) register sets.  Tiles beyond the reduction are requested
                    // at their own k: the row loaders' TAIL rule turns every such load into an out-of-range one -- no
                    // traffic, zeros nobody stages.  Same barriers as the load-free copy above: 1 + nsteps.
                    typename P::Regs pr2, pr3;
                    typename Q::Regs qr2, qr3;
                    auto tk = [&](int s2) { return kbeg + s2 * BKK; };
                    LOAD(fullc, tk(0), pr0, qr0); LOAD(fullc, tk(1), pr1, qr1);
                    LOAD(fullc, tk(2), pr2, qr2); LOAD(fullc, tk(3), pr3, qr3);
                    STORE(fullc, 0, pr0, qr0);
                    __syncthreads();
                    int s = 0;
                    for (; s + 3 < nsteps; s += 4) {
                        LOAD(fullc, tk(s + 4), pr0, qr0);
                        compute(0, no_hook);
                        STORE(fullc, 1, pr1, qr1);
                        __syncthreads();
                        LOAD(fullc, tk(s + 5), pr1, qr1);
                        compute(1, no_hook);
                        STORE(fullc, 0, pr2, qr2);
                        __syncthreads();
                        LOAD(fullc, tk(s + 6), pr2, qr2);
                        compute(0, no_hook);
                        STORE(fullc, 1, pr3, qr3);
                        __syncthreads();
                        LOAD(fullc, tk(s + 7), pr3, qr3);
                        compute(1, no_hook);
                        if (s + 4 < nsteps) STORE(fullc, 0, pr0, qr0);
                        __syncthreads();
                    }
                    const int left = nsteps - s;        // 0 .. 3 steps: tile s sits in buffer 0, s + 1 / s + 2 in sets 1 / 2
                    if (left >= 1) {
                        compute(0, no_hook);
                        if (left >= 2) STORE(fullc, 1, pr1, qr1);
                        __syncthreads();
                    }
                    if (left >= 2) {
                        compute(1, no_hook);
                        if (left >= 3) STORE(fullc, 0, pr2, qr2);
                        __syncthreads();
                    }
                    if (left >= 3) {
                        compute(0, no_hook);
                        __syncthreads();
                    }
                    return;
                }
                if (!mover) { loader_disable<PRELOAD>(p); loader_disable<PRELOAD>(q); }     // variant 1
                LOAD(fullc, tile_k(0), pr0, qr0);
                LOAD(fullc, tile_k(1), pr1, qr1);
                if (mover) STORE(fullc, 0, pr0, qr0);
                __syncthreads();
                int s = 0;
                for (; s + 1 < nsteps; s += 2) {
                    LOAD(fullc, tile_k(s + 2), pr0, qr0);
                    compute(0, no_hook);
                    if (mover) STORE(fullc, 1, pr1, qr1);
                    __syncthreads();
                    LOAD(fullc, tile_k(s + 3), pr1, qr1);
                    compute(1, no_hook);
                    if (mover && s + 2 < nsteps) STORE(fullc, 0, pr0, qr0);
                    __syncthreads();
                }
                if (s < nsteps) {   // odd number of k-steps: the last tile sits in buffer 0
                    compute(0, no_hook);
                    __syncthreads();
                }
                return;
            }
            // one tile per wave -- prologue: tiles 0 and 1 in flight, tile 0 staged
            if (mover && nsteps > 0) LOAD(fullc, kbeg, pr0, qr0);
            if (mover && nsteps > 1) LOAD(fullc, kbeg + BKK, pr1, qr1);
            if (mover && nsteps > 0) STORE(fullc, 0, pr0, qr0);
            __syncthreads();
            // two k-steps per trip (the register sets alternate); a lone last step is peeled off below so the
            // loop has ONE exit -- with a break in the middle hipcc ping-ponged the accumulator between two
            // register sets (16 v_mov + 17 wait states per k-step)
            int s = 0;
            for (; s + 1 < nsteps; s += 2) {
                // even step: MFMA on buffer 0; register set 0 is free -> fetch tile s+2; stage tile s+1
                if (mover && s + 2 < nsteps) LOAD(fullc, kbeg + (s + 2) * BKK, pr0, qr0);
                compute(0, no_hook);
                if (mover) STORE(fullc, 1, pr1, qr1);
                __syncthreads();
                // odd step
                if (mover && s + 3 < nsteps) LOAD(fullc, kbeg + (s + 3) * BKK, pr1, qr1);
                compute(1, no_hook);
                if (mover && s + 2 < nsteps) STORE(fullc, 0, pr0, qr0);
                __syncthreads();
            }
            if (s < nsteps) {       // odd number of k-steps: the last tile sits in buffer 0
                compute(0, no_hook);
                __syncthreads();
            }
        };
        constexpr bool CAN_FULL = P::PARTS && Q::PARTS && MVAE_INTERLEAVE;
        if (CAN_FULL && full) {
            if constexpr (CAN_FULL) phased(std::true_type{});
        } else {
            phased(std::false_type{});
        }
    }
    if (ROWSUM) {
        if (rs_block) {     // sum the RS_PARTS partial row sums in a fixed order (the tile buffers are free)
            if (mover) lds_raw[rs_part * BM + rs_row] = rsum;
            __syncthreads();
            if (t < BM) {
                rsum = 0.f;
#pragma unroll
                for (int pp = 0; pp < RS_PARTS; ++pp) rsum += lds_raw[pp * BM + t];
            }
            __syncthreads();
        }
    }
    constexpr bool COOP = KW > 1 && WPG < 4;        // the small layouts: tile-wide cooperative epilogue
    if (COOP) {
        // every wave parks its accumulators in LDS as tile[kg][i][j] (j along lanes: conflict-free), then all
        // threads sum the KW partials of their outputs in k-group order and run the epilogue -- 32 or 64
        // consecutive j per row, so stores stay full segments.  (With only the k-group-0 waves finishing the
        // tile, one or two of eight waves carried all the bias / Swish / store work.)
        constexpr int TP = BN + 1;                  // odd pitch: rows land on different banks
        float *tile = lds_raw + kg * (BM * TP);
#pragma unroll
        for (int x = 0; x < WM; ++x)
#pragma unroll
            for (int y = 0; y < WN; ++y)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int il = (wi * WM + x) * 32 + 4 * lrow + (r & 3) + 8 * (r >> 2);
                    tile[il * TP + (wj * WN + y) * 32 + lcol] = acc[x][y][r];
                }
        __syncthreads();
        const bool partial_c = gridDim.z > 1;
        if constexpr (COOP_PRE) {
#pragma unroll
            for (int k = 0; k < CNE; ++k) {
                const int el = t + k * NT;
                const int il = el / BN, jl = el % BN;
                float v = 0.f;
#pragma unroll
                for (int g2 = 0; g2 < KW; ++g2) v += lds_raw[g2 * (BM * TP) + il * TP + jl];
                const int i = i0 + il, j = j0 + jl;
                if constexpr (E::ROWRED) {
                    e.put_row_pre(i, j, v, cpre[k]);        // (every thread takes part in every round: see below)
                } else if (partial_c) {
                    if (i < sink.I && j < sink.J)
                        sink.ws[(size_t)cls * sink.cls_region + (size_t)split * sink.stride + (size_t)i * sink.J + j] = v;
                } else if (e.col(j)) {
                    e.put_pre(i, j, v, cpre[k]);
                }
            }
        } else
        for (int el = t; el < BM * BN; el += NT) {
            const int il = el / BN, jl = el % BN;
            float v = 0.f;
#pragma unroll
            for (int g2 = 0; g2 < KW; ++g2) v += lds_raw[g2 * (BM * TP) + il * TP + jl];
            const int i = i0 + il, j = j0 + jl;
            if constexpr (E::ROWRED) {
                // 32 consecutive threads = 32 consecutive columns of one row (BN is a multiple of 32, the trip count
                // is the same for every thread)
                static_assert((BM * BN) % NT == 0, "row-reducing epilogue: every thread takes part in every round");
                e.put_row(i, j, v);
            } else if (partial_c) {
                if (i < sink.I && j < sink.J)
                    sink.ws[(size_t)cls * sink.cls_region + (size_t)split * sink.stride + (size_t)i * sink.J + j] = v;
            } else if (e.col(j)) {
                e.put(i, j, v);
            }
        }
        if (ROWSUM) {
            if (rs_block && t < BM && i0 + t < sink.I) {
                float *dst = sink.rowsum + (size_t)cls * sink.rowsum_cls_stride + (size_t)split * sink.rowsum_stride + i0 + t;
                if (!partial_c && sink.rowsum_accumulate) rsum += *dst;
                *dst = rsum;
            }
        }
        return;
    }
    if (KW > 1) {
        // sum the k-groups' accumulators through LDS (the tile buffers are free after the last barrier)
        constexpr int PER_WAVE = WM * WN * 16 * 64;
        float *red = lds_raw;
        if (kg > 0) {
            float *dst = red + ((kg - 1) * WPG + wq) * PER_WAVE + lane;
#pragma unroll
            for (int x = 0; x < WM; ++x)
#pragma unroll
                for (int y = 0; y < WN; ++y)
#pragma unroll
                    for (int r = 0; r < 16; ++r) dst[((x * WN + y) * 16 + r) * 64] = acc[x][y][r];
        }
        __syncthreads();
        if (kg > 0) return;
#pragma unroll
        for (int g2 = 0; g2 < KW - 1; ++g2) {
            const float *src = red + (g2 * WPG + wq) * PER_WAVE + lane;
#pragma unroll
            for (int x = 0; x < WM; ++x)
#pragma unroll
                for (int y = 0; y < WN; ++y)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[x][y][r] += src[((x * WN + y) * 16 + r) * 64];
        }
    }

    // C/D layout of 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    const bool partial = gridDim.z > 1;
    if constexpr (E::ROWRED) {
        // a half wavefront holds 32 consecutive columns of row ib + ...: all lanes call (no column early-out)
#pragma unroll
        for (int y = 0; y < WN; ++y) {
            const int j = j0 + (wj * WN + y) * 32 + lcol;
#pragma unroll
            for (int x = 0; x < WM; ++x) {
                const int ib = i0 + (wi * WM + x) * 32 + 4 * lrow;
#pragma unroll
                for (int r = 0; r < 16; ++r) e.put_row(ib + (r & 3) + 8 * (r >> 2), j, acc[x][y][r]);
            }
        }
        return;
    }
    if constexpr (ep_buffer<E>::value) {
        if (!partial) {         // block-uniform
            e.tile(j0);
#pragma unroll
            for (int y = 0; y < WN; ++y) {
                (void)e.col(j0 + (wj * WN + y) * 32 + lcol);
#pragma unroll
                for (int x = 0; x < WM; ++x) {
                    const int rb = __builtin_amdgcn_readfirstlane(i0 + (wi * WM + x) * 32);
#pragma unroll
                    for (int r = 0; r < 16; ++r) e.put_b(rb, r, acc[x][y][r]);
                }
            }
            return;             // (ROWSUM launches never carry an NCHW epilogue)
        }
    }
#pragma unroll
    for (int y = 0; y < WN; ++y) {
        const int j = j0 + (wj * WN + y) * 32 + lcol;
        if (partial) {
            if (j >= sink.J) continue;
        } else if (!e.col(j)) {
            continue;
        }
#pragma unroll
        for (int x = 0; x < WM; ++x) {
            const int ib = i0 + (wi * WM + x) * 32 + 4 * lrow;
            if constexpr (ep_prefetch<E>::value && MVAE_EPI_BATCH) {
                // operands of eight outputs at a time, all fetched before the first is used (see EpRowMajor::fetch: left
                // inside put(), every output waits out its own memory latency behind a block-uniform branch)
                if (!partial) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        typename ep_pre<E>::type pr[8];
#pragma unroll
                        for (int r = 0; r < 8; ++r) pr[r] = e.fetch(ib + (r & 3) + 8 * ((8 * h + r) >> 2), j);
#pragma unroll
                        for (int r = 0; r < 8; ++r) e.put_pre(ib + (r & 3) + 8 * ((8 * h + r) >> 2), j, acc[x][y][8 * h + r], pr[r]);
                    }
                    continue;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = ib + (r & 3) + 8 * (r >> 2);
                if (partial) {
                    if (i < sink.I)
                        sink.ws[(size_t)cls * sink.cls_region + (size_t)split * sink.stride + (size_t)i * sink.J + j] = acc[x][y][r];
                } else {
                    e.put(i, j, acc[x][y][r]);
                }
            }
        }
    }
    if (ROWSUM) {
        if (rs_block && t < BM && i0 + t < sink.I) {     // t < BM <= 128: waves of k-group 0
            float *dst = sink.rowsum + (size_t)cls * sink.rowsum_cls_stride + (size_t)split * sink.rowsum_stride + i0 + t;
            if (!partial && sink.rowsum_accumulate) rsum += *dst;
            *dst = rsum;
        }
    }
}

// Sum the split partials and run the epilogue on the result.  Block = 32 consecutive outputs x 8
// split groups: group q adds splits q, q+8, ... (independent loads in flight instead of one serial
// chain of `splits` dependent round trips), the 8 group sums are combined in a fixed order.
template <class E>
__global__ __launch_bounds__(256) void finish_kernel(SplitSink sink, int splits, E e) {
    __shared__ float part[8][32];
    sink_select_class(sink, blockIdx.z); e.set_class(blockIdx.z);
    const int o = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int j = blockIdx.x * 32 + o, i = blockIdx.y;
    float s = 0.f;
    if (j < sink.J) {
        const float *src = sink.ws + (size_t)i * sink.J + j;
        // unrolled: eight loads in flight, added in the same order (a rolled loop waits a memory latency per split)
#pragma unroll 8
        for (int z = grp; z < splits; z += 8) s += src[(size_t)z * sink.stride];
    }
    part[grp][o] = s;
    __syncthreads();
    if (grp == 0 && j < sink.J && e.col(j)) {
        s = ((part[0][o] + part[1][o]) + (part[2][o] + part[3][o])) +
            ((part[4][o] + part[5][o]) + (part[6][o] + part[7][o]));
        e.put(i, j, s);
    }
    if (sink.rowsum_final && blockIdx.x == 0) {      // block-uniform: the bias gradient of row i, 8 split groups
        __syncthreads();
        if (o == 0) {
            float r = 0.f;
            for (int z = grp; z < splits; z += 8) r += sink.rowsum[(size_t)z * sink.rowsum_stride + i];
            part[grp][0] = r;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            float r = ((part[0][0] + part[1][0]) + (part[2][0] + part[3][0])) +
                      ((part[4][0] + part[5][0]) + (part[6][0] + part[7][0]));
            float *dst = sink.rowsum_final + i;
            if (sink.rowsum_final_accumulate) r += *dst;
            *dst = r;
        }
    }
}

// few splits: one thread per output, the chain is short
template <class E>
__global__ __launch_bounds__(256) void finish_few_kernel(SplitSink sink, int splits, E e) {
    sink_select_class(sink, blockIdx.z); e.set_class(blockIdx.z);
    const int j = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
    if (sink.rowsum_final && j == 0) finish_rowsum(sink, splits, i);
    if (j >= sink.J || !e.col(j)) return;
    float s = 0.f;
#pragma unroll 4
    for (int z = 0; z < splits; ++z) s += sink.ws[(size_t)z * sink.stride + (size_t)i * sink.J + j];
    e.put(i, j, s);
}

// few splits, J % 4 == 0: one thread per 4 consecutive outputs, float4 partial loads
template <class E>
__global__ __launch_bounds__(256) void finish_few_vec_kernel(SplitSink sink, int splits, E e) {
    sink_select_class(sink, blockIdx.y); e.set_class(blockIdx.y);
    const int jq = sink.J >> 2;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t nvec = (size_t)sink.I * jq;
    if (idx >= nvec) {          // the launch carries I extra threads for the bias gradient
        if (sink.rowsum_final && idx < nvec + sink.I) finish_rowsum(sink, splits, (int)(idx - nvec));
        return;
    }
    const int i = (int)(idx / jq), j = (int)(idx - (size_t)i * jq) * 4;
    const float *src = sink.ws + (size_t)i * sink.J + j;
    // MVAE_FINISH_PREFETCH (off): the four outputs' bias / pre-activation / mask requested before the partial sums instead
    // of one by one inside put() (this kernel is 2 % of the FashionMNIST and CelebA steps).  Parity green, one
    // step pair 2.3041 -> 2.3018 ms on FashionMNIST: nothing measurable (profiles/r04_epilogue_ab.txt).
    constexpr bool PRE = ep_prefetch<E>::value && !E::ROWRED && MVAE_FINISH_PREFETCH;
    typename ep_pre<E>::type pre[4];
    if constexpr (PRE) {
#pragma unroll
        for (int c = 0; c < 4; ++c) pre[c] = e.fetch(i, j + c);
    }
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
    for (int z = 0; z < splits; ++z) {       // four loads in flight, added in split order
        const float4 v = *reinterpret_cast<const float4 *>(src + (size_t)z * sink.stride);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    if constexpr (PRE) {
        if (e.col(j)) e.put_pre(i, j, s.x, pre[0]);
        if (e.col(j + 1)) e.put_pre(i, j + 1, s.y, pre[1]);
        if (e.col(j + 2)) e.put_pre(i, j + 2, s.z, pre[2]);
        if (e.col(j + 3)) e.put_pre(i, j + 3, s.w, pre[3]);
    } else {
        if (e.col(j)) e.put(i, j, s.x);
        if (e.col(j + 1)) e.put(i, j + 1, s.y);
        if (e.col(j + 2)) e.put(i, j + 2, s.z);
        if (e.col(j + 3)) e.put(i, j + 3, s.w);
    }
}

// ------------------------------------------------------------------------------------------
// host-side dispatch
// ------------------------------------------------------------------------------------------
// Tuning build + MVAE_GRID_REPORT=1 in the environment: one stderr line per GEMM-shaped launch -- blocks, blocks a CU holds
// (hipOccupancyMaxActiveBlocksPerMultiprocessor), and what share of the launch's block-slots x rounds does work: the tile /
// block-count quantisation of a launch before a single instruction runs (tools/grid_report.py tabulates it).
#ifdef MVAE_TUNING
#define MVAE_GRID_REPORT(kern, grid, NT, lds, I, J, K, items, TM, TN)                                  \
    {                                                                                            \
        static const bool rep_on = getenv("MVAE_GRID_REPORT") != nullptr;                        \
        if (rep_on) {                                                                            \
            int per_cu = 0;                                                                      \
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(kern), NT, lds); \
            const long blocks = (long)grid.x * grid.y * grid.z, slots = 256L * (per_cu > 0 ? per_cu : 1); \
            const long rounds = (blocks + slots - 1) / slots;                                    \
            /* a CU's share: blocks go round the CUs; the launch lasts as long as the CU with the most */ \
            const long busiest = (blocks + 255) / 256;                                           \
            fprintf(stderr, "[grid] tile %dx%d I %d J %d K %d  blocks %ld (%u x %u x %u) threads %d lds %zu  per_cu %d  rounds %ld  busiest_cu %ld  balance %.3f  items %d  %s\n", \
                    TM, TN, I, J, K, blocks, grid.x, grid.y, grid.z, NT, (size_t)lds, per_cu, rounds, busiest, \
                    (double)blocks / (256.0 * busiest), items, __PRETTY_FUNCTION__);              \
        }                                                                                        \
    }
#else
#define MVAE_GRID_REPORT(kern, grid, NT, lds, I, J, K, items, TM, TN)
#endif
struct Plan { int wm, wn, wgm, wgn, kw, bk, splits, klen; int xcd = 0; int items = 1; int force_items = 0; };   // xcd: launch-order re-mapping the host asks for (1 Linear sub-grids, 3 conv items, 4 split k ranges; see igemm_kernel); items: (class, j tile) items per block (conv forms)

inline long cdiv(long a, long b) { return (a + b - 1) / b; }

// Tile and split choice, from the measurements in profiles/ (tools/gemm_bench.py):
// the gather-fed forward / dgrad forms and the large Linear layers run best on 64x64 tiles (5 waves per
// SIMD hide the gather latency; 128-wide tiles drop to 2-3 waves), the conv weight-gradient form
// (two gathers feeding a small output over a huge reduction) wants the arithmetic intensity of
// 128-wide tiles.  Reductions are split until ~4 blocks per CU exist (>= 2 k-steps per split).
// `small_ok` (the float4-loadable Linear forms): outputs of fewer than 256 tiles of 64x64 whose reduction
// is short or whose 32x32 tiles still cover the chip use the small layouts -- one launch, no split.
enum PlanKind { PLAN_FWD = 0, PLAN_CONV_WGRAD = 1, PLAN_LIN_WGRAD = 2 };

inline Plan make_plan(int I, int J, int K, bool allow_split, PlanKind kind = PLAN_FWD, int ncls = 1,
                      bool small_ok = false) {
    Plan p;
    p.wm = 1; p.wn = 1; p.wgm = 2; p.wgn = 2; p.kw = 1; p.bk = BK;
    const bool forced = MVAE_TUNE(wm) || MVAE_TUNE(wn) || MVAE_TUNE(kw);
    if (small_ok && !forced && !MVAE_TUNE(small_off)) {
        const long t64 = cdiv(I, 64) * cdiv(J, 64) * ncls, t32 = cdiv(I, 32) * cdiv(J, 32) * ncls;
        if (t64 < 256 && (t32 >= 192 || K <= 1024) && K < 4096) {
            // largest tile that still gives (nearly) every CU a block; between the two 2-tile shapes the one
            // that pads less (ties: 32x64 -- the j axis is the contiguous store axis)
            const long b_tall = cdiv(I, 64) * cdiv(J, 32) * ncls, b_wide = cdiv(I, 32) * cdiv(J, 64) * ncls;
            long blocks;
            if ((b_tall > b_wide ? b_tall : b_wide) >= 224) {
                const bool tall = b_tall < b_wide;            // fewer blocks of equal size = less padding
                p.wgm = tall ? 2 : 1; p.wgn = tall ? 1 : 2;
                blocks = tall ? b_tall : b_wide;
            } else {
                p.wgm = 1; p.wgn = 1;
                blocks = t32;
            }
            // one block per CU: 8 waves (two per SIMD); more blocks than CUs: 4 waves each
            int waves = blocks <= 320 ? 8 : 4;
            if (MVAE_TUNE(small_waves)) waves = MVAE_TUNE(small_waves);
            p.kw = waves / (p.wgm * p.wgn);
            p.bk = 64;
            p.splits = 1;
            p.klen = (int)(cdiv(K, p.bk) * p.bk);
            return p;
        }
    }
    const bool narrow = (I <= 32 && J >= 128 && !forced);
    if (narrow) { p.wgm = 1; p.wgn = 4; }
    if (kind == PLAN_CONV_WGRAD && !narrow) {
        // MVAE_WGRAD_TILE (A/B builds): 128 = up to 128 x 128 tiles (default), 96 = 64 x 128, 64 = 64 x 64 -- smaller
        // tiles reach the block target with fewer k ranges, i.e. fewer partial slabs for the finish launch to re-read
        if (J >= 128 && MVAE_WGRAD_TILE >= 96) p.wn = 2;
        if (I >= 128 && J >= 128 && MVAE_WGRAD_TILE >= 128) p.wm = 2;
    }
    // long reductions into a small output (FashionMNIST's 6272-wide layers: 1024 x 512 over K = 6272): 64 x 128
    // tiles, split over K, beat the k-grouped small layouts (86 -> 101, 91 -> 100 TFLOP/s)
    if (kind == PLAN_FWD && allow_split && !narrow && !forced && K >= 4096 && J >= 128 &&
        cdiv(I, 64) * cdiv(J, 64) * ncls <= 256 && cdiv(I, 64) * cdiv(J, 64) * ncls >= 64)
        p.wn = 2;
    if (MVAE_TUNE(wm)) p.wm = MVAE_TUNE(wm);
    if (MVAE_TUNE(wn)) p.wn = MVAE_TUNE(wn);
    const long tiles = narrow ? cdiv(J, 128) * ncls : cdiv(I, 64 * p.wm) * cdiv(J, 64 * p.wn) * ncls;
    // fewer than one wave per SIMD (256 CUs x 4): let KW wave groups share each 64x64 tile
    if (!narrow && p.wm == 1 && p.wn == 1 && K >= 4 * BK) {
        if (tiles * 4 <= 256) p.kw = 4;
        else if (tiles * 2 <= 256) p.kw = 2;
    }
    if (MVAE_TUNE(kw)) p.kw = (p.wm == 1 && p.wn == 1) ? MVAE_TUNE(kw) : 1;
    long want = 1;
    if (allow_split) {
        // blocks to aim for (profiles/r01_split_sweep.txt): the Linear forward / dgrad forms want every
        // CU busy and then as FEW splits as possible (each block keeps >= 8 k-steps, the finish reads
        // less); the 128-wide conv weight-gradient tiles run two blocks per CU; everything else (Linear
        // weight gradients, small conv outputs) is fastest with ~4 blocks per CU
        const long target_tune = MVAE_TUNE(split_target);
        long target_blocks = 1024 / p.kw;
        if (kind == PLAN_FWD) target_blocks = (p.kw == 1) ? 512 : 256;
        if (kind == PLAN_CONV_WGRAD && p.wm * p.wn >= 2) target_blocks = MVAE_WGRAD_TARGET;
        if (target_tune > 0) target_blocks = target_tune / p.kw;
        want = (target_blocks + tiles / 2) / tiles;         // nearest: 800 tiles against 1024 is one round, not two
        const long maxs = (K + 2 * BK - 1) / (2 * BK);
        if (want > maxs) want = maxs;
        if (want < 1) want = 1;
        if (kind == PLAN_LIN_WGRAD && !MVAE_TUNE(kw) && !target_tune) {
            // Linear weight gradients: plain 4-wave blocks, ~2 per CU, >= 8 k-steps each beat k-wave
            // groups and deeper splits (512x512 over M = 1024: 36 vs 33 TFLOP/s; 784x512: 40 vs 32) --
            // unless the reduction is too short to make enough blocks that way
            const long maxs8 = K / (8 * BK) > 0 ? K / (8 * BK) : 1;
            long w1 = (512 + tiles / 2) / tiles;
            if (w1 > maxs8) w1 = maxs8;
            if (w1 < 1) w1 = 1;
            if (tiles * w1 >= 128) { p.kw = 1; want = w1; }
        }
        if (want > (tiles <= 4 ? 512 : 64)) want = (tiles <= 4 ? 512 : 64);   // tiny outputs may split deeper
        if (MVAE_TUNE(splits) > 0) want = MVAE_TUNE(splits);
        if (MVAE_TUNE(splits) < 0 && kind == PLAN_FWD) want = 1;      // tuning: forward / dgrad forms never split
    }
    p.klen = (int)(((K + want - 1) / want + BK - 1) / BK * BK);
    p.splits = (K + p.klen - 1) / p.klen;
    return p;
}

// PL / QL: loader templates of the BK = 32 layouts; PS / QS: their BK = 64 versions for the small
// layouts (SMALL = false: the caller has none -- the conv forms -- and the plan never asks for one).
template <template <int> class PL, template <int> class QL, class E, bool ROWSUM, bool SMALL,
          template <int> class PS, template <int> class QS, class PF, class QF>
int launch_igemm_impl(Plan pl, PF make_p, QF make_q, E e, int I, int J, int K, SplitSink sink, hipStream_t st) {
#define MVAE_LAUNCH(PLD, QLD, WM, WN, KW, WGM, WGN)                                              \
    {                                                                                            \
        constexpr int TM = 32 * WM * WGM, TN = 32 * WN * WGN, NT = 64 * WGM * WGN * KW;          \
        PLD<TM> p; make_p(p);                                                                    \
        QLD<TN> q; make_q(q);                                                                    \
        dim3 grid(((J + TN - 1) / TN) * sink.ncls, (I + TM - 1) / TM, pl.splits);                \
        sink.tiles_j = (J + TN - 1) / TN;                                                        \
        sink.xcd_map = (pl.xcd == 1 && sink.ncls == 1 && pl.splits == 1 && grid.x % 2 == 0 && grid.y % 4 == 0) ? 1 : 0; \
        if (MVAE_XCD_ROWS && pl.xcd == 1 && sink.ncls == 1 && pl.splits == 1 && grid.y % 8 == 0) sink.xcd_map = 2; \
        if (pl.xcd == 4 && pl.splits >= 8) { sink.xcd_map = 4; grid.z = (grid.z + 7) / 8 * 8; } /* k ranges XCD-local */ \
        sink.items = 1;                                                                          \
        if (pl.items > 1 && E::MULTI && KW == 1 && !ROWSUM && pl.splits == 1 && NT == NTHREADS && K % PLD<TM>::BKV == 0 && \
            K >= 2 * PLD<TM>::BKV && PLD<TM>::PARTS && QLD<TN>::PARTS) {                         \
            int items = pl.items;                      /* keep >= MVAE_MULTI_MINBLOCKS column blocks */ \
            while (!pl.force_items && items > 1 && (int)grid.x / items < MVAE_MULTI_MINBLOCKS) items >>= 1; \
            sink.items = items;                                                                  \
            grid.x = (grid.x + items - 1) / items;                                               \
        }                                                                                        \
        if (pl.xcd == 3 && pl.splits == 1 && grid.y > 1 && grid.x >= 64) {                      \
            sink.xcd_map = 3; grid.x = (grid.x + 7) / 8 * 8;    /* (class, j tile) items XCD-local */ \
        }                                                                                        \
        if (MVAE_PAIR_NEIGH && E::PAIR && WM * WN == 1 && sink.items > 1 && sink.ncls == 4 && sink.cls_minor && grid.y == 1 && \
            pl.splits == 1 && sink.xcd_map == 0) {                                                \
            sink.items = 2; sink.xcd_map = 5;                                                    \
            grid.x = ((unsigned)sink.tiles_j * 2u + 15u) / 16u * 16u;                            \
        }                                                                                        \
        constexpr size_t tile_b = 2 * (PLD<TM>::ROWS * PLD<TM>::PITCH + QLD<TN>::ROWS * QLD<TN>::PITCH) * sizeof(float); \
        constexpr size_t red_b = (WGM * WGN < 4 && KW > 1)                                       \
            ? (size_t)KW * TM * (TN + 1) * sizeof(float)              /* cooperative epilogue */ \
            : (size_t)(KW - 1) * WGM * WGN * WM * WN * 16 * 64 * sizeof(float);                  \
        constexpr size_t lds = tile_b > red_b ? tile_b : red_b;                                  \
        auto kern = igemm_kernel<PLD<TM>, QLD<TN>, E, WM, WN, ROWSUM, KW, WGM, WGN>;             \
        static bool attr_done = false;   /* idempotent: a race only repeats the same call */    \
        if (!attr_done) {                                                                        \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern),                      \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);     \
            attr_done = true;                                                                    \
        }                                                                                        \
        MVAE_GRID_REPORT(kern, grid, NT, lds, I, J, K, sink.items, TM, TN)                               \
        hipLaunchKernelGGL(kern, grid, dim3(NT), lds, st, p, q, e, K, pl.klen, sink);            \
    }
    bool launched = false;
    if constexpr (SMALL) {
        if (pl.bk == 64) {
            launched = true;
            if (pl.wgm == 2 && pl.kw == 4) MVAE_LAUNCH(PS, QS, 1, 1, 4, 2, 1)
            else if (pl.wgm == 2) MVAE_LAUNCH(PS, QS, 1, 1, 2, 2, 1)
            else if (pl.wgn == 2 && pl.kw == 4) MVAE_LAUNCH(PS, QS, 1, 1, 4, 1, 2)
            else if (pl.wgn == 2) MVAE_LAUNCH(PS, QS, 1, 1, 2, 1, 2)
            else if (pl.kw == 8) MVAE_LAUNCH(PS, QS, 1, 1, 8, 1, 1)
            else MVAE_LAUNCH(PS, QS, 1, 1, 4, 1, 1)
        }
    }
    if (!launched) {
        if (pl.bk != BK) return MVAE_ERR_ARG;
        if (pl.wgn == 4) MVAE_LAUNCH(PL, QL, 1, 1, 1, 1, 4)
        else if (pl.wm == 2 && pl.wn == 2) MVAE_LAUNCH(PL, QL, 2, 2, 1, 2, 2)
        else if (pl.wm == 2 && pl.wn == 1) MVAE_LAUNCH(PL, QL, 2, 1, 1, 2, 2)
        else if (pl.wm == 1 && pl.wn == 2) MVAE_LAUNCH(PL, QL, 1, 2, 1, 2, 2)
        else if (pl.kw == 4) MVAE_LAUNCH(PL, QL, 1, 1, 4, 2, 2)
        else if (pl.kw == 2) MVAE_LAUNCH(PL, QL, 1, 1, 2, 2, 2)
        else MVAE_LAUNCH(PL, QL, 1, 1, 1, 2, 2)
    }
#undef MVAE_LAUNCH
    if (pl.splits > 1) {
        if (pl.splits > 16) {
            dim3 grid((J + 31) / 32, I, sink.ncls);
            hipLaunchKernelGGL((finish_kernel<E>), grid, dim3(256), 0, st, sink, pl.splits, e);
        } else if (J % 4 == 0 && sink.stride % 4 == 0 && aligned16(sink.ws)) {
            const size_t nvec = (size_t)I * (J / 4) + (sink.rowsum_final ? I : 0);
            hipLaunchKernelGGL((finish_few_vec_kernel<E>), dim3((unsigned)((nvec + 255) / 256), sink.ncls), dim3(256), 0, st, sink,
                               pl.splits, e);
        } else {
            dim3 grid((J + 255) / 256, I, sink.ncls);
            hipLaunchKernelGGL((finish_few_kernel<E>), grid, dim3(256), 0, st, sink, pl.splits, e);
        }
    }
    return mvae_launch_status();
}

template <template <int> class PL, template <int> class QL, class E, bool ROWSUM, class PF, class QF>
int launch_igemm(Plan pl, PF make_p, QF make_q, E e, int I, int J, int K, SplitSink sink, hipStream_t st) {
    return launch_igemm_impl<PL, QL, E, ROWSUM, false, PL, QL>(pl, make_p, make_q, e, I, J, K, sink, st);
}

template <template <int> class PL, template <int> class QL, template <int> class PS, template <int> class QS,
          class E, bool ROWSUM, class PF, class QF>
int launch_igemm_small(Plan pl, PF make_p, QF make_q, E e, int I, int J, int K, SplitSink sink, hipStream_t st) {
    return launch_igemm_impl<PL, QL, E, ROWSUM, true, PS, QS>(pl, make_p, make_q, e, I, J, K, sink, st);
}

inline SplitSink make_sink(void *ws, int I, int J, bool rowsum) {
    SplitSink s;
    s.ws = (float *)ws; s.I = I; s.J = J;
    s.stride = (size_t)I * J + (rowsum ? I : 0);
    s.rowsum = nullptr; s.rowsum_stride = 0; s.rowsum_accumulate = 0; s.ncls = 1; s.rowsum_cls_stride = 0;
    s.rowsum_final = nullptr; s.rowsum_final_accumulate = 0; s.cls_region = 0; s.rowsum_final_cls_stride = 0;
    s.xcd_map = 0; s.tiles_j = 0; s.items = 1; s.cls_minor = 0;
    return s;
}

inline size_t split_ws_floats(int I, int J, int K) {
    size_t best = 0;      // the caller does not say which op it sizes for: take the largest plan
    for (int kind = 0; kind < 3; ++kind) {
        Plan pl = make_plan(I, J, K, true, (PlanKind)kind);
        if (pl.splits > 1 && (size_t)pl.splits > best) best = pl.splits;
    }
    return best * ((size_t)I * J + I);
}

}  // namespace
