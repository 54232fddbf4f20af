// gemm.hip -- the dense contractions of the MVAE train step on fp32 MFMA (gfx950).
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
// Tiling: 256 threads = 4 waves (2 x 2), each wave WM x WN MFMA tiles of 32x32
// (block tile 64*WM x 64*WN), BK = 32.  Software pipeline: global -> registers two k-tiles
// ahead (two register sets), registers -> LDS one tile ahead (two LDS buffers), ONE barrier per
// k-step; the fp32 MFMA (64 cycles each) of tile t hides the HBM/L2 latency of tile t+2.
// LDS tiles are [BK][tile + 4]: fragment reads are bank-conflict free (ds_read_b32, lanes
// 0..31 consecutive), float4 tile rows stay 16-byte aligned.
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
#include <cstdlib>

#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

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

// S[r * ld + k]: reduction axis contiguous (x and w of Linear fwd, dy of dgrad, conv weights).
// VEC: base 16-byte aligned, ld % 4 == 0 and Klen % 4 == 0 (float4 loads never straddle the end).
template <int TILE_, bool VEC>
struct LdRowsKT {
    static constexpr int TILE = TILE_;
    static constexpr int NV = TILE * BK / 4 / NTHREADS;
    struct Regs { float4 v[NV]; float4 m[NV]; };      // raw data + 0/1 masks (applied when staged)
    const float *src; int ld; int R; int Klen;
    size_t cls_stride = 0;                            // per-class (group) source offset
    int r0;
    __device__ void init(int tile0, int, int cls) { r0 = tile0; src += (size_t)cls * cls_stride; }
    __device__ void load(int k0, int kend, int t, Regs &rg) const {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int f = t + NTHREADS * v;
            const int r = r0 + f / (BK / 4), k = k0 + (f % (BK / 4)) * 4;
            const float *row = src + (size_t)min(r, R - 1) * ld;
            float4 x;
            if (VEC) {
                x = *reinterpret_cast<const float4 *>(row + min(k, Klen - 4));
                const float m = mask0(r < R && k < kend);
                rg.m[v] = make_float4(m, m, m, m);
            } else {
                x.x = row[min(k, Klen - 1)];     x.y = row[min(k + 1, Klen - 1)];
                x.z = row[min(k + 2, Klen - 1)]; x.w = row[min(k + 3, Klen - 1)];
                const bool rin = r < R;
                rg.m[v] = make_float4(mask0(rin && k < kend), mask0(rin && k + 1 < kend),
                                      mask0(rin && k + 2 < kend), mask0(rin && k + 3 < kend));
            }
            rg.v[v] = x;
        }
    }
    __device__ void store(float (*L)[TILE + LPAD], int t, const Regs &rg) const {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int f = t + NTHREADS * v;
            const int r = f / (BK / 4), kc = (f % (BK / 4)) * 4;
            L[kc + 0][r] = rg.v[v].x * rg.m[v].x; L[kc + 1][r] = rg.v[v].y * rg.m[v].y;
            L[kc + 2][r] = rg.v[v].z * rg.m[v].z; L[kc + 3][r] = rg.v[v].w * rg.m[v].w;
        }
    }
};
template <int T> using LdRowsK = LdRowsKT<T, true>;
template <int T> using LdRowsKS = LdRowsKT<T, false>;

// S[k * ld + r]: non-reduced axis contiguous (w of dgrad, dy and x of wgrad, repacked conv weights).
// VEC: base 16-byte aligned, ld % 4 == 0 and R % 4 == 0.
template <int TILE_, bool VEC>
struct LdRowsMNT {
    static constexpr int TILE = TILE_;
    static constexpr int NV = TILE * BK / 4 / NTHREADS;
    static constexpr int V4 = TILE / 4;     // float4 per k row
    struct Regs { float4 v[NV]; float4 m[NV]; };
    const float *src; int ld; int R; int Klen; size_t cls_stride;   // cls_stride: per-class source offset
    int r0;
    __device__ void init(int tile0, int, int cls) { r0 = tile0; src += (size_t)cls * cls_stride; }
    __device__ void load(int k0, int kend, int t, Regs &rg) const {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int f = t + NTHREADS * v;
            const int k = k0 + f / V4, r = r0 + (f % V4) * 4;
            const float *row = src + (size_t)min(k, Klen - 1) * ld;
            float4 x;
            if (VEC) {
                x = *reinterpret_cast<const float4 *>(row + min(r, R - 4));
                const float m = mask0(k < kend && r < R);
                rg.m[v] = make_float4(m, m, m, m);
            } else {
                x.x = row[min(r, R - 1)];     x.y = row[min(r + 1, R - 1)];
                x.z = row[min(r + 2, R - 1)]; x.w = row[min(r + 3, R - 1)];
                const bool kin = k < kend;
                rg.m[v] = make_float4(mask0(kin && r < R), mask0(kin && r + 1 < R),
                                      mask0(kin && r + 2 < R), mask0(kin && r + 3 < R));
            }
            rg.v[v] = x;
        }
    }
    __device__ void store(float (*L)[TILE + LPAD], int t, const Regs &rg) const {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int f = t + NTHREADS * v;
            const float4 x = rg.v[v], m = rg.m[v];
            *reinterpret_cast<float4 *>(&L[f / V4][(f % V4) * 4]) = make_float4(x.x * m.x, x.y * m.y, x.z * m.z, x.w * m.w);
        }
    }
};
template <int T> using LdRowsMN = LdRowsMNT<T, true>;
template <int T> using LdRowsMNS = LdRowsMNT<T, false>;

// Geometry of a 4x4 convolution y[B,Cout,OH,OW] = conv(x[B,Cin,H,W], w[Cout,Cin,4,4]).
struct ConvGeom {
    int B, Cin, H, W, Cout, OH, OW, stride, pad;
};

// ---- gather loaders ----------------------------------------------------------------------
// The k index of an element a thread fetches is  k0 + kq + STEP*v  with k0 a multiple of BK = 32,
// kq = thread-constant (< STEP) and v the unrolled element counter.  STEP is a power of two, so the
// (channel, tap-row, tap-col) fields of k are the OR of compile-time fields of STEP*v and the
// thread-constant fields of kq: per element the address is ONE add of a wave-uniform offset, the
// bounds test a compile-time shift of a precomputed bit mask.  (The first version decoded k and
// re-tested the image bounds per element and was VALU-bound: 5 waves x ~250 VALU ops per k-step
// against 1024 MFMA cycles.)

// im2col of x for the forward conv: element (k = (ci,kh,kw), m = (b,oh,ow)); lanes along m.
template <int TILE_>
struct LdIm2col {
    static constexpr int TILE = TILE_;
    static constexpr int NV = TILE * BK / NTHREADS;   // elements per thread
    static constexpr int KSTEP = NTHREADS / TILE;     // 2 or 4: k rows covered per pass
    struct Regs { float v[NV]; unsigned ok; };        // raw data + validity bits (applied when staged)
    const float *x; ConvGeom g; int Mtot;
    int base, kq; unsigned vh, vwq;
    __device__ void init(int tile0, int t, int) {
        const int m = tile0 + (t % TILE);
        kq = t / TILE;
        unsigned vw = 0;
        vh = 0; base = 0;
        if (m < Mtot) {
            const int ohw = g.OH * g.OW;
            const int b = m / ohw, rem = m - b * ohw;
            const int oh = rem / g.OW, ow = rem - oh * g.OW;
            const int ih0 = oh * g.stride - g.pad, iw0 = ow * g.stride - g.pad;
            base = (b * g.Cin * g.H + ih0) * g.W + iw0 + kq;      // kw = kq + (KSTEP*v & 3)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (ih0 + q >= 0 && ih0 + q < g.H) vh |= 1u << q;
                if (iw0 + q >= 0 && iw0 + q < g.W) vw |= 1u << q;
            }
        }
        vwq = vw >> kq;
    }
    __device__ void load(int k0, int kend, int t, Regs &rg) const {
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
    __device__ void store(float (*L)[TILE + LPAD], int t, const Regs &rg) const {
        const int m = t % TILE, kb = t / TILE;
#pragma unroll
        for (int v = 0; v < NV; ++v) L[kb + v * KSTEP][m] = rg.v[v] * mask0((rg.ok >> v) & 1u);
    }
};

// Transposed-conv (dgrad) gather of dy for the output parity class `cls` = (ph,pw) of dx:
// element (k = (co,a,b), m = (n, ih', iw')) with ih = ih'*s + ph, kh = kh0 + s*a,
// oh = (ih + pad - kh0)/s - a.  TPD = 4/s taps per dim (TLOG = log2 TPD); only the taps that can
// reach the class are enumerated, so stride 2 does no multiply-by-zero work.
template <int TILE_, int TLOG>
struct LdDgradDyT {
    static constexpr int TILE = TILE_;
    static constexpr int NV = TILE * BK / NTHREADS;
    static constexpr int KSTEP = NTHREADS / TILE;
    static constexpr int TMASK = (1 << TLOG) - 1;
    struct Regs { float v[NV]; unsigned ok; };        // raw data + validity bits (applied when staged)
    const float *dy; ConvGeom g; int Mtot; int H2, W2;
    int base, kq; unsigned vhq, vwq;
    __device__ void init(int tile0, int t, int cls) {
        const int ph = cls / g.stride, pw = cls % g.stride;
        const int kh0 = (ph + g.pad) % g.stride, kw0 = (pw + g.pad) % g.stride;
        const int m = tile0 + (t % TILE);
        kq = t / TILE;
        const int aq = (kq >> TLOG) & TMASK, bq = kq & TMASK;     // thread-constant tap fields
        unsigned vh = 0, vw = 0;
        base = 0;
        if (m < Mtot) {
            const int hw2 = H2 * W2;
            const int n = m / hw2, rem = m - n * hw2;
            const int ih2 = rem / W2, iw2 = rem - ih2 * W2;
            const int ohb = (ih2 * g.stride + ph + g.pad - kh0) / g.stride;
            const int owb = (iw2 * g.stride + pw + g.pad - kw0) / g.stride;
            base = (n * g.Cout * g.OH + ohb - aq) * g.OW + owb - bq;
#pragma unroll
            for (int a = 0; a <= TMASK; ++a) {
                if (ohb - a >= 0 && ohb - a < g.OH) vh |= 1u << a;
                if (owb - a >= 0 && owb - a < g.OW) vw |= 1u << a;
            }
        }
        vhq = vh >> aq; vwq = vw >> bq;
    }
    __device__ void load(int k0, int kend, int t, Regs &rg) const {
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
    __device__ void store(float (*L)[TILE + LPAD], int t, const Regs &rg) const {
        const int m = t % TILE, kb = t / TILE;
#pragma unroll
        for (int v = 0; v < NV; ++v) L[kb + v * KSTEP][m] = rg.v[v] * mask0((rg.ok >> v) & 1u);
    }
};
template <int TILE_> using LdDgradDyS2 = LdDgradDyT<TILE_, 1>;   // stride 2: 2x2 taps per class
template <int TILE_> using LdDgradDyS1 = LdDgradDyT<TILE_, 2>;   // stride 1: all 4x4 taps

// wgrad operands: the reduction runs over k = (b,oh,ow); lanes along k (spatially contiguous).
// P: element (i = co, k) = dy[b][co][oh][ow].
template <int TILE_>
struct LdWgradDy {
    static constexpr int TILE = TILE_;
    static constexpr int NV = TILE * BK / NTHREADS;
    static constexpr int ISTEP = NTHREADS / BK;
    struct Regs { float v[NV]; unsigned ok; };        // raw data + validity bits (applied when staged)
    const float *dy; ConvGeom g;
    int ioff, nvalid;       // ioff = i * OHW of element 0; nvalid = how many of the NV rows are < Cout
    __device__ void init(int tile0, int t, int) {
        const int ib = tile0 + t / BK;
        ioff = ib * g.OH * g.OW;
        nvalid = (g.Cout - ib + ISTEP - 1) / ISTEP;     // rows ib + v*ISTEP < Cout  <=>  v < nvalid
    }
    __device__ void load(int k0, int kend, int t, Regs &rg) const {
        const int k = k0 + (t % BK);
        const int ohw = g.OH * g.OW;
        const int b = k / ohw, sp = k - b * ohw;
        const float *src = dy + (size_t)b * g.Cout * ohw + sp + ioff;
        const int safe = (int)(dy - src);
        const int nv = (k < kend) ? nvalid : 0;
        unsigned okbits = 0;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const float val = src[(v < nv) ? v * ISTEP * ohw : safe];
            rg.v[v] = val;
            okbits |= ((v < nv) ? 1u : 0u) << v;
        }
        rg.ok = okbits;
    }
    __device__ void store(float (*L)[TILE + LPAD], int t, const Regs &rg) const {
        const int kl = t % BK, ib = t / BK;
#pragma unroll
        for (int v = 0; v < NV; ++v) L[kl][ib + v * ISTEP] = rg.v[v] * mask0((rg.ok >> v) & 1u);
    }
};

// Q: element (k, j = (ci,kh,kw)) = x[b][ci][oh*s-p+kh][ow*s-p+kw];  j = j0 + jq + 8*v, jq = t/32 < 8.
template <int TILE_>
struct LdWgradX {
    static constexpr int TILE = TILE_;
    static constexpr int NV = TILE * BK / NTHREADS;
    static constexpr int JSTEP = NTHREADS / BK;       // 8
    struct Regs { float v[NV]; unsigned ok; };        // raw data + validity bits (applied when staged)
    const float *x; ConvGeom g; int J;
    int joff, jq, nvalid;
    __device__ void init(int tile0, int t, int) {
        jq = t / BK;                                  // kw = jq & 3, kh = (jq >> 2) + 2*(v & 1), ci = j0/16 + (v >> 1)
        joff = (tile0 >> 4) * g.H * g.W + (jq >> 2) * g.W + (jq & 3);
        nvalid = (J - tile0 - jq + JSTEP - 1) / JSTEP;
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
            const float val = src[ok ? (v >> 1) * hw + 2 * (v & 1) * g.W : safe];
            rg.v[v] = val;
            okbits |= (ok ? 1u : 0u) << v;
        }
        rg.ok = okbits;
    }
    __device__ void store(float (*L)[TILE + LPAD], int t, const Regs &rg) const {
        const int kl = t % BK, jb = t / BK;
#pragma unroll
        for (int v = 0; v < NV; ++v) L[kl][jb + v * JSTEP] = rg.v[v] * mask0((rg.ok >> v) & 1u);
    }
};

// ------------------------------------------------------------------------------------------
// epilogues:  col(j) prepares the lane's column, put(i, j, v) consumes one element.
// ------------------------------------------------------------------------------------------

// Row-major destination D[i * ld + j] with the Linear fusions.
struct EpRowMajor {
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
};

// NCHW destination: i = channel, j = (n, row', col') of a (possibly strided) sub-lattice:
// address = (n * C + i) * HW + (row' * s + py) * Wfull + col' * s + px.
struct EpNCHW {
    float *out; float *act; const float *dpre;
    int C, HW, Wfull, H2, W2, sy, py, px, J;
    int off;   // per-lane column offset, set by col()
    __device__ void set_class(int cls) { if (sy > 1) { py = cls / sy; px = cls % sy; } }
    __device__ bool col(int j) {
        if (j >= J) return false;
        const int hw2 = H2 * W2;
        const int n = j / hw2, rem = j - n * hw2;
        const int r = rem / W2, c = rem - r * W2;
        off = n * C * HW + (r * sy + py) * Wfull + c * sy + px;
        return true;
    }
    __device__ void put(int i, int, float v) const {
        if (i >= C) return;
        const int idx = off + i * HW;
        if (dpre) v *= swish_grad_(dpre[idx]);
        if (out) out[idx] = v;
        if (act) act[idx] = swishf_(v);
    }
};

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

// ------------------------------------------------------------------------------------------
// the kernel
// ------------------------------------------------------------------------------------------
// KW > 1 (64x64 tiles only): KW groups of 4 waves share the tile; group kg takes every KW-th
// k-pair of each LDS tile, the partial accumulators are summed through LDS at the end.  Small
// GEMMs (the 512-wide MLP layers at batch 512-1024) have < 256 tiles, i.e. fewer than one wave per
// SIMD: the extra wave groups put all 1024 SIMDs to work and shorten each block's serial MFMA
// chain KW-fold without a second launch.  Waves >= 4 only issue MFMAs; waves 0-3 also move data.
// NARROW (WM = WN = 1): the four waves sit side by side along j, block tile 32 x 128 -- for outputs
// with <= 32 rows (32-channel conv layers, the 32x48 weight gradients) a 64-row tile is half padding.
template <class P, class Q, class E, int WM, int WN, bool ROWSUM, int KW, bool NARROW>
__global__ __launch_bounds__(NTHREADS * KW, (KW == 1 ? 2 : 1)) void igemm_kernel(P p, Q q, E e, int K, int klen,
                                                                               SplitSink sink) {
    constexpr int BM = NARROW ? 32 : 64 * WM, BN = NARROW ? 128 : 64 * WN;
    static_assert(P::TILE == BM && Q::TILE == BN, "loader tile mismatch");
    // dynamic LDS (the 128x128 tile needs 66 KiB, above the 64 KiB static limit); the only LDS
    // object of the kernel, so its base is 16-byte aligned (cdna_hip_programming.md G17)
    extern __shared__ __attribute__((aligned(16))) float lds_raw[];
    typedef float (*PTile)[BM + LPAD];
    typedef float (*QTile)[BN + LPAD];
    PTile Ps[2] = {reinterpret_cast<PTile>(lds_raw), reinterpret_cast<PTile>(lds_raw + BK * (BM + LPAD))};
    float *qbase = lds_raw + 2 * BK * (BM + LPAD);
    QTile Qs[2] = {reinterpret_cast<QTile>(qbase), reinterpret_cast<QTile>(qbase + BK * (BN + LPAD))};

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int kg = wave >> 2;                       // k-group of this wave (0 when KW == 1)
    const int wi = NARROW ? 0 : (wave & 3) >> 1, wj = NARROW ? (wave & 3) : (wave & 1);
    const bool mover = (KW == 1) || t < NTHREADS;   // waves 0-3 fetch and stage the tiles
    const int tiles_j = gridDim.x / sink.ncls;
    const int cls = blockIdx.x / tiles_j;
    const int i0 = blockIdx.y * BM, j0 = (blockIdx.x - cls * tiles_j) * BN, split = blockIdx.z;
    const int kbeg = split * klen;
    const int kend = min(K, kbeg + klen);
    const int nsteps = (kend - kbeg + BK - 1) / BK;

    f32x16 acc[WM][WN];
#pragma unroll
    for (int a = 0; a < WM; ++a)
#pragma unroll
        for (int b = 0; b < WN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    p.init(i0, t, cls);
    q.init(j0, t, cls);
    e.set_class(cls);
    typename P::Regs pr0, pr1;
    typename Q::Regs qr0, qr1;
    float rsum = 0.f;
    const int lrow = lane >> 5, lcol = lane & 31;

    auto compute = [&](int buf) {
        if (ROWSUM) {   // db = sum over the reduction axis of P (dy^T): the bias gradient for free
            if (blockIdx.x == cls * tiles_j && t < BM) {      // (t < BM <= 256: always a wave of k-group 0)
#pragma unroll
                for (int kk = 0; kk < BK; ++kk) rsum += Ps[buf][kk][t];
            }
        }
        // software-pipelined fragment reads: the ds_reads of k-pair kq+1 are issued before the MFMAs
        // of k-pair kq, so the LDS latency hides behind 64-cycle matrix instructions
        constexpr int NKK = BK / 2 / KW;
        float a0[WM], b0[WN];
#pragma unroll
        for (int x = 0; x < WM; ++x) a0[x] = Ps[buf][kg * 2 + lrow][(wi * WM + x) * 32 + lcol];
#pragma unroll
        for (int y = 0; y < WN; ++y) b0[y] = Qs[buf][kg * 2 + lrow][(wj * WN + y) * 32 + lcol];
#pragma unroll
        for (int kq = 0; kq < NKK; ++kq) {
            float a1[WM], b1[WN];
            if (kq + 1 < NKK) {
                const int kk = (kq + 1) * KW + kg;
#pragma unroll
                for (int x = 0; x < WM; ++x) a1[x] = Ps[buf][kk * 2 + lrow][(wi * WM + x) * 32 + lcol];
#pragma unroll
                for (int y = 0; y < WN; ++y) b1[y] = Qs[buf][kk * 2 + lrow][(wj * WN + y) * 32 + lcol];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int x = 0; x < WM; ++x)
#pragma unroll
                for (int y = 0; y < WN; ++y)
                    acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[x], b0[y], acc[x][y], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (kq + 1 < NKK) {
#pragma unroll
                for (int x = 0; x < WM; ++x) a0[x] = a1[x];
#pragma unroll
                for (int y = 0; y < WN; ++y) b0[y] = b1[y];
            }
        }
    };

    // prologue: tiles 0 and 1 in flight, tile 0 staged
    if (mover && nsteps > 0) { p.load(kbeg, kend, t, pr0); q.load(kbeg, kend, t, qr0); }
    if (mover && nsteps > 1) { p.load(kbeg + BK, kend, t, pr1); q.load(kbeg + BK, kend, t, qr1); }
    if (mover && nsteps > 0) { p.store(Ps[0], t, pr0); q.store(Qs[0], t, qr0); }
    __syncthreads();
    // two k-steps per trip (the register sets alternate); a lone last step is peeled off below so the
    // loop has ONE exit -- with a break in the middle hipcc ping-ponged the accumulator between two
    // register sets (16 v_mov + 17 wait states per k-step)
    int s = 0;
    for (; s + 1 < nsteps; s += 2) {
        // even step: MFMA on buffer 0; register set 0 is free -> fetch tile s+2; stage tile s+1
        if (mover && s + 2 < nsteps) { p.load(kbeg + (s + 2) * BK, kend, t, pr0); q.load(kbeg + (s + 2) * BK, kend, t, qr0); }
        compute(0);
        if (mover) { p.store(Ps[1], t, pr1); q.store(Qs[1], t, qr1); }
        __syncthreads();
        // odd step
        if (mover && s + 3 < nsteps) { p.load(kbeg + (s + 3) * BK, kend, t, pr1); q.load(kbeg + (s + 3) * BK, kend, t, qr1); }
        compute(1);
        if (mover && s + 2 < nsteps) { p.store(Ps[0], t, pr0); q.store(Qs[0], t, qr0); }
        __syncthreads();
    }
    if (s < nsteps) {       // odd number of k-steps: the last tile sits in buffer 0
        compute(0);
        __syncthreads();
    }
    if (KW > 1) {
        // sum the k-groups' accumulators through LDS (the tile buffers are free after the last barrier)
        constexpr int PER_WAVE = WM * WN * 16 * 64;
        float *red = lds_raw;
        if (kg > 0) {
            float *dst = red + ((kg - 1) * 4 + (wave & 3)) * PER_WAVE + lane;
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
            const float *src = red + (g2 * 4 + wave) * PER_WAVE + lane;
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
        if (blockIdx.x == cls * tiles_j && t < BM && i0 + t < sink.I) {
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
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int z = 0; z < splits; ++z) {
        const float4 v = *reinterpret_cast<const float4 *>(src + (size_t)z * sink.stride);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    if (e.col(j)) e.put(i, j, s.x);
    if (e.col(j + 1)) e.put(i, j + 1, s.y);
    if (e.col(j + 2)) e.put(i, j + 2, s.z);
    if (e.col(j + 3)) e.put(i, j + 3, s.w);
}

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

// ------------------------------------------------------------------------------------------
// host-side dispatch
// ------------------------------------------------------------------------------------------
struct Plan { int wm, wn, splits, klen, kw, narrow; };

int g_force_wm = 0, g_force_wn = 0, g_force_splits = 0, g_force_kw = 0;   // tuning hook (mvae_debug_set_tiling)

// Tile and split choice, from the measurements in profiles/r01_gemm_tiles.txt (tools/gemm_bench.py):
// the gather-fed forward / dgrad forms and the Linear layers run best on 64x64 tiles (5 waves per
// SIMD hide the gather latency; 128-wide tiles drop to 2-3 waves), the conv weight-gradient form
// (two gathers feeding a small output over a huge reduction) wants the arithmetic intensity of
// 128-wide tiles.  Reductions are split until ~4 blocks per CU exist (>= 2 k-steps per split).
enum PlanKind { PLAN_FWD = 0, PLAN_CONV_WGRAD = 1, PLAN_LIN_WGRAD = 2 };

inline Plan make_plan(int I, int J, int K, bool allow_split, PlanKind kind = PLAN_FWD, int ncls = 1) {
    Plan p;
    p.wm = 1; p.wn = 1;
    p.narrow = (I <= 32 && J >= 128 && !g_force_wm && !g_force_wn && !g_force_kw) ? 1 : 0;
    if (kind == PLAN_CONV_WGRAD && !p.narrow) {
        if (J >= 128) p.wn = 2;
        if (I >= 128 && J >= 128) p.wm = 2;
    }
    if (g_force_wm) p.wm = g_force_wm;
    if (g_force_wn) p.wn = g_force_wn;
    const long tiles = p.narrow ? (long)((J + 127) / 128) * ncls
                                : (long)((I + 64 * p.wm - 1) / (64 * p.wm)) * ((J + 64 * p.wn - 1) / (64 * p.wn)) * ncls;
    // fewer than one wave per SIMD (256 CUs x 4): let KW wave groups share each 64x64 tile
    p.kw = 1;
    if (!p.narrow && p.wm == 1 && p.wn == 1 && K >= 4 * BK) {
        if (tiles * 4 <= 256) p.kw = 4;
        else if (tiles * 2 <= 256) p.kw = 2;
    }
    if (g_force_kw) p.kw = (p.wm == 1 && p.wn == 1) ? g_force_kw : 1;
    long want = 1;
    if (allow_split) {
        // blocks to aim for (profiles/r01_split_sweep.txt): the Linear forward / dgrad forms want every
        // CU busy and then as FEW splits as possible (each block keeps >= 8 k-steps, the finish reads
        // less); the 128-wide conv weight-gradient tiles run two blocks per CU; everything else (Linear
        // weight gradients, small conv outputs) is fastest with ~4 blocks per CU
        static const long target_env = getenv("MVAE_SPLIT_TARGET") ? atol(getenv("MVAE_SPLIT_TARGET")) : 0;   // tuning
        long target_blocks = 1024 / p.kw;
        if (kind == PLAN_FWD) target_blocks = (p.kw == 1) ? 512 : 256;
        if (kind == PLAN_CONV_WGRAD && p.wm * p.wn >= 2) target_blocks = 512;
        if (target_env > 0) target_blocks = target_env / p.kw;
        want = (target_blocks + tiles / 2) / tiles;         // nearest: 800 tiles against 1024 is one round, not two
        const long maxs = (K + 2 * BK - 1) / (2 * BK);
        if (want > maxs) want = maxs;
        if (want < 1) want = 1;
        if (kind == PLAN_LIN_WGRAD && !g_force_kw && !target_env) {
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
        if (g_force_splits > 0) want = g_force_splits;
        if (g_force_splits < 0 && kind == PLAN_FWD) want = 1;      // tuning: forward / dgrad forms never split
    }
    p.klen = (int)(((K + want - 1) / want + BK - 1) / BK * BK);
    p.splits = (K + p.klen - 1) / p.klen;
    return p;
}

template <template <int> class PL, template <int> class QL, class E, bool ROWSUM, class PF, class QF>
int launch_igemm(Plan pl, PF make_p, QF make_q, E e, int I, int J, int K, SplitSink sink, hipStream_t st) {
#define MVAE_LAUNCH(WM, WN, KW, NARROW)                                                          \
    {                                                                                            \
        constexpr int TM = NARROW ? 32 : 64 * WM, TN = NARROW ? 128 : 64 * WN;                   \
        PL<TM> p; make_p(p);                                                                     \
        QL<TN> q; make_q(q);                                                                     \
        dim3 grid(((J + TN - 1) / TN) * sink.ncls, (I + TM - 1) / TM, pl.splits);                \
        constexpr size_t tile_b = 2 * BK * (TM + LPAD + TN + LPAD) * sizeof(float);              \
        constexpr size_t red_b = (size_t)(KW - 1) * 4 * WM * WN * 16 * 64 * sizeof(float);       \
        constexpr size_t lds = tile_b > red_b ? tile_b : red_b;                                  \
        auto kern = igemm_kernel<PL<TM>, QL<TN>, E, WM, WN, ROWSUM, KW, NARROW>;                 \
        static bool attr_done = false;                                                           \
        if (!attr_done) {                                                                        \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern),                      \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);     \
            attr_done = true;                                                                    \
        }                                                                                        \
        hipLaunchKernelGGL(kern, grid, dim3(NTHREADS * KW), lds, st, p, q, e, K, pl.klen, sink); \
    }
    if (pl.narrow) MVAE_LAUNCH(1, 1, 1, true)
    else if (pl.wm == 2 && pl.wn == 2) MVAE_LAUNCH(2, 2, 1, false)
    else if (pl.wm == 2 && pl.wn == 1) MVAE_LAUNCH(2, 1, 1, false)
    else if (pl.wm == 1 && pl.wn == 2) MVAE_LAUNCH(1, 2, 1, false)
    else if (pl.kw == 4) MVAE_LAUNCH(1, 1, 4, false)
    else if (pl.kw == 2) MVAE_LAUNCH(1, 1, 2, false)
    else MVAE_LAUNCH(1, 1, 1, false)
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

inline SplitSink make_sink(void *ws, int I, int J, bool rowsum) {
    SplitSink s;
    s.ws = (float *)ws; s.I = I; s.J = J;
    s.stride = (size_t)I * J + (rowsum ? I : 0);
    s.rowsum = nullptr; s.rowsum_stride = 0; s.rowsum_accumulate = 0; s.ncls = 1; s.rowsum_cls_stride = 0;
    s.rowsum_final = nullptr; s.rowsum_final_accumulate = 0; s.cls_region = 0; s.rowsum_final_cls_stride = 0;
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

inline ConvGeom make_geom(int B, int Cin, int H, int W, int Cout, int stride, int pad) {
    ConvGeom g;
    g.B = B; g.Cin = Cin; g.H = H; g.W = W; g.Cout = Cout; g.stride = stride; g.pad = pad;
    g.OH = (H + 2 * pad - 4) / stride + 1;
    g.OW = (W + 2 * pad - 4) / stride + 1;
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

// ---- conv forward form: y[n][co][oh][ow] = sum_k w[co][k] * im2col(x)[k][(n,oh,ow)] ----
int conv_fwd_impl(const float *x, const float *w, float *pre, float *act, const float *dpre,
                  ConvGeom g, hipStream_t st) {
    const int I = g.Cout, J = g.B * g.OH * g.OW, K = g.Cin * 16;
    Plan pl = make_plan(I, J, K, false);
    EpNCHW e;
    e.out = pre; e.act = act; e.dpre = dpre;
    e.C = g.Cout; e.HW = g.OH * g.OW; e.Wfull = g.OW; e.H2 = g.OH; e.W2 = g.OW;
    e.sy = 1; e.py = 0; e.px = 0; e.J = J; e.off = 0;
    auto mp = [&](auto &p) { p.src = w; p.ld = K; p.R = I; p.Klen = K; };
    auto mq = [&](auto &q) { q.x = x; q.g = g; q.Mtot = J; };
    if (aligned16(w))
        return launch_igemm<LdRowsK, LdIm2col, EpNCHW, false>(pl, mp, mq, e, I, J, K, make_sink(nullptr, I, J, false), st);
    return launch_igemm<LdRowsKS, LdIm2col, EpNCHW, false>(pl, mp, mq, e, I, J, K, make_sink(nullptr, I, J, false), st);
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

inline bool conv_dgrad_small_ok(const ConvGeom &g) {
    return g.stride == 2 && g.pad == 1 && g.Cin <= 4 && g.H == 2 * g.OH && g.W == 2 * g.OW &&
           (size_t)g.Cout * g.Cin * 16 * sizeof(float) <= 48 * 1024;
}

inline int conv_dgrad_small(const float *dy, const float *w, float *dx, float *act, const float *dpre,
                            ConvGeom g, hipStream_t st) {
    const int total = g.B * g.OH * g.OW;
    const size_t lds = (size_t)g.Cout * g.Cin * 16 * sizeof(float);
    const dim3 grid((total + 255) / 256), blk(256);
    switch (g.Cin) {
        case 1: hipLaunchKernelGGL(convT_small_kernel<1>, grid, blk, lds, st, dy, w, dx, act, dpre, g, total); break;
        case 2: hipLaunchKernelGGL(convT_small_kernel<2>, grid, blk, lds, st, dy, w, dx, act, dpre, g, total); break;
        case 3: hipLaunchKernelGGL(convT_small_kernel<3>, grid, blk, lds, st, dy, w, dx, act, dpre, g, total); break;
        default: hipLaunchKernelGGL(convT_small_kernel<4>, grid, blk, lds, st, dy, w, dx, act, dpre, g, total); break;
    }
    return mvae_launch_status();
}

// ---- stride-1 transposed conv as a DENSE GEMM + in-register/LDS col2im (ConvTranspose2d(256,128,4,1,0)
//      5x5 -> 8x8 and the dgrad of Conv2d(128,256,4,1,0): celeba/model.py:85,117).  In the gather
//      form only 39 % of the (output pixel, tap) pairs are inside the 5x5 input, so 61 % of the MFMA
//      work multiplies zeros.  Here the GEMM is  col[(n,oh,ow)][(ci,kh,kw)] = sum_co dy[n,co,oh,ow] *
//      w[co,ci,kh,kw]  (every product is real; the 25 positions of an image are padded to one 32-row
//      MFMA tile = 78 % utilisation), and the scatter-add  dx[n,ci,oh+kh,ow+kw] += col  happens inside
//      the wave that owns the tile: a wave holds one image x 2 input channels x 16 taps, i.e.
//      everything two output planes need.  Block = 2 images x 4 channels; k loop over Cout. ----
__global__ __launch_bounds__(256, 2) void convT_s1_kernel(const float *dy, const float *w, float *out, float *act,
                                                          const float *dpre, ConvGeom g) {
    constexpr int BMX = 64, BNX = 64;
    __shared__ __attribute__((aligned(16))) float Ps[2][BK][BMX + LPAD];
    __shared__ __attribute__((aligned(16))) float Qs[2][BK][BNX + LPAD];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wi = wave >> 1, wj = wave & 1;
    const int P = g.OH * g.OW;                      // positions per image (<= 32)
    const int n0 = blockIdx.y * 2, ci0 = blockIdx.x * 4;
    const int K = g.Cout, J = g.Cin * 16;
    // P loader: lanes along the position axis, 4 k rows per pass
    const int pi = t & 63, pkq = t >> 6;
    const int pimg = pi >> 5, ppos = pi & 31;
    const bool pok = ppos < P && n0 + pimg < g.B;
    const float *psrc = dy + ((size_t)(pok ? n0 + pimg : 0) * K) * P + (pok ? ppos : 0);
    // Q loader: weight rows are contiguous in (ci, tap): 16 float4 per k row, 2 per thread
    const float *qsrc = w + (size_t)ci0 * 16;
    float pr[8], pm[8];
    float4 qr[2];
    float qm[2];
    auto load = [&](int k0) {
#pragma unroll
        for (int v = 0; v < 8; ++v) {
            const int k = k0 + pkq + 4 * v;
            pm[v] = (pok && k < K) ? 1.f : 0.f;
            pr[v] = psrc[(size_t)min(k, K - 1) * P];
        }
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const int f = t + 256 * v, k = k0 + (f >> 4), c4 = (f & 15) * 4;
            qm[v] = (k < K) ? 1.f : 0.f;
            qr[v] = *reinterpret_cast<const float4 *>(qsrc + (size_t)min(k, K - 1) * J + c4);
        }
    };
    auto store = [&](int buf) {
#pragma unroll
        for (int v = 0; v < 8; ++v) Ps[buf][pkq + 4 * v][pi] = pr[v] * pm[v];
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const int f = t + 256 * v;
            *reinterpret_cast<float4 *>(&Qs[buf][f >> 4][(f & 15) * 4]) =
                make_float4(qr[v].x * qm[v], qr[v].y * qm[v], qr[v].z * qm[v], qr[v].w * qm[v]);
        }
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int lrow = lane >> 5, lcol = lane & 31;
    const int nsteps = (K + BK - 1) / BK;
    load(0);
    store(0);
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
        const int buf = s & 1;
        if (s + 1 < nsteps) load((s + 1) * BK);
        float a0 = Ps[buf][lrow][wi * 32 + lcol], b0 = Qs[buf][lrow][wj * 32 + lcol];
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            float a1 = 0.f, b1 = 0.f;
            if (kk + 1 < BK / 2) {
                a1 = Ps[buf][(kk + 1) * 2 + lrow][wi * 32 + lcol];
                b1 = Qs[buf][(kk + 1) * 2 + lrow][wj * 32 + lcol];
            }
            __builtin_amdgcn_sched_barrier(0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            a0 = a1; b0 = b1;
        }
        if (s + 1 < nsteps) store(buf ^ 1);
        __syncthreads();
    }
    // col2im inside the wave: park the 32 (positions) x 32 (2 channels x 16 taps) tile in LDS ...
    float *sc = &Ps[0][0][0] + wave * (32 * 33);        // 4 x 4224 B <= the P tile buffers
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lrow;
        sc[row * 33 + lcol] = acc[r];
    }
    __syncthreads();
    // ... and let each lane gather the <= 16 taps of its output pixels
    const int n = n0 + wi;
    if (n >= g.B) return;
    const int HW = g.H * g.W;
    for (int cl = 0; cl < 2; ++cl) {
        const int ci = ci0 + wj * 2 + cl;
        if (ci >= g.Cin) break;
        for (int px = lane; px < HW; px += 64) {
            const int ih = px / g.W, iw = px - ih * g.W;
            float v = 0.f;
#pragma unroll
            for (int kh = 0; kh < 4; ++kh) {
                const int oh = ih - kh;
                if (oh < 0 || oh >= g.OH) continue;
#pragma unroll
                for (int kw = 0; kw < 4; ++kw) {
                    const int ow = iw - kw;
                    if (ow < 0 || ow >= g.OW) continue;
                    v += sc[(oh * g.OW + ow) * 33 + cl * 16 + kh * 4 + kw];
                }
            }
            const size_t o = ((size_t)n * g.Cin + ci) * HW + px;
            if (dpre) v *= swish_grad_(dpre[o]);
            if (out) out[o] = v;
            if (act) act[o] = swishf_(v);
        }
    }
}

inline bool conv_dgrad_s1_ok(const ConvGeom &g, const float *w) {
    return g.stride == 1 && g.pad == 0 && g.OH * g.OW <= 32 && g.Cin % 4 == 0 && aligned16(w);
}

inline int conv_dgrad_s1(const float *dy, const float *w, float *dx, float *act, const float *dpre, ConvGeom g,
                         hipStream_t st) {
    dim3 grid(g.Cin / 4, (g.B + 1) / 2);
    hipLaunchKernelGGL(convT_s1_kernel, grid, dim3(256), 0, st, dy, w, dx, act, dpre, g);
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
    if (conv_dgrad_small_ok(g) && !g_force_wm) return conv_dgrad_small(dy, w, dx, act, dpre, g, st);
    if (conv_dgrad_s1_ok(g, w) && !g_force_wm) return conv_dgrad_s1(dy, w, dx, act, dpre, g, st);
    if (!ws || ws_bytes < dgrad_ws_floats(g) * sizeof(float)) return MVAE_ERR_WS;
    float *wr = (float *)ws;
    {
        const int total = s * s * K * g.Cin;
        int blocks = (total + 255) / 256;
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(repack_dgrad_weights_kernel, dim3(blocks), dim3(256), 0, st, w, wr, g.Cout, g.Cin, s, g.pad);
    }
    Plan pl = make_plan(I, J, K, false, PLAN_FWD, s * s);
    const bool vec = (g.Cin % 4 == 0) && aligned16(wr);
    EpNCHW e;
    e.out = dx; e.act = act; e.dpre = dpre;
    e.C = g.Cin; e.HW = g.H * g.W; e.Wfull = g.W; e.H2 = H2; e.W2 = W2;
    e.sy = s; e.py = 0; e.px = 0; e.J = J; e.off = 0;
    auto mp = [&](auto &p) {
        p.src = wr; p.ld = g.Cin; p.R = g.Cin; p.Klen = K; p.cls_stride = (size_t)K * g.Cin;
    };
    auto mq = [&](auto &q) { q.dy = dy; q.g = g; q.Mtot = J; q.H2 = H2; q.W2 = W2; };
    SplitSink sink = make_sink(nullptr, I, J, false);
    sink.ncls = s * s;      // all parity classes in ONE launch: s*s times the blocks
    if (vec) {
        if (s == 2) return launch_igemm<LdRowsMN, LdDgradDyS2, EpNCHW, false>(pl, mp, mq, e, I, J, K, sink, st);
        return launch_igemm<LdRowsMN, LdDgradDyS1, EpNCHW, false>(pl, mp, mq, e, I, J, K, sink, st);
    }
    if (s == 2) return launch_igemm<LdRowsMNS, LdDgradDyS2, EpNCHW, false>(pl, mp, mq, e, I, J, K, sink, st);
    return launch_igemm<LdRowsMNS, LdDgradDyS1, EpNCHW, false>(pl, mp, mq, e, I, J, K, sink, st);
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

inline bool wgrad_smallcin_ok(const ConvGeom &g) {
    return g.stride == 2 && g.pad == 1 && g.Cin <= 4 && (g.Cout == 32 || g.Cout == 64) && g.OW <= 32 &&
           (g.OW & 1) == 0 && g.W <= SC_MAXW && (g.W & 3) == 0;
}
inline int wgrad_smallcin_blocks(const ConvGeom &g) {
    const int units = g.B * g.OH;
    int blocks = (units + SC_WAVES - 1) / SC_WAVES;
    return blocks > 256 ? 256 : blocks;      // one 8-wave block per CU; more partials only slow the finish
}

// ---- conv wgrad form: dw[co][(ci,kh,kw)] = sum_(n,oh,ow) dy[n][co][oh][ow] * x[n][ci][ih][iw] ----
int conv_wgrad_impl(const float *dy, const float *x, float *dw, ConvGeom g, int flags, void *ws,
                    size_t ws_bytes, hipStream_t st) {
    const int I = g.Cout, J = g.Cin * 16, K = g.B * g.OH * g.OW;
    EpRowMajor e;
    e.out = dw; e.act = nullptr; e.ld = J; e.bias = nullptr; e.dpre = nullptr; e.ldp = 0;
    e.mask = nullptr; e.ldm = 0; e.mask_scale = 1.f; e.I = I; e.J = J;
    e.accumulate = (flags & MVAE_ACCUMULATE) ? 1 : 0;
    if (wgrad_smallcin_ok(g) && !g_force_wm && !g_force_splits && ws) {
        const int blocks = wgrad_smallcin_blocks(g);
        if (ws_bytes >= (size_t)blocks * I * J * sizeof(float)) {
            const int mt = I / 32, nt = (J + 31) / 32;
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
            SplitSink fs = make_sink(ws, I, J, false);
            if (blocks > 16) {
                hipLaunchKernelGGL((finish_kernel<EpRowMajor>), dim3((J + 31) / 32, I), dim3(256), 0, st, fs, blocks, e);
            } else {
                hipLaunchKernelGGL((finish_few_kernel<EpRowMajor>), dim3((J + 255) / 256, I), dim3(256), 0, st, fs,
                                   blocks, e);
            }
            return mvae_launch_status();
        }
    }
    Plan pl = make_plan(I, J, K, true, PLAN_CONV_WGRAD);
    SplitSink sink = make_sink(ws, I, J, false);
    if (pl.splits > 1 && (!ws || ws_bytes < pl.splits * sink.stride * sizeof(float))) return MVAE_ERR_WS;
    auto mp = [&](auto &p) { p.dy = dy; p.g = g; };
    auto mq = [&](auto &q) { q.x = x; q.g = g; q.J = J; };
    return launch_igemm<LdWgradDy, LdWgradX, EpRowMajor, false>(pl, mp, mq, e, I, J, K, sink, st);
}

}  // namespace

// ==========================================================================================
// C ABI
// ==========================================================================================
MVAE_EXPORT int mvae_abi_version(void) { return 2; }

MVAE_EXPORT void mvae_debug_set_tiling(int wm, int wn, int splits) {
    g_force_wm = wm; g_force_wn = wn; g_force_splits = splits;
}

MVAE_EXPORT void mvae_debug_set_kwaves(int kw) { g_force_kw = kw; }

MVAE_EXPORT size_t mvae_gemm_ws_bytes(int rows_out, int cols_out, int reduce_len) {
    if (rows_out <= 0 || cols_out <= 0 || reduce_len <= 0) return 0;
    size_t n = split_ws_floats(rows_out, cols_out, reduce_len);
    const size_t repack = (size_t)rows_out * cols_out;      // dgrad-form weight repack: Cin x (Cout*16)
    if (g_force_splits > 0) n = (size_t)g_force_splits * ((size_t)rows_out * cols_out + rows_out);
    const size_t smallcin = (size_t)512 * rows_out * cols_out;      // per-block partials of wgrad_smallcin_kernel
    if (rows_out <= 64 && cols_out <= 64 && smallcin > n) n = smallcin;
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
    Plan pl = make_plan(M, N, K, ws != nullptr, PLAN_FWD, gr.G);
    SplitSink sink = make_sink(ws, M, N, false);
    sink.ncls = gr.G; sink.cls_region = (size_t)pl.splits * sink.stride;
    if (pl.splits > 1 && ws_bytes < gr.G * sink.cls_region * sizeof(float)) return MVAE_ERR_WS;
    EpRowMajor e;
    e.out = pre; e.act = act; e.ld = ldy; e.bias = bias; e.dpre = nullptr; e.ldp = 0;
    e.mask = mask; e.ldm = N; e.mask_scale = mask_scale; e.I = M; e.J = N; e.accumulate = 0;
    e.out_cs = gr.d; e.bias_cs = gr.c;
    auto mp = [&](auto &p) { p.src = x; p.ld = ldx; p.R = M; p.Klen = K; p.cls_stride = gr.a; };
    auto mq = [&](auto &q) { q.src = w; q.ld = K; q.R = N; q.Klen = K; q.cls_stride = gr.b; };
    if (aligned16(x) && aligned16(w) && ldx % 4 == 0 && K % 4 == 0 && gr.a % 4 == 0 && gr.b % 4 == 0)
        return launch_igemm<LdRowsK, LdRowsK, EpRowMajor, false>(pl, mp, mq, e, M, N, K, sink, st);
    return launch_igemm<LdRowsKS, LdRowsKS, EpRowMajor, false>(pl, mp, mq, e, M, N, K, sink, st);
}

static int linear_dgrad_impl(const float *dy, int lddy, const float *w, float *dx, int lddx, const float *pre_in,
                             const float *mask, float mask_scale, int M, int N, int K, int flags, void *ws,
                             size_t ws_bytes, LinGroups gr, hipStream_t st) {
    // D[i = m][j = k] = sum_n dy[m][n] * w[n][k];  gr: a = dy stride, b = w stride, c = pre_in stride, d = dx stride
    Plan pl = make_plan(M, K, N, ws != nullptr, PLAN_FWD, gr.G);
    SplitSink sink = make_sink(ws, M, K, false);
    sink.ncls = gr.G; sink.cls_region = (size_t)pl.splits * sink.stride;
    if (pl.splits > 1 && ws_bytes < gr.G * sink.cls_region * sizeof(float)) return MVAE_ERR_WS;
    EpRowMajor e;
    e.out = dx; e.act = nullptr; e.ld = lddx; e.bias = nullptr; e.dpre = pre_in; e.ldp = K;
    e.mask = mask; e.ldm = K; e.mask_scale = mask_scale; e.I = M; e.J = K;
    e.accumulate = (flags & MVAE_ACCUMULATE) ? 1 : 0;
    e.out_cs = gr.d; e.dpre_cs = gr.c;
    auto mp = [&](auto &p) { p.src = dy; p.ld = lddy; p.R = M; p.Klen = N; p.cls_stride = gr.a; };
    auto mq = [&](auto &q) { q.src = w; q.ld = K; q.R = K; q.Klen = N; q.cls_stride = gr.b; };
    if (aligned16(dy) && aligned16(w) && lddy % 4 == 0 && N % 4 == 0 && K % 4 == 0 && gr.a % 4 == 0 && gr.b % 4 == 0)
        return launch_igemm<LdRowsK, LdRowsMN, EpRowMajor, false>(pl, mp, mq, e, M, K, N, sink, st);
    return launch_igemm<LdRowsKS, LdRowsMNS, EpRowMajor, false>(pl, mp, mq, e, M, K, N, sink, st);
}

static int linear_wgrad_impl(const float *dy, int lddy, const float *x, int ldx, float *dw, float *db, int M, int N,
                             int K, int flags, void *ws, size_t ws_bytes, LinGroups gr, hipStream_t st) {
    // D[i = n][j = k] = sum_m dy[m][n] * x[m][k];  gr: a = dy stride, b = x stride, c = db stride, d = dw stride
    Plan pl = make_plan(N, K, M, ws != nullptr, PLAN_LIN_WGRAD, gr.G);
    SplitSink sink = make_sink(ws, N, K, db != nullptr);
    sink.ncls = gr.G; sink.cls_region = (size_t)pl.splits * sink.stride;
    if (pl.splits > 1 && ws_bytes < gr.G * sink.cls_region * sizeof(float)) return MVAE_ERR_WS;
    const int acc = (flags & MVAE_ACCUMULATE) ? 1 : 0;
    EpRowMajor e;
    e.out = dw; e.act = nullptr; e.ld = K; e.bias = nullptr; e.dpre = nullptr; e.ldp = 0;
    e.mask = nullptr; e.ldm = 0; e.mask_scale = 1.f; e.I = N; e.J = K; e.accumulate = acc;
    e.out_cs = gr.d;
    auto mp = [&](auto &p) { p.src = dy; p.ld = lddy; p.R = N; p.Klen = M; p.cls_stride = gr.a; };
    auto mq = [&](auto &q) { q.src = x; q.ld = ldx; q.R = K; q.Klen = M; q.cls_stride = gr.b; };
    const bool vec = aligned16(dy) && aligned16(x) && lddy % 4 == 0 && ldx % 4 == 0 && N % 4 == 0 && K % 4 == 0 &&
                     gr.a % 4 == 0 && gr.b % 4 == 0;
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
        rc = vec ? launch_igemm<LdRowsMN, LdRowsMN, EpRowMajor, true>(pl, mp, mq, e, N, K, M, sink, st)
                 : launch_igemm<LdRowsMNS, LdRowsMNS, EpRowMajor, true>(pl, mp, mq, e, N, K, M, sink, st);
        if (rc) return rc;
        return MVAE_OK;
    }
    return vec ? launch_igemm<LdRowsMN, LdRowsMN, EpRowMajor, false>(pl, mp, mq, e, N, K, M, sink, st)
               : launch_igemm<LdRowsMNS, LdRowsMNS, EpRowMajor, false>(pl, mp, mq, e, N, K, M, sink, st);
}

static const LinGroups kOneGroup = {1, 0, 0, 0, 0};

MVAE_EXPORT int mvae_linear_fwd(const float *x, int ldx, const float *w, const float *bias,
                                float *pre, float *act, int ldy, const float *mask, float mask_scale,
                                int M, int N, int K, void *ws, size_t ws_bytes, mvae_stream_t stream) {
    if (!x || !w || (!pre && !act) || M <= 0 || N <= 0 || K <= 0 || ldx < K || ldy < N) return MVAE_ERR_ARG;
    return linear_fwd_impl(x, ldx, w, bias, pre, act, ldy, mask, mask_scale, M, N, K, ws, ws_bytes, kOneGroup,
                           (hipStream_t)stream);
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

MVAE_EXPORT int mvae_conv2d_k4_fwd(const float *x, const float *w, float *pre, float *act, int B, int Cin,
                                   int H, int W, int Cout, int stride, int pad, mvae_stream_t stream) {
    if (!x || !w || (!pre && !act) || !conv_args_ok(B, Cin, H, W, Cout, stride, pad)) return MVAE_ERR_ARG;
    return conv_fwd_impl(x, w, pre, act, nullptr, make_geom(B, Cin, H, W, Cout, stride, pad), (hipStream_t)stream);
}

MVAE_EXPORT int mvae_conv2d_k4_dgrad(const float *dy, const float *w, float *dx, const float *pre_in, int B,
                                     int Cin, int H, int W, int Cout, int stride, int pad, void *ws,
                                     size_t ws_bytes, mvae_stream_t stream) {
    if (!dy || !w || !dx || !conv_args_ok(B, Cin, H, W, Cout, stride, pad)) return MVAE_ERR_ARG;
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

MVAE_EXPORT int mvae_convT2d_k4_fwd(const float *x, const float *w, float *pre, float *act, int B, int Cin,
                                    int H, int W, int Cout, int stride, int pad, void *ws, size_t ws_bytes,
                                    mvae_stream_t stream) {
    ConvGeom g;
    if (!x || !w || (!pre && !act) || B <= 0 || !convT_geom(B, Cin, H, W, Cout, stride, pad, &g)) return MVAE_ERR_ARG;
    return conv_dgrad_impl(x, w, pre, act, nullptr, g, ws, ws_bytes, (hipStream_t)stream);
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
