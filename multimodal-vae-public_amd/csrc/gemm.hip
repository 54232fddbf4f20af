// gemm.hip -- the dense contractions of the MVAE train step on fp32 MFMA (gfx950).
//
// One LDS-tiled kernel template, `igemm_kernel`, computes D[i][j] = sum_k P(i,k) * Q(k,j)
// with v_mfma_f32_32x32x2_f32 (exact fp32, 157 TFLOP/s peak on MI355X).  What differs between
// Linear fwd/dgrad/wgrad, Conv2d 4x4 fwd/dgrad/wgrad and ConvTranspose2d is only
//   * how the P and Q tiles are fetched from HBM (loader functors: row-major vector loads,
//     implicit-im2col gathers, parity-decomposed transposed-conv gathers), and
//   * what the epilogue does with the accumulator tile (bias / swish / dropout mask / swish'
//     of the producer's pre-activation / accumulate / NCHW scatter / split-K partial).
// The j axis is the lane axis of the MFMA result (32 consecutive j per store instruction),
// so each op maps its memory-contiguous output axis to j.
//
// Tiling: 256 threads = 4 waves (2 x 2), each wave WM x WN MFMA tiles of 32x32
// (block tile 64*WM x 64*WN), BK = 16.  Global -> registers -> LDS with the next tile's
// loads in flight during the MFMAs of the current one; several blocks per CU hide the
// barriers.  LDS tiles are [BK][tile + 4]: fragment reads are bank-conflict free
// (ds_read_b32, lanes 0..31 consecutive), float4 tile rows stay 16-byte aligned.
//
// Reductions over the batch (wgrad) are split across blockIdx.z into a caller-provided
// workspace and summed by `splitk_reduce_kernel` in a fixed order: deterministic, no atomics.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int BK = 16;
constexpr int LPAD = 4;
constexpr int NTHREADS = 256;

// ------------------------------------------------------------------------------------------
// loaders.  Each exposes  init(tile0, t) / load(k0, kend, t) / store(lds, t)  and the tile
// extent TILE along its non-reduced axis.  The LDS image is always [BK][TILE + LPAD].
// ------------------------------------------------------------------------------------------

// S[r * ld + k]: reduction axis contiguous (x and w of Linear fwd, dy of dgrad, conv weights).
template <int TILE_>
struct LdRowsK {
    static constexpr int TILE = TILE_;
    static constexpr int NV = TILE / 64;
    const float *src; int ld; int R; int vec;
    int r0; float4 reg[NV];
    __device__ void init(int tile0, int) { r0 = tile0; }
    __device__ void load(int k0, int kend, int t) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int f = t + NTHREADS * v;
            const int r = r0 + (f >> 2), k = k0 + (f & 3) * 4;
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < R) {
                const float *p = src + (size_t)r * ld + k;
                if (vec && k + 3 < kend) {
                    x = *reinterpret_cast<const float4 *>(p);
                } else {
                    if (k < kend) x.x = p[0];
                    if (k + 1 < kend) x.y = p[1];
                    if (k + 2 < kend) x.z = p[2];
                    if (k + 3 < kend) x.w = p[3];
                }
            }
            reg[v] = x;
        }
    }
    __device__ void store(float (*L)[TILE + LPAD], int t) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int f = t + NTHREADS * v;
            const int r = f >> 2, kc = (f & 3) * 4;
            L[kc + 0][r] = reg[v].x; L[kc + 1][r] = reg[v].y;
            L[kc + 2][r] = reg[v].z; L[kc + 3][r] = reg[v].w;
        }
    }
};

// S[k * ld + r]: non-reduced axis contiguous (w of dgrad, dy and x of wgrad).
template <int TILE_>
struct LdRowsMN {
    static constexpr int TILE = TILE_;
    static constexpr int NV = TILE / 64;
    static constexpr int V4 = TILE / 4;     // float4 per k row
    const float *src; int ld; int R; int vec;
    int r0; float4 reg[NV];
    __device__ void init(int tile0, int) { r0 = tile0; }
    __device__ void load(int k0, int kend, int t) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int f = t + NTHREADS * v;
            const int k = k0 + f / V4, r = r0 + (f % V4) * 4;
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k < kend) {
                const float *p = src + (size_t)k * ld + r;
                if (vec && r + 3 < R) {
                    x = *reinterpret_cast<const float4 *>(p);
                } else {
                    if (r < R) x.x = p[0];
                    if (r + 1 < R) x.y = p[1];
                    if (r + 2 < R) x.z = p[2];
                    if (r + 3 < R) x.w = p[3];
                }
            }
            reg[v] = x;
        }
    }
    __device__ void store(float (*L)[TILE + LPAD], int t) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int f = t + NTHREADS * v;
            *reinterpret_cast<float4 *>(&L[f / V4][(f % V4) * 4]) = reg[v];
        }
    }
};

// Geometry of a 4x4 convolution y[B,Cout,OH,OW] = conv(x[B,Cin,H,W], w[Cout,Cin,4,4]).
struct ConvGeom {
    int B, Cin, H, W, Cout, OH, OW, stride, pad;
};

// im2col of x for the forward conv: element (k = (ci,kh,kw), m = (b,oh,ow)); lanes along m.
template <int TILE_>
struct LdIm2col {
    static constexpr int TILE = TILE_;
    static constexpr int NV = TILE / 16;              // elements per thread
    static constexpr int KSTEP = NTHREADS / TILE;     // k rows covered per pass
    const float *x; ConvGeom g; int Mtot;
    int base; unsigned vh, vw; float reg[NV];
    __device__ void init(int tile0, int t) {
        const int m = tile0 + (t % TILE);
        vh = vw = 0; base = 0;
        if (m < Mtot) {
            const int ohw = g.OH * g.OW;
            const int b = m / ohw, rem = m - b * ohw;
            const int oh = rem / g.OW, ow = rem - oh * g.OW;
            const int ih0 = oh * g.stride - g.pad, iw0 = ow * g.stride - g.pad;
            base = (b * g.Cin * g.H + ih0) * g.W + iw0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (ih0 + q >= 0 && ih0 + q < g.H) vh |= 1u << q;
                if (iw0 + q >= 0 && iw0 + q < g.W) vw |= 1u << q;
            }
        }
    }
    __device__ void load(int k0, int kend, int t) {
        const int kb = k0 + t / TILE;
        const int hw = g.H * g.W;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int k = kb + v * KSTEP;
            const int ci = k >> 4, kh = (k >> 2) & 3, kw = k & 3;
            const bool ok = k < kend && ((vh >> kh) & 1u) && ((vw >> kw) & 1u);
            reg[v] = ok ? x[base + ci * hw + kh * g.W + kw] : 0.f;
        }
    }
    __device__ void store(float (*L)[TILE + LPAD], int t) {
        const int m = t % TILE, kb = t / TILE;
#pragma unroll
        for (int v = 0; v < NV; ++v) L[kb + v * KSTEP][m] = reg[v];
    }
};

// Transposed-conv (dgrad) gather of dy for ONE output parity class (ph,pw) of dx:
// element (k = (co,a,b), m = (n, ih', iw')) with ih = ih'*s + ph, kh = kh0 + s*a,
// oh = (ih + pad - kh0)/s - a.  TPD = 4/s taps per dim; only the taps that can reach the
// class are enumerated, so stride 2 does no multiply-by-zero work.
template <int TILE_>
struct LdDgradDy {
    static constexpr int TILE = TILE_;
    static constexpr int NV = TILE / 16;
    static constexpr int KSTEP = NTHREADS / TILE;
    const float *dy; ConvGeom g; int Mtot; int H2, W2, ph, pw, kh0, kw0, tlog; // tlog = log2(TPD)
    int base; unsigned vh, vw; float reg[NV];
    __device__ void init(int tile0, int t) {
        const int m = tile0 + (t % TILE);
        vh = vw = 0; base = 0;
        if (m < Mtot) {
            const int hw2 = H2 * W2;
            const int n = m / hw2, rem = m - n * hw2;
            const int ih2 = rem / W2, iw2 = rem - ih2 * W2;
            const int ohb = (ih2 * g.stride + ph + g.pad - kh0) / g.stride;
            const int owb = (iw2 * g.stride + pw + g.pad - kw0) / g.stride;
            base = (n * g.Cout * g.OH + ohb) * g.OW + owb;
            const int tpd = 1 << tlog;
            for (int a = 0; a < tpd; ++a) {
                if (ohb - a >= 0 && ohb - a < g.OH) vh |= 1u << a;
                if (owb - a >= 0 && owb - a < g.OW) vw |= 1u << a;
            }
        }
    }
    __device__ void load(int k0, int kend, int t) {
        const int kb = k0 + t / TILE;
        const int ohw = g.OH * g.OW;
        const int tmask = (1 << tlog) - 1;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int k = kb + v * KSTEP;
            const int co = k >> (2 * tlog), a = (k >> tlog) & tmask, b = k & tmask;
            const bool ok = k < kend && ((vh >> a) & 1u) && ((vw >> b) & 1u);
            reg[v] = ok ? dy[base + co * ohw - a * g.OW - b] : 0.f;
        }
    }
    __device__ void store(float (*L)[TILE + LPAD], int t) {
        const int m = t % TILE, kb = t / TILE;
#pragma unroll
        for (int v = 0; v < NV; ++v) L[kb + v * KSTEP][m] = reg[v];
    }
};

// Weights for the same parity class: element (i = ci, k = (co,a,b)) = w[co][ci][kh0+s*a][kw0+s*b].
template <int TILE_>
struct LdDgradW {
    static constexpr int TILE = TILE_;
    static constexpr int NV = TILE / 16;
    static constexpr int KSTEP = NTHREADS / TILE;
    const float *w; int Cin, stride, kh0, kw0, tlog;
    int ci; float reg[NV];
    __device__ void init(int tile0, int t) { ci = tile0 + (t % TILE); }
    __device__ void load(int k0, int kend, int t) {
        const int kb = k0 + t / TILE;
        const int tmask = (1 << tlog) - 1;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int k = kb + v * KSTEP;
            const int co = k >> (2 * tlog), a = (k >> tlog) & tmask, b = k & tmask;
            const bool ok = k < kend && ci < Cin;
            reg[v] = ok ? w[((co * Cin + ci) * 4 + kh0 + stride * a) * 4 + kw0 + stride * b] : 0.f;
        }
    }
    __device__ void store(float (*L)[TILE + LPAD], int t) {
        const int m = t % TILE, kb = t / TILE;
#pragma unroll
        for (int v = 0; v < NV; ++v) L[kb + v * KSTEP][m] = reg[v];
    }
};

// wgrad operands: the reduction runs over k = (b,oh,ow); lanes along k (spatially contiguous).
// P: element (i = co, k) = dy[b][co][oh][ow].
template <int TILE_>
struct LdWgradDy {
    static constexpr int TILE = TILE_;
    static constexpr int NV = TILE / 16;
    const float *dy; ConvGeom g; int i0;
    float reg[NV];
    __device__ void init(int tile0, int) { i0 = tile0; }
    __device__ void load(int k0, int kend, int t) {
        const int k = k0 + (t & 15);
        const int ohw = g.OH * g.OW;
        const int b = k / ohw, sp = k - b * ohw;
        const int ib = i0 + (t >> 4);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int i = ib + v * 16;
            reg[v] = (k < kend && i < g.Cout) ? dy[(size_t)(b * g.Cout + i) * ohw + sp] : 0.f;
        }
    }
    __device__ void store(float (*L)[TILE + LPAD], int t) {
        const int kl = t & 15, ib = t >> 4;
#pragma unroll
        for (int v = 0; v < NV; ++v) L[kl][ib + v * 16] = reg[v];
    }
};

// Q: element (k, j = (ci,kh,kw)) = x[b][ci][oh*s-p+kh][ow*s-p+kw].
template <int TILE_>
struct LdWgradX {
    static constexpr int TILE = TILE_;
    static constexpr int NV = TILE / 16;
    const float *x; ConvGeom g; int J; int j0;
    float reg[NV];
    __device__ void init(int tile0, int) { j0 = tile0; }
    __device__ void load(int k0, int kend, int t) {
        const int k = k0 + (t & 15);
        const int ohw = g.OH * g.OW;
        const int b = k / ohw, sp = k - b * ohw;
        const int oh = sp / g.OW, ow = sp - oh * g.OW;
        const int ih0 = oh * g.stride - g.pad, iw0 = ow * g.stride - g.pad;
        const int hw = g.H * g.W;
        const int base = b * g.Cin * hw + ih0 * g.W + iw0;
        const int jb = j0 + (t >> 4);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int j = jb + v * 16;
            const int ci = j >> 4, kh = (j >> 2) & 3, kw = j & 3;
            const int ih = ih0 + kh, iw = iw0 + kw;
            const bool ok = k < kend && j < J && ih >= 0 && ih < g.H && iw >= 0 && iw < g.W;
            reg[v] = ok ? x[base + ci * hw + kh * g.W + kw] : 0.f;
        }
    }
    __device__ void store(float (*L)[TILE + LPAD], int t) {
        const int kl = t & 15, jb = t >> 4;
#pragma unroll
        for (int v = 0; v < NV; ++v) L[kl][jb + v * 16] = reg[v];
    }
};

// ------------------------------------------------------------------------------------------
// epilogues:  col(j) prepares the lane's column, put(i, j, v, split) consumes one element.
// ------------------------------------------------------------------------------------------

// Row-major destination D[i * ld + j] with the Linear fusions.
struct EpRowMajor {
    float *out; float *act; int ld;           // out = raw / pre-activation result, act = swish(result)
    const float *bias;                        // per column j (Linear fwd)
    const float *dpre; int ldp;               // multiply by swish'(dpre[i][j])
    const float *mask; int ldm; float mask_scale;   // dropout keep-mask (fwd on act, bwd on the product)
    int I, J; int accumulate;
    __device__ bool col(int j) const { return j < J; }
    __device__ void put(int i, int j, float v, int) const {
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
// address = (n * C + i) * HW + (row' * sy + py) * Wfull + col' * sx + px.
struct EpNCHW {
    float *out; float *act; const float *dpre;
    int C, HW, Wfull, H2, W2, sy, py, px, J;
    int off;   // per-lane column offset, set by col()
    __device__ bool col(int j) {
        if (j >= J) return false;
        const int hw2 = H2 * W2;
        const int n = j / hw2, rem = j - n * hw2;
        const int r = rem / W2, c = rem - r * W2;
        off = n * C * HW + (r * sy + py) * Wfull + c * sy + px;
        return true;
    }
    __device__ void put(int i, int, float v, int) const {
        if (i >= C) return;
        const int idx = off + i * HW;
        if (dpre) v *= swish_grad_(dpre[idx]);
        if (out) out[idx] = v;
        if (act) act[idx] = swishf_(v);
    }
};

// Split-K partial: ws[(split * I + i) * J + j]; or, with one split, the final row-major result.
struct EpPartial {
    float *ws; int I, J; size_t split_stride;
    float *direct; int accumulate;            // used when gridDim.z == 1
    __device__ bool col(int j) const { return j < J; }
    __device__ void put(int i, int j, float v, int split) const {
        if (i >= I) return;
        const size_t idx = (size_t)i * J + j;
        if (direct) {
            if (accumulate) v += direct[idx];
            direct[idx] = v;
        } else {
            ws[split * split_stride + idx] = v;
        }
    }
};

// ------------------------------------------------------------------------------------------
// the kernel
// ------------------------------------------------------------------------------------------
template <class P, class Q, class E, int WM, int WN, bool ROWSUM>
__global__ __launch_bounds__(NTHREADS) void igemm_kernel(P p, Q q, E e, int K, int klen,
                                                         float *rowsum_out, size_t rowsum_stride,
                                                         int rowsum_rows, int rowsum_accumulate) {
    constexpr int BM = 64 * WM, BN = 64 * WN;
    static_assert(P::TILE == BM && Q::TILE == BN, "loader tile mismatch");
    __shared__ __attribute__((aligned(16))) float Ps[BK][BM + LPAD];
    __shared__ __attribute__((aligned(16))) float Qs[BK][BN + LPAD];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wi = wave >> 1, wj = wave & 1;
    const int i0 = blockIdx.y * BM, j0 = blockIdx.x * BN, split = blockIdx.z;
    const int kbeg = split * klen;
    const int kend = min(K, kbeg + klen);

    f32x16 acc[WM][WN];
#pragma unroll
    for (int a = 0; a < WM; ++a)
#pragma unroll
        for (int b = 0; b < WN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    p.init(i0, t);
    q.init(j0, t);
    p.load(kbeg, kend, t);
    q.load(kbeg, kend, t);
    float rsum = 0.f;

    const int lrow = lane >> 5, lcol = lane & 31;
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        __syncthreads();
        p.store(Ps, t);
        q.store(Qs, t);
        __syncthreads();
        if (k0 + BK < kend) {
            p.load(k0 + BK, kend, t);
            q.load(k0 + BK, kend, t);
        }
        if (ROWSUM) {   // db = sum over the reduction axis of P (dy^T): bias gradient for free
            if (blockIdx.x == 0 && t < BM) {
#pragma unroll
                for (int kk = 0; kk < BK; ++kk) rsum += Ps[kk][t];
            }
        }
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            float a[WM], b[WN];
#pragma unroll
            for (int x = 0; x < WM; ++x) a[x] = Ps[kk * 2 + lrow][(wi * WM + x) * 32 + lcol];
#pragma unroll
            for (int y = 0; y < WN; ++y) b[y] = Qs[kk * 2 + lrow][(wj * WN + y) * 32 + lcol];
#pragma unroll
            for (int x = 0; x < WM; ++x)
#pragma unroll
                for (int y = 0; y < WN; ++y)
                    acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[x], b[y], acc[x][y], 0, 0, 0);
        }
    }

    // C/D layout of 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int y = 0; y < WN; ++y) {
        const int j = j0 + (wj * WN + y) * 32 + lcol;
        if (!e.col(j)) continue;
#pragma unroll
        for (int x = 0; x < WM; ++x) {
            const int ib = i0 + (wi * WM + x) * 32 + 4 * lrow;
#pragma unroll
            for (int r = 0; r < 16; ++r) e.put(ib + (r & 3) + 8 * (r >> 2), j, acc[x][y][r], split);
        }
    }
    if (ROWSUM) {
        if (blockIdx.x == 0 && t < BM && i0 + t < rowsum_rows) {
            float *dst = rowsum_out + (size_t)split * rowsum_stride + i0 + t;
            if (gridDim.z == 1 && rowsum_accumulate) rsum += *dst;
            *dst = rsum;
        }
    }
}

// out[idx] (+)= sum_s ws[s * stride + idx]
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float *ws, float *out, int n,
                                                            int splits, size_t stride, int accumulate) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    float s = 0.f;
    for (int z = 0; z < splits; ++z) s += ws[(size_t)z * stride + idx];
    if (accumulate) s += out[idx];
    out[idx] = s;
}

// ------------------------------------------------------------------------------------------
// host-side dispatch
// ------------------------------------------------------------------------------------------
struct TileChoice { int wm, wn; };

// Prefer 128-wide tiles; fall back to 64 where the extent is small or the grid would leave
// most of the 256 CUs idle.
inline TileChoice choose_tile(int I, int J, int splits_hint) {
    TileChoice c{2, 2};
    if (I <= 64) c.wm = 1;
    if (J <= 64) c.wn = 1;
    auto blocks = [&](const TileChoice &t) {
        return (long)((I + 64 * t.wm - 1) / (64 * t.wm)) * ((J + 64 * t.wn - 1) / (64 * t.wn)) * splits_hint;
    };
    while (blocks(c) < 512 && (c.wm > 1 || c.wn > 1)) {
        if (c.wm > 1 && (c.wn == 1 || I >= J)) c.wm = 1; else c.wn = 1;
    }
    return c;
}

template <template <int> class PL, template <int> class QL, class E, bool ROWSUM, class PF, class QF>
int launch_igemm(TileChoice tc, PF make_p, QF make_q, E e, int I, int J, int K, int splits, int klen,
                 float *rowsum_out, size_t rowsum_stride, int rowsum_rows, int rowsum_acc, hipStream_t st) {
#define MVAE_LAUNCH(WM, WN)                                                                      \
    {                                                                                            \
        PL<64 * WM> p; make_p(p);                                                                \
        QL<64 * WN> q; make_q(q);                                                                \
        dim3 grid((J + 64 * WN - 1) / (64 * WN), (I + 64 * WM - 1) / (64 * WM), splits);         \
        hipLaunchKernelGGL((igemm_kernel<PL<64 * WM>, QL<64 * WN>, E, WM, WN, ROWSUM>), grid,    \
                           dim3(NTHREADS), 0, st, p, q, e, K, klen, rowsum_out, rowsum_stride,   \
                           rowsum_rows, rowsum_acc);                                             \
    }
    if (tc.wm == 2 && tc.wn == 2) MVAE_LAUNCH(2, 2)
    else if (tc.wm == 2 && tc.wn == 1) MVAE_LAUNCH(2, 1)
    else if (tc.wm == 1 && tc.wn == 2) MVAE_LAUNCH(1, 2)
    else MVAE_LAUNCH(1, 1)
#undef MVAE_LAUNCH
    return mvae_launch_status();
}

// Split plan for batch reductions: enough blocks to fill the chip, splits aligned to BK.
struct SplitPlan { int splits, klen; };
inline SplitPlan plan_splits(int I, int J, int K, TileChoice tc) {
    const long tiles = (long)((I + 64 * tc.wm - 1) / (64 * tc.wm)) * ((J + 64 * tc.wn - 1) / (64 * tc.wn));
    long want = (512 + tiles - 1) / tiles;
    const long maxs = (K + 4 * BK - 1) / (4 * BK);      // at least 4 k-steps per split
    if (want > maxs) want = maxs;
    if (want < 1) want = 1;
    if (want > 256) want = 256;
    int klen = (int)(((K + want - 1) / want + BK - 1) / BK * BK);
    int splits = (K + klen - 1) / klen;
    return {splits, klen};
}

inline size_t wgrad_ws_floats(int I, int J, int K) {
    // worst case over tile choices: plan with the smallest tiles -> most splits is bounded by 256
    TileChoice tc = choose_tile(I, J, 4);
    SplitPlan sp = plan_splits(I, J, K, tc);
    return (size_t)sp.splits * ((size_t)I * J + I);
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
    TileChoice tc = choose_tile(I, J, 1);
    EpNCHW e;
    e.out = pre; e.act = act; e.dpre = dpre;
    e.C = g.Cout; e.HW = g.OH * g.OW; e.Wfull = g.OW; e.H2 = g.OH; e.W2 = g.OW;
    e.sy = 1; e.py = 0; e.px = 0; e.J = J; e.off = 0;
    auto mp = [&](auto &p) { p.src = w; p.ld = K; p.R = I; p.vec = aligned16(w) ? 1 : 0; };
    auto mq = [&](auto &q) { q.x = x; q.g = g; q.Mtot = J; };
    return launch_igemm<LdRowsK, LdIm2col, EpNCHW, false>(tc, mp, mq, e, I, J, K, 1, (K + BK - 1) / BK * BK,
                                                          nullptr, 0, 0, 0, st);
}

// ---- conv dgrad form: dx[n][ci][ih][iw] = sum_(co,kh,kw) w[co][ci][kh][kw] * dy[n][co][oh][ow],
//      one launch per output parity class (4 for stride 2, 1 for stride 1) ----
int conv_dgrad_impl(const float *dy, const float *w, float *dx, float *act, const float *dpre,
                    ConvGeom g, hipStream_t st) {
    const int s = g.stride, tlog = (s == 2) ? 1 : 2;
    const int H2 = g.H / s, W2 = g.W / s;
    const int I = g.Cin, J = g.B * H2 * W2, K = g.Cout << (2 * tlog);
    TileChoice tc = choose_tile(I, J, s * s);
    for (int ph = 0; ph < s; ++ph)
        for (int pw = 0; pw < s; ++pw) {
            const int kh0 = (ph + g.pad) % s, kw0 = (pw + g.pad) % s;
            EpNCHW e;
            e.out = dx; e.act = act; e.dpre = dpre;
            e.C = g.Cin; e.HW = g.H * g.W; e.Wfull = g.W; e.H2 = H2; e.W2 = W2;
            e.sy = s; e.py = ph; e.px = pw; e.J = J; e.off = 0;
            auto mp = [&](auto &p) {
                p.w = w; p.Cin = g.Cin; p.stride = s; p.kh0 = kh0; p.kw0 = kw0; p.tlog = tlog;
            };
            auto mq = [&](auto &q) {
                q.dy = dy; q.g = g; q.Mtot = J; q.H2 = H2; q.W2 = W2; q.ph = ph; q.pw = pw;
                q.kh0 = kh0; q.kw0 = kw0; q.tlog = tlog;
            };
            int rc = launch_igemm<LdDgradW, LdDgradDy, EpNCHW, false>(tc, mp, mq, e, I, J, K, 1,
                                                                      (K + BK - 1) / BK * BK, nullptr, 0, 0, 0, st);
            if (rc) return rc;
        }
    return MVAE_OK;
}

// ---- conv wgrad form: dw[co][(ci,kh,kw)] = sum_(n,oh,ow) dy[n][co][oh][ow] * x[n][ci][ih][iw] ----
int conv_wgrad_impl(const float *dy, const float *x, float *dw, ConvGeom g, int flags, void *ws,
                    size_t ws_bytes, hipStream_t st) {
    const int I = g.Cout, J = g.Cin * 16, K = g.B * g.OH * g.OW;
    TileChoice tc = choose_tile(I, J, 4);
    SplitPlan sp = plan_splits(I, J, K, tc);
    const size_t stride = (size_t)I * J;
    if (sp.splits > 1 && ws_bytes < sp.splits * stride * sizeof(float)) return MVAE_ERR_WS;
    EpPartial e;
    e.ws = (float *)ws; e.I = I; e.J = J; e.split_stride = stride;
    e.direct = sp.splits == 1 ? dw : nullptr; e.accumulate = (flags & MVAE_ACCUMULATE) ? 1 : 0;
    auto mp = [&](auto &p) { p.dy = dy; p.g = g; };
    auto mq = [&](auto &q) { q.x = x; q.g = g; q.J = J; };
    int rc = launch_igemm<LdWgradDy, LdWgradX, EpPartial, false>(tc, mp, mq, e, I, J, K, sp.splits, sp.klen,
                                                                 nullptr, 0, 0, 0, st);
    if (rc) return rc;
    if (sp.splits > 1) {
        const int n = I * J;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((n + 255) / 256), dim3(256), 0, st,
                           (const float *)ws, dw, n, sp.splits, stride, e.accumulate);
        return mvae_launch_status();
    }
    return MVAE_OK;
}

}  // namespace

// ==========================================================================================
// C ABI
// ==========================================================================================
MVAE_EXPORT int mvae_abi_version(void) { return 1; }

MVAE_EXPORT size_t mvae_wgrad_ws_bytes(int rows_out, int cols_out, int reduce_len) {
    if (rows_out <= 0 || cols_out <= 0 || reduce_len <= 0) return 0;
    return wgrad_ws_floats(rows_out, cols_out, reduce_len) * sizeof(float);
}

MVAE_EXPORT int mvae_linear_fwd(const float *x, int ldx, const float *w, const float *bias,
                                float *pre, float *act, int ldy, const float *mask, float mask_scale,
                                int M, int N, int K, mvae_stream_t stream) {
    if (!x || !w || (!pre && !act) || M <= 0 || N <= 0 || K <= 0 || ldx < K || ldy < N) return MVAE_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    TileChoice tc = choose_tile(M, N, 1);
    EpRowMajor e;
    e.out = pre; e.act = act; e.ld = ldy; e.bias = bias; e.dpre = nullptr; e.ldp = 0;
    e.mask = mask; e.ldm = N; e.mask_scale = mask_scale; e.I = M; e.J = N; e.accumulate = 0;
    auto mp = [&](auto &p) { p.src = x; p.ld = ldx; p.R = M; p.vec = (aligned16(x) && ldx % 4 == 0) ? 1 : 0; };
    auto mq = [&](auto &q) { q.src = w; q.ld = K; q.R = N; q.vec = (aligned16(w) && K % 4 == 0) ? 1 : 0; };
    return launch_igemm<LdRowsK, LdRowsK, EpRowMajor, false>(tc, mp, mq, e, M, N, K, 1, (K + BK - 1) / BK * BK,
                                                             nullptr, 0, 0, 0, st);
}

MVAE_EXPORT int mvae_linear_dgrad(const float *dy, int lddy, const float *w, float *dx, int lddx,
                                  const float *pre_in, const float *mask, float mask_scale,
                                  int M, int N, int K, int flags, mvae_stream_t stream) {
    if (!dy || !w || !dx || M <= 0 || N <= 0 || K <= 0 || lddy < N || lddx < K) return MVAE_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    // D[i = m][j = k] = sum_n dy[m][n] * w[n][k]
    TileChoice tc = choose_tile(M, K, 1);
    EpRowMajor e;
    e.out = dx; e.act = nullptr; e.ld = lddx; e.bias = nullptr; e.dpre = pre_in; e.ldp = K;
    e.mask = mask; e.ldm = K; e.mask_scale = mask_scale; e.I = M; e.J = K;
    e.accumulate = (flags & MVAE_ACCUMULATE) ? 1 : 0;
    auto mp = [&](auto &p) { p.src = dy; p.ld = lddy; p.R = M; p.vec = (aligned16(dy) && lddy % 4 == 0) ? 1 : 0; };
    auto mq = [&](auto &q) { q.src = w; q.ld = K; q.R = K; q.vec = (aligned16(w) && K % 4 == 0) ? 1 : 0; };
    return launch_igemm<LdRowsK, LdRowsMN, EpRowMajor, false>(tc, mp, mq, e, M, K, N, 1, (N + BK - 1) / BK * BK,
                                                              nullptr, 0, 0, 0, st);
}

MVAE_EXPORT int mvae_linear_wgrad(const float *dy, int lddy, const float *x, int ldx, float *dw, float *db,
                                  int M, int N, int K, int flags, void *ws, size_t ws_bytes,
                                  mvae_stream_t stream) {
    if (!dy || !x || !dw || M <= 0 || N <= 0 || K <= 0 || lddy < N || ldx < K) return MVAE_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    // D[i = n][j = k] = sum_m dy[m][n] * x[m][k]
    TileChoice tc = choose_tile(N, K, 4);
    SplitPlan sp = plan_splits(N, K, M, tc);
    const size_t stride = (size_t)N * K + N;          // dw partial followed by db partial
    if (sp.splits > 1 && (!ws || ws_bytes < sp.splits * stride * sizeof(float))) return MVAE_ERR_WS;
    const int acc = (flags & MVAE_ACCUMULATE) ? 1 : 0;
    EpPartial e;
    e.ws = (float *)ws; e.I = N; e.J = K; e.split_stride = stride;
    e.direct = sp.splits == 1 ? dw : nullptr; e.accumulate = acc;
    auto mp = [&](auto &p) { p.src = dy; p.ld = lddy; p.R = N; p.vec = (aligned16(dy) && lddy % 4 == 0) ? 1 : 0; };
    auto mq = [&](auto &q) { q.src = x; q.ld = ldx; q.R = K; q.vec = (aligned16(x) && ldx % 4 == 0) ? 1 : 0; };
    int rc;
    if (db) {
        // row sums of P = dy^T are the bias gradient; partials live right after each dw partial
        float *rs = sp.splits == 1 ? db : (float *)ws + (size_t)N * K;
        rc = launch_igemm<LdRowsMN, LdRowsMN, EpPartial, true>(tc, mp, mq, e, N, K, M, sp.splits, sp.klen, rs,
                                                               stride, N, acc, st);
    } else {
        rc = launch_igemm<LdRowsMN, LdRowsMN, EpPartial, false>(tc, mp, mq, e, N, K, M, sp.splits, sp.klen,
                                                                nullptr, 0, 0, 0, st);
    }
    if (rc) return rc;
    if (sp.splits > 1) {
        const int n = N * K;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((n + 255) / 256), dim3(256), 0, st, (const float *)ws, dw,
                           n, sp.splits, stride, acc);
        if (db)
            hipLaunchKernelGGL(splitk_reduce_kernel, dim3((N + 255) / 256), dim3(256), 0, st,
                               (const float *)ws + (size_t)N * K, db, N, sp.splits, stride, acc);
        return mvae_launch_status();
    }
    return MVAE_OK;
}

MVAE_EXPORT int mvae_conv2d_k4_fwd(const float *x, const float *w, float *pre, float *act, int B, int Cin,
                                   int H, int W, int Cout, int stride, int pad, mvae_stream_t stream) {
    if (!x || !w || (!pre && !act) || !conv_args_ok(B, Cin, H, W, Cout, stride, pad)) return MVAE_ERR_ARG;
    return conv_fwd_impl(x, w, pre, act, nullptr, make_geom(B, Cin, H, W, Cout, stride, pad), (hipStream_t)stream);
}

MVAE_EXPORT int mvae_conv2d_k4_dgrad(const float *dy, const float *w, float *dx, const float *pre_in, int B,
                                     int Cin, int H, int W, int Cout, int stride, int pad,
                                     mvae_stream_t stream) {
    if (!dy || !w || !dx || !conv_args_ok(B, Cin, H, W, Cout, stride, pad)) return MVAE_ERR_ARG;
    return conv_dgrad_impl(dy, w, dx, nullptr, pre_in, make_geom(B, Cin, H, W, Cout, stride, pad),
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
                                    int H, int W, int Cout, int stride, int pad, mvae_stream_t stream) {
    ConvGeom g;
    if (!x || !w || (!pre && !act) || B <= 0 || !convT_geom(B, Cin, H, W, Cout, stride, pad, &g)) return MVAE_ERR_ARG;
    return conv_dgrad_impl(x, w, pre, act, nullptr, g, (hipStream_t)stream);
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
