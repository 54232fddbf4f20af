// conv_patch.h -- 4x4 conv FORWARD form (Conv2d forward, ConvTranspose2d data gradient) from an LDS input patch (round 6).
// Conv2d(32,64,4,2,1) / (64,128,4,2,1) / (128,256,4,1,0) and the data gradients of ConvTranspose2d(256,128,4,1,0) /
// (128,64,4,2,1) / (64,32,4,2,1): celeba/model.py:78-85,117-124, fashionmnist/model.py:80-82,110-112.
//
//   out[n][co][oh][ow] = sum over (ci, kh, kw) of  w[co][ci][kh][kw] * x[n][ci][oh * s - p + kh][ow * s - p + kw]
//
// The implicit-GEMM launch of gemm_core.h builds an im2col tile per k-step: every thread gathers 8 dwords of x into registers
// and stages them through LDS (each element of x sixteen / s^2 times).  Same remedy as convt_patch.h: a block owns 64
// consecutive output positions j = (n, oh, ow) x 64 output channels;
//   * the input those positions can touch -- whole images (5 x 5, 7 x 7, 8 x 8 outputs) or a band of rows (16 x 16) -- comes
//     in ONCE per phase of 8 input channels, in its natural layout, by 16-byte LDS-DMA pieces, double-buffered (the next
//     phase's pieces are issued behind the weights of the phase's first step and have three steps to land);
//   * the B fragment of reduction index k = (ci, kh, kw) for the lane's position is ONE ds_read_b32 at a per-lane tap address
//     (8 of them: the half wave's two kh of each k-chunk x 4 kw; a tap outside the image points at a zero slot) + the
//     channel as an immediate;
//   * the weights are used as they lie in memory -- w[co][k], k contiguous -- through a 3-deep LDS-DMA ring of [64][32] tiles
//     (float4 slots XOR-swizzled on the SOURCE side: conflict-free ds_read_b128, one read per four matrix instructions);
//   * 2 x 2 waves of 32 x 32; epilogue = gemm_core.h's EpNCHW::put_b (pre-activation / Swish / Swish' outputs), unchanged.
// The k order per accumulator is the gather launch's (chunks of 8: lanes 0-31 take k = 8c + j, lanes 32-63 k = 8c + 4 + j).
#pragma once
#include "convt_patch.h"

namespace {

#ifndef MVAE_CONV_PATCH
#define MVAE_CONV_PATCH 0           // measured equal to 10 % slower than igemm_kernel<LdRowsKT, LdIm2colT> on every layer
                                    // (profiles/r06_conv_patch_bench.txt): off; -DMVAE_CONV_PATCH=1 builds it in (A/B builds)
#endif
#ifndef MVAE_CONV_PATCH_MINBLOCKS
#define MVAE_CONV_PATCH_MINBLOCKS 384
#endif

struct ConvPatchGeo {
    int B, Cin, Cout;
    int H, W, HW, OH, OW, OHW;      // input map, output map
    int stride, pad;
    int J, K;                       // B * OHW columns, Cin * 16
    int mode_a;                     // 1: whole images in the patch, 0: a band of input rows of ONE image
    int ps_raw;                     // floats per channel before the zero slot
};

constexpr int VP_KPH = 8, VP_SPP = VP_KPH / 2, VP_BK = 32;      // channels per phase, k-steps per phase (2 channels each)
constexpr int VP_WT = 64 * VP_BK;                               // floats per weight stage

// PS: floats per channel in the patch incl. the zero float4 at its end; NUI: patch DMA instructions per thread and phase
template <int PS, int NUI>
__global__ __launch_bounds__(256, 2) void conv_patch_kernel(const float *__restrict__ x, const float *__restrict__ w, EpNCHW e,
                                                            ConvPatchGeo g) {
    static_assert(PS % 4 == 0, "16-byte pieces");
    constexpr int PSV = PS / 4;
    constexpr int PATCH_FLOATS = NUI * 256 * 4;
    constexpr int NPW = 2;
    static_assert(NPW + NUI <= 63, "vmcnt");
    extern __shared__ __attribute__((aligned(16))) float lds[];     // [2 patch buffers | ring of 3 weight stages]
    const int t = threadIdx.x, lane = t & 63;
    const int wave = g2_uni(t >> 6);
    const int wi = wave >> 1, wj = wave & 1;
    const int lrow = lane >> 5, lcol = lane & 31;
    const int j0 = blockIdx.x * 64, i0 = blockIdx.y * 64;
    const unsigned lds0 = (unsigned)(unsigned long)(g2_lds_void *)lds;
    const unsigned ring0 = lds0 + 2 * PATCH_FLOATS * 4;

    // ---- the lane's output position and its tap places in a channel's patch slab
    const int j = j0 + wj * 32 + lcol;
    const bool jok = j < g.J;
    const int jj = jok ? j : 0;
    const int n = jj / g.OHW, rem = jj - n * g.OHW;
    const int oh = rem / g.OW, ow = rem - oh * g.OW;
    const int n0 = g2_uni(j0 / g.OHW);
    const int n1 = g2_uni(min(j0 + 63, g.J - 1) / g.OHW);
    const int oh0 = g2_uni((j0 - n0 * g.OHW) / g.OW);          // mode b: first output row of the block
    const int rb0 = oh0 * g.stride - g.pad;                     // ... and the input row its first tap row reads
    // the half wave's tap rows: chunk parity q = 0, 1 -> kh = 2 q + lrow
    int tp[2][4];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int kw = 0; kw < 4; ++kw) {
            const int kh = 2 * q + lrow;
            const int ih = oh * g.stride - g.pad + kh, iw = ow * g.stride - g.pad + kw;
            const bool ok = jok && ih >= 0 && ih < g.H && iw >= 0 && iw < g.W;
            const int pidx = g.mode_a ? (n - n0) * g.HW + ih * g.W + iw : (ih - rb0) * g.W + iw;
            tp[q][kw] = (ok ? pidx : PS - 1) * 4;
        }
    // ---- patch DMA: unit u = i * 256 + t -> (channel u / PSV, float4 u % PSV) -> bytes from (image n0, first channel of the phase)
    int pvoff[NUI];
#pragma unroll
    for (int i = 0; i < NUI; ++i) {
        const int u = i * 256 + t, c = u / PSV, q = (u - c * PSV) * 4;
        int off = BUF_OOB;
        if (c < VP_KPH && q < g.ps_raw) {
            if (g.mode_a) {
                const int img = q / g.HW, pos = q - img * g.HW;
                if (n0 + img <= n1) off = ((img * g.Cin + c) * g.HW + pos) * 4;
            } else {
                const int row = q / g.W, ih = rb0 + row;
                if (ih >= 0 && ih < g.H) off = (c * g.HW + ih * g.W + (q - row * g.W)) * 4;
            }
        }
        pvoff[i] = off;
    }
    const BufBase xb = buf_base(x + (size_t)n0 * g.Cin * g.HW);
    auto issue_patch = [&](int phase) {
        const i32x4_t rs = g2_rsrc(xb, (long)phase * VP_KPH * g.HW, 0x7fffffff);
        const unsigned base = lds0 + (phase & 1) * PATCH_FLOATS * 4;
        asm volatile("s_nop 4" ::: "memory");
#pragma unroll
        for (int i = 0; i < NUI; ++i) g2_dma16(rs, pvoff[i], g2_uni(base + (i * 256 + wave * 64) * 16));
    };
    // ---- weights: [64 rows][8 float4], slot f of row r holds float4 f ^ swz(r) (source-side swizzle), two pieces per thread
    auto swz = [](int r) { return (r >> 1) & 7; };
    int wvoff[2];
#pragma unroll
    for (int v = 0; v < 2; ++v) {
        const int slot = v * 256 + t, r = slot >> 3, f = (slot & 7) ^ swz(r);
        wvoff[v] = (i0 + r < g.Cout) ? (r * g.K + f * 4) * 4 : BUF_OOB;
    }
    const BufBase wb = buf_base(w + (size_t)i0 * g.K);
    const int steps_total = g.Cin / 2;
    const int nphase = g.Cin / VP_KPH;
    auto issue_w = [&](int u) {
        const unsigned dst = ring0 + ((u % 3) * VP_WT) * 4;
        const i32x4_t rs = g2_rsrc(wb, (long)u * VP_BK, 0x7fffffff);
        asm volatile("s_nop 4" ::: "memory");
        g2_dma16(rs, wvoff[0], g2_uni(dst + wave * 1024));
        g2_dma16(rs, wvoff[1], g2_uni(dst + 4096 + wave * 1024));
    };
    // fragment addresses of the four 8-k chunks of a step: row wi * 32 + lcol, float4 (2 c + lrow) ^ swz(row)
    int aoff[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) aoff[c] = ((wi * 32 + lcol) * VP_BK + 4 * ((2 * c + lrow) ^ swz(lcol))) * 4;    // (wi * 32 is a multiple of 16: the swizzle is the lane's)

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    issue_w(0);
    if (steps_total > 1) issue_w(1);
    issue_patch(0);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    int u = 0;
    for (int phase = 0; phase < nphase; ++phase) {
        const bool more = phase + 1 < nphase;               // block-uniform
        const char *Pp = reinterpret_cast<const char *>(lds) + (phase & 1) * PATCH_FLOATS * 4;
#pragma unroll
        for (int ks = 0; ks < VP_SPP; ++ks, ++u) {
            if (u > 0) {
                // what may stay in flight behind step u's weights: step u + 1's, and (steps 1, 2 of a phase) the next patch
                if (u + 1 >= steps_total) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                else if ((ks == 1 || ks == 2) && more) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(NPW + NUI) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(NPW) : "memory");
            }
            if (u + 2 < steps_total) issue_w(u + 2);
            if (ks == 0 && more) issue_patch(phase + 1);
            const char *Ws = reinterpret_cast<const char *>(lds) + (2 * PATCH_FLOATS + (u % 3) * VP_WT) * 4;
            const char *Pc = Pp + ks * 2 * PS * 4;                  // the step's first channel
#pragma unroll
            for (int c = 0; c < 4; ++c) {                       // chunk c: channel c / 2 of the step, tap rows 2 (c % 2) + lrow
                const float4 a4 = *reinterpret_cast<const float4 *>(Ws + aoff[c]);
                float bv[4];
#pragma unroll
                for (int kw = 0; kw < 4; ++kw) bv[kw] = *reinterpret_cast<const float *>(Pc + tp[c & 1][kw] + (c >> 1) * PS * 4);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, bv[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, bv[1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, bv[2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, bv[3], acc, 0, 0, 0);
            }
        }
    }
    EpNCHW et = e;
    et.set_class(0);
    et.tile(j0);
    (void)et.col(j);
    const int rb = __builtin_amdgcn_readfirstlane(i0 + wi * 32);
#pragma unroll
    for (int r = 0; r < 16; ++r) et.put_b(rb, r, acc[r]);
}

struct ConvPatchPlan { int kind; ConvPatchGeo g; dim3 grid; };
inline ConvPatchPlan conv_patch_plan(int B, int Cin, int H, int W, int Cout, int OH, int OW, int stride, int pad) {
    ConvPatchPlan pl; pl.kind = 0;
    if (!MVAE_CONV_PATCH) return pl;
#ifdef MVAE_TUNING
    if (getenv("MVAE_CONV_PATCH_OFF")) return pl;
#endif
    ConvPatchGeo &g = pl.g;
    g.B = B; g.Cin = Cin; g.Cout = Cout; g.H = H; g.W = W; g.HW = H * W; g.OH = OH; g.OW = OW; g.OHW = OH * OW;
    g.stride = stride; g.pad = pad;
    const long J = (long)B * g.OHW;
    if (J * 4 >= (1L << 31) || (long)B * Cin * g.HW >= (1L << 29) || Cin % VP_KPH != 0 || Cout % 64 != 0 || g.HW % 4 != 0) return pl;
    g.J = (int)J; g.K = Cin * 16;
    if (stride == 2 && pad == 1 && OH == 8 && OW == 8) { pl.kind = 1; g.mode_a = 1; g.ps_raw = 256; }           // one 16 x 16 image
    else if (stride == 2 && pad == 1 && OH == 16 && OW == 16) { pl.kind = 2; g.mode_a = 0; g.ps_raw = 320; }    // 10 rows of 32
    else if (stride == 2 && pad == 1 && OH == 7 && OW == 7) { pl.kind = 3; g.mode_a = 1; g.ps_raw = 588; }      // <= 3 images of 14 x 14
    else if (stride == 1 && pad == 0 && OH == 5 && OW == 5 && H == 8) { pl.kind = 4; g.mode_a = 1; g.ps_raw = 256; }   // <= 4 images of 8 x 8
    pl.grid = dim3((unsigned)cdiv(J, 64), (unsigned)(Cout / 64));
    if (pl.kind && (long)pl.grid.x * pl.grid.y < MVAE_CONV_PATCH_MINBLOCKS) pl.kind = 0;
    return pl;
}

template <int PS>
int launch_conv_patch(const ConvPatchPlan &pl, const float *x, const float *w, const EpNCHW &e, hipStream_t st) {
    constexpr int NUI = (VP_KPH * (PS / 4) + 255) / 256;
    constexpr size_t lds = ((size_t)2 * NUI * 256 * 4 + (size_t)3 * VP_WT) * sizeof(float);
    auto kern = conv_patch_kernel<PS, NUI>;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, pl.grid, dim3(256), lds, st, x, w, e, pl.g);
    return mvae_launch_status();
}

}  // namespace
