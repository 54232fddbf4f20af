// gru.hip -- the recurrent text stacks of the MultiMNIST MVAE (last item):
// multimnist/model.py:145-235 -- TextEncoder (Embedding -> bidirectional 1-layer nn.GRU -> last step, directions
// summed -> Linear) and TextDecoder (4 autoregressive steps of Embedding+Swish | z -> 2-layer nn.GRU with
// inter-layer Dropout(0.1) -> | z -> Linear, greedy arg-max feedback).
//
// The matrix products of a GRU step (x.W_ih^T + b_ih, h.W_hh^T + b_hh: [B,200..264] x [600, .]) are Linear
// launches (mvae_linear_fwd / _dgrad / _wgrad with leading dimensions, so the `torch.cat((c_in, z))` and
// `torch.cat((c_out, z))` of model.py:222,226 are column ranges of one buffer, never copies of h).  This file holds
// what is left: the gate arithmetic of a cell (forward / backward), the plain Embedding with strides (index column
// t of text[B,4]; output into a column range), a strided 2-D copy / add, and the row arg-max of the feedback.
// All HBM/latency-bound element work on [B, 200]-sized tensors; one thread per element, lanes along the hidden axis.
#include "common.h"

namespace {

inline int ew_blocks(size_t n) {
    size_t b = (n + 255) / 256;
    return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}

// libm tanh: 2 s(2x) - 1 on the fast exp / rcp loses relative accuracy near 0 (a difference of two O(1) numbers);
// these tensors are [B, 200] -- the instruction count does not matter
__device__ __forceinline__ float tanhf_(float x) { return tanhf(x); }

// nn.GRU cell (torch gate order r, z, n):  r = s(gi_r + gh_r), z = s(gi_z + gh_z), n = tanh(gi_n + r * gh_n),
// h' = (1 - z) * n + z * h.   gi / gh are [B, 3H] (biases included).  gates[B, 4H] = r | z | n | gh_n for the backward.
__global__ __launch_bounds__(256) void gru_cell_fwd_kernel(const float *gi, int ldgi, const float *gh, int ldgh,
                                                           const float *h_prev, int ldh, float *h_new, int ldo,
                                                           float *gates, int B, int H) {
    const size_t total = (size_t)B * H;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int b = (int)(i / H), j = (int)(i - (size_t)b * H);
        const float *a = gi + (size_t)b * ldgi, *c = gh + (size_t)b * ldgh;
        const float r = sigmoidf_(a[j] + c[j]);
        const float z = sigmoidf_(a[H + j] + c[H + j]);
        const float ghn = c[2 * H + j];
        const float n = tanhf_(a[2 * H + j] + r * ghn);
        const float hp = h_prev[(size_t)b * ldh + j];
        h_new[(size_t)b * ldo + j] = (1.0f - z) * n + z * hp;
        if (gates) {
            float *g = gates + (size_t)b * 4 * H;
            g[j] = r; g[H + j] = z; g[2 * H + j] = n; g[3 * H + j] = ghn;
        }
    }
}

// Backward of the cell: given dh' -> dgi[B,3H], dgh[B,3H] (gradients of the two pre-activation products) and
// dh_prev = dh' * z (the caller adds dgh . W_hh on top with an accumulating data-gradient launch).
__global__ __launch_bounds__(256) void gru_cell_bwd_kernel(const float *dh_new, int lddh, const float *dh_extra,
                                                           int ldde, const float *gates, const float *h_prev,
                                                           int ldh, float *dgi, float *dgh, float *dh_prev, int B,
                                                           int H) {
    const size_t total = (size_t)B * H;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int b = (int)(i / H), j = (int)(i - (size_t)b * H);
        const float *g = gates + (size_t)b * 4 * H;
        const float r = g[j], z = g[H + j], n = g[2 * H + j], ghn = g[3 * H + j];
        float d = dh_new[(size_t)b * lddh + j];
        if (dh_extra) d += dh_extra[(size_t)b * ldde + j];       // a second contribution to dh' (the carried state)
        const float hp = h_prev[(size_t)b * ldh + j];
        const float dn_pre = d * (1.0f - z) * (1.0f - n * n);
        const float dz_pre = d * (hp - n) * z * (1.0f - z);
        const float dr_pre = dn_pre * ghn * r * (1.0f - r);
        float *a = dgi + (size_t)b * 3 * H, *c = dgh + (size_t)b * 3 * H;
        a[j] = dr_pre; a[H + j] = dz_pre; a[2 * H + j] = dn_pre;
        c[j] = dr_pre; c[H + j] = dz_pre; c[2 * H + j] = dn_pre * r;
        dh_prev[(size_t)b * H + j] = d * z;
    }
}

// out[r, 0:width] = act(w[idx[r * idx_stride], :])   (nn.Embedding, optionally + Swish)
__global__ __launch_bounds__(256) void embedding_fwd_kernel(const int64_t *idx, int idx_stride, const float *w,
                                                            float *out, int ldo, int R, int n_classes, int width,
                                                            int swish) {
    const size_t total = (size_t)R * width;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int r = (int)(i / width), j = (int)(i - (size_t)r * width);
        int c = (int)idx[(size_t)r * idx_stride];
        c = min(max(c, 0), n_classes - 1);
        const float v = w[(size_t)c * width + j];
        out[(size_t)r * ldo + j] = swish ? swishf_(v) : v;
    }
}

// dw[c, j] (+)= act'(w[c, j]) * sum_{r : idx[r] == c} dout[r, j], rows added in order: deterministic.
// One thread per (class, column); R is a batch (<= a few thousand rows), n_classes * width a few thousand threads.
__global__ __launch_bounds__(256) void embedding_bwd_kernel(const int64_t *idx, int idx_stride, const float *w,
                                                            const float *dout, int ldd, float *dw, int R,
                                                            int n_classes, int width, int swish, int accumulate) {
    const int o = blockIdx.x * 256 + threadIdx.x;
    if (o >= n_classes * width) return;
    const int c = o / width, j = o - c * width;
    float s = 0.f;
#pragma unroll 4
    for (int r = 0; r < R; ++r) {
        int cr = (int)idx[(size_t)r * idx_stride];
        cr = min(max(cr, 0), n_classes - 1);
        const float v = dout[(size_t)r * ldd + j];
        s += (cr == c) ? v : 0.f;
    }
    if (swish) s *= swish_grad_(w[o]);
    dw[o] = accumulate ? dw[o] + s : s;
}

// dst[r, 0:cols] (+)= src[r, 0:cols] (* mask[r, 0:cols] * scale)
__global__ __launch_bounds__(256) void copy2d_kernel(const float *src, int lds, float *dst, int ldd, const float *mask,
                                                     int ldm, float scale, int rows, int cols, int accumulate) {
    const size_t total = (size_t)rows * cols;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int r = (int)(i / cols), j = (int)(i - (size_t)r * cols);
        float v = src[(size_t)r * lds + j];
        if (mask) v *= mask[(size_t)r * ldm + j] * scale;
        float *d = dst + (size_t)r * ldd + j;
        *d = accumulate ? *d + v : v;
    }
}

// out[r] = index of the first maximum of x[r, 0:K]  (torch.max(F.log_softmax(c_out, dim=1), dim=1)[1],
// multimnist/model.py:211: log_softmax is monotone per row, so the arg-max of the logits)
__global__ __launch_bounds__(256) void argmax_rows_kernel(const float *x, int ldx, int64_t *out, int R, int K) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= R) return;
    const float *row = x + (size_t)r * ldx;
    float best = row[0];
    int bi = 0;
    for (int k = 1; k < K; ++k)
        if (row[k] > best) { best = row[k]; bi = k; }
    out[r] = bi;
}

}  // namespace

MVAE_EXPORT int mvae_gru_cell_fwd(const float *gi, int ldgi, const float *gh, int ldgh, const float *h_prev, int ldh,
                                  float *h_new, int ldo, float *gates, int B, int H, mvae_stream_t stream) {
    if (!gi || !gh || !h_prev || !h_new || B <= 0 || H <= 0 || ldgi < 3 * H || ldgh < 3 * H || ldh < H || ldo < H)
        return MVAE_ERR_ARG;
    hipLaunchKernelGGL(gru_cell_fwd_kernel, dim3(ew_blocks((size_t)B * H)), dim3(256), 0, (hipStream_t)stream, gi, ldgi,
                       gh, ldgh, h_prev, ldh, h_new, ldo, gates, B, H);
    return mvae_launch_status();
}

MVAE_EXPORT int mvae_gru_cell_bwd(const float *dh_new, int lddh, const float *dh_extra, int ldde, const float *gates,
                                  const float *h_prev, int ldh, float *dgi, float *dgh, float *dh_prev, int B, int H,
                                  mvae_stream_t stream) {
    if (!dh_new || !gates || !h_prev || !dgi || !dgh || !dh_prev || B <= 0 || H <= 0 || lddh < H || ldh < H ||
        (dh_extra && ldde < H))
        return MVAE_ERR_ARG;
    hipLaunchKernelGGL(gru_cell_bwd_kernel, dim3(ew_blocks((size_t)B * H)), dim3(256), 0, (hipStream_t)stream, dh_new,
                       lddh, dh_extra, ldde, gates, h_prev, ldh, dgi, dgh, dh_prev, B, H);
    return mvae_launch_status();
}

MVAE_EXPORT int mvae_embedding_fwd(const int64_t *idx, int idx_stride, const float *w, float *out, int ldo, int R,
                                   int n_classes, int width, int flags, mvae_stream_t stream) {
    if (!idx || !w || !out || R <= 0 || n_classes <= 0 || width <= 0 || idx_stride < 1 || ldo < width) return MVAE_ERR_ARG;
    hipLaunchKernelGGL(embedding_fwd_kernel, dim3(ew_blocks((size_t)R * width)), dim3(256), 0, (hipStream_t)stream, idx,
                       idx_stride, w, out, ldo, R, n_classes, width, (flags & MVAE_ACT_SWISH) ? 1 : 0);
    return mvae_launch_status();
}

MVAE_EXPORT int mvae_embedding_bwd(const int64_t *idx, int idx_stride, const float *w, const float *dout, int ldd,
                                   float *dw, int R, int n_classes, int width, int flags, mvae_stream_t stream) {
    if (!idx || !w || !dout || !dw || R <= 0 || n_classes <= 0 || width <= 0 || idx_stride < 1 || ldd < width)
        return MVAE_ERR_ARG;
    const int n = n_classes * width;
    hipLaunchKernelGGL(embedding_bwd_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, idx, idx_stride, w,
                       dout, ldd, dw, R, n_classes, width, (flags & MVAE_ACT_SWISH) ? 1 : 0,
                       (flags & MVAE_ACCUMULATE) ? 1 : 0);
    return mvae_launch_status();
}

MVAE_EXPORT int mvae_copy2d(const float *src, int lds, float *dst, int ldd, const float *mask, int ldm, float scale,
                            int rows, int cols, int flags, mvae_stream_t stream) {
    if (!src || !dst || rows <= 0 || cols <= 0 || lds < cols || ldd < cols || (mask && ldm < cols)) return MVAE_ERR_ARG;
    hipLaunchKernelGGL(copy2d_kernel, dim3(ew_blocks((size_t)rows * cols)), dim3(256), 0, (hipStream_t)stream, src, lds,
                       dst, ldd, mask, ldm, scale, rows, cols, (flags & MVAE_ACCUMULATE) ? 1 : 0);
    return mvae_launch_status();
}

MVAE_EXPORT int mvae_argmax_rows(const float *x, int ldx, int64_t *out, int R, int K, mvae_stream_t stream) {
    if (!x || !out || R <= 0 || K <= 0 || ldx < K) return MVAE_ERR_ARG;
    hipLaunchKernelGGL(argmax_rows_kernel, dim3((R + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, ldx, out, R, K);
    return mvae_launch_status();
}
