// gather.hip -- block gather / scatter-add between the per-term latent buffer z[T,B,D] and the
// per-attribute-decoder inputs of the CelebA-19 step, and the table-driven ELBO sums.
//
// celeba19/model.py:56-60 runs all 18 attribute decoders on every one of the 20+M model() calls
// (378+ tiny MLP forwards per step); only the (decoder, term) pairs that enter an ELBO need to
// exist.  The step gathers, for decoder i, the z blocks of exactly those terms
// (complete, the sampled subsets, single-attribute i) into zcat[i], runs each decoder once on its
// rows, and scatter-adds the input gradients back per term -- in a fixed order, no atomics.
#include "common.h"

namespace {

// dst[j] = src[idx[j]] for blocks of `block` floats
__global__ __launch_bounds__(256) void block_gather_kernel(const float *src, const int *idx, float *dst,
                                                           int n_dst, size_t block) {
    const size_t total = (size_t)n_dst * block;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t j = i / block, o = i - j * block;
        dst[i] = src[(size_t)idx[j] * block + o];
    }
}

// dst[t] += sum_{j : idx[j] == t} src[j]   (dst blocks t < n_dst; j ascending -> deterministic)
__global__ __launch_bounds__(256) void block_scatter_add_kernel(const float *src, const int *idx, float *dst,
                                                                int n_src, int n_dst, size_t block) {
    const size_t total = (size_t)n_dst * block;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t t = i / block, o = i - t * block;
        float s = 0.f;
        for (int j = 0; j < n_src; ++j)
            if (idx[j] == (int)t) s += src[(size_t)j * block + o];
        dst[i] += s;
    }
}

// out[idx[j]] (+)= coef[j] * vals[j];  *total (+)= sum_j coef[j] * vals[j]   (n is a few hundred)
__global__ void scatter_sums_kernel(const float *vals, const float *coef, const int *idx, float *out,
                                    float *total_out, int n, int accumulate_total) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float tot = 0.f;
    for (int j = 0; j < n; ++j) {
        const float v = (coef ? coef[j] : 1.f) * vals[j];
        if (out) out[idx[j]] += v;
        tot += v;
    }
    if (total_out) *total_out = accumulate_total ? *total_out + tot : tot;
}

inline int blocks_for(size_t n) {
    size_t b = (n + 255) / 256;
    return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}

}  // namespace

MVAE_EXPORT int mvae_block_gather(const float *src, const int *idx_dev, float *dst, int n_dst, size_t block_elems,
                                  mvae_stream_t stream) {
    if (!src || !idx_dev || !dst || n_dst <= 0 || block_elems == 0) return MVAE_ERR_ARG;
    hipLaunchKernelGGL(block_gather_kernel, dim3(blocks_for((size_t)n_dst * block_elems)), dim3(256), 0,
                       (hipStream_t)stream, src, idx_dev, dst, n_dst, block_elems);
    return mvae_launch_status();
}

MVAE_EXPORT int mvae_block_scatter_add(const float *src, const int *idx_dev, float *dst, int n_src, int n_dst,
                                       size_t block_elems, mvae_stream_t stream) {
    if (!src || !idx_dev || !dst || n_src <= 0 || n_dst <= 0 || block_elems == 0) return MVAE_ERR_ARG;
    hipLaunchKernelGGL(block_scatter_add_kernel, dim3(blocks_for((size_t)n_dst * block_elems)), dim3(256), 0,
                       (hipStream_t)stream, src, idx_dev, dst, n_src, n_dst, block_elems);
    return mvae_launch_status();
}

MVAE_EXPORT int mvae_scatter_sums(const float *vals, const float *coef_dev, const int *idx_dev, float *out,
                                  float *total_out, int n, int flags, mvae_stream_t stream) {
    if (!vals || !idx_dev || (!out && !total_out) || n <= 0) return MVAE_ERR_ARG;
    hipLaunchKernelGGL(scatter_sums_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, vals, coef_dev, idx_dev, out,
                       total_out, n, (flags & MVAE_ACCUMULATE) ? 1 : 0);
    return mvae_launch_status();
}
