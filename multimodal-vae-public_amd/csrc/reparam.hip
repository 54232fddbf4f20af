// reparam.hip -- stand-alone reparameterised draw for the reference's public
// MVAE.reparametrize(mu, logvar) (mnist/model.py:29-35).  The train step itself never
// launches these: poe.hip fuses the draw into the product-of-experts kernel.
#include "common.h"

namespace {
__global__ __launch_bounds__(256) void reparam_fwd_kernel(const float *mu, const float *logvar, const float *eps,
                                                          float *z, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        z[i] = eps[i] * expf(0.5f * logvar[i]) + mu[i];
}
__global__ __launch_bounds__(256) void reparam_bwd_kernel(const float *dz, const float *logvar, const float *eps,
                                                          float *dmu, float *dlogvar, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float g = dz[i];
        dmu[i] = g;
        dlogvar[i] = g * eps[i] * 0.5f * expf(0.5f * logvar[i]);
    }
}
inline int blocks_for(size_t n) {
    size_t b = (n + 255) / 256;
    return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}
}  // namespace

MVAE_EXPORT int mvae_reparam_fwd(const float *mu, const float *logvar, const float *eps, float *z, size_t n,
                                 mvae_stream_t stream) {
    if (!mu || !logvar || !eps || !z) return MVAE_ERR_ARG;
    if (n == 0) return MVAE_OK;
    hipLaunchKernelGGL(reparam_fwd_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, mu, logvar, eps,
                       z, n);
    return mvae_launch_status();
}

MVAE_EXPORT int mvae_reparam_bwd(const float *dz, const float *logvar, const float *eps, float *dmu,
                                 float *dlogvar, size_t n, mvae_stream_t stream) {
    if (!dz || !logvar || !eps || !dmu || !dlogvar) return MVAE_ERR_ARG;
    if (n == 0) return MVAE_OK;
    hipLaunchKernelGGL(reparam_bwd_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, dz, logvar, eps,
                       dmu, dlogvar, n);
    return mvae_launch_status();
}
