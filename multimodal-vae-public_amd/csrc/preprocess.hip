// Input pipeline on the GPU: the transform the reference's loaders run per image on the CPU --
// Compose([Resize(64), CenterCrop(64), ToTensor()]) for CelebA (celeba/train.py:146-148,
// celeba19/train.py:200-202) and ToTensor for MNIST / FashionMNIST (mnist/train.py:160,164) -- as one
// launch over a batch of raw uint8 images already resident in HBM.
//
// Resize is Pillow's 8-bit separable BILINEAR resample (torchvision delegates to it): a triangle
// filter whose support grows with the reduction factor, coefficients normalised in double and rounded
// to 22-bit fixed point ON THE HOST (mvae_resample_coeffs, the same formulas), integer multiply-
// accumulate with a uint8 intermediate between the horizontal and the vertical pass.  Byte-exact with
// Pillow (tests/golden/preprocess.npz); the final float is the IEEE quotient u8 / 255.0f, as ToTensor's.
//
// One 1024-thread block per image.  HBM-bound integer work: the source (116 KB for a 218x178 CelebA
// image) is read once, as whole dwords, 32 KiB of rows at a time into LDS; the filter taps read bytes
// from LDS; the horizontally resampled rows the crop needs stay in LDS too (<= 150 KiB in all); only
// the S x S x 3 floats go back, lanes along the output column.
#include <cmath>

#include "common.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;
constexpr size_t PP_MAX_LDS = 150 * 1024;

struct ResizeArgs {
    const uint8_t *src; float *dst;
    const int *kx, *bx, *ky, *by;     // coefficient tables [out, ksize] and (first tap, taps) [out, 2]
    int H, W, ksx, ksy, S, crop_top, crop_left, y0, y1;
};

__device__ __forceinline__ int clip8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

constexpr int PP_THREADS = 1024;
constexpr int PP_CHUNK_BYTES = 32 * 1024;      // source rows staged per round

__global__ __launch_bounds__(PP_THREADS) void resize_crop_kernel(ResizeArgs a) {
    // [y1 - y0][S][3] horizontally resampled rows, then a staging buffer for raw source rows
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int b = blockIdx.x, S = a.S, t = threadIdx.x;
    const uint8_t *img = a.src + (size_t)b * a.H * a.W * 3;
    const int rows = a.y1 - a.y0, row_bytes = a.W * 3;
    uint8_t *tmp = lds;
    uint8_t *chunk = lds + (((size_t)rows * S * 3 + 15) & ~(size_t)15);
    const int RC = max(1, PP_CHUNK_BYTES / row_bytes);
    for (int r0 = 0; r0 < rows; r0 += RC) {
        const int nr = min(RC, rows - r0);
        // stage source rows [y0 + r0, +nr): whole dwords from the 4-byte boundary below the first byte
        // (coalesced 256-byte wave loads instead of one byte load per filter tap), bytes for the tail
        const uint8_t *g = img + (size_t)(a.y0 + r0) * row_bytes;
        const int mis = (int)(reinterpret_cast<uintptr_t>(g) & 3);
        const int nbytes = nr * row_bytes + mis, ndw = nbytes >> 2;
        const uint32_t *g4 = reinterpret_cast<const uint32_t *>(g - mis);
        uint32_t *c4 = reinterpret_cast<uint32_t *>(chunk);
        for (int i = t; i < ndw; i += PP_THREADS) c4[i] = g4[i];
        for (int i = (ndw << 2) + t; i < nbytes; i += PP_THREADS) chunk[i] = (g - mis)[i];
        __syncthreads();
        // horizontal pass over these rows
        for (int idx = t; idx < nr * S * 3; idx += PP_THREADS) {
            const int c = idx % 3, j = (idx / 3) % S, r = idx / (3 * S);
            const int ox = a.crop_left + j;
            const int xmin = a.bx[2 * ox], xmax = a.bx[2 * ox + 1];
            const int *k = a.kx + (size_t)ox * a.ksx;
            const uint8_t *row = chunk + mis + (r * a.W + xmin) * 3 + c;
            int acc = 1 << (PRECISION_BITS - 1);
            for (int x = 0; x < xmax; ++x) acc += (int)row[3 * x] * k[x];
            tmp[(size_t)(r0 + r) * S * 3 + j * 3 + c] = (uint8_t)clip8(acc >> PRECISION_BITS);
        }
        __syncthreads();
    }
    // vertical pass + ToTensor: dst[b][c][i][j]
    float *out = a.dst + (size_t)b * 3 * S * S;
    for (int idx = t; idx < 3 * S * S; idx += PP_THREADS) {
        const int j = idx % S, i = (idx / S) % S, c = idx / (S * S);
        const int oy = a.crop_top + i;
        const int ymin = a.by[2 * oy], ymax = a.by[2 * oy + 1];
        const int *k = a.ky + (size_t)oy * a.ksy;
        const uint8_t *col = tmp + ((size_t)(ymin - a.y0) * S + j) * 3 + c;
        int acc = 1 << (PRECISION_BITS - 1);
        for (int y = 0; y < ymax; ++y) acc += (int)col[(size_t)y * S * 3] * k[y];
        out[idx] = (float)clip8(acc >> PRECISION_BITS) / 255.0f;
    }
}

__global__ __launch_bounds__(256) void u8_to_f32_kernel(const uint8_t *src, float *dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        dst[i] = (float)src[i] / 255.0f;
}

inline double triangle(double x) {
    if (x < 0.0) x = -x;
    return x < 1.0 ? 1.0 - x : 0.0;
}

}  // namespace

MVAE_EXPORT int mvae_resample_ksize(int in_size, int out_size) {
    if (in_size <= 0 || out_size <= 0) return MVAE_ERR_ARG;
    double filterscale = (double)in_size / out_size;
    if (filterscale < 1.0) filterscale = 1.0;
    return (int)ceil(filterscale) * 2 + 1;
}

// HOST function: fixed-point bilinear coefficients of one axis (Pillow Resample.c precompute_coeffs +
// normalize_coeffs_8bpc).  kk is [out_size, ksize] with ksize = mvae_resample_ksize, bounds [out_size, 2].
MVAE_EXPORT int mvae_resample_coeffs(int in_size, int out_size, int *kk, int *bounds) {
    if (in_size <= 0 || out_size <= 0 || !kk || !bounds) return MVAE_ERR_ARG;
    const double scale = (double)in_size / out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 1.0 * filterscale, ss = 1.0 / filterscale;
    const int ksize = (int)ceil(support) * 2 + 1;
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = (xx + 0.5) * scale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        double w[64], ww = 0.0;
        if (ksize > 64) return MVAE_ERR_ARG;          // reduction factors above ~31x are not a loader's job
        for (int x = 0; x < xmax; ++x) {
            w[x] = triangle((x + xmin - center + 0.5) * ss);
            ww += w[x];
        }
        int *k = kk + (size_t)xx * ksize;
        for (int x = 0; x < ksize; ++x) {
            double v = 0.0;
            if (x < xmax) v = (ww != 0.0) ? w[x] / ww : w[x];
            k[x] = v < 0 ? (int)(-0.5 + v * (1 << PRECISION_BITS)) : (int)(0.5 + v * (1 << PRECISION_BITS));
        }
        bounds[2 * xx] = xmin;
        bounds[2 * xx + 1] = xmax;
    }
    return ksize;
}

MVAE_EXPORT int mvae_resize_crop_u8_to_f32(const uint8_t *src, float *dst, int B, int H, int W, int out_h, int out_w,
                                           int S, int crop_top, int crop_left, const int *kx_dev, const int *bx_dev,
                                           int ksx, const int *ky_dev, const int *by_dev, int ksy, int y0, int y1,
                                           mvae_stream_t stream) {
    if (!src || !dst || !kx_dev || !bx_dev || !ky_dev || !by_dev || B <= 0 || H <= 0 || W <= 0 || S <= 0 ||
        crop_top < 0 || crop_left < 0 || crop_top + S > out_h || crop_left + S > out_w || y0 < 0 || y1 > H || y1 <= y0)
        return MVAE_ERR_ARG;
    const int rc = PP_CHUNK_BYTES / (W * 3) > 0 ? PP_CHUNK_BYTES / (W * 3) : 1;
    const size_t lds = (((size_t)(y1 - y0) * S * 3 + 15) & ~(size_t)15) + (size_t)rc * W * 3 + 16;
    if (lds > PP_MAX_LDS) return MVAE_ERR_ARG;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(resize_crop_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)PP_MAX_LDS);
        attr_done = true;
    }
    ResizeArgs a;
    a.src = src; a.dst = dst; a.kx = kx_dev; a.bx = bx_dev; a.ky = ky_dev; a.by = by_dev;
    a.H = H; a.W = W; a.ksx = ksx; a.ksy = ksy; a.S = S; a.crop_top = crop_top; a.crop_left = crop_left;
    a.y0 = y0; a.y1 = y1;
    hipLaunchKernelGGL(resize_crop_kernel, dim3(B), dim3(PP_THREADS), lds, (hipStream_t)stream, a);
    return mvae_launch_status();
}

MVAE_EXPORT int mvae_u8_to_f32(const uint8_t *src, float *dst, size_t n, mvae_stream_t stream) {
    if (!src || !dst) return MVAE_ERR_ARG;
    if (n == 0) return MVAE_OK;
    size_t blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(u8_to_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, dst, n);
    return mvae_launch_status();
}
