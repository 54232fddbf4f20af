/*
 * mvae_hip.h -- C ABI of libmvae_hip.so: the MVAE train-step kernels for MI355X (gfx950).
 *
 * The reference (mhw32/multimodal-vae-public) has no FFI: its hot path is a chain of
 * PyTorch ATen ops dispatched from model.py / train.py.  Each entry point below
 * replaces the ATen op(s) at the cited reference call sites (paths relative to the
 * reference repository root; SURVEY.md section 2.2 is the op inventory K1..K15).
 *
 * Conventions
 *   - every tensor is fp32, contiguous unless a leading dimension is given, NCHW for images;
 *     class labels are int64;
 *   - all pointers are DEVICE pointers borrowed for the duration of the enqueue; the
 *     library never allocates, frees or retains device memory; scratch is passed in
 *     (`ws`, `ws_bytes`; size it with the matching *_ws_bytes query);
 *   - `stream` is a hipStream_t passed as void*; kernels are asynchronous, no host sync;
 *   - return 0 on success, <0 on error (MVAE_ERR_*); no C++ exception crosses the ABI;
 *   - re-entrant: no global mutable state (launch plans are pure functions of the shapes).  The
 *     tuning overrides at the end of this header exist only in the separate -DMVAE_TUNING build
 *     (libmvae_hip_tuning.so, used by tools/gemm_bench.py); libmvae_hip.so does not export them;
 *   - the data-parallel gradient exchange (RCCL over xGMI) is part of the ABI: mvae_comm_* at the end of this
 *     header; RCCL is bound at run time, so the library has no link-time dependency on it.
 */
#ifndef MVAE_HIP_H
#define MVAE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MVAE_OK          0
#define MVAE_ERR_ARG    (-1)   /* bad shape / null pointer / unsupported stride */
#define MVAE_ERR_LAUNCH (-2)   /* hipLaunchKernel reported an error */
#define MVAE_ERR_WS     (-3)   /* workspace too small */
#define MVAE_ERR_COMM   (-4)   /* RCCL not loadable / an RCCL or HIP runtime call of mvae_comm_* failed (mvae_comm_last_error) */

/* flags */
#define MVAE_ACT_SWISH   1     /* fwd: also write swish(out); bwd: multiply by swish'(preact) */
#define MVAE_ACCUMULATE  2     /* add into the destination instead of overwriting it */

#define MVAE_POE_VARIANT_A 0   /* mnist/model.py:156-163, fashionmnist/model.py:175-182 */
#define MVAE_POE_VARIANT_B 1   /* celeba/model.py:200-207, celeba19/model.py:219-226 */
#define MVAE_POE_NO_PRIOR  4   /* OR-ed into `variant`: no built-in N(0,1) prior -- the caller's stack carries every expert,
                                  as when ProductOfExperts.forward is called on a [M,B,D] stack whose row 0 is what
                                  prior_expert returned (mnist/model.py:50-63,156-163,172-185).  REQUIREMENT: every
                                  term's mask must then select at least one of the E experts -- an empty product has
                                  no precision (0/0: NaN mu, inf logvar) and the launch cannot see a device mask; the
                                  Python binding checks host-built masks (kernels._check_no_prior_masks) */
#define MVAE_MAX_EXPERTS  32

typedef void *mvae_stream_t;   /* hipStream_t */

int mvae_abi_version(void);
/* bytes of scratch a GEMM-shaped call below may need for an output of rows_out x cols_out reduced
 * over reduce_len (split reductions, or the repacked weights of a transposed conv: pass
 * rows_out = Cin, cols_out = Cout*16 of the mirrored conv) */
size_t mvae_gemm_ws_bytes(int rows_out, int cols_out, int reduce_len);
/* ------------------------------------------------------------------------------------
 * K1  Linear (nn.Linear forward / backward): mnist/model.py:75-78,95-98,117-119,136-139;
 *     fashionmnist/model.py:84-86,107-109,135-137,155-161; celeba/model.py:89-92,114,
 *     148-154,175-184; celeba19/model.py:115-118,140,176-178,199-205.
 * K5  Swish (x*sigmoid(x), mnist/model.py:166-169) and K7 Dropout(0.1) (celeba/model.py:91)
 *     are fused into the epilogues.
 *
 * fwd : pre[M,N] = x[M,K] . w[N,K]^T + bias[N]         (pre may be NULL when act given)
 *       act[M,N] = swish(pre) * (mask ? mask*mask_scale : 1)   (if act != NULL)
 * dgrad: dx[M,K] (+)= (dy[M,N] . w[N,K]) * (mask ? mask*mask_scale : 1) * swish'(pre_in)
 *        (pre_in = pre-activation that produced this layer's INPUT, NULL for none)
 * wgrad: dw[N,K] (+)= dy^T . x ;  db[N] (+)= sum_m dy   (db may be NULL)
 * ws (mvae_gemm_ws_bytes): scratch for split reductions; NULL disables splitting.
 * ---------------------------------------------------------------------------------- */
int mvae_linear_fwd(const float *x, int ldx, const float *w, const float *bias,
                    float *pre, float *act, int ldy,
                    const float *mask, float mask_scale,
                    int M, int N, int K, void *ws, size_t ws_bytes, mvae_stream_t stream);
int mvae_linear_dgrad(const float *dy, int lddy, const float *w,
                      float *dx, int lddx,
                      const float *pre_in, const float *mask, float mask_scale,
                      int M, int N, int K, int flags, void *ws, size_t ws_bytes,
                      mvae_stream_t stream);
int mvae_linear_wgrad(const float *dy, int lddy, const float *x, int ldx,
                      float *dw, float *db, int M, int N, int K, int flags,
                      void *ws, size_t ws_bytes, mvae_stream_t stream);

/* Several Linear weight gradients of DIFFERENT shapes in one launch: dw_q[N_q][K_q] (+)= dy_q^T x_q and, where
 * db_q != NULL, db_q[N_q] (+)= column sums of dy_q -- the loss.backward() work of up to
 * MVAE_WGRAD_BATCH_MAX nn.Linear layers (mnist/model.py:75-78,95-104,137-145) that nothing but the optimizer
 * waits for.  Each item: N*K <= 2048 tiles of 32x32, M <= 4096, operands below 4 GiB; distinct dw / db per
 * item (MVAE_ERR_ARG otherwise -- use mvae_linear_wgrad for those).  Same arithmetic as mvae_linear_wgrad's
 * direct path: per output the batch rows are summed in a fixed order (deterministic). */
#define MVAE_WGRAD_BATCH_MAX 16
typedef struct {
    const float *dy; int lddy;      /* [M, N] upstream gradient, row stride lddy */
    const float *x; int ldx;        /* [M, K] layer input, row stride ldx */
    float *dw;                      /* [N, K] contiguous */
    float *db;                      /* [N] or NULL */
    int M, N, K;
    int flags;                      /* MVAE_ACCUMULATE: add to dw / db */
} mvae_wgrad_item;
int mvae_linear_wgrad_batched(const mvae_wgrad_item *items, int n_items, mvae_stream_t stream);

/* The same launch, which also runs optimizer.step() (torch.optim.Adam defaults, mnist/train.py:168,219) on the
 * parameters whose gradients it has just produced: the weight gradients are the last thing loss.backward() computes
 * for a layer and nothing but the optimizer reads them, so the update rides the gradient's epilogue instead of
 * waiting for one arena-wide launch behind the whole backward pass (15 us + a join at the very end of a 296-us
 * MNIST step).  The gradient is still written.  The four arenas are laid out alike: the parameter / moment element
 * of gradient address g is at the same offset from its base as g is from grad_base.  coef2: the step's two
 * bias-correction factors, left there by mvae_adam_prepare earlier on a stream this one is ordered behind.
 * MVAE_ACCUMULATE is refused (the update needs the step's whole gradient; the CALLER guarantees that no other
 * launch of the step adds to these gradients or reads these parameters afterwards).  An item with dy == x == NULL
 * is a finished gradient of K elements at dw (written earlier on this stream, e.g. an Embedding's): it only takes
 * the update.  Same arithmetic, bit for bit, as mvae_linear_wgrad_batched followed by mvae_adam_apply_at. */
typedef struct {
    const float *grad_base;         /* gradient arena */
    float *param_base;              /* parameter arena */
    float *exp_avg_base;            /* first / second moment arenas */
    float *exp_avg_sq_base;
    const float *coef2;             /* device: { lr / (1 - beta1^t), 1 / sqrt(1 - beta2^t) } (mvae_adam_prepare) */
    double beta1, beta2, eps;
    float grad_scale;               /* gradients are multiplied by this first (1 / world size; 1 here) */
} mvae_adam_fuse;
int mvae_linear_wgrad_batched_adam(const mvae_wgrad_item *items, int n_items, const mvae_adam_fuse *adam,
                                   mvae_stream_t stream);

/* Grouped forms: G independent Linear problems of ONE shape in one launch.  celeba19 builds 18
 * identical attribute encoders / decoders (celeba19/model.py:29-30, 173-196) and the reference runs
 * them one after another (celeba19/model.py:78-81, 53-54); here operand g of every array is at
 * base + g * <name>_gs floats (the expert's slice of the parameter arena / of a [G, rows, width]
 * activation buffer; any two same-shaped problems qualify: the stride is the pointer difference,
 * modulo 2^64).  No dropout mask.  ws (NULL = never split): G * mvae_gemm_ws_bytes of the single
 * problem is always enough.  Same maths as the single forms. */
int mvae_linear_fwd_grouped(const float *x, int ldx, size_t x_gs, const float *w, size_t w_gs,
                            const float *bias, size_t bias_gs, float *pre, float *act, int ldy,
                            size_t y_gs, int G, int M, int N, int K, void *ws, size_t ws_bytes,
                            mvae_stream_t stream);
int mvae_linear_dgrad_grouped(const float *dy, int lddy, size_t dy_gs, const float *w, size_t w_gs,
                              float *dx, int lddx, size_t dx_gs, const float *pre_in, size_t pre_gs,
                              int G, int M, int N, int K, int flags, void *ws, size_t ws_bytes,
                              mvae_stream_t stream);
int mvae_linear_wgrad_grouped(const float *dy, int lddy, size_t dy_gs, const float *x, int ldx,
                              size_t x_gs, float *dw, size_t dw_gs, float *db, size_t db_gs, int G,
                              int M, int N, int K, int flags, void *ws, size_t ws_bytes,
                              mvae_stream_t stream);

/* ------------------------------------------------------------------------------------
 * K2  Conv2d 4x4, bias=False, (stride,pad) in {(2,1),(1,0)}: fashionmnist/model.py:79,81;
 *     celeba/model.py:77,79,82,85; celeba19/model.py:103,105,108,111.
 *     Implicit im2col GEMM on fp32 MFMA; x[B,Cin,H,W], w[Cout,Cin,4,4], y[B,Cout,OH,OW].
 *     fwd  : pre = conv(x,w) ; act = swish(pre) (either may be NULL)
 *     dgrad: dx (+)= conv^T(dy,w) * swish'(pre_in)   (pre_in NULL for none)
 *     wgrad: dw (+)= correlate(x, dy)
 * K3  ConvTranspose2d 4x4, bias=False: fashionmnist/model.py:112,114; celeba/model.py:117,
 *     120,123,126; celeba19/model.py:143,146,149,152.  w[Cin,Cout,4,4].  Forward of the
 *     transpose is the dgrad of the mirrored conv and vice versa.
 * ---------------------------------------------------------------------------------- */
int mvae_conv2d_k4_fwd(const float *x, const float *w, float *pre, float *act,
                       int B, int Cin, int H, int W, int Cout, int stride, int pad,
                       mvae_stream_t stream);
int mvae_conv2d_k4_dgrad(const float *dy, const float *w, float *dx, const float *pre_in,
                         int B, int Cin, int H, int W, int Cout, int stride, int pad,
                         void *ws, size_t ws_bytes, mvae_stream_t stream);
int mvae_conv2d_k4_wgrad(const float *dy, const float *x, float *dw,
                         int B, int Cin, int H, int W, int Cout, int stride, int pad,
                         int flags, void *ws, size_t ws_bytes, mvae_stream_t stream);
int mvae_convT2d_k4_fwd(const float *x, const float *w, float *pre, float *act,
                        int B, int Cin, int H, int W, int Cout, int stride, int pad,
                        void *ws, size_t ws_bytes, mvae_stream_t stream);
int mvae_convT2d_k4_dgrad(const float *dy, const float *w, float *dx, const float *pre_in,
                          int B, int Cin, int H, int W, int Cout, int stride, int pad,
                          mvae_stream_t stream);
int mvae_convT2d_k4_wgrad(const float *dy, const float *x, float *dw,
                          int B, int Cin, int H, int W, int Cout, int stride, int pad,
                          int flags, void *ws, size_t ws_bytes, mvae_stream_t stream);

/* The two dgrad-form launches -- mvae_conv2d_k4_dgrad and mvae_convT2d_k4_fwd -- read the weights from a
 * repacked copy (parity-class major, input channel contiguous) in `ws`.  With `w` given they make that copy
 * themselves, a small launch in front of every call.  A caller that runs many steps can make the copies of all
 * its layers in ONE launch per step instead (the weights only change in the optimizer) and pass w = NULL with
 * ws = the layer's copy:
 *   mvae_conv_k4_repack_floats   floats of the copy this launch would read, or 0 if it reads `w` directly (the
 *                                <= 4-channel and stride-1 5x5 shapes have their own kernels) -- then w = NULL is
 *                                MVAE_ERR_ARG.  transposed: 0 = Conv2d data gradient, 1 = ConvTranspose2d forward;
 *                                B, Cin, H, W, Cout, stride, pad exactly as in that call.
 *   mvae_conv_k4_repack_batched  up to 16 copies in one launch (Cin, Cout: the module's own). */
typedef struct {
    const float *w; float *wr;
    int transposed, Cin, Cout, stride, pad;
} mvae_repack_item;
size_t mvae_conv_k4_repack_floats(int transposed, const float *w, int B, int Cin, int H, int W, int Cout,
                                  int stride, int pad);
int mvae_conv_k4_repack_batched(const mvae_repack_item *items, int n_items, mvae_stream_t stream);

/* Statistics-only ConvTranspose2d forward: the transposed conv in FRONT OF THE LAST BatchNorm of a decoder pass whose
 * output the reference never reads -- celeba19/train.py:278-283 runs the image decoder for the 18 attribute-only terms
 * too (celeba19/model.py:52-61), celeba/train.py:195 for the attribute-only term; the only effect is the BatchNorm
 * running-statistics update (reproduced by default).  The launch computes the layer but STORES
 * NOTHING: each block leaves one (mean, M2) record per output channel over MVAE_STATS_TILE_ELEMS elements,
 *     part[tile][Cout][2],   tiles of one image -- hence of one batch group -- contiguous,
 * and mvae_bn_stats_merge turns the records of each group into that BatchNorm's saved + running statistics exactly
 * as mvae_bn_train_fwd(y = NULL) would from the activations (equal-count merge: mean = avg(mean_i),
 * M2 = sum(M2_i) + n * sum((mean_i - mean)^2); groups in order, n_updates times each).
 *   mvae_convT2d_k4_stats_tiles  number of records the launch writes, or 0 if the shape is not covered (covered:
 *                                stride 2, pad 1, <= 32 output channels, B*H*W a multiple of 128) -- the caller then
 *                                uses mvae_convT2d_k4_fwd + mvae_bn_train_fwd(y = NULL).
 *   w / ws                       as in mvae_convT2d_k4_fwd (w = NULL: ws holds the repacked copy). */
#define MVAE_STATS_TILE_ELEMS 512
size_t mvae_convT2d_k4_stats_tiles(int B, int Cin, int H, int W, int Cout, int stride, int pad);
int mvae_convT2d_k4_fwd_stats(const float *x, const float *w, float *part, size_t part_floats,
                              int B, int Cin, int H, int W, int Cout, int stride, int pad,
                              void *ws, size_t ws_bytes, mvae_stream_t stream);

/* ------------------------------------------------------------------------------------
 * K4  BatchNorm2d / BatchNorm1d (training mode, eps 1e-5, momentum 0.1) + fused Swish:
 *     celeba/model.py:80,83,86,118,121,124,149,152,176,179,182; celeba19/model.py:106,109,
 *     112,144,147,150.  x is [G*B, C, HW] (HW = 1 for BatchNorm1d): G independent groups
 *     of B samples, each normalised with its own batch statistics -- G > 1 lets one launch
 *     serve the G model() calls of a train step that the reference issues separately.
 *     Running statistics are updated sequentially for g = 0..G-1, each `n_updates` times
 *     (unbiased variance for the running estimate, biased for normalisation).
 *     save_mean / save_invstd are [G, C].  ws: mvae_bn_ws_bytes(G, C, B*HW).
 *     y == NULL: statistics only (saved + running), for decoder passes the reference runs but
 *     whose output it never reads (celeba19/train.py:277-283 -- SURVEY Appendix B-4).
 *     eval: y = swish?((x - running_mean) / sqrt(running_var + eps) * gamma + beta).
 * ---------------------------------------------------------------------------------- */
size_t mvae_bn_ws_bytes(int G, int C, int n_per_group);
int mvae_bn_train_fwd(const float *x, const float *gamma, const float *beta, float *y,
                      float *save_mean, float *save_invstd,
                      float *running_mean, float *running_var,
                      int G, int B, int C, int HW, float eps, float momentum,
                      int n_updates, const int *n_updates_dev /* nullable: overrides n_updates */,
                      int flags, void *ws, size_t ws_bytes, mvae_stream_t stream);
int mvae_bn_train_bwd(const float *dy, const float *x, const float *gamma, const float *beta,
                      const float *save_mean, const float *save_invstd,
                      float *dx, float *dgamma, float *dbeta,
                      int G, int B, int C, int HW, int flags,
                      void *ws, size_t ws_bytes, mvae_stream_t stream);
/* saved + running statistics of G groups from the records of mvae_convT2d_k4_fwd_stats (`tiles` records of
 * `elems_per_tile` elements per channel; tiles % G == 0, a group's tiles contiguous).  save_* may be NULL. */
int mvae_bn_stats_merge(const float *part, int tiles, int elems_per_tile, int G, int C,
                        float *save_mean, float *save_invstd, float *running_mean, float *running_var,
                        float eps, float momentum, int n_updates, const int *n_updates_dev, mvae_stream_t stream);
int mvae_bn_eval_fwd(const float *x, const float *gamma, const float *beta, float *y,
                     const float *running_mean, const float *running_var,
                     int N, int C, int HW, float eps, int flags, mvae_stream_t stream);

/* ------------------------------------------------------------------------------------
 * K5  stand-alone Swish (only where no producer epilogue can host it).
 * K6  Embedding lookup + Swish and its scatter-add backward: mnist/model.py:116,123;
 *     fashionmnist/model.py:133; celeba19/model.py:174,183.
 *     fwd: act[r,:] = swish(w[idx[r],:]) ; bwd: dw[c,:] (+)= sum_{r: idx[r]==c} dact[r,:]*swish'(w[c,:])
 *     idx is int64 (labels) or, with idx_is_float, the {0,1} floats celeba19 casts by .long().
 * ---------------------------------------------------------------------------------- */
int mvae_swish_fwd(const float *x, float *y, size_t n, mvae_stream_t stream);
int mvae_swish_bwd(const float *dy, const float *x, float *dx, size_t n, mvae_stream_t stream);
/* sample.py (mnist/sample.py:100-112, celeba/sample.py): std = exp(logvar/2) comes from mvae_reparam_fwd with
 * mu = 0, eps = 1; z = eps * std + mu for n_samples draws of ONE posterior row is the periodic affine map
 * below (period = n_latents; 1 for the prior N(0,1)); F.sigmoid on the image decoder's logits. */
int mvae_sigmoid_fwd(const float *x, float *y, size_t n, mvae_stream_t stream);
int mvae_affine_fwd(const float *x, const float *scale, const float *shift, float *y, size_t n, size_t period,
                    mvae_stream_t stream);
int mvae_embedding_swish_fwd(const void *idx, int idx_is_float, const float *w, float *act,
                             int R, int n_classes, int width, mvae_stream_t stream);
int mvae_embedding_swish_bwd(const void *idx, int idx_is_float, const float *w,
                             const float *dact, float *dw,
                             int R, int n_classes, int width, int flags, mvae_stream_t stream);
/* grouped: table / output / index g at base + g * <name>_gs elements (celeba19: idx = attrs[B,18],
 * idx_is_float = 18 (row stride), idx_gs = 1 (column g)); dw shares w's group stride. */
int mvae_embedding_swish_fwd_grouped(const void *idx, int idx_is_float, size_t idx_gs, const float *w,
                                     size_t w_gs, float *act, size_t act_gs, int G, int R,
                                     int n_classes, int width, mvae_stream_t stream);
int mvae_embedding_swish_bwd_grouped(const void *idx, int idx_is_float, size_t idx_gs, const float *w,
                                     size_t w_gs, const float *dact, size_t dact_gs, float *dw, int G,
                                     int R, int n_classes, int width, int flags, mvae_stream_t stream);

/* ------------------------------------------------------------------------------------
 * K8-K10, K12  prior expert + product of experts + reparameterise + analytic KL, fused:
 *     mnist/model.py:29-35,46-64,156-163,172-185; celeba/model.py:200-207;
 *     celeba19/model.py:63-89,219-226; KL: mnist/train.py:56, celeba/train.py:62,
 *     celeba19/train.py:58.
 *     E experts (the N(0,1) prior is implicit and always present) given as mu_e/logvar_e
 *     pointers with a common row stride `ld`; T terms, term t fuses the experts whose bit
 *     is set in masks[t] (DEVICE uint32[T], so a captured graph can be replayed with new
 *     subsets).  noise is [T,B,D] or NULL (eval: z = mu).
 *     fwd outputs mu/logvar/z [T,B,D] and kl[T,B] = -0.5*sum_d(1+lv-mu^2-exp(lv)).
 *     bwd inputs: dz, dmu, dlogvar [T,B,D] and dkl [T,B] = d loss / d kl (any may be NULL);
 *     outputs the per-expert gradients, summed over the terms that contain the expert.
 * ---------------------------------------------------------------------------------- */
typedef struct {
    const float *mu[MVAE_MAX_EXPERTS];
    const float *logvar[MVAE_MAX_EXPERTS];
} mvae_experts_t;
typedef struct {
    float *dmu[MVAE_MAX_EXPERTS];
    float *dlogvar[MVAE_MAX_EXPERTS];
} mvae_expert_grads_t;

int mvae_poe_fwd(const mvae_experts_t *experts, int ld, int E,
                 const uint32_t *masks_dev, int T,
                 const float *noise, float *mu, float *logvar, float *z, float *kl,
                 int B, int D, int variant, mvae_stream_t stream);
/* mvae_poe_fwd that draws its own reparameterisation noise (mnist/model.py:32: eps = std.data.new(size).normal_()):
 * eps[T,B,D] = exactly the standard normals mvae_philox_fill(seed, *counter_dev + counter_offset) would write,
 * generated inside the launch and stored to noise_out for the backward.  The counter is not advanced. */
int mvae_poe_fwd_draw(const mvae_experts_t *experts, int ld, int E,
                      const uint32_t *masks_dev, int T,
                      float *noise_out, uint64_t seed, const uint64_t *counter_dev, uint64_t counter_offset,
                      float *mu, float *logvar, float *z, float *kl,
                      int B, int D, int variant, mvae_stream_t stream);
int mvae_poe_bwd(const mvae_experts_t *experts, int ld, int E,
                 const uint32_t *masks_dev, int T,
                 const float *noise, const float *mu, const float *logvar,
                 const float *dz, const float *dmu, const float *dlogvar,
                 const float *dkl, int dkl_per_term /* dkl is [T] (beta/B per term) instead of [T,B] */,
                 const mvae_expert_grads_t *grads, int ldg,
                 int B, int D, int variant, mvae_stream_t stream);

/* mvae_poe_bwd with the latent gradient in TWO buffers: dz of term t = dz_a[slot_a[t]] + dz_b[slot_b[t]], each
 * buffer [n_slots, B, D] holding only the terms its decoder saw (slot -1: the term is not in that buffer;
 * slot_* are HOST arrays of T ints).  The fused step's two decoders run on two streams; with one buffer each,
 * neither their first layers' data gradients nor a cleared shared dz sit on the joined chain.  The sum is taken
 * a-then-b: same bits as accumulating b onto a. */
int mvae_poe_bwd_split(const mvae_experts_t *experts, int ld, int E,
                       const uint32_t *masks_dev, int T,
                       const float *noise, const float *mu, const float *logvar,
                       const float *dz_a, const int *slot_a, const float *dz_b, const int *slot_b,
                       const float *dkl, int dkl_per_term,
                       const mvae_expert_grads_t *grads, int ldg,
                       int B, int D, int variant, mvae_stream_t stream);

/* stand-alone reparameterised draw for the public MVAE.reparametrize (mnist/model.py:29-35):
 * z = eps * exp(0.5 * logvar) + mu */
int mvae_reparam_fwd(const float *mu, const float *logvar, const float *eps, float *z, size_t n,
                     mvae_stream_t stream);
int mvae_reparam_bwd(const float *dz, const float *logvar, const float *eps,
                     float *dmu, float *dlogvar, size_t n, mvae_stream_t stream);

/* stand-alone KL rows for the reference-surface elbo_loss(mu, logvar) (mnist/train.py:56) */
int mvae_kl_rows_fwd(const float *mu, const float *logvar, float *kl, int B, int D,
                     mvae_stream_t stream);
int mvae_kl_rows_bwd(const float *mu, const float *logvar, const float *dkl,
                     float *dmu, float *dlogvar, int B, int D, mvae_stream_t stream);

/* ------------------------------------------------------------------------------------
 * K11 BCE-with-logits row sums: mnist/train.py:47-49,62-74; celeba/train.py:50-58,68-80;
 *     celeba19/train.py:52-57,63-75.  logits/target [R,P];
 *     rowsum[r] = sum_p colw[(r / rows_per_group), p] * bce(logits[r,p], target[r % target_rows, p])
 *     (colw NULL -> 1).  target row of logits row r is (r / target_div) % target_rows, its element p
 *     lives at row * target_row_stride + p * target_col_stride (contiguous: div 1, strides P and 1;
 *     celeba19 reads column i of attrs[B,18] for decoder i with strides 1 and 18).
 *     bwd: dlogits[r,p] = drow[r / rows_per_group] * colw * dbce/dx  (drow is a DEVICE array;
 *     d loss / d rowsum is the constant lambda/B, so the fwd entry can emit it in the same pass).
 * K13 categorical CE: mnist/train.py:52,77-94.  row[r] = -log_softmax(logits[r,:]+1e-6)[label[r % label_rows]]
 * K14 ELBO combine: total = sum_r coef[r / rows_per_group] * rows[r]  (mnist/train.py:57-58,214)
 * ---------------------------------------------------------------------------------- */
int mvae_bce_rowsum_fwd(const float *logits, const float *target, const float *colw,
                        float *rowsum, const float *drow_dev, float *dlogits /* nullable: fused bwd */,
                        int R, int P, int rows_per_group, int target_rows,
                        int target_div, int target_row_stride, int target_col_stride,
                        mvae_stream_t stream);
int mvae_bce_rowsum_bwd(const float *logits, const float *target, const float *colw,
                        const float *drow_dev, float *dlogits,
                        int R, int P, int rows_per_group, int target_rows,
                        int target_div, int target_row_stride, int target_col_stride,
                        mvae_stream_t stream);
int mvae_ce_fwd(const float *logits, const int64_t *label, float *row,
                const float *drow_dev, float *dlogits /* nullable: fused bwd */,
                int R, int K, int rows_per_group, int label_rows, mvae_stream_t stream);
int mvae_ce_bwd(const float *logits, const int64_t *label, const float *drow_dev,
                float *dlogits, int R, int K, int rows_per_group, int label_rows,
                mvae_stream_t stream);
/* K1 + K12 / K13 in one launch: a decoder's LAST Linear whose only consumer is its reconstruction term
 * (mnist/model.py:104 -> mnist/train.py:47-49,62-74; mnist/model.py:146 -> mnist/train.py:52,77-94).  The logits
 * stay in the GEMM's accumulators: the epilogue writes d loss / d logits (what the backward consumes) and the loss.
 *   bce: dlogits[m][n] = drow[m / rows_per_group] * dBCE(logit, target[(m % target_rows) * target_row_stride + n]),
 *        partial[m][n / 32] = sum of the 32 columns' terms  (partial is [M, ceil(N / 32)] floats; a row's term is
 *        the sum of its partials -- mvae_elbo_reduce with rows_per_group = B * ceil(N / 32) adds them up);
 *   ce : N <= 32 classes; row[m] and dlogits as mvae_ce_fwd defines them, label[m % label_rows].
 * `logits` (nullable): also store the logits (tests).  The reduction is never split. */
int mvae_linear_bce_fwd(const float *x, int ldx, const float *w, const float *bias, const float *target,
                        int target_rows, int target_row_stride, const float *drow_dev, int rows_per_group,
                        float *dlogits, int ldy, float *logits, float *partial, int M, int N, int K,
                        mvae_stream_t stream);
int mvae_linear_ce_fwd(const float *x, int ldx, const float *w, const float *bias, const int64_t *label,
                       int label_rows, const float *drow_dev, int rows_per_group, float *dlogits, int ldy,
                       float *logits, float *row, int M, int N, int K, mvae_stream_t stream);
/* The ELBO of a whole fused step in one launch (mnist/train.py:57-58 batch mean per term, :214 sum of terms;
 * celeba19/train.py:59,265-302).  Each part is a vector of loss rows in `groups` groups of `rows_per_group`;
 * group g feeds term `term_of[g]` (device table) or `first_term + g`:
 *     elbo[t] = sum over parts and groups of term t of  coef[g] * sum(rows of g),   elbo[T] = sum over everything,
 * accumulated part by part, group by group (fixed order).  Optionally clears `zero[0..zero_n)` (the latent
 * gradient the decoders' first layers accumulate into) and advances a Philox launch counter by `counter_inc`
 * -- the step's bookkeeping that would otherwise be 5-6 single-purpose launches.  At most 1024 groups over all
 * parts; a part of more than one row per group has at most MVAE_ELBO_MAX_TERMS groups. */
#define MVAE_ELBO_MAX_PARTS 4
#define MVAE_ELBO_MAX_TERMS 40
typedef struct {
    const float *rows;      /* [groups * rows_per_group] */
    const float *coef;      /* per group, device; NULL = 1 */
    const int *term_of;     /* per group, device; NULL = first_term + group */
    int first_term, groups, rows_per_group;
} mvae_elbo_part;
int mvae_elbo_reduce(const mvae_elbo_part *parts, int n_parts, float *elbo, int T, float *zero, size_t zero_n,
                     uint64_t *counter_dev, uint64_t counter_inc, mvae_stream_t stream);
/* out[g] (+)= coef[g] * sum_{r in group g} rows[r] for g < G; *total_out (+)= sum_g of those
 * (either destination may be NULL) */
int mvae_group_sums(const float *rows, const float *coef_dev, float *out, float *total_out,
                    int G, int rows_per_group, int flags, mvae_stream_t stream);

/* ------------------------------------------------------------------------------------
 * K10 noise: counter-based Philox4x32-10 on device (perf mode; parity mode passes the
 *     host-drawn noise of the reference's CPU generator instead).  `counter_dev` is a
 *     device uint64 advanced by the kernel itself, so a captured hipGraph replays fresh noise.
 * K15 Adam (torch.optim.Adam defaults, mnist/train.py:168,219), one launch over the flat
 *     parameter arena; `step_dev` is a device int64 incremented by the kernel.
 * ---------------------------------------------------------------------------------- */
int mvae_randn(float *out, size_t n, uint64_t seed, uint64_t *counter_dev, mvae_stream_t stream);
int mvae_bernoulli(float *out, size_t n, float keep_prob, uint64_t seed, uint64_t *counter_dev,
                   mvae_stream_t stream);
/* the same draws at launch index *counter_dev + counter_offset WITHOUT advancing the counter (a fused step
 * advances it once, in mvae_elbo_reduce) */
int mvae_philox_fill(float *out, size_t n, int bernoulli, float keep_prob, uint64_t seed,
                     const uint64_t *counter_dev, uint64_t counter_offset, mvae_stream_t stream);
int mvae_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq,
                   size_t n, double lr, double beta1, double beta2, double eps, float grad_scale,
                   int64_t *step_dev, mvae_stream_t stream);
/* Adam over a range of the arena WITHOUT advancing `step_dev` (t = *step_dev + 1): data-parallel replicas
 * update each gradient bucket as its all-reduce lands and advance the counter once per step with
 * mvae_counter_add (mnist/train.py:219 is ONE optimizer.step(); the ranges partition it). */
int mvae_adam_apply(float *param, const float *grad, float *exp_avg, float *exp_avg_sq,
                    size_t n, double lr, double beta1, double beta2, double eps, float grad_scale,
                    const int64_t *step_dev, mvae_stream_t stream);
/* mvae_adam_apply at step t = *step_dev + step_add (step_add = 0: the caller advanced the counter earlier in
 * the step, off the critical chain, and needs no counter launch behind the update) */
int mvae_adam_apply_at(float *param, const float *grad, float *exp_avg, float *exp_avg_sq,
                       size_t n, double lr, double beta1, double beta2, double eps, float grad_scale,
                       const int64_t *step_dev, int64_t step_add, mvae_stream_t stream);
/* *step_dev += delta, then coef2[0..1] = { lr / (1 - beta1^t), 1 / sqrt(1 - beta2^t) } at t = the new counter value:
 * the counter launch of a step whose weight-gradient launches apply Adam themselves
 * (mvae_linear_wgrad_batched_adam); a later mvae_adam_apply_at(..., step_add = 0) on the rest of the arena computes
 * the same two factors from the counter. */
int mvae_adam_prepare(int64_t *step_dev, int64_t delta, double lr, double beta1, double beta2, float *coef2,
                      mvae_stream_t stream);
int mvae_counter_add(int64_t *counter_dev, int64_t delta, mvae_stream_t stream);
/* mvae_adam_apply_at(..., step_add = 0) with the step's two bias-correction factors READ from coef2 instead of recomputed
 * from the counter (ABI 6): the fused step's counter launch is mvae_adam_prepare -- early in the step, off the critical
 * chain -- and the update at the end of the chain (mnist/train.py:219) starts streaming at once: the two double-precision
 * pow() behind the factors were ~300 fp64 instructions in front of every wavefront of the 72-MB MNIST update.  Same
 * arithmetic as mvae_adam_apply_at, bit for bit. */
int mvae_adam_apply_coef(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, size_t n,
                         const float *coef2, double beta1, double beta2, double eps, float grad_scale,
                         mvae_stream_t stream);

/* Measurement aid (no reference counterpart): an EMPTY kernel on `stream`.  A profiling host launches one in front of
 * every call of a single-stream step; in the rocprofv3 kernel trace the k-th marker dispatch then separates the kernels
 * of call k - 1 from those of call k (tools/step_by_shape.py -> profiles/r04_*_by_shape.txt, the per-(call, shape)
 * table bench.py's roofline.rocprof_avg_us is read from).  `tag` is not interpreted. */
int mvae_trace_marker(int tag, mvae_stream_t stream);
int mvae_fill(float *out, size_t n, float value, mvae_stream_t stream);
/* One launch that puts a step's inputs where a captured graph reads them -- what `image.cuda()`, `text.cuda()`
 * (mnist/train.py:188-190) and the per-step scalars (annealing factor :180-186) amount to for a replayed step:
 * image_dst <- image_src (device, 16-byte aligned, a multiple of 4 floats), label_dst <- label_src (device, a multiple
 * of 4 bytes), table_dst <- table_src_host: PINNED HOST memory (hipHostMalloc / torch pin_memory) read by the kernel
 * itself.  Any of the three may be empty (size 0). */
int mvae_ingest(const float *image_src, float *image_dst, size_t image_floats,
                const void *label_src, void *label_dst, size_t label_bytes,
                const void *table_src_host, void *table_dst, size_t table_bytes, mvae_stream_t stream);

/* CelebA-19 term plumbing (celeba19/train.py:264-302, celeba19/model.py:56-60): gather the z blocks
 * a decoder needs, scatter-add the gradients back per term, and sum ELBO pieces by term table.
 *   block_gather      : dst[j] = src[idx[j]]                     (blocks of block_elems floats)
 *   block_scatter_add : dst[t] += sum_{j: idx[j]==t} src[j]      (j ascending: deterministic)
 *   scatter_sums      : out[idx[j]] += coef[j]*vals[j]; *total (+)= sum_j coef[j]*vals[j]
 * idx tables are DEVICE int32 so a captured graph follows each step's sampled subsets. */
int mvae_block_gather(const float *src, const int *idx_dev, float *dst, int n_dst, size_t block_elems,
                      mvae_stream_t stream);
int mvae_block_scatter_add(const float *src, const int *idx_dev, float *dst, int n_src, int n_dst,
                           size_t block_elems, mvae_stream_t stream);
int mvae_scatter_sums(const float *vals, const float *coef_dev, const int *idx_dev, float *out,
                      float *total_out, int n, int flags, mvae_stream_t stream);

/* K7 Dropout fan-out (celeba/model.py:89-92 runs twice per step on the same batch; only the
 *    Bernoulli draw differs): out[g,b,:] = h[b,:] * masks[g,b,:] * scale, and its backward
 *    dh[b,:] = sum_g dout[g,b,:] * masks[g,b,:] * scale. */
int mvae_dropout_fanout_fwd(const float *h, const float *masks, float *out, float scale,
                            int G, int B, int N, mvae_stream_t stream);
int mvae_dropout_fanin_bwd(const float *dout, const float *masks, float *dh, float scale,
                           int G, int B, int N, mvae_stream_t stream);
/* elementwise BCE-with-logits, the reference helper's own shape (mnist/train.py:62-74) */
int mvae_bce_elem_fwd(const float *logits, const float *target, float *out, size_t n,
                      mvae_stream_t stream);
int mvae_bce_elem_bwd(const float *logits, const float *target, const float *g,
                      float *dlogits, float *dtarget, size_t n, mvae_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Input pipeline: what the reference's DataLoader workers do per image on
 * the CPU, on a batch of raw uint8 images resident in HBM.
 *   mvae_u8_to_f32            : ToTensor (mnist/train.py:160,164): dst = (float)src / 255
 *   mvae_resize_crop_u8_to_f32: Compose([Resize(S), CenterCrop(S), ToTensor()]) (celeba/train.py:
 *       146-148, celeba19/train.py:200-202) on src[B,H,W,3] -> dst[B,3,S,S].  Resize is Pillow's
 *       8-bit separable BILINEAR resample (torchvision delegates to it), byte-exact: the caller
 *       builds the fixed-point coefficient tables of both axes once per image size with the HOST
 *       function mvae_resample_coeffs (kk[out, ksize], bounds[out, 2] = first tap, taps; ksize =
 *       mvae_resample_ksize) and uploads them; out_h/out_w = the resized size, (crop_top,
 *       crop_left) the crop origin in it, [y0, y1) the source rows the cropped rows depend on
 *       (min/max over their bounds).  One block per image, (y1-y0)*S*3 bytes of LDS <= 150 KiB.
 * ---------------------------------------------------------------------------------- */
int mvae_resample_ksize(int in_size, int out_size);
int mvae_resample_coeffs(int in_size, int out_size, int *kk_host, int *bounds_host);
int mvae_resize_crop_u8_to_f32(const uint8_t *src, float *dst, int B, int H, int W, int out_h,
                               int out_w, int S, int crop_top, int crop_left, const int *kx_dev,
                               const int *bx_dev, int ksx, const int *ky_dev, const int *by_dev,
                               int ksy, int y0, int y1, mvae_stream_t stream);
int mvae_u8_to_f32(const uint8_t *src, float *dst, size_t n, mvae_stream_t stream);

/* ------------------------------------------------------------------------------------
 * K16  The recurrent text stacks of the MultiMNIST MVAE: multimnist/model.py:145-235 --
 *      TextEncoder: nn.Embedding(12, 200) -> nn.GRU(200, 200, 1, bidirectional) -> x[-1], directions summed ->
 *      nn.Linear(200, 2D) (:162-179); TextDecoder: nn.Linear(D, 200) -> 4 steps of
 *      swish(nn.Embedding) | z -> nn.GRU(200 + D, 200, 2, dropout=0.1) -> | z -> nn.Linear(200 + D, 12), greedy
 *      arg-max feedback (:196-228).  The matrix products are mvae_linear_* launches with leading dimensions (the
 *      torch.cat's of :222,226 are column ranges of one buffer); these are the remaining element kernels.
 *   gru_cell_fwd  torch's gate order r | z | n in gi = x.W_ih^T + b_ih and gh = h.W_hh^T + b_hh ([B, 3H] each):
 *                 r = s(gi_r + gh_r), z = s(gi_z + gh_z), n = tanh(gi_n + r * gh_n), h' = (1 - z) n + z h;
 *                 gates [B, 4H] (r | z | n | gh_n, nullable) is what the backward needs.
 *   gru_cell_bwd  dh' = dh_new (+ dh_extra, nullable) -> dgi, dgh [B, 3H] contiguous and dh_prev [B, H] = dh' * z
 *                 (the caller adds dgh.W_hh with an accumulating mvae_linear_dgrad).
 *   embedding_fwd / _bwd   nn.Embedding with an index stride (column t of text[B, 4]: stride 4) and an output
 *                 leading dimension; flags: MVAE_ACT_SWISH (swish(embed(c)), :220), bwd also MVAE_ACCUMULATE.
 *                 The backward adds the rows of a class in row order: deterministic.
 *   copy2d        dst[r, :cols] (+)= src[r, :cols] (* mask[r, :cols] * scale): column ranges of the cat buffers, the
 *                 inter-layer Dropout of nn.GRU (mask in {0,1}, scale 1 / 0.9) and its backward, the direction sum.
 *   argmax_rows   first maximum per row (torch.max(F.log_softmax(c_out, dim=1), dim=1)[1], :211).
 * ------------------------------------------------------------------------------------ */
int mvae_gru_cell_fwd(const float *gi, int ldgi, const float *gh, int ldgh, const float *h_prev, int ldh,
                      float *h_new, int ldo, float *gates /* nullable */, int B, int H, mvae_stream_t stream);
int mvae_gru_cell_bwd(const float *dh_new, int lddh, const float *dh_extra /* nullable */, int ldde,
                      const float *gates, const float *h_prev, int ldh, float *dgi, float *dgh, float *dh_prev,
                      int B, int H, mvae_stream_t stream);
int mvae_embedding_fwd(const int64_t *idx, int idx_stride, const float *w, float *out, int ldo, int R,
                       int n_classes, int width, int flags, mvae_stream_t stream);
int mvae_embedding_bwd(const int64_t *idx, int idx_stride, const float *w, const float *dout, int ldd, float *dw,
                       int R, int n_classes, int width, int flags, mvae_stream_t stream);
int mvae_copy2d(const float *src, int lds, float *dst, int ldd, const float *mask /* nullable */, int ldm,
                float scale, int rows, int cols, int flags, mvae_stream_t stream);
int mvae_argmax_rows(const float *x, int ldx, int64_t *out, int R, int K, mvae_stream_t stream);

/* ------------------------------------------------------------------------------------
 * C1  Gradient exchange of data-parallel replicas -- RCCL over xGMI.
 *     The reference has NO counterpart: it is single-process, single-device (no DataParallel, no
 *     torch.distributed anywhere; README.md:47 `CUDA_VISIBLE_DEVICES=0`).  What this replaces is what N
 *     independent copies of mnist/train.py:197-219 would have to add between `backward()` (:218) and
 *     `optimizer.step()` (:219): the average of every parameter's gradient over the replicas.
 *
 *   One process per GPU, one communicator per process.  Start-up: rank 0 calls mvae_comm_unique_id and hands the
 *   128 bytes to its peers over any host transport (a file, a socket, MPI, torch.distributed's store); every rank
 *   calls mvae_comm_init (collective: returns when all `world` ranks joined); mvae_comm_broadcast makes parameters
 *   and buffers equal.  Per step: after the last weight-gradient launch of a bucket (a contiguous fp32 range of the
 *   gradient arena) was enqueued on `stream`, mvae_comm_allreduce_async sums that range over the ranks IN PLACE on
 *   the communicator's own stream -- ordered after `stream`'s work so far, concurrent with what `stream` gets next --
 *   and returns a ticket; mvae_comm_wait(ticket, stream) makes `stream` wait for it (then: mvae_adam_apply on the
 *   range with grad_scale = 1 / world).  Enqueue-only, no host sync, and capturable: issued between
 *   hipStreamBeginCapture / EndCapture on `stream` the collectives become nodes of the step's hipGraph (the
 *   communicator's stream forks from and joins back into the capturing stream -- every ticket must be waited on
 *   before EndCapture).  RCCL is dlopen-ed on first use: mvae_comm_use_library(path) > $MVAE_RCCL_LIB >
 *   "librccl.so.1" on the loader path.  At most 16 tickets may be outstanding; tickets are per communicator.
 *   Returns MVAE_OK / MVAE_ERR_ARG / MVAE_ERR_COMM (text: mvae_comm_last_error).
 * ------------------------------------------------------------------------------------ */
#define MVAE_COMM_ID_BYTES 128
typedef struct mvae_comm mvae_comm_t;
int mvae_comm_use_library(const char *librccl_path);         /* before first use; MVAE_ERR_ARG once another one is bound */
int mvae_comm_rccl_version(void);                            /* RCCL's version code (e.g. 22606), or MVAE_ERR_COMM */
int mvae_comm_unique_id(void *id_out, size_t id_bytes);      /* id_bytes >= MVAE_COMM_ID_BYTES */
int mvae_comm_init(mvae_comm_t **comm, const void *id, size_t id_bytes, int rank, int world, int device);
int mvae_comm_rank(const mvae_comm_t *comm);
int mvae_comm_world(const mvae_comm_t *comm);
const char *mvae_comm_last_error(const mvae_comm_t *comm);
int mvae_comm_broadcast(mvae_comm_t *comm, void *buf, size_t bytes, int root, mvae_stream_t stream);
int mvae_comm_allreduce_async(mvae_comm_t *comm, float *buf, size_t count, mvae_stream_t stream, int *ticket);
int mvae_comm_wait(mvae_comm_t *comm, int ticket /* < 0: everything issued so far */, mvae_stream_t stream);
/* Failure detection (ABI 5).  mvae_comm_async_error: MVAE_ERR_COMM once a collective of this communicator has failed
 * asynchronously (ncclCommGetAsyncError; MVAE_OK when the bound library has no such entry).  mvae_comm_synchronize is
 * the watchdog a host puts where it would otherwise call hipStreamSynchronize: it blocks until everything enqueued
 * on `stream` so far -- after mvae_comm_wait that includes the collectives -- has finished, for at most timeout_ms,
 * polling the asynchronous error state meanwhile; MVAE_ERR_COMM = a peer failed or the budget ran out (the
 * collective of a dead peer never completes), after which the communicator must be abandoned.  Not capturable. */
int mvae_comm_async_error(mvae_comm_t *comm);
int mvae_comm_synchronize(mvae_comm_t *comm, mvae_stream_t stream, int timeout_ms);
int mvae_comm_destroy(mvae_comm_t *comm);

#ifdef MVAE_TUNING
/* ------------------------------------------------------------------------------------
 * Tuning overrides -- libmvae_hip_tuning.so only (the same sources built with -DMVAE_TUNING).
 * Process-global, not thread-safe, not part of the product ABI.
 * ---------------------------------------------------------------------------------- */
/* force the per-wave tile (wm, wn in {1,2}: 64/128 rows/cols per block) and the number of reduction
 * splits of every subsequent GEMM-shaped launch; 0 = automatic; splits < 0: forward forms never split */
void mvae_debug_set_tiling(int wm, int wn, int splits);
/* force the number of k-wave groups (1, 2, 4) of 64x64-tile launches; 0 = automatic */
void mvae_debug_set_kwaves(int kw);
/* off != 0: never use the small (64x32 / 32x64 / 32x32, BK = 64) Linear layouts; waves in {4, 8}: force
 * their block size; 0 = automatic */
void mvae_debug_set_small(int off, int waves);
/* blocks a split reduction aims for; 0 = automatic */
void mvae_debug_set_split_target(long blocks);
/* knock-out study of the direct Linear weight-gradient kernel: 1 = loads without MFMAs, 2 = MFMAs without
 * loads (results are then meaningless); 0 = normal */
void mvae_debug_set_knockout(int mode);
#endif

#ifdef __cplusplus
}
#endif
#endif /* MVAE_HIP_H */
