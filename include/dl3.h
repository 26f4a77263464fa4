/*
 * dl3.h — C ABI of libdl3.so: the MI355X (gfx950) native operator set behind
 * Deeplabv3() (reference: deeplabv3p.py:209-466, subpixel.py:41-103, utils.py:127-130,169-214).
 *
 * The reference has no FFI of its own: every FLOP of its hot path runs inside
 * Keras/TensorFlow layer objects (Conv2D, DepthwiseConv2D, BatchNormalization, ...,
 * imported at deeplabv3p.py:29-40).  Each entry point below replaces the device-side
 * work of one of those layer types (forward and gradients); the Python host layer in
 * keras-segmentation-deeplab-v3.1_amd/ re-exposes the reference's Deeplabv3()/Subpixel/
 * SegModel surface on top of them.  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *  - every op returns int: 0 OK, -1 bad argument, -2 HIP error, -3 workspace too small,
 *    -4 unsupported configuration; dl3_last_error() gives a thread-local message.
 *  - all tensor pointers are caller-owned DEVICE pointers (fp32 unless noted), NHWC,
 *    weights in Keras layouts (pointwise [K][N]; depthwise [3][3][C]; dense [3][3][Cin][Cout]).
 *  - no allocation, no synchronisation, no global mutable state inside ops; the last
 *    argument is the hipStream_t to enqueue on.  Ops are hipGraph-capturable.
 *    Two documented exceptions, both opt-in and off by default: dl3_set_gemm_math() is a
 *    process-wide switch (by design: it selects the arithmetic of every later launch), and in
 *    split math (DL3_MATH_SPLIT) the dl3_pwconv_fwd / _bwd_data launches keep ONE device
 *    scratch per device for the launch's pre-split weights, grown with hipMalloc on first use —
 *    so the first split-math launch of a process must not be issued inside a stream capture
 *    (it returns -1 with a message; the engine's eager warm-up step takes care of it).  The
 *    f32 path — the default, and the only one any benchmark headline uses — keeps the rule.
 *  - "input transform": a tensor is handed over as (raw, scale[C], shift[C], act) and is
 *    read as act(scale*raw+shift) — BatchNorm + ReLU/ReLU6 of the PRODUCER are applied on
 *    load by the consumer, so an activation is written once and read once.  scale==NULL
 *    means identity affine.
 *  - "gradient operand": dY is handed over as (g, yraw, cA[C], cB[C], cC[C]) and read as
 *    cA*g + cB*yraw + cC — the BatchNorm backward of the producer applied on load
 *    (cA==NULL means dY = g).
 *  - "partials": per-channel reductions (BN batch statistics, BN backward sums, weight
 *    gradients) are written as P deterministic partial rows; dl3_*_partials() returns P
 *    for a shape, dl3_bn_finalize()/dl3_bn_bwd_finalize()/dl3_reduce_partials() fold them
 *    in a fixed order.  No float atomics anywhere: results are run-to-run bit-identical.
 */
#ifndef DL3_H
#define DL3_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

#define DL3_OK 0
#define DL3_EINVAL (-1)
#define DL3_EHIP (-2)
#define DL3_EWORKSPACE (-3)
#define DL3_EUNSUPPORTED (-4)

#define DL3_ACT_NONE 0
#define DL3_ACT_RELU 1  /* Activation('relu')            deeplabv3p.py:72,77,82,287,380,387,409 */
#define DL3_ACT_RELU6 2 /* relu(x, max_value=6.)         deeplabv3p.py:181,192,325 */

#define DL3_IMPL_AUTO 0
#define DL3_IMPL_GATHER 1 /* generic 9-tap gather kernel (any stride / rate / pad) */
#define DL3_IMPL_MARCH 2  /* stride-1 row-marching kernel (each input row read once) */

int dl3_version(void);
const char *dl3_last_error(void);

/* ---- DepthwiseConv2D 3x3 (deeplabv3p.py:73-74, :186-188) ------------------------------ */
/* number of partial rows P written by dwconv fwd (stat_partial [P][C][2]) and bwd
 * (dstat_partial [P][C][2], dw_partial [P][9][C]) for this shape/impl */
int dl3_dwconv3x3_partials(int N, int H, int W, int C, int stride, int rate, int Ho, int Wo, int impl);
/* y[n,oy,ox,c] = sum_{i,j} T(x)[n, oy*stride-pad_t+i*rate, ox*stride-pad_l+j*rate, c] * w[i][j][c];
 * stat_partial (nullable): per-channel sum(y), sum(y^2) partials */
int dl3_dwconv3x3_fwd(const float *x, const float *in_scale, const float *in_shift, int in_act,
                      const float *w, float *y, int N, int H, int W, int C, int stride, int rate,
                      int pad_t, int pad_l, int Ho, int Wo, float *stat_partial, int impl, void *stream);
/* fused bwd-data + bwd-weight.
 *   dY = cA*g + cB*yraw + cC                                   [N,Ho,Wo,C]
 *   dw_partial[p][i][j][c] = partial sum_{n,oy,ox} T(x)[...tap...] * dY
 *   dx = mask_{in_act}(conv^T(dY, w)) + dx_add                 [N,H,W,C]   (dx nullable)
 *   dstat_partial (nullable): sum(dx), sum(dx * (x-x_mean)*x_invstd) partials */
int dl3_dwconv3x3_bwd(const float *g, const float *yraw, const float *cA, const float *cB, const float *cC,
                      const float *x, const float *in_scale, const float *in_shift, int in_act,
                      const float *w, float *dx, const float *dx_add, const float *x_mean,
                      const float *x_invstd, float *dstat_partial, float *dw_partial, int N, int H, int W,
                      int C, int stride, int rate, int pad_t, int pad_l, int Ho, int Wo, int impl, void *stream);

/* the same launch with the BatchNorm-backward sums of dx taken against ANOTHER tensor: dstat_partial = sum(dx),
 * sum(dx * (stat_x - x_mean) * x_invstd), stat_x [N,H,W,C] — the other input of the residual Add whose output gradient
 * this launch completes (Xception's `sum` shortcuts, deeplabv3p.py:147-149: the gradient reaches the block's last
 * pointwise BatchNorm unchanged, so its sums need no pass of their own; round 4).  March kernel only (stride 1, SAME). */
int dl3_dwconv3x3_bwd_sx(const float *g, const float *yraw, const float *cA, const float *cB, const float *cC,
                         const float *x, const float *in_scale, const float *in_shift, int in_act, const float *w,
                         float *dx, const float *dx_add, const float *stat_x, const float *x_mean,
                         const float *x_invstd, float *dstat_partial, float *dw_partial, int N, int H, int W, int C,
                         int stride, int rate, int pad_t, int pad_l, int Ho, int Wo, int impl, void *stream);

/* ---- Conv2D 1x1 = GEMM on fp32 MFMA (deeplabv3p.py:78-79,:175,:194,:377,:385,:406,:420,:438) -- */
#define DL3_MATH_ENV (-1)  /* follow the environment (DL3_GEMM_MATH=split selects split math); the default */
#define DL3_MATH_F32 0     /* v_mfma_f32_32x32x2_f32 */
#define DL3_MATH_SPLIT 1   /* fp32 operands cut exactly into three bf16 pieces, six of the nine piece products on
                              v_mfma_f32_32x32x16_bf16, fp32 accumulate (same error class as the f32 MFMA) */
/* matrix math of the dl3_pwconv_* launches issued after the call (process-wide; launches already captured in a
 * hipGraph keep what they were captured with); dl3_get_gemm_math returns the mode in effect (0 or 1) */
int dl3_set_gemm_math(int mode);
int dl3_get_gemm_math(void);
/* P for the [M,K]x[K,N] GEMM's per-output-channel partials */
int dl3_pwconv_partials(int M, int K, int N);
/* which kernel a dl3_pwconv_fwd launch of this shape takes (16-byte aligned operands, leading dimensions multiples of 4,
 * no addend): 0 the tiled MFMA GEMM, 1 the weight-stationary streaming kernel of the HBM-bound layers (round 5: the
 * whole K x N matrix in LDS, every wave walks 32-row tiles on its own, 16-byte stores; deeplabv3p.py:175-198 at
 * 16..192 channels, M >= 32768), 2 the weight-stationary kernel of the MFMA-bound short reductions (round 6: K = 160 / 96 /
 * 64 into an output at least twice as wide, M >= 98304 forward / 65536 bwd-data, 131072 for K = 64; DL3_WS2=0 disables it — and its packed-output variant for the
 * logits layer, K = 256, N <= 32, ldy == N, M >= 8192; DL3_NARROW=0).  Diagnostic only. */
int dl3_pwconv_fwd_impl(int M, int K, int N);
/* ... and the route of a launch by name, for plans that want to assert what they benchmark (tests/test_host.py):
 * dir 0 = forward (as dl3_pwconv_fwd_impl), 1 = bwd-data with the single-tensor dY, a mask operand and no addend, 2 = the
 * weight gradient (two-tensor operand), 3 = bwd-data with the single-tensor dY (rows back to back) and neither mask, addend
 * nor BatchNorm sums, 4 = the weight gradient with a single-tensor dY and a bias gradient (3, 4: the logits layer).
 * Returns DL3_ROUTE_*.  Diagnostic only. */
#define DL3_ROUTE_TILED 0      /* pw_gemm_stream_kernel / pw_gemm_kernel / pw_wgrad_kernel */
#define DL3_ROUTE_WS_HBM 1     /* pw_fwd_ws_kernel */
#define DL3_ROUTE_WS_MFMA 2    /* pw_ws2_kernel */
#define DL3_ROUTE_KSPLIT 3     /* pw_ksplit32_kernel (1 024 - 16 384 rows) */
#define DL3_ROUTE_WGRAD_ROW 4  /* pw_wgrad_row_kernel (one tile row over K) */
#define DL3_ROUTE_NARROW 5     /* the logits layer, N = classes <= 32 off a 256-wide input: pw_ws2_kernel FLAT (dir 0),
                                  pw_narrowk_kernel (dir 3), pw_wgrad_narrow_kernel (dir 4) */
int dl3_pwconv_route(int dir, int M, int K, int N);
/* y[M,N](ldy) = T(x)[M,K](ldx) . w[K,N] (+bias[N]); stat_partial (nullable) [P][N][2] = sum(y), sum(y^2) */
int dl3_pwconv_fwd(const float *x, int ldx, const float *in_scale, const float *in_shift, int in_act,
                   const float *w, const float *bias, float *y, int ldy, int M, int K, int N,
                   float *stat_partial, void *stream);
/* the same with an addend: y[m,:] += add[(m / add_div) * ldadd + :] — add_div = 1: a residual tensor; add_div = H*W: one
 * row per image, i.e. the contribution of channels that are constant over the image (the ASPP image-pooling branch,
 * deeplabv3p.py:375-382,:402-406: the broadcast 1x1 feature never has to be materialised or multiplied per pixel) */
int dl3_pwconv_fwd_add(const float *x, int ldx, const float *in_scale, const float *in_shift, int in_act,
                       const float *w, const float *bias, float *y, int ldy, int M, int K, int N,
                       float *stat_partial, const float *add, int ldadd, int add_div, void *stream);
/* the same product for a FEW rows — one per image: the ASPP image-pooling branch (deeplabv3p.py:375-382) and its share of
 * concat_projection (:402-406) — accumulated in DOUBLE and rounded once (round 5).  The result is a per-image constant the
 * network adds to every pixel of the map: its rounding error does not average out over pixels, and a reduction of 2 048 on
 * the f32 MFMA was where the path's distance to float64 left torch-fp32's (tools/r5/xception_layer_distance.py).  add
 * (nullable): y[m,:] += add[(m / add_div) * ldadd + :].  No BatchNorm partial sums (few rows: dl3_bn_finalize_direct). */
int dl3_pwconv_fwd_rows(const float *x, int ldx, const float *in_scale, const float *in_shift, int in_act,
                        const float *w, const float *bias, float *y, int ldy, int M, int K, int N, const float *add,
                        int ldadd, int add_div, void *stream);
/* dx[M,K](lddx) = mask_{in_act}(dY[M,N] . wT[N,K]) + add_scale*dx_add ; dY = cA*g + cB*yraw + cC.
 * x/in_scale/in_shift/in_act describe the FORWARD input (needed for the mask and x_hat);
 * dx_add row address = dx_add + (m / add_div)*ldadd (add_div = H*W broadcasts a per-image vector);
 * dstat_partial (nullable) [P'][K][2] = sum(dx), sum(dx*(x-x_mean)*x_invstd), P' = dl3_pwconv_partials(M,N,K) */
int dl3_pwconv_bwd_data(const float *g, int ldg, const float *yraw, int ldyraw, const float *cA,
                        const float *cB, const float *cC, const float *wT, float *dx, int lddx,
                        const float *x, int ldx, const float *in_scale, const float *in_shift, int in_act,
                        const float *dx_add, int ldadd, int add_div, float add_scale, const float *x_mean,
                        const float *x_invstd, float *dstat_partial, int M, int K, int N, void *stream);
size_t dl3_pwconv_bwd_weight_workspace(int M, int K, int N);
/* dw[K,N] = T(x)^T . dY ; dbias[N] (nullable) = colsum(dY).  The launch reduces over M in S =
 * dl3_pwconv_bwd_weight_splits(M, K, N, cA != NULL) deterministic slabs [S][K][N] at the head of the workspace and folds
 * them into dw.  dw == NULL (then dbias must be NULL too) leaves the slabs where they are: the caller folds them later —
 * dl3_reduce_partials(workspace, S, K*N, dw), or together with every other weight gradient of the step in ONE
 * dl3_reduce_partials_batched launch (the workspace must then be the launch's own until that fold has run). */
int dl3_pwconv_bwd_weight_splits(int M, int K, int N, int two_tensor_dy);
int dl3_pwconv_bwd_weight(const float *x, int ldx, const float *in_scale, const float *in_shift, int in_act,
                          const float *g, int ldg, const float *yraw, int ldyraw, const float *cA,
                          const float *cB, const float *cC, float *dw, float *dbias, int M, int K, int N,
                          void *workspace, size_t workspace_bytes, void *stream);
/* the same launch, which also writes the gradient operand it assembles anyway, dy_out[M][N](lddy) = cA*g + cB*yraw + cC
 * (round 4): dl3_pwconv_bwd_data of the same layer can then be handed (g = dy_out, cA = NULL) — ONE operand tensor, no
 * operand transform, and with it the room for a straight-line masked epilogue.  Costs one write of dY; the bwd-data GEMM
 * reads one tensor less.  dy_out 16-byte aligned, lddy >= N and a multiple of 4.  Replaces the BatchNormalization
 * backward TF runs as a separate op in front of both Conv2D gradients (deeplabv3p.py:175-201 expand / project convs). */
int dl3_pwconv_bwd_weight_dy(const float *x, int ldx, const float *in_scale, const float *in_shift, int in_act,
                             const float *g, int ldg, const float *yraw, int ldyraw, const float *cA,
                             const float *cB, const float *cC, float *dw, float *dbias, int M, int K, int N,
                             void *workspace, size_t workspace_bytes, float *dy_out, int lddy, void *stream);
/* Both gradients of a 1x1 convolution in ONE pass over (g, yraw, x) — for the HBM-bound layers with a small weight matrix
 * (round 4: the expand / project convolutions of the first inverted-residual blocks, deeplabv3p.py:175-201, 16..144
 * channels on 256x256 / 128x128 maps, where _bwd_weight and _bwd_data each read the wide gradient operand once):
 *   dw[K,N] = T(x)^T . dY                                  (dw == NULL: the [S][K][N] slabs stay in the workspace)
 *   dx[M,K](lddx) = mask_{in_act}(dY . wT[N,K]) + dx_add   dY = cA*g + cB*yraw + cC
 *   dstat_partial (nullable) [S][K][2] = sum(dx), sum(dx * (stat_x - x_mean) * x_invstd) — stat_x is the forward input
 *   itself or, when the gradient reaches another BatchNorm'ed tensor unchanged through a residual Add, that tensor.
 * S = dl3_pwconv_bwd_fused_splits(M, K, N) workgroups / slabs / partial rows; workspace >= _workspace(M, K, N) bytes.
 * _supported: 0 unless K, N are multiples of 4 with ceil(K/32) * ceil(N/32) <= 6; 2: any epilogue; 1 (K > 64): without
 * dx_add and with stat_x == x only (-4 otherwise).  All operands 16-byte aligned, leading dimensions multiples of 4. */
int dl3_pwconv_bwd_fused_supported(int M, int K, int N);
int dl3_pwconv_bwd_fused_splits(int M, int K, int N);
size_t dl3_pwconv_bwd_fused_workspace(int M, int K, int N);
int dl3_pwconv_bwd_fused(const float *x, int ldx, const float *in_scale, const float *in_shift, int in_act,
                         const float *g, int ldg, const float *yraw, int ldyraw, const float *cA, const float *cB,
                         const float *cC, const float *wT, float *dw, float *dx, int lddx, const float *dx_add,
                         int ldadd, const float *stat_x, int ldstatx, const float *x_mean, const float *x_invstd,
                         float *dstat_partial, int M, int K, int N, void *workspace, size_t workspace_bytes,
                         void *stream);
/* out[cols][rows] = in[rows][cols]^T  (W[K,N] -> WT[N,K] for bwd_data) */
int dl3_transpose(const float *in, float *out, int rows, int cols, void *stream);
/* n transposes in one launch (all W -> WT of a backward pass).  desc (device, int64 [n][6]): in pointer, out pointer,
 * rows, cols, index of the matrix's first 32x32 tile, tiles per tile-row (= ceil(cols/32)); matrices in ascending
 * first-tile order; total_tiles = sum of ceil(rows/32)*ceil(cols/32). */
int dl3_transpose_batched(const long long *desc, int n, int total_tiles, void *stream);

/* ---- dense Conv2D 3x3 (stem convs: deeplabv3p.py:283,:289,:318) ----------------------- */
int dl3_conv3x3_partials(int N, int Ho, int Wo, int Cout);
int dl3_conv3x3_fwd(const float *x, const float *in_scale, const float *in_shift, int in_act, const float *w,
                    float *y, int N, int H, int W, int Cin, int Cout, int stride, int pad_t, int pad_l,
                    int Ho, int Wo, float *stat_partial, void *stream);
/* dw_partial [P][3][3][Cin][Cout] */
int dl3_conv3x3_bwd_weight(const float *x, const float *in_scale, const float *in_shift, int in_act,
                           const float *g, const float *yraw, const float *cA, const float *cB,
                           const float *cC, float *dw_partial, int N, int H, int W, int Cin, int Cout,
                           int stride, int pad_t, int pad_l, int Ho, int Wo, void *stream);
/* dx = mask(conv^T(dY,w)) + dx_add, optional dstat partials ([P'][Cin][2], P' = dl3_conv3x3_partials(N,H,W,Cin)) */
int dl3_conv3x3_bwd_data(const float *g, const float *yraw, const float *cA, const float *cB, const float *cC,
                         const float *w, float *dx, const float *x, const float *in_scale,
                         const float *in_shift, int in_act, const float *dx_add, const float *x_mean,
                         const float *x_invstd, float *dstat_partial, int N, int H, int W, int Cin, int Cout,
                         int stride, int pad_t, int pad_l, int Ho, int Wo, void *stream);

/* Matrix-pipe route for a dense 3x3 conv between 32- or 64-channel tensors (xception entry_flow_conv1_2 32 -> 64,
 * deeplabv3p.py:289), im2col-free: the taps are gathered straight into the MFMA A operand, the weight slice of one tap
 * is the LDS-staged B operand.  Same semantics as the three ops above except: the stat partial buffers have
 * P = dl3_conv3x3_mfma_partials(N, Ho, Wo) (forward) / (N, H, W) (bwd-data) rows; bwd_weight writes the finished
 * dw [3][3][Cin][Cout] (its per-workgroup slabs live in the caller's workspace); bwd_data takes
 * wT [Cout][9*Cin] (dl3_transpose of w). */
int dl3_conv3x3_mfma_supported(int Cin, int Cout);
int dl3_conv3x3_mfma_partials(int N, int Hc, int Wc);
int dl3_conv3x3_mfma_fwd(const float *x, const float *in_scale, const float *in_shift, int in_act, const float *w,
                         float *y, int N, int H, int W, int Cin, int Cout, int stride, int pad_t, int pad_l, int Ho,
                         int Wo, float *stat_partial, void *stream);
size_t dl3_conv3x3_mfma_bwd_weight_workspace(int N, int H, int W, int Cin, int Cout, int stride, int Ho, int Wo);
int dl3_conv3x3_mfma_bwd_weight(const float *x, const float *in_scale, const float *in_shift, int in_act,
                                const float *g, const float *yraw, const float *cA, const float *cB, const float *cC,
                                float *dw, int N, int H, int W, int Cin, int Cout, int stride, int pad_t, int pad_l,
                                int Ho, int Wo, void *workspace, size_t workspace_bytes, void *stream);
int dl3_conv3x3_mfma_bwd_data(const float *g, const float *yraw, const float *cA, const float *cB, const float *cC,
                              const float *wT, float *dx, const float *x, const float *in_scale,
                              const float *in_shift, int in_act, const float *dx_add, const float *x_mean,
                              const float *x_invstd, float *dstat_partial, int N, int H, int W, int Cin, int Cout,
                              int stride, int pad_t, int pad_l, int Ho, int Wo, void *stream);

/* ---- BatchNormalization (54 layers mnv2 / 146 xception; eps 1e-3 or 1e-5) -------------- */
/* training mode: fold stat partials [P][ldc][2] (channels c0..c0+C-1 at partial + 2*c0) into
 * scale = gamma*invstd, shift = beta - mean*scale, mean, invstd (biased batch variance) and
 * update the moving statistics: moving_mean = m*moving_mean + (1-m)*mean,
 * moving_var = m*moving_var + (1-m)*var_unbias*biased_var.  count = n = N*H*W.  var_unbias is the framework's
 * sample-variance factor, chosen by the host: Keras 2.2.4 on TF 1.13 (the reference's environment) multiplies
 * FusedBatchNorm's already Bessel-corrected batch variance by n/(n-(1+eps)) once more (keras/layers/normalization.py
 * `variance *= sample_size / (sample_size - (1.0 + self.epsilon))`), i.e. var_unbias = n/(n-1) * n/(n-(1+eps));
 * tf.keras uses n/(n-1).  moving_mean/moving_var NULL (together): a non-trainable layer, whose moving statistics
 * Keras 2.2.x leaves untouched. */
int dl3_bn_finalize(const float *stat_partial, int P, int ldc, int C, double count, const float *gamma,
                    const float *beta, float eps, float momentum, double var_unbias, float *scale, float *shift,
                    float *mean, float *invstd, float *moving_mean, float *moving_var, void *stream);
/* the same outputs for a SMALL tensor y[M][ldy] (channels 0..C-1), statistics taken straight from its values in two
 * double-precision passes (mean, then squared deviations).  The partial-sum form computes sum(y^2)/n - mean^2, which
 * cancels catastrophically when |mean| >> spread: the image-pooling BatchNorm (deeplabv3p.py:375-379) sees ONE value per
 * image.  The host uses it for M <= 4096 rows. */
int dl3_bn_finalize_direct(const float *y, int ldy, int M, int C, const float *gamma, const float *beta, float eps,
                           float momentum, double var_unbias, float *scale, float *shift, float *mean, float *invstd,
                           float *moving_mean, float *moving_var, void *stream);
/* inference / frozen mode: scale, shift, mean, invstd from the moving statistics */
int dl3_bn_frozen(const float *gamma, const float *beta, const float *moving_mean, const float *moving_var,
                  float eps, int C, float *scale, float *shift, float *mean, float *invstd, void *stream);
/* the same for a tensor whose PRODUCER subtracts the mean: neg_mean[C] = -moving_mean is handed to the producing
 * convolution as its bias (a BatchNorm'ed Conv2D has none of its own, deeplabv3p.py:78-79), the tensor holds y - mean,
 * and consumers read scale*(y - mean) + beta with scale = gamma*invstd, shift = beta, mean = 0.  scale*y + (beta -
 * mean*scale) rounds the large product mean*scale into the shift once and for all — half an ulp of |mean*scale| on
 * every element, the size of the tensor's own rounding noise when |mean| >> sigma; (y - mean) is exact there.  This is
 * the form the reference's BatchNormalization evaluates in inference, gamma*(x - mean)/sqrt(var + eps) + beta. */
int dl3_bn_frozen_centered(const float *gamma, const float *beta, const float *moving_mean, const float *moving_var,
                           float eps, int C, float *scale, float *shift, float *mean, float *invstd, float *neg_mean,
                           void *stream);
/* fold backward partials (sum g, sum g*x_hat) into dgamma, dbeta and the on-load coefficients
 * dY = cA*g + cB*yraw + cC.  batch_mode=1: full BN backward; 0: frozen statistics (cB=cC=0). */
int dl3_bn_bwd_finalize(const float *dstat_partial, int P, int ldc, int C, double count, const float *gamma,
                        const float *mean, const float *invstd, int batch_mode, float *cA, float *cB,
                        float *cC, float *dgamma, float *dbeta, void *stream);

/* ---- element-wise / reductions ------------------------------------------------------- */
int dl3_rows_partials(int M); /* P for row-wise reducing kernels over M rows */
/* out = act_a(sa*a+ta) + act_b(sb*b+tb)  (b nullable): Add (deeplabv3p.py:147-149,:201) and
 * materialisation of a BN(+act) output.  Optional Dropout (deeplabv3p.py:410): if drop_rate>0,
 * out *= keep_mask(seed, step, element index)/(1-drop_rate). */
int dl3_affine_add(const float *a, int lda, const float *sa, const float *ta, int act_a, const float *b,
                   int ldb, const float *sb, const float *tb, int act_b, float *out, int ldo, int M, int C,
                   float drop_rate, unsigned long long drop_seed, const unsigned long long *drop_step, void *stream);
/* *counter += inc (device memory).  The dropout step number: launch arguments are frozen inside a captured hipGraph,
 * so kernels read the step from *drop_step (nullable) and use the seed drop_seed + step * 0xD1B54A32D192ED03. */
int dl3_counter_add(unsigned long long *counter, unsigned long long inc, void *stream);
/* gout = mask_{act}(gin_scale * gin[m / gin_div] * dropmask/(1-rate)) + add ; gin_div = H*W broadcasts a
 * per-image gradient vector (backward of the global average pool); dstat partials [P][C][2] (nullable),
 * P = dl3_rows_partials(M).  gout may alias gin (gin_div == 1) and/or add. */
int dl3_grad_finish(const float *gin, int ldgin, int gin_div, float gin_scale, float *gout, int ldgout,
                    const float *add, int ldadd, const float *xraw, int ldx, const float *scale,
                    const float *shift, int act, const float *mean, const float *invstd, float *dstat_partial,
                    int M, int C, float drop_rate, unsigned long long drop_seed, const unsigned long long *drop_step,
                    void *stream);
/* strided 1x1 convolutions (Xception shortcuts, _conv2d_same(kernel_size=1, stride=2), deeplabv3p.py:143-145) sample
 * rows/cols 0, s, 2s, ...: y[n,oy,ox,c] = T(x)[n, oy*s, ox*s, c] compacts the sampled pixels for the GEMM;
 * the backward scatters the compact gradient back (zeros elsewhere). */
int dl3_subsample_fwd(const float *x, int ldx, const float *in_scale, const float *in_shift, int in_act, float *y,
                      int N, int H, int W, int C, int stride, int Ho, int Wo, void *stream);
int dl3_subsample_bwd(const float *g, float *dx, int N, int H, int W, int C, int stride, int Ho, int Wo, void *stream);
/* AveragePooling2D over the whole map (deeplabv3p.py:375): out[n][c] = out_scale * sum_hw T(x)[n,hw,c] */
int dl3_gap_fwd(const float *x, int ldx, const float *in_scale, const float *in_shift, int in_act, float *out,
                int N, int HW, int C, float out_scale, void *stream);
/* tf.image.resize_bilinear, TF1 legacy (align_corners=False, no half-pixel): deeplabv3p.py:382,:418,:439 */
int dl3_resize_bilinear_fwd(const float *x, int ldx, const float *in_scale, const float *in_shift, int in_act,
                            float *y, int ldy, int N, int Hi, int Wi, int Ho, int Wo, int C, void *stream);
/* dx[N,Hi,Wi,C] (+= if accumulate) = resize^T(dy) — deterministic gather form.  With a workspace of
 * dl3_resize_bilinear_bwd_workspace() bytes the transpose runs separably (columns, then rows: each dy element is
 * read once); workspace == NULL selects the single-pass 2-D gather. */
size_t dl3_resize_bilinear_bwd_workspace(int N, int Hi, int Wi, int Ho, int Wo, int C);
int dl3_resize_bilinear_bwd(const float *dy, int lddy, float *dx, int lddx, int N, int Hi, int Wi, int Ho,
                            int Wo, int C, int accumulate, void *workspace, size_t workspace_bytes, void *stream);
/* Subpixel._phase_shift (subpixel.py:77-88): out[n,ia*r+q,ib*r+p,ch] = in[n,ia,ib,ch*r*r+p*r+q];
 * inverse!=0 applies the inverse permutation (the backward pass) */
int dl3_phase_shift(const float *in, float *out, int N, int H, int W, int Cout, int r, int inverse, void *stream);
/* Subpixel head, training tail (utils.py:195-198 + the loss): softmax cross-entropy evaluated on the UNshuffled output u
 * [N,H,W,C*r*r] of the Subpixel convolution — labels / weights live at the shuffled resolution [N, H*r, W*r] — writing
 * du in u's own layout, so that neither the shuffled logits nor the shuffled gradient nor the two phase-shift passes
 * exist.  loss_partial [P], P = dl3_shuffle_xent_partials(...) (0: shape not supported, fall back to dl3_phase_shift +
 * dl3_softmax_xent).  C <= 32. */
int dl3_shuffle_xent_partials(int N, int H, int W, int C, int r);
int dl3_shuffle_softmax_xent(const float *u, const float *labels, const float *weights, const float *nnz, float *du,
                             float *loss_partial, int N, int H, int W, int C, int r, void *stream);
/* Subpixel with kernel_size > 1 (subpixel.py:42-58: Subpixel IS a Conv2D with any kernel_size; icnr_weights' default
 * shape is 3x3, subpixel.py:9): the k x k taps of T(x) = act(scale*x+shift), zero outside the image, gathered next to
 * each other — cols[(n,oy,ox)][(i*k+j)*C + c] = T(x)[n, oy-pad_t+i, ox-pad_l+j, c] (stride 1) — so that the Keras
 * kernel [k][k][C][F] read as a [k*k*C][F] matrix drives the ordinary 1x1 GEMM entry points.  _bwd is the transpose:
 * dx[n,iy,ix,c] = sum over taps of dcols (deterministic gather, no atomics). */
int dl3_conv_taps_fwd(const float *x, int ldx, const float *in_scale, const float *in_shift, int in_act, float *cols,
                      int N, int H, int W, int C, int k, int pad_t, int pad_l, int Ho, int Wo, void *stream);
int dl3_conv_taps_bwd(const float *dcols, float *dx, int N, int H, int W, int C, int k, int pad_t, int pad_l, int Ho,
                      int Wo, void *stream);
/* softmax over the last axis (deeplabv3p.py:441,:444) */
int dl3_softmax_fwd(const float *logits, float *probs, int M, int C, void *stream);
/* argmax over the last axis -> int32 (first maximum wins, like np.argmax) */
int dl3_argmax(const float *x, int *out, int M, int C, void *stream);
/* count of non-zero sample weights -> *out (float) ; sparse_crossentropy_ignoring_last_label
 * (utils.py:127-130) with Keras temporal sample weights:
 *   loss = sum_m w[m]*(-log clip(p[m,label]))/nnz ; dlogits = (p - onehot)*w/nnz ; label==C (void): onehot=0.
 * loss_partial [P] (P = dl3_rows_partials(M)) holds per-block shares of the loss (already divided by nnz);
 * probs nullable. */
int dl3_count_nonzero(const float *w, int M, float *out, void *stream);
int dl3_softmax_xent(const float *logits, const float *labels, const float *weights, const float *nnz,
                     float *probs, float *dlogits, float *loss_partial, int M, int C, void *stream);
/* the same loss with the final resize_bilinear (deeplabv3p.py:439, utils.py:190) fused in: logits_lo is the
 * [N,Hi,Wi,C] output of the logits conv, labels / weights / dlogits live at [N,Ho,Wo]; the full-resolution logits
 * are interpolated in registers and never touch HBM.  C <= 32. */
int dl3_upsample_softmax_xent(const float *logits_lo, const float *labels, const float *weights, const float *nnz,
                              float *probs, float *dlogits, float *loss_partial, int N, int Hi, int Wi, int Ho, int Wo,
                              int C, void *stream);
/* the training tail without the full-resolution gradient: loss as above, and the x half of the transposed resize
 * applied on chip — dlogits_xfold [N,Ho,Wi,C] = sum over the output columns of each output row (fixed order);
 * dl3_resize_bilinear_bwd_rows folds the rows (the y half) into dx [N,Hi,Wi,C].  loss_partial has
 * P = dl3_xent_fold_partials(N, Ho) entries.  C <= 32 and (Wo + 2*Wi)*C*4 <= 64 KB (an output row and its two
 * source rows live in LDS). */
int dl3_xent_fold_partials(int N, int Ho);
int dl3_upsample_softmax_xent_fold(const float *logits_lo, const float *labels, const float *weights, const float *nnz,
                                   float *dlogits_xfold, float *loss_partial, int N, int Hi, int Wi, int Ho, int Wo,
                                   int C, void *stream);
int dl3_resize_bilinear_bwd_rows(const float *xfold, float *dx, int lddx, int N, int Hi, int Wi, int Ho, int C,
                                 int accumulate, void *stream);
/* out[i] = sum_p partial[p][i]  (fixed order) */
int dl3_reduce_partials(const float *partial, int P, int n, float *out, void *stream);
/* m folds in one launch, each bit-identical to its own dl3_reduce_partials call.  desc (device, int64 [m][5]): partial
 * pointer, out pointer, P, n, index of the fold's first workgroup; folds in ascending first-workgroup order; a fold takes
 * dl3_reduce_partials_blocks(P, n) workgroups and total_blocks is their sum.  (All weight-gradient slab folds of a
 * backward pass: 55 launches less per step for MobileNetV2.) */
int dl3_reduce_partials_blocks(int P, int n);
int dl3_reduce_partials_batched(const long long *desc, int m, int total_blocks, void *stream);
int dl3_fill(float *p, float value, size_t n, void *stream);
/* p[i] *= value (the data-parallel loss normaliser: count_all(w != 0) summed over ranks on the stream, then / world) */
int dl3_scale(float *p, float value, size_t n, void *stream);
/* Keras Adam with decay (notebook cell 2): lr_t is computed on the host;
 * g is first multiplied by grad_scale (1/world_size after the RCCL sum) */
int dl3_adam_step(float *p, const float *g, float *m, float *v, size_t n, float lr_t, float beta1, float beta2,
                  float eps, float grad_scale, void *stream);
/* the same with the scale finished on the device: g is multiplied by grad_scale / max(denom[0], 1e-20).  Data-parallel
 * step without a host round trip (round 5; utils.py:209-211 merges the towers and evaluates ONE loss over the global
 * batch): every rank differentiates sum_shard(l*w) / c0 with a FIXED c0, its count(w != 0) and loss sum ride behind the
 * gradients in the same all-reduce, denom = the summed count, grad_scale = c0. */
int dl3_adam_step_norm(float *p, const float *g, float *m, float *v, size_t n, float lr_t, float beta1, float beta2,
                       float eps, float grad_scale, const float *denom, void *stream);

/* ---- either side of the network: targets in, metric counts out -------------------------- */
#define DL3_LABEL_U8 0  /* cv2.imread(path, 0) label maps (utils.py:314) */
#define DL3_LABEL_I32 1 /* label.astype('int32') (utils.py:371) */
/* Tensor contract of SegmentationGenerator.__getitem__ (utils.py:375-402) from raw label maps labels[B][HW]:
 *   y = label, values > C-1 (and negative ones) -> C ("void")            utils.py:377
 *   Y[B][HW]  (nullable) = y as float (the [B,HW,1] target of train_on_batch)   utils.py:379
 *   SW[B][HW] (nullable) = per-image 'balanced' class weight n_valid / (n_present * count[y]) evaluated in double and
 *                          rounded to float; 0 on void pixels                  utils.py:391-400
 *   hist[B][C+1] (required, int32) = per-image label histogram (bin C = void), also the scratch of the op.
 * C <= 255.  Integer counting with int atomics: results are exact and run-to-run identical. */
int dl3_prepare_targets(const void *labels, int label_dtype, int B, int HW, int C, float *Y, float *SW, int *hist,
                        void *stream);
/* Pixel counts behind Jaccard / sparse_accuracy_ignoring_last_label (utils.py:132-157), per image b and class c:
 *   counts[b][0][c] = #(y_true == c), counts[b][1][c] = #(pred == c), counts[b][2][c] = #(y_true == c && pred == c)
 * pred = dl3_argmax output (int32), y_true as fed to the loss (float, void = C).  The ratios stay on the host
 * (utils.Jaccard_from_counts): union = true + pred - inter, exactly the reference's inter/union sums. */
int dl3_seg_counts(const int *pred, const float *y_true, int B, int HW, int C, int *counts, void *stream);

/* ---- data-parallel gradient exchange over RCCL / xGMI (replaces keras.utils.multi_gpu_model, utils.py:209-211) ----
 * One process per GPU.  Rank 0 draws a 128-byte id (dl3_comm_unique_id) and hands it to the other ranks over any host
 * channel; every rank then calls dl3_comm_init with its HIP device current.  The collectives are enqueued on the
 * caller's stream (no sync): ONE all-reduce(sum) of the flat fp32 gradient arena per step, the 1/world factor goes
 * into dl3_adam_step(grad_scale).  RCCL is bound with dlopen at the first call: -4 (DL3_EUNSUPPORTED) if absent. */
#define DL3_COMM_ID_BYTES 128
int dl3_comm_unique_id(void *id128);
int dl3_comm_init(void **comm, const void *id128, int rank, int world);
int dl3_comm_allreduce_f32(void *comm, const float *send, float *recv, size_t n, void *stream);
int dl3_comm_broadcast_f32(void *comm, float *buf, size_t n, int root, void *stream);
/* number of ranks RCCL itself reports for the communicator (ncclCommCount): lets a benchmark line state which data
 * plane produced it */
int dl3_comm_count(void *comm, int *count);
int dl3_comm_destroy(void *comm);

#ifdef __cplusplus
}
#endif
#endif
