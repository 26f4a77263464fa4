// bn.hip — BatchNormalization folding kernels, deterministic partial reductions and the
// element-wise glue (Add / materialise / gradient finish) of the DeepLabV3+ path.
//
// BatchNorm never runs as its own pass over an activation: the producer conv reduces
// sum / sum-of-squares partials in its epilogue, dl3_bn_finalize folds them into
// (scale, shift) and every consumer applies act(scale*x+shift) on load.  The backward pass
// mirrors it: consumers' bwd-data epilogues reduce (sum g, sum g*x_hat), dl3_bn_bwd_finalize
// folds them into the three per-channel coefficients of dY = cA*g + cB*y + cC that the
// producer's backward kernels apply on load.
// Reference layers: BatchNormalization at deeplabv3p.py:75,:80,:146,:178,:189,:197,:286,:290,
// :322,:379,:386,:408,:422; training semantics [TF 1.13 FusedBatchNorm]: biased batch variance
// for normalisation, Bessel-corrected variance into moving_variance.
#include "common.h"

namespace {

constexpr int FOLD_U = 8;

// sum over P partial rows of two interleaved values per channel; 8 channels x 32 row lanes
// per block, fixed order, double accumulation.  Result valid for threads with pl == 0.
__device__ __forceinline__ void fold2(const float *part, int P, int ldc, int c, bool cok, double &o1, double &o2,
                                      double *red /* [32][8][2] */) {
  const int cl = threadIdx.x & 7, pl = threadIdx.x >> 3;
  double s1 = 0.0, s2 = 0.0;
  if (cok) {
    // FOLD_U rows per trip: their loads (clamped row index, always in bounds) are issued together and added in row
    // order afterwards — the same sum as a one-row loop, without one exposed L2 round trip per row (P/32 = 8-43 of
    // them made these launches 8-10 us each; a rows-past-the-end slot adds +0.0)
    for (int p0 = pl; p0 < P; p0 += 32 * FOLD_U) {
      float2 v[FOLD_U];
#pragma unroll
      for (int u = 0; u < FOLD_U; u++) {
        const int p = p0 + 32 * u;
        v[u] = *reinterpret_cast<const float2 *>(part + ((size_t)min(p, P - 1) * ldc + c) * 2);
        if (p >= P) v[u] = make_float2(0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < FOLD_U; u++) {
        s1 += (double)v[u].x;
        s2 += (double)v[u].y;
      }
    }
  }
  red[(pl * 8 + cl) * 2 + 0] = s1;
  red[(pl * 8 + cl) * 2 + 1] = s2;
  __syncthreads();
  o1 = 0.0;
  o2 = 0.0;
  if (pl == 0)
    for (int q = 0; q < 32; q++) {
      o1 += red[(q * 8 + cl) * 2 + 0];
      o2 += red[(q * 8 + cl) * 2 + 1];
    }
}

__global__ __launch_bounds__(256) void bn_finalize_kernel(const float *__restrict__ part, int P, int ldc, int C,
                                                          double count, const float *__restrict__ gamma,
                                                          const float *__restrict__ beta, float eps,
                                                          float momentum, double unbias, float *scale, float *shift,
                                                          float *mean, float *invstd, float *mmean, float *mvar) {
  __shared__ double red[32 * 8 * 2];
  const int c = blockIdx.x * 8 + (threadIdx.x & 7);
  const bool cok = c < C;
  double s1, s2;
  // per-channel operands requested before the fold (its barrier keeps them there): their round trips overlap the fold's
  // (unconditional, clamped index, a stand-in pointer when there are no moving statistics: a predicated load gets its
  // own exec-masked block and an s_waitcnt vmcnt(0) behind it)
  const bool fin = (threadIdx.x >> 3) == 0 && cok;
  const int cc = min(c, C - 1);
  const float ga = gamma[cc], be = beta[cc];
  const float mm0 = (mmean ? mmean : gamma)[cc], mv0 = (mmean ? mvar : gamma)[cc];
  fold2(part, P, ldc, c, cok, s1, s2, red);
  if (fin) {
    const double m = s1 / count;
    double var = s2 / count - m * m;
    if (var < 0.0) var = 0.0;
    const double is = 1.0 / sqrt(var + (double)eps);
    const double sc = (double)ga * is;
    scale[c] = (float)sc;
    shift[c] = (float)((double)be - m * sc);
    mean[c] = (float)m;
    invstd[c] = (float)is;
    if (mmean) {
      const double unb = var * unbias;  // the framework's sample-variance factor (see dl3_bn_finalize in dl3.h)
      mmean[c] = (float)((double)momentum * mm0 + (1.0 - (double)momentum) * m);
      mvar[c] = (float)((double)momentum * mv0 + (1.0 - (double)momentum) * unb);
    }
  }
}

// Batch statistics of a SMALL tensor straight from its values, two passes in double (mean, then sum of squared
// deviations): sum(y^2)/n - mean^2 in fp32 partials loses everything when a channel's |mean| >> its spread — the
// image-pooling BatchNorm (deeplabv3p.py:375-379) normalises ONE value per image, at B = 2 the variance of two nearly
// equal numbers (measured: mean^2/var up to 8.6e6, relative variance error 0.8 with the partial sums).
__global__ __launch_bounds__(256) void bn_finalize_direct_kernel(const float *__restrict__ y, int ldy, int M, int C,
                                                                 const float *__restrict__ gamma,
                                                                 const float *__restrict__ beta, float eps,
                                                                 float momentum, double unbias, float *scale,
                                                                 float *shift, float *mean, float *invstd, float *mmean,
                                                                 float *mvar) {
  __shared__ double red[32 * 8];
  __shared__ double mu[8];
  const int cl = threadIdx.x & 7, pl = threadIdx.x >> 3;
  const int c = blockIdx.x * 8 + cl;
  const bool cok = c < C;
  double s = 0.0;
  if (cok)
    for (int m = pl; m < M; m += 32) s += (double)y[(size_t)m * ldy + c];
  red[pl * 8 + cl] = s;
  __syncthreads();
  if (pl == 0) {
    double t = 0.0;
    for (int q = 0; q < 32; q++) t += red[q * 8 + cl];
    mu[cl] = t / (double)M;
  }
  __syncthreads();
  const double m0 = mu[cl];
  s = 0.0;
  if (cok)
    for (int m = pl; m < M; m += 32) {
      const double d = (double)y[(size_t)m * ldy + c] - m0;
      s += d * d;
    }
  __syncthreads();
  red[pl * 8 + cl] = s;
  __syncthreads();
  if (pl == 0 && cok) {
    double t = 0.0;
    for (int q = 0; q < 32; q++) t += red[q * 8 + cl];
    const double var = t / (double)M;
    const double is = 1.0 / sqrt(var + (double)eps);
    const double sc = (double)gamma[c] * is;
    scale[c] = (float)sc;
    shift[c] = (float)((double)beta[c] - m0 * sc);
    mean[c] = (float)m0;
    invstd[c] = (float)is;
    if (mmean) {
      mmean[c] = (float)((double)momentum * mmean[c] + (1.0 - (double)momentum) * m0);
      mvar[c] = (float)((double)momentum * mvar[c] + (1.0 - (double)momentum) * var * unbias);
    }
  }
}

__global__ __launch_bounds__(256) void bn_frozen_kernel(const float *__restrict__ gamma,
                                                        const float *__restrict__ beta,
                                                        const float *__restrict__ mmean,
                                                        const float *__restrict__ mvar, float eps, int C,
                                                        float *scale, float *shift, float *mean, float *invstd) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c < C) {
    const double is = 1.0 / sqrt((double)mvar[c] + (double)eps);
    const double sc = (double)gamma[c] * is;
    scale[c] = (float)sc;
    shift[c] = (float)((double)beta[c] - (double)mmean[c] * sc);
    if (mean) mean[c] = mmean[c];
    if (invstd) invstd[c] = (float)is;
  }
}

// the same with the mean subtraction moved into the PRODUCER: neg_mean = -moving_mean goes into the bias slot of the
// convolution that writes the tensor, which then stores y - mean; consumers read scale * (y - mean) + beta
__global__ __launch_bounds__(256) void bn_frozen_centered_kernel(const float *__restrict__ gamma,
                                                                 const float *__restrict__ beta,
                                                                 const float *__restrict__ mmean,
                                                                 const float *__restrict__ mvar, float eps, int C,
                                                                 float *scale, float *shift, float *mean, float *invstd,
                                                                 float *neg_mean) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c < C) {
    const double is = 1.0 / sqrt((double)mvar[c] + (double)eps);
    scale[c] = (float)((double)gamma[c] * is);
    shift[c] = beta[c];
    neg_mean[c] = -mmean[c];
    if (mean) mean[c] = 0.f;
    if (invstd) invstd[c] = (float)is;
  }
}

__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float *__restrict__ part, int P, int ldc, int C,
                                                              double count, const float *__restrict__ gamma,
                                                              const float *__restrict__ mean,
                                                              const float *__restrict__ invstd, int batch_mode,
                                                              float *cA, float *cB, float *cC, float *dgamma,
                                                              float *dbeta) {
  __shared__ double red[32 * 8 * 2];
  const int c = blockIdx.x * 8 + (threadIdx.x & 7);
  const bool cok = c < C;
  double s1, s2;
  const bool fin = (threadIdx.x >> 3) == 0 && cok;  // operands requested before the fold (see bn_finalize_kernel)
  const int cc = min(c, C - 1);
  const float ga = gamma[cc], is0 = invstd[cc], mu0 = mean[cc];
  fold2(part, P, ldc, c, cok, s1, s2, red);
  if (fin) {
    // s1 = sum g = dbeta ; s2 = sum g * x_hat = dgamma
    if (dbeta) dbeta[c] = (float)s1;
    if (dgamma) dgamma[c] = (float)s2;
    const double a = (double)ga * (double)is0;
    if (batch_mode) {
      // dy = a*(g - s1/M - x_hat*s2/M), x_hat = (y - mean)*invstd
      const double b = -a * (double)is0 * s2 / count;
      cA[c] = (float)a;
      cB[c] = (float)b;
      cC[c] = (float)(-a * s1 / count - b * (double)mu0);
    } else {
      cA[c] = (float)a;
      cB[c] = 0.f;
      cC[c] = 0.f;
    }
  }
}

// out[i] = sum_p part[p][i]; COLS columns x PL row lanes per block
template <int COLS, int PL>
__device__ __forceinline__ void reduce_partials_block(const float *__restrict__ part, int P, int n,
                                                      float *__restrict__ out, int block, float *red /* [256] */) {
  static_assert(COLS * PL == 256, "256 threads");
  const int cl = threadIdx.x % COLS, pl = threadIdx.x / COLS;
  const long col = (long)block * COLS + cl;
  float s = 0.f;
  if (col < n)
    for (int p0 = pl; p0 < P; p0 += PL * FOLD_U) {  // loads of FOLD_U rows in flight, added in row order (see fold2)
      float v[FOLD_U];
#pragma unroll
      for (int u = 0; u < FOLD_U; u++) {
        const int p = p0 + PL * u;
        v[u] = part[(size_t)min(p, P - 1) * n + col];
        if (p >= P) v[u] = 0.f;
      }
#pragma unroll
      for (int u = 0; u < FOLD_U; u++) s += v[u];
    }
  red[pl * COLS + cl] = s;
  __syncthreads();
  if (pl == 0 && col < n) {
    float t = 0.f;
    for (int q = 0; q < PL; q++) t += red[q * COLS + cl];
    out[col] = t;
  }
}

template <int COLS, int PL>
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float *__restrict__ part, int P, int n,
                                                              float *__restrict__ out) {
  __shared__ float red[256];
  reduce_partials_block<COLS, PL>(part, P, n, out, blockIdx.x, red);
}

// columns per workgroup of a fold over P partial rows (few rows: wide blocks; many rows: more row lanes)
__host__ __device__ inline int reduce_cols(int P) { return P <= 32 ? 64 : (P <= 256 ? 32 : 8); }

// several folds in one launch: desc[m] = {partial, out, P, n, first workgroup}
__global__ __launch_bounds__(256) void reduce_partials_batched_kernel(const long long *__restrict__ desc, int m) {
  __shared__ float red[256];
  int e = 0;
  while (e + 1 < m && (long long)blockIdx.x >= desc[(e + 1) * 5 + 4]) ++e;  // m is a few dozen
  const float *part = (const float *)desc[e * 5 + 0];
  float *out = (float *)desc[e * 5 + 1];
  const int P = (int)desc[e * 5 + 2], n = (int)desc[e * 5 + 3], block = blockIdx.x - (int)desc[e * 5 + 4];
  const int cols = reduce_cols(P);
  if (cols == 64) reduce_partials_block<64, 4>(part, P, n, out, block, red);
  else if (cols == 32) reduce_partials_block<32, 8>(part, P, n, out, block, red);
  else reduce_partials_block<8, 32>(part, P, n, out, block, red);
}

// out = act_a(sa*a+ta) + act_b(sb*b+tb), optional dropout; one thread per element, channel fastest
__global__ __launch_bounds__(256) void affine_add_kernel(const float *__restrict__ a, int lda,
                                                         const float *__restrict__ sa,
                                                         const float *__restrict__ ta, int act_a,
                                                         const float *__restrict__ b, int ldb,
                                                         const float *__restrict__ sb,
                                                         const float *__restrict__ tb, int act_b,
                                                         float *__restrict__ out, int ldo, long M, int C,
                                                         float drop_rate, unsigned long long seed,
                                                         const unsigned long long *__restrict__ step) {
  const long total = M * C;
  seed = dl3_step_seed(seed, step);
  const float keep_scale = drop_rate > 0.f ? 1.f / (1.f - drop_rate) : 1.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long m = i / C;
    const int c = (int)(i - m * C);
    float v = a[(size_t)m * lda + c];
    if (sa) v = sa[c] * v + ta[c];
    v = dl3_act(v, act_a);
    if (b) {
      float u = b[(size_t)m * ldb + c];
      if (sb) u = sb[c] * u + tb[c];
      v += dl3_act(u, act_b);
    }
    if (drop_rate > 0.f) v = (dl3_uniform(seed, (unsigned long long)i) >= drop_rate) ? v * keep_scale : 0.f;
    out[(size_t)m * ldo + c] = v;
  }
}

// gout = mask_act(gin*dropmask) + add ; stats partial [gridDim.y][C][2]; 32 columns x 8 row lanes
__global__ __launch_bounds__(256) void grad_finish_kernel(const float *gin, int ldgin, int gin_div,
                                                          float gin_scale, float *gout, int ldgout,
                                                          const float *add, int ldadd,
                                                          const float *__restrict__ xraw, int ldx,
                                                          const float *__restrict__ scale,
                                                          const float *__restrict__ shift, int act,
                                                          const float *__restrict__ mean,
                                                          const float *__restrict__ invstd,
                                                          float *__restrict__ part, long M, int C, float drop_rate,
                                                          unsigned long long seed,
                                                          const unsigned long long *__restrict__ step) {
  __shared__ float red[256 * 2];
  seed = dl3_step_seed(seed, step);
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  const bool cok = c < C;
  const float keep_scale = drop_rate > 0.f ? 1.f / (1.f - drop_rate) : 1.f;
  float es = 1.f, et = 0.f, mu = 0.f, is = 0.f;
  if (cok) {
    if (scale) { es = scale[c]; et = shift[c]; }
    if (mean) { mu = mean[c]; is = invstd[c]; }
  }
  float s1 = 0.f, s2 = 0.f;
  const int cc = min(c, C - 1);
  const long stride = (long)gridDim.y * 8;
  // 4 rows per iteration, every load issued before the first use (clamped addresses, no branches)
  for (long m0 = (long)blockIdx.y * 8 + rl; m0 < M; m0 += 4 * stride) {
    float gv[4], xv[4], av[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const long m = min(m0 + q * stride, M - 1);
      gv[q] = gin[(size_t)(gin_div > 1 ? m / gin_div : m) * ldgin + cc];
      xv[q] = xraw ? xraw[(size_t)m * ldx + cc] : 0.f;
      av[q] = add ? add[(size_t)m * ldadd + cc] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const long m = m0 + q * stride;
      float v = gin_scale * gv[q];
      if (drop_rate > 0.f)
        v = (dl3_uniform(seed, (unsigned long long)(min(m, M - 1) * C + cc)) >= drop_rate) ? v * keep_scale : 0.f;
      if (xraw) v *= dl3_act_mask(es * xv[q] + et, act);
      v += av[q];
      if (cok && m < M) {
        gout[(size_t)m * ldgout + c] = v;
        s1 += v;
        s2 += v * ((xv[q] - mu) * is);
      }
    }
  }
  if (part) {
    red[threadIdx.x * 2] = s1;
    red[threadIdx.x * 2 + 1] = s2;
    __syncthreads();
    if (rl == 0 && cok) {
      float a1 = 0.f, a2 = 0.f;
      for (int q = 0; q < 8; q++) {
        a1 += red[(q * 32 + cl) * 2];
        a2 += red[(q * 32 + cl) * 2 + 1];
      }
      part[((size_t)blockIdx.y * C + c) * 2] = a1;
      part[((size_t)blockIdx.y * C + c) * 2 + 1] = a2;
    }
  }
}

// out[n][c] = out_scale * sum_hw T(x)[n,hw,c]; grid (ceil(C/32), N); 32 columns x RL row lanes (RL = 8: 256 threads; RL = 32:
// 1 024 threads for small batches — a (column block, image) pair is ONE workgroup, at B = 2 the launch is 20 workgroups and
// its time is the chain of dependent round trips one lane walks: 512 rows per lane at RL = 8, 128 at RL = 32)
template <int RL>
__global__ __launch_bounds__(32 * RL) void gap_kernel(const float *__restrict__ x, int ldx,
                                                  const float *__restrict__ sc, const float *__restrict__ sh,
                                                  int act, float *__restrict__ out, int HW, int C, float out_scale) {
  // Sums in DOUBLE (round 5): the pooled feature is ONE value per image and channel that the ASPP adds to every pixel
  // (deeplabv3p.py:375-382) — its rounding error is coherent over the whole map, and the per-layer distance to float64
  // of the Xception OS=8 inference (tools/r5/xception_layer_distance.py) parted from torch-fp32's exactly here.  4 096
  // terms per (image, channel): the double adds are free next to the loads.
  __shared__ double red[32 * RL];
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl, n = blockIdx.y;
  double s = 0.0;
  if (c < C) {
    float es = 1.f, et = 0.f;
    if (sc) { es = sc[c]; et = sh[c]; }
    const float *p = x + (size_t)n * HW * ldx + c;
    int i = rl;
    // 16 independent loads per iteration: a (column block, image) pair is ONE workgroup (20-40 workgroups in all, the
    // kernel is pure latency: 72 us at 4 loads in flight per lane, measured on the 64x64x320 map)
    for (; i + 15 * RL < HW; i += 16 * RL) {
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; u++) v[u] = p[(size_t)(i + RL * u) * ldx];
      double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
      for (int u = 0; u < 16; u += 4) {
        a0 += (double)dl3_act(es * v[u] + et, act);
        a1 += (double)dl3_act(es * v[u + 1] + et, act);
        a2 += (double)dl3_act(es * v[u + 2] + et, act);
        a3 += (double)dl3_act(es * v[u + 3] + et, act);
      }
      s += (a0 + a1) + (a2 + a3);
    }
    for (; i < HW; i += RL) s += (double)dl3_act(es * p[(size_t)i * ldx] + et, act);
  }
  red[threadIdx.x] = s;
  __syncthreads();
  if (rl == 0 && c < C) {
    double t = 0.0;
    for (int q = 0; q < RL; q++) t += red[q * 32 + cl];
    out[(size_t)n * C + c] = (float)(t * (double)out_scale);
  }
}

// y[n,oy,ox,c] = T(x)[n, oy*s, ox*s, c]; grid (ceil(Wo*C/256), N*Ho)
__global__ __launch_bounds__(256) void subsample_fwd_kernel(const float *__restrict__ x, int ldx,
                                                            const float *__restrict__ sc,
                                                            const float *__restrict__ sh, int act,
                                                            float *__restrict__ y, int H, int W, int C, int stride,
                                                            int Ho, int Wo) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= Wo * C) return;
  const int ox = i / C, c = i - ox * C;
  const int n = blockIdx.y / Ho, oy = blockIdx.y - n * Ho;
  float v = x[(((size_t)n * H + (size_t)oy * stride) * W + (size_t)ox * stride) * ldx + c];
  if (sc) v = sc[c] * v + sh[c];
  y[((size_t)blockIdx.y * Wo) * C + i] = dl3_act(v, act);
}

// dx[n,iy,ix,c] = g[n,iy/s,ix/s,c] if iy%s==0 && ix%s==0 else 0; grid (ceil(W*C/256), N*H)
__global__ __launch_bounds__(256) void subsample_bwd_kernel(const float *__restrict__ g, float *__restrict__ dx,
                                                            int H, int W, int C, int stride, int Ho, int Wo) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= W * C) return;
  const int ix = i / C, c = i - ix * C;
  const int n = blockIdx.y / H, iy = blockIdx.y - n * H;
  float v = 0.f;
  if (iy % stride == 0 && ix % stride == 0 && iy / stride < Ho && ix / stride < Wo)
    v = g[(((size_t)n * Ho + iy / stride) * Wo + ix / stride) * C + c];
  dx[((size_t)blockIdx.y * W) * C + i] = v;
}

__global__ __launch_bounds__(256) void fill_kernel(float *p, float v, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = v;
}

__global__ __launch_bounds__(256) void scale_kernel(float *p, float v, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] *= v;
}

// Keras Adam (notebook cell 2): p -= lr_t * m / (sqrt(v) + eps), bias correction folded into lr_t
__global__ __launch_bounds__(256) void adam_kernel(float *__restrict__ p, const float *__restrict__ g,
                                                   float *__restrict__ m, float *__restrict__ v, size_t n,
                                                   float lr_t, float b1, float b2, float eps, float gs) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float gi = g[i] * gs;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] -= lr_t * mi / (sqrtf(vi) + eps);
  }
}

// the same with the gradient scale finished on the device: g * gs / max(denom[0], floor) — data parallel: the arena holds
// sum over ranks of (sum over the shard of dloss / c0), denom the global count(w != 0) that travelled in the same
// all-reduce, gs = c0: the optimizer sees the gradient of ONE loss over the global batch with no host round trip
__global__ __launch_bounds__(256) void adam_norm_kernel(float *__restrict__ p, const float *__restrict__ g,
                                                        float *__restrict__ m, float *__restrict__ v, size_t n,
                                                        float lr_t, float b1, float b2, float eps, float gs,
                                                        const float *__restrict__ denom) {
  const float sc = gs / fmaxf(denom[0], 1e-20f);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float gi = g[i] * sc;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] -= lr_t * mi / (sqrtf(vi) + eps);
  }
}

// float4 variant: C % 4 == 0, all leading dimensions % 4 == 0, fewer than 2^31 elements
__global__ __launch_bounds__(256) void affine_add4_kernel(const float *__restrict__ a, int lda,
                                                          const float *__restrict__ sa,
                                                          const float *__restrict__ ta, int act_a,
                                                          const float *__restrict__ b, int ldb,
                                                          const float *__restrict__ sb,
                                                          const float *__restrict__ tb, int act_b,
                                                          float *__restrict__ out, int ldo, unsigned total4,
                                                          unsigned C4, float drop_rate, unsigned long long seed,
                                                          const unsigned long long *__restrict__ step) {
  seed = dl3_step_seed(seed, step);
  const float keep_scale = drop_rate > 0.f ? 1.f / (1.f - drop_rate) : 1.f;
  for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < total4; i += gridDim.x * 256) {
    const unsigned m = i / C4, c = (i - m * C4) * 4;
    f32x4 v = ld4(a + (size_t)m * lda + c);
    if (sa) v = ld4(sa + c) * v + ld4(ta + c);
    v = dl3_act4(v, act_a);
    if (b) {
      f32x4 u = ld4(b + (size_t)m * ldb + c);
      if (sb) u = ld4(sb + c) * u + ld4(tb + c);
      v += dl3_act4(u, act_b);
    }
    if (drop_rate > 0.f) {
      const unsigned long long e0 = (unsigned long long)m * (C4 * 4) + c;
#pragma unroll
      for (int j = 0; j < 4; j++) v[j] = (dl3_uniform(seed, e0 + j) >= drop_rate) ? v[j] * keep_scale : 0.f;
    }
    st4(out + (size_t)m * ldo + c, v);
  }
}

__global__ void counter_add_kernel(unsigned long long *c, unsigned long long inc) { *c += inc; }

inline int ew_blocks(size_t n) {
  size_t b = (n + 255) / 256;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

extern "C" int dl3_rows_partials(int M) {
  int p = M / 64;
  if (p < 1) p = 1;
  if (p > 512) p = 512;
  return p;
}

extern "C" int dl3_bn_finalize(const float *stat_partial, int P, int ldc, int C, double count, const float *gamma,
                               const float *beta, float eps, float momentum, double var_unbias, float *scale,
                               float *shift, float *mean, float *invstd, float *moving_mean, float *moving_var,
                               void *stream) {
  DL3_CHECK_ARG(stat_partial && gamma && beta && scale && shift && mean && invstd, "bn_finalize: null pointer");
  DL3_CHECK_ARG(P > 0 && C > 0 && ldc >= C && count > 0, "bn_finalize: bad dimension");
  DL3_CHECK_ARG((moving_mean == nullptr) == (moving_var == nullptr), "bn_finalize: moving stats come together");
  DL3_CHECK_ARG(var_unbias > 0.0, "bn_finalize: var_unbias must be positive");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(dl3_cdiv(C, 8)), dim3(256), 0, (hipStream_t)stream, stat_partial, P,
                     ldc, C, count, gamma, beta, eps, momentum, var_unbias, scale, shift, mean, invstd, moving_mean,
                     moving_var);
  DL3_LAUNCH_CHECK("bn_finalize");
  return DL3_OK;
}

extern "C" int dl3_bn_finalize_direct(const float *y, int ldy, int M, int C, const float *gamma, const float *beta,
                                      float eps, float momentum, double var_unbias, float *scale, float *shift,
                                      float *mean, float *invstd, float *moving_mean, float *moving_var, void *stream) {
  DL3_CHECK_ARG(y && gamma && beta && scale && shift && mean && invstd, "bn_finalize_direct: null pointer");
  DL3_CHECK_ARG(M > 0 && C > 0 && ldy >= C, "bn_finalize_direct: bad dimension");
  DL3_CHECK_ARG((moving_mean == nullptr) == (moving_var == nullptr), "bn_finalize_direct: moving stats come together");
  DL3_CHECK_ARG(var_unbias > 0.0, "bn_finalize_direct: var_unbias must be positive");
  hipLaunchKernelGGL(bn_finalize_direct_kernel, dim3(dl3_cdiv(C, 8)), dim3(256), 0, (hipStream_t)stream, y, ldy, M, C,
                     gamma, beta, eps, momentum, var_unbias, scale, shift, mean, invstd, moving_mean, moving_var);
  DL3_LAUNCH_CHECK("bn_finalize_direct");
  return DL3_OK;
}

extern "C" int dl3_bn_frozen(const float *gamma, const float *beta, const float *moving_mean,
                             const float *moving_var, float eps, int C, float *scale, float *shift, float *mean,
                             float *invstd, void *stream) {
  DL3_CHECK_ARG(gamma && beta && moving_mean && moving_var && scale && shift && C > 0, "bn_frozen: bad argument");
  hipLaunchKernelGGL(bn_frozen_kernel, dim3(dl3_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, gamma, beta,
                     moving_mean, moving_var, eps, C, scale, shift, mean, invstd);
  DL3_LAUNCH_CHECK("bn_frozen");
  return DL3_OK;
}

extern "C" int dl3_bn_frozen_centered(const float *gamma, const float *beta, const float *moving_mean,
                                      const float *moving_var, float eps, int C, float *scale, float *shift, float *mean,
                                      float *invstd, float *neg_mean, void *stream) {
  DL3_CHECK_ARG(gamma && beta && moving_mean && moving_var && scale && shift && neg_mean && C > 0,
                "bn_frozen_centered: bad argument");
  hipLaunchKernelGGL(bn_frozen_centered_kernel, dim3(dl3_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, gamma, beta,
                     moving_mean, moving_var, eps, C, scale, shift, mean, invstd, neg_mean);
  DL3_LAUNCH_CHECK("bn_frozen_centered");
  return DL3_OK;
}

extern "C" int dl3_bn_bwd_finalize(const float *dstat_partial, int P, int ldc, int C, double count,
                                   const float *gamma, const float *mean, const float *invstd, int batch_mode,
                                   float *cA, float *cB, float *cC, float *dgamma, float *dbeta, void *stream) {
  DL3_CHECK_ARG(dstat_partial && gamma && mean && invstd && cA && cB && cC, "bn_bwd_finalize: null pointer");
  DL3_CHECK_ARG(P > 0 && C > 0 && ldc >= C && count > 0, "bn_bwd_finalize: bad dimension");
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(dl3_cdiv(C, 8)), dim3(256), 0, (hipStream_t)stream,
                     dstat_partial, P, ldc, C, count, gamma, mean, invstd, batch_mode, cA, cB, cC, dgamma, dbeta);
  DL3_LAUNCH_CHECK("bn_bwd_finalize");
  return DL3_OK;
}

extern "C" int dl3_reduce_partials(const float *partial, int P, int n, float *out, void *stream) {
  DL3_CHECK_ARG(partial && out && P > 0 && n > 0, "reduce_partials: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const int cols = reduce_cols(P);
  if (cols == 64)
    hipLaunchKernelGGL((reduce_partials_kernel<64, 4>), dim3(dl3_cdiv(n, 64)), dim3(256), 0, st, partial, P, n, out);
  else if (cols == 32)
    hipLaunchKernelGGL((reduce_partials_kernel<32, 8>), dim3(dl3_cdiv(n, 32)), dim3(256), 0, st, partial, P, n, out);
  else
    hipLaunchKernelGGL((reduce_partials_kernel<8, 32>), dim3(dl3_cdiv(n, 8)), dim3(256), 0, st, partial, P, n, out);
  DL3_LAUNCH_CHECK("reduce_partials");
  return DL3_OK;
}

extern "C" int dl3_reduce_partials_blocks(int P, int n) {
  if (P <= 0 || n <= 0) return 0;
  return dl3_cdiv(n, reduce_cols(P));
}

extern "C" int dl3_reduce_partials_batched(const long long *desc, int m, int total_blocks, void *stream) {
  DL3_CHECK_ARG(desc && m > 0 && total_blocks > 0, "reduce_partials_batched: bad argument");
  hipLaunchKernelGGL(reduce_partials_batched_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, desc, m);
  DL3_LAUNCH_CHECK("reduce_partials_batched");
  return DL3_OK;
}

extern "C" int dl3_affine_add(const float *a, int lda, const float *sa, const float *ta, int act_a,
                              const float *b, int ldb, const float *sb, const float *tb, int act_b, float *out,
                              int ldo, int M, int C, float drop_rate, unsigned long long drop_seed,
                              const unsigned long long *drop_step, void *stream) {
  DL3_CHECK_ARG(a && out && M > 0 && C > 0, "affine_add: bad argument");
  DL3_CHECK_ARG((sa == nullptr) == (ta == nullptr) && (sb == nullptr) == (tb == nullptr),
                "affine_add: scale/shift must come together");
  DL3_CHECK_ARG(drop_rate >= 0.f && drop_rate < 1.f, "affine_add: drop_rate must be in [0,1)");
  const bool v4 = C % 4 == 0 && lda % 4 == 0 && ldo % 4 == 0 && (!b || ldb % 4 == 0) && (size_t)M * C < (1ull << 31) &&
                  (((uintptr_t)a | (uintptr_t)out | (uintptr_t)b | (uintptr_t)sa | (uintptr_t)ta | (uintptr_t)sb |
                    (uintptr_t)tb) & 15) == 0;
  if (v4)
    hipLaunchKernelGGL(affine_add4_kernel, dim3(ew_blocks((size_t)M * C / 4)), dim3(256), 0, (hipStream_t)stream, a,
                       lda, sa, ta, act_a, b, ldb, sb, tb, act_b, out, ldo, (unsigned)((size_t)M * C / 4),
                       (unsigned)(C / 4), drop_rate, drop_seed, drop_step);
  else
    hipLaunchKernelGGL(affine_add_kernel, dim3(ew_blocks((size_t)M * C)), dim3(256), 0, (hipStream_t)stream, a, lda,
                       sa, ta, act_a, b, ldb, sb, tb, act_b, out, ldo, (long)M, C, drop_rate, drop_seed, drop_step);
  DL3_LAUNCH_CHECK("affine_add");
  return DL3_OK;
}

extern "C" int dl3_grad_finish(const float *gin, int ldgin, int gin_div, float gin_scale, float *gout, int ldgout,
                               const float *add, int ldadd,
                               const float *xraw, int ldx, const float *scale, const float *shift, int act,
                               const float *mean, const float *invstd, float *dstat_partial, int M, int C,
                               float drop_rate, unsigned long long drop_seed, const unsigned long long *drop_step,
                               void *stream) {
  DL3_CHECK_ARG(gin && gout && M > 0 && C > 0, "grad_finish: bad argument");
  DL3_CHECK_ARG(act == DL3_ACT_NONE || xraw, "grad_finish: activation mask needs xraw");
  DL3_CHECK_ARG(!dstat_partial || (xraw && mean && invstd), "grad_finish: dstat needs xraw, mean, invstd");
  DL3_CHECK_ARG((scale == nullptr) == (shift == nullptr), "grad_finish: scale/shift must come together");
  dim3 grid(dl3_cdiv(C, 32), dl3_rows_partials(M));
  hipLaunchKernelGGL(grad_finish_kernel, grid, dim3(256), 0, (hipStream_t)stream, gin, ldgin, gin_div, gin_scale,
                     gout, ldgout, add,
                     ldadd, xraw, ldx, scale, shift, act, mean, invstd, dstat_partial, (long)M, C, drop_rate,
                     drop_seed, drop_step);
  DL3_LAUNCH_CHECK("grad_finish");
  return DL3_OK;
}

extern "C" int dl3_gap_fwd(const float *x, int ldx, const float *in_scale, const float *in_shift, int in_act,
                           float *out, int N, int HW, int C, float out_scale, void *stream) {
  DL3_CHECK_ARG(x && out && N > 0 && HW > 0 && C > 0 && ldx >= C, "gap_fwd: bad argument");
  DL3_CHECK_ARG((in_scale == nullptr) == (in_shift == nullptr), "gap_fwd: scale/shift must come together");
  DL3_UNSUPPORTED(N > 65535, "gap_fwd: N too large");
  const dim3 grid(dl3_cdiv(C, 32), N);
  if ((long)grid.x * grid.y >= 512)
    hipLaunchKernelGGL(gap_kernel<8>, grid, dim3(256), 0, (hipStream_t)stream, x, ldx, in_scale, in_shift, in_act, out, HW,
                       C, out_scale);
  else
    hipLaunchKernelGGL(gap_kernel<32>, grid, dim3(1024), 0, (hipStream_t)stream, x, ldx, in_scale, in_shift, in_act, out, HW,
                       C, out_scale);
  DL3_LAUNCH_CHECK("gap_fwd");
  return DL3_OK;
}

extern "C" int dl3_fill(float *p, float value, size_t n, void *stream) {
  DL3_CHECK_ARG(p && n > 0, "fill: bad argument");
  hipLaunchKernelGGL(fill_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, p, value, n);
  DL3_LAUNCH_CHECK("fill");
  return DL3_OK;
}

extern "C" int dl3_scale(float *p, float value, size_t n, void *stream) {
  DL3_CHECK_ARG(p && n > 0, "scale: bad argument");
  hipLaunchKernelGGL(scale_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, p, value, n);
  DL3_LAUNCH_CHECK("scale");
  return DL3_OK;
}

extern "C" int dl3_counter_add(unsigned long long *counter, unsigned long long inc, void *stream) {
  DL3_CHECK_ARG(counter, "counter_add: null pointer");
  hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, counter, inc);
  DL3_LAUNCH_CHECK("counter_add");
  return DL3_OK;
}

extern "C" int dl3_adam_step(float *p, const float *g, float *m, float *v, size_t n, float lr_t, float beta1,
                             float beta2, float eps, float grad_scale, void *stream) {
  DL3_CHECK_ARG(p && g && m && v && n > 0, "adam_step: bad argument");
  hipLaunchKernelGGL(adam_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr_t, beta1,
                     beta2, eps, grad_scale);
  DL3_LAUNCH_CHECK("adam_step");
  return DL3_OK;
}

extern "C" int dl3_adam_step_norm(float *p, const float *g, float *m, float *v, size_t n, float lr_t, float beta1,
                                  float beta2, float eps, float grad_scale, const float *denom, void *stream) {
  DL3_CHECK_ARG(p && g && m && v && denom && n > 0, "adam_step_norm: bad argument");
  hipLaunchKernelGGL(adam_norm_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr_t, beta1,
                     beta2, eps, grad_scale, denom);
  DL3_LAUNCH_CHECK("adam_step_norm");
  return DL3_OK;
}

extern "C" int dl3_subsample_fwd(const float *x, int ldx, const float *in_scale, const float *in_shift, int in_act,
                                 float *y, int N, int H, int W, int C, int stride, int Ho, int Wo, void *stream) {
  DL3_CHECK_ARG(x && y && N > 0 && H > 0 && W > 0 && C > 0 && stride >= 1 && Ho > 0 && Wo > 0, "subsample_fwd: bad argument");
  DL3_CHECK_ARG((Ho - 1) * stride < H && (Wo - 1) * stride < W && ldx >= C, "subsample_fwd: geometry out of range");
  DL3_CHECK_ARG((in_scale == nullptr) == (in_shift == nullptr), "subsample_fwd: scale/shift must come together");
  DL3_UNSUPPORTED((long)N * Ho > 65535L * 32768L, "subsample_fwd: too many rows");
  hipLaunchKernelGGL(subsample_fwd_kernel, dim3(dl3_cdiv(Wo * C, 256), N * Ho), dim3(256), 0, (hipStream_t)stream, x,
                     ldx, in_scale, in_shift, in_act, y, H, W, C, stride, Ho, Wo);
  DL3_LAUNCH_CHECK("subsample_fwd");
  return DL3_OK;
}

extern "C" int dl3_subsample_bwd(const float *g, float *dx, int N, int H, int W, int C, int stride, int Ho, int Wo,
                                 void *stream) {
  DL3_CHECK_ARG(g && dx && N > 0 && H > 0 && W > 0 && C > 0 && stride >= 1 && Ho > 0 && Wo > 0, "subsample_bwd: bad argument");
  hipLaunchKernelGGL(subsample_bwd_kernel, dim3(dl3_cdiv(W * C, 256), N * H), dim3(256), 0, (hipStream_t)stream, g, dx,
                     H, W, C, stride, Ho, Wo);
  DL3_LAUNCH_CHECK("subsample_bwd");
  return DL3_OK;
}
