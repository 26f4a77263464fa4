// misc.hip — the tail of the DeepLabV3+ path: TF1-legacy bilinear resize (deeplabv3p.py:382,:418,
// :439; utils.py:190), Subpixel phase shift (subpixel.py:77-88), softmax (deeplabv3p.py:441,:444),
// sparse_crossentropy_ignoring_last_label with temporal sample weights (utils.py:127-130) and argmax.
// All HBM-bound, one pass each; backward kernels are gather-form (no atomics, deterministic).
#include "common.h"

namespace {

// Keras categorical_crossentropy clips the renormalised probability with tf.clip_by_value before the log
// (utils.py:130 -> keras/backend/tensorflow_backend.py), and the gradient of clip_by_value is ZERO where its input lies
// outside [1e-7, 1 - 1e-7]: a pixel whose true-class probability has left that interval has a constant loss and hands
// no gradient to any logit; inside it the gradient is (p - onehot).  1.f inside (bounds included, as TF), 0.f outside.
__device__ __forceinline__ float dl3_clip_pass(float q) { return (q >= 1e-7f && q <= 1.f - 1e-7f) ? 1.f : 0.f; }
// The loss normaliser count(w != 0): an integer count in a single process, count_all / world under data parallelism
// (fractional, below 1 when the global count is smaller than the world): only a zero count is replaced (no weight is
// non-zero -> every term is zero anyway)
#define DL3_NNZ_FLOOR 1e-20f

// tf.image.resize_bilinear(align_corners=False) of TF 1.x: src = dst * (in/out), no half-pixel
// offset; lower = floor(src), upper = min(lower+1, in-1), lerp = src - lower.
struct Lerp {
  int lo, hi;
  float w;
};
__device__ __forceinline__ Lerp tf1_lerp(int o, float scale, int in_size) {
  const float f = (float)o * scale;
  Lerp r;
  r.lo = (int)floorf(f);
  if (r.lo > in_size - 1) r.lo = in_size - 1;
  r.hi = min(r.lo + 1, in_size - 1);
  r.w = f - (float)r.lo;
  return r;
}

__global__ __launch_bounds__(256) void resize_fwd_kernel(const float *__restrict__ x, int ldx,
                                                         const float *__restrict__ sc,
                                                         const float *__restrict__ sh, int act,
                                                         float *__restrict__ y, int ldy, int N, int Hi, int Wi,
                                                         int Ho, int Wo, int C, float sy, float sx) {
  // grid.y = output row (n, oy); threads cover (ox, c) of that row: no 64-bit divisions per element
  const int n = blockIdx.y / Ho, oy = blockIdx.y - n * Ho;
  const Lerp ly = tf1_lerp(oy, sy, Hi);
  const float *b = x + (size_t)n * Hi * Wi * ldx;
  const float *r0 = b + (size_t)ly.lo * Wi * ldx, *r1 = b + (size_t)ly.hi * Wi * ldx;
  float *yo = y + (size_t)blockIdx.y * Wo * ldy;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < Wo * C; i += gridDim.x * 256) {
    const int ox = i / C, c = i - ox * C;
    const Lerp lx = tf1_lerp(ox, sx, Wi);
    float es = 1.f, et = 0.f;
    if (sc) { es = sc[c]; et = sh[c]; }
    const float tl = dl3_act(es * r0[(size_t)lx.lo * ldx + c] + et, act);
    const float tr = dl3_act(es * r0[(size_t)lx.hi * ldx + c] + et, act);
    const float bl = dl3_act(es * r1[(size_t)lx.lo * ldx + c] + et, act);
    const float br = dl3_act(es * r1[(size_t)lx.hi * ldx + c] + et, act);
    const float top = tl + (tr - tl) * lx.w;
    const float bot = bl + (br - bl) * lx.w;
    yo[(size_t)ox * ldy + c] = top + (bot - top) * ly.w;
  }
}

// gather form of the transpose: input pixel (iy,ix) collects from every output pixel whose
// lower/upper source index hits it; weights are recomputed with the forward formula.
__global__ __launch_bounds__(256) void resize_bwd_kernel(const float *__restrict__ dy, int lddy,
                                                         float *__restrict__ dx, int lddx, int N, int Hi, int Wi,
                                                         int Ho, int Wo, int C, float sy, float sx, int accumulate) {
  // grid.y = input row (n, iy); threads cover (ix, c)
  const int n = blockIdx.y / Hi, iy = blockIdx.y - n * Hi;
  // candidate output rows: those with floor(oy*sy) in {iy-1, iy} (+1 slack each side for rounding)
  int oy0 = (int)floorf((float)(iy - 1) / sy) - 1, oy1 = (int)ceilf((float)(iy + 1) / sy) + 1;
  oy0 = max(oy0, 0);
  oy1 = min(oy1, Ho - 1);
  const float *dyn = dy + (size_t)n * Ho * Wo * lddy;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < Wi * C; i += gridDim.x * 256) {
    const int ix = i / C, c = i - ix * C;
    int ox0 = (int)floorf((float)(ix - 1) / sx) - 1, ox1 = (int)ceilf((float)(ix + 1) / sx) + 1;
    ox0 = max(ox0, 0);
    ox1 = min(ox1, Wo - 1);
    float acc = 0.f;
    for (int oy = oy0; oy <= oy1; ++oy) {
      const Lerp ly = tf1_lerp(oy, sy, Hi);
      const float wy = (ly.lo == iy ? 1.f - ly.w : 0.f) + (ly.hi == iy ? ly.w : 0.f);
      if (wy == 0.f) continue;
      float racc = 0.f;
      const float *row = dyn + (size_t)oy * Wo * lddy + c;
      for (int ox = ox0; ox <= ox1; ++ox) {
        const Lerp lx = tf1_lerp(ox, sx, Wi);
        const float wx = (lx.lo == ix ? 1.f - lx.w : 0.f) + (lx.hi == ix ? lx.w : 0.f);
        racc += wx * row[(size_t)ox * lddy];
      }
      acc += wy * racc;
    }
    float *o = dx + ((size_t)blockIdx.y * Wi + ix) * lddx + c;
    *o = accumulate ? (*o + acc) : acc;
  }
}

// separable form of the same transpose: pass X folds the output columns of every output row onto the input columns
// (tmp [N,Ho,Wi,C]), pass Y folds the output rows onto the input rows.  Each dy element is read once instead of four
// times and the per-element loop is 2f+3 long instead of (2f+3)^2.
__global__ __launch_bounds__(256) void resize_bwd_x_kernel(const float *__restrict__ dy, int lddy,
                                                           float *__restrict__ tmp, int Wi, int Wo, int C, float sx) {
  // grid.y = output row (n, oy); threads cover (ix, c)
  const float *row = dy + (size_t)blockIdx.y * Wo * lddy;
  float *trow = tmp + (size_t)blockIdx.y * Wi * C;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < Wi * C; i += gridDim.x * 256) {
    const int ix = i / C, c = i - ix * C;
    int ox0 = (int)floorf((float)(ix - 1) / sx) - 1, ox1 = (int)ceilf((float)(ix + 1) / sx) + 1;
    ox0 = max(ox0, 0);
    ox1 = min(ox1, Wo - 1);
    float acc = 0.f;
    for (int ox = ox0; ox <= ox1; ++ox) {
      const Lerp lx = tf1_lerp(ox, sx, Wi);
      const float wx = (lx.lo == ix ? 1.f - lx.w : 0.f) + (lx.hi == ix ? lx.w : 0.f);
      acc += wx * row[(size_t)ox * lddy + c];
    }
    trow[i] = acc;
  }
}
__global__ __launch_bounds__(256) void resize_bwd_y_kernel(const float *__restrict__ tmp, float *__restrict__ dx,
                                                           int lddx, int Hi, int Wi, int Ho, int C, float sy,
                                                           int accumulate) {
  // grid.y = input row (n, iy); threads cover (ix, c)
  const int n = blockIdx.y / Hi, iy = blockIdx.y - n * Hi;
  int oy0 = (int)floorf((float)(iy - 1) / sy) - 1, oy1 = (int)ceilf((float)(iy + 1) / sy) + 1;
  oy0 = max(oy0, 0);
  oy1 = min(oy1, Ho - 1);
  const float *tn = tmp + (size_t)n * Ho * Wi * C;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < Wi * C; i += gridDim.x * 256) {
    const int ix = i / C, c = i - ix * C;
    float acc = 0.f;
    for (int oy = oy0; oy <= oy1; ++oy) {
      const Lerp ly = tf1_lerp(oy, sy, Hi);
      const float wy = (ly.lo == iy ? 1.f - ly.w : 0.f) + (ly.hi == iy ? ly.w : 0.f);
      acc += wy * tn[(size_t)oy * Wi * C + i];
    }
    float *o = dx + ((size_t)blockIdx.y * Wi + ix) * lddx + c;
    *o = accumulate ? (*o + acc) : acc;
  }
}

// out[n, ia*r+q, ib*r+p, ch] = in[n, ia, ib, ch*r*r + p*r + q]   (subpixel.py:81-87)
// Generic fallback: one thread per element of the shuffled tensor (its reads are r*r floats apart: ~1 TB/s).
__global__ __launch_bounds__(256) void phase_shift_kernel(const float *__restrict__ in, float *__restrict__ out,
                                                          int N, int H, int W, int Cout, int r, int inverse) {
  const long total = (long)N * H * W * Cout * r * r;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    // i indexes the shuffled tensor [N, H*r, W*r, Cout]
    const int ch = (int)(i % Cout);
    long p = i / Cout;
    const int X = (int)(p % ((long)W * r));
    p /= (long)W * r;
    const int Y = (int)(p % ((long)H * r));
    const int n = (int)(p / ((long)H * r));
    const int ia = Y / r, q = Y % r, ib = X / r, pp = X % r;
    const size_t j = (((size_t)n * H + ia) * W + ib) * ((size_t)Cout * r * r) + (size_t)ch * r * r + pp * r + q;
    if (inverse) out[j] = in[i];
    else out[i] = in[j];
  }
}

// The same permutation with both sides coalesced (round 3; the Subpixel head moved 2.8 GB each way at 0.98 TB/s: 12 ms of
// a 135 ms step).  A workgroup owns PB consecutive pixels (ib0 .. ib0+PB-1) of one row (n, ia) of the UNshuffled
// tensor: PB * Cout*r*r contiguous floats on that side; on the shuffled side they are r runs (q = 0..r-1, image row
// ia*r+q) of PB*r*Cout contiguous floats.  The tile goes through LDS, element (pixel, ch, pq) at
// pixel*Cout*(r*r+1) + ch*(r*r+1) + pq: the channel stride is odd, so the transposing access (consecutive lanes =
// consecutive ch) touches distinct banks.
__global__ __launch_bounds__(256) void phase_shift_lds_kernel(const float *__restrict__ in, float *__restrict__ out,
                                                              int H, int W, int Cout, int r, int PB, int inverse) {
  extern __shared__ float tile[];
  const int rr = r * r, P = Cout * rr, LP = Cout * (rr + 1);
  const int wblocks = (W + PB - 1) / PB;
  const int ib0 = (blockIdx.x % wblocks) * PB;
  const long row = blockIdx.x / wblocks;  // n * H + ia
  const int ia = (int)(row % H);
  const long n = row / H;
  const int pb = min(PB, W - ib0);
  const float *flat_src = inverse ? nullptr : in + ((size_t)row * W + ib0) * P;    // unshuffled side, contiguous
  float *flat_dst = inverse ? out + ((size_t)row * W + ib0) * P : nullptr;
  const int run = pb * r * Cout;                                                    // one shuffled-side run
  const size_t run0 = (((size_t)n * H * r + (size_t)ia * r) * ((size_t)W * r) + (size_t)ib0 * r) * Cout;  // q = 0
  const size_t run_stride = (size_t)W * r * Cout;                                   // next image row (q + 1)
  if (!inverse) {
    for (int t = threadIdx.x; t < pb * P; t += 256) {
      const int px = t / P, e = t % P;
      tile[px * LP + (e / rr) * (rr + 1) + e % rr] = flat_src[t];
    }
    __syncthreads();
    for (int t = threadIdx.x; t < r * run; t += 256) {
      const int q = t / run, u = t % run;
      const int ch = u % Cout, xx = u / Cout;  // xx = ib_local * r + p
      out[run0 + (size_t)q * run_stride + u] = tile[(xx / r) * LP + ch * (rr + 1) + (xx % r) * r + q];
    }
  } else {
    for (int t = threadIdx.x; t < r * run; t += 256) {
      const int q = t / run, u = t % run;
      const int ch = u % Cout, xx = u / Cout;
      tile[(xx / r) * LP + ch * (rr + 1) + (xx % r) * r + q] = in[run0 + (size_t)q * run_stride + u];
    }
    __syncthreads();
    for (int t = threadIdx.x; t < pb * P; t += 256) {
      const int px = t / P, e = t % P;
      flat_dst[t] = tile[px * LP + (e / rr) * (rr + 1) + e % rr];
    }
  }
}

// Subpixel head, training tail in one pass (round 3): softmax cross-entropy evaluated straight on the UNshuffled output
// of the Subpixel convolution.  out[n, ia*r+q, ib*r+p, ch] = u[n, ia, ib, ch*r*r + p*r + q] (subpixel.py:81-87), so the C
// logits of one output pixel sit r*r floats apart in u: a workgroup stages PB consecutive pixels of one row (n, ia) of u
// in LDS (coalesced; element (pixel, ch, pq) at pixel*C*(r*r+1) + ch*(r*r+1) + pq as in phase_shift_lds_kernel), every
// thread evaluates output pixels (pixel, pq) from there (lanes = consecutive pq: conflict-free), writes the gradient
// back into the same LDS cells, and the tile leaves coalesced as du in u's own layout — the shuffled logits, the
// shuffled gradient and both phase-shift passes (4 x 2.8 GB at 128 x 512 x 512 x 21) never touch HBM.
template <int MAXC>
__global__ __launch_bounds__(256) void shuffle_xent_kernel(const float *__restrict__ u, const float *__restrict__ labels,
                                                           const float *__restrict__ weights,
                                                           const float *__restrict__ nnz, float *__restrict__ du,
                                                           float *__restrict__ loss_part, int H, int W, int C, int r,
                                                           int PB) {
  extern __shared__ float tile[];
  __shared__ float red[4];
  const int rr = r * r, P = C * rr, LP = C * (rr + 1);
  const int wblocks = (W + PB - 1) / PB;
  const int ib0 = (blockIdx.x % wblocks) * PB;
  const long row = blockIdx.x / wblocks;  // n * H + ia
  const int ia = (int)(row % H);
  const long n = row / H;
  const int pb = min(PB, W - ib0);
  const size_t flat = ((size_t)row * W + ib0) * P;
  for (int t = threadIdx.x; t < pb * P; t += 256) {
    const int px = t / P, e = t % P;
    tile[px * LP + (e / rr) * (rr + 1) + e % rr] = u[flat + t];
  }
  __syncthreads();
  const float inv_nnz = 1.f / fmaxf(*nnz, DL3_NNZ_FLOOR);
  const int Wr = W * r;
  float lsum = 0.f;
  for (int o = threadIdx.x; o < pb * rr; o += 256) {
    const int px = o / rr, pq = o % rr, pp = pq / r, q = pq % r;
    float *cell = tile + px * LP + pq;
    float z[MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; c++) z[c] = cell[min(c, C - 1) * (rr + 1)];
    float mx = z[0];
#pragma unroll
    for (int c = 1; c < MAXC; c++) mx = fmaxf(mx, (c < C) ? z[c] : z[0]);
    float ssum = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; c++) {
      z[c] = (c < C) ? expf(z[c] - mx) : 0.f;
      ssum += z[c];
    }
    const float inv = 1.f / ssum;
    const size_t m = ((size_t)n * H * r + (size_t)ia * r + q) * Wr + (size_t)(ib0 + px) * r + pp;  // output pixel
    const int t = (int)labels[m];
    const float w = (t >= 0 && t < C) ? (weights ? weights[m] : 1.f) : 0.f;  // void rows: zero loss and gradient
    float psum = 0.f, pt = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; c++) {
      z[c] *= inv;
      psum += z[c];
      pt = (c == t) ? z[c] : pt;
    }
    float inside = 1.f;
    if (t >= 0 && t < C) {
      float qq = pt / psum;
      inside = dl3_clip_pass(qq);
      qq = fminf(fmaxf(qq, 1e-7f), 1.f - 1e-7f);
      lsum += -logf(qq) * w * inv_nnz;
    }
    const float gs = w * inv_nnz * inside;
#pragma unroll
    for (int c = 0; c < MAXC; c++)
      if (c < C) cell[c * (rr + 1)] = (z[c] - (c == t ? 1.f : 0.f)) * gs;
  }
  __syncthreads();
  for (int t = threadIdx.x; t < pb * P; t += 256) {
    const int px = t / P, e = t % P;
    du[flat + t] = tile[px * LP + (e / rr) * (rr + 1) + e % rr];
  }
  lsum = wave_sum(lsum);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = lsum;
  __syncthreads();
  if (threadIdx.x == 0) loss_part[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}

// k x k taps of T(x) side by side (Subpixel with kernel_size > 1): one thread per (output pixel, tap, channel)
__global__ __launch_bounds__(256) void conv_taps_fwd_kernel(const float *__restrict__ x, int ldx,
                                                            const float *__restrict__ sc, const float *__restrict__ sh,
                                                            int act, float *__restrict__ cols, int N, int H, int W, int C,
                                                            int k, int pt, int pl, int Ho, int Wo) {
  const long total = (long)N * Ho * Wo * k * k * C;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C);
    long p = i / C;
    const int t = (int)(p % (k * k));
    p /= k * k;
    const int ox = (int)(p % Wo);
    p /= Wo;
    const int oy = (int)(p % Ho), n = (int)(p / Ho);
    const int iy = oy - pt + t / k, ix = ox - pl + t % k;
    float v = 0.f;
    if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
      v = x[(((size_t)n * H + iy) * W + ix) * ldx + c];
      if (sc) v = sc[c] * v + sh[c];
      v = dl3_act(v, act);
    }
    cols[i] = v;
  }
}

__global__ __launch_bounds__(256) void conv_taps_bwd_kernel(const float *__restrict__ dcols, float *__restrict__ dx,
                                                            int N, int H, int W, int C, int k, int pt, int pl, int Ho,
                                                            int Wo) {
  const long total = (long)N * H * W * C;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C);
    long p = i / C;
    const int ix = (int)(p % W);
    p /= W;
    const int iy = (int)(p % H), n = (int)(p / H);
    float s = 0.f;
    for (int t = 0; t < k * k; t++) {  // fixed tap order: deterministic
      const int oy = iy + pt - t / k, ox = ix + pl - t % k;
      if (oy >= 0 && oy < Ho && ox >= 0 && ox < Wo)
        s += dcols[((((size_t)n * Ho + oy) * Wo + ox) * (k * k) + t) * C + c];
    }
    dx[i] = s;
  }
}

__global__ __launch_bounds__(256) void softmax_kernel(const float *__restrict__ x, float *__restrict__ p, long M,
                                                      int C) {
  for (long m = (long)blockIdx.x * 256 + threadIdx.x; m < M; m += (long)gridDim.x * 256) {
    const float *r = x + (size_t)m * C;
    float mx = r[0];
    for (int c = 1; c < C; c++) mx = fmaxf(mx, r[c]);
    float s = 0.f;
    for (int c = 0; c < C; c++) s += expf(r[c] - mx);
    const float inv = 1.f / s;
    for (int c = 0; c < C; c++) p[(size_t)m * C + c] = expf(r[c] - mx) * inv;
  }
}

__global__ __launch_bounds__(256) void argmax_kernel(const float *__restrict__ x, int *__restrict__ out, long M,
                                                     int C) {
  for (long m = (long)blockIdx.x * 256 + threadIdx.x; m < M; m += (long)gridDim.x * 256) {
    const float *r = x + (size_t)m * C;
    float mx = r[0];
    int am = 0;
    for (int c = 1; c < C; c++)
      if (r[c] > mx) { mx = r[c]; am = c; }
    out[m] = am;
  }
}

__global__ __launch_bounds__(256) void count_nz_kernel(const float *__restrict__ w, long M, unsigned int *cnt) {
  __shared__ unsigned int red[4];
  unsigned int c = 0;
  for (long m = (long)blockIdx.x * 256 + threadIdx.x; m < M; m += (long)gridDim.x * 256) c += (w[m] != 0.f);
  // integer adds commute: the atomic result is order-independent (deterministic)
  for (int o = 32; o >= 1; o >>= 1) c += __shfl_xor(c, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(cnt, red[0] + red[1] + red[2] + red[3]);
}
__global__ void count_to_float_kernel(unsigned int *cnt) {
  const unsigned int v = *cnt;
  *reinterpret_cast<float *>(cnt) = (float)v;
}

// loss_partial[blockIdx.x] = sum over the block's rows of w*(-log clip(p_label))/nnz;
// dlogits = (p - onehot) * w / nnz ; label == C (void) -> the one-hot row is all zero (utils.py:129), so the row
// contributes neither loss nor gradient whatever its sample weight.
__global__ __launch_bounds__(256) void softmax_xent_kernel(const float *__restrict__ x,
                                                           const float *__restrict__ labels,
                                                           const float *__restrict__ weights,
                                                           const float *__restrict__ nnz, float *__restrict__ probs,
                                                           float *__restrict__ dl, float *__restrict__ loss_part,
                                                           long M, int C) {
  __shared__ float red[4];
  const float inv_nnz = 1.f / fmaxf(*nnz, DL3_NNZ_FLOOR);
  float lsum = 0.f;
  for (long m = (long)blockIdx.x * 256 + threadIdx.x; m < M; m += (long)gridDim.x * 256) {
    const float *r = x + (size_t)m * C;
    float mx = r[0];
    for (int c = 1; c < C; c++) mx = fmaxf(mx, r[c]);
    float s = 0.f;
    for (int c = 0; c < C; c++) s += expf(r[c] - mx);
    const float inv = 1.f / s;
    const int t = (int)labels[m];
    // void rows (label == C or out of range): the one-hot row is all zero, so loss AND gradient are zero whatever the
    // sample weight (utils.py:129 drops the last one-hot column)
    const float w = (t >= 0 && t < C) ? (weights ? weights[m] : 1.f) : 0.f;
    float psum = 0.f, pt = 0.f;
    for (int c = 0; c < C; c++) {
      const float pc = expf(r[c] - mx) * inv;
      psum += pc;
      if (c == t) pt = pc;
      if (probs) probs[(size_t)m * C + c] = pc;
    }
    float inside = 1.f;
    if (t >= 0 && t < C) {
      // Keras categorical_crossentropy on probabilities: renormalise, clip to [1e-7, 1-1e-7], -log
      float q = pt / psum;
      inside = dl3_clip_pass(q);
      q = fminf(fmaxf(q, 1e-7f), 1.f - 1e-7f);
      lsum += -logf(q) * w * inv_nnz;
    }
    if (dl) {
      const float gs = w * inv_nnz * inside;
      for (int c = 0; c < C; c++) dl[(size_t)m * C + c] = (expf(r[c] - mx) * inv - (c == t ? 1.f : 0.f)) * gs;
    }
  }
  lsum = wave_sum(lsum);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = lsum;
  __syncthreads();
  if (threadIdx.x == 0) loss_part[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}

// Loss tail, one pass over HBM.  Thread = pixel, logits held in registers (C <= 32); dlogits (and probs) leave
// through an LDS transpose so that the 256 x C floats of a workgroup are stored as contiguous float4s.
// UPSAMPLE: the logits are bilinearly interpolated on the fly from the low-resolution map (the resize_bilinear of
// deeplabv3p.py:439 / utils.py:190 fused in), so the full-resolution logits are never written or re-read.
template <bool UPSAMPLE>
__global__ __launch_bounds__(256) void xent32_kernel(const float *__restrict__ x, const float *__restrict__ labels,
                                                     const float *__restrict__ weights,
                                                     const float *__restrict__ nnz, float *__restrict__ probs,
                                                     float *__restrict__ dl, float *__restrict__ loss_part, long M,
                                                     int C, int Hi, int Wi, int Ho, int Wo, float sy, float sx) {
  constexpr int MAXC = 32;
  __shared__ float tile[256 * MAXC];
  __shared__ float red[4];
  const float inv_nnz = 1.f / fmaxf(*nnz, DL3_NNZ_FLOOR);
  float lsum = 0.f;
  for (long base = (long)blockIdx.x * 256; base < M; base += (long)gridDim.x * 256) {
    const long m = base + threadIdx.x;
    const bool ok = m < M;
    const long mc = ok ? m : M - 1;
    float z[MAXC];
    if (UPSAMPLE) {
      const int ox = (int)(mc % Wo);
      const long q = mc / Wo;
      const int oy = (int)(q % Ho), n = (int)(q / Ho);
      const Lerp ly = tf1_lerp(oy, sy, Hi), lx = tf1_lerp(ox, sx, Wi);
      const float *b = x + (size_t)n * Hi * Wi * C;
      const float *tl = b + ((size_t)ly.lo * Wi + lx.lo) * C, *tr = b + ((size_t)ly.lo * Wi + lx.hi) * C;
      const float *bl = b + ((size_t)ly.hi * Wi + lx.lo) * C, *br = b + ((size_t)ly.hi * Wi + lx.hi) * C;
#pragma unroll
      for (int c = 0; c < MAXC; c++) {
        const int cc = min(c, C - 1);
        const float top = tl[cc] + (tr[cc] - tl[cc]) * lx.w;
        const float bot = bl[cc] + (br[cc] - bl[cc]) * lx.w;
        z[c] = top + (bot - top) * ly.w;
      }
    } else {
      // coalesced copy of the workgroup's 256 x C logits through LDS, then one row per thread (stride C is odd for
      // C = 21: conflict-free)
      const long nrem = min((long)256, M - base) * C;
      for (long i = threadIdx.x; i < nrem; i += 256) tile[i] = x[(size_t)base * C + i];
      __syncthreads();
#pragma unroll
      for (int c = 0; c < MAXC; c++) z[c] = tile[(ok ? threadIdx.x : 0) * C + min(c, C - 1)];
      __syncthreads();
    }
    float mx = z[0];
#pragma unroll
    for (int c = 1; c < MAXC; c++) mx = fmaxf(mx, (c < C) ? z[c] : z[0]);
    float ssum = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; c++) {
      z[c] = (c < C) ? expf(z[c] - mx) : 0.f;
      ssum += z[c];
    }
    const float inv = 1.f / ssum;
    const int t = (int)labels[mc];
    const float w = (t >= 0 && t < C) ? (weights ? weights[mc] : 1.f) : 0.f;  // void rows: zero loss and gradient
    float psum = 0.f, pt = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; c++) {
      z[c] *= inv;  // probability
      psum += z[c];
      pt = (c == t) ? z[c] : pt;
    }
    float inside = 1.f;
    if (ok && t >= 0 && t < C) {
      float q = pt / psum;
      inside = dl3_clip_pass(q);
      q = fminf(fmaxf(q, 1e-7f), 1.f - 1e-7f);
      lsum += -logf(q) * w * inv_nnz;
    }
    const long nrem = min((long)256, M - base) * C;
    if (probs) {
#pragma unroll
      for (int c = 0; c < MAXC; c++)
        if (c < C) tile[threadIdx.x * C + c] = z[c];
      __syncthreads();
      for (long i = threadIdx.x; i < nrem; i += 256) probs[(size_t)base * C + i] = tile[i];
      __syncthreads();
    }
    if (dl) {
      const float gs = w * inv_nnz * inside;
#pragma unroll
      for (int c = 0; c < MAXC; c++)
        if (c < C) tile[threadIdx.x * C + c] = (z[c] - (c == t ? 1.f : 0.f)) * gs;
      __syncthreads();
      for (long i = threadIdx.x; i < nrem; i += 256) dl[(size_t)base * C + i] = tile[i];
      __syncthreads();
    }
  }
  lsum = wave_sum(lsum);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = lsum;
  __syncthreads();
  if (threadIdx.x == 0) loss_part[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}

#ifndef XENT_PS
#define XENT_PS 12  // prefetching form of the row kernel: source floats per thread and source row ...
#endif
#ifndef XENT_PP
#define XENT_PP 2   // ... and output pixels per thread
#endif
// The training tail in one pass per output ROW: a workgroup owns output row (n, oy); phase 1 interpolates the
// low-resolution logits, evaluates softmax / loss / dlogits for the Wo pixels of the row and leaves dlogits in LDS;
// phase 2 folds the row onto the Wi input columns (the x half of the transposed bilinear resize, fixed summation
// order).  The full-resolution dlogits [N,Ho,Wo,C] (704 MB at 32x512x512x21) never exist: the kernel writes the
// x-folded rows [N,Ho,Wi,C] (8x smaller) and dl3_resize_bilinear_bwd_rows finishes with the y half.
// The two low-resolution logit rows an output row interpolates between are staged in LDS first: the 4 x C neighbour
// reads per pixel then come from LDS instead of the texture-address path (8 neighbouring pixels share them).
template <int MAXC, bool PREF>
__global__ __launch_bounds__(256, 2) void xent32_fold_kernel(const float *__restrict__ x, const float *__restrict__ labels,
                                                          const float *__restrict__ weights,
                                                          const float *__restrict__ nnz, float *__restrict__ xfold,
                                                          float *__restrict__ loss_part, int N, int C, int Hi, int Wi,
                                                          int Ho, int Wo, float sy, float sx) {
  extern __shared__ __attribute__((aligned(16))) float fold_tile[];  // [Wo][C] dlogits of the row, [2][Wi][C] source rows, [Wo] x lerp table
  __shared__ float red[4];
  float *src = fold_tile + (size_t)Wo * C;
  float *xw = src + 2 * (size_t)Wi * C;  // fractional x weight of every output column
  int *xlo = (int *)(xw + Wo);           // its lower source column
  for (int ox = threadIdx.x; ox < Wo; ox += 256) {
    const Lerp lx = tf1_lerp(ox, sx, Wi);
    xw[ox] = lx.w;
    xlo[ox] = lx.lo;
  }
  const float inv_nnz = 1.f / fmaxf(*nnz, DL3_NNZ_FLOOR);
  float lsum = 0.f;
  // PREF (2 * Wi * C <= 2 * 256 * XENT_PS source floats, Wo <= 256 * XENT_PP pixels per row — the host checks): a row's global
  // operands — its two source rows, its labels and weights — are requested one row AHEAD into registers and only stored
  // to the LDS / consumed when their row starts.  Three workgroups of four waves per CU (52 KB of LDS each for a 512 x 21 row
  // over 32 source columns) hide little of a row's dependent round trips (source rows, labels inside phase 1).  Measured
  // (round 4, call 24, B=128: 65536 rows): 1.54 -> 1.36 ms per step; bit-identical results (tests).
  // (named registers, not arrays: the arrays stayed in scratch — 112 B per lane — whatever the indexing)
  static_assert(XENT_PS == 12 && XENT_PP == 2, "three float4 per thread and source row, two pixels per thread");
  float4 plo0, plo1, plo2, phi0, phi1, phi2;
  float plab0, plab1, pwt0, pwt1;
  plo0 = plo1 = plo2 = phi0 = phi1 = phi2 = make_float4(0.f, 0.f, 0.f, 0.f);
  plab0 = plab1 = pwt0 = pwt1 = 0.f;
  const int nq = Wi * C / 4;
  auto prefetch = [&](int row) __attribute__((always_inline)) {
    const int n = row / Ho, oy = row - n * Ho;
    const Lerp ly = tf1_lerp(oy, sy, Hi);
    const float4 *blo = reinterpret_cast<const float4 *>(x + ((size_t)n * Hi + ly.lo) * Wi * C);
    const float4 *bhi = reinterpret_cast<const float4 *>(x + ((size_t)n * Hi + ly.hi) * Wi * C);
    const int i0 = min((int)threadIdx.x, nq - 1), i1 = min((int)threadIdx.x + 256, nq - 1), i2 = min((int)threadIdx.x + 512, nq - 1);
    plo0 = blo[i0]; phi0 = bhi[i0];
    plo1 = blo[i1]; phi1 = bhi[i1];
    plo2 = blo[i2]; phi2 = bhi[i2];
    const size_t m0 = (size_t)row * Wo + min((int)threadIdx.x, Wo - 1), m1 = (size_t)row * Wo + min((int)threadIdx.x + 256, Wo - 1);
    plab0 = labels[m0]; plab1 = labels[m1];
    pwt0 = weights ? weights[m0] : 1.f; pwt1 = weights ? weights[m1] : 1.f;
  };
  if (PREF && (int)blockIdx.x < N * Ho) prefetch(blockIdx.x);
  for (int row = blockIdx.x; row < N * Ho; row += gridDim.x) {
    const int n = row / Ho, oy = row - n * Ho;
    const Lerp ly = tf1_lerp(oy, sy, Hi);
    float clab0 = 0.f, clab1 = 0.f, cwt0 = 0.f, cwt1 = 0.f;
    if constexpr (PREF) {
      float4 *s4lo = reinterpret_cast<float4 *>(src), *s4hi = reinterpret_cast<float4 *>(src + Wi * C);
      const int i0 = threadIdx.x, i1 = threadIdx.x + 256, i2 = threadIdx.x + 512;
      if (i0 < nq) { s4lo[i0] = plo0; s4hi[i0] = phi0; }
      if (i1 < nq) { s4lo[i1] = plo1; s4hi[i1] = phi1; }
      if (i2 < nq) { s4lo[i2] = plo2; s4hi[i2] = phi2; }
      clab0 = plab0; clab1 = plab1; cwt0 = pwt0; cwt1 = pwt1;
    } else {
      const float *b = x + (size_t)n * Hi * Wi * C;
      for (int i = threadIdx.x; i < Wi * C; i += 256) {
        src[i] = b[(size_t)ly.lo * Wi * C + i];
        src[Wi * C + i] = b[(size_t)ly.hi * Wi * C + i];
      }
    }
    __syncthreads();
    if (PREF && row + (int)gridDim.x < N * Ho) prefetch(row + gridDim.x);
    // one output pixel: interpolate, softmax, loss, dlogits into the LDS row (labf / wf: its label and sample weight)
    auto pixel = [&](int ox, float labf, float wf) __attribute__((always_inline)) {
      const Lerp lx = tf1_lerp(ox, sx, Wi);
      const float *tl = src + lx.lo * C, *tr = src + lx.hi * C;
      const float *bl = src + (Wi + lx.lo) * C, *br = src + (Wi + lx.hi) * C;
      float z[MAXC];
#pragma unroll
      for (int c = 0; c < MAXC; c++) {
        const int cc = min(c, C - 1);
        const float top = tl[cc] + (tr[cc] - tl[cc]) * lx.w;
        const float bot = bl[cc] + (br[cc] - bl[cc]) * lx.w;
        z[c] = top + (bot - top) * ly.w;
      }
      float mx = z[0];
#pragma unroll
      for (int c = 1; c < MAXC; c++) mx = fmaxf(mx, (c < C) ? z[c] : z[0]);
      float ssum = 0.f;
#pragma unroll
      for (int c = 0; c < MAXC; c++) {
        z[c] = (c < C) ? __expf(z[c] - mx) : 0.f;
        ssum += z[c];
      }
      const float inv = 1.f / ssum;
      const int t = (int)labf;
      const float w = (t >= 0 && t < C) ? wf : 0.f;  // void rows: zero loss and gradient
      float psum = 0.f, pt = 0.f;
#pragma unroll
      for (int c = 0; c < MAXC; c++) {
        z[c] *= inv;
        psum += z[c];
        pt = (c == t) ? z[c] : pt;
      }
      float inside = 1.f;
      if (t >= 0 && t < C) {
        float q = pt / psum;
        inside = dl3_clip_pass(q);
        q = fminf(fmaxf(q, 1e-7f), 1.f - 1e-7f);
        lsum += -logf(q) * w * inv_nnz;
      }
      const float gs = w * inv_nnz * inside;
#pragma unroll
      for (int c = 0; c < MAXC; c++)
        if (c < C) fold_tile[ox * C + c] = (z[c] - (c == t ? 1.f : 0.f)) * gs;
    };
    if constexpr (PREF) {
      // (a real loop: unrolled, the scheduler hoists every LDS operand of the row's pixels and spills — 256 VGPRs + scratch
      // against 65 for the loop)
#pragma nounroll
      for (int j = 0; j < 2; j++) {
        const int ox = threadIdx.x + 256 * j;
        if (ox < Wo) pixel(ox, j == 0 ? clab0 : clab1, j == 0 ? cwt0 : cwt1);
      }
    } else {
      for (int ox = threadIdx.x; ox < Wo; ox += 256) {
        const size_t m = (size_t)row * Wo + ox;
        pixel(ox, labels[m], weights ? weights[m] : 1.f);
      }
    }
    __syncthreads();
    // x fold: work item = (input column, group of 3 channels); the lerp of an output column comes from the table
    float *orow = xfold + (size_t)row * Wi * C;
    const int CG = (C + 2) / 3;
    for (int i = threadIdx.x; i < Wi * CG; i += 256) {
      const int ix = i / CG, c0 = (i - ix * CG) * 3;
      int ox0 = (int)floorf((float)(ix - 1) / sx) - 1, ox1 = (int)ceilf((float)(ix + 1) / sx) + 1;
      ox0 = max(ox0, 0);
      ox1 = min(ox1, Wo - 1);
      const int c1 = min(c0 + 1, C - 1), c2 = min(c0 + 2, C - 1);
      float a0 = 0.f, a1 = 0.f, a2 = 0.f;
      for (int ox = ox0; ox <= ox1; ++ox) {
        const int lo = xlo[ox], hi = min(lo + 1, Wi - 1);
        const float w = xw[ox];
        const float wx = (lo == ix ? 1.f - w : 0.f) + (hi == ix ? w : 0.f);
        const float *t = fold_tile + ox * C;
        a0 += wx * t[c0];
        a1 += wx * t[c1];
        a2 += wx * t[c2];
      }
      orow[ix * C + c0] = a0;
      if (c0 + 1 < C) orow[ix * C + c0 + 1] = a1;
      if (c0 + 2 < C) orow[ix * C + c0 + 2] = a2;
    }
    __syncthreads();
  }
  lsum = wave_sum(lsum);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = lsum;
  __syncthreads();
  if (threadIdx.x == 0) loss_part[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}

inline int ew_blocks(size_t n) {
  size_t b = (n + 255) / 256;
  if (b > 8192) b = 8192;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

extern "C" int dl3_resize_bilinear_fwd(const float *x, int ldx, const float *in_scale, const float *in_shift,
                                       int in_act, float *y, int ldy, int N, int Hi, int Wi, int Ho, int Wo, int C,
                                       void *stream) {
  DL3_CHECK_ARG(x && y && N > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && C > 0, "resize_fwd: bad argument");
  DL3_CHECK_ARG(ldx >= C && ldy >= C, "resize_fwd: leading dimension too small");
  DL3_CHECK_ARG((in_scale == nullptr) == (in_shift == nullptr), "resize_fwd: scale/shift must come together");
  const float sy = (float)Hi / (float)Ho, sx = (float)Wi / (float)Wo;
  DL3_UNSUPPORTED((long)N * Ho > 2147483647L, "resize_fwd: too many rows");
  int gx = dl3_cdiv(Wo * C, 256);
  if (gx > 64) gx = 64;
  hipLaunchKernelGGL(resize_fwd_kernel, dim3(gx, N * Ho), dim3(256), 0, (hipStream_t)stream, x, ldx, in_scale,
                     in_shift, in_act, y, ldy, N, Hi, Wi, Ho, Wo, C, sy, sx);
  DL3_LAUNCH_CHECK("resize_fwd");
  return DL3_OK;
}

extern "C" size_t dl3_resize_bilinear_bwd_workspace(int N, int Hi, int Wi, int Ho, int Wo, int C) {
  (void)Hi; (void)Wo;
  return (size_t)N * Ho * Wi * C * sizeof(float);
}

extern "C" int dl3_resize_bilinear_bwd(const float *dy, int lddy, float *dx, int lddx, int N, int Hi, int Wi,
                                       int Ho, int Wo, int C, int accumulate, void *workspace,
                                       size_t workspace_bytes, void *stream) {
  DL3_CHECK_ARG(dy && dx && N > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && C > 0, "resize_bwd: bad argument");
  DL3_CHECK_ARG(lddy >= C && lddx >= C, "resize_bwd: leading dimension too small");
  const float sy = (float)Hi / (float)Ho, sx = (float)Wi / (float)Wo;
  if (workspace && workspace_bytes >= dl3_resize_bilinear_bwd_workspace(N, Hi, Wi, Ho, Wo, C)) {
    int gx = dl3_cdiv(Wi * C, 256);
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(resize_bwd_x_kernel, dim3(gx, N * Ho), dim3(256), 0, (hipStream_t)stream, dy, lddy,
                       (float *)workspace, Wi, Wo, C, sx);
    hipLaunchKernelGGL(resize_bwd_y_kernel, dim3(gx, N * Hi), dim3(256), 0, (hipStream_t)stream,
                       (const float *)workspace, dx, lddx, Hi, Wi, Ho, C, sy, accumulate);
    DL3_LAUNCH_CHECK("resize_bwd(separable)");
    return DL3_OK;
  }
  int gx = dl3_cdiv(Wi * C, 256);
  if (gx > 64) gx = 64;
  hipLaunchKernelGGL(resize_bwd_kernel, dim3(gx, N * Hi), dim3(256), 0, (hipStream_t)stream, dy, lddy, dx, lddx, N,
                     Hi, Wi, Ho, Wo, C, sy, sx, accumulate);
  DL3_LAUNCH_CHECK("resize_bwd");
  return DL3_OK;
}

extern "C" int dl3_resize_bilinear_bwd_rows(const float *xfold, float *dx, int lddx, int N, int Hi, int Wi, int Ho,
                                            int C, int accumulate, void *stream) {
  DL3_CHECK_ARG(xfold && dx && N > 0 && Hi > 0 && Wi > 0 && Ho > 0 && C > 0, "resize_bwd_rows: bad argument");
  DL3_CHECK_ARG(lddx >= C, "resize_bwd_rows: leading dimension too small");
  const float sy = (float)Hi / (float)Ho;
  int gx = dl3_cdiv(Wi * C, 256);
  if (gx > 64) gx = 64;
  hipLaunchKernelGGL(resize_bwd_y_kernel, dim3(gx, N * Hi), dim3(256), 0, (hipStream_t)stream, xfold, dx, lddx, Hi, Wi,
                     Ho, C, sy, accumulate);
  DL3_LAUNCH_CHECK("resize_bwd_rows");
  return DL3_OK;
}

extern "C" int dl3_phase_shift(const float *in, float *out, int N, int H, int W, int Cout, int r, int inverse,
                               void *stream) {
  DL3_CHECK_ARG(in && out && N > 0 && H > 0 && W > 0 && Cout > 0 && r > 0, "phase_shift: bad argument");
  // pixels per workgroup: as many as fit 48 KB of LDS (three workgroups per CU), at most 8
  const size_t per_px = (size_t)Cout * (r * r + 1) * sizeof(float);
  int PB = (int)((48u << 10) / per_px);
  if (PB > 8) PB = 8;
  if (PB > W) PB = W;
  const long blocks = (long)N * H * dl3_cdiv(W, PB > 0 ? PB : 1);
  if (PB >= 1 && blocks < (1L << 31)) {
    hipLaunchKernelGGL(phase_shift_lds_kernel, dim3((unsigned)blocks), dim3(256), PB * per_px, (hipStream_t)stream, in,
                       out, H, W, Cout, r, PB, inverse);
  } else {
    hipLaunchKernelGGL(phase_shift_kernel, dim3(ew_blocks((size_t)N * H * W * Cout * r * r)), dim3(256), 0,
                       (hipStream_t)stream, in, out, N, H, W, Cout, r, inverse);
  }
  DL3_LAUNCH_CHECK("phase_shift");
  return DL3_OK;
}

static int shuffle_xent_pb(int W, int C, int r) {
  const size_t per_px = (size_t)C * (r * r + 1) * sizeof(float);
  int PB = (int)((48u << 10) / per_px);
  if (PB > 8) PB = 8;
  if (PB > W) PB = W;
  return PB;
}

extern "C" int dl3_shuffle_xent_partials(int N, int H, int W, int C, int r) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || r <= 0) return 0;
  const int PB = shuffle_xent_pb(W, C, r);
  if (PB < 1) return 0;
  const long blocks = (long)N * H * dl3_cdiv(W, PB);
  return blocks < (1L << 30) ? (int)blocks : 0;
}

extern "C" int dl3_shuffle_softmax_xent(const float *u, const float *labels, const float *weights, const float *nnz,
                                        float *du, float *loss_partial, int N, int H, int W, int C, int r,
                                        void *stream) {
  DL3_CHECK_ARG(u && labels && nnz && du && loss_partial && N > 0 && H > 0 && W > 0 && C > 0 && r > 0,
                "shuffle_softmax_xent: bad argument");
  DL3_UNSUPPORTED(C > 32, "shuffle_softmax_xent: C=%d > 32 (use phase_shift + softmax_xent)", C);
  const int P = dl3_shuffle_xent_partials(N, H, W, C, r);
  DL3_UNSUPPORTED(P <= 0, "shuffle_softmax_xent: a pixel of %d x %d x %d floats does not fit the LDS tile", C, r, r);
  const int PB = shuffle_xent_pb(W, C, r);
  const size_t lds = (size_t)PB * C * (r * r + 1) * sizeof(float);
  if (C <= 24)
    hipLaunchKernelGGL(shuffle_xent_kernel<24>, dim3(P), dim3(256), lds, (hipStream_t)stream, u, labels, weights, nnz, du,
                       loss_partial, H, W, C, r, PB);
  else
    hipLaunchKernelGGL(shuffle_xent_kernel<32>, dim3(P), dim3(256), lds, (hipStream_t)stream, u, labels, weights, nnz, du,
                       loss_partial, H, W, C, r, PB);
  DL3_LAUNCH_CHECK("shuffle_softmax_xent");
  return DL3_OK;
}

extern "C" int dl3_conv_taps_fwd(const float *x, int ldx, const float *in_scale, const float *in_shift, int in_act,
                                 float *cols, int N, int H, int W, int C, int k, int pad_t, int pad_l, int Ho, int Wo,
                                 void *stream) {
  DL3_CHECK_ARG(x && cols && N > 0 && H > 0 && W > 0 && C > 0 && k > 0 && Ho > 0 && Wo > 0 && ldx >= C,
                "conv_taps_fwd: bad argument");
  DL3_CHECK_ARG((in_scale == nullptr) == (in_shift == nullptr), "conv_taps_fwd: scale/shift must come together");
  DL3_CHECK_ARG(pad_t >= 0 && pad_l >= 0 && Ho - 1 - pad_t + k - 1 < H + k && Wo - 1 - pad_l + k - 1 < W + k,
                "conv_taps_fwd: geometry out of range");
  hipLaunchKernelGGL(conv_taps_fwd_kernel, dim3(ew_blocks((size_t)N * Ho * Wo * k * k * C)), dim3(256), 0,
                     (hipStream_t)stream, x, ldx, in_scale, in_shift, in_act, cols, N, H, W, C, k, pad_t, pad_l, Ho, Wo);
  DL3_LAUNCH_CHECK("conv_taps_fwd");
  return DL3_OK;
}

extern "C" int dl3_conv_taps_bwd(const float *dcols, float *dx, int N, int H, int W, int C, int k, int pad_t, int pad_l,
                                 int Ho, int Wo, void *stream) {
  DL3_CHECK_ARG(dcols && dx && N > 0 && H > 0 && W > 0 && C > 0 && k > 0 && Ho > 0 && Wo > 0 && pad_t >= 0 && pad_l >= 0,
                "conv_taps_bwd: bad argument");
  hipLaunchKernelGGL(conv_taps_bwd_kernel, dim3(ew_blocks((size_t)N * H * W * C)), dim3(256), 0, (hipStream_t)stream,
                     dcols, dx, N, H, W, C, k, pad_t, pad_l, Ho, Wo);
  DL3_LAUNCH_CHECK("conv_taps_bwd");
  return DL3_OK;
}

extern "C" int dl3_softmax_fwd(const float *logits, float *probs, int M, int C, void *stream) {
  DL3_CHECK_ARG(logits && probs && M > 0 && C > 0, "softmax_fwd: bad argument");
  hipLaunchKernelGGL(softmax_kernel, dim3(ew_blocks((size_t)M)), dim3(256), 0, (hipStream_t)stream, logits, probs,
                     (long)M, C);
  DL3_LAUNCH_CHECK("softmax_fwd");
  return DL3_OK;
}

extern "C" int dl3_argmax(const float *x, int *out, int M, int C, void *stream) {
  DL3_CHECK_ARG(x && out && M > 0 && C > 0, "argmax: bad argument");
  hipLaunchKernelGGL(argmax_kernel, dim3(ew_blocks((size_t)M)), dim3(256), 0, (hipStream_t)stream, x, out, (long)M,
                     C);
  DL3_LAUNCH_CHECK("argmax");
  return DL3_OK;
}

extern "C" int dl3_count_nonzero(const float *w, int M, float *out, void *stream) {
  DL3_CHECK_ARG(w && out && M > 0, "count_nonzero: bad argument");
  hipStream_t st = (hipStream_t)stream;
  (void)hipMemsetAsync(out, 0, sizeof(float), st);
  int blocks = ew_blocks((size_t)M);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(count_nz_kernel, dim3(blocks), dim3(256), 0, st, w, (long)M, (unsigned int *)out);
  hipLaunchKernelGGL(count_to_float_kernel, dim3(1), dim3(1), 0, st, (unsigned int *)out);
  DL3_LAUNCH_CHECK("count_nonzero");
  return DL3_OK;
}

extern "C" int dl3_softmax_xent(const float *logits, const float *labels, const float *weights, const float *nnz,
                                float *probs, float *dlogits, float *loss_partial, int M, int C, void *stream) {
  DL3_CHECK_ARG(logits && labels && nnz && loss_partial && M > 0 && C > 0, "softmax_xent: bad argument");
  if (C <= 32)
    hipLaunchKernelGGL((xent32_kernel<false>), dim3(dl3_rows_partials(M)), dim3(256), 0, (hipStream_t)stream, logits,
                       labels, weights, nnz, probs, dlogits, loss_partial, (long)M, C, 0, 0, 0, 0, 0.f, 0.f);
  else
    hipLaunchKernelGGL(softmax_xent_kernel, dim3(dl3_rows_partials(M)), dim3(256), 0, (hipStream_t)stream, logits,
                       labels, weights, nnz, probs, dlogits, loss_partial, (long)M, C);
  DL3_LAUNCH_CHECK("softmax_xent");
  return DL3_OK;
}

extern "C" int dl3_upsample_softmax_xent(const float *logits_lo, const float *labels, const float *weights,
                                         const float *nnz, float *probs, float *dlogits, float *loss_partial, int N,
                                         int Hi, int Wi, int Ho, int Wo, int C, void *stream) {
  DL3_CHECK_ARG(logits_lo && labels && nnz && loss_partial && N > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && C > 0,
                "upsample_softmax_xent: bad argument");
  DL3_UNSUPPORTED(C > 32, "upsample_softmax_xent: C=%d > 32 (use resize_bilinear_fwd + softmax_xent)", C);
  const long M = (long)N * Ho * Wo;
  const float sy = (float)Hi / (float)Ho, sx = (float)Wi / (float)Wo;
  hipLaunchKernelGGL((xent32_kernel<true>), dim3(dl3_rows_partials((int)M)), dim3(256), 0, (hipStream_t)stream,
                     logits_lo, labels, weights, nnz, probs, dlogits, loss_partial, M, C, Hi, Wi, Ho, Wo, sy, sx);
  DL3_LAUNCH_CHECK("upsample_softmax_xent");
  return DL3_OK;
}

extern "C" int dl3_xent_fold_partials(int N, int Ho) {
  const long rows = (long)N * Ho;
  if (rows <= 0) return 0;
  return (int)(rows < 4096 ? rows : 4096);
}

extern "C" int dl3_upsample_softmax_xent_fold(const float *logits_lo, const float *labels, const float *weights,
                                              const float *nnz, float *dlogits_xfold, float *loss_partial, int N,
                                              int Hi, int Wi, int Ho, int Wo, int C, void *stream) {
  DL3_CHECK_ARG(logits_lo && labels && nnz && dlogits_xfold && loss_partial && N > 0 && Hi > 0 && Wi > 0 && Ho > 0 &&
                    Wo > 0 && C > 0,
                "upsample_softmax_xent_fold: bad argument");
  DL3_UNSUPPORTED(C > 32, "upsample_softmax_xent_fold: C=%d > 32", C);
  const size_t lds = (((size_t)Wo + 2 * (size_t)Wi) * C + 2 * (size_t)Wo) * sizeof(float);
  DL3_UNSUPPORTED(lds > 64 * 1024, "upsample_softmax_xent_fold: (%d + 2*%d) x %d floats exceed 64 KB of LDS", Wo, Wi, C);
  const float sy = (float)Hi / (float)Ho, sx = (float)Wi / (float)Wo;
  const dim3 grid(dl3_xent_fold_partials(N, Ho));
  const char *pe = getenv("DL3_XENT_PREF");  // 0 = every row requests its own operands (tuning / test aid)
  // (float4 requests and LDS stores: source rows of whole float4s, 16-byte aligned; the source tile starts Wo * C floats in)
  const bool pref = (long)Wi * C <= 256L * XENT_PS && Wo <= 256 * XENT_PP && (Wi * C) % 4 == 0 && ((long)Wo * C) % 4 == 0 &&
                    (((uintptr_t)logits_lo) & 15) == 0 && !(pe && atoi(pe) == 0);
#define DL3_XENT(MAXC_, PREF_)                                                                                         \
  hipLaunchKernelGGL((xent32_fold_kernel<MAXC_, PREF_>), grid, dim3(256), lds, (hipStream_t)stream, logits_lo, labels, \
                     weights, nnz, dlogits_xfold, loss_partial, N, C, Hi, Wi, Ho, Wo, sy, sx)
  if (C <= 24) { if (pref) DL3_XENT(24, true); else DL3_XENT(24, false); }
  else { if (pref) DL3_XENT(32, true); else DL3_XENT(32, false); }
#undef DL3_XENT
  DL3_LAUNCH_CHECK("upsample_softmax_xent_fold");
  return DL3_OK;
}
