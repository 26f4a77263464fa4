/* dl3_ops_impl.h — body of the C/OpenMP operator restatement, included once per precision (REAL, FN).
 * TEST INFRASTRUCTURE (see dl3_ops.c).  NHWC, Keras weight layouts, same semantics as oracle/dl3_oracle.py's
 * numpy operators; every function cites the reference lines its numpy twin cites. */

/* DepthwiseConv2D 3x3, depth multiplier 1 (deeplabv3p.py:73-74,:186-188):
 * y[n,oy,ox,c] = sum_ij x[n, oy*s-pt+i*r, ox*s-pl+j*r, c] * w[i][j][c] */
void FN(dw3x3_fwd)(const REAL *x, const REAL *w, REAL *y, long N, long H, long W, long C, long s, long r, long pt,
                   long pl, long Ho, long Wo) {
#pragma omp parallel for collapse(2) schedule(static)
  for (long n = 0; n < N; n++)
    for (long oy = 0; oy < Ho; oy++)
      for (long ox = 0; ox < Wo; ox++) {
        REAL *yo = y + ((n * Ho + oy) * Wo + ox) * C;
        for (long c = 0; c < C; c++) yo[c] = 0;
        for (long i = 0; i < 3; i++) {
          const long iy = oy * s - pt + i * r;
          if (iy < 0 || iy >= H) continue;
          for (long j = 0; j < 3; j++) {
            const long ix = ox * s - pl + j * r;
            if (ix < 0 || ix >= W) continue;
            const REAL *xi = x + ((n * H + iy) * W + ix) * C, *wk = w + (i * 3 + j) * C;
            for (long c = 0; c < C; c++) yo[c] += xi[c] * wk[c];
          }
        }
      }
}

/* dx = conv^T(g, w) (gather form over input pixels), dw[i][j][c] = sum x[tap] * g */
void FN(dw3x3_bwd)(const REAL *x, const REAL *w, const REAL *g, REAL *dx, REAL *dw, long N, long H, long W, long C,
                   long s, long r, long pt, long pl, long Ho, long Wo) {
#pragma omp parallel for collapse(2) schedule(static)
  for (long n = 0; n < N; n++)
    for (long iy = 0; iy < H; iy++)
      for (long ix = 0; ix < W; ix++) {
        REAL *d = dx + ((n * H + iy) * W + ix) * C;
        for (long c = 0; c < C; c++) d[c] = 0;
        for (long i = 0; i < 3; i++) {
          const long ty = iy + pt - i * r;
          if (ty < 0 || ty % s) continue;
          const long oy = ty / s;
          if (oy >= Ho) continue;
          for (long j = 0; j < 3; j++) {
            const long tx = ix + pl - j * r;
            if (tx < 0 || tx % s) continue;
            const long ox = tx / s;
            if (ox >= Wo) continue;
            const REAL *go = g + ((n * Ho + oy) * Wo + ox) * C, *wk = w + (i * 3 + j) * C;
            for (long c = 0; c < C; c++) d[c] += go[c] * wk[c];
          }
        }
      }
  /* weight gradient: every thread owns a block of channels (no shared accumulators: deterministic) */
#pragma omp parallel
  {
    const long T = omp_get_num_threads(), t = omp_get_thread_num();
    const long c0 = C * t / T, c1 = C * (t + 1) / T;
    for (long k = 0; k < 9; k++)
      for (long c = c0; c < c1; c++) dw[k * C + c] = 0;
    for (long n = 0; n < N; n++)
      for (long oy = 0; oy < Ho; oy++)
        for (long ox = 0; ox < Wo; ox++) {
          const REAL *go = g + ((n * Ho + oy) * Wo + ox) * C;
          for (long i = 0; i < 3; i++) {
            const long iy = oy * s - pt + i * r;
            if (iy < 0 || iy >= H) continue;
            for (long j = 0; j < 3; j++) {
              const long ix = ox * s - pl + j * r;
              if (ix < 0 || ix >= W) continue;
              const REAL *xi = x + ((n * H + iy) * W + ix) * C;
              REAL *dk = dw + (i * 3 + j) * C;
              for (long c = c0; c < c1; c++) dk[c] += xi[c] * go[c];
            }
          }
        }
  }
}

/* Conv2D k x k, cross-correlation, HWIO kernel (deeplabv3p.py:99-116,:283,:318,:377,...), optional bias */
void FN(conv_fwd)(const REAL *x, const REAL *w, const REAL *bias, REAL *y, long N, long H, long W, long Ci, long Co,
                  long k, long s, long pt, long pl, long Ho, long Wo) {
#pragma omp parallel for collapse(2) schedule(static)
  for (long n = 0; n < N; n++)
    for (long oy = 0; oy < Ho; oy++)
      for (long ox = 0; ox < Wo; ox++) {
        REAL *yo = y + ((n * Ho + oy) * Wo + ox) * Co;
        for (long co = 0; co < Co; co++) yo[co] = bias ? bias[co] : 0;
        for (long i = 0; i < k; i++) {
          const long iy = oy * s - pt + i;
          if (iy < 0 || iy >= H) continue;
          for (long j = 0; j < k; j++) {
            const long ix = ox * s - pl + j;
            if (ix < 0 || ix >= W) continue;
            const REAL *xi = x + ((n * H + iy) * W + ix) * Ci, *wk = w + (i * k + j) * Ci * Co;
            for (long ci = 0; ci < Ci; ci++) {
              const REAL xv = xi[ci];
              const REAL *wr = wk + ci * Co;
              for (long co = 0; co < Co; co++) yo[co] += xv * wr[co];
            }
          }
        }
      }
}

void FN(conv_bwd)(const REAL *x, const REAL *w, const REAL *g, REAL *dx, REAL *dw, REAL *dbias, long N, long H, long W,
                  long Ci, long Co, long k, long s, long pt, long pl, long Ho, long Wo) {
#pragma omp parallel for collapse(2) schedule(static)
  for (long n = 0; n < N; n++)
    for (long iy = 0; iy < H; iy++)
      for (long ix = 0; ix < W; ix++) {
        REAL *d = dx + ((n * H + iy) * W + ix) * Ci;
        for (long ci = 0; ci < Ci; ci++) d[ci] = 0;
        for (long i = 0; i < k; i++) {
          const long ty = iy + pt - i;
          if (ty < 0 || ty % s) continue;
          const long oy = ty / s;
          if (oy >= Ho) continue;
          for (long j = 0; j < k; j++) {
            const long tx = ix + pl - j;
            if (tx < 0 || tx % s) continue;
            const long ox = tx / s;
            if (ox >= Wo) continue;
            const REAL *go = g + ((n * Ho + oy) * Wo + ox) * Co, *wk = w + (i * k + j) * Ci * Co;
            for (long ci = 0; ci < Ci; ci++) {
              const REAL *wr = wk + ci * Co;
              REAL a = 0;
              for (long co = 0; co < Co; co++) a += go[co] * wr[co];
              d[ci] += a;
            }
          }
        }
      }
  /* weight gradient: output pixels split statically over the threads, one private accumulator each, folded in thread
   * order (deterministic for a given thread count) */
  {
    const long nw = k * k * Ci * Co;
    const long T = omp_get_max_threads();
    REAL *acc = (REAL *)calloc((size_t)T * (nw + Co), sizeof(REAL));
    const long NP = N * Ho * Wo;
#pragma omp parallel num_threads(T)
    {
      const long t = omp_get_thread_num(), nt = omp_get_num_threads();
      REAL *a = acc + t * (nw + Co), *ab = a + nw;
      const long p0 = NP * t / nt, p1 = NP * (t + 1) / nt;
      for (long p = p0; p < p1; p++) {
        const long ox = p % Wo, oy = (p / Wo) % Ho, n = p / (Wo * Ho);
        const REAL *go = g + p * Co;
        if (dbias)
          for (long co = 0; co < Co; co++) ab[co] += go[co];
        for (long i = 0; i < k; i++) {
          const long iy = oy * s - pt + i;
          if (iy < 0 || iy >= H) continue;
          for (long j = 0; j < k; j++) {
            const long ix = ox * s - pl + j;
            if (ix < 0 || ix >= W) continue;
            const REAL *xi = x + ((n * H + iy) * W + ix) * Ci;
            REAL *dk = a + (i * k + j) * Ci * Co;
            for (long ci = 0; ci < Ci; ci++) {
              const REAL xv = xi[ci];
              REAL *dr = dk + ci * Co;
              for (long co = 0; co < Co; co++) dr[co] += xv * go[co];
            }
          }
        }
      }
    }
#pragma omp parallel for schedule(static)
    for (long q = 0; q < nw; q++) {
      REAL v = 0;
      for (long t = 0; t < T; t++) v += acc[t * (nw + Co) + q];
      dw[q] = v;
    }
    if (dbias)
      for (long co = 0; co < Co; co++) {
        REAL v = 0;
        for (long t = 0; t < T; t++) v += acc[t * (nw + Co) + nw + co];
        dbias[co] = v;
      }
    free(acc);
  }
}

/* ReLU / relu(x, max_value=6.) (deeplabv3p.py:72,:181,:192,:325): y = min(max(x, 0), hi); hi <= 0 means no upper clamp */
void FN(relu_fwd)(const REAL *x, REAL *y, REAL hi, long n) {
#pragma omp parallel for schedule(static)
  for (long i = 0; i < n; i++) {
    REAL v = x[i] > 0 ? x[i] : 0;
    y[i] = (hi > 0 && v > hi) ? hi : v;
  }
}
void FN(relu_bwd)(const REAL *x, const REAL *g, REAL *dx, REAL hi, long n) {
#pragma omp parallel for schedule(static)
  for (long i = 0; i < n; i++) dx[i] = (x[i] > 0 && (hi <= 0 || x[i] < hi)) ? g[i] : 0;
}

/* BatchNormalization over the last axis, training mode [TF FusedBatchNorm]: biased batch variance; statistics in double */
void FN(bn_train_fwd)(const REAL *x, const REAL *gamma, const REAL *beta, REAL eps, REAL *y, REAL *xhat, double *mean,
                      double *var, double *invstd, long M, long C) {
#pragma omp parallel
  {
    const long T = omp_get_num_threads(), t = omp_get_thread_num();
    const long c0 = C * t / T, c1 = C * (t + 1) / T;
    for (long c = c0; c < c1; c++) { mean[c] = 0; var[c] = 0; }
    for (long m = 0; m < M; m++)
      for (long c = c0; c < c1; c++) mean[c] += (double)x[m * C + c];
    for (long c = c0; c < c1; c++) mean[c] /= (double)M;
    for (long m = 0; m < M; m++)
      for (long c = c0; c < c1; c++) {
        const double d = (double)x[m * C + c] - mean[c];
        var[c] += d * d;
      }
    for (long c = c0; c < c1; c++) { var[c] /= (double)M; invstd[c] = 1.0 / sqrt(var[c] + (double)eps); }
  }
#pragma omp parallel for schedule(static)
  for (long m = 0; m < M; m++)
    for (long c = 0; c < C; c++) {
      const REAL xh = (REAL)(((double)x[m * C + c] - mean[c]) * invstd[c]);
      xhat[m * C + c] = xh;
      y[m * C + c] = xh * gamma[c] + beta[c];
    }
}

/* dx = gamma*invstd*(g - dbeta/M - xhat*dgamma/M), dgamma = sum g*xhat, dbeta = sum g */
void FN(bn_train_bwd)(const REAL *g, const REAL *xhat, const REAL *gamma, const double *invstd, REAL *dx, double *dgamma,
                      double *dbeta, long M, long C) {
#pragma omp parallel
  {
    const long T = omp_get_num_threads(), t = omp_get_thread_num();
    const long c0 = C * t / T, c1 = C * (t + 1) / T;
    for (long c = c0; c < c1; c++) { dgamma[c] = 0; dbeta[c] = 0; }
    for (long m = 0; m < M; m++)
      for (long c = c0; c < c1; c++) {
        dbeta[c] += (double)g[m * C + c];
        dgamma[c] += (double)g[m * C + c] * (double)xhat[m * C + c];
      }
  }
#pragma omp parallel for schedule(static)
  for (long m = 0; m < M; m++)
    for (long c = 0; c < C; c++)
      dx[m * C + c] = (REAL)((double)gamma[c] * invstd[c] *
                             ((double)g[m * C + c] - dbeta[c] / (double)M - (double)xhat[m * C + c] * (dgamma[c] / (double)M)));
}

/* tf.image.resize_bilinear, TF 1.x legacy (align_corners=False, no half-pixel centres): deeplabv3p.py:382,:418,:439.
 * lo/hi/wt tables come from the caller (float32 source coordinates, as TF computes them). */
void FN(resize_fwd)(const REAL *x, REAL *y, const long *ylo, const long *yhi, const REAL *wy, const long *xlo,
                    const long *xhi, const REAL *wx, long N, long Hi, long Wi, long Ho, long Wo, long C) {
#pragma omp parallel for collapse(2) schedule(static)
  for (long n = 0; n < N; n++)
    for (long oy = 0; oy < Ho; oy++)
      for (long ox = 0; ox < Wo; ox++) {
        const REAL *tl = x + ((n * Hi + ylo[oy]) * Wi + xlo[ox]) * C, *tr = x + ((n * Hi + ylo[oy]) * Wi + xhi[ox]) * C;
        const REAL *bl = x + ((n * Hi + yhi[oy]) * Wi + xlo[ox]) * C, *br = x + ((n * Hi + yhi[oy]) * Wi + xhi[ox]) * C;
        REAL *o = y + ((n * Ho + oy) * Wo + ox) * C;
        for (long c = 0; c < C; c++) {
          const REAL top = tl[c] + (tr[c] - tl[c]) * wx[ox], bot = bl[c] + (br[c] - bl[c]) * wx[ox];
          o[c] = top + (bot - top) * wy[oy];
        }
      }
}

/* transpose of the above: scatter per image (images in parallel, sequential inside: deterministic) */
void FN(resize_bwd)(const REAL *g, REAL *dx, const long *ylo, const long *yhi, const REAL *wy, const long *xlo,
                    const long *xhi, const REAL *wx, long N, long Hi, long Wi, long Ho, long Wo, long C) {
#pragma omp parallel for schedule(static)
  for (long n = 0; n < N; n++) {
    REAL *d = dx + n * Hi * Wi * C;
    for (long i = 0; i < Hi * Wi * C; i++) d[i] = 0;
    for (long oy = 0; oy < Ho; oy++)
      for (long ox = 0; ox < Wo; ox++) {
        const REAL *go = g + ((n * Ho + oy) * Wo + ox) * C;
        REAL *tl = d + (ylo[oy] * Wi + xlo[ox]) * C, *tr = d + (ylo[oy] * Wi + xhi[ox]) * C;
        REAL *bl = d + (yhi[oy] * Wi + xlo[ox]) * C, *br = d + (yhi[oy] * Wi + xhi[ox]) * C;
        const REAL a = wx[ox], b = wy[oy];
        for (long c = 0; c < C; c++) {
          const REAL v = go[c];
          tl[c] += v * (1 - a) * (1 - b);
          tr[c] += v * a * (1 - b);
          bl[c] += v * (1 - a) * b;
          br[c] += v * a * b;
        }
      }
  }
}
