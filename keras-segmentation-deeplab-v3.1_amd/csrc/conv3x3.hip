// conv3x3.hip — dense Conv2D 3x3 for the three stem convolutions (deeplabv3p.py:283 entry_flow_conv1_1,
// :289 entry_flow_conv1_2, :318 Conv).  Cin is 3 or 32: direct convolution on the vector ALU (the
// north star keeps MFMA for the 1x1 GEMMs only); the input scale x/127.5-1 (deeplabv3p.py:270)
// arrives as the input transform and is applied on load, so the raw 0-255 image is read once.
// Thread = (4 output channels, output pixel); weights [3][3][Cin][Cout] are read as float4 along
// Cout (L1 resident: 3.4 KB for the MobileNetV2 stem).
#include "common.h"

namespace {

struct CGeom {
  int N, H, W, Cin, Cout, stride, pad_t, pad_l, Ho, Wo;
};

// block-level fixed-order reduction of NV floats over the pixel lanes that share a channel quad.
// tid = pl*CQ + cq.  Result valid for tid < CQ.
template <int NV>
__device__ __forceinline__ void reduce_over_pixels(float (&v)[NV], float *lds /* [256][NV] */, int CQ) {
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NV; i++) lds[threadIdx.x * NV + i] = v[i];
  __syncthreads();
  if ((int)threadIdx.x < CQ) {
    const int PL = 256 / CQ;
#pragma unroll
    for (int i = 0; i < NV; i++) {
      float s = 0.f;
      for (int q = 0; q < PL; q++) s += lds[(q * CQ + threadIdx.x) * NV + i];
      v[i] = s;
    }
  }
}

__global__ __launch_bounds__(256) void conv3x3_fwd_kernel(const float *__restrict__ x, const float *__restrict__ sc,
                                                          const float *__restrict__ sh, int act,
                                                          const float *__restrict__ w, float *__restrict__ y,
                                                          CGeom G, float *__restrict__ part) {
  __shared__ float red[256 * 8];
  const int CQ = G.Cout / 4;
  const int cq = threadIdx.x % CQ, pl = threadIdx.x / CQ, PL = 256 / CQ;
  const int co = cq * 4;
  f32x4 s1 = splat4(0.f), s2 = splat4(0.f);
  const long NP = (long)G.N * G.Ho * G.Wo;
  for (long p = (long)blockIdx.x * PL + pl; p < NP; p += (long)gridDim.x * PL) {
    const int ox = (int)(p % G.Wo);
    const int oy = (int)((p / G.Wo) % G.Ho);
    const int n = (int)(p / ((long)G.Wo * G.Ho));
    f32x4 acc = splat4(0.f);
    // clamped tap coordinates + 0/1 weights instead of branches: all loads of a pixel are issued together
#pragma unroll
    for (int i = 0; i < 3; i++) {
      const int iy = oy * G.stride - G.pad_t + i;
      const int iyc = min(max(iy, 0), G.H - 1);
#pragma unroll
      for (int j = 0; j < 3; j++) {
        const int ix = ox * G.stride - G.pad_l + j;
        const int ixc = min(max(ix, 0), G.W - 1);
        const float live = (iy == iyc && ix == ixc) ? 1.f : 0.f;
        const float *xp = x + (((size_t)n * G.H + iyc) * G.W + ixc) * G.Cin;
        const float *wp = w + ((size_t)(i * 3 + j) * G.Cin) * G.Cout + co;
        for (int ci = 0; ci < G.Cin; ci++) {
          float v = xp[ci];
          if (sc) v = sc[ci] * v + sh[ci];
          v = dl3_act(v, act) * live;
          acc += splat4(v) * ld4(wp + (size_t)ci * G.Cout);
        }
      }
    }
    st4(y + (size_t)p * G.Cout + co, acc);
    s1 += acc;
    s2 += acc * acc;
  }
  if (part) {
    float v[8] = {s1.x, s1.y, s1.z, s1.w, s2.x, s2.y, s2.z, s2.w};
    reduce_over_pixels<8>(v, red, CQ);
    if ((int)threadIdx.x < CQ) {
      float *d = part + ((size_t)blockIdx.x * G.Cout + co) * 2;
      d[0] = v[0]; d[1] = v[4]; d[2] = v[1]; d[3] = v[5];
      d[4] = v[2]; d[5] = v[6]; d[6] = v[3]; d[7] = v[7];
    }
  }
}

// dW partial [gridDim.x][3][3][Cin][Cout]; blockIdx.y selects a chunk of 3 input channels
constexpr int CI_CHUNK = 3;
__global__ __launch_bounds__(256) void conv3x3_wgrad_kernel(const float *__restrict__ x,
                                                            const float *__restrict__ sc,
                                                            const float *__restrict__ sh, int act,
                                                            const float *__restrict__ g,
                                                            const float *__restrict__ yraw,
                                                            const float *__restrict__ cA,
                                                            const float *__restrict__ cB,
                                                            const float *__restrict__ cC, float *__restrict__ wpart,
                                                            CGeom G) {
  __shared__ float red[256 * 12];
  const int CQ = G.Cout / 4;
  const int cq = threadIdx.x % CQ, pl = threadIdx.x / CQ, PL = 256 / CQ;
  const int co = cq * 4;
  const int ci0 = blockIdx.y * CI_CHUNK;
  f32x4 kA = splat4(1.f), kB = splat4(0.f), kC = splat4(0.f);
  const bool two = cA != nullptr;
  if (two) { kA = ld4(cA + co); kB = ld4(cB + co); kC = ld4(cC + co); }
  f32x4 acc[9][CI_CHUNK];
#pragma unroll
  for (int t = 0; t < 9; t++)
#pragma unroll
    for (int k = 0; k < CI_CHUNK; k++) acc[t][k] = splat4(0.f);
  const long NP = (long)G.N * G.Ho * G.Wo;
  for (long p = (long)blockIdx.x * PL + pl; p < NP; p += (long)gridDim.x * PL) {
    const int ox = (int)(p % G.Wo);
    const int oy = (int)((p / G.Wo) % G.Ho);
    const int n = (int)(p / ((long)G.Wo * G.Ho));
    f32x4 dd = ld4(g + (size_t)p * G.Cout + co);
    if (two) dd = kA * dd + kB * ld4(yraw + (size_t)p * G.Cout + co) + kC;
    float xv[9][CI_CHUNK];
#pragma unroll
    for (int i = 0; i < 3; i++) {
      const int iy = oy * G.stride - G.pad_t + i;
      const int iyc = min(max(iy, 0), G.H - 1);
#pragma unroll
      for (int j = 0; j < 3; j++) {
        const int ix = ox * G.stride - G.pad_l + j;
        const int ixc = min(max(ix, 0), G.W - 1);
        const float live = (iy == iyc && ix == ixc) ? 1.f : 0.f;
        const float *xp = x + (((size_t)n * G.H + iyc) * G.W + ixc) * G.Cin;
#pragma unroll
        for (int k = 0; k < CI_CHUNK; k++) {
          const int ci = min(ci0 + k, G.Cin - 1);
          float v = xp[ci];
          if (sc) v = sc[ci] * v + sh[ci];
          xv[i * 3 + j][k] = dl3_act(v, act) * ((ci0 + k < G.Cin) ? live : 0.f);
        }
      }
    }
#pragma unroll
    for (int t = 0; t < 9; t++)
#pragma unroll
      for (int k = 0; k < CI_CHUNK; k++) acc[t][k] += splat4(xv[t][k]) * dd;
  }
#pragma unroll
  for (int t = 0; t < 9; t++) {
    float v[4 * CI_CHUNK];
#pragma unroll
    for (int k = 0; k < CI_CHUNK; k++) {
      v[k * 4 + 0] = acc[t][k].x; v[k * 4 + 1] = acc[t][k].y;
      v[k * 4 + 2] = acc[t][k].z; v[k * 4 + 3] = acc[t][k].w;
    }
    reduce_over_pixels<4 * CI_CHUNK>(v, red, CQ);
    if ((int)threadIdx.x < CQ) {
#pragma unroll
      for (int k = 0; k < CI_CHUNK; k++) {
        const int ci = ci0 + k;
        if (ci < G.Cin) {
          f32x4 o = {v[k * 4], v[k * 4 + 1], v[k * 4 + 2], v[k * 4 + 3]};
          st4(wpart + (((size_t)blockIdx.x * 9 + t) * G.Cin + ci) * G.Cout + co, o);
        }
      }
    }
  }
}

// bwd-data (xception entry_flow_conv1_2 only): thread = (input pixel, 4 input channels)
__global__ __launch_bounds__(256) void conv3x3_dgrad_kernel(
    const float *__restrict__ g, const float *__restrict__ yraw, const float *__restrict__ cA,
    const float *__restrict__ cB, const float *__restrict__ cC, const float *__restrict__ w, float *__restrict__ dx,
    const float *__restrict__ x, const float *__restrict__ sc, const float *__restrict__ sh, int act,
    const float *__restrict__ dx_add, const float *__restrict__ xmean, const float *__restrict__ xinvstd,
    float *__restrict__ part, CGeom G) {
  __shared__ float red[256 * 8];
  const int CQ = G.Cin / 4;
  const int cq = threadIdx.x % CQ, pl = threadIdx.x / CQ, PL = 256 / CQ;
  const int ci = cq * 4;
  const bool two = cA != nullptr;
  f32x4 s = splat4(1.f), t = splat4(0.f), mu = splat4(0.f), is = splat4(0.f);
  if (sc) { s = ld4(sc + ci); t = ld4(sh + ci); }
  if (part) { mu = ld4(xmean + ci); is = ld4(xinvstd + ci); }
  f32x4 s1 = splat4(0.f), s2 = splat4(0.f);
  const long NP = (long)G.N * G.H * G.W;
  for (long p = (long)blockIdx.x * PL + pl; p < NP; p += (long)gridDim.x * PL) {
    const int ix = (int)(p % G.W);
    const int iy = (int)((p / G.W) % G.H);
    const int n = (int)(p / ((long)G.W * G.H));
    f32x4 acc = splat4(0.f);
    for (int i = 0; i < 3; i++) {
      const int ty = iy + G.pad_t - i;
      if (ty < 0 || (ty % G.stride) != 0) continue;
      const int oy = ty / G.stride;
      if (oy >= G.Ho) continue;
      for (int j = 0; j < 3; j++) {
        const int tx = ix + G.pad_l - j;
        if (tx < 0 || (tx % G.stride) != 0) continue;
        const int ox = tx / G.stride;
        if (ox >= G.Wo) continue;
        const size_t off = (((size_t)n * G.Ho + oy) * G.Wo + ox) * G.Cout;
        const float *wp = w + ((size_t)(i * 3 + j) * G.Cin + ci) * G.Cout;
        for (int co = 0; co < G.Cout; co++) {
          float dd = g[off + co];
          if (two) dd = cA[co] * dd + cB[co] * yraw[off + co] + cC[co];
          f32x4 wv = {wp[co], wp[G.Cout + co], wp[2 * (size_t)G.Cout + co], wp[3 * (size_t)G.Cout + co]};
          acc += splat4(dd) * wv;
        }
      }
    }
    f32x4 out = acc;
    f32x4 xr = splat4(0.f);
    if (x) {
      xr = ld4(x + (size_t)p * G.Cin + ci);
      out = out * dl3_mask4(s * xr + t, act);
    }
    if (dx_add) out += ld4(dx_add + (size_t)p * G.Cin + ci);
    st4(dx + (size_t)p * G.Cin + ci, out);
    s1 += out;
    s2 += out * ((xr - mu) * is);
  }
  if (part) {
    float v[8] = {s1.x, s1.y, s1.z, s1.w, s2.x, s2.y, s2.z, s2.w};
    reduce_over_pixels<8>(v, red, CQ);
    if ((int)threadIdx.x < CQ) {
      float *d = part + ((size_t)blockIdx.x * G.Cin + ci) * 2;
      d[0] = v[0]; d[1] = v[4]; d[2] = v[1]; d[3] = v[5];
      d[4] = v[2]; d[5] = v[6]; d[6] = v[3]; d[7] = v[7];
    }
  }
}

// ---------------------------------------------------------------------------------------
// The two stem convolutions on the raw image (Cin = 3 -> 32: deeplabv3p.py:283 entry_flow_conv1_1, :318 Conv).
// ---------------------------------------------------------------------------------------
// Forward on the matrix pipe (default for Cin*9 <= 32, Cout == 32): y[p][co] = sum_t col[p][t] * W[t][co] with the
// 27 (padded to 28) taps as the reduction axis of 14 v_mfma_f32_32x32x2_f32 steps per 32-pixel tile.  The column
// matrix is never formed: lane (pixel = lane & 31, half = lane >> 5) gathers tap 2s+half of its own pixel as the A
// operand; the weights sit in 14 registers per lane for the whole kernel.  The C layout has one output channel per
// lane, so every store instruction writes two full 128-byte pixel rows and the BN sums are in-lane.
template <int COUT>
__global__ __launch_bounds__(256) void conv3x3_stem_fwd_mfma_kernel(const float *__restrict__ x,
                                                                    const float *__restrict__ sc,
                                                                    const float *__restrict__ sh, int act,
                                                                    const float *__restrict__ w,
                                                                    float *__restrict__ y, CGeom G,
                                                                    float *__restrict__ part) {
  static_assert(COUT == 32, "one 32-column MFMA tile");
  constexpr int KS = 14;  // k-steps: 28 >= 9*Cin taps
  __shared__ float red[4][2 * COUT];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, lhi = lane >> 5;
  const int T = 9 * G.Cin;
  float bw[KS], fs[KS], ft[KS];
  int dy[KS], dx_[KS], dc[KS];
#pragma unroll
  for (int s = 0; s < KS; s++) {
    const int kk = 2 * s + lhi, t = min(kk, T - 1);
    bw[s] = (kk < T) ? w[(size_t)t * COUT + l31] : 0.f;
    dy[s] = t / (3 * G.Cin);
    dx_[s] = (t / G.Cin) % 3;
    dc[s] = t % G.Cin;
    fs[s] = sc ? sc[dc[s]] : 1.f;
    ft[s] = sc ? sh[dc[s]] : 0.f;
    if (kk >= T) { fs[s] = 0.f; ft[s] = 0.f; }  // act(0) = 0: the padding tap contributes nothing
  }
  float s1 = 0.f, s2 = 0.f;
  const long NP = (long)G.N * G.Ho * G.Wo;
  const long ntiles = (NP + 31) / 32;
  for (long tile = (long)blockIdx.x * 4 + wave; tile < ntiles; tile += (long)gridDim.x * 4) {
    const long p = min(tile * 32 + l31, NP - 1);
    const int ox = (int)(p % G.Wo), oy = (int)((p / G.Wo) % G.Ho), n = (int)(p / ((long)G.Wo * G.Ho));
    const int iy0 = oy * G.stride - G.pad_t, ix0 = ox * G.stride - G.pad_l;
    float av[KS];
#pragma unroll
    for (int s = 0; s < KS; s++) {
      const int iy = iy0 + dy[s], ix = ix0 + dx_[s];
      const int iyc = min(max(iy, 0), G.H - 1), ixc = min(max(ix, 0), G.W - 1);
      const float live = (iy == iyc && ix == ixc) ? 1.f : 0.f;
      av[s] = x[(((size_t)n * G.H + iyc) * G.W + ixc) * G.Cin + dc[s]] * live;
      // (live scales the raw value; the shift ft is masked below)
      av[s] = dl3_act(fs[s] * av[s] + ft[s] * live, act);
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < KS; s++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bw[s], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const long row = tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
      if (row < NP) {
        __builtin_nontemporal_store(acc[r], &y[(size_t)row * COUT + l31]);
        s1 += acc[r];
        s2 += acc[r] * acc[r];
      }
    }
  }
  if (part) {
    s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 32, 64);
    if (lhi == 0) { red[wave][2 * l31] = s1; red[wave][2 * l31 + 1] = s2; }
    __syncthreads();
    if (threadIdx.x < 2 * COUT)
      part[(size_t)blockIdx.x * COUT * 2 + threadIdx.x] =
          red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
  }
}

// Weight gradient of the same convolutions on the matrix pipe: dW[t][co] = sum_p T(x)[p, tap t] * dY[p][co] with
// t = (i*3+j)*Cin + ci < 32 rows of one v_mfma_f32_32x32x2_f32 tile and the pixel index p as the reduction axis
// (two pixels per MFMA).  A lane supplies, for its own pixel (parity = lane >> 5), the input value under tap
// (lane & 31) on the A side and dY of channel (lane & 31) on the B side: three dword loads per MFMA, no LDS.
// grid = conv_blocks; every workgroup owns a contiguous range of output pixels, its 4 waves interleave pixel pairs.
template <int COUT>
__global__ __launch_bounds__(256) void conv3x3_stem_wgrad_kernel(
    const float *__restrict__ x, const float *__restrict__ sc, const float *__restrict__ sh, int act,
    const float *__restrict__ g, const float *__restrict__ yraw, const float *__restrict__ cA,
    const float *__restrict__ cB, const float *__restrict__ cC, float *__restrict__ wpart, CGeom G) {
  constexpr int NJ = COUT / 32, U = 4;
  __shared__ float red[4][32][COUT];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, lhi = lane >> 5;
  const int T = 9 * G.Cin;
  const int t = min(l31, T - 1), ti = t / (3 * G.Cin), tj = (t / G.Cin) % 3, tc = t % G.Cin;
  const float tlive = (l31 < T) ? 1.f : 0.f;
  const float fs = sc ? sc[tc] : 1.f, ft = sc ? sh[tc] : 0.f;
  const bool two = cA != nullptr;
  float kA[NJ], kB[NJ], kC[NJ];
#pragma unroll
  for (int j = 0; j < NJ; j++) {
    kA[j] = two ? cA[j * 32 + l31] : 1.f;
    kB[j] = two ? cB[j * 32 + l31] : 0.f;
    kC[j] = two ? cC[j * 32 + l31] : 0.f;
  }
  const float *yr = two ? yraw : g;
  f32x16 acc[NJ];
#pragma unroll
  for (int j = 0; j < NJ; j++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[j][r] = 0.f;

  const long NP = (long)G.N * G.Ho * G.Wo;
  const long per = ((NP + gridDim.x - 1) / gridDim.x + 7) & ~7l;  // pixels per workgroup, multiple of 8
  const long pbeg = (long)blockIdx.x * per, pend = min(NP, pbeg + per);
  // wave w takes pixel pairs w, w+4, w+8, ... of the range; U pairs are in flight per iteration
  for (long q = pbeg + 2 * wave; q < pend; q += 8 * U) {
    float av[U], gv[U][NJ], yv[U][NJ], lv[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const long p = q + 8 * u + lhi;
      const long pc = min(p, NP - 1);
      const int ox = (int)(pc % G.Wo), oy = (int)((pc / G.Wo) % G.Ho), n = (int)(pc / ((long)G.Wo * G.Ho));
      const int iy = oy * G.stride - G.pad_t + ti, ix = ox * G.stride - G.pad_l + tj;
      const int iyc = min(max(iy, 0), G.H - 1), ixc = min(max(ix, 0), G.W - 1);
      lv[u] = (p < pend && iy == iyc && ix == ixc) ? tlive : 0.f;
      av[u] = x[(((size_t)n * G.H + iyc) * G.W + ixc) * G.Cin + tc];
#pragma unroll
      for (int j = 0; j < NJ; j++) {
        gv[u][j] = g[(size_t)pc * COUT + j * 32 + l31];
        yv[u][j] = yr[(size_t)pc * COUT + j * 32 + l31];
      }
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const float a = dl3_act(fs * av[u] + ft, act) * lv[u];
#pragma unroll
      for (int j = 0; j < NJ; j++) {
        const float b = kA[j] * gv[u][j] + kB[j] * yv[u][j] + kC[j];
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
      }
    }
  }
  // fixed-order sum of the 4 waves, then one partial row [9*Cin][COUT] per workgroup
#pragma unroll
  for (int j = 0; j < NJ; j++)
#pragma unroll
    for (int r = 0; r < 16; r++) red[wave][(r & 3) + 8 * (r >> 2) + 4 * lhi][j * 32 + l31] = acc[j][r];
  __syncthreads();
  for (int i = threadIdx.x; i < T * COUT; i += 256) {
    const int row = i / COUT, col = i % COUT;
    wpart[(size_t)blockIdx.x * T * COUT + i] = red[0][row][col] + red[1][row][col] + red[2][row][col] + red[3][row][col];
  }
}

// ---------------------------------------------------------------------------------------
// MFMA route for a dense 3x3 conv with many input channels (xception entry_flow_conv1_2: 32 -> 64 at 256x256,
// 1.2 GMAC / image, deeplabv3p.py:289): im2col -> the 1x1-conv GEMM kernels -> col2im.  The column matrix
// [N*Ho*Wo][9*Cin] costs one extra write + read of 9x the input, which the matrix pipe wins back ~10x over.
// ---------------------------------------------------------------------------------------
// col[m][(i*3+j)*Cin + c] = T(x)[n, oy*stride-pad_t+i, ox*stride-pad_l+j, c] (0 outside the image).
// thread = one float4 of one (output pixel, tap); consecutive threads walk the row of the column matrix.
__global__ __launch_bounds__(256) void im2col3x3_kernel(const float *__restrict__ x, const float *__restrict__ sc,
                                                        const float *__restrict__ sh, int act,
                                                        float *__restrict__ col, CGeom G) {
  const int CQ = G.Cin / 4, RQ = 9 * CQ;
  const long total = (long)G.N * G.Ho * G.Wo * RQ;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int q = (int)(idx % RQ);
    const long m = idx / RQ;
    const int tap = q / CQ, c = (q % CQ) * 4;
    const int ox = (int)(m % G.Wo), oy = (int)((m / G.Wo) % G.Ho), n = (int)(m / ((long)G.Wo * G.Ho));
    const int iy = oy * G.stride - G.pad_t + tap / 3, ix = ox * G.stride - G.pad_l + tap % 3;
    const int iyc = min(max(iy, 0), G.H - 1), ixc = min(max(ix, 0), G.W - 1);
    const float live = (iy == iyc && ix == ixc) ? 1.f : 0.f;
    f32x4 v = ld4(x + (((size_t)n * G.H + iyc) * G.W + ixc) * G.Cin + c);
    if (sc) v = ld4(sc + c) * v + ld4(sh + c);
    v = dl3_act4(v, act) * splat4(live);
    st4_nt(col + (size_t)idx * 4, v);
  }
}

// dx[n,iy,ix,c] = mask(x) * sum over the taps (i,j) whose output pixel exists of dcol[(n,oy,ox)][(i*3+j)*Cin + c]
// (+ dx_add); deterministic gather form of the transposed im2col.  Same epilogue / partial layout as the direct
// bwd-data kernel: thread = (input pixel, 4 input channels).
__global__ __launch_bounds__(256) void col2im3x3_kernel(const float *__restrict__ dcol, float *__restrict__ dx,
                                                        const float *__restrict__ x, const float *__restrict__ sc,
                                                        const float *__restrict__ sh, int act,
                                                        const float *__restrict__ dx_add,
                                                        const float *__restrict__ xmean,
                                                        const float *__restrict__ xinvstd, float *__restrict__ part,
                                                        CGeom G) {
  __shared__ float red[256 * 8];
  const int CQ = G.Cin / 4;
  const int cq = threadIdx.x % CQ, pl = threadIdx.x / CQ, PL = 256 / CQ;
  const int ci = cq * 4;
  f32x4 s = splat4(1.f), t = splat4(0.f), mu = splat4(0.f), is = splat4(0.f);
  if (sc) { s = ld4(sc + ci); t = ld4(sh + ci); }
  if (part) { mu = ld4(xmean + ci); is = ld4(xinvstd + ci); }
  f32x4 s1 = splat4(0.f), s2 = splat4(0.f);
  const long NP = (long)G.N * G.H * G.W;
  const size_t ldc = (size_t)9 * G.Cin;
  for (long p = (long)blockIdx.x * PL + pl; p < NP; p += (long)gridDim.x * PL) {
    const int ix = (int)(p % G.W), iy = (int)((p / G.W) % G.H), n = (int)(p / ((long)G.W * G.H));
    f32x4 acc = splat4(0.f);
#pragma unroll
    for (int i = 0; i < 3; i++) {
      const int ty = iy + G.pad_t - i;
      const int oy = ty / G.stride;
      const bool yok = ty >= 0 && (ty % G.stride) == 0 && oy < G.Ho;
      const int oyc = min(max(oy, 0), G.Ho - 1);
#pragma unroll
      for (int j = 0; j < 3; j++) {
        const int tx = ix + G.pad_l - j;
        const int ox = tx / G.stride;
        const bool ok = yok && tx >= 0 && (tx % G.stride) == 0 && ox < G.Wo;
        const int oxc = min(max(ox, 0), G.Wo - 1);
        const f32x4 v = ld4(dcol + (((size_t)n * G.Ho + oyc) * G.Wo + oxc) * ldc + (i * 3 + j) * G.Cin + ci);
        acc += v * splat4(ok ? 1.f : 0.f);
      }
    }
    f32x4 out = acc, xr = splat4(0.f);
    if (x) {
      xr = ld4(x + (size_t)p * G.Cin + ci);
      out = out * dl3_mask4(s * xr + t, act);
    }
    if (dx_add) out += ld4(dx_add + (size_t)p * G.Cin + ci);
    st4(dx + (size_t)p * G.Cin + ci, out);
    s1 += out;
    s2 += out * ((xr - mu) * is);
  }
  if (part) {
    float v[8] = {s1.x, s1.y, s1.z, s1.w, s2.x, s2.y, s2.z, s2.w};
    reduce_over_pixels<8>(v, red, CQ);
    if ((int)threadIdx.x < CQ) {
      float *d = part + ((size_t)blockIdx.x * G.Cin + ci) * 2;
      d[0] = v[0]; d[1] = v[4]; d[2] = v[1]; d[3] = v[5];
      d[4] = v[2]; d[5] = v[6]; d[6] = v[3]; d[7] = v[7];
    }
  }
}

int conv_blocks(long NP) {
  long b = (NP + 31) / 32;
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  return (int)b;
}

int conv_check(const char *name, const CGeom &G) {
  DL3_CHECK_ARG(G.N > 0 && G.H > 0 && G.W > 0 && G.Cin > 0 && G.Cout > 0 && G.Ho > 0 && G.Wo > 0 && G.stride >= 1,
                "%s: bad dimension", name);
  DL3_UNSUPPORTED(G.Cout % 4 != 0 || 256 % (G.Cout / 4) != 0 || G.Cout / 4 > 256,
                  "%s: Cout=%d must be 4*(a divisor of 256)", name, G.Cout);
  return DL3_OK;
}

}  // namespace

extern "C" int dl3_conv3x3_partials(int N, int Ho, int Wo, int Cout) {
  (void)Cout;
  return conv_blocks((long)N * Ho * Wo);
}

extern "C" int dl3_conv3x3_fwd(const float *x, const float *in_scale, const float *in_shift, int in_act,
                               const float *w, float *y, int N, int H, int W, int Cin, int Cout, int stride,
                               int pad_t, int pad_l, int Ho, int Wo, float *stat_partial, void *stream) {
  CGeom G{N, H, W, Cin, Cout, stride, pad_t, pad_l, Ho, Wo};
  int rc = conv_check("conv3x3_fwd", G);
  if (rc) return rc;
  DL3_CHECK_ARG(x && w && y, "conv3x3_fwd: null pointer");
  DL3_CHECK_ARG((in_scale == nullptr) == (in_shift == nullptr), "conv3x3_fwd: scale/shift must come together");
  if (9 * Cin <= 28 && Cout == 32)
    hipLaunchKernelGGL((conv3x3_stem_fwd_mfma_kernel<32>), dim3(conv_blocks((long)N * Ho * Wo)), dim3(256), 0,
                       (hipStream_t)stream, x, in_scale, in_shift, in_act, w, y, G, stat_partial);
  else
    hipLaunchKernelGGL(conv3x3_fwd_kernel, dim3(conv_blocks((long)N * Ho * Wo)), dim3(256), 0, (hipStream_t)stream,
                       x, in_scale, in_shift, in_act, w, y, G, stat_partial);
  DL3_LAUNCH_CHECK("conv3x3_fwd");
  return DL3_OK;
}

extern "C" int dl3_conv3x3_bwd_weight(const float *x, const float *in_scale, const float *in_shift, int in_act,
                                      const float *g, const float *yraw, const float *cA, const float *cB,
                                      const float *cC, float *dw_partial, int N, int H, int W, int Cin, int Cout,
                                      int stride, int pad_t, int pad_l, int Ho, int Wo, void *stream) {
  CGeom G{N, H, W, Cin, Cout, stride, pad_t, pad_l, Ho, Wo};
  int rc = conv_check("conv3x3_bwd_weight", G);
  if (rc) return rc;
  DL3_CHECK_ARG(x && g && dw_partial, "conv3x3_bwd_weight: null pointer");
  DL3_CHECK_ARG(!cA || (yraw && cB && cC), "conv3x3_bwd_weight: cA needs yraw, cB, cC");
  DL3_CHECK_ARG((in_scale == nullptr) == (in_shift == nullptr), "conv3x3_bwd_weight: scale/shift must come together");
  if (9 * Cin <= 32 && Cout == 32) {
    hipLaunchKernelGGL((conv3x3_stem_wgrad_kernel<32>), dim3(conv_blocks((long)N * Ho * Wo)), dim3(256), 0,
                       (hipStream_t)stream, x, in_scale, in_shift, in_act, g, yraw, cA, cB, cC, dw_partial, G);
  } else {
    dim3 grid(conv_blocks((long)N * Ho * Wo), dl3_cdiv(Cin, CI_CHUNK));
    hipLaunchKernelGGL(conv3x3_wgrad_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, in_scale, in_shift, in_act,
                       g, yraw, cA, cB, cC, dw_partial, G);
  }
  DL3_LAUNCH_CHECK("conv3x3_bwd_weight");
  return DL3_OK;
}

extern "C" int dl3_conv3x3_bwd_data(const float *g, const float *yraw, const float *cA, const float *cB,
                                    const float *cC, const float *w, float *dx, const float *x,
                                    const float *in_scale, const float *in_shift, int in_act, const float *dx_add,
                                    const float *x_mean, const float *x_invstd, float *dstat_partial, int N, int H,
                                    int W, int Cin, int Cout, int stride, int pad_t, int pad_l, int Ho, int Wo,
                                    void *stream) {
  CGeom G{N, H, W, Cin, Cout, stride, pad_t, pad_l, Ho, Wo};
  DL3_CHECK_ARG(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && Ho > 0 && Wo > 0 && stride >= 1,
                "conv3x3_bwd_data: bad dimension");
  DL3_UNSUPPORTED(Cin % 4 != 0 || 256 % (Cin / 4) != 0, "conv3x3_bwd_data: Cin=%d must be 4*(a divisor of 256)", Cin);
  DL3_CHECK_ARG(g && w && dx, "conv3x3_bwd_data: null pointer");
  DL3_CHECK_ARG(!cA || (yraw && cB && cC), "conv3x3_bwd_data: cA needs yraw, cB, cC");
  DL3_CHECK_ARG(in_act == DL3_ACT_NONE || x, "conv3x3_bwd_data: activation mask needs x");
  DL3_CHECK_ARG(!dstat_partial || (x && x_mean && x_invstd), "conv3x3_bwd_data: dstat needs x, x_mean, x_invstd");
  const int blocks = conv_blocks((long)N * H * W);
  hipLaunchKernelGGL(conv3x3_dgrad_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g, yraw, cA, cB, cC, w,
                     dx, x, in_scale, in_shift, in_act, dx_add, x_mean, x_invstd, dstat_partial, G);
  DL3_LAUNCH_CHECK("conv3x3_bwd_data");
  return DL3_OK;
}

// ---- MFMA route (im2col + the pointwise GEMM kernels) -----------------------------------
namespace {
size_t col_bytes(int N, int Ho, int Wo, int Cin) { return (size_t)N * Ho * Wo * 9 * Cin * sizeof(float); }
size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }
int gemm_check(const char *name, const CGeom &G) {
  DL3_CHECK_ARG(G.N > 0 && G.H > 0 && G.W > 0 && G.Cin > 0 && G.Cout > 0 && G.Ho > 0 && G.Wo > 0 && G.stride >= 1,
                "%s: bad dimension", name);
  DL3_UNSUPPORTED(G.Cin % 4 != 0 || 256 % (G.Cin / 4) != 0 || G.Cout % 4 != 0,
                  "%s: needs Cin = 4*(a divisor of 256) and Cout %% 4 == 0 (got %d, %d)", name, G.Cin, G.Cout);
  DL3_UNSUPPORTED((long)G.N * G.Ho * G.Wo * 9 * G.Cin >= (1l << 31) * 4, "%s: column matrix too large", name);
  return DL3_OK;
}
int im2col_blocks(long total) {
  long b = (total + 255) / 256;
  return (int)(b > 16384 ? 16384 : b);
}
}  // namespace

extern "C" size_t dl3_conv3x3_gemm_workspace(int N, int H, int W, int Cin, int Cout, int stride, int Ho, int Wo) {
  (void)H; (void)W; (void)stride;
  if (N <= 0 || Ho <= 0 || Wo <= 0 || Cin <= 0 || Cout <= 0) return 0;
  return align256(col_bytes(N, Ho, Wo, Cin)) + dl3_pwconv_bwd_weight_workspace(N * Ho * Wo, 9 * Cin, Cout);
}

extern "C" int dl3_conv3x3_gemm_fwd(const float *x, const float *in_scale, const float *in_shift, int in_act,
                                    const float *w, float *y, int N, int H, int W, int Cin, int Cout, int stride,
                                    int pad_t, int pad_l, int Ho, int Wo, float *stat_partial, void *workspace,
                                    size_t workspace_bytes, void *stream) {
  CGeom G{N, H, W, Cin, Cout, stride, pad_t, pad_l, Ho, Wo};
  int rc = gemm_check("conv3x3_gemm_fwd", G);
  if (rc) return rc;
  DL3_CHECK_ARG(x && w && y && workspace, "conv3x3_gemm_fwd: null pointer");
  DL3_CHECK_ARG((in_scale == nullptr) == (in_shift == nullptr), "conv3x3_gemm_fwd: scale/shift must come together");
  if (workspace_bytes < col_bytes(N, Ho, Wo, Cin)) {
    dl3_set_error("conv3x3_gemm_fwd: workspace %zu < %zu bytes", workspace_bytes, col_bytes(N, Ho, Wo, Cin));
    return DL3_EWORKSPACE;
  }
  float *col = (float *)workspace;
  const long M = (long)N * Ho * Wo;
  hipLaunchKernelGGL(im2col3x3_kernel, dim3(im2col_blocks(M * 9 * (Cin / 4))), dim3(256), 0, (hipStream_t)stream, x,
                     in_scale, in_shift, in_act, col, G);
  DL3_LAUNCH_CHECK("conv3x3_gemm_fwd(im2col)");
  return dl3_pwconv_fwd(col, 9 * Cin, nullptr, nullptr, DL3_ACT_NONE, w, nullptr, y, Cout, (int)M, 9 * Cin, Cout,
                        stat_partial, stream);
}

extern "C" int dl3_conv3x3_gemm_bwd_weight(const float *x, const float *in_scale, const float *in_shift, int in_act,
                                           const float *g, const float *yraw, const float *cA, const float *cB,
                                           const float *cC, float *dw, int N, int H, int W, int Cin, int Cout,
                                           int stride, int pad_t, int pad_l, int Ho, int Wo, void *workspace,
                                           size_t workspace_bytes, void *stream) {
  CGeom G{N, H, W, Cin, Cout, stride, pad_t, pad_l, Ho, Wo};
  int rc = gemm_check("conv3x3_gemm_bwd_weight", G);
  if (rc) return rc;
  DL3_CHECK_ARG(x && g && dw && workspace, "conv3x3_gemm_bwd_weight: null pointer");
  DL3_CHECK_ARG(!cA || (yraw && cB && cC), "conv3x3_gemm_bwd_weight: cA needs yraw, cB, cC");
  const size_t need = dl3_conv3x3_gemm_workspace(N, H, W, Cin, Cout, stride, Ho, Wo);
  if (workspace_bytes < need) {
    dl3_set_error("conv3x3_gemm_bwd_weight: workspace %zu < %zu bytes", workspace_bytes, need);
    return DL3_EWORKSPACE;
  }
  float *col = (float *)workspace;
  const size_t cb = align256(col_bytes(N, Ho, Wo, Cin));
  const long M = (long)N * Ho * Wo;
  hipLaunchKernelGGL(im2col3x3_kernel, dim3(im2col_blocks(M * 9 * (Cin / 4))), dim3(256), 0, (hipStream_t)stream, x,
                     in_scale, in_shift, in_act, col, G);
  DL3_LAUNCH_CHECK("conv3x3_gemm_bwd_weight(im2col)");
  return dl3_pwconv_bwd_weight(col, 9 * Cin, nullptr, nullptr, DL3_ACT_NONE, g, Cout, yraw, Cout, cA, cB, cC, dw,
                               nullptr, (int)M, 9 * Cin, Cout, (char *)workspace + cb, workspace_bytes - cb, stream);
}

extern "C" int dl3_conv3x3_gemm_bwd_data(const float *g, const float *yraw, const float *cA, const float *cB,
                                         const float *cC, const float *wT, float *dx, const float *x,
                                         const float *in_scale, const float *in_shift, int in_act,
                                         const float *dx_add, const float *x_mean, const float *x_invstd,
                                         float *dstat_partial, int N, int H, int W, int Cin, int Cout, int stride,
                                         int pad_t, int pad_l, int Ho, int Wo, void *workspace,
                                         size_t workspace_bytes, void *stream) {
  CGeom G{N, H, W, Cin, Cout, stride, pad_t, pad_l, Ho, Wo};
  int rc = gemm_check("conv3x3_gemm_bwd_data", G);
  if (rc) return rc;
  DL3_CHECK_ARG(g && wT && dx && workspace, "conv3x3_gemm_bwd_data: null pointer");
  DL3_CHECK_ARG(!cA || (yraw && cB && cC), "conv3x3_gemm_bwd_data: cA needs yraw, cB, cC");
  DL3_CHECK_ARG(in_act == DL3_ACT_NONE || x, "conv3x3_gemm_bwd_data: activation mask needs x");
  DL3_CHECK_ARG(!dstat_partial || (x && x_mean && x_invstd), "conv3x3_gemm_bwd_data: dstat needs x, x_mean, x_invstd");
  if (workspace_bytes < col_bytes(N, Ho, Wo, Cin)) {
    dl3_set_error("conv3x3_gemm_bwd_data: workspace %zu < %zu bytes", workspace_bytes, col_bytes(N, Ho, Wo, Cin));
    return DL3_EWORKSPACE;
  }
  float *dcol = (float *)workspace;
  const long M = (long)N * Ho * Wo;
  rc = dl3_pwconv_bwd_data(g, Cout, yraw, Cout, cA, cB, cC, wT, dcol, 9 * Cin, nullptr, 0, nullptr, nullptr,
                           DL3_ACT_NONE, nullptr, 0, 1, 1.f, nullptr, nullptr, nullptr, (int)M, 9 * Cin, Cout, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(col2im3x3_kernel, dim3(conv_blocks((long)N * H * W)), dim3(256), 0, (hipStream_t)stream, dcol, dx,
                     x, in_scale, in_shift, in_act, dx_add, x_mean, x_invstd, dstat_partial, G);
  DL3_LAUNCH_CHECK("conv3x3_gemm_bwd_data(col2im)");
  return DL3_OK;
}
