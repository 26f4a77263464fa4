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
  int off[KS];                                   // tap offset from the pixel's top-left input element, in floats
  unsigned m_top = 0u, m_bot = 0u, m_left = 0u, m_right = 0u, m_valid = 0u;  // bit s: tap s sits in row 0 / row 2 / column 0 / column 2
#pragma unroll
  for (int s = 0; s < KS; s++) {
    const int kk = 2 * s + lhi, t = min(kk, T - 1);
    bw[s] = (kk < T) ? w[(size_t)t * COUT + l31] : 0.f;
    const int dy = t / (3 * G.Cin), dx = (t / G.Cin) % 3, dc = t % G.Cin;
    off[s] = (dy * G.W + dx) * G.Cin + dc;
    m_top |= (dy == 0) ? (1u << s) : 0u;
    m_bot |= (dy == 2) ? (1u << s) : 0u;
    m_left |= (dx == 0) ? (1u << s) : 0u;
    m_right |= (dx == 2) ? (1u << s) : 0u;
    m_valid |= (kk < T) ? (1u << s) : 0u;
    fs[s] = sc ? sc[dc] : 1.f;
    ft[s] = sc ? sh[dc] : 0.f;
  }
  float s1 = 0.f, s2 = 0.f;
  // Round 5: the tile loop is software-pipelined around its stores.  vmcnt retires in order and counts stores: with the
  // next tile's 14 gathers requested BEHIND this tile's 16 stores (and the stores inside `if (row < NP)`, which makes the
  // compiler wait with vmcnt(0)), every wave drained its own stores before it saw its next pixels.  Now: full tiles
  // store unconditionally, the next tile's gathers are in flight before the stores are issued (always requested — the
  // last tile asks for itself again — so that no load sits in a branch), the ragged last tile is peeled.  Pixel and
  // element arithmetic in 32 bits (the host checks the sizes), one offset and five bit masks per lane instead of three
  // index arrays.  Same sums in the same order: bit-identical outputs.
  const unsigned NP = (unsigned)((long)G.N * G.Ho * G.Wo);
  const unsigned nfull = NP / 32u, tstride = gridDim.x * 4u;
  const unsigned HoWo = (unsigned)G.Ho * (unsigned)G.Wo;
  float av[KS];
  unsigned live = 0u;
  auto gather = [&](unsigned tile) __attribute__((always_inline)) {
    const unsigned p = min(tile * 32u + (unsigned)l31, NP - 1u);
    const unsigned n = p / HoWo, rem = p - n * HoWo;
    const int oy = (int)(rem / (unsigned)G.Wo), ox = (int)(rem - (unsigned)oy * (unsigned)G.Wo);
    const int iy0 = oy * G.stride - G.pad_t, ix0 = ox * G.stride - G.pad_l;
    // the 3x3 window hangs over at most one edge per axis (pad <= 1, H, W >= 2: the host checks)
    const unsigned dead = (iy0 < 0 ? m_top : 0u) | (iy0 + 2 > G.H - 1 ? m_bot : 0u) | (ix0 < 0 ? m_left : 0u) |
                          (ix0 + 2 > G.W - 1 ? m_right : 0u);
    live = ~dead & m_valid;
    const int base = ((int)n * G.H + iy0) * G.W * G.Cin + ix0 * G.Cin;     // top-left element of the window
    const int safe = ((int)n * G.H + iy0 + 1) * G.W * G.Cin + (ix0 + 1) * G.Cin;  // its centre: always inside
#pragma unroll
    for (int s = 0; s < KS; s++) av[s] = x[(unsigned)((dead >> s) & 1u ? safe : base + off[s])];
  };
  auto product = [&](f32x16 &acc) __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < KS; s++) {
      const float lv = (live >> s) & 1u ? 1.f : 0.f;
      // (live scales the raw value and masks the shift: a padding tap contributes act(0) = 0)
      const float a = dl3_act(fs[s] * (av[s] * lv) + ft[s] * lv, act);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bw[s], acc, 0, 0, 0);
    }
  };
  unsigned tile = blockIdx.x * 4u + (unsigned)wave;
  if (nfull > 0u) gather(min(tile, nfull - 1u));
  // (the first tile's pixels are waited for HERE: entering the loop with loads but no stores outstanding, the compiler
  // must size the loop-top wait for that state — "all but the 13 youngest", which on the back edge means every load and
  // three of the stores)
  __builtin_amdgcn_s_waitcnt(0x0F70);
  for (; tile < nfull; tile += tstride) {
    f32x16 acc;
    product(acc);
    gather(min(tile + tstride, nfull - 1u));
    float *yt = y + ((size_t)tile * 32 + 4 * lhi) * COUT + l31;
#pragma unroll
    for (int r = 0; r < 16; r++) {
      __builtin_nontemporal_store(acc[r], yt + (size_t)((r & 3) + 8 * (r >> 2)) * COUT);
      s1 += acc[r];
      s2 += acc[r] * acc[r];
    }
  }
  if (tile == nfull && nfull * 32u < NP) {  // the ragged last tile (one wave of the grid)
    f32x16 acc;
    gather(tile);
    product(acc);
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const unsigned row = tile * 32u + (r & 3) + 8 * (r >> 2) + 4 * lhi;
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
  constexpr int NJ = COUT / 32, U = 8;  // (round 5: 4 -> 8 pixel pairs in flight per wave — the loop only loads, it is latency-bound)
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

  // (pixel arithmetic in 32 bits: the host checks N*Ho*Wo < 2^31)
  const unsigned NP = (unsigned)((long)G.N * G.Ho * G.Wo), HoWo = (unsigned)G.Ho * (unsigned)G.Wo;
  const unsigned per = ((NP + gridDim.x - 1) / gridDim.x + 7u) & ~7u;  // pixels per workgroup, multiple of 8
  const unsigned pbeg = min(NP, blockIdx.x * per), pend = min(NP, pbeg + per);
  // wave w takes pixel pairs w, w+4, w+8, ... of the range; U pairs are in flight per iteration
  for (unsigned q = pbeg + 2 * wave; q < pend; q += 8 * U) {
    float av[U], gv[U][NJ], yv[U][NJ], lv[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const unsigned p = q + 8 * u + lhi;
      const unsigned pc = min(p, NP - 1u);
      const unsigned n = pc / HoWo, rem = pc - n * HoWo;
      const int oy = (int)(rem / (unsigned)G.Wo), ox = (int)(rem - (unsigned)oy * (unsigned)G.Wo);
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
// Dense 3x3 conv with many channels on the matrix pipe, im2col-free (xception entry_flow_conv1_2: 32 -> 64 at
// 256x256, 1.2 GMAC / image, deeplabv3p.py:289).  The reduction axis (tap, channel) is walked tap by tap: for tap t
// the A operand of v_mfma_f32_32x32x2_f32 is GATHERED from the tensor itself — lane (pixel = lane & 31, half = lane >> 5)
// loads 16 consecutive channels of its tap-shifted pixel (four 16-byte loads; the 9x overlap between taps is served by
// L1/L2, no column matrix exists) — and the 32 x CO weight slice of (tap, 32-channel chunk) is staged through
// double-buffered LDS as the B operand.  One kernel serves
//   forward   y[p][co]  = sum_{t,ci} T(x)[p (+) t][ci] * W[t][ci][co]        (epilogue: BN batch-stat partials)
//   bwd-data  dx[p][ci] = sum_{t,co} dY[p (-) t][co]   * W[t][ci][co]        (prologue: BN-backward affine of two
//             tensors on load; epilogue: activation mask, residual gradient, BN-backward stat partials)
// A wave owns TP = 2 tiles of 32 pixels (one weight fragment read feeds both); a workgroup 256 pixels per step.
// ---------------------------------------------------------------------------------------
struct TapArgs {
  const float *a, *a2;                 // gathered tensor(s) [N, Ha, Wa, CA]
  const float *ka, *kb, *kc;           // element transform act(ka*a + kb*a2 + kc) per channel (ka == nullptr: identity)
  int a_act;
  const float *w;                      // B[t][k][j] = w[t * w_ld_t + k * w_ld_k + j]
  int w_ld_t, w_ld_k;
  float *c;                            // output [N, Hc, Wc, CO]
  const float *ep_x, *ep_scale, *ep_shift;  // bwd-data epilogue: forward input (mask, x_hat)
  int ep_act;
  const float *ep_add, *ep_mean, *ep_invstd;
  float *part;                         // [gridDim.x][CO][2] (nullable)
  int N, Ha, Wa, Hc, Wc, stride, pad_t, pad_l;
};

template <int CA, int CO, bool BWD>
__global__ __launch_bounds__(256, 2) void conv3x3_tap_mfma_kernel(TapArgs P) {
  constexpr int NJ = CO / 32, NCH = CA / 32, TP = 2, NST = 9 * NCH, NW = (32 * CO / 4) / 256;
  static_assert(CA % 32 == 0 && CO % 32 == 0 && NW >= 1 && NJ <= 4, "32-channel chunks");
  __shared__ float Ws[2][32 * CO];
  __shared__ float cf[3][CA];
  __shared__ float red[4][2 * CO];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
  const bool two = BWD && P.a2 != nullptr;
  for (int i = tid; i < CA; i += 256) {
    cf[0][i] = P.ka ? P.ka[i] : 1.f;
    cf[1][i] = two ? P.kb[i] : 0.f;
    cf[2][i] = P.ka ? P.kc[i] : 0.f;
  }
  float st1[NJ], st2[NJ];
#pragma unroll
  for (int j = 0; j < NJ; j++) st1[j] = st2[j] = 0.f;
  const long NP = (long)P.N * P.Hc * P.Wc;

  f32x4 rw[NW];
  auto load_w = [&](int q) {  // stage q = (tap, chunk): 32 k-rows x CO columns
    const int t = q / NCH, ch = q % NCH;
#pragma unroll
    for (int i = 0; i < NW; i++) {
      const int idx = tid + 256 * i, kk = idx / (CO / 4), c4 = (idx % (CO / 4)) * 4;
      rw[i] = ld4(P.w + (size_t)t * P.w_ld_t + (size_t)(ch * 32 + kk) * P.w_ld_k + c4);
    }
  };
  auto store_w = [&](float *dst) {
#pragma unroll
    for (int i = 0; i < NW; i++) {
      const int idx = tid + 256 * i;
      st4(dst + (idx / (CO / 4)) * CO + (idx % (CO / 4)) * 4, rw[i]);
    }
  };

  for (long base = (long)blockIdx.x * (128 * TP); base < NP; base += (long)gridDim.x * (128 * TP)) {
    int py[TP], px[TP], pn[TP];
#pragma unroll
    for (int tp = 0; tp < TP; tp++) {
      const long p = min(base + (wave * TP + tp) * 32 + l31, NP - 1);
      px[tp] = (int)(p % P.Wc);
      py[tp] = (int)((p / P.Wc) % P.Hc);
      pn[tp] = (int)(p / ((long)P.Wc * P.Hc));
    }
    f32x16 acc[TP][NJ];
#pragma unroll
    for (int tp = 0; tp < TP; tp++)
#pragma unroll
      for (int j = 0; j < NJ; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[tp][j][r] = 0.f;

    load_w(0);
    __syncthreads();  // previous step's readers are done with both buffers (and cf is visible)
    store_w(Ws[0]);
    __syncthreads();
    for (int q = 0; q < NST; ++q) {
      const int t = q / NCH, ch = q % NCH, ti = t / 3, tj = t % 3;
      if (q + 1 < NST) load_w(q + 1);
      // gather: 16 consecutive channels (ch*32 + lhi*16 ..) of the tap-shifted pixel, all loads issued first
      f32x4 av[TP][4], bv[TP][4];
      float live[TP];
#pragma unroll
      for (int tp = 0; tp < TP; tp++) {
        int gy, gx;
        bool ok;
        if (!BWD) {
          gy = py[tp] * P.stride - P.pad_t + ti;
          gx = px[tp] * P.stride - P.pad_l + tj;
          ok = gy >= 0 && gy < P.Ha && gx >= 0 && gx < P.Wa;
        } else {  // transposed: which output pixel used this input pixel under tap (ti, tj)?
          const int ty = py[tp] + P.pad_t - ti, tx = px[tp] + P.pad_l - tj;
          gy = ty / P.stride;
          gx = tx / P.stride;
          ok = ty >= 0 && tx >= 0 && ty % P.stride == 0 && tx % P.stride == 0 && gy < P.Ha && gx < P.Wa;
        }
        const int gyc = min(max(gy, 0), P.Ha - 1), gxc = min(max(gx, 0), P.Wa - 1);
        live[tp] = ok ? 1.f : 0.f;
        const size_t off = (((size_t)pn[tp] * P.Ha + gyc) * P.Wa + gxc) * CA + ch * 32 + lhi * 16;
#pragma unroll
        for (int v = 0; v < 4; v++) {
          av[tp][v] = ld4(P.a + off + 4 * v);
          if (BWD) bv[tp][v] = two ? ld4(P.a2 + off + 4 * v) : splat4(0.f);
        }
      }
      float aop[TP][16];
#pragma unroll
      for (int v = 0; v < 4; v++) {
        const int cbase = ch * 32 + lhi * 16 + 4 * v;
        const f32x4 fa = ld4(&cf[0][cbase]), fb = ld4(&cf[1][cbase]), fc = ld4(&cf[2][cbase]);
#pragma unroll
        for (int tp = 0; tp < TP; tp++) {
          f32x4 x4 = fa * av[tp][v] + fc;
          if (BWD) x4 += fb * bv[tp][v];
          x4 = dl3_act4(x4, P.a_act) * splat4(live[tp]);
          aop[tp][4 * v + 0] = x4.x; aop[tp][4 * v + 1] = x4.y; aop[tp][4 * v + 2] = x4.z; aop[tp][4 * v + 3] = x4.w;
        }
      }
      const float *B = Ws[q & 1];
#pragma unroll
      for (int s_ = 0; s_ < 16; s_++) {
        float bf[NJ];
#pragma unroll
        for (int j = 0; j < NJ; j++) bf[j] = B[(lhi * 16 + s_) * CO + j * 32 + l31];
#pragma unroll
        for (int tp = 0; tp < TP; tp++)
#pragma unroll
          for (int j = 0; j < NJ; j++)
            acc[tp][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(aop[tp][s_], bf[j], acc[tp][j], 0, 0, 0);
      }
      if (q + 1 < NST) store_w(Ws[(q + 1) & 1]);
      __syncthreads();
    }

    // epilogue: C/D reg r of lane l is pixel row (r&3) + 8*(r>>2) + 4*(l>>5) of the tile, channel l & 31
#pragma unroll
    for (int tp = 0; tp < TP; tp++) {
      const long tbase = base + (wave * TP + tp) * 32;
#pragma unroll
      for (int j = 0; j < NJ; j++) {
        const int col = j * 32 + l31;
        float es = 1.f, et = 0.f, mu = 0.f, is = 0.f;
        if (BWD && P.ep_scale) { es = P.ep_scale[col]; et = P.ep_shift[col]; }
        if (BWD && P.ep_mean) { mu = P.ep_mean[col]; is = P.ep_invstd[col]; }
        float xr[16], ad[16];
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const long p = min(tbase + (r & 3) + 8 * (r >> 2) + 4 * lhi, NP - 1);
          xr[r] = (BWD && P.ep_x) ? P.ep_x[(size_t)p * CO + col] : 0.f;
          ad[r] = (BWD && P.ep_add) ? P.ep_add[(size_t)p * CO + col] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const long p = tbase + (r & 3) + 8 * (r >> 2) + 4 * lhi;
          float v = acc[tp][j][r];
          if (BWD && P.ep_x) v *= dl3_act_mask(es * xr[r] + et, P.ep_act);
          v += ad[r];
          if (p < NP) {
            __builtin_nontemporal_store(v, &P.c[(size_t)p * CO + col]);
            st1[j] += v;
            st2[j] += BWD ? v * ((xr[r] - mu) * is) : v * v;
          }
        }
      }
    }
  }
  if (P.part) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NJ; j++) {
      const float a1 = st1[j] + __shfl_xor(st1[j], 32, 64), a2 = st2[j] + __shfl_xor(st2[j], 32, 64);
      if (lhi == 0) { red[wave][2 * (j * 32 + l31)] = a1; red[wave][2 * (j * 32 + l31) + 1] = a2; }
    }
    __syncthreads();
    for (int i = tid; i < 2 * CO; i += 256)
      P.part[(size_t)blockIdx.x * CO * 2 + i] = ((red[0][i] + red[1][i]) + red[2][i]) + red[3][i];
  }
}

// Weight gradient of the same convolution, im2col-free: dW[t][ci][co] = sum_p T(x)[p (+) t][ci] * dY[p][co] with the
// pixel index as the reduction axis (two pixels per MFMA, as in conv3x3_stem_wgrad_kernel) and (ci, co) as the 32x32
// tile axes.  A workgroup is THREE waves, one per kernel row i: wave i accumulates the three taps (i, 0..2) for every
// (32-ci block, 32-co block) — 3*NCI*NCO accumulator tiles — so all nine taps of a pixel pair are covered by one pass
// over dY (the three waves read the same dY rows, L1 serves two of them).  Operands come straight from global memory
// with 128-byte coalesced dword loads (one channel per lane); no LDS in the main loop.
struct TapWgArgs {
  const float *x, *xs, *xt; int x_act;
  const float *g, *y, *cA, *cB, *cC;
  float *ws;  // [gridDim.x][9*CIN][COUT]
  int N, H, W, Ho, Wo, stride, pad_t, pad_l;
};

template <int CIN, int COUT>
__global__ __launch_bounds__(192, 2) void conv3x3_tap_wgrad_kernel(TapWgArgs P) {
  constexpr int NCI = CIN / 32, NCO = COUT / 32, U = (NCI * NCO >= 4) ? 2 : 4;  // pixel pairs in flight per iteration
  static_assert(NCI * NCO <= 4, "register budget: 3*NCI*NCO accumulator tiles per wave");
  const int lane = threadIdx.x & 63, ti = threadIdx.x >> 6, l31 = lane & 31, lhi = lane >> 5;
  const bool two = P.cA != nullptr;
  float xs[NCI], xt[NCI], kA[NCO], kB[NCO], kC[NCO];
#pragma unroll
  for (int a = 0; a < NCI; a++) { xs[a] = P.xs ? P.xs[a * 32 + l31] : 1.f; xt[a] = P.xs ? P.xt[a * 32 + l31] : 0.f; }
#pragma unroll
  for (int b = 0; b < NCO; b++) {
    kA[b] = two ? P.cA[b * 32 + l31] : 1.f;
    kB[b] = two ? P.cB[b * 32 + l31] : 0.f;
    kC[b] = two ? P.cC[b * 32 + l31] : 0.f;
  }
  const float *yr = two ? P.y : P.g;
  f32x16 acc[3][NCI][NCO];
#pragma unroll
  for (int j = 0; j < 3; j++)
#pragma unroll
    for (int a = 0; a < NCI; a++)
#pragma unroll
      for (int b = 0; b < NCO; b++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[j][a][b][r] = 0.f;

  const long NP = (long)P.N * P.Ho * P.Wo;
  const long per = ((NP + gridDim.x - 1) / gridDim.x + 2 * U - 1) / (2 * U) * (2 * U);
  const long pbeg = (long)blockIdx.x * per, pend = min(NP, pbeg + per);
  for (long q = pbeg; q < pend; q += 2 * U) {
    float av[U][3][NCI], gv[U][NCO], yv[U][NCO], lv[U][3];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const long p = q + 2 * u + lhi;
      const long pc = min(p, NP - 1);
      const int ox = (int)(pc % P.Wo), oy = (int)((pc / P.Wo) % P.Ho), n = (int)(pc / ((long)P.Wo * P.Ho));
      const int iy = oy * P.stride - P.pad_t + ti, iyc = min(max(iy, 0), P.H - 1);
#pragma unroll
      for (int j = 0; j < 3; j++) {
        const int ix = ox * P.stride - P.pad_l + j, ixc = min(max(ix, 0), P.W - 1);
        lv[u][j] = (p < pend && iy == iyc && ix == ixc) ? 1.f : 0.f;
        const float *xp = P.x + (((size_t)n * P.H + iyc) * P.W + ixc) * CIN + l31;
#pragma unroll
        for (int a = 0; a < NCI; a++) av[u][j][a] = xp[a * 32];
      }
#pragma unroll
      for (int b = 0; b < NCO; b++) {
        gv[u][b] = P.g[(size_t)pc * COUT + b * 32 + l31];
        yv[u][b] = yr[(size_t)pc * COUT + b * 32 + l31];
      }
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      float bo[NCO];
#pragma unroll
      for (int b = 0; b < NCO; b++) bo[b] = kA[b] * gv[u][b] + kB[b] * yv[u][b] + kC[b];
#pragma unroll
      for (int j = 0; j < 3; j++)
#pragma unroll
        for (int a = 0; a < NCI; a++) {
          const float ao = dl3_act(xs[a] * av[u][j][a] + xt[a], P.x_act) * lv[u][j];
#pragma unroll
          for (int b = 0; b < NCO; b++)
            acc[j][a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(ao, bo[b], acc[j][a][b], 0, 0, 0);
        }
    }
  }
  float *out = P.ws + (size_t)blockIdx.x * 9 * CIN * COUT;
#pragma unroll
  for (int j = 0; j < 3; j++)
#pragma unroll
    for (int a = 0; a < NCI; a++)
#pragma unroll
      for (int b = 0; b < NCO; b++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int ci = a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
          out[((size_t)(ti * 3 + j) * CIN + ci) * COUT + b * 32 + l31] = acc[j][a][b][r];
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
  // (the matrix-pipe stem kernel: 32-bit element indices, a window that leaves the image by at most one row / column)
  if (9 * Cin <= 28 && Cout == 32 && (long)N * Ho * Wo < (1L << 31) && (long)N * H * W * Cin < (1L << 30) && pad_t <= 1 &&
      pad_l <= 1 && H >= 3 && W >= 3 && (Ho - 1) * stride - pad_t + 2 <= H && (Wo - 1) * stride - pad_l + 2 <= W)
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
  if (9 * Cin <= 32 && Cout == 32 && (long)N * Ho * Wo < (1L << 31) - 64 * 2048) {
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

// ---- matrix-pipe route for many-channel 3x3 convs (tap-gather kernels above; no column matrix) -------------
namespace {
bool tap_pair_ok(int ca, int co) { return (ca == 32 || ca == 64) && (co == 32 || co == 64); }
int tap_blocks(long NP) {
  long b = (NP + 255) / 256;
  return (int)(b > 1024 ? 1024 : (b < 1 ? 1 : b));
}
int tap_wgrad_blocks(long NP) {
  long b = NP / 1024;
  return (int)(b > 512 ? 512 : (b < 1 ? 1 : b));
}
int tap_check(const char *name, const CGeom &G) {
  DL3_CHECK_ARG(G.N > 0 && G.H > 0 && G.W > 0 && G.Cin > 0 && G.Cout > 0 && G.Ho > 0 && G.Wo > 0 && G.stride >= 1,
                "%s: bad dimension", name);
  DL3_UNSUPPORTED(!tap_pair_ok(G.Cin, G.Cout), "%s: Cin, Cout must each be 32 or 64 (got %d, %d)", name, G.Cin, G.Cout);
  return DL3_OK;
}
template <bool BWD>
void launch_tap(int ca, int co, const TapArgs &A, int blocks, hipStream_t st) {
  dim3 g(blocks), b(256);
  if (ca == 32 && co == 32) hipLaunchKernelGGL((conv3x3_tap_mfma_kernel<32, 32, BWD>), g, b, 0, st, A);
  else if (ca == 32 && co == 64) hipLaunchKernelGGL((conv3x3_tap_mfma_kernel<32, 64, BWD>), g, b, 0, st, A);
  else if (ca == 64 && co == 32) hipLaunchKernelGGL((conv3x3_tap_mfma_kernel<64, 32, BWD>), g, b, 0, st, A);
  else hipLaunchKernelGGL((conv3x3_tap_mfma_kernel<64, 64, BWD>), g, b, 0, st, A);
}
}  // namespace

extern "C" int dl3_conv3x3_mfma_supported(int Cin, int Cout) { return tap_pair_ok(Cin, Cout) ? 1 : 0; }

extern "C" int dl3_conv3x3_mfma_partials(int N, int Hc, int Wc) {
  if (N <= 0 || Hc <= 0 || Wc <= 0) return 0;
  return tap_blocks((long)N * Hc * Wc);
}

extern "C" int dl3_conv3x3_mfma_fwd(const float *x, const float *in_scale, const float *in_shift, int in_act,
                                    const float *w, float *y, int N, int H, int W, int Cin, int Cout, int stride,
                                    int pad_t, int pad_l, int Ho, int Wo, float *stat_partial, void *stream) {
  CGeom G{N, H, W, Cin, Cout, stride, pad_t, pad_l, Ho, Wo};
  int rc = tap_check("conv3x3_mfma_fwd", G);
  if (rc) return rc;
  DL3_CHECK_ARG(x && w && y, "conv3x3_mfma_fwd: null pointer");
  DL3_CHECK_ARG((in_scale == nullptr) == (in_shift == nullptr), "conv3x3_mfma_fwd: scale/shift must come together");
  TapArgs A{};
  A.a = x; A.a2 = nullptr; A.ka = in_scale; A.kb = nullptr; A.kc = in_shift; A.a_act = in_act;
  A.w = w; A.w_ld_t = Cin * Cout; A.w_ld_k = Cout;
  A.c = y; A.part = stat_partial;
  A.N = N; A.Ha = H; A.Wa = W; A.Hc = Ho; A.Wc = Wo; A.stride = stride; A.pad_t = pad_t; A.pad_l = pad_l;
  launch_tap<false>(Cin, Cout, A, tap_blocks((long)N * Ho * Wo), (hipStream_t)stream);
  DL3_LAUNCH_CHECK("conv3x3_mfma_fwd");
  return DL3_OK;
}

extern "C" int dl3_conv3x3_mfma_bwd_data(const float *g, const float *yraw, const float *cA, const float *cB,
                                         const float *cC, const float *wT, float *dx, const float *x,
                                         const float *in_scale, const float *in_shift, int in_act,
                                         const float *dx_add, const float *x_mean, const float *x_invstd,
                                         float *dstat_partial, int N, int H, int W, int Cin, int Cout, int stride,
                                         int pad_t, int pad_l, int Ho, int Wo, void *stream) {
  CGeom G{N, H, W, Cin, Cout, stride, pad_t, pad_l, Ho, Wo};
  int rc = tap_check("conv3x3_mfma_bwd_data", G);
  if (rc) return rc;
  DL3_CHECK_ARG(g && wT && dx, "conv3x3_mfma_bwd_data: null pointer");
  DL3_CHECK_ARG(!cA || (yraw && cB && cC), "conv3x3_mfma_bwd_data: cA needs yraw, cB, cC");
  DL3_CHECK_ARG(in_act == DL3_ACT_NONE || x, "conv3x3_mfma_bwd_data: activation mask needs x");
  DL3_CHECK_ARG(!dstat_partial || (x && x_mean && x_invstd), "conv3x3_mfma_bwd_data: dstat needs x, x_mean, x_invstd");
  DL3_CHECK_ARG((in_scale == nullptr) == (in_shift == nullptr), "conv3x3_mfma_bwd_data: scale/shift must come together");
  TapArgs A{};
  A.a = g; A.a2 = cA ? yraw : nullptr; A.ka = cA; A.kb = cB; A.kc = cC; A.a_act = DL3_ACT_NONE;
  A.w = wT; A.w_ld_t = Cin; A.w_ld_k = 9 * Cin;  // wT [Cout][9*Cin]: B[t][k = co][j = ci]
  A.c = dx;
  const bool need_x = in_act != DL3_ACT_NONE || dstat_partial;
  A.ep_x = need_x ? x : nullptr; A.ep_scale = in_scale; A.ep_shift = in_shift; A.ep_act = in_act;
  A.ep_add = dx_add; A.ep_mean = dstat_partial ? x_mean : nullptr; A.ep_invstd = dstat_partial ? x_invstd : nullptr;
  A.part = dstat_partial;
  A.N = N; A.Ha = Ho; A.Wa = Wo; A.Hc = H; A.Wc = W; A.stride = stride; A.pad_t = pad_t; A.pad_l = pad_l;
  launch_tap<true>(Cout, Cin, A, tap_blocks((long)N * H * W), (hipStream_t)stream);
  DL3_LAUNCH_CHECK("conv3x3_mfma_bwd_data");
  return DL3_OK;
}

extern "C" int dl3_reduce_partials(const float *partial, int P, int n, float *out, void *stream);

extern "C" size_t dl3_conv3x3_mfma_bwd_weight_workspace(int N, int H, int W, int Cin, int Cout, int stride, int Ho,
                                                        int Wo) {
  (void)H; (void)W; (void)stride;
  if (N <= 0 || Ho <= 0 || Wo <= 0 || Cin <= 0 || Cout <= 0) return 0;
  return (size_t)tap_wgrad_blocks((long)N * Ho * Wo) * 9 * Cin * Cout * sizeof(float);
}

extern "C" int dl3_conv3x3_mfma_bwd_weight(const float *x, const float *in_scale, const float *in_shift, int in_act,
                                           const float *g, const float *yraw, const float *cA, const float *cB,
                                           const float *cC, float *dw, int N, int H, int W, int Cin, int Cout,
                                           int stride, int pad_t, int pad_l, int Ho, int Wo, void *workspace,
                                           size_t workspace_bytes, void *stream) {
  CGeom G{N, H, W, Cin, Cout, stride, pad_t, pad_l, Ho, Wo};
  int rc = tap_check("conv3x3_mfma_bwd_weight", G);
  if (rc) return rc;
  DL3_CHECK_ARG(x && g && dw && workspace, "conv3x3_mfma_bwd_weight: null pointer");
  DL3_CHECK_ARG(!cA || (yraw && cB && cC), "conv3x3_mfma_bwd_weight: cA needs yraw, cB, cC");
  DL3_CHECK_ARG((in_scale == nullptr) == (in_shift == nullptr), "conv3x3_mfma_bwd_weight: scale/shift must come together");
  const size_t need = dl3_conv3x3_mfma_bwd_weight_workspace(N, H, W, Cin, Cout, stride, Ho, Wo);
  if (workspace_bytes < need) {
    dl3_set_error("conv3x3_mfma_bwd_weight: workspace %zu < %zu bytes", workspace_bytes, need);
    return DL3_EWORKSPACE;
  }
  TapWgArgs A{};
  A.x = x; A.xs = in_scale; A.xt = in_shift; A.x_act = in_act;
  A.g = g; A.y = yraw; A.cA = cA; A.cB = cB; A.cC = cC;
  A.ws = (float *)workspace;
  A.N = N; A.H = H; A.W = W; A.Ho = Ho; A.Wo = Wo; A.stride = stride; A.pad_t = pad_t; A.pad_l = pad_l;
  const int blocks = tap_wgrad_blocks((long)N * Ho * Wo);
  dim3 gr(blocks), bl(192);
  hipStream_t st = (hipStream_t)stream;
  if (Cin == 32 && Cout == 32) hipLaunchKernelGGL((conv3x3_tap_wgrad_kernel<32, 32>), gr, bl, 0, st, A);
  else if (Cin == 32 && Cout == 64) hipLaunchKernelGGL((conv3x3_tap_wgrad_kernel<32, 64>), gr, bl, 0, st, A);
  else if (Cin == 64 && Cout == 32) hipLaunchKernelGGL((conv3x3_tap_wgrad_kernel<64, 32>), gr, bl, 0, st, A);
  else hipLaunchKernelGGL((conv3x3_tap_wgrad_kernel<64, 64>), gr, bl, 0, st, A);
  DL3_LAUNCH_CHECK("conv3x3_mfma_bwd_weight");
  return dl3_reduce_partials(A.ws, blocks, 9 * Cin * Cout, dw, stream);
}
