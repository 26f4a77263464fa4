// dwconv.hip — DepthwiseConv2D 3x3 (deeplabv3p.py:73-74, :186-188): forward and fused backward.
//
// HBM-bound (9 MAC per 8 B).  Layout NHWC fp32, 4 channels (16 B) per lane; a workgroup is
// 8 channel-quads x 32 pixels, so one wave-instruction touches 8 pixels x 128 B (one full
// cache line per pixel of a 32-channel slab).
//
//  * march kernels (stride 1, SAME): im2col-free direct dilated convolution.  A dilated conv
//    with rate r couples only rows y == a (mod r): each workgroup walks ONE such row phase
//    (rows a, a+r, a+2r, ...) top to bottom, so the vertical taps are the lane's own history
//    held in registers — every input row is fetched once from HBM regardless of the rate
//    (the "space-to-batch" trick of TF done virtually, no data movement).  The two horizontal
//    taps x-r / x+r are served by L1/L2 (same row, same time, neighbouring lanes/waves).
//  * gather kernels: generic 9-tap form for stride 2 / explicit pads (2 layers in MobileNetV2).
//
// BatchNorm + ReLU6 of the producer are applied on load ("input transform"); the BN batch
// statistics of THIS layer's output are reduced in the epilogue into deterministic partials.
#include "common.h"

// Contraction by the language rule only (a * b + c inside ONE expression becomes a fused multiply-add; nothing is fused
// across statements): the march kernels below carry two bodies of the same row loop (a predicated one and a straight-line
// one), and under the default -ffp-contract=fast the back end is free to fuse them differently — the two bodies of one
// kernel then rounded the same pixel differently depending on which one a batch size's row chunking sent it through
// (round 5: the B=16 and B=2 engines of tests/test_gpu_fullsize.py::test_benchmarked_plan_cfg4_b16 parted by 8e-5).
#pragma clang fp contract(on)
#include <type_traits>
#include <stdlib.h>

namespace {

// wave-uniform base pointer + 32-bit BYTE offset per lane: the form the scalar-base global_load / global_store encodes
// (a float index would be shifted left by two first and no longer be the zero-extension of a 32-bit register)
__device__ __forceinline__ f32x4 ld4_su(const float *base, unsigned byte_off) {
  return ld4(reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + byte_off));
}
__device__ __forceinline__ void st4_nt_su(float *base, unsigned byte_off, f32x4 v) {
  st4_nt(reinterpret_cast<float *>(reinterpret_cast<char *>(base) + byte_off), v);
}

struct DwGeom {
  int N, H, W, C, stride, rate, pad_t, pad_l, Ho, Wo;
  int prows;  // rows of the caller's partial buffers (dl3_dwconv3x3_partials); the launch writes gridDim.y of them
};

// ---- block reduction over the 32 pixel lanes that share a channel quad -----------------
// lane layout: tid = pl*8 + cq.  Result valid in threads tid < 8 (pl == 0), fixed order.
template <int NV>
__device__ __forceinline__ void reduce_px(float (&v)[NV], float *lds /* [4][8][NV] */) {
  const int tid = threadIdx.x;
  const int wave = tid >> 6, cq = tid & 7;
#pragma unroll
  for (int i = 0; i < NV; i++) {
    float s = v[i];
    s += __shfl_xor(s, 8, 64);
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    v[i] = s;
  }
  __syncthreads();
  if ((tid & 63) < 8) {
#pragma unroll
    for (int i = 0; i < NV; i++) lds[(wave * 8 + cq) * NV + i] = v[i];
  }
  __syncthreads();
  if (tid < 8) {
#pragma unroll
    for (int i = 0; i < NV; i++)
      v[i] = ((lds[(0 * 8 + cq) * NV + i] + lds[(1 * 8 + cq) * NV + i]) + lds[(2 * 8 + cq) * NV + i]) +
             lds[(3 * 8 + cq) * NV + i];
  }
}

__device__ __forceinline__ void write_stat_partial(float *part, int p, int C, int c, f32x4 s1, f32x4 s2) {
  // part [P][C][2]
  float *d = part + ((size_t)p * C + c) * 2;
  d[0] = s1.x; d[1] = s2.x; d[2] = s1.y; d[3] = s2.y;
  d[4] = s1.z; d[5] = s2.z; d[6] = s1.w; d[7] = s2.w;
}

// The partial buffers are sized for the larger of the forward and backward decompositions: the owner of row p also zeroes
// rows p + written, p + 2 * written, ... < rows (instead of a memset node behind the launch).
__device__ __forceinline__ void pad_stat_partial(float *part, int p, int written, int rows, int C, int c) {
  for (int q = p + written; q < rows; q += written) write_stat_partial(part, q, C, c, splat4(0.f), splat4(0.f));
}
__device__ __forceinline__ void pad_w_partial(float *wpart, int p, int written, int rows, int C, int c) {
  for (int q = p + written; q < rows; q += written)
#pragma unroll
    for (int i = 0; i < 9; i++) st4(wpart + ((size_t)q * 9 + i) * C + c, splat4(0.f));
}

// Workgroup -> tile decode for the march kernels.  The grid is 1-D and padded to a multiple of 8: hardware hands
// consecutive workgroup ids to the 8 XCDs round-robin, so id L runs on XCD L%8.  Virtual id v = (L%8)*chunk + L/8
// makes consecutive v share an XCD (and its 4 MB L2) at about the same time; with the x segment as the fastest
// coordinate the +-rate taps that cross a 32-pixel segment boundary are served by that L2 instead of a second HBM
// fetch (at rate 36 on a 64 wide map every tap crosses).
struct DwTile { int xs, slab, pc, n; };
__device__ __forceinline__ bool dw_tile(int nxseg, int nslab, int ny, int N, int xcd, DwTile &t) {
  const int L = blockIdx.x, chunk = gridDim.x >> 3;
  int v = xcd ? (L & 7) * chunk + (L >> 3) : L;
  if (v >= nxseg * nslab * ny * N) return false;
  t.xs = v % nxseg; v /= nxseg;
  t.slab = v % nslab; v /= nslab;
  t.pc = v % ny;
  t.n = v / ny;
  // (integer division runs on the vector ALU: pin the wave-uniform results to scalar registers, so that the row pointers
  // built from them are scalar and the requests can use the scalar-base + 32-bit-offset address form)
  t.xs = __builtin_amdgcn_readfirstlane(t.xs);
  t.slab = __builtin_amdgcn_readfirstlane(t.slab);
  t.pc = __builtin_amdgcn_readfirstlane(t.pc);
  t.n = __builtin_amdgcn_readfirstlane(t.n);
  return true;
}

// ======================================================================================
// march forward: 1-D grid over (x segment, channel slab, row phase/chunk, image), block 256
// ======================================================================================
__global__ __launch_bounds__(256, 3) void dw_march_fwd(const float *__restrict__ x, const float *__restrict__ sc,
                                                    const float *__restrict__ sh, int act,
                                                    const float *__restrict__ w, float *__restrict__ y,
                                                    int H, int W, int C, int r, int nchunk, int TK, int nxseg,
                                                    int nphase, int ppb, int nslab, int ny, int N, int xcd,
                                                    float *__restrict__ part, int prows, int fast) {
  __shared__ float red[4 * 8 * 8];
  const int tid = threadIdx.x, cq = tid & 7, pl = tid >> 3;
  DwTile tile;
  if (!dw_tile(nxseg, nslab, ny, N, xcd, tile)) return;
  const int xs = tile.xs, slab = tile.slab, pc = tile.pc, n = tile.n;
  const int c = slab * 32 + cq * 4;
  const int xx = xs * 32 + pl;
  const bool active = (c < C) && (xx < W);

  f32x4 wv[9], s = splat4(1.f), t = splat4(0.f);
#pragma unroll
  for (int i = 0; i < 9; i++) wv[i] = ld4(w + (size_t)i * C + min(c, C - 4));
  if (sc) { s = ld4(sc + min(c, C - 4)); t = ld4(sh + min(c, C - 4)); }
  f32x4 s1 = splat4(0.f), s2 = splat4(0.f);
  // a workgroup owns `ppb` consecutive row phases (large rates leave only 2-3 rows per phase: several phases per
  // workgroup amortise the prologue) or, for ppb == 1, one chunk of TK rows of one phase
  for (int pi = 0; pi < ppb; ++pi) {
  const int a = __builtin_amdgcn_readfirstlane((ppb > 1) ? pc * ppb + pi : pc / nchunk);
  const int ch = __builtin_amdgcn_readfirstlane((ppb > 1) ? 0 : pc % nchunk);
  if (a >= nphase) break;
  const int Ka = __builtin_amdgcn_readfirstlane((a < H) ? (H - a + r - 1) / r : 0);
  const int k0 = ch * TK;
  const int k1 = min(k0 + TK, Ka);
  const int Kc = max(Ka - 1, 0);
  const bool xl_ok = (xx - r >= 0), xr_ok = (xx + r < W);
  // clamped coordinates: every lane always issues in-bounds loads (no branches -> the loads of a row group are
  // all in flight together); out-of-image taps and idle lanes are zeroed by a select afterwards
  const int cc = min(c, C - 4), xc = min(xx, W - 1), xlc = min(max(xx - r, 0), W - 1), xrc = min(xx + r, W - 1);
  // addresses: a wave-uniform row pointer + one 32-bit byte offset per tap (shared by both bodies of the row loop below)
  const float *xn = x + (size_t)n * H * W * C;
  float *yn = y + (size_t)n * H * W * C;
  const unsigned om = 4u * (unsigned)(xc * C + cc), ol = 4u * (unsigned)(xlc * C + cc), orr = 4u * (unsigned)(xrc * C + cc);
  const unsigned oo = 4u * (unsigned)(xx * C + c);   // own position (stores; == om in an all-active workgroup)

  auto ldrow = [&](int k, f32x4 &l, f32x4 &m, f32x4 &rr) {
    const bool rok = active && k >= 0 && k < Ka;
    const float *row = xn + (size_t)(a + min(max(k, 0), Kc) * r) * W * C;
    const f32x4 vm = dl3_act4(s * ld4_su(row, om) + t, act);
    const f32x4 vl = dl3_act4(s * ld4_su(row, ol) + t, act);
    const f32x4 vr = dl3_act4(s * ld4_su(row, orr) + t, act);
    // multiply by 0/1 instead of selecting: a select lets the compiler sink the loads into a branch again
    m = vm * splat4(rok ? 1.f : 0.f);
    l = vl * splat4((rok && xl_ok) ? 1.f : 0.f);
    rr = vr * splat4((rok && xr_ok) ? 1.f : 0.f);
  };

  f32x4 accA = splat4(0.f), accB = splat4(0.f);
  // rows are processed in groups of R: all 3*R loads of a group are issued before the first use, so every lane
  // keeps R cache-missing (centre-tap) loads in flight — the kernel is latency-bound otherwise
  constexpr int R = 4;
  // input row slots ks..ke: one halo row above / below the chunk only where the phase continues there (a slot
  // outside the phase is an all-zero row: skipped, the last output row is flushed after the loop instead)
  const bool bot_halo = k1 < Ka;
  const int ks = (k0 > 0) ? k0 - 1 : k0, ke = bot_halo ? k1 : k1 - 1;
  // Round 5: INTERIOR groups of a workgroup whose lanes are all inside the tensor take a straight-line body: every row
  // of the group is a live row whose output row belongs to the chunk, so its R stores are unconditional.  vmcnt retires
  // in order and counts stores; around a store inside a branch the compiler cannot count, assumes it away and sizes the
  // waits for the group's LAST loads as vmcnt(0) — which the hardware reads as "and every store issued so far": each group
  // ended with the wave waiting for its own first three stores to be acknowledged.  With the stores counted that wait is
  // vmcnt(3).  The first group (its first row only feeds the accumulators), the last one(s) and edge workgroups keep the
  // predicated body.  Multiplying by a validity of 1.0 is exact: same values, same sums.
  const bool allact = fast && (slab * 32 + 32 <= C) && (xs * 32 + 32 <= W);
  auto group = [&](int kg, auto fastc) __attribute__((always_inline)) {
    constexpr bool FAST = decltype(fastc)::value;
    f32x4 l[R], m[R], rr[R];
#pragma unroll
    for (int j = 0; j < R; j++) {
      if constexpr (FAST) {
        // (wave-uniform row pointer + a 32-bit lane offset: one address register per tap instead of a 64-bit pair per request)
        const float *row = xn + (size_t)(a + (kg + j) * r) * W * C;
        m[j] = dl3_act4(s * ld4_su(row, om) + t, act);
        l[j] = dl3_act4(s * ld4_su(row, ol) + t, act) * splat4(xl_ok ? 1.f : 0.f);
        rr[j] = dl3_act4(s * ld4_su(row, orr) + t, act) * splat4(xr_ok ? 1.f : 0.f);
      } else {
        ldrow((kg + j <= ke) ? kg + j : -1, l[j], m[j], rr[j]);
      }
    }
#pragma unroll
    for (int j = 0; j < R; j++) {
      const int k = kg + j;
      // (no `if (k > ke) break;` here: a path that leaves the group early leaves the group's remaining requests
      // un-waited-for, the compiler then has to wait for EVERYTHING at the head of the row loop before it may reuse their
      // registers.  Rows beyond ke are all-zero slots: predicated instead.  Round 5.)
      const bool kv = FAST || k <= ke;
      // input row k feeds out[k+1] (tap row 0), out[k] (tap row 1), out[k-1] (tap row 2)
      f32x4 h0 = wv[0] * l[j] + wv[1] * m[j] + wv[2] * rr[j];
      f32x4 h1 = wv[3] * l[j] + wv[4] * m[j] + wv[5] * rr[j];
      f32x4 h2 = wv[6] * l[j] + wv[7] * m[j] + wv[8] * rr[j];
      f32x4 out = accA + h2;
      if constexpr (FAST) {
        st4_nt_su(yn + (size_t)(a + (k - 1) * r) * W * C, oo, out);
        s1 += out;
        s2 += out * out;
        accA = accB + h1;
        accB = h0;
      } else {
        if (active && kv && k - 1 >= k0 && k - 1 < k1) {
          st4_nt_su(yn + (size_t)(a + (k - 1) * r) * W * C, oo, out);
          s1 += out;
          s2 += out * out;
        }
        accA = kv ? accB + h1 : accA;
        accB = kv ? h0 : accB;
      }
    }
  };
  if (k0 < k1) {
    int kg = ks;
    group(kg, std::false_type{});
    kg += R;
    if (allact) {
      // (enter the straight-line loop with nothing outstanding: a request of the predicated group still pending at the loop
      // head is merged into every iteration's state and comes back as a vmcnt(0) in front of each group's requests)
      __builtin_amdgcn_s_waitcnt(0x0F70);
      for (; kg + R - 1 <= ke; kg += R) group(kg, std::true_type{});
    }
    for (; kg <= ke; kg += R) group(kg, std::false_type{});
  }
  if (!bot_halo && k0 < k1 && active) {  // last row of the phase: no row below contributes
    st4_nt_su(yn + (size_t)(a + (k1 - 1) * r) * W * C, oo, accA);
    s1 += accA;
    s2 += accA * accA;
  }
  }  // phases of this workgroup
  if (part) {
    float v[8] = {s1.x, s1.y, s1.z, s1.w, s2.x, s2.y, s2.z, s2.w};
    reduce_px<8>(v, red);
    if (tid < 8 && c < C) {
      const int p = (n * ny + pc) * nxseg + xs;
      f32x4 r1 = {v[0], v[1], v[2], v[3]}, r2 = {v[4], v[5], v[6], v[7]};
      write_stat_partial(part, p, C, c, r1, r2);
      pad_stat_partial(part, p, N * ny * nxseg, prows, C, c);
    }
  }
}

// ======================================================================================
// march forward, two pixels per lane (rates that divide 32: 1, 2, 4, 8, 16, 32).  A lane owns the pixel pair
// (xa, xa + r) of a 64-pixel segment (pair index j -> xa = (j / r) * 2r + j % r), so the right tap of xa and the left
// tap of xb are the lane's own centre values: four loads per row serve two outputs (xa - r, xa, xb, xb + r) instead of
// six, and a 64-wide feature map is ONE segment — no tap crosses a workgroup boundary.  Measured on the rate-4 layer:
// the side-tap loads of the one-pixel kernel cost 12 % (4.70 -> 5.26 TB/s with them removed).
// Same grid decode / row-phase walk / partial layout as dw_march_fwd with 64-pixel segments.
// ======================================================================================
__global__ __launch_bounds__(256, 3) void dw_march2_fwd(const float *__restrict__ x, const float *__restrict__ sc,
                                                     const float *__restrict__ sh, int act,
                                                     const float *__restrict__ w, float *__restrict__ y,
                                                     int H, int W, int C, int r, int nchunk, int TK, int nxseg,
                                                     int nphase, int ppb, int nslab, int ny, int N, int xcd,
                                                     float *__restrict__ part, int prows, int fast) {
  __shared__ float red[4 * 8 * 8];
  const int tid = threadIdx.x, cq = tid & 7, pl = tid >> 3;
  DwTile tile;
  if (!dw_tile(nxseg, nslab, ny, N, xcd, tile)) return;
  const int xs = tile.xs, slab = tile.slab, pc = tile.pc, n = tile.n;
  const int c = slab * 32 + cq * 4;
  const int xa = xs * 64 + (pl / r) * 2 * r + pl % r, xb = xa + r;
  const bool ca = c < C;
  const bool act_a = ca && xa < W, act_b = ca && xb < W;

  f32x4 wv[9], s = splat4(1.f), t = splat4(0.f);
#pragma unroll
  for (int i = 0; i < 9; i++) wv[i] = ld4(w + (size_t)i * C + min(c, C - 4));
  if (sc) { s = ld4(sc + min(c, C - 4)); t = ld4(sh + min(c, C - 4)); }
  f32x4 s1 = splat4(0.f), s2 = splat4(0.f);
  const int cc = min(c, C - 4);
  // clamped columns (loads are always in bounds) and 0/1 validity of the four column positions
  const int x0c = min(max(xa - r, 0), W - 1), x1c = min(xa, W - 1), x2c = min(xb, W - 1), x3c = min(xb + r, W - 1);
  const float v0 = (ca && xa - r >= 0 && xa - r < W) ? 1.f : 0.f, v1 = act_a ? 1.f : 0.f, v2 = act_b ? 1.f : 0.f,
              v3 = (ca && xb + r < W) ? 1.f : 0.f;
  // addresses: a wave-uniform row pointer + one 32-bit byte offset per column position
  const float *xn = x + (size_t)n * H * W * C;
  float *yn = y + (size_t)n * H * W * C;
  const unsigned o0 = 4u * (unsigned)(x0c * C + cc), o1 = 4u * (unsigned)(x1c * C + cc), o2 = 4u * (unsigned)(x2c * C + cc),
                 o3 = 4u * (unsigned)(x3c * C + cc);
  const unsigned osa = 4u * (unsigned)(xa * C + c), osb = 4u * (unsigned)(xb * C + c);  // own positions (stores)
  for (int pi = 0; pi < ppb; ++pi) {
    const int a = __builtin_amdgcn_readfirstlane((ppb > 1) ? pc * ppb + pi : pc / nchunk);
    const int ch = __builtin_amdgcn_readfirstlane((ppb > 1) ? 0 : pc % nchunk);
    if (a >= nphase) break;
    const int Ka = __builtin_amdgcn_readfirstlane((a < H) ? (H - a + r - 1) / r : 0);
    const int k0 = ch * TK;
    const int k1 = min(k0 + TK, Ka);
    const int Kc = max(Ka - 1, 0);
    auto ldrow = [&](int k, f32x4 (&p)[4]) {
      const float rok = (k >= 0 && k < Ka) ? 1.f : 0.f;
      const float *row = xn + (size_t)(a + min(max(k, 0), Kc) * r) * W * C;
      p[0] = dl3_act4(s * ld4_su(row, o0) + t, act) * splat4(rok * v0);
      p[1] = dl3_act4(s * ld4_su(row, o1) + t, act) * splat4(rok * v1);
      p[2] = dl3_act4(s * ld4_su(row, o2) + t, act) * splat4(rok * v2);
      p[3] = dl3_act4(s * ld4_su(row, o3) + t, act) * splat4(rok * v3);
    };
    f32x4 aA = splat4(0.f), aB = splat4(0.f), bA = splat4(0.f), bB = splat4(0.f);
    constexpr int R = 3;
    const bool bot_halo = k1 < Ka;
    const int ks = (k0 > 0) ? k0 - 1 : k0, ke = bot_halo ? k1 : k1 - 1;
    // (interior groups of all-active workgroups: straight-line body with counted stores, see dw_march_fwd)
    const bool allact = fast && (slab * 32 + 32 <= C) && (xs * 64 + 64 <= W);
    auto group = [&](int kg, auto fastc) __attribute__((always_inline)) {
      constexpr bool FAST = decltype(fastc)::value;
      f32x4 p[R][4];
#pragma unroll
      for (int j = 0; j < R; j++) {
        if constexpr (FAST) {
          const float *row = xn + (size_t)(a + (kg + j) * r) * W * C;   // wave-uniform + 32-bit lane offsets
          p[j][0] = dl3_act4(s * ld4_su(row, o0) + t, act) * splat4(v0);
          p[j][1] = dl3_act4(s * ld4_su(row, o1) + t, act) * splat4(v1);
          p[j][2] = dl3_act4(s * ld4_su(row, o2) + t, act) * splat4(v2);
          p[j][3] = dl3_act4(s * ld4_su(row, o3) + t, act) * splat4(v3);
        } else {
          ldrow((kg + j <= ke) ? kg + j : -1, p[j]);
        }
      }
#pragma unroll
      for (int j = 0; j < R; j++) {
        const int k = kg + j;
        const bool kv = FAST || k <= ke;   // (predicated, not `break`: see dw_march_fwd)
        // pixel a: taps (p0, p1, p2); pixel b: taps (p1, p2, p3)
        const f32x4 a0 = wv[0] * p[j][0] + wv[1] * p[j][1] + wv[2] * p[j][2];
        const f32x4 a1 = wv[3] * p[j][0] + wv[4] * p[j][1] + wv[5] * p[j][2];
        const f32x4 a2 = wv[6] * p[j][0] + wv[7] * p[j][1] + wv[8] * p[j][2];
        const f32x4 b0 = wv[0] * p[j][1] + wv[1] * p[j][2] + wv[2] * p[j][3];
        const f32x4 b1 = wv[3] * p[j][1] + wv[4] * p[j][2] + wv[5] * p[j][3];
        const f32x4 b2 = wv[6] * p[j][1] + wv[7] * p[j][2] + wv[8] * p[j][3];
        const f32x4 oa = aA + a2, ob = bA + b2;
        if constexpr (FAST) {
          float *orow = yn + (size_t)(a + (k - 1) * r) * W * C;
          st4_nt_su(orow, osa, oa); s1 += oa; s2 += oa * oa;
          st4_nt_su(orow, osb, ob); s1 += ob; s2 += ob * ob;
          aA = aB + a1; aB = a0;
          bA = bB + b1; bB = b0;
        } else {
          if (kv && k - 1 >= k0 && k - 1 < k1) {
            float *orow = yn + (size_t)(a + (k - 1) * r) * W * C;
            if (act_a) { st4_nt_su(orow, osa, oa); s1 += oa; s2 += oa * oa; }
            if (act_b) { st4_nt_su(orow, osb, ob); s1 += ob; s2 += ob * ob; }
          }
          aA = kv ? aB + a1 : aA; aB = kv ? a0 : aB;
          bA = kv ? bB + b1 : bA; bB = kv ? b0 : bB;
        }
      }
    };
    if (k0 < k1) {
      int kg = ks;
      group(kg, std::false_type{});
      kg += R;
      if (allact) {
        __builtin_amdgcn_s_waitcnt(0x0F70);   // (see dw_march_fwd)
        for (; kg + R - 1 <= ke; kg += R) group(kg, std::true_type{});
      }
      for (; kg <= ke; kg += R) group(kg, std::false_type{});
    }
    if (!bot_halo && k0 < k1) {  // last row of the phase: no row below contributes
      float *orow = yn + (size_t)(a + (k1 - 1) * r) * W * C;
      if (act_a) { st4_nt_su(orow, osa, aA); s1 += aA; s2 += aA * aA; }
      if (act_b) { st4_nt_su(orow, osb, bA); s1 += bA; s2 += bA * bA; }
    }
  }
  if (part) {
    float v[8] = {s1.x, s1.y, s1.z, s1.w, s2.x, s2.y, s2.z, s2.w};
    reduce_px<8>(v, red);
    if (tid < 8 && c < C) {
      const int p = (n * ny + pc) * nxseg + xs;
      f32x4 r1 = {v[0], v[1], v[2], v[3]}, r2 = {v[4], v[5], v[6], v[7]};
      write_stat_partial(part, p, C, c, r1, r2);
      pad_stat_partial(part, p, N * ny * nxseg, prows, C, c);
    }
  }
}

// ======================================================================================
// march backward (fused bwd-data + bwd-weight): same decomposition as the forward
// ======================================================================================
// SXK / ADDK: the optional operands (sx: sums against another tensor; dx_add: residual addend) are known at compile time
template <bool SXK, bool ADDK>
__global__ __launch_bounds__(256, 2) void dw_march_bwd(
    const float *__restrict__ g, const float *__restrict__ yraw, const float *__restrict__ cA,
    const float *__restrict__ cB, const float *__restrict__ cC, const float *__restrict__ x,
    const float *__restrict__ sc, const float *__restrict__ sh, int act, const float *__restrict__ w,
    float *__restrict__ dx, const float *__restrict__ dx_add, const float *__restrict__ xmean,
    const float *__restrict__ xinvstd, float *__restrict__ dpart, float *__restrict__ wpart, int H, int W, int C,
    int r, int nchunk, int TK, int nxseg, int nphase, int ppb, int nslab, int ny, int N, int xcd, int prows,
    const float *__restrict__ sx, int fast) {
  // sx (nullable, round 4): the BatchNorm-backward sums of dx are taken against x_hat of THIS tensor instead of the
  // forward input — the other input of the residual Add whose output gradient this launch completes (Xception's `sum`
  // shortcuts: the gradient reaches the block's last pointwise BatchNorm unchanged, Engine.alias_stats_target)
  __shared__ float red[4 * 8 * 36];
  const int tid = threadIdx.x, cq = tid & 7, pl = tid >> 3;
  DwTile tile;
  if (!dw_tile(nxseg, nslab, ny, N, xcd, tile)) return;
  const int xs = tile.xs, slab = tile.slab, pc = tile.pc, n = tile.n;
  const int c = slab * 32 + cq * 4;
  const int xx = xs * 32 + pl;
  const bool active = (c < C) && (xx < W);

  f32x4 wv[9], s = splat4(1.f), t = splat4(0.f);
  f32x4 kA = splat4(1.f), kB = splat4(0.f), kC = splat4(0.f), mu = splat4(0.f), is = splat4(0.f);
  {
    const int c4 = min(c, C - 4);
#pragma unroll
    for (int i = 0; i < 9; i++) wv[i] = ld4(w + (size_t)i * C + c4);
    if (sc) { s = ld4(sc + c4); t = ld4(sh + c4); }
    if (cA) { kA = ld4(cA + c4); kB = ld4(cB + c4); kC = ld4(cC + c4); }
    if (dpart) { mu = ld4(xmean + c4); is = ld4(xinvstd + c4); }
  }
  const bool two = (cA != nullptr);
  const bool xl_ok = (xx - r >= 0), xr_ok = (xx + r < W);
  const int cc = min(c, C - 4), xc = min(xx, W - 1), xlc = min(max(xx - r, 0), W - 1), xrc = min(xx + r, W - 1);
  // addresses: wave-uniform row pointers + one 32-bit byte offset per tap (shared by both bodies of the row loop)
  const size_t nimg = (size_t)n * H * W * C;
  const unsigned om = 4u * (unsigned)(xc * C + cc), ol = 4u * (unsigned)(xlc * C + cc), orr = 4u * (unsigned)(xrc * C + cc);
  const unsigned oo = 4u * (unsigned)(xx * C + c);   // own position (stores; == om in an all-active workgroup)
  const bool allact = fast && (slab * 32 + 32 <= C) && (xs * 32 + 32 <= W);
  f32x4 dwv[9];
#pragma unroll
  for (int i = 0; i < 9; i++) dwv[i] = splat4(0.f);
  f32x4 s1 = splat4(0.f), s2 = splat4(0.f);
  for (int pi = 0; pi < ppb; ++pi) {  // row phases owned by this workgroup (see dw_march_fwd)
  const int a = __builtin_amdgcn_readfirstlane((ppb > 1) ? pc * ppb + pi : pc / nchunk);
  const int ch = __builtin_amdgcn_readfirstlane((ppb > 1) ? 0 : pc % nchunk);
  if (a >= nphase) break;
  const int Ka = __builtin_amdgcn_readfirstlane((a < H) ? (H - a + r - 1) / r : 0);
  const int k0 = ch * TK;
  const int k1 = min(k0 + TK, Ka);
  const int Kc = max(Ka - 1, 0);
  const float *yr = two ? yraw : g;  // when dY = g the second stream aliases the first (coefficient 0)
  // element offset of row slot k of this phase inside the image (k clamped into the phase: always in bounds)
  auto rowoff = [&](int k) -> size_t { return nimg + (size_t)(a + min(max(k, 0), Kc) * r) * W * C; };

  auto ld_dd1 = [&](size_t row, unsigned off) -> f32x4 { return kA * ld4_su(g + row, off) + kB * ld4_su(yr + row, off) + kC; };
  auto ld_dd = [&](int k, f32x4 &l, f32x4 &m, f32x4 &rr) {
    const bool rok = active && k >= 0 && k < Ka;
    const size_t row = rowoff(k);
    const f32x4 vm = ld_dd1(row, om), vl = ld_dd1(row, ol), vr = ld_dd1(row, orr);
    m = vm * splat4(rok ? 1.f : 0.f);
    l = vl * splat4((rok && xl_ok) ? 1.f : 0.f);
    rr = vr * splat4((rok && xr_ok) ? 1.f : 0.f);
  };
  // forward input row k at own column: raw value and validity
  auto ld_e = [&](int k, f32x4 &raw, bool &ok) {
    ok = active && k >= 0 && k < Ka;
    raw = ld4_su(x + rowoff(k), om);
  };
  auto eact = [&](f32x4 raw, bool ok) -> f32x4 {
    const f32x4 v = dl3_act4(s * raw + t, act);
    return v * splat4(ok ? 1.f : 0.f);
  };

  f32x4 accA = splat4(0.f), accB = splat4(0.f);

  if (k0 < k1) {
    // groups of R dY rows: the 7*R loads of a group (3 taps x {g, yraw} + the forward input row) are issued
    // together, R x 3 of them cache-missing — twice the bytes in flight of a one-row software pipeline
    constexpr int R = 2;
    f32x4 e_prev = splat4(0.f), e_cur, e_next;
    bool ok_prev = false, ok_cur, ok_next;
    // dY row slots ks..ke (halo slots outside the phase are all-zero rows: skipped, see dw_march_fwd)
    const bool bot_halo = k1 < Ka;
    const int ks = (k0 > 0) ? k0 - 1 : k0, ke = bot_halo ? k1 : k1 - 1;
    ld_e(ks, e_cur, ok_cur);
    ld_e(ks + 1, e_next, ok_next);
    // FAST (round 5, see dw_march_fwd): an interior group of an all-active workgroup — both dY rows live, both dx rows
    // inside the chunk, the forward-input rows kg-1 .. kg+R+1 inside the phase — with the optional operands known at
    // compile time (SX: sums against another tensor, ADD: residual addend): no branch around a request or a store, so
    // the stores are counted and the group's last operands are waited for with vmcnt(1), not vmcnt(0).
    auto group = [&](int kg, auto fastc) __attribute__((always_inline)) {
      constexpr bool FAST = decltype(fastc)::value, SXC = SXK, ADDC = ADDK;
      f32x4 l[R], m[R], rr[R], e_new[R], sxv[R], addv[R];
      bool ok_new[R];
#pragma unroll
      for (int j = 0; j < R; j++) {
        if constexpr (FAST) {
          const size_t row = nimg + (size_t)(a + (kg + j) * r) * W * C;
          m[j] = ld_dd1(row, om);
          l[j] = ld_dd1(row, ol) * splat4(xl_ok ? 1.f : 0.f);
          rr[j] = ld_dd1(row, orr) * splat4(xr_ok ? 1.f : 0.f);
          ld_e(kg + j + 2, e_new[j], ok_new[j]);   // (row kg+j+2 may be the slot below the phase: clamped + flagged as ever)
          const size_t orow = nimg + (size_t)(a + (kg + j - 1) * r) * W * C;
          if constexpr (SXC) sxv[j] = ld4_su(sx + orow, om);
          if constexpr (ADDC) addv[j] = ld4_su(dx_add + orow, om);
        } else {
          ld_dd((kg + j <= ke) ? kg + j : -1, l[j], m[j], rr[j]);
          ld_e(kg + j + 2, e_new[j], ok_new[j]);
          // slot j of the group finishes dx row kg + j - 1: its x_hat operand and its residual addend (own column, clamped
          // row: always in bounds) travel with the group's requests — a load inside the predicated store block below is
          // waited for with vmcnt(0), which the in-order counter turns into "everything this wave has in flight" (round 5)
          const size_t orow = rowoff(kg + j - 1);
          if constexpr (SXC) sxv[j] = ld4_su(sx + orow, om);
          if constexpr (ADDC) addv[j] = ld4_su(dx_add + orow, om);
        }
      }
#pragma unroll
      for (int j = 0; j < R; j++) {
        const int k = kg + j;
        const bool kv = FAST || k <= ke;   // (predicated, not `break`: see dw_march_fwd)
        if (FAST || (k >= k0 && k < k1)) {
          // dW[i][j] += T(x)[k+i-1][x] * dY[k][x-(j-1)r] : j=0 -> right tap, j=2 -> left tap
          f32x4 ea0 = eact(e_prev, ok_prev), ea1 = eact(e_cur, ok_cur), ea2 = eact(e_next, ok_next);
          dwv[0] += ea0 * rr[j]; dwv[1] += ea0 * m[j]; dwv[2] += ea0 * l[j];
          dwv[3] += ea1 * rr[j]; dwv[4] += ea1 * m[j]; dwv[5] += ea1 * l[j];
          dwv[6] += ea2 * rr[j]; dwv[7] += ea2 * m[j]; dwv[8] += ea2 * l[j];
        }
        // dY row k feeds dx[k-1] (tap row 0), dx[k] (tap row 1), dx[k+1] (tap row 2)
        f32x4 h0 = wv[0] * rr[j] + wv[1] * m[j] + wv[2] * l[j];
        f32x4 h1 = wv[3] * rr[j] + wv[4] * m[j] + wv[5] * l[j];
        f32x4 h2 = wv[6] * rr[j] + wv[7] * m[j] + wv[8] * l[j];
        f32x4 out = accA + h0;
        if constexpr (FAST) {
          out = out * dl3_mask4(s * e_prev + t, act);
          if constexpr (ADDC) out += addv[j];
          st4_nt_su(dx + nimg + (size_t)(a + (k - 1) * r) * W * C, oo, out);
          s1 += out;
          s2 += out * (((SXC ? sxv[j] : e_prev) - mu) * is);
          accA = accB + h1;
          accB = h2;
          e_prev = e_cur; ok_prev = ok_cur;
          e_cur = e_next; ok_cur = ok_next;
          e_next = e_new[j]; ok_next = ok_new[j];
        } else {
          if (dx && active && kv && k - 1 >= k0 && k - 1 < k1) {
            out = out * dl3_mask4(s * e_prev + t, act);
            if constexpr (ADDC) out += addv[j];
            st4_nt_su(dx + nimg + (size_t)(a + (k - 1) * r) * W * C, oo, out);
            s1 += out;
            s2 += out * (((SXC ? sxv[j] : e_prev) - mu) * is);
          }
          accA = kv ? accB + h1 : accA;
          accB = kv ? h2 : accB;
          e_prev = kv ? e_cur : e_prev; ok_prev = kv ? ok_cur : ok_prev;
          e_cur = kv ? e_next : e_cur; ok_cur = kv ? ok_next : ok_cur;
          e_next = kv ? e_new[j] : e_next; ok_next = kv ? ok_new[j] : ok_next;
        }
      }
    };
    int kg = ks;
    group(kg, std::false_type{});
    kg += R;
    if (allact && dx) {
      __builtin_amdgcn_s_waitcnt(0x0F70);   // (enter the straight-line loop with nothing outstanding: see dw_march_fwd)
      // (dY row k1 below the chunk — the bottom halo — only finishes dx row k1-1; its dW belongs to the next chunk: not here)
      for (; kg + R - 1 < k1; kg += R) group(kg, std::true_type{});
    }
    for (; kg <= ke; kg += R) group(kg, std::false_type{});
    if (!bot_halo && dx && active) {  // last dx row of the phase (e_prev now holds forward input row k1-1)
      const size_t row = nimg + (size_t)(a + (k1 - 1) * r) * W * C;
      f32x4 out = accA * dl3_mask4(s * e_prev + t, act);
      if constexpr (ADDK) out += ld4_su(dx_add + row, oo);
      st4_nt_su(dx + row, oo, out);
      s1 += out;
      s2 += out * (((SXK ? ld4_su(sx + row, oo) : e_prev) - mu) * is);
    }
  }
  }  // phases of this workgroup
  const int p = (n * ny + pc) * nxseg + xs;
  {
    float v[36];
#pragma unroll
    for (int i = 0; i < 9; i++) {
      v[i * 4 + 0] = dwv[i].x; v[i * 4 + 1] = dwv[i].y; v[i * 4 + 2] = dwv[i].z; v[i * 4 + 3] = dwv[i].w;
    }
    reduce_px<36>(v, red);
    if (tid < 8 && c < C) {
#pragma unroll
      for (int i = 0; i < 9; i++) {
        f32x4 o = {v[i * 4], v[i * 4 + 1], v[i * 4 + 2], v[i * 4 + 3]};
        st4(wpart + ((size_t)p * 9 + i) * C + c, o);
      }
      // the partial buffers hold max(forward plan, backward plan) rows and the folds read all of them: the two-pixel
      // forward halves nxseg and may then halve TK once more, so its plan can be the LARGER one (8 x 64 x 64 x 960 at
      // rate 16: 128 forward rows, 96 backward) — the rows this grid does not own are zeroed here as in the forward
      pad_w_partial(wpart, p, N * ny * nxseg, prows, C, c);
    }
  }
  if (dpart) {
    float v[8] = {s1.x, s1.y, s1.z, s1.w, s2.x, s2.y, s2.z, s2.w};
    reduce_px<8>(v, red);
    if (tid < 8 && c < C) {
      f32x4 r1 = {v[0], v[1], v[2], v[3]}, r2 = {v[4], v[5], v[6], v[7]};
      write_stat_partial(dpart, p, C, c, r1, r2);
      pad_stat_partial(dpart, p, N * ny * nxseg, prows, C, c);
    }
  }
}

// ======================================================================================
// gather forward: grid (nslab, PB), each block loops over output pixels
// ======================================================================================
__global__ __launch_bounds__(256) void dw_gather_fwd(const float *__restrict__ x, const float *__restrict__ sc,
                                                     const float *__restrict__ sh, int act,
                                                     const float *__restrict__ w, float *__restrict__ y, DwGeom G,
                                                     float *__restrict__ part) {
  __shared__ float red[4 * 8 * 8];
  const int tid = threadIdx.x, cq = tid & 7, pl = tid >> 3;
  const int c = blockIdx.x * 32 + cq * 4;
  const bool cok = c < G.C;
  f32x4 wv[9], s = splat4(1.f), t = splat4(0.f);
#pragma unroll
  for (int i = 0; i < 9; i++) wv[i] = cok ? ld4(w + (size_t)i * G.C + c) : splat4(0.f);
  if (cok && sc) { s = ld4(sc + c); t = ld4(sh + c); }
  f32x4 s1 = splat4(0.f), s2 = splat4(0.f);
  const long NP = (long)G.N * G.Ho * G.Wo;
  for (long p = (long)blockIdx.y * 32 + pl; p < NP && cok; p += (long)gridDim.y * 32) {
    const int ox = (int)(p % G.Wo);
    const int oy = (int)((p / G.Wo) % G.Ho);
    const int n = (int)(p / ((long)G.Wo * G.Ho));
    f32x4 acc = splat4(0.f);
    f32x4 tap[9];
    float okf[9];
#pragma unroll
    for (int i = 0; i < 3; i++) {
      const int iy = oy * G.stride - G.pad_t + i * G.rate;
      const int iyc = min(max(iy, 0), G.H - 1);
#pragma unroll
      for (int j = 0; j < 3; j++) {
        const int ix = ox * G.stride - G.pad_l + j * G.rate;
        const int ixc = min(max(ix, 0), G.W - 1);
        okf[i * 3 + j] = (iy == iyc && ix == ixc) ? 1.f : 0.f;
        tap[i * 3 + j] = ld4(x + (((size_t)n * G.H + iyc) * G.W + ixc) * G.C + c);  // clamped: always in bounds
      }
    }
#pragma unroll
    for (int q = 0; q < 9; q++) acc += (wv[q] * splat4(okf[q])) * dl3_act4(s * tap[q] + t, act);
    st4_nt(y + (size_t)p * G.C + c, acc);
    s1 += acc;
    s2 += acc * acc;
  }
  if (part) {
    float v[8] = {s1.x, s1.y, s1.z, s1.w, s2.x, s2.y, s2.z, s2.w};
    reduce_px<8>(v, red);
    if (tid < 8 && cok) {
      f32x4 r1 = {v[0], v[1], v[2], v[3]}, r2 = {v[4], v[5], v[6], v[7]};
      write_stat_partial(part, blockIdx.y, G.C, c, r1, r2);
      pad_stat_partial(part, blockIdx.y, gridDim.y, G.prows, G.C, c);
    }
  }
}

// ======================================================================================
// gather backward: loops over INPUT pixels; each (input pixel, tap) pair maps to one output
// pixel and feeds both dx (x w) and dW (x T(x))
// ======================================================================================
__global__ __launch_bounds__(256) void dw_gather_bwd(
    const float *__restrict__ g, const float *__restrict__ yraw, const float *__restrict__ cA,
    const float *__restrict__ cB, const float *__restrict__ cC, const float *__restrict__ x,
    const float *__restrict__ sc, const float *__restrict__ sh, int act, const float *__restrict__ w,
    float *__restrict__ dx, const float *__restrict__ dx_add, const float *__restrict__ xmean,
    const float *__restrict__ xinvstd, float *__restrict__ dpart, float *__restrict__ wpart, DwGeom G) {
  __shared__ float red[4 * 8 * 36];
  const int tid = threadIdx.x, cq = tid & 7, pl = tid >> 3;
  const int c = blockIdx.x * 32 + cq * 4;
  const bool cok = c < G.C;
  f32x4 wv[9], s = splat4(1.f), t = splat4(0.f);
  f32x4 kA = splat4(1.f), kB = splat4(0.f), kC = splat4(0.f), mu = splat4(0.f), is = splat4(0.f);
#pragma unroll
  for (int i = 0; i < 9; i++) wv[i] = cok ? ld4(w + (size_t)i * G.C + c) : splat4(0.f);
  if (cok) {
    if (sc) { s = ld4(sc + c); t = ld4(sh + c); }
    if (cA) { kA = ld4(cA + c); kB = ld4(cB + c); kC = ld4(cC + c); }
    if (dpart) { mu = ld4(xmean + c); is = ld4(xinvstd + c); }
  }
  const bool two = (cA != nullptr);
  f32x4 dwv[9];
#pragma unroll
  for (int i = 0; i < 9; i++) dwv[i] = splat4(0.f);
  f32x4 s1 = splat4(0.f), s2 = splat4(0.f);
  const long NP = (long)G.N * G.H * G.W;
  for (long p = (long)blockIdx.y * 32 + pl; p < NP && cok; p += (long)gridDim.y * 32) {
    const int ix = (int)(p % G.W);
    const int iy = (int)((p / G.W) % G.H);
    const int n = (int)(p / ((long)G.W * G.H));
    const f32x4 eraw = ld4(x + (size_t)p * G.C + c);
    const f32x4 z = s * eraw + t;
    const f32x4 ea = dl3_act4(z, act);
    f32x4 acc = splat4(0.f);
    const float *yr = two ? yraw : g;
    f32x4 tg[9], ty_[9];
    float okf[9];
#pragma unroll
    for (int i = 0; i < 3; i++) {
      const int ty = iy + G.pad_t - i * G.rate;
      const int oy = max(ty, 0) / G.stride;
      const bool yok = ty >= 0 && (ty % G.stride) == 0 && oy < G.Ho;
      const int oyc = min(oy, G.Ho - 1);
#pragma unroll
      for (int j = 0; j < 3; j++) {
        const int tx = ix + G.pad_l - j * G.rate;
        const int ox = max(tx, 0) / G.stride;
        const bool xok = tx >= 0 && (tx % G.stride) == 0 && ox < G.Wo;
        const int oxc = min(ox, G.Wo - 1);
        const size_t off = (((size_t)n * G.Ho + oyc) * G.Wo + oxc) * G.C + c;  // clamped: always in bounds
        okf[i * 3 + j] = (yok && xok) ? 1.f : 0.f;
        tg[i * 3 + j] = ld4(g + off);
        ty_[i * 3 + j] = ld4(yr + off);
      }
    }
#pragma unroll
    for (int q = 0; q < 9; q++) {
      const f32x4 dd = (kA * tg[q] + kB * ty_[q] + kC) * splat4(okf[q]);
      acc += dd * wv[q];
      dwv[q] += ea * dd;
    }
    if (dx) {
      f32x4 out = acc * dl3_mask4(z, act);
      if (dx_add) out += ld4(dx_add + (size_t)p * G.C + c);
      st4_nt(dx + (size_t)p * G.C + c, out);
      s1 += out;
      s2 += out * ((eraw - mu) * is);
    }
  }
  {
    float v[36];
#pragma unroll
    for (int i = 0; i < 9; i++) {
      v[i * 4 + 0] = dwv[i].x; v[i * 4 + 1] = dwv[i].y; v[i * 4 + 2] = dwv[i].z; v[i * 4 + 3] = dwv[i].w;
    }
    reduce_px<36>(v, red);
    if (tid < 8 && cok) {
#pragma unroll
      for (int i = 0; i < 9; i++) {
        f32x4 o = {v[i * 4], v[i * 4 + 1], v[i * 4 + 2], v[i * 4 + 3]};
        st4(wpart + ((size_t)blockIdx.y * 9 + i) * G.C + c, o);
      }
      pad_w_partial(wpart, blockIdx.y, gridDim.y, G.prows, G.C, c);
    }
  }
  if (dpart) {
    float v[8] = {s1.x, s1.y, s1.z, s1.w, s2.x, s2.y, s2.z, s2.w};
    reduce_px<8>(v, red);
    if (tid < 8 && cok) {
      f32x4 r1 = {v[0], v[1], v[2], v[3]}, r2 = {v[4], v[5], v[6], v[7]};
      write_stat_partial(dpart, blockIdx.y, G.C, c, r1, r2);
      pad_stat_partial(dpart, blockIdx.y, gridDim.y, G.prows, G.C, c);
    }
  }
}

// ======================================================================================
// stride-2 backward (rate 1, pad 0 or 1: MobileNetV2 blocks 1 and 3, Xception entry flow): thread = one 2x2 block
// of INPUT pixels x 4 channels.  With iy0 = 2a - pad_t the block rows are "e" = iy0 (taps ky = 0 from output row a
// and ky = 2 from row a-1) and "o" = iy0+1 (tap ky = 1 from row a), same for columns, so the four dY values at
// (a, b), (a-1, b), (a, b-1), (a-1, b-1) feed the four dx pixels and all nine dW taps: 8 gradient loads per 4 output
// pixels instead of the 72 the generic gather issues (of which 9 are live).
// ======================================================================================
__global__ __launch_bounds__(256) void dw_s2_bwd(
    const float *__restrict__ g, const float *__restrict__ yraw, const float *__restrict__ cA,
    const float *__restrict__ cB, const float *__restrict__ cC, const float *__restrict__ x,
    const float *__restrict__ sc, const float *__restrict__ sh, int act, const float *__restrict__ w,
    float *__restrict__ dx, const float *__restrict__ dx_add, const float *__restrict__ xmean,
    const float *__restrict__ xinvstd, float *__restrict__ dpart, float *__restrict__ wpart, DwGeom G) {
  __shared__ float red[4 * 8 * 36];
  const int tid = threadIdx.x, cq = tid & 7, pl = tid >> 3;
  const int c = blockIdx.x * 32 + cq * 4;
  const bool cok = c < G.C;
  const int cc = min(c, G.C - 4);
  f32x4 wv[9], s = splat4(1.f), t = splat4(0.f);
  f32x4 kA = splat4(1.f), kB = splat4(0.f), kC = splat4(0.f), mu = splat4(0.f), is = splat4(0.f);
#pragma unroll
  for (int i = 0; i < 9; i++) wv[i] = ld4(w + (size_t)i * G.C + cc);
  if (sc) { s = ld4(sc + cc); t = ld4(sh + cc); }
  if (cA) { kA = ld4(cA + cc); kB = ld4(cB + cc); kC = ld4(cC + cc); }
  if (dpart) { mu = ld4(xmean + cc); is = ld4(xinvstd + cc); }
  const float *yr = cA ? yraw : g;
  f32x4 dwv[9];
#pragma unroll
  for (int i = 0; i < 9; i++) dwv[i] = splat4(0.f);
  f32x4 s1 = splat4(0.f), s2 = splat4(0.f);
  const int A = (G.H + G.pad_t + 1) / 2, Bn = (G.W + G.pad_l + 1) / 2;
  const long NB = (long)G.N * A * Bn;
  for (long q = (long)blockIdx.y * 32 + pl; q < NB; q += (long)gridDim.y * 32) {
    const int b = (int)(q % Bn), a = (int)((q / Bn) % A), n = (int)(q / ((long)Bn * A));
    const int iy0 = 2 * a - G.pad_t, ix0 = 2 * b - G.pad_l;
    // dY at (a - i, b - j), i, j in {0, 1}: clamped addresses, 0/1 validity multipliers (branch-free loads)
    f32x4 dd[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int j = 0; j < 2; j++) {
        const int oy = a - i, ox = b - j;
        const int oyc = min(max(oy, 0), G.Ho - 1), oxc = min(max(ox, 0), G.Wo - 1);
        const float ok = (cok && oy == oyc && ox == oxc) ? 1.f : 0.f;
        const size_t off = (((size_t)n * G.Ho + oyc) * G.Wo + oxc) * G.C + cc;
        dd[i][j] = (kA * ld4(g + off) + kB * ld4(yr + off) + kC) * splat4(ok);
      }
    // the 2x2 input pixels
    f32x4 xr[2][2], ad[2][2];
    float pok[2][2];
    size_t poff[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int j = 0; j < 2; j++) {
        const int iy = iy0 + i, ix = ix0 + j;
        const int iyc = min(max(iy, 0), G.H - 1), ixc = min(max(ix, 0), G.W - 1);
        pok[i][j] = (cok && iy == iyc && ix == ixc) ? 1.f : 0.f;
        poff[i][j] = (((size_t)n * G.H + iyc) * G.W + ixc) * G.C + cc;
        xr[i][j] = ld4(x + poff[i][j]);
        ad[i][j] = dx_add ? ld4(dx_add + poff[i][j]) : splat4(0.f);
      }
    f32x4 acc[2][2];
    acc[0][0] = wv[0] * dd[0][0] + wv[6] * dd[1][0] + wv[2] * dd[0][1] + wv[8] * dd[1][1];
    acc[0][1] = wv[1] * dd[0][0] + wv[7] * dd[1][0];
    acc[1][0] = wv[3] * dd[0][0] + wv[5] * dd[0][1];
    acc[1][1] = wv[4] * dd[0][0];
    f32x4 ea[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int j = 0; j < 2; j++) ea[i][j] = dl3_act4(s * xr[i][j] + t, act) * splat4(pok[i][j]);
    dwv[0] += ea[0][0] * dd[0][0]; dwv[6] += ea[0][0] * dd[1][0]; dwv[2] += ea[0][0] * dd[0][1]; dwv[8] += ea[0][0] * dd[1][1];
    dwv[1] += ea[0][1] * dd[0][0]; dwv[7] += ea[0][1] * dd[1][0];
    dwv[3] += ea[1][0] * dd[0][0]; dwv[5] += ea[1][0] * dd[0][1];
    dwv[4] += ea[1][1] * dd[0][0];
    if (dx) {
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
          f32x4 out = acc[i][j] * dl3_mask4(s * xr[i][j] + t, act) + ad[i][j];
          if (pok[i][j] != 0.f) {
            st4_nt(dx + poff[i][j], out);
            s1 += out;
            s2 += out * ((xr[i][j] - mu) * is);
          }
        }
    }
  }
  {
    float v[36];
#pragma unroll
    for (int i = 0; i < 9; i++) {
      v[i * 4 + 0] = dwv[i].x; v[i * 4 + 1] = dwv[i].y; v[i * 4 + 2] = dwv[i].z; v[i * 4 + 3] = dwv[i].w;
    }
    reduce_px<36>(v, red);
    if (tid < 8 && cok) {
#pragma unroll
      for (int i = 0; i < 9; i++) {
        f32x4 o = {v[i * 4], v[i * 4 + 1], v[i * 4 + 2], v[i * 4 + 3]};
        st4(wpart + ((size_t)blockIdx.y * 9 + i) * G.C + c, o);
      }
      pad_w_partial(wpart, blockIdx.y, gridDim.y, G.prows, G.C, c);
    }
  }
  if (dpart) {
    float v[8] = {s1.x, s1.y, s1.z, s1.w, s2.x, s2.y, s2.z, s2.w};
    reduce_px<8>(v, red);
    if (tid < 8 && cok) {
      f32x4 r1 = {v[0], v[1], v[2], v[3]}, r2 = {v[4], v[5], v[6], v[7]};
      write_stat_partial(dpart, blockIdx.y, G.C, c, r1, r2);
      pad_stat_partial(dpart, blockIdx.y, gridDim.y, G.prows, G.C, c);
    }
  }
}

// ---- host-side decomposition (shared by *_partials and the launchers) -------------------
struct DwPlan {
  int impl;                      // DL3_IMPL_MARCH / DL3_IMPL_GATHER
  int nslab, nxseg, nphase, nchunk, TK;  // march
  int ppb, ny;                           // march: row phases per workgroup, grid.y
  int two;                               // march forward: two pixels per lane (64-pixel segments)
  int PB;                        // gather
  int P;
};

unsigned march_grid(const DwPlan &p, int N) {
  const long total = (long)p.nxseg * p.nslab * p.ny * N;
  return (unsigned)((total + 7) / 8 * 8);
}
int march_fast() {
  // straight-line interior row groups (round 5): bit 0 one-pixel forward, bit 1 two-pixel forward, bit 2 backward.
  // Taken where a pixel's channels are whole 128-byte lines (C % 32 == 0).  Measured at B=128 (GPU call 9, cold operands,
  // same box; ms off -> on): backward 64x64x960 r4 1.674 -> 1.577, x576 r2 1.021 -> 0.932, x384 r2 0.802 -> 0.732, x192
  // 0.329 -> 0.293, 256x256x32 0.926 -> 0.868, but 128x128x144 (576-byte pixels: every other 128-byte store straddles two
  // lines) 1.356 -> 1.435; forward 256x256x32 0.467 -> 0.422, 128x128x144 0.569 -> 0.603, the 64x64 maps within noise.
  return 7;   // (the DL3_DW_FAST bit mask of round 5's A/B is gone: all three kernels take the fast path)
}
int march_xcd() {
  return 1;   // (XCD-aware workgroup order; the plain order was the round-1 A/B)
}

bool march_ok(int H, int W, int stride, int rate, int pad_t, int pad_l, int Ho, int Wo) {
  return stride == 1 && pad_t == rate && pad_l == rate && Ho == H && Wo == W;
}

DwPlan dw_plan(int N, int H, int W, int C, int stride, int rate, int Ho, int Wo, int impl, bool bwd) {
  DwPlan p{};
  p.nslab = dl3_cdiv(C, 32);
  p.impl = impl;
  if (impl == DL3_IMPL_MARCH) {
    p.two = (!bwd && 32 % rate == 0 && W >= 32) ? 1 : 0;
    p.nxseg = dl3_cdiv(W, p.two ? 64 : 32);
    p.nphase = rate < H ? rate : H;
    const int Kmax = dl3_cdiv(H, rate);
    // rows per block: as long as possible (halo re-read = 2/TK) while keeping >= ~1024 blocks (round 3, Xception OS=8 at
    // B=16, 64x64x736 rate 2: 2048 -> row chunks of 8 with 25 % halo, in-situ 0.573 of 8 TB/s; 1024 -> chunks of 16:
    // 0.591; 512: 0.591; 3072 / 4096: 0.543; larger batches never chunk)
    int TK = Kmax;
    const long want = 1024;  // workgroups per launch to aim for
    while (TK > 8 && (long)N * p.nslab * p.nxseg * p.nphase * dl3_cdiv(Kmax, TK) < want) TK = (TK + 1) / 2;
    p.TK = TK;
    p.nchunk = dl3_cdiv(Kmax, TK);
    // large rates: a phase holds only Kmax = 2..6 rows -> give a workgroup several phases (~12 rows of work)
    int ppb = 1;
    if (p.nchunk == 1 && Kmax < 12) {
      ppb = 12 / Kmax;
      if (ppb > p.nphase) ppb = p.nphase;
      while (ppb > 1 && (long)N * p.nslab * p.nxseg * dl3_cdiv(p.nphase, ppb) < 2048) --ppb;
      if (const char *e = getenv("DL3_DW_PPB")) ppb = atoi(e);  // tuning / test override
      if (ppb > p.nphase) ppb = p.nphase;
      if (ppb < 1) ppb = 1;
    }
    p.ppb = ppb;
    p.ny = (ppb > 1) ? dl3_cdiv(p.nphase, ppb) : p.nphase * p.nchunk;
    p.P = N * p.ny * p.nxseg;
  } else {
    const long NP = bwd ? (long)N * H * W : (long)N * Ho * Wo;
    long pb = (NP + 31) / 32;
    long cap = 4096 / p.nslab;
    if (cap < 1) cap = 1;
    if (pb > cap) pb = cap;
    p.PB = (int)pb;
    p.P = p.PB;
  }
  return p;
}

int resolve_impl(int impl, int H, int W, int stride, int rate, int pad_t, int pad_l, int Ho, int Wo) {
  const bool ok = march_ok(H, W, stride, rate, pad_t, pad_l, Ho, Wo);
  if (impl == DL3_IMPL_AUTO) return ok ? DL3_IMPL_MARCH : DL3_IMPL_GATHER;
  if (impl == DL3_IMPL_MARCH && !ok) return -1;
  return impl;
}

}  // namespace

extern "C" int dl3_dwconv3x3_partials(int N, int H, int W, int C, int stride, int rate, int Ho, int Wo,
                                      int impl) {
  // pads are implied: march needs SAME stride-1 geometry, which the caller guarantees when it asks for it
  int im = impl;
  if (im == DL3_IMPL_AUTO) im = (stride == 1 && Ho == H && Wo == W) ? DL3_IMPL_MARCH : DL3_IMPL_GATHER;
  // fwd and bwd gather plans differ in pixel count; report the larger so one buffer fits both
  DwPlan a = dw_plan(N, H, W, C, stride, rate, Ho, Wo, im, false);
  DwPlan b = dw_plan(N, H, W, C, stride, rate, Ho, Wo, im, true);
  return a.P > b.P ? a.P : b.P;
}

static int dw_check(int N, int H, int W, int C, int stride, int rate, int Ho, int Wo) {
  DL3_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && Ho > 0 && Wo > 0, "dwconv3x3: non-positive dimension");
  DL3_CHECK_ARG(stride >= 1 && rate >= 1, "dwconv3x3: stride/rate must be >= 1");
  DL3_UNSUPPORTED(C % 4 != 0, "dwconv3x3: C=%d must be a multiple of 4", C);
  DL3_UNSUPPORTED(N > 65535, "dwconv3x3: N=%d too large for grid.z", N);
  return DL3_OK;
}

extern "C" int dl3_dwconv3x3_fwd(const float *x, const float *in_scale, const float *in_shift, int in_act,
                                 const float *w, float *y, int N, int H, int W, int C, int stride, int rate,
                                 int pad_t, int pad_l, int Ho, int Wo, float *stat_partial, int impl,
                                 void *stream) {
  int rc = dw_check(N, H, W, C, stride, rate, Ho, Wo);
  if (rc) return rc;
  DL3_CHECK_ARG(x && w && y, "dwconv3x3_fwd: null pointer");
  DL3_CHECK_ARG((in_scale == nullptr) == (in_shift == nullptr), "dwconv3x3_fwd: scale/shift must come together");
  const int im = resolve_impl(impl, H, W, stride, rate, pad_t, pad_l, Ho, Wo);
  DL3_UNSUPPORTED(im < 0, "dwconv3x3_fwd: march impl needs stride 1, pad == rate, Ho == H, Wo == W");
  DwPlan p = dw_plan(N, H, W, C, stride, rate, Ho, Wo, im, false);
  hipStream_t st = (hipStream_t)stream;
  // the caller sized the partial buffer with dl3_dwconv3x3_partials (the larger of the forward and the backward
  // decomposition — either can be the larger one): the kernels zero the rows nobody owns
  const int Pmax = dl3_dwconv3x3_partials(N, H, W, C, stride, rate, Ho, Wo, im);
  if (im == DL3_IMPL_MARCH) {
    dim3 grid(march_grid(p, N));
    if (p.two)
      hipLaunchKernelGGL(dw_march2_fwd, grid, dim3(256), 0, st, x, in_scale, in_shift, in_act, w, y, H, W, C, rate,
                         p.nchunk, p.TK, p.nxseg, p.nphase, p.ppb, p.nslab, p.ny, N, march_xcd(), stat_partial, Pmax, (C % 32 == 0) ? (march_fast() & 2) : 0);
    else
      hipLaunchKernelGGL(dw_march_fwd, grid, dim3(256), 0, st, x, in_scale, in_shift, in_act, w, y, H, W, C, rate,
                         p.nchunk, p.TK, p.nxseg, p.nphase, p.ppb, p.nslab, p.ny, N, march_xcd(), stat_partial, Pmax, (C % 32 == 0) ? (march_fast() & 1) : 0);
  } else {
    DwGeom G{N, H, W, C, stride, rate, pad_t, pad_l, Ho, Wo, Pmax};
    dim3 grid(p.nslab, p.PB);
    hipLaunchKernelGGL(dw_gather_fwd, grid, dim3(256), 0, st, x, in_scale, in_shift, in_act, w, y, G,
                       stat_partial);
  }
  DL3_LAUNCH_CHECK("dwconv3x3_fwd");
  return DL3_OK;
}

static int dwconv3x3_bwd_impl(const float *g, const float *yraw, const float *cA, const float *cB,
                                 const float *cC, const float *x, const float *in_scale, const float *in_shift,
                                 int in_act, const float *w, float *dx, const float *dx_add,
                                 const float *x_mean, const float *x_invstd, float *dstat_partial,
                                 float *dw_partial, int N, int H, int W, int C, int stride, int rate, int pad_t,
                                 int pad_l, int Ho, int Wo, int impl, const float *stat_x, void *stream) {
  int rc = dw_check(N, H, W, C, stride, rate, Ho, Wo);
  if (rc) return rc;
  DL3_CHECK_ARG(!stat_x || dstat_partial, "dwconv3x3_bwd_sx: stat_x without dstat_partial");
  DL3_CHECK_ARG(g && x && w && dw_partial, "dwconv3x3_bwd: null pointer");
  DL3_CHECK_ARG(!cA || (yraw && cB && cC), "dwconv3x3_bwd: cA needs yraw, cB, cC");
  DL3_CHECK_ARG((in_scale == nullptr) == (in_shift == nullptr), "dwconv3x3_bwd: scale/shift must come together");
  DL3_CHECK_ARG(!dstat_partial || (dx && x_mean && x_invstd), "dwconv3x3_bwd: dstat needs dx, x_mean, x_invstd");
  const int im = resolve_impl(impl, H, W, stride, rate, pad_t, pad_l, Ho, Wo);
  DL3_UNSUPPORTED(im < 0, "dwconv3x3_bwd: march impl needs stride 1, pad == rate, Ho == H, Wo == W");
  DL3_UNSUPPORTED(stat_x && im != DL3_IMPL_MARCH, "dwconv3x3_bwd_sx: sums against another tensor need the march kernel (stride 1, SAME)");
  DwPlan p = dw_plan(N, H, W, C, stride, rate, Ho, Wo, im, true);
  hipStream_t st = (hipStream_t)stream;
  // (partial rows beyond this grid's: zeroed by the kernels, see pad_stat_partial)
  const int Pmax = dl3_dwconv3x3_partials(N, H, W, C, stride, rate, Ho, Wo, im);
  if (im == DL3_IMPL_MARCH) {
    dim3 grid(march_grid(p, N));
#define DL3_DW_BWD(SX_, ADD_)                                                                                              \
  hipLaunchKernelGGL((dw_march_bwd<SX_, ADD_>), grid, dim3(256), 0, st, g, yraw, cA, cB, cC, x, in_scale, in_shift, in_act, \
                     w, dx, dx_add, x_mean, x_invstd, dstat_partial, dw_partial, H, W, C, rate, p.nchunk, p.TK, p.nxseg,    \
                     p.nphase, p.ppb, p.nslab, p.ny, N, march_xcd(), Pmax, stat_x, (C % 32 == 0) ? (march_fast() & 4) : 0)
    if (stat_x) { if (dx_add) DL3_DW_BWD(true, true); else DL3_DW_BWD(true, false); }
    else { if (dx_add) DL3_DW_BWD(false, true); else DL3_DW_BWD(false, false); }
#undef DL3_DW_BWD
  } else {
    DwGeom G{N, H, W, C, stride, rate, pad_t, pad_l, Ho, Wo, Pmax};
    dim3 grid(p.nslab, p.PB);
    if (stride == 2 && rate == 1 && pad_t <= 1 && pad_l <= 1)
      hipLaunchKernelGGL(dw_s2_bwd, grid, dim3(256), 0, st, g, yraw, cA, cB, cC, x, in_scale, in_shift, in_act, w, dx,
                         dx_add, x_mean, x_invstd, dstat_partial, dw_partial, G);
    else
      hipLaunchKernelGGL(dw_gather_bwd, grid, dim3(256), 0, st, g, yraw, cA, cB, cC, x, in_scale, in_shift, in_act,
                         w, dx, dx_add, x_mean, x_invstd, dstat_partial, dw_partial, G);
  }
  DL3_LAUNCH_CHECK("dwconv3x3_bwd");
  return DL3_OK;
}

extern "C" int dl3_dwconv3x3_bwd(const float *g, const float *yraw, const float *cA, const float *cB,
                                 const float *cC, const float *x, const float *in_scale, const float *in_shift,
                                 int in_act, const float *w, float *dx, const float *dx_add,
                                 const float *x_mean, const float *x_invstd, float *dstat_partial,
                                 float *dw_partial, int N, int H, int W, int C, int stride, int rate, int pad_t,
                                 int pad_l, int Ho, int Wo, int impl, void *stream) {
  return dwconv3x3_bwd_impl(g, yraw, cA, cB, cC, x, in_scale, in_shift, in_act, w, dx, dx_add, x_mean, x_invstd,
                            dstat_partial, dw_partial, N, H, W, C, stride, rate, pad_t, pad_l, Ho, Wo, impl, nullptr, stream);
}

extern "C" int dl3_dwconv3x3_bwd_sx(const float *g, const float *yraw, const float *cA, const float *cB,
                                    const float *cC, const float *x, const float *in_scale, const float *in_shift,
                                    int in_act, const float *w, float *dx, const float *dx_add, const float *stat_x,
                                    const float *x_mean, const float *x_invstd, float *dstat_partial,
                                    float *dw_partial, int N, int H, int W, int C, int stride, int rate, int pad_t,
                                    int pad_l, int Ho, int Wo, int impl, void *stream) {
  DL3_CHECK_ARG(stat_x, "dwconv3x3_bwd_sx: stat_x is NULL (use dl3_dwconv3x3_bwd)");
  return dwconv3x3_bwd_impl(g, yraw, cA, cB, cC, x, in_scale, in_shift, in_act, w, dx, dx_add, x_mean, x_invstd,
                            dstat_partial, dw_partial, N, H, W, C, stride, rate, pad_t, pad_l, Ho, Wo, impl, stat_x, stream);
}
