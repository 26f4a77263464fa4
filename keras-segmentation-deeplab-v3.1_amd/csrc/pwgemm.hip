// pwgemm.hip — Conv2D 1x1 as fp32 GEMM on the CDNA4 matrix cores
// (deeplabv3p.py:78-79,:175,:194,:377,:385,:406,:420,:438; utils.py:189,:195).
//
// v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD).  One kernel family serves
//   forward   Y[M,N]  = T(X)[M,K] . W[K,N]            (prologue: BN+ReLU6 of the producer on load,
//                                                      epilogue: bias, BN batch-stat partials)
//   bwd-data  dX[M,K] = dY[M,N] . W^T[N,K]            (prologue: BN-backward affine of two tensors on
//                                                      load, epilogue: activation mask, residual
//                                                      gradient add, BN-backward stat partials)
// and a second one the weight gradient dW[K,N] = T(X)^T . dY (reduction over M = N*H*W, split
// over workgroups, folded deterministically).
//
// Fragment layouts (cdna_hip_programming.md §3): A lane l holds A[i=l&31][k=l>>5], B lane l holds
// B[k=l>>5][j=l&31]; C/D reg r of lane l is row (r&3)+8*(r>>2)+4*(l>>5), column l&31.
// LDS tiles are k-major (As[k][m], Bs[k][n]) so fragment reads are 32 consecutive floats per
// half-wave: conflict-free ds_read_b32.
#include <stdlib.h>

#include <mutex>
#include <type_traits>
#include <vector>

#include "common.h"

namespace {

struct GemmArgs {
  const float *a; int lda;
  const float *a2; int lda2;
  const float *ka, *kb, *kc; int a_act;
  const float *b; int ldb;
  const float *bias;
  float *c; int ldc;
  int M, K, N;
  const float *ep_x; int ld_epx;
  const float *ep_scale, *ep_shift; int ep_act;
  const float *ep_add; int ld_add; int add_div; float add_scale;
  int stat_mode;  // 0 none, 1: sum c, sum c^2 ; 2: sum c, sum c*xhat
  const float *ep_mean, *ep_invstd;
  float *part;
  int part_rows;  // rows the caller's partial buffer holds: the launch writes gridDim.y of them and zeroes the rest itself
  int part_ld;    // columns per partial row (0: N) — a launch over a column slice of the output (run_gemm) writes into the whole matrix's rows
  int mtiles;
  const void *bp; int nsub;  // split math: weights pre-split into 3 bf16 planes in MFMA fragment order (pack_b_kernel)
#ifdef DL3_PHASE_TIMING
  long long *dbg;  // probe build (tools/r3/phase_probe.py): per-workgroup cycles in prologue / K loop / epilogue
#endif
};

// probe builds only (-DDL3_PHASE_TIMING: build_variants/libdl3_timing.so, tools/r3/phase_probe.py)
#ifdef DL3_PHASE_TIMING
long long *g_phase_dbg = nullptr;
#define DL3_T(x) x
#else
#define DL3_T(x)
#endif

constexpr int BK = 16;
#ifndef DL3_STREAM_KMAX
#define DL3_STREAM_KMAX 2048  // largest reduction depth the stream kernel takes (coefficient vectors in LDS)
#endif
#ifndef DL3_STREAM_KT_FWD
#define DL3_STREAM_KT_FWD 16  // K-tile depth of the forward instantiation (32 measured in round 2: see DESIGN.md)
#endif
#ifndef DL3_STREAM_KT_BWD1
#define DL3_STREAM_KT_BWD1 16  // K-tile depth of the single-tensor bwd-data instantiation (dY materialised, round 4)
#endif
#ifndef DL3_STREAM_PD_SMALL
#define DL3_STREAM_PD_SMALL 2  // 32-row (small-M) stream kernels: K-tiles of operands in flight (register ring); 1 = the plain loop
#endif
#ifndef DL3_WGRAD_WGS_DEFAULT
#define DL3_WGRAD_WGS_DEFAULT 1024
#endif
#ifndef DL3_GEMM_PY_DEFAULT
#define DL3_GEMM_PY_DEFAULT 2048
#endif
#ifndef DL3_WGRAD_MS
#define DL3_WGRAD_MS 16
#endif


// Main loop structure (both kernels): global -> registers -> LDS, double-buffered LDS (ONE barrier per
// K-tile), MFMA operand fragments prefetched one k-step ahead.  Out-of-range rows / columns are handled by
// clamping the ADDRESS (always in bounds, so loads are unconditional and branch-free) and zeroing the VALUE
// when it is written to LDS.
// VEC: 1 = 16-byte loads on both operands, 0 = scalar loads on both, 2 (round 5) = 16-byte loads on the activation operand
// only — the logits layer (K = 256, N = classes = 21): the big operand is the aligned one
template <int TM, int TN, int WM, int WN, int VEC>
__global__ __launch_bounds__(256, 2) void pw_gemm_kernel(GemmArgs P) {
  static_assert(WM * WN == 4, "4 waves per workgroup");
  constexpr bool AVEC = VEC != 0, BVEC = VEC == 1;
  constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
  constexpr int LDA = BM + 2;  // 4*LDA == 8 (mod 32): transposed 4-float stores hit distinct banks
  constexpr int LDB = BN + 4;
  constexpr int NA = BM / 64;               // float4 A loads per thread per K-tile
  constexpr int NB = (4 * BN + 255) / 256;  // float4 B loads per thread per K-tile
  constexpr int STAGE = BK * LDA + BK * LDB;
  __shared__ float lds[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, lhi = lane >> 5;
  // XCD-aware remap (workgroup b runs on XCD b % 8): give each XCD a contiguous run of logical ids so that the
  // column tiles of one row tile — which re-read the same A rows — share one XCD's L2.  Speed only.
  const int nwg = gridDim.x * gridDim.y, b = blockIdx.y * gridDim.x + blockIdx.x;
  const int xq = nwg >> 3, xr = nwg & 7, xcd = b & 7;
  const int lid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (b >> 3);
  const int bx = lid % gridDim.x, by = lid / gridDim.x;
  const int n0 = bx * BN;
  const int akq = tid & 3, amr = tid >> 2;
  const int ktiles = (P.K + BK - 1) / BK;
  const bool xform = (P.ka != nullptr);
  const bool two = (P.a2 != nullptr);

  float st1[TN], st2[TN];
#pragma unroll
  for (int i = 0; i < TN; i++) st1[i] = st2[i] = 0.f;

  for (int mt = by; mt < P.mtiles; mt += gridDim.y) {
    const int m0 = mt * BM;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
      for (int j = 0; j < TN; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    f32x4 ra[NA], ra2[NA], rb[NB];

    auto load_tiles = [&](int kt) {
      const int k = kt * BK + akq * 4;
      if (AVEC) {
        const int kc = min(k, P.K - 4);
#pragma unroll
        for (int i = 0; i < NA; i++) {
          const int row = min(m0 + amr + 64 * i, P.M - 1);
          ra[i] = ld4(P.a + (size_t)row * P.lda + kc);
          if (two) ra2[i] = ld4(P.a2 + (size_t)row * P.lda2 + kc);
        }
      } else {
#pragma unroll
        for (int i = 0; i < NA; i++) {
          const int row = min(m0 + amr + 64 * i, P.M - 1);
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const int kc = min(k + j, P.K - 1);
            ra[i][j] = P.a[(size_t)row * P.lda + kc];
            if (two) ra2[i][j] = P.a2[(size_t)row * P.lda2 + kc];
          }
        }
      }
      if (BVEC) {
#pragma unroll
        for (int i = 0; i < NB; i++) {
          const int idx = tid + 256 * i;
          if (NB * 256 == 4 * BN || idx < 4 * BN) {
            const int kk = idx / (BN / 4), nq = idx % (BN / 4);
            const int krow = min(kt * BK + kk, P.K - 1), col = min(n0 + nq * 4, P.N - 4);
            rb[i] = ld4(P.b + (size_t)krow * P.ldb + col);
          }
        }
      } else {
#pragma unroll
        for (int i = 0; i < NB; i++) {
          const int idx = tid + 256 * i;
          if (NB * 256 == 4 * BN || idx < 4 * BN) {
            const int kk = idx / (BN / 4), nq = idx % (BN / 4);
            const int krow = min(kt * BK + kk, P.K - 1);
#pragma unroll
            for (int j = 0; j < 4; j++) rb[i][j] = P.b[(size_t)krow * P.ldb + min(n0 + nq * 4 + j, P.N - 1)];
          }
        }
      }
    };

    auto store_tiles = [&](int kt, float *As, float *Bs) {
      const int k = kt * BK + akq * 4;
      f32x4 fa = splat4(1.f), fb = splat4(0.f), fc = splat4(0.f);
      if (xform) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const int kc = min(k + j, P.K - 1);
          fa[j] = P.ka[kc];
          fc[j] = P.kc[kc];
          if (two) fb[j] = P.kb[kc];
        }
      }
#pragma unroll
      for (int i = 0; i < NA; i++) {
        const bool rok = (m0 + amr + 64 * i) < P.M;
        f32x4 v = ra[i];
        if (xform) {
          v = fa * v + fc;
          if (two) v += fb * ra2[i];
        }
        v = dl3_act4(v, P.a_act);
#pragma unroll
        for (int j = 0; j < 4; j++) As[(akq * 4 + j) * LDA + amr + 64 * i] = (rok && (k + j < P.K)) ? v[j] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < NB; i++) {
        const int idx = tid + 256 * i;
        if (NB * 256 == 4 * BN || idx < 4 * BN) {
          const int kk = idx / (BN / 4), nq = idx % (BN / 4);
          const bool kok = (kt * BK + kk) < P.K;
          f32x4 v = rb[i];
#pragma unroll
          for (int j = 0; j < 4; j++)
            if (!(kok && (n0 + nq * 4 + j < P.N))) v[j] = 0.f;
          st4(&Bs[kk * LDB + nq * 4], v);
        }
      }
    };

    load_tiles(0);
    __syncthreads();  // the previous m-tile's LDS reads (and the stat fold) are done
    store_tiles(0, lds, lds + BK * LDA);
    __syncthreads();
    for (int kt = 0; kt < ktiles; ++kt) {
      const float *As = lds + (kt & 1) * STAGE;
      const float *Bs = As + BK * LDA;
      const bool more = kt + 1 < ktiles;
      if (more) load_tiles(kt + 1);
      float af[2][TM], bf[2][TN];
#pragma unroll
      for (int i = 0; i < TM; i++) af[0][i] = As[lhi * LDA + (wm * TM + i) * 32 + l31];
#pragma unroll
      for (int j = 0; j < TN; j++) bf[0][j] = Bs[lhi * LDB + (wn * TN + j) * 32 + l31];
#pragma unroll
      for (int ks = 0; ks < BK / 2; ++ks) {
        const int cur = ks & 1, nxt = cur ^ 1;
        if (ks + 1 < BK / 2) {
#pragma unroll
          for (int i = 0; i < TM; i++) af[nxt][i] = As[(2 * ks + 2 + lhi) * LDA + (wm * TM + i) * 32 + l31];
#pragma unroll
          for (int j = 0; j < TN; j++) bf[nxt][j] = Bs[(2 * ks + 2 + lhi) * LDB + (wn * TN + j) * 32 + l31];
        }
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
          for (int j = 0; j < TN; j++)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i], bf[cur][j], acc[i][j], 0, 0, 0);
      }
      if (more) {
        float *An = lds + ((kt + 1) & 1) * STAGE;
        store_tiles(kt + 1, An, An + BK * LDA);
      }
      __syncthreads();
    }

    // ---------------- epilogue: all global reads of a 32x32 sub-tile are issued up front (clamped addresses,
    // no branches) so 16-32 loads per lane are in flight; values outside the matrix are dropped at the store
    const bool full = (m0 + BM <= P.M) && (n0 + BN <= P.N);
#pragma unroll
    for (int j = 0; j < TN; j++) {
      const int col = n0 + (wn * TN + j) * 32 + l31;
      const bool cok = col < P.N;
      const int colc = min(col, P.N - 1);
      float bias = 0.f, es = 1.f, et = 0.f, mu = 0.f, is = 0.f;
      if (P.bias) bias = P.bias[colc];
      if (P.ep_scale) { es = P.ep_scale[colc]; et = P.ep_shift[colc]; }
      if (P.stat_mode == 2) { mu = P.ep_mean[colc]; is = P.ep_invstd[colc]; }
#pragma unroll
      for (int i = 0; i < TM; i++) {
        const int rbase = m0 + (wm * TM + i) * 32 + 4 * lhi;
        float xr[16], ad[16];
        if (P.ep_x) {
#pragma unroll
          for (int r = 0; r < 16; r++) {
            const int row = min(rbase + (r & 3) + 8 * (r >> 2), P.M - 1);
            xr[r] = P.ep_x[(size_t)row * P.ld_epx + colc];
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; r++) xr[r] = 0.f;
        }
        if (P.ep_add) {
#pragma unroll
          for (int r = 0; r < 16; r++) {
            const int row = min(rbase + (r & 3) + 8 * (r >> 2), P.M - 1);
            const int arow = (P.add_div > 1) ? row / P.add_div : row;
            ad[r] = P.add_scale * P.ep_add[(size_t)arow * P.ld_add + colc];
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; r++) ad[r] = 0.f;
        }
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int row = rbase + (r & 3) + 8 * (r >> 2);
          const bool ok = full || (cok && row < P.M);
          float v = acc[i][j][r] + bias;
          if (P.ep_x) v *= dl3_act_mask(es * xr[r] + et, P.ep_act);
          v += ad[r];
          if (ok) {
            __builtin_nontemporal_store(v, &P.c[(size_t)row * P.ldc + col]);
            st1[j] += v;
            st2[j] += (P.stat_mode == 2) ? v * ((xr[r] - mu) * is) : v * v;
          }
        }
      }
    }
  }

  if (P.stat_mode != 0) {
    // fold the two half-waves (same column), then the WM waves that share columns
    float *sred = lds;  // [WM][BN][2]
    __syncthreads();
#pragma unroll
    for (int j = 0; j < TN; j++) {
      float a1 = st1[j] + __shfl_xor(st1[j], 32, 64);
      float a2 = st2[j] + __shfl_xor(st2[j], 32, 64);
      if (lhi == 0) {
        const int cl = (wn * TN + j) * 32 + l31;
        sred[(wm * BN + cl) * 2 + 0] = a1;
        sred[(wm * BN + cl) * 2 + 1] = a2;
      }
    }
    __syncthreads();
    for (int cl = tid; cl < BN; cl += 256) {
      const int col = n0 + cl;
      if (col < P.N) {
        float a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int w = 0; w < WM; w++) {
          a1 += sred[(w * BN + cl) * 2 + 0];
          a2 += sred[(w * BN + cl) * 2 + 1];
        }
        const int pld = P.part_ld ? P.part_ld : P.N;
        P.part[((size_t)by * pld + col) * 2 + 0] = a1;
        P.part[((size_t)by * pld + col) * 2 + 1] = a2;
        // rows of the buffer no workgroup owns (it is sized for the largest grid any tile choice uses): zeroed here, by
        // the row groups in turn, instead of by a memset node behind every launch
        for (int r = by + (int)gridDim.y; r < P.part_rows; r += (int)gridDim.y) {
          P.part[((size_t)r * pld + col) * 2 + 0] = 0.f;
          P.part[((size_t)r * pld + col) * 2 + 1] = 0.f;
        }
      }
    }
  }
}

// ---- split math (DL3_GEMM_MATH=split): fp32 GEMM on the bf16 matrix pipe ----------------------------------
// v_mfma_f32_32x32x2_f32 runs at the f32 VECTOR rate, 1/16 of v_mfma_f32_32x32x16_bf16.  An fp32 number is EXACTLY the
// sum of three bf16 numbers (its 24-bit significand cut into 8 + 8 + 8 bits by truncation: h = top 16 bits of x,
// m = top 16 bits of x - h, l = x - h - m, every subtraction exact), a bf16 x bf16 product is exact in fp32, and the MFMA
// accumulates in fp32.  a.b = sum over the nine piece products; the six of order >= 2^-16 are computed, the three
// dropped ones (m.l, l.m, l.l) are <= 2^-23 |a||b| — the size of ONE fp32 rounding, which the f32 MFMA commits per
// product anyway.  Six bf16 MFMAs of K=16 (6 x 32 cycles) replace eight f32 MFMAs of K=2 (8 x 64): 2.67x the matrix rate
// at fp32-roundoff-class accuracy (tests/test_gpu_ops.py::test_split_math_error measures both against float64).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// 8 consecutive-k fp32 values of one lane -> the three bf16x8 MFMA operands
__device__ __forceinline__ void split3(const f32x4 &v0, const f32x4 &v1, u32x4 &h, u32x4 &m, u32x4 &l) {
  unsigned hb[8], mb[8], lb[8];
#pragma unroll
  for (int e = 0; e < 8; e++) {
    const float x = e < 4 ? v0[e] : v1[e - 4];
    const unsigned xh = __float_as_uint(x) & 0xffff0000u;
    const float r1 = x - __uint_as_float(xh);
    const unsigned xm = __float_as_uint(r1) & 0xffff0000u;
    const float r2 = r1 - __uint_as_float(xm);
    hb[e] = xh; mb[e] = xm; lb[e] = __float_as_uint(r2);
  }
#pragma unroll
  for (int q = 0; q < 4; q++) {  // word q = [element 2q+1 : element 2q], upper halves of both
    h[q] = __builtin_amdgcn_perm(hb[2 * q + 1], hb[2 * q], 0x07060302u);
    m[q] = __builtin_amdgcn_perm(mb[2 * q + 1], mb[2 * q], 0x07060302u);
    l[q] = __builtin_amdgcn_perm(lb[2 * q + 1], lb[2 * q], 0x07060302u);
  }
}

// W[K][N] (row stride ldb) -> out[K-tile of 32][k-step s][plane][sub-tile][lane] (16 B each): lane l of sub-tile j holds
// column 32 j + (l & 31) and k = 32 kt + 16 (l >> 5) + 8 s + 0..7 — the B fragment of v_mfma_f32_32x32x16_bf16 under the
// stream kernel's k map.  Zero beyond K / N.
__global__ __launch_bounds__(256) void pack_b_kernel(const float *__restrict__ b, int ldb, int K, int N,
                                                     u32x4 *__restrict__ out, int ktiles, int nsub) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= ktiles * 2 * nsub * 64) return;
  const int lane = t & 63, j = (t >> 6) % nsub, ks = ((t >> 6) / nsub) & 1, kt = (t >> 6) / nsub / 2;
  const int col = j * 32 + (lane & 31), k0 = kt * 32 + 16 * (lane >> 5) + 8 * ks;
  f32x4 v0, v1;
#pragma unroll
  for (int e = 0; e < 8; e++) {
    const int k = k0 + e;
    const float x = (k < K && col < N) ? b[(size_t)k * ldb + col] : 0.f;
    if (e < 4) v0[e] = x; else v1[e - 4] = x;
  }
  u32x4 h, m, l;
  split3(v0, v1, h, m, l);
  const size_t base = ((size_t)(kt * 2 + ks) * 3 * nsub + j) * 64 + lane;
  out[base] = h;
  out[base + (size_t)nsub * 64] = m;
  out[base + (size_t)2 * nsub * 64] = l;
}

// Epilogue of the stream kernel for a tile that lies inside the matrix, specialised at compile time on what it has to
// do.  (The generic version below tests its run-time flags and the bounds on every element; its exec-mask branches and
// waits cost ~28k cycles per 128x160 tile — two thirds of the main loop of a K=160 GEMM.)  EPX: activation mask from
// the forward input (bwd-data); ADD: residual gradient; MODE2: second statistic is sum(c * xhat) instead of sum(c^2).
// All 16 (32 with ADD) global reads of a 32x32 sub-tile are issued before the first use.
template <int TM, int TN, bool EPX, bool ADD, bool MODE2>
__device__ __forceinline__ void stream_epilogue_full(const GemmArgs &P, const f32x16 (&acc)[TM][TN], int m0, int n0,
                                                     int wm, int l31, int lhi, float (&st1)[TN], float (&st2)[TN]) {
  // addresses = wave-uniform base (scalar registers) + ONE 32-bit lane offset per tensor: the rows of a sub-tile differ
  // by uniform multiples of the leading dimension, only (lane >> 5, lane & 31) is lane-specific
  const int wmu = __builtin_amdgcn_readfirstlane(wm);
  const unsigned lo_c = (unsigned)(4 * lhi * P.ldc + l31);
  const unsigned lo_x = EPX ? (unsigned)(4 * lhi * P.ld_epx + l31) : 0u;
  const unsigned lo_a = ADD ? (unsigned)(4 * lhi * P.ld_add + l31) : 0u;
#pragma unroll
  for (int j = 0; j < TN; j++) {
    const int col = n0 + j * 32 + l31;
    float bias = 0.f, es = 1.f, et = 0.f, mu = 0.f, is = 0.f;
    if (P.bias) bias = P.bias[col];
    if (EPX && P.ep_scale) { es = P.ep_scale[col]; et = P.ep_shift[col]; }
    if (MODE2) { mu = P.ep_mean[col]; is = P.ep_invstd[col]; }
#pragma unroll
    for (int i = 0; i < TM; i++) {
      const size_t urow = (size_t)(m0 + (wmu * TM + i) * 32);
      const int ucol = n0 + j * 32;
      float *pc = P.c + urow * P.ldc + ucol;
      const float *px = EPX ? P.ep_x + urow * P.ld_epx + ucol : nullptr;
      const float *pa = ADD ? P.ep_add + urow * P.ld_add + ucol : nullptr;
      float xr_[16], ad[16];
      if (EPX) {
#pragma unroll
        for (int r = 0; r < 16; r++) xr_[r] = (px + (size_t)((r & 3) + 8 * (r >> 2)) * P.ld_epx)[lo_x];
      }
      if (ADD) {
        if (P.add_div > 1) {
          // one addend row per image (run_gemm only sends a launch here when 32-row blocks never straddle two images)
          const float a0 = P.add_scale * P.ep_add[(urow / (size_t)P.add_div) * P.ld_add + ucol + l31];
#pragma unroll
          for (int r = 0; r < 16; r++) ad[r] = a0;
        } else {
#pragma unroll
          for (int r = 0; r < 16; r++) ad[r] = P.add_scale * (pa + (size_t)((r & 3) + 8 * (r >> 2)) * P.ld_add)[lo_a];
        }
      }
#pragma unroll
      for (int r = 0; r < 16; r++) {
        float v = acc[i][j][r] + bias;
        if (EPX) v *= dl3_act_mask(es * xr_[r] + et, P.ep_act);
        if (ADD) v += ad[r];
#ifndef DL3_DBG_NOSTORE
        __builtin_nontemporal_store(v, &(pc + (size_t)((r & 3) + 8 * (r >> 2)) * P.ldc)[lo_c]);
#else
        if (v == 1.2345e30f) __builtin_nontemporal_store(v, &(pc + (size_t)((r & 3) + 8 * (r >> 2)) * P.ldc)[lo_c]);
#endif
#ifndef DL3_DBG_NOSTAT
        st1[j] += v;
        st2[j] += MODE2 ? v * ((xr_[r] - mu) * is) : v * v;
#endif
      }
      // keep the sub-tiles apart: interleaving them would hold several sub-tiles' operands live on top of the
      // accumulators (spills)
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// FWD: the launch has no masked epilogue (no ep_x, stat_mode != 2, no per-image broadcast residual) — the forward
// GEMMs; their interior tiles take the straight-line epilogue and the masked code is not compiled in at all.
// WN: waves side by side along N (4/WN stacked along M).  WN = 4 gives 32-row tiles for small M (batch 1-4: a
// 128-row tile leaves most CUs idle and every workgroup a long serial K loop).
template <int TM, int TN, bool TWO, int KT, int EPI, int WN = 1, int MATH = 0>
__global__ __launch_bounds__(256, 2) void pw_gemm_stream_kernel(GemmArgs P) {
  // MATH 1: split math (see split3) — the A registers are split after the operand transform, the weight tile comes
  // pre-split from pack_b_kernel, six bf16 MFMAs per 16-deep K-tile and 32x32 sub-tile
  constexpr bool SPL = (MATH == 1);
  static_assert(!SPL || KT == 32, "split math: two bf16 MFMA k-steps per K-tile");
  // EPI: 0 generic epilogue only, 1 forward (see above).  A straight-line MASKED bwd-data variant was measured too:
  // on top of 80 accumulators its operand registers push long-lived values into scratch and it came out slower.
  constexpr bool FWD = (EPI == 1);
  // EPI 2 (bwd-data without residual, TM = 1, TN <= 3): the masked epilogue's operand — the forward input x of the
  // tile — is requested BEFORE the main loop and consumed after it.  The generic epilogue
  // fetches them sub-tile by sub-tile after the last MFMA, one exposed HBM round trip each (~4 us per sub-tile; for a
  // reduction of 96-320 that is as long as the main loop itself, and the in-situ table shows exactly these launches
  // running at mfma-time + hbm-time).  48 accumulators + 48 operand registers fit two waves per SIMD next to the main
  // loop's own; 80 + 80 do not (nor 48 + 96 with a residual operand: measured, spills between the loads).
  constexpr bool PRE = (EPI == 2);
  static_assert(!PRE || (TM == 1 && WN == 1 && TN <= 3), "prefetched epilogue: narrow single-row-block tiles only");
  // (round 4, single-tensor operand: the same at 128x160 — 80 accumulators + 80 prefetched operands — still does not fit:
  // 200 B of scratch, 30-40 of the operands are parked there as they arrive, which puts their HBM round trip in front of
  // the K loop again: bwd-data GEMMs 23.1 -> 25.0 ms per step, profiles/r04_ab_calls.txt call 13)
  // (EPI 3 / 4, round 4: a straight-line MASKED epilogue for the single-tensor bwd-data launches — forward input for the
  // mask and x_hat, optional residual, BatchNorm-backward sums; 0-260 B of scratch once the epilogue addresses were
  // fenced — measured no faster than the generic epilogue below: bwd-data GEMMs 25.95 ms per step with it, 25.81 without
  // (B=128, same call, profiles/r04_ab_dy_epilogue.txt); removed again)
  // KT = depth of one K-tile: each half-wave walks KT/2 consecutive k (KT/8 float4 loads per lane and tensor)
  constexpr int WM = 4 / WN;
  constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN, KH = KT / 2, NJ = KT / 8;
  constexpr int LDB = BN;
  constexpr int NS = KT / 16;                    // split math: bf16 MFMA k-steps per K-tile
  constexpr int BQ = SPL ? NS * 3 * (BN / 32) * 64 : KT * BN / 4;  // 16-byte pieces of one K-tile's weight tile
  constexpr int NB = SPL ? 1 : (KT * BN / 4 + 255) / 256;  // float4 B loads per thread per K-tile (f32 path)
  constexpr int KCS = DL3_STREAM_KMAX + KT;      // per-k operand-transform coefficients live in LDS
  __shared__ float lds[2 * BQ * 4];
  // split math: the coefficient vectors are sized by the launch (dynamic LDS, 4 * (2 or 3) * ktiles * KT bytes) so that
  // two workgroups of the 160-wide tile still fit a CU next to the 60 KB of split weight tiles
  __shared__ float cf_static[SPL ? 1 : (TWO ? 3 : 2) * KCS];
  extern __shared__ float cf_dyn[];
  float *const cf = SPL ? cf_dyn : cf_static;
  __shared__ float eco[PRE ? 4 * BN : 1];  // prefetched epilogue: per-column scale, shift, mean, invstd (n0 is fixed)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int nwg = gridDim.x * gridDim.y, b = blockIdx.y * gridDim.x + blockIdx.x;
  const int xq = nwg >> 3, xr = nwg & 7, xcd = b & 7;
  const int lid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (b >> 3);
  const int bx = lid % gridDim.x, by = lid / gridDim.x;
  const int n0 = bx * BN;
  const int ktiles = (P.K + KT - 1) / KT;
  const int KC = SPL ? ktiles * KT : KCS;
  const bool xform = (P.ka != nullptr);
  const int wm = wave / WN, wn = wave % WN;
  const int nw0 = n0 + wn * TN * 32;  // first column of this wave's sub-tile

  float st1[TN], st2[TN];
#pragma unroll
  for (int i = 0; i < TN; i++) st1[i] = st2[i] = 0.f;

  // T(a)[k] = act(ka[k]*a + kb[k]*a2 + kc[k]); k >= K gets all-zero coefficients, so the clamped (in-bounds) loads
  // beyond K contribute act(0) = 0
  for (int i = tid; i < ktiles * KT; i += 256) {
    const bool in = i < P.K;
    const int k = min(i, P.K - 1);
    cf[i] = in ? (xform ? P.ka[k] : 1.f) : 0.f;
    cf[KC + i] = (in && xform) ? P.kc[k] : 0.f;
    if (TWO) cf[2 * KC + i] = in ? P.kb[k] : 0.f;
  }
  if constexpr (PRE) {
    for (int i = tid; i < BN; i += 256) {
      const int col = min(n0 + i, P.N - 1);
      eco[i] = P.ep_scale ? P.ep_scale[col] : 1.f;
      eco[BN + i] = P.ep_scale ? P.ep_shift[col] : 0.f;
      eco[2 * BN + i] = P.stat_mode == 2 ? P.ep_mean[col] : 0.f;
      eco[3 * BN + i] = P.stat_mode == 2 ? P.ep_invstd[col] : 0.f;
    }
  }
  __syncthreads();

  DL3_T(long long tp0 = 0; long long tp1 = 0; long long tp2 = 0; long long tq0 = 0; long long tq1 = 0; long long tq2 = 0; int ntl = 0;
        long long tw_vm = 0; long long tw_bar = 0;)
  for (int mt = by; mt < P.mtiles; mt += gridDim.y) {
    DL3_T(tq0 = clock64(); ntl++;)
    const int m0 = mt * BM;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
      for (int j = 0; j < TN; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    const bool full = (m0 + BM <= P.M) && (n0 + BN <= P.N);
    float pxv[PRE ? TN : 1][16];
    if constexpr (PRE) {
      if (full) {
        const size_t urow = (size_t)(m0 + __builtin_amdgcn_readfirstlane(wm) * 32);
        const float *px = P.ep_x + urow * P.ld_epx + nw0;
        const unsigned lo_x = (unsigned)(4 * lhi * P.ld_epx + l31);
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
          for (int r = 0; r < 16; r++) pxv[j][r] = (px + (size_t)((r & 3) + 8 * (r >> 2)) * P.ld_epx + j * 32)[lo_x];
      }
    }

    const float *arow[TM], *arow2[TM];
#pragma unroll
    for (int i = 0; i < TM; i++) {
      const int row = min(m0 + (wm * TM + i) * 32 + l31, P.M - 1);
      arow[i] = P.a + (size_t)row * P.lda;
      arow2[i] = TWO ? P.a2 + (size_t)row * P.lda2 : nullptr;
    }
    f32x4 an[TM][NJ], an2[TM][NJ], ac[TM][NJ], rb[NB];

    // (the `_to` / `_from` forms name the register set: the ring loop of the 32-row kernels keeps several K-tiles in flight)
    auto load_A_to = [&](int kt, f32x4 (&da)[TM][NJ], f32x4 (&da2)[TM][NJ]) {
#pragma unroll
      for (int j = 0; j < NJ; j++) {
        const int kc = min(kt * KT + KH * lhi + 4 * j, P.K - 4);
#pragma unroll
        for (int i = 0; i < TM; i++) {
          da[i][j] = ld4(arow[i] + kc);
          if (TWO) da2[i][j] = ld4(arow2[i] + kc);
        }
      }
    };
    auto load_B_to = [&](int kt, f32x4 (&db)[NB]) {
#pragma unroll
      for (int i = 0; i < NB; i++) {
        const int idx = tid + 256 * i;
        if (NB * 256 == KT * BN / 4 || idx < KT * BN / 4) {
          const int kk = idx / (BN / 4), nq = idx % (BN / 4);
          const int krow = min(kt * KT + kk, P.K - 1), col = min(n0 + nq * 4, P.N - 4);
          db[i] = ld4(P.b + (size_t)krow * P.ldb + col);
        }
      }
    };
    auto store_B_from = [&](float *Bs, const f32x4 (&db)[NB]) {
#pragma unroll
      for (int i = 0; i < NB; i++) {
        const int idx = tid + 256 * i;
        if (NB * 256 == KT * BN / 4 || idx < KT * BN / 4) st4(&Bs[(idx / (BN / 4)) * LDB + (idx % (BN / 4)) * 4], db[i]);
      }
    };
    auto load_A = [&](int kt) { load_A_to(kt, an, an2); };
    auto load_B = [&](int kt) { load_B_to(kt, rb); };
    auto store_B = [&](float *Bs) { store_B_from(Bs, rb); };
    // loaded registers of K-tile kt -> MFMA operand values.  f32 path: straight into the operand registers ac, AFTER the
    // last MFMA of the running K-tile has been issued (round 3: a wave's own VALU work does not overlap its own MFMAs, so
    // there is nothing to gain from placing it earlier, and the separate copy an -> ac cost 8 v_mov per K-tile);
    // split path: in place (an <- T(an)), split3 reads it
    auto transform_from = [&](int kt, f32x4 (&sa)[TM][NJ], const f32x4 (&sa2)[TM][NJ]) {
#pragma unroll
      for (int j = 0; j < NJ; j++) {
        const int k = kt * KT + KH * lhi + 4 * j;
        const f32x4 fa = ld4(cf + k), fc = ld4(cf + KC + k);
#pragma unroll
        for (int i = 0; i < TM; i++) {
          f32x4 v = fa * sa[i][j] + fc;
          if (TWO) v += ld4(cf + 2 * KC + k) * sa2[i][j];
          if constexpr (SPL) sa[i][j] = dl3_act4(v, P.a_act);
          else ac[i][j] = dl3_act4(v, P.a_act);
        }
      }
    };
    auto transform = [&](int kt) { transform_from(kt, an, an2); };

    if constexpr (SPL) {
      // K-tile kt: [weights of kt+1 -> LDS by DMA] [MFMAs of k-step 0] [A of kt+1: wait, transform, split; request A of
      // kt+2] [MFMAs of k-step 1] [barrier].  The A loads have a whole K-tile of MFMAs (x2 waves per SIMD) to arrive.
      // Slot (k-step s, half h, element e) of a K-tile holds k = 16 h + 8 s + e — each lane's 16 k are consecutive in
      // memory; pack_b_kernel lays the weights out with the same map.
      u32x4 ch[NS][TM], cm[NS][TM], cl[NS][TM], nh[NS][TM], nm[NS][TM], nl[NS][TM];
      auto dma_B = [&](int kt, int buf) {
        constexpr int PIECES = NS * 3 * (BN / 32);  // 1 KB each: 64 lanes x 16 B, lane-linear in LDS
        const char *src0 = (const char *)P.bp + (size_t)lane * 16;
        for (int pc = __builtin_amdgcn_readfirstlane(wave); pc < PIECES; pc += 4) {
          const int sub = pc % (BN / 32), pl = (pc / (BN / 32)) % 3, ks = pc / (3 * (BN / 32));
          const int gsub = min(n0 / 32 + sub, P.nsub - 1);
          const size_t off = (((size_t)(kt * NS + ks) * 3 + pl) * P.nsub + gsub) * 1024;  // wave-uniform
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src0 + off),
                                           (__attribute__((address_space(3))) void *)(lds + (buf * BQ + pc * 64) * 4), 16, 0, 0);
        }
      };
      auto split_next = [&]() {
#pragma unroll
        for (int ks = 0; ks < NS; ks++)
#pragma unroll
          for (int i = 0; i < TM; i++) split3(an[i][2 * ks], an[i][2 * ks + 1], nh[ks][i], nm[ks][i], nl[ks][i]);
      };
      auto take_next = [&]() {
#pragma unroll
        for (int ks = 0; ks < NS; ks++)
#pragma unroll
          for (int i = 0; i < TM; i++) { ch[ks][i] = nh[ks][i]; cm[ks][i] = nm[ks][i]; cl[ks][i] = nl[ks][i]; }
      };
      auto mfma_step = [&](const float *Bs, int ks) {
        const u32x4 *Bq = (const u32x4 *)Bs + (ks * 3 * (BN / 32) + wn * TN) * 64 + lane;
#pragma unroll
        for (int j = 0; j < TN; j++) {
          const bf16x8 bh = __builtin_bit_cast(bf16x8, Bq[j * 64]);
          const bf16x8 bm = __builtin_bit_cast(bf16x8, Bq[((BN / 32) + j) * 64]);
          const bf16x8 bl = __builtin_bit_cast(bf16x8, Bq[(2 * (BN / 32) + j) * 64]);
#pragma unroll
          for (int i = 0; i < TM; i++) {
            const bf16x8 ah = __builtin_bit_cast(bf16x8, ch[ks][i]), am = __builtin_bit_cast(bf16x8, cm[ks][i]),
                         al = __builtin_bit_cast(bf16x8, cl[ks][i]);
            f32x16 c = acc[i][j];  // small terms first
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, c, 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c, 0, 0, 0);
          }
        }
      };
      // Every wave DMAs only its own pieces of a weight tile and, after the barrier, reads ALL of them: a wave must not
      // reach the barrier with its DMA still in flight.  s_barrier does not imply it on gfx950 (back-off barriers), so
      // the vmcnt(0) is spelled out (0x0F70 = vmcnt 0, expcnt / lgkmcnt untouched) instead of left to the fence the
      // compiler happens to emit for __syncthreads today.  (Waiting for LESS — a counted vmcnt that keeps the operand
      // request of K-tile kt+2 in flight across a raw s_barrier — was measured in round 3: split math 1 235-1 245 img/s
      // either way.)
      load_A(0);
      __syncthreads();  // the previous row tile is done with the LDS
      dma_B(0, 0);
      transform(0);
      split_next();
      take_next();
      if (ktiles > 1) load_A(1);
      __builtin_amdgcn_s_waitcnt(0x0F70);
      __syncthreads();
      DL3_T(tq1 = clock64();)
      for (int kt = 0; kt < ktiles; ++kt) {
        const float *Bs = lds + (kt & 1) * BQ * 4;
        const bool more = kt + 1 < ktiles;
        if (more) dma_B(kt + 1, (kt + 1) & 1);
        mfma_step(Bs, 0);
        if (more) {
          transform(kt + 1);
          split_next();
          if (kt + 2 < ktiles) load_A(kt + 2);
        }
        mfma_step(Bs, 1);
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();
        if (more) take_next();
      }
    } else if constexpr (WN == 4 && DL3_STREAM_PD_SMALL > 1) {
    // 32-row tiles (small M: a few hundred workgroups, each a long serial chain of short K-tiles — 8-16 MFMAs per wave):
    // a ring of PD register sets keeps PD K-tiles of both operands in flight; __syncthreads only waits for the LDS
    // (lgkmcnt) on gfx950, so the requests survive the barriers.  The loop is unrolled by PD so that the set index is a
    // constant.  Measured (round 4, call 14, same box): B=2 478.5 / 477.7 img/s with PD = 1, 486.2 / 486.1 with 2, 480.2 /
    // 480.3 with 3 (251-254 VGPRs: two workgroups per CU instead of three); B=4 683 / 690 / 687; B=16 unchanged.  The exposed
    // round trip was a small part of these K-tiles after all — their 8-16 MFMAs on a 160-wide output padded to 256 columns
    // are most of the 0.7 us each takes.
    constexpr int PD = (TWO && DL3_STREAM_PD_SMALL > 2) ? 2 : DL3_STREAM_PD_SMALL;  // (two-tensor operand: a third set spills)
    f32x4 ra[PD][TM][NJ], ra2[PD][TM][NJ], rbb[PD][NB];
#pragma unroll
    for (int d = 0; d < PD; d++)
      if (d < ktiles) {
        load_B_to(d, rbb[d]);
        load_A_to(d, ra[d], ra2[d]);
      }
    __syncthreads();  // the previous row tile is done with the LDS
    store_B_from(lds, rbb[0]);
    transform_from(0, ra[0], ra2[0]);
    __syncthreads();
    DL3_T(tq1 = clock64();)
    for (int kt0 = 0; kt0 < ktiles; kt0 += PD) {
#pragma unroll
      for (int d = 0; d < PD; d++) {
        const int kt = kt0 + d;
        if (kt < ktiles) {  // (the same for every wave of the workgroup: the barrier below is safe)
          const float *Bs = lds + (kt & 1) * KT * LDB;
          // set d (K-tile kt) went to the LDS / the operand registers at the end of the previous K-tile: refill it
          if (kt + PD < ktiles) {
            load_B_to(kt + PD, rbb[d]);
            load_A_to(kt + PD, ra[d], ra2[d]);
          }
          float bf[2][TN];
#pragma unroll
          for (int j = 0; j < TN; j++) bf[0][j] = Bs[(KH * lhi) * LDB + (wn * TN + j) * 32 + l31];
#pragma unroll
          for (int s_ = 0; s_ < KH; ++s_) {
            const int cur = s_ & 1, nxt = cur ^ 1;
            if (s_ + 1 < KH) {
#pragma unroll
              for (int j = 0; j < TN; j++) bf[nxt][j] = Bs[(KH * lhi + s_ + 1) * LDB + (wn * TN + j) * 32 + l31];
            }
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
              for (int j = 0; j < TN; j++)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[i][s_ >> 2][s_ & 3], bf[cur][j], acc[i][j], 0, 0, 0);
          }
          if (kt + 1 < ktiles) {
            const int dn = (d + 1) % PD;  // (a constant once the loop is unrolled)
            store_B_from(lds + ((kt + 1) & 1) * KT * LDB, rbb[dn]);
            transform_from(kt + 1, ra[dn], ra2[dn]);
          }
          __syncthreads();
        }
      }
    }
    } else {
    load_A(0);
    load_B(0);
    __syncthreads();  // the previous row tile is done with the LDS
    store_B(lds);
    transform(0);
    __syncthreads();
    DL3_T(tq1 = clock64();)
    for (int kt = 0; kt < ktiles; ++kt) {
      const float *Bs = lds + (kt & 1) * KT * LDB;
      const bool more = kt + 1 < ktiles;
      if (more) {
        load_A(kt + 1);
        load_B(kt + 1);
      }
      float bf[2][TN];
#pragma unroll
      for (int j = 0; j < TN; j++) bf[0][j] = Bs[(KH * lhi) * LDB + (wn * TN + j) * 32 + l31];
#pragma unroll
      for (int s_ = 0; s_ < KH; ++s_) {
        const int cur = s_ & 1, nxt = cur ^ 1;
        if (s_ + 1 < KH) {
#pragma unroll
          for (int j = 0; j < TN; j++) bf[nxt][j] = Bs[(KH * lhi + s_ + 1) * LDB + (wn * TN + j) * 32 + l31];
        }
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
          for (int j = 0; j < TN; j++)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[i][s_ >> 2][s_ & 3], bf[cur][j], acc[i][j], 0, 0, 0);
      }
      // next tile's weight tile -> LDS and operand transform, behind the last MFMA of this K-tile
      if (more) {
        DL3_T(const long long w0 = clock64(); __builtin_amdgcn_s_waitcnt(0x0F70); tw_vm += clock64() - w0;)
        store_B(lds + ((kt + 1) & 1) * KT * LDB);
        transform(kt + 1);
      }
      DL3_T(const long long w1 = clock64();)
      __syncthreads();
      DL3_T(tw_bar += clock64() - w1;)
    }

    }

    // ---------------- epilogue (same C/D layout as the LDS-staged kernel)
    DL3_T(tq2 = clock64(); tp0 += tq1 - tq0; tp1 += tq2 - tq1;)
#ifndef DL3_EPI_NOFENCE
    // The epilogue's addresses depend only on the tile coordinates: left alone, the scheduler computes them BEFORE the
    // K loop and spills them across it (round 4, kernel-resource-usage: 75 scratch stores per row tile in front of the
    // loop, ~100 reloads behind it in the 128x160 masked kernel).  Laundering the coordinates through an empty asm behind
    // the last MFMA pins that arithmetic where it is used.
    int m0e = __builtin_amdgcn_readfirstlane(m0), nw0e = __builtin_amdgcn_readfirstlane(nw0);  // (wave-uniform both)
    asm volatile("" : "+s"(m0e), "+s"(nw0e));
#else
    const int m0e = m0, nw0e = nw0;
#endif
    if (FWD && full) {
      if (!P.ep_add) stream_epilogue_full<TM, TN, false, false, false>(P, acc, m0e, nw0e, wm, l31, lhi, st1, st2);
      else stream_epilogue_full<TM, TN, false, true, false>(P, acc, m0e, nw0e, wm, l31, lhi, st1, st2);
      DL3_T(tp2 += clock64() - tq2;)
      continue;
    }
    if constexpr (PRE) {
      if (full) {
        float *pc = P.c + (size_t)(m0e + __builtin_amdgcn_readfirstlane(wm) * 32) * P.ldc + nw0e;
        const unsigned lo_c = (unsigned)(4 * lhi * P.ldc + l31);
        const bool mode2 = P.stat_mode == 2;
#pragma unroll
        for (int j = 0; j < TN; j++) {
          const int cl = j * 32 + l31;
          const float es = eco[cl], et = eco[BN + cl], mu = eco[2 * BN + cl], is = eco[3 * BN + cl];
#pragma unroll
          for (int r = 0; r < 16; r++) {
            const float xv = pxv[j][r];
            float v = acc[0][j][r] * dl3_act_mask(es * xv + et, P.ep_act);
            __builtin_nontemporal_store(v, &(pc + (size_t)((r & 3) + 8 * (r >> 2)) * P.ldc + j * 32)[lo_c]);
            st1[j] += v;
            st2[j] += mode2 ? v * ((xv - mu) * is) : v * v;
          }
        }
        DL3_T(tp2 += clock64() - tq2;)
        continue;
      }
    }
    // everything else: generic path, every element predicated
#pragma unroll
    for (int j = 0; j < TN; j++) {
      const int col = nw0e + j * 32 + l31;
      const bool cok = col < P.N;
      const int colc = min(col, P.N - 1);
      float bias = 0.f, es = 1.f, et = 0.f, mu = 0.f, is = 0.f;
      if (P.bias) bias = P.bias[colc];
      if (!FWD && P.ep_scale) { es = P.ep_scale[colc]; et = P.ep_shift[colc]; }
      if (!FWD && P.stat_mode == 2) { mu = P.ep_mean[colc]; is = P.ep_invstd[colc]; }
#pragma unroll
      for (int i = 0; i < TM; i++) {
        const int rbase = m0e + (wm * TM + i) * 32 + 4 * lhi;
        float xr_[16], ad[16];
        if (!FWD && P.ep_x) {
#pragma unroll
          for (int r = 0; r < 16; r++) {
            const int row = min(rbase + (r & 3) + 8 * (r >> 2), P.M - 1);
            xr_[r] = P.ep_x[(size_t)row * P.ld_epx + colc];
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; r++) xr_[r] = 0.f;
        }
        if (P.ep_add) {
#pragma unroll
          for (int r = 0; r < 16; r++) {
            const int row = min(rbase + (r & 3) + 8 * (r >> 2), P.M - 1);
            const int arow_ = (P.add_div > 1) ? row / P.add_div : row;
            ad[r] = P.add_scale * P.ep_add[(size_t)arow_ * P.ld_add + colc];
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; r++) ad[r] = 0.f;
        }
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int row = rbase + (r & 3) + 8 * (r >> 2);
          const bool ok = full || (cok && row < P.M);
          float v = acc[i][j][r] + bias;
          if (!FWD && P.ep_x) v *= dl3_act_mask(es * xr_[r] + et, P.ep_act);
          v += ad[r];
          if (ok) {
            __builtin_nontemporal_store(v, &P.c[(size_t)row * P.ldc + col]);
            st1[j] += v;
            st2[j] += (!FWD && P.stat_mode == 2) ? v * ((xr_[r] - mu) * is) : v * v;
          }
        }
      }
    }
    DL3_T(tp2 += clock64() - tq2;)
  }
#ifdef DL3_PHASE_TIMING
  if (P.dbg && lane == 0) {
    long long *d = P.dbg + ((size_t)b * 4 + wave) * 8;
    d[0] = tp0; d[1] = tp1; d[2] = tp2; d[3] = ntl; d[4] = tw_vm; d[5] = tw_bar;
  }
#endif

  if (P.stat_mode != 0) {
    float *sred = lds;  // [WM][BN][2]
    __syncthreads();
#pragma unroll
    for (int j = 0; j < TN; j++) {
      float a1 = st1[j] + __shfl_xor(st1[j], 32, 64);
      float a2 = st2[j] + __shfl_xor(st2[j], 32, 64);
      if (lhi == 0) {
        sred[(wm * BN + (wn * TN + j) * 32 + l31) * 2 + 0] = a1;
        sred[(wm * BN + (wn * TN + j) * 32 + l31) * 2 + 1] = a2;
      }
    }
    __syncthreads();
    for (int cl = tid; cl < BN; cl += 256) {
      const int col = n0 + cl;
      if (col < P.N) {
        float a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int w = 0; w < WM; w++) {
          a1 += sred[(w * BN + cl) * 2 + 0];
          a2 += sred[(w * BN + cl) * 2 + 1];
        }
        const int pld = P.part_ld ? P.part_ld : P.N;
        P.part[((size_t)by * pld + col) * 2 + 0] = a1;
        P.part[((size_t)by * pld + col) * 2 + 1] = a2;
        // rows of the buffer no workgroup owns (it is sized for the largest grid any tile choice uses): zeroed here, by
        // the row groups in turn, instead of by a memset node behind every launch
        for (int r = by + (int)gridDim.y; r < P.part_rows; r += (int)gridDim.y) {
          P.part[((size_t)r * pld + col) * 2 + 0] = 0.f;
          P.part[((size_t)r * pld + col) * 2 + 1] = 0.f;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// weight gradient: dW[K,N] (+)= sum_m T(X)[m][k] * dY[m][n]; grid (ntn, ntk, S)
// ---------------------------------------------------------------------------------------
struct WgradArgs {
  const float *x; int ldx;
  const float *xs, *xt; int x_act;
  const float *g; int ldg;
  const float *y; int ldy;
  const float *cA, *cB, *cC;
  float *ws;  // [S][K][N]
  int M, K, N, Mper;
  float *dyout; int lddy;  // nullable: dY = cA*g + cB*y + cC written out [M][N] by the workgroups of the first K-tile row
#ifdef DL3_PHASE_TIMING
  long long *dbg;
#endif
};

// VEC: 1 = 16-byte loads of x and of g / y, 0 = scalar loads of both, 2 (round 5) = 16-byte loads of x only (N = classes)
template <int TA, int TB, int WA, int WB, int VEC, bool SPL = false>
__global__ __launch_bounds__(256, 2) void pw_wgrad_kernel(WgradArgs P) {
  constexpr bool XVEC = VEC != 0, DVEC = VEC == 1;
  static_assert(WA * WB == 4, "4 waves per workgroup");
  // SPL: split math (see split3).  The reduction index is the pixel row, and a bf16 MFMA wants 8 CONSECUTIVE reduction
  // elements per lane: lane (column c, half h) gathers rows 8h..8h+7 of its column from the fp32 stage in LDS (the same
  // number of ds_read_b32 per pixel as the f32 MFMA's one-per-k-step), splits them in registers, and one 16-row stage is
  // one k-step of six bf16 MFMAs per 32x32 sub-tile.
  static_assert(!SPL || DL3_WGRAD_MS == 16, "split math: a stage is one 16-deep bf16 MFMA k-step");
  constexpr int BKT = 32 * TA * WA, BNT = 32 * TB * WB, MS = DL3_WGRAD_MS;
  constexpr int LDX = BKT + 4, LDD = BNT + 4;
  constexpr int XQ = MS * BKT / 4, DQ = MS * BNT / 4;  // float4s per stage
  constexpr int NX = (XQ + 255) / 256, ND = (DQ + 255) / 256;
  constexpr int STAGE = MS * LDX + MS * LDD;
  __shared__ float lds[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wa = wave / WB, wb = wave % WB;
  const int l31 = lane & 31, lhi = lane >> 5;
  // XCD-aware decode: the tiles of ONE row slab (same activations / gradients, different weight tiles) get consecutive
  // virtual ids and therefore one XCD and its L2.  With the plain (x, y, z) order consecutive tiles land on different
  // XCDs and every tile re-reads its slab from HBM (TCC hit rate 0-3 % measured; the 160x960 layer moved 2.5 GB instead
  // of 1.1 GB per launch and ran at HBM speed, not MFMA speed).
  const int nwg = gridDim.x * gridDim.y * gridDim.z;
  const int lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  const int xq = nwg >> 3, xr = nwg & 7, xcd = lin & 7;
  const int vid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (lin >> 3);
  const int bx = vid % gridDim.x, by_ = (vid / gridDim.x) % gridDim.y, bz = vid / (gridDim.x * gridDim.y);
  const int kbase = by_ * BKT, nbase = bx * BNT;
  const int mbeg = bz * P.Mper;
  const int mend = min(P.M, mbeg + P.Mper);
  const bool xform = (P.xs != nullptr);
  const bool two = (P.cA != nullptr);
  // the workgroups of the first K-tile row also write the gradient operand they assemble anyway, dY = cA*g + cB*y + cC,
  // to HBM (every (row, column) is staged by exactly one (bx, by_ = 0, bz)): the bwd-data GEMM of the layer then reads ONE
  // tensor instead of two and has no operand transform (round 4)
  // ... and they take turns: the gridDim.y workgroups that share a row slab (same bx, bz: they all assemble the same dY
  // stages) each write every gridDim.y-th stage, so that no workgroup carries the whole store stream (first version, all
  // stores on by_ == 0: the launch waited for those workgroups, +4.5 ms per step for 25 GB of stores)
  const bool dy_on = (P.dyout != nullptr);

  f32x16 acc[TA][TB];
#pragma unroll
  for (int i = 0; i < TA; i++)
#pragma unroll
    for (int j = 0; j < TB; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  // per-thread column positions are the same for every M stage: hoist coefficients and validity
  f32x4 xs4[NX], xt4[NX], kA4[ND], kB4[ND], kC4[ND];
  bool xok[NX][4], dok[ND][4];
#pragma unroll
  for (int i = 0; i < NX; i++) {
    const int idx = tid + 256 * i;
    const int k = kbase + (idx % (BKT / 4)) * 4;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      xok[i][j] = (k + j) < P.K;
      const int kc = min(k + j, P.K - 1);
      xs4[i][j] = xform ? P.xs[kc] : 1.f;
      xt4[i][j] = xform ? P.xt[kc] : 0.f;
    }
  }
#pragma unroll
  for (int i = 0; i < ND; i++) {
    const int idx = tid + 256 * i;
    const int col = nbase + (idx % (BNT / 4)) * 4;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      dok[i][j] = (col + j) < P.N;
      const int cc = min(col + j, P.N - 1);
      kA4[i][j] = two ? P.cA[cc] : 1.f;
      kB4[i][j] = two ? P.cB[cc] : 0.f;
      kC4[i][j] = two ? P.cC[cc] : 0.f;
    }
  }

  f32x4 rx[NX], rg[ND], ry[ND];

  auto load_tiles = [&](int m0) {
#pragma unroll
    for (int i = 0; i < NX; i++) {
      const int idx = tid + 256 * i;
      if (NX * 256 == XQ || idx < XQ) {
        const int mr = idx / (BKT / 4), kq = idx % (BKT / 4);
        const int row = min(m0 + mr, P.M - 1), k = kbase + kq * 4;
        if (XVEC) {
          rx[i] = ld4(P.x + (size_t)row * P.ldx + min(k, P.K - 4));
        } else {
#pragma unroll
          for (int j = 0; j < 4; j++) rx[i][j] = P.x[(size_t)row * P.ldx + min(k + j, P.K - 1)];
        }
      }
    }
#pragma unroll
    for (int i = 0; i < ND; i++) {
      const int idx = tid + 256 * i;
      if (ND * 256 == DQ || idx < DQ) {
        const int mr = idx / (BNT / 4), nq = idx % (BNT / 4);
        const int row = min(m0 + mr, P.M - 1), col = nbase + nq * 4;
        if (DVEC) {
          const int cc = min(col, P.N - 4);
          rg[i] = ld4(P.g + (size_t)row * P.ldg + cc);
          if (two) ry[i] = ld4(P.y + (size_t)row * P.ldy + cc);
        } else {
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const int cc = min(col + j, P.N - 1);
            rg[i][j] = P.g[(size_t)row * P.ldg + cc];
            if (two) ry[i][j] = P.y[(size_t)row * P.ldy + cc];
          }
        }
      }
    }
  };

  auto store_tiles = [&](int m0, float *Xs, float *Ds) {
    const bool dy_owner = dy_on && (((m0 - mbeg) / MS) % (int)gridDim.y == by_);
#pragma unroll
    for (int i = 0; i < NX; i++) {
      const int idx = tid + 256 * i;
      if (NX * 256 == XQ || idx < XQ) {
        const int mr = idx / (BKT / 4), kq = idx % (BKT / 4);
        const bool rok = (m0 + mr) < mend;
        f32x4 v = dl3_act4(xs4[i] * rx[i] + xt4[i], P.x_act);
#pragma unroll
        for (int j = 0; j < 4; j++)
          if (!(rok && xok[i][j])) v[j] = 0.f;
        st4(&Xs[mr * LDX + kq * 4], v);
      }
    }
#pragma unroll
    for (int i = 0; i < ND; i++) {
      const int idx = tid + 256 * i;
      if (ND * 256 == DQ || idx < DQ) {
        const int mr = idx / (BNT / 4), nq = idx % (BNT / 4);
        const bool rok = (m0 + mr) < mend;
        f32x4 v = kA4[i] * rg[i] + kC4[i];
        if (two) v += kB4[i] * ry[i];
        if (dy_owner && rok) {
          float *dp = P.dyout + (size_t)(m0 + mr) * P.lddy + nbase + nq * 4;
          if (DVEC) {
            if (dok[i][0]) st4_nt(dp, v);
          } else {
#pragma unroll
            for (int j = 0; j < 4; j++)
              if (dok[i][j]) dp[j] = v[j];
          }
        }
#pragma unroll
        for (int j = 0; j < 4; j++)
          if (!(rok && dok[i][j])) v[j] = 0.f;
        st4(&Ds[mr * LDD + nq * 4], v);
      }
    }
  };

  DL3_T(long long tw_vm = 0; long long tw_bar = 0; int nst = 0; const long long tstart = clock64();)
  if (mbeg < mend) {
    load_tiles(mbeg);
    store_tiles(mbeg, lds, lds + MS * LDX);
    __syncthreads();
    int stage = 0;
    for (int m0 = mbeg; m0 < mend; m0 += MS, stage ^= 1) {
      const float *Xs = lds + stage * STAGE;
      const float *Ds = Xs + MS * LDX;
      const bool more = (m0 + MS < mend);
      if (more) load_tiles(m0 + MS);
      if constexpr (SPL) {
        u32x4 ah[TA], am[TA], al[TA];
#pragma unroll
        for (int i = 0; i < TA; i++) {
          f32x4 v0, v1;
#pragma unroll
          for (int e = 0; e < 4; e++) {
            v0[e] = Xs[(8 * lhi + e) * LDX + (wa * TA + i) * 32 + l31];
            v1[e] = Xs[(8 * lhi + 4 + e) * LDX + (wa * TA + i) * 32 + l31];
          }
          split3(v0, v1, ah[i], am[i], al[i]);
        }
#pragma unroll
        for (int j = 0; j < TB; j++) {
          f32x4 v0, v1;
#pragma unroll
          for (int e = 0; e < 4; e++) {
            v0[e] = Ds[(8 * lhi + e) * LDD + (wb * TB + j) * 32 + l31];
            v1[e] = Ds[(8 * lhi + 4 + e) * LDD + (wb * TB + j) * 32 + l31];
          }
          u32x4 bh_, bm_, bl_;
          split3(v0, v1, bh_, bm_, bl_);
          const bf16x8 bh = __builtin_bit_cast(bf16x8, bh_), bm = __builtin_bit_cast(bf16x8, bm_),
                       bl = __builtin_bit_cast(bf16x8, bl_);
#pragma unroll
          for (int i = 0; i < TA; i++) {
            const bf16x8 xh = __builtin_bit_cast(bf16x8, ah[i]), xm = __builtin_bit_cast(bf16x8, am[i]),
                         xl = __builtin_bit_cast(bf16x8, al[i]);
            f32x16 c = acc[i][j];  // small terms first
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl, bh, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, bl, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xm, bm, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xm, bh, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, bm, c, 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, bh, c, 0, 0, 0);
          }
        }
        if (more) {
          float *Xn = lds + (stage ^ 1) * STAGE;
          store_tiles(m0 + MS, Xn, Xn + MS * LDX);
        }
        __syncthreads();
        continue;
      }
      float af[2][TA], bf[2][TB];
#pragma unroll
      for (int i = 0; i < TA; i++) af[0][i] = Xs[lhi * LDX + (wa * TA + i) * 32 + l31];
#pragma unroll
      for (int j = 0; j < TB; j++) bf[0][j] = Ds[lhi * LDD + (wb * TB + j) * 32 + l31];
#pragma unroll
      for (int ks = 0; ks < MS / 2; ++ks) {
        const int cur = ks & 1, nxt = cur ^ 1;
        if (ks + 1 < MS / 2) {
#pragma unroll
          for (int i = 0; i < TA; i++) af[nxt][i] = Xs[(2 * ks + 2 + lhi) * LDX + (wa * TA + i) * 32 + l31];
#pragma unroll
          for (int j = 0; j < TB; j++) bf[nxt][j] = Ds[(2 * ks + 2 + lhi) * LDD + (wb * TB + j) * 32 + l31];
        }
#pragma unroll
        for (int i = 0; i < TA; i++)
#pragma unroll
          for (int j = 0; j < TB; j++)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i], bf[cur][j], acc[i][j], 0, 0, 0);
      }
      if (more) {
        float *Xn = lds + (stage ^ 1) * STAGE;
        DL3_T(const long long w0 = clock64(); __builtin_amdgcn_s_waitcnt(0x0F70); tw_vm += clock64() - w0;)
        store_tiles(m0 + MS, Xn, Xn + MS * LDX);
      }
      DL3_T(const long long w1 = clock64();)
      __syncthreads();
      DL3_T(tw_bar += clock64() - w1; nst++;)
    }
  }
#ifdef DL3_PHASE_TIMING
  if (P.dbg && lane == 0) {
    long long *d = P.dbg + ((size_t)lin * 4 + wave) * 8;
    d[0] = 0; d[1] = clock64() - tstart; d[2] = 0; d[3] = nst; d[4] = tw_vm; d[5] = tw_bar;
  }
#endif
  float *out = P.ws + (size_t)bz * P.K * P.N;
#pragma unroll
  for (int i = 0; i < TA; i++)
#pragma unroll
    for (int j = 0; j < TB; j++) {
      const int col = nbase + (wb * TB + j) * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int krow = kbase + (wa * TA + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        if (krow < P.K && col < P.N) out[(size_t)krow * P.N + col] = acc[i][j][r];
      }
    }
}

// ---- round 6: the weight gradient of a layer whose whole reduction-side width fits ONE tile row (K <= 32 TA WA: the expand
// convolutions 96 -> 576 and 160 -> 960, deeplabv3p.py:175-178) — the HBM-heavy half of the family (a narrow x, a wide g / y,
// and the dY store on top: 3.4 - 3.9 TB/s in the tiled kernel).  One workgroup per row slab stages every dY stage exactly
// once, so the dY store needs no ownership at all and the stage loop is straight-line:
//   * the dY store (DYS) is a template parameter: every lane stores every piece (rows beyond M and column groups beyond N are clamped onto
//     valid ones: same address, same bits) — no store sits in a branch;
//   * the NEXT stage's operands are requested BEFORE this stage's stores: vmcnt retires in order and counts stores, so a
//     request issued behind a store cannot be waited for without waiting for the store; this way the wait is a counted
//     vmcnt(#stores) that leaves them in flight for a whole further stage (ISA: s_waitcnt vmcnt(2) in the loop);
//   * the per-column coefficient vectors live in LDS instead of 40-52 hoisted registers;
//   * waves whose 32-column blocks lie wholly beyond N (the half-empty last column tile of N = 960 / 576 on 128-wide
//     tiles) skip their MFMAs and leave the SIMD's matrix pipe to the co-resident workgroup.
// Same-box microbenchmark against pw_wgrad_kernel (profiles/r06_ab_calls.txt call 3): 96 -> 576 0.941 -> 0.860 ms, 160 -> 960
// 1.945 -> 1.841 ms.  With several workgroups per row slab (K > one tile row) the same structure LOST 2-9 % — whichever
// way the stores were shared out, the slab's workgroups fell out of step and re-read g / y from HBM instead of the XCD's
// L2 — so those launches stay on pw_wgrad_kernel.
template <int TA, int TB, int WA, int WB, bool DYS>
__global__ __launch_bounds__(256, 2) void pw_wgrad_row_kernel(WgradArgs P) {
  constexpr int VEC = 1;
  constexpr bool SPL = false;
  constexpr bool XVEC = VEC != 0, DVEC = VEC == 1;
  static_assert(WA * WB == 4, "4 waves per workgroup");
  // SPL: split math (see split3).  The reduction index is the pixel row, and a bf16 MFMA wants 8 CONSECUTIVE reduction
  // elements per lane: lane (column c, half h) gathers rows 8h..8h+7 of its column from the fp32 stage in LDS (the same
  // number of ds_read_b32 per pixel as the f32 MFMA's one-per-k-step), splits them in registers, and one 16-row stage is
  // one k-step of six bf16 MFMAs per 32x32 sub-tile.
  static_assert(!SPL || DL3_WGRAD_MS == 16, "split math: a stage is one 16-deep bf16 MFMA k-step");
  constexpr int BKT = 32 * TA * WA, BNT = 32 * TB * WB, MS = DL3_WGRAD_MS;
  constexpr int LDX = BKT + 4, LDD = BNT + 4;
  constexpr int XQ = MS * BKT / 4, DQ = MS * BNT / 4;  // float4s per stage
  constexpr int NX = (XQ + 255) / 256, ND = (DQ + 255) / 256;
  constexpr int STAGE = MS * LDX + MS * LDD;
  __shared__ float lds[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wa = wave / WB, wb = wave % WB;
  const int l31 = lane & 31, lhi = lane >> 5;
  // XCD-aware decode: the tiles of ONE row slab (same activations / gradients, different weight tiles) get consecutive
  // virtual ids and therefore one XCD and its L2.  With the plain (x, y, z) order consecutive tiles land on different
  // XCDs and every tile re-reads its slab from HBM (TCC hit rate 0-3 % measured; the 160x960 layer moved 2.5 GB instead
  // of 1.1 GB per launch and ran at HBM speed, not MFMA speed).
  const int nwg = gridDim.x * gridDim.y * gridDim.z;
  const int lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  const int xq = nwg >> 3, xr = nwg & 7, xcd = lin & 7;
  const int vid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (lin >> 3);
  const int bx = vid % gridDim.x, by_ = (vid / gridDim.x) % gridDim.y, bz = vid / (gridDim.x * gridDim.y);
  const int kbase = by_ * BKT, nbase = bx * BNT;
  const int mbeg = bz * P.Mper;
  const int mend = min(P.M, mbeg + P.Mper);
  const bool xform = (P.xs != nullptr);
  const bool two = (P.cA != nullptr);
  // 32 x 32 blocks of this wave's sub-tile that lie inside the K x N matrix (wave-uniform)
  const int ia = max(0, min(TA, (P.K - kbase - wa * TA * 32 + 31) >> 5));
  const int jb = max(0, min(TB, (P.N - nbase - wb * TB * 32 + 31) >> 5));
  const bool wfull = (ia == TA) && (jb == TB);

  f32x16 acc[TA][TB];
#pragma unroll
  for (int i = 0; i < TA; i++)
#pragma unroll
    for (int j = 0; j < TB; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  // per-column coefficient vectors of the tile (input transform of x; BatchNorm-backward affine of g, y) live in LDS — as
  // hoisted registers they were 40-52 VGPRs of every instantiation (an occupancy step for the 128 x 96 / 128 x 128 tiles).
  // Columns beyond K / N carry the LAST valid column group's values shifted in: a lane whose 16-byte piece lies beyond N
  // works on a duplicate of the last valid group (load_tiles clamps it there) and must assemble the same bits for it.
  __shared__ __attribute__((aligned(16))) float cfx[2][BKT];
  __shared__ __attribute__((aligned(16))) float cfd[3][BNT];
  for (int c = tid; c < BKT; c += 256) {
    const int kc = min(kbase + c, P.K - 1);
    cfx[0][c] = xform ? P.xs[kc] : 1.f;
    cfx[1][c] = xform ? P.xt[kc] : 0.f;
  }
  for (int c = tid; c < BNT; c += 256) {
    const int cc = min(nbase + c, P.N - 1);
    cfd[0][c] = two ? P.cA[cc] : 1.f;
    cfd[1][c] = two ? P.cB[cc] : 0.f;
    cfd[2][c] = two ? P.cC[cc] : 0.f;
  }
  __syncthreads();

  f32x4 rx[NX], rg[ND], ry[ND];

  // requests of the stage that starts at row m0 (all unconditional: clamped rows / columns / piece indices)
  auto load_tiles = [&](int m0) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NX; i++) {
      const int idx = min(tid + 256 * i, XQ - 1);
      const int mr = idx / (BKT / 4), kq = idx % (BKT / 4);
      const int row = min(m0 + mr, P.M - 1), k = kbase + kq * 4;
      if (XVEC) {
        rx[i] = ld4(P.x + (size_t)row * P.ldx + min(k, P.K - 4));
      } else {
#pragma unroll
        for (int j = 0; j < 4; j++) rx[i][j] = P.x[(size_t)row * P.ldx + min(k + j, P.K - 1)];
      }
    }
#pragma unroll
    for (int i = 0; i < ND; i++) {
      const int idx = min(tid + 256 * i, DQ - 1);
      const int mr = idx / (BNT / 4), nq = idx % (BNT / 4);
      const int row = min(m0 + mr, P.M - 1), col = nbase + nq * 4;
      if (DVEC) {
        const int cc = min(col, P.N - 4);
        rg[i] = ld4(P.g + (size_t)row * P.ldg + cc);
        if (two) ry[i] = ld4(P.y + (size_t)row * P.ldy + cc);
      } else {
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const int cc = min(col + j, P.N - 1);
          rg[i][j] = P.g[(size_t)row * P.ldg + cc];
          if (two) ry[i][j] = P.y[(size_t)row * P.ldy + cc];
        }
      }
    }
  };

  // the loaded stage (rows m0 ..) -> MFMA operands in LDS [+ dY to HBM]; the requests of the stage at mnext are issued in
  // between: behind the last use of the registers they land in, in front of this stage's stores
  auto load_x = [&](int m0) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NX; i++) {
      const int idx = min(tid + 256 * i, XQ - 1);
      const int mr = idx / (BKT / 4), kq = idx % (BKT / 4);
      const int row = min(m0 + mr, P.M - 1), k = kbase + kq * 4;
      if (XVEC) {
        rx[i] = ld4(P.x + (size_t)row * P.ldx + min(k, P.K - 4));
      } else {
#pragma unroll
        for (int j = 0; j < 4; j++) rx[i][j] = P.x[(size_t)row * P.ldx + min(k + j, P.K - 1)];
      }
    }
  };
  auto load_d = [&](int m0) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < ND; i++) {
      const int idx = min(tid + 256 * i, DQ - 1);
      const int mr = idx / (BNT / 4), nq = idx % (BNT / 4);
      const int row = min(m0 + mr, P.M - 1), col = nbase + nq * 4;
      if (DVEC) {
        const int cc = min(col, P.N - 4);
        rg[i] = ld4(P.g + (size_t)row * P.ldg + cc);
        if (two) ry[i] = ld4(P.y + (size_t)row * P.ldy + cc);
      } else {
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const int cc = min(col + j, P.N - 1);
          rg[i][j] = P.g[(size_t)row * P.ldg + cc];
          if (two) ry[i][j] = P.y[(size_t)row * P.ldy + cc];
        }
      }
    }
  };
  auto prepare = [&](int m0, int mnext, float *Xs, float *Ds, auto store_tag) __attribute__((always_inline)) {
    constexpr bool STORE = decltype(store_tag)::value;
    // x: transform -> LDS (not a vector-memory operation: it may sit anywhere), then its registers take the next stage's
#pragma unroll
    for (int i = 0; i < NX; i++) {
      const int idx = tid + 256 * i;
      const int mr = min(idx, XQ - 1) / (BKT / 4), kq = min(idx, XQ - 1) % (BKT / 4);
      const bool rok = (m0 + mr) < mend;
      const int kl = XVEC ? min(kbase + kq * 4, P.K - 4) - kbase : kq * 4;   // (the column group the load was clamped to)
      f32x4 v = dl3_act4(ld4(&cfx[0][kl]) * rx[i] + ld4(&cfx[1][kl]), P.x_act);
#pragma unroll
      for (int j = 0; j < 4; j++)
        if (!(rok && (kbase + kq * 4 + j) < P.K)) v[j] = 0.f;
      if (NX * 256 == XQ || idx < XQ) st4(&Xs[mr * LDX + kq * 4], v);
    }
    f32x4 td[ND];
#pragma unroll
    for (int i = 0; i < ND; i++) {
      const int nq_ = min(tid + 256 * i, DQ - 1) % (BNT / 4);
      const int nl = DVEC ? min(nbase + nq_ * 4, P.N - 4) - nbase : nq_ * 4;
      td[i] = ld4(&cfd[0][nl]) * rg[i] + ld4(&cfd[2][nl]);
      if (two) td[i] += ld4(&cfd[1][nl]) * ry[i];
    }
    __builtin_amdgcn_sched_barrier(0);
    load_x(mnext);
    load_d(mnext);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (STORE) {
#pragma unroll
      for (int i = 0; i < ND; i++) {
        const int idx = min(tid + 256 * i, DQ - 1);
        const int mr = idx / (BNT / 4), nq = idx % (BNT / 4);
        if constexpr (DVEC) {
          float *dp = P.dyout + (size_t)min(m0 + mr, P.M - 1) * P.lddy + min(nbase + nq * 4, P.N - 4);
          st4_nt(dp, td[i]);
        } else {
          if (m0 + mr < mend) {
            float *dp = P.dyout + (size_t)(m0 + mr) * P.lddy + nbase + nq * 4;
#pragma unroll
            for (int j = 0; j < 4; j++)
              if ((nbase + nq * 4 + j) < P.N) dp[j] = td[i][j];
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 0; i < ND; i++) {
      const int idx = tid + 256 * i;
      if (ND * 256 == DQ || idx < DQ) {
        const int mr = idx / (BNT / 4), nq = idx % (BNT / 4);
        const bool rok = (m0 + mr) < mend;
        f32x4 v = td[i];
#pragma unroll
        for (int j = 0; j < 4; j++)
          if (!(rok && (nbase + nq * 4 + j) < P.N)) v[j] = 0.f;
        st4(&Ds[mr * LDD + nq * 4], v);
      }
    }
  };

  // the MFMAs of one staged 16-row slice
  auto mfma_stage = [&](const float *Xs, const float *Ds) __attribute__((always_inline)) {
    if constexpr (SPL) {
      u32x4 ah[TA], am[TA], al[TA];
#pragma unroll
      for (int i = 0; i < TA; i++) {
        f32x4 v0, v1;
#pragma unroll
        for (int e = 0; e < 4; e++) {
          v0[e] = Xs[(8 * lhi + e) * LDX + (wa * TA + i) * 32 + l31];
          v1[e] = Xs[(8 * lhi + 4 + e) * LDX + (wa * TA + i) * 32 + l31];
        }
        split3(v0, v1, ah[i], am[i], al[i]);
      }
#pragma unroll
      for (int j = 0; j < TB; j++) {
        f32x4 v0, v1;
#pragma unroll
        for (int e = 0; e < 4; e++) {
          v0[e] = Ds[(8 * lhi + e) * LDD + (wb * TB + j) * 32 + l31];
          v1[e] = Ds[(8 * lhi + 4 + e) * LDD + (wb * TB + j) * 32 + l31];
        }
        u32x4 bh_, bm_, bl_;
        split3(v0, v1, bh_, bm_, bl_);
        const bf16x8 bh = __builtin_bit_cast(bf16x8, bh_), bm = __builtin_bit_cast(bf16x8, bm_),
                     bl = __builtin_bit_cast(bf16x8, bl_);
#pragma unroll
        for (int i = 0; i < TA; i++) {
          const bf16x8 xh = __builtin_bit_cast(bf16x8, ah[i]), xm = __builtin_bit_cast(bf16x8, am[i]),
                       xl = __builtin_bit_cast(bf16x8, al[i]);
          f32x16 c = acc[i][j];  // small terms first
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl, bh, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, bl, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xm, bm, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xm, bh, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, bm, c, 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, bh, c, 0, 0, 0);
        }
      }
    } else if (ia > 0 && jb > 0) {
      float af[2][TA], bf[2][TB];
#pragma unroll
      for (int i = 0; i < TA; i++) af[0][i] = Xs[lhi * LDX + (wa * TA + i) * 32 + l31];
#pragma unroll
      for (int j = 0; j < TB; j++) bf[0][j] = Ds[lhi * LDD + (wb * TB + j) * 32 + l31];
#pragma unroll
      for (int ks = 0; ks < MS / 2; ++ks) {
        const int cur = ks & 1, nxt = cur ^ 1;
        if (ks + 1 < MS / 2) {
#pragma unroll
          for (int i = 0; i < TA; i++) af[nxt][i] = Xs[(2 * ks + 2 + lhi) * LDX + (wa * TA + i) * 32 + l31];
#pragma unroll
          for (int j = 0; j < TB; j++) bf[nxt][j] = Ds[(2 * ks + 2 + lhi) * LDD + (wb * TB + j) * 32 + l31];
        }
#pragma unroll
        for (int i = 0; i < TA; i++)
#pragma unroll
          for (int j = 0; j < TB; j++)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i], bf[cur][j], acc[i][j], 0, 0, 0);
      }
    }
  };

    if (mbeg < mend) {
    const int ns = (mend - mbeg + MS - 1) / MS;          // stages of this workgroup
    const int mlast = mbeg + (ns - 1) * MS;              // (the last stage asks for itself again instead of for nothing)
    auto stage_x = [&](int t) { return lds + (t & 1) * STAGE; };
    load_x(mbeg);
    load_d(mbeg);
    prepare(mbeg, min(mbeg + MS, mlast), stage_x(0), stage_x(0) + MS * LDX, std::integral_constant<bool, DYS>{});
    __syncthreads();
    // (requests in flight at the loop's entry would make its header the pessimistic merge of entry and back edge — every
    // wait inside vmcnt(0): wait once here)
    __builtin_amdgcn_s_waitcnt(0x0F70);
    for (int t = 1; t < ns; ++t) {
      mfma_stage(stage_x(t - 1), stage_x(t - 1) + MS * LDX);
      prepare(mbeg + t * MS, min(mbeg + (t + 1) * MS, mlast), stage_x(t), stage_x(t) + MS * LDX, std::integral_constant<bool, DYS>{});
      __syncthreads();
    }
    mfma_stage(stage_x(ns - 1), stage_x(ns - 1) + MS * LDX);
  }
  float *out = P.ws + (size_t)bz * P.K * P.N;
#pragma unroll
  for (int i = 0; i < TA; i++)
#pragma unroll
    for (int j = 0; j < TB; j++) {
      const int col = nbase + (wb * TB + j) * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int krow = kbase + (wa * TA + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        if (krow < P.K && col < P.N) out[(size_t)krow * P.N + col] = acc[i][j][r];
      }
    }
}

// column sums of dY over row ranges: partial [PR][N]; block = 64 columns x 4 row lanes
__global__ __launch_bounds__(256) void colsum_kernel(const float *__restrict__ g, int ldg, int M, int N,
                                                     float *__restrict__ part) {
  __shared__ float red[256];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + cl;
  float s = 0.f;
  if (col < N)
    for (long m = (long)blockIdx.y * 4 + rl; m < M; m += (long)gridDim.y * 4) s += g[(size_t)m * ldg + col];
  red[threadIdx.x] = s;
  __syncthreads();
  if (rl == 0 && col < N)
    part[(size_t)blockIdx.y * N + col] = ((red[cl] + red[64 + cl]) + red[128 + cl]) + red[192 + cl];
}

__global__ __launch_bounds__(256) void transpose_kernel(const float *__restrict__ in, float *__restrict__ out,
                                                        int rows, int cols) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  for (int i = ty; i < 32; i += 8)
    if (r0 + i < rows && c0 + tx < cols) tile[i][tx] = in[(size_t)(r0 + i) * cols + c0 + tx];
  __syncthreads();
  for (int i = ty; i < 32; i += 8)
    if (c0 + i < cols && r0 + tx < rows) out[(size_t)(c0 + i) * rows + r0 + tx] = tile[tx][i];
}

// all weight transposes of a backward pass in one launch: desc[m] = {in, out, rows, cols, first tile, tiles per row}
__global__ __launch_bounds__(256) void transpose_batched_kernel(const long long *__restrict__ desc, int n) {
  __shared__ float tile[32][33];
  int m = 0;
  while (m + 1 < n && (long long)blockIdx.x >= desc[(m + 1) * 6 + 4]) ++m;  // n is a few dozen
  const float *in = (const float *)desc[m * 6 + 0];
  float *out = (float *)desc[m * 6 + 1];
  const int rows = (int)desc[m * 6 + 2], cols = (int)desc[m * 6 + 3];
  const int t = blockIdx.x - (int)desc[m * 6 + 4], tx_ = (int)desc[m * 6 + 5];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int c0 = (t % tx_) * 32, r0 = (t / tx_) * 32;
  for (int i = ty; i < 32; i += 8)
    if (r0 + i < rows && c0 + tx < cols) tile[i][tx] = in[(size_t)(r0 + i) * cols + c0 + tx];
  __syncthreads();
  for (int i = ty; i < 32; i += 8)
    if (c0 + i < cols && r0 + tx < rows) out[(size_t)(c0 + i) * rows + r0 + tx] = tile[tx][i];
}


int env_int(const char *name);

// ---------------------------------------------------------------------------------------
// round 5: weight-stationary streaming forward kernel for the HBM-bound layers
// ---------------------------------------------------------------------------------------
// The expand / project convolutions of the first blocks (deeplabv3p.py:175-198: 16..192 channels on 256x256 .. 64x64 maps)
// carry 5-14 FLOP per byte: they are streaming kernels, and the tiled GEMM above runs them at 2.3-4.6 TB/s (24 -> 144 at
// 0.29 of the HBM peak) because a 128- or 256-row tile of a reduction of 16..32 is all prologue and epilogue — one or two
// K-tiles, a weight tile re-staged per row tile, 16 four-byte stores per 32x32 sub-tile and lane, three column tiles
// re-reading the rows.  Here the WHOLE weight matrix sits in LDS for the life of the workgroup (K x N <= 6 blocks of
// 32 x 32), and every WAVE walks 32-row tiles on its own — no barrier in the loop:
//   * the lane (row l & 31, half l >> 5) loads its K/2 consecutive floats of the NEXT tile (K/8 16-byte loads) while the
//     current one is in the matrix pipe; with 3-4 workgroups per CU that keeps 12-16 tiles in flight per CU;
//   * the producer's BatchNorm + ReLU6 is applied in registers, coefficient vectors in LDS;
//   * K/2 x TN MFMAs (v_mfma_f32_32x32x2_f32) against B fragments read from LDS (32 consecutive floats per half-wave:
//     conflict-free);
//   * epilogue per 32-column block: bias, BatchNorm partial sums (per lane: its column), then the block goes through a
//     4 KB per-wave LDS tile and leaves as 16-byte non-temporal stores — 4 instead of 16 store instructions per lane and
//     block, 128 contiguous bytes per row and instruction (the epilogue of an HBM-bound kernel is store-ISSUE-bound).
// Tile -> wave assignment is static (tile t belongs to wave t mod 4G): the statistic partial rows are deterministic.
// KQ = K / 8 (16-byte loads per lane and tile), TN = ceil(N / 32).
template <int KQ, int TN, int OC>
__global__ __launch_bounds__(256, OC) void pw_fwd_ws_kernel(GemmArgs P) {
  constexpr int KH = 4 * KQ, K = 8 * KQ, NP = 32 * TN;
  __shared__ float Ws[K * NP];        // W[k][n], zero beyond N
  __shared__ float cf[2 * K];         // scale | shift of the input transform
  __shared__ float bs[NP];            // bias
  __shared__ float Cs[4 * 1024];      // per wave: one 32x32 block on its way out; at the end: the statistic fold
  // gfx950 only: <4,6,3> keeps 41 KB per workgroup at 3 workgroups per CU (160 KB LDS); a 64 KB-LDS part would get one
  static_assert(sizeof(float) * (K * NP + 2 * K + NP + 4 * 1024) * OC <= 160 * 1024, "weight-stationary kernel: OC workgroups do not fit a gfx950 CU's LDS");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  const bool xform = (P.ka != nullptr);

  for (int i = tid; i < K * NP; i += 256) {
    const int k = i / NP, n = i % NP;
    Ws[i] = n < P.N ? P.b[(size_t)k * P.ldb + n] : 0.f;
  }
  for (int i = tid; i < K; i += 256) {
    cf[i] = xform ? P.ka[i] : 1.f;
    cf[K + i] = xform ? P.kc[i] : 0.f;
  }
  for (int i = tid; i < NP; i += 256) bs[i] = (P.bias && i < P.N) ? P.bias[i] : 0.f;
  float st1[TN], st2[TN];
#pragma unroll
  for (int j = 0; j < TN; j++) st1[j] = st2[j] = 0.f;
  __syncthreads();

  const int nfull = P.M >> 5;  // whole 32-row tiles: the loop below; a ragged last tile is handled behind it
  const int gw = blockIdx.x * 4 + wave, GW = gridDim.x * 4;
  float *const cw = Cs + wave * 1024;
  const float *const wfrag = Ws + KH * lhi * NP + l31;
  const float *const cfs = cf + KH * lhi, *const cft = cf + K + KH * lhi;
  // 16-byte stores: lane -> (row r0 + 8 p, columns c4 .. c4 + 3) of a 32x32 block.  In the last block only N - 32 (TN - 1)
  // columns exist: the lanes beyond them repeat a valid column group (same address, same data) instead of being
  // predicated off, so that EVERY lane issues every store: with a fixed number of stores per tile the wait for the next
  // tile's rows is a counted s_waitcnt vmcnt(stores) — a branch around a store makes it vmcnt(0), i.e. every wave
  // drains its own stores (microseconds under load) once per tile.
  const int c4 = (lane & 7) * 4, r0 = lane >> 3;
  const int c4l = c4 % (P.N - 32 * (TN - 1));

  f32x4 nx[KQ];
  auto load_tile = [&](int t) {
    const int row = min(t * 32 + l31, P.M - 1);
    const float *p = P.a + (size_t)row * P.lda + KH * lhi;
#pragma unroll
    for (int j = 0; j < KQ; j++) nx[j] = ld4(p + 4 * j);
  };
  // MFMAs of the tile whose rows are in nx; then, the registers being free again, ALL K/8 requests of tile `next` (>= 0) —
  // in front of this tile's stores, so that the wait for them at the next tile's start is a counted one.  (Requested a
  // whole tile ahead through a second register set — measured: no faster for K <= 32, spills from K = 96; left alone the
  // scheduler sinks the requests between the MFMAs one by one to save registers — twelve exposed round trips per tile.)
  auto mfma_tile = [&](f32x16 (&acc)[TN], int next) {
#pragma unroll
    for (int q = 0; q < KQ; q++) {
      const f32x4 v = dl3_act4(ld4(cfs + 4 * q) * nx[q] + ld4(cft + 4 * q), P.a_act);
#pragma unroll
      for (int e = 0; e < 4; e++)
#pragma unroll
        for (int j = 0; j < TN; j++)
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[e], wfrag[(4 * q + e) * NP + j * 32], acc[j], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (next >= 0) load_tile(next);
    __builtin_amdgcn_sched_barrier(0);
  };

  if (gw < nfull) load_tile(gw);
  // (the first tile's rows are waited for HERE: entering the loop with requests in flight and no stores behind them, the
  // compiler has to merge that state with the back edge's — requests followed by 4 TN stores — and settles for vmcnt(0))
  __builtin_amdgcn_s_waitcnt(0x0F70);
  for (int t = gw; t < nfull; t += GW) {
    f32x16 acc[TN];
#pragma unroll
    for (int j = 0; j < TN; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[j][r] = 0.f;
    mfma_tile(acc, t + GW < nfull ? t + GW : -1);
    float *const cp = P.c + (size_t)(t * 32 + r0) * P.ldc;
#pragma unroll
    for (int j = 0; j < TN; j++) {
      const float bj = bs[j * 32 + l31];
      // BatchNorm partial sums of this lane's column; C layout (row (r & 3) + 8 (r >> 2) + 4 lhi, column l31) -> rows of
      // 32 floats in LDS -> 16 bytes per lane
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const float v = acc[j][r] + bj;
        st1[j] += v;
        st2[j] += v * v;
        cw[((r & 3) + 8 * (r >> 2) + 4 * lhi) * 32 + l31] = v;
      }
      __builtin_amdgcn_wave_barrier();
      const int cc = (j == TN - 1) ? c4l : c4;
      f32x4 o[4];
#pragma unroll
      for (int p = 0; p < 4; p++) o[p] = ld4(cw + (r0 + 8 * p) * 32 + cc);
#pragma unroll
      for (int p = 0; p < 4; p++) st4_nt(cp + (size_t)(8 * p) * P.ldc + j * 32 + cc, o[p]);
      __builtin_amdgcn_wave_barrier();
    }
  }
  if ((P.M & 31) != 0 && gw == nfull % GW) {
    // the one ragged tile of the launch: every element predicated
    f32x16 acc[TN];
#pragma unroll
    for (int j = 0; j < TN; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[j][r] = 0.f;
    load_tile(nfull);
    mfma_tile(acc, -1);
    const int m0 = nfull * 32;
#pragma unroll
    for (int j = 0; j < TN; j++) {
      const int col = j * 32 + l31;
      const float bj = bs[col];
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int row = m0 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        const float v = acc[j][r] + bj;
        if (row < P.M && col < P.N) {
          st1[j] += v;
          st2[j] += v * v;
          P.c[(size_t)row * P.ldc + col] = v;
        }
      }
    }
  }

  if (P.stat_mode != 0) {
    // half-waves, then the four waves in wave order: one partial row per workgroup
    __syncthreads();
    float *sred = Cs;  // [4][NP][2] (NP <= 192: 6 KB)
#pragma unroll
    for (int j = 0; j < TN; j++) {
      const float a1 = st1[j] + __shfl_xor(st1[j], 32, 64), a2 = st2[j] + __shfl_xor(st2[j], 32, 64);
      if (lhi == 0) {
        sred[(wave * NP + j * 32 + l31) * 2 + 0] = a1;
        sred[(wave * NP + j * 32 + l31) * 2 + 1] = a2;
      }
    }
    __syncthreads();
    for (int cl = tid; cl < NP; cl += 256) {
      if (cl < P.N) {
        float a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int w = 0; w < 4; w++) {
          a1 += sred[(w * NP + cl) * 2 + 0];
          a2 += sred[(w * NP + cl) * 2 + 1];
        }
        P.part[((size_t)blockIdx.x * P.N + cl) * 2 + 0] = a1;
        P.part[((size_t)blockIdx.x * P.N + cl) * 2 + 1] = a2;
        for (int r = blockIdx.x + (int)gridDim.x; r < P.part_rows; r += (int)gridDim.x) {
          P.part[((size_t)r * P.N + cl) * 2 + 0] = 0.f;
          P.part[((size_t)r * P.N + cl) * 2 + 1] = 0.f;
        }
      }
    }
  }
}

// ---- round 6: weight-stationary GEMM for the MFMA-bound short reductions ------------------------------------------------
// The expand convolutions forward / the project convolutions' bwd-data (deeplabv3p.py:175-198: reduction 64 / 96 / 160 into a
// 384 / 576 / 960-wide output at 64 x 64) spend a third of every row tile outside the K loop in the tiled stream kernel: ten
// K-tiles of a reduction of 160 are 25 us of MFMAs between a prologue (first operand round trip, two barriers) and an
// epilogue of 6-8 us during which the wave issues none (DESIGN.md, round 3 phase table: matrix pipe 71 % / 50 % busy).
// Here a workgroup of NW = 8 waves keeps its WHOLE weight slice W[K][32 TN] in LDS for its lifetime (102 KB at K = 160,
// TN = 5: one workgroup per CU, two waves per SIMD) and every wave walks 32-row tiles on its own, like pw_fwd_ws_kernel: all
// K/8 16-byte requests of a tile up front (the lane's half of its row: K/2 registers), K/2 x TN MFMAs against fragments read
// straight from the resident slice — no weight staging, no barrier anywhere in the loop — then the epilogue, block by block
// through a per-wave LDS square (16-byte stores of whole rows).  The two waves of a SIMD drift apart by themselves: while one
// is in its epilogue or waits for its rows, the other one owns the matrix pipe.
// BWD: the bwd-data instantiation (dX = dY . W^T with the single-tensor dY the weight-gradient launch materialised): no operand
// transform; the epilogue multiplies by the activation mask of the forward input x and reduces the BatchNorm-backward sums
// sum(v), sum(v * x_hat).  Both need x element by element — it is requested 16 bytes at a time in the SAME row-piece layout the
// output leaves in (block j + 1's pieces while block j is finished), and the sums are kept per lane for its four columns,
// folded over the eight row lanes once at the end.
// TWO (BWD only): the two-tensor gradient operand dY = cA*g + cB*y + cC assembled on load — the expand convolutions 64 -> 384,
// whose weight-gradient launch writes no dY (a narrow x against a wide dY: the store does not pay, round 4): reduction 384
// into a 64-wide gradient, HBM-bound, 98 KB of W^T resident.  ADD (BWD only): a residual gradient joins in the epilogue
// (requested in the same row-piece layout as the forward input).
// FLAT (forward only, one column block): the logits layer (deeplabv3p.py:438) — N = classes <= 32 columns, rows of N floats
// back to back (ldc == N, N any number): a 32-row tile of the output is ONE contiguous run of 128 N bytes, which starts on a
// 16-byte boundary whatever N is.  The block goes through the wave's LDS square packed N floats per row and leaves as 8 N
// 16-byte pieces — the tiled kernel's scalar stores into 84-byte rows were what held this HBM-bound launch at 0.40 of peak.
#ifndef DL3_WS2_XPRE_TN
#define DL3_WS2_XPRE_TN 4
#endif
template <int KQ, int TN, int NW, bool BWD = false, bool TWO = false, bool ADD = false, bool FLAT = false>
__global__ __launch_bounds__(64 * NW) void pw_ws2_kernel(GemmArgs P) {
  static_assert(BWD || (!TWO && !ADD), "two-tensor operand / residual addend: bwd-data only");
  static_assert(!FLAT || (!BWD && TN == 1), "packed narrow output: forward, one column block");
  constexpr bool XPRE = BWD && !TWO && TN <= DL3_WS2_XPRE_TN;   // the epilogue's forward-input pieces requested two column blocks ahead (below)
  constexpr int KH = 4 * KQ, K = 8 * KQ, NP = 32 * TN;
  __shared__ __attribute__((aligned(16))) float Ws[K * NP];   // W[k][n0 + n], zero beyond N
  __shared__ __attribute__((aligned(16))) float cf[(TWO ? 3 : 2) * K];    // scale | shift of the input transform (TWO: cA | cC | cB)
  __shared__ __attribute__((aligned(16))) float bs[BWD ? 4 * NP : NP];   // bias | BWD: mask scale, mask shift, mean, 1 / sigma of x
  __shared__ __attribute__((aligned(16))) float Cs[NW * 1024];  // per wave: one 32x32 block on its way out; at the end: the statistic fold
  static_assert(sizeof(float) * (K * NP + 3 * K + 4 * NP + NW * 1024) <= 160 * 1024, "gfx950: 160 KB of LDS per CU");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  // XCD-aware decode: the column tiles of one row group walk the same rows at the same time — one XCD, one L2
  const int nwg = gridDim.x * gridDim.y, b = blockIdx.y * gridDim.x + blockIdx.x;
  const int xq = nwg >> 3, xr = nwg & 7, xcd = b & 7;
  const int lid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (b >> 3);
  const int ct = lid % gridDim.x, rg = lid / gridDim.x, nrg = gridDim.y;
  const int n0 = ct * NP;
  const bool xform = (P.ka != nullptr);

  for (int i = tid; i < K * NP; i += 64 * NW) {
    const int k = i / NP, n = n0 + i % NP;
    Ws[i] = n < P.N ? P.b[(size_t)k * P.ldb + n] : 0.f;
  }
  for (int i = tid; i < K; i += 64 * NW) {
    cf[i] = xform ? P.ka[i] : 1.f;
    cf[K + i] = xform ? P.kc[i] : 0.f;
    if constexpr (TWO) cf[2 * K + i] = P.kb[i];
  }
  for (int i = tid; i < NP; i += 64 * NW) {
    const int col = min(n0 + i, P.N - 1);
    if constexpr (BWD) {
      bs[i] = P.ep_scale ? P.ep_scale[col] : 1.f;
      bs[NP + i] = P.ep_scale ? P.ep_shift[col] : 0.f;
      bs[2 * NP + i] = P.stat_mode == 2 ? P.ep_mean[col] : 0.f;
      bs[3 * NP + i] = P.stat_mode == 2 ? P.ep_invstd[col] : 0.f;
    } else {
      bs[i] = (P.bias && n0 + i < P.N) ? P.bias[n0 + i] : 0.f;
    }
  }
  float st1[TN], st2[TN];
#pragma unroll
  for (int j = 0; j < TN; j++) st1[j] = st2[j] = 0.f;
  f32x4 q1[BWD ? TN : 1], q2[BWD ? TN : 1];   // BWD: the lane's sums for the four columns of its pieces, per column block
  if constexpr (BWD) {
#pragma unroll
    for (int j = 0; j < TN; j++) q1[j] = q2[j] = splat4(0.f);
  }
  __syncthreads();

  const int nfull = P.M >> 5;
  const int gw = rg * NW + wave, GW = nrg * NW;
  float *const cw = Cs + wave * 1024;
  const float *const wfrag = Ws + KH * lhi * NP + l31;
  const float *const cfs = cf + KH * lhi, *const cft = cf + K + KH * lhi;
  const int c4 = (lane & 7) * 4, r0 = lane >> 3;
  const int ncol = min(NP, P.N - n0);                   // columns of this tile inside the matrix (a multiple of 4; FLAT: any)
  const int jl = (ncol - 1) >> 5;                        // its last column block
  const int c4l = FLAT ? 0 : c4 % (ncol - 32 * jl);      // lanes beyond the last block's columns repeat a valid column group

  // The lane's K/2 operand values arrive as a stream of NCH chunks per tile through a ring of two register slots: while the
  // MFMAs of chunk c run, chunk c + 1 is in flight and chunk c + 2 — of this tile or, behind its last two chunks, of the wave's
  // NEXT tile — is requested as soon as slot c & 1 has been read.  The stream never drains at a tile boundary: the next tile's
  // first two chunks are requested before this tile's epilogue (and in front of its stores: counted waits).
  // chunks per tile (even: the slot of a chunk is c & 1 in every tile).  BWD: two pieces per chunk — its epilogue needs the
  // registers (the forward-input pieces of two column blocks, the per-column sums)
  constexpr int NCH = BWD ? KQ / 2 : 4;
  static_assert(NCH % 2 == 0, "an even number of chunks per tile");
  constexpr int SQ = KQ / NCH;                       // 16-byte pieces per chunk
  static_assert(KQ % NCH == 0, "the reduction must split into an even number of equal chunks");
  f32x4 nx[2][SQ], ny[2][TWO ? SQ : 1];
  auto req = [&](int t, int c, int sl) __attribute__((always_inline)) {
    const int row = min(t * 32 + l31, P.M - 1);
    const float *p = P.a + (size_t)row * P.lda + KH * lhi + 4 * SQ * c;
#pragma unroll
    for (int j = 0; j < SQ; j++) nx[sl][j] = ld4(p + 4 * j);
    if constexpr (TWO) {
      const float *p2 = P.a2 + (size_t)row * P.lda2 + KH * lhi + 4 * SQ * c;
#pragma unroll
      for (int j = 0; j < SQ; j++) ny[sl][j] = ld4(p2 + 4 * j);
    }
  };
  // tile t: its chunks 0 and 1 are in flight (or landed) on entry; tn = the wave's next tile (itself again at the end:
  // harmless duplicate requests instead of a branch around requests).  The B fragments of k-step s + 1 are read from the
  // resident slice while the TN MFMAs of k-step s run (two register sets, the order pinned with sched_group_barrier: left
  // alone the scheduler sinks every ds_read next to its MFMA — read, s_waitcnt lgkmcnt(0), two MFMAs — and with eight
  // waves on the LDS that round trip is longer than the two MFMAs)
  DL3_T(long long tp0 = 0; long long tp1 = 0; long long tp2 = 0; int ntl = 0;)
  auto mfma_tile = [&](f32x16 (&acc)[TN], int t, int tn) __attribute__((always_inline)) {
    float bf[2][TN];
#pragma unroll
    for (int j = 0; j < TN; j++) bf[0][j] = wfrag[j * 32];
    DL3_T(const long long w1 = clock64();)
#pragma unroll
    for (int c = 0; c < NCH; c++) {
#pragma unroll
      for (int q = 0; q < SQ; q++) {
        const int kq = SQ * c + q;
        f32x4 v;
        if constexpr (TWO) {
          v = ld4(cfs + 4 * kq) * nx[c & 1][q] + ld4(cft + 4 * kq);
          v += ld4(cf + 2 * K + KH * lhi + 4 * kq) * ny[c & 1][q];
        } else if constexpr (BWD) {
          v = nx[c & 1][q];
        } else {
          v = dl3_act4(ld4(cfs + 4 * kq) * nx[c & 1][q] + ld4(cft + 4 * kq), P.a_act);
        }
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const int ks = 4 * kq + e, cur = ks & 1;
          if (ks + 1 < KH) {
#pragma unroll
            for (int j = 0; j < TN; j++) bf[cur ^ 1][j] = wfrag[(ks + 1) * NP + j * 32];
          }
#pragma unroll
          for (int j = 0; j < TN; j++) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[e], bf[cur][j], acc[j], 0, 0, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, TN, 0);   // TN LDS reads (k-step s + 1) ...
          __builtin_amdgcn_sched_group_barrier(0x008, TN, 0);   // ... then the TN MFMAs of k-step s
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (c + 2 < NCH) req(t, c + 2, c & 1);
      else req(tn, c + 2 - NCH, c & 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    DL3_T(tp1 += clock64() - w1;)
  };

  if (gw < nfull) {
    req(gw, 0, 0);
    req(gw, 1, 1);
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);
  for (int t = gw; t < nfull; t += GW) {
    f32x16 acc[TN];
#pragma unroll
    for (int j = 0; j < TN; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[j][r] = 0.f;
    // BWD, XPRE: the forward-input (and addend) pieces of the first TWO column blocks are requested in front of the tile's MFMAs
    // (they retire before the operand ring's requests: in-order vmcnt) and block j + 2's as soon as block j has used its slot —
    // the epilogue runs two blocks ahead of its HBM round trip instead of one, which a block's own ~1 us of work did not cover
    // (same-call A/B, 524 288 rows: 576 <- 96 0.737 -> 0.703 ms, 384 <- 96 0.495 -> 0.471, 384 <- 64 0.392 -> 0.383).  Not at
    // TN = 5 and not for the two-tensor operand: 32 more live registers across the MFMA loop spill there (144 -> 236 B of scratch,
    // 960 <- 160 1.93 -> 1.99 ms; 64 <- 384 0.505 -> 0.516)
    f32x4 xq[BWD ? 2 : 1][BWD ? 4 : 1], aq[BWD ? 2 : 1][ADD ? 4 : 1];
    if constexpr (XPRE) {
      int le = lane, c4le = c4l;   // (laundered like the epilogue's: see there)
      asm volatile("" : "+v"(le), "+v"(c4le));
      const int c4 = (le & 7) * 4, r0 = le >> 3, c4l = c4le;
      const float *const xp0 = P.ep_x + (size_t)(t * 32 + r0) * P.ld_epx + n0;
      const float *const ap0 = ADD ? P.ep_add + (size_t)(t * 32 + r0) * P.ld_add + n0 : nullptr;
#pragma unroll
      for (int jj = 0; jj < (TN < 2 ? TN : 2); jj++) {
        // (a block beyond the tile's last one: block jl's columns again — a valid address, never used)
        const int cc_ = jj < jl ? jj * 32 + c4 : jl * 32 + c4l;
#pragma unroll
        for (int p = 0; p < 4; p++) {
          xq[jj][p] = ld4(xp0 + (size_t)(8 * p) * P.ld_epx + cc_);
          if constexpr (ADD) aq[jj][p] = ld4(ap0 + (size_t)(8 * p) * P.ld_add + cc_);
        }
      }
    }
    mfma_tile(acc, t, t + GW < nfull ? t + GW : t);
    DL3_T(const long long e0 = clock64(); ntl++;)
#ifdef DL3_WS2_DIRECT_EPILOGUE   // (measured: 160 -> 960 at 524 288 rows 1.51 -> 1.59 ms; kept for the record)
    // straight from the accumulator layout: register r of lane l is row (r & 3) + 8 (r >> 2) + 4 (l >> 5), column l & 31 — a
    // store instruction writes two 128-byte row segments.  (The HBM-bound sibling pw_fwd_ws_kernel takes the block through
    // LDS for 16-byte stores of whole rows — four times fewer store instructions, but an LDS round trip per block in a
    // phase during which the wave issues no MFMA; this kernel is matrix-bound, its epilogue wants to be SHORT.)
    float *const cp = P.c + (size_t)(t * 32) * P.ldc + n0;        // (wave-uniform)
    const unsigned lo = (unsigned)(4 * lhi * P.ldc + l31);
#pragma unroll
    for (int j = 0; j < TN; j++) {
      if (j < jl || (j == jl && j * 32 + l31 < ncol)) {           // (full tiles: uniformly true)
        const float bj = bs[j * 32 + l31];
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const float v = acc[j][r] + bj;
          st1[j] += v;
          st2[j] += v * v;
          __builtin_nontemporal_store(v, &(cp + (size_t)((r & 3) + 8 * (r >> 2)) * P.ldc + j * 32)[lo]);
        }
      }
    }
#else
    if constexpr (BWD) {
      // The epilogue's lane-derived addresses depend only on the lane: left alone, the scheduler computes them once in front of
      // the tile loop and — at TN = 5, 256 registers — parks 25 of them in scratch, reloaded here tile after tile.  Laundering
      // the lane id through an empty asm behind the last MFMA makes them a handful of VALU instructions per tile instead.
      int le = lane, c4le = c4l;
      asm volatile("" : "+v"(le), "+v"(c4le));
      const int c4 = (le & 7) * 4, r0 = le >> 3, l31 = le & 31, lhi = le >> 5, c4l = c4le;
      float *const cp = P.c + (size_t)(t * 32 + r0) * P.ldc + n0;
      const float *const xp = P.ep_x + (size_t)(t * 32 + r0) * P.ld_epx + n0;
      const float *const ap = ADD ? P.ep_add + (size_t)(t * 32 + r0) * P.ld_add + n0 : nullptr;
      if constexpr (!XPRE) {
#pragma unroll
        for (int p = 0; p < 4; p++) {
          xq[0][p] = ld4(xp + (size_t)(8 * p) * P.ld_epx + (0 == jl ? c4l : c4));
          if constexpr (ADD) aq[0][p] = ld4(ap + (size_t)(8 * p) * P.ld_add + (0 == jl ? c4l : c4));
        }
      }
#pragma unroll
      for (int j = 0; j < TN; j++) {
        if (j <= jl) {
          if (!XPRE && j + 1 <= jl) {
#pragma unroll
            for (int p = 0; p < 4; p++) {
              xq[(j + 1) & 1][p] = ld4(xp + (size_t)(8 * p) * P.ld_epx + (j + 1) * 32 + (j + 1 == jl ? c4l : c4));
              if constexpr (ADD) aq[(j + 1) & 1][p] = ld4(ap + (size_t)(8 * p) * P.ld_add + (j + 1) * 32 + (j + 1 == jl ? c4l : c4));
            }
          }
#pragma unroll
          for (int r = 0; r < 16; r++) cw[((r & 3) + 8 * (r >> 2) + 4 * lhi) * 32 + l31] = acc[j][r];
          __builtin_amdgcn_wave_barrier();
          const int cc = (j == jl) ? c4l : c4;
          const f32x4 es = ld4(bs + j * 32 + cc), et = ld4(bs + NP + j * 32 + cc);
          const f32x4 mu = ld4(bs + 2 * NP + j * 32 + cc), is = ld4(bs + 3 * NP + j * 32 + cc);
          // (a lane that repeats a column group — cc != c4 in the last block — stores duplicates but must not count them)
          const float own = (cc == c4) ? 1.f : 0.f;
#pragma unroll
          for (int p = 0; p < 4; p++) {
            const f32x4 x = xq[j & 1][p];
            f32x4 o = ld4(cw + (r0 + 8 * p) * 32 + cc) * dl3_mask4(es * x + et, P.ep_act);
            if constexpr (ADD) o += P.add_scale * aq[j & 1][p];
            st4_nt(cp + (size_t)(8 * p) * P.ldc + j * 32 + cc, o);
            q1[j] += own * o;
            q2[j] += own * (o * ((x - mu) * is));
          }
          if (XPRE && j + 2 <= jl) {   // (slot j & 1 is free again: block j + 2)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int p = 0; p < 4; p++) {
              xq[j & 1][p] = ld4(xp + (size_t)(8 * p) * P.ld_epx + (j + 2) * 32 + (j + 2 == jl ? c4l : c4));
              if constexpr (ADD) aq[j & 1][p] = ld4(ap + (size_t)(8 * p) * P.ld_add + (j + 2) * 32 + (j + 2 == jl ? c4l : c4));
            }
          }
          __builtin_amdgcn_wave_barrier();
        }
      }
    } else if constexpr (FLAT) {
      const float bj = bs[l31];
      float *const cq = cw + 4 * lhi * P.N + l31;
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const float v = acc[0][r] + bj;
        st1[0] += v;
        st2[0] += v * v;
        if (l31 < P.N) cq[((r & 3) + 8 * (r >> 2)) * P.N] = v;
      }
      __builtin_amdgcn_wave_barrier();
      float *const cp = P.c + (size_t)t * 32 * P.N;
#pragma unroll
      for (int p = 0; p < 4; p++) {
        const int piece = lane + 64 * p;
        if (piece < 8 * P.N) st4_nt(cp + 4 * piece, ld4(cw + 4 * piece));
      }
      __builtin_amdgcn_wave_barrier();
    } else {
    float *const cp = P.c + (size_t)(t * 32 + r0) * P.ldc + n0;
#pragma unroll
    for (int j = 0; j < TN; j++) {
      if (j <= jl) {   // (workgroup-uniform: column blocks wholly beyond N — only in the last column tile — are skipped)
        const float bj = bs[j * 32 + l31];
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const float v = acc[j][r] + bj;
          st1[j] += v;
          st2[j] += v * v;
          cw[((r & 3) + 8 * (r >> 2) + 4 * lhi) * 32 + l31] = v;
        }
        __builtin_amdgcn_wave_barrier();
        const int cc = (j == jl) ? c4l : c4;
        f32x4 o[4];
#pragma unroll
        for (int p = 0; p < 4; p++) o[p] = ld4(cw + (r0 + 8 * p) * 32 + cc);
#pragma unroll
        for (int p = 0; p < 4; p++) st4_nt(cp + (size_t)(8 * p) * P.ldc + j * 32 + cc, o[p]);
        __builtin_amdgcn_wave_barrier();
      }
    }
    }
#endif
    DL3_T(tp2 += clock64() - e0;)
  }
#ifdef DL3_PHASE_TIMING
  if (P.dbg && lane == 0) {
    long long *d = P.dbg + ((size_t)b * NW + wave) * 8;
    d[0] = tp0; d[1] = tp1; d[2] = tp2; d[3] = ntl; d[4] = 0; d[5] = 0;
  }
#endif
  if ((P.M & 31) != 0 && gw == nfull % GW) {
    f32x16 acc[TN];
#pragma unroll
    for (int j = 0; j < TN; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[j][r] = 0.f;
    __builtin_amdgcn_s_waitcnt(0x0F70);
    req(nfull, 0, 0);
    req(nfull, 1, 1);
    mfma_tile(acc, nfull, nfull);
    const int m0 = nfull * 32;
#pragma unroll
    for (int j = 0; j < TN; j++) {
      const int col = n0 + j * 32 + l31;
      const float bj = BWD ? 0.f : bs[j * 32 + l31];
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int row = m0 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        float v = acc[j][r] + bj;
        if (row < P.M && col < P.N) {
          if constexpr (BWD) {
            const float x = P.ep_x[(size_t)row * P.ld_epx + col];
            v *= dl3_act_mask(bs[j * 32 + l31] * x + bs[NP + j * 32 + l31], P.ep_act);
            if constexpr (ADD) v += P.add_scale * P.ep_add[(size_t)row * P.ld_add + col];
            st1[j] += v;
            st2[j] += v * ((x - bs[2 * NP + j * 32 + l31]) * bs[3 * NP + j * 32 + l31]);
          } else {
            st1[j] += v;
            st2[j] += v * v;
          }
          P.c[(size_t)row * P.ldc + col] = v;
        }
      }
    }
  }

  if (P.stat_mode != 0) {
    // half-waves, then the NW waves in wave order: one partial row per row group
    __syncthreads();
    float *sred = Cs;  // [NW][NP][2]
    if constexpr (BWD) {
      // the lane's sums for its four columns: over the eight row lanes (lane bits 3-5), then lanes 0-7 own 4 columns each
#pragma unroll
      for (int j = 0; j < TN; j++) {
#pragma unroll
        for (int o = 8; o <= 32; o <<= 1) {
#pragma unroll
          for (int e = 0; e < 4; e++) {
            q1[j][e] += __shfl_xor(q1[j][e], o, 64);
            q2[j][e] += __shfl_xor(q2[j][e], o, 64);
          }
        }
        if (lane < 8) {
#pragma unroll
          for (int e = 0; e < 4; e++) {
            sred[(wave * NP + j * 32 + c4 + e) * 2 + 0] = q1[j][e];
            sred[(wave * NP + j * 32 + c4 + e) * 2 + 1] = q2[j][e];
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int j = 0; j < TN; j++) {
      const float a1 = st1[j] + __shfl_xor(st1[j], 32, 64), a2 = st2[j] + __shfl_xor(st2[j], 32, 64);
      if (lhi == 0) {
        // (BWD: st1 / st2 only hold the ragged last tile's share, on top of the row-piece sums written above)
        sred[(wave * NP + j * 32 + l31) * 2 + 0] = (BWD ? sred[(wave * NP + j * 32 + l31) * 2 + 0] : 0.f) + a1;
        sred[(wave * NP + j * 32 + l31) * 2 + 1] = (BWD ? sred[(wave * NP + j * 32 + l31) * 2 + 1] : 0.f) + a2;
      }
    }
    __syncthreads();
    for (int cl = tid; cl < NP; cl += 64 * NW) {
      const int col = n0 + cl;
      if (col < P.N) {
        float a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int w = 0; w < NW; w++) {
          a1 += sred[(w * NP + cl) * 2 + 0];
          a2 += sred[(w * NP + cl) * 2 + 1];
        }
        P.part[((size_t)rg * P.N + col) * 2 + 0] = a1;
        P.part[((size_t)rg * P.N + col) * 2 + 1] = a2;
        for (int r = rg + nrg; r < P.part_rows; r += nrg) {
          P.part[((size_t)r * P.N + col) * 2 + 0] = 0.f;
          P.part[((size_t)r * P.N + col) * 2 + 1] = 0.f;
        }
      }
    }
  }
}

// ---- round 6: the logits layer's backward (deeplabv3p.py:438: Conv2D(classes, (1, 1)) on the 256-wide decoder output) --------
// Both are HBM-bound streams whose narrow side is `classes` = 21 floats per row — 84-byte rows that no 16-byte access lines up
// with, which is why the tiled kernels ran them at 0.35 / 0.21 of the HBM peak (scalar loads of the narrow operand, 16-row
// stages with eight MFMAs per barrier).  The narrow operand is handled as what it is in memory: one contiguous run.
//
// bwd-data: dX[M, 32 TN] = dY[M, K] . W^T[K, 32 TN] with K = classes <= 32 and dY rows back to back (lda == K): a wave takes the
// 32 K floats of its row tile as 8 K 16-byte pieces of one contiguous, 16-byte-aligned run (three loads per lane at K = 21,
// requested one tile ahead), parks them in a wave-private LDS strip and reads its MFMA A-fragments from there (row l31, k =
// lhi ceil(K/2) + s: stride K, conflict-free for odd K); W^T sits in LDS for the workgroup's lifetime, zero rows up to 2 ceil(K/2).
// No transform, mask, addend or BatchNorm sums: the layer's input is the Dropout output (a materialised buffer).
template <int TN, int NW>
__global__ __launch_bounds__(64 * NW) void pw_narrowk_kernel(GemmArgs P) {
  constexpr int NP = 32 * TN;
  __shared__ __attribute__((aligned(16))) float Bs[32 * NP];     // W^T[k][n], zero for k >= K
  __shared__ __attribute__((aligned(16))) float As[NW * 1024];   // per wave: its tile of dY, 32 rows of K floats, packed
  __shared__ __attribute__((aligned(16))) float Cs[NW * 1024];   // per wave: one 32x32 block on its way out
  static_assert(sizeof(float) * (32 * NP + 2 * NW * 1024) <= 160 * 1024, "gfx950: 160 KB of LDS per CU");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  const int n0 = blockIdx.y * NP;
  for (int i = tid; i < 32 * NP; i += 64 * NW) {
    const int k = i / NP, n = n0 + i % NP;
    Bs[i] = (k < P.K && n < P.N) ? P.b[(size_t)k * P.ldb + n] : 0.f;
  }
  __syncthreads();
  const int KS = (P.K + 1) >> 1;                 // k-steps: lane half lhi owns k = lhi KS .. lhi KS + KS - 1
  const int npc = 8 * P.K;                       // 16-byte pieces per row tile (<= 256)
  const int nfull = P.M >> 5;
  const int gw = blockIdx.x * NW + wave, GW = gridDim.x * NW;
  float *const as = As + wave * 1024, *const cw = Cs + wave * 1024;
  const int c4 = (lane & 7) * 4, r0 = lane >> 3;
  f32x4 f[4];
  auto req = [&](int t) __attribute__((always_inline)) {
    const float *p = P.a + (size_t)t * 32 * P.K;
#pragma unroll
    for (int q = 0; q < 4; q++)
      if (lane + 64 * q < npc) f[q] = ld4(p + 4 * (lane + 64 * q));
  };
  auto products = [&](f32x16 (&acc)[TN]) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < TN; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[j][r] = 0.f;
    const float *ap = as + l31 * P.K + lhi * KS;
    const float *bp = Bs + lhi * KS * NP + l31;
    for (int s = 0; s < KS; s++) {
      const float a = (lhi * KS + s < P.K) ? ap[s] : 0.f;
#pragma unroll
      for (int j = 0; j < TN; j++) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bp[s * NP + j * 32], acc[j], 0, 0, 0);
    }
  };
  if (gw < nfull) req(gw);
  for (int t = gw; t < nfull; t += GW) {
#pragma unroll
    for (int q = 0; q < 4; q++)
      if (lane + 64 * q < npc) st4(as + 4 * (lane + 64 * q), f[q]);
    __builtin_amdgcn_wave_barrier();
    req(t + GW < nfull ? t + GW : t);
    f32x16 acc[TN];
    products(acc);
    float *const cp = P.c + (size_t)(t * 32 + r0) * P.ldc + n0 + c4;
#pragma unroll
    for (int j = 0; j < TN; j++) {
      if (n0 + j * 32 < P.N) {      // (the output width is a multiple of 32: whole blocks)
#pragma unroll
        for (int r = 0; r < 16; r++) cw[((r & 3) + 8 * (r >> 2) + 4 * lhi) * 32 + l31] = acc[j][r];
        __builtin_amdgcn_wave_barrier();
        f32x4 o[4];
#pragma unroll
        for (int p = 0; p < 4; p++) o[p] = ld4(cw + (r0 + 8 * p) * 32 + c4);
#pragma unroll
        for (int p = 0; p < 4; p++) st4_nt(cp + (size_t)(8 * p) * P.ldc + j * 32, o[p]);
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
  if ((P.M & 31) != 0 && gw == nfull % GW) {   // the ragged last tile: element by element
    const int m0 = nfull * 32;
    const size_t total = (size_t)P.M * P.K;
    for (int i = lane; i < 32 * P.K; i += 64) {
      const size_t e = (size_t)m0 * P.K + i;
      as[i] = e < total ? P.a[e] : 0.f;
    }
    __builtin_amdgcn_wave_barrier();
    f32x16 acc[TN];
    products(acc);
#pragma unroll
    for (int j = 0; j < TN; j++) {
      const int col = n0 + j * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int row = m0 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        if (row < P.M && col < P.N) P.c[(size_t)row * P.ldc + col] = acc[j][r];
      }
    }
  }
}

// bwd-weight: dW[K][N] = T(x)^T[K, M] . dY[M, N] (+ db[N] = column sums of dY) with N = classes <= 32 and K = 128 HB.  No LDS
// stage, no barrier in the loop: the reduction index of the MFMA is the pixel row, so a lane's A operand is x[row + lhi][its
// channel] — and WHICH channel an accumulator row stands for is ours to choose.  Lane (l31, lhi) requests 16 bytes at channel
// 128 h + 4 l31 of row r + lhi (a half-wave reads 512 contiguous bytes): element e of that piece is the A operand of block
// (h, e), whose accumulator row i is channel 128 h + 4 i + e.  Every 2-row k-step is HB 16-byte loads + one 4-byte load of dY
// and 4 HB MFMAs; a wave keeps all 4 HB accumulators (the whole K x 32 gradient), walks 4-row chunks chunk = gw, gw + GW, ...
// through a ring of D register slots (three chunks in flight while one is multiplied), and the NW waves of a workgroup meet
// in LDS once at the end, in wave order: one [K][N] slab (and one row of column sums) per workgroup.
template <int HB, int NW>
__global__ __launch_bounds__(64 * NW) void pw_wgrad_narrow_kernel(WgradArgs P, float *cpart) {
  constexpr int NB = 4 * HB, K = 128 * HB, D = 4;
  __shared__ __attribute__((aligned(16))) float red[NB * 16 * 64];   // the workgroup's sum, accumulator layout
  __shared__ __attribute__((aligned(16))) float cfx[2 * K];
  __shared__ float dbs[NW * 32];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  const bool xform = P.xs != nullptr;
  for (int i = tid; i < K; i += 64 * NW) {
    cfx[i] = xform ? P.xs[i] : 1.f;
    cfx[K + i] = xform ? P.xt[i] : 0.f;
  }
  __syncthreads();
  const int nchunks = (P.M + 3) >> 2;
  const int gw = blockIdx.x * NW + wave, GW = gridDim.x * NW;
  const int gcol = min(l31, P.N - 1);
  const float colmask = l31 < P.N ? 1.f : 0.f;
  f32x16 acc[NB];
#pragma unroll
  for (int b = 0; b < NB; b++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[b][r] = 0.f;
  float gsum = 0.f;
  f32x4 xr[D][2][HB];
  float gr[D][2];
  auto req = [&](int c, int sl) __attribute__((always_inline)) {
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
      const int row = min(4 * c + 2 * ks + lhi, P.M - 1);   // (rows beyond M: a valid row, masked when it is multiplied)
      const float *xp = P.x + (size_t)row * P.ldx + 4 * l31;
#pragma unroll
      for (int h = 0; h < HB; h++) xr[sl][ks][h] = ld4(xp + 128 * h);
      gr[sl][ks] = P.g[(size_t)row * P.ldg + gcol];
    }
  };
  auto multiply = [&](int c, int sl) __attribute__((always_inline)) {
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
      const float gv = (4 * c + 2 * ks + lhi < P.M) ? gr[sl][ks] * colmask : 0.f;
      gsum += gv;
#pragma unroll
      for (int h = 0; h < HB; h++) {
        f32x4 v = xr[sl][ks][h];
        if (xform) v = dl3_act4(ld4(cfx + 128 * h + 4 * l31) * v + ld4(cfx + K + 128 * h + 4 * l31), P.x_act);
#pragma unroll
        for (int e = 0; e < 4; e++) acc[4 * h + e] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[e], gv, acc[4 * h + e], 0, 0, 0);
      }
    }
  };
#pragma unroll
  for (int d = 0; d < D; d++) req(gw + d * GW, d);
  for (int c = gw; c < nchunks; c += D * GW) {
#pragma unroll
    for (int d = 0; d < D; d++) {
      multiply(c + d * GW, d);
      __builtin_amdgcn_sched_barrier(0);
      req(c + (d + D) * GW, d);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // the NW waves' accumulators, in wave order
  for (int w = 0; w < NW; w++) {
    if (wave == w) {
#pragma unroll
      for (int b = 0; b < NB; b++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int i = (b * 16 + r) * 64 + lane;
          red[i] = (w == 0 ? 0.f : red[i]) + acc[b][r];
        }
    }
    __syncthreads();
  }
  gsum += __shfl_xor(gsum, 32, 64);
  if (lhi == 0) dbs[wave * 32 + l31] = gsum;
  float *slab = P.ws + (size_t)blockIdx.x * K * P.N;
  for (int i = tid; i < K * P.N; i += 64 * NW) {
    const int ch = i / P.N, n = i - ch * P.N;
    const int h = ch >> 7, ii = (ch & 127) >> 2, e = ch & 3;
    const int r = (ii & 3) + 4 * (ii >> 3), hi = (ii >> 2) & 1;
    slab[i] = red[((4 * h + e) * 16 + r) * 64 + hi * 32 + n];
  }
  __syncthreads();
  if (cpart && tid < P.N) {
    float a = 0.f;
#pragma unroll
    for (int w = 0; w < NW; w++) a += dbs[w * 32 + tid];
    cpart[(size_t)blockIdx.x * P.N + tid] = a;
  }
}

// ---- small batches (round 5): 32 rows x all N <= 160 columns per workgroup, the REDUCTION split over the four
// waves.  At B = 2 - 4 the 64x64 maps are 8 192 - 16 384 rows: the 32-row stream kernels put their four waves side by side
// along N (64 columns each), which pads a 160-wide output to 256 columns (1.6x the MFMAs) and a 96-wide one to 128, on a
// launch that is one workgroup per CU walking K/16 barrier-separated K-tiles.  Here every wave owns whole 32 x (32 TN)
// accumulators and walks every fourth K-tile on its own — weights through a wave-private double buffer in LDS, no barrier
// in the loop —, the four partial accumulators meet in LDS once ((w0 + w1) + (w2 + w3), fixed order) and column block j is
// finished (bias, mask, addend, store, BatchNorm sums) by wave j % 4.  One partial-sum row per workgroup (= 32-row tile).
template <int TN, bool TWO>
__global__ __launch_bounds__(256) void pw_ksplit32_kernel(GemmArgs P) {
  constexpr int KT = 16, BN = 32 * TN, BQ = KT * BN;
  constexpr int NB = (BQ / 4 + 63) / 64;             // float4 weight loads per LANE: one wave loads a whole K-tile
  constexpr int KCS = DL3_STREAM_KMAX + KT;
  __shared__ float cf[(TWO ? 3 : 2) * KCS];
  __shared__ __attribute__((aligned(16))) float wbuf[4][2 * BQ];  // per wave: two weight tiles; later its TN accumulators
  static_assert(2 * BQ == 1024 * TN, "the reduction reuses the weight buffers");
  // gfx950 only: TN = 5 holds 80 KB of weight tiles + up to 24 KB of coefficients in static LDS — inside the 160 KB of a
  // CDNA4 CU, beyond the 64 KB of gfx90a / gfx942 (this library is built for gfx950 alone: csrc/Makefile)
  static_assert(sizeof(float) * ((TWO ? 3 : 2) * KCS + 4 * 2 * BQ) <= 160 * 1024, "K-split kernel: static LDS exceeds a gfx950 CU's 160 KB");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
  const int ktiles = (P.K + KT - 1) / KT;
  const bool xform = (P.ka != nullptr);
  for (int i = tid; i < ktiles * KT; i += 256) {
    const bool in = i < P.K;
    const int k = min(i, P.K - 1);
    cf[i] = in ? (xform ? P.ka[k] : 1.f) : 0.f;
    cf[KCS + i] = (in && xform) ? P.kc[k] : 0.f;
    if (TWO) cf[2 * KCS + i] = in ? P.kb[k] : 0.f;
  }
  __syncthreads();
  const int m0 = blockIdx.x * 32;
  const int row = min(m0 + l31, P.M - 1);
  const float *ar = P.a + (size_t)row * P.lda;
  const float *ar2 = TWO ? P.a2 + (size_t)row * P.lda2 : nullptr;
  float *wb = wbuf[wave];
  f32x16 acc[TN];
#pragma unroll
  for (int j = 0; j < TN; j++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[j][r] = 0.f;
  f32x4 an[2], an2[2], rb[NB];
  float ac[8];
  auto request = [&](int kt) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int kc = min(kt * KT + 8 * lhi + 4 * j, P.K - 4);
      an[j] = ld4(ar + kc);
      if (TWO) an2[j] = ld4(ar2 + kc);
    }
#pragma unroll
    for (int i = 0; i < NB; i++) {
      const int idx = lane + 64 * i;
      if (NB * 64 == BQ / 4 || idx < BQ / 4) {
        const int kk = idx / (BN / 4), nq = idx % (BN / 4);
        const int krow = min(kt * KT + kk, P.K - 1), col = min(nq * 4, P.N - 4);
        rb[i] = ld4(P.b + (size_t)krow * P.ldb + col);
      }
    }
  };
  auto stash = [&](int kt, float *Bs) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NB; i++) {
      const int idx = lane + 64 * i;
      if (NB * 64 == BQ / 4 || idx < BQ / 4) st4(&Bs[(idx / (BN / 4)) * BN + (idx % (BN / 4)) * 4], rb[i]);
    }
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int k = kt * KT + 8 * lhi + 4 * j;
      f32x4 v = ld4(cf + k) * an[j] + ld4(cf + KCS + k);
      if (TWO) v += ld4(cf + 2 * KCS + k) * an2[j];
      v = dl3_act4(v, P.a_act);
      ac[4 * j + 0] = v.x; ac[4 * j + 1] = v.y; ac[4 * j + 2] = v.z; ac[4 * j + 3] = v.w;
    }
  };
  int kt = wave, buf = 0;
  if (kt < ktiles) {
    request(kt);
    stash(kt, wb);
  }
  for (; kt < ktiles; kt += 4) {
    const bool more = kt + 4 < ktiles;
    if (more) request(kt + 4);
    __builtin_amdgcn_wave_barrier();   // (a wave's LDS operations complete in order: the tile it stashed is there)
    const float *Bs = wb + buf * BQ;
    float av[8];
#pragma unroll
    for (int s_ = 0; s_ < 8; s_++) av[s_] = ac[s_];
#pragma unroll
    for (int s_ = 0; s_ < 8; s_++) {
#pragma unroll
      for (int j = 0; j < TN; j++)
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s_], Bs[(8 * lhi + s_) * BN + j * 32 + l31], acc[j], 0, 0, 0);
    }
    if (more) stash(kt + 4, wb + (buf ^ 1) * BQ);
    buf ^= 1;
  }
  // the four partial accumulators meet: wave w parks its TN blocks in its own buffer, block j is summed by wave j % 4
  __syncthreads();
#pragma unroll
  for (int j = 0; j < TN; j++)
#pragma unroll
    for (int r = 0; r < 16; r++) wb[(j * 16 + r) * 64 + lane] = acc[j][r];
  __syncthreads();
  const bool mode2 = P.stat_mode == 2;
#pragma unroll
  for (int j = 0; j < TN; j++) {
    if ((j & 3) != wave) continue;     // (wave-uniform)
    float v16[16];
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int o = (j * 16 + r) * 64 + lane;
      v16[r] = (wbuf[0][o] + wbuf[1][o]) + (wbuf[2][o] + wbuf[3][o]);
    }
    const int col = j * 32 + l31;
    const bool cok = col < P.N;
    const int colc = min(col, P.N - 1);
    float bias = 0.f, es = 1.f, et = 0.f, mu = 0.f, is = 0.f;
    if (P.bias) bias = P.bias[colc];
    if (P.ep_x && P.ep_scale) { es = P.ep_scale[colc]; et = P.ep_shift[colc]; }
    if (mode2) { mu = P.ep_mean[colc]; is = P.ep_invstd[colc]; }
    const int rbase = m0 + 4 * lhi;
    float xr_[16], ad[16];
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int rw = min(rbase + (r & 3) + 8 * (r >> 2), P.M - 1);
      xr_[r] = P.ep_x ? P.ep_x[(size_t)rw * P.ld_epx + colc] : 0.f;
      const int arow_ = (P.add_div > 1) ? rw / P.add_div : rw;
      ad[r] = P.ep_add ? P.add_scale * P.ep_add[(size_t)arow_ * P.ld_add + colc] : 0.f;
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int rw = rbase + (r & 3) + 8 * (r >> 2);
      float v = v16[r] + bias;
      if (P.ep_x) v *= dl3_act_mask(es * xr_[r] + et, P.ep_act);
      v += ad[r];
      if (cok && rw < P.M) {
        __builtin_nontemporal_store(v, &P.c[(size_t)rw * P.ldc + col]);
        s1 += v;
        s2 += mode2 ? v * ((xr_[r] - mu) * is) : v * v;
      }
    }
    if (P.stat_mode != 0) {
      s1 += __shfl_xor(s1, 32, 64);
      s2 += __shfl_xor(s2, 32, 64);
      if (lhi == 0 && cok) {
        P.part[((size_t)blockIdx.x * P.N + col) * 2 + 0] = s1;
        P.part[((size_t)blockIdx.x * P.N + col) * 2 + 1] = s2;
        for (int rr = blockIdx.x + (int)gridDim.x; rr < P.part_rows; rr += (int)gridDim.x) {
          P.part[((size_t)rr * P.N + col) * 2 + 0] = 0.f;
          P.part[((size_t)rr * P.N + col) * 2 + 1] = 0.f;
        }
      }
    }
  }
}

// shapes and row counts the K-split kernel takes: 1 024 - 16 384 rows
inline int ksplit_tn(int M, int K, int N) {
  // (a pure function of the shape: the answer sizes the statistic partial buffers when a plan is lowered — dl3_pwconv_partials —
  // and picks the kernel at every launch; the DL3_KSPLIT / DL3_KSPLIT_ROWS knobs of round 5's A/B are gone)
  if (M > 16384 || M < 1024 || K < 192 || K > DL3_STREAM_KMAX || K % 4 != 0 || N % 4 != 0) return 0;
  const int tn = dl3_cdiv(N, 32);
  // 64-, 96- and 160-wide outputs: the ones the 64-column-per-wave kernels pad (to 128, 128 and 256 columns)
  return (tn == 2 || tn == 3 || tn == 5) ? tn : 0;
}

// ---- dl3_pwconv_fwd_rows: a handful of rows (the ASPP image-pooling branch: ONE row per image, deeplabv3p.py:375-382,
// and its share of concat_projection, :402-406): Y[m][n] = act(ka*x + kc)[m][:] . W[:][n] + bias + addend, in DOUBLE.  The
// result is a per-image constant the network adds to every pixel of the 64x64 map: its rounding error does not average
// out over pixels, and a reduction of 2 048 on the f32 MFMA was where the HIP path's distance to float64 left torch-fp32's
// (tools/r5/xception_layer_distance.py: ratio 1.00 up to the exit flow, 1.08 behind image_pooling, 1.15 at the logits).
// 16 columns x 16 slices of the reduction per workgroup (a lane's chain is K/16 long: at M = 2 rows the launch is eight to
// thirty-two workgroups and its time is that chain), four independent partial sums per lane, slices folded in slice order.
constexpr int ROWS_CL = 16, ROWS_SL = 16;
__global__ __launch_bounds__(256) void pw_rows_f64_kernel(GemmArgs P) {
  __shared__ double red[ROWS_SL][ROWS_CL];
  const int cl = threadIdx.x % ROWS_CL, sl = threadIdx.x / ROWS_CL;
  const int n = blockIdx.x * ROWS_CL + cl, m = blockIdx.y;
  const int kper = (P.K + ROWS_SL - 1) / ROWS_SL, k0 = sl * kper, k1 = min(P.K, k0 + kper);
  const float *xr = P.a + (size_t)m * P.lda;
  const bool xform = P.ka != nullptr;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  if (n < P.N) {
    auto term = [&](int k) -> double {
      float v = xr[k];
      if (xform) v = P.ka[k] * v + P.kc[k];
      v = dl3_act(v, P.a_act);
      return (double)v * (double)P.b[(size_t)k * P.ldb + n];
    };
    int k = k0;
    for (; k + 4 <= k1; k += 4) {
      a0 += term(k);
      a1 += term(k + 1);
      a2 += term(k + 2);
      a3 += term(k + 3);
    }
    for (; k < k1; k++) a0 += term(k);
  }
  red[sl][cl] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (sl == 0 && n < P.N) {
    double t = 0.0;
#pragma unroll
    for (int i = 0; i < ROWS_SL; i++) t += red[i][cl];
    if (P.bias) t += (double)P.bias[n];
    if (P.ep_add) t += (double)(P.add_scale * P.ep_add[(size_t)(m / P.add_div) * P.ld_add + n]);
    P.c[(size_t)m * P.ldc + n] = (float)t;
  }
}

// the layer shapes the kernel is instantiated for (K, N) -> (KQ, TN); 0: not served
struct WsShape { int K, TN; };
inline int ws_tn(int K, int N) {
  if (K % 8 != 0 || N % 4 != 0) return 0;
  const int tn = dl3_cdiv(N, 32);
  // MobileNetV2's HBM-bound 1x1 convolutions (alpha = 1): 32->16, 16->96, 96->24, 24->144, 144->24, 144->32, 32->192, 192->32
  switch (K) {
    case 16: return tn == 3 ? tn : 0;
    case 24: return tn == 5 ? tn : 0;
    case 32: return (tn == 1 || tn == 6) ? tn : 0;
    case 96: return tn == 1 ? tn : 0;
    case 144: return tn == 1 ? tn : 0;
    case 192: return tn == 1 ? tn : 0;
    default: return 0;
  }
}
int ws_grid(int M) {
  // persistent: three workgroups per CU, never more waves than 32-row tiles
  const int tiles = dl3_cdiv(M, 32);
  const int g = dl3_cdiv(tiles, 4);
  return g < 768 ? g : 768;
}
#ifndef DL3_WS2_NW
#define DL3_WS2_NW 8
#endif
// round 6: the weight-stationary kernel for MFMA-bound short reductions (pw_ws2_kernel): row groups per column tile — one
// workgroup per CU over all column tiles
int ws2_groups(int M, int ntn) {
  const int tiles = dl3_cdiv(M, 32);
  int g = 256 / ntn;
  if (g < 1) g = 1;
  const int cap = dl3_cdiv(tiles, DL3_WS2_NW);
  return g < cap ? g : cap;
}
// shapes it takes: a reduction of 160, 96 or 64 (MobileNetV2's 64 x 64 blocks, deeplabv3p.py:175-198) into an output at least twice
// as wide, from a row count below which a wave walks too few tiles to pay for loading its 60-100 KB weight slice — measured per
// direction (profiles/r06_ab_calls.txt calls 34 / 35, tiled -> weight-stationary): forward 160 -> 960 at 65 536 rows 0.218 -> 0.228 ms,
// at 98 304 0.339 -> 0.312; bwd-data 960 <- 160 at 32 768 rows 0.151 -> 0.158, at 65 536 0.295 -> 0.261 (the tiled kernel's masked
// epilogue is the longer one); reduction 64: 131 072 rows either way.  DL3_WS2=0: the tiled stream kernel serves everything.
bool ws2_shape(int M, int K, int N, bool bwd = false) {
  static const int env = env_int("DL3_WS2");
  if (env == 0) return false;
  const int minrows = K == 64 ? 131072 : (bwd ? 65536 : 98304);
  return M >= minrows && N % 4 == 0 && ((K == 160 && N >= 320) || (K == 96 && N >= 192) || (K == 64 && N >= 128));
}
inline int ws2_tn(int K) { return K == 160 ? 5 : (K == 96 ? 3 : 4); }
// 1: forward, 2: bwd-data (single-tensor dY, mask and BatchNorm-backward sums from the forward input, no addend), 3: bwd-data
// of the expand convolutions 64 -> 384 (two-tensor operand over a reduction of 384, 64-wide gradient, optional residual), 0: no
int ws2_wanted(const GemmArgs &A, bool fwd, bool vec) {
  static const int env = env_int("DL3_WS2");
  if (env == 0 || !vec || (A.bias && !fwd) || A.ldc % 4 != 0 || (((uintptr_t)A.c) & 15) != 0) return 0;
  if (!fwd && A.a2) {
    if (A.M < 131072 || A.K != 384 || A.N != 64 || !A.ka || A.bias || A.stat_mode == 1 || !A.ep_x) return 0;
    if (A.ld_epx % 4 != 0 || (((uintptr_t)A.ep_x) & 15) != 0) return 0;
    if (A.ep_add && (A.add_div != 1 || A.ld_add % 4 != 0 || (((uintptr_t)A.ep_add) & 15) != 0)) return 0;
    return 3;
  }
  if (A.ep_add || A.a2 || !ws2_shape(A.M, A.K, A.N, !fwd)) return 0;
  if (fwd) return 1;
  if (A.ka || !A.ep_x || A.ld_epx % 4 != 0 || (((uintptr_t)A.ep_x) & 15) != 0 || A.stat_mode == 1) return 0;
  return 2;
}

// forward launch served by the weight-stationary kernel of the HBM-bound layers?
bool ws_wanted(const GemmArgs &A, bool fwd, bool vec) {
  if (!fwd || !vec || A.ep_add || A.a2) return false;
  if (A.ldc % 4 != 0 || (((uintptr_t)A.c) & 15) != 0) return false;  // 16-byte stores
  if (A.M < 32768) return false;
  return ws_tn(A.K, A.N) != 0;
}

// round 6: the logits layer (N = classes <= 32 off a 256-wide input): 1 = forward with the packed narrow output (pw_ws2_kernel
// FLAT), 2 = bwd-data over the narrow reduction (pw_narrowk_kernel), 0 = no.  DL3_NARROW=0: the tiled kernels (A/B aid).
bool narrow_on() {
  static const int env = env_int("DL3_NARROW");
  return env != 0;
}
bool narrow_shape(int M, int K, int N) { return narrow_on() && M >= 8192 && K == 256 && N <= 32; }
int narrow_wanted(const GemmArgs &A, bool fwd, bool avec) {
  if (A.a2 || A.ep_add || A.M < 8192 || !narrow_on()) return 0;
  if (fwd && avec && narrow_shape(A.M, A.K, A.N) && A.ldc == A.N && (((uintptr_t)A.c) & 15) == 0) return 1;
  // (16 384 rows: measured 8 192 rows 0.011 -> 0.013 ms, 65 536 rows 0.039 -> 0.023)
  if (A.M >= 16384 && A.K <= 32 && A.N == 256 && !A.ka && A.a_act == DL3_ACT_NONE && !A.ep_x && !A.bias && A.stat_mode == 0 && A.lda == A.K &&
      A.ldc % 4 == 0 && ((((uintptr_t)A.a) | ((uintptr_t)A.c)) & 15) == 0)
    return 2;
  return 0;
}
int narrow_groups(int M) {
  const int cap = dl3_cdiv(dl3_cdiv(M, 32), DL3_WS2_NW);
  return cap < 256 ? cap : 256;
}
// ... and its weight gradient (pw_wgrad_narrow_kernel): slabs = workgroups, one per CU, never more waves than 16-row spans
// (from 131 072 rows: at 65 536 the eight-wave meeting and the 256-slab fold cost what the loop saves, 0.053 -> 0.056 ms)
bool wgrad_narrow_shape(int M, int K, int N) { return narrow_on() && M >= 131072 && K == 256 && N <= 32; }
int wgrad_narrow_slabs(int M) {
  const int cap = dl3_cdiv(dl3_cdiv(M, 16), DL3_WS2_NW);
  return cap < 256 ? cap : 256;
}

// ---- configuration choice -------------------------------------------------------------
struct GemmCfg { int id, BM, BN; };
const GemmCfg kGemmCfgs[] = {{0, 128, 128}, {1, 256, 64}, {2, 256, 32}, {3, 128, 160}, {4, 128, 96},
                             {5, 32, 256},  {6, 32, 128}};  // 5, 6: four waves side by side, for small M
constexpr int kNumGemmCfgs = 7;

int env_int(const char *name) {
  const char *e = getenv(name);
  return e ? atoi(e) : -1;
}

// small: the 32-row configurations may be chosen (they exist for the stream kernel only)
GemmCfg pick_gemm(int M, int K, int N, bool two, bool small, bool fwd = true) {
  const int forced = env_int("DL3_GEMM_CFG");  // tuning aid (tools/gemm_tune.py)
  if (forced >= 0 && forced < kNumGemmCfgs && (small || kGemmCfgs[forced].BM != 32)) return kGemmCfgs[forced];
  // measured exception (tools/gemm_tune.py): a forward GEMM with a very short reduction and a wide output
  // (24 -> 144 at 128x128) is a pure streaming kernel and wants the tall 256x64 tile
  // (forward launches only: the single-tensor bwd-data launches of round 4 are `!two` as well — 144 <- 24 at 128x128 ran
  // 0.71 -> 0.83 ms through this exception)
  if (fwd && !two && K < 32 && N > 128 && M >= 65536) return kGemmCfgs[1];
  double best = 1e30;
  GemmCfg bc = kGemmCfgs[0];
  for (const GemmCfg &c : kGemmCfgs) {
    if (c.BM == 32 && !small) continue;
    // measured on MI355X (tools/gemm_tune.py): ~80 TFLOP/s sustained fp32 MFMA, ~3 TB/s streaming; re-reads of A by
    // the other column tiles of a row tile are L2 hits thanks to the XCD remap (charged at 1/4)
    const double ntn = dl3_cdiv(N, c.BN), mp = (double)dl3_cdiv(M, c.BM) * c.BM;
    // fewer workgroups than the chip holds (2 per CU) leave matrix pipes idle; the 32-row configs pay ~15 % more
    // per MFMA (every wave re-reads the shared A rows through L1) and are for exactly that case
    const double blocks = (double)dl3_cdiv(M, c.BM) * ntn;
    const double util = blocks < 512.0 ? blocks / 512.0 : 1.0;
    const double t_mfma = 2.0 * mp * K * ntn * c.BN / 80e12 / util * (c.BM == 32 ? 1.15 : 1.0);
    const double t_mem = 4.0 * ((double)M * K * (1.0 + 0.25 * (ntn - 1)) * (two ? 2 : 1) + (double)M * N) / 3e12;
    const double cost = (t_mfma > t_mem ? t_mfma : t_mem) + 0.25 * (t_mfma + t_mem);
    if (cost < best) { best = cost; bc = c; }
  }
  return bc;
}

// masked bwd-data launch whose epilogue operand can be prefetched (pw_gemm_stream_kernel EPI 2)
bool pre_ok(const GemmArgs &A) {
  return A.ep_x && !A.bias && !A.ep_add && env_int("DL3_GEMM_PRE") != 0;
}
// ... and for which the 128x96 prefetching tile beats the cost model's choice: short reductions (the epilogue is as
// long as the main loop) into a wide output made of whole 96-column tiles, with enough row tiles to fill the chip
bool pre_wanted(const GemmArgs &A) {
  return A.K <= 320 && A.N % 96 == 0 && A.N >= 2 * A.K && A.M >= 65536;
}

int gemm_grid_y(int M, int N, const GemmCfg &c, int K = 0) {
  const int mtiles = dl3_cdiv(M, c.BM), ntn = dl3_cdiv(N, c.BN);
  // target number of workgroups per launch (DL3_GEMM_PY overrides: tuning aid).  Measured on MI355X, whole step:
  const int pytot = env_int("DL3_GEMM_PY");
  // plenty of row tiles (>= 4 per resident workgroup): 512 workgroups = exactly two per CU, each a persistent loop over
  // its row tiles (sweep at B=64: 256 -> 64.3 ms, 384 -> 66.4, 512 -> 57.7, 640 -> 63.0, 768 -> 60.3, 2048 -> 58.4: any
  // count that is not a whole number of waves of the chip leaves a ragged last wave); fewer tiles: several waves of
  // short workgroups balance better (B=2: 2048 -> 4.97 ms, 512 -> 5.08)
  // round 6: ... and 5-16 LONG tiles per resident workgroup (reduction >= 512, at most 1 024 row tiles: Xception at B = 16 / 32)
  // do not want the persistent loop either: a workgroup that walks 5 or 6 such tiles wastes up to a fifth of the launch on
  // the quantisation, which the dispatcher's own balancing of one- or two-tile workgroups does not (same-call A/B, cfg4 B=16:
  // 736 -> 736 forward 0.680 -> 0.630 ms, 1536 -> 1536 2.65 -> 2.46; 85.2 -> 87.4 img/s; MobileNetV2 B=16 / 128 unchanged / -0.5 %)
  const long tiles = (long)mtiles * ntn;
  const bool long_few = K >= 512 && mtiles <= 1024 && tiles < 8192;
  const int dflt = long_few ? 4096 : (tiles >= 2048 ? 512 : DL3_GEMM_PY_DEFAULT);
  int py = (pytot > 0 ? pytot : dflt) / ntn;
  if (py < 32) py = 32;
  if (mtiles <= py) return mtiles;
  const int iters = dl3_cdiv(mtiles, py);  // every workgroup loops over the same number of row tiles ...
  const int even = dl3_cdiv(mtiles, iters);
  // ... unless that leaves more than a tenth of the chip's 512 slots empty (512 row tiles x 5 column tiles: 86 x 5 = 430
  // workgroups of 6 tiles, most CUs carry 12 tile-times; 102 x 5 = 510 workgroups of 5 or 6 carry 10-11: the Xception
  // 736 -> 736 GEMM at 65536 rows 0.810 -> 0.733 ms forward, 0.924 -> 0.869 bwd-data).
  if ((long)even * ntn * 10 < 512L * 9) return py;
  return even;
}

// split math (opt-in): device scratch for the packed weights of the launch in flight, one buffer per DEVICE.  Launches on
// one stream are ordered, so one buffer serves them all — split-math launches of a device must not be issued from two
// streams at once (the engine issues them on one stream, eagerly or under capture; the weight-gradient kernels, the
// only launches that may run on a second stream, split in registers and do not use it).  Keying by stream as well
// (tried in round 3) breaks hipGraph capture: the capture stream is not the stream the eager warm-up step grew the
// buffer on, and nothing can be allocated while capturing.  A buffer only ever grows, and a superseded one stays
// allocated (a captured hipGraph may still point at it).
void *pack_scratch(size_t bytes, hipStream_t st) {
  struct Slot { int dev; void *buf; size_t cap; };
  static std::mutex mu;
  static std::vector<Slot> slots;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lock(mu);
  Slot *s = nullptr;
  for (Slot &q : slots)
    if (q.dev == dev) { s = &q; break; }
  if (s && bytes <= s->cap) return s->buf;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(st, &cs);
  if (cs != hipStreamCaptureStatusNone) return nullptr;
  size_t want = bytes < (32u << 20) ? (32u << 20) : bytes * 2;
  void *nb = nullptr;
  if (hipMalloc(&nb, want) != hipSuccess) return nullptr;
  if (!s) {
    slots.push_back(Slot{dev, nullptr, 0});
    s = &slots.back();
  }
  s->buf = nb;
  s->cap = want;
  return nb;
}

int g_gemm_math = -1;  // DL3_MATH_ENV
bool split_math() {
  if (g_gemm_math >= 0) return g_gemm_math == 1;
  const char *e = getenv("DL3_GEMM_MATH");  // "split": fp32 as 3 x bf16 on the bf16 matrix pipe; default: f32 MFMA
  return e && e[0] == 's';
}

template <int TM, int TN, int WM, int WN>
void launch_gemm(const GemmArgs &A, dim3 grid, hipStream_t st, int vec) {
  // 16-byte loads on both operands (1), scalar loads on both (0: tiny GEMMs with K = number of classes), or on the
  // activation operand only (2: N = number of classes — the logits layer's forward)
  if (vec == 1) hipLaunchKernelGGL((pw_gemm_kernel<TM, TN, WM, WN, 1>), grid, dim3(256), 0, st, A);
  else if (vec == 2) hipLaunchKernelGGL((pw_gemm_kernel<TM, TN, WM, WN, 2>), grid, dim3(256), 0, st, A);
  else hipLaunchKernelGGL((pw_gemm_kernel<TM, TN, WM, WN, 0>), grid, dim3(256), 0, st, A);
}

inline bool al16(const void *p) { return (((uintptr_t)p) & 15) == 0; }

// returns the number of stat partial rows the launch writes (= grid.y)
int run_gemm_one(GemmArgs A, hipStream_t st) {
  const bool two = A.a2 != nullptr;
  const bool avec = (A.K % 4 == 0) && (A.lda % 4 == 0) && al16(A.a) && (!two || ((A.lda2 % 4 == 0) && al16(A.a2)));
  const bool bvec = (A.N % 4 == 0) && (A.ldb % 4 == 0) && al16(A.b);
  const bool vec = avec && bvec;
  const int vmode = vec ? 1 : (avec ? 2 : 0);
  const bool stream = vec && A.K <= DL3_STREAM_KMAX;
  // (a per-image addend stays on the forward instantiation when its straight-line epilogue can take it: 32-row blocks
  // inside one image)
  const bool fwd = !two && !A.ep_x && A.stat_mode != 2 && !(A.ep_add && A.add_div > 1 && A.add_div % 32 != 0);
  if (const int nr = split_math() ? 0 : narrow_wanted(A, fwd, avec)) {
    if (nr == 1) {
      const int nrg = ws2_groups(A.M, 1);
      hipLaunchKernelGGL((pw_ws2_kernel<32, 1, DL3_WS2_NW, false, false, false, true>), dim3(1, nrg), dim3(64 * DL3_WS2_NW), 0, st, A);
      return nrg;
    }
    hipLaunchKernelGGL((pw_narrowk_kernel<8, DL3_WS2_NW>), dim3(narrow_groups(A.M), A.N / 256), dim3(64 * DL3_WS2_NW), 0, st, A);
    return 0;
  }
  if (const int w2 = split_math() ? 0 : ws2_wanted(A, fwd, vec)) {
    if (w2 == 3) {
      const int nrg = ws2_groups(A.M, 1);
      const dim3 grid(1, nrg);
      if (A.ep_add) hipLaunchKernelGGL((pw_ws2_kernel<48, 2, DL3_WS2_NW, true, true, true>), grid, dim3(64 * DL3_WS2_NW), 0, st, A);
      else hipLaunchKernelGGL((pw_ws2_kernel<48, 2, DL3_WS2_NW, true, true, false>), grid, dim3(64 * DL3_WS2_NW), 0, st, A);
      return nrg;
    }
    const int tn = ws2_tn(A.K), ntn = dl3_cdiv(A.N, 32 * tn), nrg = ws2_groups(A.M, ntn);
    if (w2 == 2) {
      const dim3 grid(ntn, nrg);
      if (A.K == 160) hipLaunchKernelGGL((pw_ws2_kernel<20, 5, DL3_WS2_NW, true>), grid, dim3(64 * DL3_WS2_NW), 0, st, A);
      else if (A.K == 96) hipLaunchKernelGGL((pw_ws2_kernel<12, 3, DL3_WS2_NW, true>), grid, dim3(64 * DL3_WS2_NW), 0, st, A);
      else hipLaunchKernelGGL((pw_ws2_kernel<8, 4, DL3_WS2_NW, true>), grid, dim3(64 * DL3_WS2_NW), 0, st, A);
      return nrg;
    }
    const dim3 grid(ntn, nrg), blk(512);
    DL3_T(A.dbg = g_phase_dbg;)
#ifndef DL3_WS2_NW
#define DL3_WS2_NW 8
#endif
    if (A.K == 160) hipLaunchKernelGGL((pw_ws2_kernel<20, 5, DL3_WS2_NW>), grid, dim3(64 * DL3_WS2_NW), 0, st, A);
    else if (A.K == 96) hipLaunchKernelGGL((pw_ws2_kernel<12, 3, DL3_WS2_NW>), grid, dim3(64 * DL3_WS2_NW), 0, st, A);
    else hipLaunchKernelGGL((pw_ws2_kernel<8, 4, DL3_WS2_NW>), grid, dim3(64 * DL3_WS2_NW), 0, st, A);
    return nrg;
  }
  if (ws_wanted(A, fwd, vec) && !split_math()) {
    const dim3 grid(ws_grid(A.M)), blk(256);
#define DL3_WS(KQ_, TN_, OC_) hipLaunchKernelGGL((pw_fwd_ws_kernel<KQ_, TN_, OC_>), grid, blk, 0, st, A)
    const int tn = ws_tn(A.K, A.N);
    // (workgroups per CU: the most the registers allow without spilling)
    if (A.K == 16) DL3_WS(2, 3, 4);
    else if (A.K == 24) DL3_WS(3, 5, 3);
    else if (A.K == 32 && tn == 1) DL3_WS(4, 1, 4);
    else if (A.K == 32) DL3_WS(4, 6, 3);
    else if (A.K == 96) DL3_WS(12, 1, 4);
    else if (A.K == 144) DL3_WS(18, 1, 3);
    else DL3_WS(24, 1, 3);
#undef DL3_WS
    return (int)grid.x;
  }
  if (vec && !split_math()) {
    const int ktn = ksplit_tn(A.M, A.K, A.N);
    if (ktn) {
      const dim3 grid(dl3_cdiv(A.M, 32)), blk(256);
      if (ktn == 2) {
        if (two) hipLaunchKernelGGL((pw_ksplit32_kernel<2, true>), grid, blk, 0, st, A);
        else hipLaunchKernelGGL((pw_ksplit32_kernel<2, false>), grid, blk, 0, st, A);
      } else if (ktn == 3) {
        if (two) hipLaunchKernelGGL((pw_ksplit32_kernel<3, true>), grid, blk, 0, st, A);
        else hipLaunchKernelGGL((pw_ksplit32_kernel<3, false>), grid, blk, 0, st, A);
      } else {
        if (two) hipLaunchKernelGGL((pw_ksplit32_kernel<5, true>), grid, blk, 0, st, A);
        else hipLaunchKernelGGL((pw_ksplit32_kernel<5, false>), grid, blk, 0, st, A);
      }
      return (int)grid.x;
    }
  }
  GemmCfg c = pick_gemm(A.M, A.K, A.N, two, stream, fwd);
  if (stream && pre_ok(A) && pre_wanted(A)) c = kGemmCfgs[4];
  A.mtiles = dl3_cdiv(A.M, c.BM);
  DL3_T(A.dbg = g_phase_dbg;)
  dim3 grid(dl3_cdiv(A.N, c.BN), gemm_grid_y(A.M, A.N, c, A.K));
  // stream-A kernel: 10-25 % faster than the LDS-staged kernel on every layer shape, forward and bwd-data
  // (tools/gemm_tune.py).  The two-tensor bwd-data operand uses 16-deep K-tiles so that its register budget does not
  // spill.  (Unaligned operands take the LDS-staged kernel below.)
  if (stream) {
    dim3 blk(256);
    if (split_math() && c.id != 5 && c.id != 6) {  // (the 32-row small-M tiles keep the f32 MFMA: their split weight tiles exceed the LDS)
      const unsigned dyn = 4u * (two ? 3 : 2) * (unsigned)dl3_cdiv(A.K, 32) * 32;
      const int ktiles = dl3_cdiv(A.K, 32);
      A.nsub = dl3_cdiv(A.N, 32);
      const size_t pieces = (size_t)ktiles * 2 * 3 * A.nsub * 64;
      void *ws = pack_scratch(pieces * 16, st);
      if (!ws) return -1;
      A.bp = ws;
      hipLaunchKernelGGL(pack_b_kernel, dim3(dl3_cdiv(ktiles * 2 * A.nsub * 64, 256)), blk, 0, st, A.b, A.ldb, A.K, A.N,
                         (u32x4 *)ws, ktiles, A.nsub);
#define DL3_SPLIT(TM_, TN_, WN_)                                                                                     \
  do {                                                                                                               \
    if (two) hipLaunchKernelGGL((pw_gemm_stream_kernel<TM_, TN_, true, 32, 0, WN_, 1>), grid, blk, dyn, st, A);        \
    else if (fwd) hipLaunchKernelGGL((pw_gemm_stream_kernel<TM_, TN_, false, 32, 1, WN_, 1>), grid, blk, dyn, st, A);  \
    else hipLaunchKernelGGL((pw_gemm_stream_kernel<TM_, TN_, false, 32, 0, WN_, 1>), grid, blk, dyn, st, A);           \
  } while (0)
      if (c.id == 4 && pre_ok(A)) {
        if (two) hipLaunchKernelGGL((pw_gemm_stream_kernel<1, 3, true, 32, 2, 1, 1>), grid, blk, dyn, st, A);
        else hipLaunchKernelGGL((pw_gemm_stream_kernel<1, 3, false, 32, 2, 1, 1>), grid, blk, dyn, st, A);
        return (int)grid.y;
      }
      switch (c.id) {
        case 0: DL3_SPLIT(1, 4, 1); break;
        case 1: DL3_SPLIT(2, 2, 1); break;
        case 2: DL3_SPLIT(2, 1, 1); break;
        case 3: DL3_SPLIT(1, 5, 1); break;
        default: DL3_SPLIT(1, 3, 1); break;
      }
#undef DL3_SPLIT
      return (int)grid.y;
    }
#define DL3_STREAM(TM_, TN_, WN_)                                                                                    \
  do {                                                                                                               \
    if (two) hipLaunchKernelGGL((pw_gemm_stream_kernel<TM_, TN_, true, 16, 0, WN_>), grid, blk, 0, st, A);           \
    else if (fwd) hipLaunchKernelGGL((pw_gemm_stream_kernel<TM_, TN_, false, DL3_STREAM_KT_FWD, 1, WN_>), grid, blk, 0, st, A); \
    else hipLaunchKernelGGL((pw_gemm_stream_kernel<TM_, TN_, false, DL3_STREAM_KT_BWD1, 0, WN_>), grid, blk, 0, st, A); \
  } while (0)
    if (c.id == 4 && pre_ok(A)) {
      if (two) hipLaunchKernelGGL((pw_gemm_stream_kernel<1, 3, true, 16, 2, 1>), grid, blk, 0, st, A);
      else hipLaunchKernelGGL((pw_gemm_stream_kernel<1, 3, false, 16, 2, 1>), grid, blk, 0, st, A);
      return (int)grid.y;
    }
    switch (c.id) {
      case 0: DL3_STREAM(1, 4, 1); break;
      case 1: DL3_STREAM(2, 2, 1); break;
      case 2: DL3_STREAM(2, 1, 1); break;
      case 3: DL3_STREAM(1, 5, 1); break;
      case 5: DL3_STREAM(1, 2, 4); break;
      case 6: DL3_STREAM(1, 1, 4); break;
      default: DL3_STREAM(1, 3, 1); break;
    }
#undef DL3_STREAM
    return (int)grid.y;
  }
  switch (c.id) {
    case 0: launch_gemm<2, 2, 2, 2>(A, grid, st, vmode); break;
    case 1: launch_gemm<2, 2, 4, 1>(A, grid, st, vmode); break;
    case 2: launch_gemm<2, 1, 4, 1>(A, grid, st, vmode); break;
    case 3: launch_gemm<1, 5, 4, 1>(A, grid, st, vmode); break;
    default: launch_gemm<1, 3, 4, 1>(A, grid, st, vmode); break;
  }
  return (int)grid.y;
}

// round 6: Xception's 728-channel layers (deeplabv3p.py:300-306) are stored 736 wide — 23 column blocks of 32,
// which no tile width divides: six 128-wide tiles compute 24 (the stream kernel multiplies zero weight columns like any other).
// 23 = 2 x 4 + 3 x 5: such a launch is issued as TWO over disjoint column ranges of the same output, [0, N - 480) on the
// 128-wide tiles and the last 480 columns on the 160-wide ones — 23 blocks exactly, both launches several full rounds of the
// chip.  The operand rows are read by both (the second pass over a 65 536 x 736 operand is 0.19 GB against 0.65 ms of MFMA
// time); partial sums land in disjoint columns of the same rows.  (First cut, 640 + 96 columns: the 128 x 96 tile of the
// narrow launch costs 0.78 of a 128-wide one — forward 0.749 -> 0.712 ms, bwd-data 0.798 -> 0.796.)  DL3_COLSPLIT=0: one launch.
constexpr int kColsplitTail = 480;
bool colsplit_shape(int M, int K, int N) {
  static const int env = env_int("DL3_COLSPLIT");
  return env != 0 && M >= 32768 && K >= 256 && K <= DL3_STREAM_KMAX && N >= kColsplitTail + 128 && N % 128 == 96 && N % 160 != 0;
}
int run_gemm(GemmArgs A, hipStream_t st) {
  const bool two = A.a2 != nullptr;
  const bool vec = (A.K % 4 == 0) && (A.lda % 4 == 0) && al16(A.a) && (!two || ((A.lda2 % 4 == 0) && al16(A.a2))) &&
                   (A.N % 4 == 0) && (A.ldb % 4 == 0) && al16(A.b);
  // (single-tensor operand only: the two-tensor bwd-data form, which the engine does not use once the weight-gradient launch
  // has written dY, came out 2 % slower in two launches — 0.834 -> 0.851 ms)
  if (!vec || two || split_math() || !colsplit_shape(A.M, A.K, A.N)) return run_gemm_one(A, st);
  const int n1 = A.N - kColsplitTail;
  GemmArgs S[2] = {A, A};
  S[0].N = n1;
  S[1].N = kColsplitTail;
  S[1].b += n1; S[1].c += n1;
  if (A.bias) S[1].bias += n1;
  if (A.ep_x) S[1].ep_x += n1;
  if (A.ep_scale) { S[1].ep_scale += n1; S[1].ep_shift += n1; }
  if (A.ep_mean) { S[1].ep_mean += n1; S[1].ep_invstd += n1; }
  if (A.ep_add) S[1].ep_add += n1;
  if (A.part) S[1].part += 2 * (size_t)n1;
  int rows = 0;
  for (GemmArgs &q : S) {
    q.part_ld = A.N;
    const int r = run_gemm_one(q, st);
    if (r < 0) return r;
    rows = r > rows ? r : rows;
  }
  return rows;
}

struct WgCfg { int id, BKT, BNT; };
const WgCfg kWgCfgs[] = {{0, 64, 64},  {1, 128, 128}, {2, 160, 128}, {3, 128, 160}, {4, 64, 128},
                         {5, 128, 64}, {6, 32, 128},  {7, 128, 32},  {8, 96, 128},  {9, 128, 96}};

// M splits the tile cost model reasons with: ~1024 workgroups, capped by slab traffic and by rows per split
int wgrad_splits_model(int M, int K, int N, const WgCfg &c, int target = DL3_WGRAD_WGS_DEFAULT) {
  const long tiles = (long)dl3_cdiv(K, c.BKT) * dl3_cdiv(N, c.BNT);
  long S = target / tiles;
  const long cap_traffic = (long)((double)M * (K + N) / (4.0 * K * N));
  const long cap_rows = M / 64;
  if (S > cap_traffic) S = cap_traffic;
  if (S > cap_rows) S = cap_rows;
  if (S < 1) S = 1;
  return (int)S;
}

// ... and the M splits a launch gets.  Every split writes a K x N slab that the fold reads back; for the 160-wide tiles
// on a small weight matrix (<= 160 x 960) at 32k-128k rows that is a fifth of the launch's bytes, and two workgroups per CU
// (half the slabs) are faster.  Measured in situ (round 4, calls 16 / 17, B=16: M = 65536): 160 x 960 291 -> 258 us,
// 960 x 160 253 -> 230, 576 x 160 180 -> 156; whole step 1 055 / 1 068 / 1 071 -> 1 068 / 1 082 / 1 086 img/s.  NOT a general
// rule: the same halving on Xception's 736 x 736 at the same M costs 24 % of the launch, on 64 x 384 5 %, and at
// M = 16384 (B=4) the 160-wide shapes lose 12 % (profiles/r04_ab_calls.txt, call 17).
int wgrad_splits(int M, int K, int N, const WgCfg &c) {
  const int S = wgrad_splits_model(M, K, N, c);
  const bool wide = (c.id == 2 || c.id == 3) && (long)K * N <= 160L * 960 && M >= 32768;
  if (wide && 2.0 * (double)S * K * N > 0.1 * (double)M * (K + N)) return wgrad_splits_model(M, K, N, c, DL3_WGRAD_WGS_DEFAULT / 2);
  return S;
}

WgCfg pick_wgrad(int M, int K, int N, bool two) {
  const int forced = env_int("DL3_WGRAD_CFG");
  if (forced >= 0 && forced < 10) return kWgCfgs[forced];
  // measured shortcuts (tools/gemm_tune.py, MI355X, M >= 32k rows): small weight matrices want the 64x64 tile (more
  // workgroups per M split), 160-multiples want the 160-wide tiles so the big operand is read once
  if (M >= 32768) {
    if (K >= 32 && N >= 32 && (long)K * N <= 32768) return kWgCfgs[0];
    if (N % 160 == 0 && K >= 128 && K % 160 != 0) return kWgCfgs[3];
    if (K % 160 == 0 && K % 128 != 0 && N >= 128 && N % 160 != 0) return kWgCfgs[2];
    if (K % 160 == 0 && N % 160 == 0 && K >= 320 && N >= 320) return kWgCfgs[3];
  }
  double best = 1e30;
  WgCfg bc = kWgCfgs[0];
  for (const WgCfg &c : kWgCfgs) {
    const double ntk = dl3_cdiv(K, c.BKT), ntn = dl3_cdiv(N, c.BNT);
    // few rows: the M split is capped (partial-slab traffic), so small tiles are what fills the chip
    const double blocks = ntk * ntn * wgrad_splits_model(M, K, N, c);
    const double util = blocks < 512.0 ? blocks / 512.0 : 1.0;
    const double t_mfma = 2.0 * M * ntk * c.BKT * ntn * c.BNT / 80e12 / util;
    const double t_mem = 4.0 * ((double)M * K * (1.0 + 0.25 * (ntn - 1)) +
                                (double)M * N * (1.0 + 0.25 * (ntk - 1)) * (two ? 2 : 1)) / 3e12;
    const double cost = (t_mfma > t_mem ? t_mfma : t_mem) + 0.25 * (t_mfma + t_mem);
    if (cost < best) { best = cost; bc = c; }
  }
  return bc;
}

int colsum_rows(int M) {
  int pr = M / 256;
  if (pr < 1) pr = 1;
  if (pr > 256) pr = 256;
  return pr;
}

template <int TA, int TB, int WA, int WB>
void launch_wgrad(const WgradArgs &A, dim3 grid, hipStream_t st, int vec) {
  if (vec == 1 && split_math())
    hipLaunchKernelGGL((pw_wgrad_kernel<TA, TB, WA, WB, 1, true>), grid, dim3(256), 0, st, A);
  else if (vec == 1) hipLaunchKernelGGL((pw_wgrad_kernel<TA, TB, WA, WB, 1>), grid, dim3(256), 0, st, A);
  else if (vec == 2) hipLaunchKernelGGL((pw_wgrad_kernel<TA, TB, WA, WB, 2>), grid, dim3(256), 0, st, A);
  else hipLaunchKernelGGL((pw_wgrad_kernel<TA, TB, WA, WB, 0>), grid, dim3(256), 0, st, A);
}

}  // namespace

extern "C" int dl3_reduce_partials(const float *partial, int P, int n, float *out, void *stream);

#ifdef DL3_PHASE_TIMING
extern "C" int dl3_debug_phase_buffer(long long *buf) {
  g_phase_dbg = buf;
  return DL3_OK;
}
#endif

extern "C" int dl3_set_gemm_math(int mode) {
  DL3_CHECK_ARG(mode >= -1 && mode <= 1, "set_gemm_math: mode must be DL3_MATH_ENV, DL3_MATH_F32 or DL3_MATH_SPLIT");
  g_gemm_math = mode;
  return DL3_OK;
}

extern "C" int dl3_get_gemm_math(void) { return split_math() ? 1 : 0; }

extern "C" int dl3_pwconv_partials(int M, int K, int N) {
  if (M <= 0 || K <= 0 || N <= 0) return 0;
  // the stat partial row count must not depend on which operand form is used: take the max
  int p = 0;
  for (int two = 0; two < 2; two++)
    for (int small = 0; small < 2; small++)
      for (int fwd = 0; fwd < 2; fwd++) {
        const int q = gemm_grid_y(M, N, pick_gemm(M, K, N, two != 0, small != 0, fwd != 0), K);
        p = q > p ? q : p;
      }
  if (ws_tn(K, N)) {  // the weight-stationary forward kernel writes one row per workgroup
    const int q = ws_grid(M);
    p = q > p ? q : p;
  }
  if (ws2_shape(M, K, N, true)) {   // (the lower of the two directions' row thresholds)
    const int q = ws2_groups(M, dl3_cdiv(N, 32 * ws2_tn(K)));
    p = q > p ? q : p;
  }
  if (narrow_shape(M, K, N)) {   // (the logits layer's forward: one column tile)
    const int q = ws2_groups(M, 1);
    p = q > p ? q : p;
  }
  if (M >= 131072 && K == 384 && N == 64) {   // (the two-tensor bwd-data route of the expand convolutions: one column tile)
    const int q = ws2_groups(M, 1);
    p = q > p ? q : p;
  }
  if (ksplit_tn(M, K, N)) {  // the K-split kernel of the small batches: one row per 32-row tile
    const int q = dl3_cdiv(M, 32);
    p = q > p ? q : p;
  }
  if (colsplit_shape(M, K, N)) {   // (two launches over column slices: each sizes its own grid)
    const int a = dl3_pwconv_partials(M, K, N - kColsplitTail), b = dl3_pwconv_partials(M, K, kColsplitTail);
    p = a > p ? a : p;
    p = b > p ? b : p;
  }
  if (N % 96 == 0) {  // the prefetching bwd-data variant overrides the choice with the 128x96 tile (run_gemm)
    const int q = gemm_grid_y(M, N, kGemmCfgs[4], K);
    p = q > p ? q : p;
  }
  return p;
}

extern "C" int dl3_pwconv_fwd_impl(int M, int K, int N) {
  if (M <= 0 || K <= 0 || N <= 0) return 0;
  GemmArgs A{};
  A.M = M; A.K = K; A.N = N; A.ldc = N;
  if (split_math()) return 0;
  if (narrow_wanted(A, true, true) == 1) return 2;
  if (ws2_wanted(A, true, true)) return 2;
  return ws_wanted(A, true, true) ? 1 : 0;
}

// the weight gradient of a launch with tile configuration c takes the one-tile-row kernel (DL3_WGRAD_ROW=0: off — A/B aid)
static bool wgrad_row_ok(const WgCfg &c, int M, int K, int N) {
  static const int row_env = env_int("DL3_WGRAD_ROW");
  return row_env != 0 && !split_math() && dl3_cdiv(K, c.BKT) == 1 && K % 4 == 0 && N % 4 == 0 && (c.id == 2 || c.id == 8) && M >= 32768;
}

extern "C" int dl3_pwconv_route(int dir, int M, int K, int N) {
  if (M <= 0 || K <= 0 || N <= 0 || split_math()) return DL3_ROUTE_TILED;
  if (dir == 2) return wgrad_row_ok(pick_wgrad(M, K, N, true), M, K, N) ? DL3_ROUTE_WGRAD_ROW : DL3_ROUTE_TILED;
  if (dir == 4) return wgrad_narrow_shape(M, K, N) ? DL3_ROUTE_NARROW : DL3_ROUTE_TILED;
  GemmArgs A{};
  if (dir == 3) {   // bwd-data of a layer K -> N without a mask operand: reduces over N, K wide
    A.M = M; A.K = N; A.N = K; A.ldc = K; A.lda = N;
    return narrow_wanted(A, true, false) == 2 ? DL3_ROUTE_NARROW : DL3_ROUTE_TILED;
  }
  if (dir == 0) {
    A.M = M; A.K = K; A.N = N; A.ldc = N;
    if (narrow_wanted(A, true, true) == 1) return DL3_ROUTE_NARROW;
    if (ws2_wanted(A, true, true)) return DL3_ROUTE_WS_MFMA;
    if (ws_wanted(A, true, true)) return DL3_ROUTE_WS_HBM;
  } else {
    // bwd-data of a layer K -> N: the GEMM reduces over N and is K wide
    A.M = M; A.K = N; A.N = K; A.ldc = K; A.ld_epx = K;
    A.ep_x = reinterpret_cast<const float *>(uintptr_t(16));   // (an aligned mask operand is present: only its presence matters)
    if (ws2_wanted(A, false, true) == 2) return DL3_ROUTE_WS_MFMA;
  }
  return ksplit_tn(A.M, A.K, A.N) ? DL3_ROUTE_KSPLIT : DL3_ROUTE_TILED;
}

static int gemm_common_check(const char *name, int M, int K, int N) {
  DL3_CHECK_ARG(M > 0 && K > 0 && N > 0, "%s: non-positive dimension", name);
  return DL3_OK;
}

extern "C" int dl3_pwconv_fwd(const float *x, int ldx, const float *in_scale, const float *in_shift, int in_act,
                              const float *w, const float *bias, float *y, int ldy, int M, int K, int N,
                              float *stat_partial, void *stream) {
  int rc = gemm_common_check("pwconv_fwd", M, K, N);
  if (rc) return rc;
  DL3_CHECK_ARG(x && w && y, "pwconv_fwd: null pointer");
  DL3_CHECK_ARG(ldx >= K && ldy >= N, "pwconv_fwd: leading dimension too small");
  DL3_CHECK_ARG((in_scale == nullptr) == (in_shift == nullptr), "pwconv_fwd: scale/shift must come together");
  GemmArgs A{};
  A.a = x; A.lda = ldx; A.a2 = nullptr; A.lda2 = 0;
  A.ka = in_scale; A.kb = nullptr; A.kc = in_shift; A.a_act = in_act;
  A.b = w; A.ldb = N; A.bias = bias; A.c = y; A.ldc = ldy;
  A.M = M; A.K = K; A.N = N;
  A.add_div = 1; A.add_scale = 1.f;
  A.stat_mode = stat_partial ? 1 : 0;
  A.part = stat_partial;
  A.part_rows = stat_partial ? dl3_pwconv_partials(M, K, N) : 0;
  hipStream_t st = (hipStream_t)stream;
  const int written = run_gemm(A, st);
  DL3_CHECK_ARG(written >= 0, "pwconv_fwd: split-math weight scratch unavailable (first launch inside a stream capture?)");
  DL3_LAUNCH_CHECK("pwconv_fwd");
  return DL3_OK;
}

extern "C" int dl3_pwconv_fwd_add(const float *x, int ldx, const float *in_scale, const float *in_shift, int in_act,
                                  const float *w, const float *bias, float *y, int ldy, int M, int K, int N,
                                  float *stat_partial, const float *add, int ldadd, int add_div, void *stream) {
  int rc = gemm_common_check("pwconv_fwd_add", M, K, N);
  if (rc) return rc;
  DL3_CHECK_ARG(x && w && y && add, "pwconv_fwd_add: null pointer");
  DL3_CHECK_ARG(ldx >= K && ldy >= N && ldadd >= N && add_div >= 1, "pwconv_fwd_add: bad leading dimension / add_div");
  DL3_CHECK_ARG((in_scale == nullptr) == (in_shift == nullptr), "pwconv_fwd_add: scale/shift must come together");
  GemmArgs A{};
  A.a = x; A.lda = ldx; A.a2 = nullptr; A.lda2 = 0;
  A.ka = in_scale; A.kb = nullptr; A.kc = in_shift; A.a_act = in_act;
  A.b = w; A.ldb = N; A.bias = bias; A.c = y; A.ldc = ldy;
  A.M = M; A.K = K; A.N = N;
  A.ep_add = add; A.ld_add = ldadd; A.add_div = add_div; A.add_scale = 1.f;
  A.stat_mode = stat_partial ? 1 : 0;
  A.part = stat_partial;
  A.part_rows = stat_partial ? dl3_pwconv_partials(M, K, N) : 0;
  hipStream_t st = (hipStream_t)stream;
  const int written = run_gemm(A, st);
  DL3_CHECK_ARG(written >= 0, "pwconv_fwd_add: split-math weight scratch unavailable (first launch inside a stream capture?)");
  DL3_LAUNCH_CHECK("pwconv_fwd_add");
  return DL3_OK;
}

extern "C" int dl3_pwconv_fwd_rows(const float *x, int ldx, const float *in_scale, const float *in_shift, int in_act,
                                   const float *w, const float *bias, float *y, int ldy, int M, int K, int N,
                                   const float *add, int ldadd, int add_div, void *stream) {
  int rc = gemm_common_check("pwconv_fwd_rows", M, K, N);
  if (rc) return rc;
  DL3_CHECK_ARG(x && w && y, "pwconv_fwd_rows: null pointer");
  DL3_CHECK_ARG(ldx >= K && ldy >= N && (!add || (ldadd >= N && add_div >= 1)), "pwconv_fwd_rows: bad leading dimension / add_div");
  DL3_CHECK_ARG((in_scale == nullptr) == (in_shift == nullptr), "pwconv_fwd_rows: scale/shift must come together");
  DL3_UNSUPPORTED(M > 65535, "pwconv_fwd_rows: a few rows (one per image), not %d", M);
  GemmArgs A{};
  A.a = x; A.lda = ldx; A.ka = in_scale; A.kc = in_shift; A.a_act = in_act;
  A.b = w; A.ldb = N; A.bias = bias; A.c = y; A.ldc = ldy;
  A.M = M; A.K = K; A.N = N;
  A.ep_add = add; A.ld_add = ldadd; A.add_div = add_div < 1 ? 1 : add_div; A.add_scale = 1.f;
  hipLaunchKernelGGL(pw_rows_f64_kernel, dim3(dl3_cdiv(N, ROWS_CL), M), dim3(256), 0, (hipStream_t)stream, A);
  DL3_LAUNCH_CHECK("pwconv_fwd_rows");
  return DL3_OK;
}

extern "C" int dl3_pwconv_bwd_data(const float *g, int ldg, const float *yraw, int ldyraw, const float *cA,
                                   const float *cB, const float *cC, const float *wT, float *dx, int lddx,
                                   const float *x, int ldx, const float *in_scale, const float *in_shift,
                                   int in_act, const float *dx_add, int ldadd, int add_div, float add_scale,
                                   const float *x_mean, const float *x_invstd, float *dstat_partial, int M,
                                   int K, int N, void *stream) {
  int rc = gemm_common_check("pwconv_bwd_data", M, K, N);
  if (rc) return rc;
  DL3_CHECK_ARG(g && wT && dx, "pwconv_bwd_data: null pointer");
  DL3_CHECK_ARG(!cA || (yraw && cB && cC), "pwconv_bwd_data: cA needs yraw, cB, cC");
  DL3_CHECK_ARG(in_act == DL3_ACT_NONE || x, "pwconv_bwd_data: activation mask needs x");
  DL3_CHECK_ARG(!dstat_partial || (x && x_mean && x_invstd), "pwconv_bwd_data: dstat needs x, x_mean, x_invstd");
  DL3_CHECK_ARG((in_scale == nullptr) == (in_shift == nullptr), "pwconv_bwd_data: scale/shift must come together");
  // GEMM [M, N] x [N, K] -> [M, K]
  GemmArgs A{};
  A.a = g; A.lda = ldg;
  A.a2 = cA ? yraw : nullptr; A.lda2 = ldyraw;
  A.ka = cA; A.kb = cB; A.kc = cC; A.a_act = DL3_ACT_NONE;
  A.b = wT; A.ldb = K; A.bias = nullptr; A.c = dx; A.ldc = lddx;
  A.M = M; A.K = N; A.N = K;
  const bool need_x = (in_act != DL3_ACT_NONE) || dstat_partial;
  A.ep_x = need_x ? x : nullptr; A.ld_epx = ldx;
  A.ep_scale = in_scale; A.ep_shift = in_shift; A.ep_act = in_act;
  A.ep_add = dx_add; A.ld_add = ldadd; A.add_div = add_div < 1 ? 1 : add_div; A.add_scale = add_scale;
  A.stat_mode = dstat_partial ? 2 : 0;
  A.ep_mean = x_mean; A.ep_invstd = x_invstd;
  A.part = dstat_partial;
  A.part_rows = dstat_partial ? dl3_pwconv_partials(M, N, K) : 0;
  hipStream_t st = (hipStream_t)stream;
  const int written = run_gemm(A, st);
  DL3_CHECK_ARG(written >= 0, "pwconv_bwd_data: split-math weight scratch unavailable (first launch inside a stream capture?)");
  DL3_LAUNCH_CHECK("pwconv_bwd_data");
  return DL3_OK;
}

extern "C" size_t dl3_pwconv_bwd_weight_workspace(int M, int K, int N) {
  if (M <= 0 || K <= 0 || N <= 0) return 0;
  // S depends on the operand form; size for the larger
  const WgCfg c1 = pick_wgrad(M, K, N, false), c2 = pick_wgrad(M, K, N, true);
  int S1 = wgrad_splits(M, K, N, c1), S2 = wgrad_splits(M, K, N, c2);
  int S = S1 > S2 ? S1 : S2, crows = colsum_rows(M);
  if (wgrad_narrow_shape(M, K, N)) {   // (one slab and one row of column sums per workgroup)
    const int Sn = wgrad_narrow_slabs(M);
    S = Sn > S ? Sn : S;
    crows = Sn > crows ? Sn : crows;
  }
  return ((size_t)S * K * N + (size_t)crows * N) * sizeof(float);
}

extern "C" int dl3_pwconv_bwd_weight_splits(int M, int K, int N, int two_tensor_dy) {
  if (M <= 0 || K <= 0 || N <= 0) return 0;
  return wgrad_splits(M, K, N, pick_wgrad(M, K, N, two_tensor_dy != 0));
}

// one weight-gradient launch on tile configuration c (row: K fits one tile row of it — the straight-line kernel of the
// expand convolutions, with the requests in front of the dY stores)
static void launch_wgrad_cfg(const WgradArgs &A, const WgCfg &c, dim3 grid, hipStream_t st, int vec, bool row) {
  if (row) {   // (c.id 2 or 8, 16-byte loads on both operands: wgrad_row_ok)
    if (c.id == 2) {
      if (A.dyout) hipLaunchKernelGGL((pw_wgrad_row_kernel<5, 1, 1, 4, true>), grid, dim3(256), 0, st, A);
      else hipLaunchKernelGGL((pw_wgrad_row_kernel<5, 1, 1, 4, false>), grid, dim3(256), 0, st, A);
    } else {
      if (A.dyout) hipLaunchKernelGGL((pw_wgrad_row_kernel<3, 1, 1, 4, true>), grid, dim3(256), 0, st, A);
      else hipLaunchKernelGGL((pw_wgrad_row_kernel<3, 1, 1, 4, false>), grid, dim3(256), 0, st, A);
    }
    return;
  }
  switch (c.id) {
    case 0: launch_wgrad<1, 1, 2, 2>(A, grid, st, vec); break;
    case 1: launch_wgrad<2, 2, 2, 2>(A, grid, st, vec); break;
    case 2: launch_wgrad<5, 1, 1, 4>(A, grid, st, vec); break;
    case 3: launch_wgrad<1, 5, 4, 1>(A, grid, st, vec); break;
    case 4: launch_wgrad<2, 1, 1, 4>(A, grid, st, vec); break;
    case 5: launch_wgrad<1, 2, 4, 1>(A, grid, st, vec); break;
    case 6: launch_wgrad<1, 1, 1, 4>(A, grid, st, vec); break;
    case 7: launch_wgrad<1, 1, 4, 1>(A, grid, st, vec); break;
    case 8: launch_wgrad<3, 1, 1, 4>(A, grid, st, vec); break;
    default: launch_wgrad<1, 3, 4, 1>(A, grid, st, vec); break;
  }
}

static int pwconv_bwd_weight_impl(const float *x, int ldx, const float *in_scale, const float *in_shift,
                                     int in_act, const float *g, int ldg, const float *yraw, int ldyraw,
                                     const float *cA, const float *cB, const float *cC, float *dw, float *dbias,
                                     int M, int K, int N, void *workspace, size_t workspace_bytes, float *dy_out, int lddy,
                                     void *stream) {
  int rc = gemm_common_check("pwconv_bwd_weight", M, K, N);
  if (rc) return rc;
  DL3_CHECK_ARG(!dy_out || (lddy >= N && lddy % 4 == 0 && al16(dy_out)),
                "pwconv_bwd_weight_dy: dy_out must be 16-byte aligned with a leading dimension >= N that is a multiple of 4");
  DL3_CHECK_ARG(x && g && workspace, "pwconv_bwd_weight: null pointer");
  DL3_CHECK_ARG(dw || !dbias, "pwconv_bwd_weight: dbias without dw (slabs left in the workspace) is not supported");
  DL3_CHECK_ARG(!cA || (yraw && cB && cC), "pwconv_bwd_weight: cA needs yraw, cB, cC");
  DL3_CHECK_ARG((in_scale == nullptr) == (in_shift == nullptr), "pwconv_bwd_weight: scale/shift must come together");
  DL3_UNSUPPORTED(dbias && cA, "pwconv_bwd_weight: dbias with a BN-backward gradient operand is not needed by the path");
  if (workspace_bytes < dl3_pwconv_bwd_weight_workspace(M, K, N)) {
    dl3_set_error("pwconv_bwd_weight: workspace %zu < %zu bytes", workspace_bytes,
                  dl3_pwconv_bwd_weight_workspace(M, K, N));
    return DL3_EWORKSPACE;
  }
  const bool two = cA != nullptr;
  const WgCfg c = pick_wgrad(M, K, N, two);
  const int S = wgrad_splits(M, K, N, c);
  WgradArgs A{};
  A.x = x; A.ldx = ldx; A.xs = in_scale; A.xt = in_shift; A.x_act = in_act;
  A.g = g; A.ldg = ldg; A.y = two ? yraw : nullptr; A.ldy = ldyraw;
  A.cA = cA; A.cB = cB; A.cC = cC;
  A.ws = (float *)workspace;
  A.M = M; A.K = K; A.N = N;
  A.dyout = dy_out; A.lddy = lddy;
  A.Mper = dl3_cdiv(dl3_cdiv(M, S), DL3_WGRAD_MS) * DL3_WGRAD_MS;
  DL3_T(A.dbg = g_phase_dbg;)
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(dl3_cdiv(N, c.BNT), dl3_cdiv(K, c.BKT), S);
  const bool xvec = (K % 4 == 0) && (ldx % 4 == 0) && al16(x);
  const bool dvec = (N % 4 == 0) && (ldg % 4 == 0) && al16(g) && (!two || ((ldyraw % 4 == 0) && al16(yraw)));
  // the logits layer: N = classes, the whole K x N gradient in every wave's accumulators
  if (xvec && !two && !dy_out && dw && !split_math() && wgrad_narrow_shape(M, K, N)) {
    const int Sn = wgrad_narrow_slabs(M);
    float *cpart = dbias ? A.ws + (size_t)Sn * K * N : nullptr;
    hipLaunchKernelGGL((pw_wgrad_narrow_kernel<2, DL3_WS2_NW>), dim3(Sn), dim3(64 * DL3_WS2_NW), 0, st, A, cpart);
    DL3_LAUNCH_CHECK("pwconv_bwd_weight(narrow)");
    rc = dl3_reduce_partials(A.ws, Sn, K * N, dw, stream);
    if (rc || !dbias) return rc;
    return dl3_reduce_partials(cpart, Sn, N, dbias, stream);
  }
  launch_wgrad_cfg(A, c, grid, st, (xvec && dvec) ? 1 : (xvec ? 2 : 0), xvec && dvec && wgrad_row_ok(c, M, K, N));
  DL3_LAUNCH_CHECK("pwconv_bwd_weight");
  if (!dw) return DL3_OK;  // the caller folds the [S][K][N] slabs itself (dl3_reduce_partials / _batched)
  rc = dl3_reduce_partials(A.ws, S, K * N, dw, stream);
  if (rc) return rc;
  if (dbias) {
    float *cpart = A.ws + (size_t)S * K * N;
    const int pr = colsum_rows(M);
    hipLaunchKernelGGL(colsum_kernel, dim3(dl3_cdiv(N, 64), pr), dim3(256), 0, st, g, ldg, M, N, cpart);
    DL3_LAUNCH_CHECK("pwconv_bwd_weight(colsum)");
    rc = dl3_reduce_partials(cpart, pr, N, dbias, stream);
    if (rc) return rc;
  }
  return DL3_OK;
}

extern "C" int dl3_pwconv_bwd_weight(const float *x, int ldx, const float *in_scale, const float *in_shift,
                                     int in_act, const float *g, int ldg, const float *yraw, int ldyraw,
                                     const float *cA, const float *cB, const float *cC, float *dw, float *dbias,
                                     int M, int K, int N, void *workspace, size_t workspace_bytes, void *stream) {
  return pwconv_bwd_weight_impl(x, ldx, in_scale, in_shift, in_act, g, ldg, yraw, ldyraw, cA, cB, cC, dw, dbias, M, K, N,
                                workspace, workspace_bytes, nullptr, 0, stream);
}

extern "C" int dl3_pwconv_bwd_weight_dy(const float *x, int ldx, const float *in_scale, const float *in_shift,
                                        int in_act, const float *g, int ldg, const float *yraw, int ldyraw,
                                        const float *cA, const float *cB, const float *cC, float *dw, float *dbias,
                                        int M, int K, int N, void *workspace, size_t workspace_bytes, float *dy_out,
                                        int lddy, void *stream) {
  DL3_CHECK_ARG(dy_out, "pwconv_bwd_weight_dy: dy_out is NULL (use dl3_pwconv_bwd_weight)");
  return pwconv_bwd_weight_impl(x, ldx, in_scale, in_shift, in_act, g, ldg, yraw, ldyraw, cA, cB, cC, dw, dbias, M, K, N,
                                workspace, workspace_bytes, dy_out, lddy, stream);
}

extern "C" int dl3_transpose(const float *in, float *out, int rows, int cols, void *stream) {
  DL3_CHECK_ARG(in && out && rows > 0 && cols > 0, "transpose: bad argument");
  dim3 grid(dl3_cdiv(cols, 32), dl3_cdiv(rows, 32));
  hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, (hipStream_t)stream, in, out, rows, cols);
  DL3_LAUNCH_CHECK("transpose");
  return DL3_OK;
}

extern "C" int dl3_transpose_batched(const long long *desc, int n, int total_tiles, void *stream) {
  DL3_CHECK_ARG(desc && n > 0 && total_tiles > 0, "transpose_batched: bad argument");
  hipLaunchKernelGGL(transpose_batched_kernel, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, desc, n);
  DL3_LAUNCH_CHECK("transpose_batched");
  return DL3_OK;
}
