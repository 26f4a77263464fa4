// pwfused.hip — Conv2D 1x1 backward, BOTH gradients in one pass over (g, y, x), for the HBM-bound layers
// (deeplabv3p.py:175-201: the expand / project convolutions of the first inverted-residual blocks, 16..144 channels
// on 256x256 / 128x128 maps; round 4).
//
// dl3_pwconv_bwd_weight + dl3_pwconv_bwd_data read the gradient operand (g, yraw: the wide side of an expand
// convolution) twice, once each.  Where K and N are both small the whole weight matrix fits in LDS and both products
// can be formed from ONE staged tile of 32 pixel rows:
//     dW[K,N] += T(X)^T . dY           (reduction over the rows; accumulators live across the workgroup's row range)
//     dX[32,K] = dY . W^T              (reduction over N; finished, masked, summed and stored per stage)
// with dY = cA*g + cB*yraw + cC assembled once on the way into LDS.  Traffic per row: K + 2N reads + K (+K, +K) writes
// instead of (K + 2N) + (2N + K ...): 16 -> 96 at 256x256 moves 8.6 GB instead of 15 GB per launch pair at B = 128.
// These launches sit at 20 % of the matrix pipe: v_mfma_f32_32x32x2_f32 on fragments read straight from the staged
// tiles (the dX operand with a 4-way bank conflict — not what an HBM-bound launch waits for).
// K, N multiples of 4, ceil(K/32) * ceil(N/32) <= 6 (the weight-gradient accumulators of one wave: round 5).
//
// Round 5 (pw_bwd_fused2_kernel, the default; DL3_FUSED_V=1 keeps the round-4 kernel for same-call A/B): the round-4
// kernel gave every wave WHOLE blocks — for K <= 32 ONE wave carried the entire dX reduction (80 dependent MFMAs per
// 32-row stage at N = 144, 5 000 cycles) next to three idle ones, and the launch ran at the speed of that wave
// (24 <-> 144 at 2.9 TB/s).  Now every wave owns 8 of the stage's 32 rows for dW (all K x N blocks, accumulated across
// the workgroup's row range and summed over the four waves once, at the end, in a fixed order) and, where there are
// fewer than four dX blocks, a slice of the dX reduction (partial 32x32 sums exchanged through LDS, summed in a fixed
// order): 40 MFMAs per wave and stage instead of 96 for 24 -> 144.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace {

struct FusedArgs {
  const float *x; int ldx; const float *xs, *xt; int x_act;                       // forward input view T(x)
  const float *g; int ldg; const float *y; int ldy; const float *cA, *cB, *cC;    // gradient operand dY
  const float *wT;                                                                // [N][K]
  float *slab;                                                                    // [S][K][N]
  float *dx; int lddx;
  const float *add; int ldadd;
  const float *sx; int ldsx; const float *mean, *invstd;  // x_hat = (sx - mean) * invstd for the BatchNorm-backward sums
  float *part;                                            // [S][K][2]
  int M, K, N, Mper;
};

constexpr int FMS = 32;  // rows per stage

// AUX: the epilogue has operands of its own — a residual addend and / or BatchNorm-backward sums against ANOTHER tensor
// than the forward input (the Add's other input, Engine.alias_stats_target) — staged through LDS next to x (expand
// convolutions: K <= 64, the tiles are small).  Without AUX the sums, if any, are taken against the forward input itself,
// which is staged RAW (the input transform is applied where the fragments are read: it is two VALU operations per
// element, and the epilogue needs the raw value for the activation mask and x_hat anyway).
// (four workgroups per CU where the registers allow it without spilling — <= 128 VGPRs, no scratch: these launches wait
// for memory, not for the matrix pipe; 16 -> 96 at 256x256: 2.03 ms at three, 1.81 ms at four; the five-block and AUX
// shapes keep 2 — at 3 they spill 20-150 B per lane)
template <int KB, int NB, bool AUX>
__global__ __launch_bounds__(256, (KB * NB == 5 || AUX) ? 2 : 4) void pw_bwd_fused_kernel(FusedArgs P) {
  constexpr int KP = KB * 32, NP = NB * 32;
  constexpr int LDX = KP + 4, LDD = NP + 4, LDW = KP;
  constexpr int NBLK = KB * NB;
  constexpr int WB = (NBLK + 3) / 4;  // dW blocks per wave (at most)
  constexpr int XB = (KB + 3) / 4;    // dX blocks per wave (at most)
  __shared__ float Xr[FMS * LDX];                 // x of the stage, raw, row-major
  __shared__ float Ds[FMS * LDD];                 // dY of the stage, row-major
  __shared__ float Sx[AUX ? FMS * LDX : 1];       // the other stat tensor
  __shared__ float Ad[AUX ? FMS * LDX : 1];       // the addend
  __shared__ float Wl[NP * LDW];                  // W^T [n][k], zero beyond K / N
  __shared__ float cfx[2 * KP];                   // input transform scale | shift
  __shared__ float cfd[3 * NP];                   // cA | cB | cC

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int bz = blockIdx.x;
  const int mbeg = bz * P.Mper, mend = min(P.M, mbeg + P.Mper);
  const bool two = (P.cA != nullptr);
  const bool sep = AUX && P.sx != nullptr && P.sx != P.x;
  const float hi = (P.x_act == DL3_ACT_RELU6) ? 6.f : __builtin_inff();

  for (int i = tid; i < KP; i += 256) {
    const int k = min(i, P.K - 1);
    cfx[i] = P.xs ? P.xs[k] : 1.f;
    cfx[KP + i] = P.xs ? P.xt[k] : 0.f;
  }
  for (int i = tid; i < NP; i += 256) {
    const int n = min(i, P.N - 1);
    cfd[i] = two ? P.cA[n] : 1.f;
    cfd[NP + i] = two ? P.cB[n] : 0.f;
    cfd[2 * NP + i] = two ? P.cC[n] : 0.f;
  }
  for (int i = tid; i < NP * KP; i += 256) {
    const int n = i / KP, k = i % KP;
    Wl[n * LDW + k] = (n < P.N && k < P.K) ? P.wT[(size_t)n * P.K + k] : 0.f;
  }

  f32x16 accw[WB];
#pragma unroll
  for (int b = 0; b < WB; b++)
#pragma unroll
    for (int r = 0; r < 16; r++) accw[b][r] = 0.f;
  float st1[XB], st2[XB];
#pragma unroll
  for (int b = 0; b < XB; b++) st1[b] = st2[b] = 0.f;

  f32x4 rx[KB], rg[NB], ry[NB], rs[AUX ? KB : 1], ra[AUX ? KB : 1];
  auto load_regs = [&](int m0) {
#pragma unroll
    for (int i = 0; i < KB; i++) {
      const int idx = tid + 256 * i;
      const int mr = idx / (KP / 4), kq = idx % (KP / 4);
      const int row = min(m0 + mr, P.M - 1), k = min(kq * 4, P.K - 4);
      rx[i] = ld4(P.x + (size_t)row * P.ldx + k);
      if constexpr (AUX) {
        if (sep) rs[i] = ld4(P.sx + (size_t)row * P.ldsx + k);
        if (P.add) ra[i] = ld4(P.add + (size_t)row * P.ldadd + k);
      }
    }
#pragma unroll
    for (int i = 0; i < NB; i++) {
      const int idx = tid + 256 * i;
      const int mr = idx / (NP / 4), nq = idx % (NP / 4);
      const int row = min(m0 + mr, P.M - 1), n = min(nq * 4, P.N - 4);
      rg[i] = ld4(P.g + (size_t)row * P.ldg + n);
      if (two) ry[i] = ld4(P.y + (size_t)row * P.ldy + n);
    }
  };
  auto store_lds = [&](int m0) {
    // x stays raw and needs no zeroing: rows beyond the range meet an all-zero dY row, columns beyond K only feed
    // accumulator rows / output columns that are never written (the addresses were clamped: every value is finite)
#pragma unroll
    for (int i = 0; i < KB; i++) {
      const int idx = tid + 256 * i;
      const int mr = idx / (KP / 4), kq = idx % (KP / 4);
      st4(&Xr[mr * LDX + kq * 4], rx[i]);
      if constexpr (AUX) {
        if (sep) st4(&Sx[mr * LDX + kq * 4], rs[i]);
        if (P.add) st4(&Ad[mr * LDX + kq * 4], ra[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < NB; i++) {
      const int idx = tid + 256 * i;
      const int mr = idx / (NP / 4), nq = idx % (NP / 4);
      const bool ok = (m0 + mr) < mend && nq * 4 < P.N;
      f32x4 v = ld4(cfd + nq * 4) * rg[i] + ld4(cfd + 2 * NP + nq * 4);
      if (two) v += ld4(cfd + NP + nq * 4) * ry[i];
      if (!ok) v = splat4(0.f);
      st4(&Ds[mr * LDD + nq * 4], v);
    }
  };

  __syncthreads();  // coefficient vectors and W^T are in LDS
  if (mbeg < mend) load_regs(mbeg);
  for (int m0 = mbeg; m0 < mend; m0 += FMS) {
    store_lds(m0);
    __syncthreads();
    if (m0 + FMS < mend) load_regs(m0 + FMS);  // in flight under this stage's MFMAs

    // ---- dW += T(X)^T . dY: block b of the wave = (kb, nb) = ((wave + 4 b) / NB, (wave + 4 b) % NB)
#pragma unroll
    for (int b = 0; b < WB; b++) {
      const int blk = wave + 4 * b;
      if (blk < NBLK) {
        const int kb = blk / NB, nb = blk % NB;
        const float sk = cfx[kb * 32 + l31], tk = cfx[KP + kb * 32 + l31];
#pragma unroll
        for (int s = 0; s < FMS / 2; s++) {
          const float a = dl3_act(sk * Xr[(2 * s + lhi) * LDX + kb * 32 + l31] + tk, P.x_act);
          const float bb = Ds[(2 * s + lhi) * LDD + nb * 32 + l31];
          accw[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bb, accw[b], 0, 0, 0);
        }
      }
    }
    // ---- dX = dY . W^T, finished per stage.  Block kx of the wave covers columns (4 kx + 3 - wave) * 32 ..: the waves
    // with the fewest dW blocks take them
#pragma unroll
    for (int kx = 0; kx < XB; kx++) {
      const int kbx = 4 * kx + (3 - wave);
      if (kbx < KB) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] = 0.f;
#pragma unroll 8
        for (int s = 0; s < NP / 2; s++) {
          const float a = Ds[l31 * LDD + 2 * s + lhi];
          const float bb = Wl[(2 * s + lhi) * LDW + kbx * 32 + l31];
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bb, acc, 0, 0, 0);
        }
        const int col = kbx * 32 + l31;
        const bool cok = col < P.K;
        const float sc = cfx[col], tc = cfx[KP + col];
        const float mu = (P.part && cok) ? P.mean[col] : 0.f, is = (P.part && cok) ? P.invstd[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int rl = (r & 3) + 8 * (r >> 2) + 4 * lhi;
          const int row = m0 + rl;
          const float raw = Xr[rl * LDX + col];
          float v = acc[r];
          if (P.x_act != DL3_ACT_NONE) {
            const float z = sc * raw + tc;
            v = (z > 0.f && z < hi) ? v : 0.f;
          }
          float xh = raw;
          if constexpr (AUX) {
            if (P.add) v += Ad[rl * LDX + col];
            if (sep) xh = Sx[rl * LDX + col];
          }
          if (cok && row < mend) {
            __builtin_nontemporal_store(v, &P.dx[(size_t)row * P.lddx + col]);
            st1[kx] += v;
            st2[kx] += v * ((xh - mu) * is);
          }
        }
      }
    }
    __syncthreads();  // every wave is done with the stage before it is overwritten
  }

  // ---- weight-gradient slab of this workgroup (all of it: zeros where it saw no rows)
  float *out = P.slab + (size_t)bz * P.K * P.N;
#pragma unroll
  for (int b = 0; b < WB; b++) {
    const int blk = wave + 4 * b;
    if (blk < NBLK) {
      const int kb = blk / NB, nb = blk % NB;
      const int col = nb * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int krow = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        if (krow < P.K && col < P.N) out[(size_t)krow * P.N + col] = accw[b][r];
      }
    }
  }
  if (P.part) {
#pragma unroll
    for (int kx = 0; kx < XB; kx++) {
      const int kbx = 4 * kx + (3 - wave);
      if (kbx < KB) {
        const float a1 = st1[kx] + __shfl_xor(st1[kx], 32, 64), a2 = st2[kx] + __shfl_xor(st2[kx], 32, 64);
        const int col = kbx * 32 + l31;
        if (lhi == 0 && col < P.K) {
          P.part[((size_t)bz * P.K + col) * 2 + 0] = a1;
          P.part[((size_t)bz * P.K + col) * 2 + 1] = a2;
        }
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------
// round 5: the same two products with the matrix work spread evenly over the four waves, W^T in registers
// ---------------------------------------------------------------------------------------------------------------
// dW keeps the round-4 assignment (whole 32x32 blocks: wave w owns blocks w, w + 4 — at most 32 accumulators).  dX:
//   KB >= 3  whole blocks per wave, the waves with the fewest dW blocks first (as in round 4);
//   KB <= 2  block kbx belongs to the GROUP of waves {kbx, kbx + KB, ...} (4 / KB of them): each takes a slice of the
//            block's reduction over N — slices sized at compile time so that dW + dX MFMAs per wave come out even —, the
//            partial 32x32 sums meet in LDS, and every wave of the group finishes (sum in slice order, mask, addend,
//            BatchNorm-backward sums, store) 16 / SPLIT of the block's 16 row registers.
// A wave's slice of W^T never changes: it lives in registers (<= 32 of them), not in LDS.
struct FusedSlices { int lo[4], n[4], maxn; };

constexpr int fused_dw_blocks(int nblk, int w) { return nblk > w ? (nblk - w + 3) / 4 : 0; }

constexpr FusedSlices fused_slices(int KB, int NB) {
  FusedSlices s{};
  const int nblk = KB * NB, TS = 16 * NB, split = 4 / KB;
  int load[4] = {0, 0, 0, 0};
  for (int w = 0; w < 4; w++) { load[w] = 16 * fused_dw_blocks(nblk, w); s.n[w] = 0; s.lo[w] = 0; }
  for (int g = 0; g < KB; g++) {
    for (int t = 0; t < TS; t++) {
      int best = g;
      for (int p = 1; p < split; p++) {
        const int w = g + KB * p;
        if (load[w] <= load[best]) best = w;
      }
      s.n[best]++;
      load[best]++;
    }
    int lo = 0;
    for (int p = 0; p < split; p++) {
      const int w = g + KB * p;
      s.lo[w] = lo;
      lo += s.n[w];
    }
  }
  s.maxn = 0;
  for (int w = 0; w < 4; w++) s.maxn = s.n[w] > s.maxn ? s.n[w] : s.maxn;
  return s;
}

constexpr int cmax(int a, int b) { return a > b ? a : b; }

template <int KB, int NB, bool AUX>
struct Fused2 {
  static constexpr int KP = KB * 32, NP = NB * 32;
  static constexpr int LDX = KP + 4, LDD = NP + 4;
  static constexpr int NBLK = KB * NB;
  static constexpr int WB = (NBLK + 3) / 4;                      // dW blocks per wave (at most)
  static constexpr int SPLIT = KB <= 2 ? 4 / KB : 1;             // waves that share one dX block
  static constexpr int XB = SPLIT > 1 ? 1 : (KB + 3) / 4;        // dX blocks per wave (at most)
  static constexpr int RPW = 16 / SPLIT;                         // row registers of a block a wave finishes
  static constexpr FusedSlices SL = fused_slices(KB <= 2 ? KB : 1, NB);
  static constexpr int NWB = SPLIT > 1 ? SL.maxn : 1;            // W^T registers (SPLIT > 1: the wave's slice)
  static constexpr int LDW = KP;
  // LDS carve (floats)
  static constexpr int O_X = 0;                                  // x of the stage, raw, row-major [32][LDX]
  static constexpr int O_D = O_X + FMS * LDX;                    // dY of the stage [32][LDD]
  static constexpr int O_S = O_D + FMS * LDD;                    // AUX: the other stat tensor
  static constexpr int O_A = O_S + (AUX ? FMS * LDX : 0);        // AUX: the addend
  static constexpr int O_W = O_A + (AUX ? FMS * LDX : 0);        // SPLIT == 1: W^T [NP][LDW], zero beyond K / N
  static constexpr int O_CX = O_W + (SPLIT > 1 ? 0 : NP * LDW);  // input transform scale | shift
  static constexpr int O_CD = O_CX + 2 * KP;                     // cA | cB | cC
  static constexpr int O_P = O_CD + 3 * NP;                      // dX slices [4][16][64]
  static constexpr int O_R = O_P + (SPLIT > 1 ? 4 * 1024 : 0);   // statistic fold [4][32][2]
  static constexpr int TOTAL = O_R + (SPLIT > 1 ? 256 : 0);
};

// OC: workgroups per CU the registers are budgeted for.
// (Round 5 also built the wide-dX shapes — K > 64 — with the finished dX values written back into the x tile in LDS and
// stored 16 bytes per lane by all 256 threads at the top of the next stage, 18 store instructions per stage instead of
// 80 at K = 144: one more barrier per stage, and slower — 144 -> 24 0.785 -> 0.911 ms, 192 -> 32 0.249 -> 0.284 ms,
// whole step 1 287 -> 1 280 img/s, same call, profiles/r05_ab_calls.txt call 4.  Removed.)
template <int KB, int NB, bool AUX, int OC>
__global__ __launch_bounds__(256, OC) void pw_bwd_fused2_kernel(FusedArgs P) {
  using C = Fused2<KB, NB, AUX>;
  constexpr int KP = C::KP, NP = C::NP, LDX = C::LDX, LDD = C::LDD, LDW = C::LDW, NBLK = C::NBLK;
  constexpr int SPLIT = C::SPLIT, XB = C::XB, WB = C::WB, NWB = C::NWB;
  __shared__ float smem[C::TOTAL];
  float *const Xr = smem + C::O_X, *const Ds = smem + C::O_D, *const Sx = smem + C::O_S, *const Ad = smem + C::O_A;
  float *const Wl = smem + C::O_W, *const cfx = smem + C::O_CX, *const cfd = smem + C::O_CD;
  float *const Pp = smem + C::O_P, *const red = smem + C::O_R;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  const int bz = blockIdx.x;
  const int mbeg = bz * P.Mper, mend = min(P.M, mbeg + P.Mper);
  const bool two = (P.cA != nullptr);
  const bool sep = AUX && P.sx != nullptr && P.sx != P.x;
  const float hi = (P.x_act == DL3_ACT_RELU6) ? 6.f : __builtin_inff();

  for (int i = tid; i < KP; i += 256) {
    const int k = min(i, P.K - 1);
    cfx[i] = P.xs ? P.xs[k] : 1.f;
    cfx[KP + i] = P.xs ? P.xt[k] : 0.f;
  }
  for (int i = tid; i < NP; i += 256) {
    const int n = min(i, P.N - 1);
    cfd[i] = two ? P.cA[n] : 1.f;
    cfd[NP + i] = two ? P.cB[n] : 0.f;
    cfd[2 * NP + i] = two ? P.cC[n] : 0.f;
  }

  // ---- this wave's share of dX and its W^T operands (B fragment: lane (k = l & 31, n-pair = l >> 5))
  // Columns beyond K (the last 32-column block of dX may be partial) are not predicated off: their lanes compute, mask and
  // STORE the value of column (c mod valid columns) a second time — same operands, same address, same bits — so that
  // every lane issues every store of a whole stage.  With a fixed number of stores behind the next stage's requests the
  // wait for those is a counted s_waitcnt vmcnt(stores); a branch around a store turns it into vmcnt(0): every wave
  // drains its own dX stores (microseconds under load) once per stage.
  const int klast = P.K - 32 * (KB - 1);  // valid columns of the last block (1..32)
  int x_kbx[XB], x_col[XB], x_lo = 0, x_n = NP / 2, x_part = 0;
  if constexpr (SPLIT > 1) {
    x_kbx[0] = wave % KB;
    x_part = wave / KB;
    // (a four-entry table lookup with a wave-uniform index: scalar selects)
    x_lo = wave == 0 ? C::SL.lo[0] : wave == 1 ? C::SL.lo[1] : wave == 2 ? C::SL.lo[2] : C::SL.lo[3];
    x_n = wave == 0 ? C::SL.n[0] : wave == 1 ? C::SL.n[1] : wave == 2 ? C::SL.n[2] : C::SL.n[3];
  } else {
#pragma unroll
    for (int kx = 0; kx < XB; kx++) x_kbx[kx] = 4 * kx + (3 - wave);
  }
#pragma unroll
  for (int kx = 0; kx < XB; kx++) x_col[kx] = x_kbx[kx] * 32 + (x_kbx[kx] == KB - 1 ? l31 % klast : l31);
  float x_sc[XB], x_tc[XB], x_mu[XB], x_is[XB];  // per-column constants of the wave's dX blocks (filled below, once)
  float wb[NWB];
  if constexpr (SPLIT > 1) {
#pragma unroll
    for (int s = 0; s < NWB; s++) {
      const int n = 2 * (x_lo + s) + lhi;
      const bool ok = s < x_n && n < P.N;
      const float v = P.wT[(size_t)min(n, P.N - 1) * P.K + x_col[0]];  // (clamped address: no branch per load)
      wb[s] = ok ? v : 0.f;
    }
  } else {
    wb[0] = 0.f;
    for (int i = tid; i < NP * KP; i += 256) {
      const int n = i / KP, k = i % KP;
      Wl[n * LDW + k] = (n < P.N && k < P.K) ? P.wT[(size_t)n * P.K + k] : 0.f;
    }
  }

  f32x16 accw[WB];
#pragma unroll
  for (int b = 0; b < WB; b++)
#pragma unroll
    for (int r = 0; r < 16; r++) accw[b][r] = 0.f;
  float st1[XB], st2[XB];
#pragma unroll
  for (int b = 0; b < XB; b++) st1[b] = st2[b] = 0.f;

  f32x4 rx[KB], rg[NB], ry[NB], rs[AUX ? KB : 1], ra[AUX ? KB : 1];
  auto load_regs = [&](int m0) {
#pragma unroll
    for (int i = 0; i < KB; i++) {
      const int idx = tid + 256 * i;
      const int mr = idx / (KP / 4), kq = idx % (KP / 4);
      const int row = min(m0 + mr, P.M - 1), k = min(kq * 4, P.K - 4);
      rx[i] = ld4(P.x + (size_t)row * P.ldx + k);
      if constexpr (AUX) {
        if (sep) rs[i] = ld4(P.sx + (size_t)row * P.ldsx + k);
        if (P.add) ra[i] = ld4(P.add + (size_t)row * P.ldadd + k);
      }
    }
#pragma unroll
    for (int i = 0; i < NB; i++) {
      const int idx = tid + 256 * i;
      const int mr = idx / (NP / 4), nq = idx % (NP / 4);
      const int row = min(m0 + mr, P.M - 1), n = min(nq * 4, P.N - 4);
      rg[i] = ld4(P.g + (size_t)row * P.ldg + n);
      if (two) ry[i] = ld4(P.y + (size_t)row * P.ldy + n);
    }
  };
  auto store_lds = [&](int m0) {
    // x stays raw and needs no zeroing: rows beyond the range meet an all-zero dY row, columns beyond K only feed
    // accumulator rows / output columns that are never written (the addresses were clamped: every value is finite)
#pragma unroll
    for (int i = 0; i < KB; i++) {
      const int idx = tid + 256 * i;
      const int mr = idx / (KP / 4), kq = idx % (KP / 4);
      st4(&Xr[mr * LDX + kq * 4], rx[i]);
      if constexpr (AUX) {
        if (sep) st4(&Sx[mr * LDX + kq * 4], rs[i]);
        if (P.add) st4(&Ad[mr * LDX + kq * 4], ra[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < NB; i++) {
      const int idx = tid + 256 * i;
      const int mr = idx / (NP / 4), nq = idx % (NP / 4);
      const bool ok = (m0 + mr) < mend && nq * 4 < P.N;
      f32x4 v = ld4(cfd + nq * 4) * rg[i] + ld4(cfd + 2 * NP + nq * 4);
      if (two) v += ld4(cfd + NP + nq * 4) * ry[i];
      if (!ok) v = splat4(0.f);
      st4(&Ds[mr * LDD + nq * 4], v);
    }
  };

  // one value of dX: mask, addend, store, BatchNorm-backward sums.  rl: row inside the stage; col: the column the lane
  // computes (its own, or the valid column it doubles); own: the lane's own column exists (statistics).  FULL: every row
  // of the stage exists — the store is unconditional.
  auto finish = [&](auto fullc, float v, int m0, int rl, int col, bool own, float sc, float tc, float mu, float is,
                    float &s1, float &s2) {
    constexpr bool FULL = decltype(fullc)::value;
    const int row = m0 + rl;
    const float raw = Xr[rl * LDX + col];
    if (P.x_act != DL3_ACT_NONE) {
      const float z = sc * raw + tc;
      v = (z > 0.f && z < hi) ? v : 0.f;
    }
    float xh = raw;
    if constexpr (AUX) {
      if (P.add) v += Ad[rl * LDX + col];
      if (sep) xh = Sx[rl * LDX + col];
    }
    if constexpr (FULL) {
      __builtin_nontemporal_store(v, &P.dx[(size_t)row * P.lddx + col]);
      s1 += own ? v : 0.f;
      s2 += own ? v * ((xh - mu) * is) : 0.f;
    } else {
      if (own && row < mend) {
        __builtin_nontemporal_store(v, &P.dx[(size_t)row * P.lddx + col]);
        s1 += v;
        s2 += v * ((xh - mu) * is);
      }
    }
  };

  // the MFMAs and the epilogue of the stage in LDS
  auto stage = [&](auto fullc, int m0) {
    // ---- dW += T(X)^T . dY: block b of the wave = (kb, nb) = ((wave + 4 b) / NB, (wave + 4 b) % NB)
#pragma unroll
    for (int b = 0; b < WB; b++) {
      const int blk = wave + 4 * b;
      if (blk < NBLK) {
        const int kb = blk / NB, nb = blk % NB;
        const float sk = cfx[kb * 32 + l31], tk = cfx[KP + kb * 32 + l31];
#pragma unroll 4
        for (int s = 0; s < FMS / 2; s++) {
          const float a = dl3_act(sk * Xr[(2 * s + lhi) * LDX + kb * 32 + l31] + tk, P.x_act);
          const float bb = Ds[(2 * s + lhi) * LDD + nb * 32 + l31];
          accw[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bb, accw[b], 0, 0, 0);
        }
      }
    }

    // ---- dX = dY . W^T, finished per stage
    if constexpr (SPLIT > 1) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; r++) acc[r] = 0.f;
      const float *dq = Ds + l31 * LDD + 2 * x_lo + lhi;
#pragma unroll
      for (int s = 0; s < NWB; s++)
        if (s < x_n) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(dq[2 * s], wb[s], acc, 0, 0, 0);
      float *pp = Pp + wave * 1024 + lane;
#pragma unroll
      for (int r = 0; r < 16; r++) pp[r * 64] = acc[r];
      __syncthreads();
      // the slices of block kbx, summed in slice order; this wave finishes RPW of the block's 16 row registers
      const int kbx = x_kbx[0], col = x_col[0];
      const float *pq = Pp + kbx * 1024 + lane;
      const bool own = kbx * 32 + l31 < P.K;
      const float sc = x_sc[0], tc = x_tc[0], mu = x_mu[0], is = x_is[0];
#pragma unroll
      for (int rr = 0; rr < C::RPW; rr++) {
        const int r = x_part * C::RPW + rr;
        float v = pq[r * 64];
#pragma unroll
        for (int p = 1; p < SPLIT; p++) v += pq[p * KB * 1024 + r * 64];
        finish(fullc, v, m0, (r & 3) + 8 * (r >> 2) + 4 * lhi, col, own, sc, tc, mu, is, st1[0], st2[0]);
      }
    } else {
#pragma unroll
      for (int kx = 0; kx < XB; kx++) {
        const int kbx = x_kbx[kx];
        if (kbx < KB) {
          f32x16 acc;
#pragma unroll
          for (int r = 0; r < 16; r++) acc[r] = 0.f;
          const int col = x_col[kx];
#pragma unroll 8
          for (int s = 0; s < NP / 2; s++) {
            const float a = Ds[l31 * LDD + 2 * s + lhi];
            const float bb = Wl[(2 * s + lhi) * LDW + col];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bb, acc, 0, 0, 0);
          }
          const bool own = kbx * 32 + l31 < P.K;
          const float sc = x_sc[kx], tc = x_tc[kx], mu = x_mu[kx], is = x_is[kx];
#pragma unroll
          for (int r = 0; r < 16; r++)
            finish(fullc, acc[r], m0, (r & 3) + 8 * (r >> 2) + 4 * lhi, col, own, sc, tc, mu, is, st1[kx], st2[kx]);
        }
      }
    }
  };

  __syncthreads();  // the coefficient vectors (and W^T) are in LDS
  // per-column constants of this wave's dX blocks, ONCE: a global load inside the stage loop (x_mean / x_invstd used to be
  // fetched per stage) is waited for with vmcnt(0) — and the in-order counter then drains the next stage's requests, the
  // very loads the stage is there to hide
#pragma unroll
  for (int kx = 0; kx < XB; kx++) {
    const int col = min(x_col[kx], KP - 1);
    x_sc[kx] = cfx[col];
    x_tc[kx] = cfx[KP + col];
    x_mu[kx] = P.part ? P.mean[min(col, P.K - 1)] : 0.f;
    x_is[kx] = P.part ? P.invstd[min(col, P.K - 1)] : 0.f;
  }
  const int mfull = mbeg + ((mend > mbeg ? mend - mbeg : 0) / FMS) * FMS;  // end of the whole stages
  if (mbeg < mend) load_regs(mbeg);
  auto whole_stage = [&](int m0) {
    store_lds(m0);
    __syncthreads();
    if (m0 + FMS < mend) load_regs(m0 + FMS);  // in flight under this stage's MFMAs
    stage(std::true_type{}, m0);
    __syncthreads();  // every wave is done with the stage before it is overwritten
  };
  for (int m0 = mbeg; m0 < mfull; m0 += FMS) whole_stage(m0);
  if (mfull < mend) {  // the ragged last stage of the launch's last workgroup: every element predicated
    store_lds(mfull);
    __syncthreads();
    stage(std::false_type{}, mfull);
    __syncthreads();
  }

  // ---- weight-gradient slab of this workgroup (all of it: zeros where it saw no rows)
  float *out = P.slab + (size_t)bz * P.K * P.N;
#pragma unroll
  for (int b = 0; b < WB; b++) {
    const int blk = wave + 4 * b;
    if (blk < NBLK) {
      const int kb = blk / NB, nb = blk % NB;
      const int col = nb * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int krow = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        if (krow < P.K && col < P.N) out[(size_t)krow * P.N + col] = accw[b][r];
      }
    }
  }
  // ---- BatchNorm-backward sums of this workgroup: half-waves, then (SPLIT > 1) the waves of a group in group order
  if (P.part) {
    if constexpr (SPLIT > 1) {
      const float a1 = st1[0] + __shfl_xor(st1[0], 32, 64), a2 = st2[0] + __shfl_xor(st2[0], 32, 64);
      if (lhi == 0) {
        red[(wave * 32 + l31) * 2 + 0] = a1;
        red[(wave * 32 + l31) * 2 + 1] = a2;
      }
      __syncthreads();
      const int kbx = x_kbx[0], col = kbx * 32 + l31;
      if (x_part == 0 && lhi == 0 && col < P.K) {
        float b1 = red[(kbx * 32 + l31) * 2 + 0], b2 = red[(kbx * 32 + l31) * 2 + 1];
#pragma unroll
        for (int p = 1; p < SPLIT; p++) {
          b1 += red[((kbx + KB * p) * 32 + l31) * 2 + 0];
          b2 += red[((kbx + KB * p) * 32 + l31) * 2 + 1];
        }
        P.part[((size_t)bz * P.K + col) * 2 + 0] = b1;
        P.part[((size_t)bz * P.K + col) * 2 + 1] = b2;
      }
    } else {
#pragma unroll
      for (int kx = 0; kx < XB; kx++) {
        const int kbx = x_kbx[kx];
        if (kbx < KB) {
          const float a1 = st1[kx] + __shfl_xor(st1[kx], 32, 64), a2 = st2[kx] + __shfl_xor(st2[kx], 32, 64);
          const int col = kbx * 32 + l31;
          if (lhi == 0 && col < P.K) {
            P.part[((size_t)bz * P.K + col) * 2 + 0] = a1;
            P.part[((size_t)bz * P.K + col) * 2 + 1] = a2;
          }
        }
      }
    }
  }
}

// tuning environment, read ONCE per process: the engine sizes workspaces and partial buffers from the _splits /
// _workspace queries when it lowers the plan, and a value that changed before the launch would move S under it
int fused_env(const char *name) {
  const char *e = getenv(name);
  return e ? atoi(e) : -1;
}
// (DL3_FUSED_V / DL3_FUSED_OCC are read per call: they choose a kernel, not a size; a layer that one version supports and
// the other does not fails loudly at launch if the variable changes under a lowered plan)
int fused_version() { return fused_env("DL3_FUSED_V") == 1 ? 1 : 2; }  // 1: the round-4 kernel (A/B aid); else round 5

int fused_rows_per_wg(int M) {
  long want = 2048;  // workgroups (B = 128, 16 -> 96 at 256x256: 1024 -> 3.06 ms, 2048 -> 2.45 ms)
  long stages = dl3_cdiv(M, FMS);
  // every workgroup leaves a K x N slab behind: with few rows (B <= 16) keep at least 512 rows per workgroup, down to two
  // workgroups per CU (round 4, calls 19 / 20)
  {
    long cap = (long)M / 512;
    if (cap < 512) cap = 512;
    if (want > cap) want = cap;
  }
  if (want > stages) want = stages;
  return (int)(dl3_cdiv((int)stages, (int)want) * FMS);
}

inline bool al16(const void *p) { return (((uintptr_t)p) & 15) == 0; }

}  // namespace

extern "C" int dl3_reduce_partials(const float *partial, int P, int n, float *out, void *stream);

// 1: supported; 2: supported including a residual addend / sums against another tensor than x (K <= 64)
extern "C" int dl3_pwconv_bwd_fused_supported(int M, int K, int N) {
  if (M <= 0 || K < 4 || N < 4 || K % 4 || N % 4) return 0;
  if (dl3_cdiv(K, 32) * dl3_cdiv(N, 32) > (fused_version() == 1 ? 5 : 6)) return 0;
  return K <= 64 ? 2 : 1;
}

extern "C" int dl3_pwconv_bwd_fused_splits(int M, int K, int N) {
  if (!dl3_pwconv_bwd_fused_supported(M, K, N)) return 0;
  return dl3_cdiv(M, fused_rows_per_wg(M));
}

extern "C" size_t dl3_pwconv_bwd_fused_workspace(int M, int K, int N) {
  return (size_t)dl3_pwconv_bwd_fused_splits(M, K, N) * K * N * sizeof(float);
}

extern "C" int dl3_pwconv_bwd_fused(const float *x, int ldx, const float *in_scale, const float *in_shift, int in_act,
                                    const float *g, int ldg, const float *yraw, int ldyraw, const float *cA,
                                    const float *cB, const float *cC, const float *wT, float *dw, float *dx, int lddx,
                                    const float *dx_add, int ldadd, const float *stat_x, int ldstatx,
                                    const float *x_mean, const float *x_invstd, float *dstat_partial, int M, int K,
                                    int N, void *workspace, size_t workspace_bytes, void *stream) {
  DL3_CHECK_ARG(M > 0 && K > 0 && N > 0, "pwconv_bwd_fused: non-positive dimension");
  DL3_UNSUPPORTED(!dl3_pwconv_bwd_fused_supported(M, K, N),
                  "pwconv_bwd_fused: K=%d, N=%d must be multiples of 4 with ceil(K/32)*ceil(N/32) <= 6", K, N);
  DL3_CHECK_ARG(x && g && wT && dx && workspace, "pwconv_bwd_fused: null pointer");
  DL3_CHECK_ARG(!cA || (yraw && cB && cC), "pwconv_bwd_fused: cA needs yraw, cB, cC");
  DL3_CHECK_ARG((in_scale == nullptr) == (in_shift == nullptr), "pwconv_bwd_fused: scale/shift must come together");
  DL3_CHECK_ARG(!dstat_partial || (stat_x && x_mean && x_invstd), "pwconv_bwd_fused: dstat needs stat_x, x_mean, x_invstd");
  DL3_CHECK_ARG(ldx >= K && ldx % 4 == 0 && al16(x) && ldg >= N && ldg % 4 == 0 && al16(g) &&
                    (!cA || (ldyraw >= N && ldyraw % 4 == 0 && al16(yraw))) && lddx >= K,
                "pwconv_bwd_fused: operands must be 16-byte aligned with leading dimensions that are multiples of 4");
  if (workspace_bytes < dl3_pwconv_bwd_fused_workspace(M, K, N)) {
    dl3_set_error("pwconv_bwd_fused: workspace %zu < %zu bytes", workspace_bytes, dl3_pwconv_bwd_fused_workspace(M, K, N));
    return DL3_EWORKSPACE;
  }
  FusedArgs A{};
  A.x = x; A.ldx = ldx; A.xs = in_scale; A.xt = in_shift; A.x_act = in_act;
  A.g = g; A.ldg = ldg; A.y = cA ? yraw : nullptr; A.ldy = ldyraw; A.cA = cA; A.cB = cB; A.cC = cC;
  A.wT = wT; A.slab = (float *)workspace; A.dx = dx; A.lddx = lddx; A.add = dx_add; A.ldadd = ldadd;
  A.sx = dstat_partial ? stat_x : nullptr; A.ldsx = ldstatx; A.mean = x_mean; A.invstd = x_invstd;
  A.part = dstat_partial;
  A.M = M; A.K = K; A.N = N; A.Mper = fused_rows_per_wg(M);
  const int S = dl3_cdiv(M, A.Mper);
  hipStream_t st = (hipStream_t)stream;
  const int kb = dl3_cdiv(K, 32), nb = dl3_cdiv(N, 32);
  const bool aux = dx_add != nullptr || (A.sx != nullptr && A.sx != x);
  DL3_UNSUPPORTED(aux && kb > 2, "pwconv_bwd_fused: a residual addend / sums against another tensor need K <= 64 (K=%d)", K);
  DL3_CHECK_ARG(!aux || ((!dx_add || (ldadd % 4 == 0 && al16(dx_add))) && (!A.sx || (ldstatx % 4 == 0 && al16(A.sx)))),
                "pwconv_bwd_fused: dx_add / stat_x must be 16-byte aligned with leading dimensions that are multiples of 4");
  if (fused_version() == 1) {
#define DL3_FUSED(KB_, NB_, AUX_) \
  if (kb == KB_ && nb == NB_ && aux == AUX_) \
    hipLaunchKernelGGL((pw_bwd_fused_kernel<KB_, NB_, AUX_>), dim3(S), dim3(256), 0, st, A)
  DL3_FUSED(1, 1, false); DL3_FUSED(1, 2, false); DL3_FUSED(1, 3, false); DL3_FUSED(1, 4, false); DL3_FUSED(1, 5, false);
  DL3_FUSED(2, 1, false); DL3_FUSED(3, 1, false); DL3_FUSED(4, 1, false); DL3_FUSED(5, 1, false); DL3_FUSED(2, 2, false);
  DL3_FUSED(1, 1, true); DL3_FUSED(1, 2, true); DL3_FUSED(1, 3, true); DL3_FUSED(1, 4, true); DL3_FUSED(1, 5, true);
  DL3_FUSED(2, 1, true); DL3_FUSED(2, 2, true);
#undef DL3_FUSED
  } else {
    // (workgroups per CU the registers are budgeted for: the largest count that does not spill — the five / six-block
    // shapes at three spill 60-150 bytes per lane and ran 1.3-2x slower, profiles/r05_ab_calls.txt call 1 / 2)
#define DL3_FUSED(KB_, NB_, AUX_, OC_) \
  if (kb == KB_ && nb == NB_ && aux == AUX_) \
    hipLaunchKernelGGL((pw_bwd_fused2_kernel<KB_, NB_, AUX_, OC_>), dim3(S), dim3(256), 0, st, A)
  DL3_FUSED(1, 1, false, 4); DL3_FUSED(1, 2, false, 4); DL3_FUSED(1, 3, false, 3); DL3_FUSED(1, 4, false, 3);
  DL3_FUSED(1, 5, false, 2); DL3_FUSED(1, 6, false, 2);
  DL3_FUSED(2, 1, false, 4); DL3_FUSED(2, 2, false, 3); DL3_FUSED(2, 3, false, 2);
  DL3_FUSED(3, 1, false, 4); DL3_FUSED(4, 1, false, 4); DL3_FUSED(5, 1, false, 2); DL3_FUSED(6, 1, false, 2);
  DL3_FUSED(3, 2, false, 3);
  DL3_FUSED(1, 1, true, 3); DL3_FUSED(1, 2, true, 3); DL3_FUSED(1, 3, true, 3); DL3_FUSED(1, 4, true, 3);
  DL3_FUSED(1, 5, true, 2); DL3_FUSED(1, 6, true, 2); DL3_FUSED(2, 1, true, 3); DL3_FUSED(2, 2, true, 3);
  DL3_FUSED(2, 3, true, 2);
#undef DL3_FUSED
  }
  DL3_LAUNCH_CHECK("pwconv_bwd_fused");
  if (!dw) return DL3_OK;  // the caller folds the [S][K][N] slabs (dl3_reduce_partials / _batched)
  return dl3_reduce_partials(A.slab, S, K * N, dw, stream);
}
