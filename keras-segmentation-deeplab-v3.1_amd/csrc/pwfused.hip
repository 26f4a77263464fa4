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
// K, N multiples of 4, ceil(K/32) * ceil(N/32) <= 5 (the accumulators of one workgroup).
#include <stdlib.h>

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

int fused_rows_per_wg(int M) {
  const char *e = getenv("DL3_FUSED_WGS");  // tuning aid: target number of workgroups
  long want = e && atol(e) > 0 ? atol(e) : 2048;  // (B = 128, 16 -> 96 at 256x256: 1024 -> 3.06 ms, 2048 -> 2.45 ms)
  long stages = dl3_cdiv(M, FMS);
  // every workgroup leaves a K x N slab behind: with few rows (B <= 16) keep at least DL3_FUSED_MINROWS rows per workgroup,
  // down to two workgroups per CU
  if (!(e && atol(e) > 0)) {
    const char *r = getenv("DL3_FUSED_MINROWS");
    const long minrows = r && atol(r) > 0 ? atol(r) : 512;
    long cap = (long)M / minrows;
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
  if (dl3_cdiv(K, 32) * dl3_cdiv(N, 32) > 5) return 0;
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
                  "pwconv_bwd_fused: K=%d, N=%d must be multiples of 4 with ceil(K/32)*ceil(N/32) <= 5", K, N);
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
#define DL3_FUSED(KB_, NB_, AUX_) \
  if (kb == KB_ && nb == NB_ && aux == AUX_) \
    hipLaunchKernelGGL((pw_bwd_fused_kernel<KB_, NB_, AUX_>), dim3(S), dim3(256), 0, st, A)
  DL3_FUSED(1, 1, false); DL3_FUSED(1, 2, false); DL3_FUSED(1, 3, false); DL3_FUSED(1, 4, false); DL3_FUSED(1, 5, false);
  DL3_FUSED(2, 1, false); DL3_FUSED(3, 1, false); DL3_FUSED(4, 1, false); DL3_FUSED(5, 1, false); DL3_FUSED(2, 2, false);
  DL3_FUSED(1, 1, true); DL3_FUSED(1, 2, true); DL3_FUSED(1, 3, true); DL3_FUSED(1, 4, true); DL3_FUSED(1, 5, true);
  DL3_FUSED(2, 1, true); DL3_FUSED(2, 2, true);
#undef DL3_FUSED
  DL3_LAUNCH_CHECK("pwconv_bwd_fused");
  if (!dw) return DL3_OK;  // the caller folds the [S][K][N] slabs (dl3_reduce_partials / _batched)
  return dl3_reduce_partials(A.slab, S, K * N, dw, stream);
}
