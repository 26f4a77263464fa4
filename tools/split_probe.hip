// split_probe.hip — which stage of the split-math main loop costs the matrix pipe its throughput?
// Builds the K-tile loop of pw_gemm_stream_kernel<MATH=1> (TN = 5, KT = 32, 60 bf16 MFMAs per K-tile and wave) up from
// its parts and times each stage on the whole chip (512 workgroups x 4 waves, two per SIMD):
//   0  MFMAs only, operands fixed in registers                      -> the pipe's ceiling at this occupancy
//   1  + B fragments read from LDS every K-tile (ds_read_b128)
//   2  + the K-tile barrier
//   3  + the weight tile re-filled by DMA (global_load_lds) every K-tile
//   4  + A stream: 4 x 16-byte global loads per lane and K-tile, one K-tile ahead, NO transform / split (bit-cast)
//   5  + operand transform and split3 (the real VALU work)
//   6  = 5 with a raw s_barrier behind a COUNTED s_waitcnt vmcnt(4): the A request survives the barrier
//  10  = 3 with all 15 B fragments of a k-step requested before its first MFMA (60 registers)
//   8  = 3 but the DMA is never waited for (raw s_barrier)      9  = 3 with half the DMA bytes
//   7  = 6 re-ordered: [A(kt+1): transform, split] [DMA] [request A(kt+2)] [MFMAs] [vmcnt(4); s_barrier]
// usage: hipcc -O3 --offload-arch=gfx950 tools/split_probe.hip -o /tmp/split_probe && /tmp/split_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

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
  for (int q = 0; q < 4; q++) {
    h[q] = __builtin_amdgcn_perm(hb[2 * q + 1], hb[2 * q], 0x07060302u);
    m[q] = __builtin_amdgcn_perm(mb[2 * q + 1], mb[2 * q], 0x07060302u);
    l[q] = __builtin_amdgcn_perm(lb[2 * q + 1], lb[2 * q], 0x07060302u);
  }
}

constexpr int TN = 5, NS = 2, BQ = NS * 3 * TN * 64;  // 16-byte pieces of one K-tile's weight tile

template <int STAGE>
__global__ __launch_bounds__(256, 2) void probe(const float *__restrict__ a, int lda, const u32x4 *__restrict__ bp,
                                                const float *__restrict__ coef, float *__restrict__ out, int ktiles,
                                                int rows) {
  __shared__ float lds[2 * BQ * 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
  f32x16 acc[TN];
#pragma unroll
  for (int j = 0; j < TN; j++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[j][r] = 0.f;
  for (int i = tid; i < 2 * BQ; i += 256) ((u32x4 *)lds)[i] = bp[i % BQ];
  __syncthreads();
  const int row = (blockIdx.x * 128 + wave * 32 + l31) % rows;
  const float *arow = a + (size_t)row * lda + 16 * lhi;
  f32x4 an[4];
  u32x4 ch[NS], cm[NS], cl[NS], nh[NS], nm[NS], nl[NS];
#pragma unroll
  for (int s = 0; s < NS; s++) {
    ch[s] = bp[lane]; cm[s] = bp[64 + lane]; cl[s] = bp[128 + lane];
    nh[s] = ch[s]; nm[s] = cm[s]; nl[s] = cl[s];
  }
  if (STAGE >= 4) {
#pragma unroll
    for (int j = 0; j < 4; j++) an[j] = *(const f32x4 *)(arow + 4 * j);
  }
  if (STAGE == 7) __syncthreads();
  u32x4 bh[TN], bm[TN], bl[TN];
#pragma unroll
  for (int j = 0; j < TN; j++) { bh[j] = bp[(j) * 64 + lane]; bm[j] = bp[(TN + j) * 64 + lane]; bl[j] = bp[(2 * TN + j) * 64 + lane]; }

  auto mfma_step = [&](const float *Bs, int ks) {
    const u32x4 *Bq = (const u32x4 *)Bs + (ks * 3 * TN) * 64 + lane;
    if (STAGE == 10) {  // all 15 fragments of the k-step requested before the first MFMA (60 registers)
      u32x4 fh[TN], fm[TN], fl[TN];
#pragma unroll
      for (int j = 0; j < TN; j++) { fh[j] = Bq[j * 64]; fm[j] = Bq[(TN + j) * 64]; fl[j] = Bq[(2 * TN + j) * 64]; }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < TN; j++) {
        const bf16x8 xh = __builtin_bit_cast(bf16x8, fh[j]), xm = __builtin_bit_cast(bf16x8, fm[j]), xl = __builtin_bit_cast(bf16x8, fl[j]);
        const bf16x8 ah = __builtin_bit_cast(bf16x8, ch[ks]), am = __builtin_bit_cast(bf16x8, cm[ks]),
                     al = __builtin_bit_cast(bf16x8, cl[ks]);
        f32x16 c = acc[j];
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, xh, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, xl, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, xm, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, xh, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, xm, c, 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, xh, c, 0, 0, 0);
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < TN; j++) {
      u32x4 h_ = bh[j], m_ = bm[j], l_ = bl[j];
      if (STAGE >= 1) { h_ = Bq[j * 64]; m_ = Bq[(TN + j) * 64]; l_ = Bq[(2 * TN + j) * 64]; }
      const bf16x8 xh = __builtin_bit_cast(bf16x8, h_), xm = __builtin_bit_cast(bf16x8, m_), xl = __builtin_bit_cast(bf16x8, l_);
      const bf16x8 ah = __builtin_bit_cast(bf16x8, ch[ks]), am = __builtin_bit_cast(bf16x8, cm[ks]),
                   al = __builtin_bit_cast(bf16x8, cl[ks]);
      f32x16 c = acc[j];
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, xh, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, xl, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, xm, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, xh, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, xm, c, 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, xh, c, 0, 0, 0);
    }
  };

  if (STAGE == 7) {
    // order that never uses an ordinary load result while a DMA is in flight (hipcc would drain everything there) and
    // keeps the DMA OLDER than the A request (so a counted wait at the barrier covers exactly the DMA):
    //   [A(kt+1): transform, split] [DMA B(kt+1)] [request A(kt+2)] [MFMAs of kt] [vmcnt(4); s_barrier]
    for (int kt = 0; kt < ktiles; ++kt) {
      const float *Bs = lds + (kt & 1) * BQ * 4;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const f32x4 fa = *(const f32x4 *)(coef + 4 * j), fc = *(const f32x4 *)(coef + 64 + 4 * j);
        f32x4 v = fa * an[j] + fc;
#pragma unroll
        for (int e = 0; e < 4; e++) v[e] = fminf(fmaxf(v[e], 0.f), 6.f);
        an[j] = v;
      }
      split3(an[0], an[1], nh[0], nm[0], nl[0]);
      split3(an[2], an[3], nh[1], nm[1], nl[1]);
      __builtin_amdgcn_sched_barrier(0);
      for (int pc = __builtin_amdgcn_readfirstlane(wave); pc < NS * 3 * TN; pc += 4) {
        const char *src = (const char *)bp + ((size_t)((kt + 1) % 8) * BQ + pc * 64 + lane) * 16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)(lds + (((kt + 1) & 1) * BQ + pc * 64) * 4), 16, 0, 0);
      }
      const float *nx = arow + (size_t)((kt + 2) % 16) * 32;
#pragma unroll
      for (int j = 0; j < 4; j++) an[j] = *(const f32x4 *)(nx + 4 * j);
      __builtin_amdgcn_sched_barrier(0);
      mfma_step(Bs, 0);
      mfma_step(Bs, 1);
      asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int s = 0; s < NS; s++) { ch[s] = nh[s]; cm[s] = nm[s]; cl[s] = nl[s]; }
    }
  } else
  for (int kt = 0; kt < ktiles; ++kt) {
    const float *Bs = lds + (kt & 1) * BQ * 4;
    if (STAGE >= 3) {
      for (int pc = __builtin_amdgcn_readfirstlane(wave); pc < (STAGE == 9 ? NS * 3 * TN / 2 : NS * 3 * TN); pc += 4) {
        const char *src = (const char *)bp + ((size_t)((kt + 1) % 8) * BQ + pc * 64 + lane) * 16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)(lds + (((kt + 1) & 1) * BQ + pc * 64) * 4), 16, 0, 0);
      }
    }
    mfma_step(Bs, 0);
    if (STAGE >= 4 && STAGE < 8) {
      if (STAGE >= 5) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const f32x4 fa = *(const f32x4 *)(coef + 4 * j), fc = *(const f32x4 *)(coef + 64 + 4 * j);
          f32x4 v = fa * an[j] + fc;
#pragma unroll
          for (int e = 0; e < 4; e++) v[e] = fminf(fmaxf(v[e], 0.f), 6.f);
          an[j] = v;
        }
        split3(an[0], an[1], nh[0], nm[0], nl[0]);
        split3(an[2], an[3], nh[1], nm[1], nl[1]);
      } else {
        nh[0] = __builtin_bit_cast(u32x4, an[0]); nm[0] = __builtin_bit_cast(u32x4, an[1]);
        nl[0] = __builtin_bit_cast(u32x4, an[2]); nh[1] = __builtin_bit_cast(u32x4, an[3]);
        nm[1] = nh[0]; nl[1] = nm[0];
      }
      const float *nx = arow + (size_t)((kt + 2) % 16) * 32;
#pragma unroll
      for (int j = 0; j < 4; j++) an[j] = *(const f32x4 *)(nx + 4 * j);
      if (STAGE >= 6) __builtin_amdgcn_sched_barrier(0);  // the request stays HERE, a half K-tile before the barrier
    }
    mfma_step(Bs, 1);
    if (STAGE == 8) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // DMA never waited for (the probe does not care what it reads)
      __builtin_amdgcn_s_barrier();
    } else if (STAGE == 9 || STAGE == 10) {
      __syncthreads();
    } else if (STAGE >= 6) {
      // the weight DMA (older) must have landed before the barrier; the four A loads (younger) stay in flight across it
      asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    } else if (STAGE >= 2) __syncthreads();
    if (STAGE >= 4 && STAGE < 8) {
#pragma unroll
      for (int s = 0; s < NS; s++) { ch[s] = nh[s]; cm[s] = nm[s]; cl[s] = nl[s]; }
    }
  }
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < TN; j++)
#pragma unroll
    for (int r = 0; r < 16; r++) sum += acc[j][r];
  out[blockIdx.x * 256 + tid] = sum;
}

template <int STAGE>
static void run(const float *a, int lda, const u32x4 *bp, const float *coef, float *out, int ktiles, int rows) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int i = 0; i < 2; i++) hipLaunchKernelGGL(probe<STAGE>, dim3(512), dim3(256), 0, 0, a, lda, bp, coef, out, ktiles, rows);
  hipEventRecord(e0, 0);
  const int reps = 5;
  for (int i = 0; i < reps; i++) hipLaunchKernelGGL(probe<STAGE>, dim3(512), dim3(256), 0, 0, a, lda, bp, coef, out, ktiles, rows);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  const double mfma = 512.0 * 4 * ktiles * 60;                     // bf16 MFMA instructions per launch
  const double ideal_ms = mfma * 32 / (1024.0 * 2.4e9) * 1e3;      // 32 cycles each, 1024 SIMDs, 2.4 GHz
  printf("stage %d: %.3f ms  (matrix pipe %.0f %% of 32-cycle issue at 2.4 GHz; %.0f TFLOP/s bf16)\n", STAGE, ms,
         100.0 * ideal_ms / ms, mfma * 32768.0 / ms / 1e9);
}

int main() {
  const int rows = 65536, lda = 1024, ktiles = 400;
  float *a, *coef, *out;
  u32x4 *bp;
  hipMalloc(&a, (size_t)rows * lda * 4);
  hipMalloc(&coef, 1024);
  hipMalloc(&out, 512 * 256 * 4);
  hipMalloc(&bp, (size_t)9 * BQ * 16);
  hipMemset(a, 0, (size_t)rows * lda * 4);
  hipMemset(coef, 0, 1024);
  hipMemset(bp, 0, (size_t)9 * BQ * 16);
  run<0>(a, lda, bp, coef, out, ktiles, rows);
  run<1>(a, lda, bp, coef, out, ktiles, rows);
  run<2>(a, lda, bp, coef, out, ktiles, rows);
  run<3>(a, lda, bp, coef, out, ktiles, rows);
  run<4>(a, lda, bp, coef, out, ktiles, rows);
  run<5>(a, lda, bp, coef, out, ktiles, rows);
  run<6>(a, lda, bp, coef, out, ktiles, rows);
  run<7>(a, lda, bp, coef, out, ktiles, rows);
  run<8>(a, lda, bp, coef, out, ktiles, rows);
  run<9>(a, lda, bp, coef, out, ktiles, rows);
  run<10>(a, lda, bp, coef, out, ktiles, rows);
  hipDeviceSynchronize();
  return 0;
}
