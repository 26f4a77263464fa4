// f32_probe.hip — where does the f32 stream GEMM (pw_gemm_stream_kernel, 128x160 tile, KT = 16) lose the matrix pipe?
// The K-tile loop rebuilt from its parts on the whole chip (512 workgroups x 4 waves, two per SIMD, 40 f32 MFMAs of 64
// cycles per K-tile and wave), each stage timed:
//   0  MFMAs only (operands fixed in registers)
//   1  + B operand read from LDS, one k-step ahead (5 ds_read_b32 per k-step)
//   2  + one barrier per K-tile            3  + a second barrier per K-tile (as the kernel has)
//   4  + weight tile staged global -> registers -> LDS every K-tile (2.5 float4 per thread)
//   5  + A stream (2 x 16-byte loads per lane and K-tile, one K-tile ahead) and the BN + ReLU6 transform
//   6  = 5, but the loop is cut into row tiles of TILE_K K-tiles with the tile epilogue (80 nt stores per lane, BN sums)
//        and prologue (accumulators zeroed, first K-tile loaded, waited for, staged) between them
//   7  = 6 with the next tile's first K-tile requested before the epilogue     8  = 7 without the epilogue's stores
// usage: hipcc -O3 --offload-arch=gfx950 tools/f32_probe.hip -o /tmp/f32_probe && /tmp/f32_probe [K-tiles per row tile [K-tiles per workgroup]]
//        -DPROBE_TN=1 -DPROBE_WN=4: the 32x128 small-M tile (B=2: one row tile of 60 K-tiles per workgroup: `f32_probe 60 60`)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifndef PROBE_TN
#define PROBE_TN 5
#endif
#ifndef PROBE_WN
#define PROBE_WN 1  // waves side by side along N (4: the 32-row small-M tile, every wave streams the same 32 A rows)
#endif
constexpr int TN = PROBE_TN, WN = PROBE_WN, KT = 16, KH = 8, BN = 32 * TN * WN, LDB = BN, NB = (KT * BN / 4 + 255) / 256;
constexpr int MFMA_PER_KT = 8 * TN;

template <int STAGE>
__global__ __launch_bounds__(256, 2) void probe(const float *__restrict__ a, int lda, const float *__restrict__ b,
                                                const float *__restrict__ coef, float *__restrict__ c, float *__restrict__ out,
                                                int ktiles, int rows, int tile_k) {
  __shared__ float lds[2 * KT * LDB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
  f32x16 acc[TN];
#pragma unroll
  for (int j = 0; j < TN; j++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[j][r] = 0.f;
  for (int i = tid; i < 2 * KT * LDB; i += 256) lds[i] = b[i % (KT * BN)];
  __syncthreads();
  const int wn = WN == 1 ? 0 : wave;
  int row = (WN == 1 ? blockIdx.x * 128 + wave * 32 + l31 : blockIdx.x * 32 + l31) % rows;
  const float *arow = a + (size_t)row * lda;
  f32x4 an[2], ac[2], rb[NB];
  an[0] = an[1] = ac[0] = ac[1] = f32x4{1.f, 2.f, 3.f, 4.f};
  float st1[TN], st2[TN];
#pragma unroll
  for (int j = 0; j < TN; j++) st1[j] = st2[j] = 0.f;

  auto load_A = [&](int kt) {
#pragma unroll
    for (int j = 0; j < 2; j++) an[j] = *(const f32x4 *)(arow + (size_t)(kt % 60) * KT + KH * lhi + 4 * j);
  };
  auto load_B = [&](int kt) {
#pragma unroll
    for (int i = 0; i < NB; i++) {
      const int idx = tid + 256 * i;
      if (idx < KT * BN / 4) rb[i] = *(const f32x4 *)(b + (size_t)(kt % 8) * KT * BN + idx * 4);
    }
  };
  auto store_B = [&](float *Bs) {
#pragma unroll
    for (int i = 0; i < NB; i++) {
      const int idx = tid + 256 * i;
      if (idx < KT * BN / 4) *(f32x4 *)(Bs + idx * 4) = rb[i];
    }
  };
  auto transform = [&]() {
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const f32x4 fa = *(const f32x4 *)(coef + 4 * j), fc = *(const f32x4 *)(coef + 64 + 4 * j);
      f32x4 v = fa * an[j] + fc;
#pragma unroll
      for (int e = 0; e < 4; e++) v[e] = fminf(fmaxf(v[e], 0.f), 6.f);
      an[j] = v;
    }
  };
  auto epilogue = [&](int tile) {
    float *pc = c + ((size_t)((blockIdx.x * 7 + tile) % 4096) * 128 + (WN == 1 ? wave * 32 : 0)) * BN + wn * TN * 32;
    const unsigned lo = 4 * lhi * BN + l31;
#pragma unroll
    for (int j = 0; j < TN; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const float v = acc[j][r];
        if (STAGE != 8) __builtin_nontemporal_store(v, &(pc + (size_t)((r & 3) + 8 * (r >> 2)) * BN + j * 32)[lo]);
        st1[j] += v;
        st2[j] += v * v;
      }
  };

  if (STAGE >= 5) load_A(0);
  int tile = 0;
  for (int kt = 0; kt < ktiles; ++kt) {
    const float *Bs = lds + (kt & 1) * KT * LDB;
    if (STAGE >= 5) load_A(kt + 1);
    if (STAGE >= 4) load_B(kt + 1);
    float bf[2][TN];
#pragma unroll
    for (int j = 0; j < TN; j++) bf[0][j] = STAGE >= 1 ? Bs[(KH * lhi) * LDB + (wn * TN + j) * 32 + l31] : 1.f + j;
#pragma unroll
    for (int s_ = 0; s_ < KH; ++s_) {
      const int cur = s_ & 1, nxt = cur ^ 1;
      if (s_ + 1 < KH) {
#pragma unroll
        for (int j = 0; j < TN; j++) bf[nxt][j] = STAGE >= 1 ? Bs[(KH * lhi + s_ + 1) * LDB + (wn * TN + j) * 32 + l31] : 2.f + j;
      }
#pragma unroll
      for (int j = 0; j < TN; j++) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[s_ >> 2][s_ & 3], bf[cur][j], acc[j], 0, 0, 0);
      if (s_ == KH - 3) {
        if (STAGE >= 4) store_B(lds + ((kt + 1) & 1) * KT * LDB);
        if (STAGE >= 5) transform();
      }
    }
    if (STAGE >= 5) { ac[0] = an[0]; ac[1] = an[1]; }
    if (STAGE >= 2) __syncthreads();
    if (STAGE >= 3) __syncthreads();
    if (STAGE >= 6 && (kt + 1) % tile_k == 0) {
      if (STAGE >= 7) {  // the next tile's first K-tile is requested BEFORE the epilogue
        row = (row + 128 * 512) % rows;
        arow = a + (size_t)row * lda;
        load_A(kt + 1);
        load_B(kt + 1);
      }
      epilogue(tile++);
      if (STAGE < 7) {
        row = (row + 128 * 512) % rows;
        arow = a + (size_t)row * lda;
      }
#pragma unroll
      for (int j = 0; j < TN; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[j][r] = 0.f;
      if (STAGE < 7) {
        load_A(kt + 1);  // the tile starts with an exposed request, as in the kernel
        load_B(kt + 1);
      }
      __syncthreads();
      store_B(lds + ((kt + 1) & 1) * KT * LDB);
      transform();
      ac[0] = an[0]; ac[1] = an[1];
      __syncthreads();
    }
  }
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < TN; j++) {
    sum += st1[j] + st2[j];
#pragma unroll
    for (int r = 0; r < 16; r++) sum += acc[j][r];
  }
  out[blockIdx.x * 256 + tid] = sum;
}

template <int STAGE>
static void run(const float *a, int lda, const float *b, const float *coef, float *c, float *out, int ktiles, int rows,
                int tile_k) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int i = 0; i < 2; i++) hipLaunchKernelGGL(probe<STAGE>, dim3(512), dim3(256), 0, 0, a, lda, b, coef, c, out, ktiles, rows, tile_k);
  hipEventRecord(e0, 0);
  const int reps = 5;
  for (int i = 0; i < reps; i++) hipLaunchKernelGGL(probe<STAGE>, dim3(512), dim3(256), 0, 0, a, lda, b, coef, c, out, ktiles, rows, tile_k);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  const double mfma = 512.0 * 4 * ktiles * MFMA_PER_KT;
  const double ideal_ms = mfma * 64 / (1024.0 * 2.4e9) * 1e3;
  printf("stage %d: %.3f ms  (matrix pipe %.0f %% of 64-cycle issue at 2.4 GHz; %.1f TFLOP/s f32)\n", STAGE, ms,
         100.0 * ideal_ms / ms, mfma * 4096.0 / ms / 1e9);
}

int main(int argc, char **argv) {
  const int rows = 524288, lda = 960, tile_k = argc > 1 ? atoi(argv[1]) : 10, ktiles = argc > 2 ? atoi(argv[2]) : 480;
  float *a, *b, *coef, *c, *out;
  hipMalloc(&a, (size_t)rows * lda * 4);
  hipMalloc(&b, (size_t)9 * KT * BN * 4);
  hipMalloc(&coef, 1024);
  hipMalloc(&c, (size_t)4096 * 128 * 640 * 4);
  hipMalloc(&out, 512 * 256 * 4);
  hipMemset(a, 0, (size_t)rows * lda * 4);
  hipMemset(b, 0, (size_t)9 * KT * BN * 4);
  hipMemset(coef, 0, 1024);
  printf("tile 32*%d rows x %d columns, TN = %d; row tiles of %d K-tiles (K = %d), %d K-tiles per workgroup\n", WN == 1 ? 4 : 1, BN, TN, tile_k, tile_k * KT, ktiles);
  run<0>(a, lda, b, coef, c, out, ktiles, rows, tile_k);
  run<1>(a, lda, b, coef, c, out, ktiles, rows, tile_k);
  run<2>(a, lda, b, coef, c, out, ktiles, rows, tile_k);
  run<3>(a, lda, b, coef, c, out, ktiles, rows, tile_k);
  run<4>(a, lda, b, coef, c, out, ktiles, rows, tile_k);
  run<5>(a, lda, b, coef, c, out, ktiles, rows, tile_k);
  run<6>(a, lda, b, coef, c, out, ktiles, rows, tile_k);
  run<7>(a, lda, b, coef, c, out, ktiles, rows, tile_k);
  run<8>(a, lda, b, coef, c, out, ktiles, rows, tile_k);
  hipDeviceSynchronize();
  return 0;
}
