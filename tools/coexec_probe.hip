// coexec_probe.hip — development aid (round 3): do VALU / LDS / VMEM instructions of one wave issue while ANOTHER wave
// of the same SIMD keeps the matrix pipe busy with v_mfma_f32_32x32x2_f32?  Grid = 256 CUs x 2 workgroups of 4 waves
// (two waves per SIMD, like the GEMM kernels).  Workgroups alternate roles by blockIdx parity... the hardware places two
// consecutive workgroups on different CUs, so the role is taken from the hardware wave slot instead: waves whose HW_ID
// wave_id is even run role A, odd run role B (mode 3), or every wave runs the same role (modes 0-2).
//   mode 0: every wave: MFMA loop                      -> T_mfma  (two waves share the pipe)
//   mode 1: every wave: VALU loop (v_fma chains)       -> T_valu
//   mode 2: every wave: MFMA and independent VALU interleaved in ONE wave
//   mode 3: even slots MFMA loop, odd slots VALU loop  -> if ~max(T_mfma/2.., T_valu): co-execution; if ~sum: serialised
// hipcc --offload-arch=gfx950 -O3 tools/coexec_probe.hip -o /tmp/coexec_probe && /tmp/coexec_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f16v __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void mfma_block(f16v (&acc)[5], float a, float b) {
#pragma unroll
  for (int u = 0; u < 8; u++)
#pragma unroll
    for (int j = 0; j < 5; j++) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
}
__device__ __forceinline__ void valu_block(float (&v)[8], float c) {
#pragma unroll
  for (int u = 0; u < 40; u++)
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = __builtin_fmaf(v[j], c, 0.5f);  // 320 independent-ish FMAs (8 chains)
}

__global__ __launch_bounds__(256, 2) void probe(float *out, int iters, int mode, float seed, long long *cyc) {
  f16v acc[5];
  for (int j = 0; j < 5; j++)
    for (int r = 0; r < 16; r++) acc[j][r] = 0.f;
  float v[8];
  for (int j = 0; j < 8; j++) v[j] = seed + j + threadIdx.x * 1e-3f;
  float a = seed + threadIdx.x * 1e-3f, b = seed * 0.5f;
  const unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));
  const int role = mode == 3 ? (int)(hw & 1u) : mode;  // 0 mfma, 1 valu, 2 both
  const long long t0 = clock64();
  if (role == 0) {
    for (int i = 0; i < iters; i++) mfma_block(acc, a, b);
  } else if (role == 1) {
    for (int i = 0; i < iters; i++) valu_block(v, b);
  } else {
    for (int i = 0; i < iters; i++) { mfma_block(acc, a, b); valu_block(v, b); }
  }
  const long long t1 = clock64();
  float s = 0.f;
  for (int j = 0; j < 5; j++)
    for (int r = 0; r < 16; r++) s += acc[j][r];
  for (int j = 0; j < 8; j++) s += v[j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) {
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    cyc[w * 2] = t1 - t0;
    cyc[w * 2 + 1] = role | ((long long)(hw & 0xf) << 8);
  }
}

int main() {
  float *out; long long *cyc;
  const int blocks = 512;
  hipMalloc(&out, blocks * 256 * 4);
  hipMalloc(&cyc, blocks * 4 * 2 * 8);
  long long *h = (long long *)malloc(blocks * 4 * 2 * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4000;
  for (int mode = 0; mode < 4; mode++) {
    for (int rep = 0; rep < 2; rep++) {
      hipMemset(cyc, 0, blocks * 4 * 2 * 8);
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), 0, 0, out, iters, mode, 1.0f + rep, cyc);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      hipMemcpy(h, cyc, blocks * 4 * 2 * 8, hipMemcpyDeviceToHost);
      double c[3] = {0, 0, 0}; int n[3] = {0, 0, 0}; int slots[16] = {0};
      for (int w = 0; w < blocks * 4; w++) { int r = (int)(h[w * 2 + 1] & 0xff); c[r] += (double)h[w * 2]; n[r]++; slots[(h[w * 2 + 1] >> 8) & 0xf]++; }
      printf("mode %d: kernel %.3f ms |", mode, ms);
      for (int r = 0; r < 3; r++) if (n[r]) printf(" role %d: %d waves, %.0f cycles per iteration |", r, n[r], c[r] / n[r] / iters);
      printf(" wave slots seen:");
      for (int q = 0; q < 16; q++) if (slots[q]) printf(" %d:%d", q, slots[q]);
      printf("\n");
    }
  }
  printf("per iteration: 40 MFMAs (2560 matrix-pipe cycles) and / or 320 v_fma_f32 (1280 VALU cycles at 4 per wave64 op)\n");
  return 0;
}
