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

// sibling instruction classes (mode 4..8: even slots MFMA, odd slots this class; mode 10+class: every wave this class alone)
//   4: ds_read_b32 (320 per iteration)   5: SALU s_add/s_xor (320)   6: v_mov_b32 (320)   7: v_pk_fma_f32 (160 = 320 fp32 FMAs)
//   8: global_load_dword, L2-resident (64 per iteration)
template <int CLS>
__device__ __forceinline__ void class_block(float (&v)[8], float c, const float *lds, const float *gl, int &sacc) {
  if constexpr (CLS == 4) {
#pragma unroll
    for (int u = 0; u < 40; u++)
#pragma unroll
      for (int j = 0; j < 8; j++) v[j] += lds[(threadIdx.x + 64 * (u * 8 + j)) & 4095];
  } else if constexpr (CLS == 5) {
#pragma unroll
    for (int u = 0; u < 320; u++) asm volatile("s_add_u32 %0, %0, 3" : "+s"(sacc));
  } else if constexpr (CLS == 6) {
#pragma unroll
    for (int u = 0; u < 40; u++)
#pragma unroll
      for (int j = 0; j < 8; j++) asm volatile("v_mov_b32 %0, %1" : "=v"(v[j]) : "v"(v[(j + 1) & 7]));
  } else if constexpr (CLS == 7) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 w[4];
#pragma unroll
    for (int j = 0; j < 4; j++) w[j] = f2{v[2 * j], v[2 * j + 1]};
    const f2 cc = {c, c}, hh = {0.5f, 0.5f};
#pragma unroll
    for (int u = 0; u < 40; u++)
#pragma unroll
      for (int j = 0; j < 4; j++) w[j] = __builtin_elementwise_fma(w[j], cc, hh);
#pragma unroll
    for (int j = 0; j < 4; j++) { v[2 * j] = w[j][0]; v[2 * j + 1] = w[j][1]; }
  } else {
#pragma unroll
    for (int u = 0; u < 8; u++)
#pragma unroll
      for (int j = 0; j < 8; j++) v[j] += gl[(threadIdx.x + 256 * (u * 8 + j)) & 65535];
  }
}

template <int CLS>
__global__ __launch_bounds__(256, 2) void probe_cls(float *out, int iters, int alone, float seed, long long *cyc, const float *gl) {
  __shared__ float lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = seed + i;
  __syncthreads();
  f16v acc[5];
  for (int j = 0; j < 5; j++)
    for (int r = 0; r < 16; r++) acc[j][r] = 0.f;
  float v[8];
  for (int j = 0; j < 8; j++) v[j] = seed + j + threadIdx.x * 1e-3f;
  float a = seed + threadIdx.x * 1e-3f, b = seed * 0.5f;
  int sacc = 1;
  const unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));
  const int role = alone ? 1 : (int)(hw & 1u);
  const long long t0 = clock64();
  if (role == 0) {
    for (int i = 0; i < iters; i++) mfma_block(acc, a, b);
  } else {
    for (int i = 0; i < iters; i++) class_block<CLS>(v, b, lds, gl, sacc);
  }
  const long long t1 = clock64();
  float s = (float)sacc;
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
  float *gl; hipMalloc(&gl, 65536 * 4); hipMemset(gl, 0, 65536 * 4);
  const char *names[5] = {"ds_read_b32 x320", "s_add_u32 x320", "v_mov_b32 x320", "v_pk_fma_f32 x160", "global_load_dword x64 (L2)"};
  for (int cls = 4; cls <= 8; cls++) {
    for (int alone = 1; alone >= 0; alone--) {
      hipMemset(cyc, 0, blocks * 4 * 2 * 8);
      hipEventRecord(e0, 0);
      switch (cls) {
        case 4: hipLaunchKernelGGL(probe_cls<4>, dim3(blocks), dim3(256), 0, 0, out, iters, alone, 1.0f, cyc, gl); break;
        case 5: hipLaunchKernelGGL(probe_cls<5>, dim3(blocks), dim3(256), 0, 0, out, iters, alone, 1.0f, cyc, gl); break;
        case 6: hipLaunchKernelGGL(probe_cls<6>, dim3(blocks), dim3(256), 0, 0, out, iters, alone, 1.0f, cyc, gl); break;
        case 7: hipLaunchKernelGGL(probe_cls<7>, dim3(blocks), dim3(256), 0, 0, out, iters, alone, 1.0f, cyc, gl); break;
        default: hipLaunchKernelGGL(probe_cls<8>, dim3(blocks), dim3(256), 0, 0, out, iters, alone, 1.0f, cyc, gl); break;
      }
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      hipMemcpy(h, cyc, blocks * 4 * 2 * 8, hipMemcpyDeviceToHost);
      double c[2] = {0, 0}; int n[2] = {0, 0};
      for (int w = 0; w < blocks * 4; w++) { int r = (int)(h[w * 2 + 1] & 0xff); c[r] += (double)h[w * 2]; n[r]++; }
      printf("%-28s %s: kernel %.3f ms |", names[cls - 4], alone ? "both waves this class " : "next to an MFMA sibling", ms);
      if (n[0]) printf(" MFMA wave %.0f cycles per iteration |", c[0] / n[0] / iters);
      printf(" class wave %.0f cycles per iteration\n", c[1] / n[1] / iters);
    }
  }
  return 0;
}
