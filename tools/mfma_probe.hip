// mfma_probe.hip — development aid: sustained v_mfma_f32_32x32x2_f32 rate of the whole chip (register-resident
// operands, no memory traffic), i.e. the practical fp32 matrix ceiling under the chip's power management.
// hipcc --offload-arch=gfx950 -O3 tools/mfma_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f16v __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256, 2) void mfma_loop(float *out, int iters, float seed) {
  f16v acc[NACC];
  for (int j = 0; j < NACC; j++)
    for (int r = 0; r < 16; r++) acc[j][r] = 0.f;
  float a = seed + threadIdx.x * 1e-3f, b = seed * 0.5f + threadIdx.x * 2e-3f;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
#pragma unroll
      for (int j = 0; j < NACC; j++) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
      a += 1e-6f;
    }
  }
  float s = 0.f;
  for (int j = 0; j < NACC; j++)
    for (int r = 0; r < 16; r++) s += acc[j][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  float *out;
  const int blocks = 256 * 2 * 4;
  hipMalloc(&out, blocks * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; rep++) {
    for (int iters : {2000, 20000}) {
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(mfma_loop<5>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f + rep);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double flop = (double)blocks * 4 * iters * 8 * 5 * 4096.0;
      printf("5 accumulators, %d iters: %.3f ms  %.1f TFLOP/s\n", iters, ms, flop / ms / 1e9);
    }
  }
  return 0;
}
