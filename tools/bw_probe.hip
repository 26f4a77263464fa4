// bw_probe.hip — development aid: HBM streaming rate of the NHWC channel-slab access pattern the depthwise kernels
// use, against a flat float4 copy.  hipcc --offload-arch=gfx950 -O3 tools/bw_probe.hip -o /tmp/bw_probe && /tmp/bw_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void copy_flat(const f4 *x, f4 *y, size_t n4, int nt) {
  size_t i = (size_t)blockIdx.x * 256 * 4 + threadIdx.x;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    if (i + j * 256 < n4) {
      f4 v = x[i + j * 256];
      if (nt) __builtin_nontemporal_store(v, &y[i + j * 256]); else y[i + j * 256] = v;
    }
  }
}

// block = LQ channel-quads x (256/LQ) pixels, walks ROWS rows of one row phase (rate r) like dw_march_fwd.
template <int LQ, int ROWS>
__global__ __launch_bounds__(256) void copy_slab(const float *x, float *y, int H, int W, int C, int r, int nt) {
  constexpr int PX = 256 / LQ;
  const int tid = threadIdx.x, cq = tid % LQ, pl = tid / LQ;
  const int nxseg = W / PX;
  const int xs = blockIdx.x % nxseg, slab = blockIdx.x / nxseg;
  const int nchunk = (H / r) / ROWS;
  const int a = blockIdx.y / nchunk, ch = blockIdx.y % nchunk, n = blockIdx.z;
  const int c = slab * LQ * 4 + cq * 4, xx = xs * PX + pl;
  const size_t base = ((size_t)n * H * W) * C + c;
  f4 v[ROWS];
#pragma unroll
  for (int k = 0; k < ROWS; k++) {
    const int row = a + (ch * ROWS + k) * r;
    v[k] = *(const f4 *)(x + base + ((size_t)row * W + xx) * C);
  }
#pragma unroll
  for (int k = 0; k < ROWS; k++) {
    const int row = a + (ch * ROWS + k) * r;
    f4 *p = (f4 *)(y + base + ((size_t)row * W + xx) * C);
    if (nt) __builtin_nontemporal_store(v[k], p); else *p = v[k];
  }
}

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(_e), __LINE__); return 1; } } while (0)

template <typename F> float timeit(F f) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; i++) f();
  hipEventRecord(a, 0);
  for (int i = 0; i < 20; i++) f();
  hipEventRecord(b, 0); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms / 20;
}

int main() {
  const int N = 30, H = 64, W = 64, C = 1024;
  const size_t n = (size_t)N * H * W * C;
  float *x, *y;
  CK(hipMalloc(&x, n * 4)); CK(hipMalloc(&y, n * 4));
  CK(hipMemset(x, 0, n * 4)); CK(hipMemset(y, 0, n * 4));
  const double gb = 2.0 * n * 4 / 1e9;
  for (int nt = 0; nt < 2; nt++) {
    float ms = timeit([&] { hipLaunchKernelGGL(copy_flat, dim3((n / 4 + 1023) / 1024), dim3(256), 0, 0, (const f4 *)x, (f4 *)y, n / 4, nt); });
    printf("flat copy            nt=%d  %.3f ms  %.2f TB/s\n", nt, ms, gb / ms);
    for (int r : {1, 4}) {
#define RUN(LQ, ROWS)                                                                                              \
  {                                                                                                                \
    dim3 grid((C / (LQ * 4)) * (W / (256 / LQ)), r * ((H / r) / ROWS), N);                                         \
    float ms = timeit([&] { hipLaunchKernelGGL((copy_slab<LQ, ROWS>), grid, dim3(256), 0, 0, x, y, H, W, C, r, nt); }); \
    printf("slab %4d B x %2d px, %2d rows/blk, rate %d nt=%d  %.3f ms  %.2f TB/s  (%d blocks)\n", LQ * 16, 256 / LQ, ROWS, r, \
           nt, ms, gb / ms, grid.x * grid.y * grid.z);                                                              \
  }
      RUN(8, 4) RUN(8, 8) RUN(8, 16) RUN(16, 8) RUN(16, 16) RUN(32, 8) RUN(32, 16) RUN(64, 8) RUN(64, 16)
    }
  }
  return 0;
}
