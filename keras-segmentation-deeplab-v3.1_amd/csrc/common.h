// common.h — shared host/device helpers for libdl3 (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/dl3.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

void dl3_set_error(const char *fmt, ...);

#define DL3_CHECK_ARG(cond, ...)      \
  do {                                \
    if (!(cond)) {                    \
      dl3_set_error(__VA_ARGS__);     \
      return DL3_EINVAL;              \
    }                                 \
  } while (0)

#define DL3_UNSUPPORTED(cond, ...)    \
  do {                                \
    if (cond) {                       \
      dl3_set_error(__VA_ARGS__);     \
      return DL3_EUNSUPPORTED;        \
    }                                 \
  } while (0)

#define DL3_LAUNCH_CHECK(name)                                              \
  do {                                                                      \
    hipError_t e__ = hipGetLastError();                                     \
    if (e__ != hipSuccess) {                                                \
      dl3_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return DL3_EHIP;                                                      \
    }                                                                       \
  } while (0)

static inline int dl3_cdiv(int a, int b) { return (a + b - 1) / b; }

// ---------------------------------------------------------------- device helpers
// Branch-free activation: act is wave-uniform, so the clamp bounds are scalar selects and the activation is
// one v_max + one v_min — no control flow that would make hipcc serialise neighbouring loads behind s_waitcnt.
//   DL3_ACT_NONE: [-inf, +inf]   DL3_ACT_RELU: [0, +inf] (deeplabv3p.py:72)   DL3_ACT_RELU6: [0, 6] (deeplabv3p.py:181)
__device__ __forceinline__ float dl3_act(float v, int act) {
  const float lo = (act != DL3_ACT_NONE) ? 0.f : -__builtin_inff();
  const float hi = (act == DL3_ACT_RELU6) ? 6.f : __builtin_inff();
#ifdef DL3_ACT_MINMAX
  return fminf(fmaxf(v, lo), hi);
#else
  return __builtin_amdgcn_fmed3f(v, lo, hi);  // v_med3_f32: the clamp in ONE VALU instruction (lo <= hi always)
#endif
}
// derivative mask of the activation at pre-activation z: 1 inside the open interval (lo, hi)
__device__ __forceinline__ float dl3_act_mask(float z, int act) {
  const float lo = (act != DL3_ACT_NONE) ? 0.f : -__builtin_inff();
  const float hi = (act == DL3_ACT_RELU6) ? 6.f : __builtin_inff();
  return (z > lo && z < hi) ? 1.f : 0.f;
}
__device__ __forceinline__ f32x4 dl3_act4(f32x4 v, int act) {
  f32x4 r;
  r.x = dl3_act(v.x, act);
  r.y = dl3_act(v.y, act);
  r.z = dl3_act(v.z, act);
  r.w = dl3_act(v.w, act);
  return r;
}
__device__ __forceinline__ f32x4 dl3_mask4(f32x4 z, int act) {
  f32x4 r;
  r.x = dl3_act_mask(z.x, act);
  r.y = dl3_act_mask(z.y, act);
  r.z = dl3_act_mask(z.z, act);
  r.w = dl3_act_mask(z.w, act);
  return r;
}
__device__ __forceinline__ f32x4 ld4(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }
__device__ __forceinline__ void st4(float *p, f32x4 v) { *reinterpret_cast<f32x4 *>(p) = v; }
// streaming store (global_store_dwordx4 ... nt): activations written once and next read by a LATER kernel should not
// evict the lines the current kernel still re-reads from L2
__device__ __forceinline__ void st4_nt(float *p, f32x4 v) { __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(p)); }
__device__ __forceinline__ f32x4 splat4(float v) {
  f32x4 r = {v, v, v, v};
  return r;
}

// counter-based keep-mask for Dropout: splitmix64 hash of (seed, element index) -> uniform [0,1)
__device__ __forceinline__ float dl3_uniform(unsigned long long seed, unsigned long long idx) {
  unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (idx + 1ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (float)(z >> 40) * (1.0f / 16777216.0f);
}

// Dropout draws a fresh mask every training step: the launch arguments are frozen inside a captured hipGraph, so the
// step number lives in device memory (dl3_counter_add bumps it inside the graph) and is mixed into the seed here.
__device__ __forceinline__ unsigned long long dl3_step_seed(unsigned long long seed, const unsigned long long *step) {
  return step ? seed + step[0] * 0xD1B54A32D192ED03ull : seed;
}

__device__ __forceinline__ float wave_sum(float v) {
  // 64-lane butterfly
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
