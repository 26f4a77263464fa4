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
  return fminf(fmaxf(v, lo), hi);
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

// ---------------------------------------------------------------- last-arriving-workgroup reductions ("tails")
// Partial rows that another workgroup of the SAME launch will read are published with agent-scope (write-through)
// stores: the 8 XCDs have private, mutually non-coherent L2s (cdna_hip_programming.md, Guideline 16, recipe R1).  The
// reader — the workgroup that drew the last ticket — executes ONE agent-scope acquire (drops its stale L1 / L2 lines)
// and then uses plain vector loads, which the compiler is free to batch.  (Agent-scope atomic LOADS in the fold loops
// were measured first: every one is waited for individually, ~1.5 us each — 0.1-1 ms per launch, 2.5x the whole step.)
__device__ __forceinline__ void dl3_pub(float *p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float dl3_sub(const float *p) { return *p; }  // after the reader's acquire (dl3_last_arrival)

// All threads of the workgroup call this AFTER their dl3_pub stores.  Returns true (to every thread) in exactly one
// workgroup per group: the one that drew ticket `expected - 1`, i.e. after which all `expected` rows are visible.
__device__ __forceinline__ bool dl3_last_arrival(unsigned int *ticket, unsigned expected) {
  __shared__ unsigned int s_last;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's published stores have completed
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned old = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned last = (old + 1u == expected) ? 1u : 0u;
    if (last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
    s_last = last;
  }
  __syncthreads();
  if (s_last == 0u) return false;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // every wave of the reader: stale lines out, then plain loads
  return true;
}

// BatchNorm finalize of channels [c0, c0 + nc) from partial rows part[P][ldc][2], by the whole workgroup (256 threads =
// 8 row lanes x 32 channels; row lane q sums rows q, q+8, ... in double, the lanes are combined in the order 0..7).
// The arithmetic is that of bn_finalize_kernel / bn_bwd_finalize_kernel (bn.hip).
__device__ __forceinline__ void dl3_tail_bn(const dl3_tail &T, const float *part, int P, int ldc, int c0, int nc) {
  __shared__ double s_red[256 * 2];
  const int cl = threadIdx.x & 31, q = threadIdx.x >> 5;
  for (int cb = 0; cb < nc; cb += 32) {
    const int c = c0 + cb + cl;
    const bool cok = (cb + cl) < nc;
    double a1 = 0.0, a2 = 0.0;
    if (cok)
#pragma unroll 8
      for (int p = q; p < P; p += 8) {
        const float *r = part + ((size_t)p * ldc + c) * 2;
        a1 += (double)dl3_sub(r);
        a2 += (double)dl3_sub(r + 1);
      }
    s_red[threadIdx.x * 2] = a1;
    s_red[threadIdx.x * 2 + 1] = a2;
    __syncthreads();
    if (q == 0 && cok) {
      double s1 = 0.0, s2 = 0.0;
      for (int j = 0; j < 8; j++) {
        s1 += s_red[(j * 32 + cl) * 2];
        s2 += s_red[(j * 32 + cl) * 2 + 1];
      }
      if (T.kind == DL3_TAIL_BN_FWD) {
        const double m = s1 / T.count;
        double var = s2 / T.count - m * m;
        if (var < 0.0) var = 0.0;
        const double is = 1.0 / sqrt(var + (double)T.eps);
        const double sc = (double)T.gamma[c] * is;
        T.o[0][c] = (float)sc;
        T.o[1][c] = (float)((double)T.beta[c] - m * sc);
        T.o[2][c] = (float)m;
        T.o[3][c] = (float)is;
        if (T.o[4]) {
          const double mom = (double)T.momentum;
          T.o[4][c] = (float)(mom * T.o[4][c] + (1.0 - mom) * m);
          T.o[5][c] = (float)(mom * T.o[5][c] + (1.0 - mom) * var * T.var_unbias);
        }
      } else {  // DL3_TAIL_BN_BWD: s1 = sum g = dbeta, s2 = sum g * x_hat = dgamma
        if (T.o[4]) T.o[4][c] = (float)s1;
        if (T.o[3]) T.o[3][c] = (float)s2;
        const double a = (double)T.gamma[c] * (double)T.invstd[c];
        if (T.batch_mode) {
          const double b = -a * (double)T.invstd[c] * s2 / T.count;
          T.o[0][c] = (float)a;
          T.o[1][c] = (float)b;
          T.o[2][c] = (float)(-a * s1 / T.count - b * (double)T.mean[c]);
        } else {
          T.o[0][c] = (float)a;
          T.o[1][c] = 0.f;
          T.o[2][c] = 0.f;
        }
      }
    }
    __syncthreads();
  }
}

// out[i] = sum_p part[p * stride + idx(i)] for the n = rows * nc values {r * ldr + c0 + j}: plain fixed-order sums
// (row lanes as above, double accumulation) — the depthwise weight gradient of a channel slab.
__device__ __forceinline__ void dl3_tail_sum(const float *part, int P, size_t pstride, int rows, int ldr, int c0, int nc,
                                             float *out) {
  __shared__ double s_sum[256];
  const int cl = threadIdx.x & 31, q = threadIdx.x >> 5;
  for (int r = 0; r < rows; r++)
    for (int cb = 0; cb < nc; cb += 32) {
      const bool cok = (cb + cl) < nc;
      const size_t idx = (size_t)r * ldr + c0 + cb + cl;
      double a = 0.0;
      if (cok)
#pragma unroll 8
        for (int p = q; p < P; p += 8) a += (double)dl3_sub(part + (size_t)p * pstride + idx);
      s_sum[threadIdx.x] = a;
      __syncthreads();
      if (q == 0 && cok) {
        double s = 0.0;
        for (int j = 0; j < 8; j++) s += s_sum[j * 32 + cl];
        out[idx] = (float)s;
      }
      __syncthreads();
    }
}
