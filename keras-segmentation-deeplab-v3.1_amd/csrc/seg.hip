// seg.hip — the integer work either side of the network: target preparation (the tensor contract of
// SegmentationGenerator.__getitem__, reference utils.py:375-402) and the per-image / per-class pixel counts behind the
// Jaccard and accuracy metrics (reference utils.py:132-157).  Byte / integer kernels: HBM-bound, int atomics only
// (exact and order-independent), one pass over the pixels each.
#include "common.h"

namespace {

constexpr int kMaxClasses = 255;  // labels arrive as uint8 (cv2.imread(path, 0)); bins 0..C with C <= 255

// ---------------------------------------------------------------------------------------
// per-image histogram of clamped labels: y = min(label, C)  (utils.py:377 "y[y>(n_classes-1)] = n_classes")
// grid (chunks, B); LDS histogram per workgroup, then one global int atomic per non-empty bin
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void label_hist_kernel(const T *__restrict__ labels, int HW, int C,
                                                         int *__restrict__ hist) {
  __shared__ int h[kMaxClasses + 1];
  const int b = blockIdx.y;
  for (int i = threadIdx.x; i <= C; i += 256) h[i] = 0;
  __syncthreads();
  const T *src = labels + (size_t)b * HW;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
    const int v = (int)src[i];
    const int y = (v < 0 || v > C - 1) ? C : v;
    atomicAdd(&h[y], 1);
  }
  __syncthreads();
  for (int i = threadIdx.x; i <= C; i += 256)
    if (h[i]) atomicAdd(&hist[b * (C + 1) + i], h[i]);
}

// sklearn.utils.class_weight.compute_class_weight('balanced', unique(y_valid), y_valid) as the reference calls it
// (utils.py:393-396): w_c = n_valid / (n_present * count_c), evaluated in float64 and rounded to float32 when it is
// stored into the float32 SW buffer (np.putmask, utils.py:398); void pixels get weight 0 (utils.py:400).
template <typename T>
__global__ __launch_bounds__(256) void label_apply_kernel(const T *__restrict__ labels, int HW, int C,
                                                          const int *__restrict__ hist, float *__restrict__ Y,
                                                          float *__restrict__ SW) {
  __shared__ float wt[kMaxClasses + 1];
  const int b = blockIdx.y;
  const int *hb = hist + b * (C + 1);
  if (threadIdx.x == 0) {
    long nvalid = 0;
    int npresent = 0;
    for (int c = 0; c < C; c++) {
      nvalid += hb[c];
      npresent += hb[c] > 0;
    }
    for (int c = 0; c < C; c++)
      wt[c] = hb[c] > 0 ? (float)((double)nvalid / ((double)npresent * (double)hb[c])) : 0.f;
    wt[C] = 0.f;
  }
  __syncthreads();
  const T *src = labels + (size_t)b * HW;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
    const int v = (int)src[i];
    const int y = (v < 0 || v > C - 1) ? C : v;
    if (Y) Y[(size_t)b * HW + i] = (float)y;
    if (SW) SW[(size_t)b * HW + i] = wt[y];
  }
}

// ---------------------------------------------------------------------------------------
// counts[b][0][c] = #(true == c), [1][c] = #(pred == c), [2][c] = #(true == c and pred == c)   (utils.py:143-148;
// the union of the reference is true + pred - inter, void pixels are NOT excluded from the predicted side)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void seg_counts_kernel(const int *__restrict__ pred, const float *__restrict__ ytrue,
                                                         int HW, int C, int *__restrict__ counts) {
  __shared__ int h[3 * kMaxClasses];
  const int b = blockIdx.y;
  for (int i = threadIdx.x; i < 3 * C; i += 256) h[i] = 0;
  __syncthreads();
  const size_t base = (size_t)b * HW;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
    const int t = (int)ytrue[base + i];
    const int p = pred[base + i];
    const bool tok = (t >= 0 && t < C), pok = (p >= 0 && p < C);
    if (tok) atomicAdd(&h[t], 1);
    if (pok) atomicAdd(&h[C + p], 1);
    if (tok && t == p) atomicAdd(&h[2 * C + t], 1);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 3 * C; i += 256)
    if (h[i]) atomicAdd(&counts[b * 3 * C + i], h[i]);
}

int chunks_for(int HW) {
  int c = dl3_cdiv(HW, 256 * 16);  // ~16 pixels per thread
  if (c < 1) c = 1;
  if (c > 256) c = 256;
  return c;
}

}  // namespace

extern "C" int dl3_prepare_targets(const void *labels, int label_dtype, int B, int HW, int C, float *Y, float *SW,
                                   int *hist, void *stream) {
  DL3_CHECK_ARG(labels && hist && B > 0 && HW > 0, "prepare_targets: bad argument");
  DL3_CHECK_ARG(C > 0 && C <= kMaxClasses, "prepare_targets: classes must be in 1..255, got %d", C);
  DL3_CHECK_ARG(label_dtype == DL3_LABEL_U8 || label_dtype == DL3_LABEL_I32, "prepare_targets: unknown label dtype %d",
                label_dtype);
  hipStream_t st = (hipStream_t)stream;
  (void)hipMemsetAsync(hist, 0, (size_t)B * (C + 1) * sizeof(int), st);
  dim3 grid(chunks_for(HW), B);
  if (label_dtype == DL3_LABEL_U8) {
    hipLaunchKernelGGL(label_hist_kernel<unsigned char>, grid, dim3(256), 0, st, (const unsigned char *)labels, HW, C,
                       hist);
    if (Y || SW)
      hipLaunchKernelGGL(label_apply_kernel<unsigned char>, grid, dim3(256), 0, st, (const unsigned char *)labels, HW,
                         C, hist, Y, SW);
  } else {
    hipLaunchKernelGGL(label_hist_kernel<int>, grid, dim3(256), 0, st, (const int *)labels, HW, C, hist);
    if (Y || SW)
      hipLaunchKernelGGL(label_apply_kernel<int>, grid, dim3(256), 0, st, (const int *)labels, HW, C, hist, Y, SW);
  }
  DL3_LAUNCH_CHECK("prepare_targets");
  return DL3_OK;
}

extern "C" int dl3_seg_counts(const int *pred, const float *y_true, int B, int HW, int C, int *counts, void *stream) {
  DL3_CHECK_ARG(pred && y_true && counts && B > 0 && HW > 0, "seg_counts: bad argument");
  DL3_CHECK_ARG(C > 0 && C <= kMaxClasses, "seg_counts: classes must be in 1..255, got %d", C);
  hipStream_t st = (hipStream_t)stream;
  (void)hipMemsetAsync(counts, 0, (size_t)B * 3 * C * sizeof(int), st);
  hipLaunchKernelGGL(seg_counts_kernel, dim3(chunks_for(HW), B), dim3(256), 0, st, pred, y_true, HW, C, counts);
  DL3_LAUNCH_CHECK("seg_counts");
  return DL3_OK;
}
