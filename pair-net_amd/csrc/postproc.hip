// Device-side triplet / panoptic post-processing (pairnet_head.py:788-924), replacing
// the reference's host loops and .item() syncs.
#include "common.h"

// One wave per row.  prob = exp(x - max) / sum over all C logits; the label is the
// argmax of prob over the first C-1 columns (first index on ties), score its prob.
__global__ __launch_bounds__(256) void k_cls_argmax(const float* __restrict__ logits,
                                                    int64_t* __restrict__ label,
                                                    float* __restrict__ score, int64_t rows,
                                                    int C) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* x = logits + row * C;
  float v[4];
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = lane + 64 * j;
    v[j] = (c < C) ? x[c] : -INFINITY;
    mx = fmaxf(mx, v[j]);
  }
  mx = wave_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    v[j] = (lane + 64 * j < C) ? expf(v[j] - mx) : 0.f;
    sum += v[j];
  }
  sum = wave_sum(sum);
  float best = -1.f;
  int bi = 0x7fffffff;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = lane + 64 * j;
    if (c < C - 1) {
      const float pr = v[j] / sum;
      if (pr > best) { best = pr; bi = c; }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
  }
  if (lane == 0) { label[row] = bi; score[row] = best; }
}

extern "C" int pn_cls_argmax_f32(const float* logits, int64_t* label, float* score,
                                 int64_t rows, int C, void* stream) {
  if (!logits || !label || !score || rows <= 0 || C < 2 || C > 256) return PN_BAD_ARG;
  hipLaunchKernelGGL(k_cls_argmax, dim3(pn_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream,
                     logits, label, score, rows, C);
  return PN_LAUNCH_CHECK();
}

__global__ __launch_bounds__(256) void k_rel_dists(const float* __restrict__ logits,
                                                   float* __restrict__ out, int64_t rows, int C) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* x = logits + row * C;
  float v[4];
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = lane + 64 * j;
    v[j] = (c < C) ? x[c] : -INFINITY;
    mx = fmaxf(mx, v[j]);
  }
  mx = wave_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    v[j] = (lane + 64 * j < C) ? expf(v[j] - mx) : 0.f;
    sum += v[j];
  }
  sum = wave_sum(sum);
  float* o = out + row * (C + 1);
  if (lane == 0) o[0] = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = lane + 64 * j;
    if (c < C) o[1 + c] = v[j] / sum;
  }
}

extern "C" int pn_rel_dists_f32(const float* logits, float* out, int64_t rows, int C,
                                void* stream) {
  if (!logits || !out || rows <= 0 || C < 1 || C > 256) return PN_BAD_ARG;
  hipLaunchKernelGGL(k_rel_dists, dim3(pn_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream,
                     logits, out, rows, C);
  return PN_LAUNCH_CHECK();
}

// Panoptic id map.  softmax over the n kept masks is monotone, so the per-pixel
// argmax is taken on the logits (first index on ties).  `area` must be zeroed by the
// caller; integer atomics keep it deterministic.
__global__ __launch_bounds__(256) void k_panoptic(const float* __restrict__ masks,
                                                  const int64_t* __restrict__ labels,
                                                  const int32_t* __restrict__ remap,
                                                  int64_t* __restrict__ seg,
                                                  int32_t* __restrict__ area, int n, int64_t HW) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= HW) return;
  float best = masks[p];
  int bi = 0;
  for (int i = 1; i < n; ++i) {
    const float v = masks[(int64_t)i * HW + p];
    if (v > best) { best = v; bi = i; }
  }
  if (remap) bi = remap[bi];
  seg[p] = (int64_t)bi * 1000 + labels[bi];
  atomicAdd(&area[bi], 1);
}

extern "C" int pn_panoptic_f32(const float* masks, const int64_t* labels, const int32_t* remap,
                               int64_t* seg, int32_t* area, int n, int64_t HW, void* stream) {
  if (!masks || !labels || !seg || !area || n <= 0 || HW <= 0) return PN_BAD_ARG;
  hipLaunchKernelGGL(k_panoptic, dim3(pn_cdiv(HW, 256)), dim3(256), 0, (hipStream_t)stream,
                     masks, labels, remap, seg, area, n, HW);
  return PN_LAUNCH_CHECK();
}
