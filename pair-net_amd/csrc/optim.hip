// The optimizer step of a training iteration (reference: mmcv's OptimizerHook with
// grad_clip = dict(max_norm=0.1, norm_type=2) around torch.optim.AdamW(lr=1e-4, weight_decay=1e-4),
// configs/mask2former/pairnet.py:353-368) over ONE flat fp32 parameter buffer with per-segment
// learning-rate / weight-decay multipliers (`paramwise_cfg`: lr_mult per module, norm_decay_mult):
// two launches for the global gradient norm and its clip coefficient (deterministic two-stage
// sum in double), one launch for the AdamW update of every parameter.  The gradient's data-
// parallel average rides along as `pre` (the all-reduce SUMS; 1 / world is applied where the
// gradient is read), and the clip coefficient stays on the device (no host round trip).
#include "common.h"

__global__ __launch_bounds__(256) void k_sumsq_partial(const float* __restrict__ x, int64_t n,
                                                       double* __restrict__ partial, float pre) {
  __shared__ double red[4];
  const int64_t chunk = (n + gridDim.x - 1) / gridDim.x;
  const int64_t beg = (int64_t)blockIdx.x * chunk, end = min(beg + chunk, n);
  double s = 0.0;
  for (int64_t i = beg + threadIdx.x; i < end; i += 256) {
    const double v = (double)(x[i] * pre);
    s += v * v;
  }
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// out[0] = ||g||_2, out[1] = clip coefficient min(1, max_norm / (norm + 1e-6))
// (torch.nn.utils.clip_grad_norm_; max_norm <= 0: no clipping)
__global__ void k_clip_coef(const double* __restrict__ partial, int nparts, float max_norm,
                            float* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double t = 0.0;
  for (int i = 0; i < nparts; ++i) t += partial[i];
  const float norm = (float)sqrt(t);
  out[0] = norm;
  out[1] = max_norm > 0.f ? fminf(1.f, max_norm / (norm + 1e-6f)) : 1.f;
}

extern "C" int pn_grad_norm_clip_f32(const float* g, int64_t n, float pre, float max_norm,
                                     float* out, double* scratch, void* stream) {
  if (!g || !out || !scratch || n <= 0) return PN_BAD_ARG;
  const int parts = (int)min((int64_t)256, (n + 4095) / 4096);     // scratch: 256 doubles
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_sumsq_partial, dim3(parts), dim3(256), 0, s, g, n, scratch, pre);
  hipLaunchKernelGGL(k_clip_coef, dim3(1), dim3(64), 0, s, scratch, parts, max_norm, out);
  return PN_LAUNCH_CHECK();
}

// torch.optim.AdamW's single-tensor update, element for element:
//   p *= 1 - lr wd;  m += (g - m)(1 - b1);  v = b2 v + (1 - b2) g g;
//   p -= (lr / bc1) m / (sqrt(v) / sqrt(bc2) + eps)
// with lr = base lr x the segment's lr multiplier, wd = base wd x the segment's decay multiplier.
__global__ __launch_bounds__(256) void k_adamw(float* __restrict__ p, const float* __restrict__ g,
                                               float* __restrict__ m, float* __restrict__ v,
                                               int64_t n, const int64_t* __restrict__ seg_off,
                                               const float* __restrict__ seg_lr,
                                               const float* __restrict__ seg_wd, int nseg,
                                               float lr, float b1, float b2, float eps, float wd,
                                               float bc1, float bc2_sqrt,
                                               const float* __restrict__ clip, float pre) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  int lo = 0, hi = nseg - 1;                 // the segment with seg_off[s] <= i < seg_off[s + 1]
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (seg_off[mid] <= i) lo = mid; else hi = mid - 1;
  }
  const float lr_s = lr * seg_lr[lo], wd_s = wd * seg_wd[lo];
  const float gi = g[i] * pre * (clip ? clip[1] : 1.f);
  float pi = p[i] * (1.f - lr_s * wd_s);
  const float mi = m[i] + (gi - m[i]) * (1.f - b1);
  const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
  const float denom = sqrtf(vi) / bc2_sqrt + eps;
  pi -= (lr_s / bc1) * (mi / denom);
  p[i] = pi; m[i] = mi; v[i] = vi;
}

extern "C" int pn_adamw_f32(float* p, const float* g, float* m, float* v, int64_t n,
                            const int64_t* seg_off, const float* seg_lr, const float* seg_wd,
                            int nseg, float lr, float beta1, float beta2, float eps,
                            float weight_decay, int step, const float* clip, float pre,
                            void* stream) {
  if (!p || !g || !m || !v || !seg_off || !seg_lr || !seg_wd || n <= 0 || nseg <= 0 || step <= 0)
    return PN_BAD_ARG;
  if (!(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f)) return PN_BAD_ARG;
  const float bc1 = 1.f - powf(beta1, (float)step);
  const float bc2_sqrt = sqrtf(1.f - powf(beta2, (float)step));
  hipLaunchKernelGGL(k_adamw, dim3(pn_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v,
                     n, seg_off, seg_lr, seg_wd, nseg, lr, beta1, beta2, eps, weight_decay, bc1,
                     bc2_sqrt, clip, pre);
  return PN_LAUNCH_CHECK();
}
