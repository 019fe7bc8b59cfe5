// Loss FORWARD of CrossHead2 on device outputs (first slice: target costs and
// loss values; no backward).  What the reference computes with torch ops + scipy
// (pairnet_head.py:419-718) becomes six small kernels; only the two Hungarian cost matrices
// (Q x G and R x G floats) go to the host, where the reference solves them too (`cost.cpu()`,
// matcher.py:262-264).  All of it is a few hundred KB per image: latency-sized kernels, fp32
// arithmetic in the reference's formulas, deterministic reductions (fixed order).
#include "common.h"

__device__ __forceinline__ float block_sum(float v, float* red) {   // <= 1024 threads
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float t = 0.f;
  for (int w = 0; w < nw; ++w) t += red[w];      // every thread, wave order: deterministic
  return t;
}
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float t = -INFINITY;
  for (int w = 0; w < nw; ++w) t = fmaxf(t, red[w]);
  return t;
}

// ---- PSGTr.forward_train's ground-truth mask preparation (frameworks/psgtr.py:126-141):
// F.pad(mask [G][h][w], right / bottom to the batch's [H][W]) then F.interpolate(size =
// (Ho, Wo), mode = "nearest") in one pass over the OUTPUT: ATen's legacy nearest reads source
// index min(floor(dst * (float)in / out), in - 1); a source pixel outside [h) x [w) is padding.
__global__ __launch_bounds__(256) void k_gt_mask_prepare(const uint8_t* __restrict__ in,
                                                         uint8_t* __restrict__ out, int h, int w,
                                                         int H, int W, int Ho, int Wo) {
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, g = blockIdx.z;
  if (x >= Wo) return;
  const float sy_f = (float)H / (float)Ho, sx_f = (float)W / (float)Wo;
  const int sy = min((int)floorf((float)y * sy_f), H - 1);
  const int sx = min((int)floorf((float)x * sx_f), W - 1);
  out[((int64_t)g * Ho + y) * Wo + x] =
      (sy < h && sx < w) ? in[((int64_t)g * h + sy) * w + sx] : (uint8_t)0;
}

extern "C" int pn_gt_mask_prepare_u8(const uint8_t* masks, uint8_t* out, int G, int h, int w,
                                     int H, int W, int Ho, int Wo, void* stream) {
  if (!masks || !out || G <= 0 || G > 65535 || h <= 0 || w <= 0 || H < h || W < w || Ho <= 0 ||
      Wo <= 0 || Ho > 65535)
    return PN_BAD_ARG;
  hipLaunchKernelGGL(k_gt_mask_prepare, dim3(pn_cdiv(Wo, 256), Ho, G), dim3(256), 0,
                     (hipStream_t)stream, masks, out, h, w, H, W, Ho, Wo);
  return PN_LAUNCH_CHECK();
}

// ---- mmcv point_sample = F.grid_sample(input, 2 p - 1, bilinear, zeros, align_corners=False)
// maps [P][h][w] (float, or uint8 0/1 for ground-truth masks), pts [Np][2] (x, y) in [0, 1],
// shared by all P maps (pairnet_head.py:630-638: the same random points for every query and
// every ground truth); out [P][Np].
template <typename T>
__global__ __launch_bounds__(256) void k_point_sample(const T* __restrict__ maps,
                                                      const float* __restrict__ pts,
                                                      float* __restrict__ out, int P, int h,
                                                      int w, int Np) {
  const int i = blockIdx.x * 256 + threadIdx.x, p = blockIdx.y;
  if (i >= Np) return;
  const float2 pt = *reinterpret_cast<const float2*>(pts + 2 * i);
  const float cx = 2.f * pt.x - 1.f, cy = 2.f * pt.y - 1.f;
  const float ix = ((cx + 1.f) * (float)w - 1.f) / 2.f, iy = ((cy + 1.f) * (float)h - 1.f) / 2.f;
  const float fx = floorf(ix), fy = floorf(iy);
  const int x0 = (int)fminf(fmaxf(fx, -2.f), (float)w), y0 = (int)fminf(fmaxf(fy, -2.f), (float)h);
  const float x1f = fx + 1.f, y1f = fy + 1.f;
  const float nw = (x1f - ix) * (y1f - iy), ne = (ix - fx) * (y1f - iy);
  const float sw = (x1f - ix) * (iy - fy), se = (ix - fx) * (iy - fy);
  const T* m = maps + (int64_t)p * h * w;
  auto at = [&](int y, int x) -> float {
    return (x >= 0 && x < w && y >= 0 && y < h) ? (float)m[(int64_t)y * w + x] : 0.f;
  };
  float v = 0.f;       // accumulated in ATen's order: nw, ne, sw, se
  v += at(y0, x0) * nw;
  v += at(y0, x0 + 1) * ne;
  v += at(y0 + 1, x0) * sw;
  v += at(y0 + 1, x0 + 1) * se;
  out[(int64_t)p * Np + i] = v;
}

extern "C" int pn_point_sample_f32(const void* maps, int maps_are_u8, const float* pts, float* out,
                                   int P, int h, int w, int Np, void* stream) {
  if (!maps || !pts || !out || P <= 0 || h <= 0 || w <= 0 || Np <= 0 || ((uintptr_t)pts & 7))
    return PN_BAD_ARG;
  const dim3 grid(pn_cdiv(Np, 256), P);
  if (maps_are_u8)
    hipLaunchKernelGGL(k_point_sample<uint8_t>, grid, dim3(256), 0, (hipStream_t)stream,
                       (const uint8_t*)maps, pts, out, P, h, w, Np);
  else
    hipLaunchKernelGGL(k_point_sample<float>, grid, dim3(256), 0, (hipStream_t)stream,
                       (const float*)maps, pts, out, P, h, w, Np);
  return PN_LAUNCH_CHECK();
}

// ---- MaskHungarianAssigner's cost matrix (mmdet 2.25.1; cfg pairnet.py:200-206):
//   cost[q][g] = -softmax(cls[q])[label[g]] w_cls
//              + mean_p BCE(x[q][p], t[g][p]) w_mask        (CrossEntropyLossCost, sigmoid)
//              + (1 - (2 sum_p s t + eps) / (sum_p s + sum_p t + eps)) w_dice,  s = sigmoid(x)
// x [Q][Np] sampled mask logits, t [G][Np] sampled ground-truth masks.  With
// BCE(x, 1) - BCE(x, 0) = -x the mask term is (sum_p BCE(x, 0) - sum_p x t) / Np.
// One workgroup per query; ground truths in groups of 8 (24 accumulators per thread).
__global__ __launch_bounds__(256) void k_mask_match_cost(
    const float* __restrict__ cls, int ncls, const int64_t* __restrict__ labels,
    const float* __restrict__ x, const float* __restrict__ t, float* __restrict__ cost, int G,
    int Np, float w_cls, float w_mask, float w_dice, float dice_eps) {
  __shared__ float red[8];
  const int q = blockIdx.x, tid = threadIdx.x;
  const float* xr = x + (int64_t)q * Np;
  // class term: softmax over the ncls logits of this query
  const float* cr = cls + (int64_t)q * ncls;
  float mx = -INFINITY;
  for (int c = tid; c < ncls; c += 256) mx = fmaxf(mx, cr[c]);
  mx = block_max(mx, red);
  float den = 0.f;
  for (int c = tid; c < ncls; c += 256) den += expf(cr[c] - mx);
  den = block_sum(den, red);
  float s_neg = 0.f, s_sig = 0.f;
  for (int i = tid; i < Np; i += 256) {
    const float v = xr[i];
    s_neg += fmaxf(v, 0.f) + log1pf(expf(-fabsf(v)));       // BCE-with-logits against target 0
    s_sig += 1.f / (1.f + expf(-v));
  }
  s_neg = block_sum(s_neg, red);
  s_sig = block_sum(s_sig, red);
  for (int g0 = 0; g0 < G; g0 += 8) {
    float dxt[8], dst[8], st[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) dxt[j] = dst[j] = st[j] = 0.f;
    for (int i = tid; i < Np; i += 256) {
      const float v = xr[i];
      const float s = 1.f / (1.f + expf(-v));
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float tv = t[(int64_t)min(g0 + j, G - 1) * Np + i];
        dxt[j] += v * tv; dst[j] += s * tv; st[j] += tv;
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float a = block_sum(dxt[j], red), b = block_sum(dst[j], red), c = block_sum(st[j], red);
      if (tid == 0 && g0 + j < G) {
        const float c_cls = -(expf(cr[labels[g0 + j]] - mx) / den) * w_cls;
        const float c_mask = (s_neg - a) / (float)Np * w_mask;
        const float c_dice = (1.f - (2.f * b + dice_eps) / (s_sig + c + dice_eps)) * w_dice;
        cost[(int64_t)q * G + g0 + j] = (c_cls + c_mask) + c_dice;
      }
    }
  }
}

extern "C" int pn_mask_match_cost_f32(const float* cls, int ncls, const int64_t* gt_labels,
                                      const float* pred_pts, const float* gt_pts, float* cost,
                                      int Q, int G, int Np, float w_cls, float w_mask,
                                      float w_dice, float dice_eps, void* stream) {
  if (!cls || !gt_labels || !pred_pts || !gt_pts || !cost || Q <= 0 || G <= 0 || Np <= 0 ||
      ncls <= 0)
    return PN_BAD_ARG;
  hipLaunchKernelGGL(k_mask_match_cost, dim3(Q), dim3(256), 0, (hipStream_t)stream, cls, ncls,
                     gt_labels, pred_pts, gt_pts, cost, G, Np, w_cls, w_mask, w_dice, dice_eps);
  return PN_LAUNCH_CHECK();
}

// ---- IdMatcher's cost matrix (approaches/matcher.py:250-258): three ClassificationCosts
//   cost[r][g] = -softmax(sub[r])[gs[g]] w_s - softmax(obj[r])[go[g]] w_o - softmax(rel[r])[gr[g]] w_r
__global__ __launch_bounds__(64) void k_id_match_cost(
    const float* __restrict__ sub, const float* __restrict__ obj, const float* __restrict__ rel,
    int ncls, int nrel, const int64_t* __restrict__ gs, const int64_t* __restrict__ go,
    const int64_t* __restrict__ gr, float* __restrict__ cost, int G, float ws, float wo, float wr) {
  const int r = blockIdx.x, lane = threadIdx.x;
  const float* rows[3] = {sub + (int64_t)r * ncls, obj + (int64_t)r * ncls, rel + (int64_t)r * nrel};
  const int n[3] = {ncls, ncls, nrel};
  float mx[3], den[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float m = -INFINITY;
    for (int c = lane; c < n[k]; c += 64) m = fmaxf(m, rows[k][c]);
    m = wave_max(m);
    float d = 0.f;
    for (int c = lane; c < n[k]; c += 64) d += expf(rows[k][c] - m);
    mx[k] = m; den[k] = wave_sum(d);
  }
  for (int g = lane; g < G; g += 64) {
    const float a = -(expf(rows[0][gs[g]] - mx[0]) / den[0]) * ws;
    const float b = -(expf(rows[1][go[g]] - mx[1]) / den[1]) * wo;
    const float c = -(expf(rows[2][gr[g]] - mx[2]) / den[2]) * wr;
    cost[(int64_t)r * G + g] = (a + b) + c;
  }
}

extern "C" int pn_id_match_cost_f32(const float* sub, const float* obj, const float* rel, int ncls,
                                    int nrel, const int64_t* gt_sub, const int64_t* gt_obj,
                                    const int64_t* gt_rel, float* cost, int R, int G, float w_sub,
                                    float w_obj, float w_rel, void* stream) {
  if (!sub || !obj || !rel || !gt_sub || !gt_obj || !gt_rel || !cost || R <= 0 || G <= 0 ||
      ncls <= 0 || nrel <= 0)
    return PN_BAD_ARG;
  hipLaunchKernelGGL(k_id_match_cost, dim3(R), dim3(64), 0, (hipStream_t)stream, sub, obj, rel,
                     ncls, nrel, gt_sub, gt_obj, gt_rel, cost, G, w_sub, w_obj, w_rel);
  return PN_LAUNCH_CHECK();
}

// ---- mmdet CrossEntropyLoss (softmax form), reduction "mean" over the KEPT rows:
//   out = loss_weight / n_kept * sum_{rows with target >= 0} cw[y] (logsumexp(x) - x[y])
// One workgroup; wave w takes rows w, w + 4, ...; rows are summed in row order.
#define LOSS_MAX_ROWS 4096
__global__ __launch_bounds__(256) void k_ce_mean(const float* __restrict__ logits, int64_t ld,
                                                 const int64_t* __restrict__ target,
                                                 const float* __restrict__ class_weight,
                                                 float* __restrict__ out, int rows, int C,
                                                 float loss_weight) {
  __shared__ float per_row[LOSS_MAX_ROWS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int r = wave; r < rows; r += 4) {
    const int64_t y = target[r];
    float l = 0.f;
    if (y >= 0) {                                   // (wave-uniform)
      const float* xr = logits + (int64_t)r * ld;
      float m = -INFINITY;
      for (int c = lane; c < C; c += 64) m = fmaxf(m, xr[c]);
      m = wave_max(m);
      float d = 0.f;
      for (int c = lane; c < C; c += 64) d += expf(xr[c] - m);
      d = wave_sum(d);
      l = (logf(d) + m) - xr[y];
      if (class_weight) l *= class_weight[y];
    }
    if (lane == 0) per_row[r] = l;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    int n = 0;
    for (int r = 0; r < rows; ++r) {
      if (target[r] >= 0) { s += per_row[r]; ++n; }
    }
    out[0] = n ? loss_weight * (s / (float)n) : 0.f;
  }
}

extern "C" int pn_ce_mean_f32(const float* logits, int64_t ld, const int64_t* target,
                              const float* class_weight, float* out, int rows, int C,
                              float loss_weight, void* stream) {
  if (!logits || !target || !out || rows <= 0 || rows > LOSS_MAX_ROWS || C <= 0 || ld < C)
    return PN_BAD_ARG;
  hipLaunchKernelGGL(k_ce_mean, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, ld, target,
                     class_weight, out, rows, C, loss_weight);
  return PN_LAUNCH_CHECK();
}

// ---- mmdet SeesawLoss, class part (`loss_cls_classes`; Wang et al. 2021), over the kept rows:
//   w[j] = (cum[j] / cum[y])^p  if cum[j] < cum[y]  (counts clamped to >= 1), else 1
//        * (s[j] / max(s[y], eps))^q  if that ratio > 1, else 1,        s = softmax(x)
//   loss = logsumexp(x') - x'[y],  x'[j] = x[j] + log w[j]  (j != y),  x'[y] = x[y]
// C <= 64: a wave per row, a lane per class.  `cum` already includes this batch's labels
// (seesaw_loss.py accumulates before it weighs).
__global__ __launch_bounds__(256) void k_seesaw_mean(const float* __restrict__ logits, int64_t ld,
                                                     const int64_t* __restrict__ target,
                                                     const float* __restrict__ cum,
                                                     float* __restrict__ out, int rows, int C,
                                                     float p, float q, float eps,
                                                     float loss_weight) {
  __shared__ float per_row[LOSS_MAX_ROWS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int r = wave; r < rows; r += 4) {
    const int64_t y = target[r];
    float l = 0.f;
    if (y >= 0) {
      const bool on = lane < C;
      const float xv = on ? logits[(int64_t)r * ld + lane] : -INFINITY;
      const float m = wave_max(xv);
      const float e = on ? expf(xv - m) : 0.f;
      const float s = e / wave_sum(e);
      const float sy = __shfl(s, (int)y, 64);
      float w = 1.f;
      if (p > 0.f) {
        const float cj = fmaxf(on ? cum[lane] : 1.f, 1.f), cy = fmaxf(cum[y], 1.f);
        const float ratio = cj / cy;
        if (ratio < 1.f) w *= powf(ratio, p);
      }
      if (q > 0.f) {
        const float ratio = s / fmaxf(sy, eps);
        if (ratio > 1.f) w *= powf(ratio, q);
      }
      const float xs = on ? (lane == (int)y ? xv : xv + logf(w)) : -INFINITY;
      const float m2 = wave_max(xs);
      const float d = wave_sum(on ? expf(xs - m2) : 0.f);
      l = (logf(d) + m2) - __shfl(xs, (int)y, 64);
    }
    if (lane == 0) per_row[r] = l;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    int n = 0;
    for (int r = 0; r < rows; ++r) {
      if (target[r] >= 0) { s += per_row[r]; ++n; }
    }
    out[0] = n ? loss_weight * (s / (float)n) : 0.f;
  }
}

extern "C" int pn_seesaw_mean_f32(const float* logits, int64_t ld, const int64_t* target,
                                  const float* cum_samples, float* out, int rows, int C, float p,
                                  float q, float eps, float loss_weight, void* stream) {
  if (!logits || !target || !cum_samples || !out || rows <= 0 || rows > LOSS_MAX_ROWS || C <= 0 ||
      C > 64 || ld < C)
    return PN_BAD_ARG;
  hipLaunchKernelGGL(k_seesaw_mean, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, ld, target,
                     cum_samples, out, rows, C, p, q, eps, loss_weight);
  return PN_LAUNCH_CHECK();
}

// ---- nn.BCEWithLogitsLoss(pos_weight, reduction="mean") * loss_weight (seg_losses.py:153-166):
//   l = (1 - t) x + (1 + (pw - 1) t) (log1p(exp(-|x|)) + max(-x, 0)),  pw = n / #(t > 0)
// (pairnet_head.py:541-542).  One workgroup of 1024 threads, fixed-order reduction; pw is
// computed here from the targets.
__global__ __launch_bounds__(1024) void k_bce_posw_mean(const float* __restrict__ x,
                                                        const float* __restrict__ t,
                                                        float* __restrict__ out, int64_t n,
                                                        float loss_weight) {
  __shared__ float red[16];
  float cnt = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += 1024) cnt += t[i] > 0.f ? 1.f : 0.f;
  cnt = block_sum(cnt, red);
  const float pw = (float)n / cnt;
  float s = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += 1024) {
    const float xv = x[i], tv = t[i];
    const float lw = 1.f + (pw - 1.f) * tv;
    s += (1.f - tv) * xv + lw * (log1pf(expf(-fabsf(xv))) + fmaxf(-xv, 0.f));
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) { out[0] = loss_weight * (s / (float)n); out[1] = pw; }
}

extern "C" int pn_bce_posw_mean_f32(const float* logits, const float* target, float* out /* [2] */,
                                    int64_t n, float loss_weight, void* stream) {
  if (!logits || !target || !out || n <= 0) return PN_BAD_ARG;
  hipLaunchKernelGGL(k_bce_posw_mean, dim3(1), dim3(1024), 0, (hipStream_t)stream, logits, target,
                     out, n, loss_weight);
  return PN_LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------------
// Gradients of the three loss reductions above with respect to their logits (the
// first backward slice: d loss / d {sub, obj, rel, importance}; pairnet_head.py:518-552).  Each is
// the analytic derivative of the forward kernel beside it, in the same row / lane layout; rows
// the reference masks out (target < 0) get zeros.
//   CE:      g[r][c] = lw / n * w[y] * (softmax(x_r)[c] - [c == y])
//   Seesaw:  g[r][j] = lw / n * (softmax(x'_r)[j] - [j == y]);  the seesaw weights are constants of
//            the backward pass ([3P] seesaw_ce_loss takes softmax(cls_score.detach()))
//   BCE:     g[i]    = lw / n * ((1 - t) - (1 + (pw - 1) t) sigmoid(-x))
__global__ __launch_bounds__(256) void k_ce_mean_grad(const float* __restrict__ logits, int64_t ld,
                                                      const int64_t* __restrict__ target,
                                                      const float* __restrict__ class_weight,
                                                      float* __restrict__ grad, int64_t ldg,
                                                      int rows, int C, float loss_weight) {
  __shared__ int kept;
  if (threadIdx.x == 0) {
    int n = 0;
    for (int r = 0; r < rows; ++r) n += target[r] >= 0;
    kept = n;
  }
  __syncthreads();
  const float scale = kept ? loss_weight / (float)kept : 0.f;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int r = wave; r < rows; r += 4) {
    const int64_t y = target[r];
    float* gr = grad + (int64_t)r * ldg;
    if (y < 0) {                                    // (wave-uniform)
      for (int c = lane; c < C; c += 64) gr[c] = 0.f;
      continue;
    }
    const float* xr = logits + (int64_t)r * ld;
    float m = -INFINITY;
    for (int c = lane; c < C; c += 64) m = fmaxf(m, xr[c]);
    m = wave_max(m);
    float d = 0.f;
    for (int c = lane; c < C; c += 64) d += expf(xr[c] - m);
    d = wave_sum(d);
    const float w = scale * (class_weight ? class_weight[y] : 1.f);
    for (int c = lane; c < C; c += 64)
      gr[c] = w * (expf(xr[c] - m) / d - (c == (int)y ? 1.f : 0.f));
  }
}

extern "C" int pn_ce_mean_grad_f32(const float* logits, int64_t ld, const int64_t* target,
                                   const float* class_weight, float* grad, int64_t ldg, int rows,
                                   int C, float loss_weight, void* stream) {
  if (!logits || !target || !grad || rows <= 0 || rows > LOSS_MAX_ROWS || C <= 0 || ld < C || ldg < C)
    return PN_BAD_ARG;
  hipLaunchKernelGGL(k_ce_mean_grad, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, ld, target,
                     class_weight, grad, ldg, rows, C, loss_weight);
  return PN_LAUNCH_CHECK();
}

__global__ __launch_bounds__(256) void k_seesaw_mean_grad(const float* __restrict__ logits, int64_t ld,
                                                          const int64_t* __restrict__ target,
                                                          const float* __restrict__ cum,
                                                          float* __restrict__ grad, int64_t ldg,
                                                          int rows, int C, float p, float q,
                                                          float eps, float loss_weight) {
  __shared__ int kept;
  if (threadIdx.x == 0) {
    int n = 0;
    for (int r = 0; r < rows; ++r) n += target[r] >= 0;
    kept = n;
  }
  __syncthreads();
  const float scale = kept ? loss_weight / (float)kept : 0.f;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int r = wave; r < rows; r += 4) {
    const int64_t y = target[r];
    const bool on = lane < C;
    float g = 0.f;
    if (y >= 0) {
      const float xv = on ? logits[(int64_t)r * ld + lane] : -INFINITY;
      const float m = wave_max(xv);
      const float e = on ? expf(xv - m) : 0.f;
      const float s = e / wave_sum(e);
      const float sy = __shfl(s, (int)y, 64);
      float w = 1.f;
      if (p > 0.f) {
        const float cj = fmaxf(on ? cum[lane] : 1.f, 1.f), cy = fmaxf(cum[y], 1.f);
        const float ratio = cj / cy;
        if (ratio < 1.f) w *= powf(ratio, p);
      }
      if (q > 0.f) {
        const float ratio = s / fmaxf(sy, eps);
        if (ratio > 1.f) w *= powf(ratio, q);
      }
      const float xs = on ? (lane == (int)y ? xv : xv + logf(w)) : -INFINITY;
      const float m2 = wave_max(xs);
      const float e2 = on ? expf(xs - m2) : 0.f;
      g = scale * (e2 / wave_sum(e2) - (lane == (int)y ? 1.f : 0.f));
    }
    if (on) grad[(int64_t)r * ldg + lane] = g;
  }
}

extern "C" int pn_seesaw_mean_grad_f32(const float* logits, int64_t ld, const int64_t* target,
                                       const float* cum_samples, float* grad, int64_t ldg, int rows,
                                       int C, float p, float q, float eps, float loss_weight,
                                       void* stream) {
  if (!logits || !target || !cum_samples || !grad || rows <= 0 || rows > LOSS_MAX_ROWS || C <= 0 ||
      C > 64 || ld < C || ldg < C)
    return PN_BAD_ARG;
  hipLaunchKernelGGL(k_seesaw_mean_grad, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, ld,
                     target, cum_samples, grad, ldg, rows, C, p, q, eps, loss_weight);
  return PN_LAUNCH_CHECK();
}

__global__ __launch_bounds__(1024) void k_bce_posw_mean_grad(const float* __restrict__ x,
                                                             const float* __restrict__ t,
                                                             float* __restrict__ grad, int64_t n,
                                                             float loss_weight) {
  __shared__ float red[16];
  float cnt = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += 1024) cnt += t[i] > 0.f ? 1.f : 0.f;
  cnt = block_sum(cnt, red);
  const float pw = (float)n / cnt, scale = loss_weight / (float)n;
  for (int64_t i = threadIdx.x; i < n; i += 1024) {
    const float xv = x[i], tv = t[i];
    const float sneg = 1.f / (1.f + expf(xv));                // sigmoid(-x)
    grad[i] = scale * ((1.f - tv) - (1.f + (pw - 1.f) * tv) * sneg);
  }
}

extern "C" int pn_bce_posw_mean_grad_f32(const float* logits, const float* target, float* grad,
                                         int64_t n, float loss_weight, void* stream) {
  if (!logits || !target || !grad || n <= 0) return PN_BAD_ARG;
  hipLaunchKernelGGL(k_bce_posw_mean_grad, dim3(1), dim3(1024), 0, (hipStream_t)stream, logits,
                     target, grad, n, loss_weight);
  return PN_LAUNCH_CHECK();
}
