// Device-side triplet / panoptic post-processing (pairnet_head.py:788-924), replacing
// the reference's host loops and .item() syncs.
#include "common.h"

// One wave per row.  prob = exp(x - max) / sum over all C logits; the label is the
// argmax of prob over the first C-1 columns (first index on ties), score its prob.
__global__ __launch_bounds__(256) void k_cls_argmax(const float* __restrict__ logits,
                                                    int64_t* __restrict__ label,
                                                    float* __restrict__ score, int64_t rows,
                                                    int C, int label_offset) {
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
  if (lane == 0) { label[row] = bi + label_offset; score[row] = best; }
}

extern "C" int pn_cls_argmax_f32(const float* logits, int64_t* label, float* score,
                                 int64_t rows, int C, int label_offset, void* stream) {
  if (!logits || !label || !score || rows <= 0 || C < 2 || C > 256) return PN_BAD_ARG;
  hipLaunchKernelGGL(k_cls_argmax, dim3(pn_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream,
                     logits, label, score, rows, C, label_offset);
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

// ---- CrossHeadBaseline triplet ranking (baseline.py:1025-1046) -----------------
// probs = softmax over all C logits; fg = probs without column 0 (the "no relation"
// class), packed [rows][C-1] so the flat top-k index is row*(C-1) + (label-1).
__global__ __launch_bounds__(256) void k_softmax_fg(const float* __restrict__ logits,
                                                    float* __restrict__ probs,
                                                    float* __restrict__ fg, int64_t rows, int C) {
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
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = lane + 64 * j;
    if (c < C) {
      const float pr = v[j] / sum;
      probs[row * C + c] = pr;
      if (c > 0) fg[row * (C - 1) + c - 1] = pr;
    }
  }
}

extern "C" int pn_softmax_fg_f32(const float* logits, float* probs, float* fg, int64_t rows,
                                 int C, void* stream) {
  if (!logits || !probs || !fg || rows <= 0 || C < 2 || C > 256) return PN_BAD_ARG;
  hipLaunchKernelGGL(k_softmax_fg, dim3(pn_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream,
                     logits, probs, fg, rows, C);
  return PN_LAUNCH_CHECK();
}

// First-index argmax of each row (torch.max(-1)[1] on the matching scores,
// baseline.py:398-399).  One wave per row.
__global__ __launch_bounds__(256) void k_row_argmax(const float* __restrict__ x,
                                                    int64_t* __restrict__ idx, int64_t rows,
                                                    int n) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* r = x + row * n;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int c = lane; c < n; c += 64) {
    const float v = r[c];
    if (v > best || bi == 0x7fffffff) { best = v; bi = c; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (oi != 0x7fffffff && (bi == 0x7fffffff || ob > best || (ob == best && oi < bi))) {
      best = ob; bi = oi;
    }
  }
  if (lane == 0) idx[row] = bi;
}

extern "C" int pn_row_argmax_f32(const float* x, int64_t* idx, int64_t rows, int n,
                                 void* stream) {
  if (!x || !idx || rows <= 0 || n <= 0) return PN_BAD_ARG;
  hipLaunchKernelGGL(k_row_argmax, dim3(pn_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x,
                     idx, rows, n);
  return PN_LAUNCH_CHECK();
}

// Ranked triplets: tri[j] = relation query of the j-th best (query, predicate) pair,
// rem[j] its predicate - 1.  labels = [s_label[tri]+1 | o_label[tri]+1],
// r_labels = rem + 1, r_scores[j] = probs[tri[j]][rem[j]+1], r_dists[j] = probs[tri[j]]
// (baseline.py:1035-1046).
__global__ __launch_bounds__(256) void k_triplet_finish(
    const int64_t* __restrict__ s_label, const int64_t* __restrict__ o_label,
    const float* __restrict__ probs, const int64_t* __restrict__ tri,
    const int64_t* __restrict__ rem, int64_t* __restrict__ labels,
    int64_t* __restrict__ r_labels, float* __restrict__ r_scores, float* __restrict__ r_dists,
    int k, int C) {
  const int j = blockIdx.x;
  const int64_t t = tri[j];
  if (threadIdx.x == 0) {
    labels[j] = s_label[t] + 1;
    labels[k + j] = o_label[t] + 1;
    r_labels[j] = rem[j] + 1;
    r_scores[j] = probs[t * C + rem[j] + 1];
  }
  for (int c = threadIdx.x; c < C; c += 256) r_dists[(int64_t)j * C + c] = probs[t * C + c];
}

extern "C" int pn_triplet_finish(const int64_t* s_label, const int64_t* o_label,
                                 const float* probs, const int64_t* tri, const int64_t* rem,
                                 int64_t* labels, int64_t* r_labels, float* r_scores,
                                 float* r_dists, int k, int C, void* stream) {
  if (!s_label || !o_label || !probs || !tri || !rem || !labels || !r_labels || !r_scores ||
      !r_dists || k <= 0 || C <= 0)
    return PN_BAD_ARG;
  hipLaunchKernelGGL(k_triplet_finish, dim3(k), dim3(256), 0, (hipStream_t)stream, s_label,
                     o_label, probs, tri, rem, labels, r_labels, r_scores, r_dists, k, C);
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

// ---------------------------------------------------------------------------------
// Sync-free panoptic post-processing (pairnet_head.py:845-905 without host round
// trips).  The reference filters queries on the host (`keep`), merges duplicate stuff
// classes, and re-runs the per-pixel argmax while any segment has area <= 4, reading
// areas back with .item() each time.  Here the keep list, the duplicate map, the alive
// flags and the compact ranks live in device memory; a bounded number of argmax/filter
// rounds is enqueued, a round after convergence returns at once, and `active` tells the
// caller (at whatever D2H point it already has) whether the loop still has to go on
// (pn_panoptic_continue_f32).
// ---------------------------------------------------------------------------------
struct PanState {      // device-resident, Q <= 256
  int32_t nkeep;       // number of kept queries
  int32_t active;      // 1: the last filter round dropped something -> another round is due
  int32_t rounds;      // filter rounds that dropped something so far
  int32_t all_gone;    // every kept segment was filtered (the reference raises here)
  int32_t first;       // 1 until the first argmax round has run (stuff merging applies to it)
  int32_t pad[11];
  int32_t kept[256];   // kept position -> query index
  int32_t remap[256];  // kept position -> first kept position of the same stuff class
  int32_t alive[256];
  int32_t rank[256];   // compact index among alive positions
  int64_t klab[256];   // label of kept position
};

__global__ __launch_bounds__(256) void k_pan_select(const int64_t* __restrict__ labels,
                                                    const float* __restrict__ scores, int Q,
                                                    int last_real_class, PanState* st) {
  __shared__ int flag[256], pos[256];
  const int t = threadIdx.x;
  const bool keep = t < Q && labels[t] != last_real_class && scores[t] > 0.5f;
  flag[t] = keep ? 1 : 0;
  __syncthreads();
  if (t == 0) {
    int c = 0;
    for (int i = 0; i < 256; ++i) { pos[i] = c; c += flag[i]; }
    st->nkeep = c;
    st->active = 1;
    st->rounds = 0;
    st->all_gone = 0;
    st->first = 1;
  }
  __syncthreads();
  if (keep) {
    st->kept[pos[t]] = t;
    st->klab[pos[t]] = labels[t];
  }
  __syncthreads();
  const int n = st->nkeep;
  if (t < n) {
    const int64_t lab = st->klab[t];
    int first = t;
    if (lab >= 80)
      for (int i = 0; i < t; ++i)
        if (st->klab[i] == lab) { first = i; break; }
    st->remap[t] = first;
    st->alive[t] = 1;
    st->rank[t] = t;
  }
}

// bilinear resize of the kept planes only: out[j] = resize(in[kept[j]]), j < nkeep
__global__ __launch_bounds__(256) void k_resize_kept(const float* __restrict__ in,
                                                     float* __restrict__ out,
                                                     const PanState* __restrict__ st, int hi,
                                                     int wi, int ho, int wo) {
  const int j = blockIdx.y;
  if (j >= st->nkeep) return;
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t per_plane = (int64_t)ho * wo;
  if (e >= per_plane) return;
  const int oy = (int)(e / wo), ox = (int)(e - (int64_t)oy * wo);
  // same index / lambda arithmetic as k_bilinear_planar (resize.hip)
  const float sy = (float)hi / (float)ho, sx = (float)wi / (float)wo;
  float fy = sy * ((float)oy + 0.5f) - 0.5f, fx = sx * ((float)ox + 0.5f) - 0.5f;
  if (fy < 0.f) fy = 0.f;
  if (fx < 0.f) fx = 0.f;
  int y0 = (int)fy, x0 = (int)fx;
  if (y0 > hi - 1) y0 = hi - 1;
  if (x0 > wi - 1) x0 = wi - 1;
  const int y1 = y0 + (y0 < hi - 1 ? 1 : 0), x1 = x0 + (x0 < wi - 1 ? 1 : 0);
  const float ly1 = fy - (float)y0, lx1 = fx - (float)x0;
  const float ly0 = 1.f - ly1, lx0 = 1.f - lx1;
  const float* ib = in + (int64_t)st->kept[j] * hi * wi;
  const float v00 = ib[(int64_t)y0 * wi + x0], v01 = ib[(int64_t)y0 * wi + x1];
  const float v10 = ib[(int64_t)y1 * wi + x0], v11 = ib[(int64_t)y1 * wi + x1];
  out[(int64_t)j * per_plane + e] = ly0 * (lx0 * v00 + lx1 * v01) + ly1 * (lx0 * v10 + lx1 * v11);
}

__global__ __launch_bounds__(256) void k_pan_argmax(const float* __restrict__ up,
                                                    const PanState* __restrict__ st,
                                                    int64_t* __restrict__ seg,
                                                    int32_t* __restrict__ area, int64_t HW) {
  if (!st->active) return;
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= HW) return;
  const int n = st->nkeep;
  if (n == 0) { seg[p] = 1; return; }   // torch.ones(mask_size) (:850)
  float best = 0.f;
  int bi = -1;
  for (int i = 0; i < n; ++i) {
    if (!st->alive[i]) continue;
    const float v = up[(int64_t)i * HW + p];
    if (bi < 0 || v > best) { best = v; bi = i; }
  }
  if (bi < 0) return;                   // nothing alive: all_gone is set by the filter
  if (st->first) bi = st->remap[bi];    // merge duplicate stuff classes (:873-878)
  seg[p] = (int64_t)st->rank[bi] * 1000 + st->klab[bi];
  atomicAdd(&area[bi], 1);
}

// One workgroup: drop the segments of area <= 4 (:893-905); nothing to drop -> converged.
// Also clears the area counters for the next round.
__global__ __launch_bounds__(256) void k_pan_filter(PanState* st, int32_t* __restrict__ area) {
  if (!st->active) return;
  const int t = threadIdx.x, n = st->nkeep;
  bool a = t < n && st->alive[t] != 0;
  const bool small = a && area[t] <= 4;
  if (small) a = false;
  area[t] = 0;
  const int some_small = __syncthreads_or(small ? 1 : 0);   // block-wide OR: no racing stores
  if (t == 0) st->first = 0;
  if (!some_small || n == 0) {
    if (t == 0) st->active = 0;
    return;
  }
  if (t < n) st->alive[t] = a ? 1 : 0;
  __syncthreads();
  if (t == 0) {
    int c = 0;
    for (int i = 0; i < n; ++i) { st->rank[i] = c; c += st->alive[i]; }
    st->rounds += 1;
    if (c == 0) { st->all_gone = 1; st->active = 0; }
  }
}

extern "C" int64_t pn_panoptic_state_bytes(void) { return (int64_t)sizeof(PanState); }

static void pan_rounds(PanState* st, const float* up, int32_t* area, int64_t* seg, int64_t HW,
                       int rounds, hipStream_t s) {
  for (int p = 0; p < rounds; ++p) {
    hipLaunchKernelGGL(k_pan_argmax, dim3(pn_cdiv(HW, 256)), dim3(256), 0, s, up, st, seg, area,
                       HW);
    hipLaunchKernelGGL(k_pan_filter, dim3(1), dim3(256), 0, s, st, area);
  }
}

extern "C" int pn_panoptic_device_f32(const float* masks, const int64_t* labels,
                                      const float* scores, int Q, int num_classes, int hi,
                                      int wi, int ho, int wo, void* state, float* up_scratch,
                                      int32_t* area_scratch, int64_t* seg, int rounds,
                                      void* stream) {
  if (!masks || !labels || !scores || !state || !up_scratch || !area_scratch || !seg)
    return PN_BAD_ARG;
  if (Q <= 0 || Q > 256 || rounds < 1 || rounds > 256 || hi <= 0 || wi <= 0 || ho <= 0 ||
      wo <= 0)
    return PN_BAD_ARG;
  hipStream_t s = (hipStream_t)stream;
  PanState* st = (PanState*)state;
  const int64_t HW = (int64_t)ho * wo;
  (void)hipMemsetAsync(area_scratch, 0, sizeof(int32_t) * 256, s);
  hipLaunchKernelGGL(k_pan_select, dim3(1), dim3(256), 0, s, labels, scores, Q, num_classes - 1,
                     st);
  hipLaunchKernelGGL(k_resize_kept, dim3(pn_cdiv(HW, 256), Q), dim3(256), 0, s, masks, up_scratch,
                     st, hi, wi, ho, wo);
  pan_rounds(st, up_scratch, area_scratch, seg, HW, rounds, s);
  return PN_LAUNCH_CHECK();
}

extern "C" int pn_panoptic_continue_f32(void* state, const float* up_scratch,
                                        int32_t* area_scratch, int64_t* seg, int ho, int wo,
                                        int rounds, void* stream) {
  if (!state || !up_scratch || !area_scratch || !seg || ho <= 0 || wo <= 0 || rounds < 1 ||
      rounds > 256)
    return PN_BAD_ARG;
  pan_rounds((PanState*)state, up_scratch, area_scratch, seg, (int64_t)ho * wo, rounds,
             (hipStream_t)stream);
  return PN_LAUNCH_CHECK();
}

// One fixed-shape record per image for the all-gather of predicted triplets:
// [labels 2R | rel_dists R*(C+1) | sub_pos R | obj_pos R] as fp32 (indices < 2^24 are exact).
__global__ __launch_bounds__(256) void k_pack_triplets(const int64_t* __restrict__ labels,
                                                       const float* __restrict__ r_dists,
                                                       const int64_t* __restrict__ sub_pos,
                                                       const int64_t* __restrict__ obj_pos,
                                                       float* __restrict__ rec, int R, int C1) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int n0 = 2 * R, n1 = n0 + R * C1, n2 = n1 + R, n3 = n2 + R;
  if (i >= n3) return;
  float v;
  if (i < n0) v = (float)labels[i];
  else if (i < n1) v = r_dists[i - n0];
  else if (i < n2) v = (float)sub_pos[i - n1];
  else v = (float)obj_pos[i - n2];
  rec[i] = v;
}

extern "C" int pn_pack_triplets_f32(const int64_t* labels, const float* r_dists,
                                    const int64_t* sub_pos, const int64_t* obj_pos, float* rec,
                                    int R, int C1, void* stream) {
  if (!labels || !r_dists || !sub_pos || !obj_pos || !rec || R <= 0 || C1 <= 0) return PN_BAD_ARG;
  const int n = 4 * R + R * C1;
  hipLaunchKernelGGL(k_pack_triplets, dim3(pn_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream,
                     labels, r_dists, sub_pos, obj_pos, rec, R, C1);
  return PN_LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------
// Evaluator feed (pairnet/evaluation/sgg_metrics.py:1276-1380): bit-packed masks and
// the integer counts behind mask_iou (intersection and areas), exact.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pack_bits(const uint8_t* __restrict__ m,
                                                   unsigned long long* __restrict__ words,
                                                   int64_t HW, int64_t nwords) {
  const int64_t row = blockIdx.y;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool on = i < HW && m[row * HW + i] != 0;
  const unsigned long long bal = __ballot(on);
  const int64_t w = i >> 6;
  if ((threadIdx.x & 63) == 0 && w < nwords) words[row * nwords + w] = bal;
}

extern "C" int pn_pack_mask_bits(const uint8_t* masks, uint64_t* words, int64_t rows, int64_t HW,
                                 void* stream) {
  if (!masks || !words || rows <= 0 || rows > 65535 || HW <= 0) return PN_BAD_ARG;
  const int64_t nwords = (HW + 63) / 64;
  hipLaunchKernelGGL(k_pack_bits, dim3(pn_cdiv(nwords * 64, 256), (unsigned)rows), dim3(256), 0,
                     (hipStream_t)stream, masks, (unsigned long long*)words, HW, nwords);
  return PN_LAUNCH_CHECK();
}

// inter[i][j] = popcount(pred_i & gt_j); area_p[i], area_g[j].  One wave per (i, j).
__global__ __launch_bounds__(256) void k_mask_iou_counts(const unsigned long long* __restrict__ pw,
                                                         const unsigned long long* __restrict__ gw,
                                                         int P, int G, int64_t nwords,
                                                         int32_t* __restrict__ inter,
                                                         int32_t* __restrict__ area_p,
                                                         int32_t* __restrict__ area_g) {
  const int lane = threadIdx.x & 63;
  const int pair = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pair >= P * G) return;
  const int i = pair / G, j = pair - i * G;
  const unsigned long long* a = pw + (int64_t)i * nwords;
  const unsigned long long* b = gw + (int64_t)j * nwords;
  int ci = 0, ca = 0, cb = 0;
  for (int64_t w = lane; w < nwords; w += 64) {
    const unsigned long long x = a[w], y = b[w];
    ci += __popcll(x & y);
    ca += __popcll(x);
    cb += __popcll(y);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    ci += __shfl_xor(ci, o, 64);
    ca += __shfl_xor(ca, o, 64);
    cb += __shfl_xor(cb, o, 64);
  }
  if (lane == 0) {
    inter[pair] = ci;
    if (j == 0) area_p[i] = ca;
    if (i == 0) area_g[j] = cb;
  }
}

extern "C" int pn_mask_iou_counts(const uint64_t* pred_words, int P, const uint64_t* gt_words,
                                  int G, int64_t nwords, int32_t* inter, int32_t* area_pred,
                                  int32_t* area_gt, void* stream) {
  if (!pred_words || !gt_words || !inter || !area_pred || !area_gt || P <= 0 || G <= 0 ||
      nwords <= 0)
    return PN_BAD_ARG;
  hipLaunchKernelGGL(k_mask_iou_counts, dim3(pn_cdiv((int64_t)P * G, 4)), dim3(256), 0,
                     (hipStream_t)stream, (const unsigned long long*)pred_words,
                     (const unsigned long long*)gt_words, P, G, nwords, inter, area_pred, area_gt);
  return PN_LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------
// Evaluator feed, part 2: the triplet matching of SGRecall (pairnet/evaluation/
// sgg_metrics.py:173-252 calculate_recall, :1311-1371 _compute_pred_matches_panseg) on the
// device, from the integer counts of pn_mask_iou_counts.
// ---------------------------------------------------------------------------------
// Predicted triplets (sgg_metrics.py:207-209, :1292-1294): predicate = 1 + argmax of
// rel_dists[:, 1:] (first index on ties, numpy argmax), score = that maximum;
// triplet = (labels[r], predicate, labels[R + r]) since rel_pairs[r] = (r, R + r).
__global__ __launch_bounds__(256) void k_pred_triplets(const int64_t* __restrict__ labels,
                                                       const float* __restrict__ r_dists,
                                                       int32_t* __restrict__ trip,
                                                       float* __restrict__ score, int R, int C1) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  const float* x = r_dists + (int64_t)r * C1;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int c = 1 + lane; c < C1; c += 64) {
    const float v = x[c];
    if (v > best || bi == 0x7fffffff) { best = v; bi = c; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (oi != 0x7fffffff && (bi == 0x7fffffff || ob > best || (ob == best && oi < bi))) {
      best = ob; bi = oi;
    }
  }
  if (lane == 0) {
    trip[3 * r + 0] = (int32_t)labels[r];
    trip[3 * r + 1] = bi;
    trip[3 * r + 2] = (int32_t)labels[R + r];
    score[r] = best;
  }
}

extern "C" int pn_pred_triplets(const int64_t* labels, const float* r_dists, int32_t* triplets,
                                float* scores, int R, int C1, void* stream) {
  if (!labels || !r_dists || !triplets || !scores || R <= 0 || C1 < 2) return PN_BAD_ARG;
  hipLaunchKernelGGL(k_pred_triplets, dim3(pn_cdiv(R, 4)), dim3(256), 0, (hipStream_t)stream,
                     labels, r_dists, triplets, scores, R, C1);
  return PN_LAUNCH_CHECK();
}

// words_out[r] = words[a[r]] | words[b[r]]  (union masks of the phrase-detection mode)
__global__ __launch_bounds__(256) void k_or_rows(const unsigned long long* __restrict__ words,
                                                 const int32_t* __restrict__ a,
                                                 const int32_t* __restrict__ b,
                                                 unsigned long long* __restrict__ out,
                                                 int64_t nwords) {
  const int r = blockIdx.y;
  const int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (w >= nwords) return;
  out[(int64_t)r * nwords + w] = words[(int64_t)a[r] * nwords + w] | words[(int64_t)b[r] * nwords + w];
}

extern "C" int pn_mask_or_rows(const uint64_t* words, const int32_t* a, const int32_t* b,
                               uint64_t* out, int rows, int64_t nwords, void* stream) {
  if (!words || !a || !b || !out || rows <= 0 || rows > 65535 || nwords <= 0) return PN_BAD_ARG;
  hipLaunchKernelGGL(k_or_rows, dim3(pn_cdiv(nwords, 256), rows), dim3(256), 0,
                     (hipStream_t)stream, (const unsigned long long*)words, a, b,
                     (unsigned long long*)out, nwords);
  return PN_LAUNCH_CHECK();
}

// match[p][g] = triplet classes equal (the predicate ignored with ignore_rel) and
//   both masks overlap: inter / (area_p + area_g - inter) >= thr for the subject pair AND
//   the object pair (phrdet == 0), or for the one union pair (phrdet != 0: pass the union
//   counts as the "subject" arguments).  inter_s is indexed [p_row][g_row] through the row
//   tables: subject of prediction p = row ps[p], of gt relation g = row gs[g], etc.
// IoU is compared as the reference does (float64 division, >=); an empty union is no match.
__global__ __launch_bounds__(256) void k_triplet_match(
    const int32_t* __restrict__ ptrip, const int32_t* __restrict__ gtrip, int P, int G,
    const int32_t* __restrict__ inter, const int32_t* __restrict__ area_p,
    const int32_t* __restrict__ area_g, int ldi, const int32_t* __restrict__ ps,
    const int32_t* __restrict__ po, const int32_t* __restrict__ gs,
    const int32_t* __restrict__ go, double thr, int phrdet, int ignore_rel,
    uint8_t* __restrict__ match) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= P * G) return;
  const int p = e / G, g = e - p * G;
  bool ok = ptrip[3 * p] == gtrip[3 * g] && ptrip[3 * p + 2] == gtrip[3 * g + 2] &&
            (ignore_rel || ptrip[3 * p + 1] == gtrip[3 * g + 1]);
  auto iou_ok = [&](int pr, int gr) {
    const int it = inter[(int64_t)pr * ldi + gr];
    const int un = area_p[pr] + area_g[gr] - it;
    return un > 0 && (double)it / (double)un >= thr;
  };
  if (ok) ok = iou_ok(ps[p], gs[g]);
  if (ok && !phrdet) ok = iou_ok(po[p], go[g]);
  match[e] = ok ? 1 : 0;
}

extern "C" int pn_triplet_match(const int32_t* pred_triplets, const int32_t* gt_triplets, int P,
                                int G, const int32_t* inter, const int32_t* area_pred,
                                const int32_t* area_gt, int ld_inter, const int32_t* pred_sub_row,
                                const int32_t* pred_obj_row, const int32_t* gt_sub_row,
                                const int32_t* gt_obj_row, double iou_thr, int phrdet,
                                int ignore_rel, uint8_t* match, void* stream) {
  if (!pred_triplets || !gt_triplets || !inter || !area_pred || !area_gt || !pred_sub_row ||
      !gt_sub_row || !match || P <= 0 || G <= 0 || ld_inter <= 0)
    return PN_BAD_ARG;
  if (!phrdet && (!pred_obj_row || !gt_obj_row)) return PN_BAD_ARG;
  hipLaunchKernelGGL(k_triplet_match, dim3(pn_cdiv((int64_t)P * G, 256)), dim3(256), 0,
                     (hipStream_t)stream, pred_triplets, gt_triplets, P, G, inter, area_pred,
                     area_gt, ld_inter, pred_sub_row, pred_obj_row, gt_sub_row, gt_obj_row,
                     iou_thr, phrdet, ignore_rel, match);
  return PN_LAUNCH_CHECK();
}

// ---- the same match on BOXES (sgg_metrics.py `_compute_pred_matches_bbox` :1212-1273, for the
// box-trunk sibling head's results): IoU as mmdet's `bbox_overlaps(mode="iou", eps=1e-6)` in
// float32 -- area = (x2 - x1) * (y2 - y1), overlap = clamp(rb - lt, 0) product, union clamped
// to eps -- and for phrase detection on the union boxes of subject and object (:1239-1259).
__device__ __forceinline__ float box_iou_f32(float ax1, float ay1, float ax2, float ay2, float bx1,
                                             float by1, float bx2, float by2) {
  // no fma contraction: `union - w * h` must round the product first, like the reference's
  // separate tensor ops.  Plain operators under the pragma: the *_rn intrinsics are inline
  // header functions that carry the header's own (contractable) mode into the caller.
#pragma clang fp contract(off)
  const float a1 = (ax2 - ax1) * (ay2 - ay1);
  const float a2 = (bx2 - bx1) * (by2 - by1);
  const float w = fmaxf(fminf(ax2, bx2) - fmaxf(ax1, bx1), 0.f);
  const float h = fmaxf(fminf(ay2, by2) - fmaxf(ay1, by1), 0.f);
  const float ov = w * h;
  const float sum = a1 + a2;
  const float un = fmaxf(sum - ov, 1e-6f);
  return ov / un;
}

__global__ __launch_bounds__(256) void k_triplet_match_boxes(
    const int32_t* __restrict__ ptrip, const int32_t* __restrict__ gtrip, int P, int G,
    const float* __restrict__ pbox, int ldp, const float* __restrict__ gbox, int ldg,
    const int32_t* __restrict__ ps, const int32_t* __restrict__ po,
    const int32_t* __restrict__ gs, const int32_t* __restrict__ go, float thr, int phrdet,
    int ignore_rel, uint8_t* __restrict__ match) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= P * G) return;
  const int p = e / G, g = e - p * G;
  bool ok = ptrip[3 * p] == gtrip[3 * g] && ptrip[3 * p + 2] == gtrip[3 * g + 2] &&
            (ignore_rel || ptrip[3 * p + 1] == gtrip[3 * g + 1]);
  if (ok) {
    const float* a = pbox + (int64_t)ps[p] * ldp;
    const float* b = pbox + (int64_t)po[p] * ldp;
    const float* c = gbox + (int64_t)gs[g] * ldg;
    const float* d = gbox + (int64_t)go[g] * ldg;
    if (phrdet)
      ok = box_iou_f32(fminf(c[0], d[0]), fminf(c[1], d[1]), fmaxf(c[2], d[2]), fmaxf(c[3], d[3]),
                       fminf(a[0], b[0]), fminf(a[1], b[1]), fmaxf(a[2], b[2]),
                       fmaxf(a[3], b[3])) >= thr;
    else
      ok = box_iou_f32(c[0], c[1], c[2], c[3], a[0], a[1], a[2], a[3]) >= thr &&
           box_iou_f32(d[0], d[1], d[2], d[3], b[0], b[1], b[2], b[3]) >= thr;
  }
  match[e] = ok ? 1 : 0;
}

extern "C" int pn_triplet_match_boxes(const int32_t* pred_triplets, const int32_t* gt_triplets,
                                      int P, int G, const float* pred_boxes, int ld_pred,
                                      const float* gt_boxes, int ld_gt,
                                      const int32_t* pred_sub_row, const int32_t* pred_obj_row,
                                      const int32_t* gt_sub_row, const int32_t* gt_obj_row,
                                      float iou_thr, int phrdet, int ignore_rel, uint8_t* match,
                                      void* stream) {
  if (!pred_triplets || !gt_triplets || !pred_boxes || !gt_boxes || !pred_sub_row ||
      !pred_obj_row || !gt_sub_row || !gt_obj_row || !match || P <= 0 || G <= 0 || ld_pred < 4 ||
      ld_gt < 4)
    return PN_BAD_ARG;
  hipLaunchKernelGGL(k_triplet_match_boxes, dim3(pn_cdiv((int64_t)P * G, 256)), dim3(256), 0,
                     (hipStream_t)stream, pred_triplets, gt_triplets, P, G, pred_boxes, ld_pred,
                     gt_boxes, ld_gt, pred_sub_row, pred_obj_row, gt_sub_row, gt_obj_row, iou_thr,
                     phrdet, ignore_rel, match);
  return PN_LAUNCH_CHECK();
}

// ---- result copy with a bounded CU footprint -------------------------------------------
// `triplet2Result` (psgtr.py:15-51) moves 51 MB per 800x1333 image to the host (2R x H0 x W0
// bool masks).  hipMemcpyAsync to pinned memory runs as a chip-wide blit kernel on this
// stack (rocprofv3: __amd_rocclr_copyBuffer, up to 0.9 ms each) that takes workgroup slots
// from the persistent GEMMs of the other streams: +1.1 ms per pipelined step.  PCIe needs no
// width: `wgs` workgroups (16 by default: 16 of 1024 slots) stream the bytes with 16-byte
// loads / stores into the pinned buffer, which is mapped in the device's address space.
// src / dst 16-byte aligned; the last bytes % 16 go one by one.
__global__ __launch_bounds__(256) void k_copy_stream(const uint4* __restrict__ src,
                                                     uint4* __restrict__ dst, int64_t n16,
                                                     int64_t bytes) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  // four independent 16-byte transfers in flight per thread
  for (; i + 3 * stride < n16; i += 4 * stride) {
    const uint4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
    dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
  }
  for (; i < n16; i += stride) dst[i] = src[i];
  if (blockIdx.x == 0) {
    const unsigned char* s8 = reinterpret_cast<const unsigned char*>(src);
    unsigned char* d8 = reinterpret_cast<unsigned char*>(dst);
    for (int64_t t = n16 * 16 + threadIdx.x; t < bytes; t += 256) d8[t] = s8[t];
  }
}

extern "C" int pn_copy_stream(const void* src, void* dst, int64_t bytes, int wgs, void* stream) {
  if (!src || !dst || bytes <= 0 || wgs <= 0 || wgs > 1024 ||
      (((uintptr_t)src | (uintptr_t)dst) & 15))
    return PN_BAD_ARG;
  const int64_t n16 = bytes / 16;
  int grid = (int)((n16 + 255) / 256);
  if (grid < 1) grid = 1;
  if (grid > wgs) grid = wgs;
  hipLaunchKernelGGL(k_copy_stream, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                     (const uint4*)src, (uint4*)dst, n16, bytes);
  return PN_LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------
// triplet2Result's masks (psgtr.py:38-46: 2R x H0 x W0 numpy bool, 49 MB per 800x1333 image)
// cross PCIe as BITS: packed on the device (8 mask bytes -> 1 byte, byte i bit j = element
// 8 i + j), copied (6 MB), expanded to the reference's bool array by the host helper below.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pack_bool_bits(const uint8_t* __restrict__ m,
                                                        uint8_t* __restrict__ bits, int64_t n) {
  const int64_t nb = (n + 7) >> 3, stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nb; i += stride) {
    unsigned long long x = 0;
    if (8 * i + 8 <= n) {
      x = *reinterpret_cast<const unsigned long long*>(m + 8 * i);
    } else {
      for (int j = 0; 8 * i + j < n; ++j) x |= (unsigned long long)m[8 * i + j] << (8 * j);
    }
    // any non-zero byte -> 1, then gather the eight low bits (byte j -> bit j)
    const unsigned long long lo = 0x7f7f7f7f7f7f7f7fULL;
    x = ((((x & lo) + lo) | x) >> 7) & 0x0101010101010101ULL;
    bits[i] = (uint8_t)((x * 0x0102040810204080ULL) >> 56);
  }
}

extern "C" int pn_pack_bool_bits(const uint8_t* bools, uint8_t* bits, int64_t n, void* stream) {
  if (!bools || !bits || n <= 0 || ((uintptr_t)bools & 7)) return PN_BAD_ARG;
  const int64_t nb = (n + 7) / 8;
  int64_t grid = (nb + 255) / 256;
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(k_pack_bool_bits, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream,
                     bools, bits, n);
  return PN_LAUNCH_CHECK();
}

// HOST side of the same transfer (no GPU involved): bits -> one byte (0 / 1) per element,
// split over `threads` host threads; the caller holds no lock (ctypes releases the GIL).
#include <string.h>
#include <thread>
#include <vector>
static void unpack_range(const uint8_t* bits, uint8_t* bools, int64_t b0, int64_t b1, int64_t n,
                         const uint64_t* lut) {
  for (int64_t i = b0; i < b1; ++i) {
    if (8 * i + 8 <= n) {
      memcpy(bools + 8 * i, &lut[bits[i]], 8);
    } else {
      for (int j = 0; 8 * i + j < n; ++j) bools[8 * i + j] = (bits[i] >> j) & 1;
    }
  }
}

extern "C" int pn_unpack_bits_host(const uint8_t* bits, uint8_t* bools, int64_t n, int threads) {
  if (!bits || !bools || n <= 0 || threads < 1 || threads > 64) return PN_BAD_ARG;
  struct Lut {               // byte -> its 8 bits as 8 bytes (little-endian lanes)
    uint64_t v[256];
    Lut() {
      for (int b = 0; b < 256; ++b) {
        v[b] = 0;
        for (int j = 0; j < 8; ++j) v[b] |= (uint64_t)((b >> j) & 1) << (8 * j);
      }
    }
  };
  static const Lut table;    // (C++11: initialised once, thread-safe)
  const uint64_t* lut = table.v;
  const int64_t nb = (n + 7) / 8;
  if (threads == 1 || nb < (1 << 16)) {
    unpack_range(bits, bools, 0, nb, n, lut);
    return 0;
  }
  std::vector<std::thread> pool;
  const int64_t per = (nb + threads - 1) / threads;
  for (int t = 1; t < threads; ++t) {
    const int64_t b0 = t * per, b1 = b0 + per < nb ? b0 + per : nb;
    if (b0 < b1) pool.emplace_back(unpack_range, bits, bools, b0, b1, n, lut);
  }
  unpack_range(bits, bools, 0, per < nb ? per : nb, n, lut);
  for (auto& th : pool) th.join();
  return 0;
}
