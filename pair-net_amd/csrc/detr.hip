// Box-trunk glue kernels: the small, input-dependent steps between the GEMMs / deformable
// attention of the two-stage, box-refining Deformable-DETR trunk under the reference's
// CrossHeadBBox (pairnet/models/relation_heads/pairnet_bbox_head.py:193-359; the trunk itself
// is mmdet's DeformableDetrTransformer, built at :66 -- restated in oracle/deformable_detr.py).
// All of them are row-parallel, HBM-bound and tiny next to the encoder; they exist so that the
// whole step stays on the device with no host round trip.
#include "common.h"

struct MsdaLevelsLite { int h[4], w[4], start[4]; };

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }
// mmdet `inverse_sigmoid(x, eps=1e-5)`
__device__ __forceinline__ float inv_sigmoidf_(float x) {
  x = fminf(fmaxf(x, 0.f), 1.f);
  return logf(fmaxf(x, 1e-5f) / fmaxf(1.f - x, 1e-5f));
}

// ---- out[b][r][:] = valid[b][r] ? x[b][r][:] : 0  (gen_encoder_output_proposals: tokens whose
// proposal box leaves (0.01, 0.99), and padded tokens, are zeroed before enc_output; the
// deformable attentions zero the value rows of padded tokens).  Rows of C floats at stride ld;
// `valid` advances by vstride per image (0: one table for the whole batch); in place allowed. ----
// (x and out may be the same buffer: no __restrict__ on them)
__global__ __launch_bounds__(256) void k_zero_rows(const float* x,
                                                   const uint8_t* __restrict__ valid,
                                                   float* out, int64_t rows, int C4,
                                                   int64_t ld, int64_t vstride) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= rows * C4) return;
  const int64_t r = e / C4;
  const int64_t o = ((int64_t)blockIdx.y * rows + r) * ld + (e - r * C4) * 4;
  const bool v = valid[(int64_t)blockIdx.y * vstride + r];
  if (x == out) {
    if (!v) st4(out + o, make_float4(0.f, 0.f, 0.f, 0.f));
  } else {
    st4(out + o, v ? ld4(x + o) : make_float4(0.f, 0.f, 0.f, 0.f));
  }
}

extern "C" int pn_zero_rows_f32(const float* x, const uint8_t* valid, float* out, int B,
                                int64_t rows, int C, int64_t ld, int64_t valid_bstride,
                                void* stream) {
  if (!x || !valid || !out || B <= 0 || rows <= 0 || C <= 0 || (C & 3) || ld < C || (ld & 3) ||
      valid_bstride < 0 || (((uintptr_t)x | (uintptr_t)out) & 15))
    return PN_BAD_ARG;
  hipLaunchKernelGGL(k_zero_rows, dim3(pn_cdiv(rows * (C / 4), 256), B), dim3(256), 0,
                     (hipStream_t)stream, x, valid, out, rows, C / 4, ld, valid_bstride);
  return PN_LAUNCH_CHECK();
}

// ---- y = sigmoid(x), elementwise (enc_bbox_preds; sigmoid(+inf) = 1) ----
__global__ void k_sigmoid(const float* __restrict__ x, float* __restrict__ y, int64_t n) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e < n) y[e] = sigmoidf_(x[e]);
}

extern "C" int pn_sigmoid_f32(const float* x, float* y, int64_t n, void* stream) {
  if (!x || !y || n <= 0) return PN_BAD_ARG;
  hipLaunchKernelGGL(k_sigmoid, dim3(pn_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, y, n);
  return PN_LAUNCH_CHECK();
}

// ---- two-stage queries: reference boxes and their sine embedding from the selected
// proposals' logits (DeformableDetrTransformer: reference_points = unact.sigmoid();
// get_proposal_pos_embed: 128 features per coordinate, temperature 1e4, scale 2 pi) ----
__global__ void k_box_pos_embed(const float* __restrict__ unact, float* __restrict__ ref,
                                float* __restrict__ emb, int64_t rows) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= rows * 512) return;
  const int64_t r = e >> 9;
  const int c = (int)(e & 511), j = c >> 7, i = c & 127;
  const float s = sigmoidf_(unact[r * 4 + j]);
  if (i == 0) ref[r * 4 + j] = s;
  const float dim_t = powf(10000.f, (float)(2 * (i / 2)) / 128.f);
  const float v = (s * 6.283185307179586f) / dim_t;
  emb[e] = (i & 1) ? cosf(v) : sinf(v);
}

extern "C" int pn_box_pos_embed_f32(const float* unact, float* ref, float* emb, int64_t rows,
                                    void* stream) {
  if (!unact || !ref || !emb || rows <= 0) return PN_BAD_ARG;
  hipLaunchKernelGGL(k_box_pos_embed, dim3(pn_cdiv(rows * 512, 256)), dim3(256), 0,
                     (hipStream_t)stream, unact, ref, emb, rows);
  return PN_LAUNCH_CHECK();
}

// ---- decoder cross-attention operands from the query projection and the reference boxes
// (mmcv MultiScaleDeformableAttention.forward with 4-d reference points):
//   weights  = softmax over the L*4 logits of each head
//   location = ref.xy + offset / 4 * ref.wh * 0.5
// offaw row: [offsets 8*L*4*2 | logits 8*L*4]; thread = (row, head) ----
template <int L>
__global__ __launch_bounds__(256) void k_box_sampling(const float* __restrict__ offaw, int64_t ld,
                                                      const float* __restrict__ ref,
                                                      const float* __restrict__ vr, int rows_per_img,
                                                      float* __restrict__ loc,
                                                      float* __restrict__ aw, int64_t rows) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= rows * 8) return;
  const int64_t r = t >> 3;
  const int h = (int)(t & 7);
  constexpr int NP = L * 4;
  const float* off = offaw + r * ld + h * NP * 2;
  const float* lg = offaw + r * ld + 8 * NP * 2 + h * NP;
  const float cx = ref[r * 4], cy = ref[r * 4 + 1], bw = ref[r * 4 + 2], bh = ref[r * 4 + 3];
  const float* v2 = vr ? vr + (r / rows_per_img) * (L * 2) : nullptr;
  float v[NP], m = -INFINITY;
#pragma unroll
  for (int i = 0; i < NP; ++i) { v[i] = lg[i]; m = fmaxf(m, v[i]); }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NP; ++i) { v[i] = expf(v[i] - m); s += v[i]; }
  float* lo = loc + (r * 8 + h) * NP * 2;
  float* ao = aw + (r * 8 + h) * NP;
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    // reference_points_input = ref * (valid ratio of the sampled level), x and y apart
    const float sx = v2 ? v2[(i >> 2) * 2] : 1.f, sy = v2 ? v2[(i >> 2) * 2 + 1] : 1.f;
    ao[i] = v[i] / s;
    lo[2 * i] = cx * sx + off[2 * i] / 4.f * (bw * sx) * 0.5f;
    lo[2 * i + 1] = cy * sy + off[2 * i + 1] / 4.f * (bh * sy) * 0.5f;
  }
}

extern "C" int pn_box_sampling_f32(const float* offaw, int64_t ld, const float* ref,
                                   const float* valid_ratios, int rows_per_image, float* loc,
                                   float* aw, int64_t rows, int L, void* stream) {
  if (!offaw || !ref || !loc || !aw || rows <= 0 || L <= 0 || L > 4 || ld < 8 * L * 12 ||
      (valid_ratios && rows_per_image <= 0))
    return PN_BAD_ARG;
  const dim3 grid(pn_cdiv(rows * 8, 256));
  hipStream_t s = (hipStream_t)stream;
#define PN_BOX_SAMPLING(LL)                                                                 \
  hipLaunchKernelGGL(k_box_sampling<LL>, grid, dim3(256), 0, s, offaw, ld, ref, valid_ratios, \
                     rows_per_image, loc, aw, rows)
  switch (L) {
    case 1: PN_BOX_SAMPLING(1); break;
    case 2: PN_BOX_SAMPLING(2); break;
    case 3: PN_BOX_SAMPLING(3); break;
    default: PN_BOX_SAMPLING(4); break;
  }
#undef PN_BOX_SAMPLING
  return PN_LAUNCH_CHECK();
}

// ---- encoder self-attention operands on a PADDED batch: the sampling locations and softmax
// weights of every token, with the per-image valid ratios of mmdet's get_reference_points
//   ref(b, token at level lq, sampled level ls) = (x + .5) / (vr[b][lq].x * W_lq) * vr[b][ls].x
//   location = ref + offset / (W_ls, H_ls)
// -> the operands of pn_msda_loc_f32.  (Unpadded batches use the fused pn_msda_f32.) ----
template <int L>
__global__ __launch_bounds__(256) void k_token_sampling(const float* __restrict__ offaw, int64_t ld,
                                                        const float* __restrict__ vr,
                                                        const MsdaLevelsLite lv,
                                                        float* __restrict__ loc,
                                                        float* __restrict__ aw, int64_t N) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= N * 8) return;
  const int b = blockIdx.y;
  const int64_t n = t >> 3;
  const int h = (int)(t & 7);
  constexpr int NP = L * 4;
  int lq = 0;
#pragma unroll
  for (int k = 1; k < L; ++k) if (n >= lv.start[k]) lq = k;
  const int idx = (int)(n - lv.start[lq]);
  const int qy = idx / lv.w[lq], qx = idx - qy * lv.w[lq];
  const float* v2 = vr + (int64_t)b * L * 2;
  const float rx = ((float)qx + 0.5f) / (v2[lq * 2] * (float)lv.w[lq]);
  const float ry = ((float)qy + 0.5f) / (v2[lq * 2 + 1] * (float)lv.h[lq]);
  const int64_t r = (int64_t)b * N + n;
  const float* off = offaw + r * ld + h * NP * 2;
  const float* lg = offaw + r * ld + 8 * NP * 2 + h * NP;
  float v[NP], m = -INFINITY;
#pragma unroll
  for (int i = 0; i < NP; ++i) { v[i] = lg[i]; m = fmaxf(m, v[i]); }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NP; ++i) { v[i] = expf(v[i] - m); s += v[i]; }
  float* lo = loc + (r * 8 + h) * NP * 2;
  float* ao = aw + (r * 8 + h) * NP;
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int ls = i >> 2;
    ao[i] = v[i] / s;
    lo[2 * i] = rx * v2[ls * 2] + off[2 * i] / (float)lv.w[ls];
    lo[2 * i + 1] = ry * v2[ls * 2 + 1] + off[2 * i + 1] / (float)lv.h[ls];
  }
}

extern "C" int pn_token_sampling_f32(const float* offaw, int64_t ld, const float* valid_ratios,
                                     float* loc, float* aw, int B, int L, const int32_t* level_h,
                                     const int32_t* level_w, void* stream) {
  if (!offaw || !valid_ratios || !loc || !aw || B <= 0 || L <= 0 || L > 4 || !level_h ||
      !level_w || ld < 8 * L * 12)
    return PN_BAD_ARG;
  MsdaLevelsLite lv{};
  int64_t n = 0;
  for (int l = 0; l < L; ++l) {
    if (level_h[l] <= 0 || level_w[l] <= 0) return PN_BAD_ARG;
    lv.h[l] = level_h[l]; lv.w[l] = level_w[l]; lv.start[l] = (int)n;
    n += (int64_t)level_h[l] * level_w[l];
  }
  const dim3 grid(pn_cdiv(n * 8, 256), B);
  hipStream_t s = (hipStream_t)stream;
#define PN_TOKEN_SAMPLING(LL)                                                                  \
  hipLaunchKernelGGL(k_token_sampling<LL>, grid, dim3(256), 0, s, offaw, ld, valid_ratios, lv, \
                     loc, aw, n)
  switch (L) {
    case 1: PN_TOKEN_SAMPLING(1); break;
    case 2: PN_TOKEN_SAMPLING(2); break;
    case 3: PN_TOKEN_SAMPLING(3); break;
    default: PN_TOKEN_SAMPLING(4); break;
  }
#undef PN_TOKEN_SAMPLING
  return PN_LAUNCH_CHECK();
}

// ---- iterative box refinement (DeformableDetrTransformerDecoder.forward):
// ref_out = sigmoid(delta + inverse_sigmoid(ref_in)) ----
__global__ void k_box_refine(const float* __restrict__ delta, const float* __restrict__ ref_in,
                             float* __restrict__ ref_out, int64_t n) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e < n) ref_out[e] = sigmoidf_(delta[e] + inv_sigmoidf_(ref_in[e]));
}

extern "C" int pn_box_refine_f32(const float* delta, const float* ref_in, float* ref_out,
                                 int64_t rows, void* stream) {
  if (!delta || !ref_in || !ref_out || rows <= 0) return PN_BAD_ARG;
  hipLaunchKernelGGL(k_box_refine, dim3(pn_cdiv(rows * 4, 256)), dim3(256), 0,
                     (hipStream_t)stream, delta, ref_in, ref_out, rows * 4);
  return PN_LAUNCH_CHECK();
}

// ---- query ranking of CrossHeadBBox.forward (pairnet_bbox_head.py:252-254):
// score[b][q] = max_c softmax_over_QUERIES(logits[b][:, c])[q]   (dim=1 is the query axis) ----
__global__ __launch_bounds__(256) void k_query_score(const float* __restrict__ logits,
                                                     float* __restrict__ score, int Nq, int C) {
  __shared__ float cmax[256], csum[256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* lb = logits + (int64_t)blockIdx.x * Nq * C;
  if (tid < C) {
    float m = -INFINITY;
    for (int q = 0; q < Nq; ++q) m = fmaxf(m, lb[(int64_t)q * C + tid]);
    float s = 0.f;
    for (int q = 0; q < Nq; ++q) s += expf(lb[(int64_t)q * C + tid] - m);
    cmax[tid] = m;
    csum[tid] = s;
  }
  __syncthreads();
  for (int q = wave; q < Nq; q += 4) {
    float best = -INFINITY;
    for (int c = lane; c < C; c += 64)
      best = fmaxf(best, expf(lb[(int64_t)q * C + c] - cmax[c]) / csum[c]);
    best = wave_max(best);
    if (lane == 0) score[(int64_t)blockIdx.x * Nq + q] = best;
  }
}

extern "C" int pn_query_score_f32(const float* logits, float* score, int B, int Nq, int C,
                                  void* stream) {
  if (!logits || !score || B <= 0 || Nq <= 0 || C <= 0 || C > 256) return PN_BAD_ARG;
  hipLaunchKernelGGL(k_query_score, dim3(B), dim3(256), 0, (hipStream_t)stream, logits, score, Nq,
                     C);
  return PN_LAUNCH_CHECK();
}

// ---- CrossHeadBBox._get_bboxes_single (pairnet_bbox_head.py:1056-1086): per subject / object
// row: label = argmax softmax + 1, score = max softmax, box = cxcywh -> xyxy, scaled to the
// image, clamped, optionally divided by scale_factor.  One wave per row of [subjects | objects].
__global__ __launch_bounds__(256) void k_box_triplets(const float* __restrict__ s_cls,
                                                      const float* __restrict__ o_cls,
                                                      const float* __restrict__ s_box,
                                                      const float* __restrict__ o_box,
                                                      float* __restrict__ det,
                                                      int64_t* __restrict__ labels, int R, int C,
                                                      float img_h, float img_w, float4 sf,
                                                      int rescale) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= 2 * R) return;
  const bool subj = row < R;
  const int r = subj ? row : row - R;
  const float* lg = (subj ? s_cls : o_cls) + (int64_t)r * C;
  float m = -INFINITY;
  int am = 0;
  for (int c = lane; c < C; c += 64) {
    const float v = lg[c];
    if (v > m) { m = v; am = c; }
  }
  // wave argmax, ties -> smaller class index
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float om = __shfl_xor(m, o, 64);
    const int oa = __shfl_xor(am, o, 64);
    if (om > m || (om == m && oa < am)) { m = om; am = oa; }
  }
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += expf(lg[c] - m);
  s = wave_sum(s);
  if (lane == 0) {
    const float* b = (subj ? s_box : o_box) + (int64_t)r * 4;
    const float cx = b[0], cy = b[1], w = b[2], h = b[3];
    float x1 = (cx - 0.5f * w) * img_w, y1 = (cy - 0.5f * h) * img_h;
    float x2 = (cx + 0.5f * w) * img_w, y2 = (cy + 0.5f * h) * img_h;
    x1 = fminf(fmaxf(x1, 0.f), img_w); x2 = fminf(fmaxf(x2, 0.f), img_w);
    y1 = fminf(fmaxf(y1, 0.f), img_h); y2 = fminf(fmaxf(y2, 0.f), img_h);
    if (rescale) { x1 /= sf.x; y1 /= sf.y; x2 /= sf.z; y2 /= sf.w; }
    float* d = det + (int64_t)row * 5;
    d[0] = x1; d[1] = y1; d[2] = x2; d[3] = y2;
    d[4] = 1.f / s;                         // exp(m - m) / sum
    labels[row] = am + 1;
  }
}

extern "C" int pn_box_triplets_f32(const float* s_cls, const float* o_cls, const float* s_box,
                                   const float* o_box, float* det, int64_t* labels, int R, int C,
                                   float img_h, float img_w, const float* scale_factor,
                                   int rescale, void* stream) {
  if (!s_cls || !o_cls || !s_box || !o_box || !det || !labels || R <= 0 || C <= 0 ||
      (rescale && !scale_factor))
    return PN_BAD_ARG;
  float4 sf = make_float4(1.f, 1.f, 1.f, 1.f);
  if (scale_factor) sf = make_float4(scale_factor[0], scale_factor[1], scale_factor[2], scale_factor[3]);
  hipLaunchKernelGGL(k_box_triplets, dim3(pn_cdiv(2 * R, 4)), dim3(256), 0, (hipStream_t)stream,
                     s_cls, o_cls, s_box, o_box, det, labels, R, C, img_h, img_w, sf, rescale);
  return PN_LAUNCH_CHECK();
}
