// Pair Proposal Network kernels: Matrix Learner edge layers, top-k pair selection,
// row gathers.  (The 64->64 middle layer of the Matrix Learner is the implicit-GEMM
// convolution in gemm.hip.)
#include "common.h"

// ---- first layer: 1 -> C (C == 64), 7x7, pad 3, ReLU; out channel-last ----
__global__ __launch_bounds__(256) void k_ml_first(const float* __restrict__ in,
                                                  const float* __restrict__ w1,
                                                  const float* __restrict__ b1,
                                                  float* __restrict__ out, int S) {
  __shared__ float ws[49 * 64];
  const int tid = threadIdx.x, c = tid & 63, pl = tid >> 6;
  for (int e = tid; e < 49 * 64; e += 256) ws[(e % 49) * 64 + (e / 49)] = w1[e];  // [tap][c]
  __syncthreads();
  const int b = blockIdx.y;
  const int pix = blockIdx.x * 4 + pl;
  if (pix >= S * S) return;
  const int y = pix / S, x = pix - y * S;
  const float* ib = in + (int64_t)b * S * S;
  float acc = 0.f;
#pragma unroll
  for (int ky = 0; ky < 7; ++ky) {
    const int yy = y + ky - 3;
    if (yy < 0 || yy >= S) continue;
#pragma unroll
    for (int kx = 0; kx < 7; ++kx) {
      const int xx = x + kx - 3;
      if (xx < 0 || xx >= S) continue;
      acc += ib[yy * S + xx] * ws[(ky * 7 + kx) * 64 + c];
    }
  }
  out[((int64_t)b * S * S + pix) * 64 + c] = fmaxf(acc + b1[c], 0.f);
}

extern "C" int pn_mlearner_first_f32(const float* in, const float* w1, const float* b1,
                                     float* out, int B, int S, int C, void* stream) {
  if (!in || !w1 || !b1 || !out || C != 64 || B <= 0 || S <= 0) return PN_BAD_ARG;
  hipLaunchKernelGGL(k_ml_first, dim3(pn_cdiv((int64_t)S * S, 4), B), dim3(256), 0,
                     (hipStream_t)stream, in, w1, b1, out, S);
  return PN_LAUNCH_CHECK();
}

// ---- fused front of the Pair Proposal Network -------------------------------------
// pairnet_head.py:325-333 + cnn_factory.py:22-29 in one kernel: F.normalize of the subject /
// object embeddings, the Q x Q cosine matrix, and the Matrix Learner's first layer
// (7x7, 1 -> 64 channels, ReLU).  A workgroup owns an 8 x 8 tile of (subject i, object j)
// pairs (169 workgroups at Q = 100): the 14 subject rows and 14 object rows its 7x7 halo reaches are staged in LDS as
// L2-NORMALISED query tiles, their 14 x 14 block of cosines is one 32 x 32 MFMA tile (the
// four waves split K = 256 and reduce through LDS in wave order), the block stays in LDS
// and every lane = output channel convolves it with its 49 weights from registers.  Rows
// outside [0, Q) are the convolution's zero padding.
#define PPN_T 8
#define PPN_H (PPN_T + 6)
__global__ __launch_bounds__(256) void k_ppn_front(const float* __restrict__ se,
                                                   const float* __restrict__ oe,
                                                   const float* __restrict__ w1,
                                                   const float* __restrict__ b1,
                                                   float* __restrict__ raw_out,
                                                   float* __restrict__ c1, int Q, float eps) {
  __shared__ __attribute__((aligned(16))) float sS[32 * 260], sO[32 * 260];
  __shared__ float red[3][1024];
  __shared__ float raw[PPN_H][PPN_H + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.z, i0 = blockIdx.y * PPN_T - 3, j0 = blockIdx.x * PPN_T - 3;
  const float* sb = se + (int64_t)b * Q * 256;
  const float* ob = oe + (int64_t)b * Q * 256;
  // ---- stage the normalised rows: one wave per row (k_l2norm256's arithmetic), rows
  // beyond the halo / outside [0, Q) are zero ----
  for (int r = wave; r < 64; r += 4) {
    const bool subj = r < 32;
    const int rr = subj ? r : r - 32;
    const int q = (subj ? i0 : j0) + rr;
    float4 y = make_float4(0.f, 0.f, 0.f, 0.f);
    if (rr < PPN_H && q >= 0 && q < Q) {
      const float4 v = ld4((subj ? sb : ob) + (int64_t)q * 256 + lane * 4);
      const float n = sqrtf(wave_sum((v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w)));
      const float d = fmaxf(n, eps);
      y = make_float4(v.x / d, v.y / d, v.z / d, v.w / d);
    }
    st4((subj ? sS : sO) + rr * 260 + lane * 4, y);
  }
  __syncthreads();
  // ---- cosines: 32 x 32 tile, wave w contracts k in [64 w, 64 w + 64) ----
  {
    const int li = lane & 31, lh = lane >> 5;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float* ar = sS + li * 260 + wave * 64 + 4 * lh;
    const float* br = sO + li * 260 + wave * 64 + 4 * lh;
#pragma unroll
    for (int s8 = 0; s8 < 8; ++s8) {
      const float4 av = ld4(ar + 8 * s8), bv = ld4(br + 8 * s8);
      acc = mfma32(av.x, bv.x, acc);
      acc = mfma32(av.y, bv.y, acc);
      acc = mfma32(av.z, bv.z, acc);
      acc = mfma32(av.w, bv.w, acc);
    }
    if (wave > 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) red[wave - 1][r * 64 + lane] = acc[r];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = ((acc[r] + red[0][r * 64 + lane]) + red[1][r * 64 + lane]) + red[2][r * 64 + lane];
        const int row = mfma32_row(r, lh);           // subject index inside the halo block
        if (row < PPN_H && li < PPN_H) {
          raw[row][li] = v;
          const int gi = i0 + row, gj = j0 + li;
          // the tile's own interior is this workgroup's part of the Q x Q matrix
          if (row >= 3 && row < 3 + PPN_T && li >= 3 && li < 3 + PPN_T && gi < Q && gj < Q)
            raw_out[((int64_t)b * Q + gi) * Q + gj] = v;
        }
      }
    }
  }
  __syncthreads();
  // ---- first Matrix Learner layer: lane = output channel, 49 weights in registers; the
  // taps are accumulated in k_ml_first's order (ky, kx ascending; out-of-matrix taps are
  // exact zeros here - the staged rows outside [0, Q) are zero - and add nothing) ----
  float wt[49];
#pragma unroll
  for (int t = 0; t < 49; ++t) wt[t] = w1[lane * 49 + t];
  const float bias = b1[lane];
  for (int p = wave; p < PPN_T * PPN_T; p += 4) {
    const int y = p / PPN_T, x = p - y * PPN_T;
    const int gi = i0 + 3 + y, gj = j0 + 3 + x;
    if (gi >= Q || gj >= Q) continue;               // (uniform per wave)
    float acc = 0.f;
#pragma unroll
    for (int ky = 0; ky < 7; ++ky)
#pragma unroll
      for (int kx = 0; kx < 7; ++kx) acc += raw[y + ky][x + kx] * wt[ky * 7 + kx];
    c1[(((int64_t)b * Q + gi) * Q + gj) * 64 + lane] = fmaxf(acc + bias, 0.f);
  }
}

extern "C" int pn_ppn_front_f32(const float* sub_embed, const float* obj_embed, const float* w1,
                                const float* b1, float* importance_raw, float* c1, int B, int Q,
                                float eps, void* stream) {
  if (!sub_embed || !obj_embed || !w1 || !b1 || !importance_raw || !c1 || B <= 0 || Q <= 0 ||
      B > 65535 || (((uintptr_t)sub_embed | (uintptr_t)obj_embed) & 15))
    return PN_BAD_ARG;
  const int nt = pn_cdiv(Q, PPN_T);
  hipLaunchKernelGGL(k_ppn_front, dim3(nt, nt, B), dim3(256), 0, (hipStream_t)stream, sub_embed,
                     obj_embed, w1, b1, importance_raw, c1, Q, eps);
  return PN_LAUNCH_CHECK();
}

// ---- last layer: C -> 1 (C == 64), 7x7, pad 3 ----
// A workgroup owns an 8 x 8 tile of output pixels: the 14 x 14 x 64 input patch its halo
// reaches is staged in LDS once (coalesced 256-byte rows; positions outside the map are the
// convolution's zero padding) together with the 49 x 64 weights, then each of the 4 waves
// takes 16 pixels with lane = input channel: 49 LDS multiply-adds in (ky, kx) order and one
// butterfly sum per pixel.  (Round 2 ran one wave per pixel straight from L2: 49 dependent
// 256-byte loads per wave, 30 us for the 100 x 100 map; this form takes ~8.)
#define MLL_T 8
#define MLL_H (MLL_T + 6)
__global__ __launch_bounds__(256) void k_ml_last(const float* __restrict__ in,
                                                 const float* __restrict__ w3,
                                                 const float* __restrict__ b3,
                                                 float* __restrict__ out, int S) {
  __shared__ __attribute__((aligned(16))) float patch[MLL_H * MLL_H * 64];
  __shared__ __attribute__((aligned(16))) float wt[49 * 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.z, y0 = blockIdx.y * MLL_T - 3, x0 = blockIdx.x * MLL_T - 3;
  const float* ib = in + (int64_t)b * S * S * 64;
  for (int e = tid; e < MLL_H * MLL_H * 16; e += 256) {     // one float4 per (position, 4 ch)
    const int pos = e >> 4, c4 = (e & 15) * 4;
    const int yy = y0 + pos / MLL_H, xx = x0 + pos % MLL_H;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (yy >= 0 && yy < S && xx >= 0 && xx < S) v = ld4(ib + ((int64_t)yy * S + xx) * 64 + c4);
    st4(patch + pos * 64 + c4, v);
  }
  for (int e = tid; e < 49 * 16; e += 256) st4(wt + e * 4, ld4(w3 + e * 4));
  __syncthreads();
  const float bias = b3[0];
  for (int p = wave; p < MLL_T * MLL_T; p += 4) {
    const int y = p / MLL_T, x = p - y * MLL_T;
    const int gy = y0 + 3 + y, gx = x0 + 3 + x;
    if (gy >= S || gx >= S) continue;                       // (uniform per wave)
    float acc = 0.f;
#pragma unroll
    for (int ky = 0; ky < 7; ++ky)
#pragma unroll
      for (int kx = 0; kx < 7; ++kx)
        acc += patch[((y + ky) * MLL_H + (x + kx)) * 64 + lane] * wt[(ky * 7 + kx) * 64 + lane];
    acc = wave_sum(acc);
    if (lane == 0) out[(int64_t)b * S * S + (int64_t)gy * S + gx] = acc + bias;
  }
}

extern "C" int pn_mlearner_last_f32(const float* in, const float* w3, const float* b3,
                                    float* out, int B, int S, int C, void* stream) {
  if (!in || !w3 || !b3 || !out || C != 64 || B <= 0 || S <= 0 || B > 65535 ||
      (((uintptr_t)in | (uintptr_t)w3) & 15))
    return PN_BAD_ARG;
  const int nt = pn_cdiv(S, MLL_T);
  hipLaunchKernelGGL(k_ml_last, dim3(nt, nt, B), dim3(256), 0, (hipStream_t)stream, in, w3, b3,
                     out, S);
  return PN_LAUNCH_CHECK();
}

// ---- top-k pair selection ------------------------------------------------------
// 48-bit unique keys: (order-preserving float bits << 16) | (0xFFFF - flat index), so
// "larger key" == larger score, ties -> smaller index.  The keys live in registers (formed
// once); an MSB-first 8-bit radix select (LDS histogram + one-wave suffix scan per pass,
// stopping as soon as a whole bucket is taken) finds the key of the k-th largest; the k
// survivors are compacted into LDS and ordered by rank counting (unique keys).
__device__ __forceinline__ unsigned long long topk_key(float v, int idx) {
  uint32_t u = __float_as_uint(v + 0.0f);  // -0 -> +0
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  return ((unsigned long long)u << 16) | (unsigned long long)(0xFFFF - idx);
}

// KPT = keys per thread held in registers (n <= 1024 * KPT); the keys are formed once.
template <int KPT>
__global__ __launch_bounds__(1024) void k_topk_pairs(const float* __restrict__ scores,
                                                     int64_t* __restrict__ idx_out,
                                                     int64_t* __restrict__ sub_out,
                                                     int64_t* __restrict__ obj_out,
                                                     int64_t* __restrict__ pair_out, int n, int Q,
                                                     int k, int64_t estride, int64_t rstride,
                                                     int cap) {
  // scores of row b: scores[b * rstride + i * estride]; cap = 256 or 512 >= k
  __shared__ int hist[256];
  __shared__ unsigned long long sel[512];
  __shared__ unsigned long long s_prefix;
  __shared__ int s_remaining, s_count, s_done;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* sc = scores + (int64_t)blockIdx.x * rstride;
  unsigned long long key[KPT];
#pragma unroll
  for (int j = 0; j < KPT; ++j) {
    const int i = tid + 1024 * j;
    key[j] = i < n ? topk_key(sc[(int64_t)min(i, n - 1) * estride], i) : 0ull;
  }
  if (tid == 0) { s_prefix = 0ull; s_remaining = k; s_count = 0; s_done = 0; }
  __syncthreads();
  for (int pass = 0; pass < 6; ++pass) {
    const int shift = 40 - 8 * pass;
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    const unsigned long long prefix = s_prefix;
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
      if (tid + 1024 * j < n && (pass == 0 || (key[j] >> (shift + 8)) == (prefix >> (shift + 8))))
        atomicAdd(&hist[(int)((key[j] >> shift) & 255ull)], 1);
    }
    __syncthreads();
    if (wave == 0) {
      const int remaining = s_remaining;
      const int h0 = hist[4 * lane], h1 = hist[4 * lane + 1];
      const int h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
      const int s = (h0 + h1) + (h2 + h3);
      int suf = s;  // inclusive suffix sum over lanes >= this one
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_down(suf, o, 64);
        if (lane + o < 64) suf += t;
      }
      const int above = suf - s;
      if (above < remaining && remaining <= suf) {
        int cum = above, digit, hb[4] = {h0, h1, h2, h3};
        digit = 4 * lane;
#pragma unroll
        for (int bq = 3; bq >= 0; --bq) {
          if (cum + hb[bq] >= remaining) { digit = 4 * lane + bq; break; }
          cum += hb[bq];
        }
        s_prefix = prefix | ((unsigned long long)digit << shift);
        s_remaining = remaining - cum;
        // the whole bucket of this digit is taken: every key >= the prefix (lower bits zero)
        // is selected and nothing remains to refine (distinct scores get here after the
        // passes over the 32 value bits: four instead of six)
        if (hb[digit & 3] == remaining - cum) s_done = 1;
      }
    }
    __syncthreads();
    if (s_done) break;
  }
  const unsigned long long kth = s_prefix;
#pragma unroll
  for (int j = 0; j < KPT; ++j) {
    if (tid + 1024 * j < n && key[j] >= kth) {
      const int slot = atomicAdd(&s_count, 1);
      if (slot < cap) sel[slot] = key[j];
    }
  }
  __syncthreads();
  // order: the keys are unique, so the rank of a selected key = the number of selected keys
  // above it; thread t counts for key t (k broadcast LDS reads, no barrier) and writes its
  // row of the output at that rank
  if (tid < k) {
    const unsigned long long mine = sel[tid];
    int rank = 0;
    for (int j = 0; j < k; ++j) rank += sel[j] > mine ? 1 : 0;
    const int64_t idx = 0xFFFF - (int)(mine & 0xFFFFull);
    const int64_t o = (int64_t)blockIdx.x * k + rank;
    idx_out[o] = idx;
    sub_out[o] = idx / Q;
    obj_out[o] = idx % Q;
    if (pair_out) {   // [B][sub k | obj k]: the row list of the pair-feature gather
      pair_out[(int64_t)blockIdx.x * 2 * k + rank] = idx / Q;
      pair_out[(int64_t)blockIdx.x * 2 * k + k + rank] = idx % Q;
    }
  }
}

static void launch_topk(const float* scores, int64_t* idx, int64_t* quot, int64_t* rem,
                        int64_t* pair, int B, int n, int div, int k, int64_t estride,
                        int64_t rstride, int cap, hipStream_t s) {
#define PN_TOPK(KPT)                                                                           \
  hipLaunchKernelGGL(k_topk_pairs<KPT>, dim3(B), dim3(1024), 0, s, scores, idx, quot, rem, pair, \
                     n, div, k, estride, rstride, cap)
  if (n <= 1024 * 10) PN_TOPK(10);
  else if (n <= 1024 * 24) PN_TOPK(24);
  else if (n <= 1024 * 40) PN_TOPK(40);
  else PN_TOPK(64);
#undef PN_TOPK
}

extern "C" int pn_topk_pairs(const float* scores, int64_t* idx, int64_t* sub, int64_t* obj,
                             int64_t* pair, int B, int Q, int k, void* stream) {
  if (!scores || !idx || !sub || !obj || B <= 0 || Q <= 0) return PN_BAD_ARG;
  if ((int64_t)Q * Q > 65536 || k <= 0 || k > 256 || k > Q * Q) return PN_BAD_ARG;
  launch_topk(scores, idx, sub, obj, pair, B, Q * Q, Q, k, 1, (int64_t)Q * Q, 256,
              (hipStream_t)stream);
  return PN_LAUNCH_CHECK();
}

// General form: the k largest of n scores per row, sorted descending (ties -> smaller
// index); quot = idx / div, rem = idx % div.  (CrossHeadBaseline's triplet ranking,
// pairnet/models/relation_heads/baseline.py:1033-1037: n = R * num_relations.)
extern "C" int pn_topk_f32(const float* scores, int64_t* idx, int64_t* quot, int64_t* rem, int B,
                           int n, int div, int k, void* stream) {
  if (!scores || !idx || !quot || !rem || B <= 0 || n <= 0 || div <= 0) return PN_BAD_ARG;
  if (n > 65536 || k <= 0 || k > 256 || k > n) return PN_BAD_ARG;
  launch_topk(scores, idx, quot, rem, nullptr, B, n, div, k, 1, (int64_t)n, 256,
              (hipStream_t)stream);
  return PN_LAUNCH_CHECK();
}

// Strided form with k <= 512: score i of row b is scores[b * row_stride + i * elem_stride]
// (the two-stage proposal selection of a Deformable-DETR trunk: the 300 best of column 0 of
// the per-token class logits, `torch.topk(enc_outputs_class[..., 0], 300, dim=1)` behind
// pairnet_bbox_head.py:215-228).
extern "C" int pn_topk_strided_f32(const float* scores, int64_t elem_stride, int64_t row_stride,
                                   int64_t* idx, int64_t* quot, int64_t* rem, int B, int n,
                                   int div, int k, void* stream) {
  if (!scores || !idx || !quot || !rem || B <= 0 || n <= 0 || div <= 0 || elem_stride <= 0)
    return PN_BAD_ARG;
  if (n > 65536 || k <= 0 || k > 512 || k > n) return PN_BAD_ARG;
  launch_topk(scores, idx, quot, rem, nullptr, B, n, div, k, elem_stride, row_stride,
              k > 256 ? 512 : 256, (hipStream_t)stream);
  return PN_LAUNCH_CHECK();
}

// ---- row gather ----------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gather_rows(const float* __restrict__ in,
                                                     const int64_t* __restrict__ index,
                                                     float* __restrict__ out, int rows_in,
                                                     int rows_out, int64_t len, int vec) {
  const int r = blockIdx.y, b = blockIdx.z;
  int64_t src = index[(int64_t)b * rows_out + r];
  if (src < 0) src = 0;
  if (src >= rows_in) src = rows_in - 1;
  const float* ip = in + ((int64_t)b * rows_in + src) * len;
  float* op = out + ((int64_t)b * rows_out + r) * len;
  if (vec) {
    const int64_t n4 = len >> 2;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n4; e += (int64_t)gridDim.x * 256)
      st4(op + 4 * e, ld4(ip + 4 * e));
  } else {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < len; e += (int64_t)gridDim.x * 256)
      op[e] = ip[e];
  }
}

extern "C" int pn_gather_rows_f32(const float* in, const int64_t* index, float* out, int B,
                                  int rows_in, int rows_out, int64_t len, void* stream) {
  if (!in || !index || !out || B <= 0 || rows_in <= 0 || rows_out <= 0 || len <= 0)
    return PN_BAD_ARG;
  const int vec = (len % 4 == 0) && (((uintptr_t)in | (uintptr_t)out) & 15) == 0;
  int gx = pn_cdiv(vec ? len / 4 : len, 256);
  if (gx > 64) gx = 64;
  hipLaunchKernelGGL(k_gather_rows, dim3(gx, rows_out, B), dim3(256), 0, (hipStream_t)stream,
                     in, index, out, rows_in, rows_out, len, vec);
  return PN_LAUNCH_CHECK();
}
