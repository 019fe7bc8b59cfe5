// Backward kernels of Pair-Net's own tail (second slice): the Relation Fusion
// decoder (pairnet_head.py:353-378, layer semantics facebook_detr.py:378-432), the Pair Proposal
// Network's two MLPs and cosine block (pairnet_head.py:322-333) and the Matrix Learner ConvTiny
// (frameworks/cnn_factory.py:6-53).  pair-net_amd/grad.py composes them with the library's
// forward kernels: the large contractions of a linear layer's backward (dX = dY W, dW = dY^T X)
// are pn_gemm_f32 calls on transposed operands (pn_transpose_f32 here; A read column-major
// there), the 64 -> 64 convolution's data gradient is pn_conv2d_nhwc_ex_f32 on a re-laid-out
// weight (pn_conv_weight_bwd_layout_f32).  What is new here is everything else: reductions over
// rows / batch, LayerNorm / softmax-attention / ReLU / L2-normalise derivatives, the scatter that
// undoes a row gather, and the convolutions' weight gradients (MFMA outer products over pixels).
// All sums run in a fixed order (no atomics): a gradient is bitwise reproducible.
#include "common.h"

// ---- out[c][r] = in[r][c], rows of `out` zero-filled from `rows` up to `out_cols` -----------
__global__ __launch_bounds__(256) void k_transpose(const float* __restrict__ in, int64_t ldi,
                                                   float* __restrict__ out, int64_t ldo,
                                                   int rows, int cols, int out_cols) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  for (int j = ty; j < 32; j += 8) {
    const int r = r0 + j, c = c0 + tx;
    tile[j][tx] = (r < rows && c < cols) ? in[(int64_t)r * ldi + c] : 0.f;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int c = c0 + j, r = r0 + tx;
    if (c < cols && r < out_cols) out[(int64_t)c * ldo + r] = tile[tx][j];
  }
}

extern "C" int pn_transpose_f32(const float* in, int64_t ldi, float* out, int64_t ldo, int rows,
                                int cols, int out_cols, void* stream) {
  if (!in || !out || rows <= 0 || cols <= 0 || out_cols < rows || ldo < out_cols || ldi < cols)
    return PN_BAD_ARG;
  if (pn_cdiv(out_cols, 32) > 65535) return PN_BAD_ARG;
  hipLaunchKernelGGL(k_transpose, dim3(pn_cdiv(cols, 32), pn_cdiv(out_cols, 32)), dim3(256), 0,
                     (hipStream_t)stream, in, ldi, out, ldo, rows, cols, out_cols);
  return PN_LAUNCH_CHECK();
}

// ---- out[c] (+)= sum_r x[r][c]: a bias gradient, a LayerNorm weight gradient, the second
// stage of the tap correlations.  64 columns per workgroup; 16 waves stride the rows, their
// partials are added in wave order.  Tall matrices (the pixel decoder's 21 950 token rows) are cut
// into row chunks (blockIdx.y) whose sums go to a caller scratch [chunks][cols] and are summed
// by a second launch of the same kernel: fixed order either way.
__global__ __launch_bounds__(1024) void k_colsum(const float* __restrict__ x, int64_t ld,
                                                 float* __restrict__ out, int rows, int cols,
                                                 int accumulate, int rows_per_chunk) {
  __shared__ float red[16][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  const int r0 = blockIdx.y * rows_per_chunk, r1 = min(r0 + rows_per_chunk, rows);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < cols) {
    const float* p = x + c;
    int r = r0 + wave;
    for (; r + 48 < r1; r += 64) {
      s0 += p[(int64_t)r * ld];
      s1 += p[(int64_t)(r + 16) * ld];
      s2 += p[(int64_t)(r + 32) * ld];
      s3 += p[(int64_t)(r + 48) * ld];
    }
    for (; r < r1; r += 16) s0 += p[(int64_t)r * ld];
  }
  red[wave][lane] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (wave == 0 && c < cols) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += red[w][lane];
    float* o = out + (int64_t)blockIdx.y * cols + c;         // (blockIdx.y > 0 only into scratch)
    *o = accumulate ? *o + t : t;
  }
}

extern "C" int pn_colsum_f32(const float* x, int64_t ld, float* out, int rows, int cols,
                             int accumulate, float* scratch, int64_t scratch_floats,
                             void* stream) {
  if (!x || !out || rows <= 0 || cols <= 0 || ld < cols) return PN_BAD_ARG;
  hipStream_t s = (hipStream_t)stream;
  int chunks = scratch ? min(64, rows / 512) : 1;
  if (chunks > 1 && (int64_t)chunks * cols > scratch_floats) chunks = (int)(scratch_floats / cols);
  if (chunks < 2) {
    hipLaunchKernelGGL(k_colsum, dim3(pn_cdiv(cols, 64), 1), dim3(1024), 0, s, x, ld, out, rows,
                       cols, accumulate, rows);
    return PN_LAUNCH_CHECK();
  }
  const int per = pn_cdiv(rows, chunks);
  chunks = pn_cdiv(rows, per);
  hipLaunchKernelGGL(k_colsum, dim3(pn_cdiv(cols, 64), chunks), dim3(1024), 0, s, x, ld, scratch,
                     rows, cols, 0, per);
  hipLaunchKernelGGL(k_colsum, dim3(pn_cdiv(cols, 64), 1), dim3(1024), 0, s, scratch, (int64_t)cols,
                     out, chunks, cols, accumulate, chunks);
  return PN_LAUNCH_CHECK();
}

// ---- dx = dy where y > 0 (y: the ReLU's OUTPUT), else 0; dx may alias dy --------------------
__global__ __launch_bounds__(256) void k_relu_bwd(const float* dy, const float* __restrict__ y,
                                                  float* dx, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dx[i] = y[i] > 0.f ? dy[i] : 0.f;
}

extern "C" int pn_relu_bwd_f32(const float* dy, const float* y, float* dx, int64_t n,
                               void* stream) {
  if (!dy || !y || !dx || n <= 0 || n > ((int64_t)1 << 38)) return PN_BAD_ARG;
  hipLaunchKernelGGL(k_relu_bwd, dim3(pn_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, dy, y,
                     dx, n);
  return PN_LAUNCH_CHECK();
}

// ---- out[i] = a[i] + b[i % bn]: `x + pos` with a row-periodic table (bn = its element count),
// or a gradient accumulation (bn = n, out may alias a)
__global__ __launch_bounds__(256) void k_add_periodic(const float* a, const float* __restrict__ b,
                                                      float* out, int64_t n, int64_t bn) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = a[i] + b[i % bn];
}

extern "C" int pn_add_periodic_f32(const float* a, const float* b, float* out, int64_t n,
                                   int64_t bn, void* stream) {
  if (!a || !b || !out || n <= 0 || bn <= 0 || bn > n || n > ((int64_t)1 << 38)) return PN_BAD_ARG;
  hipLaunchKernelGGL(k_add_periodic, dim3(pn_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, a,
                     b, out, n, bn);
  return PN_LAUNCH_CHECK();
}

// ---- out[i] (+)= sum_b x[b][i], b = 0 .. B-1 in order: the gradient of a table broadcast over
// the batch (query embeddings, positional embeddings)
__global__ __launch_bounds__(256) void k_batch_sum(const float* __restrict__ x, float* out,
                                                   int B, int64_t n, int accumulate) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float t = accumulate ? out[i] : 0.f;
  for (int b = 0; b < B; ++b) t += x[(int64_t)b * n + i];
  out[i] = t;
}

extern "C" int pn_batch_sum_f32(const float* x, float* out, int B, int64_t n, int accumulate,
                                void* stream) {
  if (!x || !out || B <= 0 || n <= 0 || n > ((int64_t)1 << 36)) return PN_BAD_ARG;
  hipLaunchKernelGGL(k_batch_sum, dim3(pn_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, out,
                     B, n, accumulate);
  return PN_LAUNCH_CHECK();
}

// ---- LayerNorm(256) backward, one wave per row.  y = xhat * gamma + beta with
// xhat = (x - mean) rstd (the moments are recomputed from the saved input x):
//   dx = rstd (g - mean(g) - xhat mean(g xhat)),  g = dy gamma;
//   gxhat = dy xhat   (its column sum is d gamma; d beta is the column sum of dy)
__global__ __launch_bounds__(256) void k_ln256_bwd(const float* __restrict__ dy,
                                                   const float* __restrict__ x,
                                                   const float* __restrict__ gamma,
                                                   float* __restrict__ dx,
                                                   float* __restrict__ gxhat, int rows, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const int64_t o = (int64_t)row * 256 + lane * 4;
  const float4 v = ld4(x + o), d = ld4(dy + o), g = ld4(gamma + lane * 4);
  const float mean = wave_sum((v.x + v.y) + (v.z + v.w)) * (1.f / 256.f);
  const float cx = v.x - mean, cy = v.y - mean, cz = v.z - mean, cw = v.w - mean;
  const float var = wave_sum((cx * cx + cy * cy) + (cz * cz + cw * cw)) * (1.f / 256.f);
  const float rstd = 1.f / sqrtf(var + eps);
  const float hx = cx * rstd, hy = cy * rstd, hz = cz * rstd, hw = cw * rstd;
  const float gx = d.x * g.x, gy = d.y * g.y, gz = d.z * g.z, gw = d.w * g.w;
  const float m1 = wave_sum((gx + gy) + (gz + gw)) * (1.f / 256.f);
  const float m2 = wave_sum((gx * hx + gy * hy) + (gz * hz + gw * hw)) * (1.f / 256.f);
  st4(dx + o, make_float4(rstd * (gx - m1 - hx * m2), rstd * (gy - m1 - hy * m2),
                          rstd * (gz - m1 - hz * m2), rstd * (gw - m1 - hw * m2)));
  st4(gxhat + o, make_float4(d.x * hx, d.y * hy, d.z * hz, d.w * hw));
}

extern "C" int pn_layernorm256_bwd_f32(const float* dy, const float* x, const float* gamma,
                                       float* dx, float* gxhat, int rows, float eps,
                                       void* stream) {
  if (!dy || !x || !gamma || !dx || !gxhat || rows <= 0) return PN_BAD_ARG;
  hipLaunchKernelGGL(k_ln256_bwd, dim3(pn_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, dy, x,
                     gamma, dx, gxhat, rows, eps);
  return PN_LAUNCH_CHECK();
}

// ---- multi-head attention backward (8 heads x 32 channels; optional bit-packed boolean mask:
// the Relation Fusion decoder's attentions, Nk <= a few hundred, and the masked decoder's cross-
// attentions over a level's 1 050 - 16 700 keys).  Four passes over a [B][8][Nq][Nk] scratch
// pair: P = softmax(scale q k^T) recomputed from the saved projections, dS = P (dP - sum P dP)
// with dP = dO v^T, then dq = scale dS k, dk = scale dS^T q, dv = P^T dO.  One wave per
// (query | key, head); latency-sized (3 MFLOP per head), so plain FMA rows, not MFMA tiles.
__global__ __launch_bounds__(64) void k_mha_probs(const float* __restrict__ q, int64_t ldq,
                                                  const float* __restrict__ k, int64_t ldk,
                                                  float* __restrict__ P, int Nq, int Nk,
                                                  float scale, const uint32_t* __restrict__ bits,
                                                  const int32_t* __restrict__ rowall) {
  const int i = blockIdx.x, h = blockIdx.y, b = blockIdx.z, lane = threadIdx.x;
  // the boolean attention mask of pn_mask_pack (bit set = key not attendable), shared by the
  // heads; a row whose keys are ALL masked attends to everything (pairnet_head.py:300)
  const int64_t mrow = (int64_t)b * Nq + i;
  const uint32_t* mb = (bits && !rowall[mrow]) ? bits + mrow * ((Nk + 31) >> 5) : nullptr;
  const float* qr = q + ((int64_t)b * Nq + i) * ldq + h * 32;
  float4 qv[8];
#pragma unroll
  for (int d = 0; d < 8; ++d) qv[d] = ld4(qr + 4 * d);
  float* Pr = P + (((int64_t)b * 8 + h) * Nq + i) * Nk;
  float mx = -INFINITY;
  for (int j = lane; j < Nk; j += 64) {
    const float* kr = k + ((int64_t)b * Nk + j) * ldk + h * 32;
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < 8; ++d) {
      const float4 kv = ld4(kr + 4 * d);
      s += qv[d].x * kv.x + qv[d].y * kv.y + qv[d].z * kv.z + qv[d].w * kv.w;
    }
    s *= scale;
    if (mb && ((mb[j >> 5] >> (j & 31)) & 1u)) s = -INFINITY;
    Pr[j] = s;
    mx = fmaxf(mx, s);
  }
  mx = wave_max(mx);
  float l = 0.f;
  for (int j = lane; j < Nk; j += 64) {
    const float e = expf(Pr[j] - mx);
    Pr[j] = e;
    l += e;
  }
  const float inv = 1.f / wave_sum(l);
  for (int j = lane; j < Nk; j += 64) Pr[j] *= inv;
}

__global__ __launch_bounds__(64) void k_mha_bwd_ds(const float* __restrict__ dout, int64_t ldo,
                                                   const float* __restrict__ v, int64_t ldv,
                                                   const float* __restrict__ P,
                                                   float* __restrict__ dS, int Nq, int Nk) {
  const int i = blockIdx.x, h = blockIdx.y, b = blockIdx.z, lane = threadIdx.x;
  const float* dor = dout + ((int64_t)b * Nq + i) * ldo + h * 32;
  float4 dv_[8];
#pragma unroll
  for (int d = 0; d < 8; ++d) dv_[d] = ld4(dor + 4 * d);
  const int64_t ro = (((int64_t)b * 8 + h) * Nq + i) * Nk;
  float part = 0.f;
  for (int j = lane; j < Nk; j += 64) {
    const float* vr = v + ((int64_t)b * Nk + j) * ldv + h * 32;
    float dp = 0.f;
#pragma unroll
    for (int d = 0; d < 8; ++d) {
      const float4 vv = ld4(vr + 4 * d);
      dp += dv_[d].x * vv.x + dv_[d].y * vv.y + dv_[d].z * vv.z + dv_[d].w * vv.w;
    }
    dS[ro + j] = dp;
    part += P[ro + j] * dp;
  }
  const float delta = wave_sum(part);
  for (int j = lane; j < Nk; j += 64) dS[ro + j] = P[ro + j] * (dS[ro + j] - delta);
}

// dq[i][d] = scale sum_j dS[i][j] k[j][d].  A workgroup owns one head and a tile of 8 queries:
// 8 key groups x 32 channels, a key row is loaded once for the 8 queries (the 16 700-key level
// otherwise re-reads its 2 MB of keys per query), group partials summed in group order.
__global__ __launch_bounds__(256) void k_mha_bwd_dq(const float* __restrict__ dS,
                                                    const float* __restrict__ k, int64_t ldk,
                                                    float* __restrict__ dq, int64_t lddq, int Nq,
                                                    int Nk, float scale) {
  __shared__ float red[8][8][32];
  const int i0 = blockIdx.x * 8, h = blockIdx.y, b = blockIdx.z;
  const int d = threadIdx.x & 31, g = threadIdx.x >> 5;
  const float* dsr[8];
#pragma unroll
  for (int q = 0; q < 8; ++q)
    dsr[q] = dS + (((int64_t)b * 8 + h) * Nq + min(i0 + q, Nq - 1)) * Nk;
  float acc[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) acc[q] = 0.f;
  const float* kb = k + (int64_t)b * Nk * ldk + h * 32 + d;
  for (int j = g; j < Nk; j += 8) {
    const float kv = kb[(int64_t)j * ldk];
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] += dsr[q][j] * kv;
  }
#pragma unroll
  for (int q = 0; q < 8; ++q) red[g][q][d] = acc[q];
  __syncthreads();
  const int q = g;
  if (i0 + q < Nq) {
    float t = 0.f;
#pragma unroll
    for (int gg = 0; gg < 8; ++gg) t += red[gg][q][d];
    dq[((int64_t)b * Nq + i0 + q) * lddq + h * 32 + d] = scale * t;
  }
}

__global__ __launch_bounds__(64) void k_mha_bwd_dkv(const float* __restrict__ dS,
                                                    const float* __restrict__ P,
                                                    const float* __restrict__ q, int64_t ldq,
                                                    const float* __restrict__ dout, int64_t ldo,
                                                    float* __restrict__ dk, int64_t lddk,
                                                    float* __restrict__ dv, int64_t lddv, int Nq,
                                                    int Nk, float scale) {
  const int j = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int d = threadIdx.x & 31, half = threadIdx.x >> 5;
  const int64_t base = ((int64_t)b * 8 + h) * Nq * (int64_t)Nk + j;
  float ak = 0.f, av = 0.f;
  for (int i = half; i < Nq; i += 2) {
    const int64_t row = (int64_t)b * Nq + i;
    ak += dS[base + (int64_t)i * Nk] * q[row * ldq + h * 32 + d];
    av += P[base + (int64_t)i * Nk] * dout[row * ldo + h * 32 + d];
  }
  ak += __shfl_xor(ak, 32, 64);
  av += __shfl_xor(av, 32, 64);
  if (half == 0) {
    dk[((int64_t)b * Nk + j) * lddk + h * 32 + d] = scale * ak;
    dv[((int64_t)b * Nk + j) * lddv + h * 32 + d] = av;
  }
}

extern "C" int pn_mha_bwd_f32(const float* q, int64_t ldq, const float* k, int64_t ldk,
                              const float* v, int64_t ldv, const float* dout, int64_t ldo,
                              float* dq, int64_t lddq, float* dk, int64_t lddk, float* dv,
                              int64_t lddv, const uint32_t* bits, const int32_t* rowall,
                              float* scratch, int B, int Nq, int Nk, float scale, void* stream) {
  if (!q || !k || !v || !dout || !dq || !dk || !dv || !scratch || B <= 0 || Nq <= 0 || Nk <= 0 ||
      B > 65535 || (bits && !rowall))
    return PN_BAD_ARG;
  if ((ldq | ldk | ldv | ldo) % 4 || ldq < 256 || ldk < 256 || ldv < 256 || ldo < 256 ||
      lddq < 256 || lddk < 256 || lddv < 256)
    return PN_BAD_ARG;
  if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)dout) & 15) return PN_BAD_ARG;
  hipStream_t s = (hipStream_t)stream;
  float* P = scratch;
  float* dS = scratch + (int64_t)B * 8 * Nq * Nk;
  hipLaunchKernelGGL(k_mha_probs, dim3(Nq, 8, B), dim3(64), 0, s, q, ldq, k, ldk, P, Nq, Nk, scale,
                     bits, rowall);
  hipLaunchKernelGGL(k_mha_bwd_ds, dim3(Nq, 8, B), dim3(64), 0, s, dout, ldo, v, ldv, P, dS, Nq, Nk);
  hipLaunchKernelGGL(k_mha_bwd_dq, dim3(pn_cdiv(Nq, 8), 8, B), dim3(256), 0, s, dS, k, ldk, dq, lddq,
                     Nq, Nk, scale);
  hipLaunchKernelGGL(k_mha_bwd_dkv, dim3(Nk, 8, B), dim3(64), 0, s, dS, P, q, ldq, dout, ldo, dk,
                     lddk, dv, lddv, Nq, Nk, scale);
  return PN_LAUNCH_CHECK();
}

// ---- the backward of pn_gather_rows_f32: out[b][row][:] (+)= sum over the slots s with
// index[b][s] == row of src[b][s][:], slots in order (a query picked as subject AND object, or
// by several pairs, collects every contribution; no atomics)
__global__ __launch_bounds__(256) void k_scatter_rows_add(const float* __restrict__ src,
                                                          int64_t ld_src,
                                                          const int64_t* __restrict__ index,
                                                          float* out, int64_t ld_out, int rows_out,
                                                          int slots, int length, int accumulate) {
  const int row = blockIdx.x, b = blockIdx.y;
  const int64_t* ix = index + (int64_t)b * slots;
  for (int c = threadIdx.x; c < length; c += 256) {
    float* o = out + ((int64_t)b * rows_out + row) * ld_out + c;
    float t = accumulate ? *o : 0.f;
    for (int s = 0; s < slots; ++s)
      if (ix[s] == row) t += src[((int64_t)b * slots + s) * ld_src + c];
    *o = t;
  }
}

extern "C" int pn_scatter_rows_add_f32(const float* src, int64_t ld_src, const int64_t* index,
                                       float* out, int64_t ld_out, int B, int rows_out, int slots,
                                       int length, int accumulate, void* stream) {
  if (!src || !index || !out || B <= 0 || B > 65535 || rows_out <= 0 || slots <= 0 || length <= 0 ||
      ld_src < length || ld_out < length)
    return PN_BAD_ARG;
  hipLaunchKernelGGL(k_scatter_rows_add, dim3(rows_out, B), dim3(256), 0, (hipStream_t)stream, src,
                     ld_src, index, out, ld_out, rows_out, slots, length, accumulate);
  return PN_LAUNCH_CHECK();
}

// ---- cosine block backward (pairnet_head.py:325-333): raw[b][i][j] = shat_i . ohat_j with
// shat = s / max(|s|, eps).  For side 0 (subjects): d shat_i = sum_j draw[i][j] ohat_j; for side 1
// (objects) d ohat_j = sum_i draw[i][j] shat_i; then through F.normalize:
// d s = (d shat - shat (shat . d shat)) / |s|   (|s| > eps).
// x: this side's UN-normalised rows, other_hat: the other side's normalised rows; one workgroup
// per row, thread = channel (256).
__global__ __launch_bounds__(256) void k_cos_bwd(const float* __restrict__ draw,
                                                 const float* __restrict__ x,
                                                 const float* __restrict__ other_hat,
                                                 float* __restrict__ dx, int Q, int transposed,
                                                 float eps) {
  __shared__ float red[4];
  const int i = blockIdx.x, b = blockIdx.y, d = threadIdx.x;
  const float* dr = draw + (int64_t)b * Q * Q;
  const float* oh = other_hat + (int64_t)b * Q * 256;
  float g = 0.f;
  for (int j = 0; j < Q; ++j)
    g += (transposed ? dr[(int64_t)j * Q + i] : dr[(int64_t)i * Q + j]) * oh[(int64_t)j * 256 + d];
  const float xv = x[((int64_t)b * Q + i) * 256 + d];
  auto bsum = [&](float v) {
    v = wave_sum(v);
    __syncthreads();
    if ((d & 63) == 0) red[d >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
  };
  const float nrm = sqrtf(bsum(xv * xv));
  const float den = fmaxf(nrm, eps);
  const float hat = xv / den;
  const float dot = bsum(hat * g);
  // (below eps the forward divides by the constant eps: d s = d shat / eps)
  dx[((int64_t)b * Q + i) * 256 + d] = nrm > eps ? (g - hat * dot) / den : g / den;
}

extern "C" int pn_cosine_bwd_f32(const float* draw, const float* x, const float* other_hat,
                                 float* dx, int B, int Q, int transposed, float eps,
                                 void* stream) {
  if (!draw || !x || !other_hat || !dx || B <= 0 || B > 65535 || Q <= 0) return PN_BAD_ARG;
  hipLaunchKernelGGL(k_cos_bwd, dim3(Q, B), dim3(256), 0, (hipStream_t)stream, draw, x, other_hat,
                     dx, Q, transposed, eps);
  return PN_LAUNCH_CHECK();
}

// ---- Matrix Learner, last layer (64 -> 1, 7x7, cnn_factory.py:42-48) data gradient:
// dc[b][y][x][ci] = [c[b][y][x][ci] > 0] sum_taps g[b][y + 3 - kh][x + 3 - kw] w3[kh*7+kw][ci]
// (c: the ReLU output that fed the layer; w3 [49][64] as pn_mlearner_last_f32 takes it)
__global__ __launch_bounds__(256) void k_ml_last_bwd_data(const float* __restrict__ g,
                                                          const float* __restrict__ w3,
                                                          const float* __restrict__ c,
                                                          float* __restrict__ dc, int S,
                                                          int64_t npix) {
  __shared__ float w[49 * 64];
  for (int e = threadIdx.x; e < 49 * 64; e += 256) w[e] = w3[e];
  __syncthreads();
  const int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int ci = threadIdx.x & 63;
  if (p >= npix) return;
  const int x = (int)(p % S), y = (int)((p / S) % S);
  const float* gb = g + (p - (int64_t)y * S - x);
  float acc = 0.f;
  for (int kh = 0; kh < 7; ++kh) {
    const int yy = y + 3 - kh;
    if (yy < 0 || yy >= S) continue;
    for (int kw = 0; kw < 7; ++kw) {
      const int xx = x + 3 - kw;
      if (xx < 0 || xx >= S) continue;
      acc += gb[yy * S + xx] * w[(kh * 7 + kw) * 64 + ci];
    }
  }
  dc[p * 64 + ci] = c[p * 64 + ci] > 0.f ? acc : 0.f;
}

extern "C" int pn_mlearner_last_bwd_data_f32(const float* g, const float* w3, const float* c,
                                             float* dc, int B, int S, void* stream) {
  if (!g || !w3 || !c || !dc || B <= 0 || S <= 0) return PN_BAD_ARG;
  const int64_t npix = (int64_t)B * S * S;
  hipLaunchKernelGGL(k_ml_last_bwd_data, dim3(pn_cdiv(npix, 4)), dim3(256), 0, (hipStream_t)stream,
                     g, w3, c, dc, S, npix);
  return PN_LAUNCH_CHECK();
}

// ---- tap correlation of a 64-channel map F with a 1-channel map g (the weight gradients of
// the Matrix Learner's first and last layers):
//   part[b * S + y][tap][c] = sum_x F[b][y][x][c] g[b][y + sgn (kh - 3)][x + sgn (kw - 3)]
// (zero outside the map); the caller column-sums `part` over its B * S rows.
//   last layer:  d w3[tap][ci] : F = c2 (the layer's input), g = d importance, sgn = -1
//   first layer: d w1[co][tap] : F = d c1 (pre-ReLU gradient), g = importance_raw, sgn = +1
__global__ __launch_bounds__(256) void k_tapcorr1(const float* __restrict__ F,
                                                  const float* __restrict__ g,
                                                  float* __restrict__ part, int S, int sgn) {
  __shared__ float red[4][64];
  const int tap = blockIdx.x, by = blockIdx.y;          // by = b * S + y
  const int y = by % S, b = by / S;
  const int kh = tap / 7, kw = tap % 7;
  const int c = threadIdx.x & 63, xl = threadIdx.x >> 6;
  const int yy = y + sgn * (kh - 3), dxs = sgn * (kw - 3);
  float acc = 0.f;
  if (yy >= 0 && yy < S) {
    const float* Fr = F + ((int64_t)by * S) * 64 + c;
    const float* gr = g + ((int64_t)b * S + yy) * S;
    for (int x = xl; x < S; x += 4) {
      const int xx = x + dxs;
      if (xx >= 0 && xx < S) acc += Fr[(int64_t)x * 64] * gr[xx];
    }
  }
  red[xl][c] = acc;
  __syncthreads();
  if (xl == 0)
    part[((int64_t)by * 49 + tap) * 64 + c] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
}

extern "C" int pn_tapcorr1_f32(const float* F, const float* g, float* part, int B, int S, int sgn,
                               void* stream) {
  if (!F || !g || !part || B <= 0 || S <= 0 || (int64_t)B * S > 65535 || (sgn != 1 && sgn != -1))
    return PN_BAD_ARG;
  hipLaunchKernelGGL(k_tapcorr1, dim3(49, B * S), dim3(256), 0, (hipStream_t)stream, F, g, part, S,
                     sgn);
  return PN_LAUNCH_CHECK();
}

// ---- a convolution weight in the layout its DATA gradient needs (a "same" convolution's data
// gradient is a convolution with the taps reversed and the channel roles swapped):
//   out[ci][T - 1 - t][co] = in[co][t][ci]      (in [Co][T][Ci] -> out [Ci][T][Co])
__global__ __launch_bounds__(256) void k_conv_w_bwd_layout(const float* __restrict__ in,
                                                           float* __restrict__ out, int Co, int T,
                                                           int Ci) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)Co * T * Ci) return;
  const int ci = (int)(i % Ci), t = (int)((i / Ci) % T), co = (int)(i / ((int64_t)Ci * T));
  out[((int64_t)ci * T + (T - 1 - t)) * Co + co] = in[i];
}

extern "C" int pn_conv_weight_bwd_layout_f32(const float* in, float* out, int Co, int T, int Ci,
                                             void* stream) {
  if (!in || !out || Co <= 0 || T <= 0 || Ci <= 0) return PN_BAD_ARG;
  hipLaunchKernelGGL(k_conv_w_bwd_layout, dim3(pn_cdiv((int64_t)Co * T * Ci, 256)), dim3(256), 0,
                     (hipStream_t)stream, in, out, Co, T, Ci);
  return PN_LAUNCH_CHECK();
}

// ---- pixel decoder: from the sampling operator's gradients back to the projection that made
// its operands (mmcv MultiScaleDeformableAttention.forward: sampling_locations = reference_points
// + sampling_offsets / (W_l, H_l), attention_weights = softmax over each head's L * 4 logits):
//   d offset[h][l][p][xy] = d loc[h][l][p][xy] / (W_l | H_l)
//   d logit[h][l][p]      = aw (d aw - sum_{l', p'} aw d aw)
// offaw rows [offsets 8*L*4*2 | logits 8*L*4] at stride ld, as pn_token_sampling_f32 reads them.
struct LevelDims { int h[4], w[4]; };
__global__ __launch_bounds__(256) void k_msda_offaw_bwd(const float* __restrict__ grad_loc,
                                                        const float* __restrict__ grad_aw,
                                                        const float* __restrict__ aw,
                                                        float* __restrict__ d_offaw, int64_t ld,
                                                        int64_t rows, int L, LevelDims dims) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;      // one thread per (row, head)
  if (t >= rows * 8) return;
  const int64_t r = t >> 3;
  const int h = (int)(t & 7), LP = L * 4;
  const float* gl = grad_loc + (r * 8 + h) * LP * 2;
  const float* ga = grad_aw + (r * 8 + h) * LP;
  const float* a = aw + (r * 8 + h) * LP;
  float* doff = d_offaw + r * ld + h * LP * 2;
  float* dlg = d_offaw + r * ld + 8 * LP * 2 + h * LP;
  float dot = 0.f;
  for (int i = 0; i < LP; ++i) dot += a[i] * ga[i];
  for (int i = 0; i < LP; ++i) {
    const int l = i >> 2;
    doff[2 * i] = gl[2 * i] / (float)dims.w[l];
    doff[2 * i + 1] = gl[2 * i + 1] / (float)dims.h[l];
    dlg[i] = a[i] * (ga[i] - dot);
  }
}

extern "C" int pn_msda_offaw_bwd_f32(const float* grad_loc, const float* grad_aw, const float* aw,
                                     float* d_offaw, int64_t ld, int64_t rows, int L,
                                     const int32_t* level_h, const int32_t* level_w, void* stream) {
  if (!grad_loc || !grad_aw || !aw || !d_offaw || rows <= 0 || L <= 0 || L > 4 || !level_h ||
      !level_w || ld < 8 * L * 12)
    return PN_BAD_ARG;
  LevelDims d;
  for (int l = 0; l < 4; ++l) {
    d.h[l] = l < L ? level_h[l] : 1;
    d.w[l] = l < L ? level_w[l] : 1;
    if (d.h[l] <= 0 || d.w[l] <= 0) return PN_BAD_ARG;
  }
  hipLaunchKernelGGL(k_msda_offaw_bwd, dim3(pn_cdiv(rows * 8, 256)), dim3(256), 0,
                     (hipStream_t)stream, grad_loc, grad_aw, aw, d_offaw, ld, rows, L, d);
  return PN_LAUNCH_CHECK();
}

// ---- GroupNorm backward over channel-last x[b][HW][256], G groups (the pixel decoder's
// ConvModule norm).  Pass 1, one workgroup per (image, group): the group's moments and the two
// means of the backward formula, in double, fixed order -> stats[b][g] = (mean, rstd, m1, m2)
// with g_c = gamma_c dy, m1 = mean(g), m2 = mean(g xhat).  Pass 2, elementwise:
//   dx = rstd (g - m1 - xhat m2);  gxhat = dy xhat  (column sums: d gamma; d beta = colsum dy)
__global__ __launch_bounds__(1024) void k_gn_bwd_stats(const float* __restrict__ x,
                                                       const float* __restrict__ dy,
                                                       const float* __restrict__ gamma,
                                                       float* __restrict__ stats, int64_t HW,
                                                       int G, int64_t x_bstride,
                                                       int64_t dy_bstride, float eps) {
  __shared__ double red[16][2];
  const int g = blockIdx.x, b = blockIdx.y, cpg = 256 / G;
  const float* xb = x + (int64_t)b * x_bstride + g * cpg;
  const float* db = dy + (int64_t)b * dy_bstride + g * cpg;
  const int64_t n = HW * cpg;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  auto reduce2 = [&](double a, double c, double* oa, double* oc) {
    a = wave_sum_d(a); c = wave_sum_d(c);
    __syncthreads();
    if (lane == 0) { red[wave][0] = a; red[wave][1] = c; }
    __syncthreads();
    double ta = 0.0, tc = 0.0;
    for (int w = 0; w < 16; ++w) { ta += red[w][0]; tc += red[w][1]; }
    *oa = ta; *oc = tc;
  };
  double s = 0.0, ss = 0.0;
  for (int64_t e = threadIdx.x; e < n; e += 1024) {
    const double v = (double)xb[(e / cpg) * 256 + (e % cpg)];
    s += v; ss += v * v;
  }
  double S, SS;
  reduce2(s, ss, &S, &SS);
  const double mean = S / (double)n;
  const double var = fmax(SS / (double)n - mean * mean, 0.0);
  const double rstd = 1.0 / sqrt(var + (double)eps);
  double a1 = 0.0, a2 = 0.0;
  for (int64_t e = threadIdx.x; e < n; e += 1024) {
    const int c = (int)(e % cpg);
    const int64_t o = (e / cpg) * 256 + c;
    const double gd = (double)gamma[g * cpg + c] * (double)db[o];
    a1 += gd;
    a2 += gd * ((double)xb[o] - mean) * rstd;
  }
  double A1, A2;
  reduce2(a1, a2, &A1, &A2);
  if (threadIdx.x == 0) {
    float* st = stats + ((int64_t)b * G + g) * 4;
    st[0] = (float)mean; st[1] = (float)rstd;
    st[2] = (float)(A1 / (double)n); st[3] = (float)(A2 / (double)n);
  }
}

__global__ __launch_bounds__(256) void k_gn_bwd_apply(const float* __restrict__ x,
                                                      const float* __restrict__ dy,
                                                      const float* __restrict__ gamma,
                                                      const float* __restrict__ stats,
                                                      float* __restrict__ dx,
                                                      float* __restrict__ gxhat, int64_t HW, int G,
                                                      int64_t x_bstride, int64_t dy_bstride) {
  const int64_t p = blockIdx.x;           // pixel
  const int b = blockIdx.y, c = threadIdx.x, cpg = 256 / G;
  const float* st = stats + ((int64_t)b * G + c / cpg) * 4;
  const float xv = x[(int64_t)b * x_bstride + p * 256 + c];
  const float d = dy[(int64_t)b * dy_bstride + p * 256 + c];
  const float xh = (xv - st[0]) * st[1];
  const int64_t o = ((int64_t)b * HW + p) * 256 + c;
  dx[o] = st[1] * (gamma[c] * d - st[2] - xh * st[3]);
  gxhat[o] = d * xh;
}

extern "C" int pn_groupnorm_nhwc_bwd_f32(const float* x, const float* dy, const float* gamma,
                                         float* dx, float* gxhat, float* stats, int B, int64_t HW,
                                         int G, float eps, int64_t x_bstride, int64_t dy_bstride,
                                         void* stream) {
  if (!x || !dy || !gamma || !dx || !gxhat || !stats || B <= 0 || B > 65535 || HW <= 0 ||
      HW > 2147483647 || G <= 0 || 256 % G)
    return PN_BAD_ARG;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_gn_bwd_stats, dim3(G, B), dim3(1024), 0, s, x, dy, gamma, stats, HW, G,
                     x_bstride, dy_bstride, eps);
  hipLaunchKernelGGL(k_gn_bwd_apply, dim3((unsigned)HW, B), dim3(256), 0, s, x, dy, gamma, stats,
                     dx, gxhat, HW, G, x_bstride, dy_bstride);
  return PN_LAUNCH_CHECK();
}

// ---- backbone (mmdet ResNet, frozen BatchNorm folded into the convolutions): the pieces its
// backward needs beside the GEMMs -- and the Matrix Learner's 64 -> 64 7x7 layer.
// Weight gradient of a K x K convolution (stride s, padding p) between channel-last maps, as
// MFMA outer products over output pixels:
//   dW[co][tap][ci] = sum_{b,y,x} dY[b][y][x][co] X[b][y s + kh - p][x s + kw - p][ci]
// A workgroup owns one tap, one 64 x 64 (co, ci) block and one chunk of `rows_per` output rows;
// its four waves own the four 32 x 32 quadrants; a k-step is two neighbouring output pixels, and
// both operands are read straight from the channel-last maps (a lane's operand element is
// [pixel = lane / 32][channel = lane % 32]: two 128-byte lines per wave per map, no LDS, no
// transposes).  part[chunk][co][tap][ci], column-summed over the chunks by the caller (fixed order).
__global__ __launch_bounds__(256) void k_conv_wgrad(const float* __restrict__ dY,
                                                    const float* __restrict__ X,
                                                    float* __restrict__ part, int Hi, int Wi,
                                                    int Ho, int Wo, int Ci, int Co, int K,
                                                    int stride, int pad, int rows_per,
                                                    int chunks_per_image) {
  const int T = K * K, nci = Ci >> 6;
  int id = blockIdx.x;
  const int cib = id % nci; id /= nci;
  const int tap = id % T;
  const int cob = id / T;
  const int chunk = blockIdx.y;
  const int b = chunk / chunks_per_image, y0 = (chunk % chunks_per_image) * rows_per;
  const int kh = tap / K, kw = tap % K;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int co0 = cob * 64 + (wave & 1) * 32, ci0 = cib * 64 + (wave >> 1) * 32;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int y1 = min(y0 + rows_per, Ho);
  for (int y = y0; y < y1; ++y) {
    const int yy = y * stride + kh - pad;
    if (yy < 0 || yy >= Hi) continue;                      // (workgroup-uniform)
    const float* dr = dY + (((int64_t)b * Ho + y) * Wo) * Co + co0 + li;
    const float* xr = X + (((int64_t)b * Hi + yy) * Wi) * Ci + ci0 + li;
    for (int x = 0; x < Wo; x += 2) {
      const int xp = x + lh, xx = xp * stride + kw - pad;
      const float a = xp < Wo ? dr[(int64_t)xp * Co] : 0.f;
      const float w = (xp < Wo && xx >= 0 && xx < Wi) ? xr[(int64_t)xx * Ci] : 0.f;
      acc = mfma32(a, w, acc);
    }
  }
  float* out = part + (int64_t)chunk * Co * T * Ci;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int co = co0 + mfma32_row(r, lh), ci = ci0 + li;
    out[((int64_t)co * T + tap) * Ci + ci] = acc[r];
  }
}

extern "C" int pn_conv_wgrad_f32(const float* dY, const float* X, float* part, int B, int Hi,
                                 int Wi, int Ho, int Wo, int Ci, int Co, int K, int stride, int pad,
                                 int rows_per, void* stream) {
  if (!dY || !X || !part || B <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0 || Ci <= 0 ||
      Co <= 0 || (Ci & 63) || (Co & 63) || K <= 0 || K > 7 || stride <= 0 || pad < 0 || rows_per <= 0)
    return PN_BAD_ARG;
  const int cpi = pn_cdiv(Ho, rows_per);
  const int64_t gx = (int64_t)K * K * (Co >> 6) * (Ci >> 6);
  if ((int64_t)B * cpi > 65535 || gx > 2147483647) return PN_BAD_ARG;
  hipLaunchKernelGGL(k_conv_wgrad, dim3((unsigned)gx, B * cpi), dim3(256), 0, (hipStream_t)stream,
                     dY, X, part, Hi, Wi, Ho, Wo, Ci, Co, K, stride, pad, rows_per, cpi);
  return PN_LAUNCH_CHECK();
}

// out[b][y][x][:] (+)= (y, x both even and inside) ? in[b][y/2][x/2][:] : 0 -- the zero-dilated
// gradient map of a stride-2 layer ([B][Ho][Wo][C] -> [B][Hi][Wi][C]): its data gradient is then
// the stride-1 convolution with the reversed taps; with `accumulate` the backward of the
// stride-2 subsampling in front of a 1x1 projection shortcut.
__global__ __launch_bounds__(256) void k_dilate2(const float* __restrict__ in, float* out, int Hi,
                                                 int Wi, int Ho, int Wo, int C, int64_t n,
                                                 int accumulate) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int c = (int)(i % C);
  const int64_t p = i / C;
  const int x = (int)(p % Wi), y = (int)((p / Wi) % Hi);
  const int64_t b = p / ((int64_t)Wi * Hi);
  float v = 0.f;
  if (!(y & 1) && !(x & 1) && (y >> 1) < Ho && (x >> 1) < Wo)
    v = in[((b * Ho + (y >> 1)) * Wo + (x >> 1)) * C + c];
  out[i] = accumulate ? out[i] + v : v;
}

extern "C" int pn_dilate2_f32(const float* in, float* out, int B, int Hi, int Wi, int Ho, int Wo,
                              int C, int accumulate, void* stream) {
  if (!in || !out || B <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0 || C <= 0 ||
      2 * (Ho - 1) > Hi - 1 || 2 * (Wo - 1) > Wi - 1)
    return PN_BAD_ARG;
  const int64_t n = (int64_t)B * Hi * Wi * C;
  hipLaunchKernelGGL(k_dilate2, dim3(pn_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, in, out,
                     Hi, Wi, Ho, Wo, C, n, accumulate);
  return PN_LAUNCH_CHECK();
}

// out[b][i][j][:] = in[b][2 i][2 j][:]  (what a stride-2 1x1 convolution reads)
__global__ __launch_bounds__(256) void k_subsample2(const float* __restrict__ in,
                                                    float* __restrict__ out, int Hi, int Wi,
                                                    int Ho, int Wo, int C, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int c = (int)(i % C);
  const int64_t p = i / C;
  const int x = (int)(p % Wo), y = (int)((p / Wo) % Ho);
  const int64_t b = p / ((int64_t)Wo * Ho);
  out[i] = in[((b * Hi + 2 * y) * Wi + 2 * x) * C + c];
}

extern "C" int pn_subsample2_f32(const float* in, float* out, int B, int Hi, int Wi, int Ho, int Wo,
                                 int C, void* stream) {
  if (!in || !out || B <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0 || C <= 0 ||
      2 * (Ho - 1) > Hi - 1 || 2 * (Wo - 1) > Wi - 1)
    return PN_BAD_ARG;
  const int64_t n = (int64_t)B * Ho * Wo * C;
  hipLaunchKernelGGL(k_subsample2, dim3(pn_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, in, out,
                     Hi, Wi, Ho, Wo, C, n);
  return PN_LAUNCH_CHECK();
}

// x[r][:] *= s[r]  (a folded convolution's weight gradient back to the un-folded weight:
// W' = W * gamma / sqrt(var + eps) per output channel)
__global__ __launch_bounds__(256) void k_scale_rows(float* x, const float* __restrict__ s,
                                                    int64_t cols, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) x[i] *= s[i / cols];
}

extern "C" int pn_scale_rows_f32(float* x, const float* s, int64_t rows, int64_t cols,
                                 void* stream) {
  if (!x || !s || rows <= 0 || cols <= 0) return PN_BAD_ARG;
  hipLaunchKernelGGL(k_scale_rows, dim3(pn_cdiv(rows * cols, 256)), dim3(256), 0,
                     (hipStream_t)stream, x, s, cols, rows * cols);
  return PN_LAUNCH_CHECK();
}
