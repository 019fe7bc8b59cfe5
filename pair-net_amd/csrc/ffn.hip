// Fused feed-forward block for the M ~ 100 query rows of the decoders
// (FFN of BaseTransformerLayer: x + W2 relu(W1 x + b1) + b2, then LayerNorm).
//
// As two plain GEMMs the second one (K = 2048, 100 x 256 outputs) has only 32 output
// tiles, so 32 CUs stream the whole 2 MB of W2 (23 us measured).  Here the hidden
// dimension is the parallel axis instead:
//   k_ffn_partial   workgroup (c, mt): h = relu(x[32 rows] . W1[64c..64c+64]^T + b1) is
//                   built in LDS, then contracted with the matching 64-column slab of
//                   W2 into a 32 x 256 partial product -> partial[c][row][256].
//                   hidden/64 x M/32 workgroups, every W1/W2 element is read by M/32
//                   workgroups only.
//   k_reduce_ln     one wave per row: sum of the partials (fixed order) + b2 +
//                   residual, then LayerNorm -- replaces FFN2's epilogue AND the norm
//                   launch that followed it.
#include "common.h"

#define FFN_HC 64  // hidden columns per workgroup

__global__ __launch_bounds__(512) void k_ffn_partial(const float* __restrict__ x,
                                                     const float* __restrict__ W1,
                                                     const float* __restrict__ b1,
                                                     const float* __restrict__ W2,
                                                     float* __restrict__ partial, int M,
                                                     int hidden) {
  // red: 8 partial 32x32 tiles of phase 1; hs: the relu'd 32 x 64 hidden tile
  __shared__ __attribute__((aligned(16))) float red[8 * 1024];
  __shared__ __attribute__((aligned(16))) float hs[32 * (FFN_HC + 4)];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int c = blockIdx.x, m0 = blockIdx.y * 32;
  const int am = min(m0 + li, M - 1);
  // phase 2's W2 fragment does not depend on phase 1: its loads are issued first and land
  // under phase 1's loads / MFMAs / reduction
  float4 w2f[8];
  {
    const float* wrow = W2 + (int64_t)(wave * 32 + li) * hidden + c * FFN_HC;
#pragma unroll
    for (int s = 0; s < 8; ++s) w2f[s] = ld4(wrow + 8 * s + 4 * lh);
  }

  // ---- phase 1: wave = (column tile ct, K quarter kq); K = 256 ----
  {
    const int ct = wave & 1, kq = wave >> 1;
    const float* arow = x + (int64_t)am * 256 + kq * 64;
    const float* wrow = W1 + (int64_t)(c * FFN_HC + ct * 32 + li) * 256 + kq * 64;
    float4 a[8], b[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      a[s] = ld4(arow + 8 * s + 4 * lh);
      b[s] = ld4(wrow + 8 * s + 4 * lh);
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      acc = mfma32(a[s].x, b[s].x, acc);
      acc = mfma32(a[s].y, b[s].y, acc);
      acc = mfma32(a[s].z, b[s].z, acc);
      acc = mfma32(a[s].w, b[s].w, acc);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave * 1024 + mfma32_row(r, lh) * 32 + li] = acc[r];
  }
  __syncthreads();
  for (int e = tid; e < 2048; e += 512) {   // 2 column tiles x 1024 elements
    const int ct = e >> 10, r = e & 1023;
    const int row = r >> 5, col = ct * 32 + (r & 31);
    float v = ((red[(ct + 0) * 1024 + r] + red[(ct + 2) * 1024 + r]) +
               red[(ct + 4) * 1024 + r]) + red[(ct + 6) * 1024 + r];
    v += b1[c * FFN_HC + col];
    hs[row * (FFN_HC + 4) + col] = fmaxf(v, 0.f);
  }
  __syncthreads();

  // ---- phase 2: wave w owns output columns [32w, 32w+32); K = 64 hidden columns ----
  {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const float4 b = w2f[s];
      const float4 a = ld4(hs + li * (FFN_HC + 4) + 8 * s + 4 * lh);
      acc = mfma32(a.x, b.x, acc);
      acc = mfma32(a.y, b.y, acc);
      acc = mfma32(a.z, b.z, acc);
      acc = mfma32(a.w, b.w, acc);
    }
    float* out = partial + ((int64_t)c * M) * 256 + wave * 32 + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + mfma32_row(r, lh);
      if (row < M) out[(int64_t)row * 256] = acc[r];
    }
  }
}

// y[row] = LayerNorm(sum_s partial[s][row] + bias + res[row]) * gamma + beta, C = 256
__global__ __launch_bounds__(256) void k_reduce_ln(const float* __restrict__ partial, int S,
                                                   const float* __restrict__ bias,
                                                   const float* __restrict__ res,
                                                   const float* __restrict__ g,
                                                   const float* __restrict__ b,
                                                   float* __restrict__ y, int64_t rows,
                                                   float eps, const float* __restrict__ g2,
                                                   const float* __restrict__ b2n,
                                                   float* __restrict__ y2) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  // the S partials are added in slice order; their loads are issued eight at a time (the same
  // sum as a one-by-one loop, without 31 dependent round trips to L2)
  const float* pp = partial + row * 256 + lane * 4;
  const int64_t ps = rows * 256;
  float4 v = ld4(pp);
  int s = 1;
  for (; s + 8 <= S; s += 8) {
    float4 t[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = ld4(pp + (s + j) * ps);
#pragma unroll
    for (int j = 0; j < 8; ++j) v = add4(v, t[j]);
  }
  for (; s < S; ++s) v = add4(v, ld4(pp + s * ps));
  if (bias) v = add4(v, ld4(bias + lane * 4));
  if (res) v = add4(ld4(res + row * 256 + lane * 4), v);
  const float4 gg = ld4(g + lane * 4), bb = ld4(b + lane * 4);
  const float4 o = ln256_row(v, gg, bb, eps);
  st4(y + row * 256 + lane * 4, o);
  if (y2) {   // a second LayerNorm of the result (the decoder's post_norm): k_layernorm256's
              // arithmetic (the shared ln256_row) on the row this wave already holds
    const float4 g4 = ld4(g2 + lane * 4), b4 = ld4(b2n + lane * 4);
    st4(y2 + row * 256 + lane * 4, ln256_row(o, g4, b4, eps));
  }
}

extern "C" int64_t pn_ffn_scratch_floats(int M, int hidden) {
  return (int64_t)(hidden / FFN_HC) * M * 256;
}

extern "C" int pn_ffn_ln2_f32(const float* x, const float* W1, const float* b1, const float* W2,
                              const float* b2, const float* gamma, const float* beta, float* y,
                              const float* gamma2, const float* beta2, float* y2,
                              float* scratch, int M, int C, int hidden, float eps,
                              void* stream) {
  if (!x || !W1 || !b1 || !W2 || !b2 || !gamma || !beta || !y || !scratch) return PN_BAD_ARG;
  if (C != 256 || M <= 0 || hidden <= 0 || hidden % FFN_HC) return PN_BAD_ARG;
  if (((uintptr_t)x | (uintptr_t)W1 | (uintptr_t)W2 | (uintptr_t)scratch | (uintptr_t)y) & 15)
    return PN_BAD_ARG;
  if (y2 && (!gamma2 || !beta2 || ((uintptr_t)y2 & 15))) return PN_BAD_ARG;
  hipStream_t s = (hipStream_t)stream;
  const int S = hidden / FFN_HC;
  hipLaunchKernelGGL(k_ffn_partial, dim3(S, pn_cdiv(M, 32)), dim3(512), 0, s, x, W1, b1, W2,
                     scratch, M, hidden);
  hipLaunchKernelGGL(k_reduce_ln, dim3(pn_cdiv(M, 4)), dim3(256), 0, s, scratch, S, b2, x, gamma,
                     beta, y, (int64_t)M, eps, gamma2, beta2, y2);
  return PN_LAUNCH_CHECK();
}

extern "C" int pn_ffn_ln_f32(const float* x, const float* W1, const float* b1, const float* W2,
                             const float* b2, const float* gamma, const float* beta, float* y,
                             float* scratch, int M, int C, int hidden, float eps,
                             void* stream) {
  return pn_ffn_ln2_f32(x, W1, b1, W2, b2, gamma, beta, y, nullptr, nullptr, nullptr, scratch, M,
                        C, hidden, eps, stream);
}
