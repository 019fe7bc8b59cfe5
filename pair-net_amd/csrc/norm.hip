// Row / group normalisations (HBM-bound; one pass over the data each, fp32 I/O).
#include "common.h"

// One wave per 256-channel row: lane holds one float4.  Two-pass moments in
// registers (mean, then centred variance), like torch's CPU LayerNorm.
__global__ __launch_bounds__(256) void k_layernorm256(const float* __restrict__ x,
                                                       const float* __restrict__ g,
                                                       const float* __restrict__ b,
                                                       float* __restrict__ y,
                                                       int64_t rows, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float4 v = ld4(x + row * 256 + lane * 4);
  const float4 gg = ld4(g + lane * 4), bb = ld4(b + lane * 4);
  st4(y + row * 256 + lane * 4, ln256_row(v, gg, bb, eps));
}

extern "C" int pn_layernorm_f32(const float* x, const float* gamma, const float* beta,
                                float* y, int64_t rows, int C, float eps, void* stream) {
  if (!x || !gamma || !beta || !y || C != 256 || rows <= 0) return PN_BAD_ARG;
  hipLaunchKernelGGL(k_layernorm256, dim3(pn_cdiv(rows, 4)), dim3(256), 0,
                     (hipStream_t)stream, x, gamma, beta, y, rows, eps);
  return PN_LAUNCH_CHECK();
}

// ---- GroupNorm, channel-last, C == 256 (64 float4 lanes per pixel) ----
// Pass 1: each block reduces GN_PIX pixels x 256 channels to per-group (sum, sumsq)
// in fp64 (exact enough that E[x^2]-E[x]^2 is safe) and writes them to `partials`.
// Pass 2: every block re-reduces the (small) partials of its image in a fixed order
// -> mean / rstd per group, then normalises its own pixels.  Deterministic.
// 16 waves per block: a 66 800-pixel map is only 261 blocks (about one per CU), so the
// block itself has to supply the memory-level parallelism.
#define GN_PIX 256
#define GN_SUB 16   // waves per block = pixel sub-lanes

extern "C" int pn_groupnorm_nblk(int64_t HW) { return pn_cdiv(HW, GN_PIX); }

__global__ __launch_bounds__(64 * GN_SUB) void k_gn_partial(const float* __restrict__ x,
                                                            double* __restrict__ partials,
                                                            int64_t HW, int G, int64_t xbs) {
  __shared__ double red[GN_SUB][64][2];
  const int tid = threadIdx.x, c4 = tid & 63, sub = tid >> 6;
  const int b = blockIdx.y;
  const int64_t p0 = (int64_t)blockIdx.x * GN_PIX;
  const float* xb = x + (int64_t)b * xbs;
  // all loads first, unconditional (clamped; hipcc serialises predicated loads)
  constexpr int NP = GN_PIX / GN_SUB;
  float4 v[NP];
#pragma unroll
  for (int j = 0; j < NP; ++j)
    v[j] = ld4(xb + min(p0 + sub + j * GN_SUB, HW - 1) * 256 + c4 * 4);
  double s = 0.0, ss = 0.0;
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    if (p0 + sub + j * GN_SUB < HW) {
      s += ((double)v[j].x + (double)v[j].y) + ((double)v[j].z + (double)v[j].w);
      ss += ((double)v[j].x * v[j].x + (double)v[j].y * v[j].y) +
            ((double)v[j].z * v[j].z + (double)v[j].w * v[j].w);
    }
  }
  red[sub][c4][0] = s;
  red[sub][c4][1] = ss;
  __syncthreads();
  // group g owns channels [g*cpg, (g+1)*cpg) = float4 lanes [g*cpg/4, ...)
  const int lanes_per_group = (256 / G) / 4;
  if (tid < G * 2) {
    const int g = tid >> 1, which = tid & 1;
    double t = 0.0;
    for (int l = 0; l < lanes_per_group; ++l)
      for (int k = 0; k < GN_SUB; ++k) t += red[k][g * lanes_per_group + l][which];
    partials[(((int64_t)b * gridDim.x + blockIdx.x) * G + g) * 2 + which] = t;
  }
}

__global__ __launch_bounds__(64 * GN_SUB) void k_gn_apply(const float* __restrict__ x,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta,
                                                          float* __restrict__ y,
                                                          const double* __restrict__ partials,
                                                          int64_t HW, int G, float eps, int relu,
                                                          int64_t xbs, int64_t ybs) {
  __shared__ double acc[GN_SUB][64];
  __shared__ float stat[64][2];
  const int tid = threadIdx.x, b = blockIdx.y;
  const int nblk = gridDim.x;
  {  // reduce partials of image b: 2G columns, 4 row-strided lanes each
    const int col = tid & 63, part = tid >> 6;
    double t = 0.0;
    if (col < 2 * G)
      for (int i = part; i < nblk; i += GN_SUB)
        t += partials[((int64_t)b * nblk + i) * G * 2 + col];
    acc[part][col] = t;
  }
  __syncthreads();
  if (tid < G) {
    const double n = (double)HW * (256 / G);
    double s = 0.0, ss = 0.0;
    for (int k = 0; k < GN_SUB; ++k) {
      s += acc[k][2 * tid];
      ss += acc[k][2 * tid + 1];
    }
    const double mean = s / n;
    double var = ss / n - mean * mean;
    if (var < 0.0) var = 0.0;
    stat[tid][0] = (float)mean;
    stat[tid][1] = (float)(1.0 / sqrt(var + (double)eps));
  }
  __syncthreads();
  const int c4 = tid & 63, sub = tid >> 6;
  const int g = (c4 * 4) / (256 / G);
  const float mean = stat[g][0], rstd = stat[g][1];
  const float4 gg = ld4(gamma + c4 * 4), bb = ld4(beta + c4 * 4);
  const int64_t p0 = (int64_t)blockIdx.x * GN_PIX;
  const float* xb = x + (int64_t)b * xbs;
  float* yb = y + (int64_t)b * ybs;
  constexpr int NP = GN_PIX / GN_SUB;
  float4 v[NP];
#pragma unroll
  for (int j = 0; j < NP; ++j)
    v[j] = ld4(xb + min(p0 + sub + j * GN_SUB, HW - 1) * 256 + c4 * 4);
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    const int64_t pix = p0 + sub + j * GN_SUB;
    float4 o = make_float4((v[j].x - mean) * rstd * gg.x + bb.x, (v[j].y - mean) * rstd * gg.y + bb.y,
                           (v[j].z - mean) * rstd * gg.z + bb.z, (v[j].w - mean) * rstd * gg.w + bb.w);
    if (relu) {
      o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f);
      o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
    }
    if (pix < HW) st4(yb + pix * 256 + c4 * 4, o);
  }
}

extern "C" int pn_groupnorm_nhwc_f32(const float* x, const float* gamma, const float* beta,
                                     float* y, double* partials, int B, int64_t HW, int C,
                                     int G, float eps, int relu, int64_t x_bstride,
                                     int64_t y_bstride, void* stream) {
  if (!x || !gamma || !beta || !y || !partials) return PN_BAD_ARG;
  if (C != 256 || G <= 0 || G > 32 || 256 % G || (256 / G) % 4 || B <= 0 || HW <= 0)
    return PN_BAD_ARG;
  if ((x_bstride | y_bstride) & 3) return PN_BAD_ARG;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(pn_groupnorm_nblk(HW), B);
  hipLaunchKernelGGL(k_gn_partial, grid, dim3(64 * GN_SUB), 0, s, x, partials, HW, G, x_bstride);
  hipLaunchKernelGGL(k_gn_apply, grid, dim3(64 * GN_SUB), 0, s, x, gamma, beta, y, partials, HW, G,
                     eps, relu, x_bstride, y_bstride);
  return PN_LAUNCH_CHECK();
}

__global__ __launch_bounds__(256) void k_l2norm256(const float* __restrict__ x,
                                                   float* __restrict__ y, int64_t rows,
                                                   float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float4 v = ld4(x + row * 256 + lane * 4);
  const float n = sqrtf(wave_sum((v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w)));
  const float d = fmaxf(n, eps);
  st4(y + row * 256 + lane * 4, make_float4(v.x / d, v.y / d, v.z / d, v.w / d));
}

extern "C" int pn_l2normalize_f32(const float* x, float* y, int64_t rows, int C, float eps,
                                  void* stream) {
  if (!x || !y || C != 256 || rows <= 0) return PN_BAD_ARG;
  hipLaunchKernelGGL(k_l2norm256, dim3(pn_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream,
                     x, y, rows, eps);
  return PN_LAUNCH_CHECK();
}
