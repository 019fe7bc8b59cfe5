// Sine positional encoding and bilinear resampling (align_corners=False), HBM-bound.
#include "common.h"

// SinePositionalEncoding(num_feats = C/2, normalize=True, scale=2pi, eps=1e-6)
// for an all-valid mask: channels [pos_y | pos_x], each
// interleaved sin (even) / cos (odd) over dim_t[i] = T^(2*(i/2)/num_feats).
__global__ void k_sine_pe(float* __restrict__ out, const float* __restrict__ add, int h,
                          int w, int C, float temperature, float offset, int vh, int vw) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)h * w * C;
  if (e >= total) return;
  const int c = (int)(e % C);
  const int pix = (int)(e / C);
  const int y = pix / w, x = pix - y * w;
  const int nf = C / 2;
  const int i = (c < nf) ? c : c - nf;
  const float scale = 6.283185307179586f;
  // cumulative sums of the not-padded flags down a column / along a row: inside the valid
  // region they count, past it they stay at its size, and a fully padded column / row sums to
  // zero everywhere, normaliser included (the reference then encodes -0.5 / eps there; those
  // tokens are masked out of every consumer)
  const int ye = x < vw ? min(y + 1, vh) : 0, ny = x < vw ? vh : 0;
  const int xe = y < vh ? min(x + 1, vw) : 0, nx = y < vh ? vw : 0;
  const float embed = (c < nf) ? (((float)ye + offset) / ((float)ny + 1e-6f)) * scale
                               : (((float)xe + offset) / ((float)nx + 1e-6f)) * scale;
  const float dim_t = powf(temperature, (float)(2 * (i / 2)) / (float)nf);
  const float v = embed / dim_t;
  float r = (i & 1) ? cosf(v) : sinf(v);
  if (add) r += add[c];
  out[e] = r;
}

extern "C" int pn_sine_pe_valid_f32(float* out, const float* add, int h, int w, int valid_h,
                                    int valid_w, int C, float temperature, float offset,
                                    void* stream) {
  if (!out || h <= 0 || w <= 0 || C <= 0 || (C & 3) || valid_h <= 0 || valid_h > h ||
      valid_w <= 0 || valid_w > w)
    return PN_BAD_ARG;
  const int64_t total = (int64_t)h * w * C;
  hipLaunchKernelGGL(k_sine_pe, dim3(pn_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream,
                     out, add, h, w, C, temperature, offset, valid_h, valid_w);
  return PN_LAUNCH_CHECK();
}

extern "C" int pn_sine_pe_offset_f32(float* out, const float* add, int h, int w, int C,
                                     float temperature, float offset, void* stream) {
  return pn_sine_pe_valid_f32(out, add, h, w, h, w, C, temperature, offset, stream);
}

extern "C" int pn_sine_pe_f32(float* out, const float* add, int h, int w, int C,
                              float temperature, void* stream) {
  return pn_sine_pe_offset_f32(out, add, h, w, C, temperature, 0.f, stream);
}

__global__ __launch_bounds__(256) void k_bilinear_nhwc(const float* __restrict__ in,
                                                       float* __restrict__ out, int hi, int wi,
                                                       int ho, int wo, int C4, int accumulate,
                                                       int64_t ibs, int64_t obs) {
  // thread = (output pixel, float4 channel group)
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t per_img = (int64_t)ho * wo * C4;
  if (e >= per_img) return;
  const int b = blockIdx.y;
  const int c4 = (int)(e % C4);
  const int pix = (int)(e / C4);
  const int oy = pix / wo, ox = pix - oy * wo;
  const Tap ty = make_tap(oy, hi, ho), tx = make_tap(ox, wi, wo);
  const float* ib = in + (int64_t)b * ibs + c4 * 4;
  const int64_t rs = (int64_t)wi * C4 * 4, cs = (int64_t)C4 * 4;
  const float4 v00 = ld4(ib + ty.i0 * rs + tx.i0 * cs), v01 = ld4(ib + ty.i0 * rs + tx.i1 * cs);
  const float4 v10 = ld4(ib + ty.i1 * rs + tx.i0 * cs), v11 = ld4(ib + ty.i1 * rs + tx.i1 * cs);
  float4 r;
  r.x = ty.l0 * (tx.l0 * v00.x + tx.l1 * v01.x) + ty.l1 * (tx.l0 * v10.x + tx.l1 * v11.x);
  r.y = ty.l0 * (tx.l0 * v00.y + tx.l1 * v01.y) + ty.l1 * (tx.l0 * v10.y + tx.l1 * v11.y);
  r.z = ty.l0 * (tx.l0 * v00.z + tx.l1 * v01.z) + ty.l1 * (tx.l0 * v10.z + tx.l1 * v11.z);
  r.w = ty.l0 * (tx.l0 * v00.w + tx.l1 * v01.w) + ty.l1 * (tx.l0 * v10.w + tx.l1 * v11.w);
  float* o = out + (int64_t)b * obs + (int64_t)pix * C4 * 4 + c4 * 4;
  if (accumulate) r = add4(ld4(o), r);
  st4(o, r);
}

extern "C" int pn_bilinear_nhwc_f32(const float* in, float* out, int B, int hi, int wi, int ho,
                                    int wo, int C, int accumulate, int64_t in_bstride,
                                    int64_t out_bstride, void* stream) {
  if (!in || !out || B <= 0 || hi <= 0 || wi <= 0 || ho <= 0 || wo <= 0 || C <= 0 || (C & 3))
    return PN_BAD_ARG;
  if ((in_bstride | out_bstride) & 3) return PN_BAD_ARG;
  const int64_t per_img = (int64_t)ho * wo * (C / 4);
  hipLaunchKernelGGL(k_bilinear_nhwc, dim3(pn_cdiv(per_img, 256), B), dim3(256), 0,
                     (hipStream_t)stream, in, out, hi, wi, ho, wo, C / 4, accumulate, in_bstride,
                     out_bstride);
  return PN_LAUNCH_CHECK();
}

template <bool GT0>
__global__ __launch_bounds__(256) void k_bilinear_planar(const float* __restrict__ in,
                                                         void* __restrict__ outv, int hi, int wi,
                                                         int ho, int wo) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t per_plane = (int64_t)ho * wo;
  if (e >= per_plane) return;
  const int64_t pl = blockIdx.y;
  const int oy = (int)(e / wo), ox = (int)(e - (int64_t)oy * wo);
  const Tap ty = make_tap(oy, hi, ho), tx = make_tap(ox, wi, wo);
  const float* ib = in + pl * hi * wi;
  const float v00 = ib[(int64_t)ty.i0 * wi + tx.i0], v01 = ib[(int64_t)ty.i0 * wi + tx.i1];
  const float v10 = ib[(int64_t)ty.i1 * wi + tx.i0], v11 = ib[(int64_t)ty.i1 * wi + tx.i1];
  const float r = tap_blend(ty, tx, v00, v01, v10, v11);
  if (GT0) ((uint8_t*)outv)[pl * per_plane + e] = r > 0.f ? 1 : 0;
  else     ((float*)outv)[pl * per_plane + e] = r;
}

// wo % 4 == 0: thread = 4 consecutive pixels of a row (one row tap, 32-bit index math, one
// 4- or 16-byte store): the per-pixel form above spends most of its time in the 64-bit
// division and in byte stores.  Same arithmetic per pixel.
template <bool GT0>
__global__ __launch_bounds__(256) void k_bilinear_planar4(const float* __restrict__ in,
                                                          void* __restrict__ outv, int hi, int wi,
                                                          int ho, int wo) {
  const int wq = wo >> 2;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= ho * wq) return;
  const int64_t pl = blockIdx.y;
  const int oy = e / wq, ox = (e - oy * wq) * 4;
  const Tap ty = make_tap(oy, hi, ho);
  const float* r0 = in + pl * hi * wi + (int64_t)ty.i0 * wi;
  const float* r1 = in + pl * hi * wi + (int64_t)ty.i1 * wi;
  float r[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const Tap tx = make_tap(ox + j, wi, wo);
    const float v00 = r0[tx.i0], v01 = r0[tx.i1], v10 = r1[tx.i0], v11 = r1[tx.i1];
    r[j] = tap_blend(ty, tx, v00, v01, v10, v11);
  }
  const int64_t o = pl * ((int64_t)ho * wo) + (int64_t)oy * wo + ox;
  if (GT0) {
    const uint32_t m = (r[0] > 0.f ? 1u : 0u) | (r[1] > 0.f ? 0x100u : 0u) |
                       (r[2] > 0.f ? 0x10000u : 0u) | (r[3] > 0.f ? 0x1000000u : 0u);
    *reinterpret_cast<uint32_t*>((uint8_t*)outv + o) = m;
  } else {
    st4((float*)outv + o, make_float4(r[0], r[1], r[2], r[3]));
  }
}

static int launch_planar(const float* in, void* out, int64_t P, int hi, int wi, int ho, int wo,
                         bool gt0, void* stream) {
  if (!in || !out || P <= 0 || P > 65535 || hi <= 0 || wi <= 0 || ho <= 0 || wo <= 0)
    return PN_BAD_ARG;
  hipStream_t s = (hipStream_t)stream;
  if ((wo & 3) == 0 && ((uintptr_t)out & 15) == 0 && (int64_t)ho * wo < ((int64_t)1 << 31)) {
    dim3 grid(pn_cdiv((int64_t)ho * (wo >> 2), 256), (unsigned)P);
    if (gt0)
      hipLaunchKernelGGL(k_bilinear_planar4<true>, grid, dim3(256), 0, s, in, out, hi, wi, ho, wo);
    else
      hipLaunchKernelGGL(k_bilinear_planar4<false>, grid, dim3(256), 0, s, in, out, hi, wi, ho, wo);
    return PN_LAUNCH_CHECK();
  }
  dim3 grid(pn_cdiv((int64_t)ho * wo, 256), (unsigned)P);
  if (gt0)
    hipLaunchKernelGGL(k_bilinear_planar<true>, grid, dim3(256), 0, s, in, out,
                       hi, wi, ho, wo);
  else
    hipLaunchKernelGGL(k_bilinear_planar<false>, grid, dim3(256), 0, s, in, out,
                       hi, wi, ho, wo);
  return PN_LAUNCH_CHECK();
}

extern "C" int pn_bilinear_planar_f32(const float* in, float* out, int64_t P, int hi, int wi,
                                      int ho, int wo, void* stream) {
  return launch_planar(in, out, P, hi, wi, ho, wo, false, stream);
}
extern "C" int pn_bilinear_planar_gt0_u8(const float* in, uint8_t* out, int64_t P, int hi, int wi,
                                         int ho, int wo, void* stream) {
  return launch_planar(in, out, P, hi, wi, ho, wo, true, stream);
}

// ---- the rows a bilinear resize reads: out[b][t][p][:] = in[b][tap t of output pixel p][:],
// t = 0..3 <-> (i0,j0), (i0,j1), (i1,j0), (i1,j1) of make_tap.  With these rows as the W
// operand of the mask-logit GEMM, the FULL-RESOLUTION logits a layer's attention mask needs
// (pairnet_head.py:244-246: interpolate(mask_pred) reads 4 of them per output pixel) come out
// of a Q x 4 N_l GEMM instead of the Q x H2 W2 one; pn_mask_pack_stencil blends them. ----
__global__ __launch_bounds__(256) void k_stencil_rows(const float* __restrict__ in,
                                                      float* __restrict__ out, int hi, int wi,
                                                      int ho, int wo, int C4, int64_t ibs,
                                                      int64_t obs) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;   // (tap, pixel, float4 group)
  const int64_t n = (int64_t)ho * wo;
  if (e >= 4 * n * C4) return;
  const int c = (int)(e % C4);
  const int64_t tp = e / C4;
  const int t = (int)(tp / n);
  const int pix = (int)(tp - (int64_t)t * n);
  const int oy = pix / wo, ox = pix - oy * wo;
  const Tap ty = make_tap(oy, hi, ho), tx = make_tap(ox, wi, wo);
  const int sy = (t & 2) ? ty.i1 : ty.i0, sx = (t & 1) ? tx.i1 : tx.i0;
  const float* ib = in + (int64_t)blockIdx.y * ibs;
  st4(out + (int64_t)blockIdx.y * obs + e * 4, ld4(ib + ((int64_t)sy * wi + sx) * C4 * 4 + c * 4));
}

extern "C" int pn_bilinear_stencil_rows_f32(const float* in, float* out, int B, int hi, int wi,
                                            int ho, int wo, int C, int64_t in_bstride,
                                            int64_t out_bstride, void* stream) {
  if (!in || !out || B <= 0 || B > 65535 || hi <= 0 || wi <= 0 || ho <= 0 || wo <= 0 || C <= 0 ||
      (C & 3) || ((in_bstride | out_bstride) & 3))
    return PN_BAD_ARG;
  const int64_t total = (int64_t)4 * ho * wo * (C / 4);
  hipLaunchKernelGGL(k_stencil_rows, dim3(pn_cdiv(total, 256), B), dim3(256), 0,
                     (hipStream_t)stream, in, out, hi, wi, ho, wo, C / 4, in_bstride, out_bstride);
  return PN_LAUNCH_CHECK();
}

// ---- 3x3 stride-2 pad-1 max pooling, channel-last (ResNet stem) ----
__global__ __launch_bounds__(256) void k_maxpool3x3s2(const float* __restrict__ in,
                                                      float* __restrict__ out, int H, int W,
                                                      int Ho, int Wo, int C4) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;   // (pixel, float4 channel group)
  const int64_t total = (int64_t)Ho * Wo * C4;
  if (e >= total) return;
  const int b = blockIdx.y;
  const int c = (int)(e % C4);
  const int pix = (int)(e / C4);
  const int oy = pix / Wo, ox = pix - oy * Wo;
  const float* ib = in + (int64_t)b * H * W * C4 * 4;
  float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int iy = oy * 2 - 1 + ky;
    if (iy < 0 || iy >= H) continue;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int ix = ox * 2 - 1 + kx;
      if (ix < 0 || ix >= W) continue;
      const float4 v = ld4(ib + ((int64_t)iy * W + ix) * C4 * 4 + c * 4);
      m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
    }
  }
  st4(out + ((int64_t)b * Ho * Wo * C4 + e) * 4, m);
}

extern "C" int pn_maxpool3x3s2_nhwc_f32(const float* in, float* out, int B, int H, int W, int C,
                                        void* stream) {
  if (!in || !out || B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 4 ||
      (((uintptr_t)in | (uintptr_t)out) & 15))
    return PN_BAD_ARG;
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const int64_t total = (int64_t)Ho * Wo * (C / 4);
  hipLaunchKernelGGL(k_maxpool3x3s2, dim3(pn_cdiv(total, 256), B), dim3(256), 0,
                     (hipStream_t)stream, in, out, H, W, Ho, Wo, C / 4);
  return PN_LAUNCH_CHECK();
}
