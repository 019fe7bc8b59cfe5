// Shared device helpers for the gfx950 kernels (wave64, CDNA4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pairnet_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define PN_LAUNCH_CHECK() ((int)hipGetLastError())
#define PN_BAD_ARG (-1)

static inline int pn_cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// 32x32x2 f32 MFMA: D = A(32x2) * B(2x32) + C.  Lane l supplies A[l&31][l>>5] and
// B[l>>5][l&31]; D/C: col = l&31, row = (r&3) + 8*(r>>2) + 4*(l>>5), r in [0,16).
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int mfma32_row(int r, int lane_hi) {
  return (r & 3) + 8 * (r >> 2) + 4 * lane_hi;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// LayerNorm of one 256-channel row held as a float4 per lane (one wave = one row): two-pass
// moments like torch's CPU LayerNorm.  ONE definition, compiled WITHOUT implicit fused
// multiply-add contraction: hipcc otherwise picks the products it fuses per call site, and
// kernels that are documented as bit-identical (pn_layernorm_f32, the second norm of
// pn_ffn_ln2_f32, the epilogue of pn_linear_res_ln_f32) would only agree by luck.
__device__ __forceinline__ float4 ln256_row(const float4 v, const float4 gg, const float4 bb,
                                            const float eps) {
#pragma clang fp contract(off)
  const float mean = wave_sum((v.x + v.y) + (v.z + v.w)) * (1.f / 256.f);
  const float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
  const float var = wave_sum((dx * dx + dy * dy) + (dz * dz + dw * dw)) * (1.f / 256.f);
  const float rstd = 1.f / sqrtf(var + eps);
  return make_float4((dx * rstd) * gg.x + bb.x, (dy * rstd) * gg.y + bb.y,
                     (dz * rstd) * gg.z + bb.z, (dw * rstd) * gg.w + bb.w);
}

__device__ __forceinline__ float4 ld4(const float* p) {
  return *reinterpret_cast<const float4*>(p);
}
__device__ __forceinline__ void st4(float* p, float4 v) {
  *reinterpret_cast<float4*>(p) = v;
}
__device__ __forceinline__ float4 add4(float4 a, float4 b) {
  return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}

// 16-byte load through a buffer descriptor: address = rsrc.base + voff (per lane, bytes)
// + soff (wave-uniform SGPR, bytes).  The k advance of the GEMM loops rides in soff, so
// the loop issues NO per-lane address arithmetic (flat loads need a 64-bit VALU add per
// load, and VALU issued between MFMAs on one accumulator costs ~43 cycles a piece).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* base) {
  // raw buffer, no bounds clamp (callers keep offsets inside the operand)
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ float buf_ld1(__amdgpu_buffer_rsrc_t r, unsigned voff_bytes,
                                         int soff_bytes) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)voff_bytes, soff_bytes, 0));
}
__device__ __forceinline__ float4 buf_ld4(__amdgpu_buffer_rsrc_t r, unsigned voff_bytes,
                                          int soff_bytes) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff_bytes, soff_bytes, 0);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z),
                     __uint_as_float(v.w));
}

// XCD-aware tile order.  MI355X has 8 XCDs with private 4 MB L2s and the dispatcher
// places workgroup L on XCD L % 8, so consecutive workgroups (which want to share an
// operand panel) land on eight different L2s.  Map the launch index L to the logical
// tile index T so that each XCD walks a CONTIGUOUS range of T (bijective for any
// count; performance only -- any placement is correct).
__device__ __forceinline__ int xcd_tile_index(int L, int ntiles) {
  const int xcd = L & 7, j = L >> 3;
  const int q = ntiles >> 3, r = ntiles & 7;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + j;
}

// ATen's upsample_bilinear2d source index (align_corners=False, no scale factor):
//   src = max(scale * (dst + 0.5) - 0.5, 0), scale = in / out;
//   i0 = (int)src, i1 = i0 + (i0 < in-1), l1 = src - i0, l0 = 1 - l1.
struct Tap { int i0, i1; float l0, l1; };
__device__ __forceinline__ Tap make_tap(int dst, int in, int outn) {
  const float scale = (float)in / (float)outn;
  float src = scale * ((float)dst + 0.5f) - 0.5f;
  if (src < 0.f) src = 0.f;
  Tap t;
  t.i0 = (int)src;
  if (t.i0 > in - 1) t.i0 = in - 1;
  t.i1 = t.i0 + ((t.i0 < in - 1) ? 1 : 0);
  t.l1 = src - (float)t.i0;
  t.l0 = 1.f - t.l1;
  return t;
}
// out = l0y (l0x v00 + l1x v01) + l1y (l0x v10 + l1x v11): ONE definition for every kernel
// that resamples, so that all of them round alike
__device__ __forceinline__ float tap_blend(const Tap& ty, const Tap& tx, float v00, float v01,
                                           float v10, float v11) {
  return ty.l0 * (tx.l0 * v00 + tx.l1 * v01) + ty.l1 * (tx.l0 * v10 + tx.l1 * v11);
}
