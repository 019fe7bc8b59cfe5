// The S3 operand format's element helpers (csrc/gemm_s3.hip has the format's description):
// shared by the GEMM and by producers that write their output pre-split (csrc/msda.hip).
#pragma once
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
union s3_frag { uint4 u; bf16x8 v; __bf16 e[8]; };

__device__ __forceinline__ void s3_split(float x, __bf16& a, __bf16& b, __bf16& c) {
  a = (__bf16)x;
  float r = x - (float)a;      // exact
  b = (__bf16)r;
  r = r - (float)b;            // exact
  c = (__bf16)r;
}
// 8 floats -> three planes
__device__ __forceinline__ void s3_split8(const float (&v)[8], s3_frag& p0, s3_frag& p1, s3_frag& p2) {
#pragma unroll
  for (int i = 0; i < 8; ++i) s3_split(v[i], p0.e[i], p1.e[i], p2.e[i]);
}
// three planes -> 8 floats, exactly the fp32 values that were split
__device__ __forceinline__ void s3_join8(const s3_frag& p0, const s3_frag& p1, const s3_frag& p2,
                                         float (&v)[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = ((float)p2.e[i] + (float)p1.e[i]) + (float)p0.e[i];
}
