// A kernel that does nothing but issue MFMAs (no memory, no LDS, no barrier), to run BESIDE another
// kernel: does a workgroup of dense matrix instructions on the CU change what its neighbours
// compute?  (LABNOTES R5.12: k_msda beside the split-bf16 GEMM.)   Built by tools/build_probe.sh
// into tools/bin/libmfma_hammer.so; driven by tools/attic/dbg_mfma_hammer.py.
#include <hip/hip_runtime.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KIND>   // 0: v_mfma_f32_32x32x16_bf16, 1: v_mfma_f32_32x32x2_f32, 2: VALU fma only
__global__ __launch_bounds__(512, 2) void k_hammer(float* sink, int iters) {
  f32x16 acc[2];
  for (int j = 0; j < 2; ++j)
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  union { bf16x8 b; unsigned u[4]; } a, b;
  for (int i = 0; i < 4; ++i) { a.u[i] = 0x3f803f80u + threadIdx.x; b.u[i] = 0x3c003c00u + blockIdx.x; }
  float fa = 1.f + threadIdx.x * 1e-3f, fb = 0.5f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (KIND == 0) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.b, b.b, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b.b, a.b, acc[1], 0, 0, 0);
      } else if (KIND == 1) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb, fa, acc[1], 0, 0, 0);
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[0][r] = fmaf(acc[0][r], fa, fb); acc[1][r] = fmaf(acc[1][r], fb, fa); }
      }
    }
  }
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += acc[0][r] + acc[1][r];
  if (s == 12345.678f) sink[0] = s;     // keep the work alive
}

// KIND 3: six bf16-MFMA accumulators (96 registers) + operands: ~124 VGPRs at 512 threads, the
// register footprint of k_gemm_split; KIND 4: the same footprint, VALU only
template <int KIND>
__global__ __launch_bounds__(512, 2) void k_hammer_wide(float* sink, int iters) {
  f32x16 acc[6];
  for (int j = 0; j < 6; ++j)
    for (int r = 0; r < 16; ++r) acc[j][r] = (float)(threadIdx.x + j + r);
  union { bf16x8 b; unsigned u[4]; } a[3];
  for (int q = 0; q < 3; ++q)
    for (int i = 0; i < 4; ++i) a[q].u[i] = 0x3f803f80u + threadIdx.x * (q + 1) + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      if (KIND == 3) {
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[j % 3].b, a[(j + 1) % 3].b, acc[j], 0, 0, 0);
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = fmaf(acc[j][r], 0.999f, 0.5f);
      }
    }
  }
  float s = 0.f;
  for (int j = 0; j < 6; ++j)
    for (int r = 0; r < 16; ++r) s += acc[j][r];
  if (s == 12345.678f) sink[0] = s;
}

extern "C" int mfma_hammer(void* stream, float* sink, int kind, int grid, int iters) {
  hipStream_t s = (hipStream_t)stream;
  if (kind == 0) k_hammer<0><<<grid, 512, 0, s>>>(sink, iters);
  else if (kind == 1) k_hammer<1><<<grid, 512, 0, s>>>(sink, iters);
  else if (kind == 2) k_hammer<2><<<grid, 512, 0, s>>>(sink, iters);
  else if (kind == 3) k_hammer_wide<3><<<grid, 512, 0, s>>>(sink, iters);
  else k_hammer_wide<4><<<grid, 512, 0, s>>>(sink, iters);
  return (int)hipGetLastError();
}
