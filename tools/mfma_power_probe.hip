// Micro-probe (tuning aid, not part of the library): sustained f32 MFMA throughput with
// constant vs random operands, for the 32x32x2 and 16x16x4 shapes -- how much of the
// 157 TFLOP/s roof survives the power management under realistic operand toggling.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define NV 16
template <int SHAPE>
__global__ __launch_bounds__(256) void k(const float* __restrict__ src, float* out, int iters) {
  float a[NV], b[NV];
  for (int i = 0; i < NV; ++i) {
    a[i] = src[(threadIdx.x * NV + i) & 4095];
    b[i] = src[(threadIdx.x * NV + i + 2048) & 4095];
  }
  float s = 0.f;
  if (SHAPE == 32) {
    f32x16 acc[2];
    for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[i], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[(i + 1) % NV], acc[1], 0, 0, 0);
      }
    }
    for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  } else {
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[i], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[(i + 1) % NV], acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(i + 1) % NV], b[i], acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(i + 1) % NV], b[(i + 1) % NV], acc[3], 0, 0, 0);
      }
    }
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
  }
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int SHAPE> void run(const char* what, const float* src, float* out, int wps) {
  const int iters = 6000;
  dim3 grid(256 * wps);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<SHAPE>, grid, dim3(256), 0, 0, src, out, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<SHAPE>, grid, dim3(256), 0, 0, src, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double per = SHAPE == 32 ? 2.0 * 2 * 32 * 32 * 2 : 4.0 * 2 * 16 * 16 * 4;
  double flops = 5.0 * grid.x * 4 * (double)iters * NV * per;
  printf("%-8s mfma %dx%d waves/SIMD %d: %6.1f TFLOP/s (%.1f ms)\n", what, SHAPE, SHAPE, wps, flops / ms / 1e9, ms);
}
int main() {
  float *src, *out; hipMalloc(&src, 4096 * 4); hipMalloc(&out, sizeof(float) * 256 * 256 * 8);
  float h[4096];
  for (int mode = 0; mode < 3; ++mode) {
    for (int i = 0; i < 4096; ++i) {
      float u1 = (rand() + 1.0f) / (RAND_MAX + 2.0f), u2 = (rand() + 1.0f) / (RAND_MAX + 2.0f);
      float g = sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2);
      h[i] = mode == 0 ? 1.0f : mode == 1 ? (float)((int)(g * 4)) : g;   // const | few bits | N(0,1)
    }
    hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice);
    const char* what = mode == 0 ? "const" : mode == 1 ? "lowbits" : "normal";
    for (int wps : {2, 4}) { run<32>(what, src, out, wps); run<16>(what, src, out, wps); }
  }
  return 0;
}
