// Micro-probe (tuning aid, not part of the library): f32 MFMA throughput vs number of
// independent accumulators per wave and waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = a0 + threadIdx.x * 1e-6f, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC> void run(int wps, float* out) {
  const int iters = 4096 / NACC * 4;
  dim3 grid(256 * wps);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NACC>, grid, dim3(256), 0, 0, out, iters, 1.0f, 0.5f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NACC>, grid, dim3(256), 0, 0, out, iters, 1.0f, 0.5f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flops = (double)grid.x * 4 * iters * NACC * 2.0 * 32 * 32 * 2;
  printf("nacc %d waves/SIMD %d: %.1f TFLOP/s (%.3f ms)\n", NACC, wps, flops / ms / 1e9, ms);
}
int main() {
  float* out; hipMalloc(&out, sizeof(float) * 256 * 256 * 8);
  for (int wps : {1, 2, 4, 7}) { run<1>(wps, out); run<2>(wps, out); run<4>(wps, out); }
  return 0;
}
