// Micro-probe (tuning aid): cost of a software grid barrier (one atomic counter, bounded
// spin) for a few dozen co-resident workgroups -- alone, and while a chip-filling kernel
// with 4 resident workgroups per CU runs on another stream (the situation of a per-layer
// persistent kernel for the query chains beside stage A's persistent GEMMs).
// Every spin is BOUNDED: a barrier that does not complete sets a flag and the kernel exits.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void k_phases(unsigned* counter, int phases, int* failed,
                                                float* sink, unsigned long long* stamps) {
  float acc = 0.f;
#ifdef PRIO
  __builtin_amdgcn_s_setprio(3);
#endif
  if (blockIdx.x == 0 && threadIdx.x == 0) stamps[0] = wall_clock64();
  for (int p = 0; p < phases; ++p) {
    for (int i = 0; i < 64; ++i) acc += __sinf(acc + i);        // a little work per phase
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence();
      atomicAdd(counter, 1u);
      const unsigned target = (unsigned)(p + 1) * gridDim.x;
      int spins = 0;
      while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > (1 << 22)) { *failed = 1; break; }         // ~ tens of ms: give up
      }
    }
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x == 0 && p == 0) stamps[1] = wall_clock64();
    if (*failed) break;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) stamps[2] = wall_clock64();
  if (acc == 12345.f) sink[0] = acc;
}

// chip filler: 1024 workgroups x 4 waves of dependent MFMAs, ~`iters` x 16 x 64 cycles each
__global__ __launch_bounds__(256) void k_filler(float* out, int iters) {
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float a = threadIdx.x * 1e-3f, b = 0.5f;
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += acc[r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  unsigned* counter; int* failed; float *sink, *out; unsigned long long* stamps;
  hipMalloc(&stamps, 24);
  hipMalloc(&counter, 4); hipMalloc(&failed, 4); hipMalloc(&sink, 4); hipMalloc(&out, 1024 * 256 * 4);
  hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int P = 200;
  for (int wgs : {8, 32, 64, 128}) {
    for (int busy = 0; busy < 2; ++busy) {
      hipMemsetAsync(counter, 0, 4, s2); hipMemsetAsync(failed, 0, 4, s2);
      hipStreamSynchronize(s2);
      if (busy)   // ~100 us kernels back to back, like stage A's GEMMs
        for (int r = 0; r < 40; ++r) hipLaunchKernelGGL(k_filler, dim3(1024), dim3(256), 0, s1, out, 60);
      hipEventRecord(e0, s2);
      hipLaunchKernelGGL(k_phases, dim3(wgs), dim3(256), 0, s2, counter, P, failed, sink, stamps);
      hipEventRecord(e1, s2);
      hipEventSynchronize(e1);
      hipDeviceSynchronize();
      float ms; hipEventElapsedTime(&ms, e0, e1);
      int f; hipMemcpy(&f, failed, 4, hipMemcpyDeviceToHost);
      unsigned long long st[3]; hipMemcpy(st, stamps, 24, hipMemcpyDeviceToHost);
      // wall_clock64 ticks at 100 MHz
      printf("%3d workgroups, %s: first barrier after %8.1f us, then %6.2f us per phase+barrier%s\n", wgs,
             busy ? "beside a chip-filling kernel stream" : "alone                             ",
             (st[1] - st[0]) / 100.0, (st[2] - st[1]) / 100.0 / (P - 1), f ? "  (a barrier timed out)" : "");
    }
  }
  return 0;
}
