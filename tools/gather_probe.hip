// Micro-probe (tuning aid): rate of the deformable-attention access pattern -- W lanes read
// one random 128-byte line (W = 8: float4 per lane, the kernel's shape; 16: float2; 32: float),
// 12 * W / 8 independent lines per lane group -- drawn from a 32 MB window (HBM-bound) or a
// 512 KB one (L2-resident).  All index loads are hoisted before the gathers.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
template <int W>
__global__ __launch_bounds__(256) void k(const float* __restrict__ v, const int* __restrict__ idx,
                                         float* __restrict__ out, int mask) {
  constexpr int VEC = 32 / W;                 // floats per lane
  constexpr int NL = 12 * (W / 8);            // lines per lane group (same bytes per thread)
  const int tid = threadIdx.x, c = tid % W, grp = tid / W;
  const int wg = blockIdx.x;
  float r[NL][VEC];
  float acc = 0.f;
#pragma unroll
  for (int j = 0; j < NL; ++j) {
    const int line = idx[((size_t)wg * (256 / W) + grp) * NL + j] & mask;
    const float* p = v + (size_t)line * 32 + c * VEC;
    if (VEC == 4) { const float4 t = *reinterpret_cast<const float4*>(p); r[j][0] = t.x; r[j][1 % VEC] = t.y; r[j][2 % VEC] = t.z; r[j][3 % VEC] = t.w; }
    else if (VEC == 2) { const float2 t = *reinterpret_cast<const float2*>(p); r[j][0] = t.x; r[j][1 % VEC] = t.y; }
    else r[j][0] = *p;
  }
#pragma unroll
  for (int j = 0; j < NL; ++j)
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc += r[j][e];
  out[(size_t)wg * 256 + tid] = acc;
}
template <int W> void run(const float* v, const int* idx, float* out, int NWG, int mask) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<W>, dim3(NWG), dim3(256), 0, 0, v, idx, out, mask);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<W>, dim3(NWG), dim3(256), 0, 0, v, idx, out, mask);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  const double bytes = (double)NWG * 256 * 12 * 16;
  printf("%2d lanes x %2d B per line, window %7.1f KB: %7.1f us per layer-equivalent, %6.2f TB/s\n", W, 128 / W,
         (mask + 1) * 128 / 1024.0, ms * 1e3 / 4, bytes / ms / 1e9);
}
int main() {
  const int NWG = 21950 * 4;
  float* v; int* idx; float* out;
  hipMalloc(&v, (size_t)(1 << 18) * 128); hipMemset(v, 0, (size_t)(1 << 18) * 128);
  hipMalloc(&out, (size_t)NWG * 256 * 4);
  std::vector<int> h((size_t)NWG * 32 * 12 * 4);
  for (auto& x : h) x = rand();
  hipMalloc(&idx, h.size() * 4); hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  for (int mask : {(1 << 18) - 1, (1 << 12) - 1}) {
    run<8>(v, idx, out, NWG, mask); run<16>(v, idx, out, NWG, mask); run<32>(v, idx, out, NWG, mask);
  }
  return 0;
}
