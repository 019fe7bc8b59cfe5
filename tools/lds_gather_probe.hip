// Micro-probe (tuning aid): the deformable-attention gather served from LDS instead of L1:
// a workgroup stages a window of 128-byte rows with coalesced loads, then every 8-lane group
// reads random rows of it as float4s (ds_read_b128).  Compare with tools/gather_probe.hip
// (26 TB/s L2-resident, 7.8 TB/s from HBM for the same pattern straight from global memory).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
// one workgroup = 64 "queries" x 1 head: per level stage `rows` rows, then 64 q x 4 pts x 4 taps
__global__ __launch_bounds__(256) void k(const float* __restrict__ v, const int* __restrict__ idx,
                                         float* __restrict__ out, int rows) {
  extern __shared__ __attribute__((aligned(16))) float win[];
  const int tid = threadIdx.x, c4 = tid & 7, grp = tid >> 3;   // 32 groups of 8 lanes
  const int wg = blockIdx.x;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int lvl = 0; lvl < 3; ++lvl) {
    __syncthreads();
    const float* src = v + ((size_t)((wg * 131 + lvl * 7919) & 0xffff) * 32);   // window origin
    for (int e = tid; e < rows * 8; e += 256)
      *reinterpret_cast<float4*>(win + e * 4) = *reinterpret_cast<const float4*>(src + e * 4);
    __syncthreads();
    // 64 queries x 4 points = 256 (q, pt) items, 32 groups -> 8 passes, 4 taps each
#pragma unroll 2
    for (int pass = 0; pass < 8; ++pass) {
      const int* ip = idx + (((size_t)wg * 3 + lvl) * 256 + pass * 32 + grp) * 4;
      float4 r[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int row = (unsigned)ip[t] % (unsigned)rows;
        r[t] = *reinterpret_cast<const float4*>(win + row * 32 + c4 * 4);
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) { acc.x += r[t].x; acc.y += r[t].y; acc.z += r[t].z; acc.w += r[t].w; }
    }
  }
  out[(size_t)wg * 256 + tid] = acc.x + acc.y + acc.z + acc.w;
}
int main() {
  const int NQ = 21950, NWG = (NQ / 64) * 8 * 4;   // (patches of 64 queries) x 8 heads, x4 repeats
  float* v; int* idx; float* out;
  hipMalloc(&v, (size_t)(0x10000 + 4096) * 128); hipMemset(v, 0, (size_t)(0x10000 + 4096) * 128);
  hipMalloc(&out, (size_t)NWG * 256 * 4);
  std::vector<int> h((size_t)NWG * 3 * 256 * 4);
  for (auto& x : h) x = rand();
  hipMalloc(&idx, h.size() * 4); hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rows : {324, 196, 144, 64}) {
    hipLaunchKernelGGL(k, dim3(NWG), dim3(256), rows * 128, 0, v, idx, out, rows);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k, dim3(NWG), dim3(256), rows * 128, 0, v, idx, out, rows);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("window %4d rows (%5.1f KB) x 3 levels: %7.1f us per layer-equivalent\n", rows, rows * 128 / 1024.0, ms * 1e3 / 4);
  }
  return 0;
}
