// Micro-probe (tuning aid): rate of the deformable-attention access pattern -- 8 lanes read
// one 128-byte line as float4s, 12 independent lines per lane group -- against the size of
// the window the lines are drawn from (HBM/L2-resident 22 MB ... L1-resident 16 KB).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ __launch_bounds__(256) void k(const float* __restrict__ v, const int* __restrict__ idx,
                                         float* __restrict__ out, int nlines_mask, int per_wg_window) {
  const int tid = threadIdx.x, c4 = tid & 7, grp = tid >> 3;
  const int wg = blockIdx.x;
  // window base: per_wg_window != 0 -> each workgroup draws from its own window (spatial
  // locality: neighbouring workgroups overlap), else all from the whole buffer
  const int base = per_wg_window ? (((wg * 37) & ~nlines_mask) & ((1 << 18) - 1)) : 0;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 r[12];
#pragma unroll
  for (int j = 0; j < 12; ++j) {
    const int line = base + (idx[(wg * 32 + grp) * 12 + j] & nlines_mask);
    r[j] = *reinterpret_cast<const float4*>(v + (size_t)line * 32 + c4 * 4);
  }
#pragma unroll
  for (int j = 0; j < 12; ++j) { acc.x += r[j].x; acc.y += r[j].y; acc.z += r[j].z; acc.w += r[j].w; }
  acc.x += __shfl_xor(acc.x, 8, 64); acc.x += __shfl_xor(acc.x, 16, 64);
  if (((tid >> 3) & 3) == 0) out[(size_t)wg * 64 + (tid >> 5) * 8 + c4] = acc.x + acc.y + acc.z + acc.w;
}
int main() {
  const int NWG = 21950 * 4;               // 4x the layer's work for a stable timing
  const size_t total_lines = 1 << 18;      // 32 MB buffer
  float* v; int* idx; float* out;
  hipMalloc(&v, total_lines * 128 + (1 << 20)); hipMemset(v, 0, total_lines * 128 + (1 << 20));
  hipMalloc(&out, (size_t)NWG * 64 * 4);
  std::vector<int> h((size_t)NWG * 32 * 12);
  for (auto& x : h) x = rand();
  hipMalloc(&idx, h.size() * 4); hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 2; ++mode)
    for (int lines : {1 << 18, 1 << 15, 1 << 12, 1 << 9, 1 << 7, 1 << 5}) {
      if (mode == 1 && lines > (1 << 12)) continue;
      hipLaunchKernelGGL(k, dim3(NWG), dim3(256), 0, 0, v, idx, out, lines - 1, mode);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k, dim3(NWG), dim3(256), 0, 0, v, idx, out, lines - 1, mode);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
      const double bytes = (double)NWG * 256 * 12 * 16;
      printf("%s window %8.1f KB: %7.1f us per layer-equivalent, %6.2f TB/s gathered\n",
             mode ? "per-WG " : "global ", lines * 128 / 1024.0, ms * 1e3 / 4, bytes / ms / 1e9);
    }
  return 0;
}
