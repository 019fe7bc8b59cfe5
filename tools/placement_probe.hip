// Tuning aid: where does the hardware put the workgroups of a grid that does NOT fill the
// chip?  Each workgroup (256 threads, 18 KB of LDS, ~96 VGPRs' worth of occupancy: at most 5
// per CU, like k_gemm_tile) records the XCD / shader engine / CU it runs on and spins ~30 us so
// that the whole grid is resident at once; the host prints the histogram of workgroups per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

__global__ __launch_bounds__(256, 5) void k_where(unsigned* out, long long spin) {
  __shared__ float pad[18 * 256];
  pad[threadIdx.x] = (float)threadIdx.x;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    out[2 * blockIdx.x] = hw;
    out[2 * blockIdx.x + 1] = xcc;
  }
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin) {}
  if (pad[(threadIdx.x * 7) & 255] < 0.f) out[0] = 0;
}

int main(int argc, char** argv) {
  std::vector<int> grids = {264, 528, 792, 1056, 1216, 1280, 2088};
  unsigned* d;
  hipMalloc(&d, 2 * 4096 * sizeof(unsigned));
  std::vector<unsigned> h(2 * 4096);
  for (int g : grids) {
    hipMemset(d, 0, 2 * 4096 * sizeof(unsigned));
    hipLaunchKernelGGL(k_where, dim3(g), dim3(256), 0, 0, d, 3000LL);   // 100 MHz clock: 30 us
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d, 2 * g * sizeof(unsigned), hipMemcpyDeviceToHost);
    std::map<unsigned, int> per_cu;
    std::map<unsigned, int> per_xcd;
    for (int b = 0; b < g; ++b) {
      const unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 15;
      const unsigned cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
      per_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu]++;
      per_xcd[xcc]++;
    }
    int hist[16] = {0};
    for (auto& kv : per_cu) hist[kv.second < 15 ? kv.second : 15]++;
    printf("grid %5d: %3zu CUs used; workgroups per CU histogram:", g, per_cu.size());
    for (int i = 1; i < 16; ++i)
      if (hist[i]) printf(" %dx%d", hist[i], i);
    printf("   per XCD:");
    for (auto& kv : per_xcd) printf(" %d", kv.second);
    printf("\n");
  }
  return 0;
}
