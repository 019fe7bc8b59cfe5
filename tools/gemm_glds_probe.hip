// Tuning aid (not part of the library): the plain row-major 64x64 fp32-MFMA GEMM tile with
// its operands loaded global -> LDS directly (global_load_lds_dwordx4, "glds"), two LDS
// stages, ONE barrier per 32-deep chunk, against the production kernel (register staging,
// one LDS stage, two barriers).  LDS image: 128-byte rows [row][8 x float4], the float4 slot
// XOR-swizzled with (row >> 1) & 7 -- applied on the per-lane SOURCE address, since a glds
// wave-instruction writes lane-linear -- which makes the ds_read_b128 fragment reads
// conflict-free.
//   tools/build_probe.sh glds && tools/bin/gemm_glds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "pairnet_hip.h"
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ void glds16(const float* g, float* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// PERSIST: workgroups walk tiles t = blockIdx.x, + gridDim.x, ... (n fastest)
template <int WGS_PER_CU>
__global__ __launch_bounds__(256, WGS_PER_CU) void k_glds(const float* __restrict__ A,
                                                          const float* __restrict__ W,
                                                          const float* __restrict__ bias,
                                                          float* __restrict__ C, int M, int N, int K) {
  __shared__ __attribute__((aligned(1024))) float smem[2][2][64 * 32];   // [stage][A|B][row*32]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int nt = (N + 63) / 64, mt = (M + 63) / 64, ntiles = nt * mt;
  const int nk = K / 32;
  // loader role: wave-instruction j (0, 1) of this wave covers tile rows (j * 4 + wave) * 8 .. + 8;
  // lane -> (row = r0 + lane / 8, LDS slot = lane % 8) loads global piece slot ^ f(row)
  int lrow[2], lpiece[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    lrow[j] = (j * 4 + wave) * 8 + (lane >> 3);
    lpiece[j] = (lane & 7) ^ ((lrow[j] >> 1) & 7);
  }
  // fragment read offsets (floats): row * 32 + ((kb * 2 + lh) ^ f(row)) * 4
  const int arow = wm * 32 + li, brow = wn * 32 + li;
  const int fa = (arow >> 1) & 7, fb = (brow >> 1) & 7;
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int tm = t / nt, tn = t - tm * nt;
    const int m0 = tm * 64, n0 = tn * 64;
    const float* ga[2]; const float* gw[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      ga[j] = A + (int64_t)min(m0 + lrow[j], M - 1) * K + lpiece[j] * 4;
      gw[j] = W + (int64_t)min(n0 + lrow[j], N - 1) * K + lpiece[j] * 4;
    }
    auto issue = [&](int kt, int st) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        glds16(ga[j] + kt * 32, &smem[st][0][(j * 4 + wave) * 8 * 32]);
        glds16(gw[j] + kt * 32, &smem[st][1][(j * 4 + wave) * 8 * 32]);
      }
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    __syncthreads();            // previous tile's last reads are done before its stage is refilled
    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      const int st = kt & 1;
      if (kt + 1 < nk) issue(kt + 1, st ^ 1);
      const float* sA = smem[st][0] + arow * 32;
      const float* sB = smem[st][1] + brow * 32;
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        const float4 av = ld4(sA + (((kb * 2 + lh) ^ fa) << 2));
        const float4 bv = ld4(sB + (((kb * 2 + lh) ^ fb) << 2));
        acc = mfma32(av.x, bv.x, acc);
        acc = mfma32(av.y, bv.y, acc);
        acc = mfma32(av.z, bv.z, acc);
        acc = mfma32(av.w, bv.w, acc);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    const int col = n0 + wn * 32 + li;
    const float bv = bias ? bias[min(col, N - 1)] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (row < M && col < N) C[(int64_t)row * N + col] = acc[r] + bv;
    }
  }
}

static float *dA, *dW, *dC, *dC2;
template <typename F> static double time_us(F launch, int n = 50) {
  for (int i = 0; i < 5; ++i) launch();
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  for (int i = 0; i < n; ++i) launch();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3 / n;
}
static void prod(int M, int N, int K) {
  pn_gemm_desc d{};
  d.A = dA; d.lda = K; d.W = dW; d.ldw = K; d.C = dC2; d.ldc = N; d.M = M; d.N = N; d.K = K; d.batch = 1;
  pn_gemm_f32(&d, nullptr);
}
int main() {
  const size_t maxA = (size_t)66800 * 1024, maxW = (size_t)1024 * 1024, maxC = (size_t)66800 * 1024;
  hipMalloc(&dA, maxA * 4); hipMalloc(&dW, maxW * 4); hipMalloc(&dC, maxC * 4); hipMalloc(&dC2, maxC * 4);
  std::vector<float> h(maxA);
  srand(1);
  for (size_t i = 0; i < maxA; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
  hipMemcpy(dA, h.data(), maxA * 4, hipMemcpyHostToDevice);
  hipMemcpy(dW, h.data() + 12345, maxW * 4, hipMemcpyHostToDevice);
  const int shapes[][3] = {{21950, 1024, 256}, {21950, 256, 1024}, {21950, 544, 256}, {21950, 256, 256},
                           {66800, 256, 256}, {16700, 512, 128}, {4200, 1024, 256}};
  for (auto& s : shapes) {
    const int M = s[0], N = s[1], K = s[2];
    const int tiles = ((M + 63) / 64) * ((N + 63) / 64);
    printf("M=%d N=%d K=%d\n", M, N, K);
    double us = time_us([&] { prod(M, N, K); });
    printf("  %-34s %8.1f us %6.1f TF\n", "production (reg-staged, persistent)", us, 2.0 * M * N * K / us / 1e6);
    for (int per_cu : {4, 5}) {
      for (int persist : {1, 0}) {
        const int grid = persist ? (tiles < 256 * per_cu ? tiles : 256 * per_cu) : tiles;
        auto launch = [&] {
          if (per_cu == 4) hipLaunchKernelGGL(k_glds<4>, dim3(grid), dim3(256), 0, 0, dA, dW, (const float*)nullptr, dC, M, N, K);
          else hipLaunchKernelGGL(k_glds<5>, dim3(grid), dim3(256), 0, 0, dA, dW, (const float*)nullptr, dC, M, N, K);
        };
        us = time_us(launch);
        char name[64]; snprintf(name, 64, "glds 2-stage, %d WG/CU, %s", per_cu, persist ? "persistent" : "1 tile/WG");
        printf("  %-34s %8.1f us %6.1f TF\n", name, us, 2.0 * M * N * K / us / 1e6);
      }
    }
    // correctness vs production
    std::vector<float> c1((size_t)M * N), c2((size_t)M * N);
    hipMemcpy(c1.data(), dC, c1.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(c2.data(), dC2, c2.size() * 4, hipMemcpyDeviceToHost);
    double e = 0, mx = 0;
    for (size_t i = 0; i < c1.size(); ++i) { e = fmax(e, fabs((double)c1[i] - c2[i])); mx = fmax(mx, fabs((double)c2[i])); }
    printf("  max |glds - production| = %.3e (max |value| %.2f)\n", e, mx);
  }
  return 0;
}
