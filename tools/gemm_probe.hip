// Tuning aid (not part of the library): where does the fp32 MFMA GEMM lose time?
// A stripped copy of the row-major 64x64 tile of csrc/gemm.hip with phases that can be
// switched off at compile time, plus candidate structures, timed on the encoder shapes.
//   hipcc --offload-arch=gfx950 -O3 tools/gemm_probe.hip -o /tmp/gemm_probe && /tmp/gemm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "pairnet_hip.h"
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int xcd_tile_index(int L, int ntiles) {
  const int xcd = L & 7, j = L >> 3;
  const int q = ntiles >> 3, r = ntiles & 7;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + j;
}
__device__ unsigned long long g_clk[2];
enum { NO_GLOAD = 1, NO_MFMA = 2, NO_LDSST = 4, NO_EPI = 8, NO_LDSRD = 16 };

// ---- V0: the production structure (single LDS stage, register prefetch) ----
template <int BM, int BN, int WM, int WN, int FLAGS>
__global__ __launch_bounds__(256) void k_v0(const float* __restrict__ A, const float* __restrict__ W,
                                            float* __restrict__ C, int M, int N, int K) {
  constexpr int BK = 32, LD = BK + 4;
  constexpr int WAVES_N = BN / WN, TM = WM / 32, TN = WN / 32;
  constexpr int NA = (BM * BK / 4) / 256, NB = (BN * BK / 4) / 256;
  __shared__ __attribute__((aligned(16))) float smem[(BM + BN) * LD];
  float* sA = smem;
  float* sB = smem + BM * LD;
  const long long c0 = clock64(), w0 = wall_clock64();
  const int nt = (N + BN - 1) / BN, mt = (M + BM - 1) / BM;
  const int T = xcd_tile_index(blockIdx.x, nt * mt);
  const int tm = T / nt, tn = T - tm * nt;
  const int m0 = tm * BM, n0 = tn * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int kc = (tid & 7) * 4;
  const float* a_row[NA]; bool a_ok[NA];
  const float* w_row[NB]; bool w_ok[NB];
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    const int gm = m0 + (tid >> 3) + 32 * j;
    a_ok[j] = gm < M; a_row[j] = A + (int64_t)(a_ok[j] ? gm : 0) * K;
  }
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int gn = n0 + (tid >> 3) + 32 * j;
    w_ok[j] = gn < N; w_row[j] = W + (int64_t)(w_ok[j] ? gn : 0) * K;
  }
  float4 ra[NA], rb[NB];
  auto load_tile = [&](int kt) {
    const int k0 = kt * BK;
#pragma unroll
    for (int j = 0; j < NA; ++j) ra[j] = a_ok[j] ? ld4(a_row[j] + k0 + kc) : make_float4(0, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < NB; ++j) rb[j] = w_ok[j] ? ld4(w_row[j] + k0 + kc) : make_float4(0, 0, 0, 0);
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int j = 0; j < NA; ++j) st4(sA + ((tid >> 3) + 32 * j) * LD + kc, ra[j]);
#pragma unroll
    for (int j = 0; j < NB; ++j) st4(sB + ((tid >> 3) + 32 * j) * LD + kc, rb[j]);
  };
  f32x16 acc[TM][TN];
#pragma unroll
  for (int mi = 0; mi < TM; ++mi)
#pragma unroll
    for (int ni = 0; ni < TN; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
  float a[TM][4], b[TN][4];
#pragma unroll
  for (int mi = 0; mi < TM; ++mi) for (int t = 0; t < 4; ++t) a[mi][t] = 1.f + lane;
#pragma unroll
  for (int ni = 0; ni < TN; ++ni) for (int t = 0; t < 4; ++t) b[ni][t] = 0.5f;
  auto compute = [&]() {
#pragma unroll
    for (int kb = 0; kb < BK; kb += 8) {
      if (!(FLAGS & NO_LDSRD)) {
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) {
          const float4 v = ld4(sA + (wm * WM + mi * 32 + li) * LD + kb + 4 * lh);
          a[mi][0] = v.x; a[mi][1] = v.y; a[mi][2] = v.z; a[mi][3] = v.w;
        }
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
          const float4 v = ld4(sB + (wn * WN + ni * 32 + li) * LD + kb + 4 * lh);
          b[ni][0] = v.x; b[ni][1] = v.y; b[ni][2] = v.z; b[ni][3] = v.w;
        }
      }
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
          for (int ni = 0; ni < TN; ++ni) {
            if (FLAGS & NO_MFMA) acc[mi][ni][t] += a[mi][t] * b[ni][t];
            else acc[mi][ni] = mfma32(a[mi][t], b[ni][t], acc[mi][ni]);
          }
    }
  };
  const int nk = K / BK;
  load_tile(0);
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();
    if (!(FLAGS & NO_LDSST) || kt == 0) store_tile();
    __syncthreads();
    if (kt + 1 < nk && !(FLAGS & NO_GLOAD)) load_tile(kt + 1);
    compute();
  }
  if (FLAGS & NO_EPI) {
    float s = 0.f;
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
      for (int ni = 0; ni < TN; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[mi][ni][r];
    if (s == 123.456f) C[tid] = s;
    if (threadIdx.x == 0 && (blockIdx.x % 97) == 0) {
      atomicAdd(&g_clk[0], (unsigned long long)(clock64() - c0));
      atomicAdd(&g_clk[1], (unsigned long long)(wall_clock64() - w0));
    }
    return;
  }
#pragma unroll
  for (int ni = 0; ni < TN; ++ni) {
    const int col = n0 + wn * WN + ni * 32 + li;
    if (col >= N) continue;
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * WM + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row < M) C[(int64_t)row * N + col] = acc[mi][ni][r];
      }
  }
  if (threadIdx.x == 0 && (blockIdx.x % 97) == 0) {
    atomicAdd(&g_clk[0], (unsigned long long)(clock64() - c0));
    atomicAdd(&g_clk[1], (unsigned long long)(wall_clock64() - w0));
  }
}

static float* dA; static float* dW; static float* dC;

template <typename F> static double time_us(F launch, int n = 20) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) launch();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < n; ++i) launch();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return 1e3 * ms / n;
}

// ---- V1: persistent workgroups; the first chunk of the NEXT tile is prefetched while the
// last chunk of the current one is contracted, and the epilogue's stores drain under the
// next tile's MFMAs (no per-tile prologue / epilogue bubble).  BK = 32 | 64; DB = LDS
// double buffer with ONE barrier per chunk.
template <int BM, int BN, int WM, int WN, int BK, bool DB, int FLAGS>
__global__ __launch_bounds__(256) void k_v1(const float* __restrict__ A, const float* __restrict__ W,
                                            float* __restrict__ C, int M, int N, int K) {
  constexpr int LD = BK + 4;
  constexpr int WAVES_N = BN / WN, TM = WM / 32, TN = WN / 32;
  constexpr int TPR = BK / 4;            // threads per row
  constexpr int RPP = 256 / TPR;         // rows per pass
  constexpr int NA = BM / RPP, NB = BN / RPP;
  constexpr int STAGE = (BM + BN) * LD;
  __shared__ __attribute__((aligned(16))) float smem[(DB ? 2 : 1) * STAGE];
  const int nt = (N + BN - 1) / BN, mt = (M + BM - 1) / BM, ntiles = nt * mt;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per = gridDim.x >> 3;
  const int q = ntiles >> 3, rr = ntiles & 7;
  const int base = (xcd < rr) ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q;
  const int cnt = q + (xcd < rr ? 1 : 0);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int kc = (tid % TPR) * 4, lr = tid / TPR;
  const float* a_row[NA]; const float* w_row[NB];
  auto set_tile = [&](int t) {
    const int T = base + t;
    const int tm = T / nt, tn = T - tm * nt;
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const int gm = min(tm * BM + lr + RPP * j, M - 1);
      a_row[j] = A + (int64_t)gm * K + kc;
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int gn = min(tn * BN + lr + RPP * j, N - 1);
      w_row[j] = W + (int64_t)gn * K + kc;
    }
  };
  float4 ra[NA], rb[NB];
  auto load_tile = [&](int kt) {
#pragma unroll
    for (int j = 0; j < NA; ++j) ra[j] = ld4(a_row[j] + kt * BK);
#pragma unroll
    for (int j = 0; j < NB; ++j) rb[j] = ld4(w_row[j] + kt * BK);
  };
  auto store_tile = [&](int buf) {
    float* sA = smem + buf * STAGE;
    float* sB = sA + BM * LD;
#pragma unroll
    for (int j = 0; j < NA; ++j) st4(sA + (lr + RPP * j) * LD + kc, ra[j]);
#pragma unroll
    for (int j = 0; j < NB; ++j) st4(sB + (lr + RPP * j) * LD + kc, rb[j]);
  };
  f32x16 acc[TM][TN];
  auto zero = [&]() {
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
      for (int ni = 0; ni < TN; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
  };
  auto compute = [&](int buf) {
    const float* sA = smem + buf * STAGE;
    const float* sB = sA + BM * LD;
#pragma unroll
    for (int kb = 0; kb < BK; kb += 8) {
      float a[TM][4], b[TN][4];
#pragma unroll
      for (int mi = 0; mi < TM; ++mi) {
        const float4 v = ld4(sA + (wm * WM + mi * 32 + li) * LD + kb + 4 * lh);
        a[mi][0] = v.x; a[mi][1] = v.y; a[mi][2] = v.z; a[mi][3] = v.w;
      }
#pragma unroll
      for (int ni = 0; ni < TN; ++ni) {
        const float4 v = ld4(sB + (wn * WN + ni * 32 + li) * LD + kb + 4 * lh);
        b[ni][0] = v.x; b[ni][1] = v.y; b[ni][2] = v.z; b[ni][3] = v.w;
      }
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
          for (int ni = 0; ni < TN; ++ni) acc[mi][ni] = mfma32(a[mi][t], b[ni][t], acc[mi][ni]);
    }
  };
  const int nk = K / BK;
  if (slot >= cnt) return;
  set_tile(slot);
  load_tile(0);
  int it = 0;   // global chunk counter (LDS buffer parity)
  if (DB) { store_tile(0); __syncthreads(); }
  for (int t = slot; t < cnt; t += per) {
    const int T = base + t;
    const int tm = T / nt, tn = T - tm * nt;
    const int m0 = tm * BM, n0 = tn * BN;
    zero();
    for (int kt = 0; kt < nk; ++kt, ++it) {
      const bool more = (kt + 1 < nk) || (t + per < cnt);
      if (DB) {
        if (kt + 1 < nk) load_tile(kt + 1);
        else if (t + per < cnt) { set_tile(t + per); load_tile(0); }
        compute(it & 1);
        if (more) store_tile((it + 1) & 1);
        __syncthreads();
      } else {
        __syncthreads();
        store_tile(0);
        __syncthreads();
        if (kt + 1 < nk) load_tile(kt + 1);
        else if (t + per < cnt) { set_tile(t + per); load_tile(0); }
        compute(0);
      }
    }
    if (FLAGS & NO_EPI) continue;
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) {
      const int col = n0 + wn * WN + ni * 32 + li;
      if (col >= N) continue;
#pragma unroll
      for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + wm * WM + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (row < M) C[(int64_t)row * N + col] = acc[mi][ni][r];
        }
    }
  }
}
template <int BM, int BN, int WM, int WN, int BK, bool DB, int FLAGS>
static void run_v1(const char* name, int M, int N, int K, int occ) {
  const int grid = 256 * occ;
  double us = time_us([&] { hipLaunchKernelGGL((k_v1<BM, BN, WM, WN, BK, DB, FLAGS>), dim3(grid), dim3(256), 0, 0, dA, dW, dC, M, N, K); });
  printf("  %-30s occ %d %8.1f us %6.1f TF\n", name, occ, us, 2.0 * M * N * K / us / 1e6);
}

// ---- V2: V1 + MFMA operand fragments double-buffered in registers (the ds_reads of
// k-step kb+1 are issued before the MFMAs of kb), branch-free next-tile pointers.
template <int BM, int BN, int WM, int WN, int BK, bool DB, int PRIO>
__global__ __launch_bounds__(256) void k_v2(const float* __restrict__ A, const float* __restrict__ W,
                                            float* __restrict__ C, int M, int N, int K) {
  constexpr int LD = BK + 4;
  constexpr int WAVES_N = BN / WN, TM = WM / 32, TN = WN / 32;
  constexpr int TPR = BK / 4, RPP = 256 / TPR;
  constexpr int NA = BM / RPP, NB = BN / RPP;
  constexpr int STAGE = (BM + BN) * LD;
  constexpr int NKB = BK / 8;
  __shared__ __attribute__((aligned(16))) float smem[(DB ? 2 : 1) * STAGE];
  const int nt = (N + BN - 1) / BN, mt = (M + BM - 1) / BM, ntiles = nt * mt;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per = gridDim.x >> 3;
  const int q = ntiles >> 3, rr = ntiles & 7;
  const int base = (xcd < rr) ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q;
  const int cnt = q + (xcd < rr ? 1 : 0);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int kc = (tid % TPR) * 4, lr = tid / TPR;
  // per-lane row offsets (elements) of the tile being LOADED; the tile origin is added per tile
  int a_off[NA], w_off[NB];
  auto set_tile = [&](int t) {
    const int T = base + min(t, cnt - 1);
    const int tm = T / nt, tn = T - tm * nt;
#pragma unroll
    for (int j = 0; j < NA; ++j) a_off[j] = min(tm * BM + lr + RPP * j, M - 1) * K + kc;
#pragma unroll
    for (int j = 0; j < NB; ++j) w_off[j] = min(tn * BN + lr + RPP * j, N - 1) * K + kc;
  };
  float4 ra[NA], rb[NB];
  auto load_tile = [&](int k0) {
#pragma unroll
    for (int j = 0; j < NA; ++j) ra[j] = ld4(A + a_off[j] + k0);
#pragma unroll
    for (int j = 0; j < NB; ++j) rb[j] = ld4(W + w_off[j] + k0);
  };
  auto store_tile = [&](int buf) {
    float* sA = smem + buf * STAGE;
    float* sB = sA + BM * LD;
#pragma unroll
    for (int j = 0; j < NA; ++j) st4(sA + (lr + RPP * j) * LD + kc, ra[j]);
#pragma unroll
    for (int j = 0; j < NB; ++j) st4(sB + (lr + RPP * j) * LD + kc, rb[j]);
  };
  f32x16 acc[TM][TN];
  auto zero = [&]() {
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
      for (int ni = 0; ni < TN; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
  };
  auto compute = [&](int buf) {
    const float* sA = smem + buf * STAGE + (wm * WM + li) * LD + 4 * lh;
    const float* sB = smem + buf * STAGE + BM * LD + (wn * WN + li) * LD + 4 * lh;
    float4 fa[2][TM], fb[2][TN];
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) fa[0][mi] = ld4(sA + mi * 32 * LD);
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) fb[0][ni] = ld4(sB + ni * 32 * LD);
    if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      const int cur = kb & 1;
      if (kb + 1 < NKB) {
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) fa[cur ^ 1][mi] = ld4(sA + mi * 32 * LD + (kb + 1) * 8);
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) fb[cur ^ 1][ni] = ld4(sB + ni * 32 * LD + (kb + 1) * 8);
      }
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
          for (int ni = 0; ni < TN; ++ni) {
            const float av = t == 0 ? fa[cur][mi].x : t == 1 ? fa[cur][mi].y : t == 2 ? fa[cur][mi].z : fa[cur][mi].w;
            const float bv = t == 0 ? fb[cur][ni].x : t == 1 ? fb[cur][ni].y : t == 2 ? fb[cur][ni].z : fb[cur][ni].w;
            acc[mi][ni] = mfma32(av, bv, acc[mi][ni]);
          }
    }
    if (PRIO) __builtin_amdgcn_s_setprio(0);
  };
  const int nk = K / BK;
  if (slot >= cnt) return;
  set_tile(slot);
  load_tile(0);
  int it = 0;
  if (DB) { store_tile(0); __syncthreads(); }
  for (int t = slot; t < cnt; t += per) {
    const int T = base + t;
    const int tm = T / nt, tn = T - tm * nt;
    const int m0 = tm * BM, n0 = tn * BN;
    zero();
    for (int kt = 0; kt < nk; ++kt, ++it) {
      const bool last = kt + 1 == nk;
      if (DB) {
        if (last) set_tile(t + per);
        load_tile(last ? 0 : (kt + 1) * BK);
        compute(it & 1);
        store_tile((it + 1) & 1);
        __syncthreads();
      } else {
        __syncthreads();
        store_tile(0);
        __syncthreads();
        if (last) set_tile(t + per);
        load_tile(last ? 0 : (kt + 1) * BK);
        compute(0);
      }
    }
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) {
      const int col = n0 + wn * WN + ni * 32 + li;
      if (col >= N) continue;
#pragma unroll
      for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + wm * WM + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (row < M) C[(int64_t)row * N + col] = acc[mi][ni][r];
        }
    }
  }
}
template <int BM, int BN, int WM, int WN, int BK, bool DB, int PRIO>
static void run_v2(const char* name, int M, int N, int K, int occ) {
  const int grid = 256 * occ;
  double us = time_us([&] { hipLaunchKernelGGL((k_v2<BM, BN, WM, WN, BK, DB, PRIO>), dim3(grid), dim3(256), 0, 0, dA, dW, dC, M, N, K); });
  printf("  %-30s occ %d %8.1f us %6.1f TF\n", name, occ, us, 2.0 * M * N * K / us / 1e6);
}

template <int BM, int BN, int WM, int WN, int FLAGS>
static void run_v0(const char* name, int M, int N, int K) {
  const int grid = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  unsigned long long z[2] = {0, 0};
  hipMemcpyToSymbol(HIP_SYMBOL(g_clk), z, sizeof(z));
  double us = time_us([&] { hipLaunchKernelGGL((k_v0<BM, BN, WM, WN, FLAGS>), dim3(grid), dim3(256), 0, 0, dA, dW, dC, M, N, K); });
  hipMemcpyFromSymbol(z, HIP_SYMBOL(g_clk), sizeof(z));
  printf("  %-34s %8.1f us %6.1f TF  shader clk %.0f MHz\n", name, us, 2.0 * M * N * K / us / 1e6,
         z[1] ? 100.0 * (double)z[0] / (double)z[1] : 0.0);
}
static void run_prod(const char* name, int M, int N, int K, int flags, bool bias, bool res) {
  pn_gemm_desc d{};
  d.A = dA; d.lda = K; d.W = dW; d.ldw = K; d.C = dC; d.ldc = N; d.M = M; d.N = N; d.K = K;
  d.batch = 1; d.flags = flags;
  if (bias) d.bias = dW;
  if (res) { d.Res = dA; d.ldres = N; }
  double us = time_us([&] { pn_gemm_f32(&d, nullptr); });
  printf("  %-36s %8.1f us %6.1f TF\n", name, us, 2.0 * M * N * K / us / 1e6);
}
static void run_conv(const char* name, int flags) {
  const int H = 200, Wd = 334, Cc = 256;
  double us = time_us([&] { pn_conv2d_nhwc_f32(dA, dW, nullptr, dC, 1, H, Wd, Cc, 256, 3, 3, 1, 0, flags, nullptr); }, 5);
  printf("  %-36s %8.1f us %6.1f TF\n", name, us, 2.0 * H * Wd * 256 * 9.0 * Cc / us / 1e6);
}
int main() {
  const size_t maxA = (size_t)66800 * 1024, maxW = (size_t)1024 * 1024, maxC = (size_t)66800 * 1024;
  hipMalloc(&dA, maxA * 4); hipMalloc(&dW, maxW * 4); hipMalloc(&dC, maxC * 4);
  std::vector<float> h(maxA);
  for (size_t i = 0; i < maxA; ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
  hipMemcpy(dA, h.data(), maxA * 4, hipMemcpyHostToDevice);
  hipMemcpy(dW, h.data(), maxW * 4, hipMemcpyHostToDevice);
  const int shapes[][3] = {{21950, 1024, 256}, {21950, 256, 1024}, {21950, 544, 256}, {66800, 256, 256}, {21950, 256, 256}};
  for (auto& s : shapes) {
    const int M = s[0], N = s[1], K = s[2];
    printf("M=%d N=%d K=%d\n", M, N, K);
    run_v0<64, 64, 32, 32, 0>("v0 64x64 full", M, N, K);
    run_v0<128, 64, 64, 32, 0>("v0 128x64 full", M, N, K);
    run_prod("prod default", M, N, K, 0, false, false);
    run_prod("prod default +bias", M, N, K, 0, true, false);
    run_prod("prod default +bias+res", M, N, K, 0, true, true);
    run_prod("prod 128x64", M, N, K, 32, false, false);
    for (int occ : {4}) {
      run_v2<64, 64, 32, 32, 32, false, 0>("v2 64x64 bk32 single", M, N, K, occ);
      run_v2<64, 64, 32, 32, 32, false, 1>("v2 64x64 bk32 single prio", M, N, K, occ);
      run_v2<64, 64, 32, 32, 64, false, 0>("v2 64x64 bk64 single", M, N, K, occ);
      run_v2<64, 64, 32, 32, 32, true, 0>("v2 64x64 bk32 double", M, N, K, occ);
      run_v2<64, 64, 32, 32, 64, true, 0>("v2 64x64 bk64 double", M, N, K, occ);
      run_v2<128, 64, 64, 32, 32, false, 0>("v2 128x64 bk32 single", M, N, K, occ);
      run_v2<128, 64, 64, 32, 64, false, 0>("v2 128x64 bk64 single", M, N, K, occ);
      run_v2<128, 128, 64, 64, 32, false, 0>("v2 128x128 bk32 single", M, N, K, occ);
    }
  }
  run_conv("conv3x3 64x64 (default)", 0);
  run_conv("conv3x3 128x64", 32);
  run_conv("conv3x3 128x128", 4);
  return 0;
}
