// Tuning aid (not part of the library): fp32 GEMM through the bf16 matrix pipe.
//
// Every fp32 number is EXACTLY the sum of three bf16 numbers (x = x0 + x1 + x2, 3 x 8
// significand bits, each piece the round-to-nearest bf16 of what is left), and a product of two
// bf16 numbers is exact in fp32.  So a.b = sum over the nine piece products, all exact; the
// three smallest (a1 b2, a2 b1, a2 b2) are together below 2^-26 |a b| -- a quarter of the
// rounding unit of ONE fp32 product -- and are dropped: six bf16 MFMAs (fp32 accumulate) per
// fp32 MFMA's work, at 16 x the rate.  Peak 2.5 PFLOP/s / 6 = 417 TFLOP/s of fp32-equivalent
// work against 157.3 for v_mfma_f32_32x32x2_f32.
//
// The probe: C = act(A W^T + bias), A [M][K] fp32 split on the fly while staging to LDS, W
// [N][K] fp32 pre-split once into three bf16 planes; timed against pn_gemm_f32 on the encoder
// shapes; both checked against an fp64 dot product on a sample of outputs.
//   sh tools/build_probe.sh && tools/bin/gemm_split_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "pairnet_hip.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ void split8(const f32x8 x, bf16x8& p0, bf16x8& p1, bf16x8& p2) {
  p0 = __builtin_convertvector(x, bf16x8);
  const f32x8 r1 = x - __builtin_convertvector(p0, f32x8);
  p1 = __builtin_convertvector(r1, bf16x8);
  const f32x8 r2 = r1 - __builtin_convertvector(p1, f32x8);
  p2 = __builtin_convertvector(r2, bf16x8);
}

// W [N][K] fp32 -> Ws [3][N][K] bf16
__global__ void k_split_w(const float* __restrict__ W, __bf16* __restrict__ Ws, long long n8, long long plane) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  const f32x8 x = *reinterpret_cast<const f32x8*>(W + i * 8);
  bf16x8 p0, p1, p2;
  split8(x, p0, p1, p2);
  *reinterpret_cast<bf16x8*>(Ws + i * 8) = p0;
  *reinterpret_cast<bf16x8*>(Ws + plane + i * 8) = p1;
  *reinterpret_cast<bf16x8*>(Ws + 2 * plane + i * 8) = p2;
}

__device__ __forceinline__ f32x16 mfma(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// 128 x 128 tile, 4 waves (2 x 2), each 64 x 64 = 2 x 2 accumulators of 32 x 32; BK = 32.
// LDS: [piece 3][row 128][40 bf16] for A and for W (80-byte rows: conflict-free b128 reads).
constexpr int BM = 128, BN = 128, BK = 32, LDK = 40;
constexpr int PIECE = BM * LDK;             // bf16 elements per piece plane in LDS

template <int TERMS>
__global__ __launch_bounds__(256, 2) void k_gemm_split(
    const float* __restrict__ A, const __bf16* __restrict__ Ws, const float* __restrict__ bias,
    float* __restrict__ C, int M, int N, int K, long long wplane, int relu) {
  __shared__ __attribute__((aligned(16))) __bf16 sA[3 * PIECE];
  __shared__ __attribute__((aligned(16))) __bf16 sW[3 * PIECE];
  const int nt = (N + BN - 1) / BN;
  const int tm = blockIdx.x / nt, tn = blockIdx.x - tm * nt;
  const int m0 = tm * BM, n0 = tn * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int lr = tid >> 2, lc = (tid & 3) * 8;     // staging: row within 64, k offset
  const float* a_ptr[2];
  const __bf16* w_ptr[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    int gm = m0 + lr + 64 * p; gm = gm < M ? gm : M - 1;
    int gn = n0 + lr + 64 * p; gn = gn < N ? gn : N - 1;
    a_ptr[p] = A + (long long)gm * K + lc;
    w_ptr[p] = Ws + (long long)gn * K + lc;
  }
  f32x8 ra[2];
  bf16x8 rw[2][3];
  auto gload = [&](int kt) {
    const int k0 = kt * BK;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      ra[p] = *reinterpret_cast<const f32x8*>(a_ptr[p] + k0);
#pragma unroll
      for (int q = 0; q < 3; ++q)
        rw[p][q] = *reinterpret_cast<const bf16x8*>(w_ptr[p] + q * wplane + k0);
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      bf16x8 p0, p1, p2;
      split8(ra[p], p0, p1, p2);
      const int off = (lr + 64 * p) * LDK + lc;
      *reinterpret_cast<bf16x8*>(sA + off) = p0;
      *reinterpret_cast<bf16x8*>(sA + PIECE + off) = p1;
      *reinterpret_cast<bf16x8*>(sA + 2 * PIECE + off) = p2;
#pragma unroll
      for (int q = 0; q < 3; ++q) *reinterpret_cast<bf16x8*>(sW + q * PIECE + off) = rw[p][q];
    }
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int nk = K / BK;
  const int fa = (wm * 64 + (lane & 31)) * LDK + (lane >> 5) * 8;
  const int fb = (wn * 64 + (lane & 31)) * LDK + (lane >> 5) * 8;
  gload(0);
  lstore();
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) gload(kt + 1);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 a[3][2], b[3][2];
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          a[q][i] = *reinterpret_cast<const bf16x8*>(sA + q * PIECE + fa + i * 32 * LDK + s * 16);
          b[q][i] = *reinterpret_cast<const bf16x8*>(sW + q * PIECE + fb + i * 32 * LDK + s * 16);
        }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          f32x16 c = acc[i][j];
          if (TERMS >= 6) {
            c = mfma(a[2][i], b[0][j], c);
            c = mfma(a[1][i], b[1][j], c);
            c = mfma(a[0][i], b[2][j], c);
          }
          if (TERMS >= 3) {
            c = mfma(a[1][i], b[0][j], c);
            c = mfma(a[0][i], b[1][j], c);
          }
          c = mfma(a[0][i], b[0][j], c);
          acc[i][j] = c;
        }
    }
    __syncthreads();
    if (kt + 1 < nk) {
      lstore();
      __syncthreads();
    }
  }
  // epilogue: C/D layout col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int gn = n0 + wn * 64 + j * 32 + (lane & 31);
    if (gn >= N) continue;
    const float bv = bias ? bias[gn] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int gm = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (gm < M) {
          float v = acc[i][j][r] + bv;
          if (relu) v = v > 0.f ? v : 0.f;
          C[(long long)gm * N + gn] = v;
        }
      }
  }
}


// ---- v2: persistent, producer / consumer waves, double-buffered LDS -----------------------
// 512 threads: waves 0-3 contract (2 x 2, 64 x 64 each), waves 4-7 stage (global -> registers
// two k-steps ahead -> split -> LDS one step ahead).  One barrier per k-step; the steps of all
// the workgroup's tiles form ONE stream, so the next tile's first loads are in flight while the
// previous tile's accumulators are written.
constexpr int STAGE = 6 * PIECE;            // bf16 elements per LDS stage (A pieces | W pieces)

__device__ __forceinline__ int xcd_range(int ntiles, int xcd, int& cnt) {
  const int q = ntiles >> 3, r = ntiles & 7;
  cnt = q + (xcd < r ? 1 : 0);
  return xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
}

struct StageRegs { f32x8 a[2]; bf16x8 w[2][3]; };

template <int TERMS>
__global__ __launch_bounds__(512, 1) void k_gemm_split2(
    const float* __restrict__ A, const __bf16* __restrict__ Ws, const float* __restrict__ bias,
    float* __restrict__ C, int M, int N, int K, long long wplane, int relu) {
  extern __shared__ __attribute__((aligned(16))) __bf16 smem[];
  const int nt = (N + BN - 1) / BN, mt = (M + BM - 1) / BM;
  const int nk = K / BK;
  const int L = blockIdx.x, per = gridDim.x >> 3, jx = L >> 3;
  int cnt;
  const int base = xcd_range(nt * mt, L & 7, cnt);
  const int mine = jx < cnt ? (cnt - jx + per - 1) / per : 0;
  const int total = mine * nk;
  if (total == 0) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool producer = wave >= 4;
  // ---- producer state ----
  const int pt = tid & 255, lr = pt >> 2, lc = (pt & 3) * 8;
  auto gload = [&](StageRegs& R, int g) {
    g = g < total ? g : total - 1;
    const int ti = g / nk, kt = g - ti * nk;
    const int T = base + jx + ti * per;
    const int tm = T / nt, tn = T - tm * nt;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      int gm = tm * BM + lr + 64 * p; gm = gm < M ? gm : M - 1;
      int gn = tn * BN + lr + 64 * p; gn = gn < N ? gn : N - 1;
      R.a[p] = *reinterpret_cast<const f32x8*>(A + (long long)gm * K + kt * BK + lc);
      const __bf16* wp = Ws + (long long)gn * K + kt * BK + lc;
#pragma unroll
      for (int q = 0; q < 3; ++q) R.w[p][q] = *reinterpret_cast<const bf16x8*>(wp + q * wplane);
    }
  };
  auto lstore = [&](const StageRegs& R, int stage) {
    __bf16* sA = smem + stage * STAGE;
    __bf16* sW = sA + 3 * PIECE;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      bf16x8 p0, p1, p2;
      split8(R.a[p], p0, p1, p2);
      const int off = (lr + 64 * p) * LDK + lc;
      *reinterpret_cast<bf16x8*>(sA + off) = p0;
      *reinterpret_cast<bf16x8*>(sA + PIECE + off) = p1;
      *reinterpret_cast<bf16x8*>(sA + 2 * PIECE + off) = p2;
#pragma unroll
      for (int q = 0; q < 3; ++q) *reinterpret_cast<bf16x8*>(sW + q * PIECE + off) = R.w[p][q];
    }
  };
  // ---- consumer state ----
  const int wm = (wave & 3) >> 1, wn = wave & 1;
  const int fa = (wm * 64 + (lane & 31)) * LDK + (lane >> 5) * 8;
  const int fb = (wn * 64 + (lane & 31)) * LDK + (lane >> 5) * 8;
  f32x16 acc[2][2];
  auto zero = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  };
  auto epilogue = [&](int ti) {
    const int T = base + jx + ti * per;
    const int tm = T / nt, tn = T - tm * nt;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int gn = tn * BN + wn * 64 + j * 32 + (lane & 31);
      if (gn >= N) continue;
      const float bv = bias ? bias[gn] : 0.f;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int gm = tm * BM + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          if (gm < M) {
            float v = acc[i][j][r] + bv;
            if (relu) v = v > 0.f ? v : 0.f;
            C[(long long)gm * N + gn] = v;
          }
        }
    }
  };
  auto compute = [&](int stage) {
    const __bf16* sA = smem + stage * STAGE;
    const __bf16* sW = sA + 3 * PIECE;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 a[3][2], b[3][2];
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          a[q][i] = *reinterpret_cast<const bf16x8*>(sA + q * PIECE + fa + i * 32 * LDK + s * 16);
          b[q][i] = *reinterpret_cast<const bf16x8*>(sW + q * PIECE + fb + i * 32 * LDK + s * 16);
        }
      // small terms first; consecutive MFMAs go to different accumulators
#define TERM(qa, qb)                                                                   \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j) \
      acc[i][j] = mfma(a[qa][i], b[qb][j], acc[i][j]);
      if (TERMS >= 6) { TERM(2, 0) TERM(1, 1) TERM(0, 2) }
      if (TERMS >= 3) { TERM(1, 0) TERM(0, 1) }
      TERM(0, 0)
#undef TERM
    }
  };
  int kt = 0, ti = 0;
  auto step = [&](int stage) {
    compute(stage);
    if (++kt == nk) { epilogue(ti); zero(); kt = 0; ++ti; }
  };
  StageRegs R0, R1;
  zero();
  if (producer) { gload(R0, 0); gload(R1, 1); lstore(R0, 0); gload(R0, 2); }
  __syncthreads();
  for (int g = 0;; g += 2) {
    if (producer) { lstore(R1, 1); gload(R1, g + 3); } else step(0);
    __syncthreads();
    if (g + 1 >= total) break;
    if (producer) { lstore(R0, 0); gload(R0, g + 4); } else step(1);
    __syncthreads();
    if (g + 2 >= total) break;
  }
}


// ---- v3: as v2, both operands fp32 split on the fly by the staging waves, loads issued as
// inline asm with hand-counted s_waitcnt (the compiler's loop model drained the queue every
// step), register ring D steps deep in front of the two LDS stages ---------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));
struct Ring { f32x4 v[8]; };     // A rows lr, lr + 64 (2 x 8 floats), W rows likewise

__device__ __forceinline__ void gl16(f32x4& d, const float* p) {
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(d) : "v"(p) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_ring(Ring& R) {
  asm volatile("s_waitcnt vmcnt(%8)"
               : "+v"(R.v[0]), "+v"(R.v[1]), "+v"(R.v[2]), "+v"(R.v[3]), "+v"(R.v[4]), "+v"(R.v[5]),
                 "+v"(R.v[6]), "+v"(R.v[7])
               : "n"(N)
               : "memory");
}
__device__ __forceinline__ void split8v(const f32x4 lo, const f32x4 hi, bf16x8& p0, bf16x8& p1, bf16x8& p2) {
  f32x8 x;
  x[0] = lo[0]; x[1] = lo[1]; x[2] = lo[2]; x[3] = lo[3];
  x[4] = hi[0]; x[5] = hi[1]; x[6] = hi[2]; x[7] = hi[3];
  split8(x, p0, p1, p2);
}

template <int TERMS, int D>
__global__ __launch_bounds__(512, 1) void k_gemm_split3(
    const float* __restrict__ A, const float* __restrict__ W, const float* __restrict__ bias,
    float* __restrict__ C, int M, int N, int K, int relu) {
  extern __shared__ __attribute__((aligned(16))) __bf16 smem[];
  const int nt = (N + BN - 1) / BN, mt = (M + BM - 1) / BM;
  const int nk = K / BK;
  const int L = blockIdx.x, per = gridDim.x >> 3, jx = L >> 3;
  int cnt;
  const int base = xcd_range(nt * mt, L & 7, cnt);
  const int mine = jx < cnt ? (cnt - jx + per - 1) / per : 0;
  const int total = mine * nk;
  if (total == 0) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool producer = wave >= 4;
  const int pt = tid & 255, lr = pt >> 2, lc = (pt & 3) * 8;
  auto gload = [&](Ring& R, int g) {
    g = g < total ? g : total - 1;
    const int ti = g / nk, kt = g - ti * nk;
    const int T = base + jx + ti * per;
    const int tm = T / nt, tn = T - tm * nt;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      int gm = tm * BM + lr + 64 * p; gm = gm < M ? gm : M - 1;
      int gn = tn * BN + lr + 64 * p; gn = gn < N ? gn : N - 1;
      const float* ap = A + (long long)gm * K + kt * BK + lc;
      const float* wp = W + (long long)gn * K + kt * BK + lc;
      gl16(R.v[2 * p], ap); gl16(R.v[2 * p + 1], ap + 4);
      gl16(R.v[4 + 2 * p], wp); gl16(R.v[4 + 2 * p + 1], wp + 4);
    }
  };
  auto lstore = [&](const Ring& R, int stage) {
    __bf16* sA = smem + stage * STAGE;
    __bf16* sW = sA + 3 * PIECE;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      bf16x8 p0, p1, p2;
      const int off = (lr + 64 * p) * LDK + lc;
      split8v(R.v[2 * p], R.v[2 * p + 1], p0, p1, p2);
      *reinterpret_cast<bf16x8*>(sA + off) = p0;
      *reinterpret_cast<bf16x8*>(sA + PIECE + off) = p1;
      *reinterpret_cast<bf16x8*>(sA + 2 * PIECE + off) = p2;
      split8v(R.v[4 + 2 * p], R.v[4 + 2 * p + 1], p0, p1, p2);
      *reinterpret_cast<bf16x8*>(sW + off) = p0;
      *reinterpret_cast<bf16x8*>(sW + PIECE + off) = p1;
      *reinterpret_cast<bf16x8*>(sW + 2 * PIECE + off) = p2;
    }
  };
  const int wm = (wave & 3) >> 1, wn = wave & 1;
  const int fa = (wm * 64 + (lane & 31)) * LDK + (lane >> 5) * 8;
  const int fb = (wn * 64 + (lane & 31)) * LDK + (lane >> 5) * 8;
  f32x16 acc[2][2];
  auto zero = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  };
  auto epilogue = [&](int ti) {
    const int T = base + jx + ti * per;
    const int tm = T / nt, tn = T - tm * nt;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int gn = tn * BN + wn * 64 + j * 32 + (lane & 31);
      if (gn >= N) continue;
      const float bv = bias ? bias[gn] : 0.f;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int gm = tm * BM + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          if (gm < M) {
            float v = acc[i][j][r] + bv;
            if (relu) v = v > 0.f ? v : 0.f;
            C[(long long)gm * N + gn] = v;
          }
        }
    }
  };
  auto compute = [&](int stage) {
    const __bf16* sA = smem + stage * STAGE;
    const __bf16* sW = sA + 3 * PIECE;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 a[3][2], b[3][2];
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          a[q][i] = *reinterpret_cast<const bf16x8*>(sA + q * PIECE + fa + i * 32 * LDK + s * 16);
          b[q][i] = *reinterpret_cast<const bf16x8*>(sW + q * PIECE + fb + i * 32 * LDK + s * 16);
        }
#define TERM(qa, qb)                                                                   \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j) \
      acc[i][j] = mfma(a[qa][i], b[qb][j], acc[i][j]);
      if (TERMS >= 6) { TERM(2, 0) TERM(1, 1) TERM(0, 2) }
      if (TERMS >= 3) { TERM(1, 0) TERM(0, 1) }
      TERM(0, 0)
#undef TERM
    }
  };
  int kt = 0, ti = 0;
  Ring R[D];
  zero();
  if (producer) {
#pragma unroll
    for (int d = 0; d < D; ++d) gload(R[d], d);
    wait_ring<(D - 1) * 8>(R[0]);
    lstore(R[0], 0);
    gload(R[0], D);
  }
  __syncthreads();
  constexpr int U = (D % 2 == 0) ? D : 2 * D;
  bool done = false;
  for (int g = 0; !done; g += U) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (producer) {
        Ring& S = R[(u + 1) % D];
        wait_ring<(D - 1) * 8>(S);
        lstore(S, (u + 1) & 1);
        gload(S, g + u + 1 + D);
      } else {
        compute(u & 1);
        if (++kt == nk) { epilogue(ti); zero(); kt = 0; ++ti; }
      }
      __syncthreads();
      if (g + u + 1 >= total) { done = true; break; }
    }
  }
  if (producer) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}


// ---- ablations of v3: FLAGS 1 = no global loads after the prologue, 2 = no split arithmetic,
// 4 = no LDS writes, 8 = no LDS reads / MFMAs in the contracting waves (timing only) ----------
template <int TERMS, int D, int FLAGS>
__global__ __launch_bounds__(512, 1) void k_gemm_ablate(
    const float* __restrict__ A, const float* __restrict__ W, const float* __restrict__ bias,
    float* __restrict__ C, int M, int N, int K, int relu) {
  extern __shared__ __attribute__((aligned(16))) __bf16 smem[];
  const int nt = (N + BN - 1) / BN, mt = (M + BM - 1) / BM;
  const int nk = K / BK;
  const int L = blockIdx.x, per = gridDim.x >> 3, jx = L >> 3;
  int cnt;
  const int base = xcd_range(nt * mt, L & 7, cnt);
  const int mine = jx < cnt ? (cnt - jx + per - 1) / per : 0;
  const int total = mine * nk;
  if (total == 0) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool producer = wave >= 4;
  const int pt = tid & 255, lr = pt >> 2, lc = (pt & 3) * 8;
  auto gload = [&](Ring& R, int g) {
    g = g < total ? g : total - 1;
    const int ti = g / nk, kt = g - ti * nk;
    const int T = base + jx + ti * per;
    const int tm = T / nt, tn = T - tm * nt;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      int gm = tm * BM + lr + 64 * p; gm = gm < M ? gm : M - 1;
      int gn = tn * BN + lr + 64 * p; gn = gn < N ? gn : N - 1;
      const float* ap = A + (long long)gm * K + kt * BK + lc;
      const float* wp = W + (long long)gn * K + kt * BK + lc;
      gl16(R.v[2 * p], ap); gl16(R.v[2 * p + 1], ap + 4);
      gl16(R.v[4 + 2 * p], wp); gl16(R.v[4 + 2 * p + 1], wp + 4);
    }
  };
  auto lstore = [&](const Ring& R, int stage) {
    __bf16* sA = smem + stage * STAGE;
    __bf16* sW = sA + 3 * PIECE;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      bf16x8 p0, p1, p2;
      const int off = (lr + 64 * p) * LDK + lc;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if (FLAGS & 2) {     // no split arithmetic: the raw words
          union { f32x4 f; bf16x8 b; } u0, u1;
          u0.f = R.v[4 * h + 2 * p]; u1.f = R.v[4 * h + 2 * p + 1];
          p0 = u0.b; p1 = u1.b; p2 = u0.b;
        } else {
          split8v(R.v[4 * h + 2 * p], R.v[4 * h + 2 * p + 1], p0, p1, p2);
        }
        __bf16* dst = h ? sW : sA;
        if (FLAGS & 4) {     // no LDS writes (the values stay alive)
          asm volatile("" :: "v"(p0), "v"(p1), "v"(p2));
        } else {
          *reinterpret_cast<bf16x8*>(dst + off) = p0;
          *reinterpret_cast<bf16x8*>(dst + PIECE + off) = p1;
          *reinterpret_cast<bf16x8*>(dst + 2 * PIECE + off) = p2;
        }
      }
    }
  };
  const int wm = (wave & 3) >> 1, wn = wave & 1;
  const int fa = (wm * 64 + (lane & 31)) * LDK + (lane >> 5) * 8;
  const int fb = (wn * 64 + (lane & 31)) * LDK + (lane >> 5) * 8;
  f32x16 acc[2][2];
  auto zero = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  };
  auto epilogue = [&](int ti) {
    const int T = base + jx + ti * per;
    const int tm = T / nt, tn = T - tm * nt;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int gn = tn * BN + wn * 64 + j * 32 + (lane & 31);
      if (gn >= N) continue;
      const float bv = bias ? bias[gn] : 0.f;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int gm = tm * BM + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          if (gm < M) {
            float v = acc[i][j][r] + bv;
            if (relu) v = v > 0.f ? v : 0.f;
            C[(long long)gm * N + gn] = v;
          }
        }
    }
  };
  auto compute = [&](int stage) {
    const __bf16* sA = smem + stage * STAGE;
    const __bf16* sW = sA + 3 * PIECE;
    bf16x8 a[2][3][2], b[2][3][2];
    // FLAGS & 16: all 24 fragment reads of the k-step first, then its 48 MFMAs (one exposed LDS
    // latency per step instead of one per 16-deep half)
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          if (FLAGS & 64) {   // MFMAs only: fragments made up in registers
            union { bf16x8 b; unsigned u[4]; } z;
            z.u[0] = z.u[1] = z.u[2] = z.u[3] = (FLAGS & 128) ? 0u : (FLAGS & 256) ? (0x3c003c00u ^ (lane * 0x9e3779b1u + (s * 6 + q * 2 + i) * 0x85ebca6bu)) & 0xbfffbfffu : 0x3c003c00u + lane + s + q + i;
            a[s][q][i] = z.b; b[s][q][i] = z.b;
            asm volatile("" : "+v"(a[s][q][i]), "+v"(b[s][q][i]));
          } else {
          a[s][q][i] = *reinterpret_cast<const bf16x8*>(sA + q * PIECE + fa + i * 32 * LDK + s * 16);
          b[s][q][i] = *reinterpret_cast<const bf16x8*>(sW + q * PIECE + fb + i * 32 * LDK + s * 16);
          }
        }
    if (FLAGS & 16) __builtin_amdgcn_sched_barrier(0);
    if (FLAGS & 32) {      // fragment reads only: keep them alive, no MFMA
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
          for (int i = 0; i < 2; ++i) asm volatile("" :: "v"(a[s][q][i]), "v"(b[s][q][i]));
      return;
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#define TERM(qa, qb)                                                                   \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j) \
      acc[i][j] = mfma(a[s][qa][i], b[s][qb][j], acc[i][j]);
      if (TERMS >= 6) { TERM(2, 0) TERM(1, 1) TERM(0, 2) }
      if (TERMS >= 3) { TERM(1, 0) TERM(0, 1) }
      TERM(0, 0)
#undef TERM
    }
  };
  int kt = 0, ti = 0;
  Ring R[D];
  zero();
  if (producer) {
#pragma unroll
    for (int d = 0; d < D; ++d) gload(R[d], d);
    wait_ring<(D - 1) * 8>(R[0]);
    lstore(R[0], 0);
    gload(R[0], D);
  }
  __syncthreads();
  constexpr int U = (D % 2 == 0) ? D : 2 * D;
  bool done = false;
  for (int g = 0; !done; g += U) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (producer) {
        Ring& S = R[(u + 1) % D];
        wait_ring<(D - 1) * 8>(S);
        lstore(S, (u + 1) & 1);
        if (!(FLAGS & 1)) gload(S, g + u + 1 + D);
      } else {
        if (!(FLAGS & 8)) compute(u & 1);
        if (++kt == nk) { epilogue(ti); zero(); kt = 0; ++ti; }
      }
      __syncthreads();
      if (g + u + 1 >= total) { done = true; break; }
    }
  }
  if (producer) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}



// ---- v5: v3 with the staging lanes laid along the rows (8 lanes x 16 B = one whole 128-byte line
// per row and instruction, as k_gemm_tile stages) instead of 4 lanes x 2 instructions ---------
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split4(const f32x4 x, bf16x4& p0, bf16x4& p1, bf16x4& p2) {
  p0 = __builtin_convertvector(x, bf16x4);
  const f32x4 r1 = x - __builtin_convertvector(p0, f32x4);
  p1 = __builtin_convertvector(r1, bf16x4);
  const f32x4 r2 = r1 - __builtin_convertvector(p1, f32x4);
  p2 = __builtin_convertvector(r2, bf16x4);
}
template <int TERMS, int D>
__global__ __launch_bounds__(512, 1) void k_gemm_split5(
    const float* __restrict__ A, const float* __restrict__ W, const float* __restrict__ bias,
    float* __restrict__ C, int M, int N, int K, int relu) {
  extern __shared__ __attribute__((aligned(16))) __bf16 smem[];
  const int nt = (N + BN - 1) / BN, mt = (M + BM - 1) / BM;
  const int nk = K / BK;
  const int L = blockIdx.x, per = gridDim.x >> 3, jx = L >> 3;
  int cnt;
  const int base = xcd_range(nt * mt, L & 7, cnt);
  const int mine = jx < cnt ? (cnt - jx + per - 1) / per : 0;
  const int total = mine * nk;
  if (total == 0) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool producer = wave >= 4;
  const int pt = tid & 255, lr = pt >> 3, lc = (pt & 7) * 4;   // 8 lanes x 16 B = one 128-byte line per row
  auto gload = [&](Ring& R, int g) {
    g = g < total ? g : total - 1;
    const int ti = g / nk, kt = g - ti * nk;
    const int T = base + jx + ti * per;
    const int tm = T / nt, tn = T - tm * nt;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      int gm = tm * BM + lr + 32 * p; gm = gm < M ? gm : M - 1;
      int gn = tn * BN + lr + 32 * p; gn = gn < N ? gn : N - 1;
      gl16(R.v[p], A + (long long)gm * K + kt * BK + lc);
      gl16(R.v[4 + p], W + (long long)gn * K + kt * BK + lc);
    }
  };
  auto lstore = [&](const Ring& R, int stage) {
    __bf16* sA = smem + stage * STAGE;
    __bf16* sW = sA + 3 * PIECE;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      bf16x4 p0, p1, p2;
      const int off = (lr + 32 * p) * LDK + lc;
      split4(R.v[p], p0, p1, p2);
      *reinterpret_cast<bf16x4*>(sA + off) = p0;
      *reinterpret_cast<bf16x4*>(sA + PIECE + off) = p1;
      *reinterpret_cast<bf16x4*>(sA + 2 * PIECE + off) = p2;
      split4(R.v[4 + p], p0, p1, p2);
      *reinterpret_cast<bf16x4*>(sW + off) = p0;
      *reinterpret_cast<bf16x4*>(sW + PIECE + off) = p1;
      *reinterpret_cast<bf16x4*>(sW + 2 * PIECE + off) = p2;
    }
  };
  const int wm = (wave & 3) >> 1, wn = wave & 1;
  const int fa = (wm * 64 + (lane & 31)) * LDK + (lane >> 5) * 8;
  const int fb = (wn * 64 + (lane & 31)) * LDK + (lane >> 5) * 8;
  f32x16 acc[2][2];
  auto zero = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  };
  auto epilogue = [&](int ti) {
    const int T = base + jx + ti * per;
    const int tm = T / nt, tn = T - tm * nt;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int gn = tn * BN + wn * 64 + j * 32 + (lane & 31);
      if (gn >= N) continue;
      const float bv = bias ? bias[gn] : 0.f;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int gm = tm * BM + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          if (gm < M) {
            float v = acc[i][j][r] + bv;
            if (relu) v = v > 0.f ? v : 0.f;
            C[(long long)gm * N + gn] = v;
          }
        }
    }
  };
  auto compute = [&](int stage) {
    const __bf16* sA = smem + stage * STAGE;
    const __bf16* sW = sA + 3 * PIECE;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 a[3][2], b[3][2];
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          a[q][i] = *reinterpret_cast<const bf16x8*>(sA + q * PIECE + fa + i * 32 * LDK + s * 16);
          b[q][i] = *reinterpret_cast<const bf16x8*>(sW + q * PIECE + fb + i * 32 * LDK + s * 16);
        }
#define TERM(qa, qb)                                                                   \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j) \
      acc[i][j] = mfma(a[qa][i], b[qb][j], acc[i][j]);
      if (TERMS >= 6) { TERM(2, 0) TERM(1, 1) TERM(0, 2) }
      if (TERMS >= 3) { TERM(1, 0) TERM(0, 1) }
      TERM(0, 0)
#undef TERM
    }
  };
  int kt = 0, ti = 0;
  Ring R[D];
  zero();
  if (producer) {
#pragma unroll
    for (int d = 0; d < D; ++d) gload(R[d], d);
    wait_ring<(D - 1) * 8>(R[0]);
    lstore(R[0], 0);
    gload(R[0], D);
  }
  __syncthreads();
  constexpr int U = (D % 2 == 0) ? D : 2 * D;
  bool done = false;
  for (int g = 0; !done; g += U) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (producer) {
        Ring& S = R[(u + 1) % D];
        wait_ring<(D - 1) * 8>(S);
        lstore(S, (u + 1) & 1);
        gload(S, g + u + 1 + D);
      } else {
        compute(u & 1);
        if (++kt == nk) { epilogue(ti); zero(); kt = 0; ++ti; }
      }
      __syncthreads();
      if (g + u + 1 >= total) { done = true; break; }
    }
  }
  if (producer) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}



// ---- v4: v3 with the weights pre-split (three bf16 planes, no conversion in the kernel),
// incremental addressing in the staging waves, and a choice of splitter (hardware
// v_cvt_pk_bf16_f32 vs round-to-nearest-even in integer arithmetic) --------------------------
__device__ __forceinline__ unsigned rne_hi(unsigned u) {   // bf16(x) as the high half of a dword
  return (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u;
}
__device__ __forceinline__ void split2i(float x0, float x1, unsigned& p0, unsigned& p1, unsigned& p2) {
  const unsigned a0 = rne_hi(__float_as_uint(x0)), b0 = rne_hi(__float_as_uint(x1));
  const float ra = x0 - __uint_as_float(a0), rb = x1 - __uint_as_float(b0);
  const unsigned a1 = rne_hi(__float_as_uint(ra)), b1 = rne_hi(__float_as_uint(rb));
  const float sa = ra - __uint_as_float(a1), sb = rb - __uint_as_float(b1);
  const unsigned ua = __float_as_uint(sa), ub = __float_as_uint(sb);
  const unsigned a2 = ua + 0x7fffu + ((ua >> 16) & 1u), b2 = ub + 0x7fffu + ((ub >> 16) & 1u);
  p0 = __builtin_amdgcn_perm(b0, a0, 0x07060302u);
  p1 = __builtin_amdgcn_perm(b1, a1, 0x07060302u);
  p2 = __builtin_amdgcn_perm(b2, a2, 0x07060302u);
}
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
struct Ring4 { f32x4 a[4]; u32x4 w[6]; };

template <int N>
__device__ __forceinline__ void wait_ring4(Ring4& R) {
  asm volatile("s_waitcnt vmcnt(%10)"
               : "+v"(R.a[0]), "+v"(R.a[1]), "+v"(R.a[2]), "+v"(R.a[3]), "+v"(R.w[0]), "+v"(R.w[1]),
                 "+v"(R.w[2]), "+v"(R.w[3]), "+v"(R.w[4]), "+v"(R.w[5])
               : "n"(N)
               : "memory");
}
__device__ __forceinline__ void gl16u(u32x4& d, const void* p) {
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(d) : "v"(p) : "memory");
}

template <int TERMS, int D, int CVT>
__global__ __launch_bounds__(512, 1) void k_gemm_split4(
    const float* __restrict__ A, const __bf16* __restrict__ Ws, const float* __restrict__ bias,
    float* __restrict__ C, int M, int N, int K, long long wplane, int relu) {
  extern __shared__ __attribute__((aligned(16))) __bf16 smem[];
  const int nt = (N + BN - 1) / BN, mt = (M + BM - 1) / BM;
  const int nk = K / BK;
  const int L = blockIdx.x, per = gridDim.x >> 3, jx = L >> 3;
  int cnt;
  const int base = xcd_range(nt * mt, L & 7, cnt);
  const int mine = jx < cnt ? (cnt - jx + per - 1) / per : 0;
  const int total = mine * nk;
  if (total == 0) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool producer = wave >= 4;
  const int pt = tid & 255, lr = pt >> 2, lc = (pt & 3) * 8;
  // staging cursor: (tile ordinal, k-step) of the NEXT load, pointers of that tile
  int p_ti = 0, p_kt = 0;
  const float* ap[2];
  const __bf16* wp[2];
  auto set_tile = [&](int ti) {
    ti = ti < mine ? ti : mine - 1;
    const int T = base + jx + ti * per;
    const int tm = T / nt, tn = T - tm * nt;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      int gm = tm * BM + lr + 64 * p; gm = gm < M ? gm : M - 1;
      int gn = tn * BN + lr + 64 * p; gn = gn < N ? gn : N - 1;
      ap[p] = A + (long long)gm * K + lc;
      wp[p] = Ws + (long long)gn * K + lc;
    }
  };
  auto gload = [&](Ring4& R) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      gl16(R.a[2 * p], ap[p]); gl16(R.a[2 * p + 1], ap[p] + 4);
#pragma unroll
      for (int q = 0; q < 3; ++q) gl16u(R.w[3 * p + q], wp[p] + q * wplane);
      ap[p] += BK; wp[p] += BK;
    }
    if (++p_kt == nk) { p_kt = 0; set_tile(++p_ti); }
  };
  auto lstore = [&](const Ring4& R, int stage) {
    __bf16* sA = smem + stage * STAGE;
    __bf16* sW = sA + 3 * PIECE;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int off = (lr + 64 * p) * LDK + lc;
      if (CVT == 0) {
        bf16x8 p0, p1, p2;
        split8v(R.a[2 * p], R.a[2 * p + 1], p0, p1, p2);
        *reinterpret_cast<bf16x8*>(sA + off) = p0;
        *reinterpret_cast<bf16x8*>(sA + PIECE + off) = p1;
        *reinterpret_cast<bf16x8*>(sA + 2 * PIECE + off) = p2;
      } else {
        u32x4 q0, q1, q2;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          unsigned x0, x1, x2;
          split2i(R.a[2 * p][2 * e], R.a[2 * p][2 * e + 1], x0, x1, x2);
          q0[e] = x0; q1[e] = x1; q2[e] = x2;
          split2i(R.a[2 * p + 1][2 * e], R.a[2 * p + 1][2 * e + 1], x0, x1, x2);
          q0[2 + e] = x0; q1[2 + e] = x1; q2[2 + e] = x2;
        }
        *reinterpret_cast<u32x4*>(sA + off) = q0;
        *reinterpret_cast<u32x4*>(sA + PIECE + off) = q1;
        *reinterpret_cast<u32x4*>(sA + 2 * PIECE + off) = q2;
      }
#pragma unroll
      for (int q = 0; q < 3; ++q) *reinterpret_cast<u32x4*>(sW + q * PIECE + off) = R.w[3 * p + q];
    }
  };
  const int wm = (wave & 3) >> 1, wn = wave & 1;
  const int fa = (wm * 64 + (lane & 31)) * LDK + (lane >> 5) * 8;
  const int fb = (wn * 64 + (lane & 31)) * LDK + (lane >> 5) * 8;
  f32x16 acc[2][2];
  auto zero = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  };
  auto epilogue = [&](int ti) {
    const int T = base + jx + ti * per;
    const int tm = T / nt, tn = T - tm * nt;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int gn = tn * BN + wn * 64 + j * 32 + (lane & 31);
      if (gn >= N) continue;
      const float bv = bias ? bias[gn] : 0.f;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int gm = tm * BM + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          if (gm < M) {
            float v = acc[i][j][r] + bv;
            if (relu) v = v > 0.f ? v : 0.f;
            C[(long long)gm * N + gn] = v;
          }
        }
    }
  };
  auto compute = [&](int stage) {
    const __bf16* sA = smem + stage * STAGE;
    const __bf16* sW = sA + 3 * PIECE;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 a[3][2], b[3][2];
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          a[q][i] = *reinterpret_cast<const bf16x8*>(sA + q * PIECE + fa + i * 32 * LDK + s * 16);
          b[q][i] = *reinterpret_cast<const bf16x8*>(sW + q * PIECE + fb + i * 32 * LDK + s * 16);
        }
#define TERM(qa, qb)                                                                   \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j) \
      acc[i][j] = mfma(a[qa][i], b[qb][j], acc[i][j]);
      if (TERMS >= 6) { TERM(2, 0) TERM(1, 1) TERM(0, 2) }
      if (TERMS >= 3) { TERM(1, 0) TERM(0, 1) }
      TERM(0, 0)
#undef TERM
    }
  };
  int kt = 0, ti = 0;
  Ring4 R[D];
  zero();
  if (producer) {
    set_tile(0);
#pragma unroll
    for (int d = 0; d < D; ++d) gload(R[d]);
    wait_ring4<(D - 1) * 10>(R[0]);
    lstore(R[0], 0);
    gload(R[0]);
  }
  __syncthreads();
  constexpr int U = (D % 2 == 0) ? D : 2 * D;
  bool done = false;
  for (int g = 0; !done; g += U) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (producer) {
        Ring4& S = R[(u + 1) % D];
        wait_ring4<(D - 1) * 10>(S);
        lstore(S, (u + 1) & 1);
        gload(S);
      } else {
        compute(u & 1);
        if (++kt == nk) { epilogue(ti); zero(); kt = 0; ++ti; }
      }
      __syncthreads();
      if (g + u + 1 >= total) { done = true; break; }
    }
  }
  if (producer) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}


// ---- v8: the plain structure again (every wave stages, splits and contracts; one LDS stage;
// two workgroups per CU), but 8 waves of 32 x 64 per 128 x 128 tile instead of 4 of 64 x 64:
// four waves per SIMD at <= 128 registers, so that one wave's split arithmetic, fragment reads
// and barrier waits run under another's MFMAs.  Both operands fp32, split on the fly. ---------
template <int TERMS, int SWZ>
__global__ __launch_bounds__(512, 2) void k_gemm_split8(
    const float* __restrict__ A, const float* __restrict__ W, const float* __restrict__ bias,
    float* __restrict__ C, int M, int N, int K, int relu) {
  __shared__ __attribute__((aligned(16))) __bf16 sA[3 * PIECE];
  __shared__ __attribute__((aligned(16))) __bf16 sW[3 * PIECE];
  const int nt = (N + BN - 1) / BN, mt = (M + BM - 1) / BM;
  int T = blockIdx.x;
  if (SWZ) {   // XCD-aware: workgroup L -> tile of its XCD's contiguous range
    int cnt;
    const int base = xcd_range(nt * mt, T & 7, cnt);
    T = base + (T >> 3);
    if ((int)(blockIdx.x >> 3) >= cnt) return;
  }
  const int tm = T / nt, tn = T - tm * nt;
  const int m0 = tm * BM, n0 = tn * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;          // 4 x 2 waves of 32 x 64
  const int lr = tid >> 2, lc = (tid & 3) * 8;      // staging: 128 rows x 4 lanes x 8 floats
  int gm = m0 + lr; gm = gm < M ? gm : M - 1;
  int gn = n0 + lr; gn = gn < N ? gn : N - 1;
  const float* a_ptr = A + (long long)gm * K + lc;
  const float* w_ptr = W + (long long)gn * K + lc;
  f32x8 ra, rw;
  auto gload = [&](int kt) {
    ra = *reinterpret_cast<const f32x8*>(a_ptr + kt * BK);
    rw = *reinterpret_cast<const f32x8*>(w_ptr + kt * BK);
  };
  auto lstore = [&]() {
    bf16x8 p0, p1, p2;
    const int off = lr * LDK + lc;
    split8(ra, p0, p1, p2);
    *reinterpret_cast<bf16x8*>(sA + off) = p0;
    *reinterpret_cast<bf16x8*>(sA + PIECE + off) = p1;
    *reinterpret_cast<bf16x8*>(sA + 2 * PIECE + off) = p2;
    split8(rw, p0, p1, p2);
    *reinterpret_cast<bf16x8*>(sW + off) = p0;
    *reinterpret_cast<bf16x8*>(sW + PIECE + off) = p1;
    *reinterpret_cast<bf16x8*>(sW + 2 * PIECE + off) = p2;
  };
  f32x16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  const int nk = K / BK;
  const int fa = (wm * 32 + (lane & 31)) * LDK + (lane >> 5) * 8;
  const int fb = (wn * 64 + (lane & 31)) * LDK + (lane >> 5) * 8;
  gload(0);
  lstore();
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) gload(kt + 1);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 a[3], b[3][2];
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        a[q] = *reinterpret_cast<const bf16x8*>(sA + q * PIECE + fa + s * 16);
#pragma unroll
        for (int j = 0; j < 2; ++j)
          b[q][j] = *reinterpret_cast<const bf16x8*>(sW + q * PIECE + fb + j * 32 * LDK + s * 16);
      }
#define TERM(qa, qb) _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[j] = mfma(a[qa], b[qb][j], acc[j]);
      if (TERMS >= 6) { TERM(2, 0) TERM(1, 1) TERM(0, 2) }
      if (TERMS >= 3) { TERM(1, 0) TERM(0, 1) }
      TERM(0, 0)
#undef TERM
    }
    __syncthreads();
    if (kt + 1 < nk) {
      lstore();
      __syncthreads();
    }
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int gcol = n0 + wn * 64 + j * 32 + (lane & 31);
    if (gcol >= N) continue;
    const float bv = bias ? bias[gcol] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int grow = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (grow < M) {
        float v = acc[j][r] + bv;
        if (relu) v = v > 0.f ? v : 0.f;
        C[(long long)grow * N + gcol] = v;
      }
    }
  }
}


// ---- v9: eight symmetric waves (32 x 64 each), ONE workgroup per CU, persistent, two LDS
// stages, one barrier per k-step: every wave splits its 16 elements of step g+1 into the other
// stage IN THE SAME BASIC BLOCK as its 24 MFMAs of step g (SCHED: the two interleaved by
// sched_group_barrier), loads two steps ahead in two register sets (inline asm, counted vmcnt) --
struct Ring9 { f32x4 v[4]; };    // A row (8 floats), W row (8 floats)
template <int N>
__device__ __forceinline__ void wait_ring9(Ring9& R) {
  asm volatile("s_waitcnt vmcnt(%4)" : "+v"(R.v[0]), "+v"(R.v[1]), "+v"(R.v[2]), "+v"(R.v[3]) : "n"(N) : "memory");
}

template <int TERMS, int SCHED>
__global__ __launch_bounds__(512, 1) void k_gemm_split9(
    const float* __restrict__ A, const float* __restrict__ W, const float* __restrict__ bias,
    float* __restrict__ C, int M, int N, int K, int relu) {
  extern __shared__ __attribute__((aligned(16))) __bf16 smem[];
  const int nt = (N + BN - 1) / BN, mt = (M + BM - 1) / BM;
  const int nk = K / BK;
  const int L = blockIdx.x, per = gridDim.x >> 3, jx = L >> 3;
  int cnt;
  const int base = xcd_range(nt * mt, L & 7, cnt);
  const int mine = jx < cnt ? (cnt - jx + per - 1) / per : 0;
  const int total = mine * nk;
  if (total == 0) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = tid >> 2, lc = (tid & 3) * 8;
  // staging cursor
  int p_ti = 0, p_kt = 0;
  const float *ap, *wp;
  auto set_tile = [&](int ti) {
    ti = ti < mine ? ti : mine - 1;
    const int T = base + jx + ti * per;
    const int tm = T / nt, tn = T - tm * nt;
    int gm = tm * BM + lr; gm = gm < M ? gm : M - 1;
    int gn = tn * BN + lr; gn = gn < N ? gn : N - 1;
    ap = A + (long long)gm * K + lc;
    wp = W + (long long)gn * K + lc;
  };
  auto gload = [&](Ring9& R) {
    gl16(R.v[0], ap); gl16(R.v[1], ap + 4);
    gl16(R.v[2], wp); gl16(R.v[3], wp + 4);
    ap += BK; wp += BK;
    if (++p_kt == nk) { p_kt = 0; set_tile(++p_ti); }
  };
  const int soff = lr * LDK + lc;
  auto lstore = [&](const Ring9& R, int stage) {
    __bf16* sA = smem + stage * STAGE;
    __bf16* sW = sA + 3 * PIECE;
    bf16x8 p0, p1, p2;
    split8v(R.v[0], R.v[1], p0, p1, p2);
    *reinterpret_cast<bf16x8*>(sA + soff) = p0;
    *reinterpret_cast<bf16x8*>(sA + PIECE + soff) = p1;
    *reinterpret_cast<bf16x8*>(sA + 2 * PIECE + soff) = p2;
    split8v(R.v[2], R.v[3], p0, p1, p2);
    *reinterpret_cast<bf16x8*>(sW + soff) = p0;
    *reinterpret_cast<bf16x8*>(sW + PIECE + soff) = p1;
    *reinterpret_cast<bf16x8*>(sW + 2 * PIECE + soff) = p2;
  };
  const int wm = wave >> 1, wn = wave & 1;
  const int fa = (wm * 32 + (lane & 31)) * LDK + (lane >> 5) * 8;
  const int fb = (wn * 64 + (lane & 31)) * LDK + (lane >> 5) * 8;
  f32x16 acc[2];
  auto zero = [&]() {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  };
  auto epilogue = [&](int ti) {
    const int T = base + jx + ti * per;
    const int tm = T / nt, tn = T - tm * nt;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int gcol = tn * BN + wn * 64 + j * 32 + (lane & 31);
      if (gcol >= N) continue;
      const float bv = bias ? bias[gcol] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int grow = tm * BM + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (grow < M) {
          float v = acc[j][r] + bv;
          if (relu) v = v > 0.f ? v : 0.f;
          C[(long long)grow * N + gcol] = v;
        }
      }
    }
  };
  auto compute = [&](int stage) {
    const __bf16* sA = smem + stage * STAGE;
    const __bf16* sW = sA + 3 * PIECE;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 a[3], b[3][2];
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        a[q] = *reinterpret_cast<const bf16x8*>(sA + q * PIECE + fa + s * 16);
#pragma unroll
        for (int j = 0; j < 2; ++j)
          b[q][j] = *reinterpret_cast<const bf16x8*>(sW + q * PIECE + fb + j * 32 * LDK + s * 16);
      }
#define TERM(qa, qb) _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[j] = mfma(a[qa], b[qb][j], acc[j]);
      if (TERMS >= 6) { TERM(2, 0) TERM(1, 1) TERM(0, 2) }
      if (TERMS >= 3) { TERM(1, 0) TERM(0, 1) }
      TERM(0, 0)
#undef TERM
    }
  };
  int kt = 0, ti = 0;
  Ring9 R[2];
  zero();
  set_tile(0);
  gload(R[0]); gload(R[1]);
  wait_ring9<4>(R[0]);
  lstore(R[0], 0);
  gload(R[0]);
  __syncthreads();
  bool done = false;
  for (int g = 0; !done; g += 2) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      Ring9& S = R[(u + 1) & 1];
      wait_ring9<4>(S);
      if (SCHED) {
        // fragment reads first (they need only the barrier), then the split arithmetic of step
        // g + 1 between the MFMAs of step g, the LDS writes last
        compute(u & 1);
        lstore(S, (u + 1) & 1);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          __builtin_amdgcn_sched_group_barrier(0x100, 9, 0);
#pragma unroll
          for (int i = 0; i < 12; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
          }
        }
        __builtin_amdgcn_sched_group_barrier(0x200, 6, 0);
      } else {
        lstore(S, (u + 1) & 1);
        compute(u & 1);
      }
      gload(S);
      if (++kt == nk) { epilogue(ti); zero(); kt = 0; ++ti; }
      __syncthreads();
      if (g + u + 1 >= total) { done = true; break; }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

static float *dA, *dW, *dC, *dC2, *dBias;
static __bf16* dWs;
static std::vector<float> hA, hW, hBias;

static double check(const float* dOut, int M, int N, int K, int relu, const char* tag) {
  std::vector<float> hC((size_t)M * N);
  CK(hipMemcpy(hC.data(), dOut, hC.size() * 4, hipMemcpyDeviceToHost));
  double maxerr = 0, sumsq = 0, maxref = 0;
  unsigned s = 12345u;
  const int NS = 20000;
  for (int t = 0; t < NS; ++t) {
    s = s * 1664525u + 1013904223u; const int m = (s >> 8) % M;
    s = s * 1664525u + 1013904223u; const int n = (s >> 8) % N;
    double ref = hBias[n];
    for (int k = 0; k < K; ++k) ref += (double)hA[(size_t)m * K + k] * (double)hW[(size_t)n * K + k];
    if (relu && ref < 0) ref = 0;
    const double e = fabs((double)hC[(size_t)m * N + n] - ref);
    if (e > maxerr) maxerr = e;
    sumsq += e * e;
    if (fabs(ref) > maxref) maxref = fabs(ref);
  }
  // the edge rows / columns too
  for (int m : {0, M - 1}) for (int n : {0, N - 1}) {
    double ref = hBias[n];
    for (int k = 0; k < K; ++k) ref += (double)hA[(size_t)m * K + k] * (double)hW[(size_t)n * K + k];
    if (relu && ref < 0) ref = 0;
    const double e = fabs((double)hC[(size_t)m * N + n] - ref);
    if (e > maxerr) maxerr = e;
  }
  printf("    %-26s max |err| %.3e   rms %.3e   (max |ref| %.2f)\n", tag, maxerr, sqrt(sumsq / NS), maxref);
  return maxerr;
}

template <int TERMS>
static void run_split(const char* tag, int M, int N, int K, int relu, bool verify) {
  const int nt = (N + BN - 1) / BN, mt = (M + BM - 1) / BM;
  const long long plane = (long long)N * K;
  k_split_w<<<(unsigned)((plane / 8 + 255) / 256), 256>>>(dW, dWs, plane / 8, plane);
  auto launch = [&]() {
    k_gemm_split<TERMS><<<nt * mt, 256>>>(dA, dWs, dBias, dC, M, N, K, plane, relu);
  };
  for (int i = 0; i < 3; ++i) launch();
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int R = 20;
  hipEventRecord(e0);
  for (int i = 0; i < R; ++i) launch();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / R, tf = 2.0 * M * N * K / (us * 1e-6) / 1e12;
  printf("  %-28s %8.1f us  %7.1f TFLOP/s fp32-equivalent (%.2f of 157.3; bf16 pipe %.2f of 2500)\n",
         tag, us, tf, tf / 157.3, tf * TERMS / 2500.0);
  if (verify) check(dC, M, N, K, relu, tag);
}


template <int TERMS>
static void run_split2(const char* tag, int M, int N, int K, int relu, bool verify, int grid) {
  const long long plane = (long long)N * K;
  k_split_w<<<(unsigned)((plane / 8 + 255) / 256), 256>>>(dW, dWs, plane / 8, plane);
  const int smem = 2 * STAGE * 2;
  CK(hipFuncSetAttribute((const void*)k_gemm_split2<TERMS>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
  CK(hipMemset(dC, 0xff, (size_t)M * N * 4));
  auto launch = [&]() {
    k_gemm_split2<TERMS><<<grid, 512, smem>>>(dA, dWs, dBias, dC, M, N, K, plane, relu);
  };
  for (int i = 0; i < 3; ++i) launch();
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int R = 20;
  hipEventRecord(e0);
  for (int i = 0; i < R; ++i) launch();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / R, tf = 2.0 * M * N * K / (us * 1e-6) / 1e12;
  printf("  %-28s %8.1f us  %7.1f TFLOP/s fp32-equivalent (%.2f of 157.3; bf16 pipe %.2f of 2500)\n",
         tag, us, tf, tf / 157.3, tf * TERMS / 2500.0);
  if (verify) check(dC, M, N, K, relu, tag);
}


template <int TERMS, int D>
static void run_split3(const char* tag, int M, int N, int K, int relu, bool verify, int grid) {
  const int smem = 2 * STAGE * 2;
  CK(hipFuncSetAttribute((const void*)k_gemm_split3<TERMS, D>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
  CK(hipMemset(dC, 0xff, (size_t)M * N * 4));
  auto launch = [&]() {
    k_gemm_split3<TERMS, D><<<grid, 512, smem>>>(dA, dW, dBias, dC, M, N, K, relu);
  };
  for (int i = 0; i < 3; ++i) launch();
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int R = 20;
  hipEventRecord(e0);
  for (int i = 0; i < R; ++i) launch();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / R, tf = 2.0 * M * N * K / (us * 1e-6) / 1e12;
  printf("  %-28s %8.1f us  %7.1f TFLOP/s fp32-equivalent (%.2f of 157.3; bf16 pipe %.2f of 2500)\n",
         tag, us, tf, tf / 157.3, tf * TERMS / 2500.0);
  if (verify) check(dC, M, N, K, relu, tag);
}


template <int TERMS, int D, int FLAGS>
static void run_ablate(const char* tag, int M, int N, int K, int relu, bool verify, int grid) {
  const int smem = 2 * STAGE * 2;
  CK(hipFuncSetAttribute((const void*)k_gemm_ablate<TERMS, D, FLAGS>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
  CK(hipMemset(dC, 0xff, (size_t)M * N * 4));
  auto launch = [&]() {
    k_gemm_ablate<TERMS, D, FLAGS><<<grid, 512, smem>>>(dA, dW, dBias, dC, M, N, K, relu);
  };
  for (int i = 0; i < 3; ++i) launch();
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int R = 20;
  hipEventRecord(e0);
  for (int i = 0; i < R; ++i) launch();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / R, tf = 2.0 * M * N * K / (us * 1e-6) / 1e12;
  printf("  %-28s %8.1f us  %7.1f TFLOP/s fp32-equivalent (%.2f of 157.3; bf16 pipe %.2f of 2500)\n",
         tag, us, tf, tf / 157.3, tf * TERMS / 2500.0);
  if (verify) check(dC, M, N, K, relu, tag);
}


template <int TERMS, int D>
static void run_split5(const char* tag, int M, int N, int K, int relu, bool verify, int grid) {
  const int smem = 2 * STAGE * 2;
  CK(hipFuncSetAttribute((const void*)k_gemm_split5<TERMS, D>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
  CK(hipMemset(dC, 0xff, (size_t)M * N * 4));
  auto launch = [&]() {
    k_gemm_split5<TERMS, D><<<grid, 512, smem>>>(dA, dW, dBias, dC, M, N, K, relu);
  };
  for (int i = 0; i < 3; ++i) launch();
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int R = 20;
  hipEventRecord(e0);
  for (int i = 0; i < R; ++i) launch();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / R, tf = 2.0 * M * N * K / (us * 1e-6) / 1e12;
  printf("  %-28s %8.1f us  %7.1f TFLOP/s fp32-equivalent (%.2f of 157.3; bf16 pipe %.2f of 2500)\n",
         tag, us, tf, tf / 157.3, tf * TERMS / 2500.0);
  if (verify) check(dC, M, N, K, relu, tag);
}


template <int TERMS, int D, int CVT>
static void run_split4(const char* tag, int M, int N, int K, int relu, bool verify, int grid) {
  const long long plane = (long long)N * K;
  k_split_w<<<(unsigned)((plane / 8 + 255) / 256), 256>>>(dW, dWs, plane / 8, plane);
  const int smem = 2 * STAGE * 2;
  CK(hipFuncSetAttribute((const void*)k_gemm_split4<TERMS, D, CVT>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
  CK(hipMemset(dC, 0xff, (size_t)M * N * 4));
  auto launch = [&]() {
    k_gemm_split4<TERMS, D, CVT><<<grid, 512, smem>>>(dA, dWs, dBias, dC, M, N, K, plane, relu);
  };
  for (int i = 0; i < 3; ++i) launch();
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int R = 20;
  hipEventRecord(e0);
  for (int i = 0; i < R; ++i) launch();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / R, tf = 2.0 * M * N * K / (us * 1e-6) / 1e12;
  printf("  %-28s %8.1f us  %7.1f TFLOP/s fp32-equivalent (%.2f of 157.3; bf16 pipe %.2f of 2500)\n",
         tag, us, tf, tf / 157.3, tf * TERMS / 2500.0);
  if (verify) check(dC, M, N, K, relu, tag);
}


template <int TERMS, int SWZ>
static void run_split8(const char* tag, int M, int N, int K, int relu, bool verify) {
  const int nt = (N + BN - 1) / BN, mt = (M + BM - 1) / BM;
  const int grid = SWZ ? ((nt * mt + 7) / 8) * 8 : nt * mt;
  CK(hipMemset(dC, 0xff, (size_t)M * N * 4));
  auto launch = [&]() { k_gemm_split8<TERMS, SWZ><<<grid, 512>>>(dA, dW, dBias, dC, M, N, K, relu); };
  for (int i = 0; i < 3; ++i) launch();
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int R = 20;
  hipEventRecord(e0);
  for (int i = 0; i < R; ++i) launch();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / R, tf = 2.0 * M * N * K / (us * 1e-6) / 1e12;
  printf("  %-28s %8.1f us  %7.1f TFLOP/s fp32-equivalent (%.2f of 157.3; bf16 pipe %.2f of 2500)\n",
         tag, us, tf, tf / 157.3, tf * TERMS / 2500.0);
  if (verify) check(dC, M, N, K, relu, tag);
}


template <int TERMS, int SCHED>
static void run_split9(const char* tag, int M, int N, int K, int relu, bool verify, int grid) {
  const int smem = 2 * STAGE * 2;
  CK(hipFuncSetAttribute((const void*)k_gemm_split9<TERMS, SCHED>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
  CK(hipMemset(dC, 0xff, (size_t)M * N * 4));
  auto launch = [&]() { k_gemm_split9<TERMS, SCHED><<<grid, 512, smem>>>(dA, dW, dBias, dC, M, N, K, relu); };
  for (int i = 0; i < 3; ++i) launch();
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int R = 20;
  hipEventRecord(e0);
  for (int i = 0; i < R; ++i) launch();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / R, tf = 2.0 * M * N * K / (us * 1e-6) / 1e12;
  printf("  %-28s %8.1f us  %7.1f TFLOP/s fp32-equivalent (%.2f of 157.3; bf16 pipe %.2f of 2500)\n",
         tag, us, tf, tf / 157.3, tf * TERMS / 2500.0);
  if (verify) check(dC, M, N, K, relu, tag);
}

static void run_prod(int M, int N, int K, int relu, bool verify) {
  pn_gemm_desc d;
  memset(&d, 0, sizeof(d));
  d.A = dA; d.lda = K; d.W = dW; d.ldw = K; d.bias = dBias; d.C = dC2; d.ldc = N;
  d.M = M; d.N = N; d.K = K; d.batch = 1; d.flags = relu ? 1 : 0;
  for (int i = 0; i < 3; ++i) pn_gemm_f32(&d, nullptr);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int R = 20;
  hipEventRecord(e0);
  for (int i = 0; i < R; ++i) pn_gemm_f32(&d, nullptr);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / R, tf = 2.0 * M * N * K / (us * 1e-6) / 1e12;
  printf("  %-28s %8.1f us  %7.1f TFLOP/s (%.2f of 157.3)\n", "pn_gemm_f32 (fp32 MFMA)", us, tf, tf / 157.3);
  if (verify) check(dC2, M, N, K, relu, "pn_gemm_f32");
}

static float gauss(unsigned& s) {
  float acc = 0.f;
  for (int i = 0; i < 12; ++i) { s = s * 1664525u + 1013904223u; acc += (float)(s >> 8) / 16777216.f; }
  return acc - 6.f;
}

int main(int argc, char** argv) {
  setvbuf(stdout, NULL, _IONBF, 0);
  const bool verify = argc < 2 || strcmp(argv[1], "noverify") != 0;
  const size_t maxA = (size_t)66800 * 1024, maxW = (size_t)2048 * 2048, maxC = (size_t)66800 * 1024;
  CK(hipMalloc(&dA, maxA * 4)); CK(hipMalloc(&dW, maxW * 4)); CK(hipMalloc(&dC, maxC * 4));
  CK(hipMalloc(&dC2, maxC * 4)); CK(hipMalloc(&dWs, maxW * 6)); CK(hipMalloc(&dBias, 4096 * 4));
  hA.resize(maxA); hW.resize(maxW); hBias.resize(4096);
  unsigned s = 1u;
  for (auto& v : hA) v = gauss(s);
  for (auto& v : hW) v = gauss(s) * 0.0625f;
  for (auto& v : hBias) v = gauss(s);
  CK(hipMemcpy(dA, hA.data(), maxA * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dW, hW.data(), maxW * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dBias, hBias.data(), 4096 * 4, hipMemcpyHostToDevice));
  const int shapes[][4] = {{21950, 1024, 256, 1}, {21950, 256, 1024, 0}, {21950, 544, 256, 0},
                           {66800, 256, 256, 0}, {21950, 256, 256, 0}, {16700, 512, 1024, 1},
                           {4200, 1024, 2048, 1}, {333, 200, 64, 0}};
  const int only = argc > 2 ? atoi(argv[2]) : -1;
  int si = -1;
  for (auto& sh : shapes) {
    if (++si != only && only >= 0) continue;
    const int M = sh[0], N = sh[1], K = sh[2], relu = sh[3];
    printf("M=%d N=%d K=%d relu=%d\n", M, N, K, relu);
    run_prod(M, N, K, relu, verify);
    run_split<6>("v1 x6", M, N, K, relu, verify);
    run_split3<6, 2>("v3 x6 D=2", M, N, K, relu, verify, 256);
    run_split8<6, 0>("v8 x6 (8 waves of 32x64)", M, N, K, relu, verify);
    run_split8<6, 1>("v8 x6 XCD-aware", M, N, K, relu, verify);
    run_split8<1, 1>("v8 x1 XCD-aware", M, N, K, relu, false);
    run_split9<6, 0>("v9 x6 compiler order", M, N, K, relu, verify, 256);
    run_split9<6, 1>("v9 x6 MFMA/VALU interleaved", M, N, K, relu, verify, 256);
    run_split9<1, 0>("v9 x1", M, N, K, relu, false, 256);
    if (only >= 0) {
      run_split5<6, 2>("v5 x6 D=2 (whole lines)", M, N, K, relu, verify, 256);
      run_split5<1, 2>("v5 x1 D=2", M, N, K, relu, false, 256);
    }
    if (only >= 0) {
      run_ablate<6, 2, 0>("ablate: none", M, N, K, relu, false, 256);
      run_ablate<6, 2, 1>("ablate: no global loads", M, N, K, relu, false, 256);
      run_ablate<6, 2, 2>("ablate: no split arithmetic", M, N, K, relu, false, 256);
      run_ablate<6, 2, 4>("ablate: no LDS writes", M, N, K, relu, false, 256);
      run_ablate<6, 2, 8>("ablate: no LDS reads / MFMA", M, N, K, relu, false, 256);
      run_ablate<6, 2, 3>("ablate: no loads, no split", M, N, K, relu, false, 256);
      run_ablate<6, 2, 7>("ablate: staging waves idle", M, N, K, relu, false, 256);
      run_ablate<6, 2, 14>("ablate: loads only", M, N, K, relu, false, 256);
      run_ablate<6, 2, 16>("reads first: full kernel", M, N, K, relu, verify, 256);
      run_ablate<6, 2, 23>("reads first: staging waves idle", M, N, K, relu, false, 256);
      run_ablate<6, 2, 7 + 32>("staging idle, LDS reads only", M, N, K, relu, false, 256);
      run_ablate<6, 2, 7 + 64>("staging idle, MFMA only", M, N, K, relu, false, 256);
      run_ablate<1, 2, 7 + 64>("staging idle, MFMA only x1", M, N, K, relu, false, 256);
      run_ablate<6, 2, 15>("barriers + epilogue only", M, N, K, relu, false, 256);
      run_ablate<6, 2, 7 + 64 + 128>("staging idle, MFMA only, zero operands", M, N, K, relu, false, 256);
      run_ablate<6, 2, 7 + 64 + 256>("staging idle, MFMA only, random bits", M, N, K, relu, false, 256);
      run_split4<6, 2, 0>("v4 x6 D=2 cvt_pk", M, N, K, relu, verify, 256);
      run_split4<6, 2, 1>("v4 x6 D=2 int RNE", M, N, K, relu, verify, 256);
    }
  }
  return 0;
}
