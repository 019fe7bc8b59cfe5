// Tuning aid (not part of the library): the fp32 GEMM C = act(A W^T + bias) computed on the bf16
// matrix pipe from operands stored PRE-SPLIT ("S3" format): every fp32 number is exactly the sum of
// three bf16 numbers (x = x0 + x1 + x2, each the round-to-nearest bf16 of what is left), products
// of bf16 pieces are exact in fp32, and the six largest of the nine piece products are summed by
// v_mfma_f32_32x32x16_bf16 into fp32 accumulators (dropped terms < 2^-26 |ab|).
//
// S3 layout of an [R x K] operand (K % 16 == 0, R padded to 32): blocks of 32 rows x 16 k; block
// (rb, kb) at ((rb * K/16 + kb) * 3 KiB); inside a block three 1 KiB planes (x0, x1, x2); inside a
// plane the MFMA fragment order: 16 bytes (8 consecutive k) per lane, lane = (k % 16 / 8) * 32 +
// r % 32.  One wave-wide 16-byte load / LDS-DMA piece / ds_read_b128 moves one plane of one block:
// fully coalesced in HBM, lane-linear (conflict-free) in LDS, no swizzle anywhere.
//
//   hipcc --offload-arch=gfx950 -O3 tools/gemm_s3_probe.hip -o tools/bin/gemm_s3_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

union frag_t { uint4 u; bf16x8 v; __bf16 e[8]; };

__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
__device__ __forceinline__ f32x16 mfma(const frag_t& a, const frag_t& b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, c, 0, 0, 0);
}
__device__ __forceinline__ void split3(float x, __bf16& a, __bf16& b, __bf16& c) {
  a = (__bf16)x;
  float r = x - (float)a;
  b = (__bf16)r;
  r = r - (float)b;
  c = (__bf16)r;
}

// fp32 row-major [R x K] (ld) (+ optional addend, row index modulo add_rows) -> S3
__global__ __launch_bounds__(256) void k_s3_split(const float* __restrict__ X, int ld,
                                                  const float* __restrict__ add, int add_rows,
                                                  uint4* __restrict__ S, int R, int K) {
  const int KB = K >> 4;
  const int lane = threadIdx.x & 63;
  const int64_t piece = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);   // (rb, kb)
  const int64_t npieces = (int64_t)((R + 31) >> 5) * KB;
  if (piece >= npieces) return;
  const int rb = (int)(piece / KB), kb = (int)(piece - (int64_t)rb * KB);
  const int r = rb * 32 + (lane & 31), k = kb * 16 + (lane >> 5) * 8;
  float v[8];
  if (r < R) {
    const float4 a = *(const float4*)(X + (int64_t)r * ld + k);
    const float4 b = *(const float4*)(X + (int64_t)r * ld + k + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    if (add) {
      const float* p = add + (int64_t)(r % add_rows) * K + k;
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] += p[i];
    }
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = 0.f;
  }
  frag_t p0, p1, p2;
#pragma unroll
  for (int i = 0; i < 8; ++i) split3(v[i], p0.e[i], p1.e[i], p2.e[i]);
  uint4* o = S + piece * 192 + lane;
  o[0] = p0.u; o[64] = p1.u; o[128] = p2.u;
}

// One workgroup = 4 waves side by side along N: tile (32 MB) x 256, wave tile (32 MB) x 64.
// A and W: LDS-DMA into a two-slot ring (one 16-deep k-block per slot); A is shared by the four
// waves, each wave reads its own 64 columns of W.
// NPROD: 6 = the product form, 1 = timing only.  ABL: 0 everything, 1 no MFMAs, 2 no loads.
// EPI: 0 = fp32 row-major output, 1 = S3 output (the next GEMM's A operand).
template <int MB, int NPROD, int ABL, int EPI, int NS = 2, int PRIO = 0>
__global__ __launch_bounds__(256, 2) void k_gemm_s3(const uint4* __restrict__ A,
                                                    const uint4* __restrict__ W,
                                                    const float* __restrict__ bias,
                                                    float* __restrict__ C, uint4* __restrict__ CS,
                                                    int M, int N, int K, int ldc, int relu, long long* stats = nullptr) {
  constexpr int SLOT = (MB + 8) * 3 * 64;           // uint4 per ring slot: A pieces, then W pieces
  constexpr int TR = 32 * 68;                       // floats per wave of transposition scratch
  constexpr int SMEM_U4 = (EPI == 1 && TR > NS * SLOT) ? TR : NS * SLOT;   // 4 waves x TR floats = TR uint4
  __shared__ __attribute__((aligned(1024))) uint4 smem[SMEM_U4];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int KB = K >> 4;
  const int RB = (M + 31) >> 5, CB = (N + 31) >> 5;
  const int nt = (N + 255) >> 8, mt = (RB + MB - 1) / MB;
  // XCD-aware: workgroup b runs on XCD b % 8; give each XCD a contiguous tile range
  const int ntiles = nt * mt;
  int t;
  {
    const int b = blockIdx.x, x = b & 7, i = b >> 3, q = ntiles >> 3, r = ntiles & 7;
    t = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
  }
  const int tm = t / nt, tn = t - tm * nt;
  const int rb0 = tm * MB;

  f32x16 acc[MB][2];
#pragma unroll
  for (int m = 0; m < MB; ++m)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][j][r] = 0.f;

  // every operand piece (one plane of one 32 x 16 block, 1 KiB) by LDS-DMA; the compiler's
  // counter model drains the whole queue in front of the first use of any ordinary load
  // issued beside an LDS-DMA, so there are no ordinary loads in the loop.  Wave w moves pieces
  // w, w + 4, ...: their source bases are wave-uniform and computed once.
  constexpr int NP = (MB + 8) * 3, NPW = (NP + 3) / 4;
  const uint4* gp[NPW];
#pragma unroll
  for (int i = 0; i < NPW; ++i) {
    const int p = min(wave + 4 * i, NP - 1), blk = p / 3, plane = p - blk * 3;
    const int64_t rowblk = blk < MB ? min(rb0 + blk, RB - 1) : min(tn * 8 + blk - MB, CB - 1);
    gp[i] = (blk < MB ? A : W) + rowblk * KB * 192 + plane * 64 + lane;
  }
  auto issue = [&](int kb, int slot) {
    if (ABL == 2) return;
#pragma unroll
    for (int i = 0; i < NPW; ++i)
      if (i * 4 + 3 < NP || wave + 4 * i < NP)
        glds16(gp[i] + kb * 192, &smem[slot * SLOT + (wave + 4 * i) * 64]);
  };
  auto readB = [&](int slot, frag_t (&b)[2][3]) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int p = 0; p < 3; ++p) b[j][p].u = smem[slot * SLOT + (MB * 3 + (wave * 2 + j) * 3 + p) * 64 + lane];
  };
  auto readA = [&](int slot, frag_t (&a)[MB][3]) {
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
      for (int p = 0; p < 3; ++p) a[m][p].u = smem[slot * SLOT + (m * 3 + p) * 64 + lane];
  };
  auto mma = [&](frag_t (&a)[MB][3], frag_t (&b)[2][3]) {
    if (ABL == 1) {
#pragma unroll
      for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int p = 0; p < 3; ++p) acc[m][j][p] += __uint_as_float(a[m][p].u.x ^ b[j][p].u.y);
      return;
    }
    // small terms first; consecutive MFMAs go to different accumulators
    constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
    for (int q = 6 - NPROD; q < 6; ++q)
#pragma unroll
      for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[m][j] = mfma(a[m][PA[q]], b[j][PB[q]], acc[m][j]);
  };
  // vmcnt(0) as a real s_waitcnt the compiler's counter model sees (an asm statement is opaque to
  // it: it would then wait again, for the LDS-DMA it believes pending, in front of every ds_read)
  auto drain = [&] {
    __builtin_amdgcn_sched_barrier(0);    // the MFMAs stay in front of the wait
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), expcnt / lgkmcnt untouched
    __syncthreads();
  };

  frag_t a[MB][3], b[2][3];
  if (PRIO) {
    // two workgroups share a CU, i.e. two waves share each SIMD's matrix pipe.  Started together
    // they stay in lockstep (the arbiter alternates between their MFMAs, both reach their
    // load / LDS-read segment at the same time and the pipe idles through it); a static priority
    // for one of the two -- the hardware wave slot's parity -- makes them alternate instead
    const unsigned slot = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4);   // HW_ID.wave_id
    if (slot & 1) __builtin_amdgcn_s_setprio(1);
  }
  if (NS == 2) {
    issue(0, 0);
    drain();
    for (int kb = 0; kb < KB; kb += 2) {
      readA(0, a); readB(0, b);
      if (kb + 1 < KB) issue(kb + 1, 1);
      mma(a, b);
      drain();
      if (kb + 1 < KB) {
        readA(1, a); readB(1, b);
        if (kb + 2 < KB) issue(kb + 2, 0);
        mma(a, b);
        drain();
      }
    }
  } else {
    // three slots: the pieces of step s + 2 are in flight across the barrier that ends step s
    const bool short_wave = wave + 4 * (NPW - 1) >= NP;    // this wave moves NPW - 1 pieces per step
    auto wait_but_one_step = [&] {
      if (short_wave) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NPW - 1) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NPW) : "memory");
    };
    auto step = [&](int s, int u) {
      readA(u, a); readB(u, b);
      if (s + 2 < KB) issue(s + 2, (u + 2) % 3);
      mma(a, b);
      __builtin_amdgcn_sched_barrier(0);
      if (s + 2 < KB) wait_but_one_step();
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    };
    issue(0, 0);
    if (KB > 1) { issue(1, 1); wait_but_one_step(); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    for (int kb = 0; kb < KB; kb += 3) {
      step(kb, 0);
      if (kb + 1 < KB) step(kb + 1, 1);
      if (kb + 2 < KB) step(kb + 2, 2);
    }
  }
  if (PRIO) __builtin_amdgcn_s_setprio(0);
  __syncthreads();

  // ---- epilogue ----
  const int li = lane & 31, lh = lane >> 5;
  if (EPI == 0) {
    const bool full = (rb0 + MB) * 32 <= M;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = tn * 256 + wave * 64 + j * 32 + li;
      if (col >= N) continue;
      const float bv = bias ? bias[col] : 0.f;
      float* cp = C + (int64_t)(rb0 * 32 + 4 * lh) * ldc + col;
#pragma unroll
      for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int dr = m * 32 + (r & 3) + 8 * (r >> 2);
          float v = acc[m][j][r] + bv;
          if (relu) v = fmaxf(v, 0.f);
          if (full || rb0 * 32 + 4 * lh + dr < M) cp[(int64_t)dr * ldc] = v;
        }
    }
  } else {
    // transpose each 32 x 64 piece through LDS: fp32 [32][68], then lane (row, k-half) reads 8
    // consecutive columns, splits them and writes one 16-byte piece per plane
    float* tr = (float*)smem + wave * TR;
    const int KBo = N >> 4;
    float bv[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = tn * 256 + wave * 64 + j * 32 + li;
      bv[j] = bias && col < N ? bias[col] : 0.f;
    }
#pragma unroll
    for (int m = 0; m < MB; ++m) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = acc[m][j][r] + bv[j];
          if (relu) v = fmaxf(v, 0.f);
          tr[((r & 3) + 8 * (r >> 2) + 4 * lh) * 68 + j * 32 + li] = v;
        }
      __builtin_amdgcn_wave_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (rb0 + m < RB) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int kbo = tn * 16 + wave * 4 + q;
          if (kbo >= KBo) break;
          const float4 x = *(const float4*)(tr + li * 68 + q * 16 + lh * 8);
          const float4 y = *(const float4*)(tr + li * 68 + q * 16 + lh * 8 + 4);
          const float v[8] = {x.x, x.y, x.z, x.w, y.x, y.y, y.z, y.w};
          frag_t p0, p1, p2;
#pragma unroll
          for (int i = 0; i < 8; ++i) split3(v[i], p0.e[i], p1.e[i], p2.e[i]);
          uint4* o = CS + ((int64_t)(rb0 + m) * KBo + kbo) * 192 + lane;
          o[0] = p0.u; o[64] = p1.u; o[128] = p2.u;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
    }
  }
}


// ---------------------------------------------------------------------------------------------
// Ping-pong form: ONE workgroup of 8 waves per CU, tile 96 x 256.  Waves 0-3 (group 0) own
// columns 0..127, waves 4-7 (group 1) columns 128..255; a wave's tile is 96 x 32 (3 accumulators).
// Time is cut into phases by workgroup barriers; in every phase one group issues the 36 MFMAs of a
// 32-deep k-stage from registers while the other group, on the same SIMDs, reads its fragments of
// the next stage from LDS (24 ds_read_b128) and issues the LDS-DMA of a later 16-deep k-step
// (33 pieces of 1 KiB over its 4 waves).  The ring holds four 16-deep k-steps (132 KiB).
//   phase 2s-1: G0 READ(s), issues k-step 2s+2      | G1 MMA(s-1)
//   phase 2s  : G0 MMA(s)                           | G1 READ(s), issues k-step 2s+3
// Every read happens after a barrier that follows the issuing waves' vmcnt(0); every slot is
// overwritten at least one barrier after its last reader finished (see the probe's notes).
template <int NPROD, int EPI, int ROT = 0, int ABL = 0>
__global__ __launch_bounds__(512, 1) void k_gemm_s3p(const uint4* __restrict__ A,
                                                     const uint4* __restrict__ W,
                                                     const float* __restrict__ bias,
                                                     float* __restrict__ C, uint4* __restrict__ CS,
                                                     int M, int N, int K, int ldc, int relu, long long* stats = nullptr) {
  constexpr int MB = 3, NP = (MB + 8) * 3, NPW = (NP + 3) / 4;   // 33 pieces per k-step, <= 9 per wave
  constexpr int SLOT = NP * 64;                                 // uint4 per ring slot (33 KiB)
  constexpr int TLD = 260;                                      // floats per row of the epilogue tile
  __shared__ __attribute__((aligned(1024))) uint4 smem[4 * SLOT];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wq = wave & 3;
  const int KB = K >> 4, S = KB >> 1;
  const int RB = (M + 31) >> 5, CB = (N + 31) >> 5;
  const int nt = (N + 255) >> 8, mt = (RB + MB - 1) / MB;
  const int ntiles = nt * mt;
  int t;
  {
    const int b = blockIdx.x, x = b & 7, i = b >> 3, q = ntiles >> 3, r = ntiles & 7;
    t = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
  }
  const int tm = t / nt, tn = t - tm * nt;
  const int rb0 = tm * MB;
  const long long c0 = clock64(), w0 = wall_clock64();
  const uint4* gp[NPW];
#pragma unroll
  for (int i = 0; i < NPW; ++i) {
    const int p = min(wq + 4 * i, NP - 1), blk = p / 3, plane = p - blk * 3;
    const int64_t rowblk = blk < MB ? min(rb0 + blk, RB - 1) : min(tn * 8 + blk - MB, CB - 1);
    gp[i] = (blk < MB ? A : W) + rowblk * KB * 192 + plane * 64 + lane;
  }
  // ROT: every tile starts its k loop somewhere else (in whole stages), so that at any moment
  // the CUs of an XCD read different parts of W instead of all hitting the same L2 channels
  const int krot = ROT ? 2 * ((t * 5) % S) : 0;
  auto issue = [&](int kb, int slot) {      // this wave's share of k-step kb -> ring slot
    if (ABL & 1) return;
    int kbp = kb + krot;
    if (kbp >= KB) kbp -= KB;
#pragma unroll
    for (int i = 0; i < NPW; ++i)
      if (i * 4 + 3 < NP || wq + 4 * i < NP)
        glds16(gp[i] + kbp * 192, &smem[slot * SLOT + (wq + 4 * i) * 64]);
  };
  frag_t a[2][MB][3], b[2][3];
  if (ABL & 2) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        b[h][p].u = make_uint4(lane, h, p, 7);
#pragma unroll
        for (int m = 0; m < MB; ++m) a[h][m][p].u = make_uint4(lane, h, p, m);
      }
  }
  f32x16 acc[MB];
#pragma unroll
  for (int m = 0; m < MB; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
  const int bpiece = MB * 3 + (grp * 4 + wq) * 3;
  auto read = [&](int slot0) {              // fragments of the stage in ring slots slot0, slot0 + 1
    if (ABL & 2) return;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int p = 0; p < 3; ++p) a[h][m][p].u = smem[(slot0 + h) * SLOT + (m * 3 + p) * 64 + lane];
#pragma unroll
      for (int p = 0; p < 3; ++p) b[h][p].u = smem[(slot0 + h) * SLOT + (bpiece + p) * 64 + lane];
    }
  };
  auto mma = [&] {
    constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int q = 6 - NPROD; q < 6; ++q)
#pragma unroll
        for (int m = 0; m < MB; ++m) acc[m] = mfma(a[h][m][PA[q]], b[h][PB[q]], acc[m]);
  };
  auto bar = [&] {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  auto drain_bar = [&] {                    // this wave's LDS-DMA has landed, then the barrier
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0)
    bar();
  };
  // READ(s): fragments of stage s, then the LDS-DMA of k-step kq into ring slot kq & 3
  long long tq[5] = {0, 0, 0, 0, 0};
  auto READ = [&](int s, int kq) {
    if (ABL & 4) {
      const long long t0 = clock64();
      read((s & 1) * 2);
      __builtin_amdgcn_sched_barrier(0);
      const long long t1 = clock64();
      if (kq < KB) issue(kq, kq & 3);
      __builtin_amdgcn_sched_barrier(0);
      const long long t2 = clock64();
      __builtin_amdgcn_s_waitcnt(0x0F70);
      __builtin_amdgcn_sched_barrier(0);
      const long long t3 = clock64();
      bar();
      const long long t4 = clock64();
      tq[0] += t1 - t0; tq[1] += t2 - t1; tq[2] += t3 - t2; tq[3] += t4 - t3; tq[4] += 1;
      return;
    }
    read((s & 1) * 2);
    if (kq < KB) issue(kq, kq & 3);
    drain_bar();
  };
  issue(grp, grp);                          // k-steps 0 and 1
  drain_bar();
  if (grp == 0) {
    for (int s = 0; s < S; s += 2) {
      READ(s, 2 * s + 2);
      mma(); bar();
      if (s + 1 < S) {
        READ(s + 1, 2 * s + 4);
        mma(); bar();
      }
    }
  } else {
    bar();
    for (int s = 0; s < S; s += 2) {
      READ(s, 2 * s + 3);
      mma();
      if (s + 1 < S) {
        bar();
        READ(s + 1, 2 * s + 5);
        mma();
        if (s + 2 < S) bar();
      }
    }
  }
  const long long c1 = clock64();
  // ---- epilogue: the tile as fp32 rows in LDS (over the ring), then row-wise by all 8 waves ----
  float* T = (float*)smem;
  const int li = lane & 31, lh = lane >> 5;
  {
    const int col = (grp * 4 + wq) * 32 + li;
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) T[(m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * TLD + col] = acc[m][r];
  }
  __syncthreads();
  const int row0 = rb0 * 32, col0 = tn * 256;
  {
    // wave w: rows w, w + 8, ...; lane: 4 consecutive columns
    const int c = col0 + lane * 4;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias && c < N) bv = *(const float4*)(bias + c);     // N % 4 == 0
#pragma unroll 4
    for (int r = wave; r < MB * 32; r += 8) {
      float4 v = *(const float4*)(T + r * TLD + lane * 4);
      v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
      if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      if (EPI == 0) {
        if (row0 + r < M && c < N) *(float4*)(C + (int64_t)(row0 + r) * ldc + c) = v;
      } else {
        *(float4*)(T + r * TLD + lane * 4) = v;
      }
    }
  }
  if (EPI == 1) {
    __syncthreads();
    const int KBo = N >> 4;
    // (row block, 16-column block) pieces of the tile: 3 x 16 = 48, six per wave
#pragma unroll 2
    for (int q = wave; q < MB * 16; q += 8) {
      const int m = q >> 4, kb = q & 15;
      const int kbo = tn * 16 + kb;
      if (rb0 + m >= RB || kbo >= KBo) continue;
      const float* src = T + (m * 32 + li) * TLD + kb * 16 + lh * 8;
      const float4 x = *(const float4*)src, y = *(const float4*)(src + 4);
      const float v[8] = {x.x, x.y, x.z, x.w, y.x, y.y, y.z, y.w};
      frag_t p0, p1, p2;
#pragma unroll
      for (int i = 0; i < 8; ++i) split3(v[i], p0.e[i], p1.e[i], p2.e[i]);
      uint4* o = CS + ((int64_t)(rb0 + m) * KBo + kbo) * 192 + lane;
      o[0] = p0.u; o[64] = p1.u; o[128] = p2.u;
    }
  }
  if ((ABL & 4) && stats && tid == 0 && blockIdx.x == 100) {
    for (int i = 0; i < 5; ++i) stats[8192 * 3 + i] = tq[i];
  }
  if (stats && tid == 0) {
    stats[blockIdx.x * 3] = c1 - c0; stats[blockIdx.x * 3 + 1] = clock64() - c0;
    stats[blockIdx.x * 3 + 2] = wall_clock64() - w0;
  }
}


// ---------------------------------------------------------------------------------------------
// Ping-pong form 2: as k_gemm_s3p, but an LDS-DMA piece is waited for at the END OF THE ISSUING
// WAVE'S NEXT (MMA) PHASE, i.e. it has a whole phase to land, and the reading wave runs at raised
// priority while it issues.  That needs a piece to be issued two phases before its first read:
//   A ring: 3 stages (32-deep) x 2 halves x 9 pieces,  B ring: 2 stages x 2 halves x 24 pieces
//   G0 READ(s) (phase 2s-1) issues: its own B columns of stage s+1, A first half of stage s+1
//   G1 READ(s) (phase 2s)   issues: its own B columns of stage s+1, A second half of stage s+2
template <int NPROD, int EPI, int ABL = 0>
__global__ __launch_bounds__(512, 1) void k_gemm_s3q(const uint4* __restrict__ A,
                                                     const uint4* __restrict__ W,
                                                     const float* __restrict__ bias,
                                                     float* __restrict__ C, uint4* __restrict__ CS,
                                                     int M, int N, int K, int ldc, int relu, long long* stats = nullptr) {
  constexpr int MB = 3;
  constexpr int AH = 9 * 64, AST = 2 * AH;           // uint4 per A half / A stage
  constexpr int BH = 24 * 64, BST = 2 * BH;          // uint4 per B half / B stage
  constexpr int BRING = 3 * AST;                     // B ring starts here
  constexpr int TLD = 260;
  __shared__ __attribute__((aligned(1024))) uint4 smem[3 * AST + 2 * BST];   // 150 KiB
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wq = wave & 3;
  const int KB = K >> 4, S = KB >> 1;
  const int RB = (M + 31) >> 5, CB = (N + 31) >> 5;
  const int nt = (N + 255) >> 8, mt = (RB + MB - 1) / MB;
  const int ntiles = nt * mt;
  int t;
  {
    const int b = blockIdx.x, x = b & 7, i = b >> 3, q = ntiles >> 3, r = ntiles & 7;
    t = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
  }
  const int tm = t / nt, tn = t - tm * nt;
  const int rb0 = tm * MB;
  const long long c0 = clock64(), w0 = wall_clock64();
  // per wave and READ phase: 6 pieces of B (its own group's 12 column-block planes x 2 halves / 4
  // waves), 2-3 pieces of A.  B piece j of wave wq: index e = wq + 4 j in [0, 24): half = e / 12,
  // (column block, plane) = e % 12.  A piece: index e = wq + 4 j in [0, 9): (row block, plane).
  const uint4* gB[6]; int lB[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const int e = wq + 4 * j, half = e / 12, r = e - half * 12, cbl = r / 3, plane = r - cbl * 3;
    const int64_t cb = min(tn * 8 + grp * 4 + cbl, CB - 1);
    gB[j] = W + (cb * KB + half) * 192 + plane * 64 + lane;
    lB[j] = half * BH + ((grp * 4 + cbl) * 3 + plane) * 64;
  }
  const uint4* gA[3]; int lA[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int e = min(wq + 4 * j, 8), m = e / 3, plane = e - m * 3;
    gA[j] = A + (int64_t)min(rb0 + m, RB - 1) * KB * 192 + plane * 64 + lane;
    lA[j] = e * 64;
  }
  const int nA = wq == 0 ? 3 : 2;
  // B columns of this wave's group for stage s
  auto issueB = [&](int s) {
    if (ABL & 1) return;
    const int base = BRING + (s & 1) * BST;
#pragma unroll
    for (int j = 0; j < 6; ++j) glds16(gB[j] + (int64_t)s * 384, &smem[base + lB[j]]);
  };
  // one 16-deep half of A for stage s (as = s % 3)
  auto issueA = [&](int s, int as, int half) {
    if (ABL & 1) return;
    const int base = as * AST + half * AH;
#pragma unroll
    for (int j = 0; j < 3; ++j)
      if (j < 2 || nA == 3) glds16(gA[j] + (int64_t)(2 * s + half) * 192, &smem[base + lA[j]]);
  };
  frag_t a[2][MB][3], b[2][3];
  if (ABL & 2) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        b[h][p].u = make_uint4(lane, h, p, 7);
#pragma unroll
        for (int m = 0; m < MB; ++m) a[h][m][p].u = make_uint4(lane, h, p, m);
      }
  }
  f32x16 acc[MB];
#pragma unroll
  for (int m = 0; m < MB; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
  const int bfrag = (grp * 4 + wq) * 3 * 64 + lane;
  auto read = [&](int s, int as) {
    if (ABL & 2) return;
    const uint4* pa = smem + as * AST + lane;
    const uint4* pb = smem + BRING + (s & 1) * BST + bfrag;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int p = 0; p < 3; ++p) a[h][m][p].u = pa[h * AH + (m * 3 + p) * 64];
#pragma unroll
      for (int p = 0; p < 3; ++p) b[h][p].u = pb[h * BH + p * 64];
    }
  };
  auto mma = [&] {
    constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int q = 6 - NPROD; q < 6; ++q)
#pragma unroll
        for (int m = 0; m < MB; ++m) acc[m] = mfma(a[h][m][PA[q]], b[h][PB[q]], acc[m]);
  };
  auto bar = [&] {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  auto drain_bar = [&] {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0)
    bar();
  };
  // ---- prologue: stage 0 whole, A second half of stage 1 ----
  {
    // group g loads its own B columns of stage 0; A of stage 0: first half by G0, second by G1;
    // A second half of stage 1 by G1
    issueB(0);
    issueA(0, 0, grp);
    if (grp == 1 && S > 1) issueA(1, 1, 1);
    drain_bar();
  }
  if (grp == 0) {
    int as = 0;
    for (int s = 0; s < S; ++s) {
      const int as1 = as == 2 ? 0 : as + 1;
      __builtin_amdgcn_s_setprio(1);
      read(s, as);
      if (s + 1 < S) { issueB(s + 1); issueA(s + 1, as1, 0); }
      __builtin_amdgcn_s_setprio(0);
      bar();
      mma();
      drain_bar();
      as = as1;
    }
  } else {
    bar();
    int as = 0;
    for (int s = 0; s < S; ++s) {
      const int as2 = as == 0 ? 2 : as - 1;          // (s + 2) % 3
      __builtin_amdgcn_s_setprio(1);
      read(s, as);
      if (s + 1 < S) issueB(s + 1);
      if (s + 2 < S) issueA(s + 2, as2, 1);
      __builtin_amdgcn_s_setprio(0);
      bar();
      mma();
      if (s + 1 < S) drain_bar();
      as = as == 2 ? 0 : as + 1;
    }
  }
  const long long c1 = clock64();
  // ---- epilogue: the tile as fp32 rows in LDS (over the rings), then row-wise by all 8 waves ----
  __syncthreads();
  float* T = (float*)smem;
  const int li = lane & 31, lh = lane >> 5;
  {
    const int col = (grp * 4 + wq) * 32 + li;
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) T[(m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * TLD + col] = acc[m][r];
  }
  __syncthreads();
  const int row0 = rb0 * 32, col0 = tn * 256;
  {
    const int c = col0 + lane * 4;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias && c < N) bv = *(const float4*)(bias + c);
#pragma unroll 4
    for (int r = wave; r < MB * 32; r += 8) {
      float4 v = *(const float4*)(T + r * TLD + lane * 4);
      v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
      if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      if (EPI == 0) {
        if (row0 + r < M && c < N) *(float4*)(C + (int64_t)(row0 + r) * ldc + c) = v;
      } else {
        *(float4*)(T + r * TLD + lane * 4) = v;
      }
    }
  }
  if (EPI == 1) {
    __syncthreads();
    const int KBo = N >> 4;
#pragma unroll 2
    for (int q = wave; q < MB * 16; q += 8) {
      const int m = q >> 4, kb = q & 15;
      const int kbo = tn * 16 + kb;
      if (rb0 + m >= RB || kbo >= KBo) continue;
      const float* src = T + (m * 32 + li) * TLD + kb * 16 + lh * 8;
      const float4 x = *(const float4*)src, y = *(const float4*)(src + 4);
      const float v[8] = {x.x, x.y, x.z, x.w, y.x, y.y, y.z, y.w};
      frag_t p0, p1, p2;
#pragma unroll
      for (int i = 0; i < 8; ++i) split3(v[i], p0.e[i], p1.e[i], p2.e[i]);
      uint4* o = CS + ((int64_t)(rb0 + m) * KBo + kbo) * 192 + lane;
      o[0] = p0.u; o[64] = p1.u; o[128] = p2.u;
    }
  }
  if (stats && tid == 0) {
    stats[blockIdx.x * 3] = c1 - c0; stats[blockIdx.x * 3 + 1] = clock64() - c0;
    stats[blockIdx.x * 3 + 2] = wall_clock64() - w0;
  }
}

// ---------------------------------------------------------------- host
static inline float gauss(unsigned& s) {
  float a = 0.f;
  for (int i = 0; i < 4; ++i) { s = s * 1664525u + 1013904223u; a += (float)(s >> 8) / 16777216.f; }
  return (a - 2.f) * 1.7320508f;
}
static float bf16_to_f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

static float *dA, *dW, *dC, *dBias;
static uint4 *dAs, *dWs, *dCs;
static long long* dStats;
static std::vector<float> hA, hW, hBias;

template <typename F> static double time_us(F launch, int n = 30) {
  for (int i = 0; i < 3; ++i) launch();
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  for (int i = 0; i < n; ++i) launch();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  CK(hipGetLastError());
  return ms * 1e3 / n;
}
static size_t s3_units(int R, int K) { return (size_t)((R + 31) / 32) * (K / 16) * 192; }
static void split(const float* X, uint4* S, int R, int K) {
  const int64_t np = (int64_t)((R + 31) / 32) * (K / 16);
  k_s3_split<<<(unsigned)((np + 3) / 4), 256>>>(X, K, nullptr, 1, S, R, K);
}
// element (r, k) of an S3 buffer copied to the host
static double s3_get(const std::vector<uint16_t>& h, int K, int r, int k, int plane) {
  const size_t blk = ((size_t)(r / 32) * (K / 16) + k / 16) * 1536;
  return bf16_to_f(h[blk + plane * 512 + ((k % 16) / 8) * 256 + (r % 32) * 8 + k % 8]);
}

template <int MB, int NPROD, int ABL, int EPI, int NS = 2, int PRIO = 0>
static void run(const char* name, int M, int N, int K, int relu, bool verify) {
  const int RB = (M + 31) / 32, mt = (RB + MB - 1) / MB, nt = (N + 255) / 256;
  auto launch = [&] {
    k_gemm_s3<MB, NPROD, ABL, EPI, NS, PRIO><<<mt * nt, 256>>>(dAs, dWs, dBias, dC, dCs, M, N, K, N, relu);
  };
  const double us = time_us(launch);
  const double tf = 2.0 * M * N * K / us * 1e-6;
  printf("  %-44s %8.1f us  %6.1f TFLOP/s fp32-eq (%.2f of 157.3)  tiles %d", name, us, tf, tf / 157.3, mt * nt);
  if (verify) {
    CK(hipDeviceSynchronize());
    double maxe = 0, sse = 0; int cnt = 0;
    std::vector<float> hc; std::vector<uint16_t> hs;
    if (EPI == 0) { hc.resize((size_t)M * N); CK(hipMemcpy(hc.data(), dC, hc.size() * 4, hipMemcpyDeviceToHost)); }
    else { hs.resize(s3_units(M, N) * 8); CK(hipMemcpy(hs.data(), dCs, hs.size() * 2, hipMemcpyDeviceToHost)); }
    unsigned s = 7u;
    for (int it = 0; it < 20000; ++it) {
      s = s * 1664525u + 1013904223u; int r = (s >> 8) % M;
      s = s * 1664525u + 1013904223u; int c = (s >> 8) % N;
      if (it < 64) { r = it < 32 ? M - 1 - it : it - 32; }
      if (it >= 64 && it < 96) c = N - 1 - (it - 64);
      double ref = hBias[c];
      for (int k = 0; k < K; ++k) ref += (double)hA[(size_t)r * K + k] * (double)hW[(size_t)c * K + k];
      if (relu && ref < 0) ref = 0;
      double got;
      if (EPI == 0) got = hc[(size_t)r * N + c];
      else got = s3_get(hs, N, r, c, 0) + s3_get(hs, N, r, c, 1) + s3_get(hs, N, r, c, 2);
      const double e = fabs(got - ref);
      maxe = fmax(maxe, e); sse += e * e; ++cnt;
    }
    printf("   err vs fp64: max %.3e rms %.3e", maxe, sqrt(sse / cnt));
  }
  printf("\n");
}


template <int NPROD, int EPI, int ROT = 0, int ABL = 0, int KV = 0>
static void runp(const char* name, int M, int N, int K, int relu, bool verify) {
  const int RB = (M + 31) / 32, mt = (RB + 2) / 3, nt = (N + 255) / 256;
  auto launch = [&] {
    if (KV == 0) k_gemm_s3p<NPROD, EPI, ROT, ABL><<<mt * nt, 512>>>(dAs, dWs, dBias, dC, dCs, M, N, K, N, relu, dStats);
    else k_gemm_s3q<NPROD, EPI, ABL & 3><<<mt * nt, 512>>>(dAs, dWs, dBias, dC, dCs, M, N, K, N, relu, dStats);
  };
  CK(hipMemset(EPI == 0 ? (void*)dC : (void*)dCs, 0xFF, EPI == 0 ? (size_t)M * N * 4 : s3_units(M, N) * 16));
  const double us = time_us(launch);
  const double tf = 2.0 * M * N * K / us * 1e-6;
  printf("  %-44s %8.1f us  %6.1f TFLOP/s fp32-eq (%.2f of 157.3)  tiles %d", name, us, tf, tf / 157.3, mt * nt);
  {
    CK(hipDeviceSynchronize());
    std::vector<long long> st((size_t)mt * nt * 3);
    CK(hipMemcpy(st.data(), dStats, st.size() * 8, hipMemcpyDeviceToHost));
    double loop = 0, all = 0, wall = 0;
    for (int i = 0; i < mt * nt; ++i) { loop += st[i * 3]; all += st[i * 3 + 1]; wall += st[i * 3 + 2]; }
    const int phases = K / 16 + 1;
    if (ABL & 4) {
      long long tq[5];
      CK(hipMemcpy(tq, dStats + 8192 * 3, 40, hipMemcpyDeviceToHost));
      printf("\n      wave 0 of tile 100, per READ phase: fragment reads %.0f, LDS-DMA issue %.0f, vmcnt(0) %.0f, barrier %.0f cycles",
             (double)tq[0] / tq[4], (double)tq[1] / tq[4], (double)tq[2] / tq[4], (double)tq[3] / tq[4]);
    }
    printf("\n      per tile: main loop %.0f cycles (%.0f per phase), with epilogue %.0f cycles; shader clock %.2f GHz",
           loop / (mt * nt), loop / (mt * nt) / phases, all / (mt * nt), all / wall * 0.1);
  }
  if (verify) {
    CK(hipDeviceSynchronize());
    double maxe = 0, sse = 0; int cnt = 0;
    std::vector<float> hc; std::vector<uint16_t> hs;
    if (EPI == 0) { hc.resize((size_t)M * N); CK(hipMemcpy(hc.data(), dC, hc.size() * 4, hipMemcpyDeviceToHost)); }
    else { hs.resize(s3_units(M, N) * 8); CK(hipMemcpy(hs.data(), dCs, hs.size() * 2, hipMemcpyDeviceToHost)); }
    unsigned s = 7u;
    for (int it = 0; it < 20000; ++it) {
      s = s * 1664525u + 1013904223u; int r = (s >> 8) % M;
      s = s * 1664525u + 1013904223u; int c = (s >> 8) % N;
      if (it < 64) { r = it < 32 ? M - 1 - it : it - 32; }
      if (it >= 64 && it < 96) c = N - 1 - (it - 64);
      double ref = hBias[c];
      for (int k = 0; k < K; ++k) ref += (double)hA[(size_t)r * K + k] * (double)hW[(size_t)c * K + k];
      if (relu && ref < 0) ref = 0;
      double got;
      if (EPI == 0) got = hc[(size_t)r * N + c];
      else got = s3_get(hs, N, r, c, 0) + s3_get(hs, N, r, c, 1) + s3_get(hs, N, r, c, 2);
      const double e = fabs(got - ref);
      maxe = fmax(maxe, e); sse += e * e; ++cnt;
    }
    printf("\n      err vs fp64: max %.3e rms %.3e", maxe, sqrt(sse / cnt));
  }
  printf("\n");
}

int main(int argc, char** argv) {
  setvbuf(stdout, NULL, _IONBF, 0);
  const size_t maxA = (size_t)66816 * 1024, maxW = (size_t)1024 * 1024, maxC = (size_t)66816 * 1024;
  CK(hipMalloc(&dA, maxA * 4)); CK(hipMalloc(&dW, maxW * 4)); CK(hipMalloc(&dC, maxC * 4));
  CK(hipMalloc(&dAs, maxA * 6)); CK(hipMalloc(&dWs, maxW * 6)); CK(hipMalloc(&dCs, maxC * 6));
  CK(hipMalloc(&dBias, 4096 * 4)); CK(hipMalloc(&dStats, (8192 * 3 + 8) * 8));
  hA.resize(maxA); hW.resize(maxW); hBias.resize(4096);
  unsigned s = 1u;
  for (auto& v : hA) v = gauss(s);
  for (auto& v : hW) v = gauss(s) * 0.0625f;
  for (auto& v : hBias) v = gauss(s);
  CK(hipMemcpy(dA, hA.data(), maxA * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dW, hW.data(), maxW * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dBias, hBias.data(), 4096 * 4, hipMemcpyHostToDevice));
  const int shapes[][4] = {{21950, 1024, 256, 1}, {21950, 256, 1024, 0}, {21950, 544, 256, 0},
                           {21950, 256, 256, 0}, {66800, 256, 256, 0}, {16700, 512, 256, 0}};
  const int only = argc > 1 ? atoi(argv[1]) : -1;
  int si = -1;
  for (auto& sh : shapes) {
    if (++si != only && only >= 0) continue;
    const int M = sh[0], N = sh[1], K = sh[2], relu = sh[3];
    printf("M=%d N=%d K=%d relu=%d\n", M, N, K, relu);
    // operands as their producers would leave them
    split(dA, dAs, M, K); split(dW, dWs, N, K);
    CK(hipDeviceSynchronize());
    {  // the split itself: exact reconstruction on samples, and its cost
      std::vector<uint16_t> hs(s3_units(M, K) * 8);
      CK(hipMemcpy(hs.data(), dAs, hs.size() * 2, hipMemcpyDeviceToHost));
      double worst = 0; unsigned q = 3u;
      for (int it = 0; it < 100000; ++it) {
        q = q * 1664525u + 1013904223u; const int r = (q >> 8) % M;
        q = q * 1664525u + 1013904223u; const int k = (q >> 8) % K;
        const double x = hA[(size_t)r * K + k];
        const double y = s3_get(hs, K, r, k, 0) + s3_get(hs, K, r, k, 1) + s3_get(hs, K, r, k, 2);
        worst = fmax(worst, fabs(x - y) / fmax(fabs(x), 1e-30));
      }
      const double us = time_us([&] { split(dA, dAs, M, K); });
      printf("  split A: %.1f us (%.2f TB/s), worst relative reconstruction error %.3e\n", us,
             (double)M * K * 10 / us * 1e-6, worst);
    }
    runp<6, 0>("s3 ping-pong 96x256 x6 fp32 out", M, N, K, relu, true);
    runp<6, 0, 0, 0, 1>("s3 ping-pong 2, x6 fp32 out", M, N, K, relu, true);
    runp<6, 1, 0, 0, 1>("s3 ping-pong 2, x6 S3 out", M, N, K, relu, true);
    runp<1, 0, 0, 0, 1>("s3 ping-pong 2, x1", M, N, K, relu, false);
    runp<6, 0, 0, 1, 1>("s3 ping-pong 2, x6, no LDS-DMA in the loop", M, N, K, relu, false);
    runp<6, 0, 0, 2, 1>("s3 ping-pong 2, x6, no fragment reads", M, N, K, relu, false);
  }
  return 0;
}
