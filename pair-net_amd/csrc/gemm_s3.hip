// fp32 GEMMs on the bf16 matrix pipe from PRE-SPLIT operands ("S3" format).
//
// Arithmetic.  Every fp32 number is exactly the sum of three bf16 numbers, x = x0 + x1 + x2, each
// piece the round-to-nearest-even bf16 of what is left (3 x 8 significand bits, same exponent
// range as fp32).  A product of two bf16 numbers is exact in fp32, so a*b is the sum of nine exact
// piece products, of which the three smallest (a1 b2, a2 b1, a2 b2) are together below
// 2^-26 |ab| -- a quarter of the rounding unit of ONE fp32 product -- and are dropped.  The other
// six go through v_mfma_f32_32x32x16_bf16 (exact products, fp32 accumulation), smallest terms
// first.  Measured against fp64 the result is at or below the error of the fp32 MFMA
// (v_mfma_f32_32x32x2_f32) on every shape of the path (profiles/r06_gemm_s3_error.txt), at six
// sixteenths of its instruction time.  Replaces the encoder GEMMs of
// reference pairnet/models/relation_heads/pairnet_head.py:262 (mmdet
// MSDeformAttnPixelDecoder, cfg configs/mask2former/pairnet.py:33-71).
//
// S3 layout of an [R x K] operand (K % 16 == 0; rows padded to 32): blocks of 32 rows x 16 k,
// block (rb, kb) at byte ((rb * K/16 + kb) * 3072); inside a block three 1 KiB planes (x0, x1,
// x2); inside a plane the MFMA fragment order: 16 bytes (8 consecutive k) per lane, lane =
// (k % 16 / 8) * 32 + r % 32.  One wave-wide 16-byte load / LDS-DMA piece / ds_read_b128 / store
// moves one plane of one block: fully coalesced in HBM, lane-linear (conflict-free) in LDS, no
// swizzle anywhere, and the SAME layout serves as the MFMA's A and B operand.  Producers write
// it (this file's epilogues, k_s3_split), so no GEMM wave converts anything in its main loop.
//
// Kernel.  ONE workgroup of 8 waves per CU, tile 96 rows x 256 columns, whole rows owned by the
// workgroup (row epilogues: residual + LayerNorm).  Waves 0-3 (group 0) own columns 0..127, waves
// 4-7 (group 1) columns 128..255; a wave's tile is 96 x 32 = 3 accumulators.  The accumulators
// hold the TRANSPOSED tile (the weight fragment is the MFMA's A operand, the activation fragment
// its B operand): a lane owns ONE output row and 16 of its columns, so bias / residual /
// LayerNorm / splitting are register work and every output -- fp32 rows or S3 pieces -- leaves in
// 16-byte pieces without an LDS round trip (four v_permlane32_swap per 16 columns give each lane 8
// consecutive columns, which IS the S3 piece order).
// Time is cut into phases by workgroup barriers; in every phase one group issues the 36 MFMAs of
// a 32-deep k-stage from registers while the other group, on the same SIMDs and at raised
// priority, reads its fragments of its next stage (24 ds_read_b128) and issues LDS-DMA:
//   phase 2s-1: G0 READ(s); issues its own W columns of stage s+1 and A's first half of s+1
//   phase 2s  : G0 MMA(s)  | G1 READ(s); issues its own W columns of s+1 and A's second half of s+2
//   phase 2s+1:            | G1 MMA(s)
// A piece is waited for (vmcnt(0)) at the end of the issuing wave's NEXT phase (its MMA phase)
// and read at least one barrier later; a ring slot is overwritten at least one barrier after its
// last reader's phase.  Rings: A 3 stages x 18 KiB, W 2 stages x 48 KiB (150 KiB of LDS).
#include "common.h"
#include "s3_common.h"

__device__ __forceinline__ void s3_glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// The `+ pos` table in "P8" order: fp32, 8-column pieces in the S3 piece order --
// [row / 32][col / 16][col % 16 / 8][row % 32][8] -- so that the lanes of an epilogue wave read 2 KiB
// contiguous (row-major rows would be 32-byte pieces 1 KiB apart).  hip.pos8() builds it.
__device__ __forceinline__ int64_t pos8_offset(int row, int col, int N) {
  return ((((int64_t)(row >> 5) * (N >> 4) + (col >> 4)) * 2 + ((col >> 3) & 1)) * 32 + (row & 31)) * 8;
}

// ---------------------------------------------------------------- fp32 rows -> S3
// One wave per (row block, k block): lane (k half, row) reads 8 consecutive floats.
__global__ __launch_bounds__(256) void k_s3_split(const float* __restrict__ X, int64_t ld,
                                                  const float* __restrict__ add, int add_rows,
                                                  uint4* __restrict__ S, int R, int K) {
  const int KB = K >> 4;
  const int lane = threadIdx.x & 63;
  const int64_t piece = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t npieces = (int64_t)((R + 31) >> 5) * KB;
  if (piece >= npieces) return;
  const int rb = (int)(piece / KB), kb = (int)(piece - (int64_t)rb * KB);
  const int r = rb * 32 + (lane & 31), k = kb * 16 + (lane >> 5) * 8;
  float v[8];
  if (r < R) {
    const float4 a = ld4(X + (int64_t)r * ld + k), b = ld4(X + (int64_t)r * ld + k + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    if (add) {
      const float* p = add + (int64_t)(r % add_rows) * K + k;
      const float4 c = ld4(p), d = ld4(p + 4);
      v[0] += c.x; v[1] += c.y; v[2] += c.z; v[3] += c.w; v[4] += d.x; v[5] += d.y; v[6] += d.z; v[7] += d.w;
    }
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = 0.f;
  }
  s3_frag p0, p1, p2;
  s3_split8(v, p0, p1, p2);
  uint4* o = S + piece * 192 + lane;
  o[0] = p0.u; o[64] = p1.u; o[128] = p2.u;
}

// S3 -> fp32 rows (tests, and consumers that have not moved to the format)
__global__ __launch_bounds__(256) void k_s3_join(const uint4* __restrict__ S, float* __restrict__ X,
                                                 int64_t ld, int R, int K) {
  const int KB = K >> 4;
  const int lane = threadIdx.x & 63;
  const int64_t piece = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t npieces = (int64_t)((R + 31) >> 5) * KB;
  if (piece >= npieces) return;
  const int rb = (int)(piece / KB), kb = (int)(piece - (int64_t)rb * KB);
  const int r = rb * 32 + (lane & 31), k = kb * 16 + (lane >> 5) * 8;
  if (r >= R) return;
  const uint4* in = S + piece * 192 + lane;
  s3_frag p0, p1, p2;
  p0.u = in[0]; p1.u = in[64]; p2.u = in[128];
  float v[8];
  s3_join8(p0, p1, p2, v);
  st4(X + (int64_t)r * ld + k, make_float4(v[0], v[1], v[2], v[3]));
  st4(X + (int64_t)r * ld + k + 4, make_float4(v[4], v[5], v[6], v[7]));
}

// ---------------------------------------------------------------- the GEMM
struct s3_args {
  const uint4* A; const uint4* A2; int a2_from_tile;
  const uint4* W; const float* bias;
  float* C; int64_t ldc; uint4* CS; uint4* CSP;
  const uint4* RES; const float* gamma; const float* beta; const float* pos; int pos_rows; float eps;
  int M, N, K, relu;      // relu: activation 0 none, 1 ReLU, 2 exact (erf) GELU, 3 ReLU after the shortcut
  const float* res; int64_t ldres;   // plain epilogue: out = act(acc + bias) + res[m][n] (fp32 rows)
  int nout;      // columns of the S3 outputs' rows (N, or the whole row when N is a column range)
};

// One 32 x 32 accumulator block's plain epilogue (bias, ReLU, fp32 rows and / or S3 pieces): the
// accumulator is the TRANSPOSED block (lane l: output row rb * 32 + l % 32; register r: column
// n0 + (r & 3) + 8 (r >> 2) + 4 (l >> 5)); four v_permlane32_swap per 16 columns give each lane 8
// consecutive columns, the S3 piece order.
__device__ __forceinline__ void s3_block_epilogue(const f32x16& acc, const s3_args& p, int rb, int n0,
                                                  int lane) {
  const int li = lane & 31, lh = lane >> 5, row = rb * 32 + li;
  const int KBo = p.nout >> 4;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    float v[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[8 * j + i]),
                                                       __float_as_uint(acc[8 * j + 4 + i]), false, false);
      v[i] = __uint_as_float(sw[0]);
      v[4 + i] = __uint_as_float(sw[1]);
    }
    if (p.bias) {
      const float4 b0 = ld4(p.bias + n0 + 16 * j + 8 * lh), b1 = ld4(p.bias + n0 + 16 * j + 8 * lh + 4);
      v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
    }
    if (p.relu == 1) {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = fmaxf(v[i], 0.f);
    } else if (p.relu == 2) {            // [3P] nn.GELU (erf form), as pn_gemm_f32's
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = 0.5f * v[i] * (1.f + erff(v[i] * 0.70710678118654752f));
    }
    if (p.res) {                         // the block's shortcut (`x = x + proj(...)`)
      const float* rp = p.res + (int64_t)min(row, p.M - 1) * p.ldres + n0 + 16 * j + 8 * lh;
      const float4 r0 = ld4(rp), r1 = ld4(rp + 4);
      v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w; v[4] += r1.x; v[5] += r1.y; v[6] += r1.z; v[7] += r1.w;
    }
    if (p.relu == 3) {                   // ReLU AFTER the shortcut (a ResNet bottleneck's last conv)
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = fmaxf(v[i], 0.f);
    }
    if (p.C && row < p.M) {
      float* c = p.C + (int64_t)row * p.ldc + n0 + 16 * j + 8 * lh;
      st4(c, make_float4(v[0], v[1], v[2], v[3]));
      st4(c + 4, make_float4(v[4], v[5], v[6], v[7]));
    }
    if (p.CS) {
      s3_frag q0, q1, q2;
      s3_split8(v, q0, q1, q2);
      uint4* o = p.CS + ((int64_t)rb * KBo + (n0 >> 4) + j) * 192 + lane;
      o[0] = q0.u; o[64] = q1.u; o[128] = q2.u;
    }
  }
}

// LN: the row epilogue out = LayerNorm(acc + bias + residual) (N == 256, one column tile)
#ifndef S3_LATE_B
#define S3_LATE_B 0
#endif
template <bool LN>
__global__ __launch_bounds__(512, 1) void k_gemm_s3(const s3_args p) {
  constexpr int MB = 3, LATE_B = S3_LATE_B;   // (A/B knob: W pieces per wave issued in the MMA phase)
  constexpr int AH = 9 * 64, AST = 2 * AH;           // uint4 per A half (16-deep) / A stage
  constexpr int BH = 24 * 64, BST = 2 * BH;          // uint4 per W half / W stage
  constexpr int BRING = 3 * AST;
  __shared__ __attribute__((aligned(1024))) uint4 smem[3 * AST + 2 * BST];   // 150 KiB
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wq = wave & 3;
  const int M = p.M, N = p.N, KB = p.K >> 4, S = KB >> 1;
  const int RB = (M + 31) >> 5, CB = (N + 31) >> 5;
  const int nt = (N + 255) >> 8, mt = (RB + MB - 1) / MB;
  const int t = xcd_tile_index(blockIdx.x, nt * mt);
  const int tm = t / nt, tn = t - tm * nt;
  const int rb0 = tm * MB;
  const uint4* Aop = (p.A2 && tn >= p.a2_from_tile) ? p.A2 : p.A;

  // per wave and READ phase: 6 pieces of W (its group's 4 column blocks x 3 planes x 2 halves over
  // 4 waves) and 2-3 pieces of A (3 row blocks x 3 planes over 4 waves), bases wave-uniform
  const uint4* gB[6]; int lB[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const int e = wq + 4 * j, half = e / 12, r = e - half * 12, cbl = r / 3, plane = r - cbl * 3;
    const int64_t cb = min(tn * 8 + grp * 4 + cbl, CB - 1);
    gB[j] = p.W + (cb * KB + half) * 192 + plane * 64 + lane;
    lB[j] = half * BH + ((grp * 4 + cbl) * 3 + plane) * 64;
  }
  const uint4* gA[3]; int lA[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int e = min(wq + 4 * j, 8), m = e / 3, plane = e - m * 3;
    gA[j] = Aop + (int64_t)min(rb0 + m, RB - 1) * KB * 192 + plane * 64 + lane;
    lA[j] = e * 64;
  }
  const bool a3 = wq == 0;
  // (diagnostic builds for tools/coresidency_probe.py: -DS3_DIAG_NO_DMA drops the LDS-DMA of the
  // main loop, -DS3_DIAG_NO_MFMA the MFMAs; results are then meaningless)
  // this group's W columns of stage s: pieces [j0, j1) of this wave's six
  auto issueB = [&](int s, int j0 = 0, int j1 = 6) {
#ifdef S3_DIAG_NO_DMA
    if (s > 0) return;
#endif
    const int base = BRING + (s & 1) * BST;
#pragma unroll
    for (int j = 0; j < 6; ++j)
      if (j >= j0 && j < j1) s3_glds16(gB[j] + (int64_t)s * 384, &smem[base + lB[j]]);
  };
  auto issueA = [&](int s, int as, int half) {   // one 16-deep half of A's stage s into ring slot as
#ifdef S3_DIAG_NO_DMA
    if (s > 0) return;
#endif
    const int base = as * AST + half * AH;
#pragma unroll
    for (int j = 0; j < 3; ++j)
      if (j < 2 || a3) s3_glds16(gA[j] + (int64_t)(2 * s + half) * 192, &smem[base + lA[j]]);
  };
  s3_frag a[2][MB][3], b[2][3];
  f32x16 acc[MB];
#pragma unroll
  for (int m = 0; m < MB; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
  const int bfrag = (grp * 4 + wq) * 3 * 64 + lane;
  auto read = [&](int s, int as) {
    const uint4* pa = smem + as * AST + lane;
    const uint4* pb = smem + BRING + (s & 1) * BST + bfrag;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int q = 0; q < 3; ++q) a[h][m][q].u = pa[h * AH + (m * 3 + q) * 64];
#pragma unroll
      for (int q = 0; q < 3; ++q) b[h][q].u = pb[h * BH + q * 64];
    }
  };
  auto mma = [&](int late) {
    // smallest terms first; consecutive MFMAs go to different accumulators.  W is the MFMA's A
    // operand: acc[m] lane l, register r = output row (rb0 + m) * 32 + l % 32, column
    // n0 + (r & 3) + 8 (r >> 2) + 4 (l >> 5)
    constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
#ifdef S3_DIAG_NO_MFMA
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int q = 0; q < 3; ++q) acc[m][q] += __uint_as_float(b[h][q].u.x ^ a[h][m][q].u.y);
    return;
#endif
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int q = 0; q < 6; ++q) {
#pragma unroll
        for (int m = 0; m < MB; ++m)
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[h][PB[q]].v, a[h][m][PA[q]].v, acc[m], 0, 0, 0);
        // the LATE pieces of this group's W columns of stage `late` (two stages ahead: their
        // ring slot is this stage's, consumed in this wave's READ phase) ride between the
        // MFMAs, one per three of them: an LDS-DMA piece costs ~120 cycles of issue in a READ
        // phase beside the other group's MFMAs and about half of that here (labnotes R6.2)
        if (LATE_B > 0 && h == 0 && q < LATE_B && late >= 0) {
          __builtin_amdgcn_sched_barrier(0);
          issueB(late, 6 - LATE_B + q, 6 - LATE_B + q + 1);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
  };
  auto bar = [&] {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  // vmcnt(0) as a real s_waitcnt the compiler's counter model sees (it would otherwise wait for
  // the LDS-DMA it believes pending in front of every ds_read), then the barrier
  auto drain_bar = [&] {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    bar();
  };
  // ---- prologue: stage 0 whole (W columns by their group, A's halves one per group), and A's
  // second half of stage 1 ----
  issueB(0);
  if (LATE_B > 0 && S > 1) issueB(1, 6 - LATE_B, 6);     // (stage 1's late pieces have no MMA phase to ride in)
  issueA(0, 0, grp);
  if (grp == 1 && S > 1) issueA(1, 1, 1);
  drain_bar();
  if (grp == 0) {
    int as = 0;
    for (int s = 0; s < S; ++s) {
      const int as1 = as == 2 ? 0 : as + 1;
      __builtin_amdgcn_s_setprio(1);
      read(s, as);
      if (s + 1 < S) { issueB(s + 1, 0, 6 - LATE_B); issueA(s + 1, as1, 0); }
      __builtin_amdgcn_s_setprio(0);
      bar();
      mma(s + 2 < S ? s + 2 : -1);
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
      if (s + 1 < S) issueB(s + 1, 0, 6 - LATE_B);
      if (s + 2 < S) issueA(s + 2, as2, 1);
      __builtin_amdgcn_s_setprio(0);
      bar();
      mma(s + 2 < S ? s + 2 : -1);
      if (s + 1 < S) drain_bar();
      as = as == 2 ? 0 : as + 1;
    }
  }

  // ---- epilogue, in registers ----
  const int n0 = tn * 256 + wave * 32;               // this wave's 32 columns
  if (!LN) {
    if (n0 < N) {
#pragma unroll
      for (int m = 0; m < MB; ++m)
        if (rb0 + m < RB) s3_block_epilogue(acc[m], p, rb0 + m, n0, lane);
    }
    return;
  }
  const int li = lane & 31, lh = lane >> 5;
  const bool cols_ok = n0 < N;                       // N % 32 == 0: a wave is all in or all out
  // v[m][j][i]: row (rb0 + m) * 32 + li, column n0 + 16 j + 8 lh + i  (after the swaps)
  float v[MB][2][8];
#pragma unroll
  for (int m = 0; m < MB; ++m)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        // lower lanes hold columns 16j + {0-3, 8-11}, upper lanes 16j + {4-7, 12-15}: exchange the
        // lower lanes' second quad with the upper lanes' first quad
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[m][8 * j + i]),
                                                         __float_as_uint(acc[m][8 * j + 4 + i]), false, false);
        v[m][j][i] = __uint_as_float(sw[0]);
        v[m][j][4 + i] = __uint_as_float(sw[1]);
      }
  if (p.bias && cols_ok) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float4 b0 = ld4(p.bias + n0 + 16 * j + 8 * lh), b1 = ld4(p.bias + n0 + 16 * j + 8 * lh + 4);
      const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
      for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int i = 0; i < 8; ++i) v[m][j][i] += bb[i];
    }
  }
  if (p.relu) {
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) v[m][j][i] = fmaxf(v[m][j][i], 0.f);
  }
  if (LN) {
#pragma clang fp contract(off)
    // residual: the exact fp32 values of an S3 operand [M x 256] (three planes summed)
    if (p.RES) {
#pragma unroll
      for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const uint4* in = p.RES + ((int64_t)min(rb0 + m, RB - 1) * 16 + wave * 2 + j) * 192 + lane;
          s3_frag q0, q1, q2;
          q0.u = in[0]; q1.u = in[64]; q2.u = in[128];
          float r[8];
          s3_join8(q0, q1, q2, r);
#pragma unroll
          for (int i = 0; i < 8; ++i) v[m][j][i] += r[i];
        }
    }
    // two-pass moments over the 256 columns of a row: 16 values in this lane, 16 in lane ^ 32, the
    // rest in the other 7 waves (through LDS, over the rings: every ring read is behind a barrier)
    float* red = (float*)smem;                       // [2][8 waves][96 rows]
    __syncthreads();
    float s1[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) {
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) s += v[m][j][i];
      s += __shfl_xor(s, 32, 64);
      if (lh == 0) red[wave * 96 + m * 32 + li] = s;
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < MB; ++m) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) s += red[w * 96 + m * 32 + li];
      s1[m] = s * (1.f / 256.f);
    }
#pragma unroll
    for (int m = 0; m < MB; ++m) {
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) { v[m][j][i] -= s1[m]; s += v[m][j][i] * v[m][j][i]; }
      s += __shfl_xor(s, 32, 64);
      if (lh == 0) red[768 + wave * 96 + m * 32 + li] = s;
    }
    __syncthreads();
    float g[2][8], be[2][8];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float4 g0 = ld4(p.gamma + n0 + 16 * j + 8 * lh), g1 = ld4(p.gamma + n0 + 16 * j + 8 * lh + 4);
      const float4 e0 = ld4(p.beta + n0 + 16 * j + 8 * lh), e1 = ld4(p.beta + n0 + 16 * j + 8 * lh + 4);
      g[j][0] = g0.x; g[j][1] = g0.y; g[j][2] = g0.z; g[j][3] = g0.w; g[j][4] = g1.x; g[j][5] = g1.y; g[j][6] = g1.z; g[j][7] = g1.w;
      be[j][0] = e0.x; be[j][1] = e0.y; be[j][2] = e0.z; be[j][3] = e0.w; be[j][4] = e1.x; be[j][5] = e1.y; be[j][6] = e1.z; be[j][7] = e1.w;
    }
#pragma unroll
    for (int m = 0; m < MB; ++m) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) s += red[768 + w * 96 + m * 32 + li];
      const float rstd = 1.f / sqrtf(s * (1.f / 256.f) + p.eps);
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) v[m][j][i] = (v[m][j][i] * rstd) * g[j][i] + be[j][i];
    }
  }
  if (!cols_ok) return;
  const int KBo = p.nout >> 4;
#pragma unroll
  for (int m = 0; m < MB; ++m) {
    if (rb0 + m >= RB) break;
    const int row = (rb0 + m) * 32 + li;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (p.C && row < M) {
        float* c = p.C + (int64_t)row * p.ldc + n0 + 16 * j + 8 * lh;
        st4(c, make_float4(v[m][j][0], v[m][j][1], v[m][j][2], v[m][j][3]));
        st4(c + 4, make_float4(v[m][j][4], v[m][j][5], v[m][j][6], v[m][j][7]));
      }
      const int64_t piece = ((int64_t)(rb0 + m) * KBo + (n0 >> 4) + j) * 192 + lane;
      if (p.CS) {
        s3_frag q0, q1, q2;
        s3_split8(v[m][j], q0, q1, q2);
        uint4* o = p.CS + piece;
        o[0] = q0.u; o[64] = q1.u; o[128] = q2.u;
      }
      if (p.CSP) {
        // the next layer's query operand: split(out + pos[row % pos_rows])
        const float* pp = p.pos + pos8_offset(min(row, M - 1) % p.pos_rows, n0 + 16 * j + 8 * lh, p.nout);
        const float4 p0 = ld4(pp), p1 = ld4(pp + 4);
        const float w[8] = {v[m][j][0] + p0.x, v[m][j][1] + p0.y, v[m][j][2] + p0.z, v[m][j][3] + p0.w,
                            v[m][j][4] + p1.x, v[m][j][5] + p1.y, v[m][j][6] + p1.z, v[m][j][7] + p1.w};
        s3_frag q0, q1, q2;
        s3_split8(w, q0, q1, q2);
        uint4* o = p.CSP + piece;
        o[0] = q0.u; o[64] = q1.u; o[128] = q2.u;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// The same ping-pong for shapes with at least two column tiles (N >= 512: FFN-1, the [value |
// offsets | logits] projection): tile 192 rows x 256 columns, the two groups own the upper and
// the lower 96 rows and SHARE W, each wave 96 x 64 (six accumulators).  42 KiB of operands per
// 16-deep k-step feed both groups' MMA phases -- 21 LDS-DMA pieces per phase instead of the 33 of
// the 96 x 256 tile, whose L2 -> LDS traffic is what bounds it (labnotes R6.2).  Phases are one
// 16-deep k-step (36 MFMAs) each:
//   phase 2j-1: G0 READ(j); issues its A rows of step j+1 and W's first 4 column blocks of j+1
//   phase 2j  : G0 MMA(j)  | G1 READ(j); issues its A rows of j+1 and W's last 4 column blocks of j+2
//   phase 2j+1:            | G1 MMA(j)
// Rings: W 3 steps x 24 KiB, A 2 groups x 2 steps x 9 KiB (108 KiB).  Waits and barriers as in
// k_gemm_s3: a piece is waited for at the end of its issuer's next phase and read a barrier later.
__global__ __launch_bounds__(512, 1) void k_gemm_s3_wide(const s3_args p) {
  constexpr int MB = 3;
  constexpr int AS = 9 * 64, WS = 24 * 64;            // uint4 per A step (one group) / W step
  constexpr int WRING = 4 * AS;                       // [A: group][slot] first, then W [3]
  __shared__ __attribute__((aligned(1024))) uint4 smem[4 * AS + 3 * WS];    // 108 KiB
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wq = wave & 3;
  const int KB = p.K >> 4;
  const int RB = (p.M + 31) >> 5, CB = (p.N + 31) >> 5;
  const int nt = (p.N + 255) >> 8, mt = (RB + 2 * MB - 1) / (2 * MB);
  const int t = xcd_tile_index(blockIdx.x, nt * mt);
  const int tm = t / nt, tn = t - tm * nt;
  const int rb0 = tm * 2 * MB + grp * MB;             // this group's first row block
  const uint4* Aop = (p.A2 && tn >= p.a2_from_tile) ? p.A2 : p.A;
  // A pieces of this wave: e = wq + 4 i < 9 -> (row block, plane); W pieces: this group's half
  // (column blocks 4 grp .. 4 grp + 3) x 3 planes = 12, three per wave
  const uint4* gA[3]; int lA[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int e = min(wq + 4 * i, 8), m = e / 3, plane = e - m * 3;
    gA[i] = Aop + (int64_t)min(rb0 + m, RB - 1) * KB * 192 + plane * 64 + lane;
    lA[i] = e * 64;
  }
  const bool a3 = wq == 0;
  const uint4* gW[3]; int lW[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int e = grp * 12 + wq + 4 * i, cbl = e / 3, plane = e - cbl * 3;
    gW[i] = p.W + (int64_t)min(tn * 8 + cbl, CB - 1) * KB * 192 + plane * 64 + lane;
    lW[i] = e * 64;
  }
  auto issueA = [&](int j) {
    const int base = (grp * 2 + (j & 1)) * AS;
#pragma unroll
    for (int i = 0; i < 3; ++i)
      if (i < 2 || a3) s3_glds16(gA[i] + (int64_t)j * 192, &smem[base + lA[i]]);
  };
  auto issueW = [&](int j, int slot) {     // this group's half of W's step j
    const int base = WRING + slot * WS;
#pragma unroll
    for (int i = 0; i < 3; ++i) s3_glds16(gW[i] + (int64_t)j * 192, &smem[base + lW[i]]);
  };
  s3_frag a[MB][3], b[2][3];
  f32x16 acc[MB][2];
#pragma unroll
  for (int m = 0; m < MB; ++m)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][c][r] = 0.f;
  auto read = [&](int j, int slot) {
    const uint4* pa = smem + (grp * 2 + (j & 1)) * AS + lane;
    const uint4* pb = smem + WRING + slot * WS + wq * 6 * 64 + lane;
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
      for (int q = 0; q < 3; ++q) a[m][q].u = pa[(m * 3 + q) * 64];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int q = 0; q < 3; ++q) b[c][q].u = pb[(c * 3 + q) * 64];
  };
  auto mma = [&] {
    constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
      for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int c = 0; c < 2; ++c)
          acc[m][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[c][PB[q]].v, a[m][PA[q]].v, acc[m][c], 0, 0, 0);
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
  // prologue: step 0 whole (each group its A rows and its half of W), W's second half of step 1
  issueA(0);
  issueW(0, 0);
  if (grp == 1 && KB > 1) issueW(1, 1);
  drain_bar();
  if (grp == 0) {
    int ws = 0;                              // j % 3
    for (int j = 0; j < KB; ++j) {
      const int ws1 = ws == 2 ? 0 : ws + 1;
      __builtin_amdgcn_s_setprio(1);
      read(j, ws);
      if (j + 1 < KB) { issueA(j + 1); issueW(j + 1, ws1); }
      __builtin_amdgcn_s_setprio(0);
      bar();
      mma();
      drain_bar();
      ws = ws1;
    }
  } else {
    bar();
    int ws = 0;
    for (int j = 0; j < KB; ++j) {
      const int ws2 = ws == 0 ? 2 : ws - 1;  // (j + 2) % 3
      __builtin_amdgcn_s_setprio(1);
      read(j, ws);
      if (j + 1 < KB) issueA(j + 1);
      if (j + 2 < KB) issueW(j + 2, ws2);
      __builtin_amdgcn_s_setprio(0);
      bar();
      mma();
      if (j + 1 < KB) drain_bar();
      ws = ws == 2 ? 0 : ws + 1;
    }
  }
#pragma unroll
  for (int m = 0; m < MB; ++m) {
    if (rb0 + m >= RB) break;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int n0 = tn * 256 + wq * 64 + c * 32;
      if (n0 < p.N) s3_block_epilogue(acc[m][c], p, rb0 + m, n0, lane);
    }
  }
}

// Leftover columns (N % 256 of at most 64, e.g. the last 32 of the encoder's 544-column
// [value | offsets | logits] projection): one wave per (row block, column block), operands straight
// to registers -- nothing is shared, 0.4 GFLOP, latency-bound and short.
__global__ __launch_bounds__(256) void k_gemm_s3_narrow(const s3_args p, const int col0) {
  const int lane = threadIdx.x & 63;
  const int KB = p.K >> 4, RB = (p.M + 31) >> 5, ncb = (p.N - col0) >> 5;
  const int unit = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (unit >= RB * ncb) return;
  const int rb = unit / ncb, cb = (col0 >> 5) + unit - rb * ncb;
  const uint4* ga = ((p.A2 && (cb >> 3) >= p.a2_from_tile) ? p.A2 : p.A) + (int64_t)rb * KB * 192 + lane;
  const uint4* gw = p.W + (int64_t)cb * KB * 192 + lane;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
  // nothing hides a load here but the wave's own earlier loads: four k-blocks (24 loads) in flight
  constexpr int D = 4;
  s3_frag a[D][3], b[D][3];
  auto load = [&](int kb, int slot) {
#pragma unroll
    for (int q = 0; q < 3; ++q) { a[slot][q].u = ga[kb * 192 + q * 64]; b[slot][q].u = gw[kb * 192 + q * 64]; }
  };
#pragma unroll
  for (int d = 0; d < D; ++d)
    if (d < KB) load(d, d);
  for (int kb = 0; kb < KB; kb += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      if (kb + d >= KB) break;
#pragma unroll
      for (int q = 0; q < 6; ++q)
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[d][PB[q]].v, a[d][PA[q]].v, acc, 0, 0, 0);
      if (kb + d + D < KB) load(kb + d + D, d);
    }
  }
  s3_block_epilogue(acc, p, rb, cb * 32, lane);
}

// ---------------------------------------------------------------- C ABI
extern "C" int64_t pn_s3_bytes(int rows, int K) {
  return (int64_t)((rows + 31) / 32) * (K / 16) * 3072;
}

extern "C" int pn_s3_split_f32(const float* X, int64_t ld, const float* add, int add_rows, void* S,
                               int rows, int K, void* stream) {
  if (!X || !S || rows <= 0 || K <= 0 || K % 16 || ld % 4 || (add && add_rows <= 0)) return PN_BAD_ARG;
  const int64_t np = (int64_t)((rows + 31) / 32) * (K / 16);
  k_s3_split<<<dim3((unsigned)((np + 3) / 4)), dim3(256), 0, (hipStream_t)stream>>>(
      X, ld, add, add ? add_rows : 1, (uint4*)S, rows, K);
  return PN_LAUNCH_CHECK();
}

extern "C" int pn_s3_join_f32(const void* S, float* X, int64_t ld, int rows, int K, void* stream) {
  if (!X || !S || rows <= 0 || K <= 0 || K % 16 || ld % 4) return PN_BAD_ARG;
  const int64_t np = (int64_t)((rows + 31) / 32) * (K / 16);
  k_s3_join<<<dim3((unsigned)((np + 3) / 4)), dim3(256), 0, (hipStream_t)stream>>>(
      (const uint4*)S, X, ld, rows, K);
  return PN_LAUNCH_CHECK();
}

extern "C" int pn_gemm_s3_f32(const pn_gemm_s3_desc* d, void* stream) {
  if (!d || !d->A || !d->W || d->M <= 0 || d->N <= 0 || d->K <= 0) return PN_BAD_ARG;
  if (d->K % 32 || d->N % 32) return PN_BAD_ARG;
  if (!d->C && !d->CS && !d->CS_pos) return PN_BAD_ARG;
  if (d->C && (d->ldc % 4 || d->ldc < d->N)) return PN_BAD_ARG;
  if (d->A2 && (d->a2_from_col <= 0 || d->a2_from_col % 256)) return PN_BAD_ARG;
  if (d->CS_pos && (!d->pos || d->pos_rows <= 0)) return PN_BAD_ARG;
  const bool ln = d->gamma != nullptr;
  if (ln && (d->N != 256 || !d->beta || d->relu)) return PN_BAD_ARG;
  if (!ln && d->res_s3) return PN_BAD_ARG;
  if (d->res && (ln || d->ldres % 4 || d->ldres < d->N)) return PN_BAD_ARG;
  if (d->act < 0 || d->act > 3 || (ln && d->act)) return PN_BAD_ARG;
  s3_args a;
  a.A = (const uint4*)d->A; a.A2 = (const uint4*)d->A2; a.a2_from_tile = d->A2 ? d->a2_from_col / 256 : 0;
  a.W = (const uint4*)d->W; a.bias = d->bias;
  a.C = d->C; a.ldc = d->ldc; a.CS = (uint4*)d->CS; a.CSP = (uint4*)d->CS_pos;
  a.RES = (const uint4*)d->res_s3; a.gamma = d->gamma; a.beta = d->beta; a.pos = d->pos;
  a.pos_rows = d->pos_rows; a.eps = d->eps;
  a.M = d->M; a.N = d->N; a.K = d->K; a.relu = d->act ? d->act : d->relu; a.nout = d->N;
  a.res = d->res; a.ldres = d->ldres;
  // columns beyond the last whole 256-column tile: at most 64 of them go to the narrow kernel
  // (a whole tile for 32 columns would run 7 of its 8 waves empty)
  const int rem = d->N % 256;
  const bool narrow = !ln && !d->CS_pos && d->N > 256 && rem > 0 && rem <= 64;
  const int RB = (d->M + 31) / 32, mt = (RB + 2) / 3;
  // two or more column tiles: the 192-row tile whose row groups share W -- unless its tile count
  // quantises worse on the chip's CUs than the 96-row tile's (rounds x phases x measured cycles
  // per phase: 1 300 with shared W, 1 550 without; labnotes R6.2)
  bool wide = !ln && !d->CS_pos && (d->N - (narrow ? rem : 0)) >= 512 && !(d->flags & PN_GEMM_S3_TILE96);
  if (wide && !(d->flags & PN_GEMM_S3_TILE192)) {
    static int cus = 0;
    if (!cus) {
      int dev = 0;
      if (hipGetDevice(&dev) != hipSuccess ||
          hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
        cus = 256;
    }
    const int64_t ntm = (d->N - (narrow ? rem : 0)) / 256, kb = d->K / 16;
    const int64_t rb = (d->M + 31) / 32;
    const int64_t t_wide = (((rb + 5) / 6 * ntm + cus - 1) / cus) * (2 * kb + 1) * 1300;
    const int64_t t_96 = (((rb + 2) / 3 * ntm + cus - 1) / cus) * (kb + 1) * 1550;
    wide = t_wide <= t_96;
  }
  if (narrow) {
    a.N = d->N - rem;                       // the main launch sees only the whole tiles ...
    const int nt = a.N / 256;
    if (wide) k_gemm_s3_wide<<<dim3((RB + 5) / 6 * nt), dim3(512), 0, (hipStream_t)stream>>>(a);
    else k_gemm_s3<false><<<dim3(mt * nt), dim3(512), 0, (hipStream_t)stream>>>(a);
    a.N = d->N;                             // ... S3 outputs keep the full row pitch
    s3_args b = a;
    const int units = RB * (rem / 32);
    k_gemm_s3_narrow<<<dim3((units + 3) / 4), dim3(256), 0, (hipStream_t)stream>>>(b, d->N - rem);
    return PN_LAUNCH_CHECK();
  }
  const int nt = (d->N + 255) / 256;
  if (wide) {
    k_gemm_s3_wide<<<dim3((RB + 5) / 6 * nt), dim3(512), 0, (hipStream_t)stream>>>(a);
    return PN_LAUNCH_CHECK();
  }
  if (ln) k_gemm_s3<true><<<dim3(mt * nt), dim3(512), 0, (hipStream_t)stream>>>(a);
  else k_gemm_s3<false><<<dim3(mt * nt), dim3(512), 0, (hipStream_t)stream>>>(a);
  return PN_LAUNCH_CHECK();
}
