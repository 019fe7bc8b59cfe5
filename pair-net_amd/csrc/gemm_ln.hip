// y = LayerNorm(res + x W^T + b) * gamma + beta for N == 256 output columns: the linear
// layer, the residual add and the post-norm of a transformer layer in ONE launch, for the
// chip-filling token counts of the pixel-decoder encoder (M = 21 950 per 800x1333 image).
//
// Why a kernel of its own: with N = 256 one workgroup can own whole output rows, so the
// LayerNorm moments never leave the chip -- the stand-alone pair (k_gemm_tile, then
// k_layernorm256) writes the 22.5 MB pre-norm map and reads it straight back, twelve times
// per image.  At M ~ 100 (the decoders' query side) a row-owning tile is 4 workgroups and
// loses (LABNOTES.md 6.0); this kernel is only launched for M >= 2048.
//
// Tile: 32 rows x 256 columns per workgroup, 4 waves, wave w owns columns [64 w, 64 w + 64)
// as two 32x32 fp32 MFMA accumulators that share every A fragment.  32-deep k-chunks are
// staged global -> registers -> LDS exactly like k_gemm_tile (same [row][36] layout, same
// k-permutation inside each 8-deep step), so every accumulator sees its products in the SAME
// order: the pre-norm values are bit for bit those of pn_gemm_f32, and the normalisation is
// k_layernorm256's arithmetic (one wave per row, a float4 per lane, two-pass moments) on a
// tile transposed through LDS -- the result equals the unfused pair bitwise.
#include "common.h"

#define RLN_BM 32
#define RLN_TLD 264   // transposed-tile row stride (floats): 4 rows apart = 32 banks apart
// Operand image in LDS: rows padded to 36 floats, as in k_gemm_tile.  288 rows x 36 floats are
// 41 472 B -- 512 B more than a quarter of the CU's 160 KB, i.e. three workgroups per CU.
// -DRLN_SWIZZLE=1 (measured in round 4, NOT adopted) stores rows unpadded (32 floats) with the
// float4 column XOR-ed by (row & 7) -- as conflict-free as the padding for the eight lanes of
// a ds_read_b128 phase and for a staging thread row -- which makes the image 36 864 B and FOUR
// workgroups fit.  The 686 row tiles of an 800x1333 image are all resident at three per CU
// already, so the occupancy buys nothing stand-alone (K = 1024: 111.5 vs 108.6 us, the XOR-ed
// addresses cost two VALU instructions per fragment), and under the pipeline it measured
// 207.6-208.0 against 208.6-208.8 images/s (three alternating runs, tools/ab_rln.sh).
#ifndef RLN_SWIZZLE
#define RLN_SWIZZLE 0
#endif
#if RLN_SWIZZLE
#define RLN_LD 32
#else
#define RLN_LD 36
#endif

__global__ __launch_bounds__(256, RLN_SWIZZLE ? 4 : 3) void k_gemm_rowln(
    const float* __restrict__ A, const int64_t lda, const float* __restrict__ W,
    const int64_t ldw, const float* __restrict__ bias, const float* __restrict__ Res,
    const int64_t ldres, const float* __restrict__ gamma, const float* __restrict__ beta,
    float* __restrict__ Y, const int64_t ldy, const int M, const int K, const float eps) {
  constexpr int kImage = (RLN_BM + 256) * RLN_LD, kT = RLN_BM * RLN_TLD;
  __shared__ __attribute__((aligned(16))) float smem[kImage > kT ? kImage : kT];
  float* const sA = smem;
  float* const sB = smem + RLN_BM * RLN_LD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int kc = (tid & 7) * 4, lrow = tid >> 3;
  const int m0 = blockIdx.x * RLN_BM;

  // loader: one float4 of A and eight of W per thread and chunk; addresses are a buffer
  // descriptor + a per-lane byte offset fixed for the tile + the chunk's scalar k offset
  const __amdgpu_buffer_rsrc_t rA = make_rsrc(A), rW = make_rsrc(W);
  const unsigned a_off = ((unsigned)min(m0 + lrow, M - 1) * (unsigned)lda + kc) * 4u;
  unsigned w_off[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) w_off[j] = ((unsigned)(lrow + 32 * j) * (unsigned)ldw + kc) * 4u;

  float4 ra, rb[8];
  auto load_chunk = [&](int kt) {
    ra = buf_ld4(rA, a_off, kt * 32 * 4);
#pragma unroll
    for (int j = 0; j < 8; ++j) rb[j] = buf_ld4(rW, w_off[j], kt * 32 * 4);
  };
#if RLN_SWIZZLE
  const int skc = ((tid & 7) ^ (lrow & 7)) * 4;       // (row + 32 j) & 7 == lrow & 7
#else
  const int skc = kc;
#endif
  auto store_chunk = [&]() {
    st4(sA + lrow * RLN_LD + skc, ra);
#pragma unroll
    for (int j = 0; j < 8; ++j) st4(sB + (lrow + 32 * j) * RLN_LD + skc, rb[j]);
  };

  f32x16 acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
#if RLN_SWIZZLE
  // float4 column (2 kb + lh) of row r sits at column (2 kb + lh) ^ (r & 7) = 2 kb ^ sx with
  // sx = lh ^ (li & 7) (every row this lane reads has r & 7 == li & 7)
  const int sx = lh ^ (li & 7);
  const float* fA = sA + li * RLN_LD;
  const float* fB = sB + (wave * 64 + li) * RLN_LD;
  int koff[4];
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) koff[kb] = ((2 * kb) ^ sx) * 4;
#else
  const float* fA = sA + li * RLN_LD + 4 * lh;
  const float* fB = sB + (wave * 64 + li) * RLN_LD + 4 * lh;
  const int koff[4] = {0, 8, 16, 24};
#endif
  auto compute = [&]() {
    float4 fa[2], fb0[2], fb1[2];
    fa[0] = ld4(fA + koff[0]); fb0[0] = ld4(fB + koff[0]); fb1[0] = ld4(fB + 32 * RLN_LD + koff[0]);
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      const int cur = kb & 1;
      if (kb + 1 < 4) {
        fa[cur ^ 1] = ld4(fA + koff[kb + 1]);
        fb0[cur ^ 1] = ld4(fB + koff[kb + 1]);
        fb1[cur ^ 1] = ld4(fB + 32 * RLN_LD + koff[kb + 1]);
      }
      const float4 av = fa[cur], b0 = fb0[cur], b1 = fb1[cur];
      acc0 = mfma32(av.x, b0.x, acc0); acc1 = mfma32(av.x, b1.x, acc1);
      acc0 = mfma32(av.y, b0.y, acc0); acc1 = mfma32(av.y, b1.y, acc1);
      acc0 = mfma32(av.z, b0.z, acc0); acc1 = mfma32(av.z, b1.z, acc1);
      acc0 = mfma32(av.w, b0.w, acc0); acc1 = mfma32(av.w, b1.w, acc1);
    }
  };

  const int nk = K / 32;
  load_chunk(0);
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();
    store_chunk();
    __syncthreads();
    if (kt + 1 < nk) load_chunk(kt + 1);
    compute();
  }

  // ---- epilogue: (acc + bias) -> LDS [32][256] (stride 264), then one wave per row ----
  // The residual and LayerNorm rows are read / written as whole 1 KB lines; all of a
  // wave's residual loads are issued before the first is used.
  const int r0 = wave * 8;
  float4 rs[8];
#pragma unroll
  for (int j = 0; j < 8; ++j)
    rs[j] = ld4(Res + (int64_t)min(m0 + r0 + j, M - 1) * ldres + lane * 4);
  const float4 gg = ld4(gamma + lane * 4), bb = ld4(beta + lane * 4);
  __syncthreads();                       // every wave has read its last fragments
  float* const T = smem;
  {
    const int c0 = wave * 64 + li;
    const float bv0 = bias ? bias[c0] : 0.f, bv1 = bias ? bias[c0 + 32] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = mfma32_row(r, lh);
      T[row * RLN_TLD + c0] = acc0[r] + bv0;
      T[row * RLN_TLD + c0 + 32] = acc1[r] + bv1;
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int grow = m0 + r0 + j;
    const float4 t = ld4(T + (r0 + j) * RLN_TLD + lane * 4);
    const float4 v = add4(t, rs[j]);
    if (grow < M) st4(Y + (int64_t)grow * ldy + lane * 4, ln256_row(v, gg, bb, eps));
  }
}

extern "C" int pn_linear_res_ln_f32(const float* x, int64_t ldx, const float* W, int64_t ldw,
                                    const float* bias, const float* res, int64_t ldres,
                                    const float* gamma, const float* beta, float* y,
                                    int64_t ldy, int M, int N, int K, float eps, void* stream) {
  if (!x || !W || !res || !gamma || !beta || !y || M <= 0) return PN_BAD_ARG;
  if (N != 256 || K <= 0 || K % 32) return PN_BAD_ARG;
  if (ldx % 4 || ldw % 4 || ldres % 4 || ldy % 4) return PN_BAD_ARG;
  if (ldx < K || ldw < K || ldres < 256 || ldy < 256 || y == x) return PN_BAD_ARG;
  if (((uintptr_t)x | (uintptr_t)W | (uintptr_t)res | (uintptr_t)gamma | (uintptr_t)beta |
       (uintptr_t)y) & 15)
    return PN_BAD_ARG;
  if ((int64_t)(M - 1) * ldx + K >= ((int64_t)1 << 29) ||
      (int64_t)255 * ldw + K >= ((int64_t)1 << 29))
    return PN_BAD_ARG;                  // 32-bit operand offsets, like pn_gemm_f32
  hipLaunchKernelGGL(k_gemm_rowln, dim3(pn_cdiv(M, RLN_BM)), dim3(256), 0, (hipStream_t)stream,
                     x, ldx, W, ldw, bias, res, ldres, gamma, beta, y, ldy, M, K, eps);
  return PN_LAUNCH_CHECK();
}
