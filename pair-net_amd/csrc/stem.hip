// ResNet stem: 7x7 stride-2 pad-3 convolution of the NCHW RGB image + folded BatchNorm + ReLU
// -> channel-last [B][Ho][Wo][64]  (configs/mask2former/pairnet.py:9-19; mmdet's ResNet is
// third party, restated in oracle/backbone.py).
//
// Round 3 rewrite.  Rounds 1-2 ran the stem as an A-operand mode of the persistent GEMM whose
// loader gathered every im2col element from global memory (16 scalar loads and a dozen integer
// divisions per thread and 32-deep chunk): 102 us, 49 TFLOP/s, load-issue bound.  Here a
// workgroup owns 4 output rows x 32 output columns (128 pixels x 64 channels):
//   * the 13 x 69 x 3 input patch those pixels read is staged in LDS once, row-contiguous
//     (coalesced) loads, zero outside the image;
//   * wave = (output row pair, channel half): two 32 x 32 accumulators, i.e. two independent
//     MFMA chains that share every weight; its 75 weights per lane (channel = lane & 31) sit in
//     registers for the whole kernel (persistent workgroups);
//   * the 147 taps are contracted two per MFMA in an order chosen so that the two lane halves
//     of an MFMA read patch addresses a CONSTANT apart: per input channel, 21 horizontal pairs
//     (kx, kx + 1), 3 vertical pairs in the last column (ky, ky + 1) and one single tap
//     (its partner's weight is zero): 75 MFMAs, each fed by ONE ds_read_b32 with an immediate
//     offset from one of two per-lane base registers -- no address arithmetic in the loop.
// (The summation order over the taps therefore differs from a k-ascending fmaf chain; it is
// fixed, fp32 throughout, and within 2e-6 of the direct form.)
#include "common.h"

#define STEM_TW 32                     // output columns per tile
#define STEM_TH 4                      // output rows per tile
#define STEM_PR (2 * STEM_TH + 5)      // patch rows: 13
#define STEM_PC (2 * STEM_TW + 5)      // patch columns: 69
#define STEM_LD 72                     // LDS row stride
#define STEM_NM 75                     // MFMAs per 32 x 32 output block

// MFMA m contracts taps k_lo (lane half 0) and k_hi (lane half 1; -1 = none, weight 0);
// off = patch offset of k_lo, the upper half reads off + 1 (horizontal pair) or + STEM_LD.
struct StemMma { int off, k_lo, k_hi; int kind; };   // kind: 0 horizontal pair, 1 vertical pair, 2 single
__host__ __device__ constexpr StemMma stem_mma(int m) {
  const int c = m / 25, q = m % 25;
  if (q < 21) {                                   // (kx, kx + 1), kx = 0, 2, 4
    const int ky = q / 3, kx = 2 * (q % 3);
    return {(c * STEM_PR + ky) * STEM_LD + kx, c * 49 + ky * 7 + kx, c * 49 + ky * 7 + kx + 1,
            0};
  }
  if (q < 24) {                                   // last column: (ky, ky + 1), ky = 0, 2, 4
    const int ky = 2 * (q - 21);
    return {(c * STEM_PR + ky) * STEM_LD + 6, c * 49 + ky * 7 + 6, c * 49 + (ky + 1) * 7 + 6, 1};
  }
  return {(c * STEM_PR + 6) * STEM_LD + 6, c * 49 + 6 * 7 + 6, -1, 2};      // single tap
}

__global__ __launch_bounds__(256) void k_stem7x7s2(const float* __restrict__ img,
                                                   const float* __restrict__ Wp,
                                                   const float* __restrict__ bias,
                                                   float* __restrict__ out, int H, int W, int Ho,
                                                   int Wo, int tiles_x, int tiles_per_img,
                                                   int ntiles) {
  // the patch; before the tile loop the same memory holds the weights [64][161] once
  __shared__ float patch[64 * 161 > 3 * STEM_PR * STEM_LD ? 64 * 161 : 3 * STEM_PR * STEM_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int rsel = wave >> 1, chalf = wave & 1;
  const int ch = chalf * 32 + li;
  // ---- this lane's weight of MFMA m: the tap (c, ky, kx) its half contributes ----
  // (through LDS: coalesced float4 rows in, a conflict-free stride-161 column out)
  for (int e = tid; e < 64 * 40; e += 256) {
    const int r = e / 40, c4 = e - r * 40;
    const float4 w4 = *reinterpret_cast<const float4*>(Wp + r * 160 + 4 * c4);
    float* d = patch + r * 161 + 4 * c4;
    d[0] = w4.x; d[1] = w4.y; d[2] = w4.z; d[3] = w4.w;
  }
  __syncthreads();
  float wf[STEM_NM];
#pragma unroll
  for (int m = 0; m < STEM_NM; ++m) {
    const StemMma d = stem_mma(m);
    const int k = lh ? d.k_hi : d.k_lo;
    wf[m] = k >= 0 ? patch[ch * 161 + (k >= 0 ? k : 0)] : 0.f;
  }
  const float bv = bias[ch];
  // per-lane bases of the two pair types: the upper half reads one column / one row further
  const float* baseH = patch + 4 * rsel * STEM_LD + 2 * li + lh;
  const float* baseV = patch + 4 * rsel * STEM_LD + 2 * li + lh * STEM_LD;
  const float* baseS = patch + 4 * rsel * STEM_LD + 2 * li;   // single tap: both halves read
                                                              // the tap, the upper weight is 0

  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int b = t / tiles_per_img, r = t - b * tiles_per_img;
    const int ty = r / tiles_x, tx = r - ty * tiles_x;
    const int oy0 = STEM_TH * ty, ox0 = STEM_TW * tx;
    const float* ib = img + (int64_t)b * 3 * H * W;
    // ---- stage the patch (unconditional clamped loads, zero outside the image) ----
    __syncthreads();   // the previous tile's reads are done
    {
      constexpr int NE = 3 * STEM_PR * STEM_LD, NL = (NE + 255) / 256;
      float v[NL];
      unsigned in = 0;
#pragma unroll
      for (int j = 0; j < NL; ++j) {       // all loads of the patch in flight together
        const int e = min(tid + 256 * j, NE - 1);
        const int c = e / (STEM_PR * STEM_LD), rem = e - c * (STEM_PR * STEM_LD);
        const int pr = rem / STEM_LD, pc = rem - pr * STEM_LD;
        const int iy = 2 * oy0 - 3 + pr, ix = 2 * ox0 - 3 + pc;
        const bool ok = pc < STEM_PC && iy >= 0 && iy < H && ix >= 0 && ix < W;
        in |= (ok ? 1u : 0u) << j;
        const int yc = min(max(iy, 0), H - 1), xc = min(max(ix, 0), W - 1);
        v[j] = ib[((int64_t)c * H + yc) * W + xc];
      }
#pragma unroll
      for (int j = 0; j < NL; ++j)
        if (tid + 256 * j < NE) patch[tid + 256 * j] = ((in >> j) & 1u) ? v[j] : 0.f;
    }
    __syncthreads();
    // ---- 2 x 75 MFMAs: output rows oy0 + 2 rsel and + 1 (patch rows two further down) ----
    f32x16 acc0, acc1;
#pragma unroll
    for (int q = 0; q < 16; ++q) { acc0[q] = 0.f; acc1[q] = 0.f; }
#pragma unroll
    for (int m = 0; m < STEM_NM; ++m) {
      const StemMma d = stem_mma(m);
      const float* bp = d.kind == 0 ? baseH : d.kind == 1 ? baseV : baseS;
      acc0 = mfma32(bp[d.off], wf[m], acc0);
      acc1 = mfma32(bp[d.off + 2 * STEM_LD], wf[m], acc1);
    }
    // ---- bias + ReLU; D[pixel][channel]: register q <-> pixel mfma32_row(q, lh) ----
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int oy = oy0 + 2 * rsel + half;
      if (oy < Ho) {
        float* ob = out + (((int64_t)b * Ho + oy) * Wo) * 64 + ch;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int ox = ox0 + mfma32_row(q, lh);
          const float a = half ? acc1[q] : acc0[q];
          if (ox < Wo) ob[(int64_t)ox * 64] = fmaxf(a + bv, 0.f);
        }
      }
    }
  }
}

// Wp is [64][160]: the PyTorch weight [64][3][7][7] flattened (k = c*49 + ky*7 + kx) and
// zero-padded from 147 to 160.  flags: 0 or PN_GEMM_RESERVE(n).
extern "C" int pn_stem7x7s2_f32(const float* img, const float* Wp, const float* bias, float* out,
                                int B, int H, int W, int flags, void* stream) {
  if (!img || !Wp || !bias || !out || B <= 0 || H <= 0 || W <= 0 || ((uintptr_t)Wp & 15))
    return PN_BAD_ARG;
  const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
  const int tiles_x = pn_cdiv(Wo, STEM_TW), tiles_y = pn_cdiv(Ho, STEM_TH);
  const int64_t ntiles = (int64_t)B * tiles_x * tiles_y;
  if (ntiles >= ((int64_t)1 << 31)) return PN_BAD_ARG;
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n >= 8)
      cus = n;
  }
  static const int wg_per_cu = [] {              // resident workgroups per CU, asked once
    int n = 0;
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_stem7x7s2, 256, 0) == hipSuccess &&
                   n > 0 ? n : 2;
  }();
  const int reserve = ((flags >> PN_GEMM_RESERVE_SHIFT) & 0x3ff) * 8;
  int64_t grid = (int64_t)cus * wg_per_cu - reserve;
  if (grid < 256) grid = 256;
  if (grid > ntiles) grid = ntiles;
  grid = pn_cdiv(ntiles, pn_cdiv(ntiles, grid));   // same number of rounds, fewer workgroups
  hipLaunchKernelGGL(k_stem7x7s2, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, img, Wp,
                     bias, out, H, W, Ho, Wo, tiles_x, tiles_x * tiles_y, (int)ntiles);
  return PN_LAUNCH_CHECK();
}
