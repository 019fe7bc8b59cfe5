// Test-time image front end on the device (configs/mask2former/pairnet.py:310-331):
// mmdet's Resize(keep_ratio) -> Normalize(mean, std, to_rgb) -> Pad -> ImageToTensor of
// one decoded uint8 HWC image, fused into ONE pass: each output element is produced from
// the four source pixels it depends on; the resized uint8 image is never materialised.
//
// Resize is OpenCV's INTER_LINEAR for 8-bit images (what mmcv.imresize calls), restated
// from its published fixed-point algorithm: 11-bit horizontal / vertical coefficients
// (cvRound(w * 2048), half-pixel centres, source index clamped at the borders), integer
// horizontal pass, vertical pass ((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2.
// Integer arithmetic: bit-exact against oracle/preprocess.py.  (cv2 is not in this image:
// unpinned against OpenCV itself.)
#include "common.h"

__device__ __forceinline__ void lin_coef(int d, double scale, int n, int& s, int& a0, int& a1,
                                         bool horizontal) {
  float f = (float)(((double)d + 0.5) * scale - 0.5);
  s = (int)floorf(f);
  f -= (float)s;
  if (horizontal) {
    if (s < 0) { f = 0.f; s = 0; }
    if (s >= n - 1) { f = 0.f; s = n - 1; }
  }
  a0 = (int)rintf((1.f - f) * 2048.f);
  a1 = (int)rintf(f * 2048.f);
}

__global__ __launch_bounds__(256) void k_preprocess(const uint8_t* __restrict__ img, int H,
                                                    int W, float* __restrict__ out, int Hn,
                                                    int Wn, int Hp, int Wp, float m0, float m1,
                                                    float m2, float s0, float s1, float s2,
                                                    int to_rgb) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)Hp * Wp) return;
  const int oy = (int)(e / Wp), ox = (int)(e - (int64_t)oy * Wp);
  const int64_t plane = (int64_t)Hp * Wp;
  if (oy >= Hn || ox >= Wn) {          // Pad: zeros AFTER normalisation (mmcv impad, pad_val 0)
    out[e] = 0.f;
    out[plane + e] = 0.f;
    out[2 * plane + e] = 0.f;
    return;
  }
  int sx, ax0, ax1, sy, by0, by1;
  lin_coef(ox, (double)W / (double)Wn, W, sx, ax0, ax1, true);
  lin_coef(oy, (double)H / (double)Hn, H, sy, by0, by1, false);
  const int x1 = min(sx + 1, W - 1);
  const int y0 = min(max(sy, 0), H - 1), y1 = min(max(sy + 1, 0), H - 1);
  const uint8_t* r0 = img + (int64_t)y0 * W * 3;
  const uint8_t* r1 = img + (int64_t)y1 * W * 3;
  const float mean[3] = {m0, m1, m2}, stdinv[3] = {s0, s1, s2};
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int h0 = (int)r0[sx * 3 + c] * ax0 + (int)r0[x1 * 3 + c] * ax1;
    const int h1 = (int)r1[sx * 3 + c] * ax0 + (int)r1[x1 * 3 + c] * ax1;
    const int v = (((by0 * (h0 >> 4)) >> 16) + ((by1 * (h1 >> 4)) >> 16) + 2) >> 2;
    const int u = min(max(v, 0), 255);
    const int oc = to_rgb ? 2 - c : c;       // BGR -> RGB; mean / std are in OUTPUT order
    out[oc * plane + e] = __fmul_rn(__fsub_rn((float)u, mean[oc]), stdinv[oc]);
  }
}

extern "C" int pn_preprocess_u8_f32(const uint8_t* img, int H, int W, float* out, int Hn, int Wn,
                                    int Hp, int Wp, const float* mean3, const float* stdinv3,
                                    int to_rgb, void* stream) {
  if (!img || !out || !mean3 || !stdinv3 || H <= 0 || W <= 0 || Hn <= 0 || Wn <= 0 || Hp < Hn ||
      Wp < Wn)
    return PN_BAD_ARG;
  hipLaunchKernelGGL(k_preprocess, dim3(pn_cdiv((int64_t)Hp * Wp, 256)), dim3(256), 0,
                     (hipStream_t)stream, img, H, W, out, Hn, Wn, Hp, Wp, mean3[0], mean3[1],
                     mean3[2], stdinv3[0], stdinv3[1], stdinv3[2], to_rgb);
  return PN_LAUNCH_CHECK();
}

// ---- ground truth from the panoptic PNG (pairnet/datasets/psg.py:354-372 for the evaluator,
// pipelines/loading.py:128-147 for the training-side loader): the decoded RGB image -> segment
// id per pixel ([3P] panopticapi rgb2id: R + 256 G + 65536 B) -> one 0/1 byte mask per listed
// segment (`seg == id`, every segment, things and stuff; a listed id absent from the image
// gives an empty mask) and, optionally, the semantic map (the category of the LAST listed
// segment that owns the pixel, 255 where none does: `np.where` applied in list order).
// HBM-bound byte work: 3 B read, G (+4) B written per pixel; a thread owns four consecutive
// pixels (three aligned 32-bit loads, one 32-bit store per segment), the <= 256 ids sit in LDS.
#define PAN_MAX_SEGMENTS 256
__global__ __launch_bounds__(256) void k_pan_masks(const uint8_t* __restrict__ rgb,
                                                   const int* __restrict__ ids,
                                                   const int* __restrict__ cats,
                                                   uint8_t* __restrict__ masks,
                                                   int* __restrict__ sem, const int G,
                                                   const int64_t HW) {
  __shared__ int s_id[PAN_MAX_SEGMENTS], s_cat[PAN_MAX_SEGMENTS];
  for (int g = threadIdx.x; g < G; g += 256) {
    s_id[g] = ids[g];
    s_cat[g] = cats ? cats[g] : 0;
  }
  __syncthreads();
  const int64_t p0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (p0 >= HW) return;
  int id[4];
  if (p0 + 4 <= HW) {             // 12 bytes = three aligned words (the image base is 4-aligned)
    const unsigned* w = reinterpret_cast<const unsigned*>(rgb + p0 * 3);
    const unsigned a = w[0], b = w[1], c = w[2];
    id[0] = (int)(a & 0xffffffu);                                   // R0 G0 B0
    id[1] = (int)((a >> 24) | ((b & 0xffffu) << 8));                // R1 | G1 B1
    id[2] = (int)((b >> 16) | ((c & 0xffu) << 16));                 // R2 G2 | B2
    id[3] = (int)(c >> 8);                                          // R3 G3 B3
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t p = p0 + j < HW ? p0 + j : HW - 1;
      id[j] = rgb[p * 3] + 256 * rgb[p * 3 + 1] + 65536 * rgb[p * 3 + 2];
    }
  }
  int sm[4] = {255, 255, 255, 255};
  const bool whole = p0 + 4 <= HW && !(HW & 3);      // (every row of `masks` 4-aligned)
  for (int g = 0; g < G; ++g) {
    const int want = s_id[g];
    const unsigned m0 = id[0] == want, m1 = id[1] == want, m2 = id[2] == want, m3 = id[3] == want;
    if (m0) sm[0] = s_cat[g];
    if (m1) sm[1] = s_cat[g];
    if (m2) sm[2] = s_cat[g];
    if (m3) sm[3] = s_cat[g];
    uint8_t* dst = masks + (int64_t)g * HW + p0;
    if (whole) {
      *reinterpret_cast<unsigned*>(dst) = m0 | (m1 << 8) | (m2 << 16) | (m3 << 24);
    } else {
      const unsigned m[4] = {m0, m1, m2, m3};
      for (int j = 0; j < 4; ++j)
        if (p0 + j < HW) dst[j] = (uint8_t)m[j];
    }
  }
  if (sem) {
    for (int j = 0; j < 4; ++j)
      if (p0 + j < HW) sem[p0 + j] = sm[j];
  }
}

extern "C" int pn_pan_masks_u8(const uint8_t* rgb, const int* ids, const int* cats, uint8_t* masks,
                               int* sem, int G, int H, int W, void* stream) {
  if (!rgb || H <= 0 || W <= 0 || G < 0 || G > PAN_MAX_SEGMENTS) return PN_BAD_ARG;
  if (G > 0 && (!ids || !masks)) return PN_BAD_ARG;
  if (sem && G > 0 && !cats) return PN_BAD_ARG;
  if (((uintptr_t)rgb | (uintptr_t)masks) & 3) return PN_BAD_ARG;
  if (G == 0 && !sem) return 0;
  const int64_t HW = (int64_t)H * W;
  hipLaunchKernelGGL(k_pan_masks, dim3(pn_cdiv(pn_cdiv(HW, 4), 256)), dim3(256), 0,
                     (hipStream_t)stream, rgb, ids, cats, masks, sem, G, HW);
  return PN_LAUNCH_CHECK();
}
