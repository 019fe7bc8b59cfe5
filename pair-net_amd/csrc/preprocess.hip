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
