// Multi-scale deformable attention sampling for the pixel-decoder encoder
// (8 heads x 32 channels, P = 4 points, L <= 4 levels), HBM/L2-bound gather.
//
// Thread = (query, head, sampling point, 4 channels): one 256-thread workgroup per
// query token, 32 lanes per head = 4 points x 8 lanes; the 8 lanes of a point read
// one 128-byte value row per bilinear tap as float4s (every tap is a full line) and
// each lane keeps its L x 4 = 12 independent 16-byte loads in flight at once (48 VGPRs,
// still 5+ waves/SIMD) on this latency-bound gather.  The softmax over
// the L*P logits and the sum over points are xor-8 / xor-16 shuffles inside the
// 32-lane group.  (Measured: the kernel sits at ~75 us/layer regardless of L2 locality
// -- XCD banding, several tokens per workgroup -- i.e. it is bound by the texture-
// addresser rate of 16-byte-per-lane gathers, ~1 GB of them per layer.)  Reference point + offset / (W_l, H_l) and the grid_sample
// un-normalisation follow mmcv's CPU formula
// (multi_scale_deformable_attn_pytorch):
//   loc = ref + off / (W_l, H_l);  g = 2 loc - 1;  ix = ((g + 1) W_l - 1) / 2
// with zero padding outside the map.
#include "common.h"

struct MsdaLevels {
  int h[4], w[4], start[4];
  int L, N;
};

__device__ __forceinline__ float grp_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 8, 64));
  return fmaxf(v, __shfl_xor(v, 16, 64));
}
__device__ __forceinline__ float grp_sum(float v) {
  v += __shfl_xor(v, 8, 64);
  return v + __shfl_xor(v, 16, 64);
}

template <int L>
__global__ __launch_bounds__(256) void k_msda(const float* __restrict__ value,
                                              const float* __restrict__ offaw,
                                              float* __restrict__ out,
                                              const MsdaLevels lv, const int64_t ldv,
                                              const int64_t ldo) {
  const int tid = threadIdx.x;
  const int c4 = tid & 7, pt = (tid >> 3) & 3, head = tid >> 5;
  const int b = blockIdx.y;
  constexpr int LP = L * 4;

  // XCD-aware token order: the dispatcher puts workgroup g on XCD g % 8; give each XCD
  // one horizontal band of the image at EVERY level (rows [k h_l/8, (k+1) h_l/8)), so
  // the value rows its tokens sample (all levels, around the same normalised position)
  // stay inside that XCD's 4 MB L2 instead of streaming the whole 22 MB map through
  // all eight L2s.  Wave-uniform integer math; tokens past a band's end exit.
  const int band = blockIdx.x & 7;
  int i = blockIdx.x >> 3;            // index inside the band
  int n = -1, qw = 1, qh = 1, qs = 0;
#pragma unroll
  for (int l = 0; l < 4; ++l) {
    if (l < L && n < 0) {
      const int r0 = (band * lv.h[l]) >> 3, r1 = ((band + 1) * lv.h[l]) >> 3;
      const int cnt = (r1 - r0) * lv.w[l];
      if (i < cnt) {
        n = lv.start[l] + r0 * lv.w[l] + i;
        qw = lv.w[l]; qh = lv.h[l]; qs = lv.start[l];
      } else {
        i -= cnt;
      }
    }
  }
  if (n < 0) return;
  const int idx = n - qs;
  const int qy = idx / qw, qx = idx - qy * qw;
  const float ref_x = ((float)qx + 0.5f) / (float)qw;
  const float ref_y = ((float)qy + 0.5f) / (float)qh;

  const float* oa = offaw + ((int64_t)b * lv.N + n) * ldo;
  const float* offp = oa + head * LP * 2 + pt * 2;
  const float* awp = oa + 8 * LP * 2 + head * LP + pt;

  // every offset / logit load is issued before the first dependent use
  float e[L];
  float2 off[L];
#pragma unroll
  for (int l = 0; l < L; ++l) {
    e[l] = awp[l * 4];
    off[l] = *reinterpret_cast<const float2*>(offp + l * 8);
  }
  float mx = -INFINITY;
#pragma unroll
  for (int l = 0; l < L; ++l) mx = fmaxf(mx, e[l]);
  mx = grp_max(mx);
  float den = 0.f;
#pragma unroll
  for (int l = 0; l < L; ++l) {
    e[l] = expf(e[l] - mx);
    den += e[l];
  }
  den = grp_sum(den);

  const float* vb = value + (int64_t)b * lv.N * ldv + head * 32 + c4 * 4;
  // All L x 4 taps are loaded UNCONDITIONALLY from clamped coordinates before any is
  // used (out-of-map taps get weight 0): one memory round trip per thread.  Predicated
  // loads would be serialised by hipcc into one branch + full vmcnt wait per level.
  float4 v[L][4];
  float wt[L][4], awl[L];
#pragma unroll
  for (int l = 0; l < L; ++l) {
    const int Hl = lv.h[l], Wl = lv.w[l];
    const float* vl = vb + (int64_t)lv.start[l] * ldv;
    awl[l] = e[l] / den;
    const float locx = ref_x + off[l].x / (float)Wl;
    const float locy = ref_y + off[l].y / (float)Hl;
    const float gx = 2.f * locx - 1.f, gy = 2.f * locy - 1.f;
    const float ix = ((gx + 1.f) * (float)Wl - 1.f) * 0.5f;
    const float iy = ((gy + 1.f) * (float)Hl - 1.f) * 0.5f;
    const float fx = floorf(ix), fy = floorf(iy);
    // (clamp before the int conversion: far-out offsets must not overflow it)
    const int x0 = (int)fminf(fmaxf(fx, -2.f), (float)Wl), y0 = (int)fminf(fmaxf(fy, -2.f), (float)Hl);
    const float tx = ix - fx, ty = iy - fy;
    const bool xin0 = x0 >= 0 && x0 < Wl, xin1 = x0 + 1 >= 0 && x0 + 1 < Wl;
    const bool yin0 = y0 >= 0 && y0 < Hl, yin1 = y0 + 1 >= 0 && y0 + 1 < Hl;
    wt[l][0] = (xin0 && yin0) ? (1.f - tx) * (1.f - ty) : 0.f;
    wt[l][1] = (xin1 && yin0) ? tx * (1.f - ty) : 0.f;
    wt[l][2] = (xin0 && yin1) ? (1.f - tx) * ty : 0.f;
    wt[l][3] = (xin1 && yin1) ? tx * ty : 0.f;
    const int xa = min(max(x0, 0), Wl - 1), xb = min(max(x0 + 1, 0), Wl - 1);
    const int ya = min(max(y0, 0), Hl - 1), yb = min(max(y0 + 1, 0), Hl - 1);
    v[l][0] = ld4(vl + (int64_t)(ya * Wl + xa) * ldv);
    v[l][1] = ld4(vl + (int64_t)(ya * Wl + xb) * ldv);
    v[l][2] = ld4(vl + (int64_t)(yb * Wl + xa) * ldv);
    v[l][3] = ld4(vl + (int64_t)(yb * Wl + xb) * ldv);
  }
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int l = 0; l < L; ++l) {
    const float w_nw = wt[l][0], w_ne = wt[l][1], w_sw = wt[l][2], w_se = wt[l][3];
    const float4 v_nw = v[l][0], v_ne = v[l][1], v_sw = v[l][2], v_se = v[l][3];
    const float aw = awl[l];
    float4 s;
    s.x = ((v_nw.x * w_nw + v_ne.x * w_ne) + v_sw.x * w_sw) + v_se.x * w_se;
    s.y = ((v_nw.y * w_nw + v_ne.y * w_ne) + v_sw.y * w_sw) + v_se.y * w_se;
    s.z = ((v_nw.z * w_nw + v_ne.z * w_ne) + v_sw.z * w_sw) + v_se.z * w_se;
    s.w = ((v_nw.w * w_nw + v_ne.w * w_ne) + v_sw.w * w_sw) + v_se.w * w_se;
    acc.x += s.x * aw; acc.y += s.y * aw; acc.z += s.z * aw; acc.w += s.w * aw;
  }
  acc.x = grp_sum(acc.x); acc.y = grp_sum(acc.y);
  acc.z = grp_sum(acc.z); acc.w = grp_sum(acc.w);
  if (pt == 0) st4(out + ((int64_t)b * lv.N + n) * 256 + head * 32 + c4 * 4, acc);
}

extern "C" int pn_msda_f32(const float* value, int64_t ld_value, const float* offaw,
                           int64_t ld_offaw, float* out, int B, int L, const int32_t* level_h,
                           const int32_t* level_w, void* stream) {
  if (!value || !offaw || !out || B <= 0 || L <= 0 || L > 4 || !level_h || !level_w)
    return PN_BAD_ARG;
  if (ld_value < 256 || (ld_value & 3) || ld_offaw < 8 * L * 12 || (ld_offaw & 1) ||
      ((uintptr_t)value & 15) || ((uintptr_t)offaw & 7))
    return PN_BAD_ARG;
  MsdaLevels lv{};
  lv.L = L;
  int n = 0;
  for (int l = 0; l < L; ++l) {
    if (level_h[l] <= 0 || level_w[l] <= 0) return PN_BAD_ARG;
    lv.h[l] = level_h[l]; lv.w[l] = level_w[l]; lv.start[l] = n;
    n += level_h[l] * level_w[l];
  }
  lv.N = n;
  int per_band = 0;
  for (int k = 0; k < 8; ++k) {
    int c = 0;
    for (int l = 0; l < L; ++l) c += ((((k + 1) * lv.h[l]) >> 3) - ((k * lv.h[l]) >> 3)) * lv.w[l];
    if (c > per_band) per_band = c;
  }
  const dim3 grid(per_band * 8, B);
  hipStream_t s = (hipStream_t)stream;
  switch (L) {
    case 1: hipLaunchKernelGGL(k_msda<1>, grid, dim3(256), 0, s, value, offaw, out, lv, ld_value, ld_offaw); break;
    case 2: hipLaunchKernelGGL(k_msda<2>, grid, dim3(256), 0, s, value, offaw, out, lv, ld_value, ld_offaw); break;
    case 3: hipLaunchKernelGGL(k_msda<3>, grid, dim3(256), 0, s, value, offaw, out, lv, ld_value, ld_offaw); break;
    default: hipLaunchKernelGGL(k_msda<4>, grid, dim3(256), 0, s, value, offaw, out, lv, ld_value, ld_offaw); break;
  }
  return PN_LAUNCH_CHECK();
}
