// Winograd F(2x2, 3x3) transforms for the pixel decoder's 3x3 output convolution
// (256 -> 256 channels on the 1/4-resolution map, 78.8 GFLOP as a direct convolution):
// 16 multiplications per 2x2 output tile and channel pair instead of 36, i.e. the
// contraction becomes 16 independent [tiles x Cin] x [Cin x Cout] GEMMs (35 GFLOP) that run
// as ONE batched launch of the persistent GEMM kernel, bracketed by the two HBM-bound
// transforms below.  fp32 throughout; the transform matrices hold only 0, +-1, +-1/2, so
// the result differs from the direct convolution by ordinary fp32 re-association.
//
//   V = B^T d B   (d: 4x4 input patch, zero outside the image)
//   M_xi = V_xi . U_xi^T   for the 16 positions xi = (i, j),  U = G g G^T (host, pack time)
//   Y = A^T M A   (2x2 outputs) + bias
//
//   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]     A^T = [1 1 1 0; 0 1 -1 -1]
#include "common.h"

__device__ __forceinline__ float4 sub4(float4 a, float4 b) {
  return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w);
}

// in [B][H][W][C] -> V [16][B * (H/2) * (W/2)][C]; thread = (tile, 4 channels)
__global__ __launch_bounds__(256) void k_wino_f23_input(const float* __restrict__ in,
                                                        float* __restrict__ V, int H, int W,
                                                        int C4, int64_t tiles_total) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= tiles_total * C4) return;
  const int c = (int)(e % C4);
  const int64_t tile = e / C4;
  const int th = (H + 1) >> 1, tw = (W + 1) >> 1;   // (odd sides: the last tile row / column
  const int b = (int)(tile / ((int64_t)th * tw));   //  sticks out, reads zeros there and
  const int r = (int)(tile - (int64_t)b * th * tw); //  stores only its inside part)
  const int ty = r / tw, tx = r - ty * tw;
  const float* ib = in + (int64_t)b * H * W * C4 * 4 + c * 4;
  float4 d[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int y = 2 * ty - 1 + i;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int x = 2 * tx - 1 + j;
      const bool ok = y >= 0 && y < H && x >= 0 && x < W;
      // unconditional load from a clamped address, zeroed by a select (hipcc serialises
      // predicated loads)
      const float4 v = ld4(ib + ((int64_t)min(max(y, 0), H - 1) * W + min(max(x, 0), W - 1)) * C4 * 4);
      d[i][j] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  float4 t[4][4];   // t = B^T d
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    t[0][j] = sub4(d[0][j], d[2][j]);
    t[1][j] = add4(d[1][j], d[2][j]);
    t[2][j] = sub4(d[2][j], d[1][j]);
    t[3][j] = sub4(d[1][j], d[3][j]);
  }
  const int64_t plane = tiles_total * C4 * 4;
  float* vb = V + tile * C4 * 4 + c * 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) {   // V = t B
    st4(vb + (int64_t)(4 * i + 0) * plane, sub4(t[i][0], t[i][2]));
    st4(vb + (int64_t)(4 * i + 1) * plane, add4(t[i][1], t[i][2]));
    st4(vb + (int64_t)(4 * i + 2) * plane, sub4(t[i][2], t[i][1]));
    st4(vb + (int64_t)(4 * i + 3) * plane, sub4(t[i][1], t[i][3]));
  }
}

// M [16][tiles][C] -> out [B][H][W][C] = act(A^T M A + bias); thread = (tile, 4 channels)
__global__ __launch_bounds__(256) void k_wino_f23_output(const float* __restrict__ Mx,
                                                         const float* __restrict__ bias,
                                                         float* __restrict__ out, int H, int W,
                                                         int C4, int64_t tiles_total, int relu) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= tiles_total * C4) return;
  const int c = (int)(e % C4);
  const int64_t tile = e / C4;
  const int th = (H + 1) >> 1, tw = (W + 1) >> 1;
  const int b = (int)(tile / ((int64_t)th * tw));
  const int r = (int)(tile - (int64_t)b * th * tw);
  const int ty = r / tw, tx = r - ty * tw;
  const int64_t plane = tiles_total * C4 * 4;
  const float* mb = Mx + tile * C4 * 4 + c * 4;
  float4 m[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) m[i][j] = ld4(mb + (int64_t)(4 * i + j) * plane);
  float4 s[2][4];   // s = A^T m
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    s[0][j] = add4(add4(m[0][j], m[1][j]), m[2][j]);
    s[1][j] = sub4(sub4(m[1][j], m[2][j]), m[3][j]);
  }
  const float4 bv = bias ? ld4(bias + c * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  float* ob = out + (int64_t)b * H * W * C4 * 4 + c * 4;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    float4 y0 = add4(add4(add4(s[i][0], s[i][1]), s[i][2]), bv);
    float4 y1 = add4(sub4(sub4(s[i][1], s[i][2]), s[i][3]), bv);
    if (relu) {
      y0 = make_float4(fmaxf(y0.x, 0.f), fmaxf(y0.y, 0.f), fmaxf(y0.z, 0.f), fmaxf(y0.w, 0.f));
      y1 = make_float4(fmaxf(y1.x, 0.f), fmaxf(y1.y, 0.f), fmaxf(y1.z, 0.f), fmaxf(y1.w, 0.f));
    }
    if (2 * ty + i < H) {
      st4(ob + ((int64_t)(2 * ty + i) * W + 2 * tx) * C4 * 4, y0);
      if (2 * tx + 1 < W) st4(ob + ((int64_t)(2 * ty + i) * W + 2 * tx + 1) * C4 * 4, y1);
    }
  }
}

extern "C" int pn_winograd_f23_input_f32(const float* in, float* V, int B, int H, int W, int C,
                                         void* stream) {
  if (!in || !V || B <= 0 || H < 2 || W < 2 || C <= 0 || (C & 3) ||
      (((uintptr_t)in | (uintptr_t)V) & 15))
    return PN_BAD_ARG;
  const int64_t tiles = (int64_t)B * ((H + 1) / 2) * ((W + 1) / 2);
  hipLaunchKernelGGL(k_wino_f23_input, dim3(pn_cdiv(tiles * (C / 4), 256)), dim3(256), 0,
                     (hipStream_t)stream, in, V, H, W, C / 4, tiles);
  return PN_LAUNCH_CHECK();
}

extern "C" int pn_winograd_f23_output_f32(const float* Mx, const float* bias, float* out, int B,
                                          int H, int W, int C, int relu, void* stream) {
  if (!Mx || !out || B <= 0 || H < 2 || W < 2 || C <= 0 || (C & 3) ||
      (((uintptr_t)Mx | (uintptr_t)out | (uintptr_t)bias) & 15))
    return PN_BAD_ARG;
  const int64_t tiles = (int64_t)B * ((H + 1) / 2) * ((W + 1) / 2);
  hipLaunchKernelGGL(k_wino_f23_output, dim3(pn_cdiv(tiles * (C / 4), 256)), dim3(256), 0,
                     (hipStream_t)stream, Mx, bias, out, H, W, C / 4, tiles, relu);
  return PN_LAUNCH_CHECK();
}

// ---- Winograd F(4x4, 3x3): 36 multiplications per 4x4 output tile instead of 144 -------
// (4x fewer than the direct form; Lavin & Gray's matrices).  Tiles that stick out of the
// image (H or W not a multiple of 4) read zeros and store only their inside part.
//   B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
//   A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
__device__ __forceinline__ float4 axpy4(float a, float4 x, float4 y) {
  return make_float4(fmaf(a, x.x, y.x), fmaf(a, x.y, y.y), fmaf(a, x.z, y.z), fmaf(a, x.w, y.w));
}
__device__ __forceinline__ float4 scale4(float a, float4 x) {
  return make_float4(a * x.x, a * x.y, a * x.z, a * x.w);
}

// one application of B^T to six values (a column or a row of the 6x6 patch)
__device__ __forceinline__ void bt6(const float4 (&d)[6], float4 (&o)[6]) {
  o[0] = axpy4(4.f, d[0], axpy4(-5.f, d[2], d[4]));
  const float4 p = axpy4(-4.f, d[2], d[4]), q = axpy4(-4.f, d[1], d[3]);
  o[1] = add4(p, q);
  o[2] = sub4(p, q);
  const float4 r = sub4(d[4], d[2]), s2 = scale4(2.f, sub4(d[3], d[1]));
  o[3] = add4(r, s2);
  o[4] = sub4(r, s2);
  o[5] = axpy4(4.f, d[1], axpy4(-5.f, d[3], d[5]));
}

// one application of A^T to six values -> four
__device__ __forceinline__ void at6(const float4 (&m)[6], float4 (&o)[4]) {
  const float4 a = add4(m[1], m[2]), b = sub4(m[1], m[2]);
  const float4 c = add4(m[3], m[4]), e = sub4(m[3], m[4]);
  o[0] = add4(add4(m[0], a), c);
  o[1] = axpy4(2.f, e, b);
  o[2] = axpy4(4.f, c, a);
  o[3] = add4(axpy4(8.f, e, b), m[5]);
}

__global__ __launch_bounds__(256, 3) void k_wino_f43_input(const float* __restrict__ in,
                                                        float* __restrict__ V, int H, int W,
                                                        int C4, int th, int tw,
                                                        int64_t tiles_total) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= tiles_total * C4) return;
  const int c = (int)(e % C4);
  const int64_t tile = e / C4;
  const int b = (int)(tile / ((int64_t)th * tw));
  const int r = (int)(tile - (int64_t)b * th * tw);
  const int ty = r / tw, tx = r - ty * tw;
  const float* ib = in + (int64_t)b * H * W * C4 * 4 + c * 4;
  float4 t[6][6];   // t = B^T d, built column by column
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const int x = 4 * tx - 1 + j;
    float4 col[6], o[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int y = 4 * ty - 1 + i;
      const bool ok = y >= 0 && y < H && x >= 0 && x < W;
      const float4 v = ld4(ib + ((int64_t)min(max(y, 0), H - 1) * W + min(max(x, 0), W - 1)) * C4 * 4);
      col[i] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    bt6(col, o);
#pragma unroll
    for (int i = 0; i < 6; ++i) t[i][j] = o[i];
  }
  const int64_t plane = tiles_total * C4 * 4;
  float* vb = V + tile * C4 * 4 + c * 4;
#pragma unroll
  for (int i = 0; i < 6; ++i) {   // V = t B  (B^T applied along the row)
    float4 o[6];
    bt6(t[i], o);
#pragma unroll
    for (int j = 0; j < 6; ++j) st4(vb + (int64_t)(6 * i + j) * plane, o[j]);
  }
}

__global__ __launch_bounds__(256) void k_wino_f43_output(const float* __restrict__ Mx,
                                                         const float* __restrict__ bias,
                                                         float* __restrict__ out, int H, int W,
                                                         int C4, int th, int tw,
                                                         int64_t tiles_total, int relu) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= tiles_total * C4) return;
  const int c = (int)(e % C4);
  const int64_t tile = e / C4;
  const int b = (int)(tile / ((int64_t)th * tw));
  const int r = (int)(tile - (int64_t)b * th * tw);
  const int ty = r / tw, tx = r - ty * tw;
  const int64_t plane = tiles_total * C4 * 4;
  const float* mb = Mx + tile * C4 * 4 + c * 4;
  float4 s[4][6];   // s = A^T m, column by column
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    float4 col[6], o[4];
#pragma unroll
    for (int i = 0; i < 6; ++i) col[i] = ld4(mb + (int64_t)(6 * i + j) * plane);
    at6(col, o);
#pragma unroll
    for (int i = 0; i < 4; ++i) s[i][j] = o[i];
  }
  const float4 bv = bias ? ld4(bias + c * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  float* ob = out + (int64_t)b * H * W * C4 * 4 + c * 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float4 o[4];
    at6(s[i], o);
    const int y = 4 * ty + i;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int x = 4 * tx + j;
      float4 v = add4(o[j], bv);
      if (relu) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
      if (y < H && x < W) st4(ob + ((int64_t)y * W + x) * C4 * 4, v);
    }
  }
}

static bool wino43_ok(const void* a, const void* b, const void* c, int B, int H, int W, int C) {
  return a && b && B > 0 && H >= 1 && W >= 1 && C > 0 && !(C & 3) &&
         !(((uintptr_t)a | (uintptr_t)b | (uintptr_t)c) & 15);
}

extern "C" int pn_winograd_f43_input_f32(const float* in, float* V, int B, int H, int W, int C,
                                         void* stream) {
  if (!wino43_ok(in, V, nullptr, B, H, W, C)) return PN_BAD_ARG;
  const int th = (H + 3) / 4, tw = (W + 3) / 4;
  const int64_t tiles = (int64_t)B * th * tw;
  hipLaunchKernelGGL(k_wino_f43_input, dim3(pn_cdiv(tiles * (C / 4), 256)), dim3(256), 0,
                     (hipStream_t)stream, in, V, H, W, C / 4, th, tw, tiles);
  return PN_LAUNCH_CHECK();
}

extern "C" int pn_winograd_f43_output_f32(const float* Mx, const float* bias, float* out, int B,
                                          int H, int W, int C, int relu, void* stream) {
  if (!wino43_ok(Mx, out, bias, B, H, W, C)) return PN_BAD_ARG;
  const int th = (H + 3) / 4, tw = (W + 3) / 4;
  const int64_t tiles = (int64_t)B * th * tw;
  hipLaunchKernelGGL(k_wino_f43_output, dim3(pn_cdiv(tiles * (C / 4), 256)), dim3(256), 0,
                     (hipStream_t)stream, Mx, bias, out, H, W, C / 4, th, tw, tiles, relu);
  return PN_LAUNCH_CHECK();
}

// ---- weight transforms U = G g G^T on the device (load time, and after every optimizer step of
// a trained backbone: pair-net_amd/train.py): one thread per (co, ci) filter, double arithmetic in
// a fixed order, rounded to fp32 once.  form 2: F(2x2,3x3), U [16][Co][Ci]; form 4: F(4x4,3x3),
// U [36][Co][Ci].  (Rounds 1-5 did this with a float64 torch.einsum, i.e. a rocBLAS call, at
// pack time; inside a long-lived training process that call was seen to abort the process.)
template <int R>
__global__ __launch_bounds__(256) void k_winograd_weights(const float* __restrict__ w,
                                                          float* __restrict__ U, int Co, int Ci) {
  const int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (f >= (int64_t)Co * Ci) return;
  double G[R][3];
  if (R == 4) {
    const double g4[4][3] = {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0.0, 0.0, 1.0}};
    for (int i = 0; i < 4; ++i) for (int k = 0; k < 3; ++k) G[i][k] = g4[i][k];
  } else {
    const double g6[6][3] = {{1.0 / 4, 0.0, 0.0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6},
                             {-1.0 / 6, 1.0 / 6, -1.0 / 6}, {1.0 / 24, 1.0 / 12, 1.0 / 6},
                             {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0.0, 0.0, 1.0}};
    for (int i = 0; i < 6; ++i) for (int k = 0; k < 3; ++k) G[i % R][k] = g6[i][k];
  }
  double g[3][3];
  for (int k = 0; k < 9; ++k) g[k / 3][k % 3] = (double)w[f * 9 + k];
  double t[R][3];
  for (int i = 0; i < R; ++i)
    for (int l = 0; l < 3; ++l) t[i][l] = G[i][0] * g[0][l] + G[i][1] * g[1][l] + G[i][2] * g[2][l];
  for (int i = 0; i < R; ++i)
    for (int j = 0; j < R; ++j)
      U[((int64_t)(i * R + j)) * Co * Ci + f] =
          (float)(t[i][0] * G[j][0] + t[i][1] * G[j][1] + t[i][2] * G[j][2]);
}

extern "C" int pn_winograd_weights_f32(const float* w, float* U, int Co, int Ci, int form,
                                       void* stream) {
  if (!w || !U || Co <= 0 || Ci <= 0 || (form != 2 && form != 4)) return PN_BAD_ARG;
  const dim3 grid(pn_cdiv((int64_t)Co * Ci, 256));
  if (form == 2)
    hipLaunchKernelGGL(k_winograd_weights<4>, grid, dim3(256), 0, (hipStream_t)stream, w, U, Co, Ci);
  else
    hipLaunchKernelGGL(k_winograd_weights<6>, grid, dim3(256), 0, (hipStream_t)stream, w, U, Co, Ci);
  return PN_LAUNCH_CHECK();
}
