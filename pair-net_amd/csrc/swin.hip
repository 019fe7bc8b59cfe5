// Swin Transformer backbone kernels (pairnet_swinb.py:203-226,
// BASELINE.json configs[3]): everything between the GEMMs of gemm.hip.
//   k_ln_rows        LayerNorm over C channels of token rows (any C % 4 == 0, C <= 3072);
//                    MERGE: the row is gathered from a 2x2 neighbourhood (patch merging)
//   k_patch_im2col   4x4/4 patch embedding: NCHW image -> [tokens][64] rows (48 + 16 zeros)
//   k_window_attn    (shifted-)window multi-head attention straight on the [tokens][3C]
//                    qkv matrix: padding to the window grid, the cyclic shift, the window
//                    partition and their inverses are index arithmetic, not copies
// Token maps are channel-last rows [B][H*W][C] throughout.
#include "common.h"
#include "s3_common.h"

#define LN_MAXV 12  // float4 per lane: C <= 64 * 4 * 12 = 3072

struct MergeP { int H, W, H2, W2, C; };

// One wave per row.  Two-pass mean / variance on the register copy of the row.
// MERGE: row (b, y2, x2) = [x(2y2, 2x2) | x(2y2, 2x2+1) | x(2y2+1, 2x2) | x(2y2+1, 2x2+1)],
// zeros beyond an odd map's edge (the reference pads before nn.Unfold); gamma / beta are
// expected in this neighbour-major order.
// S3OUT: y is an S3 operand [rows x C] (csrc/gemm_s3.hip: the qkv / FFN GEMM's A operand written
// pre-split): even lanes take their neighbour's four channels and write the three 16-byte plane
// pieces of 8 consecutive channels of this row.
template <bool MERGE, bool S3OUT = false>
__global__ __launch_bounds__(256) void k_ln_rows(const float* __restrict__ x, int64_t ldx,
                                                 const float* __restrict__ g,
                                                 const float* __restrict__ b,
                                                 float* __restrict__ y, int64_t ldy,
                                                 int64_t rows, int C, float eps, MergeP mp) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nv = (C + 255) >> 8;
  float4 v[LN_MAXV];
  int64_t img = 0;
  int y2 = 0, x2 = 0;
  if (MERGE) {
    const int64_t per = (int64_t)mp.H2 * mp.W2;
    img = row / per;
    const int r = (int)(row - img * per);
    y2 = r / mp.W2; x2 = r - y2 * mp.W2;
  }
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c = i * 256 + lane * 4;
    v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < nv && c < C) {
      if (MERGE) {
        const int q = c / mp.C, cc = c - q * mp.C;
        const int yy = 2 * y2 + (q >> 1), xx = 2 * x2 + (q & 1);
        if (yy < mp.H && xx < mp.W)
          v[i] = ld4(x + ((img * mp.H + yy) * mp.W + xx) * ldx + cc);
      } else {
        v[i] = ld4(x + row * ldx + c);
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  const float mean = wave_sum(s) / (float)C;
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c = i * 256 + lane * 4;
    if (i < nv && c < C) {
      v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
      ss += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
    }
  }
  const float rstd = 1.f / sqrtf(wave_sum(ss) / (float)C + eps);
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c = i * 256 + lane * 4;
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < nv && c < C) {
      const float4 gg = ld4(g + c), bb = ld4(b + c);
      o = make_float4(v[i].x * rstd * gg.x + bb.x, v[i].y * rstd * gg.y + bb.y,
                      v[i].z * rstd * gg.z + bb.z, v[i].w * rstd * gg.w + bb.w);
      if (!S3OUT) st4(y + row * ldy + c, o);
    }
    if (S3OUT) {
      if (i >= nv) continue;                 // (wave-uniform)
      const float4 nb = make_float4(__shfl_xor(o.x, 1, 64), __shfl_xor(o.y, 1, 64),
                                    __shfl_xor(o.z, 1, 64), __shfl_xor(o.w, 1, 64));
      if ((lane & 1) == 0 && c < C) {        // C % 8 == 0: the pair is in or out together
        const float v8[8] = {o.x, o.y, o.z, o.w, nb.x, nb.y, nb.z, nb.w};
        s3_frag q0, q1, q2;
        s3_split8(v8, q0, q1, q2);
        uint4* dst = reinterpret_cast<uint4*>(y) + ((row >> 5) * (C >> 4) + (c >> 4)) * 192 +
                     ((c >> 3) & 1) * 32 + (row & 31);
        dst[0] = q0.u; dst[64] = q1.u; dst[128] = q2.u;
      }
    }
  }
}

extern "C" int pn_layernorm_rows_s3_f32(const float* x, int64_t ldx, const float* gamma,
                                        const float* beta, void* y_s3, int64_t rows, int C,
                                        float eps, void* stream) {
  if (!x || !gamma || !beta || !y_s3 || rows <= 0 || C <= 0 || (C & 15) || C > 256 * LN_MAXV ||
      ldx < C || (ldx & 3))
    return PN_BAD_ARG;
  if (((uintptr_t)x | (uintptr_t)y_s3 | (uintptr_t)gamma | (uintptr_t)beta) & 15) return PN_BAD_ARG;
  hipLaunchKernelGGL((k_ln_rows<false, true>), dim3(pn_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream,
                     x, ldx, gamma, beta, (float*)y_s3, 0, rows, C, eps, MergeP{});
  return PN_LAUNCH_CHECK();
}

extern "C" int pn_layernorm_rows_f32(const float* x, int64_t ldx, const float* gamma,
                                     const float* beta, float* y, int64_t ldy, int64_t rows,
                                     int C, float eps, void* stream) {
  if (!x || !gamma || !beta || !y || rows <= 0 || C <= 0 || (C & 3) || C > 256 * LN_MAXV ||
      ldx < C || ldy < C || ((ldx | ldy) & 3))
    return PN_BAD_ARG;
  if (((uintptr_t)x | (uintptr_t)y | (uintptr_t)gamma | (uintptr_t)beta) & 15) return PN_BAD_ARG;
  hipLaunchKernelGGL(k_ln_rows<false>, dim3(pn_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream,
                     x, ldx, gamma, beta, y, ldy, rows, C, eps, MergeP{});
  return PN_LAUNCH_CHECK();
}

extern "C" int pn_patch_merge_ln_f32(const float* x, const float* gamma, const float* beta,
                                     float* y, int B, int H, int W, int C, float eps,
                                     void* stream) {
  if (!x || !gamma || !beta || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3) ||
      4 * C > 256 * LN_MAXV)
    return PN_BAD_ARG;
  if (((uintptr_t)x | (uintptr_t)y | (uintptr_t)gamma | (uintptr_t)beta) & 15) return PN_BAD_ARG;
  MergeP mp{H, W, (H + 1) / 2, (W + 1) / 2, C};
  const int64_t rows = (int64_t)B * mp.H2 * mp.W2;
  hipLaunchKernelGGL(k_ln_rows<true>, dim3(pn_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream,
                     x, (int64_t)C, gamma, beta, y, (int64_t)4 * C, rows, 4 * C, eps, mp);
  return PN_LAUNCH_CHECK();
}

// 16 threads per token, thread j writes the float4 of columns [4j, 4j+4): j < 12 is
// (channel j / 4, patch row j % 4) -- the flattening order of a [C][3][4][4] conv weight.
__global__ __launch_bounds__(256) void k_patch_im2col(const float* __restrict__ img,
                                                      float* __restrict__ out, int B, int H,
                                                      int W, int H4, int W4) {
  const int64_t t = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  const int j = threadIdx.x & 15;
  if (t >= (int64_t)B * H4 * W4) return;
  const int b = (int)(t / ((int64_t)H4 * W4));
  const int r = (int)(t - (int64_t)b * H4 * W4);
  const int ty = r / W4, tx = r - ty * W4;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (j < 12) {
    const int c = j >> 2, yy = ty * 4 + (j & 3), x0 = tx * 4;
    if (yy < H) {
      const float* p = img + (((int64_t)b * 3 + c) * H + yy) * W;
      v.x = x0 + 0 < W ? p[x0 + 0] : 0.f;
      v.y = x0 + 1 < W ? p[x0 + 1] : 0.f;
      v.z = x0 + 2 < W ? p[x0 + 2] : 0.f;
      v.w = x0 + 3 < W ? p[x0 + 3] : 0.f;
    }
  }
  st4(out + t * 64 + j * 4, v);
}

extern "C" int pn_patch_im2col4_f32(const float* img, float* out, int B, int H, int W,
                                    void* stream) {
  if (!img || !out || B <= 0 || H <= 0 || W <= 0 || ((uintptr_t)out & 15)) return PN_BAD_ARG;
  const int H4 = (H + 3) / 4, W4 = (W + 3) / 4;
  hipLaunchKernelGGL(k_patch_im2col, dim3(pn_cdiv((int64_t)B * H4 * W4, 16)), dim3(256), 0,
                     (hipStream_t)stream, img, out, B, H, W, H4, W4);
  return PN_LAUNCH_CHECK();
}

// ---- (shifted-)window attention, head dim 32 --------------------------------------
// Workgroup = (window, head, image), one wave per 32 queries of the window's N = ws^2
// tokens (2 waves for ws = 7, 5 for ws = 12).  Position p = (py, px) of window (wy, wx)
// sits at (y, x) = (wy ws + py, wx ws + px) of the padded, rolled map, i.e. at source
// pixel ((y + shift) mod Hp, (x + shift) mod Wp); a source pixel outside the H x W map is
// padding: the reference pads AFTER norm1, so its q / k / v rows are the qkv bias.
// Scores follow k_attn_chunk (attn.hip): S^T = K Q^T on the f32 MFMA with the query in
// lane & 31, so softmax statistics are lane-local plus one xor-32; P^T feeds the P.V MFMA
// from the accumulator registers.  Added before the softmax: the relative position bias
// table[head][(qy - ky + ws - 1)(2 ws - 1) + (qx - kx + ws - 1)] (this head's row is
// staged in LDS) and -100 between tokens of different wrap-around regions (shift > 0).
// LDS: K and V rows of 32 floats without padding (the 18-block stage of Swin-L launches 840
// workgroups: at 40 KB each all of them are resident, 4 per CU).  K row r keeps its 16-byte
// chunk c at position c ^ ((r ^ r >> 3) & 7): the per-lane-row b128 reads of the S^T MFMA
// touch every bank once per 16 lanes.  V row r lives at physical row swap_bits_0_2(r): the
// two half-waves of the P.V operand read (rows r, r + 4) land in different bank halves.
// Rows N8..32 nt of the last tile are never stored; reads there fall into the arrays that
// follow (V -> K rows, finite, times P = 0; K -> table / metadata, whose scores are
// discarded by the key < N select).
#define WA_MAXN 169  // ws <= 13
#define LOG2E 1.4426950408889634f

__device__ __forceinline__ int wa_kpos(int r, int c) { return r * 32 + 4 * (c ^ ((r ^ (r >> 3)) & 7)); }
__device__ __forceinline__ int wa_vrow(int r) { return (r & ~5) | ((r & 1) << 2) | ((r >> 2) & 1); }

struct WinP {
  const float* qkv; const float* qkv_bias; const float* table; float* out;
  int64_t ldqkv, ldo;
  int H, W, Hp, Wp, C, heads, ws, shift, nwx;
  float scale;
  int s3_out;     // out is an S3 operand [B H W x C] (ldo ignored): the proj GEMM's A, pre-split
};

__global__ __launch_bounds__(512) void k_window_attn(const WinP p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int N = p.ws * p.ws;
  const int nt = (N + 31) >> 5;            // key tiles == waves
  const int N8 = (N + 7) & ~7;
  float* Vs = smem;                        // [N8][32], rows permuted
  float* Ks = Vs + N8 * 32;                // [N8][32], chunks swizzled
  float* tab = Ks + N8 * 32;               // [(2ws-1)^2] this head's bias row
  int* meta = reinterpret_cast<int*>(tab + (2 * p.ws - 1) * (2 * p.ws - 1));  // [nt*32]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int win = blockIdx.x, head = blockIdx.y, b = blockIdx.z;
  const int wy = win / p.nwx, wx = win - wy * p.nwx;
  const int nthreads = nt * 64;
  const int nrel = (2 * p.ws - 1) * (2 * p.ws - 1);

  // token t of this window -> its q/k/v row (the qkv bias for a padding token) and its
  // metadata (relative-position key term | region label << 16); pure index arithmetic, so
  // every global load of the prologue can be issued before anything is waited for
  auto token = [&](int t, int& m, int& src) -> const float* {
    const int py = t / p.ws, px = t - py * p.ws;
    const int y = wy * p.ws + py, x = wx * p.ws + px;
    int ys = y + p.shift, xs = x + p.shift;
    if (ys >= p.Hp) ys -= p.Hp;
    if (xs >= p.Wp) xs -= p.Wp;
    int label = 0;
    if (p.shift > 0) {
      const int rh = (y >= p.Hp - p.ws) + (y >= p.Hp - p.shift);
      const int rw = (x >= p.Wp - p.ws) + (x >= p.Wp - p.shift);
      label = rh * 3 + rw;
    }
    m = (py * (2 * p.ws - 1) + px) | (label << 16);
    src = (ys < p.H && xs < p.W) ? (b * p.H + ys) * p.W + xs : -1;
    return (src >= 0 ? p.qkv + (int64_t)src * p.ldqkv : p.qkv_bias) + head * 32;
  };

  // this lane's query
  const int myq = wave * 32 + li;
  const bool q_ok = myq < N;
  int qmeta, qsrc;
  const float* qrow = token(q_ok ? myq : N - 1, qmeta, qsrc);
  float4 qv[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) qv[u] = ld4(qrow + 16 * lh + 4 * u);
  // K / V rows of the window (8 float4 per row, <= 4 per thread), the bias row, metadata
  float4 kv[4], vv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int e = tid + i * nthreads;
    int m, src;
    const float* r = token(min(e >> 3, N - 1), m, src) + (e & 7) * 4;
    kv[i] = ld4(r + p.C);
    vv[i] = ld4(r + 2 * p.C);
  }
  float tv[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) tv[i] = p.table[(int64_t)head * nrel + min(tid + i * nthreads, nrel - 1)];
  if (tid < nt * 32) {
    int m = 0, src;
    if (tid < N) token(tid, m, src);
    meta[tid] = m;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
    if (tid + i * nthreads < nrel) tab[tid + i * nthreads] = tv[i] * LOG2E;
  for (int t = tid + 2 * nthreads; t < nrel; t += nthreads)
    tab[t] = p.table[(int64_t)head * nrel + t] * LOG2E;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int e = tid + i * nthreads, t = e >> 3;
    if (e < N8 * 8) {
      const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
      st4(Ks + wa_kpos(t, e & 7), t < N ? kv[i] : z);
      st4(Vs + wa_vrow(t) * 32 + (e & 7) * 4, t < N ? vv[i] : z);
    }
  }
  for (int e = tid + 4 * nthreads; e < N8 * 8; e += nthreads) {   // (not reached for ws <= 13)
    int m, src;
    const int t = e >> 3;
    const float* r = token(min(t, N - 1), m, src) + (e & 7) * 4;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    st4(Ks + wa_kpos(t, e & 7), t < N ? ld4(r + p.C) : z);
    st4(Vs + wa_vrow(t) * 32 + (e & 7) * 4, t < N ? ld4(r + 2 * p.C) : z);
  }
  const int qlabel = qmeta >> 16;
  const int qbase = (qmeta & 0xffff) + (p.ws - 1) * (2 * p.ws - 1) + (p.ws - 1);
  const float qscale = p.scale * LOG2E;
  const bool shifted = p.shift > 0;
  float qf[16];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    qf[4 * u + 0] = qv[u].x * qscale; qf[4 * u + 1] = qv[u].y * qscale;
    qf[4 * u + 2] = qv[u].z * qscale; qf[4 * u + 3] = qv[u].w * qscale;
  }
  __syncthreads();

  f32x16 o;
#pragma unroll
  for (int r = 0; r < 16; ++r) o[r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  for (int kt = 0; kt < nt; ++kt) {
    const int k0 = kt * 32;
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float4 a = ld4(Ks + wa_kpos(k0 + li, 4 * lh + u));
      s = mfma32(a.x, qf[4 * u + 0], s);
      s = mfma32(a.y, qf[4 * u + 1], s);
      s = mfma32(a.z, qf[4 * u + 2], s);
      s = mfma32(a.w, qf[4 * u + 3], s);
    }
    // scores are in log2 units (log2(e) is folded into the q scale and the LDS copy of
    // the bias row), so the softmax numerators are single v_exp_f32 instructions
    float tmax = -INFINITY;
    const bool tail = k0 + 32 > N;              // wave-uniform: only the last tile
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = k0 + mfma32_row(r, lh);
      const int km = meta[key];
      float v = s[r] + tab[qbase - (km & 0xffff)];
      if (shifted && (km >> 16) != qlabel) v += -100.f * LOG2E;
      if (tail) v = key < N ? v : -INFINITY;
      s[r] = v;
      tmax = fmaxf(tmax, v);
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float m_new = fmaxf(m_run, tmax);     // tile 0 always holds key 0: finite
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float pv = __builtin_amdgcn_exp2f(s[r] - m_new);
      s[r] = pv;
      psum += pv;
    }
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] *= alpha;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const float a = Vs[wa_vrow(k0 + mfma32_row(t, lh)) * 32 + li];
      o = mfma32(a, s[t], o);
    }
  }
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  if (p.s3_out) {
    // the accumulator is a transposed 32 x 32 block (lane: one token, 16 of the head's channels):
    // four v_permlane32_swap per 16 channels give the lane 8 consecutive ones = one S3 piece
    // element (csrc/gemm_s3.hip); tokens of a window are not consecutive rows: 16-byte stores
    const float inv = 1.f / l_tot;
    const bool live = q_ok && qsrc >= 0;
    const int row = live ? qsrc : 0;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float v8[8];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(o[8 * j + i] * inv),
                                                         __float_as_uint(o[8 * j + 4 + i] * inv), false, false);
        v8[i] = __uint_as_float(sw[0]);
        v8[4 + i] = __uint_as_float(sw[1]);
      }
      if (live) {
        s3_frag q0, q1, q2;
        s3_split8(v8, q0, q1, q2);
        uint4* dst = reinterpret_cast<uint4*>(p.out) +
                     ((int64_t)(row >> 5) * (p.C >> 4) + head * 2 + j) * 192 + lh * 32 + (row & 31);
        dst[0] = q0.u; dst[64] = q1.u; dst[128] = q2.u;
      }
    }
    return;
  }
  if (!q_ok || qsrc < 0) return;
  const float inv = 1.f / l_tot;
  float* op = p.out + (int64_t)qsrc * p.ldo + head * 32;
#pragma unroll
  for (int g = 0; g < 4; ++g)
    st4(op + 8 * g + 4 * lh, make_float4(o[4 * g + 0] * inv, o[4 * g + 1] * inv,
                                         o[4 * g + 2] * inv, o[4 * g + 3] * inv));
}

static int window_attention(const float* qkv, int64_t ldqkv, const float* qkv_bias,
                            const float* bias_table, float* out, int64_t ldo, int B, int H, int W,
                            int C, int heads, int ws, int shift, float scale, int s3_out,
                            void* stream);
extern "C" int pn_window_attention_f32(const float* qkv, int64_t ldqkv, const float* qkv_bias,
                                       const float* bias_table, float* out, int64_t ldo, int B,
                                       int H, int W, int C, int heads, int ws, int shift,
                                       float scale, void* stream) {
  return window_attention(qkv, ldqkv, qkv_bias, bias_table, out, ldo, B, H, W, C, heads, ws, shift,
                          scale, 0, stream);
}
extern "C" int pn_window_attention_s3_f32(const float* qkv, int64_t ldqkv, const float* qkv_bias,
                                          const float* bias_table, void* out_s3, int B, int H,
                                          int W, int C, int heads, int ws, int shift, float scale,
                                          void* stream) {
  return window_attention(qkv, ldqkv, qkv_bias, bias_table, (float*)out_s3, C, B, H, W, C, heads, ws,
                          shift, scale, 1, stream);
}
static int window_attention(const float* qkv, int64_t ldqkv, const float* qkv_bias,
                            const float* bias_table, float* out, int64_t ldo, int B, int H, int W,
                            int C, int heads, int ws, int shift, float scale, int s3_out,
                            void* stream) {
  if (!qkv || !qkv_bias || !bias_table || !out || B <= 0 || H <= 0 || W <= 0 || heads <= 0 ||
      C != heads * 32 || ws < 2 || ws * ws > WA_MAXN || shift < 0 || shift >= ws ||
      ldqkv < 3 * C || ldo < C || ((ldqkv | ldo) & 3))
    return PN_BAD_ARG;
  if (((uintptr_t)qkv | (uintptr_t)qkv_bias | (uintptr_t)out) & 15) return PN_BAD_ARG;
  if ((int64_t)B * H * W >= (1ll << 31)) return PN_BAD_ARG;
  WinP p{};
  p.qkv = qkv; p.qkv_bias = qkv_bias; p.table = bias_table; p.out = out;
  p.ldqkv = ldqkv; p.ldo = ldo; p.H = H; p.W = W; p.C = C; p.heads = heads; p.ws = ws;
  p.shift = shift; p.scale = scale; p.s3_out = s3_out;
  p.Hp = (H + ws - 1) / ws * ws; p.Wp = (W + ws - 1) / ws * ws;
  p.nwx = p.Wp / ws;
  const int nt = (ws * ws + 31) / 32;
  const int N8 = (ws * ws + 7) & ~7;
  int tail = (2 * ws - 1) * (2 * ws - 1) + nt * 32;          // table + metadata
  if (tail < (nt * 32 - N8) * 32) tail = (nt * 32 - N8) * 32;  // over-read of the last K tile
  const size_t lds = (size_t)(2 * N8 * 32 + tail) * 4;
  if (lds > 65536) return PN_BAD_ARG;
  hipLaunchKernelGGL(k_window_attn, dim3(p.nwx * (p.Hp / ws), heads, B), dim3(nt * 64), lds,
                     (hipStream_t)stream, p);
  return PN_LAUNCH_CHECK();
}
