// Row-chain kernel for the query side of the decoders (M = 100 ... 400 rows, C = 256).
//
// Between two attention kernels a decoder layer applies a short chain of row-wise dense
// operations to its <= a few hundred queries: out_proj + residual, LayerNorm, the next
// attention's input projection; after the FFN the three Linears of the mask-embedding
// MLP, the next layer's query projection, the class head (facebook_detr.py:311-432 glue,
// pairnet_head.py:216-243, :322-326).  As separate launches each of them is a 3 us
// kernel plus a kernel boundary on a latency-bound dependent chain (~16 launches per
// layer).  Here a workgroup owns 32 complete rows and runs the WHOLE chain with the
// intermediates in LDS: one launch per chain segment.
//
//   program = up to PN_CHAIN_MAX_OPS ops over three 32 x 256 LDS row buffers
//     LIN     dst = act((src [+ aadd for output columns >= add_from_col]) W^T + bias) [+ res]
//             K = 256, N <= 768; dst is an LDS buffer (N == 256) and / or global memory
//     LN      LayerNorm(256) of a buffer (k_layernorm256's arithmetic, bit for bit)
//     L2NORM  x / max(||x||, eps)       (k_l2norm256's arithmetic)
//
// 16 waves.  A LIN is cut into (32-column tile, K half) units, two adjacent waves per
// tile: each contracts 128 of the 256 k (64 x v_mfma_f32_32x32x2_f32, A fragments from
// LDS as ds_read_b128, W fragments straight from global / L2 like k_gemm_skinny), the odd
// wave parks its accumulator in LDS and the even wave adds it in fixed order
// (deterministic), applies the epilogue and writes the tile.
#include "common.h"

#define CH_LD 260   // LDS row stride in floats (256 + 4: conflict-free ds_read_b128 by row)

enum { CH_LIN = 0, CH_LN = 1, CH_L2NORM = 2 };

struct ChainOp {
  int kind, src, dst, res;      // LDS buffer ids 0..2; dst / res may be -1
  int N, relu, add_from_col, aadd_rows;
  const float* W; const float* bias; const float* aadd;
  const float* gamma; const float* beta;
  float* out; int64_t ldo;      // optional global destination, row-major
  float eps;
};

struct ChainP {
  int nops, M;
  const float* in0; int64_t ld0;   // -> buffer 0
  const float* in1; int64_t ld1;   // -> buffer 1 (optional)
  ChainOp op[PN_CHAIN_MAX_OPS];
};

__global__ __launch_bounds__(1024) void k_rowchain(const ChainP p) {
  __shared__ __attribute__((aligned(16))) float buf[3][32 * CH_LD];
  __shared__ __attribute__((aligned(16))) float red[8][1024];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int m0 = blockIdx.x * 32;

  // ---- load the input rows (clamped past M; their results are never stored) ----
  {
    const int r = tid >> 5, c = (tid & 31) * 8;      // 32 rows x 32 lanes x 8 floats
    const int gr = min(m0 + r, p.M - 1);
    const float* s0 = p.in0 + (int64_t)gr * p.ld0 + c;
    st4(&buf[0][r * CH_LD + c], ld4(s0));
    st4(&buf[0][r * CH_LD + c + 4], ld4(s0 + 4));
    if (p.in1) {
      const float* s1 = p.in1 + (int64_t)gr * p.ld1 + c;
      st4(&buf[1][r * CH_LD + c], ld4(s1));
      st4(&buf[1][r * CH_LD + c + 4], ld4(s1 + 4));
    }
  }
  __syncthreads();

  for (int oi = 0; oi < p.nops; ++oi) {
    const ChainOp& op = p.op[oi];
    if (op.kind == CH_LIN) {
      const float* A = buf[op.src];
      const int ntile = (op.N + 31) >> 5;
      const int nunit = ntile * 2;
      for (int u0 = 0; u0 < nunit; u0 += 16) {
        const int u = u0 + wave;
        const bool live = u < nunit;
        const int t = u >> 1, h = u & 1;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        if (live) {
          const int n0 = t * 32;
          const float* wrow = op.W + (int64_t)min(n0 + li, op.N - 1) * 256;
          const float* arow = A + li * CH_LD;
          const bool add = op.aadd && n0 >= op.add_from_col;
          const float* addrow = add ? op.aadd + (int64_t)((m0 + li) % op.aadd_rows) * 256 : nullptr;
          const int k0 = h * 128;
          // 64-deep blocks: 8 W + 8 A fragments (float4) in flight, then their 32 MFMAs; the
          // workgroup's 16 waves cap a lane at 128 registers
#pragma unroll
          for (int kb = 0; kb < 2; ++kb) {
            float4 wv[8], av[8];
#pragma unroll
            for (int s = 0; s < 8; ++s) wv[s] = ld4(wrow + k0 + kb * 64 + 8 * s + 4 * lh);
#pragma unroll
            for (int s = 0; s < 8; ++s) {
              av[s] = ld4(arow + k0 + kb * 64 + 8 * s + 4 * lh);
              if (add) av[s] = add4(av[s], ld4(addrow + k0 + kb * 64 + 8 * s + 4 * lh));
            }
#pragma unroll
            for (int s = 0; s < 8; ++s) {
              acc = mfma32(av[s].x, wv[s].x, acc);
              acc = mfma32(av[s].y, wv[s].y, acc);
              acc = mfma32(av[s].z, wv[s].z, acc);
              acc = mfma32(av[s].w, wv[s].w, acc);
            }
          }
          if (h == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) red[wave >> 1][r * 64 + lane] = acc[r];
          }
        }
        __syncthreads();
        if (live && h == 0) {
          const int col = t * 32 + li;
          const bool cok = col < op.N;
          const float bv = (op.bias && cok) ? op.bias[col] : 0.f;
          float* D = op.dst >= 0 ? buf[op.dst] : nullptr;
          const float* R = op.res >= 0 ? buf[op.res] : nullptr;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = mfma32_row(r, lh);
            float v = (acc[r] + red[wave >> 1][r * 64 + lane]) + bv;
            if (op.relu) v = fmaxf(v, 0.f);
            if (R) v += R[row * CH_LD + col];
            if (D && cok) D[row * CH_LD + col] = v;
            if (op.out && cok && m0 + row < p.M) op.out[(int64_t)(m0 + row) * op.ldo + col] = v;
          }
        }
        __syncthreads();
      }
    } else {
      // LN / L2NORM: one wave per row, two rows per wave (k_layernorm256 / k_l2norm256)
      const float* S = buf[op.src];
      float* D = op.dst >= 0 ? buf[op.dst] : nullptr;
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {
        const int row = wave * 2 + rr;
        const float4 v = ld4(S + row * CH_LD + lane * 4);
        float4 y;
        if (op.kind == CH_LN) {
          const float mean = wave_sum((v.x + v.y) + (v.z + v.w)) * (1.f / 256.f);
          const float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
          const float var = wave_sum((dx * dx + dy * dy) + (dz * dz + dw * dw)) * (1.f / 256.f);
          const float rstd = 1.f / sqrtf(var + op.eps);
          const float4 gg = ld4(op.gamma + lane * 4), bb = ld4(op.beta + lane * 4);
          y = make_float4(dx * rstd * gg.x + bb.x, dy * rstd * gg.y + bb.y,
                          dz * rstd * gg.z + bb.z, dw * rstd * gg.w + bb.w);
        } else {
          const float n = sqrtf(wave_sum((v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w)));
          const float d = fmaxf(n, op.eps);
          y = make_float4(v.x / d, v.y / d, v.z / d, v.w / d);
        }
        // (src and dst may be the same buffer: each lane rewrites exactly what it read)
        if (D) st4(D + row * CH_LD + lane * 4, y);
        if (op.out && m0 + row < p.M) st4(op.out + (int64_t)(m0 + row) * op.ldo + lane * 4, y);
      }
      __syncthreads();
    }
  }
}

extern "C" int pn_rowchain_f32(const pn_chain_desc* d, void* stream) {
  if (!d || !d->in0 || d->M <= 0 || d->nops <= 0 || d->nops > PN_CHAIN_MAX_OPS) return PN_BAD_ARG;
  if (d->ld0 < 256 || (d->ld0 & 3) || ((uintptr_t)d->in0 & 15)) return PN_BAD_ARG;
  if (d->in1 && (d->ld1 < 256 || (d->ld1 & 3) || ((uintptr_t)d->in1 & 15))) return PN_BAD_ARG;
  ChainP p{};
  p.nops = d->nops; p.M = d->M;
  p.in0 = d->in0; p.ld0 = d->ld0; p.in1 = d->in1; p.ld1 = d->ld1;
  for (int i = 0; i < d->nops; ++i) {
    const pn_chain_op& s = d->op[i];
    ChainOp& o = p.op[i];
    if (s.kind < 0 || s.kind > 2 || s.src < 0 || s.src > 2 || s.dst > 2 || s.res > 2) return PN_BAD_ARG;
    if (s.kind == CH_LIN) {
      if (!s.W || s.N <= 0 || s.N > 768 || ((uintptr_t)s.W & 15)) return PN_BAD_ARG;
      if (s.dst >= 0 && s.N != 256) return PN_BAD_ARG;       // an LDS destination is a full row
      if (s.dst == s.src || (s.res >= 0 && s.dst == s.res && s.dst >= 0 && false)) return PN_BAD_ARG;
      if (s.aadd && (s.aadd_rows <= 0 || ((uintptr_t)s.aadd & 15))) return PN_BAD_ARG;
      if (s.dst < 0 && !s.out) return PN_BAD_ARG;
    } else {
      if (s.kind == CH_LN && (!s.gamma || !s.beta)) return PN_BAD_ARG;
      if (s.out && ((s.ldo & 3) || ((uintptr_t)s.out & 15))) return PN_BAD_ARG;
    }
    if (s.out && s.ldo < (s.kind == CH_LIN ? s.N : 256)) return PN_BAD_ARG;
    o.kind = s.kind; o.src = s.src; o.dst = s.dst; o.res = s.res;
    o.N = s.N; o.relu = s.relu;
    o.add_from_col = s.aadd ? s.add_from_col : 0x7fffffff;
    o.aadd_rows = s.aadd_rows > 0 ? s.aadd_rows : 1;
    o.W = s.W; o.bias = s.bias; o.aadd = s.aadd; o.gamma = s.gamma; o.beta = s.beta;
    o.out = s.out; o.ldo = s.ldo; o.eps = s.eps;
  }
  hipLaunchKernelGGL(k_rowchain, dim3(pn_cdiv(d->M, 32)), dim3(1024), 0, (hipStream_t)stream, p);
  return PN_LAUNCH_CHECK();
}
