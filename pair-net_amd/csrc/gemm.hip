// fp32 contractions on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32,
// bitwise an fmaf chain).  Two kernels behind pn_gemm_f32 / pn_conv2d_nhwc_f32:
//
//   k_gemm_tile   LDS-tiled BMxBNx32, 4 waves, register-prefetched double-buffered
//                 LDS.  A rows come from a row-major matrix, a column-major matrix
//                 (an NCHW feature map read as [K][M]) or an on-the-fly im2col of
//                 a channel-last image (implicit-GEMM convolution).
//   k_gemm_skinny 32x32 output tile per workgroup, the 4 waves split K and reduce
//                 through LDS; operands go global->VGPR directly.  For the M~100
//                 query-side GEMMs of the decoders, where a 128-row tile would
//                 leave the chip empty.
//
// LDS layout [row][32+4]: a lane's MFMA operand for four consecutive k-steps is one
// ds_read_b128 (k-permutation: lanes 0-31 take k = kb..kb+3, lanes 32-63 take
// kb+4..kb+7 for BOTH operands, so the pairing inside each MFMA stays consistent);
// the 36-float stride makes the four 16-lane groups of a b128 read hit all 64 banks
// exactly once.
#include "common.h"
#include "gemm_common.h"

template <int BM, int BN, int WM, int WN, int AMODE, bool DB = true>
__device__ __forceinline__ void gemm_tile_body(const GemmP& p, const int m0, const int n0,
                                               const int bz, float* smem) {
  constexpr int BK = 32, LD = BK + 4;
  constexpr int WAVES_N = BN / WN;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int LDK = BM + 4;  // k-major A tile stride (A_COL)
  constexpr int A_ELEMS = (AMODE == A_COL) ? BK * LDK : BM * LD;
  constexpr int B_ELEMS = BN * LD;
  constexpr int STAGE = A_ELEMS + B_ELEMS;
  constexpr int NA = (BM * BK / 4) / 256;
  constexpr int NB = (BN * BK / 4) / 256;
  constexpr int QM = BM / 4;         // float4 per k-row of a column-major A tile
  constexpr int KSTEP = 256 / QM;    // k rows covered per pass
  static_assert((BM / WM) * WAVES_N == 4, "4 waves");

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const float* __restrict__ Aadd = (p.Aadd && n0 >= p.aadd_from_col) ? p.Aadd : nullptr;
  const float* __restrict__ A = p.A + (int64_t)bz * p.sA;
  const float* __restrict__ W = p.W + (int64_t)bz * p.sW;

  // ---- per-thread loader state (row indices do not change over k) ----
  const float* a_row[NA];
  const float* add_row[NA];
  int cy[NA], cx[NA];
  bool a_ok[NA];
  const int kc = (tid & 7) * 4;  // row-major / conv: k offset inside the tile
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    a_row[j] = nullptr; add_row[j] = nullptr; cy[j] = cx[j] = 0; a_ok[j] = false;
    if (AMODE != A_COL) {
      const int gm = m0 + (tid >> 3) + 32 * j;
      a_ok[j] = gm < p.M;
      if (AMODE == A_ROW) {
        a_row[j] = A + (int64_t)(a_ok[j] ? gm : 0) * p.lda;
        if (Aadd) add_row[j] = Aadd + (int64_t)((a_ok[j] ? gm : 0) % p.aadd_rows) * p.ldaadd;
      } else {
        cy[j] = gm / p.Wd;
        cx[j] = gm - cy[j] * p.Wd;
      }
    }
  }
  const float* w_row[NB];
  bool w_ok[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int gn = n0 + (tid >> 3) + 32 * j;
    w_ok[j] = gn < p.N;
    w_row[j] = W + (int64_t)(w_ok[j] ? gn : 0) * p.ldw;
  }

  float4 ra[NA], rb[NB];
  auto load_tile = [&](int kt) {
    const int k0 = kt * BK;
    if (AMODE == A_ROW) {
#pragma unroll
      for (int j = 0; j < NA; ++j) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a_ok[j] && k0 + kc < p.K) {
          v = ld4(a_row[j] + k0 + kc);
          if (Aadd) v = add4(v, ld4(add_row[j] + k0 + kc));
        }
        ra[j] = v;
      }
    } else if (AMODE == A_CONV) {
      const int tap = k0 / p.Cin;
      const int ci = k0 - tap * p.Cin + kc;
      const int dy = tap / p.KW - p.pad, dx = tap % p.KW - p.pad;
#pragma unroll
      for (int j = 0; j < NA; ++j) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        const int yy = cy[j] + dy, xx = cx[j] + dx;
        if (a_ok[j] && yy >= 0 && yy < p.H && xx >= 0 && xx < p.Wd && k0 + kc < p.K)
          v = ld4(A + ((int64_t)yy * p.Wd + xx) * p.Cin + ci);
        ra[j] = v;
      }
    } else {
      const int mc = m0 + (tid % QM) * 4;
#pragma unroll
      for (int j = 0; j < NA; ++j) {
        const int kk = k0 + tid / QM + KSTEP * j;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (kk < p.K) {
          const float* src = A + (int64_t)kk * p.lda + mc;
          if (p.a_vec && mc + 3 < p.M) {
            v = ld4(src);
          } else {
            if (mc + 0 < p.M) v.x = src[0];
            if (mc + 1 < p.M) v.y = src[1];
            if (mc + 2 < p.M) v.z = src[2];
            if (mc + 3 < p.M) v.w = src[3];
          }
        }
        ra[j] = v;
      }
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (w_ok[j] && k0 + kc < p.K) v = ld4(w_row[j] + k0 + kc);
      rb[j] = v;
    }
  };
  auto store_tile = [&](int buf) {
    float* sA = smem + buf * STAGE;
    float* sB = sA + A_ELEMS;
    if (AMODE == A_COL) {
#pragma unroll
      for (int j = 0; j < NA; ++j)
        st4(sA + (tid / QM + KSTEP * j) * LDK + (tid % QM) * 4, ra[j]);
    } else {
#pragma unroll
      for (int j = 0; j < NA; ++j)
        st4(sA + ((tid >> 3) + 32 * j) * LD + kc, ra[j]);
    }
#pragma unroll
    for (int j = 0; j < NB; ++j)
      st4(sB + ((tid >> 3) + 32 * j) * LD + kc, rb[j]);
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int mi = 0; mi < TM; ++mi)
#pragma unroll
    for (int ni = 0; ni < TN; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  const int nk = (p.K + BK - 1) / BK;
  auto compute = [&](int buf) {
    const float* sA = smem + buf * STAGE;
    const float* sB = sA + A_ELEMS;
#pragma unroll
    for (int kb = 0; kb < BK; kb += 8) {
      float a[TM][4], b[TN][4];
#pragma unroll
      for (int mi = 0; mi < TM; ++mi) {
        if (AMODE == A_COL) {
#pragma unroll
          for (int t = 0; t < 4; ++t)
            a[mi][t] = sA[(kb + t + 4 * lh) * LDK + wm * WM + mi * 32 + li];
        } else {
          const float4 v = ld4(sA + (wm * WM + mi * 32 + li) * LD + kb + 4 * lh);
          a[mi][0] = v.x; a[mi][1] = v.y; a[mi][2] = v.z; a[mi][3] = v.w;
        }
      }
#pragma unroll
      for (int ni = 0; ni < TN; ++ni) {
        const float4 v = ld4(sB + (wn * WN + ni * 32 + li) * LD + kb + 4 * lh);
        b[ni][0] = v.x; b[ni][1] = v.y; b[ni][2] = v.z; b[ni][3] = v.w;
      }
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
          for (int ni = 0; ni < TN; ++ni)
            acc[mi][ni] = mfma32(a[mi][t], b[ni][t], acc[mi][ni]);
    }
  };
  load_tile(0);
  if (DB) {
    store_tile(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      const int buf = kt & 1;
      if (kt + 1 < nk) load_tile(kt + 1);
      compute(buf);
      if (kt + 1 < nk) store_tile(buf ^ 1);
      __syncthreads();
    }
  } else {
    // one LDS stage: half the LDS, so twice as many workgroups are resident per CU to
    // fill the matrix pipe while others sit in their load / barrier phases
    for (int kt = 0; kt < nk; ++kt) {
      __syncthreads();
      store_tile(0);
      __syncthreads();
      if (kt + 1 < nk) load_tile(kt + 1);
      compute(0);
    }
  }

  // ---- epilogue: bias -> act -> residual ----
  float* __restrict__ C = p.C + (int64_t)bz * p.sC;
  const float* __restrict__ Res = p.Res ? p.Res + (int64_t)bz * p.sRes : nullptr;
#pragma unroll
  for (int ni = 0; ni < TN; ++ni) {
    const int col = n0 + wn * WN + ni * 32 + li;
    if (col >= p.N) continue;
    const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * WM + mi * 32 + mfma32_row(r, lh);
        if (row < p.M) {
          float v = acc[mi][ni][r] + bv;
          if (p.relu) v = fmaxf(v, 0.f);
          if (Res) v += Res[(int64_t)row * p.ldres + col];
          C[(int64_t)row * p.ldc + col] = v;
        }
      }
    }
  }
}

template <int BM, int BN, int AMODE, bool DB = true>
struct TileSmem {
  static constexpr int A_ELEMS = (AMODE == A_COL) ? 32 * (BM + 4) : BM * 36;
  static constexpr int FLOATS = (DB ? 2 : 1) * (A_ELEMS + BN * 36);
};

template <int BM, int BN, int WM, int WN, int AMODE, bool DB = true>
__global__ __launch_bounds__(256) void k_gemm_tile(const GemmP p) {
  __shared__ __attribute__((aligned(16))) float smem[TileSmem<BM, BN, AMODE, DB>::FLOATS];
  // 1-D launch over tiles; T enumerates tiles n-fastest, so one XCD's contiguous range
  // of T shares A row-panels (and, for the conv, image rows with their halo) in its L2
  const int nt = (p.N + BN - 1) / BN, mt = (p.M + BM - 1) / BM;
  const int T = xcd_tile_index(blockIdx.x, nt * mt);
  const int tm = T / nt, tn = T - tm * nt;
  gemm_tile_body<BM, BN, WM, WN, AMODE, DB>(p, tm * BM, tn * BN, blockIdx.z, smem);
}

// Several independent row-major GEMMs in ONE launch: the 64x64 tiles of all
// problems are enumerated together (n-tile fastest, so blocks that share an A panel
// are neighbours), which fills the chip where a single M x 256 problem leaves a
// ragged second round of workgroups.
#define GEMM_GROUP_MAX 16
struct GroupP {
  int n;
  int tile_start[GEMM_GROUP_MAX + 1];
  int mt[GEMM_GROUP_MAX], nt[GEMM_GROUP_MAX];
  GemmP p[GEMM_GROUP_MAX];
};

__global__ __launch_bounds__(256) void k_gemm_group(const GroupP g) {
  __shared__ __attribute__((aligned(16))) float smem[TileSmem<64, 64, A_ROW, false>::FLOATS];
  const int bid = xcd_tile_index(blockIdx.x, gridDim.x);
  int i = 0;
#pragma unroll
  for (int j = 1; j < GEMM_GROUP_MAX; ++j)
    if (j < g.n && bid >= g.tile_start[j]) i = j;
  const int local = bid - g.tile_start[i];
  const int per = g.mt[i] * g.nt[i];
  const int bz = local / per, r = local - bz * per;
  const int tm = r / g.nt[i], tn = r - tm * g.nt[i];
  gemm_tile_body<64, 64, 32, 32, A_ROW, false>(g.p[i], tm * 64, tn * 64, bz, smem);
}

// 32x32 output tile per workgroup; NW waves split K (wave w contracts a contiguous
// K/NW slice, rounded to 32) and the partial tiles are summed through LDS in wave
// order (deterministic).  Operands go global -> VGPR directly, one 32-deep chunk
// (4 float4 per operand per lane) prefetched ahead of the 16 MFMAs that consume the
// previous one, so a wave never waits on a single load round trip per MFMA group.
template <int AMODE, int NW>
__global__ __launch_bounds__(64 * NW) void k_gemm_skinny(const GemmP p) {
  __shared__ float red[NW * 1024];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32, bz = blockIdx.z;
  const float* __restrict__ A = p.A + (int64_t)bz * p.sA;
  const float* __restrict__ W = p.W + (int64_t)bz * p.sW;
  const int am = min(m0 + li, p.M - 1);
  const int wn_ = min(n0 + li, p.N - 1);
  const float* arow = (AMODE == A_ROW) ? A + (int64_t)am * p.lda : A + am;
  const float* addrow = (p.Aadd && n0 >= p.aadd_from_col)
                            ? p.Aadd + (int64_t)(am % p.aadd_rows) * p.ldaadd : nullptr;
  const float* wrow = W + (int64_t)wn_ * p.ldw;

  int ks = (p.K + NW - 1) / NW;
  ks = (ks + 31) & ~31;
  const int kbeg = min(wave * ks, p.K), kend = min(kbeg + ks, p.K);

  auto load_chunk = [&](int kc, float4 (&a)[4], float4 (&b)[4]) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int kk = kc + 8 * s + 4 * lh;
      const bool ok = kk < kend;
      float4 av = make_float4(0.f, 0.f, 0.f, 0.f), bv = av;
      if (ok) {
        bv = ld4(wrow + kk);
        if (AMODE == A_ROW) {
          av = ld4(arow + kk);
          if (addrow) av = add4(av, ld4(addrow + kk));
        } else {
          av.x = arow[(int64_t)(kk + 0) * p.lda];
          av.y = arow[(int64_t)(kk + 1) * p.lda];
          av.z = arow[(int64_t)(kk + 2) * p.lda];
          av.w = arow[(int64_t)(kk + 3) * p.lda];
        }
      }
      a[s] = av; b[s] = bv;
    }
  };
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  auto mma_chunk = [&](const float4 (&a)[4], const float4 (&b)[4]) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      acc = mfma32(a[s].x, b[s].x, acc);
      acc = mfma32(a[s].y, b[s].y, acc);
      acc = mfma32(a[s].z, b[s].z, acc);
      acc = mfma32(a[s].w, b[s].w, acc);
    }
  };

  float4 a0[4], b0[4], a1[4], b1[4];
  if (kbeg < kend) load_chunk(kbeg, a0, b0);
  for (int kc = kbeg; kc < kend; kc += 64) {
    const bool more1 = kc + 32 < kend;
    if (more1) load_chunk(kc + 32, a1, b1);
    mma_chunk(a0, b0);
    if (more1) {
      if (kc + 64 < kend) load_chunk(kc + 64, a0, b0);
      mma_chunk(a1, b1);
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wave * 1024 + mfma32_row(r, lh) * 32 + li] = acc[r];
  __syncthreads();

  float* __restrict__ C = p.C + (int64_t)bz * p.sC;
  const float* __restrict__ Res = p.Res ? p.Res + (int64_t)bz * p.sRes : nullptr;
  for (int e = tid; e < 1024; e += 64 * NW) {
    const int row = m0 + (e >> 5), col = n0 + (e & 31);
    if (row < p.M && col < p.N) {
      float v = red[e];
#pragma unroll
      for (int w = 1; w < NW; ++w) v += red[w * 1024 + e];
      if (p.bias) v += p.bias[col];
      if (p.relu) v = fmaxf(v, 0.f);
      if (Res) v += Res[(int64_t)row * p.ldres + col];
      C[(int64_t)row * p.ldc + col] = v;
    }
  }
}

template <int AMODE>
static int launch_skinny(const GemmP& p, int batch, hipStream_t s) {
  dim3 grid(pn_cdiv(p.N, 32), pn_cdiv(p.M, 32), batch);
  if (p.K <= 128)
    hipLaunchKernelGGL((k_gemm_skinny<AMODE, 4>), grid, dim3(256), 0, s, p);
  else if (p.K <= 256)
    hipLaunchKernelGGL((k_gemm_skinny<AMODE, 8>), grid, dim3(512), 0, s, p);
  else
    hipLaunchKernelGGL((k_gemm_skinny<AMODE, 16>), grid, dim3(1024), 0, s, p);
  return PN_LAUNCH_CHECK();
}


template <int BM, int BN, int WM, int WN, int AMODE>
static int launch_tile(const GemmP& p, int batch, hipStream_t s) {
  dim3 grid(pn_cdiv(p.N, BN) * pn_cdiv(p.M, BM), 1, batch);
  hipLaunchKernelGGL((k_gemm_tile<BM, BN, WM, WN, AMODE>), grid, dim3(256), 0, s, p);
  return PN_LAUNCH_CHECK();
}

static bool gemm_use_skinny(const pn_gemm_desc* d) {
  const int64_t tiles128 = (int64_t)pn_cdiv(d->M, 128) * pn_cdiv(d->N, 128) * d->batch;
  bool skinny = tiles128 < 96;
  if (d->flags & (PN_GEMM_FORCE_TILE | PN_GEMM_FORCE_TILE64 | PN_GEMM_FORCE_TILE128x64))
    skinny = false;
  if (d->flags & PN_GEMM_FORCE_SKINNY) skinny = true;
  return skinny;
}

extern "C" int pn_gemm_variant(const pn_gemm_desc* d) {
  if (!d || d->M <= 0 || d->N <= 0 || d->batch <= 0) return PN_BAD_ARG;
  const int col = (d->flags & PN_GEMM_A_COLMAJOR) ? 1 : 0;
  if (gemm_use_skinny(d)) return PN_GEMM_VARIANT_SKINNY + col;
  if ((d->flags & PN_GEMM_SPLIT_BF16) && !col) return PN_GEMM_VARIANT_SPLIT;
  if (d->flags & PN_GEMM_FORCE_TILE128x64) return PN_GEMM_VARIANT_TILE_128x64 + col;
  if (d->flags & PN_GEMM_FORCE_TILE) return PN_GEMM_VARIANT_TILE_128x128 + col;
  return PN_GEMM_VARIANT_TILE_64x64 + col;
}

int pn_fill_params(const pn_gemm_desc* d, GemmP* out) {
  if (!d || !d->A || !d->W || !d->C) return PN_BAD_ARG;
  if (d->M <= 0 || d->N <= 0 || d->K <= 0 || d->batch <= 0) return PN_BAD_ARG;
  const bool colmajor = d->flags & PN_GEMM_A_COLMAJOR;
  if (d->K % 4 || d->ldw % 4 || d->strideW % 4 || !aligned16(d->W)) return PN_BAD_ARG;
  if (!colmajor && (d->lda % 4 || d->strideA % 4 || !aligned16(d->A))) return PN_BAD_ARG;
  if (d->Aadd && (colmajor || d->ldaadd % 4 || !aligned16(d->Aadd) || d->aadd_rows <= 0))
    return PN_BAD_ARG;
  if (d->Aadd && (d->aadd_from_col < 0 || d->aadd_from_col % 64)) return PN_BAD_ARG;
  GemmP p{};
  p.A = d->A; p.Aadd = d->Aadd; p.W = d->W; p.bias = d->bias; p.Res = d->Res; p.C = d->C;
  p.lda = d->lda; p.ldaadd = d->ldaadd; p.ldw = d->ldw; p.ldres = d->ldres; p.ldc = d->ldc;
  p.sA = d->strideA; p.sW = d->strideW; p.sRes = d->strideRes; p.sC = d->strideC;
  p.M = d->M; p.N = d->N; p.K = d->K; p.aadd_rows = d->Aadd ? d->aadd_rows : 1;
  p.aadd_from_col = d->Aadd ? d->aadd_from_col : 0;
  p.relu = (d->flags & PN_GEMM_RELU) ? 1 : 0;
  p.a_vec = colmajor && d->lda % 4 == 0 && d->strideA % 4 == 0 && aligned16(d->A);
  *out = p;
  return 0;
}

extern "C" int pn_gemm_f32(const pn_gemm_desc* d, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  GemmP p;
  if (int rc = pn_fill_params(d, &p)) return rc;
  const bool colmajor = d->flags & PN_GEMM_A_COLMAJOR;
  if (gemm_use_skinny(d)) {
    return colmajor ? launch_skinny<A_COL>(p, d->batch, s) : launch_skinny<A_ROW>(p, d->batch, s);
  }
  if ((d->flags & PN_GEMM_SPLIT_BF16) && !colmajor)
    return pn_launch_gemm_split(p, d->batch, /*conv=*/false, (d->flags & PN_GEMM_FORCE_TILE) != 0, s);
  // Tile choice (measured on MI355X, tools/gemm_sweep.py): with K = 256..1024 and
  // M x N of a few hundred 128x128 tiles, the 64x64 tile wins everywhere (75-94 vs
  // 58-80 TFLOP/s): 4x more workgroups even out the last round over 256 CUs and four
  // of them fit a CU (36.8 KB LDS, 58 VGPRs).
  if (d->flags & PN_GEMM_FORCE_TILE128x64)
    return colmajor ? launch_tile<128, 64, 32, 64, A_COL>(p, d->batch, s)
                    : launch_tile<128, 64, 32, 64, A_ROW>(p, d->batch, s);
  if (d->flags & PN_GEMM_FORCE_TILE)
    return colmajor ? launch_tile<128, 128, 64, 64, A_COL>(p, d->batch, s)
                    : launch_tile<128, 128, 64, 64, A_ROW>(p, d->batch, s);
  // ... and with ONE LDS stage (18 KB, two barriers per k-tile) seven workgroups fit a
  // CU instead of four: +3-9 % on the same shapes (more waves to fill the matrix pipe
  // while others sit in their load / barrier phases).
  dim3 grid(pn_cdiv(p.N, 64) * pn_cdiv(p.M, 64), 1, d->batch);
  if (colmajor)
    hipLaunchKernelGGL((k_gemm_tile<64, 64, 32, 32, A_COL, false>), grid, dim3(256), 0, s, p);
  else
    hipLaunchKernelGGL((k_gemm_tile<64, 64, 32, 32, A_ROW, false>), grid, dim3(256), 0, s, p);
  return PN_LAUNCH_CHECK();
}

extern "C" int pn_gemm_group_f32(const pn_gemm_desc* d, int count, void* stream) {
  if (!d || count <= 0 || count > GEMM_GROUP_MAX) return PN_BAD_ARG;
  GroupP g{};
  g.n = count;
  int tiles = 0;
  for (int i = 0; i < count; ++i) {
    if (d[i].flags & PN_GEMM_A_COLMAJOR) return PN_BAD_ARG;
    if (int rc = pn_fill_params(&d[i], &g.p[i])) return rc;
    g.tile_start[i] = tiles;
    g.mt[i] = pn_cdiv(d[i].M, 64);
    g.nt[i] = pn_cdiv(d[i].N, 64);
    tiles += g.mt[i] * g.nt[i] * d[i].batch;
  }
  for (int i = count; i <= GEMM_GROUP_MAX; ++i) g.tile_start[i] = tiles;
  hipLaunchKernelGGL(k_gemm_group, dim3(tiles), dim3(256), 0, (hipStream_t)stream, g);
  return PN_LAUNCH_CHECK();
}

extern "C" int pn_conv2d_nhwc_f32(const float* in, const float* Wp, const float* bias,
                                  float* out, int B, int H, int W, int Cin, int Cout,
                                  int KH, int KW, int pad, int relu, int flags,
                                  void* stream) {
  if (!in || !Wp || !out || B <= 0 || H <= 0 || W <= 0) return PN_BAD_ARG;
  if (Cin % 32 || !aligned16(in) || !aligned16(Wp)) return PN_BAD_ARG;
  GemmP p{};
  p.A = in; p.W = Wp; p.bias = bias; p.C = out;
  p.M = H * W; p.N = Cout; p.K = KH * KW * Cin;
  p.lda = Cin; p.ldw = p.K; p.ldc = Cout;
  p.sA = (int64_t)H * W * Cin; p.sW = 0; p.sC = (int64_t)H * W * Cout;
  p.relu = relu ? 1 : 0; p.aadd_rows = 1;
  p.H = H; p.Wd = W; p.Cin = Cin; p.KW = KW; p.pad = pad;
  hipStream_t s = (hipStream_t)stream;
  if (flags & PN_GEMM_SPLIT_BF16) {
    return pn_launch_gemm_split(p, B, /*conv=*/true, (flags & PN_GEMM_FORCE_TILE) != 0, s);
  }
  // single LDS stage everywhere (more resident workgroups; 817 vs 944 us on the 3x3 FPN
  // conv); 128x128 for the 256-channel conv, 64x64 for the 64-channel Matrix Learner layer
  if (Cout <= 64) {
    dim3 grid(pn_cdiv(p.N, 64) * pn_cdiv(p.M, 64), 1, B);
    hipLaunchKernelGGL((k_gemm_tile<64, 64, 32, 32, A_CONV, false>), grid, dim3(256), 0, s, p);
  } else {
    dim3 grid(pn_cdiv(p.N, 128) * pn_cdiv(p.M, 128), 1, B);
    hipLaunchKernelGGL((k_gemm_tile<128, 128, 64, 64, A_CONV, false>), grid, dim3(256), 0, s, p);
  }
  return PN_LAUNCH_CHECK();
}

extern "C" int pn_abi_version(void) { return PN_ABI_VERSION; }
