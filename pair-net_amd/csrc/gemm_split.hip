// fp32-accurate contraction on the bf16 matrix pipe ("3 x bf16 split"), opt-in via
// PN_GEMM_SPLIT_BF16.
//
// Every fp32 operand is split EXACTLY into three bf16 pieces x = h + m + l (two
// mask/subtract steps: 8 + 8 + 8 significand bits), and a product is taken as the six
// partial products of order <= 2^-16:
//     a b ~= h_a h_b + (h_a m_b + m_a h_b) + (h_a l_b + l_a h_b + m_a m_b)
// each an exact bf16 x bf16 -> fp32 product accumulated in fp32 by
// v_mfma_f32_32x32x16_bf16.  The dropped terms are <= 3 * 2^-24 |a b|, the size of one
// fp32 rounding: the result is fp32-class (measured max error vs fp64 equal to or below
// the fp32 MFMA path's), though not bitwise the fmaf chain of the default path.  Six
// bf16 MFMAs at 32 cycles replace eight fp32 MFMAs at 64 cycles.
//
// BMxBNx32 tile, 4 waves; A/B staged in LDS as three bf16 planes [row][32 k]: 64-byte
// rows, the four 16-byte chunks of a row XOR-swizzled with (row >> 2) & 3, which makes
// both the b128 fragment reads and the b64 stores bank-conflict free.  (Pre-splitting
// the weights into planes was measured SLOWER: three 8-byte loads per thread instead
// of one 16-byte load cost more than the 20 VALU ops they save.)
#include "gemm_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// high halves of two dwords -> one dword {lo16 = x0 >> 16, hi16 = x1 >> 16}
__device__ __forceinline__ uint32_t pack_hi16(uint32_t x0, uint32_t x1) {
  return __builtin_amdgcn_perm(x1, x0, 0x07060302u);
}

__device__ __forceinline__ void split3_pack(const float4 v, uint2& hi, uint2& mid, uint2& lo) {
  const uint32_t M16 = 0xFFFF0000u;
  const float x[4] = {v.x, v.y, v.z, v.w};
  uint32_t h[4], m[4], l[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    h[i] = __float_as_uint(x[i]) & M16;
    const float r = x[i] - __uint_as_float(h[i]);   // exact
    m[i] = __float_as_uint(r) & M16;
    l[i] = __float_as_uint(r - __uint_as_float(m[i]));  // exact, <= 8 significant bits
  }
  hi = make_uint2(pack_hi16(h[0], h[1]), pack_hi16(h[2], h[3]));
  mid = make_uint2(pack_hi16(m[0], m[1]), pack_hi16(m[2], m[3]));
  lo = make_uint2(pack_hi16(l[0], l[1]), pack_hi16(l[2], l[3]));
}

template <int BM, int BN, int WM, int WN, int AMODE>
__global__ __launch_bounds__(256) void k_gemm_split(const GemmP p) {
  constexpr int BK = 32;
  constexpr int TM = WM / 32, TN = WN / 32, WAVES_N = BN / WN;
  constexpr int NA = BM / 32, NB = BN / 32;
  constexpr int ROWB = 64;
  constexpr int PLANE_A = BM * ROWB, PLANE_B = BN * ROWB;
  static_assert((BM / WM) * WAVES_N == 4, "4 waves");
  __shared__ __attribute__((aligned(16))) unsigned char smem[3 * (PLANE_A + PLANE_B)];
  unsigned char* sA = smem;                 // planes hi, mid, lo
  unsigned char* sB = smem + 3 * PLANE_A;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int nt = (p.N + BN - 1) / BN, mt = (p.M + BM - 1) / BM;
  const int T = xcd_tile_index(blockIdx.x, nt * mt);   // see common.h
  const int m0 = (T / nt) * BM, n0 = (T % nt) * BN, bz = blockIdx.z;
  const float* __restrict__ A = p.A + (int64_t)bz * p.sA;
  const float* __restrict__ W = p.W + (int64_t)bz * p.sW;
  const float* __restrict__ Aadd = (p.Aadd && n0 >= p.aadd_from_col) ? p.Aadd : nullptr;

  const float* a_row[NA];
  const float* add_row[NA];
  int64_t w_off[NB];
  int cy[NA], cx[NA];
  bool a_ok[NA], w_ok[NB];
  const int kc = (tid & 7) * 4;
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int gn = n0 + (tid >> 3) + 32 * j;
    w_ok[j] = gn < p.N;
    w_off[j] = (int64_t)(w_ok[j] ? gn : 0) * p.ldw;
  }
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    const int gm = m0 + (tid >> 3) + 32 * j;
    a_ok[j] = gm < p.M;
    a_row[j] = add_row[j] = nullptr; cy[j] = cx[j] = 0;
    if (AMODE == A_ROW) {
      a_row[j] = A + (int64_t)(a_ok[j] ? gm : 0) * p.lda;
      if (Aadd) add_row[j] = Aadd + (int64_t)((a_ok[j] ? gm : 0) % p.aadd_rows) * p.ldaadd;
    } else {
      cy[j] = gm / p.Wd;
      cx[j] = gm - cy[j] * p.Wd;
    }
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
    } else {
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
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const bool ok = w_ok[j] && k0 + kc < p.K;
      rb[j] = ok ? ld4(W + w_off[j] + k0 + kc) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto store_tile = [&]() {
    uint2 h, m, l;
    // thread (row, 4 k) -> 8 bytes of chunk (tid&7)>>1; rows 4 apart land on different
    // chunks (conflict-free b128 reads), consecutive rows on different bank halves
    const int wr = tid >> 3;
    const int wsub = (tid & 1) * 8, wch = (tid & 7) >> 1;
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const int r = wr + 32 * j;
      const int off = r * ROWB + ((wch ^ ((r >> 2) & 3)) << 4) + wsub;
      split3_pack(ra[j], h, m, l);
      *reinterpret_cast<uint2*>(sA + off) = h;
      *reinterpret_cast<uint2*>(sA + PLANE_A + off) = m;
      *reinterpret_cast<uint2*>(sA + 2 * PLANE_A + off) = l;
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int r = wr + 32 * j;
      const int off = r * ROWB + ((wch ^ ((r >> 2) & 3)) << 4) + wsub;
      split3_pack(rb[j], h, m, l);
      *reinterpret_cast<uint2*>(sB + off) = h;
      *reinterpret_cast<uint2*>(sB + PLANE_B + off) = m;
      *reinterpret_cast<uint2*>(sB + 2 * PLANE_B + off) = l;
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int mi = 0; mi < TM; ++mi)
#pragma unroll
    for (int ni = 0; ni < TN; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  const int nk = (p.K + BK - 1) / BK;
  load_tile(0);
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();           // previous tile's fragments fully read
    store_tile();
    __syncthreads();
    if (kt + 1 < nk) load_tile(kt + 1);
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      bf16x8 a[TM][3], b[TN][3];
      const int chunk = ((2 * st + lh) ^ ((li >> 2) & 3)) << 4;
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
        for (int t = 0; t < TM; ++t)
          a[t][pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(
              sA + pl * PLANE_A + (wm * WM + t * 32 + li) * ROWB + chunk));
#pragma unroll
        for (int t = 0; t < TN; ++t)
          b[t][pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(
              sB + pl * PLANE_B + (wn * WN + t * 32 + li) * ROWB + chunk));
      }
      // six partial products, smallest first; round-robin over the accumulators so
      // that consecutive MFMAs are independent
      constexpr int PA[6] = {2, 0, 1, 1, 0, 0};
      constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
          for (int ni = 0; ni < TN; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][PA[t]], b[ni][PB[t]],
                                                                  acc[mi][ni], 0, 0, 0);
    }
  }

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
          if (p.relu) v = fmaxf(v, 0.f);  // (relu_after: not offered by the split path)
          if (Res) v += Res[(int64_t)row * p.ldres + col];
          C[(int64_t)row * p.ldc + col] = v;
        }
      }
    }
  }
}

template <int BM, int BN, int WM, int WN, int AMODE>
static int launch(const GemmP& p, int batch, hipStream_t s) {
  dim3 grid(pn_cdiv(p.N, BN) * pn_cdiv(p.M, BM), 1, batch);
  hipLaunchKernelGGL((k_gemm_split<BM, BN, WM, WN, AMODE>), grid, dim3(256), 0, s, p);
  return PN_LAUNCH_CHECK();
}

int pn_launch_gemm_split(const GemmP& p, int batch, bool conv, bool big_tile, hipStream_t s) {
  if (conv) return big_tile ? launch<128, 128, 64, 64, A_CONV>(p, batch, s)
                            : launch<64, 64, 32, 32, A_CONV>(p, batch, s);
  return big_tile ? launch<128, 128, 64, 64, A_ROW>(p, batch, s)
                  : launch<64, 64, 32, 32, A_ROW>(p, batch, s);
}
