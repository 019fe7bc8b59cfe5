// fp32 contractions on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32,
// bitwise an fmaf chain).  The kernels behind pn_gemm_f32 / pn_gemm_group_f32 /
// pn_conv2d_nhwc(_ex)_f32 (the ResNet stem has its own kernel, stem.hip):
//
//   k_gemm_tile   persistent workgroups (4 waves) walking BMxBN output tiles, 32-deep
//                 k-chunks staged global -> registers -> one LDS stage, MFMA operand
//                 fragments double-buffered in registers.  A rows come from a
//                 row-major matrix, a column-major matrix (an NCHW feature map read as
//                 [K][M]) or an on-the-fly im2col of a channel-last image (implicit-GEMM
//                 convolution, any stride).
//   k_splitk_reduce  second pass of the deterministic split-K used when a problem has
//                 too few output tiles to fill 256 CUs.
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

// Resident workgroups per CU the persistent tile kernels are sized for.  64x64 tiles: 5 since
// round 4 (rounds 1-3: 4).  In the pipelined bench, three alternating runs per build on one
// box (tools/bench_variant.py): 4 everywhere 206.1 / 206.2 / 206.1 images/s; 5 for the
// instantiations that fit 96 VGPRs without spilling (plain row-major A, implicit-GEMM conv) and
// 4 for the rest 207.6 / 207.5 / 207.3; 5 everywhere (positional-addend, column-major and
// grouped instantiations then spill 4-8 registers) 208.3 / 207.9 / 208.0: +0.9 %.  1216
// instead of 960 workgroup slots turn the 1056- and 2088-tile backbone launches into one and
// two full rounds, and a fifth wave per SIMD covers more of each tile's load latency.  Six
// would need 80 VGPRs (9-26 spilled).  Build-time constants so that a variant library can be
// A/B'd against the product one; > 4 also bounds the kernel's registers to that many waves
// per SIMD.  Results do not depend on them (a tile's arithmetic is the same wherever it runs).
#ifndef PN_GEMM_WGS64
#define PN_GEMM_WGS64 5
#endif
#ifndef PN_GEMM_WGS128x64           // 128x64 tiles (126 VGPRs: up to 4 waves per SIMD, 27 KB of LDS)
#define PN_GEMM_WGS128x64 3
#endif
#ifndef PN_GEMM_BIGM_128x64         // 1: plain row-major launches with M >= 16384, N >= 256 take
#define PN_GEMM_BIGM_128x64 0       // 128x64 tiles (power experiment, LABNOTES.md 6.0-r4)
#endif
#ifndef PN_GEMM_WGS64_SPILLING      // the instantiations that spill a few registers at 5
#define PN_GEMM_WGS64_SPILLING 5
#endif

// -DPN_GEMM_LDS2=1 (round-5 experiment): TWO LDS stages and ONE barrier per 32-deep chunk -- the
// chunk after next goes global -> registers and the next chunk registers -> the other stage
// while the current stage is multiplied -- instead of one stage with two barriers per chunk.
// 36.9 KB per 64x64 workgroup: four resident workgroups per CU instead of five.  See LABNOTES
// R5.8 for the measurement.
#ifndef PN_GEMM_LDS2
#define PN_GEMM_LDS2 0
#endif

template <int BM, int BN, int AMODE>
struct TileSmem {
  static constexpr int A_ELEMS = (AMODE == A_COL) ? 32 * (BM + 4) : BM * 36;
  static constexpr int STAGE = A_ELEMS + BN * 36;
  static constexpr int FLOATS = STAGE * (PN_GEMM_LDS2 ? 2 : 1);
};

// -DPN_GEMM_W2=1 (round-4 experiment, NOT adopted): the default 64x64 launches run as TWO waves
// of 64x32 (two accumulators per wave, a quarter fewer LDS fragment reads per flop) instead of
// four waves of 32x32 at 5 workgroups = 20 waves per CU.  149 VGPRs: 6 workgroups = 12 waves
// per CU without spilling, 8 = 16 waves with 72-224 B of scratch.  Exact on the integer-operand
// tests; stand-alone within -4 .. +3 % on the M = 21 950 shapes and 5-20 % slower on the
// backbone's; pipelined 191.3 (6 / CU) and 175.6 (8 / CU) against 207.4 images/s.
#ifndef PN_GEMM_W2
#define PN_GEMM_W2 0
#endif
#ifndef PN_GEMM_W2_WGS
#define PN_GEMM_W2_WGS 8
#endif

template <int BM, int BN, int AMODE, bool ADD>
struct TileWgs {   // resident workgroups per CU this instantiation is sized (and register-bounded) for
  static constexpr int value =
      (BM * BN > 128 * 64) ? 2 : (BM * BN > 64 * 64) ? PN_GEMM_WGS128x64
      : ((AMODE == A_ROW && !ADD) || AMODE == A_CONV) ? PN_GEMM_WGS64 : PN_GEMM_WGS64_SPILLING;
  static constexpr int min_waves = (BM * BN <= 64 * 64 && value > 4) ? value : 1;
};

struct TileRef {
  int pi;  // problem index (grouped launches), resolved through Locator::P
  int bz, tm, tn;
};

// Persistent tile loop.  A workgroup walks a strided subsequence of its XCD's contiguous
// tile range (xcd_tile_index's partition: neighbouring tiles share operand panels in that
// XCD's L2).  Per 32-deep k-chunk: barrier, registers -> LDS, barrier, global loads of
// the NEXT chunk into registers, 16*TM*TN MFMAs fed by register-double-buffered
// ds_read_b128 fragments.  The next chunk after a tile's last one is the first chunk of
// the workgroup's next tile, so there is no per-tile load prologue, and the epilogue's
// stores drain under the next tile's MFMAs.  (Measured on MI355X with
// tools/gemm_probe.hip: 83-96 -> 99-112 TFLOP/s on the encoder shapes versus one
// workgroup per tile; the matrix pipe alone peaks at ~140.)
template <int BM, int BN, int WM, int WN, int AMODE, bool ADD, typename Locator, int EPI = 0>
__device__ __forceinline__ void gemm_persistent(const Locator& loc, const int ntiles,
                                                float* smem) {
  constexpr int BK = 32, LD = BK + 4;
  constexpr int WAVES_N = BN / WN;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int LDK = BM + 4;  // k-major A tile stride (A_COL)
  constexpr int A_ELEMS = TileSmem<BM, BN, AMODE>::A_ELEMS;
  constexpr int NT = 64 * (BM / WM) * WAVES_N;   // threads: one wave per WM x WN sub-tile
  constexpr int RPP = NT / 8;        // tile rows staged per pass (8 lanes per 32-float row)
  constexpr int NA = (BM * BK / 4) / NT;
  constexpr int NB = (BN * BK / 4) / NT;
  constexpr int QM = BM / 4;         // float4 per k-row of a column-major A tile
  constexpr int KSTEP = NT / QM;     // k rows covered per pass
  static_assert(NT == 128 || NT == 256 || NT == 512, "2, 4 or 8 waves");

  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per = gridDim.x >> 3;
  const int q8 = ntiles >> 3, r8 = ntiles & 7;
  const int base = (xcd < r8) ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  const int cnt = q8 + (xcd < r8 ? 1 : 0);
  if (slot >= cnt) return;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int kc = (tid & 7) * 4;  // row-major / conv: k offset inside the chunk
  float* const sA = smem;
  float* const sB = smem + A_ELEMS;

  // ---- loader state: describes the tile whose chunks are being LOADED ----
  // Addresses are (buffer descriptor on the per-batch operand base) + (per-lane 32-bit
  // byte offset, fixed per tile) + (wave-uniform k offset): the K loop carries no
  // per-lane address arithmetic.  pn_fill_params guarantees that every per-batch
  // operand spans < 2^29 elements.
  // Every load is UNCONDITIONAL (hipcc serialises predicated loads: a branch and a full
  // vmcnt wait per element): out-of-range rows are clamped to the last valid row (their
  // products land in output rows / columns that are never stored); a ragged K tail and
  // the conv halo read a clamped address and are zeroed when the chunk goes to LDS.
  int lpi = 0, lK = 0, lkt0 = 0;   // problem, its K, first chunk of this split
  const float* bA = nullptr;      // batch base of A
  const float* bW = nullptr;
  const float* bAdd = nullptr;
  __amdgpu_buffer_rsrc_t rA = make_rsrc(nullptr), rW = rA, rAdd = rA;
  unsigned a_off[NA], add_off[ADD ? NA : 1], w_off[NB];   // BYTE offsets (A_COL: element m)
  int cy[NA], cx[NA];
  unsigned a_tap[AMODE == A_CONV ? NA : 1];   // conv: a_off shifted to the current tap
  unsigned a_off1[AMODE == A_COL ? NA : 1][4];  // column-major, unaligned rows: per element
  unsigned in_mask = 0;  // conv: bit j = the current tap of row j lies inside the image
  bool l_add = false;   // the positional add feeds this tile's columns
  bool l_vec = false;   // column-major A: whole tile inside M and 16-byte aligned rows
  auto set_tile = [&](const TileRef& tr) {
    const GemmP& p = loc.P(tr.pi);
    lpi = tr.pi;
    lK = p.K;
    const int m0 = tr.tm * BM, n0 = tr.tn * BN;
    const int ks = max(p.ksplit, 1);
    const int bb = tr.bz / ks;                 // batch index (bz = bb * ksplit + split)
    lkt0 = (tr.bz - bb * ks) * p.split_chunks;
    bA = p.A + (int64_t)bb * p.sA;
    bW = p.W + (int64_t)bb * p.sW;
    bAdd = p.Aadd ? p.Aadd : bA;   // (without an addend it only has to be readable)
    rA = make_rsrc(bA);
    rW = make_rsrc(bW);
    if (ADD) rAdd = make_rsrc(bAdd);
    l_add = p.Aadd && n0 >= p.aadd_from_col;
    l_vec = p.a_vec && m0 + BM <= p.M;
    if (AMODE == A_COL) {
      // lane covers 4 consecutive m of k-row (tid / QM + KSTEP * j) of the chunk
      const int mq = m0 + (tid % QM) * 4;
#pragma unroll
      for (int j = 0; j < NA; ++j) {
        const unsigned krow = (unsigned)(min(tid / QM + KSTEP * j, p.K - 1)) * (unsigned)p.lda;
        a_off[j] = (krow + (unsigned)min(mq, p.M - 1)) * 4u;
#pragma unroll
        for (int e = 0; e < 4; ++e) a_off1[j][e] = (krow + (unsigned)min(mq + e, p.M - 1)) * 4u;
      }
    } else {
#pragma unroll
      for (int j = 0; j < NA; ++j) {
        const int gm = min(m0 + (tid >> 3) + RPP * j, p.M - 1);
        if (AMODE == A_CONV) {
          // output pixel -> input coordinates of tap (0, 0) + pad
          const int oy = gm / p.Wo, ox = gm - oy * p.Wo;
          cy[j] = oy * p.stride;
          cx[j] = ox * p.stride;
          a_off[j] = ((unsigned)(cy[j] * p.Wd + cx[j]) * (unsigned)p.Cin + kc) * 4u;
        } else {
          a_off[j] = ((unsigned)gm * (unsigned)p.lda + kc) * 4u;
        }
        if (ADD) add_off[j] = ((unsigned)(p.Aadd ? gm % p.aadd_rows : gm) *
                                   (unsigned)(p.Aadd ? p.ldaadd : p.lda) + kc) * 4u;
      }
    }
    if constexpr (EPI == 1) {
      const StencilP& st = loc.st;
      // stencil-mask mode: tile column c <-> (wave sub-tile c / 32, tap (c % 32) / 8, key
      // 16 tn + 8 (c / 32) + c % 8): the four taps of a key sit in ONE wave's 32 columns, 8
      // lanes apart, so the epilogue blends them with three lane shuffles.  W rows are
      // tap-major (tap t of key p at row t * Nk + p, pn_bilinear_stencil_rows_f32).
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int c = (tid >> 3) + RPP * j;
        const int key = min(tr.tn * 16 + (c >> 5) * 8 + (c & 7), st.nk - 1);
        w_off[j] = ((unsigned)(((c >> 3) & 3) * st.nk + key) * (unsigned)p.ldw + kc) * 4u;
      }
    } else {
#pragma unroll
      for (int j = 0; j < NB; ++j)
        w_off[j] = ((unsigned)min(n0 + (tid >> 3) + RPP * j, p.N - 1) * (unsigned)p.ldw + kc) * 4u;
    }
  };

  // Raw loaded chunk + what store_chunk has to do to it.  Nothing here may CONSUME a
  // loaded value at load time (a select or add right after the load makes the wave wait
  // out the whole memory latency before its MFMAs): the positional add, the K-tail and
  // halo zeroing all happen when the chunk is written to LDS, one chunk later.
  float4 ra[NA], rad[ADD ? NA : 1], rb[NB];
  bool st_ragged = false, st_add = false;
  int st_k0 = 0;
  unsigned st_in = 0;
  auto load_chunk = [&](int kt) {
    const int k0 = (lkt0 + kt) * BK;
    // a ragged last chunk (K % 32 != 0, rare) re-reads the previous 32 columns' address
    // range shifted back to stay in bounds; store_chunk zeroes what lies past K
    const bool ragged = k0 + BK > lK;
    st_ragged = ragged;
    st_add = l_add;
    st_k0 = k0;
    if (AMODE == A_COL) {
      // element (k, m) at k * lda + m: voffset = (k row inside the chunk) * lda + m,
      // soffset = k0 * lda (a ragged last chunk loads chunk 0 instead and is re-read
      // with bounds in store_chunk)
      const GemmP& p = loc.P(lpi);
      const int soff = (ragged ? 0 : k0) * (int)p.lda * 4;
      if (l_vec) {
#pragma unroll
        for (int j = 0; j < NA; ++j) ra[j] = buf_ld4(rA, a_off[j], soff);
      } else {
#pragma unroll
        for (int j = 0; j < NA; ++j) {
          ra[j].x = buf_ld1(rA, a_off1[j][0], soff);
          ra[j].y = buf_ld1(rA, a_off1[j][1], soff);
          ra[j].z = buf_ld1(rA, a_off1[j][2], soff);
          ra[j].w = buf_ld1(rA, a_off1[j][3], soff);
        }
      }
    } else if (AMODE == A_CONV) {
      const GemmP& p = loc.P(lpi);
      const int tap = k0 / p.Cin;
      const int ci0 = k0 - tap * p.Cin;
      const int dy = tap / p.KW - p.pad, dx = tap % p.KW - p.pad;
      if (ci0 == 0 || kt == 0) {   // new tap, or first chunk of a tile / K split (uniform):
                                   // which rows' taps fall inside the image, and their
                                   // per-lane offsets; the channel advance rides in soffset
        in_mask = 0;
        const int tap_off = (dy * p.Wd + dx) * p.Cin;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
          const int yy = cy[j] + dy, xx = cx[j] + dx;
          const bool in = yy >= 0 && yy < p.H && xx >= 0 && xx < p.Wd;
          in_mask |= (in ? 1u : 0u) << j;
          // outside: read the (valid) centre pixel, zeroed when the chunk goes to LDS
          a_tap[j] = a_off[j] + (unsigned)(in ? tap_off : 0) * 4u;
        }
      }
      st_in = in_mask;
#pragma unroll
      for (int j = 0; j < NA; ++j) ra[j] = buf_ld4(rA, a_tap[j], ci0 * 4);
    } else {
#pragma unroll
      for (int j = 0; j < NA; ++j) ra[j] = buf_ld4(rA, a_off[j], ragged ? 0 : k0 * 4);
      if (ADD) {
#pragma unroll
        for (int j = 0; j < NA; ++j)
          rad[j] = buf_ld4(rAdd, add_off[j], ragged ? 0 : k0 * 4);
      }
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) rb[j] = buf_ld4(rW, w_off[j], ragged ? 0 : k0 * 4);
  };
  auto store_chunk = [&](const int so = 0) {     // so: float offset of the LDS stage written
    float* const sA = smem + so;
    float* const sB = smem + so + A_ELEMS;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      if (ADD && st_add) ra[j] = add4(ra[j], rad[j]);
      if (AMODE == A_CONV && !((st_in >> j) & 1u)) ra[j] = zero4;
    }
    if (st_ragged) {   // uniform, rare: the ragged chunk was loaded from k = 0; reload it
      // synchronously with per-element bounds (slow path, correctness only)
      const GemmP& p = loc.P(lpi);
#pragma unroll
      for (int j = 0; j < NA; ++j) {
        if (AMODE == A_COL) {
          const int kk = st_k0 + tid / QM + KSTEP * j;
          const int mq = (int)(a_off1[j][0] / 4u) - (int)min(tid / QM + KSTEP * j, lK - 1) * (int)p.lda;
          float4 v = zero4;
          if (kk < lK) {
            const float* src = bA + (int64_t)kk * p.lda;
            v.x = src[min(mq + 0, p.M - 1)];
            v.y = src[min(mq + 1, p.M - 1)];
            v.z = src[min(mq + 2, p.M - 1)];
            v.w = src[min(mq + 3, p.M - 1)];
          }
          ra[j] = v;
        } else if (AMODE == A_ROW) {
          const bool ok = st_k0 + kc < lK;
          float4 v = ok ? ld4(bA + st_k0 + a_off[j] / 4u) : zero4;
          if (ADD && st_add && ok) v = add4(v, ld4(bAdd + st_k0 + add_off[j] / 4u));
          ra[j] = v;
        } else {
          ra[j] = zero4;   // conv: Cin % 32 == 0 is required, so K is never ragged
        }
      }
      (void)p;
#pragma unroll
      for (int j = 0; j < NB; ++j)
        rb[j] = st_k0 + kc < lK ? ld4(bW + st_k0 + w_off[j] / 4u) : zero4;
    }
    if (AMODE == A_COL) {
#pragma unroll
      for (int j = 0; j < NA; ++j)
        st4(sA + (tid / QM + KSTEP * j) * LDK + (tid % QM) * 4, ra[j]);
#pragma unroll
      for (int j = 0; j < NB; ++j) st4(sB + ((tid >> 3) + RPP * j) * LD + kc, rb[j]);
    } else {
#pragma unroll
      for (int j = 0; j < NA; ++j) st4(sA + ((tid >> 3) + RPP * j) * LD + kc, ra[j]);
#pragma unroll
      for (int j = 0; j < NB; ++j) st4(sB + ((tid >> 3) + RPP * j) * LD + kc, rb[j]);
    }
  };

  f32x16 acc[TM][TN];
  // MFMA operand fragments, double-buffered in registers: the ds_reads of k-step kb+1
  // are in flight while the MFMAs of kb issue
  const float* fA0 = (AMODE == A_COL) ? sA + (4 * lh) * LDK + wm * WM + li
                                      : sA + (wm * WM + li) * LD + 4 * lh;
  const float* fB0 = sB + (wn * WN + li) * LD + 4 * lh;
  int rso = 0;     // float offset of the LDS stage being read (PN_GEMM_LDS2)
  auto read_frag = [&](int kb, float4 (&fa)[TM], float4 (&fb)[TN]) {
    const float* const fA = fA0 + rso;
    const float* const fB = fB0 + rso;
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
      if (AMODE == A_COL) {
        const float* s = fA + (kb * 8) * LDK + mi * 32;
        fa[mi] = make_float4(s[0], s[LDK], s[2 * LDK], s[3 * LDK]);
      } else {
        fa[mi] = ld4(fA + mi * 32 * LD + kb * 8);
      }
    }
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) fb[ni] = ld4(fB + ni * 32 * LD + kb * 8);
  };
  auto compute = [&]() {
    float4 fa[2][TM], fb[2][TN];
    read_frag(0, fa[0], fb[0]);
#pragma unroll
    for (int kb = 0; kb < BK / 8; ++kb) {
      const int cur = kb & 1;
      if (kb + 1 < BK / 8) read_frag(kb + 1, fa[cur ^ 1], fb[cur ^ 1]);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
          for (int ni = 0; ni < TN; ++ni) {
            const float4 av = fa[cur][mi], bv = fb[cur][ni];
            acc[mi][ni] = mfma32(t == 0 ? av.x : t == 1 ? av.y : t == 2 ? av.z : av.w,
                                 t == 0 ? bv.x : t == 1 ? bv.y : t == 2 ? bv.z : bv.w,
                                 acc[mi][ni]);
          }
    }
  };

  TileRef cur = loc(base + slot);
  set_tile(cur);
  load_chunk(0);
#if PN_GEMM_LDS2
  constexpr int STAGE = TileSmem<BM, BN, AMODE>::STAGE;
  int wso = 0;                     // stage the NEXT store_chunk writes
  {
    // prologue: chunk 0 -> stage 0; the chunk after it -> registers
    const GemmP& p0 = loc.P(cur.pi);
    int nk0 = (p0.K + BK - 1) / BK;
    if (p0.ksplit > 1) nk0 = min(p0.split_chunks, nk0 - (cur.bz % p0.ksplit) * p0.split_chunks);
    store_chunk(0);
    wso = STAGE;
    if (nk0 > 1) {
      load_chunk(1);
    } else {
      set_tile(loc(base + min(slot + per, cnt - 1)));
      load_chunk(0);
    }
    __syncthreads();
  }
#endif
  for (int t = slot; t < cnt; t += per) {
    const GemmP& p = loc.P(cur.pi);
    // the tile after this one (clamped: the last tile re-loads its own first chunk, unused)
    const TileRef nxt = loc(base + min(t + per, cnt - 1));
    int nk = (p.K + BK - 1) / BK;
    if (p.ksplit > 1) {
      const int sidx = cur.bz % p.ksplit;
      nk = min(p.split_chunks, nk - sidx * p.split_chunks);
    }
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
      for (int ni = 0; ni < TN; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
#if PN_GEMM_LDS2
    // Stage rso holds chunk kt (visible since the last barrier), the registers hold the chunk
    // after it (loaded one iteration ago): it goes to the OTHER stage -- which every wave
    // finished reading before the last barrier -- the chunk after that is requested, and only
    // then the current stage is multiplied.  One barrier per chunk.
    for (int kt = 0; kt < nk; ++kt) {
      store_chunk(wso);
      // the chunk two ahead: kt + 2 of this tile, or chunk 0 / 1 of the next one
      if (kt + 2 < nk) {
        load_chunk(kt + 2);
      } else if (kt + 2 == nk) {       // registers held this tile's last chunk: next tile starts
        set_tile(nxt);
        load_chunk(0);
      } else {                          // kt + 1 == nk: registers held the next tile's chunk 0
        const GemmP& pn = loc.P(nxt.pi);
        int nkn = (pn.K + BK - 1) / BK;
        if (pn.ksplit > 1) nkn = min(pn.split_chunks, nkn - (nxt.bz % pn.ksplit) * pn.split_chunks);
        if (nkn > 1) {
          load_chunk(1);
        } else {                        // single-chunk tiles: the tile after the next one
          set_tile(loc(base + min(t + 2 * per, cnt - 1)));
          load_chunk(0);
        }
      }
      compute();
      __syncthreads();
      rso = wso;
      wso = STAGE - wso;
    }
#else
    for (int kt = 0; kt < nk; ++kt) {
      __syncthreads();
      store_chunk();
      __syncthreads();
      const bool last = kt + 1 == nk;
      if (last) set_tile(nxt);
      load_chunk(last ? 0 : kt + 1);
      compute();
    }
#endif
    // ---- epilogue: bias -> act -> residual.  The residual values of a 32x32
    // accumulator are fetched as 16 independent loads (clamped, unconditional) before
    // any is used: one memory round trip per accumulator instead of sixteen. ----
    const int m0 = cur.tm * BM, n0 = cur.tn * BN;
    if constexpr (EPI == 1) {
      const StencilP& st = loc.st;
      // ---- stencil-mask epilogue: blend (the ONE tap_blend every resampling kernel uses),
      // threshold, pack; nothing of C reaches memory.  Lane (half lh, column li): key slot
      // pl = li % 8 of this wave's 8 keys; register r is query row mfma32_row(r, lh). ----
      static_assert(EPI == 0 || (TM == 1 && TN == 1 && BN == 64), "stencil mode: 64x64 tiles of 32x32 waves");
      const int pl = li & 7;
      const int key0 = cur.tn * 16 + wn * 8;
      const int key = key0 + pl;
      const bool kvalid = key < st.nk;
      const int kk = min(key, st.nk - 1);
      const int oy = kk / st.wo, ox = kk - oy * st.wo;
      const Tap ty = make_tap(oy, st.hi, st.ho), tx = make_tap(ox, st.wi, st.wo);
      const int src = (lane & 32) | pl;
      const int nwords = (st.nk + 31) / 32;
      const int nvalid = min(max(st.nk - key0, 0), 8);
      const unsigned vm = (1u << nvalid) - 1u;
      const int rbase = m0 + wm * WM;
      const bool writer = li == 0;                     // lanes 0 and 32: one per row half
      const bool last_tile = (cur.tn + 1) * 16 >= st.nk && wn == WAVES_N - 1;
      // (stores only inside the loop -- fire and forget; a LOAD of the row flag per register
      // would put 16 dependent round trips into every tile's epilogue, and keeping the 16
      // bytes for a store block afterwards spills at this kernel's 96-register bound)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float a = acc[0][0][r];
        const float v00 = __shfl(a, src, 64), v01 = __shfl(a, src + 8, 64);
        const float v10 = __shfl(a, src + 16, 64), v11 = __shfl(a, src + 24, 64);
        const float res = tap_blend(ty, tx, v00, v01, v10, v11);
        const unsigned long long bal = __ballot(kvalid && res < 0.f);
        const int row = rbase + mfma32_row(r, lh);
        if (writer && row < p.M) {
          const unsigned byte = (unsigned)(bal >> (32 * lh)) & 0xffu;
          const int64_t grow = (int64_t)cur.bz * p.M + row;
          unsigned char* bp = reinterpret_cast<unsigned char*>(st.bits + grow * nwords);
          bp[cur.tn * 2 + wn] = (unsigned char)byte;
          if (last_tile)        // padding bytes of the row's last word
            for (int e = cur.tn * 2 + wn + 1; e < nwords * 4; ++e) bp[e] = 0;
          // rowall starts at 1 (every key masked) and is cleared by any tile that finds an
          // unmasked valid key: plain stores of the same value 0 (no atomics: ~2000 tiles per
          // row would serialise on them)
          if (~byte & vm) st.rowall[grow] = 0;
        }
      }
      cur = nxt;
      continue;
    }
    float* __restrict__ C = p.C + (int64_t)cur.bz * p.sC;
    const float* __restrict__ Res = p.Res ? p.Res + (int64_t)cur.bz * p.sRes : nullptr;
    const bool full = m0 + BM <= p.M && n0 + BN <= p.N;
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) {
      const int col = n0 + wn * WN + ni * 32 + li;
      const int colc = min(col, p.N - 1);
      const float bv = p.bias ? p.bias[colc] : 0.f;
#pragma unroll
      for (int mi = 0; mi < TM; ++mi) {
        const int rbase = m0 + wm * WM + mi * 32;
        float rv[16];
        if (Res) {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            rv[r] = Res[(int64_t)min(rbase + mfma32_row(r, lh), p.M - 1) * p.ldres + colc];
        }
        float ov[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = acc[mi][ni][r] + bv;
          v = gemm_act(v, p.relu);
          if (Res) v += rv[r];
          if (p.relu_after) v = fmaxf(v, 0.f);
          ov[r] = v;
        }
        // interior tiles store without per-element predicates (hipcc puts a full
        // vmcnt(0) -- which on gfx9 also waits for the previous STORE -- in front of
        // every predicated store)
        float* cp = C + (int64_t)(rbase + 4 * lh) * p.ldc + col;
        if (full) {
#pragma unroll
          for (int r = 0; r < 16; ++r) cp[(int64_t)((r & 3) + 8 * (r >> 2)) * p.ldc] = ov[r];
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (rbase + mfma32_row(r, lh) < p.M && col < p.N)
              cp[(int64_t)((r & 3) + 8 * (r >> 2)) * p.ldc] = ov[r];
        }
      }
    }
    cur = nxt;
  }
}

// One problem (optionally batched): tiles enumerated batch-major, then m, n fastest.
struct SingleLocator {
  const GemmP& p;
  int mt, nt;
  __device__ __forceinline__ const GemmP& P(int) const { return p; }
  __device__ __forceinline__ TileRef operator()(int T) const {
    const int per_b = mt * nt;
    const int bz = T / per_b, r = T - bz * per_b;
    const int tm = r / nt;
    return TileRef{0, bz, tm, r - tm * nt};
  }
};

// ADD: the launch has a row-periodic addend on A (positional encodings), pn_gemm_desc.Aadd
template <int BM, int BN, int WM, int WN, int AMODE, bool ADD = false>
__global__ __launch_bounds__(64 * (BM / WM) * (BN / WN),
                             ((BM == 64 && WM == 64) ? PN_GEMM_W2_WGS / 2
                                                     : TileWgs<BM, BN, AMODE, ADD>::min_waves))
void k_gemm_tile(const GemmP p, const int batch) {
  __shared__ __attribute__((aligned(16))) float smem[TileSmem<BM, BN, AMODE>::FLOATS];
  const SingleLocator loc{p, (p.M + BM - 1) / BM, (p.N + BN - 1) / BN};
  gemm_persistent<BM, BN, WM, WN, AMODE, ADD>(loc, loc.mt * loc.nt * batch, smem);
}

// Attention-mask bits of one decoder layer in ONE launch (round 5): logits = me . MFs^T over
// the level's 4 N_l stencil rows -- the products and their order are k_gemm_tile's, so every
// logit is bit for bit the one pn_gemm_f32 + pn_mask_pack_stencil see -- blended, thresholded
// and packed in the epilogue: the Q x 4 N_l logit map (26.7 MB at the finest level) is neither
// written nor read back, and the pack launch is gone from the query chain.
__global__ __launch_bounds__(256) void k_fill_i32(int32_t* __restrict__ p, const int v, const int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = v;
}

struct StencilLocator {
  const GemmP& p;
  const StencilP& st;
  int mt, nt;
  __device__ __forceinline__ const GemmP& P(int) const { return p; }
  __device__ __forceinline__ TileRef operator()(int T) const {
    const int per_b = mt * nt;
    const int bz = T / per_b, r = T - bz * per_b;
    const int tm = r / nt;
    return TileRef{0, bz, tm, r - tm * nt};
  }
};

__global__ __launch_bounds__(256, (TileWgs<64, 64, A_ROW, false>::min_waves))
void k_gemm_stencil(const GemmP p, const StencilP st, const int batch) {
  __shared__ __attribute__((aligned(16))) float smem[TileSmem<64, 64, A_ROW>::FLOATS];
  const StencilLocator loc{p, st, (p.M + 63) / 64, (p.N + 63) / 64};
  gemm_persistent<64, 64, 32, 32, A_ROW, false, StencilLocator, 1>(loc, loc.mt * loc.nt * batch, smem);
}

// Several independent row-major GEMMs in ONE launch: the 64x64 tiles of all
// problems are enumerated together (n-tile fastest, so workgroups that share an A panel
// are neighbours), which fills the chip where a single M x 256 problem leaves a
// ragged second round of workgroups.
#define GEMM_GROUP_MAX 18   // (the 18 K / V projections of the 9 decoder layers in ONE launch; the
                            // descriptor block is 3.7 KB of the 4 KB kernel-argument space)
struct GroupP {
  int n;
  int tile_start[GEMM_GROUP_MAX + 1];
  int mt[GEMM_GROUP_MAX], nt[GEMM_GROUP_MAX];
  GemmP p[GEMM_GROUP_MAX];
};

static_assert(sizeof(GroupP) <= 4096, "k_gemm_group's argument block must fit the kernarg segment");

struct GroupLocator {
  const GroupP& g;
  __device__ __forceinline__ const GemmP& P(int i) const { return g.p[i]; }
  __device__ __forceinline__ TileRef operator()(int T) const {
    int i = 0;
#pragma unroll
    for (int j = 1; j < GEMM_GROUP_MAX; ++j)
      if (j < g.n && T >= g.tile_start[j]) i = j;
    const int local = T - g.tile_start[i];
    const int per_b = g.mt[i] * g.nt[i];
    const int bz = local / per_b, r = local - bz * per_b;
    const int tm = r / g.nt[i];
    return TileRef{i, bz, tm, r - tm * g.nt[i]};
  }
};

__global__ __launch_bounds__(256, (TileWgs<64, 64, A_ROW, true>::min_waves)) void k_gemm_group(const GroupP g) {
  __shared__ __attribute__((aligned(16))) float smem[TileSmem<64, 64, A_ROW>::FLOATS];
  const GroupLocator loc{g};
  gemm_persistent<64, 64, 32, 32, A_ROW, true>(loc, g.tile_start[GEMM_GROUP_MAX], smem);
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
  const bool has_add = p.Aadd && n0 >= p.aadd_from_col;          // wave-uniform
  const float* addrow = has_add ? p.Aadd + (int64_t)(am % p.aadd_rows) * p.ldaadd : arow;
  const float* wrow = W + (int64_t)wn_ * p.ldw;

  int ks = (p.K + NW - 1) / NW;
  ks = (ks + 31) & ~31;
  const int kbeg = min(wave * ks, p.K), kend = min(kbeg + ks, p.K);

  // All loads of a chunk are UNCONDITIONAL and issued together (k clamped into the row, the
  // value zeroed afterwards where k is past this wave's slice): a load under a per-lane
  // predicate is its own branch with a full vmcnt(0) behind it -- four serialised round trips
  // per chunk with a positional addend, measured 8 us per launch for a 1 us problem.
  auto load_chunk = [&](int kc, float4 (&a)[4], float4 (&b)[4]) {
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const int k0 = min(kc + 4 * lh, p.K - 4), k1 = min(kc + 8 + 4 * lh, p.K - 4);
    const int k2 = min(kc + 16 + 4 * lh, p.K - 4), k3 = min(kc + 24 + 4 * lh, p.K - 4);
    const bool ok0 = kc + 4 * lh < kend, ok1 = kc + 8 + 4 * lh < kend;
    const bool ok2 = kc + 16 + 4 * lh < kend, ok3 = kc + 24 + 4 * lh < kend;
    // (every element of a / b is written exactly once, from locals: the arrays stay in VGPRs)
    const float4 w0 = ld4(wrow + k0), w1 = ld4(wrow + k1), w2 = ld4(wrow + k2), w3 = ld4(wrow + k3);
    float4 x0, x1, x2, x3;
    if (AMODE == A_ROW) {
      x0 = ld4(arow + k0); x1 = ld4(arow + k1); x2 = ld4(arow + k2); x3 = ld4(arow + k3);
      if (has_add) {                          // ONE wave-uniform branch per chunk
        const float4 d0 = ld4(addrow + k0), d1 = ld4(addrow + k1);
        const float4 d2 = ld4(addrow + k2), d3 = ld4(addrow + k3);
        x0 = add4(x0, d0); x1 = add4(x1, d1); x2 = add4(x2, d2); x3 = add4(x3, d3);
      }
    } else {
      const int64_t l = p.lda;
      x0 = make_float4(arow[k0 * l], arow[(k0 + 1) * l], arow[(k0 + 2) * l], arow[(k0 + 3) * l]);
      x1 = make_float4(arow[k1 * l], arow[(k1 + 1) * l], arow[(k1 + 2) * l], arow[(k1 + 3) * l]);
      x2 = make_float4(arow[k2 * l], arow[(k2 + 1) * l], arow[(k2 + 2) * l], arow[(k2 + 3) * l]);
      x3 = make_float4(arow[k3 * l], arow[(k3 + 1) * l], arow[(k3 + 2) * l], arow[(k3 + 3) * l]);
    }
    a[0] = ok0 ? x0 : zero4; a[1] = ok1 ? x1 : zero4; a[2] = ok2 ? x2 : zero4; a[3] = ok3 ? x3 : zero4;
    b[0] = ok0 ? w0 : zero4; b[1] = ok1 ? w1 : zero4; b[2] = ok2 ? w2 : zero4; b[3] = ok3 ? w3 : zero4;
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
      v = gemm_act(v, p.relu);
      if (Res) v += Res[(int64_t)row * p.ldres + col];
      if (p.relu_after) v = fmaxf(v, 0.f);
      C[(int64_t)row * p.ldc + col] = v;
    }
  }
}

#ifndef PN_SKINNY_NW256
#define PN_SKINNY_NW256 8
#endif
template <int AMODE>
static int launch_skinny(const GemmP& p, int batch, hipStream_t s) {
  dim3 grid(pn_cdiv(p.N, 32), pn_cdiv(p.M, 32), batch);
  // (waves per 32x32 tile = K slices; PN_SKINNY_NW256: build-time choice for 128 < K <= 256, the
  // query chains' shape, so that variant libraries can be A/B'd)
  if (p.K <= 128)
    hipLaunchKernelGGL((k_gemm_skinny<AMODE, 4>), grid, dim3(256), 0, s, p);
  else if (p.K <= 256)
    hipLaunchKernelGGL((k_gemm_skinny<AMODE, PN_SKINNY_NW256>), grid, dim3(64 * PN_SKINNY_NW256), 0, s, p);
  else
    hipLaunchKernelGGL((k_gemm_skinny<AMODE, 16>), grid, dim3(1024), 0, s, p);
  return PN_LAUNCH_CHECK();
}


// Persistent launch: at most `wg_per_cu` workgroups per CU, a multiple of 8 so that every
// XCD gets the same number of them; `flags` may carry PN_GEMM_RESERVE(n): n workgroup slots
// (spread over the XCDs) stay unoccupied, so that the small latency-bound kernels of a
// concurrent stream (the query chains) find a resident slot without waiting for a kernel
// boundary.  The hint travels with the call: no process-wide state.
static int cu_count() {
  // immutable hardware attribute of the current device, looked up once per device
  static int cached[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (cached[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8)
      n = 256;
    cached[dev] = n;
  }
  return cached[dev];
}

static int persistent_grid(int64_t ntiles, int wg_per_cu, int flags) {
  const int reserve = ((flags >> PN_GEMM_RESERVE_SHIFT) & 0x3ff) * 8;
  int64_t cap = (int64_t)cu_count() * wg_per_cu / 8 * 8 - reserve;
  if (cap < 256) cap = 256;
  const int64_t want = (ntiles + 7) / 8 * 8;
  return (int)(want < cap ? want : cap);
}

// Resident workgroups per CU of a tile kernel (runtime occupancy query, cached per
// instantiation), capped at 4: the persistent grid must not exceed what is resident at
// once, or the surplus workgroups start a second round with their whole static share.
template <typename Kern>
static int resident_wgs(Kern kern, int cap, int threads = 256) {
  int n = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kern, threads, 0) != hipSuccess || n < 1) n = 1;
  return n < cap ? n : cap;
}

template <int BM, int BN, int WM, int WN, int AMODE>
static int launch_tile(const GemmP& p, int batch, hipStream_t s, int flags) {
  const int64_t ntiles = (int64_t)pn_cdiv(p.N, BN) * pn_cdiv(p.M, BM) * batch;
  constexpr int NT = 64 * (BM / WM) * (BN / WN);
  if (AMODE == A_ROW && p.Aadd) {
    auto kern = k_gemm_tile<BM, BN, WM, WN, AMODE, AMODE == A_ROW>;
    static const int wgs = resident_wgs(
        kern, (BM == 64 && WM == 64) ? PN_GEMM_W2_WGS : TileWgs<BM, BN, AMODE, (AMODE == A_ROW)>::value, NT);
    hipLaunchKernelGGL(kern, dim3(persistent_grid(ntiles, wgs, flags)), dim3(NT), 0, s, p, batch);
  } else {
    auto kern = k_gemm_tile<BM, BN, WM, WN, AMODE, false>;
    static const int wgs = resident_wgs(
        kern, (BM == 64 && WM == 64) ? PN_GEMM_W2_WGS : TileWgs<BM, BN, AMODE, false>::value, NT);
    hipLaunchKernelGGL(kern, dim3(persistent_grid(ntiles, wgs, flags)), dim3(NT), 0, s, p, batch);
  }
  return PN_LAUNCH_CHECK();
}

// ---- split-K for problems with too few output tiles to fill the chip -----------------
// part [batch][S][M][N] (dense) -> C = [relu_after](act(sum_s part + bias) + Res); the
// partials are summed in split order, so the result does not depend on scheduling.
__global__ __launch_bounds__(256) void k_splitk_reduce(const float* __restrict__ part,
                                                       const GemmP p, const int S) {
  const int n4 = p.N >> 2;
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)p.M * n4) return;
  const int b = blockIdx.y;
  const int row = (int)(e / n4), col = (int)(e - (int64_t)row * n4) * 4;
  const int64_t mn = (int64_t)p.M * p.N;
  const float* pp = part + (int64_t)b * S * mn + (int64_t)row * p.N + col;
  float4 v = ld4(pp);
  int s = 1;
  for (; s + 4 <= S; s += 4) {     // (slice order kept; four loads in flight)
    const float4 t0 = ld4(pp + s * mn), t1 = ld4(pp + (s + 1) * mn);
    const float4 t2 = ld4(pp + (s + 2) * mn), t3 = ld4(pp + (s + 3) * mn);
    v = add4(add4(add4(add4(v, t0), t1), t2), t3);
  }
  for (; s < S; ++s) v = add4(v, ld4(pp + s * mn));
  if (p.bias) v = add4(v, ld4(p.bias + col));
  v = make_float4(gemm_act(v.x, p.relu), gemm_act(v.y, p.relu), gemm_act(v.z, p.relu), gemm_act(v.w, p.relu));
  if (p.Res) {
    const float* r = p.Res + (int64_t)b * p.sRes + (int64_t)row * p.ldres + col;
    v = add4(v, make_float4(r[0], r[1], r[2], r[3]));
  }
  if (p.relu_after)
    v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
  float* c = p.C + (int64_t)b * p.sC + (int64_t)row * p.ldc + col;
  c[0] = v.x; c[1] = v.y; c[2] = v.z; c[3] = v.w;
}

// Number of K splits for a 64x64-tile launch; the partials must fit the caller's scratch.
// Round 4 rule, from a sweep over the 16 mid-size shapes of one image with 1216 workgroup slots
// (tools/ksplit_sweep.py: the round-3 rule -- ~1024 work items, >= 4 chunks per split -- cost
// 3412 us per image, the per-shape best 3351): aim for ~800 work items, never more than 6
// splits (the reduce pass and the partials' round trip grow with S: the 68-tile C5 convolution
// takes 22 us with 3-6 splits and 28 with 16), at least 8 chunks per split, 16 where the tile
// count alone half-fills the chip (522 tiles x 16 chunks runs best unsplit, 528 x 32 in two).
static int splitk_factor(const GemmP& p, int batch, const float* scratch, int64_t scratch_floats,
                         int* chunks_per_split, int flags) {
  const int nk = pn_cdiv(p.K, 32);
  const int64_t tiles = (int64_t)pn_cdiv(p.M, 64) * pn_cdiv(p.N, 64) * batch;
  *chunks_per_split = nk;
  const int forced = (flags >> PN_GEMM_KSPLIT_SHIFT) & 31;   // tuning: PN_GEMM_KSPLIT(n)
  if (!scratch || (p.N & 3) || !aligned16(p.bias) || ((uintptr_t)scratch & 15) || forced == 1)
    return 1;
#ifdef PN_SPLITK_ROUND3          // (the rule of rounds 1-3, for A/B builds)
  if (!forced && (tiles >= 512 || nk < 8)) return 1;
  int S = forced ? forced : (int)((1024 + tiles - 1) / tiles);
  if (S > 16) S = 16;
  if (!forced && S > nk / 4) S = nk / 4;
#else
  if (!forced && (tiles >= 1024 || nk < 8)) return 1;
  int S = forced ? forced : (int)((1600 + tiles) / (2 * tiles));      // round(800 / tiles)
  if (!forced) {
    if (S > 6) S = 6;
    const int min_chunks = tiles > 400 ? 16 : 8;
    if (S > nk / min_chunks) S = nk / min_chunks;
  }
  if (S > 16) S = 16;
#endif
  if (S > nk) S = nk;
  const int64_t per = (int64_t)batch * p.M * p.N;
  if ((int64_t)S * per > scratch_floats) S = (int)(scratch_floats / per);
  if (S < 2) return 1;
  const int cps = pn_cdiv(nk, S);
  *chunks_per_split = cps;
  return pn_cdiv(nk, cps);          // every split owns at least one chunk
}

template <int AMODE>
static int launch_tile64_splitk(const GemmP& p, int batch, float* scratch, int64_t scratch_floats,
                                hipStream_t s, int flags) {
  int cps;
  const int S = splitk_factor(p, batch, scratch, scratch_floats, &cps, flags);
#if PN_GEMM_W2
#define PN_T64_WM 64
#else
#define PN_T64_WM 32
#endif
  if (S <= 1) return launch_tile<64, 64, PN_T64_WM, 32, AMODE>(p, batch, s, flags);
  GemmP q = p;                       // pass 1: raw partial products into the scratch
  q.C = scratch; q.ldc = p.N; q.sC = (int64_t)p.M * p.N;
  q.bias = nullptr; q.Res = nullptr; q.relu = 0; q.relu_after = 0;
  q.ksplit = S; q.split_chunks = cps;
  if (int rc = launch_tile<64, 64, PN_T64_WM, 32, AMODE>(q, batch * S, s, flags)) return rc;
  const int64_t n = (int64_t)p.M * (p.N / 4);
  hipLaunchKernelGGL(k_splitk_reduce, dim3(pn_cdiv(n, 256), batch), dim3(256), 0, s, scratch, p, S);
  return PN_LAUNCH_CHECK();
}

static bool gemm_use_skinny(const pn_gemm_desc* d) {
  const int64_t tiles128 = (int64_t)pn_cdiv(d->M, 128) * pn_cdiv(d->N, 128) * d->batch;
  const int64_t tiles64 = (int64_t)pn_cdiv(d->M, 64) * pn_cdiv(d->N, 64) * d->batch;
  bool skinny = tiles128 < 96;
  // with a split-K scratch the 64x64 tile kernel takes over from ~64 tiles up
  if (d->splitk_scratch && tiles64 >= 64 && d->K >= 256 && !(d->N & 3)) skinny = false;
  if (d->flags & (PN_GEMM_FORCE_TILE | PN_GEMM_FORCE_TILE64 | PN_GEMM_FORCE_TILE128x64))
    skinny = false;
  if (d->flags & PN_GEMM_FORCE_SKINNY) skinny = true;
  // K < 32: the tile kernels' ragged-tail path loads "chunk 0" unconditionally, i.e. 32 columns
  // of every row -- with fewer than 32 it would read past the last row of W (found through the
  // backward pass's dW GEMMs, whose contraction runs over a few dozen pixels at small sizes);
  // the skinny kernel clamps every load into its row
  if (d->K < 32) skinny = true;
  return skinny;
}

extern "C" int pn_gemm_variant(const pn_gemm_desc* d) {
  if (!d || d->M <= 0 || d->N <= 0 || d->batch <= 0) return PN_BAD_ARG;
  const int col = (d->flags & PN_GEMM_A_COLMAJOR) ? 1 : 0;
  if (gemm_use_skinny(d)) return PN_GEMM_VARIANT_SKINNY + col;
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
  // the tile kernels address each per-batch operand with 32-bit element offsets
  const int64_t lim = (int64_t)1 << 29;
  if (!colmajor && (int64_t)(d->M - 1) * d->lda + d->K >= lim) return PN_BAD_ARG;
  if (colmajor && (int64_t)(d->K - 1) * d->lda + d->M >= lim) return PN_BAD_ARG;
  if ((int64_t)(d->N - 1) * d->ldw + d->K >= lim) return PN_BAD_ARG;
  if (d->Aadd && (int64_t)d->aadd_rows * d->ldaadd >= lim) return PN_BAD_ARG;
  GemmP p{};
  p.A = d->A; p.Aadd = d->Aadd; p.W = d->W; p.bias = d->bias; p.Res = d->Res; p.C = d->C;
  p.lda = d->lda; p.ldaadd = d->ldaadd; p.ldw = d->ldw; p.ldres = d->ldres; p.ldc = d->ldc;
  p.sA = d->strideA; p.sW = d->strideW; p.sRes = d->strideRes; p.sC = d->strideC;
  p.M = d->M; p.N = d->N; p.K = d->K; p.aadd_rows = d->Aadd ? d->aadd_rows : 1;
  p.aadd_from_col = d->Aadd ? d->aadd_from_col : 0;
  p.relu = (d->flags & PN_GEMM_GELU) ? 2 : (d->flags & PN_GEMM_RELU) ? 1 : 0;
  p.relu_after = (d->flags & PN_GEMM_RELU_AFTER_RES) ? 1 : 0;
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
  // Tile choice (measured on MI355X, tools/gemm_probe.hip / gemm_sweep.py): with
  // K = 256..1024 the persistent 64x64 tile (4 workgroups per CU then, 5 since round 4) is best or within 3 %
  // of the best on every encoder shape (99-112 TFLOP/s); 128x64 and 128x128 stay
  // selectable for sweeps.
  const bool big_m = PN_GEMM_BIGM_128x64 && !colmajor && !d->Aadd && d->M >= 16384 &&
                     d->N >= 256 && d->N % 64 == 0 && !(d->flags & PN_GEMM_FORCE_TILE64);
  if ((d->flags & PN_GEMM_FORCE_TILE128x64) || big_m)
    return colmajor ? launch_tile<128, 64, 64, 32, A_COL>(p, d->batch, s, d->flags)
                    : launch_tile<128, 64, 64, 32, A_ROW>(p, d->batch, s, d->flags);
  if (d->flags & PN_GEMM_FORCE_TILE)
    return colmajor ? launch_tile<128, 128, 64, 64, A_COL>(p, d->batch, s, d->flags)
                    : launch_tile<128, 128, 64, 64, A_ROW>(p, d->batch, s, d->flags);
  return colmajor ? launch_tile64_splitk<A_COL>(p, d->batch, d->splitk_scratch,
                                                d->splitk_scratch_floats, s, d->flags)
                  : launch_tile64_splitk<A_ROW>(p, d->batch, d->splitk_scratch,
                                                d->splitk_scratch_floats, s, d->flags);
}

// Grid the 64x64 tile kernel takes for `d` (introspection for tests / tuning: a function of
// the call's own descriptor only).
extern "C" int pn_gemm_grid_size(const pn_gemm_desc* d) {
  if (!d || d->M <= 0 || d->N <= 0 || d->batch <= 0) return PN_BAD_ARG;
  const int64_t tiles = (int64_t)pn_cdiv(d->M, 64) * pn_cdiv(d->N, 64) * d->batch;
  return persistent_grid(tiles, TileWgs<64, 64, A_ROW, false>::value, d->flags);
}

extern "C" int pn_gemm_wgs_per_cu(void) { return TileWgs<64, 64, A_ROW, false>::value; }

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
  hipLaunchKernelGGL(k_gemm_group, dim3(persistent_grid(tiles, TileWgs<64, 64, A_ROW, true>::value, d[0].flags)), dim3(256), 0,
                     (hipStream_t)stream, g);
  return PN_LAUNCH_CHECK();
}

extern "C" int pn_conv2d_nhwc_ex_f32(const float* in, const float* Wp, const float* bias,
                                     const float* res, float* out, int B, int H, int W,
                                     int Cin, int Cout, int KH, int KW, int stride, int pad,
                                     int flags, float* splitk_scratch,
                                     int64_t splitk_scratch_floats, void* stream) {
  if (!in || !Wp || !out || B <= 0 || H <= 0 || W <= 0 || stride <= 0 || pad < 0)
    return PN_BAD_ARG;
  if (Cin % 32 || !aligned16(in) || !aligned16(Wp)) return PN_BAD_ARG;
  if ((int64_t)H * W * Cin >= ((int64_t)1 << 29)) return PN_BAD_ARG;
  const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
  if (Ho <= 0 || Wo <= 0) return PN_BAD_ARG;
  GemmP p{};
  p.A = in; p.W = Wp; p.bias = bias; p.C = out; p.Res = res;
  p.M = Ho * Wo; p.N = Cout; p.K = KH * KW * Cin;
  p.lda = Cin; p.ldw = p.K; p.ldc = Cout; p.ldres = Cout;
  p.sA = (int64_t)H * W * Cin; p.sW = 0;
  p.sC = p.sRes = (int64_t)Ho * Wo * Cout;
  p.relu = (flags & PN_GEMM_RELU) ? 1 : 0;
  p.relu_after = (flags & PN_GEMM_RELU_AFTER_RES) ? 1 : 0;
  p.aadd_rows = 1;
  p.H = H; p.Wd = W; p.Cin = Cin; p.KW = KW; p.pad = pad; p.stride = stride; p.Wo = Wo;
  hipStream_t s = (hipStream_t)stream;
  // 64x64 tiles everywhere (3x3 FPN conv on MI355X: 625 us / 126 TFLOP/s, against 730
  // with 128x64 and 845 with 128x128 tiles, tools/gemm_probe.hip); the larger tiles stay
  // selectable for sweeps
  if (flags & PN_GEMM_FORCE_TILE128x64) return launch_tile<128, 64, 64, 32, A_CONV>(p, B, s, flags);
  if (flags & PN_GEMM_FORCE_TILE) return launch_tile<128, 128, 64, 64, A_CONV>(p, B, s, flags);
  return launch_tile64_splitk<A_CONV>(p, B, splitk_scratch, splitk_scratch_floats, s, flags);
}

extern "C" int pn_conv2d_nhwc_f32(const float* in, const float* Wp, const float* bias,
                                  float* out, int B, int H, int W, int Cin, int Cout,
                                  int KH, int KW, int pad, int relu, int flags,
                                  void* stream) {
  if (KH != 2 * pad + 1 || KW != 2 * pad + 1) return PN_BAD_ARG;   // "same" convolution
  return pn_conv2d_nhwc_ex_f32(in, Wp, bias, nullptr, out, B, H, W, Cin, Cout, KH, KW, 1, pad,
                               (flags & ~PN_GEMM_RELU) | (relu ? PN_GEMM_RELU : 0), nullptr, 0,
                               stream);
}

extern "C" int pn_mask_stencil_gemm_f32(const float* me, int64_t ld_me, int64_t stride_me,
                                        const float* rows, int64_t ld_rows, int64_t stride_rows,
                                        uint32_t* bits, int32_t* rowall, int B, int Q, int Nk,
                                        int K, int hi, int wi, int ho, int wo, int flags,
                                        void* stream) {
  if (!me || !rows || !bits || !rowall || B <= 0 || Q <= 0 || Nk <= 0 || K <= 0 || K % 32)
    return PN_BAD_ARG;
  if (hi <= 0 || wi <= 0 || ho <= 0 || wo <= 0 || (int64_t)ho * wo != Nk) return PN_BAD_ARG;
  if (ld_me % 4 || ld_rows % 4 || stride_me % 4 || stride_rows % 4 || !aligned16(me) || !aligned16(rows))
    return PN_BAD_ARG;
  const int64_t lim = (int64_t)1 << 29;
  if ((int64_t)(Q - 1) * ld_me + K >= lim || ((int64_t)4 * Nk - 1) * ld_rows + K >= lim)
    return PN_BAD_ARG;
  GemmP p{};
  p.A = me; p.W = rows; p.lda = ld_me; p.ldw = ld_rows; p.sA = stride_me; p.sW = stride_rows;
  p.M = Q; p.K = K; p.aadd_rows = 1;
  p.N = pn_cdiv(Nk, 16) * 64;            // 16 keys x 4 taps per 64-column tile
  const StencilP st{bits, rowall, Nk, hi, wi, ho, wo};
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_fill_i32, dim3(pn_cdiv(B * Q, 256)), dim3(256), 0, s, rowall, 1, B * Q);
  const int64_t ntiles = (int64_t)pn_cdiv(Q, 64) * (p.N / 64) * B;
  static const int wgs = resident_wgs(k_gemm_stencil, TileWgs<64, 64, A_ROW, false>::value, 256);
  hipLaunchKernelGGL(k_gemm_stencil, dim3(persistent_grid(ntiles, wgs, flags)), dim3(256), 0, s, p, st, B);
  return PN_LAUNCH_CHECK();
}

extern "C" int pn_abi_version(void) { return PN_ABI_VERSION; }
