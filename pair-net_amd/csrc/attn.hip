// Multi-head attention core for the Mask2Former / Relation Fusion decoders:
// 8 heads x 32, any number of queries, Nk keys from 100 to 16 700, optional boolean mask
// shared by the heads.  Both contractions run on the f32 MFMA; per 32-key tile and
// 32-query wave:
//   S^T[key][q] = K . Q^T       A = K tile, B = Q rows held in VGPRs
//   online softmax               the S^T accumulator layout puts query = lane&31 in
//                                every register, so max/sum are in-lane + one xor-32
//   O^T[d][q]  += V^T . P^T      P^T is consumed straight from the S^T registers as
//                                the MFMA B operand (k-permutation: MFMA t pairs the
//                                keys held in register t of the two half-waves); A = V
// Two kernels (round 3):
//   k_attn_small  Nk <= 512 (self-attention, the relation decoder's 200 pair features):
//                 workgroup = (32 queries, head, image), its NW = 4 /
//                 8 / 16 waves SPLIT THE KEYS (wave w takes tiles w, w + NW, ...), operands go
//                 global -> VGPR directly with the next tile prefetched under the current
//                 one's MFMAs (no LDS staging, no barrier in the loop), and the waves'
//                 partial (O, m, l) are merged through LDS in wave order: one launch, no
//                 combine pass, 1-3 tiles of latency instead of 4-33.
//   k_attn_chunk  flash-style over key chunks so the masked cross-attention over the
//                 4 200 / 16 700-token levels fills the chip: workgroup = (key chunk, head,
//                 image, 128-query group), 4 waves = 4 query groups sharing the K / V tiles
//                 through a DOUBLE-BUFFERED LDS stage: tile t+1 is written to the other stage
//                 and tile t+2's global loads (and mask words) are issued before tile t's
//                 MFMAs, one barrier per tile.  Partial (O, m, l) per chunk go to scratch;
//                 k_attn_combine (1024 threads per query: the chunk sweep is split four
//                 ways, fixed-order LDS reduction) merges them.
#include "common.h"

#define ATT_LD 36  // LDS row stride (floats): conflict-free b128 reads, see gemm.hip

__global__ __launch_bounds__(256) void k_mask_pack(const float* __restrict__ logits,
                                                   uint32_t* __restrict__ bits,
                                                   int32_t* __restrict__ rowall, int Nk) {
  const int64_t row = blockIdx.x;
  const int nwords = (Nk + 31) / 32;
  const float* lr = logits + row * Nk;
  uint32_t* br = bits + row * nwords;
  bool seen = false;
  // four 256-key strips per iteration: their loads are issued together (a 16 700-key row is
  // 17 iterations of independent loads instead of 66 dependent round trips)
  for (int base = 0; base < Nk; base += 1024) {
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = base + 256 * j + threadIdx.x;
      v[j] = i < Nk ? lr[i] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = base + 256 * j + threadIdx.x;
      const bool valid = i < Nk;
      const bool masked = valid && v[j] < 0.f;
      seen |= valid && !masked;
      const unsigned long long bal = __ballot(masked);
      const int lane = threadIdx.x & 63;
      const int w0 = (base + 256 * j + (threadIdx.x & ~63)) / 32;
      if (lane == 0 && w0 < nwords) br[w0] = (uint32_t)bal;
      if (lane == 32 && w0 + 1 < nwords) br[w0 + 1] = (uint32_t)(bal >> 32);
    }
  }
  const int some = __syncthreads_or(seen ? 1 : 0);
  if (threadIdx.x == 0) rowall[row] = some ? 0 : 1;
}

extern "C" int pn_mask_pack(const float* logits, uint32_t* bits, int32_t* rowall, int64_t R,
                            int Nk, void* stream) {
  if (!logits || !bits || !rowall || R <= 0 || Nk <= 0) return PN_BAD_ARG;
  hipLaunchKernelGGL(k_mask_pack, dim3((unsigned)R), dim3(256), 0, (hipStream_t)stream, logits,
                     bits, rowall, Nk);
  return PN_LAUNCH_CHECK();
}

// The same from the four full-resolution logits of each key's bilinear stencil (logits4
// [R][4][Nk], tap-major as pn_bilinear_stencil_rows_f32 orders the rows): the resized logit
// is blended exactly as pn_bilinear_planar_f32 blends it, thresholded, packed.
#define MPS_MAX_SIDE 1024   // output rows / columns whose blend weights fit the LDS tables
#define MPS_NT 1024         // threads: a 16 700-key row is 5 rounds of 16 loads per thread
__global__ __launch_bounds__(MPS_NT) void k_mask_pack_stencil(const float* __restrict__ logits4,
                                                           uint32_t* __restrict__ bits,
                                                           int32_t* __restrict__ rowall, int hi,
                                                           int wi, int ho, int wo) {
  // the blend weights depend on the output row / column only: make_tap once per row and per
  // column (the same function, hence the same weights, as pn_bilinear_planar_f32), not per key
  __shared__ float ly0[MPS_MAX_SIDE], ly1[MPS_MAX_SIDE], lx0[MPS_MAX_SIDE], lx1[MPS_MAX_SIDE];
  for (int t = threadIdx.x; t < ho; t += MPS_NT) {
    const Tap ty = make_tap(t, hi, ho);
    ly0[t] = ty.l0; ly1[t] = ty.l1;
  }
  for (int t = threadIdx.x; t < wo; t += MPS_NT) {
    const Tap tx = make_tap(t, wi, wo);
    lx0[t] = tx.l0; lx1[t] = tx.l1;
  }
  __syncthreads();
  const int Nk = ho * wo;
  const float inv_wo = 1.f / (float)wo;
  const int64_t row = blockIdx.x;
  const int nwords = (Nk + 31) / 32;
  const float* lr = logits4 + row * 4 * Nk;
  uint32_t* br = bits + row * nwords;
  bool seen = false;
  for (int base = 0; base < Nk; base += 4 * MPS_NT) {   // four strips, 16 loads in flight
    float v[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = min(base + MPS_NT * j + (int)threadIdx.x, Nk - 1);
#pragma unroll
      for (int t = 0; t < 4; ++t) v[j][t] = lr[(int64_t)t * Nk + i];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = base + MPS_NT * j + threadIdx.x;
      const bool valid = i < Nk;
      const int ic = valid ? i : Nk - 1;
      int oy = (int)((float)ic * inv_wo);              // estimate, then exact
      if (oy * wo > ic) --oy;
      else if ((oy + 1) * wo <= ic) ++oy;
      const int ox = ic - oy * wo;
      Tap ty, tx;
      ty.i0 = ty.i1 = tx.i0 = tx.i1 = 0;
      ty.l0 = ly0[oy]; ty.l1 = ly1[oy]; tx.l0 = lx0[ox]; tx.l1 = lx1[ox];
      const float r = tap_blend(ty, tx, v[j][0], v[j][1], v[j][2], v[j][3]);
      const bool masked = valid && r < 0.f;
      seen |= valid && !masked;
      const unsigned long long bal = __ballot(masked);
      const int lane = threadIdx.x & 63;
      const int w0 = (base + MPS_NT * j + (threadIdx.x & ~63)) / 32;
      if (lane == 0 && w0 < nwords) br[w0] = (uint32_t)bal;
      if (lane == 32 && w0 + 1 < nwords) br[w0 + 1] = (uint32_t)(bal >> 32);
    }
  }
  const int some = __syncthreads_or(seen ? 1 : 0);
  if (threadIdx.x == 0) rowall[row] = some ? 0 : 1;
}

extern "C" int pn_mask_pack_stencil(const float* logits4, uint32_t* bits, int32_t* rowall,
                                    int64_t R, int hi, int wi, int ho, int wo, void* stream) {
  if (!logits4 || !bits || !rowall || R <= 0 || hi <= 0 || wi <= 0 || ho <= 0 || wo <= 0 ||
      ho > MPS_MAX_SIDE || wo > MPS_MAX_SIDE || (int64_t)ho * wo >= ((int64_t)1 << 24))
    return PN_BAD_ARG;
  hipLaunchKernelGGL(k_mask_pack_stencil, dim3((unsigned)R), dim3(MPS_NT), 0, (hipStream_t)stream,
                     logits4, bits, rowall, hi, wi, ho, wo);
  return PN_LAUNCH_CHECK();
}

struct AttnP {
  const float* q; const float* k; const float* v;
  const uint32_t* bits; const int32_t* rowall;
  float* opart; float* ml; float* out;
  int64_t ldq, ldk, ldv, ldo;
  int Q, Nk, chunk, nchunks, nwords;
  float scale;
};

// ---- per-wave pieces shared by the two kernels ------------------------------------------
// Scores are kept in LOG2 units (the query fragment is pre-multiplied by scale * log2 e), so
// that every exponential of the online softmax is one v_exp_f32.
#define ATT_LOG2E 1.4426950408889634f
__device__ __forceinline__ float exp2_fast(float x) { return __builtin_amdgcn_exp2f(x); }

// Q fragment (B operand of S^T): MFMA t of k-group u uses d = 4u + t + 16*lh.
__device__ __forceinline__ void attn_load_q(const AttnP& p, int b, int qc, int head, int lh,
                                            float (&qf)[16]) {
  const float* qp = p.q + ((int64_t)b * p.Q + qc) * p.ldq + head * 32 + 16 * lh;
  const float sc = p.scale * ATT_LOG2E;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const float4 v = ld4(qp + 4 * u);
    qf[4 * u + 0] = v.x * sc; qf[4 * u + 1] = v.y * sc;
    qf[4 * u + 2] = v.z * sc; qf[4 * u + 3] = v.w * sc;
  }
}

// Mask word of a 32-key tile: bit j = key k0 + j is masked or lies past `kend`.
__device__ __forceinline__ uint32_t attn_dead_bits(uint32_t mw, int k0, int kend) {
  const int nvalid = kend - k0;
  return nvalid >= 32 ? mw : (mw | (0xffffffffu << nvalid));
}

// mask + online softmax on the S^T accumulator (register r <-> key k0 + mfma32_row(r, lh);
// `dead` = attn_dead_bits of the tile), rescaling o; afterwards s holds P^T.  Per element: one
// bit test + select, max, subtract, v_exp_f32, add.
__device__ __forceinline__ void attn_softmax(f32x16& s, f32x16& o, float& m_run, float& l_run,
                                             const uint32_t dead, const int lh) {
  const uint32_t dm = dead >> (4 * lh);     // bit (r & 3) + 8 * (r >> 2) <-> register r
  float tmax = -INFINITY;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    s[r] = (dm & (1u << ((r & 3) + 8 * (r >> 2)))) ? -INFINITY : s[r];
    tmax = fmaxf(tmax, s[r]);
  }
  tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
  const float m_new = fmaxf(m_run, tmax);
  // (nothing alive so far: subtract 0, every exponential below is exp2(-inf) = 0)
  const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
  const float alpha = exp2_fast(m_run - m_use);
  float psum = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float pv = exp2_fast(s[r] - m_use);
    s[r] = pv;
    psum += pv;
  }
  l_run = l_run * alpha + psum;
  m_run = m_new;
#pragma unroll
  for (int r = 0; r < 16; ++r) o[r] *= alpha;
}

// =========================================================================================
// k_attn_small: keys split over the NW waves of a (32 queries, head, image) workgroup
// =========================================================================================
template <int NW, bool PF = true>
__global__ __launch_bounds__(64 * NW) void k_attn_small(const AttnP p) {
  __shared__ float Os[NW][32][32];
  __shared__ float Ms[NW][32], Ls[NW][32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int head = blockIdx.y, b = blockIdx.z;
  const int q0 = blockIdx.x * 32;
  const int myq = q0 + li;
  const int qc = min(myq, p.Q - 1);
  float qf[16];
  attn_load_q(p, b, qc, head, lh, qf);
  const bool use_mask = p.bits != nullptr;
  bool row_unmask = true;
  const uint32_t* brow = nullptr;
  if (use_mask) {
    row_unmask = p.rowall[(int64_t)b * p.Q + qc] != 0;
    brow = p.bits + ((int64_t)b * p.Q + qc) * p.nwords;
  }
  // buffer descriptors on this (image, head)'s K / V rows + 32-bit per-lane byte offsets: no
  // 64-bit address registers per load (pn_attention_f32 checks Nk * ld < 2^29)
  const __amdgpu_buffer_rsrc_t rk = make_rsrc(p.k + (int64_t)b * p.Nk * p.ldk + head * 32);
  const __amdgpu_buffer_rsrc_t rv = make_rsrc(p.v + (int64_t)b * p.Nk * p.ldv + head * 32);
  const unsigned ldk4 = (unsigned)p.ldk * 4u, ldv4 = (unsigned)p.ldv * 4u;
  const int ntiles = (p.Nk + 31) >> 5;

  // operands of one tile, straight from global memory (rows past Nk: the last row, whose
  // scores are masked out below): K fragment = A of S^T, V fragment = A of O^T
  auto load_tile = [&](int t, auto& kf, auto& vf, uint32_t& mw) {
    const int k0 = t * 32;
    const unsigned ko = (unsigned)min(k0 + li, p.Nk - 1) * ldk4 + 64u * lh;
#pragma unroll
    for (int u = 0; u < 4; ++u) kf[u] = buf_ld4(rk, ko, 16 * u);
#pragma unroll
    for (int r = 0; r < 16; ++r)
      vf[r] = buf_ld1(rv, (unsigned)min(k0 + mfma32_row(r, lh), p.Nk - 1) * ldv4 + 4u * li, 0);
    mw = row_unmask ? 0u : brow[t];
  };

  f32x16 o;
#pragma unroll
  for (int r = 0; r < 16; ++r) o[r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  auto compute_tile = [&](int t, const auto& kf, const auto& vf, uint32_t mw) {
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float4 a = kf[u];
      s = mfma32(a.x, qf[4 * u + 0], s);
      s = mfma32(a.y, qf[4 * u + 1], s);
      s = mfma32(a.z, qf[4 * u + 2], s);
      s = mfma32(a.w, qf[4 * u + 3], s);
    }
    attn_softmax(s, o, m_run, l_run, attn_dead_bits(mw, t * 32, p.Nk), lh);
#pragma unroll
    for (int r = 0; r < 16; ++r) o = mfma32(vf[r], s[r], o);
  };
  // two statically indexed operand sets: while one tile is multiplied the next one's loads
  // are in flight (NW = 16 runs at the 128-register cap of a 1024-thread workgroup: there
  // the four waves per SIMD hide the latency instead)
  float4 kfa[4], kfb[PF ? 4 : 1];
  float vfa[16], vfb[PF ? 16 : 1];
  uint32_t mwa = 0u, mwb = 0u;
  if (PF) {
    int t = wave;
    if (t < ntiles) load_tile(t, kfa, vfa, mwa);
    for (; t < ntiles; t += 2 * NW) {
      const bool more = t + NW < ntiles;
      if (more) load_tile(t + NW, kfb, vfb, mwb);
      compute_tile(t, kfa, vfa, mwa);
      if (more) {
        if (t + 2 * NW < ntiles) load_tile(t + 2 * NW, kfa, vfa, mwa);
        compute_tile(t + NW, kfb, vfb, mwb);
      }
    }
  } else {
    for (int t = wave; t < ntiles; t += NW) {
      load_tile(t, kfa, vfa, mwa);
      compute_tile(t, kfa, vfa, mwa);
    }
  }
  // ---- merge the waves' partials (fixed wave order) ----
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
#pragma unroll
  for (int r = 0; r < 16; ++r) Os[wave][mfma32_row(r, lh)][li] = o[r];
  if (lh == 0) { Ms[wave][li] = m_run; Ls[wave][li] = l_tot; }
  __syncthreads();
  // thread (query q, 4 consecutive d); with NW > 4 the surplus waves are done
  if (tid >= 256) return;
  const int q = tid & 31, d0 = (tid >> 5) * 4;
  if (q0 + q >= p.Q) return;
  float M = -INFINITY;
#pragma unroll
  for (int w = 0; w < NW; ++w) M = fmaxf(M, Ms[w][q]);
  float den = 0.f, n0 = 0.f, n1 = 0.f, n2 = 0.f, n3 = 0.f;
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    const float mwv = Ms[w][q];
    const float wt = exp2_fast(mwv - M);   // (a wave without live keys: exp2(-inf) = 0)
    den += wt * Ls[w][q];
    n0 += wt * Os[w][d0 + 0][q];
    n1 += wt * Os[w][d0 + 1][q];
    n2 += wt * Os[w][d0 + 2][q];
    n3 += wt * Os[w][d0 + 3][q];
  }
  const float inv = 1.f / den;
  st4(p.out + ((int64_t)b * p.Q + q0 + q) * p.ldo + head * 32 + d0,
      make_float4(n0 * inv, n1 * inv, n2 * inv, n3 * inv));
}

// =========================================================================================
// k_attn_chunk: key chunks across workgroups, K / V tiles shared by 4 query waves via LDS
// =========================================================================================
__global__ __launch_bounds__(256) void k_attn_chunk(const AttnP p) {
  __shared__ __attribute__((aligned(16))) float smem[2][2 * 32 * ATT_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int c = blockIdx.x, head = blockIdx.y;
  const int qgroups = (p.Q + 127) / 128;
  const int b = blockIdx.z / qgroups, qg = blockIdx.z % qgroups;
  const int q0 = qg * 128 + wave * 32;
  const int myq = q0 + li;                 // this lane's query column
  const bool q_ok = myq < p.Q;
  const int qc = q_ok ? myq : p.Q - 1;

  float qf[16];
  attn_load_q(p, b, qc, head, lh, qf);
  const bool use_mask = p.bits != nullptr;
  bool row_unmask = true;
  const uint32_t* brow = nullptr;
  if (use_mask) {
    row_unmask = p.rowall[(int64_t)b * p.Q + qc] != 0;
    brow = p.bits + ((int64_t)b * p.Q + qc) * p.nwords;
  }

  f32x16 o;
#pragma unroll
  for (int r = 0; r < 16; ++r) o[r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  const int kbeg = c * p.chunk;
  const int kend = min(kbeg + p.chunk, p.Nk);
  const int ntiles = (kend - kbeg + 31) >> 5;
  const int lrow = tid >> 3, lcol = (tid & 7) * 4;  // staging: one float4 of K and V per thread
  const float* kb = p.k + (int64_t)b * p.Nk * p.ldk + head * 32 + lcol;
  const float* vb = p.v + (int64_t)b * p.Nk * p.ldv + head * 32 + lcol;
  // (rows past the chunk's end read its last key; their scores are masked out)
  auto gload = [&](int t, float4& kv, float4& vv) {
    const int key = min(kbeg + 32 * t + lrow, kend - 1);
    kv = ld4(kb + (int64_t)key * p.ldk);
    vv = ld4(vb + (int64_t)key * p.ldv);
  };
  auto sstore = [&](int stage, const float4& kv, const float4& vv) {
    st4(&smem[stage][lrow * ATT_LD + lcol], kv);
    st4(&smem[stage][32 * ATT_LD + lrow * ATT_LD + lcol], vv);
  };
  auto mask_word = [&](int t) { return row_unmask ? 0u : brow[(kbeg >> 5) + t]; };

  float4 kreg, vreg;
  gload(0, kreg, vreg);
  uint32_t mw = mask_word(0), mw_next = 0u;
  sstore(0, kreg, vreg);
  if (ntiles > 1) { gload(1, kreg, vreg); mw_next = mask_word(1); }
  __syncthreads();
  for (int t = 0; t < ntiles; ++t) {
    // stage t&1 holds tile t; the registers hold tile t+1 (loaded one tile ago); tile t+2's
    // loads are issued before this tile's MFMAs
    uint32_t mw_next2 = 0u;
    if (t + 1 < ntiles) {
      sstore((t + 1) & 1, kreg, vreg);
      if (t + 2 < ntiles) { gload(t + 2, kreg, vreg); mw_next2 = mask_word(t + 2); }
    }
    const float* Ks = smem[t & 1];
    const float* Vs = Ks + 32 * ATT_LD;
    const int k0 = kbeg + 32 * t;
    const uint32_t dead = attn_dead_bits(mw, k0, kend);
    // A tile none of this wave's 32 queries may attend to contributes exactly nothing
    // (every p = 0, running max / sum unchanged): skip its 32 MFMAs.  Trained checkpoints
    // attend inside the predicted mask only, so most tiles of the large levels go this way.
    if (!__all(dead == 0xffffffffu)) {
    // ---- S^T = K Q^T ----
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float4 a = ld4(Ks + li * ATT_LD + 16 * lh + 4 * u);
      s = mfma32(a.x, qf[4 * u + 0], s);
      s = mfma32(a.y, qf[4 * u + 1], s);
      s = mfma32(a.z, qf[4 * u + 2], s);
      s = mfma32(a.w, qf[4 * u + 3], s);
    }
    attn_softmax(s, o, m_run, l_run, dead, lh);
    // ---- O^T += V^T P^T : MFMA t pairs keys mfma32_row(t, 0) / mfma32_row(t, 1) ----
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float a = Vs[mfma32_row(r, lh) * ATT_LD + li];
      o = mfma32(a, s[r], o);
    }
    }
    mw = mw_next;
    mw_next = mw_next2;
    __syncthreads();   // tile t fully consumed; stage (t+1)&1 visible
  }
  // ---- O^T register r is d = mfma32_row(r, lh) of query lane&31 ----
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  if (!q_ok) return;
  if (p.nchunks == 1) {  // single chunk: normalise and write the final rows directly
    const float inv = 1.f / l_tot;
    float* op = p.out + ((int64_t)b * p.Q + myq) * p.ldo + head * 32;
#pragma unroll
    for (int g = 0; g < 4; ++g)
      st4(op + 8 * g + 4 * lh, make_float4(o[4 * g + 0] * inv, o[4 * g + 1] * inv,
                                           o[4 * g + 2] * inv, o[4 * g + 3] * inv));
    return;
  }
  const int64_t slot = (((int64_t)b * 8 + head) * p.nchunks + c) * p.Q + myq;
  float* op = p.opart + slot * 32;
#pragma unroll
  for (int g = 0; g < 4; ++g)
    st4(op + 8 * g + 4 * lh, make_float4(o[4 * g + 0], o[4 * g + 1], o[4 * g + 2], o[4 * g + 3]));
  if (lh == 0) {
    p.ml[slot * 2 + 0] = m_run;
    p.ml[slot * 2 + 1] = l_tot;
  }
}

// out[b][q][h*32+d] = sum_c e^{m_c-M} O_c[d] / sum_c e^{m_c-M} l_c.
// One workgroup of 1024 threads per (b, q): thread = (chunk phase cp of 4, head, d).  The
// 128 lanes of a head first sweep the chunk statistics (lane = chunk) to get M, the
// weights (kept in LDS) and the denominator; then thread (cp, head, d) accumulates the
// weighted partials of the chunks c = cp (mod 4) with independent loads, and the four
// phases are summed through LDS in phase order (deterministic).
#define ATT_MAXCH 256
__global__ __launch_bounds__(1024) void k_attn_combine(const float* __restrict__ opart,
                                                       const float* __restrict__ ml,
                                                       float* __restrict__ out, int64_t ldo, int Q,
                                                       int nchunks) {
  __shared__ float wts[8][ATT_MAXCH];
  __shared__ float red[4][8][2];        // per (phase, head): partial max / denominator
  __shared__ float acc[4][8][32];
  const int q = blockIdx.x, b = blockIdx.y;
  const int tid = threadIdx.x, cp = tid >> 8, head = (tid >> 5) & 7, d = tid & 31;
  const int64_t base = ((int64_t)b * 8 + head) * nchunks;
  const int lc = cp * 32 + d;           // this lane's chunk slot among the head's 128 lanes
  float M = -INFINITY;
  for (int c = lc; c < nchunks; c += 128) M = fmaxf(M, ml[((base + c) * Q + q) * 2]);
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) M = fmaxf(M, __shfl_xor(M, o, 32));
  if (d == 0) red[cp][head][0] = M;
  __syncthreads();
  M = fmaxf(fmaxf(red[0][head][0], red[1][head][0]), fmaxf(red[2][head][0], red[3][head][0]));
  float den = 0.f;
  for (int c = lc; c < nchunks; c += 128) {
    const int64_t slot = (base + c) * Q + q;
    const float m = ml[slot * 2];
    const float w = exp2_fast(m - M);     // (log2 units; a chunk without live keys: 0)
    wts[head][c] = w;
    den += w * ml[slot * 2 + 1];
  }
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) den += __shfl_xor(den, o, 32);
  if (d == 0) red[cp][head][1] = den;
  __syncthreads();
  float n0 = 0.f, n1 = 0.f, n2 = 0.f, n3 = 0.f;
  int c = cp;
  for (; c + 12 < nchunks; c += 16) {
    n0 += wts[head][c + 0] * opart[((base + c + 0) * Q + q) * 32 + d];
    n1 += wts[head][c + 4] * opart[((base + c + 4) * Q + q) * 32 + d];
    n2 += wts[head][c + 8] * opart[((base + c + 8) * Q + q) * 32 + d];
    n3 += wts[head][c + 12] * opart[((base + c + 12) * Q + q) * 32 + d];
  }
  for (; c < nchunks; c += 4) n0 += wts[head][c] * opart[((base + c) * Q + q) * 32 + d];
  acc[cp][head][d] = (n0 + n1) + (n2 + n3);
  __syncthreads();
  if (cp != 0) return;
  den = ((red[0][head][1] + red[1][head][1]) + red[2][head][1]) + red[3][head][1];
  const float num = ((acc[0][head][d] + acc[1][head][d]) + acc[2][head][d]) + acc[3][head][d];
  out[((int64_t)b * Q + q) * ldo + head * 32 + d] = num / den;
}

#ifndef ATT_SMALL_MAX
#define ATT_SMALL_MAX 512    // keys up to which one workgroup's waves split the key range
#endif
#define ATT_MIN_CHUNK 64     // keys per chunk at least (two tiles)

// Workgroups a long-key launch aims for (build-time, so that variant libraries can be A/B'd).
// Round 4: 512 instead of 1024 -- 53 instead of 105 chunks of the 16 700-key level halve the
// partial (O, m, l) round trip through k_attn_combine while each workgroup's five-tile life
// becomes ten tiles; bench `roofline_attention` over the 30 calls of an image, two runs each
// on one box: 1024: 12.86 / 12.85 us, 768: 12.66 / 12.62, 512: 12.29 / 12.22 (second box:
// 512: 12.37 / 12.26, 384: 12.62 / 12.63, 256: 12.52 / 12.55).
#ifndef ATT_WANT_WGS
#define ATT_WANT_WGS 512
#endif
static int attn_chunking(int Nk, int B, int Q, int* chunk) {
  // long key sets: ~ATT_WANT_WGS workgroups, chunks of at least 2 tiles (64 keys) and a multiple
  // of 32 keys, at most ATT_MAXCH of them
  const int qgroups = (Q + 127) / 128;
  int want = ATT_WANT_WGS / (8 * B * qgroups);
  if (want < 1) want = 1;
  if (want > ATT_MAXCH) want = ATT_MAXCH;
  int ch = ((Nk + want - 1) / want + 31) & ~31;
  if (ch < ATT_MIN_CHUNK) ch = ATT_MIN_CHUNK;
  *chunk = ch;
  return (Nk + ch - 1) / ch;
}

extern "C" int64_t pn_attn_scratch_floats(int B, int Q, int Nk) {
  if (Nk <= ATT_SMALL_MAX) return 64;   // (k_attn_small needs none; never a zero-size buffer)
  int chunk;
  const int nch = attn_chunking(Nk, B, Q, &chunk);
  return (int64_t)B * 8 * nch * Q * 34;
}

extern "C" int pn_attention_f32(const float* q, int64_t ldq, const float* k, int64_t ldk,
                                const float* v, int64_t ldv, const uint32_t* maskbits,
                                const int32_t* rowall, float* out, int64_t ldo, float* scratch,
                                int B, int Q, int Nk, float scale, void* stream) {
  if (!q || !k || !v || !out || !scratch || B <= 0 || Q <= 0 || Nk <= 0) return PN_BAD_ARG;
  if ((ldq | ldk | ldv | ldo) & 3) return PN_BAD_ARG;
  if ((int64_t)Nk * ldk >= ((int64_t)1 << 29) || (int64_t)Nk * ldv >= ((int64_t)1 << 29))
    return PN_BAD_ARG;
  if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)scratch | (uintptr_t)out) & 15)
    return PN_BAD_ARG;
  if ((maskbits == nullptr) != (rowall == nullptr)) return PN_BAD_ARG;
  AttnP p{};
  p.q = q; p.k = k; p.v = v; p.bits = maskbits; p.rowall = rowall;
  p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.Q = Q; p.Nk = Nk; p.scale = scale;
  p.nwords = (Nk + 31) / 32;
  p.out = out;
  p.ldo = ldo;
  hipStream_t s = (hipStream_t)stream;
  if (Nk <= ATT_SMALL_MAX) {
    // waves per workgroup: at most ~3 tiles per wave
    const int ntiles = (Nk + 31) / 32;
    const dim3 grid((Q + 31) / 32, 8, B);
    p.nchunks = 1;
    if (ntiles <= 4)
      hipLaunchKernelGGL(k_attn_small<4>, grid, dim3(256), 0, s, p);
    else if (ntiles <= 16)
      hipLaunchKernelGGL(k_attn_small<8>, grid, dim3(512), 0, s, p);
    else   // (only with a larger ATT_SMALL_MAX: 16 waves at the 128-register cap, no prefetch)
      hipLaunchKernelGGL((k_attn_small<16, false>), grid, dim3(1024), 0, s, p);
    return PN_LAUNCH_CHECK();
  }
  p.nchunks = attn_chunking(Nk, B, Q, &p.chunk);
  p.opart = scratch;
  p.ml = scratch + (int64_t)B * 8 * p.nchunks * Q * 32;
  const int qgroups = (Q + 127) / 128;
  hipLaunchKernelGGL(k_attn_chunk, dim3(p.nchunks, 8, B * qgroups), dim3(256), 0, s, p);
  if (p.nchunks > 1)
    hipLaunchKernelGGL(k_attn_combine, dim3(Q, B), dim3(1024), 0, s, p.opart, p.ml, out, ldo, Q,
                       p.nchunks);
  return PN_LAUNCH_CHECK();
}
