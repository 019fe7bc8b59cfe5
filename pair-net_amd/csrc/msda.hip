// Multi-scale deformable attention sampling for the pixel-decoder encoder
// (8 heads x 32 channels, P = 4 points, L <= 4 levels): an L2-resident gather.
//
// Two phases per workgroup (k_msda below).  The gather threads are (head, sampling point,
// 4 channels): 32 lanes per head = 4 points x 8 lanes; the 8 lanes of a point read one
// 128-byte value row per bilinear tap as float4s (every tap is a full line), L x 4 = 12
// independent 16-byte loads in flight per lane, all unconditional (clamped rows, weight 0
// outside the map).  The sum over points is an xor-8 / xor-16 shuffle.
// Reference point + offset / (W_l, H_l) and the grid_sample un-normalisation follow mmcv's
// CPU formula (multi_scale_deformable_attn_pytorch):
//   loc = ref + off / (W_l, H_l);  g = 2 loc - 1;  ix = ((g + 1) W_l - 1) / 2
// with zero padding outside the map.
#include "common.h"
#include "s3_common.h"

struct MsdaLevels {
  int h[4], w[4], start[4];
  int L, N;
};

// 16-lane groups (one head's L x 4 logits, padded to 16)
__device__ __forceinline__ float grp16_max(float v) {
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float pt_sum(float v) {   // over the 4 points: lanes ^8, ^16
  v += __shfl_xor(v, 8, 64);
  return v + __shfl_xor(v, 16, 64);
}

struct __attribute__((aligned(16))) MsdaTap {
  unsigned off[4];   // byte offsets of the 4 taps' value rows (clamped into the map)
  float w[4];        // bilinear weight x attention weight (0 for taps outside the map)
};

#define MSDA_TQ 2   // queries per workgroup

// Workgroup = MSDA_TQ query tokens.  Phase 1: thread (query, head, point, level) -- 128
// threads per query, a head's 16 (point, level) slots in one 16-lane group -- computes the
// softmax weight, the sampling location and the four tap offsets / weights ONCE and parks
// them in LDS (the gather threads used to recompute them eight times over).  Phase 2:
// thread (head, point, 4 channels) reads them back (broadcast within its 8 lanes), issues
// the L x 4 float4 gathers of a query back to back and accumulates.
// -DMSDA_NT=1 (round 4, measured, NOT adopted): the offsets / logits are loaded and the output
// is stored with the non-temporal hint, so that those 48 MB of stream do not push a band's
// value rows out of its XCD's L2.  FETCH_SIZE per launch falls by 8 % (tools/ab_msda_nt.sh:
// 117.4 -> 108.0 MB over the probe's mix of init and N(0, 8 px) offsets), the kernel does not
// get faster (init offsets 46.9 -> 49.8 us, N(0, 8 px) 63.0 -> 61.9 us) and the pipelined bench
// does not move (208.5 vs 208.4 images/s, three alternating runs): it is bound by the L1
// request rate, not by the L2 misses.
#ifndef MSDA_NT
#define MSDA_NT 0
#endif
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int L, bool S3OUT = false>
__device__ __forceinline__ void msda_one_shot(const float* __restrict__ value,
                                              const float* __restrict__ offaw,
                                              float* __restrict__ out,
                                              const MsdaLevels& lv, const int64_t ldv,
                                              const int64_t ldo) {
  __shared__ MsdaTap taps[MSDA_TQ][8][4][4];   // [query][head][point][level]
  __shared__ float attw[MSDA_TQ][8][4][4];     // softmax weight of the same slot
  __shared__ int tok[MSDA_TQ];
  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  constexpr int LP = L * 4;

  // XCD-aware token order: the dispatcher puts workgroup g on XCD g % 8; give each XCD
  // one horizontal band of the image at EVERY level (rows [k h_l/8, (k+1) h_l/8)), so
  // the value rows its tokens sample (all levels, around the same normalised position)
  // stay inside that XCD's 4 MB L2 instead of streaming the whole 22 MB map through
  // all eight L2s.  Tokens past a band's end are skipped.
  const int band = blockIdx.x & 7;
  // ---- phase 1 ----
  {
    const int qi = tid >> 7, head = (tid >> 4) & 7, pt = (tid >> 2) & 3, l = tid & 3;
    int i = (blockIdx.x >> 3) * MSDA_TQ + qi;            // index inside the band
    int n = -1, qw = 1, qh = 1, qs = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (k < L && n < 0) {
        const int r0 = (band * lv.h[k]) >> 3, r1 = ((band + 1) * lv.h[k]) >> 3;
        const int cnt = (r1 - r0) * lv.w[k];
        if (i < cnt) {
          n = lv.start[k] + r0 * lv.w[k] + i;
          qw = lv.w[k]; qh = lv.h[k]; qs = lv.start[k];
        } else {
          i -= cnt;
        }
      }
    }
    if ((tid & 127) == 0) tok[qi] = n;
    const bool live = n >= 0 && l < L;
    const int nc = max(n, 0), lc = min(l, L - 1);
    const float* oa = offaw + ((int64_t)b * lv.N + nc) * ldo;
#if MSDA_NT
    // offsets / logits are read once and the output is written once: non-temporal accesses,
    // so that 48 MB of stream do not push the band's value rows out of the XCD's 4 MB L2
    const float e = __builtin_nontemporal_load(oa + 8 * LP * 2 + head * LP + lc * 4 + pt);
    const f32x2 off_v = __builtin_nontemporal_load(
        reinterpret_cast<const f32x2*>(oa + head * LP * 2 + lc * 8 + pt * 2));
    const float2 off = make_float2(off_v[0], off_v[1]);
#else
    const float e = oa[8 * LP * 2 + head * LP + lc * 4 + pt];
    const float2 off = *reinterpret_cast<const float2*>(oa + head * LP * 2 + lc * 8 + pt * 2);
#endif
    // softmax over the head's L x 4 logits: levels summed in order per point, then the
    // points pairwise (the summation order of the reference-checked first version)
    const float mx = grp16_max(l < L ? e : -INFINITY);
    const float ex = l < L ? expf(e - mx) : 0.f;
    const int g0 = (tid & 63) & ~3;
    float den = __shfl(ex, g0, 64);
#pragma unroll
    for (int k = 1; k < L; ++k) den += __shfl(ex, g0 + k, 64);
    den += __shfl_xor(den, 4, 64);
    den += __shfl_xor(den, 8, 64);
    const float aw = ex / den;
    if (live) {
      const int idx = n - qs;
      const int qy = idx / qw, qx = idx - qy * qw;
      const float ref_x = ((float)qx + 0.5f) / (float)qw;
      const float ref_y = ((float)qy + 0.5f) / (float)qh;
      const int Hl = lv.h[lc], Wl = lv.w[lc];
      const float locx = ref_x + off.x / (float)Wl;
      const float locy = ref_y + off.y / (float)Hl;
      const float gx = 2.f * locx - 1.f, gy = 2.f * locy - 1.f;
      const float ix = ((gx + 1.f) * (float)Wl - 1.f) * 0.5f;
      const float iy = ((gy + 1.f) * (float)Hl - 1.f) * 0.5f;
      const float fx = floorf(ix), fy = floorf(iy);
      // (clamp before the int conversion: far-out offsets must not overflow it)
      const int x0 = (int)fminf(fmaxf(fx, -2.f), (float)Wl), y0 = (int)fminf(fmaxf(fy, -2.f), (float)Hl);
      const float tx = ix - fx, ty = iy - fy;
      const bool xin0 = x0 >= 0 && x0 < Wl, xin1 = x0 + 1 >= 0 && x0 + 1 < Wl;
      const bool yin0 = y0 >= 0 && y0 < Hl, yin1 = y0 + 1 >= 0 && y0 + 1 < Hl;
      const int xa = min(max(x0, 0), Wl - 1), xb = min(max(x0 + 1, 0), Wl - 1);
      const int ya = min(max(y0, 0), Hl - 1), yb = min(max(y0 + 1, 0), Hl - 1);
      MsdaTap t;
      // out-of-map taps read a clamped row with weight 0: every gather is unconditional
      t.w[0] = (xin0 && yin0) ? (1.f - tx) * (1.f - ty) : 0.f;
      t.w[1] = (xin1 && yin0) ? tx * (1.f - ty) : 0.f;
      t.w[2] = (xin0 && yin1) ? (1.f - tx) * ty : 0.f;
      t.w[3] = (xin1 && yin1) ? tx * ty : 0.f;
      const unsigned row = (unsigned)ldv * 4u, base = (unsigned)lv.start[lc];
      t.off[0] = (base + (unsigned)(ya * Wl + xa)) * row;
      t.off[1] = (base + (unsigned)(ya * Wl + xb)) * row;
      t.off[2] = (base + (unsigned)(yb * Wl + xa)) * row;
      t.off[3] = (base + (unsigned)(yb * Wl + xb)) * row;
      taps[qi][head][pt][lc] = t;
      attw[qi][head][pt][lc] = aw;   // multiplies the bilinear sum afterwards (rounding order)
    }
    __syncthreads();
    // ---- phase 2 ----
    const int c4 = tid & 7, p2 = (tid >> 3) & 3, h2 = tid >> 5;
    const char* vb = reinterpret_cast<const char*>(value + (int64_t)b * lv.N * ldv + h2 * 32 + c4 * 4);
#pragma unroll
    for (int q = 0; q < MSDA_TQ; ++q) {
      const int nq = tok[q];
      if (nq < 0) continue;                   // (uniform)
      float4 v[L][4];
      MsdaTap t[L];
#pragma unroll
      for (int k = 0; k < L; ++k) {
        t[k] = taps[q][h2][p2][k];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[k][j] = *reinterpret_cast<const float4*>(vb + t[k].off[j]);
      }
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int k = 0; k < L; ++k) {
        const float a = attw[q][h2][p2][k];
        float4 s4;
        s4.x = ((v[k][0].x * t[k].w[0] + v[k][1].x * t[k].w[1]) + v[k][2].x * t[k].w[2]) + v[k][3].x * t[k].w[3];
        s4.y = ((v[k][0].y * t[k].w[0] + v[k][1].y * t[k].w[1]) + v[k][2].y * t[k].w[2]) + v[k][3].y * t[k].w[3];
        s4.z = ((v[k][0].z * t[k].w[0] + v[k][1].z * t[k].w[1]) + v[k][2].z * t[k].w[2]) + v[k][3].z * t[k].w[3];
        s4.w = ((v[k][0].w * t[k].w[0] + v[k][1].w * t[k].w[1]) + v[k][2].w * t[k].w[2]) + v[k][3].w * t[k].w[3];
        acc.x += s4.x * a; acc.y += s4.y * a; acc.z += s4.z * a; acc.w += s4.w * a;
      }
      acc.x = pt_sum(acc.x); acc.y = pt_sum(acc.y);
      acc.z = pt_sum(acc.z); acc.w = pt_sum(acc.w);
#if MSDA_NT
      if (p2 == 0) {
        f32x4 o4;
        o4[0] = acc.x; o4[1] = acc.y; o4[2] = acc.z; o4[3] = acc.w;
        __builtin_nontemporal_store(
            o4, reinterpret_cast<f32x4*>(out + ((int64_t)b * lv.N + nq) * 256 + h2 * 32 + c4 * 4));
      }
#else
      if (S3OUT) {
        // the output_proj GEMM's A operand, pre-split (csrc/gemm_s3.hip): even-c4 lanes take
        // their neighbour's four channels and write the three 16-byte plane pieces of 8
        // consecutive channels of this token (the fp32 map is then never written)
        const float4 nb = make_float4(__shfl_xor(acc.x, 1, 64), __shfl_xor(acc.y, 1, 64),
                                      __shfl_xor(acc.z, 1, 64), __shfl_xor(acc.w, 1, 64));
        if (p2 == 0 && (c4 & 1) == 0) {
          const float v8[8] = {acc.x, acc.y, acc.z, acc.w, nb.x, nb.y, nb.z, nb.w};
          s3_frag q0, q1, q2;
          s3_split8(v8, q0, q1, q2);
          const int64_t row = (int64_t)b * lv.N + nq;
          const int k0 = h2 * 32 + c4 * 4;
          uint4* o = reinterpret_cast<uint4*>(out) + ((row >> 5) * 16 + (k0 >> 4)) * 192 +
                     ((k0 >> 3) & 1) * 32 + (row & 31);
          o[0] = q0.u; o[64] = q1.u; o[128] = q2.u;
        }
      } else if (p2 == 0) {
        st4(out + ((int64_t)b * lv.N + nq) * 256 + h2 * 32 + c4 * 4, acc);
      }
#endif
    }
  }
}

// Two register budgets of the same body (round 4).  k_msda, the default: bounded to 6+ waves
// per SIMD -- hipcc then finds 62 VGPRs without spilling, 8 workgroups per CU (the hardware
// maximum of 32 waves; 9 KB of LDS each).  k_msda_lo: the compiler's own choice (84 VGPRs, 5
// workgroups per CU), rounds 1-3, selectable with PN_MSDA_LOW_OCCUPANCY.  Same arithmetic,
// bit-identical output; tools/msda_ab.py, same run: 45.9 / 55.9 us against 49.9 / 59.5 us at
// the init offsets (second / first pass of the probe), 63.7 against 65.3 with N(0, 8 px)
// offsets: 1.08 GB of value rows pass the vector L1 in 45.9 us = 23.6 TB/s, 0.9 of what the
// bare gather pattern reaches (26 TB/s, tools/gather_probe.hip).
template <int L, bool S3OUT = false>
__global__ __launch_bounds__(256, 6) void k_msda(const float* __restrict__ value,
                                                 const float* __restrict__ offaw,
                                                 float* __restrict__ out, const MsdaLevels lv,
                                                 const int64_t ldv, const int64_t ldo) {
  msda_one_shot<L, S3OUT>(value, offaw, out, lv, ldv, ldo);
}
template <int L>
__global__ __launch_bounds__(256) void k_msda_lo(const float* __restrict__ value,
                                                 const float* __restrict__ offaw,
                                                 float* __restrict__ out, const MsdaLevels lv,
                                                 const int64_t ldv, const int64_t ldo) {
  msda_one_shot<L>(value, offaw, out, lv, ldv, ldo);
}

// The same two phases as a PERSISTENT, software-pipelined loop (round 4; MEASURED SLOWER,
// kept selectable as the evidence: PN_MSDA_PERSISTENT / PN_MSDA_PERSISTENT_BATCHED).  The
// idea: k_msda above spends a workgroup's whole life on two queries -- dispatch, one round
// trip for the offsets / logits, one for the gathers, exit -- so let a workgroup walk its
// band's query pairs instead: the offsets / logits of pair i + 1 are fetched while pair i
// gathers (registers), the tap records are double-buffered in LDS (one barrier per pair),
// and with BATCH both queries of a pair gather in one unconditional batch (2 x L x 4 float4
// loads in flight per lane; a dead query reads token 0 and is not stored).  Arithmetic and
// summation order are k_msda's: bit-identical output.  On MI355X (tools/msda_ab.py, same
// run): batched 57.5-62.2 us against 49.9-57.5 us one-shot at the init offsets, 70.0 against
// 65.6 with N(0, 8 px) offsets -- more loads in flight per CU do not help a kernel that is
// bound by the vector L1's request rate, and 4 resident workgroups instead of 5 cost more
// than the hidden launch / phase-1 latency buys (LABNOTES.md 6.0).
template <int L, bool BATCH>
__global__ __launch_bounds__(256) void k_msda_pipe(const float* __restrict__ value,
                                                   const float* __restrict__ offaw,
                                                   float* __restrict__ out,
                                                   const MsdaLevels lv, const int64_t ldv,
                                                   const int64_t ldo, const int pairs) {
  __shared__ MsdaTap taps[2][MSDA_TQ][8][4][4];   // [buffer][query][head][point][level]
  __shared__ float attw[2][MSDA_TQ][8][4][4];
  __shared__ int tok[2][MSDA_TQ];
  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  constexpr int LP = L * 4;
  const int band = blockIdx.x & 7, slot = blockIdx.x >> 3, per = gridDim.x >> 3;
  const int qi = tid >> 7, head = (tid >> 4) & 7, pt = (tid >> 2) & 3, l = tid & 3;
  const int lc = min(l, L - 1);
  const int c4 = tid & 7, p2 = (tid >> 3) & 3, h2 = tid >> 5;
  const char* vb = reinterpret_cast<const char*>(value + (int64_t)b * lv.N * ldv + h2 * 32 + c4 * 4);

  // phase-1 inputs of one pair, fetched one iteration ahead
  struct Pre { int n, qw, qh, qs; float e; float2 off; };
  auto fetch = [&](int pair) {
    Pre r;
    int i = pair * MSDA_TQ + qi;            // index inside the band
    r.n = -1; r.qw = 1; r.qh = 1; r.qs = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (k < L && r.n < 0) {
        const int r0 = (band * lv.h[k]) >> 3, r1 = ((band + 1) * lv.h[k]) >> 3;
        const int cnt = (r1 - r0) * lv.w[k];
        if (i < cnt) {
          r.n = lv.start[k] + r0 * lv.w[k] + i;
          r.qw = lv.w[k]; r.qh = lv.h[k]; r.qs = lv.start[k];
        } else {
          i -= cnt;
        }
      }
    }
    const float* oa = offaw + ((int64_t)b * lv.N + max(r.n, 0)) * ldo;
    r.e = oa[8 * LP * 2 + head * LP + lc * 4 + pt];
    r.off = *reinterpret_cast<const float2*>(oa + head * LP * 2 + lc * 8 + pt * 2);
    return r;
  };

  if (slot >= pairs) return;
  Pre cur = fetch(slot);
  int buf = 0;
  for (int it = slot; it < pairs; it += per, buf ^= 1) {
    // ---- phase 1 of pair `it` (from registers) ----
    {
      const int n = cur.n;
      if ((tid & 127) == 0) tok[buf][qi] = n;
      const bool live = n >= 0 && l < L;
      const float e = cur.e;
      const float2 off = cur.off;
      const float mx = grp16_max(l < L ? e : -INFINITY);
      const float ex = l < L ? expf(e - mx) : 0.f;
      const int g0 = (tid & 63) & ~3;
      float den = __shfl(ex, g0, 64);
#pragma unroll
      for (int k = 1; k < L; ++k) den += __shfl(ex, g0 + k, 64);
      den += __shfl_xor(den, 4, 64);
      den += __shfl_xor(den, 8, 64);
      const float aw = ex / den;
      MsdaTap t;
      t.w[0] = t.w[1] = t.w[2] = t.w[3] = 0.f;
      t.off[0] = t.off[1] = t.off[2] = t.off[3] = 0u;
      if (live) {
        const int idx = n - cur.qs;
        const int qy = idx / cur.qw, qx = idx - qy * cur.qw;
        const float ref_x = ((float)qx + 0.5f) / (float)cur.qw;
        const float ref_y = ((float)qy + 0.5f) / (float)cur.qh;
        const int Hl = lv.h[lc], Wl = lv.w[lc];
        const float locx = ref_x + off.x / (float)Wl;
        const float locy = ref_y + off.y / (float)Hl;
        const float gx = 2.f * locx - 1.f, gy = 2.f * locy - 1.f;
        const float ix = ((gx + 1.f) * (float)Wl - 1.f) * 0.5f;
        const float iy = ((gy + 1.f) * (float)Hl - 1.f) * 0.5f;
        const float fx = floorf(ix), fy = floorf(iy);
        const int x0 = (int)fminf(fmaxf(fx, -2.f), (float)Wl), y0 = (int)fminf(fmaxf(fy, -2.f), (float)Hl);
        const float tx = ix - fx, ty = iy - fy;
        const bool xin0 = x0 >= 0 && x0 < Wl, xin1 = x0 + 1 >= 0 && x0 + 1 < Wl;
        const bool yin0 = y0 >= 0 && y0 < Hl, yin1 = y0 + 1 >= 0 && y0 + 1 < Hl;
        const int xa = min(max(x0, 0), Wl - 1), xb = min(max(x0 + 1, 0), Wl - 1);
        const int ya = min(max(y0, 0), Hl - 1), yb = min(max(y0 + 1, 0), Hl - 1);
        t.w[0] = (xin0 && yin0) ? (1.f - tx) * (1.f - ty) : 0.f;
        t.w[1] = (xin1 && yin0) ? tx * (1.f - ty) : 0.f;
        t.w[2] = (xin0 && yin1) ? (1.f - tx) * ty : 0.f;
        t.w[3] = (xin1 && yin1) ? tx * ty : 0.f;
        const unsigned row = (unsigned)ldv * 4u, base = (unsigned)lv.start[lc];
        t.off[0] = (base + (unsigned)(ya * Wl + xa)) * row;
        t.off[1] = (base + (unsigned)(ya * Wl + xb)) * row;
        t.off[2] = (base + (unsigned)(yb * Wl + xa)) * row;
        t.off[3] = (base + (unsigned)(yb * Wl + xb)) * row;
      }
      if (l < L) {          // (dead queries park row 0 with weight 0: gathers stay unconditional)
        taps[buf][qi][head][pt][lc] = t;
        attw[buf][qi][head][pt][lc] = live ? aw : 0.f;
      }
    }
    __syncthreads();
    // the next pair's offsets / logits travel while this pair gathers
    if (it + per < pairs) cur = fetch(it + per);
    // ---- phase 2: BATCH: both queries' gathers in one batch (2 x L x 4 loads in flight per
    // lane); otherwise one query at a time like k_msda (L x 4 in flight, fewer registers) ----
    constexpr int NQ = BATCH ? MSDA_TQ : 1;
#pragma unroll
    for (int qb = 0; qb < MSDA_TQ; qb += NQ) {
      float4 v[NQ][L][4];
      MsdaTap t[NQ][L];
#pragma unroll
      for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int k = 0; k < L; ++k) {
          t[q][k] = taps[buf][qb + q][h2][p2][k];
#pragma unroll
          for (int j = 0; j < 4; ++j)
            v[q][k][j] = *reinterpret_cast<const float4*>(vb + t[q][k].off[j]);
        }
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int nq = tok[buf][qb + q];
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < L; ++k) {
          const float a = attw[buf][qb + q][h2][p2][k];
          const float4* vv = v[q][k];
          const float* w = t[q][k].w;
          float4 s4;
          s4.x = ((vv[0].x * w[0] + vv[1].x * w[1]) + vv[2].x * w[2]) + vv[3].x * w[3];
          s4.y = ((vv[0].y * w[0] + vv[1].y * w[1]) + vv[2].y * w[2]) + vv[3].y * w[3];
          s4.z = ((vv[0].z * w[0] + vv[1].z * w[1]) + vv[2].z * w[2]) + vv[3].z * w[3];
          s4.w = ((vv[0].w * w[0] + vv[1].w * w[1]) + vv[2].w * w[2]) + vv[3].w * w[3];
          acc.x += s4.x * a; acc.y += s4.y * a; acc.z += s4.z * a; acc.w += s4.w * a;
        }
        acc.x = pt_sum(acc.x); acc.y = pt_sum(acc.y);
        acc.z = pt_sum(acc.z); acc.w = pt_sum(acc.w);
        if (p2 == 0 && nq >= 0) st4(out + ((int64_t)b * lv.N + nq) * 256 + h2 * 32 + c4 * 4, acc);
      }
    }
  }
}

template <bool BATCH>
static int msda_resident_wgs() {
  static int n = 0;
  if (n == 0) {
    int k = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&k, k_msda_pipe<3, BATCH>, 256, 0) != hipSuccess || k < 1)
      k = 2;
    n = k > 8 ? 8 : k;
  }
  return n;
}

extern "C" int pn_msda_ex_f32(const float* value, int64_t ld_value, const float* offaw,
                              int64_t ld_offaw, float* out, int B, int L, const int32_t* level_h,
                              const int32_t* level_w, int flags, void* stream) {
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
  if ((int64_t)n * ld_value * 4 >= ((int64_t)1 << 32)) return PN_BAD_ARG;   // 32-bit tap offsets
  hipStream_t s = (hipStream_t)stream;
  const int pairs = (per_band + MSDA_TQ - 1) / MSDA_TQ;
  if ((flags & PN_MSDA_S3_OUT) && (flags & (PN_MSDA_PERSISTENT | PN_MSDA_PERSISTENT_BATCHED | PN_MSDA_LOW_OCCUPANCY)))
    return PN_BAD_ARG;                 // the pre-split output exists in the default form only
  if (!(flags & (PN_MSDA_PERSISTENT | PN_MSDA_PERSISTENT_BATCHED))) {
    // one workgroup per query pair
    const dim3 grid(pairs * 8, B);
    if (flags & PN_MSDA_LOW_OCCUPANCY) {
      switch (L) {
        case 1: hipLaunchKernelGGL(k_msda_lo<1>, grid, dim3(256), 0, s, value, offaw, out, lv, ld_value, ld_offaw); break;
        case 2: hipLaunchKernelGGL(k_msda_lo<2>, grid, dim3(256), 0, s, value, offaw, out, lv, ld_value, ld_offaw); break;
        case 3: hipLaunchKernelGGL(k_msda_lo<3>, grid, dim3(256), 0, s, value, offaw, out, lv, ld_value, ld_offaw); break;
        default: hipLaunchKernelGGL(k_msda_lo<4>, grid, dim3(256), 0, s, value, offaw, out, lv, ld_value, ld_offaw); break;
      }
      return PN_LAUNCH_CHECK();
    }
    if (flags & PN_MSDA_S3_OUT) {      // `out` is an S3 operand [B * N x 256] (pn_gemm_s3_f32's A)
      switch (L) {
        case 1: hipLaunchKernelGGL((k_msda<1, true>), grid, dim3(256), 0, s, value, offaw, out, lv, ld_value, ld_offaw); break;
        case 2: hipLaunchKernelGGL((k_msda<2, true>), grid, dim3(256), 0, s, value, offaw, out, lv, ld_value, ld_offaw); break;
        case 3: hipLaunchKernelGGL((k_msda<3, true>), grid, dim3(256), 0, s, value, offaw, out, lv, ld_value, ld_offaw); break;
        default: hipLaunchKernelGGL((k_msda<4, true>), grid, dim3(256), 0, s, value, offaw, out, lv, ld_value, ld_offaw); break;
      }
      return PN_LAUNCH_CHECK();
    }
    switch (L) {
      case 1: hipLaunchKernelGGL(k_msda<1>, grid, dim3(256), 0, s, value, offaw, out, lv, ld_value, ld_offaw); break;
      case 2: hipLaunchKernelGGL(k_msda<2>, grid, dim3(256), 0, s, value, offaw, out, lv, ld_value, ld_offaw); break;
      case 3: hipLaunchKernelGGL(k_msda<3>, grid, dim3(256), 0, s, value, offaw, out, lv, ld_value, ld_offaw); break;
      default: hipLaunchKernelGGL(k_msda<4>, grid, dim3(256), 0, s, value, offaw, out, lv, ld_value, ld_offaw); break;
    }
    return PN_LAUNCH_CHECK();
  }
  // persistent: every CU's resident slots, spread evenly over the 8 bands (XCDs)
  const bool batched = flags & PN_MSDA_PERSISTENT_BATCHED;
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess)
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  int slots = cus * (batched ? msda_resident_wgs<true>() : msda_resident_wgs<false>()) / 8;
  if (B > 1) slots = (slots + B - 1) / B;
  if (slots > pairs) slots = pairs;
  if (slots < 1) slots = 1;
  const dim3 grid(slots * 8, B);
#define PN_MSDA_PIPE(LL)                                                                       \
  if (batched)                                                                                 \
    hipLaunchKernelGGL((k_msda_pipe<LL, true>), grid, dim3(256), 0, s, value, offaw, out, lv,  \
                       ld_value, ld_offaw, pairs);                                             \
  else                                                                                         \
    hipLaunchKernelGGL((k_msda_pipe<LL, false>), grid, dim3(256), 0, s, value, offaw, out, lv, \
                       ld_value, ld_offaw, pairs)
  switch (L) {
    case 1: PN_MSDA_PIPE(1); break;
    case 2: PN_MSDA_PIPE(2); break;
    case 3: PN_MSDA_PIPE(3); break;
    default: PN_MSDA_PIPE(4); break;
  }
#undef PN_MSDA_PIPE
  return PN_LAUNCH_CHECK();
}

extern "C" int pn_msda_f32(const float* value, int64_t ld_value, const float* offaw,
                           int64_t ld_offaw, float* out, int B, int L, const int32_t* level_h,
                           const int32_t* level_w, void* stream) {
  return pn_msda_ex_f32(value, ld_value, offaw, ld_offaw, out, B, L, level_h, level_w, 0, stream);
}

// ---------------------------------------------------------------------------------
// The operator in mmcv's own shape: arbitrary queries, explicit (already normalised)
// sampling locations and (already soft-maxed) attention weights, level geometry read from
// the DEVICE tensors mmcv passes -- a drop-in behind the unmodified
// MultiScaleDeformableAttention.forward, and the cross-attention of Deformable-DETR
// decoders (query-side sampling).  Same two phases as k_msda; the output of query q is
//   out[b][q][h][:] = sum_{l,p} w[b][q][h][l][p] * bilinear(value_l[b][:, h, :], loc[b][q][h][l][p])
// with zero padding, `loc` in [0, 1] x [0, 1] as (x, y).
template <int L>
__global__ __launch_bounds__(256, 6) void k_msda_loc(const float* __restrict__ value,
                                                  const int64_t* __restrict__ shapes,
                                                  const int64_t* __restrict__ starts,
                                                  const float* __restrict__ loc,
                                                  const float* __restrict__ aw,
                                                  float* __restrict__ out, const int N,
                                                  const int Nq, const int64_t ldv) {
  __shared__ MsdaTap taps[MSDA_TQ][8][4][4];   // [query][head][point][level]
  __shared__ float attw[MSDA_TQ][8][4][4];
  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  constexpr int LP = L * 4;
  {
    const int qi = tid >> 7, head = (tid >> 4) & 7, pt = (tid >> 2) & 3, l = tid & 3;
    const int q = blockIdx.x * MSDA_TQ + qi;
    if (q < Nq && l < L) {
      const int Hl = (int)shapes[2 * l], Wl = (int)shapes[2 * l + 1];
      const int64_t slot = (((int64_t)b * Nq + q) * 8 + head) * LP + l * 4 + pt;
      const float2 lc = *reinterpret_cast<const float2*>(loc + 2 * slot);
      const float gx = 2.f * lc.x - 1.f, gy = 2.f * lc.y - 1.f;
      const float ix = ((gx + 1.f) * (float)Wl - 1.f) * 0.5f;
      const float iy = ((gy + 1.f) * (float)Hl - 1.f) * 0.5f;
      const float fx = floorf(ix), fy = floorf(iy);
      const int x0 = (int)fminf(fmaxf(fx, -2.f), (float)Wl), y0 = (int)fminf(fmaxf(fy, -2.f), (float)Hl);
      const float tx = ix - fx, ty = iy - fy;
      const bool xin0 = x0 >= 0 && x0 < Wl, xin1 = x0 + 1 >= 0 && x0 + 1 < Wl;
      const bool yin0 = y0 >= 0 && y0 < Hl, yin1 = y0 + 1 >= 0 && y0 + 1 < Hl;
      const int xa = min(max(x0, 0), Wl - 1), xb = min(max(x0 + 1, 0), Wl - 1);
      const int ya = min(max(y0, 0), Hl - 1), yb = min(max(y0 + 1, 0), Hl - 1);
      MsdaTap t;
      t.w[0] = (xin0 && yin0) ? (1.f - tx) * (1.f - ty) : 0.f;
      t.w[1] = (xin1 && yin0) ? tx * (1.f - ty) : 0.f;
      t.w[2] = (xin0 && yin1) ? (1.f - tx) * ty : 0.f;
      t.w[3] = (xin1 && yin1) ? tx * ty : 0.f;
      const unsigned row = (unsigned)ldv * 4u, base = (unsigned)starts[l];
      t.off[0] = (base + (unsigned)(ya * Wl + xa)) * row;
      t.off[1] = (base + (unsigned)(ya * Wl + xb)) * row;
      t.off[2] = (base + (unsigned)(yb * Wl + xa)) * row;
      t.off[3] = (base + (unsigned)(yb * Wl + xb)) * row;
      taps[qi][head][pt][l] = t;
      attw[qi][head][pt][l] = aw[slot];
    }
  }
  __syncthreads();
  const int c4 = tid & 7, p2 = (tid >> 3) & 3, h2 = tid >> 5;
  const char* vb = reinterpret_cast<const char*>(value + (int64_t)b * N * ldv + h2 * 32 + c4 * 4);
#pragma unroll
  for (int qi = 0; qi < MSDA_TQ; ++qi) {
    const int q = blockIdx.x * MSDA_TQ + qi;
    if (q >= Nq) continue;                    // (uniform)
    float4 v[L][4];
    MsdaTap t[L];
#pragma unroll
    for (int k = 0; k < L; ++k) {
      t[k] = taps[qi][h2][p2][k];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[k][j] = *reinterpret_cast<const float4*>(vb + t[k].off[j]);
    }
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < L; ++k) {
      const float a = attw[qi][h2][p2][k];
      float4 s4;
      s4.x = ((v[k][0].x * t[k].w[0] + v[k][1].x * t[k].w[1]) + v[k][2].x * t[k].w[2]) + v[k][3].x * t[k].w[3];
      s4.y = ((v[k][0].y * t[k].w[0] + v[k][1].y * t[k].w[1]) + v[k][2].y * t[k].w[2]) + v[k][3].y * t[k].w[3];
      s4.z = ((v[k][0].z * t[k].w[0] + v[k][1].z * t[k].w[1]) + v[k][2].z * t[k].w[2]) + v[k][3].z * t[k].w[3];
      s4.w = ((v[k][0].w * t[k].w[0] + v[k][1].w * t[k].w[1]) + v[k][2].w * t[k].w[2]) + v[k][3].w * t[k].w[3];
      acc.x += s4.x * a; acc.y += s4.y * a; acc.z += s4.z * a; acc.w += s4.w * a;
    }
    acc.x = pt_sum(acc.x); acc.y = pt_sum(acc.y);
    acc.z = pt_sum(acc.z); acc.w = pt_sum(acc.w);
    if (p2 == 0) st4(out + ((int64_t)b * Nq + q) * 256 + h2 * 32 + c4 * 4, acc);
  }
}

extern "C" int pn_msda_loc_f32(const float* value, int64_t ld_value,
                               const int64_t* spatial_shapes, const int64_t* level_start_index,
                               const float* sampling_locations, const float* attention_weights,
                               float* out, int B, int N, int Nq, int L, void* stream) {
  if (!value || !spatial_shapes || !level_start_index || !sampling_locations ||
      !attention_weights || !out || B <= 0 || N <= 0 || Nq <= 0 || L <= 0 || L > 4)
    return PN_BAD_ARG;
  if (ld_value < 256 || (ld_value & 3) || ((uintptr_t)value & 15) ||
      ((uintptr_t)sampling_locations & 7))
    return PN_BAD_ARG;
  if ((int64_t)N * ld_value * 4 >= ((int64_t)1 << 32)) return PN_BAD_ARG;   // 32-bit tap offsets
  const dim3 grid((Nq + MSDA_TQ - 1) / MSDA_TQ, B);
  hipStream_t s = (hipStream_t)stream;
#define PN_MSDA_LOC(LL)                                                                       \
  hipLaunchKernelGGL(k_msda_loc<LL>, grid, dim3(256), 0, s, value, spatial_shapes,            \
                     level_start_index, sampling_locations, attention_weights, out, N, Nq,    \
                     ld_value)
  switch (L) {
    case 1: PN_MSDA_LOC(1); break;
    case 2: PN_MSDA_LOC(2); break;
    case 3: PN_MSDA_LOC(3); break;
    default: PN_MSDA_LOC(4); break;
  }
#undef PN_MSDA_LOC
  return PN_LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------
// Backward of the operator above, in mmcv's shape (`ext_module.ms_deform_attn_backward`,
// mmcv/ops/csrc/pytorch/cuda/ms_deform_attn_cuda.cu: ms_deformable_col2im): for every
// (batch, query, head, level, point) with sampling location inside (-1, H) x (-1, W) in
// pixel coordinates
//   g      = grad_output[b][q][h][:]                                     (32 channels)
//   grad_value[b][tap_i][h][:]       += w_i * a * g        for the 4 bilinear taps (atomic)
//   grad_attn_weight[b][q][h][l][p]   = sum_c g_c * bilinear(value)_c
//   grad_sampling_loc[...][0] (x)     = W_l * sum_c a g_c * (-hh v1 + hh v2 - lh v3 + lh v4)_c
//   grad_sampling_loc[...][1] (y)     = H_l * sum_c a g_c * (-hw v1 - lw v2 + hw v3 + lw v4)_c
// with (lh, lw) the fractional parts, hh = 1 - lh, hw = 1 - lw, taps outside the map
// contributing value 0 -- mmcv's ms_deform_attn_col2im_bilinear.  Thread layout of the forward
// kernel: (head, point, 4 channels), the 8 lanes of a point own one 128-byte value row per
// tap; the channel sums are xor-1/2/4 shuffles (mmcv reduces through shared memory in the
// same way, one thread per channel).  grad_value is accumulated with hardware fp32 atomics
// into a buffer the CALLER zeroes (mmcv: at::zeros_like(value)): like mmcv's, its summation
// order is not deterministic; the other two outputs are.
template <int L>
__global__ __launch_bounds__(256) void k_msda_bwd(const float* __restrict__ value,
                                                  const int64_t* __restrict__ shapes,
                                                  const int64_t* __restrict__ starts,
                                                  const float* __restrict__ loc,
                                                  const float* __restrict__ aw,
                                                  const float* __restrict__ gout,
                                                  float* __restrict__ gvalue,
                                                  float* __restrict__ gloc,
                                                  float* __restrict__ gaw, const int N,
                                                  const int Nq, const int64_t ldv) {
  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  const int q = blockIdx.x;
  constexpr int LP = L * 4;
  const int c4 = tid & 7, pt = (tid >> 3) & 3, head = tid >> 5;
  const float4 g = ld4(gout + ((int64_t)b * Nq + q) * 256 + head * 32 + c4 * 4);
  const float* vb = value + (int64_t)b * N * ldv + head * 32 + c4 * 4;
  float* gvb = gvalue + (int64_t)b * N * ldv + head * 32 + c4 * 4;
#pragma unroll
  for (int l = 0; l < L; ++l) {
    const int Hl = (int)shapes[2 * l], Wl = (int)shapes[2 * l + 1];
    const int64_t slot = (((int64_t)b * Nq + q) * 8 + head) * LP + l * 4 + pt;
    const float2 lc = *reinterpret_cast<const float2*>(loc + 2 * slot);
    const float a = aw[slot];
    const float h_im = lc.y * (float)Hl - 0.5f, w_im = lc.x * (float)Wl - 0.5f;
    float ga = 0.f, gx = 0.f, gy = 0.f;
    if (h_im > -1.f && w_im > -1.f && h_im < (float)Hl && w_im < (float)Wl) {   // (uniform in the 8 lanes)
      const float fy = floorf(h_im), fx = floorf(w_im);
      const int y0 = (int)fy, x0 = (int)fx;
      const float lh = h_im - fy, lw = w_im - fx, hh = 1.f - lh, hw = 1.f - lw;
      const bool yin0 = y0 >= 0, yin1 = y0 + 1 <= Hl - 1, xin0 = x0 >= 0, xin1 = x0 + 1 <= Wl - 1;
      const int64_t base = starts[l];
      const int ya = max(y0, 0), yb = min(y0 + 1, Hl - 1), xa = max(x0, 0), xb = min(x0 + 1, Wl - 1);
      const int64_t r1 = (base + (int64_t)ya * Wl + xa) * ldv, r2 = (base + (int64_t)ya * Wl + xb) * ldv;
      const int64_t r3 = (base + (int64_t)yb * Wl + xa) * ldv, r4 = (base + (int64_t)yb * Wl + xb) * ldv;
      const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
      // unconditional (clamped) loads, zeroed where the tap lies outside the map
      float4 v1 = ld4(vb + r1), v2 = ld4(vb + r2), v3 = ld4(vb + r3), v4 = ld4(vb + r4);
      const bool in1 = yin0 && xin0, in2 = yin0 && xin1, in3 = yin1 && xin0, in4 = yin1 && xin1;
      v1 = in1 ? v1 : z; v2 = in2 ? v2 : z; v3 = in3 ? v3 : z; v4 = in4 ? v4 : z;
      const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
      const float4 tg = make_float4(g.x * a, g.y * a, g.z * a, g.w * a);   // top_grad * attn_weight
#define PN_BWD_CH(c)                                                                          \
      {                                                                                       \
        const float gh = -hw * v1.c - lw * v2.c + hw * v3.c + lw * v4.c;                      \
        const float gw = -hh * v1.c - lh * v3.c + hh * v2.c + lh * v4.c;                      \
        const float val = w1 * v1.c + w2 * v2.c + w3 * v3.c + w4 * v4.c;                      \
        ga += g.c * val;                                                                      \
        gx += (float)Wl * gw * tg.c;                                                          \
        gy += (float)Hl * gh * tg.c;                                                          \
      }
      PN_BWD_CH(x) PN_BWD_CH(y) PN_BWD_CH(z) PN_BWD_CH(w)
#undef PN_BWD_CH
      if (in1) { unsafeAtomicAdd(gvb + r1 + 0, w1 * tg.x); unsafeAtomicAdd(gvb + r1 + 1, w1 * tg.y);
                 unsafeAtomicAdd(gvb + r1 + 2, w1 * tg.z); unsafeAtomicAdd(gvb + r1 + 3, w1 * tg.w); }
      if (in2) { unsafeAtomicAdd(gvb + r2 + 0, w2 * tg.x); unsafeAtomicAdd(gvb + r2 + 1, w2 * tg.y);
                 unsafeAtomicAdd(gvb + r2 + 2, w2 * tg.z); unsafeAtomicAdd(gvb + r2 + 3, w2 * tg.w); }
      if (in3) { unsafeAtomicAdd(gvb + r3 + 0, w3 * tg.x); unsafeAtomicAdd(gvb + r3 + 1, w3 * tg.y);
                 unsafeAtomicAdd(gvb + r3 + 2, w3 * tg.z); unsafeAtomicAdd(gvb + r3 + 3, w3 * tg.w); }
      if (in4) { unsafeAtomicAdd(gvb + r4 + 0, w4 * tg.x); unsafeAtomicAdd(gvb + r4 + 1, w4 * tg.y);
                 unsafeAtomicAdd(gvb + r4 + 2, w4 * tg.z); unsafeAtomicAdd(gvb + r4 + 3, w4 * tg.w); }
    }
    // sum over the point's 32 channels: its 8 lanes (fixed order: deterministic)
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
      ga += __shfl_xor(ga, o, 64);
      gx += __shfl_xor(gx, o, 64);
      gy += __shfl_xor(gy, o, 64);
    }
    if (c4 == 0) {
      gaw[slot] = ga;
      *reinterpret_cast<float2*>(gloc + 2 * slot) = make_float2(gx, gy);
    }
  }
}

// The same backward with lane = channel (round 6): a wave owns one head and two points at a
// time, its 32-lane halves each cover ONE 128-byte value row per tap, so every atomic instruction
// touches two whole lines instead of eight 16-byte-strided partial ones (4x fewer line visits at
// the L2's atomic units, which is where this kernel's time goes: 270 M fp32 atomics per launch at
// 21 950 queries).  The channel sums are 5-step shuffles inside a half.  -DPN_MSDA_BWD_LANES=0
// builds the first form instead (A/B with a variant library).
template <int L>
__global__ __launch_bounds__(512) void k_msda_bwd_lanes(const float* __restrict__ value,
                                                        const int64_t* __restrict__ shapes,
                                                        const int64_t* __restrict__ starts,
                                                        const float* __restrict__ loc,
                                                        const float* __restrict__ aw,
                                                        const float* __restrict__ gout,
                                                        float* __restrict__ gvalue,
                                                        float* __restrict__ gloc,
                                                        float* __restrict__ gaw, const int N,
                                                        const int Nq, const int64_t ldv) {
  const int tid = threadIdx.x;
  const int b = blockIdx.y, q = blockIdx.x;
  constexpr int LP = L * 4;
  const int c = tid & 31, half = (tid >> 5) & 1, head = tid >> 6;
  const float g = gout[((int64_t)b * Nq + q) * 256 + head * 32 + c];
  const float* vb = value + (int64_t)b * N * ldv + head * 32 + c;
  float* gvb = gvalue + (int64_t)b * N * ldv + head * 32 + c;
#pragma unroll
  for (int l = 0; l < L; ++l) {
    const int Hl = (int)shapes[2 * l], Wl = (int)shapes[2 * l + 1];
    const int64_t base = starts[l];
#pragma unroll
    for (int pp = 0; pp < 2; ++pp) {
      const int pt = pp * 2 + half;
      const int64_t slot = (((int64_t)b * Nq + q) * 8 + head) * LP + l * 4 + pt;
      const float2 lc = *reinterpret_cast<const float2*>(loc + 2 * slot);
      const float a = aw[slot];
      const float h_im = lc.y * (float)Hl - 0.5f, w_im = lc.x * (float)Wl - 0.5f;
      float ga = 0.f, gx = 0.f, gy = 0.f;
      if (h_im > -1.f && w_im > -1.f && h_im < (float)Hl && w_im < (float)Wl) {   // (uniform in the half)
        const float fy = floorf(h_im), fx = floorf(w_im);
        const int y0 = (int)fy, x0 = (int)fx;
        const float lh = h_im - fy, lw = w_im - fx, hh = 1.f - lh, hw = 1.f - lw;
        const bool yin0 = y0 >= 0, yin1 = y0 + 1 <= Hl - 1, xin0 = x0 >= 0, xin1 = x0 + 1 <= Wl - 1;
        const int ya = max(y0, 0), yb = min(y0 + 1, Hl - 1), xa = max(x0, 0), xb = min(x0 + 1, Wl - 1);
        const int64_t r1 = (base + (int64_t)ya * Wl + xa) * ldv, r2 = (base + (int64_t)ya * Wl + xb) * ldv;
        const int64_t r3 = (base + (int64_t)yb * Wl + xa) * ldv, r4 = (base + (int64_t)yb * Wl + xb) * ldv;
        const bool in1 = yin0 && xin0, in2 = yin0 && xin1, in3 = yin1 && xin0, in4 = yin1 && xin1;
        float v1 = vb[r1], v2 = vb[r2], v3 = vb[r3], v4 = vb[r4];
        v1 = in1 ? v1 : 0.f; v2 = in2 ? v2 : 0.f; v3 = in3 ? v3 : 0.f; v4 = in4 ? v4 : 0.f;
        const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
        const float tg = g * a;                               // top_grad * attn_weight
        const float gh = -hw * v1 - lw * v2 + hw * v3 + lw * v4;
        const float gw = -hh * v1 - lh * v3 + hh * v2 + lh * v4;
        ga = g * (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4);
        gx = (float)Wl * gw * tg;
        gy = (float)Hl * gh * tg;
        if (in1) unsafeAtomicAdd(gvb + r1, w1 * tg);
        if (in2) unsafeAtomicAdd(gvb + r2, w2 * tg);
        if (in3) unsafeAtomicAdd(gvb + r3, w3 * tg);
        if (in4) unsafeAtomicAdd(gvb + r4, w4 * tg);
      }
      // sum over the point's 32 channels: the half's lanes (fixed order: deterministic)
#pragma unroll
      for (int o = 1; o < 32; o <<= 1) {
        ga += __shfl_xor(ga, o, 64);
        gx += __shfl_xor(gx, o, 64);
        gy += __shfl_xor(gy, o, 64);
      }
      if (c == 0) {
        gaw[slot] = ga;
        *reinterpret_cast<float2*>(gloc + 2 * slot) = make_float2(gx, gy);
      }
    }
  }
}

#ifndef PN_MSDA_BWD_LANES        // build-time A/B knob (tools/build_variant.py): 0 = the first form
#define PN_MSDA_BWD_LANES 1
#endif
static constexpr int msda_bwd_lanes() { return PN_MSDA_BWD_LANES; }

extern "C" int pn_msda_bwd_f32(const float* value, int64_t ld_value,
                               const int64_t* spatial_shapes, const int64_t* level_start_index,
                               const float* sampling_locations, const float* attention_weights,
                               const float* grad_output, float* grad_value,
                               float* grad_sampling_loc, float* grad_attn_weight, int B, int N,
                               int Nq, int L, void* stream) {
  if (!value || !spatial_shapes || !level_start_index || !sampling_locations ||
      !attention_weights || !grad_output || !grad_value || !grad_sampling_loc ||
      !grad_attn_weight || B <= 0 || N <= 0 || Nq <= 0 || L <= 0 || L > 4)
    return PN_BAD_ARG;
  if (ld_value < 256 || (ld_value & 3) || ((uintptr_t)value & 15) || ((uintptr_t)grad_value & 15) ||
      ((uintptr_t)grad_output & 15) || ((uintptr_t)sampling_locations & 7) ||
      ((uintptr_t)grad_sampling_loc & 7))
    return PN_BAD_ARG;
  const dim3 grid(Nq, B);
  hipStream_t s = (hipStream_t)stream;
#define PN_MSDA_BWD(LL)                                                                        \
  if (msda_bwd_lanes())                                                                        \
    hipLaunchKernelGGL(k_msda_bwd_lanes<LL>, grid, dim3(512), 0, s, value, spatial_shapes,     \
                       level_start_index, sampling_locations, attention_weights, grad_output,  \
                       grad_value, grad_sampling_loc, grad_attn_weight, N, Nq, ld_value);      \
  else                                                                                         \
    hipLaunchKernelGGL(k_msda_bwd<LL>, grid, dim3(256), 0, s, value, spatial_shapes,           \
                       level_start_index, sampling_locations, attention_weights, grad_output,  \
                       grad_value, grad_sampling_loc, grad_attn_weight, N, Nq, ld_value)
  switch (L) {
    case 1: PN_MSDA_BWD(1); break;
    case 2: PN_MSDA_BWD(2); break;
    case 3: PN_MSDA_BWD(3); break;
    default: PN_MSDA_BWD(4); break;
  }
#undef PN_MSDA_BWD
  return PN_LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------
// Diagnostic: the BARE access pattern of the sampling kernels -- 8 lanes read one random
// 128-byte line as float4s, 12 independent lines per 8-lane group (3 levels x 4 taps), index
// loads hoisted in front of the gathers, a sum as the only arithmetic -- so that bench.py can
// measure, in the same run and on the same board, the rate the vector L1 / L2 deliver for
// this pattern (`roofline_deformable_sampling.gather_peak`): the roof k_msda is actually under
// once its value rows are L2-resident.  lines = line_mask + 1 (a power of two) 128-byte lines
// are drawn from; idx [workgroups][32][12] int32; out [workgroups][256].
__global__ __launch_bounds__(256) void k_gather_probe(const float* __restrict__ v,
                                                      const int* __restrict__ idx,
                                                      float* __restrict__ out, const int mask) {
  const int tid = threadIdx.x, c = tid & 7, grp = tid >> 3;
  float4 r[12];
#pragma unroll
  for (int j = 0; j < 12; ++j) {
    const int line = idx[((int64_t)blockIdx.x * 32 + grp) * 12 + j] & mask;
    r[j] = ld4(v + (int64_t)line * 32 + c * 4);
  }
  float acc = 0.f;
#pragma unroll
  for (int j = 0; j < 12; ++j) acc += (r[j].x + r[j].y) + (r[j].z + r[j].w);
  out[(int64_t)blockIdx.x * 256 + tid] = acc;
}

extern "C" int pn_gather_probe_f32(const float* lines, const int32_t* idx, float* out,
                                   int workgroups, int line_mask, void* stream) {
  if (!lines || !idx || !out || workgroups <= 0 || line_mask < 0 || (line_mask & (line_mask + 1)))
    return PN_BAD_ARG;
  hipLaunchKernelGGL(k_gather_probe, dim3(workgroups), dim3(256), 0, (hipStream_t)stream, lines,
                     idx, out, line_mask);
  return PN_LAUNCH_CHECK();
}
