// Multi-head attention core for the Mask2Former / Relation Fusion decoders:
// 8 heads x 32, Q <= 256 queries, Nk keys from 100 to 16 700, optional boolean mask
// shared by the heads.  Flash-style over key chunks so the masked cross-attention
// over the 16 700-token level fills the chip; both contractions run on the f32 MFMA.
//
// Workgroup = (key chunk, head, image, 128-query group), 4 waves, wave w owns queries
// [32w, 32w+32).  Per 32-key tile, with K/V tiles staged in LDS:
//   S^T[key][q] = K . Q^T       A = K tile (b128 LDS reads), B = Q rows held in VGPRs
//   online softmax               the S^T accumulator layout puts query = lane&31 in
//                                every register, so max/sum are in-lane + one xor-32
//   O^T[d][q]  += V^T . P^T      P^T is consumed straight from the S^T registers as
//                                the MFMA B operand (k-permutation: MFMA t pairs the
//                                keys held in register t of the two half-waves); A = V
// Partial (O, m, l) per chunk go to scratch; k_attn_combine merges the chunks.
#include "common.h"

#define ATT_LD 36  // LDS row stride (floats): conflict-free b128 reads, see gemm.hip

__global__ __launch_bounds__(256) void k_mask_pack(const float* __restrict__ logits,
                                                   uint32_t* __restrict__ bits,
                                                   int32_t* __restrict__ rowall, int Nk) {
  __shared__ int any_unmasked;
  const int64_t row = blockIdx.x;
  const int nwords = (Nk + 31) / 32;
  if (threadIdx.x == 0) any_unmasked = 0;
  __syncthreads();
  const float* lr = logits + row * Nk;
  uint32_t* br = bits + row * nwords;
  bool seen = false;
  for (int base = 0; base < Nk; base += 256) {
    const int i = base + threadIdx.x;
    const bool valid = i < Nk;
    const bool masked = valid ? (lr[i] < 0.f) : false;
    seen |= valid && !masked;
    const unsigned long long bal = __ballot(masked);
    const int lane = threadIdx.x & 63;
    const int w0 = (base + (threadIdx.x & ~63)) / 32;
    if (lane == 0 && w0 < nwords) br[w0] = (uint32_t)bal;
    if (lane == 32 && w0 + 1 < nwords) br[w0 + 1] = (uint32_t)(bal >> 32);
  }
  if (seen) any_unmasked = 1;  // benign race: all writers store 1
  __syncthreads();
  if (threadIdx.x == 0) rowall[row] = any_unmasked ? 0 : 1;
}

extern "C" int pn_mask_pack(const float* logits, uint32_t* bits, int32_t* rowall, int64_t R,
                            int Nk, void* stream) {
  if (!logits || !bits || !rowall || R <= 0 || Nk <= 0) return PN_BAD_ARG;
  hipLaunchKernelGGL(k_mask_pack, dim3((unsigned)R), dim3(256), 0, (hipStream_t)stream, logits,
                     bits, rowall, Nk);
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

__global__ __launch_bounds__(256) void k_attn_chunk(const AttnP p) {
  __shared__ __attribute__((aligned(16))) float smem[2 * 32 * ATT_LD];
  float* Ks = smem;
  float* Vs = smem + 32 * ATT_LD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int c = blockIdx.x, head = blockIdx.y;
  const int b = blockIdx.z / ((p.Q + 127) / 128), qg = blockIdx.z % ((p.Q + 127) / 128);
  const int q0 = qg * 128 + wave * 32;
  const int myq = q0 + li;                 // this lane's query column
  const bool q_ok = myq < p.Q;
  const int qc = q_ok ? myq : p.Q - 1;

  // Q fragment: B operand of S^T.  MFMA t uses d = t + 16*lh.
  float qf[16];
  {
    const float* qp = p.q + ((int64_t)b * p.Q + qc) * p.ldq + head * 32 + 16 * lh;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float4 v = ld4(qp + 4 * u);
      qf[4 * u + 0] = v.x * p.scale; qf[4 * u + 1] = v.y * p.scale;
      qf[4 * u + 2] = v.z * p.scale; qf[4 * u + 3] = v.w * p.scale;
    }
  }
  const bool use_mask = p.bits != nullptr;
  bool row_unmask = false;
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
  const float* kb = p.k + (int64_t)b * p.Nk * p.ldk + head * 32;
  const float* vb = p.v + (int64_t)b * p.Nk * p.ldv + head * 32;
  const int lrow = tid >> 3, lcol = (tid & 7) * 4;  // staging: one float4 per thread

  for (int k0 = kbeg; k0 < kend; k0 += 32) {
    {
      const int key = k0 + lrow;
      float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
      if (key < kend) {
        kv = ld4(kb + (int64_t)key * p.ldk + lcol);
        vv = ld4(vb + (int64_t)key * p.ldv + lcol);
      }
      __syncthreads();  // previous tile fully consumed
      st4(Ks + lrow * ATT_LD + lcol, kv);
      st4(Vs + lrow * ATT_LD + lcol, vv);
      __syncthreads();
    }
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
    // ---- mask + online softmax (register r <-> key k0 + mfma32_row(r, lh)) ----
    uint32_t mw = 0;
    if (use_mask && !row_unmask) mw = brow[k0 >> 5];
    float tmax = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int kk = mfma32_row(r, lh);
      const bool dead = (k0 + kk >= kend) || ((mw >> kk) & 1u);
      s[r] = dead ? -INFINITY : s[r];
      tmax = fmaxf(tmax, s[r]);
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float m_new = fmaxf(m_run, tmax);
    const float alpha = (m_new == -INFINITY) ? 1.f : expf(m_run - m_new);
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float pv = (s[r] == -INFINITY) ? 0.f : expf(s[r] - m_new);
      s[r] = pv;
      psum += pv;
    }
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] *= alpha;
    // ---- O^T += V^T P^T : MFMA t pairs keys mfma32_row(t, 0) / mfma32_row(t, 1) ----
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const float a = Vs[mfma32_row(t, lh) * ATT_LD + li];
      o = mfma32(a, s[t], o);
    }
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
// One workgroup per (b, q); the 32 lanes of head h first sweep the chunk statistics
// in parallel (lane = chunk) to get M, the weights (kept in LDS) and the denominator,
// then lane d accumulates the weighted partials with independent loads.
#define ATT_MAXCH 256
__global__ __launch_bounds__(256) void k_attn_combine(const float* __restrict__ opart,
                                                      const float* __restrict__ ml,
                                                      float* __restrict__ out, int64_t ldo, int Q,
                                                      int nchunks) {
  __shared__ float wts[8][ATT_MAXCH];
  const int q = blockIdx.x, b = blockIdx.y;
  const int head = threadIdx.x >> 5, d = threadIdx.x & 31;
  const int64_t base = ((int64_t)b * 8 + head) * nchunks;
  float M = -INFINITY;
  for (int c = d; c < nchunks; c += 32) M = fmaxf(M, ml[((base + c) * Q + q) * 2]);
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) M = fmaxf(M, __shfl_xor(M, o, 32));
  float den = 0.f;
  for (int c = d; c < nchunks; c += 32) {
    const int64_t slot = (base + c) * Q + q;
    const float m = ml[slot * 2];
    const float w = (m == -INFINITY) ? 0.f : expf(m - M);
    wts[head][c] = w;
    den += w * ml[slot * 2 + 1];
  }
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) den += __shfl_xor(den, o, 32);
  __syncthreads();
  float n0 = 0.f, n1 = 0.f, n2 = 0.f, n3 = 0.f;
  int c = 0;
  for (; c + 4 <= nchunks; c += 4) {
    n0 += wts[head][c + 0] * opart[((base + c + 0) * Q + q) * 32 + d];
    n1 += wts[head][c + 1] * opart[((base + c + 1) * Q + q) * 32 + d];
    n2 += wts[head][c + 2] * opart[((base + c + 2) * Q + q) * 32 + d];
    n3 += wts[head][c + 3] * opart[((base + c + 3) * Q + q) * 32 + d];
  }
  for (; c < nchunks; ++c) n0 += wts[head][c] * opart[((base + c) * Q + q) * 32 + d];
  out[((int64_t)b * Q + q) * ldo + head * 32 + d] = ((n0 + n1) + (n2 + n3)) / den;
}

static int attn_chunking(int Nk, int B, int Q, int* chunk) {
  // short key sets (self-attention, relation decoder): one chunk, no combine pass;
  // long ones: ~1000 workgroups, chunk a multiple of 32 keys, at most ATT_MAXCH chunks
  const int qgroups = (Q + 127) / 128;
  if (Nk <= 512) {
    *chunk = (Nk + 31) & ~31;
    return 1;
  }
  int want = 1024 / (8 * B * qgroups);
  if (want < 1) want = 1;
  if (want > ATT_MAXCH) want = ATT_MAXCH;
  int ch = ((Nk + want - 1) / want + 31) & ~31;
  if (ch < 64) ch = 64;
  *chunk = ch;
  return (Nk + ch - 1) / ch;
}

extern "C" int64_t pn_attn_scratch_floats(int B, int Q, int Nk) {
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
  if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)scratch) & 15) return PN_BAD_ARG;
  if ((maskbits == nullptr) != (rowall == nullptr)) return PN_BAD_ARG;
  AttnP p{};
  p.q = q; p.k = k; p.v = v; p.bits = maskbits; p.rowall = rowall;
  p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.Q = Q; p.Nk = Nk; p.scale = scale;
  p.nchunks = attn_chunking(Nk, B, Q, &p.chunk);
  p.nwords = (Nk + 31) / 32;
  p.opart = scratch;
  p.ml = scratch + (int64_t)B * 8 * p.nchunks * Q * 32;
  p.out = out;
  p.ldo = ldo;
  hipStream_t s = (hipStream_t)stream;
  const int qgroups = (Q + 127) / 128;
  hipLaunchKernelGGL(k_attn_chunk, dim3(p.nchunks, 8, B * qgroups), dim3(256), 0, s, p);
  if (p.nchunks > 1)
    hipLaunchKernelGGL(k_attn_combine, dim3(Q, B), dim3(256), 0, s, p.opart, p.ml, out, ldo, Q,
                       p.nchunks);
  return PN_LAUNCH_CHECK();
}
