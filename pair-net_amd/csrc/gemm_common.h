// Shared between gemm.hip and gemm_split.hip.
#pragma once
#include "common.h"

struct GemmP {
  const float* A; const float* Aadd; const float* W; const float* bias;
  const float* Res; float* C;
  int64_t lda, ldaadd, ldw, ldres, ldc;
  int64_t sA, sW, sRes, sC;
  int M, N, K, aadd_rows, aadd_from_col, relu /* act: 0 none, 1 ReLU, 2 GELU (erf) */, a_vec;
  int relu_after;           // ReLU after the residual add (ResNet bottleneck output)
  // split-K: the launch's batch index bz = (batch b) * ksplit + (split s); split s
  // contracts the 32-deep chunks [s * split_chunks, (s+1) * split_chunks) into its own
  // partial C (k_splitk_reduce sums the partials in split order and applies the epilogue)
  int ksplit, split_chunks;
  // conv mode: input H x Wd x Cin channel-last, output rows M = Ho * Wo
  int H, Wd, Cin, KW, pad, stride, Wo;
};

// stencil-mask mode (k_gemm_stencil): W = the 4 * nk tap-major stencil rows of one level; the
// epilogue blends each key's four logits like pn_bilinear_planar_f32, thresholds and packs bits
// instead of storing C.  (Its own struct: GemmP x 18 must fit k_gemm_group's kernarg block.)
struct StencilP {
  uint32_t* bits; int32_t* rowall;
  int nk, hi, wi, ho, wo;
};

// A operand: row-major matrix | column-major matrix | implicit im2col of a channel-last
// image | implicit im2col of the NCHW RGB image for ResNet's 7x7/2 stem (K = 147 -> 160)
enum { A_ROW = 0, A_COL = 1, A_CONV = 2 };


static inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

int pn_fill_params(const pn_gemm_desc* d, GemmP* out);

// act of the epilogue: the code is wave-uniform (a kernel argument)
__device__ __forceinline__ float gemm_act(float v, int act) {
  if (act == 1) return fmaxf(v, 0.f);
  if (act == 2) return 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
  return v;
}
