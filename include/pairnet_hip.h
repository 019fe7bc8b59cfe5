/*
 * pairnet_hip.h -- C ABI of libpairnet_hip.so: hand-written gfx950 (MI355X) kernels
 * for the Pair-Net inference hot path.
 *
 * Conventions (SURVEY.md 8b "operator-level boundary"):
 *   - every entry point is extern "C", returns a hipError_t value as int (0 = ok;
 *     -1 = argument contract violated, nothing launched);
 *   - plain device pointers + sizes, no torch types; `stream` is a hipStream_t;
 *   - no allocation, no global state, caller-owned buffers, re-entrant per stream,
 *     asynchronous (the caller synchronises), graph-capturable;
 *   - all floating-point data is fp32 (the reference runs fp32, no AMP:
 *     configs/mask2former/pairnet.py has no fp16 key); indices are int64, as torch
 *     produces them; token / pixel tensors are channel-last ("token-major").
 *
 * The reference has exactly one native operator boundary on this path, the
 * third-party mmcv extension `ms_deform_attn_forward` (reached from
 * configs/mask2former/pairnet.py:43-54 via pairnet_head.py:94,262); every other
 * native kernel it runs is an ATen/cuDNN/cuBLAS kernel behind a torch call.  Each
 * entry below names the reference call site(s) (file:line under /root/reference)
 * whose arithmetic it replaces.  INTEGRATION.md shows the reference-side binding.
 */
#ifndef PAIRNET_HIP_H_
#define PAIRNET_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PN_ABI_VERSION 28
int pn_abi_version(void);

/* ------------------------------------------------------------------------- *
 * Dense contraction (f32 MFMA 32x32x2, exact fp32 accumulate)
 *   C[z][m][n] = act( sum_k (A[z][m][k] + Aadd[m % aadd_rows][k]) * W[z][n][k]
 *                     + bias[n] ) + Res[z][m][n]
 * Replaces every nn.Linear / 1x1 Conv2d / einsum / matmul on the path:
 *   pairnet_head.py:240-243 (cls_embed, mask_embed, einsum "bqc,bchw->bqhw"),
 *   :323-327 (sub/obj MLPs, importance matmul), :378 (rel_cls_embed), the packed
 *   in_proj / out_proj of nn.MultiheadAttention and the FFNs behind :302-312 and
 *   :367-376, and the pixel decoder's value_proj / sampling_offsets /
 *   attention_weights / output_proj / FFN / 1x1 convs behind :262.
 * `Aadd` fuses the positional add the reference does before a projection
 * (query + query_pos, key + key_pos: facebook_detr.py:329-332).
 * ------------------------------------------------------------------------- */
#define PN_GEMM_RELU       1   /* act = ReLU (else identity)                        */
#define PN_GEMM_A_COLMAJOR 2   /* A is stored [K][lda] (an NCHW feature map)        */
#define PN_GEMM_FORCE_TILE 4   /* testing: force the 128x128 tile of the LDS kernel */
#define PN_GEMM_FORCE_SKINNY 8 /* testing: force the 32x32 split-K-in-block kernel  */
#define PN_GEMM_FORCE_TILE64 16     /* tuning: 64x64 tile (row-major A)             */
#define PN_GEMM_FORCE_TILE128x64 32 /* tuning: 128x64 tile                          */
#define PN_GEMM_RELU_AFTER_RES 128  /* ReLU after the residual add: relu(act(..)+Res)  */
#define PN_GEMM_GELU 256           /* act = exact (erf) GELU; Swin FFN                  */
/* Scheduling hint carried PER CALL in `flags` (performance only, results unchanged): leave
 * `n` (a multiple of 8, < 8192) of the persistent tile kernels' resident workgroup slots
 * unoccupied, so that the small latency-bound kernels of a concurrent stream (the query
 * chains) find a free slot without waiting for a kernel boundary.  There is no
 * process-wide knob: two callers in one process cannot change each other's grids. */
#define PN_GEMM_RESERVE_SHIFT 16
#define PN_GEMM_RESERVE(n) ((((n) / 8) & 0x3ff) << PN_GEMM_RESERVE_SHIFT)
/* Tuning: force the number of K slices of the split-K path (1 = never split; 0 = the
 * library's own choice; needs splitk_scratch). */
#define PN_GEMM_KSPLIT_SHIFT 26
#define PN_GEMM_KSPLIT(n) (((n) & 31) << PN_GEMM_KSPLIT_SHIFT)

typedef struct pn_gemm_desc {
  const float* A;     int64_t lda;    int64_t strideA;    /* [M][K] (or [K][M])    */
  const float* Aadd;  int64_t ldaadd; int32_t aadd_rows;  /* optional, may be NULL */
  int32_t aadd_from_col;  /* Aadd only feeds output columns >= this (multiple of 64;
                             0 = all): one GEMM for [value_proj | offsets+logits]       */
  const float* W;     int64_t ldw;    int64_t strideW;    /* [N][K]                */
  const float* bias;                                      /* [N] or NULL           */
  const float* Res;   int64_t ldres;  int64_t strideRes;  /* optional residual     */
  float*       C;     int64_t ldc;    int64_t strideC;    /* [M][N]                */
  int32_t M, N, K, batch, flags;
  /* optional split-K workspace (NULL = never split): problems whose M x N gives too few
   * 64x64 tiles to fill 256 CUs (the backbone's late stages, the C5/C4 input convs)
   * contract K in up to 16 slices into this scratch and sum them in slice order */
  float* splitk_scratch;  int64_t splitk_scratch_floats;
} pn_gemm_desc;

int pn_gemm_f32(const pn_gemm_desc* d, void* stream);

/* `count` (<= 18) independent row-major problems in ONE launch of the persistent 64x64
 * tile kernel (all their tiles share the grid): used for the 18 key/value projections of
 * the 9 decoder layers, whose per-problem tile counts do not fill 256 CUs evenly. */
int pn_gemm_group_f32(const pn_gemm_desc* d, int count, void* stream);

/* Resident workgroups per CU the persistent 64x64-tile kernel (plain row-major A) is sized
 * for: pn_gemm_grid_size(d) = min(tiles, CUs * this - reserved slots) rounded to 8. */
int pn_gemm_wgs_per_cu(void);

/* Which kernel pn_gemm_f32 would launch for `d` (for profiling / roofline
 * attribution; +1 = the column-major-A instantiation). */
#define PN_GEMM_VARIANT_SKINNY        0  /* k_gemm_skinny<A>                 */
#define PN_GEMM_VARIANT_TILE_128x64   2  /* k_gemm_tile<128,64,64,32,A>      */
#define PN_GEMM_VARIANT_TILE_128x128  4  /* k_gemm_tile<128,128,64,64,A>     */
#define PN_GEMM_VARIANT_TILE_64x64    6  /* k_gemm_tile<64,64,32,32,A> (default) */
int pn_gemm_variant(const pn_gemm_desc* d);
/* Workgroups the persistent 64x64 tile kernel is launched with for `d` (introspection: a
 * function of the descriptor alone, PN_GEMM_RESERVE included). */
int pn_gemm_grid_size(const pn_gemm_desc* d);

/* Implicit-GEMM KHxKW convolution, stride 1, zero padding, channel-last:
 *   out[b][y][x][co] = act(sum_{ky,kx,ci} in[b][y+ky-pad][x+kx-pad][ci]
 *                          * Wp[co][(ky*KW+kx)*Cin+ci] + bias[co])
 * Replaces the pixel decoder's 3x3 output conv (behind pairnet_head.py:262) and
 * the 64->64 7x7 layer of the Matrix Learner (cnn_factory.py:31-41).
 * Cin % 32 == 0. */
int pn_conv2d_nhwc_f32(const float* in, const float* Wp, const float* bias,
                       float* out, int B, int H, int W, int Cin, int Cout,
                       int KH, int KW, int pad, int relu, int flags /* 0, PN_GEMM_FORCE_TILE*,
                       PN_GEMM_RESERVE(n) */, void* stream);

/* Winograd F(2x2, 3x3) form of the same 3x3 "same" convolution (C % 4 == 0; odd H / W: the
 * last tile row / column is padded with zeros and clipped), T = B*ceil(H/2)*ceil(W/2) tiles:
 *   pn_winograd_f23_input_f32:  V [16][T][Cin]  = B^T d B of every 4x4 patch
 *   16 GEMMs (one batched pn_gemm_f32 call, batch = 16):  M_xi = V_xi . U_xi^T with
 *       U [16][Cout][Cin] = G g G^T of the weights (computed once by the caller)
 *   pn_winograd_f23_output_f32: out [B][H][W][Cout] = act(A^T M A + bias)
 * 2.25x fewer multiplications than the direct form; fp32, differs from it by fp32
 * re-association only. */
int pn_winograd_f23_input_f32(const float* in, float* V, int B, int H, int W, int C,
                              void* stream);
int pn_winograd_f23_output_f32(const float* M, const float* bias, float* out, int B, int H,
                               int W, int C, int relu, void* stream);
/* Weight transform U = G g G^T of a 3x3 convolution weight w [Co][Ci][3][3] on the device (double
 * arithmetic, rounded once): form 2 -> F(2x2,3x3), U [16][Co][Ci]; form 4 -> F(4x4,3x3), U [36][Co][Ci]
 * (what pn_conv3x3_winograd* take).  Used at pack time and after every optimizer step of a trained
 * backbone. */
int pn_winograd_weights_f32(const float* w, float* U, int Co, int Ci, int form, void* stream);
/* F(4x4, 3x3): 36 positions (V, M: [36][B*ceil(H/4)*ceil(W/4)][C], U [36][Cout][Cin]), 4x fewer
 * multiplications than the direct form; any H, W (edge tiles are padded / clipped).  Its
 * transform constants (up to 8 and 1/24) cost about one more decimal digit than F(2x2). */
int pn_winograd_f43_input_f32(const float* in, float* V, int B, int H, int W, int C,
                              void* stream);
int pn_winograd_f43_output_f32(const float* M, const float* bias, float* out, int B, int H,
                               int W, int C, int relu, void* stream);

/* ------------------------------------------------------------------------- *
 * Backbone: ResNet-50, style "pytorch", frozen BatchNorm
 * (configs/mask2former/pairnet.py:9-19; mmdet's ResNet is third-party, restated in
 * oracle/backbone.py).  BatchNorm is folded into the convolution weights / bias by the
 * caller; activations are channel-last.
 * ------------------------------------------------------------------------- */
/* General form of the convolution above: stride >= 1, any padding,
 *   Ho = (H + 2 pad - KH) / stride + 1 (same for W),
 *   out = [relu_after](act(conv + bias) + res),  res/out [B][Ho][Wo][Cout],
 * flags: PN_GEMM_RELU (act), PN_GEMM_RELU_AFTER_RES, tile selectors, PN_GEMM_RESERVE(n). */
int pn_conv2d_nhwc_ex_f32(const float* in, const float* Wp, const float* bias,
                          const float* res, float* out, int B, int H, int W, int Cin,
                          int Cout, int KH, int KW, int stride, int pad, int flags,
                          float* splitk_scratch /* or NULL */,
                          int64_t splitk_scratch_floats, void* stream);
/* Stem: relu(conv7x7/2 pad 3 (NCHW RGB image) + bias) -> [B][Ho][Wo][64] channel-last.
 * Wp [64][160] = conv1.weight [64][3][7][7] flattened, zero-padded 147 -> 160. */
int pn_stem7x7s2_f32(const float* img_nchw, const float* Wp, const float* bias,
                     float* out, int B, int H, int W, int flags /* 0 or PN_GEMM_RESERVE(n) */,
                     void* stream);
/* F.max_pool2d(x, 3, stride=2, padding=1) on channel-last data, C % 4 == 0. */
int pn_maxpool3x3s2_nhwc_f32(const float* in, float* out, int B, int H, int W, int C,
                             void* stream);

/* ------------------------------------------------------------------------- *
 * Normalisation
 * ------------------------------------------------------------------------- */
/* y[r][:] = LayerNorm(x[r][:]) * gamma + beta, C == 256, eps 1e-5.
 * (norms of BaseTransformerLayer, facebook_detr.py:406-408; post_norm,
 * pairnet_head.py:236.) */
int pn_layernorm_f32(const float* x, const float* gamma, const float* beta,
                     float* y, int64_t rows, int C, float eps, void* stream);

/* The same over rows of any width C % 4 == 0, C <= 3072, with row strides (floats):
 * the norms of the Swin backbone ([3P] mmdet SwinTransformer configured at
 * configs/mask2former/pairnet_swinb.py:203-226: patch_embed.norm, norm1 / norm2 of every
 * block, the output norm of every stage). */
int pn_layernorm_rows_f32(const float* x, int64_t ldx, const float* gamma, const float* beta,
                          float* y, int64_t ldy, int64_t rows, int C, float eps, void* stream);
/* The same LayerNorm written as an S3 operand [rows x C] (three bf16 planes, pn_gemm_s3_f32's A:
 * the Swin blocks' norm1 / norm2 in front of the qkv / FFN GEMMs); C % 16 == 0.  Rows of the last
 * 32-row block beyond `rows` are not written (the GEMM never stores what it computes from them). */
int pn_layernorm_rows_s3_f32(const float* x, int64_t ldx, const float* gamma, const float* beta,
                             void* y_s3, int64_t rows, int C, float eps, void* stream);

/* Swin patch merging up to its Linear: x [B][H*W][C] -> y [B][H2*W2][4C], H2 = ceil(H/2),
 * row (b, y2, x2) = LayerNorm_4C([x(2y2,2x2) | x(2y2,2x2+1) | x(2y2+1,2x2) | x(2y2+1,2x2+1)])
 * with zeros beyond an odd map's edge.  gamma / beta (and the columns of the reduction
 * weight that follows) are in this neighbour-major order: index (row*2+col)*C + c, the
 * permutation of mmdet's nn.Unfold order c*4 + row*2 + col. */
int pn_patch_merge_ln_f32(const float* x, const float* gamma, const float* beta, float* y,
                          int B, int H, int W, int C, float eps, void* stream);

/* Swin patch embedding, im2col half: NCHW RGB image [B][3][H][W] -> rows
 * [B*ceil(H/4)*ceil(W/4)][64], column c*16 + ky*4 + kx (the flattening of the
 * [C][3][4][4] projection weight), columns 48..63 zero, pixels beyond H / W zero (the
 * reference pads bottom / right to a multiple of the patch). */
int pn_patch_im2col4_f32(const float* img, float* out, int B, int H, int W, void* stream);

/* Swin (shifted-)window multi-head attention, head dim 32, on the qkv rows
 * [B*H*W][ldqkv] (q | k | v, each C = heads*32 wide) -> out [B*H*W][ldo]:
 * zero-padding of the map to a multiple of `ws` (padded tokens' q/k/v = qkv_bias, as the
 * reference pads after norm1), cyclic shift by `shift`, window partition, softmax(q k^T *
 * scale + bias_table[head][(dy+ws-1)(2ws-1)+(dx+ws-1)] + (-100 between different
 * wrap-around regions)) v, window merge, un-shift and crop are all index arithmetic.
 * bias_table is [heads][(2ws-1)^2]: the TRANSPOSE of mmdet's
 * relative_position_bias_table parameter.  ws*ws <= 169. */
int pn_window_attention_f32(const float* qkv, int64_t ldqkv, const float* qkv_bias,
                            const float* bias_table, float* out, int64_t ldo, int B, int H,
                            int W, int C, int heads, int ws, int shift, float scale,
                            void* stream);
/* The same with the output written as an S3 operand [B H W x C] (three bf16 planes: the proj GEMM's A
 * operand of pn_gemm_s3_f32, pre-split; bit for bit the split of the fp32 form's output). */
int pn_window_attention_s3_f32(const float* qkv, int64_t ldqkv, const float* qkv_bias,
                               const float* bias_table, void* out_s3, int B, int H, int W, int C,
                               int heads, int ws, int shift, float scale, void* stream);

/* GroupNorm over channel-last x[b][HW][C] with G groups (+ optional ReLU); image b
 * starts at x + b*x_bstride / y + b*y_bstride (floats).  `partials` is caller
 * scratch of B * nblk * G * 2 doubles, nblk = pn_groupnorm_nblk(HW).  (ConvModule norm of the pixel decoder, behind
 * pairnet_head.py:262.) */
int pn_groupnorm_nblk(int64_t HW);
int pn_groupnorm_nhwc_f32(const float* x, const float* gamma, const float* beta,
                          float* y, double* partials, int B, int64_t HW, int C,
                          int G, float eps, int relu, int64_t x_bstride,
                          int64_t y_bstride, void* stream);

/* Linear + residual + post-norm of a transformer layer in one launch, N == 256:
 *   y[r][:] = LayerNorm(res[r][:] + x[r][:] W^T + bias) * gamma + beta
 * -- `output_proj` + identity + norms.0, and ffns.0.layers.1 + identity + norms.1, of the
 * pixel decoder's six encoder layers (MultiScaleDeformableAttention.forward /
 * BaseTransformerLayer, [3P] mmcv 1.7.0; configured at configs/mask2former/pairnet.py:40-66,
 * run behind pairnet_head.py:262).  A workgroup owns 32 whole rows, so the pre-norm map never
 * reaches HBM.  Bit for bit pn_gemm_f32 (row-major, residual epilogue) followed by
 * pn_layernorm_f32.  x [M][ldx] (K used), W [256][ldw], res [M][ldres], y [M][ldy]; K % 32 == 0;
 * 16-byte aligned operands, leading dimensions % 4 == 0; y must not alias x.  Meant for
 * chip-filling M (>= 2048 rows: M / 32 workgroups); the query-side chains keep their own
 * kernels. */
int pn_linear_res_ln_f32(const float* x, int64_t ldx, const float* W, int64_t ldw,
                         const float* bias /* nullable */, const float* res, int64_t ldres,
                         const float* gamma, const float* beta, float* y, int64_t ldy,
                         int M, int N, int K, float eps, void* stream);

/* Fused FFN block of a decoder layer for M ~ 100 rows (facebook_detr.py:425-427 +
 * the norm that follows):  y = LayerNorm(x + W2 relu(W1 x + b1) + b2) * gamma + beta.
 * W1 [hidden][256], W2 [256][hidden]; hidden % 64 == 0; scratch =
 * pn_ffn_scratch_floats(M, hidden) floats.  y may alias nothing it reads except x. */
int64_t pn_ffn_scratch_floats(int M, int hidden);
int pn_ffn_ln_f32(const float* x, const float* W1, const float* b1, const float* W2,
                  const float* b2, const float* gamma, const float* beta, float* y,
                  float* scratch, int M, int C, int hidden, float eps, void* stream);
/* The same with a SECOND LayerNorm of the result written to y2 (nullable): the decoder's
 * post_norm that `forward_head` applies to every layer's output (pairnet_head.py:236), bit
 * for bit the arithmetic of pn_layernorm_f32 on y. */
int pn_ffn_ln2_f32(const float* x, const float* W1, const float* b1, const float* W2,
                   const float* b2, const float* gamma, const float* beta, float* y,
                   const float* gamma2, const float* beta2, float* y2, float* scratch, int M,
                   int C, int hidden, float eps, void* stream);

/* y[r][:] = x[r][:] / max(||x[r]||_2, eps)   (F.normalize, pairnet_head.py:325-326) */
int pn_l2normalize_f32(const float* x, float* y, int64_t rows, int C, float eps,
                       void* stream);

/* ------------------------------------------------------------------------- *
 * Multi-scale deformable attention sampling.  THE reference's native boundary:
 * replaces mmcv `ext_module.ms_deform_attn_forward(value, spatial_shapes,
 * level_start_index, sampling_locations, attention_weights, im2col_step)` and
 * fuses what precedes it in MultiScaleDeformableAttention.forward: softmax over
 * the L*P logits and reference_point + offset / (W_l, H_l).
 *   value   [B][N][ld_value]  projected value tokens (first H*D floats of each
 *                             row), levels concatenated
 *   offaw   [B][N][ld_offaw]  raw Linear outputs: H*L*P*2 offsets (x,y) followed
 *                             by H*L*P attention logits
 *   (row strides let both live in one [value | offsets | logits] GEMM output)
 *   out     [B][N][H*D]
 * Queries are the N tokens themselves (encoder self-attention); the reference
 * point of token n is its own pixel centre ((x+.5)/w, (y+.5)/h).
 * H == 8, D == 32, P == 4, L <= 4. */
int pn_msda_f32(const float* value, int64_t ld_value, const float* offaw,
                int64_t ld_offaw, float* out, int B, int L,
                const int32_t* level_h /* host */, const int32_t* level_w /* host */,
                void* stream);
/* The same with a tuning flag: 0 = one workgroup per query pair (what pn_msda_f32 launches);
 * PN_MSDA_PERSISTENT / PN_MSDA_PERSISTENT_BATCHED = persistent workgroups walking their XCD
 * band's query pairs with the next pair's offsets / logits in flight (one / both queries of a
 * pair gathering at a time): measured slower in round 4 (csrc/msda.hip), kept as the A/B.
 * All three give bit-identical output. */
#define PN_MSDA_PERSISTENT 1
#define PN_MSDA_PERSISTENT_BATCHED 2
#define PN_MSDA_LOW_OCCUPANCY 4    /* one-shot form with the compiler's register choice (84 VGPRs, 5
                                      workgroups per CU: rounds 1-3); default: 62 VGPRs, 8 per CU */
#define PN_MSDA_S3_OUT 8           /* `out` is an S3 operand [B * N x 256] (three bf16 planes, see
                                    * pn_gemm_s3_f32): the output_proj GEMM's A operand written pre-split,
                                    * the fp32 map is not written; default (one-shot) form only */
int pn_msda_ex_f32(const float* value, int64_t ld_value, const float* offaw,
                   int64_t ld_offaw, float* out, int B, int L,
                   const int32_t* level_h /* host */, const int32_t* level_w /* host */,
                   int flags, void* stream);

/* The same operator in mmcv's OWN shape -- the entry a maintainer binds behind the
 * unmodified `MultiScaleDeformableAttention.forward` (mmcv/ops/multi_scale_deform_attn.py,
 * `ext_module.ms_deform_attn_forward`, configured at configs/mask2former/pairnet.py:43-54;
 * INTEGRATION.md shows the binding), and the query-side cross-attention of the
 * Deformable-DETR trunk under CrossHeadBBox (pairnet_bbox_head.py:193-359):
 *   value               [B][N][ld_value]      first 8*32 floats of each row
 *   spatial_shapes      [L][2] int64, DEVICE  (H_l, W_l), as mmcv passes them
 *   level_start_index   [L]    int64, DEVICE
 *   sampling_locations  [B][Nq][8][L][4][2]   normalised (x, y) in [0, 1]
 *   attention_weights   [B][Nq][8][L][4]      already soft-maxed
 *   out                 [B][Nq][256]
 * Any number of queries Nq, bilinear sampling with zero padding (grid_sample,
 * align_corners=False).  H == 8, D == 32, P == 4, L <= 4.  (im2col_step is a batching
 * knob of mmcv's CUDA kernel without an arithmetic effect: there is none here.) */
int pn_msda_loc_f32(const float* value, int64_t ld_value, const int64_t* spatial_shapes,
                    const int64_t* level_start_index, const float* sampling_locations,
                    const float* attention_weights, float* out, int B, int N, int Nq, int L,
                    void* stream);

/* Backward of pn_msda_loc_f32, in the shape of mmcv's `ext_module.ms_deform_attn_backward`
 * (mmcv/ops/multi_scale_deform_attn.py, `MultiScaleDeformableAttnFunction.backward`; the second
 * half of the reference's one native boundary -- what the training step of
 * pairnet_head.py:420-757 / tools/train.py:115-241 differentiates through):
 *   grad_output         [B][Nq][256]
 *   grad_value          [B][N][ld_value]   ACCUMULATED into (hardware fp32 atomics): the caller
 *                                          zeroes it first, as mmcv does (zeros_like(value))
 *   grad_sampling_loc   [B][Nq][8][L][4][2]  written
 *   grad_attn_weight    [B][Nq][8][L][4]     written
 * Arithmetic of mmcv's ms_deform_attn_col2im_bilinear (locations outside (-1, H) x (-1, W)
 * contribute nothing); grad_value's summation order is not deterministic (nor is mmcv's), the
 * other two outputs are. */
int pn_msda_bwd_f32(const float* value, int64_t ld_value, const int64_t* spatial_shapes,
                    const int64_t* level_start_index, const float* sampling_locations,
                    const float* attention_weights, const float* grad_output, float* grad_value,
                    float* grad_sampling_loc, float* grad_attn_weight, int B, int N, int Nq, int L,
                    void* stream);

/* Diagnostic (bench.py's `roofline_deformable_sampling.gather_peak`): the bare access pattern of
 * the sampling kernels above -- 8 lanes x 16 B per random 128-byte line, 12 independent lines per
 * lane group, no arithmetic but a sum -- over `workgroups` x 256 threads, lines drawn from the
 * first line_mask + 1 (a power of two) lines of `lines`; idx [workgroups][32][12] int32 (any
 * values: masked), out [workgroups][256].  Replaces nothing in the reference: it measures the
 * roof the deformable-attention gather (mmcv ms_deform_attn_forward) sits under here. */
int pn_gather_probe_f32(const float* lines, const int32_t* idx, float* out, int workgroups,
                        int line_mask, void* stream);

/* ------------------------------------------------------------------------- *
 * Positional encoding / resampling
 * ------------------------------------------------------------------------- */
/* out[(y*w+x)][c] = SinePositionalEncoding(num_feats=C/2, normalize=True,
 * temperature, scale=2pi, eps=1e-6)(y,x)[c] + add[c]  (add may be NULL).
 * (pairnet_head.py:278; level_encoding add in the pixel decoder.) */
int pn_sine_pe_f32(float* out, const float* add, int h, int w, int C,
                   float temperature, void* stream);
/* The same with SinePositionalEncoding's `offset` (added to the 1-based row / column index
 * before normalisation; configs/deformable_detr/cross_r101_vg.py:118-120 uses -0.5). */
int pn_sine_pe_offset_f32(float* out, const float* add, int h, int w, int C, float temperature,
                          float offset, void* stream);
/* The same on a padded map: rows >= valid_h / columns >= valid_w are padding
 * (SinePositionalEncoding on a mask: the cumulative sums stop growing there, and the
 * normalisation divides by the valid size). */
int pn_sine_pe_valid_f32(float* out, const float* add, int h, int w, int valid_h, int valid_w,
                         int C, float temperature, float offset, void* stream);

/* Bilinear resize, align_corners=False (F.interpolate semantics).
 * nhwc:   in [b][hi][wi][C] -> out [b][ho][wo][C], out = (accumulate? out:0)+v;
 *         image b at in + b*in_bstride / out + b*out_bstride (floats)
 * planar: in [P][hi][wi]    -> out [P][ho][wo]
 * (FPN top-down add behind pairnet_head.py:262; attention-mask resize :244-246;
 * mask upsampling in _get_bboxes_single :826-843.) */
int pn_bilinear_nhwc_f32(const float* in, float* out, int B, int hi, int wi,
                         int ho, int wo, int C, int accumulate, int64_t in_bstride,
                         int64_t out_bstride, void* stream);
int pn_bilinear_planar_f32(const float* in, float* out, int64_t P, int hi,
                           int wi, int ho, int wo, void* stream);
/* The rows a bilinear resize of a channel-last map reads: out[b][t][p][:] = in[b][tap t of
 * output pixel p][:], t = 0..3 <-> (y0,x0), (y0,x1), (y1,x0), (y1,x1).  As the W operand of the
 * mask-logit GEMM they yield exactly the full-resolution logits `F.interpolate(mask_pred)`
 * reads for a level (pairnet_head.py:244-246) -- Q x 4 N_l instead of Q x H2 W2 per layer;
 * pn_mask_pack_stencil blends, thresholds and packs them. */
int pn_bilinear_stencil_rows_f32(const float* in, float* out, int B, int hi, int wi, int ho,
                                 int wo, int C, int64_t in_bstride, int64_t out_bstride,
                                 void* stream);
/* planar resize + (sigmoid(v) > 0.5  <=>  v > 0) -> uint8 {0,1}  (:834,:842) */
int pn_bilinear_planar_gt0_u8(const float* in, uint8_t* out, int64_t P, int hi,
                              int wi, int ho, int wo, void* stream);

/* ------------------------------------------------------------------------- *
 * Attention
 * ------------------------------------------------------------------------- */
/* Bit-pack the boolean attention mask and apply the reference's all-masked fix:
 *   bits[r][w] bit j = (logits[r][32w+j] < 0)   (== sigmoid < 0.5, :256)
 *   rowall[r] = 1 if every one of the Nk keys of row r is masked (:300)
 * logits [R][Nk]; bits [R][nwords], nwords = (Nk+31)/32. */
int pn_mask_pack(const float* logits, uint32_t* bits, int32_t* rowall,
                 int64_t R, int Nk, void* stream);
/* The same from logits4 [R][4][Nk = ho*wo], the four stencil logits per key (tap-major, see
 * pn_bilinear_stencil_rows_f32): bit = (bilinear blend of the four, as pn_bilinear_planar_f32
 * computes it from the (hi, wi) map) < 0.  ho, wo <= 1024. */
int pn_mask_pack_stencil(const float* logits4, uint32_t* bits, int32_t* rowall, int64_t R,
                         int hi, int wi, int ho, int wo, void* stream);

/* The two steps above -- a layer's stencil logits and their blend / threshold / pack -- as ONE
 * launch (round 5): logits = me . rows^T over the level's 4 Nk tap-major stencil rows
 * (pn_bilinear_stencil_rows_f32) on the 64x64 tile loop of pn_gemm_f32 (the same products in
 * the same order: every logit is the bit pattern pn_gemm_f32 would have stored), blended with
 * the single tap_blend definition, thresholded (`< 0`) and bit-packed in the epilogue; the
 * Q x 4 Nk logit map is never written.  Output exactly pn_mask_pack_stencil's: bits
 * [B*Q][ceil(Nk/32)] (padding bits 0), rowall [B*Q] = 1 where every key of the row is masked
 * (set to 1 by a fill launch in front, cleared by any tile that finds an unmasked key).
 *   me [B][Q][ld_me] (K used), rows [B][4*Nk][ld_rows]; (hi, wi) the full-resolution mask map,
 *   (ho, wo) the level's map, Nk == ho * wo; K % 32 == 0.  `flags`: PN_GEMM_RESERVE(n) only.
 * Replaces pairnet_head.py:244-256 (interpolate -> sigmoid < 0.5) + :300 for one layer. */
int pn_mask_stencil_gemm_f32(const float* me, int64_t ld_me, int64_t stride_me, const float* rows,
                             int64_t ld_rows, int64_t stride_rows, uint32_t* bits,
                             int32_t* rowall, int B, int Q, int Nk, int K, int hi, int wi, int ho,
                             int wo, int flags, void* stream);

/* softmax(q k^T * scale + mask) v per head, flash-style over key chunks
 * (f32 MFMA for both contractions), then a combine pass.
 *   q [B][Q][ldq], k [B][Nk][ldk], v [B][Nk][ldv]: projected, head h at columns
 *   [32h, 32h+32); out [B][Q][ldo]; 8 heads x 32.
 *   maskbits/rowall from pn_mask_pack (NULL = no mask), shared by the 8 heads.
 *   scratch: pn_attn_scratch_floats(B, Q, Nk) floats.
 * (nn.MultiheadAttention core behind pairnet_head.py:302-312, :367-376.) */
int64_t pn_attn_scratch_floats(int B, int Q, int Nk);
int pn_attention_f32(const float* q, int64_t ldq, const float* k, int64_t ldk,
                     const float* v, int64_t ldv, const uint32_t* maskbits,
                     const int32_t* rowall, float* out, int64_t ldo,
                     float* scratch, int B, int Q, int Nk, float scale,
                     void* stream);

/* ------------------------------------------------------------------------- *
 * Pair Proposal Network
 * ------------------------------------------------------------------------- */
/* Matrix Learner edge layers (cnn_factory.py:22-29, 42-48):
 *   first: in [B][S][S]     -> out [B][S][S][C] = relu(conv7x7(1->C) + b)
 *          w1 [C][49]
 *   last:  in [B][S][S][C]  -> out [B][S][S]    = conv7x7(C->1) + b
 *          w3 [49][C]
 * The C->C middle layer is pn_conv2d_nhwc_f32. */
int pn_mlearner_first_f32(const float* in, const float* w1, const float* b1,
                          float* out, int B, int S, int C, void* stream);
/* Fused front of the PPN (pairnet_head.py:325-333, cnn_factory.py:22-29): per 8 x 8 tile of
 * (subject, object) pairs the L2-normalised query rows its halo needs are staged in LDS,
 * their cosine block is one MFMA tile, and the first Matrix Learner layer is applied to it
 * on chip.  sub_embed / obj_embed [B][Q][256] are the MLP outputs BEFORE F.normalize;
 * importance_raw [B][Q][Q] (the cosine matrix) and c1 [B][Q][Q][64] (first-layer output,
 * ReLU) are written; w1 [64][49], b1 [64].  Equivalent to pn_l2normalize_f32 x 2 +
 * the batched pn_gemm_f32 + pn_mlearner_first_f32. */
int pn_ppn_front_f32(const float* sub_embed, const float* obj_embed, const float* w1,
                     const float* b1, float* importance_raw, float* c1, int B, int Q, float eps,
                     void* stream);
int pn_mlearner_last_f32(const float* in, const float* w3, const float* b3,
                         float* out, int B, int S, int C, void* stream);

/* Top-k pair selection (pairnet_head.py:334-340): for each image the k largest of
 * the n = Q*Q scores, sorted descending; ties broken by the smaller flat index
 * (torch leaves tie order unspecified).  idx/sub/obj [B][k] int64:
 * sub = idx / Q (trunc), obj = idx % Q; pair (nullable) [B][2k] = [sub | obj], the row
 * list of the pair-feature gather (:342-351).  n <= 65536, k <= 256. */
int pn_topk_pairs(const float* scores, int64_t* idx, int64_t* sub, int64_t* obj,
                  int64_t* pair, int B, int Q, int k, void* stream);
/* General form: k largest of n scores per row; quot = idx / div, rem = idx % div
 * (triplet ranking of the sibling head, relation_heads/baseline.py:1033-1037). */
int pn_topk_f32(const float* scores, int64_t* idx, int64_t* quot, int64_t* rem, int B,
                int n, int div, int k, void* stream);
/* Strided form, k <= 512: score i of row b is scores[b*row_stride + i*elem_stride] (floats).
 * (The two-stage proposal selection of the Deformable-DETR trunk under CrossHeadBBox,
 * `torch.topk(enc_outputs_class[..., 0], 300, dim=1)`: mmdet DeformableDetrTransformer,
 * called at pairnet_bbox_head.py:215-228.) */
int pn_topk_strided_f32(const float* scores, int64_t elem_stride, int64_t row_stride,
                        int64_t* idx, int64_t* quot, int64_t* rem, int B, int n, int div, int k,
                        void* stream);

/* out[b][r][:] = in[b][index[b][r]][:], rows of `len` floats
 * (torch.gather at pairnet_head.py:342-351, 380-403). */
int pn_gather_rows_f32(const float* in, const int64_t* index, float* out, int B,
                       int rows_in, int rows_out, int64_t len, void* stream);

/* ------------------------------------------------------------------------- *
 * Post-processing (pairnet_head.py:788-924)
 * ------------------------------------------------------------------------- */
/* softmax over C logits, drop the last (background) column, max/argmax
 * (:811-815, :823-825).  label = argmax + label_offset (the triplet labels are
 * 1-based, `+ 1` at :812 / :815; the panoptic branch :823-825 is 0-based),
 * score = max prob. */
int pn_cls_argmax_f32(const float* logits, int64_t* label, float* score,
                      int64_t rows, int C, int label_offset, void* stream);
/* r_dists[r][0] = 0, r_dists[r][1:] = softmax(logits[r][:])  (:817-820) */
int pn_rel_dists_f32(const float* logits, float* out, int64_t rows, int C,
                     void* stream);
/* CrossHeadBaseline triplet ranking (pairnet/models/relation_heads/baseline.py).
 * probs [rows][C] = softmax(logits); fg [rows][C-1] = probs[:, 1:]  (:1033-1034) */
int pn_softmax_fg_f32(const float* logits, float* probs, float* fg, int64_t rows,
                      int C, void* stream);
/* idx[r] = first index of max(x[r][:])  (torch.max(-1)[1], :398-399) */
int pn_row_argmax_f32(const float* x, int64_t* idx, int64_t rows, int n,
                      void* stream);
/* labels [2k] = [s_label[tri]+1 | o_label[tri]+1]; r_labels = rem+1;
 * r_scores = probs[tri][rem+1]; r_dists [k][C] = probs[tri]  (:1035-1046) */
int pn_triplet_finish(const int64_t* s_label, const int64_t* o_label,
                      const float* probs, const int64_t* tri, const int64_t* rem,
                      int64_t* labels, int64_t* r_labels, float* r_scores,
                      float* r_dists, int k, int C, void* stream);
/* Panoptic id map (:866-871): masks [n][HW] fp32 logits ->
 * m_id[p] = argmax_i softmax_i(masks[:, p]); remap[i] merges stuff duplicates
 * (:873-878); seg[p] = id*1000 + labels[id]; area[i] = #pixels with id i. */
int pn_panoptic_f32(const float* masks, const int64_t* labels,
                    const int32_t* remap, int64_t* seg, int32_t* area, int n,
                    int64_t HW, void* stream);

/* The whole panoptic branch of _get_bboxes_single (:845-905) with no host round trip:
 * keep = (label != num_classes-1) & (score > 0.5) in query order, duplicate stuff
 * classes (label >= 80) merged into their first occurrence, kept masks resized to
 * (ho, wo), per-pixel argmax -> seg = id*1000 + label, then up to `rounds` rounds of
 * "drop segments with area <= 4 and redo the argmax" (the `while True` of :893-905; a
 * round after convergence returns at once).  The reference loops until nothing is
 * dropped: if the enqueued rounds were not enough, state.active is still 1 and
 * pn_panoptic_continue_f32 runs further rounds on the same buffers; state.all_gone
 * means every segment was filtered (the reference raises IndexError there, :882).
 * No kept query -> seg = 1 everywhere (:850).
 *   masks [Q][hi][wi] mask logits; labels/scores from pn_cls_argmax_f32 over all_cls
 *   state: pn_panoptic_state_bytes() bytes; its first int32 words, readable after the
 *          stream has drained: nkeep, active, rounds (that dropped something), all_gone
 *   up_scratch Q*ho*wo floats; area_scratch 256 int32; seg [ho*wo] int64 */
int64_t pn_panoptic_state_bytes(void);
int pn_panoptic_device_f32(const float* masks, const int64_t* labels, const float* scores,
                           int Q, int num_classes, int hi, int wi, int ho, int wo,
                           void* state, float* up_scratch, int32_t* area_scratch,
                           int64_t* seg, int rounds, void* stream);
int pn_panoptic_continue_f32(void* state, const float* up_scratch, int32_t* area_scratch,
                             int64_t* seg, int ho, int wo, int rounds, void* stream);

/* One fixed-shape fp32 record per image for the all-gather of predicted triplets that
 * replaces mmdet's pickled collect_results_gpu (tools/test.py:256-267):
 * rec = [labels 2R | rel_dists R*C1 | sub_pos R | obj_pos R], 4R + R*C1 floats. */
int pn_pack_triplets_f32(const int64_t* labels, const float* r_dists, const int64_t* sub_pos,
                         const int64_t* obj_pos, float* rec, int R, int C1, void* stream);

/* Byte copy with a bounded CU footprint (`wgs` workgroups): the device -> pinned-host copy of
 * `triplet2Result`'s fields (psgtr.py:15-51; 51 MB per 800x1333 image), which as a
 * hipMemcpyAsync runs as a chip-wide blit kernel that takes workgroup slots from concurrent
 * GEMMs.  `dst` may be pinned host memory (mapped in the device's address space) or device
 * memory; src / dst 16-byte aligned. */
int pn_copy_stream(const void* src, void* dst, int64_t bytes, int wgs, void* stream);

/* `triplet2Result`'s masks (psgtr.py:38-46: 2R x H0 x W0 numpy bool, 49 of the 51 MB) cross
 * PCIe as bits.  Device: `n` mask bytes (any non-zero = set; `bools` 8-byte aligned) ->
 * (n + 7) / 8 bytes, byte i bit j = element 8 i + j.  Host (no GPU call, runs on `threads`
 * host threads, 1..64): the inverse, one 0 / 1 byte per element = numpy bool. */
int pn_pack_bool_bits(const uint8_t* bools, uint8_t* bits, int64_t n, void* stream);
int pn_unpack_bits_host(const uint8_t* bits, uint8_t* bools, int64_t n, int threads);

/* Ground truth from the decoded panoptic PNG (the evaluator's side: pairnet/datasets/psg.py:
 * 354-372; the training-side loader: pairnet/datasets/pipelines/loading.py:128-147):
 *   rgb   [H][W][3] uint8, RGB                 ids / cats [G] int32: the annotation's segments
 *   masks [G][H][W] uint8 0/1 = (id(pixel) == ids[g]),  id = R + 256 G + 65536 B ([3P]
 *         panopticapi rgb2id); every listed segment, an id absent from the image -> zeros
 *   sem   [H][W] int32 or NULL: cats[g] of the LAST listed segment owning the pixel, else 255
 * G <= 256; rgb and masks 4-byte aligned.  G == 0 with sem: sem is filled with 255. */
int pn_pan_masks_u8(const uint8_t* rgb, const int* ids, const int* cats, uint8_t* masks, int* sem,
                    int G, int H, int W, void* stream);

/* Test-time image front end (configs/mask2former/pairnet.py:310-331, mean / std :229-231):
 * mmdet Resize(keep_ratio) [= mmcv.imresize = OpenCV INTER_LINEAR on uint8, fixed point] ->
 * Normalize(to_rgb) -> Pad -> ImageToTensor of one decoded image, fused.
 *   img  [H][W][3] uint8, BGR (cv2 order)        out [3][Hp][Wp] fp32, zero beyond (Hn, Wn)
 *   (Hn, Wn) = mmcv.rescale_size of (H, W); mean3 / stdinv3: HOST pointers, output order */
int pn_preprocess_u8_f32(const uint8_t* img, int H, int W, float* out, int Hn, int Wn, int Hp,
                         int Wp, const float* mean3, const float* stdinv3, int to_rgb,
                         void* stream);

/* Evaluator feed (pairnet/evaluation/sgg_metrics.py:1276-1380, mask_iou :1374-1380):
 * masks as bit rows (bit i of word w = pixel 64w+i) and the exact integer counts
 * behind IoU: inter[i][j] = |pred_i & gt_j|, area_pred[i], area_gt[j]. */
int pn_pack_mask_bits(const uint8_t* masks, uint64_t* words, int64_t rows, int64_t HW,
                      void* stream);
int pn_mask_iou_counts(const uint64_t* pred_words, int P, const uint64_t* gt_words, int G,
                       int64_t nwords, int32_t* inter, int32_t* area_pred, int32_t* area_gt,
                       void* stream);

/* Evaluator feed, part 2: SGRecall's triplet matching (sgg_metrics.py:173-252, :1311-1371).
 *   pn_pred_triplets   triplets[r] = (labels[r], 1 + argmax(rel_dists[r][1:]), labels[R+r]),
 *                      scores[r] = that maximum (:207-209, :1292-1294)
 *   pn_mask_or_rows    out[r] = words[a[r]] | words[b[r]]: the union masks of phrase
 *                      detection (:1343-1350)
 *   pn_triplet_match   match[p][g] = classes equal && IoU(subject) >= thr && IoU(object) >= thr
 *                      from the counts of pn_mask_iou_counts (inter [*][ld_inter], areas),
 *                      rows chosen through the *_row tables; phrdet: one IoU (the union
 *                      counts passed as the subject arguments). */
int pn_pred_triplets(const int64_t* labels, const float* r_dists, int32_t* triplets,
                     float* scores, int R, int C1, void* stream);
int pn_mask_or_rows(const uint64_t* words, const int32_t* a, const int32_t* b, uint64_t* out,
                    int rows, int64_t nwords, void* stream);
int pn_triplet_match(const int32_t* pred_triplets, const int32_t* gt_triplets, int P, int G,
                     const int32_t* inter, const int32_t* area_pred, const int32_t* area_gt,
                     int ld_inter, const int32_t* pred_sub_row, const int32_t* pred_obj_row,
                     const int32_t* gt_sub_row, const int32_t* gt_obj_row, double iou_thr,
                     int phrdet, int ignore_rel, uint8_t* match, void* stream);
/* The same triplet match on boxes, for `detection_method="bbox"` (the box-trunk sibling head):
 * `_compute_pred_matches_bbox` (sgg_metrics.py:1212-1273) with mmdet's
 * `bbox_overlaps(mode="iou", eps=1e-6)` in float32; boxes (x1, y1, x2, y2) in rows of ld_*
 * floats (so `refine_bboxes` [2R][5] can be passed as it is); phrdet: union boxes. */
int pn_triplet_match_boxes(const int32_t* pred_triplets, const int32_t* gt_triplets, int P, int G,
                           const float* pred_boxes, int ld_pred, const float* gt_boxes, int ld_gt,
                           const int32_t* pred_sub_row, const int32_t* pred_obj_row,
                           const int32_t* gt_sub_row, const int32_t* gt_obj_row, float iou_thr,
                           int phrdet, int ignore_rel, uint8_t* match, void* stream);

/* ------------------------------------------------------------------------- *
 * Box trunk of the sibling head CrossHeadBBox (pairnet_bbox_head.py:193-359): the
 * input-dependent glue of mmdet's two-stage, box-refining DeformableDetrTransformer
 * (built at :66, called at :215-228) between this library's GEMM / deformable-attention
 * entries.  All row-parallel and HBM-bound.
 * ------------------------------------------------------------------------- */
/* out[b][r][:] = valid[b][r] ? x[b][r][:] : 0 on rows of C floats at stride ld (C % 4 == 0);
 * valid: bytes, advancing by valid_bstride per image (0: one table for the batch); x == out
 * allowed.  (gen_encoder_output_proposals: tokens whose proposal box leaves (0.01, 0.99) and
 * padded tokens are zeroed before enc_output; mmcv MultiScaleDeformableAttention zeroes the
 * value rows of padded tokens, `value.masked_fill(key_padding_mask[..., None], 0)`.) */
int pn_zero_rows_f32(const float* x, const uint8_t* valid, float* out, int B, int64_t rows,
                     int C, int64_t ld, int64_t valid_bstride, void* stream);
/* y = sigmoid(x) elementwise (enc_bbox_preds, pairnet_bbox_head.py:345-347; sigmoid(inf) = 1) */
int pn_sigmoid_f32(const float* x, float* y, int64_t n, void* stream);
/* Two-stage queries: ref[r][4] = sigmoid(unact[r][4]) and emb[r][512] =
 * get_proposal_pos_embed(unact) (128 sine features per coordinate, temperature 1e4). */
int pn_box_pos_embed_f32(const float* unact, float* ref, float* emb, int64_t rows, void* stream);
/* Decoder cross-attention operands (mmcv MultiScaleDeformableAttention.forward with 4-d
 * reference boxes): offaw row = [offsets 8*L*4*2 | logits 8*L*4] (stride ld floats);
 * aw [rows][8][L][4] = softmax over each head's L*4 logits; loc [rows][8][L][4][2] =
 * ref.xy + offset / 4 * ref.wh * 0.5 -- the operands of pn_msda_loc_f32. */
int pn_box_sampling_f32(const float* offaw, int64_t ld, const float* ref,
                        const float* valid_ratios, int rows_per_image, float* loc, float* aw,
                        int64_t rows, int L, void* stream);
/* (valid_ratios, nullable: [B][L][2] = (valid width / W_l, valid height / H_l) of a padded
 * batch -- the decoder's `reference_points * cat([valid_ratios, valid_ratios])`; image of row
 * r = r / rows_per_image.)
 * The encoder's self-attention operands on a PADDED batch: for every token of every image the
 * sampling locations / softmax weights with mmdet's get_reference_points,
 * ref = (x + .5) / (vr[lq] * W_lq) * vr[ls], location = ref + offset / (W_ls, H_ls); offaw rows
 * as above, tokens level by level; outputs are pn_msda_loc_f32's operands.  (Unpadded batches
 * use the fused pn_msda_f32, where every valid ratio is 1.) */
int pn_token_sampling_f32(const float* offaw, int64_t ld, const float* valid_ratios, float* loc,
                          float* aw, int B, int L, const int32_t* level_h, const int32_t* level_w,
                          void* stream);
/* Iterative box refinement: ref_out = sigmoid(delta + inverse_sigmoid(ref_in, eps=1e-5)),
 * [rows][4] (DeformableDetrTransformerDecoder.forward; also the last layer's
 * `outputs_coord`, pairnet_bbox_head.py:236-246). */
int pn_box_refine_f32(const float* delta, const float* ref_in, float* ref_out, int64_t rows,
                      void* stream);
/* Query ranking (pairnet_bbox_head.py:252-254): score[b][q] = max over classes of
 * softmax(logits[b], dim = the QUERY axis)[q]; logits [B][Nq][C], C <= 256. */
int pn_query_score_f32(const float* logits, float* score, int B, int Nq, int C, void* stream);
/* CrossHeadBBox._get_bboxes_single (:1056-1086): rows [subjects R | objects R]:
 * labels = argmax softmax + 1, det[row] = (x1, y1, x2, y2, max softmax), boxes cxcywh ->
 * xyxy * (img_w, img_h), clamped to the image, divided by scale_factor[4] (a HOST pointer)
 * when rescale. */
int pn_box_triplets_f32(const float* s_cls, const float* o_cls, const float* s_box,
                        const float* o_box, float* det, int64_t* labels, int R, int C,
                        float img_h, float img_w, const float* scale_factor, int rescale,
                        void* stream);

/* ------------------------------------------------------------------------- *
 * Loss forward of CrossHead2 on device outputs (first slice: the
 * values of `CrossHead2.loss`, pairnet_head.py:419-718; no backward).  The two
 * Hungarian assignments themselves run on the host, as in the reference
 * (`linear_sum_assignment(cost.cpu())`, matcher.py:262-264, and [3P] mmdet
 * MaskHungarianAssigner): these entries produce their cost matrices and the loss
 * scalars.
 * ------------------------------------------------------------------------- */
/* `PSGTr.forward_train`'s ground-truth mask preparation (frameworks/psgtr.py:126-141):
 * masks [G][h][w] uint8 0/1 -> zero-padded on the right / bottom to the batch tensor's
 * [H][W] (F.pad) -> nearest-neighbour resize to [Ho][Wo] (F.interpolate(mode="nearest"):
 * source index min(floor(dst * (float)in / out), in - 1); the reference passes
 * (Ho, Wo) = (H // 2, W // 2)).  out [G][Ho][Wo] uint8.  G, Ho <= 65535. */
int pn_gt_mask_prepare_u8(const uint8_t* masks, uint8_t* out, int G, int h, int w, int H, int W,
                          int Ho, int Wo, void* stream);

/* [3P] mmcv `point_sample` (pairnet_head.py:631-638) = grid_sample(maps, 2 p - 1, bilinear,
 * zero padding, align_corners=False) with ONE point set shared by all maps:
 * maps [P][h][w] float32, or uint8 0/1 when maps_are_u8 (ground-truth masks); pts [Np][2]
 * (x, y) in [0, 1]; out [P][Np]. */
int pn_point_sample_f32(const void* maps, int maps_are_u8, const float* pts, float* out, int P,
                        int h, int w, int Np, void* stream);
/* [3P] mmdet MaskHungarianAssigner's cost matrix (cfg configs/mask2former/pairnet.py:200-206;
 * called at pairnet_head.py:641-643): cost [Q][G] =
 *   -softmax(cls[q])[gt_labels[g]] w_cls + mean_p BCEwithLogits(pred_pts[q][p], gt_pts[g][p]) w_mask
 *   + (1 - (2 sum s t + eps) / (sum s + sum t + eps)) w_dice,  s = sigmoid(pred_pts[q]). */
int pn_mask_match_cost_f32(const float* cls /* [Q][ncls] */, int ncls,
                           const int64_t* gt_labels /* [G] */, const float* pred_pts /* [Q][Np] */,
                           const float* gt_pts /* [G][Np] */, float* cost, int Q, int G, int Np,
                           float w_cls, float w_mask, float w_dice, float dice_eps, void* stream);
/* `IdMatcher.assign` cost (approaches/matcher.py:250-258; called at pairnet_head.py:662-671):
 * cost [R][G] = -softmax(sub[r])[gt_sub[g]] w_sub - softmax(obj[r])[gt_obj[g]] w_obj
 *               - softmax(rel[r])[gt_rel[g]] w_rel. */
int pn_id_match_cost_f32(const float* sub, const float* obj /* [R][ncls] */, const float* rel
                         /* [R][nrel] */, int ncls, int nrel, const int64_t* gt_sub,
                         const int64_t* gt_obj, const int64_t* gt_rel, float* cost, int R, int G,
                         float w_sub, float w_obj, float w_rel, void* stream);
/* [3P] mmdet CrossEntropyLoss, softmax form, reduction "mean" (`subobj_cls_loss`,
 * pairnet_head.py:518-527): out[0] = loss_weight * mean over rows with target >= 0 of
 * class_weight[y] (logsumexp(x) - x[y]); rows with target < 0 are the unmatched queries the
 * reference masks out (`r_label_weights_mask`).  rows <= 4096. */
int pn_ce_mean_f32(const float* logits, int64_t ld, const int64_t* target,
                   const float* class_weight /* [C] or NULL */, float* out, int rows, int C,
                   float loss_weight, void* stream);
/* [3P] mmdet SeesawLoss, `loss_cls_classes` (`rel_cls_loss`, pairnet_head.py:529-536) over the
 * rows with target >= 0; cum_samples [C] = the loss's persistent label counts INCLUDING this
 * batch (the caller accumulates them, as seesaw_loss.py does before weighing).  C <= 64. */
int pn_seesaw_mean_f32(const float* logits, int64_t ld, const int64_t* target,
                       const float* cum_samples, float* out, int rows, int C, float p, float q,
                       float eps, float loss_weight, void* stream);
/* `BCEWithLogitsLoss` (losses/seg_losses.py:153-166) with the reference's
 * pos_weight = numel / #(target > 0) (pairnet_head.py:541-552): out[0] = loss_weight * mean,
 * out[1] = pos_weight. */
int pn_bce_posw_mean_f32(const float* logits, const float* target, float* out /* [2] */, int64_t n,
                         float loss_weight, void* stream);

/* SURVEY 8 f-4, first backward slice: the gradients of the three reductions above with respect to
 * their logits (`loss_sub_cls` / `loss_obj_cls`, `loss_r_cls`, `loss_match`; pairnet_head.py:518-552):
 *   CE      g[r][c] = loss_weight / n * class_weight[y] * (softmax(x_r)[c] - [c == y])
 *   Seesaw  g[r][j] = loss_weight / n * (softmax(x'_r)[j] - [j == y]), the seesaw weights constants of
 *           the backward pass ([3P] seesaw_ce_loss uses softmax(cls_score.detach()))
 *   BCE     g[i]    = loss_weight / n * ((1 - t) - (1 + (pos_weight - 1) t) sigmoid(-x))
 * rows with target < 0 get zeros; n = rows with target >= 0 (CE, Seesaw) / all elements (BCE).
 * Checked against autograd through the reference-pinned loss oracle (tests/test_losses_gpu.py). */
int pn_ce_mean_grad_f32(const float* logits, int64_t ld, const int64_t* target,
                        const float* class_weight /* [C] or NULL */, float* grad, int64_t ldg,
                        int rows, int C, float loss_weight, void* stream);
int pn_seesaw_mean_grad_f32(const float* logits, int64_t ld, const int64_t* target,
                            const float* cum_samples, float* grad, int64_t ldg, int rows, int C,
                            float p, float q, float eps, float loss_weight, void* stream);
int pn_bce_posw_mean_grad_f32(const float* logits, const float* target, float* grad, int64_t n,
                              float loss_weight, void* stream);

/* ------------------------------------------------------------------------- *
 * SURVEY 8 f-4, second backward slice (csrc/grad.hip): what the backward of Pair-Net's own tail
 * needs beside the forward kernels above -- the Relation Fusion decoder
 * (pairnet_head.py:353-378; layers: facebook_detr.py:378-432), the Pair Proposal Network
 * (pairnet_head.py:322-333) and the Matrix Learner (frameworks/cnn_factory.py:6-53).  In the
 * reference this is torch.autograd behind `losses.backward()` (mmcv's OptimizerHook, configs/
 * _base_/schedules/schedule_1x.py); pair-net_amd/grad.py composes these entry points with
 * pn_gemm_f32 (dX = dY W and dW = dY^T X on transposed operands) and pn_conv2d_nhwc_ex_f32.
 * Every reduction runs in a fixed order (no atomics).  Checked against autograd through the
 * reference-pinned oracle (tests/test_grad_gpu.py).
 * ------------------------------------------------------------------------- */
/* out[c][r] = in[r][c] for in [rows][ldi >= cols]; out [cols][ldo], columns rows .. out_cols-1 of
 * every output row are zero-filled (out_cols >= rows: pads a contraction length to 4). */
int pn_transpose_f32(const float* in, int64_t ldi, float* out, int64_t ldo, int rows, int cols,
                     int out_cols, void* stream);
/* out[c] (+)= sum_r x[r][c]  (bias / LayerNorm-weight gradients, second stage of the tap correlations);
 * scratch (nullable): up to 64 * cols floats, lets a tall matrix be summed in row chunks by many
 * workgroups + one combining launch (fixed order either way). */
int pn_colsum_f32(const float* x, int64_t ld, float* out, int rows, int cols, int accumulate,
                  float* scratch, int64_t scratch_floats, void* stream);
/* dx[i] = y[i] > 0 ? dy[i] : 0  (y: the ReLU's output; dx may alias dy) */
int pn_relu_bwd_f32(const float* dy, const float* y, float* dx, int64_t n, void* stream);
/* out[i] = a[i] + b[i % bn]  (x + row-periodic table; bn == n: a plain sum, out may alias a) */
int pn_add_periodic_f32(const float* a, const float* b, float* out, int64_t n, int64_t bn,
                        void* stream);
/* out[i] (+)= sum_b x[b][i], i < n  (gradient of a table broadcast over the batch) */
int pn_batch_sum_f32(const float* x, float* out, int B, int64_t n, int accumulate, void* stream);
/* nn.LayerNorm(256) backward from the saved INPUT x [rows][256]: dx, and gxhat = dy * xhat whose
 * column sum is d weight (d bias = column sum of dy). */
int pn_layernorm256_bwd_f32(const float* dy, const float* x, const float* gamma, float* dx,
                            float* gxhat, int rows, float eps, void* stream);
/* nn.MultiheadAttention core backward, 8 heads x 32 channels, from the saved projections
 * q [B*Nq][ldq], k / v [B*Nk][ldk / ldv] and the gradient of the concatenated head outputs dout
 * [B*Nq][ldo]: dq, dk, dv (same row layouts).  bits / rowall: the forward's boolean mask as
 * pn_mask_pack wrote it (both NULL: no mask).  scratch: 2 * B * 8 * Nq * Nk floats. */
int pn_mha_bwd_f32(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v,
                   int64_t ldv, const float* dout, int64_t ldo, float* dq, int64_t lddq, float* dk,
                   int64_t lddk, float* dv, int64_t lddv, const uint32_t* bits,
                   const int32_t* rowall, float* scratch, int B, int Nq, int Nk, float scale,
                   void* stream);
/* Backward of pn_gather_rows_f32: out[b][row][0:length] (+)= sum_{s: index[b][s] == row}
 * src[b][s][0:length]; src [B*slots][ld_src], out [B*rows_out][ld_out]. */
int pn_scatter_rows_add_f32(const float* src, int64_t ld_src, const int64_t* index, float* out,
                            int64_t ld_out, int B, int rows_out, int slots, int length,
                            int accumulate, void* stream);
/* Cosine block backward (pairnet_head.py:325-333): x [B][Q][256] this side's rows BEFORE
 * F.normalize, other_hat [B][Q][256] the other side's normalised rows, draw [B][Q][Q] the gradient
 * of importance_raw (transposed != 0: this side indexes draw's columns, i.e. the objects). */
int pn_cosine_bwd_f32(const float* draw, const float* x, const float* other_hat, float* dx, int B,
                      int Q, int transposed, float eps, void* stream);
/* Matrix Learner last layer (64 -> 1): gradient w.r.t. its input c [B][S][S][64] (a ReLU output:
 * masked where c == 0) from g [B][S][S]; w3 [49][64] as pn_mlearner_last_f32 takes it. */
int pn_mlearner_last_bwd_data_f32(const float* g, const float* w3, const float* c, float* dc, int B,
                                  int S, void* stream);
/* part[b*S + y][tap][c] = sum_x F[b][y][x][c] g[b][y + sgn (kh-3)][x + sgn (kw-3)], F [B][S][S][64],
 * g [B][S][S]; column-summed over its B*S rows it is d w3 [49][64] (F = the layer's input, g = d
 * importance, sgn = -1) or (d w1)^T (F = d c1, g = importance_raw, sgn = +1). */
int pn_tapcorr1_f32(const float* F, const float* g, float* part, int B, int S, int sgn,
                    void* stream);
/* Pixel decoder (pairnet_head.py:262; mmcv MultiScaleDeformableAttention.forward): from pn_msda_bwd_f32's
 * grad_sampling_loc / grad_attn_weight back to the gradient of the [offsets 8*L*4*2 | logits 8*L*4]
 * projection rows (stride ld) that pn_token_sampling_f32 turned into those operands on an unpadded
 * batch: d offset = d loc / (W_l, H_l), d logit = aw (d aw - sum aw d aw) per head. */
int pn_msda_offaw_bwd_f32(const float* grad_loc, const float* grad_aw, const float* aw,
                          float* d_offaw, int64_t ld, int64_t rows, int L,
                          const int32_t* level_h /* host */, const int32_t* level_w /* host */,
                          void* stream);
/* Backward of pn_groupnorm_nhwc_f32 (no ReLU) from its INPUT x: dx [B][HW][256] (dense), gxhat = dy *
 * xhat (column sum over B*HW rows: d weight; d bias: column sum of dy); stats: B * G * 4 floats of
 * scratch; x / dy images start at multiples of x_bstride / dy_bstride floats. */
int pn_groupnorm_nhwc_bwd_f32(const float* x, const float* dy, const float* gamma, float* dx,
                              float* gxhat, float* stats, int B, int64_t HW, int G, float eps,
                              int64_t x_bstride, int64_t dy_bstride, void* stream);
/* Backbone backward (mmdet ResNet behind configs/mask2former/pairnet.py:9-19, frozen BatchNorm folded
 * into the convolutions; pair-net_amd/grad.py BackboneGrad) and the Matrix Learner's 64 -> 64 7x7 layer
 * (cnn_factory.py:30-41).  Weight gradient of a K x K convolution
 * (stride, pad) between channel-last maps dY [B][Ho][Wo][Co] and X [B][Hi][Wi][Ci] (Ci, Co % 64 == 0):
 * part[chunk][co][tap][ci] over chunks of rows_per output rows per image, B * ceil(Ho / rows_per)
 * chunks of Co*K*K*Ci floats; their column sum is dW [Co][K*K][Ci]. */
int pn_conv_wgrad_f32(const float* dY, const float* X, float* part, int B, int Hi, int Wi, int Ho,
                      int Wo, int Ci, int Co, int K, int stride, int pad, int rows_per, void* stream);
/* out[b][y][x][:] (+)= in[b][y/2][x/2][:] at even (y, x), 0 elsewhere: [B][Ho][Wo][C] -> [B][Hi][Wi][C] */
int pn_dilate2_f32(const float* in, float* out, int B, int Hi, int Wi, int Ho, int Wo, int C,
                   int accumulate, void* stream);
/* out[b][i][j][:] = in[b][2i][2j][:]: [B][Hi][Wi][C] -> [B][Ho][Wo][C] */
int pn_subsample2_f32(const float* in, float* out, int B, int Hi, int Wi, int Ho, int Wo, int C,
                      void* stream);
/* x[r][0:cols] *= s[r] */
int pn_scale_rows_f32(float* x, const float* s, int64_t rows, int64_t cols, void* stream);
/* out[ci][T-1-t][co] = in[co][t][ci]: a "same" convolution's weight as its data gradient reads it */
int pn_conv_weight_bwd_layout_f32(const float* in, float* out, int Co, int T, int Ci, void* stream);

/* ------------------------------------------------------------------------- *
 * The optimizer step (csrc/optim.hip) over one flat fp32 parameter buffer: what mmcv's
 * OptimizerHook(grad_clip=dict(max_norm=0.1, norm_type=2)) + torch.optim.AdamW do per iteration
 * (configs/mask2former/pairnet.py:353-368; paramwise lr_mult / norm_decay_mult as per-segment
 * multipliers).  pair-net_amd/train.py drives it.
 * ------------------------------------------------------------------------- */
/* out[0] = || pre * g ||_2, out[1] = min(1, max_norm / (out[0] + 1e-6)) (1 if max_norm <= 0);
 * scratch: 256 doubles.  Deterministic (two-stage sum in double, fixed order). */
int pn_grad_norm_clip_f32(const float* g, int64_t n, float pre, float max_norm, float* out,
                          double* scratch, void* stream);
/* AdamW (torch.optim.AdamW's arithmetic) on p / g / m / v [n] with the effective gradient
 * pre * clip[1] * g (clip: the device pair written by pn_grad_norm_clip_f32, or NULL); segment s
 * covers [seg_off[s], seg_off[s+1]) and uses lr * seg_lr[s], weight_decay * seg_wd[s]; `step`
 * counts from 1 (bias corrections). */
int pn_adamw_f32(float* p, const float* g, float* m, float* v, int64_t n, const int64_t* seg_off,
                 const float* seg_lr, const float* seg_wd, int nseg, float lr, float beta1,
                 float beta2, float eps, float weight_decay, int step, const float* clip, float pre,
                 void* stream);

/* ------------------------------------------------------------------------- *
 * fp32 GEMMs on the bf16 matrix pipe from PRE-SPLIT operands (csrc/gemm_s3.hip, round 6)
 *
 * Arithmetic: every fp32 number is exactly x0 + x1 + x2 with bf16 pieces (each the
 * round-to-nearest-even bf16 of what is left); a product of bf16 numbers is exact in fp32; of the
 * nine piece products of a*b the six largest are summed by v_mfma_f32_32x32x16_bf16 into fp32
 * accumulators, the three dropped ones are together < 2^-26 |ab|.  Measured against fp64 the
 * result is at or below the error of the exact-fp32 MFMA of pn_gemm_f32 on every shape of the
 * path (profiles/r06_gemm_s3_error.txt).  Same call sites as pn_gemm_f32 /
 * pn_linear_res_ln_f32 for the pixel decoder's encoder (value_proj / sampling_offsets /
 * attention_weights / output_proj / FFN behind pairnet_head.py:262).
 *
 * "S3" operand of an [R x K] matrix (K % 16 == 0): blocks of 32 rows x 16 k, block (rb, kb) at
 * byte (rb * K/16 + kb) * 3072; three 1 KiB planes (x0, x1, x2) per block; inside a plane 16
 * bytes (8 consecutive k) per MFMA lane, lane = (k % 16 / 8) * 32 + r % 32.  Rows are padded to
 * a multiple of 32 (pn_s3_bytes); A and W operands use the same layout.
 * ------------------------------------------------------------------------- */
int64_t pn_s3_bytes(int rows, int K);
/* S = split(X[r][0:K] + add[r % add_rows][0:K]) for fp32 rows X [rows][ld] (add may be NULL;
 * ld % 4 == 0, 16-byte aligned).  Rows of the last block beyond `rows` are written as zeros. */
int pn_s3_split_f32(const float* X, int64_t ld, const float* add, int add_rows, void* S, int rows,
                    int K, void* stream);
/* The exact fp32 values back: X[r][0:K] = x0 + x1 + x2. */
int pn_s3_join_f32(const void* S, float* X, int64_t ld, int rows, int K, void* stream);
/*   out[m][n] = act( sum_k A*[m][k] W[n][k] + bias[n] ),   A* = A2 for n >= a2_from_col, else A
 * or, with gamma != NULL (N == 256):
 *   out[m][:] = LayerNorm(sum_k A[m][k] W[:][k] + bias + res[m][:]) * gamma + beta
 * (two-pass moments, eps inside the square root: [3P] nn.LayerNorm).  Any subset of three outputs:
 *   C      fp32 rows [M][ldc]
 *   CS     S3 [M x N] of out                      (the next GEMM's A operand)
 *   CS_pos S3 [M x N] of out + pos[m % pos_rows]  (the next layer's query operand, `query + query_pos`)
 * A, A2, W, res_s3 are S3 operands ([M x K], [M x K], [N x K], [M x 256]); K % 32 == 0, N % 32 == 0,
 * a2_from_col % 256 == 0.  pos is in "P8" order: [pos_rows / 32][N / 16][2][32][8] (the S3 piece
 * order with fp32 elements; pair-net_amd/hip.py pos8()).  One workgroup of 8 waves per 96 x 256 tile
 * (192 x 256 with W shared by two row groups when N >= 512 and there is no row epilogue); up to 64
 * columns beyond the last whole 256-column tile go to a one-wave-per-block kernel. */
typedef struct pn_gemm_s3_desc {
  const void* A;  const void* A2;  int32_t a2_from_col;
  const void* W;  const float* bias;            /* bias [N] or NULL */
  int32_t M, N, K;
  int32_t relu;
  float* C;       int64_t ldc;
  void* CS;
  void* CS_pos;   const float* pos;  int32_t pos_rows;   /* pos [pos_rows][N] fp32 */
  const void* res_s3;  const float* gamma;  const float* beta;  float eps;
  int32_t flags;                                 /* PN_GEMM_S3_* (tuning / tests), else 0 */
  /* plain epilogue only (round 6, the Swin blocks): out = act(sum + bias) + res[m][n]; act 0 = `relu`
   * decides, 1 ReLU, 2 exact (erf) GELU, 3 = relu(sum + bias + res) (ReLU after the shortcut: a
   * ResNet bottleneck's last convolution); res fp32 rows [M][ldres] or NULL */
  int32_t act;  const float* res;  int64_t ldres;
} pn_gemm_s3_desc;
#define PN_GEMM_S3_TILE96 1   /* force the 96 x 256 tile where the 192 x 256 one would be taken */
#define PN_GEMM_S3_TILE192 2  /* force the 192 x 256 tile (N >= 512, plain epilogue) whatever its tile count */
int pn_gemm_s3_f32(const pn_gemm_s3_desc* d, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PAIRNET_HIP_H_ */
