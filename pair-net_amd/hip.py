"""ctypes binding of libpairnet_hip.so (include/pairnet_hip.h).

torch is used here only for what the boundary needs from it: device memory
(`Tensor.data_ptr()`) and the current HIP stream.  There is NO fallback: if the
library is missing or a launch fails, a RuntimeError is raised.
"""
import contextlib
import ctypes as C
import os
import threading

import torch

from .build import LIB_PATH

# (tuning aid: PAIRNET_LIB=<path> runs the package against an alternative BUILD of the same
# library -- a kernel variant compiled with a build-time knob, tools/bench_variant.py -- with
# the same ABI check; there is still no fallback)
LIB_PATH = os.environ.get("PAIRNET_LIB") or LIB_PATH

_i32, _i64, _f32, _vp = C.c_int32, C.c_int64, C.c_float, C.c_void_p

GEMM_RELU, GEMM_A_COLMAJOR, GEMM_FORCE_TILE, GEMM_FORCE_SKINNY = 1, 2, 4, 8
GEMM_GELU = 256
GEMM_FORCE_TILE64, GEMM_FORCE_TILE128x64 = 16, 32
GEMM_RELU_AFTER_RES = 128
GEMM_GROUP_MAX = 18       # problems per pn_gemm_group_f32 launch (csrc/gemm.hip)


class GemmDesc(C.Structure):
    _fields_ = [("A", _vp), ("lda", _i64), ("strideA", _i64),
                ("Aadd", _vp), ("ldaadd", _i64), ("aadd_rows", _i32),
                ("aadd_from_col", _i32),
                ("W", _vp), ("ldw", _i64), ("strideW", _i64),
                ("bias", _vp),
                ("Res", _vp), ("ldres", _i64), ("strideRes", _i64),
                ("C", _vp), ("ldc", _i64), ("strideC", _i64),
                ("M", _i32), ("N", _i32), ("K", _i32), ("batch", _i32),
                ("flags", _i32),
                ("splitk_scratch", _vp), ("splitk_scratch_floats", _i64)]


class GemmS3Desc(C.Structure):
    _fields_ = [("A", _vp), ("A2", _vp), ("a2_from_col", _i32),
                ("W", _vp), ("bias", _vp),
                ("M", _i32), ("N", _i32), ("K", _i32), ("relu", _i32),
                ("C", _vp), ("ldc", _i64),
                ("CS", _vp),
                ("CS_pos", _vp), ("pos", _vp), ("pos_rows", _i32),
                ("res_s3", _vp), ("gamma", _vp), ("beta", _vp), ("eps", _f32), ("flags", _i32),
                ("act", _i32), ("res", _vp), ("ldres", _i64)]


_SIGS = {
    "pn_abi_version": (C.c_int, []),
    "pn_s3_bytes": (_i64, [_i32, _i32]),
    "pn_s3_split_f32": (C.c_int, [_vp, _i64, _vp, _i32, _vp, _i32, _i32, _vp]),
    "pn_s3_join_f32": (C.c_int, [_vp, _vp, _i64, _i32, _i32, _vp]),
    "pn_gemm_s3_f32": (C.c_int, [C.POINTER(GemmS3Desc), _vp]),
    "pn_gemm_f32": (C.c_int, [C.POINTER(GemmDesc), _vp]),
    "pn_gemm_variant": (C.c_int, [C.POINTER(GemmDesc)]),
    "pn_gemm_grid_size": (C.c_int, [C.POINTER(GemmDesc)]),
    "pn_gemm_wgs_per_cu": (C.c_int, []),
    "pn_gemm_group_f32": (C.c_int, [C.POINTER(GemmDesc), C.c_int, _vp]),
    "pn_conv2d_nhwc_f32": (C.c_int, [_vp, _vp, _vp, _vp] + [_i32] * 10 + [_vp]),
    "pn_conv2d_nhwc_ex_f32": (C.c_int, [_vp] * 5 + [_i32] * 10 + [_vp, _i64, _vp]),
    "pn_stem7x7s2_f32": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "pn_maxpool3x3s2_nhwc_f32": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "pn_winograd_f23_input_f32": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "pn_winograd_f23_output_f32": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "pn_winograd_f43_input_f32": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "pn_winograd_f43_output_f32": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "pn_layernorm_f32": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i32, _f32, _vp]),
    "pn_layernorm_rows_s3_f32": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _i64, _i32, _f32, _vp]),
    "pn_layernorm_rows_f32": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _i64, _i64, _i32, _f32, _vp]),
    "pn_patch_merge_ln_f32": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _vp]),
    "pn_patch_im2col4_f32": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp]),
    "pn_window_attention_f32": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _i64] + [_i32] * 7 +
                                [_f32, _vp]),
    "pn_window_attention_s3_f32": (C.c_int, [_vp, _i64, _vp, _vp, _vp] + [_i32] * 7 + [_f32, _vp]),
    "pn_groupnorm_nblk": (C.c_int, [_i64]),
    "pn_groupnorm_nhwc_f32": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i64, _i32,
                                        _i32, _f32, _i32, _i64, _i64, _vp]),
    "pn_l2normalize_f32": (C.c_int, [_vp, _vp, _i64, _i32, _f32, _vp]),
    "pn_ffn_scratch_floats": (_i64, [_i32, _i32]),
    "pn_ffn_ln_f32": (C.c_int, [_vp] * 9 + [_i32, _i32, _i32, _f32, _vp]),
    "pn_ffn_ln2_f32": (C.c_int, [_vp] * 12 + [_i32, _i32, _i32, _f32, _vp]),
    "pn_msda_f32": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i32, _i32, C.POINTER(_i32),
                              C.POINTER(_i32), _vp]),
    "pn_msda_ex_f32": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i32, _i32, C.POINTER(_i32),
                                 C.POINTER(_i32), _i32, _vp]),
    "pn_linear_res_ln_f32": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _vp, _i64,
                                       _i32, _i32, _i32, _f32, _vp]),
    "pn_gt_mask_prepare_u8": (C.c_int, [_vp, _vp] + [_i32] * 7 + [_vp]),
    "pn_point_sample_f32": (C.c_int, [_vp, _i32, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "pn_mask_match_cost_f32": (C.c_int, [_vp, _i32, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _f32,
                                         _f32, _f32, _f32, _vp]),
    "pn_id_match_cost_f32": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _i32, _i32,
                                       _f32, _f32, _f32, _vp]),
    "pn_ce_mean_f32": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _i32, _i32, _f32, _vp]),
    "pn_seesaw_mean_f32": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _i32, _i32, _f32, _f32, _f32, _f32,
                                     _vp]),
    "pn_bce_posw_mean_f32": (C.c_int, [_vp, _vp, _vp, _i64, _f32, _vp]),
    "pn_ce_mean_grad_f32": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _i64, _i32, _i32, _f32, _vp]),
    "pn_seesaw_mean_grad_f32": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _i64, _i32, _i32, _f32, _f32,
                                          _f32, _f32, _vp]),
    "pn_bce_posw_mean_grad_f32": (C.c_int, [_vp, _vp, _vp, _i64, _f32, _vp]),
    "pn_transpose_f32": (C.c_int, [_vp, _i64, _vp, _i64, _i32, _i32, _i32, _vp]),
    "pn_colsum_f32": (C.c_int, [_vp, _i64, _vp, _i32, _i32, _i32, _vp, _i64, _vp]),
    "pn_relu_bwd_f32": (C.c_int, [_vp, _vp, _vp, _i64, _vp]),
    "pn_add_periodic_f32": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _vp]),
    "pn_batch_sum_f32": (C.c_int, [_vp, _vp, _i32, _i64, _i32, _vp]),
    "pn_layernorm256_bwd_f32": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _f32, _vp]),
    "pn_mha_bwd_f32": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64,
                                 _vp, _i64, _vp, _vp, _vp, _i32, _i32, _i32, _f32, _vp]),
    "pn_scatter_rows_add_f32": (C.c_int, [_vp, _i64, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _vp]),
    "pn_cosine_bwd_f32": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _f32, _vp]),
    "pn_mlearner_last_bwd_data_f32": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _vp]),
    "pn_tapcorr1_f32": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "pn_conv_weight_bwd_layout_f32": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp]),
    "pn_msda_offaw_bwd_f32": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i32, C.POINTER(_i32),
                                        C.POINTER(_i32), _vp]),
    "pn_groupnorm_nhwc_bwd_f32": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i64, _i32, _f32, _i64,
                                            _i64, _vp]),
    "pn_winograd_weights_f32": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp]),
    "pn_conv_wgrad_f32": (C.c_int, [_vp, _vp, _vp] + [_i32] * 11 + [_vp]),
    "pn_dilate2_f32": (C.c_int, [_vp, _vp] + [_i32] * 7 + [_vp]),
    "pn_subsample2_f32": (C.c_int, [_vp, _vp] + [_i32] * 6 + [_vp]),
    "pn_scale_rows_f32": (C.c_int, [_vp, _vp, _i64, _i64, _vp]),
    "pn_grad_norm_clip_f32": (C.c_int, [_vp, _i64, _f32, _f32, _vp, _vp, _vp]),
    "pn_adamw_f32": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _i32, _f32, _f32, _f32, _f32,
                               _f32, _i32, _vp, _f32, _vp]),
    "pn_msda_loc_f32": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "pn_msda_bwd_f32": (C.c_int, [_vp, _i64] + [_vp] * 8 + [_i32, _i32, _i32, _i32, _vp]),
    "pn_gather_probe_f32": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _vp]),
    "pn_sine_pe_f32": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _f32, _vp]),
    "pn_sine_pe_offset_f32": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _f32, _f32, _vp]),
    "pn_bilinear_nhwc_f32": (C.c_int, [_vp, _vp] + [_i32] * 7 + [_i64, _i64, _vp]),
    "pn_bilinear_planar_f32": (C.c_int, [_vp, _vp, _i64] + [_i32] * 4 + [_vp]),
    "pn_bilinear_planar_gt0_u8": (C.c_int, [_vp, _vp, _i64] + [_i32] * 4 + [_vp]),
    "pn_mask_pack": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _vp]),
    "pn_mask_pack_stencil": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _vp]),
    "pn_mask_stencil_gemm_f32": (C.c_int, [_vp, _i64, _i64, _vp, _i64, _i64, _vp, _vp] +
                                 [_i32] * 9 + [_vp]),
    "pn_bilinear_stencil_rows_f32": (C.c_int, [_vp, _vp] + [_i32] * 6 + [_i64, _i64, _vp]),
    "pn_attn_scratch_floats": (_i64, [_i32, _i32, _i32]),
    "pn_attention_f32": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _vp,
                                   _i64, _vp, _i32, _i32, _i32, _f32, _vp]),
    "pn_ppn_front_f32": (C.c_int, [_vp] * 6 + [_i32, _i32, _f32, _vp]),
    "pn_mlearner_first_f32": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "pn_mlearner_last_f32": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "pn_topk_pairs": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "pn_topk_f32": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "pn_topk_strided_f32": (C.c_int, [_vp, _i64, _i64, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "pn_gather_rows_f32": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i64, _vp]),
    "pn_cls_argmax_f32": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _i32, _vp]),
    "pn_rel_dists_f32": (C.c_int, [_vp, _vp, _i64, _i32, _vp]),
    "pn_softmax_fg_f32": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _vp]),
    "pn_row_argmax_f32": (C.c_int, [_vp, _vp, _i64, _i32, _vp]),
    "pn_triplet_finish": (C.c_int, [_vp] * 9 + [_i32, _i32, _vp]),
    "pn_panoptic_f32": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i64, _vp]),
    "pn_panoptic_state_bytes": (_i64, []),
    "pn_panoptic_device_f32": (C.c_int, [_vp, _vp, _vp] + [_i32] * 6 + [_vp, _vp, _vp, _vp,
                                                                        _i32, _vp]),
    "pn_panoptic_continue_f32": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "pn_pack_triplets_f32": (C.c_int, [_vp] * 5 + [_i32, _i32, _vp]),
    "pn_copy_stream": (C.c_int, [_vp, _vp, _i64, _i32, _vp]),
    "pn_pack_bool_bits": (C.c_int, [_vp, _vp, _i64, _vp]),
    "pn_unpack_bits_host": (C.c_int, [_vp, _vp, _i64, _i32]),
    "pn_pred_triplets": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _vp]),
    "pn_mask_or_rows": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i64, _vp]),
    "pn_triplet_match": (C.c_int, [_vp, _vp, _i32, _i32, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp,
                                   C.c_double, _i32, _i32, _vp, _vp]),
    "pn_triplet_match_boxes": (C.c_int, [_vp, _vp, _i32, _i32, _vp, _i32, _vp, _i32, _vp, _vp, _vp,
                                         _vp, _f32, _i32, _i32, _vp, _vp]),
    "pn_pan_masks_u8": (C.c_int, [_vp] * 5 + [_i32] * 3 + [_vp]),
    "pn_preprocess_u8_f32": (C.c_int, [_vp, _i32, _i32, _vp] + [_i32] * 4 + [C.POINTER(_f32),
                                                                            C.POINTER(_f32), _i32, _vp]),
    "pn_pack_mask_bits": (C.c_int, [_vp, _vp, _i64, _i64, _vp]),
    "pn_mask_iou_counts": (C.c_int, [_vp, _i32, _vp, _i32, _i64, _vp, _vp, _vp, _vp]),
    "pn_zero_rows_f32": (C.c_int, [_vp, _vp, _vp, _i32, _i64, _i32, _i64, _i64, _vp]),
    "pn_token_sampling_f32": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _i32, _i32, C.POINTER(_i32),
                                        C.POINTER(_i32), _vp]),
    "pn_sine_pe_valid_f32": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _f32, _vp]),
    "pn_sigmoid_f32": (C.c_int, [_vp, _vp, _i64, _vp]),
    "pn_box_pos_embed_f32": (C.c_int, [_vp, _vp, _vp, _i64, _vp]),
    "pn_box_sampling_f32": (C.c_int, [_vp, _i64, _vp, _vp, _i32, _vp, _vp, _i64, _i32, _vp]),
    "pn_box_refine_f32": (C.c_int, [_vp, _vp, _vp, _i64, _vp]),
    "pn_query_score_f32": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp]),
    "pn_box_triplets_f32": (C.c_int, [_vp] * 5 + [_vp, _i32, _i32, _f32, _f32, C.POINTER(_f32),
                                                _i32, _vp]),
}
EXPORTS = tuple(_SIGS)
ABI_VERSION = 28   # PN_ABI_VERSION of include/pairnet_hip.h these bindings were written for

_lib = None


def lib():
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libpairnet_hip.so is missing (%s): run __graft_entry__.build(); "
                "there is no CPU / PyTorch fallback for the Pair-Net hot path" % LIB_PATH)
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(handle, name)
            fn.restype, fn.argtypes = res, args
        if handle.pn_abi_version() != ABI_VERSION:
            raise RuntimeError("libpairnet_hip.so ABI mismatch; rebuild")
        _lib = handle
    return _lib


class _Reserve(threading.local):
    slots = 0


_reserve = _Reserve()


@contextlib.contextmanager
def reserve_slots(n):
    """Within the block, every persistent tile-kernel launch of THIS thread carries
    PN_GEMM_RESERVE(n): n resident workgroup slots stay free for the kernels of concurrent
    streams.  A per-call hint in the descriptor's flags (include/pairnet_hip.h); nothing
    process-wide is changed, and a captured hipGraph keeps the grids it was captured with."""
    old = _reserve.slots
    _reserve.slots = max(0, int(n)) // 8 * 8
    try:
        yield
    finally:
        _reserve.slots = old


def with_reserve(fn):
    """Method decorator: run under `reserve_slots(self.grid_reserve)`."""
    import functools

    @functools.wraps(fn)
    def wrapper(self, *args, **kw):
        with reserve_slots(getattr(self, "grid_reserve", 0)):
            return fn(self, *args, **kw)
    return wrapper


def _reserve_flag():
    return ((_reserve.slots // 8) & 0x3ff) << 16


def _stream():
    return torch.cuda.current_stream().cuda_stream


def on_device(fn):
    """Method decorator: run with the object's GPU as the current device, so that the
    launches go to THAT device's current stream (a head on cuda:1 called while cuda:0 is
    current would otherwise launch on device 0 against device-1 pointers).  The device is
    `self.device`, or the first device tensor among the arguments before `.to()` was called."""
    import functools

    def first_tensor(objs):
        for o in objs:
            if isinstance(o, torch.Tensor) and o.is_cuda:
                return o.device
            if isinstance(o, (list, tuple)):
                d = first_tensor(o)
                if d is not None:
                    return d
            if isinstance(o, dict):
                d = first_tensor(o.values())
                if d is not None:
                    return d
        return None

    @functools.wraps(fn)
    def wrapper(self, *args, **kw):
        dev = getattr(self, "device", None)
        if dev is None or dev.type != "cuda":
            dev = first_tensor(args)
        if dev is None or dev.type != "cuda" or dev.index == torch.cuda.current_device():
            return fn(self, *args, **kw)
        with torch.cuda.device(dev):
            return fn(self, *args, **kw)
    return wrapper


GEMM_KERNELS = {0: "k_gemm_skinny<A_ROW>", 1: "k_gemm_skinny<A_COL>",
                2: "k_gemm_tile<128,64,64,32,A_ROW>", 3: "k_gemm_tile<128,64,64,32,A_COL>",
                4: "k_gemm_tile<128,128,64,64,A_ROW>", 5: "k_gemm_tile<128,128,64,64,A_COL>",
                6: "k_gemm_tile<64,64,32,32,A_ROW>", 7: "k_gemm_tile<64,64,32,32,A_COL>"}


class KernelTimer:
    """Optional per-launch HIP-event timing (bench.py's roofline leg).  Events are
    recorded on the stream the kernels are launched on (torch's current stream).
    `only`: kernel names to bracket (None = all instrumented launches)."""

    def __init__(self, only=None):
        self.only = set(only) if only else None
        self.records = []   # (name, flops, bytes, start_event, end_event)
        self.meta = []      # per record: problem description (GEMMs: M, N, K, batch) or None

    def summary(self):
        torch.cuda.synchronize()
        agg = {}
        for name, flops, nbytes, s, e in self.records:
            a = agg.setdefault(name, dict(launches=0, ms=0.0, flops=0.0, bytes=0.0))
            a["launches"] += 1
            a["ms"] += s.elapsed_time(e)
            a["flops"] += flops
            a["bytes"] += nbytes
        return agg


TIMER = None


def _launch(name, flops, nbytes, fn, meta=None):
    t = TIMER
    if t is None or (t.only is not None and name not in t.only):
        return fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    r = fn()
    e.record()
    t.records.append((name, flops, nbytes, s, e))
    t.meta.append(meta)
    return r


def _ptr(t, dtype=torch.float32):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("pairnet_hip kernels take device tensors (got %s)" % t.device)
    if t.dtype != dtype:
        raise RuntimeError("expected %s, got %s" % (dtype, t.dtype))
    return t.data_ptr()


def _check(code, what):
    if code != 0:
        raise RuntimeError("%s failed: %s" % (
            what, "argument contract violated" if code == -1 else "hipError %d" % code))


def _rowmajor(t):
    """(rows, ld) of a 2-D view whose last dim is contiguous."""
    assert t.dim() == 2 and t.stride(1) == 1, (t.shape, t.stride())
    return t.shape[0], t.stride(0)


def gemm_desc(A, W, Cout, *, M, N, K, lda, ldw, ldc, bias=None, aadd=None, ldaadd=0,
              aadd_rows=0, aadd_from_col=0, res=None, ldres=0, batch=1, sA=0, sW=0, sC=0,
              sRes=0, relu=False, colmajor=False, force=None, into=None,
              relu_after=False, scratch=None, gelu=False, ksplit=0):
    """Fill a pn_gemm_desc; tensors only supply base pointers."""
    d = into if into is not None else GemmDesc()
    d.A, d.lda, d.strideA = _ptr(A), lda, sA
    d.Aadd, d.ldaadd, d.aadd_rows, d.aadd_from_col = _ptr(aadd), ldaadd, aadd_rows, aadd_from_col
    d.W, d.ldw, d.strideW = _ptr(W), ldw, sW
    d.bias = _ptr(bias)
    d.Res, d.ldres, d.strideRes = _ptr(res), ldres, sRes
    d.C, d.ldc, d.strideC = _ptr(Cout), ldc, sC
    d.M, d.N, d.K, d.batch = M, N, K, batch
    d.flags = (GEMM_RELU if relu else 0) | (GEMM_A_COLMAJOR if colmajor else 0) | \
        {None: 0, "tile": GEMM_FORCE_TILE, "skinny": GEMM_FORCE_SKINNY, "tile64": 16,
         "tile128x64": 32}[force] | _reserve_flag() | ((ksplit & 31) << 26) | \
        (GEMM_RELU_AFTER_RES if relu_after else 0) | (GEMM_GELU if gelu else 0)
    d.splitk_scratch = _ptr(scratch)
    d.splitk_scratch_floats = scratch.numel() if scratch is not None else 0
    return d


def gemm(A, W, Cout, **kw):
    """Raw pn_gemm_f32 call."""
    d = gemm_desc(A, W, Cout, **kw)
    if TIMER is None:
        _check(lib().pn_gemm_f32(C.byref(d), _stream()), "pn_gemm_f32")
        return
    name = GEMM_KERNELS[lib().pn_gemm_variant(C.byref(d))]
    # every operand of the fused op once: A, W, C, and where present the residual read by the
    # epilogue and the row-periodic addend on A (positional encodings)
    nbytes = 4.0 * d.batch * (d.M * d.K + d.N * d.K + d.M * d.N)
    if d.Res:
        nbytes += 4.0 * d.batch * d.M * d.N
    if d.Aadd:
        nbytes += 4.0 * min(d.aadd_rows, d.M) * d.K
    _check(_launch(name, 2.0 * d.M * d.N * d.K * d.batch, nbytes,
                   lambda: lib().pn_gemm_f32(C.byref(d), _stream()),
                   meta=(d.M, d.N, d.K, d.batch, bool(d.splitk_scratch))), "pn_gemm_f32")


def gemm_group(problems):
    """`problems`: list of kwargs dicts for gemm_desc (+ 'A','W','C'), <= 16, row-major:
    one launch of the grouped (persistent 64x64 tile) kernel."""
    n = len(problems)
    arr = (GemmDesc * n)()
    flops = nbytes = 0.0
    for i, pr in enumerate(problems):
        pr = dict(pr)
        d = gemm_desc(pr.pop("A"), pr.pop("W"), pr.pop("C"), into=arr[i], **pr)
        flops += 2.0 * d.M * d.N * d.K * d.batch
        nbytes += 4.0 * d.batch * (d.M * d.K + d.N * d.K + d.M * d.N)
    _check(_launch("k_gemm_group", flops, nbytes,
                   lambda: lib().pn_gemm_group_f32(arr, n, _stream())), "pn_gemm_group_f32")


def linear(x, weight, bias, out, *, aadd=None, aadd_from_col=0, res=None, relu=False,
           force=None, relu_after=False, scratch=None, gelu=False):
    """out = act((x + aadd[row % len(aadd)]) @ weight.T + bias) + res on 2-D views
    (aadd only feeds output columns >= aadd_from_col)."""
    M, lda = _rowmajor(x)
    N, ldw = _rowmajor(weight)
    Mo, ldc = _rowmajor(out)
    assert Mo == M and out.shape[1] == N and weight.shape[1] == x.shape[1]
    kw = {}
    if aadd is not None:
        ra, lda2 = _rowmajor(aadd)
        kw.update(aadd=aadd, ldaadd=lda2, aadd_rows=ra, aadd_from_col=aadd_from_col)
    if res is not None:
        rr, ldr = _rowmajor(res)
        assert rr == M
        kw.update(res=res, ldres=ldr)
    gemm(x, weight, out, M=M, N=N, K=x.shape[1], lda=lda, ldw=ldw, ldc=ldc, bias=bias,
         relu=relu, force=force, relu_after=relu_after, scratch=scratch, gelu=gelu,
         **kw)


def conv2d_nhwc(x, wp, bias, out, B, H, W, Cin, Cout, KH, KW, pad, relu, big_tile=False,
                tile=None):
    """tile: None (library default: 64x64), "128x64" or "128" (sweeps)."""
    tflag = {None: 0, "128": GEMM_FORCE_TILE, "128x64": GEMM_FORCE_TILE128x64}[tile]
    name = ("k_gemm_tile<128,64,64,32,A_CONV>" if tile == "128x64" else
            "k_gemm_tile<128,128,64,64,A_CONV>" if tile == "128" else
            "k_gemm_tile<64,64,32,32,A_CONV>")
    flops = 2.0 * B * H * W * Cout * KH * KW * Cin
    nbytes = 4.0 * (B * H * W * (Cin + Cout) + Cout * KH * KW * Cin)
    _check(_launch(name, flops, nbytes, lambda: lib().pn_conv2d_nhwc_f32(
        _ptr(x), _ptr(wp), _ptr(bias), _ptr(out), B, H, W, Cin, Cout, KH, KW, pad, int(relu),
        (GEMM_FORCE_TILE if big_tile else 0) | tflag | _reserve_flag(),
        _stream())), "pn_conv2d_nhwc_f32")


def conv2d_ex(x, wp, bias, res, out, B, H, W, Cin, Cout, KH, KW, stride, pad, relu=False,
              relu_after=False, scratch=None):
    """General channel-last convolution (stride, residual, ReLU before / after it)."""
    Ho, Wo = (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1
    flops = 2.0 * B * Ho * Wo * Cout * KH * KW * Cin
    nbytes = 4.0 * (B * (H * W * Cin + Ho * Wo * Cout * (2 if res is not None else 1))
                    + Cout * KH * KW * Cin)
    flags = (GEMM_RELU if relu else 0) | (GEMM_RELU_AFTER_RES if relu_after else 0) | \
        _reserve_flag()
    _check(_launch("k_gemm_tile<64,64,32,32,A_CONV>", flops, nbytes,
                   lambda: lib().pn_conv2d_nhwc_ex_f32(
                       _ptr(x), _ptr(wp), _ptr(bias), _ptr(res), _ptr(out), B, H, W, Cin, Cout,
                       KH, KW, stride, pad, flags, _ptr(scratch),
                       scratch.numel() if scratch is not None else 0, _stream())),
           "pn_conv2d_nhwc_ex_f32")


def stem7x7s2(img, wp, bias, out, B, H, W):
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    _check(_launch("k_stem7x7s2", 2.0 * B * Ho * Wo * 64 * 147,
                   4.0 * B * (3 * H * W + Ho * Wo * 64),
                   lambda: lib().pn_stem7x7s2_f32(_ptr(img), _ptr(wp), _ptr(bias), _ptr(out), B,
                                                  H, W, _reserve_flag(), _stream())),
           "pn_stem7x7s2_f32")


def maxpool3x3s2(x, out, B, H, W, Cc):
    _check(lib().pn_maxpool3x3s2_nhwc_f32(_ptr(x), _ptr(out), B, H, W, Cc, _stream()),
           "pn_maxpool3x3s2_nhwc_f32")


def _winograd_weights_dev(w, form):
    """The device form of the two transforms below (one HIP kernel, no BLAS call)."""
    w = w.contiguous()
    co, ci = w.shape[0], w.shape[1]
    U = torch.empty((16 if form == 2 else 36), co, ci, device=w.device, dtype=torch.float32)
    _check(lib().pn_winograd_weights_f32(_ptr(w), _ptr(U), co, ci, form, _stream()),
           "pn_winograd_weights_f32")
    return U


def winograd_weights(w):
    """conv weight [Cout][Cin][3][3] (fp32) -> U [16][Cout][Cin] = G g G^T.  Device tensors go
    through `pn_winograd_weights_f32`; host tensors (tests, tools) through the einsum below."""
    if w.is_cuda:
        return _winograd_weights_dev(w, 2)
    G = torch.tensor([[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]],
                     dtype=torch.float64, device=w.device)
    U = torch.einsum("ik,ockl,jl->ijoc", G, w.double(), G)
    return U.reshape(16, w.shape[0], w.shape[1]).float().contiguous()


def winograd43_weights(w):
    """conv weight [Cout][Cin][3][3] -> U [36][Cout][Cin] = G g G^T for F(4x4, 3x3)."""
    if w.is_cuda:
        return _winograd_weights_dev(w, 4)
    G = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6],
                      [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]],
                     dtype=torch.float64, device=w.device)
    U = torch.einsum("ik,ockl,jl->ijoc", G, w.double(), G)
    return U.reshape(36, w.shape[0], w.shape[1]).float().contiguous()


def conv3x3_winograd43(x, U, bias, out, V, Mbuf, B, H, W, Cin, Cout, relu):
    """F(4x4,3x3) form; V / Mbuf: 36 * B*ceil(H/4)*ceil(W/4) * Cin / Cout floats."""
    T = B * ((H + 3) // 4) * ((W + 3) // 4)
    _check(lib().pn_winograd_f43_input_f32(_ptr(x), _ptr(V), B, H, W, Cin, _stream()),
           "pn_winograd_f43_input_f32")
    gemm(V, U, Mbuf, M=T, N=Cout, K=Cin, lda=Cin, ldw=Cin, ldc=Cout, batch=36, sA=T * Cin,
         sW=Cout * Cin, sC=T * Cout)
    _check(lib().pn_winograd_f43_output_f32(_ptr(Mbuf), _ptr(bias), _ptr(out), B, H, W, Cout,
                                            int(relu), _stream()), "pn_winograd_f43_output_f32")


def conv3x3_winograd(x, U, bias, out, V, Mbuf, B, H, W, Cin, Cout, relu):
    """3x3 pad-1 stride-1 convolution as Winograd F(2x2,3x3): input transform, one batched
    GEMM over the 16 transform positions, output transform.  V / Mbuf: scratch of
    16 * B*ceil(H/2)*ceil(W/2) * Cin / Cout floats."""
    T = B * ((H + 1) // 2) * ((W + 1) // 2)
    _check(lib().pn_winograd_f23_input_f32(_ptr(x), _ptr(V), B, H, W, Cin, _stream()),
           "pn_winograd_f23_input_f32")
    gemm(V, U, Mbuf, M=T, N=Cout, K=Cin, lda=Cin, ldw=Cin, ldc=Cout, batch=16, sA=T * Cin,
         sW=Cout * Cin, sC=T * Cout)
    _check(lib().pn_winograd_f23_output_f32(_ptr(Mbuf), _ptr(bias), _ptr(out), B, H, W, Cout,
                                            int(relu), _stream()), "pn_winograd_f23_output_f32")


def layernorm(x, gamma, beta, out, eps=1e-5):
    rows = x.numel() // 256
    _check(lib().pn_layernorm_f32(_ptr(x), _ptr(gamma), _ptr(beta), _ptr(out), rows, 256,
                                  eps, _stream()), "pn_layernorm_f32")


def linear_res_ln(x, weight, bias, res, gamma, beta, out, eps=1e-5):
    """out = LayerNorm(res + x @ weight.T + bias) * gamma + beta, 256 output columns, one
    launch (csrc/gemm_ln.hip); bit for bit linear(..., res=res) followed by layernorm()."""
    M, ldx = _rowmajor(x)
    N, ldw = _rowmajor(weight)
    Mr, ldr = _rowmajor(res)
    Mo, ldo = _rowmajor(out)
    K = x.shape[1]
    assert Mr == M and Mo == M and N == 256 and weight.shape[1] == K
    # operands of the fused op once each: x, W, residual in, normalised rows out
    nbytes = 4.0 * (M * K + N * K + 2 * M * N)
    _check(_launch("k_gemm_rowln", 2.0 * M * N * K, nbytes,
                   lambda: lib().pn_linear_res_ln_f32(_ptr(x), ldx, _ptr(weight), ldw, _ptr(bias),
                                                      _ptr(res), ldr, _ptr(gamma), _ptr(beta),
                                                      _ptr(out), ldo, M, N, K, eps, _stream()),
                   meta=(M, N, K, 1, False)), "pn_linear_res_ln_f32")


def s3_floats(rows, K):
    """Size of the S3 operand of a [rows x K] matrix, in float32 elements (arena carving)."""
    return ((rows + 31) // 32) * (K // 16) * 768


def pos8(pos):
    """The `out + pos` table of gemm_s3 in the order its epilogue reads ("P8",
    csrc/gemm_s3.hip): [rows / 32][cols / 16][2][32][8] from fp32 rows [rows][cols]."""
    rows, cols = pos.shape
    rb = (rows + 31) // 32
    p = pos.new_zeros(rb * 32, cols)
    p[:rows] = pos
    return p.view(rb, 32, cols // 16, 2, 8).permute(0, 2, 3, 1, 4).contiguous()


def s3_split(x, out, add=None):
    """out (an S3 buffer, any dtype, >= s3_floats(rows, K) * 4 bytes) = split(x + add[row % len(add)])
    for 2-D fp32 rows x: the three-bf16-plane operand format of gemm_s3 (include/pairnet_hip.h)."""
    rows, ld = _rowmajor(x)
    K = x.shape[1]
    assert out.numel() * out.element_size() >= s3_floats(rows, K) * 4
    ar = 0
    if add is not None:
        ar, lda = _rowmajor(add)
        assert lda == K and add.shape[1] == K
    _check(_launch("k_s3_split", 0.0, 10.0 * rows * K,
                   lambda: lib().pn_s3_split_f32(_ptr(x), ld, _ptr(add), ar, _ptr(out), rows, K,
                                                 _stream())), "pn_s3_split_f32")


def s3_join(s3, out):
    """out (2-D fp32 rows) = the exact fp32 values of the S3 operand."""
    rows, ld = _rowmajor(out)
    _check(_launch("k_s3_join", 0.0, 10.0 * rows * out.shape[1],
                   lambda: lib().pn_s3_join_f32(_ptr(s3), _ptr(out), ld, rows, out.shape[1],
                                                _stream())), "pn_s3_join_f32")


def gemm_s3(a, w, M, N, K, *, bias=None, relu=False, out=None, out_s3=None, out_s3_pos=None,
            pos=None, a2=None, a2_from_col=0, res_s3=None, gamma=None, beta=None, eps=1e-5,
            tile96=False, tile192=False, gelu=False, res=None, relu_after=False):
    """fp32 GEMM on the bf16 matrix pipe from pre-split operands (csrc/gemm_s3.hip):
    out = act(a @ w.T + bias), or LayerNorm(a @ w.T + bias + res) * gamma + beta (N == 256).
    a, a2, w, res_s3, out_s3, out_s3_pos are S3 buffers; out is 2-D fp32 rows."""
    d = GemmS3Desc()
    d.A, d.A2, d.a2_from_col = _ptr(a), _ptr(a2), a2_from_col
    d.W, d.bias = _ptr(w), _ptr(bias)
    d.M, d.N, d.K, d.relu = M, N, K, int(relu)
    if out is not None:
        Mo, ldc = _rowmajor(out)
        assert Mo == M and out.shape[1] >= N
        d.C, d.ldc = _ptr(out), ldc
    d.CS, d.CS_pos = _ptr(out_s3), _ptr(out_s3_pos)
    if out_s3_pos is not None:
        # pos: the table in P8 order (pos8()), pos_rows its true row count
        pos, pr = pos
        assert pos.numel() == (pr + 31) // 32 * 32 * N
        d.pos, d.pos_rows = _ptr(pos), pr
    d.res_s3, d.gamma, d.beta, d.eps = _ptr(res_s3), _ptr(gamma), _ptr(beta), eps
    d.flags = (1 if tile96 else 0) | (2 if tile192 else 0)     # PN_GEMM_S3_TILE96 / _TILE192
    d.act = 3 if relu_after else 2 if gelu else 0
    if res is not None:              # fp32 rows added after the activation (a block's shortcut)
        Mr, ldr = _rowmajor(res)
        assert Mr == M and res.shape[1] >= N
        d.res, d.ldres = _ptr(res), ldr
    nbytes = 6.0 * (M * K + N * K) + (4.0 * M * N if out is not None else 0.0) + \
        6.0 * M * N * ((out_s3 is not None) + (out_s3_pos is not None) + (res_s3 is not None))
    name = "k_gemm_s3<ln>" if gamma is not None else "k_gemm_s3"    # (both tile forms)
    _check(_launch(name, 2.0 * M * N * K, nbytes, lambda: lib().pn_gemm_s3_f32(C.byref(d), _stream()),
                   meta=(M, N, K, 1, False)), "pn_gemm_s3_f32")


def layernorm_rows(x, gamma, beta, out, eps=1e-5):
    """LayerNorm over the last dim of 2-D row views (any C % 4 == 0, C <= 3072)."""
    rows, ldx = _rowmajor(x)
    ro, ldo = _rowmajor(out)
    assert ro == rows and out.shape[1] == x.shape[1]
    _check(_launch("k_ln_rows", 0.0, 8.0 * rows * x.shape[1],
                   lambda: lib().pn_layernorm_rows_f32(_ptr(x), ldx, _ptr(gamma), _ptr(beta),
                                                       _ptr(out), ldo, rows, x.shape[1], eps,
                                                       _stream())), "pn_layernorm_rows_f32")


def layernorm_rows_s3(x, gamma, beta, out_s3, eps=1e-5):
    """LayerNorm over the last dim of 2-D rows, written as an S3 operand (three bf16 planes:
    gemm_s3's A) instead of fp32 rows; C % 16 == 0."""
    rows, ldx = _rowmajor(x)
    Cc = x.shape[1]
    assert out_s3.numel() >= s3_floats(rows, Cc)
    _check(_launch("k_ln_rows<s3>", 0.0, 10.0 * rows * Cc,
                   lambda: lib().pn_layernorm_rows_s3_f32(_ptr(x), ldx, _ptr(gamma), _ptr(beta),
                                                          _ptr(out_s3), rows, Cc, eps, _stream())),
           "pn_layernorm_rows_s3_f32")


def patch_merge_ln(x, gamma, beta, out, B, H, W, C, eps=1e-5):
    """x [B][H*W][C] -> out [B][ceil(H/2)*ceil(W/2)][4C] = LayerNorm of the 2x2 neighbourhoods
    concatenated neighbour-major ((row*2+col)*C + c); gamma / beta in that order."""
    _check(_launch("k_ln_rows<merge>", 0.0, 8.0 * B * H * W * C,
                   lambda: lib().pn_patch_merge_ln_f32(_ptr(x), _ptr(gamma), _ptr(beta), _ptr(out),
                                                       B, H, W, C, eps, _stream())),
           "pn_patch_merge_ln_f32")


def patch_im2col4(img, out, B, H, W):
    """NCHW RGB image -> [B*ceil(H/4)*ceil(W/4)][64] rows of 4x4 patches (c*16+ky*4+kx, 48
    columns + 16 zeros); pixels beyond H / W read as zero."""
    _check(_launch("k_patch_im2col", 0.0, 4.0 * B * H * W * 3 * 2.4,
                   lambda: lib().pn_patch_im2col4_f32(_ptr(img), _ptr(out), B, H, W, _stream())),
           "pn_patch_im2col4_f32")


def window_attention(qkv, qkv_bias, table, out, B, H, W, C, heads, ws, shift):
    """(Shifted-)window attention on the [B*H*W][3C] qkv rows -> out [B*H*W][C];
    `table` is the relative position bias table transposed to [heads][(2ws-1)^2]."""
    n, ldq = _rowmajor(qkv)
    no, ldo = _rowmajor(out)
    assert n == no == B * H * W and qkv.shape[1] == 3 * C
    hp, wp = -(-H // ws) * ws, -(-W // ws) * ws
    flops = 4.0 * B * hp * wp * (ws * ws) * C
    _check(_launch("k_window_attn", flops, 16.0 * n * C,
                   lambda: lib().pn_window_attention_f32(_ptr(qkv), ldq, _ptr(qkv_bias),
                                                         _ptr(table), _ptr(out), ldo, B, H, W, C,
                                                         heads, ws, shift, 32 ** -0.5, _stream())),
           "pn_window_attention_f32")


def window_attention_s3(qkv, qkv_bias, table, out_s3, B, H, W, C, heads, ws, shift):
    """window_attention() with the output as an S3 operand buffer (gemm_s3's A) instead of rows."""
    n, ldq = _rowmajor(qkv)
    assert n == B * H * W and qkv.shape[1] == 3 * C and out_s3.numel() >= s3_floats(n, C)
    hp, wp = -(-H // ws) * ws, -(-W // ws) * ws
    flops = 4.0 * B * hp * wp * (ws * ws) * C
    _check(_launch("k_window_attn", flops, 18.0 * n * C,
                   lambda: lib().pn_window_attention_s3_f32(_ptr(qkv), ldq, _ptr(qkv_bias), _ptr(table),
                                                            _ptr(out_s3), B, H, W, C, heads, ws, shift,
                                                            32 ** -0.5, _stream())),
           "pn_window_attention_s3_f32")


def groupnorm_nblk(hw):
    return lib().pn_groupnorm_nblk(hw)


def groupnorm_nhwc(x, gamma, beta, out, partials, B, HW, G, relu, x_bstride, y_bstride,
                   eps=1e-5):
    _check(lib().pn_groupnorm_nhwc_f32(
        _ptr(x), _ptr(gamma), _ptr(beta), _ptr(out), _ptr(partials, torch.float64), B, HW,
        256, G, eps, int(relu), x_bstride, y_bstride, _stream()), "pn_groupnorm_nhwc_f32")


def ffn_scratch_floats(M, hidden):
    return lib().pn_ffn_scratch_floats(M, hidden)


def ffn_ln(x, W1, b1, W2, b2, gamma, beta, out, scratch, M, hidden, eps=1e-5, post=None):
    """post = (gamma2, beta2, out2): a second LayerNorm of the result into out2."""
    g2, b2n, y2 = post if post is not None else (None, None, None)
    _check(_launch("k_ffn_partial+k_reduce_ln", 4.0 * M * 256 * hidden,
                   4.0 * (2 * 256 * hidden + 2 * M * 256),
                   lambda: lib().pn_ffn_ln2_f32(_ptr(x), _ptr(W1), _ptr(b1), _ptr(W2), _ptr(b2),
                                                _ptr(gamma), _ptr(beta), _ptr(out), _ptr(g2),
                                                _ptr(b2n), _ptr(y2), _ptr(scratch), M, 256, hidden,
                                                eps, _stream())), "pn_ffn_ln2_f32")


def l2normalize(x, out, eps=1e-12):
    _check(lib().pn_l2normalize_f32(_ptr(x), _ptr(out), x.numel() // 256, 256, eps,
                                    _stream()), "pn_l2normalize_f32")


MSDA_PERSISTENT, MSDA_PERSISTENT_BATCHED, MSDA_LOW_OCCUPANCY = 1, 2, 4   # PN_MSDA_* launch forms (A/B)


MSDA_S3_OUT = 8     # PN_MSDA_S3_OUT


def msda(value, ld_value, offaw, ld_offaw, out, B, shapes, flags=0, s3_out=False):
    """`s3_out`: `out` is an S3 operand buffer [B * N x 256] (gemm_s3's A) instead of fp32 rows."""
    if s3_out:
        flags |= MSDA_S3_OUT
    L = len(shapes)
    hs = (_i32 * L)(*[s[0] for s in shapes])
    ws = (_i32 * L)(*[s[1] for s in shapes])
    n = sum(h * w for h, w in shapes)
    # algorithmic bytes: value read + offsets/logits read + output write
    nbytes = 4.0 * B * n * (256 + 8 * L * 4 * 3 + 256)
    _check(_launch("k_msda", 2.0 * B * n * 8 * L * 4 * 4 * 32 * 2, nbytes,
                   lambda: lib().pn_msda_ex_f32(_ptr(value), ld_value, _ptr(offaw), ld_offaw,
                                                _ptr(out), B, L, hs, ws, flags, _stream())),
           "pn_msda_ex_f32")


def msda_loc(value, ld_value, spatial_shapes, level_start_index, loc, aw, out, B, N, Nq, L):
    nbytes = 4.0 * B * (N * 256 + Nq * (8 * L * 4 * 3 + 256))
    _check(_launch("k_msda_loc", 2.0 * B * Nq * 8 * L * 4 * 4 * 32 * 2, nbytes,
                   lambda: lib().pn_msda_loc_f32(_ptr(value), ld_value,
                                                 _ptr(spatial_shapes, torch.int64),
                                                 _ptr(level_start_index, torch.int64), _ptr(loc),
                                                 _ptr(aw), _ptr(out), B, N, Nq, L, _stream())),
           "pn_msda_loc_f32")


def msda_bwd(value, ld_value, spatial_shapes, level_start_index, loc, aw, grad_out, grad_value,
             grad_loc, grad_aw, B, N, Nq, L):
    _check(lib().pn_msda_bwd_f32(_ptr(value), ld_value, _ptr(spatial_shapes, torch.int64),
                                 _ptr(level_start_index, torch.int64), _ptr(loc), _ptr(aw),
                                 _ptr(grad_out), _ptr(grad_value), _ptr(grad_loc), _ptr(grad_aw),
                                 B, N, Nq, L, _stream()), "pn_msda_bwd_f32")


def gather_probe(lines, idx, out, workgroups, line_mask):
    _check(lib().pn_gather_probe_f32(_ptr(lines), _ptr(idx, torch.int32), _ptr(out), workgroups,
                                     line_mask, _stream()), "pn_gather_probe_f32")


def sine_pe(out, add, h, w, C_=256, temperature=10000.0, offset=0.0, valid=None):
    """valid = (valid_h, valid_w) of a padded map (default: the whole map)."""
    vh, vw = valid if valid is not None else (h, w)
    _check(lib().pn_sine_pe_valid_f32(_ptr(out), _ptr(add), h, w, vh, vw, C_, temperature, offset,
                                      _stream()), "pn_sine_pe_valid_f32")


def bilinear_nhwc(x, out, B, hi, wi, ho, wo, Cc, accumulate, in_bstride, out_bstride):
    _check(lib().pn_bilinear_nhwc_f32(_ptr(x), _ptr(out), B, hi, wi, ho, wo, Cc,
                                      int(accumulate), in_bstride, out_bstride, _stream()),
           "pn_bilinear_nhwc_f32")


def bilinear_planar(x, out, P, hi, wi, ho, wo):
    _check(lib().pn_bilinear_planar_f32(_ptr(x), _ptr(out), P, hi, wi, ho, wo, _stream()),
           "pn_bilinear_planar_f32")


def bilinear_planar_gt0(x, out, P, hi, wi, ho, wo):
    _check(lib().pn_bilinear_planar_gt0_u8(_ptr(x), _ptr(out, torch.uint8), P, hi, wi, ho,
                                           wo, _stream()), "pn_bilinear_planar_gt0_u8")


def mask_pack(logits, bits, rowall, R, Nk):
    _check(lib().pn_mask_pack(_ptr(logits), _ptr(bits, torch.int32),
                              _ptr(rowall, torch.int32), R, Nk, _stream()), "pn_mask_pack")


def mask_pack_stencil(logits4, bits, rowall, R, hi, wi, ho, wo):
    _check(lib().pn_mask_pack_stencil(_ptr(logits4), _ptr(bits, torch.int32),
                                      _ptr(rowall, torch.int32), R, hi, wi, ho, wo, _stream()),
           "pn_mask_pack_stencil")


def mask_stencil_gemm(me, rows, bits, rowall, B, Q, hi, wi, ho, wo, K=256):
    """bits / rowall of one layer's attention mask from the mask embedding `me` [B*Q, K] and the
    level's stencil rows [B, 4*ho*wo, K] in one launch (csrc/gemm.hip k_gemm_stencil)."""
    Nk = ho * wo
    _check(_launch("k_gemm_stencil", 2.0 * B * Q * 4 * Nk * K,
                   4.0 * B * (Q * K + 4 * Nk * K) + B * Q * Nk / 8.0,
                   lambda: lib().pn_mask_stencil_gemm_f32(
                       _ptr(me), me.stride(0), Q * me.stride(0), _ptr(rows), rows.stride(-2),
                       rows.stride(0) if rows.dim() == 3 else 4 * Nk * rows.stride(-2),
                       _ptr(bits, torch.int32), _ptr(rowall, torch.int32), B, Q, Nk, K, hi, wi,
                       ho, wo, _reserve_flag(), _stream()),
                   meta=(Q, 4 * Nk, K, B, False)), "pn_mask_stencil_gemm_f32")


def bilinear_stencil_rows(x, out, B, hi, wi, ho, wo, Cc, in_bstride, out_bstride):
    _check(lib().pn_bilinear_stencil_rows_f32(_ptr(x), _ptr(out), B, hi, wi, ho, wo, Cc,
                                              in_bstride, out_bstride, _stream()),
           "pn_bilinear_stencil_rows_f32")


def attn_scratch_floats(B, Q, Nk):
    return lib().pn_attn_scratch_floats(B, Q, Nk)


def attention(q, ldq, k, ldk, v, ldv, bits, rowall, out, ldo, scratch, B, Q, Nk, scale):
    # QK^T and PV: 2 x (2 * B * 8 heads * Q * Nk * 32) flops; bytes: q, k, v, out rows once
    _check(_launch("k_attn_chunk", 8.0 * B * 8 * Q * Nk * 32 / 2,
                   4.0 * B * 256 * (2 * Q + 2 * Nk),
                   lambda: lib().pn_attention_f32(
                       _ptr(q), ldq, _ptr(k), ldk, _ptr(v), ldv, _ptr(bits, torch.int32),
                       _ptr(rowall, torch.int32), _ptr(out), ldo, _ptr(scratch), B, Q, Nk, scale,
                       _stream())), "pn_attention_f32")


def ppn_front(sub_embed, obj_embed, w1, b1, imp_raw, c1, B, Q, eps=1e-12):
    _check(lib().pn_ppn_front_f32(_ptr(sub_embed), _ptr(obj_embed), _ptr(w1), _ptr(b1),
                                  _ptr(imp_raw), _ptr(c1), B, Q, eps, _stream()),
           "pn_ppn_front_f32")


def mlearner_first(x, w1, b1, out, B, S):
    _check(lib().pn_mlearner_first_f32(_ptr(x), _ptr(w1), _ptr(b1), _ptr(out), B, S, 64,
                                       _stream()), "pn_mlearner_first_f32")


def mlearner_last(x, w3, b3, out, B, S):
    _check(lib().pn_mlearner_last_f32(_ptr(x), _ptr(w3), _ptr(b3), _ptr(out), B, S, 64,
                                      _stream()), "pn_mlearner_last_f32")


def topk_pairs(scores, idx, sub, obj, B, Q, k, pair=None):
    _check(lib().pn_topk_pairs(_ptr(scores), _ptr(idx, torch.int64), _ptr(sub, torch.int64),
                               _ptr(obj, torch.int64), _ptr(pair, torch.int64), B, Q, k,
                               _stream()), "pn_topk_pairs")


def topk(scores, idx, quot, rem, B, n, div, k):
    _check(lib().pn_topk_f32(_ptr(scores), _ptr(idx, torch.int64), _ptr(quot, torch.int64),
                             _ptr(rem, torch.int64), B, n, div, k, _stream()), "pn_topk_f32")


def topk_strided(scores, elem_stride, row_stride, idx, quot, rem, B, n, div, k):
    _check(lib().pn_topk_strided_f32(_ptr(scores), elem_stride, row_stride, _ptr(idx, torch.int64),
                                     _ptr(quot, torch.int64), _ptr(rem, torch.int64), B, n, div,
                                     k, _stream()), "pn_topk_strided_f32")


def gather_rows(x, index, out, B, rows_in, rows_out, length):
    _check(lib().pn_gather_rows_f32(_ptr(x), _ptr(index, torch.int64), _ptr(out), B, rows_in,
                                    rows_out, length, _stream()), "pn_gather_rows_f32")


def cls_argmax(logits, label, score, rows, Cc, label_offset=0):
    _check(lib().pn_cls_argmax_f32(_ptr(logits), _ptr(label, torch.int64), _ptr(score), rows,
                                   Cc, label_offset, _stream()), "pn_cls_argmax_f32")


def rel_dists(logits, out, rows, Cc):
    _check(lib().pn_rel_dists_f32(_ptr(logits), _ptr(out), rows, Cc, _stream()),
           "pn_rel_dists_f32")


def softmax_fg(logits, probs, fg, rows, Cc):
    _check(lib().pn_softmax_fg_f32(_ptr(logits), _ptr(probs), _ptr(fg), rows, Cc, _stream()),
           "pn_softmax_fg_f32")


def row_argmax(x, idx, rows, n):
    _check(lib().pn_row_argmax_f32(_ptr(x), _ptr(idx, torch.int64), rows, n, _stream()),
           "pn_row_argmax_f32")


def triplet_finish(s_label, o_label, probs, tri, rem, labels, r_labels, r_scores, r_dists, k,
                   Cc):
    i64 = torch.int64
    _check(lib().pn_triplet_finish(_ptr(s_label, i64), _ptr(o_label, i64), _ptr(probs),
                                   _ptr(tri, i64), _ptr(rem, i64), _ptr(labels, i64),
                                   _ptr(r_labels, i64), _ptr(r_scores), _ptr(r_dists), k, Cc,
                                   _stream()),
           "pn_triplet_finish")


def panoptic(masks, labels, remap, seg, area, n, HW):
    _check(lib().pn_panoptic_f32(_ptr(masks), _ptr(labels, torch.int64),
                                 _ptr(remap, torch.int32), _ptr(seg, torch.int64),
                                 _ptr(area, torch.int32), n, HW, _stream()), "pn_panoptic_f32")


# argmax / area-filter rounds enqueued up front.  The reference's loop can take at most
# three: the first round merges duplicate stuff classes into their first occurrence (the
# duplicates end with area 0 and are dropped), the second counts the survivors without the
# merge (so the first occurrence can lose the merged area and fall out), and since from
# then on dropping segments only ever grows the others, the third finds nothing to drop.
PAN_ROUNDS = 4


def panoptic_state_bytes():
    return lib().pn_panoptic_state_bytes()


def panoptic_device(masks, labels, scores, Q, num_classes, hi, wi, ho, wo, state, up, area, seg,
                    rounds=None):
    rounds = PAN_ROUNDS if rounds is None else rounds
    _check(lib().pn_panoptic_device_f32(
        _ptr(masks), _ptr(labels, torch.int64), _ptr(scores), Q, num_classes, hi, wi, ho, wo,
        _ptr(state, torch.uint8), _ptr(up), _ptr(area, torch.int32), _ptr(seg, torch.int64),
        rounds, _stream()), "pn_panoptic_device_f32")


def panoptic_continue(state, up, area, seg, ho, wo, rounds=None):
    rounds = PAN_ROUNDS if rounds is None else rounds
    _check(lib().pn_panoptic_continue_f32(
        _ptr(state, torch.uint8), _ptr(up), _ptr(area, torch.int32), _ptr(seg, torch.int64),
        ho, wo, rounds, _stream()), "pn_panoptic_continue_f32")


def pack_triplets(labels, r_dists, sub_pos, obj_pos, rec, R, C1):
    i64 = torch.int64
    _check(lib().pn_pack_triplets_f32(_ptr(labels, i64), _ptr(r_dists), _ptr(sub_pos, i64),
                                      _ptr(obj_pos, i64), _ptr(rec), R, C1, _stream()),
           "pn_pack_triplets_f32")


def copy_stream(src, dst, wgs=16):
    """dst <- src (same dtype / shape, both contiguous); `dst` may be a PINNED host tensor: the
    kernel writes it over PCIe from `wgs` workgroups, on torch's current stream."""
    if src.dtype != dst.dtype or src.numel() != dst.numel() or not (
            src.is_contiguous() and dst.is_contiguous()):
        raise RuntimeError("copy_stream takes contiguous tensors of one dtype and size")
    if not src.is_cuda or not (dst.is_cuda or dst.is_pinned()):
        raise RuntimeError("copy_stream: device source, device or pinned-host destination")
    nbytes = src.numel() * src.element_size()
    if nbytes:
        _check(lib().pn_copy_stream(src.data_ptr(), dst.data_ptr(), nbytes, int(wgs), _stream()),
               "pn_copy_stream")


def pack_bool_bits(src, bits):
    """bits[(n + 7) // 8] uint8 <- the n elements of the bool / uint8 device tensor `src` (byte i
    bit j = element 8 i + j), on torch's current stream."""
    n = src.numel()
    if not src.is_cuda or not bits.is_cuda or src.element_size() != 1 or not src.is_contiguous() \
            or bits.dtype != torch.uint8 or bits.numel() < (n + 7) // 8 or not bits.is_contiguous():
        raise RuntimeError("pack_bool_bits: contiguous 1-byte device source, uint8 device "
                           "destination of (n + 7) // 8 bytes")
    if n:
        _check(lib().pn_pack_bool_bits(src.data_ptr(), bits.data_ptr(), n, _stream()),
               "pn_pack_bool_bits")


def unpack_bits_host(bits, out, threads=4):
    """The host half: `out` (a HOST bool / uint8 tensor of n elements) <- the packed bytes in
    the host tensor `bits` (once their copy has completed).  No GPU call; ctypes drops the GIL
    for its duration."""
    n = out.numel()
    if bits.is_cuda or out.is_cuda or out.element_size() != 1 or bits.dtype != torch.uint8 \
            or not (bits.is_contiguous() and out.is_contiguous()) or bits.numel() < (n + 7) // 8:
        raise RuntimeError("unpack_bits_host: contiguous host tensors, uint8 bits of "
                           "(n + 7) // 8 bytes")
    if n:
        _check(lib().pn_unpack_bits_host(bits.data_ptr(), out.data_ptr(), n, int(threads)),
               "pn_unpack_bits_host")


def preprocess_u8(img, H, W, out, Hn, Wn, Hp, Wp, mean, stdinv, to_rgb):
    """mean / stdinv: 3 host floats each, in OUTPUT channel order."""
    m = (_f32 * 3)(*[float(v) for v in mean])
    s = (_f32 * 3)(*[float(v) for v in stdinv])
    _check(lib().pn_preprocess_u8_f32(_ptr(img, torch.uint8), H, W, _ptr(out), Hn, Wn, Hp, Wp,
                                      m, s, int(to_rgb), _stream()), "pn_preprocess_u8_f32")


def pack_mask_bits(masks_u8, words, rows, HW):
    _check(lib().pn_pack_mask_bits(_ptr(masks_u8, torch.uint8), _ptr(words, torch.int64), rows, HW,
                                   _stream()), "pn_pack_mask_bits")


def mask_iou_counts(pred_words, P, gt_words, G, nwords, inter, area_p, area_g):
    _check(lib().pn_mask_iou_counts(_ptr(pred_words, torch.int64), P, _ptr(gt_words, torch.int64),
                                    G, nwords, _ptr(inter, torch.int32),
                                    _ptr(area_p, torch.int32), _ptr(area_g, torch.int32),
                                    _stream()), "pn_mask_iou_counts")


def pred_triplets(labels, r_dists, triplets, scores, R, C1):
    _check(lib().pn_pred_triplets(_ptr(labels, torch.int64), _ptr(r_dists),
                                  _ptr(triplets, torch.int32), _ptr(scores), R, C1, _stream()),
           "pn_pred_triplets")


def mask_or_rows(words, a, b, out, rows, nwords):
    _check(lib().pn_mask_or_rows(_ptr(words, torch.int64), _ptr(a, torch.int32),
                                 _ptr(b, torch.int32), _ptr(out, torch.int64), rows, nwords,
                                 _stream()), "pn_mask_or_rows")


def triplet_match(ptrip, gtrip, P, G, inter, area_p, area_g, ld_inter, ps, po, gs, go, thr, phrdet,
                  ignore_rel, match):
    i32 = torch.int32
    _check(lib().pn_triplet_match(_ptr(ptrip, i32), _ptr(gtrip, i32), P, G, _ptr(inter, i32),
                                  _ptr(area_p, i32), _ptr(area_g, i32), ld_inter, _ptr(ps, i32),
                                  _ptr(po, i32), _ptr(gs, i32), _ptr(go, i32), float(thr),
                                  int(phrdet), int(ignore_rel), _ptr(match, torch.uint8),
                                  _stream()), "pn_triplet_match")


def triplet_match_boxes(ptrip, gtrip, P, G, pbox, ldp, gbox, ldg, ps, po, gs, go, thr, phrdet,
                        ignore_rel, match):
    i32 = torch.int32
    _check(lib().pn_triplet_match_boxes(
        _ptr(ptrip, i32), _ptr(gtrip, i32), P, G, _ptr(pbox), ldp, _ptr(gbox), ldg, _ptr(ps, i32),
        _ptr(po, i32), _ptr(gs, i32), _ptr(go, i32), float(thr), int(phrdet), int(ignore_rel),
        _ptr(match, torch.uint8), _stream()), "pn_triplet_match_boxes")


# ---- box trunk glue (csrc/detr.hip) ----
def zero_rows(x, valid_u8, out, B, rows, Cc, ld=None, per_image=False):
    """valid_u8: [rows] (shared) or, with per_image, [B][rows]; rows of Cc floats at stride ld."""
    _check(lib().pn_zero_rows_f32(_ptr(x), _ptr(valid_u8, torch.uint8), _ptr(out), B, rows, Cc,
                                  Cc if ld is None else ld, rows if per_image else 0, _stream()),
           "pn_zero_rows_f32")


def token_sampling(offaw, ld, valid_ratios, loc, aw, B, shapes):
    L = len(shapes)
    hs = (C.c_int32 * L)(*[h for h, _ in shapes])
    ws = (C.c_int32 * L)(*[w for _, w in shapes])
    _check(lib().pn_token_sampling_f32(_ptr(offaw), ld, _ptr(valid_ratios), _ptr(loc), _ptr(aw), B,
                                       L, hs, ws, _stream()), "pn_token_sampling_f32")


def sigmoid(x, out):
    _check(lib().pn_sigmoid_f32(_ptr(x), _ptr(out), x.numel(), _stream()), "pn_sigmoid_f32")


def box_pos_embed(unact, ref, emb, rows):
    _check(lib().pn_box_pos_embed_f32(_ptr(unact), _ptr(ref), _ptr(emb), rows, _stream()),
           "pn_box_pos_embed_f32")


def box_sampling(offaw, ld, ref, loc, aw, rows, L, valid_ratios=None, rows_per_image=0):
    _check(lib().pn_box_sampling_f32(_ptr(offaw), ld, _ptr(ref), _ptr(valid_ratios),
                                     rows_per_image, _ptr(loc), _ptr(aw), rows, L, _stream()),
           "pn_box_sampling_f32")


def box_refine(delta, ref_in, ref_out, rows):
    _check(lib().pn_box_refine_f32(_ptr(delta), _ptr(ref_in), _ptr(ref_out), rows, _stream()),
           "pn_box_refine_f32")


def query_score(logits, score, B, Nq, Cc):
    _check(lib().pn_query_score_f32(_ptr(logits), _ptr(score), B, Nq, Cc, _stream()),
           "pn_query_score_f32")


def box_triplets(s_cls, o_cls, s_box, o_box, det, labels, R, Cc, img_h, img_w, scale_factor,
                 rescale):
    sf = (C.c_float * 4)(*[float(v) for v in scale_factor])
    _check(lib().pn_box_triplets_f32(_ptr(s_cls), _ptr(o_cls), _ptr(s_box), _ptr(o_box), _ptr(det),
                                     _ptr(labels, torch.int64), R, Cc, float(img_h), float(img_w),
                                     sf, 1 if rescale else 0, _stream()), "pn_box_triplets_f32")


# ---- loss forward (csrc/loss.hip) ----------------------------------------------------------
def pan_masks(rgb, ids, cats, masks, sem=None):
    """rgb [H][W][3] uint8 (RGB panoptic PNG) -> masks [G][H][W] uint8 (`rgb2id(rgb) == ids[g]`)
    and optionally sem [H][W] int32 (category of the last listed owner, 255 = none)."""
    H, W, c = rgb.shape
    G = int(ids.shape[0])
    assert c == 3 and rgb.dtype == torch.uint8 and rgb.is_contiguous()
    assert masks.shape == (G, H, W) and masks.is_contiguous() and masks.dtype in (torch.uint8, torch.bool)
    _check(lib().pn_pan_masks_u8(_ptr(rgb, torch.uint8), _ptr(ids, torch.int32) if G else None,
                                 _ptr(cats, torch.int32) if (G and cats is not None) else None,
                                 _ptr(masks, masks.dtype) if G else None,
                                 _ptr(sem, torch.int32) if sem is not None else None, G, H, W,
                                 _stream()), "pn_pan_masks_u8")


def gt_mask_prepare(masks, out, H, W):
    """masks [G][h][w] bool / uint8 -> out [G][Ho][Wo] uint8: zero-padded to [H][W], then
    nearest-neighbour resized (psgtr.py:126-141)."""
    G, h, w = masks.shape
    assert masks.dtype in (torch.bool, torch.uint8) and out.dtype in (torch.bool, torch.uint8)
    assert out.shape[0] == G and masks.is_contiguous() and out.is_contiguous()
    _check(lib().pn_gt_mask_prepare_u8(_ptr(masks, masks.dtype), _ptr(out, out.dtype), G, h, w, H,
                                       W, out.shape[1], out.shape[2], _stream()),
           "pn_gt_mask_prepare_u8")


def point_sample(maps, pts, out):
    """maps [P][h][w] float32 or bool / uint8; pts [Np][2]; out [P][Np] (mmcv point_sample with
    one point set for all maps)."""
    P, h, w = maps.shape
    u8 = maps.dtype in (torch.bool, torch.uint8)
    _check(lib().pn_point_sample_f32(_ptr(maps, maps.dtype), int(u8), _ptr(pts), _ptr(out), P, h, w,
                                     pts.shape[0], _stream()), "pn_point_sample_f32")


def mask_match_cost(cls, gt_labels, pred_pts, gt_pts, cost, w_cls, w_mask, w_dice, dice_eps):
    Q, ncls = cls.shape
    G, Np = gt_pts.shape
    _check(lib().pn_mask_match_cost_f32(_ptr(cls), ncls, _ptr(gt_labels, torch.int64),
                                        _ptr(pred_pts), _ptr(gt_pts), _ptr(cost), Q, G, Np, w_cls,
                                        w_mask, w_dice, dice_eps, _stream()),
           "pn_mask_match_cost_f32")


def id_match_cost(sub, obj, rel, gt_sub, gt_obj, gt_rel, cost, w_sub, w_obj, w_rel):
    R, ncls = sub.shape
    _check(lib().pn_id_match_cost_f32(_ptr(sub), _ptr(obj), _ptr(rel), ncls, rel.shape[1],
                                      _ptr(gt_sub, torch.int64), _ptr(gt_obj, torch.int64),
                                      _ptr(gt_rel, torch.int64), _ptr(cost), R, gt_sub.shape[0],
                                      w_sub, w_obj, w_rel, _stream()), "pn_id_match_cost_f32")


def ce_mean(logits, target, class_weight, out, loss_weight):
    rows, ld = _rowmajor(logits)
    _check(lib().pn_ce_mean_f32(_ptr(logits), ld, _ptr(target, torch.int64), _ptr(class_weight),
                                _ptr(out), rows, logits.shape[1], loss_weight, _stream()),
           "pn_ce_mean_f32")


def seesaw_mean(logits, target, cum_samples, out, p, q, eps, loss_weight):
    rows, ld = _rowmajor(logits)
    _check(lib().pn_seesaw_mean_f32(_ptr(logits), ld, _ptr(target, torch.int64), _ptr(cum_samples),
                                    _ptr(out), rows, logits.shape[1], p, q, eps, loss_weight,
                                    _stream()), "pn_seesaw_mean_f32")


def ce_mean_grad(logits, target, class_weight, grad, loss_weight):
    rows, ld = _rowmajor(logits)
    rg, ldg = _rowmajor(grad)
    assert rg == rows and grad.shape[1] == logits.shape[1]
    _check(lib().pn_ce_mean_grad_f32(_ptr(logits), ld, _ptr(target, torch.int64), _ptr(class_weight),
                                     _ptr(grad), ldg, rows, logits.shape[1], loss_weight, _stream()),
           "pn_ce_mean_grad_f32")


def seesaw_mean_grad(logits, target, cum_samples, grad, p, q, eps, loss_weight):
    rows, ld = _rowmajor(logits)
    rg, ldg = _rowmajor(grad)
    assert rg == rows and grad.shape[1] == logits.shape[1]
    _check(lib().pn_seesaw_mean_grad_f32(_ptr(logits), ld, _ptr(target, torch.int64),
                                         _ptr(cum_samples), _ptr(grad), ldg, rows, logits.shape[1],
                                         p, q, eps, loss_weight, _stream()), "pn_seesaw_mean_grad_f32")


def bce_posw_mean_grad(logits, target, grad, loss_weight):
    assert grad.numel() == logits.numel()
    _check(lib().pn_bce_posw_mean_grad_f32(_ptr(logits), _ptr(target), _ptr(grad), logits.numel(),
                                           loss_weight, _stream()), "pn_bce_posw_mean_grad_f32")


# ---- backward building blocks of the Pair-Net tail (csrc/grad.hip; composed in grad.py) -----
def transpose(x, out, out_cols=None):
    """out[c][r] = x[r][c]; out [cols][ld >= out_cols], zero-filled from column rows on."""
    rows, ldi = _rowmajor(x)
    cols, ldo = _rowmajor(out)
    assert cols == x.shape[1]
    oc = out.shape[1] if out_cols is None else out_cols
    _check(lib().pn_transpose_f32(_ptr(x), ldi, _ptr(out), ldo, rows, cols, oc, _stream()),
           "pn_transpose_f32")


def colsum(x, out, accumulate=False):
    rows, ld = _rowmajor(x)
    cols = x.shape[1]
    assert out.numel() == cols and out.is_contiguous()
    scratch = torch.empty(min(64, rows // 512) * cols, device=x.device, dtype=torch.float32) \
        if rows >= 1024 else None               # tall matrices: row chunks on many workgroups
    _check(lib().pn_colsum_f32(_ptr(x), ld, _ptr(out), rows, cols, int(accumulate), _ptr(scratch),
                               scratch.numel() if scratch is not None else 0, _stream()),
           "pn_colsum_f32")


def relu_bwd(dy, y, dx):
    assert dy.is_contiguous() and y.is_contiguous() and dx.is_contiguous() and \
        dy.numel() == y.numel() == dx.numel()
    _check(lib().pn_relu_bwd_f32(_ptr(dy), _ptr(y), _ptr(dx), dy.numel(), _stream()),
           "pn_relu_bwd_f32")


def add_periodic(a, b, out):
    """out = a + b tiled over a's leading rows (b.numel() divides a.numel()); out may be a."""
    assert a.is_contiguous() and b.is_contiguous() and out.is_contiguous() and \
        a.numel() == out.numel() and a.numel() % b.numel() == 0
    _check(lib().pn_add_periodic_f32(_ptr(a), _ptr(b), _ptr(out), a.numel(), b.numel(), _stream()),
           "pn_add_periodic_f32")


def batch_sum(x, out, B, accumulate=False):
    assert x.is_contiguous() and out.is_contiguous() and x.numel() == B * out.numel()
    _check(lib().pn_batch_sum_f32(_ptr(x), _ptr(out), B, out.numel(), int(accumulate), _stream()),
           "pn_batch_sum_f32")


def layernorm256_bwd(dy, x, gamma, dx, gxhat, eps=1e-5):
    rows = x.shape[0]
    for t in (dy, x, dx, gxhat):
        assert t.is_contiguous() and tuple(t.shape) == (rows, 256)
    _check(lib().pn_layernorm256_bwd_f32(_ptr(dy), _ptr(x), _ptr(gamma), _ptr(dx), _ptr(gxhat),
                                         rows, eps, _stream()), "pn_layernorm256_bwd_f32")


def mha_bwd_scratch_floats(B, Nq, Nk):
    return 2 * B * 8 * Nq * Nk


def mha_bwd(q, k, v, dout, dq, dk, dv, scratch, B, Nq, Nk, scale, bits=None, rowall=None):
    """q / dout / dq: 2-D views [B*Nq, 256] (any row stride), k / v / dk / dv [B*Nk, 256];
    bits / rowall: the forward's packed boolean mask (pn_mask_pack's outputs) or None."""
    ld = lambda t: _rowmajor(t)[1]
    assert scratch.numel() >= mha_bwd_scratch_floats(B, Nq, Nk)
    _check(lib().pn_mha_bwd_f32(_ptr(q), ld(q), _ptr(k), ld(k), _ptr(v), ld(v), _ptr(dout), ld(dout),
                                _ptr(dq), ld(dq), _ptr(dk), ld(dk), _ptr(dv), ld(dv),
                                _ptr(bits, torch.int32), _ptr(rowall, torch.int32), _ptr(scratch),
                                B, Nq, Nk, scale, _stream()), "pn_mha_bwd_f32")


def scatter_rows_add(src, index, out, B, rows_out, slots, length, accumulate=False):
    """src [B * slots, >= length], out [B * rows_out, >= length] (2-D views, free row strides)."""
    (rs, lds), (ro, ldo) = _rowmajor(src), _rowmajor(out)
    assert rs == B * slots and ro == B * rows_out and index.numel() == B * slots
    _check(lib().pn_scatter_rows_add_f32(_ptr(src), lds, _ptr(index, torch.int64), _ptr(out), ldo,
                                         B, rows_out, slots, length, int(accumulate), _stream()),
           "pn_scatter_rows_add_f32")


def cosine_bwd(draw, x, other_hat, dx, B, Q, transposed, eps=1e-12):
    _check(lib().pn_cosine_bwd_f32(_ptr(draw), _ptr(x), _ptr(other_hat), _ptr(dx), B, Q,
                                   int(transposed), eps, _stream()), "pn_cosine_bwd_f32")


def mlearner_last_bwd_data(g, w3, c, dc, B, S):
    _check(lib().pn_mlearner_last_bwd_data_f32(_ptr(g), _ptr(w3), _ptr(c), _ptr(dc), B, S, _stream()),
           "pn_mlearner_last_bwd_data_f32")


def tapcorr1(F, g, part, B, S, sgn):
    assert part.numel() == B * S * 49 * 64
    _check(lib().pn_tapcorr1_f32(_ptr(F), _ptr(g), _ptr(part), B, S, sgn, _stream()),
           "pn_tapcorr1_f32")


def conv_weight_bwd_layout(w, out, Co, T, Ci):
    assert w.numel() == out.numel() == Co * T * Ci
    _check(lib().pn_conv_weight_bwd_layout_f32(_ptr(w), _ptr(out), Co, T, Ci, _stream()),
           "pn_conv_weight_bwd_layout_f32")


def msda_offaw_bwd(grad_loc, grad_aw, aw, d_offaw, shapes):
    """d [offsets | logits] rows (2-D view, free row stride) from pn_msda_bwd_f32's outputs."""
    rows, ld = _rowmajor(d_offaw)
    L = len(shapes)
    hs = (C.c_int32 * L)(*[h for h, _ in shapes])
    ws = (C.c_int32 * L)(*[w for _, w in shapes])
    _check(lib().pn_msda_offaw_bwd_f32(_ptr(grad_loc), _ptr(grad_aw), _ptr(aw), _ptr(d_offaw), ld,
                                       rows, L, hs, ws, _stream()), "pn_msda_offaw_bwd_f32")


def groupnorm_nhwc_bwd(x, dy, gamma, dx, gxhat, stats, B, HW, G, x_bstride, dy_bstride, eps=1e-5):
    _check(lib().pn_groupnorm_nhwc_bwd_f32(_ptr(x), _ptr(dy), _ptr(gamma), _ptr(dx), _ptr(gxhat),
                                           _ptr(stats), B, HW, G, eps, x_bstride, dy_bstride,
                                           _stream()), "pn_groupnorm_nhwc_bwd_f32")


def conv_wgrad(dY, X, part, B, Hi, Wi, Ho, Wo, Ci, Co, K, stride, pad, rows_per):
    assert part.numel() == B * ((Ho + rows_per - 1) // rows_per) * Co * K * K * Ci
    _check(lib().pn_conv_wgrad_f32(_ptr(dY), _ptr(X), _ptr(part), B, Hi, Wi, Ho, Wo, Ci, Co, K,
                                   stride, pad, rows_per, _stream()), "pn_conv_wgrad_f32")


def dilate2(x, out, B, Hi, Wi, Ho, Wo, Cc, accumulate=False):
    _check(lib().pn_dilate2_f32(_ptr(x), _ptr(out), B, Hi, Wi, Ho, Wo, Cc, int(accumulate),
                                _stream()), "pn_dilate2_f32")


def subsample2(x, out, B, Hi, Wi, Ho, Wo, Cc):
    _check(lib().pn_subsample2_f32(_ptr(x), _ptr(out), B, Hi, Wi, Ho, Wo, Cc, _stream()),
           "pn_subsample2_f32")


def scale_rows(x, s):
    rows = s.numel()
    assert x.is_contiguous() and x.numel() % rows == 0
    _check(lib().pn_scale_rows_f32(_ptr(x), _ptr(s), rows, x.numel() // rows, _stream()),
           "pn_scale_rows_f32")


def grad_norm_clip(g, out, scratch, pre=1.0, max_norm=0.0):
    """out[0] = ||pre * g||, out[1] = clip coefficient; g flat fp32, scratch >= 256 float64."""
    assert g.is_contiguous() and out.numel() >= 2 and scratch.numel() >= 256
    _check(lib().pn_grad_norm_clip_f32(_ptr(g), g.numel(), pre, max_norm, _ptr(out),
                                       _ptr(scratch, torch.float64), _stream()),
           "pn_grad_norm_clip_f32")


def adamw(p, g, m, v, seg_off, seg_lr, seg_wd, lr, beta1, beta2, eps, weight_decay, step,
          clip=None, pre=1.0):
    n = p.numel()
    assert all(t.is_contiguous() and t.numel() == n for t in (p, g, m, v))
    nseg = seg_lr.numel()
    assert seg_off.numel() == nseg + 1 and seg_wd.numel() == nseg
    _check(lib().pn_adamw_f32(_ptr(p), _ptr(g), _ptr(m), _ptr(v), n, _ptr(seg_off, torch.int64),
                              _ptr(seg_lr), _ptr(seg_wd), nseg, lr, beta1, beta2, eps, weight_decay,
                              step, _ptr(clip), pre, _stream()), "pn_adamw_f32")


def bce_posw_mean(logits, target, out, loss_weight):
    _check(lib().pn_bce_posw_mean_f32(_ptr(logits), _ptr(target), _ptr(out), logits.numel(),
                                      loss_weight, _stream()), "pn_bce_posw_mean_f32")
