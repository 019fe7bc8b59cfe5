"""CPU: the drop-in boundary -- C ABI exports, config schema, state-dict names,
loud failure without a GPU, and "the product never touches the oracle"."""
import ctypes
import os
import re

import pytest
import torch

from helpers import head_cfg
from oracle import ref_shim
from oracle.head import OracleCrossHead2

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(built_lib):
    header = open(os.path.join(ROOT, "include", "pairnet_hip.h")).read()
    declared = set(re.findall(r"\b(pn_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 20
    lib = ctypes.CDLL(built_lib)
    missing = [n for n in declared if not hasattr(lib, n)]
    assert not missing, missing
    lib.pn_abi_version.restype = ctypes.c_int
    assert lib.pn_abi_version() == int(re.search(r"#define PN_ABI_VERSION (\d+)", header).group(1))
    from pairnet_amd import hip
    assert declared == set(hip.EXPORTS)


def test_library_has_no_packed_fp32_and_bf16_mfma_only_in_the_split_gemm(built_lib):
    """Packed-fp32 VALU results were measured wrong in waves that share a CU with the bf16-MFMA
    GEMM (tools/coresidency_probe.py, profiles/r06_coresidency.txt): the library is built without
    the instructions (pair-net_amd/build.py), checked here by disassembling every embedded gfx950
    code object.  The bf16 MFMAs live in csrc/gemm_s3.hip's kernels and nowhere else."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_packed_fp32 as chk
    packed, bf16 = 0, {}
    cur = None
    for asm in chk.code_objects(built_lib):
        for line in asm.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
            if m:
                cur = m.group(1)
            elif re.search(r"v_pk_[a-z0-9]+_f32", line):
                packed += 1
            elif "v_mfma_f32_32x32x16_bf16" in line:
                bf16[cur] = bf16.get(cur, 0) + 1
    assert packed == 0
    assert bf16 and all("k_gemm_s3" in k for k in bf16), sorted(bf16)


def test_bad_arguments_are_refused_without_launching(built_lib):
    from pairnet_amd import hip
    lib = hip.lib()
    d = hip.GemmDesc()  # all NULL
    assert lib.pn_gemm_f32(ctypes.byref(d), None) == -1
    assert lib.pn_topk_pairs(None, None, None, None, None, 1, 100, 100, None) == -1
    assert lib.pn_gemm_group_f32(None, 3, None) == -1
    s3 = hip.GemmS3Desc()   # all NULL
    assert lib.pn_gemm_s3_f32(ctypes.byref(s3), None) == -1
    assert lib.pn_s3_split_f32(None, 256, None, 0, None, 32, 256, None) == -1
    assert lib.pn_layernorm_f32(None, None, None, None, 4, 256, 1e-5, None) == -1
    # round 4 entries: refused before anything is launched (NULL operands, N != 256, K % 32)
    assert lib.pn_linear_res_ln_f32(None, 256, None, 256, None, None, 256, None, None, None, 256,
                                    64, 256, 256, 1e-5, None) == -1
    assert lib.pn_linear_res_ln_f32(16, 256, 16, 256, None, 16, 256, 16, 16, 32, 256,
                                    64, 128, 256, 1e-5, None) == -1      # N != 256
    assert lib.pn_linear_res_ln_f32(16, 256, 16, 256, None, 16, 256, 16, 16, 32, 256,
                                    64, 256, 48, 1e-5, None) == -1       # K % 32
    assert lib.pn_msda_ex_f32(None, 256, None, 288, None, 1, 3, None, None, 0, None) == -1
    assert lib.pn_point_sample_f32(None, 0, None, None, 1, 8, 8, 16, None) == -1
    assert lib.pn_gt_mask_prepare_u8(None, None, 3, 8, 8, 16, 16, 8, 8, None) == -1
    assert lib.pn_pan_masks_u8(None, None, None, None, None, 0, 8, 8, None) == -1
    assert lib.pn_pan_masks_u8(16, 16, None, 32, None, 257, 8, 8, None) == -1      # G > 256
    assert lib.pn_pan_masks_u8(16, 16, None, 33, None, 2, 8, 8, None) == -1        # alignment
    assert lib.pn_gt_mask_prepare_u8(16, 32, 3, 20, 8, 16, 16, 8, 8, None) == -1   # h > H
    assert lib.pn_mask_match_cost_f32(None, 134, None, None, None, None, 100, 3, 64, 2.0, 5.0, 5.0,
                                      1.0, None) == -1
    assert lib.pn_id_match_cost_f32(None, None, None, 134, 56, None, None, None, None, 100, 3,
                                    1.0, 1.0, 0.0, None) == -1
    assert lib.pn_ce_mean_f32(None, 134, None, None, None, 10, 134, 1.0, None) == -1
    assert lib.pn_seesaw_mean_f32(None, 56, None, None, None, 10, 56, 0.8, 2.0, 0.01, 2.0, None) == -1
    assert lib.pn_bce_posw_mean_f32(None, None, None, 100, 5.0, None) == -1


def test_state_dict_names_match_reference_layout():
    from pairnet_amd import CrossHead2
    cfg = head_cfg()
    ours = {k: tuple(v.shape) for k, v in CrossHead2(**cfg).state_dict().items()}
    theirs = {k: tuple(v.shape) for k, v in OracleCrossHead2(**cfg).state_dict().items()}
    assert ours == theirs
    if ref_shim.available():
        ref = ref_shim.build_reference_head()
        assert ours == {k: tuple(v.shape) for k, v in ref.state_dict().items()}


@pytest.mark.skipif(not ref_shim.available(), reason="reference tree not present")
def test_reference_config_file_drops_in():
    """The reference's own configs/mask2former/pairnet.py builds our detector, and our
    restated schema equals it key for key."""
    from pairnet_amd import CrossHead2, load_config, pairnet_r50
    cfg = load_config(os.path.join(ref_shim.REF_ROOT, "configs/mask2former/pairnet.py"))
    ref_head = dict(cfg.model.bbox_head)
    assert ref_head == dict(pairnet_r50().bbox_head)
    ref_head.pop("type")
    head = CrossHead2(**ref_head)
    assert head.num_classes == 133 and head.num_rel_query == 100 and head.use_mask


@pytest.mark.skipif(not ref_shim.available(), reason="reference tree not present")
def test_reference_swinb_config_file_drops_in():
    """The reference's own configs/mask2former/pairnet_swinb.py (Swin-B backbone, :203-226,
    under the same head, :227-...) builds our detector, and `pairnet_swin("B")` equals it
    key for key -- backbone and head sections."""
    from pairnet_amd import (CrossHead2, SwinTransformerHip, build_detector, load_config,
                             pairnet_swin)
    cfg = load_config(os.path.join(ref_shim.REF_ROOT, "configs/mask2former/pairnet_swinb.py"))
    ours = pairnet_swin("B")
    strip = lambda d: {k: v for k, v in dict(d).items()
                       if k not in ("object_classes", "predicate_classes")}
    as_plain = lambda d: {k: (list(v) if isinstance(v, (tuple, list)) else v) for k, v in d.items()}
    assert as_plain(dict(cfg.model.backbone)) == as_plain(dict(ours.backbone))
    assert strip(cfg.model.bbox_head) == strip(ours.bbox_head)
    det = build_detector(dict(cfg.model))
    assert isinstance(det.backbone, SwinTransformerHip) and isinstance(det.bbox_head, CrossHead2)
    assert det.bbox_head.in_channels == [128, 256, 512, 1024]
    # mmdet's SwinTransformer state-dict layout (names from memory of mmdet 2.25.1, see
    # SURVEY N7): what a converted checkpoint of this config would carry
    keys = set(det.backbone.state_dict())
    for k in ("patch_embed.projection.weight", "stages.2.blocks.17.attn.w_msa.qkv.weight",
              "stages.0.downsample.reduction.weight", "norm3.weight"):
        assert k in keys, k


@pytest.mark.skipif(not ref_shim.available(), reason="reference tree not present")
def test_sibling_heads_drop_in():
    """BASELINE config #5 / SURVEY 8f rank 3: CrossHeadBaseline and PSGTrHead2 build from
    the reference's own config files with the reference's state-dict layout."""
    from pairnet_amd import (CrossHeadBaseline, PSGTrHead2, baseline_r50, build_detector,
                             load_config, psgtr2_r50)
    for path, restated, cls, ref_build in (
            ("configs/mask2former/baseline_r50_psg.py", baseline_r50, CrossHeadBaseline,
             ref_shim.build_reference_baseline_head),
            ("configs/psgtr/psgtr_r50_psg_plus.py", psgtr2_r50, PSGTrHead2,
             ref_shim.build_reference_psgtr2_head)):
        cfg = load_config(os.path.join(ref_shim.REF_ROOT, path))
        ref_head = dict(cfg.model.bbox_head)
        ours = dict(restated().bbox_head)
        assert {k: v for k, v in ref_head.items()
                if k not in ("object_classes", "predicate_classes")} == ours, path
        det = build_detector(dict(cfg.model))
        assert isinstance(det.bbox_head, cls)
        ref = ref_build()
        assert {k: tuple(v.shape) for k, v in det.bbox_head.state_dict().items()} == \
            {k: tuple(v.shape) for k, v in ref.state_dict().items()}, path
        for name in ("forward", "get_bboxes", "simple_test_bboxes", "simple_test"):
            assert callable(getattr(det.bbox_head, name))


def test_no_cpu_fallback():
    from pairnet_amd import CrossHead2
    head = CrossHead2(**head_cfg())
    feats = [torch.zeros(1, c, 8 // (2 ** i) or 1, 8 // (2 ** i) or 1)
             for i, c in enumerate((256, 512, 1024, 2048))]
    with pytest.raises(RuntimeError):
        head.forward(feats, [dict(img_shape=(32, 32, 3), scale_factor=[1, 1, 1, 1])])
    with pytest.raises(RuntimeError):
        head.to("cpu")._pack()


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "pair-net_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, re.M), f
                assert "/root/reference" not in text, f


def test_detector_builds_from_config_and_result_container():
    import numpy as np
    from pairnet_amd import build_detector, pairnet_r50, triplet2Result
    det = build_detector(pairnet_r50())
    assert det.num_classes == 133
    keys = list(det.backbone.state_dict())
    assert "conv1.weight" in keys and "layer4.2.bn3.running_var" in keys
    t = (torch.zeros(200, 5), torch.ones(200, dtype=torch.long),
         torch.arange(200, dtype=torch.int).reshape(2, -1).T, torch.zeros(200, 4, 4, dtype=torch.bool),
         torch.ones(4, 4, dtype=torch.long), torch.zeros(100), torch.zeros(100),
         torch.zeros(100, 57))
    r = triplet2Result(t, True)
    assert isinstance(r.rel_dists, np.ndarray) and r.rel_pair_idxes.shape == (100, 2)
    assert len(r) == 1 and r[0] is r and r.formatted_masks["pan_results"] is r.pan_results


@pytest.mark.skipif(not ref_shim.available(), reason="reference tree not present")
def test_reference_bbox_config_file_drops_in():
    """configs/deformable_detr/cross_r101_vg.py (ResNet-101 C3-C5 -> ChannelMapper ->
    CrossHeadBBox) builds our detector; the restated schema equals the file on every key except
    the training losses, and the head / neck carry the reference class's state-dict layout."""
    from pairnet_amd import (ChannelMapper, CrossHeadBBox, build_detector, cross_r101_vg,
                             load_config)
    from oracle.deformable_detr import ChannelMapper as OracleMapper
    cfg = load_config(os.path.join(ref_shim.REF_ROOT, "configs/deformable_detr/cross_r101_vg.py"))
    ours = cross_r101_vg().model
    plain = lambda d: {k: (list(v) if isinstance(v, (tuple, list)) else v) for k, v in dict(d).items()}
    bb = plain(cfg.model.backbone)
    bb.pop("init_cfg")
    assert bb == plain(ours.backbone)
    assert plain(cfg.model.neck) == plain(ours.neck)
    ref_head, our_head = dict(cfg.model.bbox_head), dict(ours.bbox_head)
    assert {k: ref_head[k] for k in our_head} == our_head
    assert set(ref_head) - set(our_head) == {"rel_cls_loss", "subobj_cls_loss",
                                             "importance_match_loss", "loss_bbox", "loss_iou"}
    det = build_detector(dict(cfg.model))
    assert isinstance(det.bbox_head, CrossHeadBBox) and isinstance(det.neck, ChannelMapper)
    assert det.backbone.depth == 101 and det.out_indices == (1, 2, 3)
    ref = ref_shim.build_reference_bbox_head()
    assert list((k, tuple(v.shape)) for k, v in det.bbox_head.state_dict().items()) == \
        list((k, tuple(v.shape)) for k, v in ref.state_dict().items())
    ncfg = dict(cfg.model.neck)
    ncfg.pop("type")
    assert list((k, tuple(v.shape)) for k, v in det.neck.state_dict().items()) == \
        list((k, tuple(v.shape)) for k, v in OracleMapper(**ncfg).state_dict().items())
    for name in ("forward", "get_bboxes", "simple_test_bboxes", "simple_test"):
        assert callable(getattr(det.bbox_head, name))
    # the configs that are NOT consistent with the class fail loudly, like the reference
    bad = load_config(os.path.join(ref_shim.REF_ROOT, "configs/deformable_detr/cross_r50_coco.py"))
    with pytest.raises((ValueError, NotImplementedError, TypeError)):
        build_detector(dict(bad.model))


def test_detector_checkpoint_round_trip(tmp_path):
    """tools/test.py:235-248: `load_checkpoint(model, path, map_location="cpu")` on an
    mmdet-layout file (`backbone.* / neck.* / bbox_head.*`, DDP `module.` prefix, `meta`)."""
    import torch
    from pairnet_amd import build_detector, cross_r101_vg, load_checkpoint, pairnet_r50
    for cfg, n_parts in ((cross_r101_vg().model, 3), (pairnet_r50(), 2)):
        det = build_detector(dict(cfg))
        sd = det.state_dict()
        assert len({k.split(".")[0] for k in sd}) == n_parts
        g = torch.Generator().manual_seed(1)
        new = {"module." + k: torch.randn(v.shape, generator=g) for k, v in sd.items()}
        path = str(tmp_path / "ckpt.pth")
        torch.save(dict(state_dict=new, meta=dict(CLASSES=("a", "b"), PREDICATES=("on",))), path)
        ckpt = load_checkpoint(det, path, map_location="cpu", strict=True)
        assert ckpt["meta"]["CLASSES"] == ("a", "b")
        got = det.state_dict()
        assert all(torch.equal(got[k], new["module." + k].to(got[k].dtype)) for k in got)
        with pytest.raises(RuntimeError):
            det.load_state_dict({k: v for k, v in list(new.items())[1:]}, strict=True)


def test_no_process_wide_scheduling_state(built_lib):
    """The header promises "no global state": the persistent-grid hint travels in each call's
    descriptor (PN_GEMM_RESERVE), there is no setter, and the grid of one caller's GEMM does
    not depend on what another caller in the process asked for."""
    from pairnet_amd import hip
    lib = hip.lib()
    assert not hasattr(lib, "pn_gemm_set_grid_trim") and not hasattr(lib, "pn_gemm_set_grid_scale")
    header = open(os.path.join(ROOT, "include", "pairnet_hip.h")).read()
    assert "pn_gemm_set" not in header and "process-wide knob" in header

    def desc():
        d = hip.GemmDesc()
        d.M, d.N, d.K, d.batch = 21950, 1024, 256, 1
        d.flags = hip._reserve_flag()
        return d
    alone = lib.pn_gemm_grid_size(ctypes.byref(desc()))
    full = 256 * lib.pn_gemm_wgs_per_cu()                  # 256 CUs x resident workgroups (5)
    assert alone == full and lib.pn_gemm_wgs_per_cu() in (4, 5)
    with hip.reserve_slots(64):
        assert lib.pn_gemm_grid_size(ctypes.byref(desc())) == full - 64
        with hip.reserve_slots(128):
            assert lib.pn_gemm_grid_size(ctypes.byref(desc())) == full - 128
        assert lib.pn_gemm_grid_size(ctypes.byref(desc())) == full - 64
        # another thread (another head / pipeline of the process) is not affected
        import threading
        seen = []
        t = threading.Thread(target=lambda: seen.append(lib.pn_gemm_grid_size(ctypes.byref(desc()))))
        t.start()
        t.join()
        assert seen == [alone]
    assert lib.pn_gemm_grid_size(ctypes.byref(desc())) == alone
    small = desc()
    small.M = 640                                          # 10 x 16 tiles: grid = tile count
    assert lib.pn_gemm_grid_size(ctypes.byref(small)) == 160


def test_plan_caches_are_bounded():
    """Per-shape plans (device buffers + graphs) live in LRU caches (a keep-ratio
    evaluation pass meets hundreds of shapes)."""
    from pairnet_amd.plans import PlanCache
    c = PlanCache(max_plans=3)
    for i in range(5):
        c[("shape", i)] = i
    assert list(c) == [("shape", 2), ("shape", 3), ("shape", 4)] and c.evictions == 2
    assert c[("shape", 2)] == 2                            # a hit refreshes the entry
    c[("shape", 5)] = 5
    assert ("shape", 2) in c and ("shape", 3) not in c
    from pairnet_amd import CrossHead2, CrossHeadBBox, ResNet50Hip, SwinTransformerHip, ChannelMapper
    from pairnet_amd import bbox_head_cfg, channel_mapper_cfg
    head = CrossHead2(**head_cfg())
    assert isinstance(head._plans, PlanCache)
    head.load_state_dict(head.state_dict())
    assert isinstance(head._plans, PlanCache)
    assert isinstance(ResNet50Hip()._plans, PlanCache)
    assert isinstance(SwinTransformerHip(embed_dims=32, depths=(2, 2, 2, 2),
                                         num_heads=(1, 2, 4, 8), window_size=4)._plans, PlanCache)
    ncfg = channel_mapper_cfg()
    ncfg.pop("type")
    assert isinstance(ChannelMapper(**ncfg)._plans, PlanCache)
    bcfg = bbox_head_cfg()
    bcfg.pop("type")
    assert isinstance(CrossHeadBBox(**bcfg)._plans, PlanCache)


def test_load_checkpoint_refuses_pickled_code_by_default(tmp_path):
    from pairnet_amd import CrossHead2, load_checkpoint

    class Payload:                       # an arbitrary pickled object
        def __reduce__(self):
            return (print, ("code from the checkpoint ran",))
    head = CrossHead2(**head_cfg())
    path = os.path.join(tmp_path, "evil.pth")
    torch.save(dict(state_dict=head.state_dict(), meta=dict(x=Payload())), path)
    with pytest.raises(RuntimeError):
        load_checkpoint(head, path)
    load_checkpoint(head, path, trust=True)               # the caller vouches for the file
    good = os.path.join(tmp_path, "good.pth")
    torch.save(dict(state_dict=head.state_dict(), meta=dict(CLASSES=("a", "b"))), good)
    assert load_checkpoint(head, good)["meta"]["CLASSES"] == ("a", "b")


def test_host_unpack_of_mask_bits_needs_no_gpu():
    """`pn_unpack_bits_host`, the host half of the bit-packed mask transfer of
    `triplet2Result`: numpy.unpackbits(bitorder="little"), any length, any thread count."""
    import numpy as np
    from pairnet_amd import hip
    for n in (1, 8, 13, 65536 * 8 + 5, 200 * 96 * 128):
        rng = np.random.default_rng(n)
        bits = torch.from_numpy(rng.integers(0, 256, (n + 7) // 8, dtype=np.uint8))
        want = np.unpackbits(bits.numpy(), bitorder="little")[:n].astype(bool)
        for threads in (1, 4, 7):
            out = torch.zeros(n + 2, dtype=torch.bool)
            hip.unpack_bits_host(bits, out[:n], threads)
            assert np.array_equal(out[:n].numpy(), want) and not out[n:].any()
    with pytest.raises(RuntimeError):
        hip.unpack_bits_host(bits[:3], torch.zeros(100, dtype=torch.bool))
    assert hip.lib().pn_unpack_bits_host(None, None, 8, 1) == -1
    assert hip.lib().pn_unpack_bits_host(bits.data_ptr(), out.data_ptr(), 8, 0) == -1
