"""The test-time image front end (inference half): resize (keep ratio,
(1333, 800)) / normalise / pad, configs/mask2former/pairnet.py:310-331 and :229-231.

CPU: the oracle (oracle/preprocess.py, OpenCV's fixed-point INTER_LINEAR restated -- cv2 is
not in the image, so unpinned against OpenCV itself) against torch's float bilinear
resampling; mmcv.rescale_size cases; the reference's own config file.  GPU: the HIP kernel
against the oracle, bit for bit."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import preprocess as OP
from oracle import ref_shim


def _image(seed, h, w):
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, (h // 8 + 2, w // 8 + 2, 3)).astype(np.float32)
    img = F.interpolate(torch.from_numpy(base).permute(2, 0, 1)[None], size=(h, w),
                        mode="bicubic", align_corners=False)[0].permute(1, 2, 0).numpy()
    img += rng.normal(0, 6, img.shape)                 # smooth content + sensor-like noise
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def test_rescale_size_is_mmcv_keep_ratio_arithmetic():
    assert OP.rescale_size(384, 640) == (800, 1333)     # the north-star shape (x 2.083)
    assert OP.rescale_size(480, 640) == (800, 1067)
    assert OP.rescale_size(640, 480) == (1067, 800)
    assert OP.rescale_size(500, 333) == (1201, 800)
    assert OP.rescale_size(1000, 3000) == (444, 1333)   # long-edge bound


@pytest.mark.parametrize("h,w", [(384, 640), (97, 131), (1000, 1500)])
def test_fixed_point_resize_UNPINNED_against_opencv_is_within_one_grey_level_of_float_bilinear(h, w):
    img = _image(3, h, w)
    hn, wn = OP.rescale_size(h, w)
    got = OP.resize_linear_u8(img, hn, wn).astype(np.int32)
    ref = F.interpolate(torch.from_numpy(img.astype(np.float32)).permute(2, 0, 1)[None],
                        size=(hn, wn), mode="bilinear", align_corners=False)[0]
    ref = ref.permute(1, 2, 0).numpy()
    assert np.abs(got - ref).max() <= 1.0 + 1e-3
    assert np.abs(got - np.rint(ref)).mean() < 0.2   # (the fixed-point passes truncate)
    # identity when nothing is rescaled
    assert np.array_equal(OP.resize_linear_u8(img, h, w), img)


def test_pipeline_output_layout_and_metas():
    img = _image(5, 96, 160)
    out, meta = OP.test_pipeline(img, img_scale=(333, 200), size_divisor=32)
    hn, wn = OP.rescale_size(96, 160, (333, 200))
    assert meta["img_shape"] == (hn, wn, 3) and meta["ori_shape"] == (96, 160, 3)
    assert out.shape == (1, 3, -(-hn // 32) * 32, -(-wn // 32) * 32)
    assert np.all(out[0, :, hn:, :] == 0) and np.all(out[0, :, :, wn:] == 0)
    res = OP.resize_linear_u8(img, hn, wn)
    r = (res[..., 2].astype(np.float32) - np.float32(123.675)) * np.float32(1 / 58.395)
    assert np.array_equal(out[0, 0, :hn, :wn], r)        # channel 0 is R (to_rgb)
    assert np.allclose(meta["scale_factor"], [wn / 160, hn / 96, wn / 160, hn / 96])


@pytest.mark.skipif(not ref_shim.available(), reason="reference tree not present")
def test_restated_test_pipeline_equals_the_reference_config():
    from pairnet_amd import load_config, test_pipeline_cfg
    cfg = load_config(os.path.join(ref_shim.REF_ROOT, "configs/mask2former/pairnet.py"))
    assert cfg.test_pipeline == test_pipeline_cfg()
    assert cfg.img_norm_cfg["mean"] == list(OP.MEAN) and cfg.img_norm_cfg["std"] == list(OP.STD)


@pytest.mark.gpu
@pytest.mark.parametrize("h,w,div", [(384, 640, 1), (97, 131, 32), (1000, 1500, 1), (333, 500, 32)])
def test_hip_front_end_equals_the_oracle_bit_for_bit(h, w, div):
    from pairnet_amd import TestPipeline, test_pipeline_cfg
    img = _image(7, h, w)
    pipe = TestPipeline.from_config(test_pipeline_cfg())
    pipe.size_divisor = div
    out, metas = pipe(torch.from_numpy(img).cuda())
    ref, meta = OP.test_pipeline(img, size_divisor=div)
    assert tuple(out.shape) == ref.shape
    assert metas[0]["img_shape"] == meta["img_shape"] and metas[0]["pad_shape"] == meta["pad_shape"]
    assert np.array_equal(metas[0]["scale_factor"], meta["scale_factor"])
    assert np.array_equal(out.cpu().numpy(), ref)


@pytest.mark.gpu
def test_detector_detect_runs_from_a_decoded_image():
    from pairnet_amd import build_detector, pairnet_r50
    det = build_detector(pairnet_r50()).to("cuda:0")
    img = _image(9, 120, 200)
    res = det.detect(img)
    hn, wn = OP.rescale_size(120, 200)
    assert len(res) == 1 and res[0].rel_dists.shape == (100, 57)
    assert res[0].masks.shape == (200, 120, 200)         # back at the original size
    assert res[0].pan_results.shape == (120, 200)
