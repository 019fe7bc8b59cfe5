"""CPU restatement of the reference's test-time image pipeline (TEST INFRASTRUCTURE).

configs/mask2former/pairnet.py:310-331: LoadImageFromFile (BGR uint8 HWC) ->
MultiScaleFlipAug(img_scale=(1333, 800), flip=False)[Resize(keep_ratio=True) ->
RandomFlip (off) -> Normalize(mean, std, to_rgb=True; :229-231) -> Pad(size_divisor=1) ->
ImageToTensor].  The transforms are mmdet 2.25.1 / mmcv 1.7.0 code (not vendored); mmcv's
`imresize` calls `cv2.resize(..., INTER_LINEAR)`.  cv2 is not installed in this image, so
this restates OpenCV's published algorithm for 8-bit INTER_LINEAR (fixed point, 11-bit
coefficients) in numpy integer arithmetic.  PARITY UNPINNED against OpenCV / mmcv themselves;
cross-checked in tests/test_oracle.py against torch's float bilinear resampling (same
half-pixel geometry) to within one grey level.
"""
import numpy as np

MEAN = (123.675, 116.28, 103.53)      # configs/mask2former/pairnet.py:229-231 (RGB order)
STD = (58.395, 57.12, 57.375)
IMG_SCALE = (1333, 800)


def rescale_size(h, w, scale=IMG_SCALE):
    """mmcv.rescale_size with a (long edge, short edge) tuple."""
    max_long, max_short = max(scale), min(scale)
    f = min(max_long / max(h, w), max_short / min(h, w))
    return int(h * float(f) + 0.5), int(w * float(f) + 0.5)


def _coef(n_dst, n_src, horizontal):
    d = np.arange(n_dst, dtype=np.float64)
    f = ((d + 0.5) * (n_src / n_dst) - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = f - s.astype(np.float32)
    if horizontal:
        f = np.where(s < 0, np.float32(0), f)
        s = np.maximum(s, 0)
        f = np.where(s >= n_src - 1, np.float32(0), f)
        s = np.minimum(s, n_src - 1)
    a0 = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int64)
    a1 = np.rint(f * np.float32(2048)).astype(np.int64)
    return s, a0, a1


def resize_linear_u8(img, hn, wn):
    """OpenCV resize(INTER_LINEAR) for uint8 HWC images, fixed point."""
    h, w = img.shape[:2]
    sx, ax0, ax1 = _coef(wn, w, True)
    sy, by0, by1 = _coef(hn, h, False)
    x1 = np.minimum(sx + 1, w - 1)
    src = img.astype(np.int64)
    hor = src[:, sx] * ax0[None, :, None] + src[:, x1] * ax1[None, :, None]     # (h, wn, 3)
    y0, y1 = np.clip(sy, 0, h - 1), np.clip(sy + 1, 0, h - 1)
    v = (((by0[:, None, None] * (hor[y0] >> 4)) >> 16) +
         ((by1[:, None, None] * (hor[y1] >> 4)) >> 16) + 2) >> 2
    return np.clip(v, 0, 255).astype(np.uint8)


def test_pipeline(img_bgr_u8, img_scale=IMG_SCALE, mean=MEAN, std=STD, to_rgb=True,
                  size_divisor=1):
    """-> (img float32 (1, 3, Hp, Wp), img_meta dict) as mmdet's pipeline + collate give
    `PSGTr.simple_test`."""
    h, w = img_bgr_u8.shape[:2]
    hn, wn = rescale_size(h, w, img_scale)
    res = resize_linear_u8(img_bgr_u8, hn, wn)
    x = res.astype(np.float32)
    if to_rgb:
        x = x[..., ::-1]
    stdinv = (1.0 / np.asarray(std, np.float64)).astype(np.float32)
    x = (x - np.asarray(mean, np.float32)) * stdinv
    hp = -(-hn // size_divisor) * size_divisor
    wp = -(-wn // size_divisor) * size_divisor
    out = np.zeros((1, 3, hp, wp), np.float32)
    out[0, :, :hn, :wn] = x.transpose(2, 0, 1)
    sf = np.array([wn / w, hn / h, wn / w, hn / h], np.float32)
    meta = dict(ori_shape=(h, w, 3), img_shape=(hn, wn, 3), pad_shape=(hp, wp, 3),
                scale_factor=sf, flip=False, batch_input_shape=(hp, wp))
    return out, meta
