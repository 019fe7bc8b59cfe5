"""Test-time image front end on the MI355X (inference half).

The reference feeds `PSGTr.simple_test` from mmdet's CPU data pipeline
(configs/mask2former/pairnet.py:310-331: Resize(keep_ratio, img_scale=(1333, 800)) ->
Normalize(mean / std :229-231, to_rgb) -> Pad(size_divisor) -> ImageToTensor -> collate).
`TestPipeline` does the same on the GPU in one kernel (csrc/preprocess.hip): a decoded
uint8 HWC BGR image in, the normalised NCHW float tensor and its `img_metas` entry out, so
images/s can be counted from the decoded image.  No CPU path.
"""
import torch

from . import hip

MEAN = (123.675, 116.28, 103.53)
STD = (58.395, 57.12, 57.375)


def rescale_size(h, w, scale):
    """mmcv.rescale_size for a (long edge, short edge) `img_scale` tuple."""
    max_long, max_short = max(scale), min(scale)
    f = min(max_long / max(h, w), max_short / min(h, w))
    return int(h * float(f) + 0.5), int(w * float(f) + 0.5)


class TestPipeline:
    __test__ = False   # (not a pytest class)

    def __init__(self, img_scale=(1333, 800), mean=MEAN, std=STD, to_rgb=True, size_divisor=1,
                 device="cuda:0"):
        self.img_scale, self.to_rgb, self.size_divisor = tuple(img_scale), bool(to_rgb), size_divisor
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("TestPipeline runs on an MI355X only; there is no CPU path")
        self._mean = (torch.tensor(mean, dtype=torch.float32),
                      (1.0 / torch.tensor(std, dtype=torch.float64)).to(torch.float32))
        self._out = {}

    @classmethod
    def from_config(cls, test_pipeline, device="cuda:0"):
        """Build from the reference's `test_pipeline` list (the MultiScaleFlipAug entry)."""
        aug = [t for t in test_pipeline if t.get("type") == "MultiScaleFlipAug"][0]
        kw = dict(img_scale=aug["img_scale"], device=device)
        for t in aug["transforms"]:
            if t["type"] == "Normalize":
                kw.update(mean=t["mean"], std=t["std"], to_rgb=t.get("to_rgb", True))
            elif t["type"] == "Pad":
                kw.update(size_divisor=t.get("size_divisor") or 1)
            elif t["type"] == "Resize" and not t.get("keep_ratio", False):
                raise NotImplementedError("Resize(keep_ratio=True) only")
        if aug.get("flip", False):
            raise NotImplementedError("test-time flip")
        return cls(**kw)

    @torch.no_grad()
    @hip.on_device
    def __call__(self, img, out=None):
        """img: uint8 (H, W, 3) BGR, a device tensor (or a numpy array / host tensor, copied
        over) -> (float32 (1, 3, Hp, Wp) device tensor, [img_meta]).  The output is `out` if
        given, else a per-shape buffer that the next call with the same shape overwrites."""
        if not isinstance(img, torch.Tensor):
            img = torch.from_numpy(img)
        if img.dtype != torch.uint8 or img.dim() != 3 or img.shape[2] != 3:
            raise RuntimeError("img must be uint8 (H, W, 3)")
        img = img.to(self.device).contiguous()
        H, W = int(img.shape[0]), int(img.shape[1])
        Hn, Wn = rescale_size(H, W, self.img_scale)
        d = self.size_divisor
        Hp, Wp = -(-Hn // d) * d, -(-Wn // d) * d
        key = (Hp, Wp)
        if out is None:
            if key not in self._out:
                self._out[key] = torch.empty(1, 3, Hp, Wp, device=self.device, dtype=torch.float32)
            out = self._out[key]
        elif tuple(out.shape) != (1, 3, Hp, Wp) or out.dtype != torch.float32 or not out.is_contiguous():
            raise RuntimeError("out must be a contiguous float32 (1, 3, %d, %d) tensor" % (Hp, Wp))
        hip.preprocess_u8(img, H, W, out, Hn, Wn, Hp, Wp, self._mean[0], self._mean[1],
                          self.to_rgb)
        sf = torch.tensor([Wn / W, Hn / H, Wn / W, Hn / H], dtype=torch.float32).numpy()
        meta = dict(ori_shape=(H, W, 3), img_shape=(Hn, Wn, 3), pad_shape=(Hp, Wp, 3),
                    scale_factor=sf, flip=False, batch_input_shape=(Hp, Wp))
        return out, [meta]
