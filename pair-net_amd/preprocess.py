"""Test-time image front end on the MI355X (inference half).

The reference feeds `PSGTr.simple_test` from mmdet's CPU data pipeline
(configs/mask2former/pairnet.py:310-331: Resize(keep_ratio, img_scale=(1333, 800)) ->
Normalize(mean / std :229-231, to_rgb) -> Pad(size_divisor) -> ImageToTensor -> collate).
`TestPipeline` does the same on the GPU in one kernel (csrc/preprocess.hip): a decoded
uint8 HWC BGR image in, the normalised NCHW float tensor and its `img_metas` entry out, so
images/s can be counted from the decoded image.  No CPU path.
"""
from collections import OrderedDict

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
        self._out = OrderedDict()      # synchronous calls: a few per-shape outputs (LRU)
        self._slots = {}               # pipelined calls: one grow-only buffer per slot

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

    OUT_SHAPES = 4       # per-shape output buffers kept for the synchronous call form

    def _decoded(self, img):
        if not isinstance(img, torch.Tensor):
            img = torch.from_numpy(img)
        if img.dtype != torch.uint8 or img.dim() != 3 or img.shape[2] != 3:
            raise RuntimeError("img must be uint8 (H, W, 3)")
        return img.to(self.device, non_blocking=True).contiguous()

    def sizes(self, H, W):
        """(Hn, Wn) after Resize(keep_ratio), (Hp, Wp) after Pad of an H x W image."""
        Hn, Wn = rescale_size(H, W, self.img_scale)
        d = self.size_divisor
        return (Hn, Wn), (-(-Hn // d) * d, -(-Wn // d) * d)

    @staticmethod
    def _meta(H, W, Hn, Wn, Hp, Wp, batch_shape=None):
        sf = torch.tensor([Wn / W, Hn / H, Wn / W, Hn / H], dtype=torch.float32).numpy()
        return dict(ori_shape=(H, W, 3), img_shape=(Hn, Wn, 3), pad_shape=(Hp, Wp, 3),
                    scale_factor=sf, flip=False, batch_input_shape=batch_shape or (Hp, Wp))

    @torch.no_grad()
    @hip.on_device
    def __call__(self, img, out=None):
        """img: uint8 (H, W, 3) BGR, a device tensor (or a numpy array / host tensor, copied
        over) -> (float32 (1, 3, Hp, Wp) device tensor, [img_meta]).  The output is `out` if
        given, else a per-shape buffer that the next call with the same shape overwrites (a
        few shapes are kept; a loop over many shapes uses `batch(..., slot=)`)."""
        img = self._decoded(img)
        H, W = int(img.shape[0]), int(img.shape[1])
        (Hn, Wn), (Hp, Wp) = self.sizes(H, W)
        key = (Hp, Wp)
        if out is None:
            if key not in self._out:
                self._out[key] = torch.empty(1, 3, Hp, Wp, device=self.device, dtype=torch.float32)
                while len(self._out) > self.OUT_SHAPES:
                    self._out.popitem(last=False)
            self._out.move_to_end(key)
            out = self._out[key]
        elif tuple(out.shape) != (1, 3, Hp, Wp) or out.dtype != torch.float32 or not out.is_contiguous():
            raise RuntimeError("out must be a contiguous float32 (1, 3, %d, %d) tensor" % (Hp, Wp))
        hip.preprocess_u8(img, H, W, out, Hn, Wn, Hp, Wp, self._mean[0], self._mean[1],
                          self.to_rgb)
        return out, [self._meta(H, W, Hn, Wn, Hp, Wp)]

    @torch.no_grad()
    @hip.on_device
    def batch(self, images, slot=0):
        """The loader's batch of the reference's test loop (tools/test.py:202-214: pipeline per
        image, then mmcv's `collate`, which zero-pads to the largest image of the batch) for a
        pipelined caller: `images` = decoded uint8 (H, W, 3) BGR images -> (float32
        (k, 3, Hmax, Wmax) device tensor, [img_meta] * k) in ONE grow-only buffer per `slot`
        (a view of it: the next batch of the same slot overwrites it), written on the current
        stream -- a caller that keeps several batches in flight gives each stream its own slot
        and queues this call on that stream.  No allocation once the largest batch has passed."""
        imgs = [self._decoded(im) for im in images]
        geo = []
        for im in imgs:
            H, W = int(im.shape[0]), int(im.shape[1])
            geo.append((H, W) + self.sizes(H, W))
        Hp, Wp = max(g[3][0] for g in geo), max(g[3][1] for g in geo)
        need = len(imgs) * 3 * Hp * Wp
        buf = self._slots.get(slot)
        if buf is None or buf.numel() < need:
            # (allocated on the current stream: the caching allocator hands the old block back
            # to this stream's pool, where every earlier use of it was queued)
            buf = self._slots[slot] = torch.empty(need, device=self.device, dtype=torch.float32)
        out = buf[:need].view(len(imgs), 3, Hp, Wp)
        metas = []
        for i, (im, (H, W, (Hn, Wn), _)) in enumerate(zip(imgs, geo)):
            hip.preprocess_u8(im, H, W, out[i], Hn, Wn, Hp, Wp, self._mean[0], self._mean[1],
                              self.to_rgb)
            # per-image `pad_shape` (the reference's Pad(size_divisor) runs before mmcv's collate,
            # which only adds `batch_input_shape`): consumers that derive valid regions from it
            # (the box trunk's padding masks) see what the reference pipeline gives them
            metas.append(self._meta(H, W, Hn, Wn, *geo[i][3], batch_shape=(Hp, Wp)))
        return out, metas
