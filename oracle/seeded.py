"""Deterministic synthetic weights / inputs for fixtures (TEST INFRASTRUCTURE).

Golden fixtures cannot carry the 30.8 M head parameters, and torch's RNG stream is
not a contract across builds, so weights and inputs are regenerated from a seed with
numpy's PCG64 `Generator.random` (bit-stream stable by numpy policy).  Every fixture
also stores a checksum of what was generated, so a drift would be caught as
"fixture inputs differ" rather than as a parity failure.
"""
import zlib

import numpy as np
import torch


def uniform(rng, shape, lo, hi):
    n = int(np.prod(shape)) if len(shape) else 1
    a = rng.random(n, dtype=np.float32) * np.float32(hi - lo) + np.float32(lo)
    return torch.from_numpy(a.reshape(shape).copy())


def seeded_state_dict(shapes, seed):
    """`shapes`: OrderedDict name -> shape (CrossHead2.param_shapes() or an oracle
    state_dict's shapes).  Every parameter is random so that every term of the path
    matters (in particular MSDeformAttn offsets / attention logits depend on the
    input, which mmcv's default init would zero out)."""
    rng = np.random.default_rng(seed)
    out = {}
    for name, shape in shapes.items():
        shape = tuple(shape)
        leaf = name.rsplit(".", 1)[-1]
        is_norm = ".norms." in name or ".gn." in name or "post_norm" in name
        if is_norm:
            v = uniform(rng, shape, 0.5, 1.5) if leaf == "weight" else uniform(rng, shape, -0.1, 0.1)
        elif len(shape) == 1:
            v = uniform(rng, shape, -0.1, 0.1)
            if name in ("mask_embed.4.bias", "pixel_decoder.mask_feature.bias"):
                v = v * 0.0  # keeps mask logits sign-balanced -> ~50 % dense attention masks
            if "sampling_offsets" in name:
                v = uniform(rng, shape, -2.0, 2.0)
        elif name.endswith(("query_embed.weight", "query_feat.weight", "query_embed2.weight",
                            "query_embed3.weight", "level_embed.weight",
                            "level_encoding.weight")):
            v = uniform(rng, shape, -1.0, 1.0)
        else:
            fan_in = int(np.prod(shape[1:]))
            a = float(np.sqrt(3.0 / fan_in))
            v = uniform(rng, shape, -a, a)
        out[name] = v
    return out


def seeded_feats(seed, batch, height, width, channels=(256, 512, 1024, 2048),
                 strides=(4, 8, 16, 32)):
    """Backbone-shaped feature pyramid for an image of (height, width)."""
    rng = np.random.default_rng(seed)
    sizes = []
    h, w = (height + 1) // 2, (width + 1) // 2          # conv1 s2
    h, w = (h + 1) // 2, (w + 1) // 2                    # maxpool s2  -> C2
    for _ in strides:
        sizes.append((h, w))
        h, w = (h + 1) // 2, (w + 1) // 2
    return [uniform(rng, (batch, c, hh, ww), -1.0, 1.0) for c, (hh, ww) in zip(channels, sizes)]


def checksum(tensors):
    """crc32 over the raw bytes of a list/dict of tensors (order-dependent)."""
    if isinstance(tensors, dict):
        tensors = list(tensors.values())
    c = 0
    for t in tensors:
        c = zlib.crc32(t.detach().contiguous().numpy().tobytes(), c)
    return c & 0xFFFFFFFF
