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


def seeded_param(name, shape, rng):
    """One parameter of the seeded family (distribution chosen by the parameter's name)."""
    shape = tuple(shape)
    leaf = name.rsplit(".", 1)[-1]
    is_norm = ".norms." in name or ".gn." in name or "_norm." in name
    if is_norm:
        return uniform(rng, shape, 0.5, 1.5) if leaf == "weight" else uniform(rng, shape, -0.1, 0.1)
    if len(shape) == 1:
        v = uniform(rng, shape, -0.1, 0.1)
        if name in ("mask_embed.4.bias", "pixel_decoder.mask_feature.bias"):
            v = v * 0.0  # keeps mask logits sign-balanced -> ~50 % dense attention masks
        if "sampling_offsets" in name:
            v = uniform(rng, shape, -2.0, 2.0)
        return v
    if name.endswith(("query_embed.weight", "query_feat.weight", "query_embed2.weight",
                      "query_embed3.weight", "level_embed.weight", "level_encoding.weight")):
        return uniform(rng, shape, -1.0, 1.0)
    fan_in = int(np.prod(shape[1:]))
    a = float(np.sqrt(3.0 / fan_in))
    return uniform(rng, shape, -a, a)


def seeded_state_dict(shapes, seed):
    """`shapes`: OrderedDict name -> shape (CrossHead2.param_shapes() or an oracle
    state_dict's shapes).  Every parameter is random so that every term of the path
    matters (in particular MSDeformAttn offsets / attention logits depend on the
    input, which mmcv's default init would zero out)."""
    rng = np.random.default_rng(seed)
    return {name: seeded_param(name, shape, rng) for name, shape in shapes.items()}


# ---- fixture "ops": small edits on top of the seeded weights, stored in the fixture ----
# A fixture cannot carry 30 M parameters, so it stores a weight seed plus a few edits
# (the keys below, applied in this order).  The "separated" fixtures use them to give the
# random-weight head the two properties of a trained one that exact top-k parity needs
# (LABNOTES.md section 3): queries that do not collapse onto one common vector, and an
# importance matrix whose top scores are spread far wider than fp32 rounding noise.
#   reseed_<name>   = seed            redraw that parameter from its own seed
#   keeprows_<name> = n               zero every row from n on
#   mlearner_skip   = t               add t to the centre tap of channel 0 -> 0 of each
#                                     Matrix Learner convolution (an identity path)
#   scale_<name>    = [f, r0, r1]     multiply rows r0:r1 by f
#   override_<name> = array           replace the parameter
OP_PREFIXES = ("reseed_", "keeprows_", "mlearner_skip", "scale_", "override_")


def ops_of(fx):
    """The ops stored in a fixture (np.load result or dict), in application order."""
    keys = fx.files if hasattr(fx, "files") else list(fx)
    out = {}
    for prefix in OP_PREFIXES:
        for k in keys:
            if k.startswith(prefix):
                out[k] = np.asarray(fx[k])
    return out


def apply_ops(sd, ops):
    """Apply fixture ops to a state dict in place (tensors are replaced, not mutated)."""
    for prefix in OP_PREFIXES:
        for k, v in ops.items():
            if not k.startswith(prefix):
                continue
            name = k[len(prefix):]
            if prefix == "reseed_":
                sd[name] = seeded_param(name, sd[name].shape, np.random.default_rng(int(v)))
            elif prefix == "scale_":
                f, r0, r1 = float(v[0]), int(v[1]), int(v[2])
                t = sd[name].clone()
                t[r0:r1] *= f
                sd[name] = t
            elif prefix == "keeprows_":
                t = sd[name].clone()
                t[int(v):] = 0
                sd[name] = t
            elif prefix == "mlearner_skip":
                for i in range(3):
                    n = "update_importance.conv_layers.%d.0.weight" % i
                    t = sd[n].clone()
                    t[0, 0, 3, 3] += float(v)
                    sd[n] = t
            else:
                sd[name] = torch.as_tensor(np.asarray(v)).clone()
    return sd


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


def smooth_feats(seed, batch, height, width, k, channels=(256, 512, 1024, 2048)):
    """The same pyramid with spatially SMOOTH maps: white noise on a grid k times coarser,
    bilinearly upsampled (align_corners).  Per-pixel white noise is the worst case for
    anything that samples the maps at computed positions -- the derivative of a bilinear
    sample is (neighbour difference) x (map width), which at 167 columns turns a 1e-6
    perturbation of a sampling location into 1e-4 of the sample, per decoder layer; a
    backbone's real feature maps are smooth on that scale."""
    import torch.nn.functional as F
    out = []
    for f in seeded_feats(seed, batch, height, width, channels):
        h, w = f.shape[-2:]
        hc, wc = -(-h // k) + 1, -(-w // k) + 1
        out.append(F.interpolate(f[..., :hc, :wc].contiguous(), size=(h, w), mode="bilinear",
                                 align_corners=True))
    return out


def checksum(tensors):
    """crc32 over the raw bytes of a list/dict of tensors (order-dependent)."""
    if isinstance(tensors, dict):
        tensors = list(tensors.values())
    c = 0
    for t in tensors:
        c = zlib.crc32(t.detach().contiguous().numpy().tobytes(), c)
    return c & 0xFFFFFFFF
