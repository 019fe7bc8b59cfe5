"""CPU restatement of the Swin Transformer backbone of configs/mask2former/pairnet_swinb.py:203-226
(`type="SwinTransformer"`, mmdet 2.25.1, [3P]) and BASELINE.json configs[3] (Swin-L).
TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this.

mmdet is neither vendored under /root/reference nor installed, so the published algorithm
(Liu et al. 2021; mmdet/models/backbones/swin.py) is restated from torch.nn primitives
with mmdet's module tree, so that an mmdet checkpoint's keys load unchanged:

  patch_embed.projection (Conv2d k=s=4, input padded bottom/right to a multiple of 4),
  patch_embed.norm; stages.S.blocks.J.{norm1, attn.w_msa.{relative_position_bias_table,
  relative_position_index, qkv, proj}, norm2, ffn.layers.0.0, ffn.layers.1};
  stages.S.downsample.{norm, reduction}; norm{S} on every output stage.

Block: x += W-MSA(norm1(x)); x += FFN(norm2(x)); odd blocks shift the windows by ws // 2.
(S)W-MSA: the normalised map is zero-padded (bottom/right) to a multiple of the window,
rolled by -shift, cut into ws x ws windows; attention = softmax((q * d^-0.5) k^T +
bias[relative index] + shift mask (0 / -100 between different wrap-around regions)) v; the
windows are merged, rolled back, and the padding is cropped.  Patch merging concatenates each
2 x 2 neighbourhood in nn.Unfold order (channel-major: index c * 4 + row * 2 + col) ->
LayerNorm(4C) -> Linear(4C, 2C, bias=False); odd maps are zero-padded bottom/right.

PARITY: the arithmetic is pinned against an independent implementation of the same
published model, HuggingFace `transformers.SwinBackbone` (importable in this image), by
tests/test_oracle.py::test_swin_oracle_matches_transformers_swin through `to_hf_state`
(key renaming + the unfold-order permutation of the patch-merging weights).  The mmdet KEY
NAMES are restated from memory and are unpinned (no mmdet checkpoint exists here).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class WindowMSA(nn.Module):
    def __init__(self, dims, heads, ws):
        super().__init__()
        self.heads, self.ws, self.scale = heads, ws, (dims // heads) ** -0.5
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * ws - 1) ** 2, heads))
        self.register_buffer("relative_position_index", relative_position_index(ws))
        self.qkv = nn.Linear(dims, dims * 3)
        self.proj = nn.Linear(dims, dims)

    def forward(self, x, mask):
        nwin_b, n, c = x.shape
        qkv = self.qkv(x).reshape(nwin_b, n, 3, self.heads, c // self.heads).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0] * self.scale, qkv[1], qkv[2]
        attn = q @ k.transpose(-2, -1)
        bias = self.relative_position_bias_table[self.relative_position_index.view(-1)]
        attn = attn + bias.view(n, n, -1).permute(2, 0, 1).unsqueeze(0)
        if mask is not None:
            nw = mask.shape[0]
            attn = (attn.view(nwin_b // nw, nw, self.heads, n, n) + mask[None, :, None]).view(
                -1, self.heads, n, n)
        attn = attn.softmax(-1)
        return self.proj((attn @ v).transpose(1, 2).reshape(nwin_b, n, c))


def relative_position_index(ws):
    """[ws*ws, ws*ws]: (dy + ws - 1) * (2 ws - 1) + (dx + ws - 1) for query p, key q."""
    ys, xs = torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")
    ys, xs = ys.reshape(-1), xs.reshape(-1)
    dy = ys[:, None] - ys[None, :] + ws - 1
    dx = xs[:, None] - xs[None, :] + ws - 1
    return dy * (2 * ws - 1) + dx


def windows_of(x, ws):
    b, h, w, c = x.shape
    return x.view(b, h // ws, ws, w // ws, ws, c).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws, c)


def shift_mask(hp, wp, ws, shift):
    region = lambda n: (torch.arange(n) >= n - ws).long() + (torch.arange(n) >= n - shift).long()
    label = (region(hp)[:, None] * 3 + region(wp)[None, :]).float()
    lw = windows_of(label[None, :, :, None], ws).squeeze(-1)          # [nW, ws*ws]
    diff = lw[:, None, :] - lw[:, :, None]
    return torch.where(diff != 0, torch.full_like(diff, -100.0), torch.zeros_like(diff))


class ShiftWindowMSA(nn.Module):
    def __init__(self, dims, heads, ws, shift):
        super().__init__()
        self.ws, self.shift = ws, shift
        self.w_msa = WindowMSA(dims, heads, ws)

    def forward(self, x, hw):
        b, _, c = x.shape
        h, w = hw
        ws, s = self.ws, self.shift
        x = x.view(b, h, w, c)
        x = F.pad(x, (0, 0, 0, (ws - w % ws) % ws, 0, (ws - h % ws) % ws))
        hp, wp = x.shape[1:3]
        mask = None
        if s > 0:
            x = torch.roll(x, (-s, -s), (1, 2))
            mask = shift_mask(hp, wp, ws, s)
        y = self.w_msa(windows_of(x, ws), mask)
        y = y.view(b, hp // ws, wp // ws, ws, ws, c).permute(0, 1, 3, 2, 4, 5).reshape(b, hp, wp, c)
        if s > 0:
            y = torch.roll(y, (s, s), (1, 2))
        return y[:, :h, :w].reshape(b, h * w, c)


class FFN(nn.Module):
    def __init__(self, dims, hidden):
        super().__init__()
        self.layers = nn.Sequential(nn.Sequential(nn.Linear(dims, hidden), nn.GELU()),
                                    nn.Linear(hidden, dims))

    def forward(self, x):
        return self.layers(x)


class SwinBlock(nn.Module):
    def __init__(self, dims, heads, ws, shift, mlp_ratio):
        super().__init__()
        self.norm1 = nn.LayerNorm(dims)
        self.attn = ShiftWindowMSA(dims, heads, ws, shift)
        self.norm2 = nn.LayerNorm(dims)
        self.ffn = FFN(dims, int(mlp_ratio * dims))

    def forward(self, x, hw):
        x = x + self.attn(self.norm1(x), hw)
        return x + self.ffn(self.norm2(x))


class PatchMerging(nn.Module):
    def __init__(self, dims):
        super().__init__()
        self.norm = nn.LayerNorm(4 * dims)
        self.reduction = nn.Linear(4 * dims, 2 * dims, bias=False)

    def forward(self, x, hw):
        b, _, c = x.shape
        h, w = hw
        x = F.pad(x.view(b, h, w, c), (0, 0, 0, w % 2, 0, h % 2))
        h2, w2 = x.shape[1] // 2, x.shape[2] // 2
        # nn.Unfold(2, stride 2) channel order: c * 4 + row * 2 + col
        x = x.view(b, h2, 2, w2, 2, c).permute(0, 1, 3, 5, 2, 4).reshape(b, h2 * w2, 4 * c)
        return self.reduction(self.norm(x)), (h2, w2)


class Stage(nn.Module):
    def __init__(self, dims, depth, heads, ws, mlp_ratio, downsample):
        super().__init__()
        self.blocks = nn.ModuleList(
            [SwinBlock(dims, heads, ws, 0 if j % 2 == 0 else ws // 2, mlp_ratio)
             for j in range(depth)])
        self.downsample = PatchMerging(dims) if downsample else None


class PatchEmbed(nn.Module):
    def __init__(self, dims, patch):
        super().__init__()
        self.patch = patch
        self.projection = nn.Conv2d(3, dims, patch, stride=patch)
        self.norm = nn.LayerNorm(dims)

    def forward(self, img):
        p = self.patch
        h, w = img.shape[-2:]
        x = self.projection(F.pad(img, (0, (p - w % p) % p, 0, (p - h % p) % p)))
        hw = tuple(x.shape[-2:])
        return self.norm(x.flatten(2).transpose(1, 2)), hw


class OracleSwin(nn.Module):
    """Swin-T/S/B/L by arguments; the reference's Swin-B is embed_dims=128, depths (2,2,18,2),
    num_heads (4,8,16,32), window_size 12 (pairnet_swinb.py:205-209); Swin-L is 192 /
    (2,2,18,2) / (6,12,24,48) / 12."""

    def __init__(self, embed_dims=128, depths=(2, 2, 18, 2), num_heads=(4, 8, 16, 32),
                 window_size=12, mlp_ratio=4, patch_size=4, out_indices=(0, 1, 2, 3), **unused):
        super().__init__()
        self.out_indices = tuple(out_indices)
        self.patch_embed = PatchEmbed(embed_dims, patch_size)
        self.stages = nn.ModuleList()
        dims = embed_dims
        self.num_features = []
        for i, (d, nh) in enumerate(zip(depths, num_heads)):
            self.stages.append(Stage(dims, d, nh, window_size, mlp_ratio, i < len(depths) - 1))
            self.num_features.append(dims)
            dims *= 2
        for i in self.out_indices:
            setattr(self, "norm%d" % i, nn.LayerNorm(self.num_features[i]))
        self.eval()

    @torch.no_grad()
    def forward(self, img):
        x, hw = self.patch_embed(img)
        outs = []
        for i, st in enumerate(self.stages):
            for blk in st.blocks:
                x = blk(x, hw)
            if i in self.out_indices:
                y = getattr(self, "norm%d" % i)(x)
                outs.append(y.view(-1, hw[0], hw[1], y.shape[-1]).permute(0, 3, 1, 2).contiguous())
            if st.downsample is not None:
                x, hw = st.downsample(x, hw)
        return tuple(outs)


def seeded_swin_state(model, seed):
    """Deterministic state dict for `model` (numpy PCG64): fan-in scaled weights, LayerNorm
    affine away from the identity, a relative-position table of visible size."""
    import numpy as np
    rng = np.random.default_rng(seed)
    out = {}
    for k, v in model.state_dict().items():
        shape = tuple(v.shape)
        if k.endswith("relative_position_index"):
            out[k] = v.clone()
        elif "norm" in k:
            lo, hi = (0.5, 1.5) if k.endswith("weight") else (-0.2, 0.2)
            out[k] = torch.from_numpy(rng.uniform(lo, hi, shape).astype(np.float32))
        elif k.endswith("relative_position_bias_table"):
            out[k] = torch.from_numpy(rng.normal(0, 0.5, shape).astype(np.float32))
        elif k.endswith("bias"):
            out[k] = torch.from_numpy(rng.normal(0, 0.1, shape).astype(np.float32))
        else:
            fan_in = int(np.prod(shape[1:]))
            out[k] = torch.from_numpy((rng.normal(0, 1, shape) * fan_in ** -0.5).astype(np.float32))
    return out


def to_hf_state(sd, depths):
    """mmdet-layout state dict -> `transformers.SwinBackbone` keys (pin test only).
    The patch-merging weights go from nn.Unfold order (c * 4 + row * 2 + col) to the
    original implementation's concatenation order ((col * 2 + row) * C + c)."""
    out = {}
    ren = lambda k: k.replace("norm1", "layernorm_before").replace("norm2", "layernorm_after") \
        .replace("attn.w_msa.proj", "attention.o_proj") \
        .replace("attn.w_msa.relative_position_bias_table",
                 "attention.relative_position_bias.relative_position_bias_table") \
        .replace("ffn.layers.0.0", "mlp.fc1").replace("ffn.layers.1", "mlp.fc2")

    def unfold_to_concat(t):                       # last dim 4C
        c = t.shape[-1] // 4
        return t.reshape(*t.shape[:-1], c, 2, 2).permute(*range(t.dim() - 1), t.dim() + 1,
                                                        t.dim(), t.dim() - 1) \
            .reshape(*t.shape[:-1], 4 * c)         # [.., c, row, col] -> [.., col, row, c]

    for k, v in sd.items():
        if k.endswith("relative_position_index"):
            continue
        if k.startswith("patch_embed.projection"):
            out["swin.embeddings.patch_embeddings." + k[len("patch_embed."):]] = v
        elif k.startswith("patch_embed.norm"):
            out["swin.embeddings." + k[len("patch_embed."):]] = v
        elif k.startswith("norm"):
            i = int(k[4:k.index(".")])
            out["hidden_states_norms.stage%d.%s" % (i + 1, k.split(".")[-1])] = v
        elif ".downsample." in k:
            out["swin.encoder.layers." + k[len("stages."):]] = unfold_to_concat(v)
        elif ".attn.w_msa.qkv." in k:
            base = "swin.encoder.layers." + k[len("stages."):k.index("attn.w_msa")] + "attention."
            q, kk, vv = v.chunk(3, 0)
            leaf = k.split(".")[-1]
            out[base + "q_proj." + leaf], out[base + "k_proj." + leaf] = q, kk
            out[base + "v_proj." + leaf] = vv
        else:
            out["swin.encoder.layers." + ren(k[len("stages."):])] = v
    return out
