"""Swin Transformer backbone on MI355X (BASELINE.json configs[3]):
the `type="SwinTransformer"` backbone of configs/mask2former/pairnet_swinb.py:203-226
([3P] mmdet 2.25.1 SwinTransformer) on the head's fp32-MFMA kernels.

Token maps stay channel-last rows [B][H*W][C] end to end.  Per block: `k_ln_rows` (norm1)
-> qkv GEMM -> `k_window_attn` (window padding, cyclic shift, partition, relative position
bias, shift mask, merge, un-shift and crop are index arithmetic inside the kernel: no
roll / pad / permute copies) -> proj GEMM with the residual in its epilogue -> `k_ln_rows`
(norm2) -> FFN GEMM with the erf-GELU epilogue -> FFN GEMM with the residual epilogue.
Patch embedding = `k_patch_im2col` + GEMM (K = 48 -> 64) + LayerNorm; patch merging =
`k_ln_rows<merge>` (2x2 gather + LayerNorm(4C)) + GEMM.  The outputs are NCHW-shaped
tensors in `torch.channels_last` memory format, which `CrossHead2` reads directly.

State-dict names are mmdet's (restated in oracle/swin.py): `patch_embed.projection`,
`patch_embed.norm`, `stages.S.blocks.J.{norm1,attn.w_msa.{relative_position_bias_table,
relative_position_index,qkv,proj},norm2,ffn.layers.0.0,ffn.layers.1}`,
`stages.S.downsample.{norm,reduction}`, `norm{S}`.  Head dim must be 32 (Swin-T/S/B/L).
No CPU path.
"""
from collections import OrderedDict

import torch

from . import hip
from .plans import Arena, PlanCache, measure_bytes

EPS = 1e-5


class SwinTransformerHip:
    """Drop-in for the detector's `backbone(img) -> tuple of feature maps`."""

    def __init__(self, embed_dims=128, depths=(2, 2, 18, 2), num_heads=(4, 8, 16, 32),
                 window_size=12, mlp_ratio=4, patch_size=4, out_indices=(0, 1, 2, 3),
                 qkv_bias=True, patch_norm=True, use_abs_pos_embed=False, **unused):
        if patch_size != 4 or not qkv_bias or not patch_norm or use_abs_pos_embed:
            raise NotImplementedError("patch_size 4, qkv_bias, patch_norm and no absolute "
                                      "position embedding (the reference's configuration)")
        if len(depths) != len(num_heads):
            raise ValueError("depths and num_heads must have one entry per stage")
        self.embed_dims, self.depths, self.num_heads = embed_dims, tuple(depths), tuple(num_heads)
        self.ws, self.mlp_ratio = int(window_size), mlp_ratio
        self.out_indices = tuple(out_indices)
        self.num_features = [embed_dims * 2 ** i for i in range(len(depths))]
        for c, nh in zip(self.num_features, self.num_heads):
            if c != nh * 32:
                raise NotImplementedError("head dim must be 32 (embed_dims*2^i == 32*num_heads[i])")
        if self.ws * self.ws > 169:
            raise NotImplementedError("window_size <= 13")
        self._params = OrderedDict((k, torch.zeros(s, dtype=dt))
                                   for k, (s, dt) in self._param_shapes().items())
        self.init_weights(0)
        self.device, self.w, self._plans = None, None, PlanCache()
        self.grid_reserve = 0
        # the blocks' four GEMMs (qkv, proj, FFN): "bf16x3" = on the bf16 matrix pipe from
        # operands their producers store as three exact bf16 planes (csrc/gemm_s3.hip; norm1 /
        # norm2 write S3, the FFN's hidden rows exist only as S3), "fp32" = the exact-fp32 MFMA
        # kernels.  Patch embedding and patch merging stay fp32 either way.
        self.gemm_arithmetic = "bf16x3"
        self.attn_s3_out = True      # bf16x3: window attention writes proj's operand pre-split

    # ------------------------------------------------------------------ parameters
    def _param_shapes(self):
        s, f, ws = OrderedDict(), torch.float32, self.ws
        C0 = self.embed_dims
        s["patch_embed.projection.weight"] = ((C0, 3, 4, 4), f)
        s["patch_embed.projection.bias"] = ((C0,), f)
        s["patch_embed.norm.weight"], s["patch_embed.norm.bias"] = ((C0,), f), ((C0,), f)
        for i, (d, nh) in enumerate(zip(self.depths, self.num_heads)):
            C = self.num_features[i]
            hid = int(self.mlp_ratio * C)
            for j in range(d):
                p = "stages.%d.blocks.%d." % (i, j)
                s[p + "norm1.weight"], s[p + "norm1.bias"] = ((C,), f), ((C,), f)
                s[p + "attn.w_msa.relative_position_bias_table"] = (((2 * ws - 1) ** 2, nh), f)
                s[p + "attn.w_msa.relative_position_index"] = ((ws * ws, ws * ws), torch.int64)
                s[p + "attn.w_msa.qkv.weight"], s[p + "attn.w_msa.qkv.bias"] = ((3 * C, C), f), ((3 * C,), f)
                s[p + "attn.w_msa.proj.weight"], s[p + "attn.w_msa.proj.bias"] = ((C, C), f), ((C,), f)
                s[p + "norm2.weight"], s[p + "norm2.bias"] = ((C,), f), ((C,), f)
                s[p + "ffn.layers.0.0.weight"], s[p + "ffn.layers.0.0.bias"] = ((hid, C), f), ((hid,), f)
                s[p + "ffn.layers.1.weight"], s[p + "ffn.layers.1.bias"] = ((C, hid), f), ((C,), f)
            if i < len(self.depths) - 1:
                p = "stages.%d.downsample." % i
                s[p + "norm.weight"], s[p + "norm.bias"] = ((4 * C,), f), ((4 * C,), f)
                s[p + "reduction.weight"] = ((2 * C, 4 * C), f)
        for i in self.out_indices:
            s["norm%d.weight" % i] = ((self.num_features[i],), f)
            s["norm%d.bias" % i] = ((self.num_features[i],), f)
        return s

    def _relative_index(self):
        ws = self.ws
        ys, xs = torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")
        ys, xs = ys.reshape(-1), xs.reshape(-1)
        return (ys[:, None] - ys[None, :] + ws - 1) * (2 * ws - 1) + xs[:, None] - xs[None, :] + ws - 1

    def init_weights(self, seed=0):
        """Default init in the spirit of mmdet's (trunc-normal 0.02 weights, unit norms)."""
        g = torch.Generator().manual_seed(seed)
        for k, v in self._params.items():
            if k.endswith("relative_position_index"):
                v.copy_(self._relative_index())
            elif k.endswith("weight") and v.dim() == 1:
                v.fill_(1.0)
            elif k.endswith("bias"):
                v.zero_()
            elif k.endswith("projection.weight"):
                v.copy_(torch.randn(v.shape, generator=g) * (1.0 / 48) ** 0.5)
            else:
                v.copy_((torch.randn(v.shape, generator=g) * 0.02).clamp_(-0.04, 0.04))
        self.w = None

    def state_dict(self):
        return OrderedDict((k, v.clone()) for k, v in self._params.items())

    def load_state_dict(self, sd, strict=True):
        missing = [k for k in self._params if k not in sd]
        unexpected = [k for k in sd if k not in self._params]
        if strict and (missing or unexpected):
            raise RuntimeError("state_dict mismatch: missing %s unexpected %s"
                               % (missing[:5], unexpected[:5]))
        for k, p in self._params.items():
            if k in sd:
                if tuple(sd[k].shape) != tuple(p.shape):
                    raise RuntimeError("shape mismatch for %s" % k)
                if k.endswith("relative_position_index"):
                    if not torch.equal(sd[k].cpu().long(), self._relative_index()):
                        raise RuntimeError("%s is not the standard Swin index" % k)
                    continue
                p.copy_(sd[k].detach().to(p.dtype).cpu())
        self.w = None
        return missing, unexpected

    def to(self, device):
        self.device, self.w, self._plans = torch.device(device), None, PlanCache()
        self._arenas = {}
        return self

    def eval(self):
        return self

    def _pack(self):
        if self.device is None or self.device.type != "cuda":
            raise RuntimeError("SwinTransformerHip runs on an MI355X only: call .to('cuda:0'); "
                               "there is no CPU path")
        hip.lib()
        dev, P, w = self.device, self._params, {}
        for k, v in P.items():
            if k.endswith("relative_position_index"):
                continue
            if k == "patch_embed.projection.weight":
                pw = torch.zeros(v.shape[0], 64)
                pw[:, :48] = v.reshape(v.shape[0], 48)            # c*16 + ky*4 + kx
                v = pw
            elif k.endswith("relative_position_bias_table"):
                v = v.t()                                         # [heads][(2ws-1)^2]
            elif ".downsample." in k:
                # nn.Unfold order c*4 + (row*2+col) -> neighbour-major (row*2+col)*C + c
                c = v.shape[-1] // 4
                v = v.reshape(*v.shape[:-1], c, 4).transpose(-1, -2).reshape(v.shape)
            w[k] = v.contiguous().to(dev)
        # the block GEMMs' weights as S3 operands, split once
        for k in list(w):
            if k.endswith(("qkv.weight", "proj.weight", "ffn.layers.0.0.weight", "ffn.layers.1.weight")):
                s3 = torch.empty(hip.s3_floats(*w[k].shape), device=dev, dtype=torch.float32)
                hip.s3_split(w[k], s3)
                w[k + ".s3"] = s3
        self.w = w

    class _Plan:
        def busy_events(self):
            out = []
            for st in getattr(self, "streams", {}).values():
                ev = torch.cuda.Event()
                ev.record(st)
                out.append(ev)
            return out

    def _arena(self, slot):
        """One flat buffer per slot; every (batch, image size) plan of the slot is a set of
        views of it (plans.py: a keep-ratio evaluation pass meets hundreds of sizes)."""
        arenas = self.__dict__.setdefault("_arenas", {})
        a = arenas.get(slot)
        if a is None:
            a = arenas[slot] = Arena(self.device, on_grow=lambda a, s=slot: self._plans.drop(
                lambda k: k[3] == s))
        return a

    def _layout_for(self, dims):
        def layout(E):
            pl = SwinTransformerHip._Plan()
            self._layout(pl, E, *dims)
            return pl
        return layout

    def _measure(self, dims):
        return measure_bytes(self._layout_for(dims))

    def reserve(self, batch, H, W, slots=(0,)):
        """Size the arenas of `slots` up front for [batch, 3, H, W] images (optional)."""
        if self.w is None:
            self._pack()
        for s in slots:
            self._arena(s).reserve((batch, H, W), self._measure)

    def arena_bytes(self):
        return sum(a.capacity for a in self.__dict__.get("_arenas", {}).values())

    def feature_shapes(self, H, W):
        """[(h, w)] of the output stages for an H x W image (4x4 patches, then /2 per stage)."""
        h, w = -(-H // 4), -(-W // 4)
        out = []
        for i in range(len(self.depths)):
            if i in self.out_indices:
                out.append((h, w))
            h, w = -(-h // 2), -(-w // 2)
        return out

    def _plan(self, B, H, W, slot=0):
        key = (B, H, W, slot)
        if key in self._plans:
            return self._plans[key]
        if self.w is None:
            self._pack()
        dims = (B, H, W)
        pl = self._arena(slot).carve(self._layout_for(dims), dims, self._measure)
        pl.streams = {}
        self._plans[key] = pl
        return pl

    def _layout(self, pl, E, B, H, W):
        """Every activation buffer as a view of the slot's arena (sizes non-decreasing in B, H, W)."""
        h, wd = -(-H // 4), -(-W // 4)
        pl.hw = []
        for i in range(len(self.depths)):
            pl.hw.append((h, wd))
            h, wd = -(-h // 2), -(-wd // 2)
        n0 = B * pl.hw[0][0] * pl.hw[0][1]
        pl.cols = E(n0, 64)
        # the widest intermediate of every stage has the same size: tokens halve*halve, C doubles
        tok_c = max(B * hh * ww * c for (hh, ww), c in zip(pl.hw, self.num_features))
        pl.x = [E(B * hh * ww, c) for (hh, ww), c in zip(pl.hw, self.num_features)]
        pl.xn = E(tok_c)                       # normalised tokens / attention output
        pl.ao = E(tok_c)
        pl.qkv = E(3 * tok_c)
        pl.hid = E(int(self.mlp_ratio) * tok_c)
        # S3 operands (gemm_arithmetic == "bf16x3"): normalised tokens / attention output (shared:
        # never live together), hidden rows; 1.5 x the fp32 size + the padding of the last row block
        s3max = lambda mul: max(hip.s3_floats(B * hh * ww, mul * c)
                                for (hh, ww), c in zip(pl.hw, self.num_features))
        pl.xn_s3, pl.hid_s3 = E(s3max(1)), E(s3max(int(self.mlp_ratio)))
        pl.merged = E(max([B * hh * ww * 2 * c                 # [ceil-halved tokens][4C]
                           for (hh, ww), c in zip(pl.hw[1:], self.num_features[1:])] + [4]))
        pl.out = {i: E(B, pl.hw[i][0], pl.hw[i][1], self.num_features[i]) for i in self.out_indices}
        pl.scratch = E(B * 8 * 1024 * 1024)

    # ------------------------------------------------------------------ forward
    @torch.no_grad()
    @hip.on_device
    @hip.with_reserve
    def forward(self, img, slot=0):
        """img [B,3,H,W] fp32 NCHW on the GPU -> one NCHW-shaped, channels_last feature map
        per out index (views of per-shape buffers that the next call overwrites)."""
        if not img.is_cuda or img.dtype != torch.float32 or img.dim() != 4 or img.shape[1] != 3:
            raise RuntimeError("img must be a [B,3,H,W] fp32 device tensor")
        if self.device is None:
            self.to(img.device)
        img = img.contiguous()
        B, _, H, W = img.shape
        pl = self._plan(B, H, W, slot)
        cur = torch.cuda.current_stream(self.device)
        pl.streams[cur.cuda_stream] = cur
        w, ws = self.w, self.ws
        lin = lambda x, wk, bk, out, **kw: hip.linear(x, w[wk], w[bk] if bk else None, out,
                                                      scratch=pl.scratch, **kw)
        hip.patch_im2col4(img, pl.cols, B, H, W)
        x = pl.x[0]
        hip.linear(pl.cols, w["patch_embed.projection.weight"], w["patch_embed.projection.bias"],
                   pl.xn[:x.numel()].view_as(x))
        hip.layernorm_rows(pl.xn[:x.numel()].view_as(x), w["patch_embed.norm.weight"],
                           w["patch_embed.norm.bias"], x, EPS)
        for i, (d, nh) in enumerate(zip(self.depths, self.num_heads)):
            C, (h, wd) = self.num_features[i], pl.hw[i]
            n = B * h * wd
            x = pl.x[i]
            xn, ao = pl.xn[:n * C].view(n, C), pl.ao[:n * C].view(n, C)
            qkv = pl.qkv[:3 * n * C].view(n, 3 * C)
            hid = pl.hid[:n * int(self.mlp_ratio * C)].view(n, -1)
            s3 = self.gemm_arithmetic == "bf16x3"
            for j in range(d):
                p = "stages.%d.blocks.%d." % (i, j)
                if s3:
                    F = int(self.mlp_ratio * C)
                    hip.layernorm_rows_s3(x, w[p + "norm1.weight"], w[p + "norm1.bias"], pl.xn_s3, EPS)
                    hip.gemm_s3(pl.xn_s3, w[p + "attn.w_msa.qkv.weight.s3"], n, 3 * C, C,
                                bias=w[p + "attn.w_msa.qkv.bias"], out=qkv)
                    if self.attn_s3_out:
                        hip.window_attention_s3(qkv, w[p + "attn.w_msa.qkv.bias"],
                                                w[p + "attn.w_msa.relative_position_bias_table"],
                                                pl.xn_s3, B, h, wd, C, nh, ws,
                                                0 if j % 2 == 0 else ws // 2)
                    else:
                        hip.window_attention(qkv, w[p + "attn.w_msa.qkv.bias"],
                                             w[p + "attn.w_msa.relative_position_bias_table"], ao,
                                             B, h, wd, C, nh, ws, 0 if j % 2 == 0 else ws // 2)
                        hip.s3_split(ao, pl.xn_s3)
                    hip.gemm_s3(pl.xn_s3, w[p + "attn.w_msa.proj.weight.s3"], n, C, C,
                                bias=w[p + "attn.w_msa.proj.bias"], out=x, res=x)
                    hip.layernorm_rows_s3(x, w[p + "norm2.weight"], w[p + "norm2.bias"], pl.xn_s3, EPS)
                    hip.gemm_s3(pl.xn_s3, w[p + "ffn.layers.0.0.weight.s3"], n, F, C,
                                bias=w[p + "ffn.layers.0.0.bias"], gelu=True, out_s3=pl.hid_s3)
                    hip.gemm_s3(pl.hid_s3, w[p + "ffn.layers.1.weight.s3"], n, C, F,
                                bias=w[p + "ffn.layers.1.bias"], out=x, res=x)
                    continue
                hip.layernorm_rows(x, w[p + "norm1.weight"], w[p + "norm1.bias"], xn, EPS)
                lin(xn, p + "attn.w_msa.qkv.weight", p + "attn.w_msa.qkv.bias", qkv)
                hip.window_attention(qkv, w[p + "attn.w_msa.qkv.bias"],
                                     w[p + "attn.w_msa.relative_position_bias_table"], ao, B, h, wd,
                                     C, nh, ws, 0 if j % 2 == 0 else ws // 2)
                lin(ao, p + "attn.w_msa.proj.weight", p + "attn.w_msa.proj.bias", x, res=x)
                hip.layernorm_rows(x, w[p + "norm2.weight"], w[p + "norm2.bias"], xn, EPS)
                lin(xn, p + "ffn.layers.0.0.weight", p + "ffn.layers.0.0.bias", hid, gelu=True)
                lin(hid, p + "ffn.layers.1.weight", p + "ffn.layers.1.bias", x, res=x)
            if i in pl.out:
                hip.layernorm_rows(x, w["norm%d.weight" % i], w["norm%d.bias" % i],
                                   pl.out[i].view(n, C), EPS)
            if i < len(self.depths) - 1:
                p = "stages.%d.downsample." % i
                n2 = B * pl.hw[i + 1][0] * pl.hw[i + 1][1]
                mg = pl.merged[:n2 * 4 * C].view(n2, 4 * C)
                hip.patch_merge_ln(x, w[p + "norm.weight"], w[p + "norm.bias"], mg, B, h, wd, C, EPS)
                lin(mg, p + "reduction.weight", None, pl.x[i + 1])
        return tuple(pl.out[i].permute(0, 3, 1, 2) for i in self.out_indices)

    __call__ = forward
