"""ResNet-50 on MI355X: the reference's backbone
(configs/mask2former/pairnet.py:9-19: mmdet ResNet depth 50, style "pytorch", frozen
BatchNorm, out_indices 0-3) on the head's own fp32-MFMA kernels.

Every convolution is the persistent implicit-GEMM kernel of csrc/gemm.hip (1x1 layers are
plain GEMMs over the channel-last pixels); BatchNorm (eval mode) is folded into the weights
and a bias at pack time, ReLU and the residual add live in the GEMM epilogue, activations
stay channel-last end to end.  The four outputs are returned as NCHW-shaped tensors in
`torch.channels_last` memory format: mmdet's contract for `feats`, and `CrossHead2` reads
that format directly (row-major 1x1 input convolutions, no transpose pass).

State-dict names are mmdet's / torchvision's (`conv1.weight`, `bn1.running_mean`,
`layer3.4.conv2.weight`, `layer2.0.downsample.0.weight`, ...).  No CPU path.
"""
from collections import OrderedDict

import torch

from . import hip
from . import plans
from .plans import Arena, PlanCache, measure_bytes

BLOCKS = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3)}   # bottleneck ResNets (mmdet arch_settings)
EPS = 1e-5


def _stages(depth):
    return tuple(zip((64, 128, 256, 512), BLOCKS[depth]))


def _param_shapes(depth=50):
    STAGES = _stages(depth)
    s = OrderedDict()

    def bn(prefix, c):
        for n in ("weight", "bias", "running_mean", "running_var"):
            s["%s.%s" % (prefix, n)] = (c,)
        s[prefix + ".num_batches_tracked"] = ()

    s["conv1.weight"] = (64, 3, 7, 7)
    bn("bn1", 64)
    cin = 64
    for i, (planes, blocks) in enumerate(STAGES):
        for b in range(blocks):
            p = "layer%d.%d." % (i + 1, b)
            s[p + "conv1.weight"] = (planes, cin, 1, 1)
            bn(p + "bn1", planes)
            s[p + "conv2.weight"] = (planes, planes, 3, 3)
            bn(p + "bn2", planes)
            s[p + "conv3.weight"] = (planes * 4, planes, 1, 1)
            bn(p + "bn3", planes * 4)
            if b == 0:
                s[p + "downsample.0.weight"] = (planes * 4, cin, 1, 1)
                bn(p + "downsample.1", planes * 4)
            cin = planes * 4
    return s


class ResNet50Hip:
    """Drop-in for the detector's `backbone(img) -> (C2, C3, C4, C5)`."""

    def __init__(self, depth=50, **unused):
        if depth not in BLOCKS:
            raise NotImplementedError("bottleneck ResNets: depth 50 or 101")
        self.depth, self.stages = depth, _stages(depth)
        self._params = OrderedDict(
            (k, torch.zeros(shape, dtype=torch.int64 if k.endswith("tracked") else torch.float32))
            for k, shape in _param_shapes(depth).items())
        for k, v in self._params.items():          # identity BatchNorm until weights are loaded
            if k.endswith(("running_var",)) or (k.endswith(".weight") and v.dim() == 1):
                v.fill_(1.0)
        g = torch.Generator().manual_seed(0)
        for k, v in self._params.items():
            if v.dim() == 4:
                fan_in = v.shape[1] * v.shape[2] * v.shape[3]
                v.copy_(torch.randn(v.shape, generator=g) * (2.0 / fan_in) ** 0.5)
        self.device, self.w, self._plans, self._arenas = None, None, PlanCache(), {}
        # replay the forward pass as one hipGraph once a (shape, slot) has run `graph_after`
        # times eagerly (no host synchronisation either way: plans are views of the slot's
        # arena, plans.py)
        self.use_graphs = False
        self.graph_after = 1
        # persistent-GEMM workgroup slots left free for concurrent streams (hip.reserve_slots)
        self.grid_reserve = 0
        # 3x3 stride-1 layers of >= 128 channels (stages 2-4): "winograd" = F(2x2,3x3) around
        # one batched GEMM (2.25x fewer multiplications); "winograd4" = F(4x4,3x3) (4x fewer)
        # on the maps where its 36 GEMMs of ceil(h/4)*ceil(w/4) rows cover fewer padded 64-row
        # tiles than the 16 of F(2x2,3x3) -- stages 2 and 3 at 800x1333, not the 25x42 map of
        # stage 4; "direct" = implicit GEMM
        self.conv_algo = "winograd4"
        # narrowest 3x3 layer that takes the Winograd form (128: stages 2-4; 64 adds the three
        # K = 64 layers of stage 1 -- measured in round 4, LABNOTES.md 6.0-r4)
        self.wino_min_planes = 128
        # each bottleneck's last 1x1 convolution (+ BN + shortcut + ReLU; N = 4 planes, K =
        # planes) on the bf16 matrix pipe from a pre-split operand (csrc/gemm_s3.hip) for
        # planes >= this; 0 = never (default): measured in round 6 with a split pass in front
        # (the 3x3 convolution's transforms write fp32), same run: 223.5 images/s without,
        # 218.4 / 221.2 / 222.2 from 64 / 128 / 256 planes on (labnotes R6.5) -- these GEMMs have
        # K = 64..512 and 4 200- / 1 050-row maps: two to sixteen k-stages per tile and half-empty
        # tile rounds
        self.s3_conv3_min_planes = 0

    def _weights_version(self):
        return sum(p._version for p in self._params.values())

    def parameters(self):
        """The live (host) parameter / buffer tensors; in-place updates are picked up by the
        next forward (folded weights and captured graphs are rebuilt)."""
        return list(self._params.values())

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
                p.copy_(sd[k].detach().to(p.dtype).cpu())
        self.w, self._plans = None, PlanCache()     # plans hold graphs captured on the old weights
        return missing, unexpected

    def to(self, device):
        self.device, self.w, self._plans = torch.device(device), None, PlanCache()
        self._arenas = {}
        return self

    def eval(self):
        return self

    # ------------------------------------------------------------------ packing
    def _fold(self, conv, bn):
        """conv weight [co][ci][kh][kw] + eval-mode BatchNorm -> ([co][(ky*KW+kx)*ci + ci'],
        bias[co]) on the device:  bn(conv(x)) = conv(x) * s + (beta - mean * s),
        s = gamma / sqrt(var + eps)."""
        P = self._params
        w = P[conv + ".weight"].double()
        s = P[bn + ".weight"].double() / torch.sqrt(P[bn + ".running_var"].double() + EPS)
        b = P[bn + ".bias"].double() - P[bn + ".running_mean"].double() * s
        w = (w * s.view(-1, 1, 1, 1)).float()
        return w, b.float()

    def _pack(self):
        if self.device is None or self.device.type != "cuda":
            raise RuntimeError("ResNet50Hip runs on an MI355X only: call .to('cuda:0'); there "
                               "is no CPU path")
        hip.lib()
        dev, w = self.device, {}
        cw, cb = self._fold("conv1", "bn1")
        stem = torch.zeros(64, 160)
        stem[:, :147] = cw.reshape(64, 147)                # k = c*49 + ky*7 + kx
        w["stem.w"], w["stem.b"] = stem.to(dev), cb.to(dev)
        for i, (planes, blocks) in enumerate(self.stages):
            for b in range(blocks):
                p = "layer%d.%d." % (i + 1, b)
                for conv, bn in (("conv1", "bn1"), ("conv2", "bn2"), ("conv3", "bn3"),
                                 ("downsample.0", "downsample.1")):
                    if p + conv + ".weight" not in self._params:
                        continue
                    cw, cb = self._fold(p + conv, p + bn)
                    co = cw.shape[0]
                    w[p + conv + ".w"] = cw.permute(0, 2, 3, 1).reshape(co, -1).contiguous().to(dev)
                    w[p + conv + ".b"] = cb.to(dev)
                    # stride-1 3x3 layers of >= 128 channels also get their Winograd F(2x2,3x3)
                    # weights (stages 2-4; the 64-channel layers of stage 1 would be K = 64
                    # GEMMs, slower than the direct form)
                    if conv == "conv2" and co >= 64 and not (b == 0 and i > 0):
                        w[p + "conv2.wino"] = hip.winograd_weights(cw.to(dev))
                        w[p + "conv2.wino4"] = hip.winograd43_weights(cw.to(dev))
        # conv3 (BN folded) as S3 operands for the bf16-pipe GEMM, split once
        for k in list(w):
            if k.endswith("conv3.w"):
                s3 = torch.empty(hip.s3_floats(*w[k].shape), device=dev, dtype=torch.float32)
                hip.s3_split(w[k], s3)
                w[k + ".s3"] = s3
        self.w = w
        self._packed_version = self._weights_version()

    class _Plan:
        """Views of one slot's arena for one (batch, image size) + the captured hipGraph."""

        def busy_events(self):
            out = []
            for st in self.streams.values():
                ev = torch.cuda.Event()
                ev.record(st)
                out.append(ev)
            return out

    def _arena(self, slot):
        a = self._arenas.get(slot)
        if a is None:
            a = self._arenas[slot] = Arena(self.device, on_grow=lambda a, s=slot: self._plans.drop(
                lambda k: k[3] == s))
        return a

    def _layout_for(self, dims):
        def layout(E):
            pl = ResNet50Hip._Plan()
            self._layout(pl, E, *dims)
            return pl
        return layout

    def _measure(self, dims):
        return measure_bytes(self._layout_for(dims))

    def reserve(self, batch, H, W, slots=(0,)):
        """Size the arenas of `slots` up front for [batch, 3, H, W] images (optional: they
        grow on demand, one device wait per growth)."""
        if self.w is None:
            self._pack()
        for s in slots:
            self._arena(s).reserve((batch, H, W), self._measure)

    def arena_bytes(self):
        return sum(a.capacity for a in self._arenas.values())

    @staticmethod
    def feature_shapes(H, W):
        """[(h, w)] of C2..C5 for an H x W image (7x7/2 stem, 3x3/2 pool, then /2 per stage)."""
        h, w = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        out = []
        for i in range(4):
            h, w = (h - 1) // 2 + 1, (w - 1) // 2 + 1
            out.append((h, w))
        return out

    def _plan(self, B, H, W, slot=0):
        key = (B, H, W, slot, self.conv_algo, self.wino_min_planes, self.s3_conv3_min_planes)   # (a plan's graph bakes them in)
        if key in self._plans:
            return self._plans[key]
        if self.w is None:
            self._pack()

        dims = (B, H, W)
        pl = self._arena(slot).carve(self._layout_for(dims), dims, self._measure)
        pl.graph = pl.outs = pl.last_img = None
        pl.calls, pl.staged, pl.streams = 0, False, {}
        self._plans[key] = pl
        return pl

    def _layout(self, pl, E, B, H, W):
        """Every activation buffer as a view of the slot's arena; sizes are non-decreasing in
        B, H and W (plans.py)."""
        pl.B = B
        pl.img = E(B, 3, H, W)        # staging copy of the image for graph replay (see forward)
        h1, w1 = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        h, wd = (h1 - 1) // 2 + 1, (w1 - 1) // 2 + 1
        pl.stem, pl.hw_stem = E(B, h1, w1, 64), (h1, w1)
        pl.pool = E(B, h, wd, 64)
        # split-K workspace for the late stages (few output tiles, long K): sized for the
        # largest S x M x N the library can ask for here
        pl.scratch = E(B * 16 * 1024 * 1024 // 2)
        pl.hw, pl.out, pl.t1, pl.t2, pl.idt, pl.ping, pl.f43 = [], [], [], [], [], [], []
        nwino = 0
        for i, (planes, blocks) in enumerate(self.stages):
            hin, win = h, wd
            if i > 0:
                h, wd = (h - 1) // 2 + 1, (wd - 1) // 2 + 1
            pl.hw.append(((hin, win), (h, wd)))
            pl.t1.append(E(B, hin, win, planes))       # conv1 output at the input resolution
            pl.t2.append(E(B, h, wd, planes))
            pl.idt.append(E(B, h, wd, planes * 4))
            pl.ping.append(E(B, h, wd, planes * 4))
            pl.out.append(E(B, h, wd, planes * 4))
            if planes >= self.wino_min_planes:
                t2, t4 = B * ((h + 1) // 2) * ((wd + 1) // 2), B * ((h + 3) // 4) * ((wd + 3) // 4)
                nwino = max(nwino, 16 * t2 * planes, 36 * t4 * planes)
                # F(4x4,3x3) where it multiplies fewer (64-row padded) GEMM rows
                pl.f43.append(36 * (-(-t4 // 64)) < 16 * (-(-t2 // 64)))
            else:
                pl.f43.append(False)
        pl.wV, pl.wM = E(max(nwino, 4)), E(max(nwino, 4))   # Winograd transform planes
        # the 3x3 convolution's output once more as an S3 operand (conv3's A, s3_conv3_min_planes)
        pl.t2_s3 = E(max(hip.s3_floats(B * hw[1][0] * hw[1][1], planes)
                         for hw, (planes, _) in zip(pl.hw, self.stages)))

    # ------------------------------------------------------------------ forward
    @torch.no_grad()
    @hip.on_device
    def forward(self, img, slot=0):
        """img [B,3,H,W] fp32 NCHW on the GPU -> (C2, C3, C4, C5) NCHW-shaped tensors in
        channels_last memory format (views of per-(shape, slot) buffers that the next call
        with the same slot overwrites: callers that keep several images in flight on
        different streams give each stream its own `slot`)."""
        if not img.is_cuda or img.dtype != torch.float32 or img.dim() != 4 or img.shape[1] != 3:
            raise RuntimeError("img must be a [B,3,H,W] fp32 device tensor")
        if self.device is None:
            self.to(img.device)
        if self.w is not None and self._packed_version != self._weights_version():
            self.w, self._plans = None, PlanCache(self._plans.max_plans)   # updated in place
        img = img.contiguous()
        B, _, H, W = img.shape
        pl = self._plan(B, H, W, slot)
        if getattr(pl, "reserve", None) != self.grid_reserve:   # a captured graph bakes it in
            pl.reserve, pl.graph = self.grid_reserve, None
        cur = torch.cuda.current_stream(self.device)
        pl.streams[cur.cuda_stream] = cur
        try:
            return self._forward_on(img, pl, cur)
        finally:
            plans.note_use(cur)

    def _forward_on(self, img, pl, cur):
        if not self.use_graphs:
            return self._run(img, pl)
        # hipGraph replay of the ~55 launches: captured on the caller's image buffer when it
        # comes back with the same one (a resident input, a preprocessing stage writing
        # into a fixed buffer), otherwise on the plan's staging view pl.img that each call is
        # copied into.  Captured after `graph_after` eager calls, with no host wait
        # (head.CrossHead2._capture).
        ptr = img.data_ptr()
        if pl.graph is not None and not pl.staged and ptr != pl.static_img.data_ptr():
            pl.graph, pl.staged = None, True        # the caller rotates buffers: stage from now on
        if pl.graph is None:
            if pl.calls < self.graph_after or not plans.quiet(cur):   # (capture at quiet points only)
                pl.calls += 1
                pl.last_img = ptr
                return self._run(img, pl)
            if not pl.staged and pl.last_img is not None and ptr != pl.last_img:
                pl.staged = True
            pl.static_img = pl.img if pl.staged else img
            pl.graph, pl.outs = self._capture(lambda: self._run(pl.static_img, pl))
        pl.last_img = ptr
        if pl.staged:
            pl.static_img.copy_(img)
        pl.graph.replay()
        return pl.outs

    @staticmethod
    def _capture(fn):
        from .head import CrossHead2
        box = {}
        g = CrossHead2._capture(lambda: box.update(out=fn()))
        return g, box["out"]

    def _run(self, img, pl):
        with hip.reserve_slots(self.grid_reserve):
            return self._run_layers(img, pl)

    def _run_layers(self, img, pl):
        B, _, H, W = img.shape
        w = self.w
        hip.stem7x7s2(img, w["stem.w"], w["stem.b"], pl.stem, B, H, W)
        hip.maxpool3x3s2(pl.stem, pl.pool, B, pl.hw_stem[0], pl.hw_stem[1], 64)
        x, cin = pl.pool, 64
        for i, (planes, blocks) in enumerate(self.stages):
            (hin, win), (h, wd) = pl.hw[i]
            for b in range(blocks):
                p = "layer%d.%d." % (i + 1, b)
                stride = 2 if (b == 0 and i > 0) else 1
                hi, wi = (hin, win) if b == 0 else (h, wd)
                t1 = pl.t1[i].view(-1)[:B * hi * wi * planes].view(B, hi, wi, planes)
                # conv1 1x1 (+BN+ReLU): a GEMM over the pixels
                hip.linear(x.view(-1, cin), w[p + "conv1.w"], w[p + "conv1.b"],
                           t1.view(-1, planes), relu=True, scratch=pl.scratch)
                # conv2 3x3, stride on this layer ("pytorch" style) (+BN+ReLU)
                wino = self.conv_algo in ("winograd", "winograd4") and p + "conv2.wino" in w \
                    and stride == 1 and planes >= self.wino_min_planes
                if wino and self.conv_algo == "winograd4" and pl.f43[i]:
                    # 4x fewer multiplications (fp32; ~1.6e-5 relative to the direct form)
                    hip.conv3x3_winograd43(t1, w[p + "conv2.wino4"], w[p + "conv2.b"], pl.t2[i],
                                           pl.wV, pl.wM, B, hi, wi, planes, planes, True)
                elif wino:
                    # 2.25x fewer multiplications (fp32; differs from the direct form by fp32
                    # re-association, ~2e-6 relative)
                    hip.conv3x3_winograd(t1, w[p + "conv2.wino"], w[p + "conv2.b"], pl.t2[i],
                                         pl.wV, pl.wM, B, hi, wi, planes, planes, True)
                else:
                    hip.conv2d_ex(t1, w[p + "conv2.w"], w[p + "conv2.b"], None, pl.t2[i], B, hi,
                                  wi, planes, planes, 3, 3, stride, 1, relu=True,
                                  scratch=pl.scratch)
                # shortcut: projection in the first block of a stage, identity after it
                if b == 0:
                    if stride == 1:
                        hip.linear(x.view(-1, cin), w[p + "downsample.0.w"],
                                   w[p + "downsample.0.b"], pl.idt[i].view(-1, planes * 4),
                                   scratch=pl.scratch)
                    else:
                        hip.conv2d_ex(x, w[p + "downsample.0.w"], w[p + "downsample.0.b"], None,
                                      pl.idt[i], B, hi, wi, cin, planes * 4, 1, 1, stride, 0,
                                      scratch=pl.scratch)
                    idt = pl.idt[i]
                else:
                    idt = x
                # conv3 1x1 (+BN) + shortcut, ReLU; the block input stays intact until here
                if b == blocks - 1:
                    dst = pl.out[i]
                else:
                    dst = pl.ping[i] if idt is not pl.ping[i] else pl.idt[i]
                if self.s3_conv3_min_planes and planes >= self.s3_conv3_min_planes:
                    t2 = pl.t2[i].view(-1, planes)
                    hip.s3_split(t2, pl.t2_s3)
                    hip.gemm_s3(pl.t2_s3, w[p + "conv3.w.s3"], t2.shape[0], planes * 4, planes,
                                bias=w[p + "conv3.b"], out=dst.view(-1, planes * 4),
                                res=idt.view(-1, planes * 4), relu_after=True)
                else:
                    hip.linear(pl.t2[i].view(-1, planes), w[p + "conv3.w"], w[p + "conv3.b"],
                               dst.view(-1, planes * 4), res=idt.view(-1, planes * 4),
                               relu_after=True, scratch=pl.scratch)
                x, cin = dst, planes * 4
        return tuple(o.permute(0, 3, 1, 2) for o in pl.out)

    __call__ = forward
