"""ChannelMapper neck on MI355X (configs/deformable_detr/cross_r101_vg.py:20-29).

mmdet's `ChannelMapper` with `norm_cfg=GN, act_cfg=None`: a 1x1 convolution + GroupNorm(32)
per backbone level and `num_outs - len(in_channels)` extra 3x3 stride-2 convolutions (+ GN)
on the last backbone level.  State-dict names are mmdet's (`convs.i.{conv,gn}`,
`extra_convs.j.{conv,gn}`; a ConvModule followed by a norm has no conv bias).

The levels are written as the TOKEN rows of one [B, sum(h*w), 256] buffer -- the layout the
Deformable-DETR encoder of `CrossHeadBBox` reads -- and returned as the reference's
(B, 256, h, w) tensors: channels_last-strided views of that buffer, which the head
recognises and uses in place.  The 1x1 convolutions are GEMMs straight off the backbone's
NCHW or channels_last outputs; the stride-2 3x3 is the implicit-GEMM convolution.
"""
from collections import OrderedDict

import math
import torch

from . import hip
from .plans import Arena, PlanCache, measure_bytes


class ChannelMapper:
    def __init__(self, in_channels, out_channels=256, kernel_size=1, num_outs=None,
                 norm_cfg=None, act_cfg=None, conv_cfg=None, init_cfg=None, **unused):
        if out_channels != 256 or kernel_size != 1 or act_cfg is not None or not norm_cfg \
                or norm_cfg.get("type") != "GN":
            raise NotImplementedError("ChannelMapper: 1x1 -> 256 channels + GN, no activation")
        self.in_channels = list(in_channels)
        self.groups = norm_cfg.get("num_groups", 32)
        self.num_outs = len(self.in_channels) if num_outs is None else num_outs
        if self.num_outs - len(self.in_channels) not in (0, 1):
            raise NotImplementedError("at most one extra level")
        self._params = OrderedDict((k, torch.zeros(s)) for k, s in self.param_shapes().items())
        self.device, self.w, self._plans = None, None, PlanCache()
        self.init_weights()

    def param_shapes(self):
        s = OrderedDict()
        for i, c in enumerate(self.in_channels):
            s["convs.%d.conv.weight" % i] = (256, c, 1, 1)
            s["convs.%d.gn.weight" % i] = (256,)
            s["convs.%d.gn.bias" % i] = (256,)
        for j in range(self.num_outs - len(self.in_channels)):
            s["extra_convs.%d.conv.weight" % j] = (256, self.in_channels[-1], 3, 3)
            s["extra_convs.%d.gn.weight" % j] = (256,)
            s["extra_convs.%d.gn.bias" % j] = (256,)
        return s

    def init_weights(self, seed=0):
        g = torch.Generator().manual_seed(seed)
        for k, p in self._params.items():
            if ".gn." in k:
                p.copy_(torch.ones(p.shape) if k.endswith("weight") else torch.zeros(p.shape))
            else:   # xavier_uniform (ChannelMapper's init_cfg)
                fan_in, fan_out = p[0].numel(), p.shape[0] * p.shape[2] * p.shape[3]
                a = math.sqrt(6.0 / (fan_in + fan_out))
                p.copy_((torch.rand(p.shape, generator=g) * 2 - 1) * a)
        self.w, self._plans = None, PlanCache()

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
                p.copy_(sd[k].detach().to(torch.float32).cpu().reshape(p.shape))
        self.w, self._plans = None, PlanCache()
        return missing, unexpected

    def eval(self):
        return self

    def to(self, device):
        self.device = torch.device(device)
        self.w, self._plans, self._arenas = None, PlanCache(), {}
        return self

    def cuda(self, index=0):
        return self.to("cuda:%d" % index)

    def _pack(self):
        if self.device is None or self.device.type != "cuda":
            raise RuntimeError("ChannelMapper runs on an MI355X only: call .to('cuda:0')")
        hip.lib()
        w = {k: v.to(self.device).contiguous() for k, v in self._params.items()}
        for i in range(len(self.in_channels)):
            w["convs.%d.conv.weight" % i] = w["convs.%d.conv.weight" % i].reshape(256, -1)
        for j in range(self.num_outs - len(self.in_channels)):
            k = "extra_convs.%d.conv.weight" % j     # [co][ci][ky][kx] -> [co][(ky*3+kx)*ci]
            w[k] = w[k].permute(0, 2, 3, 1).reshape(256, -1).contiguous()
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
        """One flat buffer per slot; a (batch, level shapes) plan is a set of views of it
        (plans.py).  The token rows a plan hands to the box trunk keep their address for a
        (shape, slot) as long as the arena does not grow."""
        arenas = self.__dict__.setdefault("_arenas", {})
        a = arenas.get(slot)
        if a is None:
            a = arenas[slot] = Arena(self.device, on_grow=lambda a, s=slot: self._plans.drop(
                lambda k: k[2] == s))
        return a

    def _layout_for(self, dims):
        B, shapes = dims[0], [(dims[1 + 2 * l], dims[2 + 2 * l]) for l in range((len(dims) - 1) // 2)]

        def layout(E):
            pl = ChannelMapper._Plan()
            self._layout(pl, E, B, shapes)
            return pl
        return layout

    def _measure(self, dims):
        return measure_bytes(self._layout_for(dims))

    def arena_bytes(self):
        return sum(a.capacity for a in self.__dict__.get("_arenas", {}).values())

    def _plan(self, B, shapes, slot):
        key = (B, tuple(shapes), slot)
        if key in self._plans:
            return self._plans[key]
        if self.w is None:
            self._pack()
        dims = (B,) + tuple(v for hw in shapes for v in hw)
        pl = self._arena(slot).carve(self._layout_for(dims), dims, self._measure)
        pl.streams = {}
        self._plans[key] = pl
        return pl

    def _layout(self, pl, E, B, shapes):
        pl.shapes = list(shapes)
        if self.num_outs > len(shapes):
            h, w = shapes[-1]
            pl.shapes.append(((h - 1) // 2 + 1, (w - 1) // 2 + 1))
        pl.N = [h * w for h, w in pl.shapes]
        pl.start = [sum(pl.N[:l]) for l in range(len(pl.N))]
        pl.SN = sum(pl.N)
        pl.tok = E(B, pl.SN, 256)
        pl.tmp = E(B, max(pl.N), 256)
        pl.splitk = E(B * 9 * 1024 * 1024)
        nblk = hip.groupnorm_nblk(max(pl.N))
        pl.gn_part = E.f64(B * nblk * self.groups * 2)
        # channel-last staging copy of the last input level for the extra 3x3 / 2 convolution
        # (contiguous NCHW inputs only; channels_last inputs are read in place)
        h, w = shapes[-1]
        pl.last_nhwc = E(B, h, w, self.in_channels[-1]) if self.num_outs > len(shapes) else None
        # the reference's (B, 256, h, w) tensors as views of the token rows
        pl.outs = tuple(pl.tok[:, s:s + n].view(B, h, w, 256).permute(0, 3, 1, 2)
                        for s, n, (h, w) in zip(pl.start, pl.N, pl.shapes))

    @torch.no_grad()
    @hip.on_device
    @hip.with_reserve
    def forward(self, inputs, slot=0):
        """inputs: the backbone levels named by `in_channels`, fp32 (B, C, h, w) device
        tensors, all contiguous or all channels_last."""
        assert len(inputs) == len(self.in_channels)
        if self.device is None:
            self.to(inputs[0].device)
        B = inputs[0].shape[0]
        nchw = all(f.is_contiguous() for f in inputs)
        nhwc = not nchw and all(f.is_contiguous(memory_format=torch.channels_last) for f in inputs)
        for f, c in zip(inputs, self.in_channels):
            if not f.is_cuda or f.dtype != torch.float32 or f.shape[1] != c or not (nchw or nhwc):
                raise RuntimeError("ChannelMapper inputs must be fp32 [B,C,H,W] device tensors "
                                   "with channels %s, all contiguous or all channels_last"
                                   % self.in_channels)
        pl = self._plan(B, [tuple(f.shape[-2:]) for f in inputs], slot)
        cur = torch.cuda.current_stream(self.device)
        pl.streams[cur.cuda_stream] = cur
        w = self.w
        for l, f in enumerate(inputs):
            cin, n = f.shape[1], pl.N[l]
            hip.gemm(f, w["convs.%d.conv.weight" % l], pl.tmp, M=n, N=256, K=cin,
                     lda=cin if nhwc else n, ldw=cin, ldc=256, batch=B, sA=cin * n, sC=n * 256,
                     colmajor=not nhwc, scratch=pl.splitk)
            hip.groupnorm_nhwc(pl.tmp, w["convs.%d.gn.weight" % l], w["convs.%d.gn.bias" % l],
                               pl.tok[:, pl.start[l]:], pl.gn_part, B, n, self.groups, False,
                               n * 256, pl.SN * 256)
        if self.num_outs > len(inputs):
            f = inputs[-1]
            h, wd = f.shape[-2:]
            l = len(inputs)
            x = f.permute(0, 2, 3, 1)
            if not nhwc:                        # the implicit-GEMM conv reads channel-last
                pl.last_nhwc.copy_(x)
                x = pl.last_nhwc
            hip.conv2d_ex(x, w["extra_convs.0.conv.weight"], None, None, pl.tmp, B, h, wd,
                          f.shape[1], 256, 3, 3, 2, 1, scratch=pl.splitk)
            hip.groupnorm_nhwc(pl.tmp, w["extra_convs.0.gn.weight"], w["extra_convs.0.gn.bias"],
                               pl.tok[:, pl.start[l]:], pl.gn_part, B, pl.N[l], self.groups,
                               False, pl.N[l] * 256, pl.SN * 256)
        return pl.outs

    __call__ = forward
