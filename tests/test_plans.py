"""Host logic of the per-slot buffer arenas and the plan cache (pair-net_amd/plans.py): the
part of the shape-polymorphic execution that needs no GPU.  (An `Arena` on the CPU device is
the same bump allocator over a host buffer; the head / backbone only ever make CUDA ones.)"""
import torch

from pairnet_amd.plans import ALIGN, Arena, Carver, PlanCache, measure_bytes


def _layout_for(dims):
    B, H, W = dims

    def layout(E):
        return dict(x=E(B, H * W, 8), idx=E.i64(B, 5), bits=E.i32((H * W + 31) // 32),
                    part=E.f64(B * 3), u8=E.u8(B, H, W), dims=dims)
    return layout


def _measure(dims):
    return measure_bytes(_layout_for(dims))


def test_carver_hands_out_aligned_typed_views_and_counts_bytes():
    buf = torch.zeros(1 << 16, dtype=torch.uint8)
    c = Carver(buf, buf.numel())
    a, b, d = c(3, 5), c.i64(7), c.u8(2, 3)
    assert a.dtype == torch.float32 and a.shape == (3, 5) and b.dtype == torch.int64
    assert a.data_ptr() == buf.data_ptr() and b.data_ptr() == buf.data_ptr() + ALIGN
    assert d.data_ptr() == buf.data_ptr() + 2 * ALIGN and c.used == 3 * ALIGN
    a.fill_(1.0)
    b.fill_(-1)                                           # (views of ONE buffer, no overlap)
    assert float(a.sum()) == 15.0 and int(buf[60:ALIGN].sum()) == 0
    m = Carver(None, 0)                                   # measuring mode: meta tensors
    t = m(4, 4)
    assert t.device.type == "meta" and m.used == ALIGN and measure_bytes(lambda E: E(100)) == 512


def test_arena_grows_with_the_envelope_only_and_plans_alias_each_other():
    grown = []
    ar = Arena("cpu", on_grow=lambda a: grown.append(a.generation))
    p1 = ar.carve(_layout_for((1, 10, 20)), (1, 10, 20), _measure)
    assert ar.grows == 1 and ar.envelope == (1, 10, 20) and ar.capacity == _measure((1, 10, 20))
    base = ar.buf.data_ptr()
    p2 = ar.carve(_layout_for((1, 8, 16)), (1, 8, 16), _measure)        # inside the envelope
    assert ar.grows == 1 and p2["x"].data_ptr() == p1["x"].data_ptr() == base
    p2["x"].fill_(7.0)
    assert float(p1["x"].view(-1)[0]) == 7.0                              # the same memory
    # a shape that exceeds ONE dimension: the envelope becomes the element-wise maximum, so
    # the first shape still fits afterwards without another growth
    p3 = ar.carve(_layout_for((1, 6, 40)), (1, 6, 40), _measure)
    assert ar.grows == 2 and ar.envelope == (1, 10, 40) and grown == [0, 1]
    assert ar.capacity >= max(_measure((1, 10, 40)), _measure((1, 6, 40)))
    assert p3["x"].data_ptr() == ar.buf.data_ptr() != base
    ar.carve(_layout_for((1, 10, 20)), (1, 10, 20), _measure)
    ar.carve(_layout_for((1, 10, 40)), (1, 10, 40), _measure)
    assert ar.grows == 2
    ar.reserve((2, 10, 40), _measure)                                     # explicit pre-sizing
    assert ar.grows == 3 and ar.envelope == (2, 10, 40)
    ar.reserve((1, 4, 4), _measure)
    assert ar.grows == 3


def test_plan_cache_parks_busy_values_until_their_events_fire():
    class Ev:
        def __init__(self):
            self.done = False

        def query(self):
            return self.done

    class Plan:
        def __init__(self):
            self.ev = Ev()

        def busy_events(self):
            return [self.ev]
    c = PlanCache(max_plans=2)
    plans = [Plan() for _ in range(4)]
    for i, p in enumerate(plans):
        c[("k", i)] = p
    assert list(c) == [("k", 2), ("k", 3)] and c.evictions == 2 and c.reap() == 2
    plans[0].ev.done = True
    assert c.reap() == 1                                  # released once its stream has passed
    c.drop(lambda k: k[1] == 3)                           # (an arena grew: its plans go, parked)
    assert list(c) == [("k", 2)] and c.evictions == 2 and c.reap() == 2
    for p in plans:
        p.ev.done = True
    assert c.reap() == 0
    assert c.get(("k", 9)) is None and c.get(("k", 2)) is plans[2]


def test_head_and_backbone_layouts_are_monotone_in_every_dimension():
    """What the envelope rule rests on: the bytes a layout carves never shrink when one
    size-driving dimension grows (checked on a grid around the production sizes, on meta
    tensors -- no GPU, no weights)."""
    from pairnet_amd import CrossHead2, ResNet50Hip, pairnet_head_cfg
    cfg = pairnet_head_cfg()
    cfg.pop("type")
    head = CrossHead2(**cfg)
    head.device = torch.device("cpu")
    head.w = {"query_feat.weight": torch.zeros(100, 256), "rel_query_feat.weight": torch.zeros(100, 256)}
    net = ResNet50Hip()

    def head_bytes(B, H, W):
        fs = net.feature_shapes(H, W)
        return head._measure(head._plan_dims(B, [fs[3], fs[2], fs[1]], fs[0]))
    for f in (head_bytes, lambda B, H, W: net._measure((B, H, W))):
        for B in (1, 2):
            for H in (749, 800, 801, 1067):
                for W in (1199, 1200, 1201, 1333):
                    here = f(B, H, W)
                    assert f(B + 1, H, W) >= here and f(B, H + 1, W) >= here and f(B, H, W + 1) >= here
    # odd and even sides of the quarter-resolution map share one Winograd scratch size rule
    assert head_bytes(1, 800, 1333) >= head_bytes(1, 799, 1332)
    # the Swin backbone, the neck and the box trunk follow the same rule
    from pairnet_amd import (ChannelMapper, CrossHeadBBox, SwinTransformerHip, bbox_head_cfg,
                             channel_mapper_cfg, swin_backbone_cfg)
    sc = swin_backbone_cfg("T")
    sc.pop("type")
    swin = SwinTransformerHip(**sc)
    swin.device, swin.w = torch.device("cpu"), {}
    nc = channel_mapper_cfg()
    nc.pop("type", None)
    neck = ChannelMapper(**nc)
    neck.device = torch.device("cpu")
    bc = bbox_head_cfg()
    bc.pop("type", None)
    box = CrossHeadBBox(**bc)
    box.device, box.w = torch.device("cpu"), {"rel_query_feat.weight": torch.zeros(100, 256)}

    def lvl(H, W):          # C3..C5 of a ResNet (the neck's inputs); + the neck's extra level
        fs = net.feature_shapes(H, W)[1:]
        return fs, fs + [((fs[-1][0] - 1) // 2 + 1, (fs[-1][1] - 1) // 2 + 1)]
    fns = (lambda B, H, W: swin._measure((B, H, W)),
           lambda B, H, W: neck._measure((B,) + tuple(v for hw in lvl(H, W)[0] for v in hw)),
           lambda B, H, W: box._measure((B,) + tuple(v for hw in lvl(H, W)[1] for v in hw)))
    for f in fns:
        for B in (1, 2):
            for H, W in ((800, 1333), (801, 1201), (1067, 800)):
                here = f(B, H, W)
                assert f(B + 1, H, W) >= here and f(B, H + 8, W) >= here and f(B, H, W + 8) >= here
