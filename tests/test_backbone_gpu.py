"""GPU: the native ResNet-50 backbone against oracle/backbone.py
(torch CPU fp32 restatement of mmdet's ResNet-50), and the kernels only it uses (strided
implicit-GEMM convolution with residual epilogue, 7x7/2 stem on the NCHW image, max pool).

Tolerance: 1e-4 of the feature scale (fp32; BatchNorm folding and a different summation
order are the only differences)."""
import math

import pytest
import torch
import torch.nn.functional as F

from oracle.backbone import OracleResNet50, seeded_backbone_state

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def R(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def rel(a, b):
    b = b.double()
    return float((a.detach().cpu().double() - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("B,H,W,Cin,Cout,k,stride,pad,res", [
    (2, 13, 17, 64, 64, 3, 1, 1, False), (1, 30, 41, 128, 128, 3, 2, 1, False),
    (2, 25, 42, 256, 512, 1, 2, 0, False), (1, 16, 16, 64, 256, 1, 1, 0, True),
    (1, 9, 11, 32, 96, 3, 2, 1, True)])
def test_conv2d_ex_matches_torch(B, H, W, Cin, Cout, k, stride, pad, res):
    from pairnet_amd import hip
    x, w, b = R(B, Cin, H, W, seed=1), R(Cout, Cin, k, k, seed=2, scale=0.05), R(Cout, seed=3)
    ref = F.conv2d(x, w, b, stride=stride, padding=pad)
    Ho, Wo = ref.shape[-2:]
    r = R(B, Cout, Ho, Wo, seed=4) if res else None
    want = F.relu(F.relu(ref) + r) if res else F.relu(ref)
    out = torch.empty(B, Ho, Wo, Cout, device=DEV)
    hip.conv2d_ex(x.permute(0, 2, 3, 1).contiguous().to(DEV),
                  w.permute(0, 2, 3, 1).reshape(Cout, -1).contiguous().to(DEV), b.to(DEV),
                  r.permute(0, 2, 3, 1).contiguous().to(DEV) if res else None, out, B, H, W, Cin,
                  Cout, k, k, stride, pad, relu=True, relu_after=res)
    assert rel(out.permute(0, 3, 1, 2), want) < 2e-5 * math.sqrt(Cin * k * k / 256)


@pytest.mark.parametrize("B,H,W", [(1, 64, 96), (2, 37, 53), (1, 7, 9)])
def test_stem_and_maxpool_match_torch(B, H, W):
    from pairnet_amd import hip
    img, w, b = R(B, 3, H, W, seed=5), R(64, 3, 7, 7, seed=6, scale=0.1), R(64, seed=7)
    ref = F.relu(F.conv2d(img, w, b, stride=2, padding=3))
    h1, w1 = ref.shape[-2:]
    wp = torch.zeros(64, 160)
    wp[:, :147] = w.reshape(64, 147)
    out = torch.empty(B, h1, w1, 64, device=DEV)
    hip.stem7x7s2(img.to(DEV), wp.to(DEV), b.to(DEV), out, B, H, W)
    assert rel(out.permute(0, 3, 1, 2), ref) < 2e-5
    pooled = F.max_pool2d(ref, 3, stride=2, padding=1)
    h2, w2 = pooled.shape[-2:]
    pout = torch.empty(B, h2, w2, 64, device=DEV)
    hip.maxpool3x3s2(ref.permute(0, 2, 3, 1).contiguous().to(DEV), pout, B, h1, w1, 64)
    assert torch.equal(pout.permute(0, 3, 1, 2).cpu(), pooled)


@pytest.mark.parametrize("B,H,W,algo", [(1, 96, 128, "winograd"), (2, 75, 101, "winograd"),
                                        (1, 128, 192, "winograd"), (1, 128, 192, "direct"),
                                        (1, 128, 192, "winograd4"), (2, 75, 101, "winograd4"),
                                        (1, 416, 544, "winograd4")])
def test_backbone_matches_oracle(B, H, W, algo):
    """(128 x 192: the stage-3 map is 8 x 12, even sides, so its stride-1 3x3 layers take the
    Winograd form when `conv_algo` says so; "winograd4", the default, takes F(4x4,3x3) only on
    maps where that multiplies fewer padded GEMM rows -- stages 2 and 3 of the 416 x 544 case.)"""
    from pairnet_amd import ResNet50Hip
    sd = seeded_backbone_state(31)
    oracle = OracleResNet50()
    oracle.load_state_dict(sd)
    net = ResNet50Hip()
    net.conv_algo = algo
    net.load_state_dict(sd)
    net.to(DEV)
    img = R(B, 3, H, W, seed=8)
    want = oracle(img)
    got = net(img.to(DEV))
    torch.cuda.synchronize()
    for i, (g, o) in enumerate(zip(got, want)):
        assert tuple(g.shape) == tuple(o.shape)
        assert g.is_contiguous(memory_format=torch.channels_last)
        e = rel(g, o)
        print("C%d %s rel err %.2e" % (i + 2, tuple(o.shape), e))
        assert e < 1e-4


@pytest.mark.parametrize("H,W", [(75, 101), (800, 1333)])
def test_resnet101_matches_oracle(H, W):
    """depth 101 (the R101 configs of the reference, e.g. configs/psgformer/psgformer_r101_psg.py,
    configs/deformable_detr/cross_r101_vg.py): same kernels, 23 blocks in stage 3 -- also at the
    production size, where 22 of them take the Winograd F(4x4,3x3) form on the 50 x 84 map."""
    from pairnet_amd import ResNet50Hip
    sd = seeded_backbone_state(37, 101)
    oracle = OracleResNet50(101)
    oracle.load_state_dict(sd)
    net = ResNet50Hip(depth=101)
    assert set(net.state_dict()) == set(sd)
    net.load_state_dict(sd)
    net.to(DEV)
    img = R(1, 3, H, W, seed=8)
    with torch.no_grad():
        want = oracle(img)
    got = net(img.to(DEV))
    torch.cuda.synchronize()
    for g, o in zip(got, want):
        print(tuple(o.shape), "rel err %.2e" % rel(g, o))
        assert tuple(g.shape) == tuple(o.shape) and rel(g, o) < 2e-4
    with pytest.raises(NotImplementedError):
        ResNet50Hip(depth=34)


def test_head_reads_channels_last_features():
    """The head's outputs do not depend on the memory format of `feats`."""
    from helpers import head_cfg, oracle_head
    from oracle import seeded
    from pairnet_amd import CrossHead2
    _, sd, _ = oracle_head(7)
    head = CrossHead2(**head_cfg())
    head.load_state_dict(sd)
    head.to(DEV)
    H, W = 64, 96
    feats = [f.to(DEV) for f in seeded.seeded_feats(8, 2, H, W)]
    metas = [dict(img_shape=(H, W, 3), scale_factor=[1.0] * 4)] * 2
    a_cls, a_mask = head.forward(feats, metas)
    a = {k: v.clone() for k, v in {**a_cls, **a_mask}.items()}
    cl = [f.contiguous(memory_format=torch.channels_last) for f in feats]
    assert not cl[0].is_contiguous()
    b_cls, b_mask = head.forward(cl, metas)
    torch.cuda.synchronize()
    for k, v in {**b_cls, **b_mask}.items():
        scale = max(1.0, float(a[k].abs().max()))
        assert float((a[k] - v).abs().max()) < 2e-4 * scale, k
    with pytest.raises(RuntimeError):
        head.forward([cl[0]] + feats[1:], metas)        # mixed layouts are refused


def test_detector_end_to_end_native_backbone():
    """The native backbone inside the detector against the PyTorch-ROCm / MIOpen ResNet-50
    (tools/torch_resnet50.py, the bench's comparison leg) with the same state dict."""
    from pairnet_amd import build_detector, pairnet_r50
    from tools.torch_resnet50 import ResNet50
    det = build_detector(pairnet_r50()).to(DEV)
    ref = ResNet50().to(DEV)
    ref.load_state_dict(det.backbone.state_dict())
    img = R(1, 3, 128, 160, seed=9).to(DEV)
    metas = [dict(img_shape=(128, 160, 3), scale_factor=[1.0] * 4)]
    fa, fb = det.extract_feat(img), ref(img)
    for x, y in zip(fa, fb):
        assert rel(x, y.cpu()) < 1e-4
    res = det.simple_test(img, metas)
    assert len(res) == 1 and res[0].rel_dists.shape == (100, 57)
    assert res[0].masks.shape == (200, 128, 160)


def test_pipelined_detector_with_reused_backbone_buffers():
    """Native backbone (whose outputs are views of reused buffers) feeding the 3-deep
    pipelined head with a DIFFERENT image every step == the one-image-at-a-time results,
    bitwise: stage A must have consumed the features before the backbone overwrites them."""
    from helpers import head_cfg, oracle_head
    from pairnet_amd import CrossHead2, PipelinedHead, ResNet50Hip
    _, sd, _ = oracle_head(7)
    head = CrossHead2(**head_cfg())
    head.load_state_dict(sd)
    head.to(DEV)
    net = ResNet50Hip()
    net.load_state_dict(seeded_backbone_state(41))
    net.to(DEV)
    H, W = 160, 224
    metas = [dict(img_shape=(H, W, 3), scale_factor=[1.0] * 4)]
    imgs = [R(1, 3, H, W, seed=50 + i).to(DEV) for i in range(6)]
    want = []
    for im in imgs:
        r = head.simple_test_bboxes(net(im), metas)[0]
        want.append([t.clone() if t.is_cuda else t for t in r])
    torch.cuda.synchronize()
    for graphs in (False, True):
        head.use_graphs = graphs
        pipe = PipelinedHead(head, depth=3)
        got = []
        for rep in range(2):                 # second pass runs on captured graphs
            got = []
            for im in imgs:
                o = pipe.submit(net(im), metas)
                if o is not None:
                    got.append([t.clone() if t.is_cuda else t for t in o[0]])
            got += [o[0] for o in pipe.flush()]
        torch.cuda.synchronize()
        assert len(got) == len(want)
        for a, b in zip(want, got):
            for x, y in zip(a, b):
                assert torch.equal(x.cpu(), y.cpu())
    # the bench's schedule: two stage-A streams, the backbone issued ON them (hipGraph replay,
    # one backbone buffer slot per stream), two chain streams
    head.use_graphs = net.use_graphs = True
    pipe = PipelinedHead(head, depth=4, a_streams=2)
    for rep in range(3):
        got = []
        for im in imgs:
            sl = pipe.count % len(pipe.streams_a)
            pipe.streams_a[sl].wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(pipe.streams_a[sl]):
                o = pipe.submit(net(im, slot=sl), metas)
                if o is not None:    # (the results are ordered behind THIS stream, the
                    # one submit() was called on: consume them on it)
                    got.append([t.clone() if t.is_cuda else t for t in o[0]])
        got += [o[0] for o in pipe.flush()]
        torch.cuda.synchronize()
        assert len(got) == len(want)
        for a, b in zip(want, got):
            for x, y in zip(a, b):
                assert torch.equal(x.cpu(), y.cpu()), rep
