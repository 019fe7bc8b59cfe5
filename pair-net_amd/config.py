"""Config boundary: the schema of the reference's configs/mask2former/pairnet.py.

The reference builds its head from nested python dicts (configs/mask2former/
pairnet.py:7-211) and reads some of them by attribute
(pairnet_head.py:83-87: `transformer_decoder.transformerlayers.attn_cfgs.num_heads`).
`ConfigDict` gives the same access; `pairnet_r50()` is this repo's own statement of
that schema (same keys and values, assembled from small helpers rather than copied),
and `load_config()` also accepts a python config file in the reference's format,
e.g. the reference's own file when it is present.
"""
import os


class ConfigDict(dict):
    """Nested dict with attribute access (the subset of mmcv.ConfigDict used)."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        for k, v in dict(*args, **kwargs).items():
            self[k] = v

    def __setitem__(self, key, value):
        super().__setitem__(key, _to_cfg(value))

    def __getattr__(self, key):
        try:
            return self[key]
        except KeyError:
            raise AttributeError(key) from None

    __setattr__ = __setitem__

    def update(self, *args, **kwargs):
        for k, v in dict(*args, **kwargs).items():
            self[k] = v

    def copy(self):
        return ConfigDict(self)


def _to_cfg(v):
    if isinstance(v, ConfigDict):
        return v
    if isinstance(v, dict):
        return ConfigDict(v)
    if isinstance(v, (list, tuple)):
        return type(v)(_to_cfg(x) for x in v)
    return v


def _mha(embed=256, heads=8):
    return dict(type="MultiheadAttention", embed_dims=embed, num_heads=heads,
                attn_drop=0.0, proj_drop=0.0, dropout_layer=None, batch_first=False)


CROSS_FIRST = ("cross_attn", "norm", "self_attn", "norm", "ffn", "norm")
SELF_FIRST = ("self_attn", "norm", "cross_attn", "norm", "ffn", "norm")


def _decoder(num_layers, return_intermediate, ffn_drop, order=CROSS_FIRST, attn=None):
    ffn = dict(embed_dims=256, feedforward_channels=2048, num_fcs=2,
               act_cfg=dict(type="ReLU", inplace=True), ffn_drop=ffn_drop,
               dropout_layer=None, add_identity=True)
    return dict(type="DetrTransformerDecoder", return_intermediate=return_intermediate,
                num_layers=num_layers,
                transformerlayers=dict(type="BaseTransformerLayer", attn_cfgs=attn or _mha(),
                                       ffn_cfgs=ffn, operation_order=order))


def _pixel_decoder(num_levels=3, num_points=4, enc_layers=6):
    attn = dict(type="MultiScaleDeformableAttention", embed_dims=256, num_heads=8,
                num_levels=num_levels, num_points=num_points, im2col_step=64,
                dropout=0.0, batch_first=False, norm_cfg=None, init_cfg=None)
    ffn = dict(type="FFN", embed_dims=256, feedforward_channels=1024, num_fcs=2,
               ffn_drop=0.0, act_cfg=dict(type="ReLU", inplace=True))
    enc = dict(type="DetrTransformerEncoder", num_layers=enc_layers, init_cfg=None,
               transformerlayers=dict(type="BaseTransformerLayer", attn_cfgs=attn,
                                      ffn_cfgs=ffn,
                                      operation_order=("self_attn", "norm", "ffn", "norm")))
    return dict(type="MSDeformAttnPixelDecoder", num_outs=3,
                norm_cfg=dict(type="GN", num_groups=32), act_cfg=dict(type="ReLU"),
                encoder=enc, init_cfg=None,
                positional_encoding=dict(type="SinePositionalEncoding", num_feats=128,
                                         normalize=True))


def pairnet_head_cfg(in_channels=(256, 512, 1024, 2048), num_obj_query=100,
                     num_rel_query=100, num_classes=133, num_relations=56):
    """bbox_head section (type CrossHead2) of the Pair-Net R50 config."""
    nc = num_classes
    return ConfigDict(
        type="CrossHead2", num_classes=nc, num_relations=num_relations,
        num_obj_query=num_obj_query, num_rel_query=num_rel_query, mapper="conv_tiny",
        in_channels=list(in_channels), feat_channels=256, out_channels=256,
        num_transformer_feat_level=3, embed_dims=256,
        enforce_decoder_input_project=False,
        pixel_decoder=_pixel_decoder(),
        transformer_decoder=_decoder(9, False, 0.0),
        relation_decoder=_decoder(6, True, 0.1),
        positional_encoding=dict(type="SinePositionalEncoding", num_feats=128,
                                 normalize=True),
        rel_cls_loss=dict(type="SeesawLoss", num_classes=num_relations,
                          return_dict=True, loss_weight=2.0),
        subobj_cls_loss=dict(type="CrossEntropyLoss", use_sigmoid=False, loss_weight=4.0,
                             reduction="mean", class_weight=[1.0] * (nc + 1)),
        importance_match_loss=dict(type="BCEWithLogitsLoss", reduction="mean",
                                   loss_weight=5.0),
        loss_cls=dict(type="CrossEntropyLoss", use_sigmoid=False, loss_weight=2.0,
                      reduction="mean", class_weight=[1.0] * nc + [0.1]),
        loss_mask=dict(type="CrossEntropyLoss", use_sigmoid=True, reduction="mean",
                       loss_weight=5.0),
        loss_dice=dict(type="DiceLoss", use_sigmoid=True, activate=True, reduction="mean",
                       naive_dice=True, eps=1.0, loss_weight=5.0))


def baseline_head_cfg(in_channels=(256, 512, 1024, 2048), num_obj_query=100,
                      num_rel_query=100, num_classes=133, num_relations=56):
    """bbox_head section (type CrossHeadBaseline) of configs/mask2former/baseline_r50_psg.py:
    the same trunk; the relation decoder runs self-attention first and cross-attends the
    pixel memories."""
    nc = num_classes
    return ConfigDict(
        type="CrossHeadBaseline", num_classes=nc, num_relations=num_relations,
        num_obj_query=num_obj_query, num_rel_query=num_rel_query,
        in_channels=list(in_channels), strides=[4, 8, 16, 32], feat_channels=256,
        out_channels=256, num_transformer_feat_level=3, embed_dims=256,
        enforce_decoder_input_project=False,
        pixel_decoder=_pixel_decoder(),
        transformer_decoder=_decoder(9, False, 0.0),
        relation_decoder=_decoder(6, True, 0.1, SELF_FIRST,
                                  dict(type="MultiheadAttention", embed_dims=256, num_heads=8)),
        positional_encoding=dict(type="SinePositionalEncoding", num_feats=128,
                                 normalize=True),
        rel_loss_cls=dict(type="CrossEntropyLoss", use_sigmoid=False, loss_weight=2.0,
                          reduction="mean", class_weight=[0.02] + [1.0] * num_relations),
        sub_id_loss=dict(type="MultilabelCrossEntropy", loss_weight=2.0),
        obj_id_loss=dict(type="MultilabelCrossEntropy", loss_weight=2.0),
        loss_cls=dict(type="CrossEntropyLoss", use_sigmoid=False, loss_weight=2.0,
                      reduction="mean", class_weight=[1.0] * nc + [0.1]),
        loss_mask=dict(type="CrossEntropyLoss", use_sigmoid=True, reduction="mean",
                       loss_weight=5.0),
        loss_dice=dict(type="DiceLoss", use_sigmoid=True, activate=True, reduction="mean",
                       naive_dice=True, eps=1.0, loss_weight=5.0))


def psgtr2_head_cfg(in_channels=(256, 512, 1024, 2048), num_obj_query=100, num_classes=133,
                    num_relations=56):
    """bbox_head section (type PSGTrHead2) of configs/psgtr/psgtr_r50_psg_plus.py."""
    pd = _pixel_decoder()
    pd.pop("positional_encoding")       # that config relies on the decoder's defaults
    pd.pop("init_cfg")
    ce = lambda w: dict(type="CrossEntropyLoss", use_sigmoid=False, loss_weight=w,
                        class_weight=1.0)
    mask = dict(type="CrossEntropyLoss", use_sigmoid=True, reduction="mean", loss_weight=5.0)
    dice = dict(type="DiceLoss", use_sigmoid=True, activate=True, reduction="mean",
                naive_dice=True, eps=1.0, loss_weight=5.0)
    cfg = ConfigDict(
        type="PSGTrHead2", num_classes=num_classes, num_relations=num_relations, use_mask=True,
        num_obj_query=num_obj_query, pixel_decoder=pd,
        transformer_decoder=_decoder(9, False, 0.0),
        positional_encoding=dict(type="SinePositionalEncoding", num_feats=128,
                                 normalize=True),
        sub_loss_cls=ce(1.0), sub_loss_mask=dict(mask), sub_loss_dice=dict(dice),
        obj_loss_cls=ce(1.0), obj_loss_mask=dict(mask), obj_loss_dice=dict(dice),
        rel_loss_cls=ce(2.0))
    if tuple(in_channels) != (256, 512, 1024, 2048):   # the reference file relies on the default
        cfg["in_channels"] = list(in_channels)
    return cfg


def pairnet_r50():
    """`model` section: PSGTr(ResNet-50, CrossHead2) as the reference configures it."""
    return ConfigDict(
        type="PSGTr",
        backbone=dict(type="ResNet", depth=50, num_stages=4, out_indices=(0, 1, 2, 3),
                      frozen_stages=1, norm_cfg=dict(type="BN", requires_grad=False),
                      norm_eval=True, style="pytorch"),
        bbox_head=pairnet_head_cfg(),
        test_cfg=dict(max_per_img=100))


def swin_backbone_cfg(variant="B"):
    """`backbone` section of configs/mask2former/pairnet_swinb.py:203-226 (Swin-B, window 12,
    pretrain 384); "L" = the Swin-L of BASELINE.json configs[3] (same file, embed 192 and
    heads 6/12/24/48), "T" = Swin-T (window 7)."""
    embed, depths, heads, ws = {"T": (96, (2, 2, 6, 2), (3, 6, 12, 24), 7),
                                "B": (128, (2, 2, 18, 2), (4, 8, 16, 32), 12),
                                "L": (192, (2, 2, 18, 2), (6, 12, 24, 48), 12)}[variant]
    return dict(type="SwinTransformer", embed_dims=embed, depths=list(depths),
                num_heads=list(heads), window_size=ws, mlp_ratio=4, qkv_bias=True, qk_scale=None,
                drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.3, patch_norm=True,
                out_indices=(0, 1, 2, 3), convert_weights=True, frozen_stages=3,
                pretrain_img_size=384)


def pairnet_swin(variant="B", num_obj_query=100):
    """`model` section: PSGTr(Swin, CrossHead2) (pairnet_swinb.py:201-240)."""
    bb = swin_backbone_cfg(variant)
    chans = [bb["embed_dims"] * 2 ** i for i in range(4)]
    head = pairnet_head_cfg(in_channels=chans, num_obj_query=num_obj_query)
    # the Swin config differs from pairnet.py in two head keys: it passes `strides`
    # (pairnet_swinb.py:237, swallowed by **kwargs) and leaves `mapper` at its default
    head.pop("mapper")
    head["strides"] = [4, 8, 16, 32]
    return ConfigDict(type="PSGTr", backbone=bb, bbox_head=head,
                      test_cfg=dict(max_per_img=100))


def baseline_r50():
    """`model` section: PSGTr(ResNet-50, CrossHeadBaseline)."""
    cfg = pairnet_r50()
    cfg["bbox_head"] = baseline_head_cfg()
    return cfg


def psgtr2_r50():
    """`model` section: PSGTr(ResNet-50, PSGTrHead2)."""
    cfg = pairnet_r50()
    cfg["bbox_head"] = psgtr2_head_cfg()
    return cfg


IMG_NORM_CFG = dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True)


def channel_mapper_cfg(in_channels=(512, 1024, 2048)):
    """configs/deformable_detr/cross_r101_vg.py:20-29."""
    return dict(type="ChannelMapper", in_channels=list(in_channels), kernel_size=1,
                out_channels=256, act_cfg=None, norm_cfg=dict(type="GN", num_groups=32),
                num_outs=4)


def bbox_head_cfg(num_obj_query=300, num_rel_query=100, num_classes=150, num_relations=50):
    """The `bbox_head` of configs/deformable_detr/cross_r101_vg.py:30-117 (inference keys):
    CrossHeadBBox on a two-stage, box-refining Deformable-DETR trunk."""
    msda = dict(type="MultiScaleDeformableAttention", embed_dims=256)
    trunk = dict(
        type="DeformableDetrTransformer", as_two_stage=True,
        encoder=dict(type="DetrTransformerEncoder", num_layers=6, transformerlayers=dict(
            type="BaseTransformerLayer", attn_cfgs=msda, feedforward_channels=1024,
            ffn_dropout=0.1, operation_order=("self_attn", "norm", "ffn", "norm"))),
        decoder=dict(type="DeformableDetrTransformerDecoder", num_layers=6,
                     return_intermediate=True, transformerlayers=dict(
                         type="DetrTransformerDecoderLayer",
                         attn_cfgs=[dict(type="MultiheadAttention", embed_dims=256, num_heads=8,
                                         dropout=0.1), msda],
                         feedforward_channels=1024, ffn_dropout=0.1, operation_order=SELF_FIRST)))
    return ConfigDict(
        type="CrossHeadBBox", num_classes=num_classes, num_relations=num_relations,
        num_obj_query=num_obj_query, num_rel_query=num_rel_query, in_channels=2048,
        sync_cls_avg_factor=True, embed_dims=256, as_two_stage=True, with_box_refine=True,
        transformer=trunk, relation_decoder=_decoder(6, True, 0.0),
        positional_encoding=dict(type="SinePositionalEncoding", num_feats=128, normalize=True,
                                 offset=-0.5),
        loss_cls=dict(type="FocalLoss", use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=2.0))


def cross_r101_vg():
    """configs/deformable_detr/cross_r101_vg.py:8-117: ResNet-101 (C3-C5) -> ChannelMapper ->
    CrossHeadBBox, the box-trunk sibling of Pair-Net on Visual Genome (150 / 50 classes)."""
    return ConfigDict(model=dict(
        type="PSGTr",
        backbone=dict(type="ResNet", depth=101, num_stages=4, out_indices=(1, 2, 3),
                      frozen_stages=1, norm_cfg=dict(type="BN", requires_grad=False),
                      norm_eval=True, style="pytorch"),
        neck=channel_mapper_cfg(), bbox_head=bbox_head_cfg(), test_cfg=dict(max_per_img=100)))


def test_pipeline_cfg():
    """`test_pipeline` of configs/mask2former/pairnet.py:310-331 (what preprocess.TestPipeline
    .from_config consumes)."""
    return [
        dict(type="LoadImageFromFile"),
        dict(type="LoadSceneGraphAnnotations", with_bbox=True, with_rel=True),
        dict(type="MultiScaleFlipAug", img_scale=(1333, 800), flip=False, transforms=[
            dict(type="Resize", keep_ratio=True),
            dict(type="RandomFlip"),
            dict(type="Normalize", **IMG_NORM_CFG),
            dict(type="Pad", size_divisor=1),
            dict(type="ImageToTensor", keys=["img"]),
            dict(type="ToTensor", keys=["gt_bboxes", "gt_labels"]),
            dict(type="ToDataContainer", fields=(dict(key="gt_bboxes"), dict(key="gt_labels"))),
            dict(type="Collect", keys=["img"]),
        ]),
    ]


test_pipeline_cfg.__test__ = False


def load_config(path):
    """exec a reference-format python config; returns ConfigDict of its globals
    (without `_base_` inheritance: the model section of pairnet.py has none)."""
    scope = {}
    with open(path) as f:
        exec(compile(f.read(), os.path.abspath(path), "exec"), scope)
    return ConfigDict({k: v for k, v in scope.items() if not k.startswith("__")})
