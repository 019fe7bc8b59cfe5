"""Import the reference's own CrossHead2 / ConvTiny from /root/reference.

TEST INFRASTRUCTURE, usable in the build container only (/root/reference does not
exist on the GPU box; nothing here is imported by `-m gpu` tests, smoke() or
bench.py).  Nothing is copied: the reference files are executed from where they
lie, with bytecode writing disabled so that no __pycache__ appears in the
read-only tree.

`mmcv` / `mmdet` are absent from the image, so name-only stub modules are put in
sys.modules.  Their builders return the restated layers of oracle/layers.py, so a
shimmed run = the reference's OWN orchestration and in-tree arithmetic
(pairnet_head.py:216-417, 760-924; cnn_factory.py:6-53) over restated
third-party layers.  This is what pins oracle/head.py and what
oracle/make_golden.py records into tests/golden/.
"""
import importlib.util
import os
import sys
import types

import torch.nn as nn

from . import layers as L

REF_ROOT = "/root/reference"


def available():
    return os.path.isfile(os.path.join(
        REF_ROOT, "pairnet/models/relation_heads/pairnet_head.py"))


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _load(name, rel):
    spec = importlib.util.spec_from_file_location(
        name, os.path.join(REF_ROOT, rel))
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


class _Registry:
    def register_module(self, *a, **kw):
        return lambda cls: cls


class _Loss:
    def __init__(self, cfg):
        self.use_sigmoid = bool(cfg.get("use_sigmoid", False))


def _passthrough_decorator(*a, **kw):
    return lambda fn: fn


def install():
    """Register the stubs and load the two reference modules. Idempotent."""
    if "pairnet.models.relation_heads.pairnet_head" in sys.modules:
        return
    if not available():
        raise RuntimeError("reference tree not present at " + REF_ROOT)
    sys.dont_write_bytecode = True
    _mod("mmcv")
    _mod("mmcv.cnn", Conv2d=nn.Conv2d, Linear=nn.Linear,
         build_plugin_layer=L.build_plugin_layer,
         caffe2_xavier_init=lambda *a, **k: None,
         bias_init_with_prob=None, constant_init=None)
    _mod("mmcv.cnn.bricks")
    _mod("mmcv.cnn.bricks.transformer",
         build_positional_encoding=L.build_positional_encoding,
         build_transformer_layer_sequence=L.build_transformer_layer_sequence)
    _mod("mmcv.ops", point_sample=None)
    _mod("mmcv.runner", ModuleList=nn.ModuleList,
         force_fp32=_passthrough_decorator)
    _mod("mmdet")
    from . import bbox_head as BH
    from . import deformable_detr as DD
    _mod("mmdet.core", build_assigner=None, build_sampler=None,
         multi_apply=None, reduce_mean=None,
         bbox_cxcywh_to_xyxy=BH.bbox_cxcywh_to_xyxy, bbox_xyxy_to_cxcywh=None)
    _mod("mmdet.models.utils", get_uncertain_point_coords_with_randomness=None,
         build_transformer=DD.build_transformer)
    _mod("mmdet.models.utils.transformer", inverse_sigmoid=DD.inverse_sigmoid)
    _mod("mmdet.datasets")
    _mod("mmdet.datasets.coco_panoptic", INSTANCE_OFFSET=1000)
    _mod("mmdet.models")
    _mod("mmdet.models.builder", HEADS=_Registry(),
         build_loss=lambda cfg: _Loss(cfg))

    # super(AnchorFreeHead, self).__init__(init_cfg) (pairnet_head.py:56)
    # resolves to the class after AnchorFreeHead in the MRO: give it one that
    # accepts init_cfg.
    class _Base(nn.Module):
        def __init__(self, init_cfg=None):
            super().__init__()

    class AnchorFreeHead(_Base):
        pass

    _mod("mmdet.models.dense_heads", AnchorFreeHead=AnchorFreeHead)
    for pkg in ("pairnet", "pairnet.models", "pairnet.models.frameworks",
                "pairnet.models.relation_heads"):
        _mod(pkg)
    _load("pairnet.models.frameworks.cnn_factory",
          "pairnet/models/frameworks/cnn_factory.py")
    _load("pairnet.models.relation_heads.pairnet_head",
          "pairnet/models/relation_heads/pairnet_head.py")
    _load("pairnet.models.relation_heads.baseline",
          "pairnet/models/relation_heads/baseline.py")
    _load("pairnet.models.relation_heads.psgtr_head2",
          "pairnet/models/relation_heads/psgtr_head2.py")
    _load("pairnet.models.relation_heads.pairnet_bbox_head",
          "pairnet/models/relation_heads/pairnet_bbox_head.py")


def reference_head_cfg():
    """exec the reference's own config file and return model.bbox_head."""
    path = os.path.join(REF_ROOT, "configs/mask2former/pairnet.py")
    scope = {}
    with open(path) as f:
        exec(compile(f.read(), path, "exec"), scope)
    return L.CfgDict(scope["model"]["bbox_head"])


def _losses(training):
    """Which `build_loss` the reference's CrossHead2 constructor sees: the name-only stub of
    the inference shims (no parameters, no buffers: the state dict is the inference one), or
    the restated mmdet losses after install_training() (SeesawLoss adds the buffer
    `rel_cls_loss.cum_samples`)."""
    mod = sys.modules["pairnet.models.relation_heads.pairnet_head"]
    if training:
        from . import mmdet_train as T
        mod.build_loss = T.build_loss
    else:
        mod.build_loss = lambda cfg: _Loss(cfg)


def build_reference_head(cfg=None):
    install()
    _losses(False)
    cls = sys.modules["pairnet.models.relation_heads.pairnet_head"].CrossHead2
    cfg = L.CfgDict(cfg if cfg is not None else reference_head_cfg())
    cfg.pop("type", None)
    return cls(**cfg, train_cfg=None).eval()


def reference_baseline_cfg():
    """model.bbox_head of the reference's configs/mask2former/baseline_r50_psg.py."""
    path = os.path.join(REF_ROOT, "configs/mask2former/baseline_r50_psg.py")
    scope = {}
    with open(path) as f:
        exec(compile(f.read(), path, "exec"), scope)
    return L.CfgDict(scope["model"]["bbox_head"])


def build_reference_baseline_head(cfg=None):
    install()
    cls = sys.modules["pairnet.models.relation_heads.baseline"].CrossHeadBaseline
    cfg = L.CfgDict(cfg if cfg is not None else reference_baseline_cfg())
    cfg.pop("type", None)
    return cls(**cfg, train_cfg=None).eval()


def reference_psgtr2_cfg():
    """model.bbox_head of the reference's configs/psgtr/psgtr_r50_psg_plus.py."""
    path = os.path.join(REF_ROOT, "configs/psgtr/psgtr_r50_psg_plus.py")
    scope = {}
    with open(path) as f:
        exec(compile(f.read(), path, "exec"), scope)
    return L.CfgDict(scope["model"]["bbox_head"])


def build_reference_psgtr2_head(cfg=None):
    install()
    cls = sys.modules["pairnet.models.relation_heads.psgtr_head2"].PSGTrHead2
    cfg = L.CfgDict(cfg if cfg is not None else reference_psgtr2_cfg())
    cfg.pop("type", None)
    return cls(**cfg, train_cfg=None).eval()


def reference_bbox_cfg(name="cross_r101_vg"):
    """model.{neck, bbox_head} of the reference's configs/deformable_detr/<name>.py."""
    path = os.path.join(REF_ROOT, "configs/deformable_detr/%s.py" % name)
    scope = {}
    with open(path) as f:
        exec(compile(f.read(), path, "exec"), scope)
    return L.CfgDict(scope["model"]["neck"]), L.CfgDict(scope["model"]["bbox_head"])


def build_reference_bbox_head(cfg=None):
    install()
    cls = sys.modules["pairnet.models.relation_heads.pairnet_bbox_head"].CrossHeadBBox
    cfg = L.CfgDict(cfg if cfg is not None else reference_bbox_cfg()[1])
    cfg.pop("type", None)
    cfg.pop("train_cfg", None)
    return cls(**cfg, train_cfg=None).eval()


def reference_conv_tiny():
    install()
    return sys.modules["pairnet.models.frameworks.cnn_factory"].creat_cnn(
        "conv_tiny").eval()


def install_training():
    """On top of install(): the names `CrossHead2.loss` needs (pairnet_head.py:419-718).  The
    [3P] ones (point_sample, MaskHungarianAssigner + its costs, MaskPseudoSampler, SeesawLoss,
    CrossEntropyLoss, multi_apply) come from oracle/mmdet_train.py (restated, unpinned); the
    reference's OWN `IdMatcher` (approaches/matcher.py:208-275) and `BCEWithLogitsLoss`
    (losses/seg_losses.py:153-166) are executed from /root/reference and register themselves
    in those builders.  Idempotent."""
    install()
    if "pairnet.models.relation_heads.approaches.matcher" in sys.modules:
        return
    from . import mmdet_train as T
    core = sys.modules["mmdet.core"]
    core.__dict__.update(build_assigner=T.build_assigner, build_sampler=T.build_sampler,
                         multi_apply=T.multi_apply, AssignResult=T.AssignResult,
                         BaseAssigner=T.BaseAssigner)
    _mod("mmdet.core.bbox")
    _mod("mmdet.core.bbox.builder", BBOX_ASSIGNERS=T.BBOX_ASSIGNERS)
    _mod("mmdet.core.bbox.match_costs", build_match_cost=T.build_match_cost)
    sys.modules["mmdet.models.builder"].__dict__.update(LOSSES=T.LOSSES, build_loss=T.build_loss)
    _mod("mmdet.models.losses")
    _mod("mmdet.models.losses.utils", weighted_loss=T.weighted_loss)
    sys.modules["mmcv.ops"].point_sample = T.point_sample
    for pkg in ("pairnet.models.relation_heads.approaches", "pairnet.models.losses"):
        _mod(pkg)
    _load("pairnet.models.relation_heads.approaches.matcher",
          "pairnet/models/relation_heads/approaches/matcher.py")
    _load("pairnet.models.losses.seg_losses", "pairnet/models/losses/seg_losses.py")
    # pairnet_head.py bound these names at import time (to the inference stubs)
    head_mod = sys.modules["pairnet.models.relation_heads.pairnet_head"]
    head_mod.__dict__.update(point_sample=T.point_sample, build_assigner=T.build_assigner,
                             build_sampler=T.build_sampler, multi_apply=T.multi_apply,
                             build_loss=T.build_loss)


def build_reference_training_head(cfg=None, train_cfg=None):
    """The reference CrossHead2 WITH its train_cfg (assigners, sampler, losses built)."""
    install_training()
    _losses(True)
    cls = sys.modules["pairnet.models.relation_heads.pairnet_head"].CrossHead2
    cfg = L.CfgDict(cfg if cfg is not None else reference_head_cfg())
    cfg.pop("type", None)
    if train_cfg is None:
        train_cfg = reference_model_cfg()["train_cfg"]
    return cls(**cfg, train_cfg=L.CfgDict(train_cfg)).eval()


def reference_model_cfg():
    """The whole `model` dict of configs/mask2former/pairnet.py."""
    path = os.path.join(REF_ROOT, "configs/mask2former/pairnet.py")
    scope = {}
    with open(path) as f:
        exec(compile(f.read(), path, "exec"), scope)
    return scope["model"]


# ---- dataset side: PanopticSceneGraphDataset / LoadPanopticSceneGraphAnnotations -------------
PAN_IMAGES = {}     # file name -> (H, W, 3) uint8 RGB array: what the stubbed image readers return
CAPTURED = {}       # last call of the stubbed sgg_evaluation (its keyword arguments)


def _ensure(name, **attrs):
    m = sys.modules.get(name) or _mod(name)
    m.__dict__.update(attrs)
    return m


def install_dataset():
    """Load the reference's pairnet/datasets/psg.py and pipelines/loading.py in place, under
    name-only stubs of what they import from mmcv / mmdet / detectron2 / panopticapi (all absent
    from the image): file and image readers that serve PAN_IMAGES, empty pipeline base classes,
    a COCOPanoptic that only keeps the dataset dict, and an `sgg_evaluation` that records what
    the dataset's `evaluate` hands it.  The in-tree code -- the constructor's filtering,
    `get_ann_info`, the ground-truth block of `evaluate`, `_load_masks_and_semantic_segs` --
    runs unmodified.  Returns (PanopticSceneGraphDataset, LoadPanopticSceneGraphAnnotations)."""
    if "pairnet.datasets.psg" in sys.modules:
        return (sys.modules["pairnet.datasets.psg"].PanopticSceneGraphDataset,
                sys.modules["pairnet.datasets.pipelines.loading"].LoadPanopticSceneGraphAnnotations)
    if not available():
        raise RuntimeError("reference tree not present at " + REF_ROOT)
    sys.dont_write_bytecode = True
    import json
    import numpy as np
    from . import dataset as D

    def _image(path, **kw):
        return PAN_IMAGES[os.path.basename(path)].copy()

    class FileClient:
        def __init__(self, **kw):
            pass

        def get(self, filename):
            return filename                 # (imfrombytes below resolves it)

    class ProgressBar:
        def __init__(self, n):
            pass

        def update(self):
            pass

    def load(path):
        with open(path) as f:
            return json.load(f)

    _ensure("mmcv", load=load, FileClient=FileClient, ProgressBar=ProgressBar,
            imfrombytes=lambda b, flag="color", channel_order="bgr": _image(b))
    _ensure("mmcv.parallel", DataContainer=object)
    _ensure("detectron2")
    _ensure("detectron2.data")
    _ensure("detectron2.data.detection_utils", read_image=_image)
    _ensure("panopticapi")
    _ensure("panopticapi.utils", rgb2id=D.rgb2id)      # [3P]: the published one-line definition

    class CocoPanopticDataset:
        def __len__(self):
            return len(self.data_infos)

        def _set_group_flag(self):
            pass

        def pre_pipeline(self, results):
            pass

    class COCOPanoptic:
        def createIndex(self):
            self.imgToAnns = {a["image_id"]: [a] for a in self.dataset["annotations"]}
            self.catToImgs = {}
            self.cats = {c["id"]: c for c in self.dataset["categories"]}

        def get_cat_ids(self):
            return sorted(self.cats)

    class _Pipe:
        def __init__(self, with_bbox=True, with_label=True, with_mask=True, with_seg=True,
                     file_client_args=None, **kw):
            self.with_bbox, self.with_label = with_bbox, with_label
            self.with_mask, self.with_seg = with_mask, with_seg
            self.file_client_args, self.file_client = dict(file_client_args or {}), None

    class BitmapMasks:
        def __init__(self, masks, height, width):
            self.masks = np.stack(masks, 0) if len(masks) else np.zeros((0, height, width), np.uint8)
            self.height, self.width = height, width

        def to_ndarray(self):
            return self.masks

    _ensure("mmdet")
    _ensure("mmdet.core", BitmapMasks=BitmapMasks)
    _ensure("mmdet.datasets", DATASETS=_Registry(), PIPELINES=_Registry(),
            CocoPanopticDataset=CocoPanopticDataset)
    _ensure("mmdet.datasets.coco_panoptic", COCOPanoptic=COCOPanoptic)
    _ensure("mmdet.datasets.pipelines", Compose=lambda p: p, DefaultFormatBundle=object,
            LoadAnnotations=_Pipe, to_tensor=None)
    _ensure("mmdet.datasets.pipelines.loading", LoadPanopticAnnotations=_Pipe)

    def sgg_evaluation(*a, **kw):
        CAPTURED.clear()
        CAPTURED.update(kw, args=a)
        return "captured"

    class Result:                           # (relation_util.Result is a field holder: keep the
        def __init__(self, **kw):           #  keyword arguments the dataset passes)
            self.__dict__.update(kw)

    for pkg in ("pairnet", "pairnet.models", "pairnet.models.relation_heads", "pairnet.datasets",
                "pairnet.datasets.pipelines"):
        _ensure(pkg)
    _ensure("pairnet.evaluation", sgg_evaluation=sgg_evaluation)
    _ensure("pairnet.models.relation_heads.approaches", Result=Result)
    loading = _load("pairnet.datasets.pipelines.loading", "pairnet/datasets/pipelines/loading.py")
    psg = _load("pairnet.datasets.psg", "pairnet/datasets/psg.py")
    return psg.PanopticSceneGraphDataset, loading.LoadPanopticSceneGraphAnnotations
