"""Public names of the package (what a user of the reference looks for)."""
from .config import (ConfigDict, baseline_head_cfg, baseline_r50, bbox_head_cfg,  # noqa: F401
                     channel_mapper_cfg, cross_r101_vg, load_config,
                     pairnet_head_cfg, pairnet_r50, pairnet_swin, psgtr2_head_cfg, psgtr2_r50,
                     swin_backbone_cfg, test_pipeline_cfg)
from .psgtr_head2 import PSGTrHead2  # noqa: F401
from .backbone import ResNet50Hip  # noqa: F401
from .swin import SwinTransformerHip  # noqa: F401
from .baseline_head import CrossHeadBaseline  # noqa: F401
from .bbox_head import CrossHeadBBox  # noqa: F401
from .neck import ChannelMapper  # noqa: F401
from .head import CrossHead2  # noqa: F401
from .pipeline import PipelinedHead  # noqa: F401
from .grad import BackboneGrad, HeadGrad, PixelDecoderGrad, RelationTailGrad  # noqa: F401
from .train import TailTrainer  # noqa: F401
from .preprocess import TestPipeline  # noqa: F401
from .detector import (PSGTr, Result, ResultStreamer, build_detector, load_checkpoint,  # noqa: F401
                       triplet2Result)
from .dist import all_gather_triplets, shard_indices  # noqa: F401
from .evaluation import SceneGraphMetrics, TripletEvaluator  # noqa: F401
from . import dataset  # noqa: F401  (PSG ground truth: load_psg, ann_info, eval_ground_truth, ...)

__all__ = ["ConfigDict", "load_config", "pairnet_head_cfg", "pairnet_r50", "CrossHead2",
           "PSGTr", "Result", "ResultStreamer", "build_detector", "load_checkpoint", "triplet2Result", "all_gather_triplets",
           "shard_indices", "PipelinedHead", "CrossHeadBaseline", "baseline_head_cfg",
           "baseline_r50", "PSGTrHead2", "psgtr2_head_cfg", "psgtr2_r50", "ResNet50Hip",
           "SwinTransformerHip", "pairnet_swin", "swin_backbone_cfg", "TestPipeline", "test_pipeline_cfg",
           "CrossHeadBBox", "ChannelMapper", "bbox_head_cfg", "channel_mapper_cfg", "cross_r101_vg",
           "TripletEvaluator", "SceneGraphMetrics", "dataset", "RelationTailGrad", "HeadGrad", "PixelDecoderGrad", "BackboneGrad", "TailTrainer"]
