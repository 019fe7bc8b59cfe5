"""Public names of the package (what a user of the reference looks for)."""
from .config import ConfigDict, load_config, pairnet_head_cfg, pairnet_r50  # noqa: F401
from .head import CrossHead2  # noqa: F401
from .pipeline import PipelinedHead  # noqa: F401
from .detector import PSGTr, Result, build_detector, triplet2Result  # noqa: F401
from .dist import all_gather_triplets, shard_indices  # noqa: F401

__all__ = ["ConfigDict", "load_config", "pairnet_head_cfg", "pairnet_r50", "CrossHead2",
           "PSGTr", "Result", "build_detector", "triplet2Result", "all_gather_triplets",
           "shard_indices", "PipelinedHead"]
