"""Feature I/O and batch construction next to the hot path (reference: dataloader.py)."""
from .shards import CaptionTable, FeatureShard, write_feature_shard  # noqa: F401
from .loader import ShardLoader  # noqa: F401
