"""Callers' side of the hot path: batch formation for multi-aspect training."""
from .sampler import AspectRatioBatchSampler, closest_ratio  # noqa: F401
from .features import FeatureDatasetMS, vae_feat_loader  # noqa: F401,E402
