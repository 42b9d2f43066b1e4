"""MI355X-native dense stereo cost-volume engine.

Host-side mirror (Python, on torch device tensors) of the reference's operator boundary:
AD_Census, CrossArmAggregation, ScanlineOptimizer, LeftRightConsistency, CrossAggregator,
SAD / NCC / ASW matchers.  All compute goes through the C ABI of libsmt_hip.so
(include/smt.h); torch only provides device memory, streams and torch.distributed.
"""
from ._lib import SmtError, LIB_PATH, VIEW_LEFT, VIEW_RIGHT, VIEW_BOTH  # noqa: F401
from .api import *  # noqa: F401,F403
