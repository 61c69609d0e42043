"""kimimaro_amd -- MI355X-native TEASAR skeletonization (the hot path of seung-lab/kimimaro).

Public names follow kimimaro/__init__.py:18-25 for the part that is in scope.
Importing the package never touches the GPU; calling skeletonize() without libkimi_hip.so + an
MI355X raises HipUnavailableError (there is deliberately no CPU fallback).
"""
import os as _os

# one hardware queue per stream of the lanes (kimimaro_amd/lanes.py): the HIP runtime reads this once, at its first call
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")

from ._abi import HipUnavailableError, KimiHipError  # noqa: F401
from .intake import DEFAULT_TEASAR_PARAMS, DimensionError, skeletonize  # noqa: F401
from .lanes import skeletonize_many  # noqa: F401
from .post import join_close_components, postprocess  # noqa: F401
from .skeleton import Skeleton  # noqa: F401

__version__ = "0.1.0"
