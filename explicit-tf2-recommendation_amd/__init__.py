"""explicit-tf2-recommendation_amd -- MI355X-native embedding + feature-interaction engine.

Drop-in for the hot path of PatrickHwang/Explicit-tf2-Recommendation's ``CustomLayers.py`` layers (sparse
embedding lookup, FM second order, DCN CrossNet, DIN ActivationUnit, DSSM two-tower) as hand-written gfx950
HIP kernels behind a C ABI (include/mi355rec.h, csrc/libmi355rec.so), with a Python mirror of the
reference's Layer / ModelManager interface on top.  Import name: ``explicit_tf2_recommendation_amd``
(the directory keeps the repository's hyphenated name; the root-level shim module maps one to the other).
"""
from . import _lib  # noqa: F401  (raises if the HIP library is missing: there is no fallback)
from . import ops  # noqa: F401
from . import functional  # noqa: F401
from . import layers  # noqa: F401
from . import data  # noqa: F401
from . import engine  # noqa: F401
from . import sharded  # noqa: F401
from . import model_manager  # noqa: F401
from .model_manager import ModelManager  # noqa: F401

__all__ = ["ops", "functional", "layers", "data", "engine", "sharded", "model_manager", "ModelManager"]
