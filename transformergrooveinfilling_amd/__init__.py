"""MI355X-native hot path for the GrooveTransformer that pelinski/TransformerGrooveInfilling's
train.py drives: hand-written HIP kernels (csrc/) behind a C ABI (include/groove_hip.h), and the
Python mirror of the reference's model / loss / train-loop interface on top of it."""
from . import _lib, layout  # noqa: F401

__all__ = ["_lib", "layout"]
