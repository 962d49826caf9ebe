"""Drop-in for the reference's un-vendored `BaseGrooveTransformers` submodule import
(ref:train.py:12 `from BaseGrooveTransformers import initialize_model, calculate_loss, train_loop`;
ref:tutorial.py:6): the same three names, served by the MI355X-native path in
transformergrooveinfilling_amd.  With this repo's root on PYTHONPATH the reference's train.py /
tutorial.py / evaluator.py run against the HIP kernels without edits to those lines."""
from transformergrooveinfilling_amd.model import GrooveTransformer, GrooveTransformerEncoder  # noqa: F401
from transformergrooveinfilling_amd.training import calculate_loss, initialize_model, train_loop  # noqa: F401

__all__ = ["initialize_model", "calculate_loss", "train_loop", "GrooveTransformer", "GrooveTransformerEncoder"]
