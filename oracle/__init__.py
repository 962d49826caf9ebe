"""CPU oracle for the GrooveTransformer train/predict hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``transformergrooveinfilling_amd/`` may import
this package; only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg do, and there only as the checker / the reported CPU baseline.

Parity status (see DESIGN.md "Oracle"):

* The reference's model / loss / train-loop source (the ``BaseGrooveTransformers`` git
  submodule, ref:.gitmodules:4-6) is NOT vendored under /root/reference, and the reference
  holds no tests or golden vectors.  The arithmetic lives in third-party PyTorch
  (pinned ``pytorch=1.10.2`` at ref:environment.yaml:61): ``nn.TransformerEncoder/Decoder``,
  ``nn.MultiheadAttention``, ``nn.LayerNorm``, ``nn.Linear``, ``BCEWithLogitsLoss``,
  ``MSELoss`` (ref:train.py:176-179), ``optim.SGD/Adam``.
* PINNED: the encoder-only module tree, parameter names/shapes, the ``pe`` buffer and the
  SGD hyper-parameters -- by strict-loading the reference's own artefact
  ``demo/transformer_run_171tyqit_Epoch_1.Model`` into ``oracle.torch_groove`` (stock torch
  modules), and by checking ``oracle.numpy_groove`` (explicit restatement, manual backward)
  against those stock torch modules (tests/test_oracle.py, tests/golden/*.npz).
* PARITY UNPINNED: the glue that only exists in the un-vendored submodule -- decoder wiring,
  output-head activations (h=logit, v=sigmoid, o=0.5*tanh), the ``calculate_loss`` formula and
  ``predict`` thresholding.  They follow the upstream behaviour as recalled plus the evidence at
  the reference's call sites (ref:train.py:55-58,176-179,195-215; ref:evaluator.py:173-177).
"""
