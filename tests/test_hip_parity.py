"""Parity of the real HIP path (libgroove_hip.so on an MI355X) against the oracle, the committed
golden vectors and the reference's demo checkpoint; then size-independent properties at
BASELINE.json's full sizes.  All calls go through the C ABI."""
import numpy as np
import pytest

import parity
from harness import Runner, cfg_dict
from oracle import numpy_groove as ng

pytestmark = pytest.mark.gpu

ENC = cfg_dict(32, 4, 16, 2)
ENCDEC = cfg_dict(32, 4, 16, 2, 2)
C1 = cfg_dict(32, 4, 16, 6)                       # InfillingClosedHH_testing_training.yaml
YAML_HH = cfg_dict(32, 16, 512, 6)                # InfillingClosedHH_training.yaml (head_dim 2)
C2 = cfg_dict(128, 4, 512, 3)                     # BASELINE configs[1]
YAML_KS = cfg_dict(256, 2, 512, 6)                # InfillingKicksAndSnares / InfillingRandom YAML (head_dim 128)
C3 = cfg_dict(256, 2, 512, 2, 2)                  # encoder-decoder (layers reduced for oracle time)
C4 = cfg_dict(512, 8, 512, 2)                     # d_model 512 / 8 heads (layers reduced for oracle time)
SYM = cfg_dict(64, 16, 64, 2, embedding_size_src=27)
YAML_LM = cfg_dict(256, 2, 2048, 2)               # InfillingRandomLow_lm_training.yaml (dim_feedforward 2048, L 8: layers reduced for oracle time)
YAML_LARGE = cfg_dict(256, 16, 64, 3)             # InfillingRandom_test_large.yaml (16 heads of 16, dim_feedforward 64, L 11: layers reduced)
CLI_DEFAULT = cfg_dict(64, 16, 256, 7)            # ref:train.py:43-62 / hyperparameter_defaults.yaml: what train.py runs without a --config (16 heads of 4)


@pytest.mark.parametrize("cfg,B,p", [(ENC, 2, 0.0), (ENC, 5, 0.25), (ENCDEC, 3, 0.0), (ENCDEC, 2, 0.25), (C1, 32, 0.18),
                                     (YAML_HH, 16, 0.24), (SYM, 3, 0.1), (cfg_dict(16, 16, 16, 1), 1, 0.0),
                                     (C2, 8, 0.0), (C2, 64, 0.24), (YAML_KS, 4, 0.3), (C3, 4, 0.0), (C3, 3, 0.3),
                                     (C4, 4, 0.0), (C4, 3, 0.15), (cfg_dict(64, 4, 2048, 1), 2, 0.16),
                                     (YAML_LM, 32, 0.16), (YAML_LARGE, 16, 0.15),       # the two YAMLs at their own batch sizes
                                     (CLI_DEFAULT, 16, 0.2), (cfg_dict(64, 16, 512, 2), 64, 0.1),    # d_model 64 / 16 heads on the SPLIT schedule (round 6)
                                     (cfg_dict(64, 8, 256, 3), 16, 0.2),                             # ... and 8 heads of 8 (zero-padded MFMA attention)
                                     (cfg_dict(64, 4, 256, 3), 16, 0.2), (cfg_dict(64, 2, 512, 2), 32, 0.1)])   # ... head-dim classes 16 / 32
def test_step_parity(cfg, B, p):
    parity.check_step("hip", cfg, B, p)


@pytest.mark.parametrize("cfg,B,p", [(cfg_dict(32, 4, 16, 1), 256, 0.1), (cfg_dict(64, 4, 64, 1), 256, 0.2), (cfg_dict(128, 4, 512, 1), 256, 0.24),
                                     (cfg_dict(256, 2, 512, 1), 256, 0.3), (cfg_dict(512, 8, 512, 1), 256, 0.15),
                                     (cfg_dict(256, 2, 512, 1, 1), 256, 0.1), (cfg_dict(512, 8, 512, 1), 512, 0.1),
                                     (cfg_dict(256, 2, 512, 1), 512, 0.2),
                                     (cfg_dict(512, 8, 512, 1), 64, 0.15)])      # a GPU's share of configs[3]: 2048 tokens, 64x64 ring tiles (gt_gemm64.h: 256 / 768 tiles)
def test_step_parity_large_batches(cfg, B, p):
    """M = 8192 / 16384 tokens: the 32- and 64-row LayerNorm-row tiles of every padded width, 128x128 GEMM tiles and the
    64x64 / 128x128-tile weight-gradient groups (mixed with 32x32-tile problems in one backward) -- tile choices the
    small cases never reach"""
    parity.check_step("hip", cfg, B, p)


@pytest.mark.parametrize("cfg,B,p,nb,seq", [(C2, 64, 0.24, 2, False), (C2, 64, 0.24, 2, True), (C2, 64, 0.24, 1, "split-noride"), (C3, 4, 0.3, 2, True), (C1, 32, 0.18, 2, False), (C1, 32, 0.18, 1, True),
                                            (cfg_dict(128, 4, 512, 1), 8, 0.1, 1, True)])
def test_bucketed_backward(cfg, B, p, nb, seq):
    """data-parallel overlap: the first half of a bucketed backward leaves bucket 0 final, both halves equal the whole
    (sequence-resident path: one bucket -- unless the weight gradients ride in the backward phases: two)"""
    parity.check_bucketed_backward("hip", cfg, B, p, nb, exact=False, seq=seq)


@pytest.mark.parametrize("path", parity.golden_files())
def test_golden_vectors(path):
    parity.check_golden("hip", path)


@pytest.mark.parametrize("path", [p for p in parity.golden_files() if any(t in p for t in ("c4", "c5_sym", "encdec_c3"))])
def test_full_depth_golden_vectors_on_the_bf16_operand_path(path):
    parity.check_golden_bf16("hip", path)


def test_demo_checkpoint():
    parity.check_demo_ckpt("hip")


def test_optimizers():
    parity.check_optimizers("hip", ENC, 4)
    parity.check_optimizers("hip", C2, 2)


@pytest.mark.parametrize("cfg,B,p", [(ENC, 4, 0.2), (ENCDEC, 2, 0.1), (C2, 16, 0.24), (cfg_dict(64, 16, 256, 2), 16, 0.2)])
def test_train_step(cfg, B, p):
    parity.check_train_step("hip", cfg, B, p)


@pytest.mark.parametrize("cfg,B,use_thres", [(ENC, 8, True), (ENC, 3, False), (ENCDEC, 4, True), (C2, 32, True), (ENCDEC, 5, False), (C3, 6, True),
                                             (cfg_dict(128, 4, 64, 2, 2), 33, True)])
def test_predict(cfg, B, use_thres):
    parity.check_predict("hip", cfg, B, use_thres)


@pytest.mark.parametrize("cfg,B", [(ENC, 8), (C2, 32), (ENCDEC, 5), (C3, 4)])
def test_predict_use_pd_samples_on_the_device(cfg, B):
    """model.predict(use_pd=True): hits sampled from the probabilities by a counter hash of the seed (gt_predict_pd), bit-exact against
    the oracle's restatement of the hash outside the decision margin; the encoder-decoder feeds the sampled hits back"""
    parity.check_predict("hip", cfg, B, pd_seed=987654321)


# ---- properties at full size (BASELINE configs[1]: d128/H4/F512/L3, bs 64, dropout 0.24) ---------------
FULL_SIZES = [(C2, 64, 0.24, 0.07),                                                      # BASELINE configs[1]
              (cfg_dict(256, 2, 512, 6, 6), 256, 0.3, 0.02),                              # configs[2]: full encoder-decoder, bs 256
              (cfg_dict(512, 8, 512, 6), 64, 0.3, 0.02),                                  # configs[3]: d512 / 8 heads / 6 layers, 64 per GPU
              (cfg_dict(512, 8, 512, 6, embedding_size_src=27, precision=1), 64, 0.24, 0.02),   # configs[4]: symbolic input, bf16 operands
              (cfg_dict(512, 8, 512, 6, precision=1), 64, 0.24, 0.02)]                    # configs[4]: audio (MSO) input, bf16 operands


@pytest.mark.parametrize("cfg0,B,p,lr", FULL_SIZES)
def test_full_size_properties(cfg0, B, p, lr):
    """size-independent properties at BASELINE's full sizes (no oracle run needed)"""
    cfg = dict(cfg0, dropout=p)
    dec = cfg.get("num_decoder_layers", 0) > 0
    P = ng.init_params(cfg, seed=0)
    x, y = ng.synthetic_batch(B, cfg["embedding_size_src"], seed=1234)
    tgt = parity.shift_right(y) if dec else None
    r = Runner(cfg, B, "hip", rng=(5, 6, 0), lr=lr)
    r.set_params(P)
    # (1) eval forward is a pure function of (weights, inputs): bitwise repeatable
    a = r.forward(x, tgt)
    b = r.forward(x, tgt)
    assert np.array_equal(a, b)
    # (2) sequences are independent: permuting the batch permutes the outputs bit for bit
    perm = np.random.default_rng(0).permutation(B)
    c = r.forward(x[perm], tgt[perm] if dec else None)
    assert np.array_equal(c, a[perm])
    # (3) v in (0,1), o in (-0.5,0.5)
    assert (a[..., 9:18] > 0).all() and (a[..., 9:18] < 1).all() and (np.abs(a[..., 18:]) < 0.5).all()
    # (4) loss is linear in the penalty mask: stats(pen) = stats(0) + pen*(stats(1)-stats(0))
    s0, _ = r.loss(y, 0.0)
    s1, _ = r.loss(y, 1.0)
    sp, _ = r.loss(y, 0.38)
    assert abs(sp[0] - (s0[0] + 0.38 * (s1[0] - s0[0]))) < 1e-5 * abs(sp[0])
    # (5) gradient of a scaled loss scales: backward is linear in d_hvo
    _, d = r.loss(y, 0.38)
    g1 = r.backward(d_hvo=d)
    g2 = r.backward(d_hvo=2.0 * d)
    for k in g1:
        # (bf16 operands: 2 x is exact in bf16, so the rounded operands scale exactly too; only fp32 accumulation differs)
        assert parity.rel_err(g2[k], 2.0 * g1[k]) < 1e-5, k
    # (6) train-mode dropout: same (seed, step) -> same masks; next step -> different masks, same keep rate
    t0 = r.forward(x, tgt, train=True)
    h0 = r.ws_get("hact", 0)
    t1 = r.forward(x, tgt, train=True)
    assert np.array_equal(t0, t1)
    r.train_step(x, y, 0.38)
    assert r.step_state().step == 1
    r.forward(x, tgt, train=True)
    h1 = r.ws_get("hact", 0)
    assert not np.array_equal(h0 == 0, h1 == 0)
    assert abs((h0 == 0).mean() - (h1 == 0).mean()) < 0.01
    # (7) a few SGD steps on a fixed batch reduce the loss
    losses = [r.train_step(x, y, 0.38)[0] for _ in range(20)]
    assert losses[-1] < losses[0]
    assert np.isfinite(r.params.numpy()).all()


# ---- bf16 operand path (gt_config.precision = 1; BASELINE configs[4]: InfillingClosedHH_Symbolic (S=27) vs audio input (S=16),
# d_model 512 / 8 heads / F 512).  Bars and their derivation: parity.check_step_bf16 ------------------------------------------
C5_SYM = cfg_dict(512, 8, 512, 2, embedding_size_src=27)     # layers reduced for oracle time; 6 layers in FULL_SIZES above
C5_MSO = cfg_dict(512, 8, 512, 2)


@pytest.mark.parametrize("cfg,B,p", [(ENC, 2, 0.0), (ENCDEC, 3, 0.25), (SYM, 3, 0.1), (C2, 8, 0.24), (C5_SYM, 4, 0.24), (C5_MSO, 3, 0.0),
                                     (C3, 3, 0.3), (cfg_dict(48, 3, 40, 1, 1), 2, 0.0)])
def test_step_parity_bf16_operands(cfg, B, p):
    parity.check_step_bf16("hip", cfg, B, p)


@pytest.mark.parametrize("cfg,B,p", [(cfg_dict(512, 8, 512, 1, embedding_size_src=27), 256, 0.24), (cfg_dict(128, 4, 512, 1), 256, 0.24),
                                     (cfg_dict(256, 2, 512, 1, 1), 256, 0.1),
                                     (cfg_dict(512, 8, 512, 1, embedding_size_src=27), 64, 0.24)])     # a GPU's share of configs[4]: both operands bf16 on 64x64 tiles (gemm64h_kernel)
def test_step_parity_bf16_operands_large_batches(cfg, B, p):
    """8192 tokens: 128x128 / 64x64 bf16 tiles and the 64- and 128-tile weight-gradient groups"""
    parity.check_step_bf16("hip", cfg, B, p)


def test_bf16_shadows_of_the_gemm_operands():
    """BASELINE configs[4] with gt_set_operand_shadows(1), at a size where every Linear runs on the big-tile kernel (>= 192 tiles of 128 x 128:
    d_model 512, F 512 at 6144 tokens = bs 192): the GEMM operands' producers also write bf16 copies and the Linears, dgrads and weight
    gradients stage those.  Every shadow is bit for bit the rounding of its fp32 tensor, and the per-operation oracle check passes on
    the same shape through that path."""
    from transformergrooveinfilling_amd import _lib
    c5 = dict(d_model=512, n_heads=8, dim_feedforward=512, num_encoder_layers=2, num_decoder_layers=0, embedding_size_src=27)
    assert parity.check_bf16_shadows("hip", c5, 192, 0.3) == 32
    assert parity.check_bf16_shadows("hip", dict(c5, n_heads=16, num_encoder_layers=1), 192, 0.0) == 16
    assert parity.check_bf16_shadows("hip", c5, 64, 0.3) == 32          # 2048 tokens: the same through the 64x64-tile kernels (gt_gemm64.h)
    lib = _lib.get_lib()
    try:
        for level in (2, 1):                        # 2 (the default): operand-only tensors in bf16 ALONE; 1: beside their fp32 tensors
            lib.cdll.gt_set_operand_shadows(level)
            parity.check_step_bf16("hip", c5, 192, 0.3)
    finally:
        lib.cdll.gt_set_operand_shadows(-1)


@pytest.mark.parametrize("cfg,B,p", [(cfg_dict(512, 8, 512, 2, embedding_size_src=27), 64, 0.24),     # a GPU's share of configs[4]: 64x64 bf16-source tiles, attn_bwd <64, 1>
                                     (cfg_dict(512, 8, 512, 1), 256, 0.1),                            # 8192 tokens: 128x128 bf16-source tiles for the QKV projection
                                     (cfg_dict(512, 8, 512, 1), 16, 0.0),                             # below the shadows' threshold: precision 2 is not in force
                                     (cfg_dict(256, 2, 512, 1), 256, 0.3)])                           # head_dim 128
def test_step_parity_precision2_bf16_storage(cfg, B, p):
    """gt_config.precision = 2: Linear outputs of the encoder layers stored in bf16 alone (qkv, pre-LayerNorm outputs, pre-LayerNorm-backward
    dgrads, dctx) -- per-operation teacher-forced parity with the hidden roundings restated, end-to-end sanity bound"""
    r, P, G = parity.check_step_bf16("hip", cfg, B, p, precision=2)
    assert r.precision_in_force() == (2 if B >= 64 else 1)      # (bs 16: below the threshold of the operand shadows -- runs as precision 1)


def test_precision2_train_step_and_autocast_anchor():
    parity.check_train_step_bf16("hip", cfg_dict(512, 8, 512, 1, embedding_size_src=27), 64, 0.2, precision=2)
    dev_ac, dev_fp32, eq = parity.check_autocast_anchor("hip", cfg_dict(512, 8, 512, 2, embedding_size_src=27), 64)
    print("precision 2 vs torch.autocast: rms(device - autocast) %.3g, rms(device - fp32) %.3g, rms(fp32 - autocast) %.3g" % (dev_ac, dev_fp32, eq))


def test_layernorm_row_exchange_of_the_64_tile_linears():
    """LayerNorm forward / backward inside the producing Linear / dgrad at 2048 tokens (the default there; forced on here so that the d_model-256 case takes it too), the 8 workgroups of a
    row block meeting through the in-launch row exchange (csrc/gt_gemm64.h) -- oracle parity at a GPU's share of configs[3] / [4], a train
    step, and the time-out path: a polling bound of one raises the error word and the update applies nothing."""
    import ctypes
    import torch
    from transformergrooveinfilling_amd import _lib
    from transformergrooveinfilling_amd.engine import StepEngine
    import warnings
    lib = _lib.get_lib()
    lib.cdll.gt_set_ln_exchange(1)
    try:
        parity.check_step("hip", cfg_dict(512, 8, 512, 1), 64, 0.15)
        parity.check_step("hip", cfg_dict(256, 2, 512, 2), 64, 0.3)
        parity.check_step_bf16("hip", cfg_dict(512, 8, 512, 1, embedding_size_src=27), 64, 0.24)
        parity.check_train_step("hip", cfg_dict(512, 8, 512, 1), 64, 0.1)
        dims = dict(d_model=512, n_heads=8, dim_feedforward=512, num_encoder_layers=2, num_decoder_layers=0, dropout=0.1, embedding_size_src=16)
        eng = StepEngine(batch_size=64, optimizer="adam", learning_rate=0.01, hit_loss_penalty=0.5, seed=3, use_graph=False, **dims)   # (a captured graph would keep the polling bound it was recorded with)
        from oracle import numpy_groove as ng
        eng.load_named(ng.init_params(dims, seed=1))
        x, y = ng.synthetic_batch(64, 16, seed=4)
        eng.x.copy_(torch.from_numpy(x)); eng.y.copy_(torch.from_numpy(y))
        eng.train_step(); eng.train_step()
        torch.cuda.synchronize()
        assert not eng.check_exchange(eng.slot(64))
        before, m0 = eng.params.clone(), eng.m.clone()
        lib.cdll.gt_set_xchg_spin_max(1)
        eng.train_step()
        torch.cuda.synchronize()
        lib.cdll.gt_set_xchg_spin_max(0)
        assert torch.equal(eng.params, before) and torch.equal(eng.m, m0) and float(eng.grads.abs().max()) == 0.0
        with warnings.catch_warnings(record=True):
            warnings.simplefilter("always")
            assert eng.check_exchange(eng.slot(64))             # noticed, region zeroed, the norm a row pass of its own again
        eng.train_step()
        torch.cuda.synchronize()
        assert not torch.equal(eng.params, before) and not eng.check_exchange(eng.slot(64))
    finally:
        lib.cdll.gt_set_xchg_spin_max(0)
        lib.cdll.gt_set_ln_exchange(-1)
        lib.cdll.gt_set_seq_quad(-1)


def test_layernorm_row_exchange_of_the_big_tile_linears(capfd):
    """Round 6: LayerNorm forward / backward inside the 128x128-tile Linears / dgrads (csrc/gt_gemm64.h gemm32_ln_epilogue) -- the default at
    d_model 512 from 8192 tokens while the grid is resident (512 tiles = two per CU at 16384 tokens).  Oracle parity at bs 256 and bs 512
    (fp32: NT forward, NN dgrad), both operands as bf16 shadows, precision 2 (the pre-norm output rounded to bf16 in registers), a train step;
    the trace proves which launches took the path; a polling bound of one: the update applies nothing, the engine falls back to the row pass."""
    import os
    import warnings
    import torch
    os.environ["GT_TRACE_GEMM64"] = "1"
    try:
        capfd.readouterr()
        parity.check_step("hip", cfg_dict(512, 8, 512, 1), 256, 0.1)
        parity.check_step("hip", cfg_dict(512, 8, 512, 1), 512, 0.15)
        parity.check_step_bf16("hip", cfg_dict(512, 8, 512, 1, embedding_size_src=27), 256, 0.24)
        parity.check_step_bf16("hip", cfg_dict(512, 8, 512, 1), 256, 0.1, precision=2)
        parity.check_train_step("hip", cfg_dict(512, 8, 512, 1), 256, 0.1)
        err = capfd.readouterr().err
    finally:
        del os.environ["GT_TRACE_GEMM64"]
    tr = [ln for ln in err.splitlines() if ln.startswith("[gemm64] ln128")]
    for want in ("fp32-source M 8192 N 512 K 512 NT epi 7 prec 0", "fp32-source M 16384 N 512 K 512 NT epi 7 prec 0", "NN epi 8 prec 0",
                 "bf16-source M 8192 N 512 K 512 NT epi 7 prec 1", "bf16-source M 8192 N 512 K 512 NT epi 8 prec 1"):
        assert any(want in ln for ln in tr), (want, tr[:6])
    # the fail-safe on this geometry: a polling bound of one -> error word -> nothing applied -> the engine's own fall-back (gt_config.flags)
    from transformergrooveinfilling_amd import _lib
    from transformergrooveinfilling_amd.engine import StepEngine
    lib = _lib.get_lib()
    dims = dict(d_model=512, n_heads=8, dim_feedforward=512, num_encoder_layers=1, num_decoder_layers=0, dropout=0.1, embedding_size_src=16)
    eng = StepEngine(batch_size=256, optimizer="sgd", learning_rate=0.01, hit_loss_penalty=0.5, seed=3, use_graph=False, **dims)
    eng.load_named(ng.init_params(dims, seed=1))
    x, y = ng.synthetic_batch(256, 16, seed=4)
    eng.x.copy_(torch.from_numpy(x)); eng.y.copy_(torch.from_numpy(y))
    try:
        eng.train_step(); torch.cuda.synchronize()
        assert not eng.check_exchange(eng.slot(256))
        before = eng.params.clone()
        lib.cdll.gt_set_xchg_spin_max(1)
        eng.train_step(); torch.cuda.synchronize()
        lib.cdll.gt_set_xchg_spin_max(0)
        assert torch.equal(eng.params, before) and float(eng.grads.abs().max()) == 0.0
        with warnings.catch_warnings(record=True):
            warnings.simplefilter("always")
            assert eng.check_exchange(eng.slot(256))
        assert eng.skipped_updates == 1 and eng.slot(256).cfg.flags == eng.FALLBACK_FLAGS
        eng.train_step(); torch.cuda.synchronize()
        assert not torch.equal(eng.params, before) and not eng.check_exchange(eng.slot(256))
    finally:
        lib.cdll.gt_set_xchg_spin_max(0)


def test_layernorm_row_exchange_of_the_32_tile_linears(capfd):
    """Round 6: the reference's d_model-256 YAML shape at its own batch size (K&S / Random: 1024 tokens) runs its LayerNorms inside the
    generic kernel's 32x32-tile Linears / dgrads (csrc/gt_gemm64.h gemm_xln32_epilogue): oracle parity of the step and of three train steps,
    the trace of the launches that took the path, 69 launches per step where there were 91."""
    import os
    os.environ["GT_TRACE_GEMM64"] = "1"
    try:
        capfd.readouterr()
        parity.check_step("hip", YAML_KS, 32, 0.3)
        parity.check_step("hip", cfg_dict(256, 16, 64, 2), 16, 0.15)          # Random_test_large's shape (128 tiles: half the CUs)
        parity.check_train_step("hip", cfg_dict(256, 2, 512, 2), 32, 0.2)
        err = capfd.readouterr().err
    finally:
        del os.environ["GT_TRACE_GEMM64"]
    tr = [ln for ln in err.splitlines() if ln.startswith("[gemm64] ln32")]
    for want in ("M 1024 N 256 K 256 NT epi 7 prec 0", "M 1024 N 256 K 512 NT epi 7 prec 0", "M 1024 N 256 K 768 NN epi 8 prec 0", "M 512 N 256"):
        assert any(want in ln for ln in tr), (want, tr[:6])


def test_ffn_keep_bits(capfd):
    """Round 6: FFN1's epilogue on the ring-tile kernels leaves one keep bit per element of the FFN activation (kept by the dropout and positive)
    and the FFN2 dgrad selects on those instead of reading the activation (csrc/gt_gemm32.h gemm32_store_epilogue, groove_hip.hip ffn_kbits):
    oracle parity of the step on the 64x64 tile (d_model 512 at 2048 tokens, fp32 and both bf16 modes) and the 128x128 tile (8192 tokens), of
    three train steps, of an encoder-decoder model (bits per decoder layer too); the trace proves which launches wrote / read them."""
    import os
    os.environ["GT_TRACE_GEMM64"] = "1"
    try:
        capfd.readouterr()
        parity.check_step("hip", cfg_dict(512, 8, 512, 1), 64, 0.3)
        parity.check_step("hip", cfg_dict(512, 8, 512, 1), 256, 0.1)
        parity.check_step("hip", cfg_dict(256, 2, 512, 1, 1), 64, 0.3)
        parity.check_step_bf16("hip", cfg_dict(512, 8, 512, 1), 64, 0.3)
        parity.check_step_bf16("hip", cfg_dict(512, 8, 512, 1), 64, 0.3, precision=2)
        parity.check_train_step("hip", cfg_dict(512, 8, 512, 2), 64, 0.2)
        err = capfd.readouterr().err
    finally:
        del os.environ["GT_TRACE_GEMM64"]
    tr = [ln for ln in err.splitlines() if ln.startswith("[gemm64] kbits")]
    for want in ("kbits write M 2048 N 512 K 512 NT epi 3 prec 0", "kbits read M 2048 N 512 K 512 NN epi 5 prec 0", "kbits write M 8192 N 512", "kbits read M 8192 N 512",
                 "kbits read M 2048 N 512 K 256", "kbits write M 2048 N 512 K 512 NT epi 3 prec 1", "kbits read M 2048 N 512 K 512 NN epi 5 prec 1"):
        assert any(want in ln for ln in tr), (want, tr[:8])


def test_train_step_bf16_operands():
    parity.check_train_step_bf16("hip", ENC, 4, 0.2)
    parity.check_train_step_bf16("hip", cfg_dict(128, 4, 512, 2), 8, 0.24)


def test_bucketed_backward_bf16_operands():
    parity.check_bucketed_backward("hip", dict(C2, precision=1), 16, 0.24, 2, exact=False)


def test_predict_bf16_operands():
    parity.check_predict("hip", dict(C2, precision=1), 16, True, out_tol=1e-2, margin_tol=5e-3)
    parity.check_predict("hip", dict(ENCDEC, precision=1), 4, True, out_tol=1e-2, margin_tol=5e-3)


# ---- sequence-resident kernels (gt_seq.h): the default for small encoder-only models, so ENC / C1 / YAML_HH / SYM above already run
# on them; here the remaining branches at real batch sizes, and the same models on the one-kernel-per-op path ------------------
@pytest.mark.parametrize("cfg,B,p", [(cfg_dict(64, 4, 64, 2), 16, 0.1), (cfg_dict(64, 2, 32, 2, embedding_size_src=27), 8, 0.2),
                                     (cfg_dict(64, 1, 16, 1), 4, 0.2), (cfg_dict(48, 3, 48, 2), 5, 0.15), (cfg_dict(32, 16, 512, 6), 16, 0.24),
                                     (cfg_dict(16, 2, 16, 1, embedding_size_src=5), 3, 0.0), (cfg_dict(32, 4, 16, 6), 64, 0.18),
                                     (cfg_dict(128, 4, 512, 3), 64, 0.24), (cfg_dict(128, 16, 48, 2), 7, 0.1),
                                     (cfg_dict(96, 6, 80, 2, embedding_size_src=27), 9, 0.1), (cfg_dict(128, 2, 32, 1), 3, 0.0)])
def test_sequence_resident_kernels(cfg, B, p):
    parity.check_step("hip", cfg, B, p)


@pytest.mark.parametrize("cfg,B,p", [(ENC, 5, 0.25), (C1, 32, 0.18), (YAML_HH, 16, 0.24), (SYM, 3, 0.1), (C2, 64, 0.24)])
def test_small_models_on_the_one_kernel_per_op_path(cfg, B, p):
    parity.check_step("hip", cfg, B, p, seq=False)


@pytest.mark.parametrize("cfg,B,p", [(cfg_dict(128, 4, 512, 3), 64, 0.24), (cfg_dict(128, 16, 48, 2), 7, 0.1), (cfg_dict(128, 2, 32, 1), 3, 0.0),
                                     (cfg_dict(128, 8, 128, 4, embedding_size_src=27), 33, 0.2), (YAML_HH, 16, 0.24), (C1, 32, 0.18),
                                     (cfg_dict(32, 2, 256, 2, embedding_size_src=27), 5, 0.1), (cfg_dict(32, 1, 16, 1), 2, 0.0),
                                     (cfg_dict(32, 16, 64, 1), 2, 0.0)])      # head_dim 2 without dropout (saved keep bits read and overruled)
def test_sequence_resident_kernels_two_workgroups_per_sequence(cfg, B, p):
    parity.check_step("hip", cfg, B, p, seq="split")
    parity.check_step("hip", cfg, B, p, seq="whole")


def test_sequence_resident_split_train_step_and_buckets():
    parity.check_train_step("hip", C2, 64, 0.24, seq="split")
    parity.check_train_step("hip", C2, 64, 0.24, seq="whole")
    parity.check_bucketed_backward("hip", C2, 64, 0.24, 2, exact=False, seq="split")
    parity.check_train_step("hip", YAML_HH, 16, 0.24, seq="split")


def test_sequence_resident_train_step():
    parity.check_train_step("hip", C1, 32, 0.18)
    parity.check_train_step("hip", ENC, 4, 0.2, seq=False)


def test_errors_are_reported():
    from transformergrooveinfilling_amd import _lib
    import ctypes
    r = Runner(ENC, 2, "hip")
    with pytest.raises(_lib.GrooveLibError, match="NULL"):
        r.lib.call("gt_forward", ctypes.byref(r.c), None, r.pe.ptr, r.hvo.ptr, None, r.hvo.ptr, r.ws.ptr, None, 0, r.stream)
    bad = _lib.make_config(2, 16, 48, 5, 16, 2)            # 48 % 5 != 0
    with pytest.raises(_lib.GrooveLibError, match="divisible"):
        r.lib.call("gt_forward", ctypes.byref(bad), r.pe.ptr, r.pe.ptr, r.hvo.ptr, None, r.hvo.ptr, r.ws.ptr, None, 0, r.stream)


def test_random_odd_shapes():
    """a fixed-seed slice of tools/fuzz_parity.py: odd widths / head counts / FFN sizes / batch sizes / input dims, encoder-only
    and encoder-decoder, each through full step parity, predict and the bucketed backward"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_parity
    assert fuzz_parity.run(20, seed=11, verbose=False) == 0


@pytest.mark.parametrize("B,S", [(64, 16), (17, 27), (80, 16)])
def test_rider_path_repeats_bit_for_bit_by_default(B, S):
    """SPLIT phases with rider weight gradients (gt_seq_wg.h): one owner per gradient tile (two partial tiles commuting onto zero in the tail)
    -> a run of train steps repeats bit for bit WITHOUT gt_set_deterministic; the packs folded into the update (GT_STEP_PACKS_CURRENT) too."""
    cfgp = dict(C2, dropout=0.24, embedding_size_src=S)
    P = ng.init_params(cfgp, seed=2)
    x, y = ng.synthetic_batch(B, S, seed=77)
    runs = []
    for _ in range(3):
        r = Runner(cfgp, B, "hip", rng=(9, 4, 0), lr=0.05, seq="split")
        assert len(r.lib.grad_buckets(r.c)) == 2                       # (riders on: the backward has two gradient buckets)
        r.set_params(P)
        for step in range(5):
            r.train_step(x, y, 0.4, skip_update=4 if step else 0)
        runs.append(r.params.numpy().copy())
    assert np.array_equal(runs[0], runs[1]) and np.array_equal(runs[0], runs[2])


@pytest.mark.parametrize("cfg,B,p,seq", [(C2, 64, 0.24, True), (C2, 64, 0.24, False), (cfg_dict(256, 2, 512, 2), 32, 0.3, True),
                                         (ENCDEC, 16, 0.2, True)])
def test_deterministic_weight_gradients_repeat_bit_for_bit(cfg, B, p, seq):
    """gt_set_deterministic(1): one workgroup per weight-gradient tile over all tokens -> the whole train step is bitwise
    repeatable (default mode: the token-split partial sums meet in fp32 atomics, last bits vary run to run); and it still is the
    same gradient (oracle parity in that mode)."""
    from transformergrooveinfilling_amd import _lib
    lib = _lib.get_lib()
    lib.cdll.gt_set_deterministic(1)
    try:
        cfgp = dict(cfg, dropout=p)
        dec = cfgp.get("num_decoder_layers", 0) > 0
        P = ng.init_params(cfgp, seed=2)
        x, y = ng.synthetic_batch(B, cfgp["embedding_size_src"], seed=77)
        runs = []
        for _ in range(2):
            r = Runner(cfgp, B, "hip", rng=(9, 4, 0), lr=0.05, seq=seq)
            r.set_params(P)
            for _ in range(4):
                r.train_step(x, y, 0.4)
            runs.append(r.params.numpy().copy())
        assert np.array_equal(runs[0], runs[1])
        if not dec:
            parity.check_step("hip", cfg, min(B, 8), p, seq=seq)
    finally:
        lib.cdll.gt_set_deterministic(0)
