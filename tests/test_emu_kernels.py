"""Kernel logic under the host fiber emulator (tests/emu/hip_emu.h): the SAME sources that build
libgroove_hip.so, compiled as host C++, checked against the oracle.  This is the CPU-side
sanitiser/debug build of the kernels -- the product never loads it.  Small shapes only."""
import pytest

import parity
from harness import cfg_dict

ENC = cfg_dict(32, 4, 16, 2)
SYM = cfg_dict(32, 16, 64, 2, embedding_size_src=27)          # head_dim 2, S=27 (unaligned rows)
ENCDEC = cfg_dict(32, 4, 16, 2, 2)


@pytest.mark.parametrize("cfg,B,p", [(ENC, 2, 0.0), (SYM, 1, 0.0), (ENCDEC, 2, 0.0), (ENC, 3, 0.25), (ENCDEC, 1, 0.25),
                                     (cfg_dict(16, 1, 24, 1), 1, 0.0),      # d=16 (row tile wider than d), head_dim 16
                                     (cfg_dict(64, 2, 40, 1), 1, 0.1),      # F not a multiple of 16
                                     (cfg_dict(128, 4, 48, 1), 1, 0.0),     # row-tile 128
                                     (cfg_dict(256, 2, 16, 1), 1, 0.0),     # row-tile 256, head_dim 128
                                     (cfg_dict(32, 2, 16, 1, 1), 2, 0.2),   # enc-dec, head_dim 16: causal + cross attention on MFMA
                                     (cfg_dict(64, 2, 16, 1, 1), 1, 0.0),   # enc-dec, head_dim 32
                                     (cfg_dict(256, 2, 16, 1), 2, 0.15),    # head_dim 128 under dropout: four wave pairs per (sequence, head), column tiles split
                                     (cfg_dict(128, 2, 16, 1, 1), 1, 0.1)]) # enc-dec, head_dim 64: the same split with causal + cross attention
def test_step_parity(cfg, B, p):
    parity.check_step("emu", cfg, B, p)


def test_step_parity_d512():
    parity.check_step("emu", cfg_dict(512, 8, 16, 1), 1, 0.0)             # row-tile 512, head_dim 64


def test_optimizers():
    parity.check_optimizers("emu", ENC, 2)


@pytest.mark.parametrize("cfg,p", [(ENC, 0.2), (ENCDEC, 0.0)])
def test_train_step(cfg, p):
    parity.check_train_step("emu", cfg, 2, p)


@pytest.mark.parametrize("cfg,use_thres", [(ENC, True), (ENC, False), (ENCDEC, True), (ENCDEC, False), (cfg_dict(64, 2, 24, 1, 2), True)])
def test_predict(cfg, use_thres):
    parity.check_predict("emu", cfg, 2, use_thres)


def test_golden_small():
    parity.check_golden("emu", [p for p in parity.golden_files() if "enc_d32h4" in p][0])


def test_demo_checkpoint():
    parity.check_demo_ckpt("emu")


@pytest.mark.parametrize("cfg,B,p,nb", [(ENC, 2, 0.25, 2), (cfg_dict(32, 4, 16, 3), 2, 0.0, 2), (ENCDEC, 2, 0.2, 2), (cfg_dict(32, 2, 16, 1, 1), 1, 0.0, 2),
                                        (cfg_dict(32, 4, 16, 1), 2, 0.1, 1)])
def test_bucketed_backward(cfg, B, p, nb):
    parity.check_bucketed_backward("emu", cfg, B, p, nb, exact=True, seq=False)      # (two buckets: the one-kernel-per-op path)


# ---- bf16 operand path (gt_config.precision = 1) ---------------------------------------------------------------------------
@pytest.mark.parametrize("cfg,B,p", [(ENC, 2, 0.0), (ENCDEC, 3, 0.25), (cfg_dict(64, 16, 64, 2, embedding_size_src=27), 3, 0.1),
                                     (cfg_dict(128, 4, 48, 1), 1, 0.2), (cfg_dict(48, 3, 40, 1, 1), 2, 0.0)])
def test_step_parity_bf16_operands(cfg, B, p):
    parity.check_step_bf16("emu", cfg, B, p)


def test_train_step_bf16_operands():
    parity.check_train_step_bf16("emu", ENC, 2, 0.2)


def test_bucketed_backward_bf16_operands():
    parity.check_bucketed_backward("emu", dict(ENC, precision=1), 2, 0.25, 2, exact=True)      # (bf16: never the sequence-resident path)


def test_predict_bf16_operands():
    parity.check_predict("emu", dict(ENCDEC, precision=1), 2, True, out_tol=1e-2, margin_tol=5e-3)


# ---- big-tile fp32 kernel (gt_gemm32.h: 32x32x2 MFMA, two-deep prefetch ring) -------------------------------------------------
# It serves problems of >= 192 tiles of 128x128 -- far beyond what the emulator can run -- so this test builds a variant of the
# emulator library whose tile rule sends EVERY eligible problem (interior tiles, K % 64 == 0) to that kernel, in a subprocess
# -- and every head_dim-64 attention backward to the LDS-staged kernel (attn_bwd_lds_kernel: batch x heads >= 1024 otherwise) --
# (the harness's library handle is process-wide).
def test_big_tile_kernel_variant():
    import os
    import subprocess
    import sys
    from harness import ROOT
    so = os.path.join(ROOT, "tests", "emu", "libgroove_emu_big.so")
    subprocess.check_call([os.path.join(ROOT, "tests", "emu", "build_emu.sh"), "-DGT_T128_BIG_MIN=1", "-DGT_T128H_MIN=1", "-DGT_WGRAD_T128_MIN=1", "-DGT_ROW32_MIN_M=64", "-DGT_ROW_FUSE_MIN_M=64", "-DGT_ATTN_BWD_LDS_MIN=1"],
                          env=dict(os.environ, GT_EMU_OUT=so),
                          stdout=subprocess.DEVNULL)
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import parity\nfrom harness import cfg_dict\n"
            "parity.check_step('emu', cfg_dict(128, 4, 128, 2), 4, 0.2)\n"          # M = 128: QKV / out-proj / FFN fwd (NT), all dgrads (NN), K = 128 / 384
            "parity.check_step('emu', cfg_dict(128, 2, 256, 1, 1), 4, 0.0)\n"        # encoder-decoder: accumulate epilogue (cross-attention dmem)
            "parity.check_step_bf16('emu', cfg_dict(128, 4, 128, 2), 4, 0.2)\n"     # the same kernels with bf16 fragments (32x32x16 MFMA)
            # bf16 SHADOWS of the GEMM operands (d_model 256 / 512 at precision 1): every shadow = the rounding of its fp32 tensor, bit for
            # bit, and the step through gemm32h_kernel against the oracle (with dropout: masked copies; without: dz itself; head_dim 64 / 32)
            "assert parity.check_bf16_shadows('emu', cfg_dict(256, 4, 128, 2), 4, 0.2) == 32\n"
            "assert parity.check_bf16_shadows('emu', cfg_dict(256, 8, 256, 2), 4, 0.0) == 32\n"
            # ... the step against the oracle: level 2 (default: ctx / hact / dhid / dqkv / masked dz copies in bf16 ALONE), level 1 (beside
            # the fp32 tensors), level 0; an encoder-decoder model (its decoder layers keep fp32 tensors)
            "import harness\n"
            "parity.check_step_bf16('emu', cfg_dict(256, 4, 128, 2), 4, 0.2)\n"      # (default = level 2)
            "parity.check_step_bf16('emu', cfg_dict(256, 8, 256, 1), 4, 0.0)\n"
            "parity.check_step_bf16('emu', cfg_dict(256, 2, 128, 1, 1), 4, 0.1)\n"   # encoder-decoder
            "harness.emu_lib().cdll.gt_set_operand_shadows(1)\n"
            "parity.check_step_bf16('emu', cfg_dict(256, 4, 128, 1), 4, 0.2)\n"
            "harness.emu_lib().cdll.gt_set_operand_shadows(-1)\n"
            "parity.check_step('emu', cfg_dict(256, 2, 64, 2), 2, 0.2)\n"           # ring-body row tiles (gemm32row_kernel): 32-row tiles at d_model 256, NT + NN, K 256 / 64 / 768
            "parity.check_step('emu', cfg_dict(256, 4, 32, 1, 1), 2, 0.0)\n"        # ... encoder-decoder (three norms per decoder layer; head_dim 64: self, causal and cross attention backward from LDS)
            "parity.check_step('emu', cfg_dict(512, 8, 32, 1), 2, 0.1)\n"           # ... 64-row tiles at d_model 512
            "print('ok')\n") % (ROOT, os.path.join(ROOT, "tests"))
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, GT_EMU_LIB_PATH=so), capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-3000:]
    # the same library (operand shadows from 1 tile) with the 64x64 rule lowered: both operands bf16 on gemm64h_kernel, bf16-only storage
    out = _emu_subprocess("assert parity.check_bf16_shadows('emu', cfg_dict(256, 4, 128, 2), 4, 0.2) == 32\n"
                          "parity.check_step_bf16('emu', cfg_dict(256, 8, 256, 1), 4, 0.0)\n"
                          "harness.emu_lib().cdll.gt_set_operand_shadows(1)\n"
                          "parity.check_step_bf16('emu', cfg_dict(256, 4, 128, 1), 4, 0.2)\n",
                          dict(GT_EMU_LIB_PATH=so, GT_T64R_MIN="1", GT_TRACE_GEMM64="1"))
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-3000:]
    assert any(ln.startswith("[gemm64] bf16-source") for ln in out.stderr.splitlines())
    # gt_config.precision = 2 (round 5): the Linear outputs of the encoder layers stored in bf16 alone -- qkv (attention kernels staging bf16 q /
    # k / v / dctx into LDS: head_dim 64 and 128), the out-proj / linear2 outputs ahead of their LayerNorm, the dgrad outputs ahead of a
    # LayerNorm backward, dctx.  Per-operation teacher-forced parity (hidden roundings restated), a train step, the torch.autocast anchor,
    # and the shapes where it falls back to precision 1 (decoder layers; narrow heads)
    out = _emu_subprocess("r, P, G = parity.check_step_bf16('emu', cfg_dict(256, 4, 128, 2), 4, 0.2, precision=2)\n"
                          "assert r.precision_in_force() == 2\n"
                          "parity.check_step_bf16('emu', cfg_dict(256, 2, 256, 1), 4, 0.0, precision=2)\n"       # head_dim 128, no dropout
                          "parity.check_train_step_bf16('emu', cfg_dict(256, 4, 128, 1), 4, 0.1, precision=2)\n"
                          "parity.check_autocast_anchor('emu', cfg_dict(256, 4, 128, 2), 4)\n"
                          "r, P, G = parity.check_step_bf16('emu', cfg_dict(256, 4, 128, 1, 1), 4, 0.1, precision=2)\n"   # encoder-decoder: the decoder layers keep fp32 tensors
                          "assert r.precision_in_force() == 2\n"
                          "r, P, G = parity.check_step_bf16('emu', cfg_dict(256, 8, 128, 1), 4, 0.0, precision=2)\n"      # head_dim 32: runs as precision 1
                          "assert r.precision_in_force() == 1\n",
                          dict(GT_EMU_LIB_PATH=so))
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-3000:]


def test_layernorm_fused_into_the_big_tile_linears():
    """gt_gemm64.h gemm32_ln_epilogue (round 6): EPI_RES_LN / EPI_RES_LNBWD on the 128x128 kernels -- a wave owns 64 x 64, parts of 64 columns,
    two rows per lane, granules [part][4 row quarters][2 values][32 rows].  The big-tile emulator library (every eligible problem on the
    128x128 kernels, operand shadows from one tile) with the shape rule's lower bound at its build-time 1: fp32 NT forward + NN dgrad, bf16
    fragments from fp32 sources, both operands as bf16 shadows (gemm32h_kernel), precision 2, N = 256 (2 column tiles) and 512 (4), one and two
    128-row blocks, with and without dropout; a train step (dgamma / dbeta partials per 128-row block through the reduce); the trace proves
    which launches took the path.  GT_LN_XCHG128=0 keeps the shape on the LayerNorm row pass."""
    import os
    import subprocess
    from harness import ROOT
    so = os.path.join(ROOT, "tests", "emu", "libgroove_emu_big.so")
    subprocess.check_call([os.path.join(ROOT, "tests", "emu", "build_emu.sh"), "-DGT_T128_BIG_MIN=1", "-DGT_T128H_MIN=1", "-DGT_WGRAD_T128_MIN=1", "-DGT_ROW32_MIN_M=64", "-DGT_ROW_FUSE_MIN_M=64", "-DGT_ATTN_BWD_LDS_MIN=1"],
                          env=dict(os.environ, GT_EMU_OUT=so), stdout=subprocess.DEVNULL)
    out = _emu_subprocess("parity.check_step('emu', cfg_dict(256, 4, 128, 2), 4, 0.2)\n"           # M = 128: one row block, two column tiles
                          "parity.check_step('emu', cfg_dict(256, 2, 64, 1), 8, 0.0)\n"            # M = 256: two row blocks, no dropout
                          "parity.check_step('emu', cfg_dict(512, 8, 128, 1), 4, 0.1)\n"           # N = 512: four column tiles, 8 parts
                          "parity.check_train_step('emu', cfg_dict(256, 2, 128, 1), 4, 0.1, seq=False)\n"
                          "parity.check_step_bf16('emu', cfg_dict(256, 4, 128, 2), 4, 0.2)\n"      # both operands as bf16 shadows
                          "parity.check_step_bf16('emu', cfg_dict(512, 8, 128, 1), 4, 0.0)\n"
                          "harness.emu_lib().cdll.gt_set_operand_shadows(0)\n"
                          "parity.check_step_bf16('emu', cfg_dict(256, 4, 128, 1), 4, 0.2)\n"      # bf16 fragments from fp32 sources (PREC 1 body)
                          "harness.emu_lib().cdll.gt_set_operand_shadows(-1)\n"
                          "r, P, G = parity.check_step_bf16('emu', cfg_dict(256, 4, 128, 2), 4, 0.2, precision=2)\n"
                          "assert r.precision_in_force() == 2\n"
                          "parity.check_train_step_bf16('emu', cfg_dict(256, 4, 128, 1), 4, 0.1, precision=2)\n",
                          dict(GT_EMU_LIB_PATH=so, GT_TRACE_GEMM64="1"))
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-3000:]
    tr = [ln for ln in out.stderr.splitlines() if ln.startswith("[gemm64] ln128")]
    for want in ("fp32-source M 128 N 256 K 256 NT epi 7 prec 0", "fp32-source M 128 N 256 K 128 NT epi 7 prec 0", "NN epi 8 prec 0", "M 256 N 256",
                 "M 128 N 512", "bf16-source M 128 N 256 K 256 NT epi 7", "bf16-source M 128 N 256 K 128 NT epi 8", "epi 7 prec 1", "bf16-source M 128 N 512"):
        assert any(want in ln for ln in tr), (want, tr[:8])
    # the keep bits of the FFN activation (gemm32_store_epilogue: EPI_RELU_DROP writes one bit per element, EPI_MASK_NZ reads them instead of hact):
    # fp32 and bf16-only activations, both tile sizes' word layout is one function of (row, col)
    kb = [ln for ln in out.stderr.splitlines() if ln.startswith("[gemm64] kbits")]
    for want in ("kbits write M 128 N 128 K 256 NT epi 3 prec 0", "kbits read M 128 N 128 K 256 NN epi 5 prec 0", "kbits write M 128 N 128 K 256 NT epi 3 prec 1",
                 "kbits read M 128 N 128 K 256 NN epi 5 prec 1", "kbits read M 128 N 128 K 512"):
        assert any(want in ln for ln in kb), (want, kb[:8])
    off = _emu_subprocess("parity.check_step('emu', cfg_dict(256, 4, 128, 1), 4, 0.2)\n", dict(GT_EMU_LIB_PATH=so, GT_TRACE_GEMM64="1", GT_LN_XCHG128="0", GT_FFN_KBITS="0"))
    assert off.returncode == 0 and "ok" in off.stdout and "ln128" not in off.stderr and "kbits" not in off.stderr, off.stderr[-3000:]


def test_layernorm_fused_into_the_32_tile_linears():
    """gt_gemm64.h gemm_xln32_epilogue (round 6): the row exchange under the generic kernel's 32x32 tiles -- the Linears of the reference's
    d_model-256 YAMLs at 512 ... 2048 tokens.  The tile's accumulators are staged in LDS, ONE wave publishes its 64 granules, all four collect
    into LDS, every 16-lane group merges the parts in order.  fp32 NT forward / NN dgrad, bf16 fragments (precision 1), an encoder-decoder
    model (three norms per decoder layer), N = 256 and 512, one to four 32-row blocks, a train step; the shape rule's lower bound (half the
    CUs get a tile) is lowered through the environment, the trace proves which launches took the path."""
    out = _emu_subprocess("parity.check_step('emu', cfg_dict(256, 2, 64, 2), 2, 0.2, seq=False)\n"
                          "parity.check_step('emu', cfg_dict(256, 4, 128, 1), 4, 0.0, seq=False)\n"
                          "parity.check_train_step('emu', cfg_dict(256, 2, 64, 1), 2, 0.1, seq=False)\n"
                          "parity.check_step('emu', cfg_dict(512, 8, 64, 1), 2, 0.1)\n"
                          "parity.check_step_bf16('emu', cfg_dict(256, 4, 64, 1), 2, 0.2)\n"
                          "parity.check_step('emu', cfg_dict(256, 2, 64, 1, 1), 2, 0.1)\n",
                          dict(GT_LN32_MIN="1", GT_TRACE_GEMM64="1"))
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-3000:]
    tr = [ln for ln in out.stderr.splitlines() if ln.startswith("[gemm64] ln32")]
    for want in ("M 64 N 256 K 256 NT epi 7 prec 0", "M 64 N 256 K 64 NN epi 8 prec 0", "M 64 N 256 K 768 NN epi 8 prec 0", "M 128 N 256",
                 "M 64 N 512 K 512 NT epi 7 prec 0", "epi 7 prec 1", "epi 8 prec 1"):
        assert any(want in ln for ln in tr), (want, tr[:8])
    off = _emu_subprocess("parity.check_step('emu', cfg_dict(256, 2, 64, 1), 2, 0.2, seq=False)\n", dict(GT_LN32_MIN="1", GT_TRACE_GEMM64="1", GT_LN_XCHG32="0"))
    assert off.returncode == 0 and "ok" in off.stdout and "ln32" not in off.stderr, off.stderr[-3000:]
    # a CU mask in the environment: the device's CU count says nothing about co-residency any more -- no exchange schedule is chosen (xchg_cus)
    masked = _emu_subprocess("parity.check_step('emu', cfg_dict(256, 2, 64, 1), 2, 0.2, seq=False)\n"
                             "parity.check_step('emu', cfg_dict(128, 4, 64, 1), 4, 0.1)\n",       # (d_model 128: the QUAD schedule's rule reads the same count)
                             dict(GT_LN32_MIN="1", GT_TRACE_GEMM64="1", ROC_GLOBAL_CU_MASK="0xffff"))
    assert masked.returncode == 0 and "ok" in masked.stdout and "ln32" not in masked.stderr and "a CU mask is set" in masked.stderr, masked.stderr[-3000:]


def _emu_subprocess(code, env):
    import os
    import subprocess
    import sys
    from harness import ROOT
    head = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import parity, harness\nfrom harness import cfg_dict\n") % (ROOT, os.path.join(ROOT, "tests"))
    return subprocess.run([sys.executable, "-c", head + code + "print('ok')\n"], env=dict(os.environ, **env), capture_output=True, text=True, timeout=1500)


def test_output_layer_kernel_with_the_loss_inside():
    """d_model 192 (outside the sequence-resident class, a multiple of 64): the OutputLayer runs on heads_fwd_kernel -- 16 rows per workgroup, the
    contraction split over four waves -- and, in the fused train step, that launch computes the loss, d loss / d logits and the statistics too
    (ticket hand-off); encoder-only and encoder-decoder, fp32 and bf16 operands; plain forward (no loss) through check_step"""
    out = _emu_subprocess("big = cfg_dict(192, 3, 32, 1)\n"
                          "parity.check_step('emu', big, 2, 0.1)\n"
                          "parity.check_train_step('emu', big, 2, 0.1)\n"
                          "parity.check_train_step('emu', cfg_dict(128, 2, 32, 1, 1), 1, 0.0)\n"
                          "parity.check_train_step_bf16('emu', big, 2, 0.1)\n", dict(GT_TRACE_HEADS="1"))
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-3000:]
    tr = [ln for ln in out.stderr.splitlines() if ln.startswith("[heads]")]
    assert any("loss 1" in ln and "precision 0" in ln for ln in tr) and any("loss 1" in ln and "precision 1" in ln for ln in tr)
    assert any("loss 0" in ln for ln in tr) and any("M 32 d 128" in ln for ln in tr)


# ---- 64x64 tiles on the ring body (gt_gemm64.h, round 5): problems of 192 .. 2047 tiles otherwise -- the tile rule is lowered through
# the environment (GT_T64R_MIN=1, read once per process: a subprocess) and GT_TRACE_GEMM64=1 proves which launches took the kernel
def test_tile64_ring_kernel():
    out = _emu_subprocess("parity.check_step('emu', cfg_dict(256, 4, 128, 2), 2, 0.2)\n"          # M = 64: NT (QKV, out-proj, FFN, K 256 / 128), NN dgrads (K 768), all four epilogues
                          "parity.check_step('emu', cfg_dict(128, 2, 256, 1, 1), 2, 0.0)\n"        # encoder-decoder: the accumulate epilogue (cross-attention dmem)
                          "parity.check_step_bf16('emu', cfg_dict(256, 4, 128, 1), 2, 0.2)\n"      # fp32 sources rounded at fragment assembly (32x32x16 bf16 MFMA)
                          "parity.check_train_step('emu', cfg_dict(256, 2, 128, 1), 2, 0.1, seq=False)\n",
                          dict(GT_T64R_MIN="1", GT_TRACE_GEMM64="1"))
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-3000:]
    tr = [ln for ln in out.stderr.splitlines() if ln.startswith("[gemm64]")]
    for want in ("NT epi 0 prec 0", "NN epi 0 prec 0", "NT epi 3 prec 0", "NN epi 5 prec 0", "NN epi 6 prec 0", "NT epi 0 prec 1", "K 768"):
        assert any(want in ln for ln in tr), (want, tr[:5])


def test_layernorm_fused_into_the_64_tile_linears():
    """gt_gemm64.h, EPI_RES_LN / EPI_RES_LNBWD: the N / 64 workgroups of a row block exchange their row partials inside the launch
    (forced here on a small shape, gt_set_ln_exchange(1); the default from 3/4 tile per CU): forward (Chan-merged mean / M2) and backward (row sums, dgamma / dbeta partials) against the oracle, NT
    and NN, fp32 and bf16 fragments, N = 256 and 512; the emulator re-runs workgroups that find a partner's ready word missing."""
    from harness import emu_lib
    lib = emu_lib()
    lib.cdll.gt_set_ln_exchange(1)
    try:
        parity.check_step("emu", cfg_dict(256, 4, 128, 2), 2, 0.2)
        parity.check_step_bf16("emu", cfg_dict(256, 4, 128, 1), 2, 0.2)
        parity.check_train_step("emu", cfg_dict(256, 2, 128, 1), 2, 0.1, seq=False)
        parity.check_step("emu", cfg_dict(512, 8, 128, 1), 2, 0.1)
    finally:
        lib.cdll.gt_set_ln_exchange(-1)


def test_gather_and_voice_metrics_kernels():
    """gt_gather_batch / gt_voice_metrics under the emulator against numpy."""
    import ctypes
    import numpy as np
    from harness import emu_lib
    lib = emu_lib()
    r = np.random.default_rng(0)
    n, S, B = 37, 27, 5
    xs, ys = r.random((n, 32, S), dtype=np.float32), r.random((n, 32, 27), dtype=np.float32)
    idx = np.array([3, 36, 0, 3, 99], np.int64)                     # a repeated and an out-of-range index (clamped)
    x, y = np.zeros((B, 32, S), np.float32), np.zeros((B, 32, 27), np.float32)
    vp = lambda a: ctypes.c_void_p(a.ctypes.data)
    lib.call("gt_gather_batch", vp(xs), vp(ys), vp(idx), ctypes.c_int64(n), B, S, vp(x), vp(y), None)
    want = np.clip(idx, 0, n - 1)
    assert np.array_equal(x, xs[want]) and np.array_equal(y, ys[want])
    rows = 3 * 32 + 7                                                # not a multiple of the 64-row workgroup chunk
    pred, gt = r.random((rows, 27), dtype=np.float32), r.random((rows, 27), dtype=np.float32)
    pred[:, :9] = (pred[:, :9] > 0.5); gt[:, :9] = (gt[:, :9] > 0.4)
    out = np.zeros(30, np.float32)
    scratch = np.zeros(int(lib.cdll.gt_voice_metrics_scratch_floats(ctypes.c_int64(rows))), np.float32)
    lib.call("gt_voice_metrics", vp(pred), vp(gt), ctypes.c_int64(rows), vp(out), vp(scratch), None)
    for g, base in enumerate((0, 10, 20)):
        cols = slice(9 * g, 9 * g + 9)
        per = (pred[:, cols] == gt[:, cols]).mean(0) if g == 0 else ((pred[:, cols] - gt[:, cols]) ** 2).mean(0)
        assert np.abs(out[base + 1:base + 10] - per).max() < 1e-6 and abs(out[base] - per.mean()) < 1e-6


# ---- sequence-resident kernels (gt_seq.h): one workgroup per sequence runs the whole forward / backward.  They are the default
# for the small encoder-only models they support, so the tests above already go through them where they apply; here the shapes
# that exercise every branch (head_dim < 16 / 16 / 32 / 64, odd head count, F up to 512, S = 27, d_model 16 ... 128),
# and the same small models on the one-kernel-per-op path (gt_set_seq(0)) ------------------------------------------------------
@pytest.mark.parametrize("cfg,B,p", [(cfg_dict(32, 4, 16, 3), 2, 0.18),                        # testing YAML shape (3 of its 6 layers)
                                     (cfg_dict(32, 16, 512, 1), 1, 0.24),                      # ClosedHH YAML: head_dim 2, F 512
                                     (cfg_dict(64, 4, 64, 2), 1, 0.1),                         # head_dim 16
                                     (cfg_dict(64, 2, 32, 1, embedding_size_src=27), 2, 0.0),  # head_dim 32, symbolic input
                                     (cfg_dict(64, 1, 16, 1), 1, 0.2),                         # head_dim 64, one head (idle wave pair)
                                     (cfg_dict(48, 3, 48, 2), 2, 0.15),                        # odd head count, d_model 48
                                     (cfg_dict(16, 2, 16, 1, embedding_size_src=5), 1, 0.0),
                                     (cfg_dict(128, 4, 512, 2), 1, 0.24),                      # the headline shape (2 of its 3 layers)
                                     (cfg_dict(128, 16, 48, 1), 1, 0.1),                       # d_model class 128, head_dim 8
                                     (cfg_dict(96, 6, 80, 1, embedding_size_src=27), 1, 0.1),  # d_model 96: five idle column tiles
                                     (cfg_dict(128, 2, 32, 1), 1, 0.0)])                       # head_dim 64
def test_sequence_resident_kernels(cfg, B, p):
    parity.check_step("emu", cfg, B, p)


@pytest.mark.parametrize("cfg,B,p", [(cfg_dict(128, 4, 512, 2), 1, 0.24), (cfg_dict(128, 16, 48, 3), 2, 0.1), (cfg_dict(128, 2, 32, 1), 1, 0.0),
                                     (cfg_dict(32, 16, 512, 2), 2, 0.24), (cfg_dict(32, 2, 32, 1), 1, 0.1), (cfg_dict(32, 1, 16, 2), 3, 0.0),
                                     # head_dim 2 in SPLIT mode (d_model 32, 16 heads) runs its attention on the vector ALU, two threads per
                                     # (row, head): the ClosedHH YAML shape above and its F 256 sibling; d_model 128 with 64 heads of 2
                                     # (32 x 64 pairs > 512 threads) falls back to the zero-padded MFMA form -- there also through QUAD
                                     (cfg_dict(32, 16, 256, 1), 3, 0.1), (cfg_dict(128, 64, 32, 1), 1, 0.1),
                                     (cfg_dict(32, 16, 64, 1), 1, 0.0),      # head_dim 2 without dropout: the saved keep bits are read and overruled
                                     # round 6: d_model 64 with 16 heads of 4 (the reference CLI's default shape) on the same schedule and the same
                                     # vector-ALU attention (seq_attn_*_small<4>), with the stage-ahead operand requests of the d_model-32 kernels
                                     (cfg_dict(64, 16, 256, 2), 2, 0.2), (cfg_dict(64, 16, 512, 1), 1, 0.0),
                                     (cfg_dict(64, 8, 256, 1), 2, 0.1),      # ... 8 heads of 8: the same instantiation, zero-padded MFMA attention
                                     (cfg_dict(64, 4, 256, 1), 2, 0.2), (cfg_dict(64, 1, 512, 1), 1, 0.0)])   # ... head-dim classes 16 and 64
def test_sequence_resident_kernels_two_workgroups_per_sequence(cfg, B, p):
    """d_model 128 / 32 (/ 64 with 16 heads), SPLIT mode: 16 token rows per workgroup, one launch per phase -- the same numbers as the whole-sequence kernels.
    At d_model 128 "split" also means: weight gradients as rider workgroups of the backward phases + the tail launch (gt_seq_wg.h)"""
    parity.check_step("emu", cfg, B, p, seq="split")
    parity.check_step("emu", cfg, B, p, seq="whole")


@pytest.mark.parametrize("cfg,B,p", [(cfg_dict(128, 4, 64, 2, embedding_size_src=27), 3, 0.2),    # packed staging of the 27-wide input / output layer
                                     (cfg_dict(128, 8, 80, 1), 5, 0.0)])                          # partial column tiles, odd slab count
def test_rider_weight_gradients_edge_shapes(cfg, B, p):
    parity.check_step("emu", cfg, B, p, seq="split")
    parity.check_step("emu", cfg, B, p, seq="split-noride")


def test_sequence_resident_train_step_and_predict():
    parity.check_train_step("emu", cfg_dict(32, 4, 16, 2), 2, 0.2)
    parity.check_train_step("emu", cfg_dict(32, 16, 16, 2), 2, 0.2)                        # 16 heads: SPLIT by shape at any F (round 6: the vector-ALU attention)
    parity.check_step("emu", cfg_dict(64, 16, 64, 1), 2, 0.1)
    parity.check_train_step("emu", cfg_dict(128, 4, 32, 2), 2, 0.2, seq="split")      # loss fused into the last forward phase, 4 workgroups
    parity.check_train_step("emu", cfg_dict(128, 8, 48, 1), 3, 0.1, seq="whole")
    parity.check_train_step("emu", cfg_dict(64, 16, 256, 2), 2, 0.2)                       # the reference CLI's default shape: SPLIT by shape (round 6)
    parity.check_train_step("emu", cfg_dict(32, 4, 16, 2), 2, 0.0, algo=1)               # the fused update + pack kernel, Adam branch
    parity.check_bucketed_backward("emu", cfg_dict(32, 4, 16, 2), 2, 0.25, 1, exact=True)      # one bucket: the backward is one launch
    # riders: after backward phase p everything from layer L - p + 1 on is final -> two buckets, bit-exact (one owner per gradient tile)
    parity.check_bucketed_backward("emu", cfg_dict(128, 4, 64, 3), 2, 0.2, 2, exact=True, seq="split")
    parity.check_bucketed_backward("emu", cfg_dict(128, 4, 64, 2), 1, 0.0, 2, exact=True, seq="split")
    parity.check_bucketed_backward("emu", cfg_dict(128, 4, 64, 2), 2, 0.1, 1, exact=False, seq="split-noride")
    parity.check_predict("emu", cfg_dict(32, 4, 16, 2), 2, True)
    parity.check_predict("emu", cfg_dict(32, 4, 16, 2), 3, pd_seed=12345)                 # use_pd: hits sampled on the device
    parity.check_predict("emu", cfg_dict(32, 4, 16, 1, 1), 2, pd_seed=777)                # ... fed back through the greedy decode


@pytest.mark.parametrize("cfg,B,p", [(ENC, 3, 0.25), (SYM, 1, 0.0), (cfg_dict(64, 4, 64, 1), 1, 0.1)])
def test_small_models_on_the_one_kernel_per_op_path(cfg, B, p):
    parity.check_step("emu", cfg, B, p, seq=False)
    if cfg is ENC:
        parity.check_train_step("emu", cfg, 2, 0.2, seq=False)
        parity.check_bucketed_backward("emu", cfg, 2, 0.25, 2, exact=True, seq=False)
