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
                                     (cfg_dict(64, 2, 16, 1, 1), 1, 0.0)])  # enc-dec, head_dim 32
def test_step_parity(cfg, B, p):
    parity.check_step("emu", cfg, B, p)


def test_step_parity_d512():
    parity.check_step("emu", cfg_dict(512, 8, 16, 1), 1, 0.0)             # row-tile 512, head_dim 64


@pytest.mark.parametrize("cfg,B,p", [(ENC, 3, 0.25), (SYM, 1, 0.0), (cfg_dict(16, 1, 24, 1), 1, 0.0), (cfg_dict(64, 2, 40, 2), 1, 0.1),
                                     (cfg_dict(128, 4, 48, 2), 1, 0.2), (cfg_dict(256, 2, 16, 1), 1, 0.0)])
def test_step_parity_row_chain_kernels(cfg, B, p):
    """the opt-in fused row-chain path (gt_chain.h: loader waves + MFMA waves), every DPAD instantiation"""
    parity.check_step("emu", cfg, B, p, chain=True)


def test_train_step_row_chain_kernels():
    parity.check_train_step("emu", ENC, 2, 0.2, chain=True)


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
    parity.check_bucketed_backward("emu", cfg, B, p, nb, exact=True)


# ---- bf16 operand path (gt_config.precision = 1) ---------------------------------------------------------------------------
@pytest.mark.parametrize("cfg,B,p", [(ENC, 2, 0.0), (ENCDEC, 3, 0.25), (cfg_dict(64, 16, 64, 2, embedding_size_src=27), 3, 0.1),
                                     (cfg_dict(128, 4, 48, 1), 1, 0.2), (cfg_dict(48, 3, 40, 1, 1), 2, 0.0)])
def test_step_parity_bf16_operands(cfg, B, p):
    parity.check_step_bf16("emu", cfg, B, p)


def test_train_step_bf16_operands():
    parity.check_train_step_bf16("emu", ENC, 2, 0.2)


def test_bucketed_backward_bf16_operands():
    parity.check_bucketed_backward("emu", dict(ENC, precision=1), 2, 0.25, 2, exact=True)


def test_predict_bf16_operands():
    parity.check_predict("emu", dict(ENCDEC, precision=1), 2, True, out_tol=1e-2, margin_tol=5e-3)
