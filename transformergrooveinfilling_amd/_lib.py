"""ctypes binding of libgroove_hip.so (C ABI declared in include/groove_hip.h).

The product path has NO CPU fallback: if the HIP library is missing this module raises, it never
substitutes another implementation.  (tests/ may call ``load(path)`` with the host-emulator build
of the same sources, tests/emu/libgroove_emu.so, to debug kernels without a GPU.)
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.environ.get("GT_LIB_PATH") or os.path.join(_HERE, "lib", "libgroove_hip.so")   # GT_LIB_PATH: experiments only

GT_T = 32
GT_TGT = 27
GT_VOICES = 9


class GtConfig(ctypes.Structure):
    _fields_ = [("batch", ctypes.c_int32), ("src_dim", ctypes.c_int32), ("d_model", ctypes.c_int32),
                ("n_heads", ctypes.c_int32), ("dim_ff", ctypes.c_int32), ("n_enc_layers", ctypes.c_int32),
                ("n_dec_layers", ctypes.c_int32), ("dropout", ctypes.c_float), ("precision", ctypes.c_int32),
                ("flags", ctypes.c_int32)]


CFG_NO_QUAD = 1          # gt_config.flags (include/groove_hip.h): per-caller schedule switches
CFG_NO_LN_XCHG = 2


class GtStepState(ctypes.Structure):
    _fields_ = [("seed_lo", ctypes.c_uint32), ("seed_hi", ctypes.c_uint32), ("step", ctypes.c_uint32),
                ("opt_step", ctypes.c_uint32), ("lr", ctypes.c_float), ("grad_scale", ctypes.c_float),
                ("beta1", ctypes.c_float), ("beta2", ctypes.c_float), ("eps", ctypes.c_float),
                ("pad2", ctypes.c_float * 3)]


STEP_STATE_BYTES = ctypes.sizeof(GtStepState)   # 48

_vp, _cfgp = ctypes.c_void_p, ctypes.POINTER(GtConfig)
_SIGS = {
    "gt_last_error": (ctypes.c_char_p, []),
    "gt_version": (ctypes.c_int, []),
    "gt_param_count": (ctypes.c_int, [_cfgp, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]),
    "gt_param_layout": (ctypes.c_int, [_cfgp, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64),
                                       ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32)]),
    "gt_workspace_bytes": (ctypes.c_size_t, [_cfgp]),
    "gt_ws_find": (ctypes.c_int, [_cfgp, ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int64),
                                  ctypes.POINTER(ctypes.c_int64)]),
    # cfg, params, pe, x, tgt_in, hvo_out, ws, state, train, stream
    "gt_forward": (ctypes.c_int, [_cfgp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_int, _vp]),
    # cfg, hvo, y, penalty, stats, d_hvo, stream
    "gt_loss": (ctypes.c_int, [_cfgp, _vp, _vp, ctypes.c_float, _vp, _vp, _vp]),
    # cfg, params, grads, x, tgt_in, hvo, d_hvo, ws, state, train, accumulate, stream
    "gt_backward": (ctypes.c_int, [_cfgp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_int, ctypes.c_int, _vp]),
    # algo, params, grads, m, v, n, state, zero_grads, stream
    "gt_optimizer_step": (ctypes.c_int, [ctypes.c_int, _vp, _vp, _vp, _vp, ctypes.c_int64, _vp, ctypes.c_int, _vp]),
    # cfg, algo, params, grads, m, v, pe, x, y, penalty, hvo_out, stats, tgt_scratch, ws, state, skip_update, stream
    "gt_train_step": (ctypes.c_int, [_cfgp, ctypes.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_float, _vp, _vp,
                                     _vp, _vp, _vp, ctypes.c_int, _vp]),
    # cfg, params, pe, x, hvo_out, thres, use_thres, tgt_scratch, ws, stream
    "gt_predict": (ctypes.c_int, [_cfgp, _vp, _vp, _vp, _vp, ctypes.c_float, ctypes.c_int, _vp, _vp, _vp]),
    # cfg, params, pe, x, hvo_out, seed, tgt_scratch, ws, stream
    "gt_predict_pd": (ctypes.c_int, [_cfgp, _vp, _vp, _vp, _vp, ctypes.c_uint32, _vp, _vp, _vp]),
    "gt_predict_pd_at": (ctypes.c_int, [_cfgp, _vp, _vp, _vp, _vp, ctypes.c_uint32, ctypes.c_int64, _vp, _vp, _vp]),
    "gt_voice_metrics_scratch_floats": (ctypes.c_int64, [ctypes.c_int64]),
    # hvo_pred, hvo_gt, n_rows, out30, scratch, stream
    "gt_voice_metrics": (ctypes.c_int, [_vp, _vp, ctypes.c_int64, _vp, _vp, _vp]),
    # xs, ys, idx, n_seq, batch, src_dim, x, y, stream
    "gt_gather_batch": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, _vp, _vp, _vp]),
    # cfg, algo, params, grads, m, v, ws, state, zero_grads, stream
    "gt_optimizer_step_ws": (ctypes.c_int, [_cfgp, ctypes.c_int, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_int, _vp]),
    "gt_grad_buckets": (ctypes.c_int, [_cfgp, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]),
    "gt_profile_enable": (ctypes.c_int, [ctypes.c_int]),
    "gt_set_overlap": (ctypes.c_int, [ctypes.c_int]),
    "gt_set_seq": (ctypes.c_int, [ctypes.c_int]),
    "gt_set_seq_split": (ctypes.c_int, [ctypes.c_int]),
    "gt_set_seq_quad": (ctypes.c_int, [ctypes.c_int]),
    "gt_set_xchg_spin_max": (ctypes.c_int, [ctypes.c_int]),
    "gt_dp_guard": (ctypes.c_int, [_cfgp, _vp, _vp, _vp]),
    "gt_set_ln_exchange": (ctypes.c_int, [ctypes.c_int]),
    "gt_debug_occupy_cus": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, _vp]),
    "gt_set_operand_shadows": (ctypes.c_int, [ctypes.c_int]),
    "gt_layout_epoch": (ctypes.c_int, []),
    "gt_operand_shadow_level": (ctypes.c_int, [_cfgp]),
    "gt_precision_in_force": (ctypes.c_int, [_cfgp]),
    "gt_workspace_init": (ctypes.c_int, [_cfgp, _vp, _vp]),
    "gt_set_seq_ride": (ctypes.c_int, [ctypes.c_int]),
    "gt_set_deterministic": (ctypes.c_int, [ctypes.c_int]),
    # cfg, params, grads, x, ws, phase, ksplit, stream
    "gt_debug_seq_wg_phase": (ctypes.c_int, [_cfgp, _vp, _vp, _vp, _vp, ctypes.c_int, ctypes.c_int, _vp]),
    "gt_step_launches": (ctypes.c_int, [ctypes.POINTER(GtConfig)]),
    "gt_profile_report": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int]),
}
EXPORTS = tuple(_SIGS)


class GrooveLibError(RuntimeError):
    pass


class GrooveLib:
    """Thin checked wrapper: every entry point raises GrooveLibError(gt_last_error()) on failure."""

    def __init__(self, path=None):
        path = path or DEFAULT_LIB
        if not os.path.exists(path):
            raise GrooveLibError(
                "HIP library %s not found. Build it with transformergrooveinfilling_amd/csrc/build.sh "
                "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback." % path)
        self.path = path
        self.cdll = ctypes.CDLL(path)
        for name, (res, args) in _SIGS.items():
            fn = getattr(self.cdll, name)      # AttributeError if the library lacks a declared symbol
            fn.restype, fn.argtypes = res, args

    def _chk(self, rc, what):
        if rc != 0:
            raise GrooveLibError("%s failed: %s" % (what, self.cdll.gt_last_error().decode()))

    def call(self, name, *args):
        self._chk(getattr(self.cdll, name)(*args), name)

    def param_layout(self, cfg):
        """-> (total_floats, [(offset, size, rows, cols)]) in state-dict order."""
        nt, nf = ctypes.c_int64(), ctypes.c_int64()
        self.call("gt_param_count", ctypes.byref(cfg), ctypes.byref(nt), ctypes.byref(nf))
        n = nt.value
        off, siz = (ctypes.c_int64 * n)(), (ctypes.c_int64 * n)()
        rows, cols = (ctypes.c_int32 * n)(), (ctypes.c_int32 * n)()
        self.call("gt_param_layout", ctypes.byref(cfg), off, siz, rows, cols)
        return nf.value, [(off[i], siz[i], rows[i], cols[i]) for i in range(n)]

    def workspace_floats(self, cfg):
        b = self.cdll.gt_workspace_bytes(ctypes.byref(cfg))
        if b == 0:
            raise GrooveLibError("gt_workspace_bytes failed: %s" % self.cdll.gt_last_error().decode())
        return b // 4

    def grad_buckets(self, cfg):
        """-> [(offset, count)] of the flat gradient buffer in the order backward completes them (1 or 2 buckets)."""
        off, cnt = (ctypes.c_int64 * 2)(), (ctypes.c_int64 * 2)()
        n = self.cdll.gt_grad_buckets(ctypes.byref(cfg), off, cnt)
        if n < 1:
            raise GrooveLibError("gt_grad_buckets failed: %s" % self.cdll.gt_last_error().decode())
        return [(off[i], cnt[i]) for i in range(n)]

    def ws_find(self, cfg, name, layer=0):
        o, c = ctypes.c_int64(), ctypes.c_int64()
        self.call("gt_ws_find", ctypes.byref(cfg), name.encode(), layer, ctypes.byref(o), ctypes.byref(c))
        return o.value, c.value


_default = None


def get_lib():
    """The process-wide HIP library (loaded on first use; raises if it is not built)."""
    global _default
    if _default is None:
        _default = GrooveLib()
    return _default


# 0 fp32 | 1 bf16 GEMM operands | 2 ... and bf16 storage of the Linear outputs ("autocast": what torch.autocast(bfloat16) keeps in bf16)
PRECISION = {"fp32": 0, "f32": 0, "float32": 0, 0: 0, None: 0, "bf16": 1, "bfloat16": 1, 1: 1, "bf16_storage": 2, "bf16s": 2, "autocast": 2, 2: 2}


def make_config(batch, src_dim, d_model, n_heads, dim_ff, n_enc_layers, n_dec_layers=0, dropout=0.0, precision=0, flags=0):
    return GtConfig(int(batch), int(src_dim), int(d_model), int(n_heads), int(dim_ff), int(n_enc_layers),
                    int(n_dec_layers), float(dropout), PRECISION[precision], int(flags))
