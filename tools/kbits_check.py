"""Keep bits of the FFN activation (GT_FFN_KBITS) select exactly what the zero test of hact selects: every shape runs forward + loss +
backward twice -- in a child process with GT_FFN_KBITS=0 and in one with the default -- and the gradients are compared (equal up to the
order of the weight gradients' atomic adds: 1e-6 of the tensor's maximum; a wrong mask bit shows as 1e-2).  GPU, through the C ABI.
usage: python tools/kbits_check.py            (parent)      python tools/kbits_check.py child OUT.npz   (one process)"""
import hashlib
import os
import subprocess
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

SHAPES = [  # d H F Le Ld B p precision
    (256, 2, 512, 2, 2, 256, 0.3, 0),     # C3's layers (64x64 tiles, encoder-decoder)
    (512, 8, 512, 2, 0, 64, 0.3, 0),      # the per-GPU share of C4 (64x64)
    (512, 8, 512, 1, 0, 512, 0.3, 0),     # C4 (128x128)
    (512, 8, 512, 2, 0, 512, 0.3, 1),     # C5 bf16 (128x128, bf16 sources)
    (512, 8, 512, 2, 0, 64, 0.3, 2),      # precision 2 (64x64, bf16 sources)
    (256, 2, 512, 2, 0, 32, 0.1, 0),      # K&S (generic kernel: no bits)
]

if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch  # noqa: F401
    import numpy as np
    from harness import Runner, cfg_dict
    from oracle import numpy_groove as ng
    out = {}
    for d, H, F, Le, Ld, B, p, prec in SHAPES:
        cfg = dict(cfg_dict(d, H, F, Le, Ld), dropout=p)
        if prec:
            cfg["precision"] = prec
        P = ng.init_params(cfg, seed=3, perturb=0.05)
        x, y = ng.synthetic_batch(B, cfg["embedding_size_src"], seed=5)
        r = Runner(cfg, B, "hip", rng=(1234, 99, 7))
        r.set_params(P)
        r.forward(x, y if Ld else None, train=True)
        r.loss(y, 0.47)
        G = r.backward(train=True)
        out["%d_%d_%d_%d_%d_%d_p%d" % (d, H, F, Le, Ld, B, prec)] = r.flatten(G)
    np.savez(sys.argv[2], **out)
    sys.exit(0)

import numpy as np
import tempfile
res = {}
tmp = tempfile.mkdtemp()
for tag, env in (("off", dict(GT_FFN_KBITS="0")), ("off2", dict(GT_FFN_KBITS="0")), ("on", {})):
    f = os.path.join(tmp, tag + ".npz")
    out = subprocess.run([sys.executable, os.path.abspath(__file__), "child", f], env=dict(os.environ, GT_TRACE_GEMM64="1", **env), capture_output=True, text=True)
    if out.returncode:
        print(out.stderr[-3000:]); sys.exit(1)
    res[tag] = dict(np.load(f))
    print(tag, "launches that read keep bits:", sum("kbits read" in ln for ln in out.stderr.splitlines()))
bad = 0
for k in res["off"]:
    ref = res["off"][k]
    e_on = float(np.abs(res["on"][k] - ref).max() / np.abs(ref).max()), float(np.abs(res["off2"][k] - ref).max() / np.abs(ref).max())
    ok = e_on[0] <= max(1e-6, 4 * e_on[1])
    bad += not ok
    print("%-28s on vs off %.2e   (off vs off, another process: %.2e)   %s" % (k, e_on[0], e_on[1], "ok" if ok else "DIFFERENT"))
print("FAIL" if bad else "ok: the keep bits select what the zero test selects")
sys.exit(1 if bad else 0)
