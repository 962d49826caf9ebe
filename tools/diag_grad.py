"""Debug aid: localise a gradient mismatch of one parity case (GPU, through the C ABI) against the fp64 oracle.
usage: python tools/diag_grad.py  [d H F L B p]   -- prints per-row error of layers.0.linear1.weight and ReLU/dropout mask flips."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # first: one HIP runtime (the one torch bundles) for both torch and libgroove_hip.so
import numpy as np
import parity
from harness import Runner, cfg_dict
from oracle import numpy_groove as ng

a = [int(v) for v in sys.argv[1:6]] if len(sys.argv) > 5 else [128, 4, 512, 3, 64]
p = float(sys.argv[6]) if len(sys.argv) > 6 else 0.24
cfg = dict(cfg_dict(a[0], a[1], a[2], a[3]), dropout=p)
B = a[4]
P = ng.init_params(cfg, seed=3, perturb=0.05)
x, y = ng.synthetic_batch(B, cfg["embedding_size_src"], seed=5)
rng = (1234, 99, 7)
r = Runner(cfg, B, "hip", rng=rng)
r.set_params(P)
hvo = r.forward(x, None, train=p > 0)
(h, v, o), C = ng.forward(P, cfg, x, tgt=None, rng=rng if p > 0 else None, dtype=np.float64)
print("forward max-abs", np.abs(hvo - np.concatenate([h, v, o], -1)).max())
for l in range(a[3]):
    hg = r.ws_get("hact", l).reshape(-1, a[2]); hr = C["enc"][l]["hact"].reshape(-1, a[2])
    flips = np.argwhere((hg == 0) != (hr == 0))
    print("layer", l, "hact mask flips:", len(flips), [(int(m), int(f), float(hg[m, f]), float(hr[m, f])) for m, f in flips[:5]])
stats, d_hvo = r.loss(y, 0.47)
rstats, dpred = ng.calculate_loss((h, v, o), y.astype(np.float64), 0.47)
G = r.backward(train=p > 0)
Gr = ng.backward(P, cfg, C, dpred, dtype=np.float64)
worst = sorted(((parity.rel_err(G[k], Gr[k]), k) for k in Gr), reverse=True)[:6]
print("worst tensors:", worst)
k = worst[0][1]
D = np.abs(G[k] - Gr[k]); 
if D.ndim == 2:
    rows = D.max(1); top = np.argsort(rows)[::-1][:5]
    print(k, "rows with largest error:", [(int(t), float(rows[t])) for t in top], "median row err", float(np.median(rows)), "ref max", float(np.abs(Gr[k]).max()))
