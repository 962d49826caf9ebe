"""Per-kernel-class time AND flop rate of one train step (eager pass, the dispatch packets' own timestamps around every launch;
flops = the algorithmic 2 M N K the library tags each launch with) for a shape of tools/shape_bench.py.
usage: python tools/class_profile.py SHAPE_INDEX [steps] [--json FILE]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

from shape_bench import SHAPES  # noqa: E402
from transformergrooveinfilling_amd import layout  # noqa: E402
from transformergrooveinfilling_amd.engine import StepEngine  # noqa: E402

import json  # noqa: E402

argv = [a for a in sys.argv[1:] if not a.startswith("--")]
jpath = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
if jpath in argv:
    argv.remove(jpath)
i = int(argv[0])
steps = int(argv[1]) if len(argv) > 1 else 20
name, dims, B = SHAPES[i]
dims = dict(dict(embedding_size_src=16), **dims)
eng = StepEngine(batch_size=B, optimizer="sgd", learning_rate=0.05, hit_loss_penalty=0.5, seed=1, **dims)
eng.load_named(layout.init_params(dims, seed=0))
x, y = layout.synthetic_batch(B, dims["embedding_size_src"], seed=2)
eng.x.copy_(torch.from_numpy(x)); eng.y.copy_(torch.from_numpy(y))
for _ in range(5):
    eng.train_step()
prof = eng.profile(steps)
tot = sum(v[1] for v in prof.values())
print(name, "-- kernel time per step %.1f us, %d launches" % (1e3 * tot / steps, sum(v[0] for v in prof.values()) // steps))
peak = 2500.0 if dims.get("precision") in ("bf16", "autocast") else 157.3
out = {}
for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1]):
    tf = v[2] / (v[1] * 1e-3) / 1e12 if v[1] > 0 else 0.0
    out[k] = {"launches_per_step": v[0] / steps, "us_per_step": 1e3 * v[1] / steps, "tflops": tf, "frac_of_mfma_peak": tf / peak}
    print("  %-20s %5.1f launches  %8.2f us/step  %7.2f us/launch  %s" % (k, v[0] / steps, 1e3 * v[1] / steps, 1e3 * v[1] / v[0],
          ("%6.1f TFLOP/s = %4.1f %% of the %s MFMA peak" % (tf, 100 * tf / peak, "bf16" if peak > 1000 else "fp32")) if tf > 0 else ""))
if jpath:
    json.dump({"shape": name, "steps": steps, "peak_tflops": peak, "classes": out}, open(jpath, "w"), indent=1)
