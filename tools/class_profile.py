"""Per-kernel-class time of one train step (eager pass, HIP events around every launch) for a shape of tools/shape_bench.py.
usage: python tools/class_profile.py SHAPE_INDEX [steps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

from shape_bench import SHAPES  # noqa: E402
from transformergrooveinfilling_amd import layout  # noqa: E402
from transformergrooveinfilling_amd.engine import StepEngine  # noqa: E402

i = int(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
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
for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1]):
    print("  %-20s %5.1f launches  %8.2f us/step  %7.2f us/launch" % (k, v[0] / steps, 1e3 * v[1] / steps, 1e3 * v[1] / v[0]))
