"""Per-stage cycle counts of the sequence-resident kernels (diagnostic build: csrc/groove_hip.hip with -DGT_SEQ_STAMPS, selected
through GT_LIB_PATH).  usage: GT_LIB_PATH=.../libgroove_stamps.so python tools/seq_stamps.py SHAPE_INDEX"""
import os
import sys

os.environ.setdefault("GT_SEQ", "1")            # the whole-sequence kernels are the stamped ones
os.environ.setdefault("GT_SEQ_SPLIT", "0")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from shape_bench import SHAPES  # noqa: E402
from transformergrooveinfilling_amd import layout  # noqa: E402
from transformergrooveinfilling_amd.engine import StepEngine  # noqa: E402

name, dims, B = SHAPES[int(sys.argv[1])]
dims = dict(dict(embedding_size_src=16), **dims)
eng = StepEngine(batch_size=B, learning_rate=0.05, seed=1, use_graph=False, **dims)
eng.load_named(layout.init_params(dims, seed=0))
x, y = layout.synthetic_batch(B, dims["embedding_size_src"], seed=2)
xs, ys = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
for _ in range(5):
    eng.train_step(xs, ys, B)
torch.cuda.synchronize()
s = eng.slot(B)
off, cnt = eng.lib.ws_find(s.cfg, "stamps")
st = s.ws[off:off + cnt].cpu().numpy().view(np.int64)
L = dims["num_encoder_layers"]
print(name, "(shader-clock cycles, workgroup 0; 1000 cycles = 0.42 us at 2.4 GHz)")
fl = ["in-proj", "attention", "out-proj(+ctx copy)", "norm1", "FFN1", "FFN2 split-K", "norm2"]
print(" forward: input tile load %d, input layer %d, total %d" % (st[1] - st[0], st[2] - st[1], st[2 + 10 * L] - st[0]))
for l in range(L):
    b = 2 + 10 * l
    d = [int(st[b + i + 1] - st[b + i]) for i in range(7)]
    print("  layer %d: " % l + "  ".join("%s %d" % (n, c) for n, c in zip(fl, d)) + "   = %d" % sum(d))
print("  final norm + output layer %d" % (st[2 + 10 * L] - st[2 + 10 * (L - 1) + 7]))
bl = ["norm2 bwd", "FFN2 dgrad", "FFN1 dgrad split-K", "norm1 bwd", "out-proj dgrad", "attention bwd", "in-proj dgrad split-K"]
print(" backward: output-layer dgrad + final norm %d, total %d" % (st[101] - st[100], st[102 + 10 * L] - st[100]))
prev = st[101]
for k in range(L):
    b = 102 + 10 * k
    t = [prev] + [int(st[b + i]) for i in range(7)]
    d = [int(t[i + 1] - t[i]) for i in range(7)]
    prev = st[b + 6]
    print("  layer %d: " % (L - 1 - k) + "  ".join("%s %d" % (n, c) for n, c in zip(bl, d)) + "   = %d" % sum(d))
print("  input-layer epilogue %d" % (st[102 + 10 * L] - prev))

sub = st[200:216]
if sub[0]:
    print(" forward layer 1 in-proj, wave 0: issue first B loads %d; then per tile [wait + MFMAs, epilogue, gap]:" % (sub[1] - sub[0]),
          " ".join("[%d %d %d]" % (sub[3 + 3 * i] - sub[2 + 3 * i], sub[4 + 3 * i] - sub[3 + 3 * i], (sub[5 + 3 * i] - sub[4 + 3 * i]) if i < 2 else 0) for i in range(3)))
