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
if os.environ.get("GT_SEQ_SPLIT") != "1":       # one workgroup per sequence: one launch forward, one backward
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
    print(" forward layer 1 FFN1, wave 0: issue first B loads %d; then per tile [wait + MFMAs, epilogue, gap]:" % (sub[1] - sub[0]),
          " ".join("[%d %d %d]" % (sub[3 + 3 * i] - sub[2 + 3 * i], sub[4 + 3 * i] - sub[3 + 3 * i], (sub[5 + 3 * i] - sub[4 + 3 * i]) if i < 3 else 0) for i in range(4)),
          " stage start -> first stamp %d" % (sub[0] - st[2 + 10 * 1 + 4]))

if os.environ.get("GT_SEQ_SPLIT") == "1":
    print(" SPLIT mode (two workgroups per sequence, one launch per phase): cycles of workgroup 0")
    for ph in range(L + 1):
        print("  forward phase %d: %d cycles" % (ph, st[61 + 2 * ph] - st[60 + 2 * ph]), end="")
        if ph >= 1:
            b = 2 + 10 * (ph - 1)
            print("   load state %d  attention %d  out-proj %d  norm1 %d  FFN1 %d  FFN2 %d  norm2 %d  then %d" %
                  (st[b + 1] - st[60 + 2 * ph], st[b + 2] - st[b + 1], st[b + 3] - st[b + 2], st[b + 4] - st[b + 3], st[b + 5] - st[b + 4],
                   st[b + 6] - st[b + 5], st[b + 7] - st[b + 6], st[61 + 2 * ph] - st[b + 7]), end="")
            q = 300 + 4 * (ph - 1)
            if st[q] > st[b + 5]:      # QUAD forward: the FFN2 stage = half-K product, send, wait for the partner's partial tile
                print("   [FFN2: product %d  send %d  wait + receive %d]" % (st[q] - st[b + 5], st[q + 1] - st[q], st[q + 2] - st[q + 1]))
            else:
                print()
        else:
            print()
    for ph in range(L + 1):
        print("  backward phase %d: %d cycles" % (ph, st[161 + 2 * ph] - st[160 + 2 * ph]), end="")
        if ph >= 1:
            k = ph - 1                                   # attention / in-proj dgrad of layer L-ph use the stamp base of that layer
            b = 102 + 10 * k
            print("   load state + attention bwd %d  in-proj dgrad %d" % (st[b + 5] - st[160 + 2 * ph], st[b + 6] - st[b + 5]), end="")
            q = 400 + 4 * ph
            if st[q] > st[160 + 2 * ph] and st[q + 1] > st[q]:   # head_dim-2 attention: state tiles in LDS, pass A (dS, dq), pass B (dk, dv), store
                print("  [state %d  pass A %d  pass B %d  store %d]" % (st[q] - st[160 + 2 * ph], st[q + 1] - st[q], st[q + 2] - st[q + 1], st[b + 5] - st[q + 2]), end="")
            elif st[q] > st[160 + 2 * ph]:
                print("  [state %d  attention bwd %d]" % (st[q] - st[160 + 2 * ph], st[b + 5] - st[q]), end="")
            if ph < L:
                c = 102 + 10 * ph
                print("  norm2 bwd %d  FFN2 dgrad %d  FFN1 dgrad %d  norm1 bwd %d  out-proj dgrad %d" %
                      (st[c] - st[b + 6], st[c + 1] - st[c], st[c + 2] - st[c + 1], st[c + 3] - st[c + 2], st[c + 4] - st[c + 3]))
            else:
                print()
        else:
            c = 102
            print("   prologue %d  norm2 bwd %d  FFN2 dgrad %d  FFN1 dgrad %d  norm1 bwd %d  out-proj dgrad %d" %
                  (st[101] - st[160], st[c] - st[101], st[c + 1] - st[c], st[c + 2] - st[c + 1], st[c + 3] - st[c + 2], st[c + 4] - st[c + 3]))

if os.environ.get("GT_SEQ_SPLIT") == "1" and len(st) >= 1024 + 4 * 512:
    w = st[1024:1024 + 4 * 512].reshape(512, 4)
    w = w[w[:, 0] != 0]
    if len(w):
        cyc, rt = w[:, 2] - w[:, 0], (w[:, 3] - w[:, 1]) * 10.0          # shader cycles; ns (100 MHz clock)
        t0 = w[:, 1].min()
        print(" forward phase 1, all %d workgroups: cycles min %d / median %d / max %d;  ns min %d / median %d / max %d;  clock %.2f GHz (median);"
              "  first start -> last end %d ns; start spread %d ns" % (len(w), cyc.min(), np.median(cyc), cyc.max(), rt.min(), np.median(rt), rt.max(),
                                                                      np.median(cyc / rt), (w[:, 3].max() - t0) * 10, (w[:, 1].max() - t0) * 10))
