"""Times the rider units of gt_seq_wg.h as launches of their own (gt_debug_seq_wg_phase): per backward phase and token split, alone on
the chip.  usage: python tools/wg_unit_bench.py [batch]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

from shape_bench import SHAPES  # noqa: E402
from transformergrooveinfilling_amd import layout  # noqa: E402
from transformergrooveinfilling_amd.engine import StepEngine, _ptr  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
name, dims, _ = SHAPES[2]
dims = dict(dict(embedding_size_src=16), **dims)
eng = StepEngine(batch_size=B, learning_rate=0.05, seed=1, use_graph=False, **dims)
eng.load_named(layout.init_params(dims, seed=0))
x, y = layout.synthetic_batch(B, 16, seed=2)
xs, ys = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
for _ in range(3):
    eng.train_step(xs, ys, B)
torch.cuda.synchronize()
s = eng.slot(B)
L = dims["num_encoder_layers"]
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for phase in range(1, L + 1):
    for ks in (1, 2):
        def go():
            eng.lib.call("gt_debug_seq_wg_phase", ctypes.byref(s.cfg), _ptr(eng.params), _ptr(eng.grads), _ptr(s.x), _ptr(s.ws), phase, ks, stream)
        for _ in range(3):
            go()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            go()
        e1.record()
        torch.cuda.synchronize()
        print("batch %d phase %d ksplit %d: %.1f us per launch" % (B, phase, ks, 1e3 * e0.elapsed_time(e1) / 20), flush=True)
