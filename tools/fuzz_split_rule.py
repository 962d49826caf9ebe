"""Randomised parity of the d_model 32 / 64 shapes the round-6 two-workgroups-per-sequence rule newly covers (GPU, through the C ABI, against the fp64 oracle)."""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa
import parity
from harness import cfg_dict
rnd = random.Random(11)
fails = n = 0
t_end = time.time() + 420
while time.time() < t_end:
    d = rnd.choice([32, 64])
    H = rnd.choice([16, 16, 16, 8, 4, 2, 1]) if d == 64 else rnd.choice([16, 16, 8, 4, 2, 1])
    F = rnd.choice([16, 32, 48, 64, 128, 256, 320, 512]) if H == 16 else rnd.choice([256, 320, 512, 128])
    L = rnd.choice([1, 2, 3]); B = rnd.choice([1, 2, 3, 5, 9, 16, 17, 33, 64, 100, 128]); S = rnd.choice([16, 27, 5]); p = rnd.choice([0.0, 0.1, 0.3])
    cfg = cfg_dict(d, H, F, L, 0, embedding_size_src=S)
    tag = "d%d H%d F%d L%d B%d S%d p%.1f" % (d, H, F, L, B, S, p)
    try:
        parity.check_step("hip", cfg, B, p, seed=n)
        if n % 3 == 0: parity.check_train_step("hip", cfg, min(B, 16), p)
        if n % 4 == 0: parity.check_predict("hip", cfg, min(B, 4), True)
        print("ok   ", tag, flush=True)
    except Exception as e:  # noqa
        fails += 1
        print("FAIL ", tag, repr(e)[:300], flush=True)
    n += 1
print("%d shapes, %d failures" % (n, fails))
sys.exit(1 if fails else 0)
