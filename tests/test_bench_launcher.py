"""bench.py's launcher from a bare shell: `python bench.py --gpus 2` must start 2 ranks itself (before any GPU call), run the
data-parallel step sequence (fwd+loss+bwd, all-reduce of the flat gradient buffer, update) and print ONE JSON line, last on
stdout, that says what the collective layer saw.  Runs here on host memory: gloo + the host-emulator build of the kernels
(GT_BENCH_EMU_LIB is a test hook of bench.py; the workload is shrunk because the emulator needs minutes per real step)."""
import json
import os
import subprocess
import sys

from harness import ROOT, emu_lib


def _run(gpus, extra_env=None):
    emu_lib()                                                    # builds tests/emu/libgroove_emu.so if stale
    env = dict(os.environ, GT_BENCH_EMU_LIB=os.path.join(ROOT, "tests", "emu", "libgroove_emu.so"),
               GT_BENCH_EMU_WORK=json.dumps(dict(d_model=32, n_heads=4, dim_feedforward=16, num_encoder_layers=2, batch=2)))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "3", "--warmup", "1",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    return json.loads(lines[-1]), lines


def test_bare_shell_launch_of_two_ranks():
    out, lines = _run(2)
    assert len(lines) == 1                                       # the JSON line is the only thing on stdout
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1 and out["scaling"] == "weak"
    d = out["distributed"]
    assert d["world"] == 2 and d["dist_world_size"] == 2 and d["backend"] == "gloo"
    assert out["config"]["global_batch"] == 4 and out["config"]["parallelism"] == "dp2"
    assert out["value"] > 0 and out["final_loss"] == out["final_loss"]
    # the first run on real ranks picks its own data-parallel recipe: the table of the recipes timed (host memory: the eager ones) and
    # the one kept are part of the record; no pair exchange timed out, no block was thrown away
    t = d["dp_tune"]
    assert t["chosen"] in t["modes"] and set(t["modes"]) <= {"plain_eager", "buckets_eager"} and all(v and v > 0 for v in t["modes"].values())
    assert out["exchange_timeouts"] == 0 and out["blocks_discarded"] == 0 and out["skipped_updates"] == 0
    # ... and what explains a scaling record: every rank's time-outs / skipped updates, the gradient all-reduce timed alone
    assert [r["rank"] for r in d["per_rank"]] == [0, 1] and all(r["exchange_timeouts"] == 0 and r["skipped_updates"] == 0 for r in d["per_rank"])
    assert d["allreduce_alone_us"] > 0 and d["grad_bytes"] > 0


def test_single_rank_and_two_ranks_compute_the_same_kind_of_step():
    one, _ = _run(1)
    assert one["n_gpus"] == 1 and one["distributed"]["dist_world_size"] == 1
    assert 0 < one["final_loss"] < 20


def test_bench_imports_the_oracle_only_in_the_cpu_baseline_leg():
    src = open(os.path.join(ROOT, "bench.py")).read()
    body = src.split("def cpu_baseline", 1)[1].split("\ndef ", 1)[0]
    rest = src.replace(body, "")
    assert "from oracle" in body
    assert "from oracle" not in rest and "import oracle" not in rest
