"""Host-side logic that needs no GPU: the train.py CLI / YAML contract, the checkpoint finder, the save
schedule and the data-parallel sampler."""
import os

import numpy as np
import pytest
import torch
import yaml

import train as train_cli
from transformergrooveinfilling_amd import parallel, training

# hyper-parameters of the reference's shipped YAMLs (ref:configs/*_training.yaml)
REF_YAMLS = {
    "InfillingClosedHH_training": dict(experiment="InfillingClosedHH", batch_size=16, d_model=32, dim_feedforward=512, dropout=0.24,
                                       optimizer_algorithm="sgd", learning_rate=0.07, n_heads=16, num_encoder_decoder_layers=6,
                                       epochs=400, encoder_only=1, hit_loss_penalty=0.38, load_model=None),
    "InfillingKicksAndSnares_training": dict(experiment="InfillingKicksAndSnares", batch_size=32, d_model=256, dim_feedforward=512,
                                             dropout=0.3, optimizer_algorithm="sgd", learning_rate=0.089, n_heads=2,
                                             num_encoder_decoder_layers=6, epochs=400, encoder_only=1, hit_loss_penalty=0.73, load_model=None),
    "InfillingClosedHH_Symbolic_training": dict(experiment="InfillingClosedHH_Symbolic", batch_size=16, d_model=32, dim_feedforward=512,
                                                dropout=0.24, optimizer_algorithm="sgd", learning_rate=0.07, n_heads=16,
                                                num_encoder_decoder_layers=6, epochs=400, encoder_only=1, hit_loss_penalty=0.38, load_model=None),
    "InfillingRandom_test_large": dict(experiment="InfillingRandom", batch_size=16, d_model=256, dim_feedforward=64, dropout=0.15,
                                       optimizer_algorithm="sgd", learning_rate=0.04, n_heads=16, num_encoder_decoder_layers=11,
                                       epochs=100, encoder_only=1, hit_loss_penalty=1),           # no load_model key
}


@pytest.mark.parametrize("name", sorted(REF_YAMLS))
def test_reference_yaml_keys_load_unchanged(tmp_path, name):
    f = tmp_path / (name + ".yaml")
    f.write_text(yaml.safe_dump(REF_YAMLS[name]))
    args = train_cli.build_parser().parse_args(["--config", str(f), "--d_model", "999"])      # CLI hparams ignored with --config
    hp = train_cli.load_hyperparameters(args)
    assert hp["d_model"] == REF_YAMLS[name]["d_model"] and hp["load_model"] is None
    p = train_cli.model_params(hp, "cuda")
    assert p["model"]["max_len"] == 32 and p["model"]["embedding_size_tgt"] == 27
    assert p["model"]["embedding_size_src"] == (27 if "Symbolic" in name else 16)
    assert p["model"]["num_decoder_layers"] == 0
    assert p["training"] == {"learning_rate": hp["learning_rate"], "batch_size": hp["batch_size"], "hit_loss_penalty": hp["hit_loss_penalty"]}


REF_CONFIGS = "/root/reference/configs"


@pytest.mark.skipif(not os.path.isdir(REF_CONFIGS), reason="the reference checkout is only present in the build container")
def test_the_reference_s_own_yaml_files_load_in_place():
    """Every *_training.yaml of the reference, read where it lies (never copied): the CLI takes it unchanged, the model parameters
    follow ref:train.py:115-143, and the retyped table above agrees with the real files."""
    import glob
    files = sorted(glob.glob(os.path.join(REF_CONFIGS, "*_training.yaml")))
    assert len(files) >= 4
    seen = set()
    for f in files:
        args = train_cli.build_parser().parse_args(["--config", f])
        hp = train_cli.load_hyperparameters(args)
        raw = yaml.safe_load(open(f))
        for k in raw:
            assert hp[k] == raw[k], (f, k)
        p = train_cli.model_params(hp, "cuda")
        m = p["model"]
        assert m["d_model"] == raw["d_model"] and m["n_heads"] == raw["n_heads"] and m["dim_feedforward"] == raw["dim_feedforward"]
        assert m["num_encoder_layers"] == raw["num_encoder_decoder_layers"]
        assert m["num_decoder_layers"] == (0 if raw["encoder_only"] else raw["num_encoder_decoder_layers"])
        assert m["embedding_size_src"] == (27 if raw["experiment"] == "InfillingClosedHH_Symbolic" else 16)     # ref:train.py:129-131
        assert m["d_model"] % m["n_heads"] == 0
        name = os.path.splitext(os.path.basename(f))[0]
        if name in REF_YAMLS:
            seen.add(name)
            for k, v in REF_YAMLS[name].items():
                assert raw.get(k) == v, (name, k, raw.get(k), v)
    assert seen >= {"InfillingClosedHH_training", "InfillingKicksAndSnares_training", "InfillingClosedHH_Symbolic_training"}


def test_cli_without_config_and_overrides(tmp_path):
    args = train_cli.build_parser().parse_args(["--experiment", "InfillingRandom", "--encoder_only", "0", "--d_model", "128",
                                                "--testing", "1", "--override", "n_heads=4"])
    hp = train_cli.load_hyperparameters(args)
    assert hp["epochs"] == 1 and hp["n_heads"] == 4 and hp["d_model"] == 128
    p = train_cli.model_params(hp, "cuda")
    assert p["model"]["num_decoder_layers"] == p["model"]["num_encoder_layers"] == 7
    with pytest.raises(AssertionError):
        train_cli.load_hyperparameters(train_cli.build_parser().parse_args([]))                  # experiment not specified


def test_save_schedule_matches_reference_rule():
    part, full = training.save_schedule(400)
    assert set(range(10)) <= part and {10, 20, 390, 399} <= part and 15 not in part
    assert {0, 9, 10, 30, 399} <= full and 20 not in full
    assert training.save_schedule(5) == (set(range(5)), set(range(5)))
    assert training.save_schedule(100, only_final=True) == ({99}, set())


def test_find_checkpoint_latest_epoch(tmp_path):
    for ep in (0, 3, 12):
        (tmp_path / training.FILE_PATTERN.format("171tyqit", ep)).write_bytes(b"x")
    (tmp_path / training.FILE_PATTERN.format("other", 40)).write_bytes(b"x")
    lm = {"location": "local", "dir": str(tmp_path), "file_pattern": "transformer_run_{}_Epoch_{}.Model", "run": "171tyqit"}
    assert training.find_checkpoint(lm).endswith("transformer_run_171tyqit_Epoch_12.Model")
    assert training.find_checkpoint(dict(lm, epoch=3)).endswith("Epoch_3.Model")
    with pytest.raises(FileNotFoundError):
        training.find_checkpoint(dict(lm, epoch=7))


def test_find_checkpoint_wandb_location(tmp_path, monkeypatch):
    """load_model.location == "wandb" (ref:tutorial.py:98-104: dir = the run path, file = file_pattern.format(run, epoch)): the file is
    fetched with wandb.restore; without the package the request fails loudly instead of looking in a local directory."""
    lm = {"location": "wandb", "dir": "mmil_infilling/InfillingClosedHH/y16izsyy", "file_pattern": "transformer_run_{}_Epoch_{}.Model",
          "epoch": 50, "run": "y16izsyy"}
    monkeypatch.setattr(training, "wandb", None)
    with pytest.raises(RuntimeError, match="wandb"):
        training.find_checkpoint(lm)
    calls = []

    class _Restored:
        name = str(tmp_path / "transformer_run_y16izsyy_Epoch_50.Model")

    class _Wandb:
        run = None

        @staticmethod
        def restore(name, run_path=None):
            calls.append((name, run_path))
            return _Restored()

    monkeypatch.setattr(training, "wandb", _Wandb)
    assert training.find_checkpoint(lm) == _Restored.name
    assert calls == [("transformer_run_y16izsyy_Epoch_50.Model", "mmil_infilling/InfillingClosedHH/y16izsyy")]


def test_sharded_sampler_partitions_every_epoch():
    n, bs, world = 1000, 16, 4
    seen = []
    for r in range(world):
        s = parallel.ShardedBatchSampler(n, bs, r, world, seed=3)
        s.set_epoch(5)
        batches = list(s)
        assert len(batches) == len(s) == (n // world) // bs and all(len(b) == bs for b in batches)
        seen.append(np.concatenate(batches))
    allidx = np.concatenate(seen)
    assert len(np.unique(allidx)) == len(allidx)                     # disjoint shards
    s0 = parallel.ShardedBatchSampler(n, bs, 0, world, seed=3)
    s0.set_epoch(6)
    assert not np.array_equal(np.concatenate(list(s0)), seen[0])     # reshuffled per epoch


def test_synthetic_generator_ranges():
    x, y = train_cli.synthetic_tensors(64, 16, 1)
    assert x.shape == (64, 32, 16) and y.shape == (64, 32, 27)
    h, v, o = y[..., :9], y[..., 9:18], y[..., 18:]
    assert set(h.unique().tolist()) <= {0.0, 1.0} and (v[h == 0] == 0).all() and (o.abs() <= 0.5).all()


def test_model_needs_gpu_and_has_no_cpu_fallback():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from transformergrooveinfilling_amd.model import GrooveTransformerEncoder
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        GrooveTransformerEncoder(d_model=32, nhead=4, num_encoder_layers=1, dim_feedforward=16)


def test_device_batch_loader_matches_sharded_sampler():
    """DeviceBatchLoader (dataset resident on the device, gathers instead of a DataLoader) visits exactly the batches the
    ShardedBatchSampler prescribes: same seeded permutation on every rank, rank-strided, every index at most once."""
    import torch
    from transformergrooveinfilling_amd.parallel import DeviceBatchLoader, ShardedBatchSampler
    n, bs = 103, 8
    x = torch.arange(n, dtype=torch.float32).reshape(n, 1, 1).repeat(1, 32, 16)
    y = -torch.arange(n, dtype=torch.float32).reshape(n, 1, 1).repeat(1, 32, 27)
    for world in (1, 2, 4):
        seen = []
        for rank in range(world):
            dl = DeviceBatchLoader(x, y, bs, "cpu", rank, world, seed=7)
            sm = ShardedBatchSampler(n, bs, rank, world, seed=7, drop_last=(world > 1))
            for ep in (0, 3):
                dl.set_epoch(ep); sm.set_epoch(ep)
                got = [(xb, yb, idx) for xb, yb, idx in dl]
                want = list(sm)
                assert len(got) == len(want) == len(dl)
                for (xb, yb, idx), w in zip(got, want):
                    assert idx.tolist() == w
                    assert torch.equal(xb[:, 0, 0], idx.float()) and torch.equal(yb[:, 0, 0], -idx.float())
                if ep == 0:
                    seen += [i for _, _, idx in got for i in idx.tolist()]
        assert len(seen) == len(set(seen))
        assert len(seen) == (n if world == 1 else (n // world // bs) * bs * world)


def test_reference_evaluator_flags_are_accepted():
    """ref:train.py:18-24: --eval_train / --eval_test / --eval_validation / --dump_eval must parse (command lines and sweep
    programs written for the reference pass them)"""
    a = train_cli.build_parser().parse_args(["--experiment", "InfillingClosedHH", "--eval_train", "0", "--eval_test", "1",
                                             "--eval_validation", "0", "--dump_eval", "0", "--only_final_eval", "1"])
    assert (a.eval_train, a.eval_test, a.eval_validation, a.dump_eval) == ("0", "1", "0", "0")
    d = train_cli.build_parser().parse_args(["--experiment", "x"])
    assert (d.eval_train, d.eval_test, d.eval_validation, d.dump_eval) == (True, False, True, True)     # the reference's defaults
