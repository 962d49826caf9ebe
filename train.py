"""train.py -- `python train.py --config configs/X.yaml`, the reference's training CLI (ref:train.py:14-101) on the
MI355X-native hot path.

Same flags and YAML keys as the reference: with --config every hyper-parameter comes from the YAML
(experiment, batch_size, d_model, dim_feedforward, dropout, optimizer_algorithm, learning_rate, n_heads,
num_encoder_decoder_layers, epochs, encoder_only, hit_loss_penalty, load_model); without it from the CLI.
The reference's own YAMLs (InfillingClosedHH / KicksAndSnares / Random ...) load unchanged.

Host side stays Python: data come from (a) the reference's processed dataset when its dataset modules are
importable (`load_processed_dataset`, ref:train.py:153-155), (b) --data-npz FILE with arrays
`inputs (N,32,S)` / `outputs (N,32,27)` (= the dataset's processed_inputs/processed_outputs tensors,
ref:dataset.py:263-264), or (c) --synthetic N sequences from the SURVEY 8(d) generator.  W&B is optional.
Data-parallel: launch with torch.distributed.run, one process per GPU.
"""
import argparse
import os
import pprint
import sys
import time

import yaml

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

KEYS = ("encoder_only", "optimizer_algorithm", "d_model", "n_heads", "dropout", "num_encoder_decoder_layers",
        "hit_loss_penalty", "batch_size", "dim_feedforward", "learning_rate", "epochs", "load_model")


def build_parser():
    p = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    p.add_argument("--paths", default="configs/paths.yaml", help="paths file (experiment -> dataset dirs)")
    p.add_argument("--testing", default=False, help="testing mode (1 epoch)")
    p.add_argument("--wandb", default=True, help="log to wandb (when installed)")
    p.add_argument("--only_final_eval", default=False)
    # which sets are scored per epoch and whether the scores are dumped (ref:train.py:18-24, 219-250).  The reference's
    # GrooveEvaluator (host-side plots / audio, un-vendored) is outside the hot path; its SCALAR leg -- predict the set,
    # per-voice hit accuracy / velocity / offset errors (ref:evaluator.py:516-525) -- runs here on the device
    # (transformergrooveinfilling_amd.metrics.evaluate: chunked predict + gt_voice_metrics, ONE 30-float D2H per set) on the epochs of
    # the reference's save schedule, logged under the reference's set identifiers (Train_Set / Test_Set / Validation_Set).
    p.add_argument("--eval_train", default=True, help="evaluator train set")
    p.add_argument("--eval_test", default=False, help="evaluator test set")
    p.add_argument("--eval_validation", default=True, help="evaluator validation set")
    p.add_argument("--dump_eval", default=True, help="dump the per-epoch evaluation scalars (eval_<Set>_Epoch_<n>.json beside the checkpoints)")
    p.add_argument("--load_model", default=None)
    p.add_argument("--notes", default=None)
    p.add_argument("--tags", default=None)
    p.add_argument("--config", default=None, help="yaml config file; if given the hyper-parameter flags are ignored")
    p.add_argument("--experiment", default=None)
    p.add_argument("--encoder_only", default=1, type=int)
    p.add_argument("--optimizer_algorithm", default="sgd", type=str)
    p.add_argument("--d_model", default=64, type=int)
    p.add_argument("--n_heads", default=16, type=int)
    p.add_argument("--dropout", default=0.2, type=float)
    p.add_argument("--num_encoder_decoder_layers", default=7, type=int)
    p.add_argument("--hit_loss_penalty", default=1, type=float)
    p.add_argument("--batch_size", default=16, type=int)
    p.add_argument("--dim_feedforward", default=256, type=int)
    p.add_argument("--learning_rate", default=0.05, type=float)
    p.add_argument("--epochs", default=100, type=int)
    # build-side additions
    p.add_argument("--data-npz", default=None, help="npz with inputs (N,32,S) and outputs (N,32,27)")
    p.add_argument("--synthetic", default=0, type=int, help="train on N synthetic sequences")
    p.add_argument("--eval-npz", default=None,
                   help="npz with the evaluation subsets: train_inputs/train_gt, test_inputs/test_gt, validation_inputs/validation_gt "
                        "((n,32,S) / (n,32,27): the evaluators' processed_inputs / processed_gt, ref:evaluator.py:509-511); missing sets are "
                        "skipped.  With --synthetic and no file: three seeded synthetic subsets")
    p.add_argument("--eval-size", default=1024, type=int, help="sequences per synthetic / train-derived evaluation subset")
    p.add_argument("--host-loader", action="store_true",
                   help="feed batches through torch's DataLoader from host memory (the reference's way) instead of keeping the "
                        "dataset in HBM and gathering batches on the device")
    p.add_argument("--watch-log-freq", type=int, default=1000, help="gradient histograms to wandb every N batches (ref:train.py:150 wandb.watch log_freq; 0 = off)")
    p.add_argument("--save-dir", default=None, help="where checkpoints go (default: wandb run dir or ./checkpoints)")
    p.add_argument("--override", action="append", default=[], metavar="KEY=VALUE", help="override a YAML key (bench shapes)")
    p.add_argument("--seed", default=0, type=int)
    p.add_argument("--precision", default="fp32", choices=["fp32", "bf16", "autocast"],
                   help="bf16: the Linear GEMMs take bf16 operands on the matrix cores (fp32 accumulate, fp32 master weights; BASELINE configs[4]); "
                        "autocast: ... and the encoder layers' Linear outputs are stored in bf16 (gt_config.precision = 2: what torch.autocast keeps in bf16)")
    p.add_argument("--deterministic", action="store_true",
                   help="bitwise-reproducible weight gradients (gt_set_deterministic: no token split in the weight-gradient kernels; +9-30 %% step time)")
    return p


def load_hyperparameters(args):
    """All from the YAML or all from the CLI (ref:train.py:69-96)."""
    if args.config is not None:
        with open(args.config, "r") as f:
            hp = yaml.safe_load(f)
    else:
        hp = {k: getattr(args, k) for k in KEYS}
    if args.testing:
        hp["epochs"] = 1
    if args.experiment is not None:
        hp["experiment"] = args.experiment
    for kv in args.override:
        k, v = kv.split("=", 1)
        hp[k] = yaml.safe_load(v)
    assert "experiment" in hp, "experiment not specified"
    hp.setdefault("load_model", None)          # InfillingRandom_test_large.yaml lacks the key
    return hp


def model_params(hp, device):
    """params dict of ref:train.py:115-143."""
    enc_only = bool(hp["encoder_only"])
    return {"model": {"experiment": hp["experiment"], "encoder_only": hp["encoder_only"], "optimizer": hp["optimizer_algorithm"],
                      "d_model": hp["d_model"], "n_heads": hp["n_heads"], "dim_feedforward": hp["dim_feedforward"],
                      "dropout": hp["dropout"], "num_encoder_layers": hp["num_encoder_decoder_layers"],
                      "num_decoder_layers": 0 if enc_only else hp["num_encoder_decoder_layers"], "max_len": 32,
                      "embedding_size_src": 27 if hp["experiment"] == "InfillingClosedHH_Symbolic" else 16,
                      "embedding_size_tgt": 27, "device": device},
            "training": {"learning_rate": hp["learning_rate"], "batch_size": hp["batch_size"],
                         "hit_loss_penalty": hp["hit_loss_penalty"]},
            "load_model": hp["load_model"]}


def synthetic_tensors(n, src_dim, seed):
    """x ~ U[0,1); hits ~ Bernoulli(0.15), vel = U*h, off = (U-0.5)*h  (SURVEY 8d)."""
    import torch
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(n, 32, src_dim, generator=g)
    h = (torch.rand(n, 32, 9, generator=g) < 0.15).float()
    v = torch.rand(n, 32, 9, generator=g) * h
    o = (torch.rand(n, 32, 9, generator=g) - 0.5) * h
    return x, torch.cat([h, v, o], -1)


def load_data(args, hp, src_dim):
    import numpy as np
    import torch
    if args.synthetic:
        return synthetic_tensors(args.synthetic, src_dim, args.seed + 1)
    if args.data_npz:
        z = np.load(args.data_npz)
        return torch.from_numpy(z["inputs"]).float(), torch.from_numpy(z["outputs"]).float()
    try:                                    # the reference's own host-side data path, if present on PYTHONPATH
        from process_dataset import load_processed_dataset
    except Exception as e:
        raise SystemExit("no data: pass --synthetic N or --data-npz FILE, or put the reference's dataset modules on "
                         "PYTHONPATH (%s)" % e)
    with open(args.paths, "r") as f:
        paths = yaml.safe_load(f)
    ds = load_processed_dataset(paths[hp["experiment"]]["datasets"]["train"], exp=hp["experiment"])
    return ds.processed_inputs, ds.processed_outputs


def _flag(v):
    """the reference's boolean flags arrive as strings from sweep command lines (`--eval_test False`)"""
    return bool(v) and str(v).lower() not in ("false", "0", "no", "none")


def load_eval_sets(args, x, y, src_dim):
    """{"Train_Set" | "Test_Set" | "Validation_Set": (inputs, gt)} for the sets the flags enable (ref:train.py:160-174: three
    pickled evaluators, each holding processed_inputs / processed_gt of a subset).  Sources: --eval-npz; else, with --synthetic,
    seeded synthetic subsets; the train subset defaults to the head of the training tensors."""
    import numpy as np
    import torch
    want = {"Train_Set": _flag(args.eval_train), "Test_Set": _flag(args.eval_test), "Validation_Set": _flag(args.eval_validation)}
    key = {"Train_Set": "train", "Test_Set": "test", "Validation_Set": "validation"}
    sets = {}
    z = np.load(args.eval_npz) if args.eval_npz else None
    for i, (name, on) in enumerate(want.items()):
        if not on:
            continue
        k = key[name]
        if z is not None and k + "_inputs" in z.files:
            sets[name] = (torch.from_numpy(z[k + "_inputs"]).float(), torch.from_numpy(z[k + "_gt"]).float())
        elif name == "Train_Set":
            n = min(args.eval_size, len(x))
            sets[name] = (x[:n], y[:n])
        elif args.synthetic:
            sets[name] = synthetic_tensors(args.eval_size, src_dim, args.seed + 100 + i)
    return sets


def main(argv=None):
    args = build_parser().parse_args(argv)
    hp = load_hyperparameters(args)
    import torch
    from torch.utils.data import DataLoader, TensorDataset
    from transformergrooveinfilling_amd import parallel
    from transformergrooveinfilling_amd import metrics
    from transformergrooveinfilling_amd.training import calculate_loss, initialize_model, save_schedule, train_loop
    rank, local, world = parallel.init_distributed()
    if rank == 0:
        pprint.pprint(hp)
    if not torch.cuda.is_available():
        raise SystemExit("train.py runs the MI355X hot path; no ROCm GPU is visible (there is no CPU fallback)")
    device = "cuda:%d" % local
    torch.manual_seed(args.seed)
    wb = None
    if args.wandb and str(args.wandb) != "False" and rank == 0:
        try:
            import wandb as wb
            wb.init(config=hp, project=hp["experiment"], job_type="train", notes=args.notes, tags=args.tags)
        except Exception:
            wb = None
    params = model_params(hp, device)
    params["model"]["precision"] = hp.get("precision", args.precision)
    params["seed"] = args.seed                # dropout stream of this run (the data-parallel rank is mixed in by the model)
    model, optimizer, initial_epoch = initialize_model(params)
    model.eval_log = []                       # the evaluation leg's records (also what the tests read)
    # ref:train.py:150 wandb.watch(model, log_freq=1000): on the fused path the gradients never pass through autograd, so the hooks
    # wandb installs would never fire -- train_loop logs the same "gradients/<name>" histograms itself every watch_log_freq batches
    model.watch_log_freq = args.watch_log_freq if wb is not None else 0
    if args.deterministic:
        model.engine.lib.cdll.gt_set_deterministic(1)
    parallel.broadcast_parameters(model.engine.params)
    x, y = load_data(args, hp, params["model"]["embedding_size_src"])

    class _Triples(TensorDataset):           # the reference's dataset yields (x, y, idx) (ref:dataset.py:355-356)
        def __getitem__(self, i):
            return self.tensors[0][i], self.tensors[1][i], i

    ds = _Triples(x, y)
    eval_sets = load_eval_sets(args, x, y, params["model"]["embedding_size_src"]) if rank == 0 else {}
    test = eval_sets.get("Test_Set", (None, None))
    val = eval_sets.get("Validation_Set", (None, None))
    if not args.host_loader:
        # the whole dataset in HBM, batches gathered on the device
        sampler = loader = parallel.DeviceBatchLoader(x, y, hp["batch_size"], device, rank, world, seed=args.seed)
    elif world > 1:
        sampler = parallel.ShardedBatchSampler(len(ds), hp["batch_size"], rank, world, seed=args.seed)
        loader = DataLoader(ds, batch_sampler=sampler, pin_memory=True)
    else:
        sampler = None
        loader = DataLoader(ds, batch_size=hp["batch_size"], shuffle=True, pin_memory=True)
    bce, mse = torch.nn.BCEWithLogitsLoss(reduction="none"), torch.nn.MSELoss(reduction="none")
    part, full = save_schedule(hp["epochs"], only_final=bool(args.only_final_eval) and str(args.only_final_eval) != "False")
    save_dir = args.save_dir or (wb.run.dir if wb else os.path.join(ROOT, "checkpoints"))
    os.makedirs(save_dir, exist_ok=True)
    for ep in range(initial_epoch, hp["epochs"]):
        if sampler:
            sampler.set_epoch(ep)
        t0 = time.perf_counter()
        m = train_loop(dataloader=loader, groove_transformer=model, encoder_only=hp["encoder_only"], opt=optimizer, epoch=ep,
                       loss_fn=calculate_loss, bce_fn=bce, mse_fn=mse, device=device, hit_loss_penalty=hp["hit_loss_penalty"],
                       # test / validation LOSS after the epoch's batches, as the reference's train_loop arguments (ref:train.py:204-212)
                       test_inputs=test[0], test_gt=test[1], validation_inputs=val[0], validation_gt=val[1],
                       save=(rank == 0 and (ep in part or ep in full)), save_dir=save_dir,
                       run_id=(wb.run.id if wb else "local"))
        torch.cuda.synchronize()
        if rank == 0:
            n = min(len(loader) * hp["batch_size"], len(ds) // world) * world
            print("Epoch %d: loss %.5f  hit_acc %.4f  (%.0f sequences/s)" % (ep, m["train/loss"], m["train/hit_accuracy"],
                                                                           n / (time.perf_counter() - t0)))
            # the evaluation leg (ref:train.py:219-250 -> log_eval, ref:evaluator.py:516-525) on the epochs of the save schedule:
            # predict each enabled set on the device, per-voice metrics on the device, one 30-float copy per set
            if ep in part or ep in full:
                for name, (xin, gt) in eval_sets.items():
                    sc = metrics.evaluate(model, xin, gt)
                    rec = {"%s/%s" % (name, k): v for k, v in sc.items()}
                    model.eval_log.append(dict(rec, epoch=ep))
                    print("  %s: hits accuracy %.4f  velocity MSE %.5f  offset MSE %.5f" %
                          (name, sc["Hits_Accuracy_Overall"], sc["Velocity_MSE_Overall"], sc["Offset_MSE_Overall"]))
                    if wb:
                        wb.log(dict(rec, epoch=ep), commit=False)
                    if _flag(args.dump_eval):
                        import json
                        with open(os.path.join(save_dir, "eval_%s_Epoch_%d.json" % (name, ep)), "w") as f:
                            json.dump(dict(sc, epoch=ep), f)
            if wb:
                wb.log({"epoch": ep}, commit=True)
    if wb:
        wb.finish()
    return model


if __name__ == "__main__":
    main()
