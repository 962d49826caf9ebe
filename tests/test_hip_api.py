"""The reference-facing Python API on the real GPU: model modules (state_dict = checkpoint keys), autograd path,
calculate_loss, fused train_loop, optimizers, checkpoints / resume, predict, train.py."""
import os

import numpy as np
import pytest
import torch
import yaml

from oracle import numpy_groove as ng

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _params(enc_only=True, algo="sgd", d=32, H=4, F=16, L=2, dropout=0.0, lr=0.094, pen=0.47, S=16, load=None):
    return {"model": {"experiment": "InfillingClosedHH", "encoder_only": int(enc_only), "optimizer": algo, "d_model": d, "n_heads": H,
                      "dim_feedforward": F, "dropout": dropout, "num_encoder_layers": L, "num_decoder_layers": 0 if enc_only else L,
                      "max_len": 32, "embedding_size_src": S, "embedding_size_tgt": 27, "device": "cuda"},
            "training": {"learning_rate": lr, "batch_size": 8, "hit_loss_penalty": pen}, "load_model": load}


def _cfg(p):
    m = p["model"]
    return dict(d_model=m["d_model"], n_heads=m["n_heads"], dim_feedforward=m["dim_feedforward"], num_encoder_layers=m["num_encoder_layers"],
                num_decoder_layers=m["num_decoder_layers"], embedding_size_src=m["embedding_size_src"], dropout=m["dropout"])


def test_demo_checkpoint_strict_loads_and_matches_golden():
    from BaseGrooveTransformers import initialize_model
    z = np.load(os.path.join(GOLD, "demo_ckpt.npz"))
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd/")}
    model, opt, ep0 = initialize_model(_params(d=32, H=4, F=16, L=6, lr=0.094))
    assert list(model.state_dict().keys()) == list(sd.keys())                       # same names, same order, pe included
    assert model.load_state_dict(sd, strict=True).missing_keys == []
    assert len(list(model.parameters())) == 78 and ep0 == 0
    osd = opt.state_dict()
    assert osd["param_groups"][0]["lr"] == 0.094 and osd["param_groups"][0]["momentum"] == 0
    assert osd["param_groups"][0]["params"] == list(range(78)) and osd["state"][0] == {"momentum_buffer": None}
    model.eval()
    with torch.no_grad():
        h, v, o = model(torch.from_numpy(z["x"]).cuda())
    for t, k in ((h, "h_H4"), (v, "v_H4"), (o, "o_H4")):
        assert t.shape == (4, 32, 9) and np.abs(t.cpu().numpy() - z[k]).max() < 2e-5


@pytest.mark.parametrize("enc_only", [True, False])
def test_autograd_path_matches_oracle(enc_only):
    from BaseGrooveTransformers import calculate_loss, initialize_model
    from transformergrooveinfilling_amd.training import shift_right
    p = _params(enc_only=enc_only)
    cfg = _cfg(p)
    model, opt, _ = initialize_model(p)
    P = ng.init_params(cfg, seed=2, perturb=0.05)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()}, strict=False)
    x, y = ng.synthetic_batch(6, 16, seed=3)
    xt, yt = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    bce, mse = torch.nn.BCEWithLogitsLoss(reduction="none"), torch.nn.MSELoss(reduction="none")
    model.train()
    opt.zero_grad()
    pred = model(xt) if enc_only else model(xt, shift_right(yt))
    out = calculate_loss(pred, yt, bce, mse, 0.47)
    out[0].backward()
    tgt = None if enc_only else np.concatenate([np.zeros_like(y[:, :1]), y[:, :-1]], 1)
    (h, v, o), C = ng.forward(P, cfg, x, tgt=tgt, dtype=np.float64)
    st, dpred = ng.calculate_loss((h, v, o), y.astype(np.float64), 0.47)
    assert abs(out[0].item() - st[0]) < 1e-5 and abs(out[1] - st[1]) < 1e-6 and abs(out[2] - st[2]) < 1e-3
    G = ng.backward(P, cfg, C, dpred, dtype=np.float64)
    for n, prm in model.named_parameters():
        g = prm.grad.cpu().numpy()
        assert np.abs(g - G[n]).max() <= 2e-4 * np.abs(G[n]).max() + 1e-9, n
    opt.step()
    torch.cuda.synchronize()
    for n, prm in model.named_parameters():
        assert np.abs(prm.detach().cpu().numpy() - (P[n] - 0.094 * G[n])).max() < 1e-5, n
    with pytest.raises(ValueError):
        calculate_loss(pred, yt, torch.nn.BCEWithLogitsLoss(), mse, 0.47)              # reduction must be 'none'


def test_train_loop_fast_path_checkpoint_and_resume(tmp_path):
    from BaseGrooveTransformers import calculate_loss, initialize_model, train_loop
    from torch.utils.data import DataLoader, TensorDataset
    import train as train_cli
    p = _params(d=64, H=4, F=64, L=2, dropout=0.1, lr=0.05, pen=0.38)
    model, opt, ep0 = initialize_model(p)
    x, y = train_cli.synthetic_tensors(256, 16, 0)

    class Triples(TensorDataset):
        def __getitem__(self, i):
            return self.tensors[0][i], self.tensors[1][i], i

    # (the shuffle is seeded and the criterion is the mean of an epoch's logged minibatch losses: the loss of ONE minibatch -- what
    #  train_loop returns -- moves by more between two batches than three epochs of SGD gain; 1 run in ~25 used to fail on that)
    dl = DataLoader(Triples(x, y), batch_size=32, shuffle=True, generator=torch.Generator().manual_seed(0))
    bce, mse = torch.nn.BCEWithLogitsLoss(reduction="none"), torch.nn.MSELoss(reduction="none")
    logs = []
    per_epoch = []
    for ep in range(3):
        n0 = len(logs)
        m = train_loop(dataloader=dl, groove_transformer=model, encoder_only=1, opt=opt, epoch=ep, loss_fn=calculate_loss, bce_fn=bce,
                       mse_fn=mse, device="cuda", test_inputs=x[:16], test_gt=y[:16], hit_loss_penalty=0.38, save=(ep == 2),
                       save_dir=str(tmp_path), run_id="abc", log_every=2, on_log=logs.append)
        ls = [r["train/loss"] for r in logs[n0:] if "train/loss" in r]
        per_epoch.append(sum(ls) / len(ls))
    assert per_epoch[2] < per_epoch[0], (per_epoch, m, [r.get("train/loss") for r in logs if "train/loss" in r])
    assert any("test/loss" in r for r in logs) and any("train/loss" in r for r in logs)
    assert model.engine.state_struct().step == 3 * 8                                 # one fused update per batch
    ck = tmp_path / "transformer_run_abc_Epoch_2.Model"
    assert ck.exists()
    payload = torch.load(ck, weights_only=True)
    assert set(payload) == {"epoch", "model_state_dict", "optimizer_state_dict", "loss"} and payload["epoch"] == 2
    assert payload["optimizer_state_dict"]["param_groups"][0]["dropout_step"] == 3 * 8       # position of the dropout stream
    lm = {"location": "local", "dir": str(tmp_path), "file_pattern": "transformer_run_{}_Epoch_{}.Model", "run": "abc"}
    model2, opt2, ep1 = initialize_model(dict(p, load_model=lm))
    assert ep1 == 3
    assert model2.engine.state_struct().step == 3 * 8                                       # resumed, not replayed from step 0
    for (n, a), (_, b) in zip(model.state_dict().items(), model2.state_dict().items()):
        assert torch.equal(a, b), n
    h, v, o = model2.predict(x[:10].cuda())
    assert h.shape == v.shape == o.shape == (10, 32, 9) and set(h.unique().tolist()) <= {0.0, 1.0}
    assert torch.equal(torch.cat([h, v, o], -1), model2.predict_hvo(x[:10]))


def test_train_loop_exposes_gradients_every_watch_log_freq_batches():
    """ref:train.py:150 wandb.watch(model, log_freq=1000): the fused step consumes its gradients inside its last launch, so every
    model.watch_log_freq batches train_loop runs the step split (backward | gradients visible | update) and logs them under wandb.watch's
    names -- same parameters afterwards as the all-fused epoch, bit for bit (one owner per gradient tile on both paths)."""
    from BaseGrooveTransformers import calculate_loss, initialize_model, train_loop
    from torch.utils.data import DataLoader, TensorDataset
    import train as train_cli
    x, y = train_cli.synthetic_tensors(128, 16, 0)

    class Triples(TensorDataset):
        def __getitem__(self, i):
            return self.tensors[0][i], self.tensors[1][i], i

    bce, mse = torch.nn.BCEWithLogitsLoss(reduction="none"), torch.nn.MSELoss(reduction="none")
    finals = []
    for freq in (0, 3):
        torch.manual_seed(5)                                                         # (the weights are drawn from torch's generator)
        model, opt, _ = initialize_model(_params(d=128, H=4, F=64, L=2, dropout=0.1, lr=0.05, pen=0.38))
        model.watch_log_freq = freq
        dl = DataLoader(Triples(x, y), batch_size=16, shuffle=False)
        train_loop(dataloader=dl, groove_transformer=model, encoder_only=1, opt=opt, epoch=0, loss_fn=calculate_loss, bce_fn=bce, mse_fn=mse,
                   device="cuda", hit_loss_penalty=0.38, log_every=4)
        torch.cuda.synchronize()
        finals.append(model.engine.params.clone())
        if freq:
            rec = model.last_watch                                                   # batches 3 and 6 of 8 took the split step
            assert set(rec) == {"gradients/" + n for n, _ in model.named_parameters()}
            g = rec["gradients/Encoder.Encoder.layers.0.linear1.weight"]
            assert g.shape == (64, 128) and np.isfinite(g).all() and np.abs(g).max() > 0
            assert model.engine.state_struct().step == 8
    assert torch.equal(finals[0], finals[1])


def test_adam_paths_agree_with_oracle():
    from BaseGrooveTransformers import calculate_loss, initialize_model
    p = _params(algo="adam", lr=1e-3)
    cfg = _cfg(p)
    model, opt, _ = initialize_model(p)
    P = ng.init_params(cfg, seed=4, perturb=0.05)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()}, strict=False)
    x, y = ng.synthetic_batch(4, 16, seed=5)
    xt, yt = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    bce, mse = torch.nn.BCEWithLogitsLoss(reduction="none"), torch.nn.MSELoss(reduction="none")
    cur = {k: v.astype(np.float64) for k, v in P.items()}
    m = {k: np.zeros_like(v) for k, v in cur.items()}
    vv = {k: np.zeros_like(v) for k, v in cur.items()}
    model.eval()                                                    # no dropout; gradients still flow
    for t in (1, 2, 3):
        opt.zero_grad()
        out = calculate_loss(model(xt), yt, bce, mse, 1.0)
        out[0].backward()
        opt.step()
        (h, v, o), C = ng.forward(cur, cfg, x, dtype=np.float64)
        _, dpred = ng.calculate_loss((h, v, o), y.astype(np.float64), 1.0)
        G = ng.backward(cur, cfg, C, dpred, dtype=np.float64)
        cur, m, vv = ng.adam_step(cur, G, m, vv, t, 1e-3)
    torch.cuda.synchronize()
    for n, prm in model.named_parameters():
        live = np.abs(G[n]) > 1e-6
        assert np.abs(prm.detach().cpu().numpy() - cur[n])[live].max(initial=0) < 3e-5, n
    sd = opt.state_dict()
    assert set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"} and float(sd["state"][0]["step"]) == 3


def test_train_py_cli_runs_a_reference_yaml(tmp_path):
    import train as train_cli
    cfg = dict(experiment="InfillingClosedHH_testing", batch_size=32, d_model=32, dim_feedforward=16, dropout=0.18, optimizer_algorithm="sgd",
               learning_rate=0.094, n_heads=4, num_encoder_decoder_layers=6, epochs=1, encoder_only=1, hit_loss_penalty=0.47, load_model=None)
    f = tmp_path / "InfillingClosedHH_testing_training.yaml"          # values of the reference's smoke config
    f.write_text(yaml.safe_dump(cfg))
    model = train_cli.main(["--config", str(f), "--synthetic", "128", "--wandb", "False", "--save-dir", str(tmp_path)])
    assert (tmp_path / "transformer_run_local_Epoch_0.Model").exists()
    assert sum(p.numel() for p in model.parameters()) == 34043         # BASELINE.md C1 parameter count


def test_train_py_cli_evaluation_leg(tmp_path):
    """ref:train.py:219-250 / ref:evaluator.py:516-525: on the epochs of the save schedule every enabled set is predicted and scored per
    voice on the device (metrics.evaluate), the flags select the sets, the scalars are dumped; train_loop also gets the test / validation
    tensors for its end-of-epoch losses."""
    import json
    import train as train_cli
    from transformergrooveinfilling_amd import metrics
    cfg = dict(experiment="InfillingClosedHH_testing", batch_size=32, d_model=32, dim_feedforward=16, dropout=0.18, optimizer_algorithm="sgd",
               learning_rate=0.094, n_heads=4, num_encoder_decoder_layers=2, epochs=2, encoder_only=1, hit_loss_penalty=0.47, load_model=None)
    f = tmp_path / "cfg.yaml"
    f.write_text(yaml.safe_dump(cfg))
    xe, ye = train_cli.synthetic_tensors(96, 16, 77)
    npz = tmp_path / "eval.npz"
    np.savez(npz, test_inputs=xe.numpy(), test_gt=ye.numpy(), validation_inputs=xe[:40].numpy(), validation_gt=ye[:40].numpy())
    model = train_cli.main(["--config", str(f), "--synthetic", "128", "--wandb", "False", "--save-dir", str(tmp_path), "--eval-npz", str(npz),
                            "--eval_train", "False", "--eval_test", "True", "--eval_validation", "True", "--eval-size", "64"])
    recs = model.eval_log
    assert [r["epoch"] for r in recs] == [0, 0, 1, 1]                                       # two enabled sets x two epochs (both on the schedule)
    assert all(any(k.startswith("Test_Set/") for k in r) or any(k.startswith("Validation_Set/") for k in r) for r in recs)
    assert not any(k.startswith("Train_Set/") for r in recs for k in r)                     # --eval_train False is honoured
    last = [r for r in recs if r["epoch"] == 1 and "Test_Set/Hits_Accuracy_Overall" in r][0]
    want = metrics.evaluate(model, xe, ye)                                                  # the same numbers, recomputed on the final weights
    assert abs(last["Test_Set/Hits_Accuracy_Overall"] - want["Hits_Accuracy_Overall"]) < 1e-6
    assert abs(last["Test_Set/Velocity_MSE_KICK"] - want["Velocity_MSE_KICK"]) < 1e-6
    dumped = json.load(open(tmp_path / "eval_Test_Set_Epoch_1.json"))
    assert dumped["epoch"] == 1 and abs(dumped["Offset_MSE_Overall"] - want["Offset_MSE_Overall"]) < 1e-6
    # against numpy on the host, from the predictions themselves
    pred = model.predict_hvo(xe).cpu().numpy()
    assert abs(want["Hits_Accuracy_Overall"] - float((pred[..., :9] == ye.numpy()[..., :9]).mean())) < 1e-6


def test_engine_bucketed_data_parallel_sequence_matches_fused_step():
    """StepEngine's data-parallel sequence (graph A, async all-reduce of bucket 0 under graph B, all-reduce of bucket 1,
    update) on a 1-rank process group must train exactly like the fused single-GPU step."""
    import socket
    import torch.distributed as dist
    from transformergrooveinfilling_amd.engine import StepEngine
    # (d_model 256: one kernel per op -- the sequence-resident kernels of smaller models run the backward as ONE launch, one bucket)
    dims = dict(d_model=256, n_heads=4, dim_feedforward=128, num_encoder_layers=3, num_decoder_layers=0, dropout=0.2, embedding_size_src=16)
    x, y = ng.synthetic_batch(8, 16, seed=4)
    outs = []
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
    try:
        for force_dp in (False, True):
            eng = StepEngine(batch_size=8, optimizer="sgd", learning_rate=0.05, hit_loss_penalty=0.5, seed=3, **dims)
            eng.force_dp = force_dp
            eng.overlap_allreduce = True                        # opt-in (GT_DP_OVERLAP=1)
            eng.load_named(ng.init_params(dims, seed=1))
            eng.x.copy_(torch.from_numpy(x)); eng.y.copy_(torch.from_numpy(y))
            if force_dp:
                assert len(eng.lib.grad_buckets(eng.slot(8).cfg)) == 2
            for _ in range(4):
                stats = eng.train_step()
            torch.cuda.synchronize()
            outs.append((eng.params.cpu().numpy().copy(), stats.cpu().numpy().copy(), eng.grads.abs().max().item()))
    finally:
        dist.destroy_process_group()
    assert outs[1][2] == 0.0                                    # the update re-zeroed the gradient buffer
    assert np.abs(outs[0][0] - outs[1][0]).max() < 2e-6 * np.abs(outs[0][0]).max()
    assert np.abs(outs[0][1] - outs[1][1]).max() < 1e-5


def test_indexed_train_step_and_voice_metrics():
    """SURVEY 8f N3 / N4 on the device: (a) a step fed by INDICES into an HBM-resident dataset (gather = first launch of the step's
    graph) equals the step fed the gathered tensors; DeviceBatchLoader's index mode drives train_loop; (b) per-voice metrics
    (gt_voice_metrics) equal a numpy evaluation of ref:evaluator.py:522-525's three quantities."""
    import numpy as np
    from transformergrooveinfilling_amd import layout, metrics
    from transformergrooveinfilling_amd.engine import StepEngine
    from transformergrooveinfilling_amd.parallel import DeviceBatchLoader
    dims = dict(d_model=64, n_heads=4, dim_feedforward=64, num_encoder_layers=2, num_decoder_layers=0, dropout=0.1, embedding_size_src=16)
    x, y = layout.synthetic_batch(200, 16, seed=3)
    xs, ys = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    a = StepEngine(batch_size=16, learning_rate=0.05, hit_loss_penalty=0.4, seed=7, **dims)
    b = StepEngine(batch_size=16, learning_rate=0.05, hit_loss_penalty=0.4, seed=7, **dims)
    P = layout.init_params(dims, seed=1)
    a.load_named(P); b.load_named(P)
    g = torch.Generator().manual_seed(0)
    for _ in range(4):
        idx = torch.randperm(200, generator=g)[:16].cuda()
        sa = a.train_step_indexed(xs, ys, idx).clone()
        sb = b.train_step(xs[idx], ys[idx]).clone()
        assert torch.allclose(sa, sb, rtol=2e-5, atol=1e-6)        # (weight gradients are fp32 atomics: later steps differ in the last bits)
    assert torch.allclose(a.params, b.params, rtol=1e-4, atol=2e-6)
    # DeviceBatchLoader in index mode through train_loop == the same loader iterated as (x, y, idx) tensors
    from BaseGrooveTransformers import calculate_loss, initialize_model, train_loop
    bce, mse = torch.nn.BCEWithLogitsLoss(reduction="none"), torch.nn.MSELoss(reduction="none")
    res = []
    for mode in ("index", "tensors"):
        model, opt, _ = initialize_model(_params(d=64, H=4, F=64, L=2, dropout=0.1, lr=0.05, pen=0.38))
        model.engine.load_named(P)
        dl = DeviceBatchLoader(xs, ys, 32, model.engine.device, seed=5)
        src = dl if mode == "index" else list(dl)
        train_loop(dataloader=src, groove_transformer=model, encoder_only=1, opt=opt, epoch=0, loss_fn=calculate_loss, bce_fn=bce,
                   mse_fn=mse, device="cuda", hit_loss_penalty=0.38)
        res.append(model.engine.params.clone())
    assert torch.allclose(res[0], res[1], rtol=1e-4, atol=2e-6)
    # per-voice metrics
    pred = model.predict_hvo(xs)
    m = metrics.voice_metrics(model, pred, ys)
    p, t = pred.cpu().numpy().reshape(-1, 27), y.reshape(-1, 27)
    assert abs(m["Hits_Accuracy_Overall"] - (p[:, :9] == t[:, :9]).mean()) < 1e-6
    for j, v in enumerate(metrics.VOICES):
        assert abs(m["Hits_Accuracy_" + v] - (p[:, j] == t[:, j]).mean()) < 1e-6
        assert abs(m["Velocity_MSE_" + v] - ((p[:, 9 + j] - t[:, 9 + j]) ** 2).mean()) < 1e-6
        assert abs(m["Offset_MSE_" + v] - ((p[:, 18 + j] - t[:, 18 + j]) ** 2).mean()) < 1e-6
    assert abs(m["Velocity_MSE_Overall"] - ((p[:, 9:18] - t[:, 9:18]) ** 2).mean()) < 1e-6
    assert metrics.evaluate(model, xs, ys) == m                                             # bitwise reproducible


def test_model_predict_use_pd_on_both_model_kinds():
    """model.predict(src, use_pd=True) (the reference's sampling mode): reproducible for a seed, different across seeds, hits in {0,1};
    no NotImplementedError for the encoder-decoder any more"""
    from BaseGrooveTransformers import initialize_model
    for enc_only in (True, False):
        model, _, _ = initialize_model(_params(enc_only=enc_only, d=32, H=4, F=16, L=2))
        x, _ = ng.synthetic_batch(6, 16, seed=3)
        xt = torch.from_numpy(x).cuda()
        h1, v1, o1 = model.predict(xt, use_pd=True, pd_seed=11)
        h2, _, _ = model.predict(xt, use_pd=True, pd_seed=11)
        h3, _, _ = model.predict(xt, use_pd=True, pd_seed=12)
        assert torch.equal(h1, h2) and not torch.equal(h1, h3)
        assert set(h1.unique().tolist()) <= {0.0, 1.0} and h1.shape == (6, 32, 9) and v1.shape == o1.shape == (6, 32, 9)
        model.predict(xt, use_pd=True)                                           # seed drawn from torch's generator
        # the samples are hashed from the element's index in the whole set: any chunking draws the same ones (gt_predict_pd_at)
        whole = model.engine.predict(xt, pd_seed=11, chunk=6)
        for chunk in (1, 4):
            assert torch.equal(model.engine.predict(xt, pd_seed=11, chunk=chunk), whole)


def test_predict_walks_large_sets_in_chunks():
    """engine.predict (ref:evaluator.py:173 hands over the whole evaluation set): any chunking gives the same HVO tensor, and the
    default chunk grows with the set while the workspace fits (greedy decoding is launch-bound per call)."""
    from transformergrooveinfilling_amd import layout
    from transformergrooveinfilling_amd.engine import StepEngine, PREDICT_CHUNK
    for Ld in (0, 1):
        dims = dict(d_model=32, n_heads=4, dim_feedforward=64, num_encoder_layers=2, num_decoder_layers=Ld, dropout=0.1, embedding_size_src=16)
        eng = StepEngine(batch_size=4, seed=2, **dims)
        eng.load_named(layout.init_params(dims, seed=5))
        x, _ = layout.synthetic_batch(70, 16, seed=9)
        whole = eng.predict(x, chunk=128).cpu()
        parts = eng.predict(x, chunk=32).cpu()                      # 32 + 32 + 6
        assert torch.equal(whole, parts)
        assert torch.equal(whole, eng.predict(x).cpu())
        assert eng.predict_chunk(70) == PREDICT_CHUNK and eng.predict_chunk(4096) == 4096


def test_quad_pair_exchange_fails_safe_under_contention_and_on_timeout():
    """VERDICT r04 #4d / ADVICE r04: a four-workgroups-per-sequence (QUAD) step while another stream holds CUs with an LDS-heavy kernel
    must give the right numbers; and when an exchange does time out (forced: a polling bound of ONE) the step must not be applied --
    parameters untouched, the engine notices without synchronising on the step path, falls back to two workgroups per sequence and
    trains on -- never silent garbage."""
    import ctypes
    import warnings
    from transformergrooveinfilling_amd import engine as E
    from transformergrooveinfilling_amd.engine import StepEngine
    dims = dict(d_model=128, n_heads=4, dim_feedforward=512, num_encoder_layers=2, num_decoder_layers=0, dropout=0.24, embedding_size_src=16)
    B = 64                                                      # 4 x 64 = 256 workgroups: the whole chip
    x, y = ng.synthetic_batch(B, 16, seed=4)

    def make():
        e = StepEngine(batch_size=B, optimizer="sgd", learning_rate=0.05, hit_loss_penalty=0.5, seed=3, **dims)
        e.load_named(ng.init_params(dims, seed=1))
        e.x.copy_(torch.from_numpy(x)); e.y.copy_(torch.from_numpy(y))
        return e

    lib = make().lib
    try:
        lib.cdll.gt_set_seq_quad(-1)
        ref = make()
        for _ in range(3):
            ref.train_step()
        torch.cuda.synchronize()
        # (1) 96 CUs held for 30 ms by another stream while three QUAD steps are enqueued: partners are adjacent blocks, dispatched
        #     together, so the exchange only ever waits for a dispatch slot -- same numbers, bit for bit, no time-out
        eng = make()
        side = torch.cuda.Stream()
        lib.call("gt_debug_occupy_cus", 96, 30000, ctypes.c_void_p(side.cuda_stream))
        for _ in range(3):
            eng.train_step()
        torch.cuda.synchronize()
        assert not eng.check_exchange(eng.slot(B)) and eng.exchange_timeouts == 0
        assert torch.equal(eng.params, ref.params)
        # (2) a polling bound of one: exchanges give up at once.  The update is skipped on the device ...
        lib.cdll.gt_set_xchg_spin_max(1)
        before = eng.params.clone()
        eng.train_step()
        torch.cuda.synchronize()
        assert torch.equal(eng.params, before) and float(eng.grads.abs().max()) == 0.0
        # ... the step path notices by itself (asynchronous 4-byte copy every XCHG_POLL_EVERY steps, looked at one poll later) ...
        with warnings.catch_warnings(record=True) as rec:
            warnings.simplefilter("always")
            for _ in range(40 * E.XCHG_POLL_EVERY):     # (the host runs a hundred steps ahead of the device: the copy it looks at is an old one)
                eng.train_step()
                if eng.exchange_timeouts:
                    break
            torch.cuda.synchronize()
        assert eng.exchange_timeouts == 1 and any("pair exchange" in str(w.message) for w in rec)
        assert torch.equal(eng.params, before)                  # every step in between was a skipped one
        assert eng.skipped_updates >= 2 and eng.cfg_flags == eng.FALLBACK_FLAGS      # ... counted by the device; the fall-back is this engine's own
        assert ref.cfg_flags == 0 and lib.cdll.gt_step_launches(ctypes.byref(ref.slot(B).cfg)) != lib.cdll.gt_step_launches(ctypes.byref(eng.slot(B).cfg))
        # ... and from there on it trains on two workgroups per sequence, with the numbers of an undisturbed engine (fp32 rounding)
        lib.cdll.gt_set_xchg_spin_max(0)
        st = eng.state_struct()
        eng.set_state(step=ref.state_struct().step, opt_step=ref.state_struct().opt_step)
        ref.train_step(); eng.train_step()
        torch.cuda.synchronize()
        assert not eng.check_exchange(eng.slot(B))
        assert (eng.params - ref.params).abs().max() < 2e-6 * ref.params.abs().max()
        # predict through a raised word (its own workspace): noticed when the call ends, repeated on the fallback schedule
        lib.cdll.gt_set_seq_quad(-1)
        e2 = make()
        want = e2.predict(torch.from_numpy(x)).clone()
        cfg2, ws2, _ = e2._predict_ws[B]
        e2._xchg_word(ws2, cfg2)[0] = 1
        with warnings.catch_warnings(record=True):
            warnings.simplefilter("always")
            got = e2.predict(torch.from_numpy(x))
        assert e2.exchange_timeouts == 1 and int(e2._xchg_word(ws2, cfg2)[0].item()) == 0
        assert torch.equal(got[..., :9], want[..., :9]) and (got - want).abs().max() < 1e-5
    finally:
        lib.cdll.gt_set_xchg_spin_max(0)
        lib.cdll.gt_set_seq_quad(-1)


@pytest.mark.parametrize("algo", ["sgd", "adam"])
def test_module_api_optimizer_honours_the_exchange_error_word(algo):
    """loss.backward(); opt.step() through a raised error word must change NOTHING -- the package's optimizers hand the update
    kernel the workspace of the last backward (gt_optimizer_step_ws): parameters and moments stay, Adam's t does not advance, the device counts
    the skipped update; after the region is cleared the same step applies."""
    from BaseGrooveTransformers import calculate_loss, initialize_model
    p = _params(algo=algo, d=128, H=4, F=64, L=1, dropout=0.1, lr=0.01)
    model, opt, _ = initialize_model(p)
    eng = model.engine
    x, y = ng.synthetic_batch(8, 16, seed=3)
    xt, yt = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    bce, mse = torch.nn.BCEWithLogitsLoss(reduction="none"), torch.nn.MSELoss(reduction="none")
    model.train()

    def step():
        opt.zero_grad()
        calculate_loss(model(xt), yt, bce, mse, 0.47)[0].backward()
        opt.step()
        torch.cuda.synchronize()
    step()
    s = eng._bwd_slot
    assert s is not None and eng._xchg_word(s) is not None and eng._fused_opt      # d_model 128: the four-workgroups-per-sequence schedule, its region
    before, t0 = eng.params.clone(), eng.state_struct().opt_step
    m0 = None if eng.m is None else eng.m.clone()
    eng._xchg_word(s)[0] = 1
    step()
    assert torch.equal(eng.params, before) and eng.state_struct().opt_step == t0
    if m0 is not None:
        assert torch.equal(eng.m, m0)
    assert eng.exchange_report(s)["skipped_updates"] == 1
    import ctypes
    eng.lib.call("gt_workspace_init", ctypes.byref(s.cfg), ctypes.c_void_p(s.ws.data_ptr()), eng.stream)
    s.xchg_skipped_seen = 0
    step()
    assert not torch.equal(eng.params, before) and eng.state_struct().opt_step == t0 + 1


def test_encoder_decoder_graph_replays_separated_by_host_syncs_stay_finite():
    """Round 5 regression: the encoder-decoder step zeroed the memory gradient with a hipMemsetAsync -- a memset node inside the captured
    step graph -- and went non-finite intermittently (2 of 3 processes, from the second or third step) when the replays were separated by a
    host synchronisation; never eagerly, never without the decoder.  The memset node is gone (the first cross-attention k / v dgrad stores,
    the others add): replayed with a synchronise after every step the model must train exactly like the eager engine."""
    from transformergrooveinfilling_amd.engine import StepEngine
    dims = dict(d_model=256, n_heads=2, dim_feedforward=512, num_encoder_layers=2, num_decoder_layers=2, dropout=0.3, embedding_size_src=16)
    x, y = ng.synthetic_batch(64, 16, seed=2)
    out = []
    for graph in (True, False):
        eng = StepEngine(batch_size=64, optimizer="sgd", learning_rate=0.05, hit_loss_penalty=0.5, seed=1, use_graph=graph, **dims)
        eng.load_named(ng.init_params(dims, seed=0))
        eng.x.copy_(torch.from_numpy(x)); eng.y.copy_(torch.from_numpy(y))
        for _ in range(10):
            eng.train_step()
            torch.cuda.synchronize()
            assert bool(torch.isfinite(eng.params).all())
        out.append((eng.params.clone(), float(eng.stats[0])))
    assert abs(out[0][1] - out[1][1]) < 1e-3 * abs(out[1][1])
    assert (out[0][0] - out[1][0]).abs().max() < 1e-3 * out[1][0].abs().max()      # (fp32 atomics in the weight gradients: last bits differ run to run)
