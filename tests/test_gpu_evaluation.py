"""GPU: the SURVEY §8(f) rows -- test.py inference semantics, on-device per-row AUC / top-1, validation.
The model_test restatement in the oracle is 'parity unpinned' (test.py is not importable here, see its docstring);
the AUC kernel is checked against the oracle's AUC, which IS pinned to sklearn by make_golden.py."""
import numpy as np
import pytest
import torch

from golden_util import load_case, rel_err
from oracle import user_model_oracle as orc

pytestmark = pytest.mark.gpu


def test_row_auc_kernel_matches_oracle_with_ties_and_padding(lib):
    from news_recommendation_model_amd import evaluation
    rng = np.random.default_rng(0)
    B, T = 37, 21
    score = rng.standard_normal((B, T)).astype(np.float32)
    score[:, 3] = score[:, 5]                       # ties
    score[4] = 0.25                                 # all tied -> 0.5
    label = np.zeros((B, T), dtype=np.float32)
    label[np.arange(B), rng.integers(0, 9, B)] = 1
    live = rng.integers(9, T + 1, B)
    auc, top1 = evaluation.row_auc_top1(torch.from_numpy(score).cuda(), torch.from_numpy(label).cuda(),
                                        torch.from_numpy(live).cuda())
    auc, top1 = auc.cpu().numpy(), top1.cpu().numpy()
    for b in range(B):
        n = live[b]
        assert abs(auc[b] - orc.row_auc(label[b, :n], score[b, :n])) < 1e-6
        assert top1[b] == int(np.argmax(score[b, :n]) == np.argmax(label[b, :n]))
    # a row with one class only: sklearn raises; the kernel marks it with -1
    label[0] = 0
    auc, _ = evaluation.row_auc_top1(torch.from_numpy(score).cuda(), torch.from_numpy(label).cuda())
    assert float(auc[0]) == -1.0


def test_predict_follows_test_py_semantics(lib):
    """Two-model ensemble on a batch with per-row candidate padding: trimming, averaged softmax, second softmax."""
    from news_recommendation_model_amd import evaluation, synth, trainer
    case, dims, batch, sd, fx = load_case("tiny_pad")
    sd2 = synth.make_state_dict(dims, seed=5, user_num=int(batch["user_num"]))
    # rows get different amounts of padding: 2 (common, trimmed), 3, 2, 3
    batch["empty_num"][:] = [2, 3, 2, 3]
    batch["label"][:] = 0
    batch["label"][:, 0] = 1                          # the clicked candidate must survive every row's padding
    for b, z in enumerate(batch["empty_num"]):
        batch["x_target"][b, case["T"] - z:] = 0
        batch["x_global"][b, case["T"] - z:] = 0
    models = [trainer.build_model(dims, int(batch["user_num"]), s, device="cuda") for s in (sd, sd2)]
    tb = trainer.batch_to_device(batch, "cuda")
    scores, live = evaluation.predict(models, tb)
    tbc = {k: torch.from_numpy(v) for k, v in batch.items() if isinstance(v, np.ndarray) and v.ndim > 0}
    ref = orc.model_test_scores([orc.to_torch_params(s, False) for s in (sd, sd2)], tbc)
    assert scores.shape[1] == case["T"] - 2
    for b, r in enumerate(ref):
        assert int(live[b]) == len(r)
        assert rel_err(scores[b, :len(r)].cpu().numpy(), r) < 1e-3
    assert evaluation.rank_row([0.1, 0.7, 0.2]) == [3, 1, 2]
    # the same call captured into a HIP graph (one per input shape and trim), fed from HOST tensors as a DataLoader would
    graphed = evaluation.GraphedPredict(models)
    host = {k: torch.from_numpy(np.ascontiguousarray(batch[k])) for k in ("x_history", "x_target", "x_global", "empty_num")}
    for rep in range(3):                                # capture, then two replays (the second with other data in between)
        s_g, live_g = graphed(host)
        assert torch.equal(live_g, live) and torch.allclose(s_g, scores, rtol=1e-5, atol=1e-7)
        if rep == 1:
            other = dict(host, x_history=host["x_history"].flip(0).contiguous())
            s_o, _ = graphed(other)
            assert not torch.allclose(s_o, scores, rtol=1e-5, atol=1e-7)
    less = dict(host, empty_num=host["empty_num"] - 1)  # another trim: another graph, one more candidate column
    s_l, live_l = graphed(less)
    assert s_l.shape[1] == scores.shape[1] + 1 and len(graphed.graphs) == 2
    # validation numbers (verify.py:19-43) against the oracle's AUC on the same scores
    auc_v, tpr_v = evaluation.validate(models, [tb])
    want_auc = np.mean([orc.row_auc(batch["label"][b, :len(r)], r) for b, r in enumerate(ref)])
    want_tpr = np.mean([float(np.argmax(r) == np.argmax(batch["label"][b])) for b, r in enumerate(ref)])
    assert abs(auc_v - want_auc) < 1e-5 and abs(tpr_v - want_tpr) < 1e-9


@pytest.mark.parametrize("name", ["refdefault_eval", "c2_small_eval", "c3_large_eval"])
def test_eval_forward_at_baseline_dims_matches_reference_fixture(lib, name):
    """The part of inference that is pinned: model.eval() forward of the imported REFERENCE at BASELINE
    dimensions on candidate lists with trailing all-zero padding (oracle/make_golden.py EVAL_CASES: logits `r` and
    nn.Softmax(dim=1)(r), reference test.py:44,61).  (1) the Module's eval forward, (2) evaluation.predict with one model and
    no declared padding -- exactly test.py:61's `softmax(model(...))` -- and (3) predict with the padding declared: its
    pre-wrapper logits are the reference's on the live columns (a candidate's logit does not depend on the other candidates in
    eval mode).  The trim / second softmax wrapped around them (test.py:48-56,:66-70) stays parity UNPINNED: checked here only
    against numpy arithmetic on the reference's logits."""
    from news_recommendation_model_amd import evaluation, trainer
    case, dims, batch, sd, fx = load_case(name)
    model = trainer.build_model(dims, int(batch["user_num"]), sd, device="cuda").eval()
    tb = trainer.batch_to_device(batch, "cuda")
    with torch.no_grad():
        r = model(tb["x_history"], tb["x_target"], tb["x_global"])
    assert rel_err(r.cpu().numpy(), fx["r"]) < 1e-3
    assert np.abs(torch.softmax(r, dim=1).cpu().numpy() - fx["softmax_r"]).max() < 1e-3 * fx["softmax_r"].max()
    # predict(), padding not declared: the reference's softmax(model(x)) line, one model
    nopad = dict(tb, empty_num=torch.zeros(case["B"], dtype=torch.int64))
    scores, live = evaluation.predict([model], nopad)
    assert scores.shape == fx["softmax_r"].shape and bool((live == case["T"]).all())
    assert np.abs(scores.cpu().numpy() - fx["softmax_r"]).max() < 1e-3 * fx["softmax_r"].max()
    # predict(), padding declared (common to all rows -> trimmed before the forward): logits of the live columns
    k = case["pad_target"]
    scores_t, live_t = evaluation.predict([model], tb)
    assert scores_t.shape[1] == case["T"] - k and bool((live_t == case["T"] - k).all())
    live_logits = fx["r"][:, :case["T"] - k].astype(np.float64)
    e = np.exp(live_logits - live_logits.max(1, keepdims=True))
    want = e / e.sum(1, keepdims=True)
    assert np.abs(scores_t.cpu().numpy() - want).max() < 1e-3 * want.max()
    # and in the arithmetic BASELINE config 2 names (bf16 matrix cores on hi/lo split operands) where the case is that config
    if name == "c2_small_eval":
        m2 = trainer.build_model(dims, int(batch["user_num"]), sd, device="cuda", attention_mma="bf16x3").eval()
        with torch.no_grad():
            r2 = m2(tb["x_history"], tb["x_target"], tb["x_global"])
        assert rel_err(r2.cpu().numpy(), fx["r"]) < 1e-3


def test_graphed_predict_follows_weight_changes(lib, tmp_path):
    """a captured predict() reads the packed weight images that ops._pack filled BEFORE the capture.  Weights that
    change afterwards -- load_state_dict of another checkpoint, a torch.optim step, FlatAdam (which re-seats the parameters), a
    dropped pack cache -- must show up in the next replay: graphed == eager after every one of them, and != the scores before."""
    from news_recommendation_model_amd import evaluation, ops, synth, trainer
    case, dims, batch, sd, fx = load_case("tiny_train")
    batch["empty_num"] = np.zeros(case["B"], dtype=np.int64)
    user_num = int(batch["user_num"])
    model = trainer.build_model(dims, user_num, sd, device="cuda").eval()
    tb = trainer.batch_to_device(batch, "cuda")
    tb["empty_num"] = torch.from_numpy(batch["empty_num"])
    graphed = evaluation.GraphedPredict([model])

    def both():
        g, _ = graphed(tb)
        g = g.clone()
        e, _ = evaluation.predict([model], tb)
        assert torch.allclose(g, e, rtol=1e-5, atol=1e-7), float((g - e).abs().max())
        return g

    s0 = both()
    assert torch.equal(both(), s0) and len(graphed.graphs) == 1
    # (1) other weights through load_state_dict (in place: same addresses, version counters move) -> re-pack, same graph
    sd2 = synth.make_state_dict(dims, seed=9, user_num=user_num)
    model.load_state_dict({k: torch.as_tensor(v) for k, v in sd2.items()}, strict=False)
    graph_before = next(iter(graphed.graphs.values()))[0]
    s1 = both()
    assert not torch.allclose(s1, s0, rtol=1e-3, atol=1e-6)
    assert next(iter(graphed.graphs.values()))[0] is graph_before          # replayed, not re-captured
    # (2) a torch.optim step
    model.train()
    opt = trainer.make_optimizer(model, lr=5e-2)
    out = model(tb["x_history"], tb["x_target"], tb["x_global"])
    model.loss(tb["user_id"], out, tb["label"]).backward()
    opt.step()
    opt.zero_grad()
    model.eval()
    s2 = both()
    assert not torch.allclose(s2, s1, rtol=1e-3, atol=1e-6)
    # (3) the checkpoint loader of evaluation.py (test.py:160)
    path = str(tmp_path / "other.pth")
    ref_model = trainer.build_model(dims, user_num, sd, device="cuda")
    evaluation.save_checkpoint(ref_model, path)
    evaluation.load_checkpoint(model, path)
    s3 = both()
    assert torch.allclose(s3, s0, rtol=1e-5, atol=1e-7)                     # back on the first weights (delta is not read by forward)
    # (4) the pack cache dropped (its buffers may be reused by the allocator): the graph must be re-captured, not replayed
    ops.invalidate_packed_weights()
    junk = [torch.randn(1 << 16, device="cuda") for _ in range(8)]          # noqa: F841  (take the freed blocks)
    s4 = both()
    assert torch.allclose(s4, s0, rtol=1e-5, atol=1e-7)
    # (5) FlatAdam re-seats every parameter into its flat buffer (new addresses) and updates through a raw pointer
    model.train()
    fopt = trainer.FlatAdam(model, lr=5e-2)
    trainer.train_step(model, fopt, tb)
    model.eval()
    s5 = both()
    assert not torch.allclose(s5, s0, rtol=1e-3, atol=1e-6)
    trainer.train_step(model.train(), fopt, tb)                             # same addresses now: FlatAdam refreshes the images itself
    model.eval()
    s6 = both()
    assert not torch.allclose(s6, s5, rtol=1e-3, atol=1e-6)


def test_checkpoint_roundtrip_drops_delta(lib, tmp_path):
    from news_recommendation_model_amd import evaluation, trainer
    case, dims, batch, sd, fx = load_case("tiny_train")
    model = trainer.build_model(dims, int(batch["user_num"]), sd, device="cuda")
    path = str(tmp_path / "ckpt_epoch_0.pth")
    evaluation.save_checkpoint(model, path)
    saved = torch.load(path, weights_only=True)
    assert "delta" not in saved and len(saved) == 37
    fresh = trainer.build_model(dims, 0, None, device="cuda")
    res = evaluation.load_checkpoint(fresh, path)
    assert res.missing_keys == ["delta"] and res.unexpected_keys == []
    assert torch.equal(fresh.gate.fc1.weight.cpu(), model.gate.fc1.weight.cpu())


def test_train_epochs_mirrors_the_reference_loop(lib, tmp_path):
    """trainer.train_epochs = train.py:52-100 (epoch loop, running loss / AUC averages, checkpoint without delta per
    epoch): the loss average falls on a small synthetic set, the AUC average rises above chance, checkpoints reload."""
    from news_recommendation_model_amd import config, evaluation, synth, trainer
    dims = config.Dims.for_emb(32, 60)
    B, H, T, user_num = 16, 6, 5, 40
    hosts = [synth.make_batch(dims, B, H, T, seed=500 + i, user_num=user_num) for i in range(4)]
    model = trainer.build_model(dims, user_num, synth.make_state_dict(dims, seed=9, user_num=user_num))
    opt = trainer.FlatAdam(model, lr=1e-3)                            # train.py's default lr
    seen = []
    hist = trainer.train_epochs(model, opt, lambda: iter(hosts), 8, ckpt_path=str(tmp_path / "ckpt_epoch_{epoch}.pth"),
                                on_batch=lambda e, i, loss, auc: seen.append((e, i)))
    assert [h["epoch"] for h in hist] == list(range(8)) and all(h["impressions"] == 4 * B for h in hist)
    assert seen[:5] == [(0, 0), (0, 1), (0, 2), (0, 3), (1, 0)]
    assert hist[-1]["loss_avg"] < 0.5 * hist[0]["loss_avg"]          # memorises 64 impressions (0.50 -> 0.18 measured)
    assert hist[-1]["auc_avg"] > 0.9 and hist[0]["auc_avg"] < 0.7
    assert abs(hist[0]["lr"] - 1e-3) < 1e-12
    sd = torch.load(tmp_path / "ckpt_epoch_7.pth", weights_only=True)
    assert "delta" not in sd and len(sd) == len(model.state_dict()) - 1
    fresh = trainer.build_model(dims, user_num)
    evaluation.load_checkpoint(fresh, tmp_path / "ckpt_epoch_7.pth")
    auc, hit = evaluation.validate([fresh], (trainer.batch_to_device(h, "cuda") for h in hosts))
    assert auc > 0.9


def test_train_epochs_raises_indexerror_for_an_out_of_range_id(lib):
    """The reference raises IndexError at the batch that holds an out-of-range category id (F.embedding); the kernels clamp
    and flag, the flag travels to pinned host memory asynchronously at the end of every train_step and the NEXT train_step
    raises before it enqueues anything (trainer.IndexErrorWatch): the loop stops one batch after the offending one, not at the
    end of the epoch, and without a host sync of its own."""
    from news_recommendation_model_amd import config, ops, synth, trainer
    dims = config.Dims.for_emb(32, 60)
    B, H, T, user_num = 8, 4, 3, 20
    hosts = [synth.make_batch(dims, B, H, T, seed=700 + i, user_num=user_num) for i in range(6)]
    cat_col = 4 + dims.pca_vector
    hosts[1]["x_history"][3, 2, cat_col] = dims.category_label_num + 5          # one bad id in the second batch
    model = trainer.build_model(dims, user_num, synth.make_state_dict(dims, seed=9, user_num=user_num))
    opt = trainer.FlatAdam(model)
    ops.check_index_errors("cuda")
    seen = []

    def on_batch(epoch, i, loss, auc):
        seen.append(i)
        torch.cuda.synchronize()              # the device keeps up with the host: the flag copy of batch i has landed

    with pytest.raises(IndexError):
        trainer.train_epochs(model, opt, lambda: iter(hosts), 1, on_batch=on_batch)
    assert seen == [0, 1]                     # batch 2's train_step raised before doing any work
    assert opt.steps == 2
    ops.check_index_errors("cuda")                                              # the flag is cleared by the raise
    # a host that runs ahead of the device still learns of it within max_lag + 1 steps
    seen.clear()
    with pytest.raises(IndexError):
        trainer.train_epochs(model, opt, lambda: iter(hosts), 1, on_batch=lambda e, i, l, a: seen.append(i))
    assert seen and seen[-1] <= 1 + 3
    ops.check_index_errors("cuda")
    # strict mode: the reference raises BEFORE any update -- ids validated on the device before the step is enqueued
    steps_before, w_before = opt.steps, opt.flat_param.clone()
    bad = trainer.batch_to_device(hosts[1], "cuda")
    with pytest.raises(IndexError, match="nothing was enqueued"):
        trainer.train_step(model, opt, bad, strict_ids=True)
    torch.cuda.synchronize()
    assert opt.steps == steps_before and torch.equal(opt.flat_param, w_before)
    bad_uid = trainer.batch_to_device(hosts[0], "cuda")
    bad_uid["user_id"] = bad_uid["user_id"].clone()
    bad_uid["user_id"][2] = user_num + 1                                        # delta has user_num + 1 entries
    with pytest.raises(IndexError, match="nothing was enqueued"):
        trainer.train_step(model, opt, bad_uid, strict_ids=True)
    trainer.train_step(model, opt, trainer.batch_to_device(hosts[0], "cuda"), strict_ids=True)     # a clean batch passes
    ops.check_index_errors("cuda")
    # ... and it says so instead of breaking a stream capture (its one host read of a device flag)
    clean = trainer.batch_to_device(hosts[0], "cuda")
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with pytest.raises(RuntimeError, match="cannot run inside a stream capture"):
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            trainer.validate_batch_ids(model, clean)
    torch.cuda.synchronize()
    # the asynchronous watch says what it could not prevent
    trainer.train_step(model, opt, bad)
    torch.cuda.synchronize()
    with pytest.raises(IndexError, match="ALREADY UPDATED"):
        trainer.train_step(model, opt, trainer.batch_to_device(hosts[0], "cuda"))
    ops.check_index_errors("cuda")
    hosts[1]["x_history"][3, 2, cat_col] = 1
    assert len(trainer.train_epochs(model, opt, lambda: iter(hosts), 1)) == 1


def test_full_size_inference_properties(lib):
    """The inference path of reference test.py:31-74 at BASELINE C3 dimensions (B=1024, H=50, T=30, D=400: too big for the CPU
    oracle as a whole): eval-mode BatchNorm makes impressions independent, so predict() of the batch must equal predict() of
    its parts bit for bit (same kernels, no atomics in the fp32 forward without the z store), per-row padding triggers the
    second softmax only on the rows that carry it, and two impressions match the oracle's model_test restatement."""
    from news_recommendation_model_amd import evaluation, synth, trainer
    from news_recommendation_model_amd.config import Dims
    B, H, T, D = 1024, 50, 30, 400
    dims = Dims.for_emb(D)
    user_num = 10 * B
    sd = synth.make_state_dict(dims, seed=1, user_num=user_num, perturb=True)
    model = trainer.build_model(dims, user_num, sd, device="cuda").eval()
    batch = synth.make_batch(dims, B, H, T, seed=3, user_num=user_num, dtype=np.float32)
    empty = np.zeros(B, dtype=np.int64)
    empty[5], empty[700] = 4, 9                                   # two rows with their own trailing padding
    for b in (5, 700):
        batch["x_target"][b, T - empty[b]:] = 0
        batch["x_global"][b, T - empty[b]:] = 0
    batch["empty_num"] = empty
    tb = trainer.batch_to_device(batch, "cuda")
    whole, live = evaluation.predict([model], tb)
    assert whole.shape == (B, T) and int(live[5]) == T - 4 and int(live[700]) == T - 9 and int(live[0]) == T
    part = lambda lo, hi: {k: (v[lo:hi] if hasattr(v, "shape") and v.ndim > 0 and v.shape[0] == B else v) for k, v in tb.items()}   # noqa: E731
    lo, _ = evaluation.predict([model], part(0, 512))
    hi, _ = evaluation.predict([model], part(512, B))
    assert torch.equal(whole[:512], lo) and torch.equal(whole[512:], hi)
    # every row is a distribution over its live candidates
    cols = torch.arange(T, device="cuda")[None, :]
    assert torch.allclose((whole * (cols < live[:, None])).sum(1), torch.ones(B, device="cuda"), atol=1e-5)
    # spot check against the oracle: one plain row, one padded row (second softmax)
    idx = [3, 700]
    tbc = {k: torch.from_numpy(np.asarray(v)[idx]) for k, v in batch.items() if isinstance(v, np.ndarray) and v.ndim > 0}
    ref = orc.model_test_scores([orc.to_torch_params(sd, False)], tbc)
    # (the oracle trims the padding COMMON to its two rows first; the scores of the live candidates do not depend on it)
    for j, b in enumerate(idx):
        n = int(live[b])
        assert len(ref[j]) == n
        assert rel_err(whole[b, :n].cpu().numpy(), ref[j]) < 1e-3


@pytest.mark.parametrize("graph", [False, True])
def test_bench_inference_mode_prints_the_contract_line(lib, graph):
    """`bench.py --mode infer [--graph]` (reference test.py:31-74: the OTHER caller of the hot path): one JSON line with the
    throughput, the forward kernel's roofline (timed on one-stream batches) and -- eager or as a captured graph -- the same
    fields."""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--mode", "infer", "--workload", "ref-test80", "--steps", "5", "--warmup", "2",
           "--no-cpu-baseline"] + (["--graph"] if graph else [])
    pr = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=420)
    assert pr.returncode == 0, pr.stderr.decode(errors="replace")[-3000:]
    lines = [ln for ln in pr.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["unit"] == "impressions/s" and line["value"] > 0 and line["n_gpus"] == 1 and line["higher_is_better"] is True
    assert line["config"]["mode"] == "infer" and line["config"]["batch"] == 80
    assert ("hipGraph" in line["config"]["launch"]) == graph
    r = line["roofline"]
    assert r["bound"] == "mfma" and 0 < r["frac"] < 1 and r["mean_launch_ms"] > 0 and "no z store" in r["kernel"]
