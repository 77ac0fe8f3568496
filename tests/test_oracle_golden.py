"""CPU: the oracle (oracle/user_model_oracle.py) reproduces every golden fixture generated from the
real reference (oracle/make_golden.py).  This is what keeps the oracle pinned on machines where the
reference itself is absent (the GPU box)."""
import numpy as np
import pytest
import torch

from golden_util import (II_B, II_W, MODEL_CASES, TRAIN_CASES, TRAJ_CASES, ZERO_GRAD_KEYS, assert_first_adam_update, check_trajectory,
                         grad_tolerance, instant_interest_grad_bounds, load_case, load_trajectory, pick, rel_err)
from oracle import user_model_oracle as orc


@pytest.mark.parametrize("name", MODEL_CASES)
def test_oracle_matches_fixture(name):
    case, dims, batch, sd, fx = load_case(name)
    tb = {k: torch.from_numpy(v) for k, v in batch.items() if isinstance(v, np.ndarray) and v.ndim > 0}
    full = case["full"]
    if case["mode"] == "eval":
        p = orc.to_torch_params(sd, requires_grad=False)
        with torch.no_grad():
            r, aux = orc.user_model_forward(p, tb["x_history"], tb["x_target"], tb["x_global"], training=False,
                                            return_aux=True)
            loss = orc.user_model_loss(p, tb["user_id"], r, tb["label"])
        assert rel_err(r.numpy(), fx["r"]) < 1e-5
        assert abs(float(loss) - float(fx["loss"])) < 1e-6
        assert rel_err(aux["eu_H"].numpy(), fx["eu_H"]) < 1e-5
        if "softmax_r" in fx.files:                # BASELINE-dims eval cases (round 5): test.py:61's softmax(model(x)) of the reference
            assert np.abs(torch.softmax(r, dim=1).numpy() - fx["softmax_r"]).max() < 1e-6
            assert rel_err(aux["ec"].numpy(), fx["ec"]) < 1e-5
        return
    p = orc.to_torch_params(sd)
    opt_state = {"step": 0, "m": {}, "v": {}}
    loss, r, grads = orc.train_step(p, opt_state, tb)
    assert rel_err(r.numpy(), fx["r"]) < 1e-5
    assert abs(float(loss) - float(fx["loss"])) < 1e-6
    gscale = max(float(fx["gradnorm/" + k]) for k in grads)
    for k, g in grads.items():
        ref = fx["grad/" + k]
        got = pick(g.numpy(), full)
        if k in ZERO_GRAD_KEYS:
            assert np.abs(got).max() < 1e-6 * max(1.0, gscale)
        else:
            assert np.abs(got - ref).max() <= 1e-2 * np.abs(ref).max() + 1e-9, k
    # Adam: the UPDATE (after - before) against the reference's, entry by entry (golden_util.assert_first_adam_update)
    checked = [assert_first_adam_update(k, sd[k], p[k].detach().numpy(), fx, full) for k in p if k not in orc.BUFFER_KEYS]
    assert max(checked) > 0.5                      # most entries of the dense tensors have a determined sign
    for k in ("bn.running_mean", "bn.running_var"):
        assert rel_err(pick(p[k].numpy(), full), fx["after/" + k]) < 1e-5, k
    np.testing.assert_allclose(orc.batch_auc(batch["label"], fx["r"]), fx["auc"], atol=1e-12)


@pytest.mark.parametrize("name", [c for c in TRAIN_CASES if c not in ("c5_long",)])
def test_instant_interest_gradient_bound_is_tight_and_sufficient(name):
    """The derived tolerance of the two instant-interest gradients (golden_util.instant_interest_grad_bounds; used by the GPU
    parity tests instead of round 3's flat floor): against the FLOAT64 gradient of the oracle, (1) the reference's own fp32
    gradient (fixture) and the fp32 oracle's are inside it with the NOISE term alone (no relative allowance at all), and
    (2) it is far below the gradient itself -- a zeroed weight or bias gradient is outside it by a wide margin."""
    case, dims, batch, sd, fx = load_case(name)
    bounds = instant_interest_grad_bounds(sd, batch)
    p64 = orc.to_torch_params(sd, dtype=torch.float64)
    tb = {k: torch.from_numpy(v) for k, v in batch.items() if isinstance(v, np.ndarray) and v.ndim > 0}
    with orc.precision(torch.float64):
        r = orc.user_model_forward(p64, tb["x_history"], tb["x_target"], tb["x_global"], training=True)
        orc.user_model_loss(p64, tb["user_id"], r, tb["label"]).backward()
    p32 = orc.to_torch_params(sd)
    _, _, g32 = orc.train_step(p32, {"step": 0, "m": {}, "v": {}}, tb, lr=0.0)
    for k in (II_W, II_B):
        g64 = p64[k].grad.numpy().reshape(-1)
        noise_only = grad_tolerance(k, np.zeros(1), 0.0, bounds)          # II_NOISE * S, per entry
        for other in (np.asarray(fx["grad/" + k]).reshape(-1), g32[k].numpy().reshape(-1)):
            assert (np.abs(other - g64) <= noise_only.reshape(-1)).all(), k
        tol = grad_tolerance(k, g64, 1e-2, bounds).reshape(-1)
        live = bounds[k].reshape(-1) > 0                                  # (dead ReLU units: gradient exactly 0)
        assert np.abs(g64)[live].max() > 20 * tol[live].max(), (k, float(np.abs(g64).max()), float(tol.max()))
        assert not (np.abs(0.0 - g64) <= tol).all(), k                    # the mutation the GPU test applies to the kernel


@pytest.mark.parametrize("name", TRAJ_CASES)
def test_oracle_follows_the_reference_trajectory(name):
    """8 steps of the reference's loop (same batch every step at C3 dimensions: the logits grow to ~240 and the loss
    reaches its guarded/clamped regime; fresh batches on the tiny model): loss, logits, BatchNorm running statistics
    per step, parameters and Adam moments at the end."""
    case, dims, user_num, batches, sd, fx = load_trajectory(name)
    p = orc.to_torch_params(sd)
    st = {"step": 0, "m": {}, "v": {}}
    losses, rs, rms, rvs = [], [], [], []
    for b in batches:
        tb = {k: torch.from_numpy(v) for k, v in b.items() if isinstance(v, np.ndarray) and v.ndim > 0}
        loss, r, _ = orc.train_step(p, st, tb)
        losses.append(float(loss)); rs.append(r.numpy().copy())
        rms.append(p["bn.running_mean"].numpy().copy()); rvs.append(p["bn.running_var"].numpy().copy())
    params = {k: v.detach().numpy() for k, v in p.items() if k not in orc.BUFFER_KEYS}
    worst = check_trajectory(fx, sd, losses, rs, rms, rvs, params, {k: v.numpy() for k, v in st["m"].items()},
                             {k: v.numpy() for k, v in st["v"].items()}, move_tol=5e-3)
    print(name, worst)


def test_oracle_attention_2d_target():
    import os
    from golden_util import GOLDEN
    fx = np.load(os.path.join(GOLDEN, "attention_2d.npz"))
    p = {"a." + k[2:]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("w/")}
    s = orc.pointwise_attention_scores(p, "a", torch.from_numpy(fx["target"]), torch.from_numpy(fx["history"]))
    assert tuple(s.shape) == tuple(fx["scores"].shape)
    assert rel_err(s.numpy(), fx["scores"]) < 1e-5
    pm = {"m." + k[6:]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("mlp_w/")}
    y = orc.mlp(pm, "m", torch.from_numpy(fx["mlp_x"]))
    assert rel_err(y.numpy(), fx["mlp_y"]) < 1e-5


def test_row_auc_ties_and_errors():
    assert orc.row_auc([0, 1, 0], [0.1, 0.9, 0.3]) == 1.0
    assert orc.row_auc([0, 1, 0], [0.5, 0.5, 0.5]) == 0.5
    assert orc.row_auc([1, 0, 0, 0], [0.2, 0.2, 0.1, 0.9]) == pytest.approx((1 + 0.5) / 3)
    with pytest.raises(ValueError):
        orc.row_auc([0, 0], [0.1, 0.2])


def test_oracle_degenerate_inputs_behave_like_the_reference():
    """MANIFEST "degenerate" = what the reference model does with no rows (recorded by oracle/make_golden.py from the
    reference itself): the oracle must raise / return the same."""
    import json
    import os
    from golden_util import GOLDEN
    from news_recommendation_model_amd import config, synth
    with open(os.path.join(GOLDEN, "MANIFEST.json")) as f:
        want = json.load(f)["degenerate"]
    dims = config.Dims.for_emb(16, 40)
    p = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.make_state_dict(dims, seed=3, user_num=5).items()}
    batch = synth.make_batch(dims, 3, 2, 4, seed=11, user_num=5)
    xh, xt, xg = (torch.from_numpy(batch[k]).float() for k in ("x_history", "x_target", "x_global"))

    def probe(fn):
        try:
            out = fn()
            return "shape:" + "x".join(str(int(d)) for d in out.shape)
        except Exception as e:                                     # noqa: BLE001
            return "raises:" + type(e).__name__
    for mode in ("train", "eval"):
        tr = mode == "train"
        got = {"empty_history_" + mode: probe(lambda: orc.user_model_forward(p, xh[:, :0], xt, xg, training=tr)),
               "empty_batch_" + mode: probe(lambda: orc.user_model_forward(p, xh[:0], xt[:0], xg[:0], training=tr)),
               "no_candidates_" + mode: probe(lambda: orc.user_model_forward(p, xh, xt[:, :0], xg[:, :0], training=tr)),
               "single_row_" + mode: probe(lambda: orc.user_model_forward(p, xh[:1], xt[:1, :1], xg[:1, :1], training=tr))}
        for k, v in got.items():
            assert v == want[k], (k, v, want[k])
    pre = "invariant_interest_model.text_img_attention"
    D = dims.pca_vector
    for name, (B, T, H) in {"B0": (0, 3, 4), "T0": (2, 0, 4), "H0": (2, 3, 0)}.items():
        assert probe(lambda: orc.pointwise_attention_scores(p, pre, torch.zeros(B, T, D), torch.zeros(B, H, D))) == want["attention_" + name]
    loss = orc.user_model_loss(p, torch.zeros(0, dtype=torch.long), torch.zeros(0, 4), torch.zeros(0, 4))
    assert bool(torch.isnan(loss)) == want["empty_batch_loss_is_nan"]
