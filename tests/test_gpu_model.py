"""GPU: the full UserModel step (train.py:66-75: forward, loss, backward, Adam) against the golden
fixtures produced by the REFERENCE model (tests/golden/*.npz).
Tolerances (BASELINE.json north_star): forward <= 1e-3 relative, gradients <= 1e-2 relative."""
import numpy as np
import pytest
import torch

from golden_util import MODEL_CASES, TRAIN_CASES, ZERO_GRAD_KEYS, load_case, pick, rel_err
from oracle import user_model_oracle as orc

pytestmark = pytest.mark.gpu
FWD_TOL, GRAD_TOL = 1e-3, 1e-2


def _model_and_batch(name):
    from news_recommendation_model_amd import trainer
    case, dims, batch, sd, fx = load_case(name)
    model = trainer.build_model(dims, int(batch["user_num"]), sd, device="cuda")
    tb = trainer.batch_to_device(batch, "cuda")            # float64 inputs, as the reference DataLoader yields
    return case, model, tb, batch, fx


@pytest.mark.parametrize("name", TRAIN_CASES)
def test_train_step_matches_reference_fixture(lib, name):
    from news_recommendation_model_amd import trainer
    case, model, tb, batch, fx = _model_and_batch(name)
    full = case["full"]
    model.train()
    opt = trainer.make_optimizer(model)
    grabbed = {}
    hook = model.invariant_interest_model.register_forward_hook(lambda m, i, o: grabbed.__setitem__("inv", o))
    out = model(tb["x_history"], tb["x_target"], tb["x_global"])
    hook.remove()
    loss = model.loss(tb["user_id"], out, tb["label"])
    loss.backward()
    torch.cuda.synchronize()
    r = out.detach().cpu().numpy()
    assert rel_err(r, fx["r"]) < FWD_TOL
    assert abs(float(loss.detach()) - float(fx["loss"])) < FWD_TOL * abs(float(fx["loss"]))
    assert rel_err(grabbed["inv"][0].detach().cpu().numpy(), fx["eu_H"]) < FWD_TOL
    assert rel_err(grabbed["inv"][1].detach().cpu().numpy(), fx["ec"]) < FWD_TOL
    # per-row AUC of the new r equals the reference's (train.py:77-80)
    np.testing.assert_allclose(orc.batch_auc(batch["label"], r), fx["auc"], atol=1e-6)

    gscale = max(float(fx["gradnorm/" + k]) for k, _ in model.named_parameters())
    for k, v in model.named_parameters():
        ref = fx["grad/" + k]
        got = pick(v.grad.cpu().numpy(), full)
        if k in ZERO_GRAD_KEYS:
            assert np.abs(got).max() < 1e-5 * max(1.0, gscale), k
        else:
            assert np.abs(got - ref).max() <= GRAD_TOL * np.abs(ref).max() + 1e-9, k
    opt.step()
    opt.zero_grad()
    torch.cuda.synchronize()
    for k, v in model.state_dict().items():
        ref = fx["after/" + k].astype(np.float64)
        got = pick(v.cpu().numpy(), full).astype(np.float64)
        # one Adam step moves a weight by <= lr = 1e-3; sign flips only where |g| ~ eps
        assert np.abs(got - ref).max() < 2.5e-3, k
        if k.startswith("bn.running"):
            assert rel_err(got, ref) < FWD_TOL, k


def test_eval_mode_uses_running_stats(lib):
    case, model, tb, batch, fx = _model_and_batch("tiny_eval")
    model.eval()
    with torch.no_grad():
        out = model(tb["x_history"], tb["x_target"], tb["x_global"])
        loss = model.loss(tb["user_id"], out, tb["label"])
    assert rel_err(out.cpu().numpy(), fx["r"]) < FWD_TOL
    assert abs(float(loss.detach()) - float(fx["loss"])) < FWD_TOL * abs(float(fx["loss"]))


def test_float32_and_float64_inputs_agree(lib):
    case, model, tb, batch, fx = _model_and_batch("tiny_train")
    model.eval()
    with torch.no_grad():
        a = model(tb["x_history"], tb["x_target"], tb["x_global"])
        b = model(tb["x_history"].float(), tb["x_target"].float(), tb["x_global"].float())
    assert torch.allclose(a, b, rtol=1e-6, atol=1e-6)


def test_variable_T_trim_like_test_py(lib):
    """reference test.py:48-56 trims trailing all-padding candidates before the forward: the scores of the
    kept candidates must not change in eval mode (impression rows are independent under running-stat BN)."""
    case, model, tb, batch, fx = _model_and_batch("tiny_pad")
    model.eval()
    k = case["pad_target"]
    with torch.no_grad():
        full_out = model(tb["x_history"], tb["x_target"], tb["x_global"])
        trim_out = model(tb["x_history"], tb["x_target"][:, :-k], tb["x_global"][:, :-k])
    assert torch.allclose(full_out[:, :-k], trim_out, rtol=1e-5, atol=1e-6)


def test_loss_kernel_matches_oracle_including_clamp(lib):
    """nrm_loss_fwd_bwd (one wave per impression) against the oracle's BCELoss/softmax restatement: value,
    dL/dout and dL/ddelta, including logits extreme enough to hit the -100 log clamp and T > 64."""
    from news_recommendation_model_amd import ops
    rng = np.random.default_rng(2)
    for B, T, scale in ((5, 7, 1.0), (3, 100, 3.0), (4, 6, 60.0)):
        out = (rng.standard_normal((B, T)) * scale).astype(np.float32)
        label = np.zeros((B, T), dtype=np.float64)
        label[np.arange(B), rng.integers(0, T, B)] = 1
        uid = rng.integers(0, 9, B)
        delta = (rng.standard_normal(9) * 0.3).astype(np.float32)
        o_c = torch.from_numpy(out).requires_grad_(True)
        d_c = torch.from_numpy(delta).requires_grad_(True)
        l_c = orc.user_model_loss({"delta": d_c}, torch.from_numpy(uid), o_c, torch.from_numpy(label))
        l_c.backward()
        o_g = torch.from_numpy(out).cuda().requires_grad_(True)
        d_g = torch.from_numpy(delta).cuda().requires_grad_(True)
        l_g = ops.softmax_bce_loss(o_g, d_g, torch.from_numpy(label).cuda(), torch.from_numpy(uid).cuda(), 0.95)
        (2.0 * l_g).backward()
        assert abs(float(l_g) - float(l_c)) <= 1e-5 * max(1.0, abs(float(l_c)))
        assert rel_err(o_g.grad.cpu().numpy() / 2.0, o_c.grad.numpy()) < 1e-4
        assert np.abs(d_g.grad.cpu().numpy() / 2.0 - d_c.grad.numpy()).max() < 1e-6


def test_pool_kernels_match_bmm(lib):
    from news_recommendation_model_amd import ops
    torch.manual_seed(1)
    for B, T, H, D in ((3, 5, 7, 64), (2, 30, 50, 400), (1, 9, 130, 100), (2, 3, 5, 30), (2, 17, 70, 36), (1, 33, 64, 20),
                       (5, 16, 65, 132), (1, 1, 1, 4)):
        s = torch.randn(B, T, H, device="cuda", requires_grad=True)
        h = torch.randn(B, H, D, device="cuda", requires_grad=True)
        g = torch.randn(B, T, D, device="cuda")
        out = ops.weighted_pool(s, h)
        out.backward(g)
        ref = torch.bmm(s.detach().double(), h.detach().double())
        assert rel_err(out.detach().cpu().numpy(), ref.cpu().numpy()) < 1e-5
        assert rel_err(s.grad.cpu().numpy(), torch.bmm(g.double(), h.detach().double().transpose(1, 2)).cpu().numpy()) < 1e-5
        assert rel_err(h.grad.cpu().numpy(), torch.bmm(s.detach().double().transpose(1, 2), g.double()).cpu().numpy()) < 1e-5


def test_flat_adam_matches_torch_adam_and_fixture(lib):
    """FlatAdam (one fused launch over a flat buffer) == torch.optim.Adam(lr 1e-3, weight_decay 1e-5) on the
    same gradients for three steps, and the first step matches the reference fixture."""
    from news_recommendation_model_amd import trainer
    case, model, tb, batch, fx = _model_and_batch("tiny_train")
    twin = _model_and_batch("tiny_train")[1]
    model.train(); twin.train()
    fopt = trainer.FlatAdam(model)
    topt = trainer.make_optimizer(twin)
    for step in range(3):
        for m, o in ((model, fopt), (twin, topt)):
            out = m(tb["x_history"], tb["x_target"], tb["x_global"])
            m.loss(tb["user_id"], out, tb["label"]).backward()
            o.step()
            o.zero_grad()
        if step == 0:
            for k, v in model.state_dict().items():
                ref = fx["after/" + k].astype(np.float64)
                assert np.abs(v.cpu().numpy().reshape(-1).astype(np.float64) - ref).max() < 2.5e-3, k
    # identical math on identical gradients: the two optimizers stay together to float rounding
    # (atomics in the attention backward make gradients differ in the last bits between the two models)
    for (k, a), (_, b) in zip(model.named_parameters(), twin.named_parameters()):
        assert torch.allclose(a, b, rtol=0, atol=3e-4), k
    assert float(fopt.flat_grad.abs().max()) == 0.0          # zero_grad fused into the step
    # state_dict round trip still works on the re-seated parameters
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model.load_state_dict(sd)
    assert model.gate.fc1.weight.data_ptr() >= fopt.flat_param.data_ptr()


def test_frontend_flags_out_of_range_index_like_indexerror(lib):
    """The reference raises IndexError for a category id >= category_label_num (nn.Embedding on CPU); the HIP
    front end clamps and raises the flag that ops.check_index_errors() turns into IndexError."""
    from news_recommendation_model_amd import ops
    case, model, tb, batch, fx = _model_and_batch("tiny_train")
    model.eval()
    ops.check_index_errors("cuda")                          # clean
    bad = tb["x_history"].clone()
    bad[0, 0, 4 + 64] = 10_000                              # category column of the first history row
    with torch.no_grad():
        model(bad, tb["x_target"], tb["x_global"])
    with pytest.raises(IndexError):
        ops.check_index_errors("cuda")
    ops.check_index_errors("cuda")                          # flag is cleared after raising


def test_graphed_train_step_replays_like_eager(lib):
    """trainer.GraphedTrainStep: the whole step (forward, loss, backward, Adam+zero_grad) captured in one HIP graph;
    three replays must land where three eager steps land (same kernels, float-atomic order noise only)."""
    from news_recommendation_model_amd import trainer
    case, eager, tb, batch, fx = _model_and_batch("tiny_train")
    graphed = _model_and_batch("tiny_train")[1]
    eager.train(); graphed.train()
    eopt, gopt = trainer.FlatAdam(eager), trainer.FlatAdam(graphed)
    for _ in range(3 + 3):                                   # GraphedTrainStep runs 3 eager warm-up steps itself
        le, _ = trainer.train_step(eager, eopt, tb)
    step = trainer.GraphedTrainStep(graphed, gopt, tb, warmup=3)
    losses = []
    for _ in range(3):
        lg, out = step.replay()
        losses.append(float(lg))
    torch.cuda.synchronize()
    assert gopt.steps == eopt.steps == 6                     # 3 warm-up steps + 3 replays (the capture does not execute)
    assert abs(losses[-1] - float(le)) < 1e-4 * abs(float(le))
    assert losses[0] > losses[-1]                            # it is really training
    for (k, a), (_, b) in zip(eager.named_parameters(), graphed.named_parameters()):
        assert torch.allclose(a, b, rtol=0, atol=5e-4), k


def test_batch_prefetcher_feeds_the_same_steps(lib):
    """Host (float64, as the DataLoader yields) -> HBM staging on a copy stream, one batch ahead, double buffered:
    the steps must see exactly the batches a plain .to(device) would hand them (train.py:66-68)."""
    from news_recommendation_model_amd import synth, trainer
    case, dims, batch, sd, fx = load_case("tiny_train")
    user_num = int(batch["user_num"])
    B, H, T = batch["x_history"].shape[0], batch["x_history"].shape[1], batch["x_target"].shape[1]
    hosts = [synth.make_batch(dims, B, H, T, seed=100 + i, user_num=user_num) for i in range(5)]
    losses = []
    for use_prefetcher in (False, True):
        model = trainer.build_model(dims, user_num, sd, device="cuda").train()
        opt = trainer.FlatAdam(model)
        out = []
        if use_prefetcher:
            pf = trainer.BatchPrefetcher(iter(hosts), "cuda")
            for b, slot in pf:
                loss, _ = trainer.train_step(model, opt, b)
                pf.release(slot)
                out.append(float(loss))
        else:
            for hb in hosts:
                loss, _ = trainer.train_step(model, opt, trainer.batch_to_device(hb, "cuda"))
                out.append(float(loss))
        losses.append(out)
    assert len(losses[1]) == len(hosts)
    np.testing.assert_allclose(losses[1], losses[0], rtol=2e-5)      # float atomics in the backward: not bit-identical


def _fuzz_models(n=10, seed=77):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        out.append((int(rng.choice([16, 24, 40, 72, 104])), int(rng.integers(2, 7)), int(rng.integers(1, 40)),
                    int(rng.integers(2, 24)), int(rng.integers(0, 3)), int(rng.integers(0, 2))))
    return out


@pytest.mark.parametrize("emb,B,H,T,pad_h,pad_t", _fuzz_models())
def test_random_models_match_oracle(lib, emb, B, H, T, pad_h, pad_t):
    """Seeded random model widths / batch shapes / padding: the whole step (forward, loss, every gradient) against the
    oracle (which is pinned to the reference by the fixtures above)."""
    from news_recommendation_model_amd import config, synth, trainer
    dims = config.Dims.for_emb(emb, 37)
    user_num = 3 * B
    pad_h, pad_t = min(pad_h, H - 1), min(pad_t, T - 1)
    batch = synth.make_batch(dims, B, H, T, seed=emb * 7 + B, user_num=user_num, pad_history=pad_h, pad_target=pad_t)
    sd = synth.make_state_dict(dims, seed=emb + T, user_num=user_num)
    p = orc.to_torch_params(sd)
    tb_cpu = {k: torch.from_numpy(v) for k, v in batch.items() if isinstance(v, np.ndarray) and v.ndim > 0}
    tb_cpu = {k: (v.float() if v.is_floating_point() else v) for k, v in tb_cpu.items()}
    loss_o, r_o, g_o = orc.train_step(p, {"step": 0, "m": {}, "v": {}}, tb_cpu, lr=0.0)      # lr 0: values and gradients only
    model = trainer.build_model(dims, user_num, sd, device="cuda").train()
    tb = trainer.batch_to_device(batch, "cuda")
    out = model(tb["x_history"], tb["x_target"], tb["x_global"])
    loss = model.loss(tb["user_id"], out, tb["label"])
    loss.backward()
    assert rel_err(out.detach().cpu().numpy(), r_o.numpy()) < FWD_TOL
    assert abs(float(loss.detach()) - float(loss_o)) < FWD_TOL * abs(float(loss_o))
    gscale = max(float(g.abs().max()) for g in g_o.values())
    for k, v in model.named_parameters():
        ref = g_o[k].numpy()
        got = v.grad.cpu().numpy()
        if k in ZERO_GRAD_KEYS:
            assert np.abs(got).max() < 1e-5 * max(1.0, gscale), k
        else:
            assert np.abs(got - ref).max() <= GRAD_TOL * np.abs(ref).max() + 1e-9, k
