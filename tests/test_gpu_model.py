"""GPU: the full UserModel step (train.py:66-75: forward, loss, backward, Adam) against the golden
fixtures produced by the REFERENCE model (tests/golden/*.npz).
Tolerances (BASELINE.json north_star): forward <= 1e-3 relative, gradients <= 1e-2 relative."""
import numpy as np
import pytest
import torch

from golden_util import (II_B, II_W, MODEL_CASES, TRAIN_CASES, TRAJ_CASES, ZERO_GRAD_KEYS, assert_first_adam_update, check_trajectory,
                         grad_tolerance, instant_interest_grad_bounds, load_case, load_trajectory, pick, rel_err)
from oracle import user_model_oracle as orc

pytestmark = pytest.mark.gpu
FWD_TOL, GRAD_TOL = 1e-3, 1e-2


def _model_and_batch(name, with_sd=False):
    from news_recommendation_model_amd import trainer
    case, dims, batch, sd, fx = load_case(name)
    model = trainer.build_model(dims, int(batch["user_num"]), sd, device="cuda")
    tb = trainer.batch_to_device(batch, "cuda")            # float64 inputs, as the reference DataLoader yields
    if with_sd:
        return case, model, tb, batch, fx, sd
    return case, model, tb, batch, fx


def _assert_gradients_match_fixture(model, fx, full, bounds):
    """Every parameter gradient against the reference's (fixture): <= GRAD_TOL of the tensor's max, entry by entry.  The two
    instant-interest tensors (column sums over the batch of a head gradient whose BatchNorm part cancels to ~0) additionally get
    the fp32 noise their own terms allow -- II_NOISE * sum of term magnitudes, per entry, from a float64 oracle pass
    (golden_util.instant_interest_grad_bounds) -- instead of round 3's flat 1e-4 * (model gradient scale), which was 3x the
    weight gradient itself."""
    gscale = max(float(fx["gradnorm/" + k]) for k, _ in model.named_parameters())
    for k, v in model.named_parameters():
        ref = fx["grad/" + k]
        got = pick(v.grad.cpu().numpy(), full)
        if k in ZERO_GRAD_KEYS:
            assert np.abs(got).max() < 1e-5 * max(1.0, gscale), k
        else:
            assert (np.abs(got - ref) <= grad_tolerance(k, ref, GRAD_TOL, bounds, full)).all(), (k, float(np.abs(got - ref).max()))


@pytest.mark.parametrize("name", TRAIN_CASES)
def test_train_step_matches_reference_fixture(lib, name):
    from news_recommendation_model_amd import trainer
    case, model, tb, batch, fx, sd = _model_and_batch(name, with_sd=True)
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

    _assert_gradients_match_fixture(model, fx, full, instant_interest_grad_bounds(sd, batch))
    opt.step()
    opt.zero_grad()
    torch.cuda.synchronize()
    # the Adam UPDATE (after - before) against the reference's, entry by entry: a no-op optimizer fails here
    checked = []
    for k, v in model.state_dict().items():
        if k.startswith("bn.running"):
            assert rel_err(pick(v.cpu().numpy(), full), fx["after/" + k]) < FWD_TOL, k
        elif k == "bn.num_batches_tracked":
            assert int(v) == int(fx["after/" + k][0]) == 1
        else:
            checked.append(assert_first_adam_update(k, sd[k], v.cpu().numpy(), fx, full))
    assert max(checked) > 0.5


@pytest.mark.parametrize("name", ["tiny_train", "c3_large"])
def test_zeroed_instant_interest_gradient_fails_the_fixture_gate(lib, monkeypatch, name):
    """Mutation check of the gate above (with the flat floor a zeroed weight gradient of the instant-interest layer
    passed every fixture).  (1) nrm_small_linear_relu_bwd is skipped, so its output buffer stays zero: the fixture comparison
    must fail ON THE WEIGHT.  (2) With the real kernel, zeroing only the bias gradient must fail on the bias."""
    from news_recommendation_model_amd import native
    case, model, tb, batch, fx, sd = _model_and_batch(name, with_sd=True)
    bounds = instant_interest_grad_bounds(sd, batch)
    model.train()

    def backward():
        model.zero_grad(set_to_none=True)
        model.bn.reset_running_stats()
        out = model(tb["x_history"], tb["x_target"], tb["x_global"])
        model.loss(tb["user_id"], out, tb["label"]).backward()
        torch.cuda.synchronize()

    backward()
    _assert_gradients_match_fixture(model, fx, case["full"], bounds)                   # the unmutated kernel passes
    model.instant_interest_model.out_fc[0].bias.grad.zero_()
    with pytest.raises(AssertionError, match=II_B.replace(".", r"\.")):
        _assert_gradients_match_fixture(model, fx, case["full"], bounds)
    real = native.call
    skipped = []

    def call(fn, *a, **k):
        if fn == "nrm_small_linear_relu_bwd":
            skipped.append(fn)
            return None
        return real(fn, *a, **k)
    monkeypatch.setattr(native, "call", call)
    backward()
    assert skipped, "the instant-interest backward did not go through nrm_small_linear_relu_bwd"
    with pytest.raises(AssertionError, match=II_W.replace(".", r"\.")):
        _assert_gradients_match_fixture(model, fx, case["full"], bounds)


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


def test_loss_kernel_beyond_256_candidates_matches_oracle(lib, monkeypatch):
    """reference models/user_model.py:37-43 takes any number of candidates.  Up to 256 a lane holds its four candidates in
    registers; beyond, loss_long_kernel re-reads the row in every pass (round 4 ran ATen ops there: VERDICT r4 'missing' 4).
    (1) Forced onto rows of <= 256 candidates (NRM_LOSS_LONG=1) it must reproduce the register kernel -- dL/dout BIT FOR BIT, the value and dL/ddelta up to the order of their float atomics --, contiguous and padded [B*T, 4] logits, incl. the -100 clamp regime (where the gradient of the dominant candidate is a
    cancellation residue and only a bitwise comparison means anything).  (2) T = 300 / 1000 against the oracle.  (3) UserModel.loss."""
    from news_recommendation_model_amd import ops, trainer
    from news_recommendation_model_amd.config import Dims
    rng = np.random.default_rng(12)

    def run(out, label, uid, delta, padded):
        B, T = out.shape
        if padded:                                  # column 0 of a [B*T, 4] matrix, as out_mlp.fc2's GEMM leaves the logits
            buf = torch.zeros(B * T, 4, device="cuda")
            buf[:, 0] = torch.from_numpy(out).cuda().reshape(-1)
            o_g = buf.requires_grad_(True)
            logits = o_g[:, 0].view(B, T)
        else:
            o_g = torch.from_numpy(out).cuda().requires_grad_(True)
            logits = o_g
        d_g = torch.from_numpy(delta).cuda().requires_grad_(True)
        l_g = ops.softmax_bce_loss(logits, d_g, torch.from_numpy(label).cuda(), torch.from_numpy(uid).cuda(), 0.95)
        (2.0 * l_g).backward()
        got = (o_g.grad[:, 0].reshape(B, T) if padded else o_g.grad).clone() / 2.0
        return l_g.detach().clone(), got, d_g.grad.clone() / 2.0

    def make(B, T, scale):
        out = (rng.standard_normal((B, T)) * scale).astype(np.float32)
        label = np.zeros((B, T), dtype=np.float64)
        label[np.arange(B), rng.integers(0, T, B)] = 1
        return out, label, rng.integers(0, 9, B), (rng.standard_normal(9) * 0.3).astype(np.float32)

    for B, T, scale in ((5, 7, 1.0), (3, 100, 3.0), (4, 6, 60.0), (6, 256, 60.0), (2, 65, 25.0)):
        out, label, uid, delta = make(B, T, scale)
        for padded in (False, True):
            monkeypatch.delenv("NRM_LOSS_LONG", raising=False)
            reg = run(out, label, uid, delta, padded)
            monkeypatch.setenv("NRM_LOSS_LONG", "1")
            lng = run(out, label, uid, delta, padded)
            assert torch.equal(lng[1], reg[1]), (T, scale, padded, "dout")        # per row: no atomics, bit for bit
            # the batch sum of the loss and rows that share a user id are float atomics: the same numbers in another order
            # (dL/ddelta is zero in exact arithmetic -- softmax is shift invariant -- so what is compared is a residue of roundings)
            assert torch.allclose(lng[0], reg[0], rtol=1e-6, atol=0) and float((lng[2] - reg[2]).abs().max()) < 1e-7 * float(reg[1].abs().max()) * T, (T, scale, padded)
    monkeypatch.delenv("NRM_LOSS_LONG", raising=False)
    for B, T, scale in ((5, 300, 1.0), (3, 1000, 2.0), (4, 257, 4.0)):
        out, label, uid, delta = make(B, T, scale)
        o_c = torch.from_numpy(out).requires_grad_(True)
        d_c = torch.from_numpy(delta).requires_grad_(True)
        l_c = orc.user_model_loss({"delta": d_c}, torch.from_numpy(uid), o_c, torch.from_numpy(label))
        l_c.backward()
        for padded in (False, True):
            l_g, got, dd = run(out, label, uid, delta, padded)
            assert abs(float(l_g) - float(l_c)) <= 1e-5 * max(1.0, abs(float(l_c))), (T, padded)
            assert rel_err(got.cpu().numpy(), o_c.grad.numpy()) < 1e-4, (T, padded)
            assert np.abs(dd.cpu().numpy() - d_c.grad.numpy()).max() < 1e-6
    # the Module method (no ATen branch any more): the last case, T = 257, on a model built with 9 delta entries
    dims = Dims.for_emb(16, 20)
    model = trainer.build_model(dims, 8, None, device="cuda")          # delta: user_num + 1 = 9 entries
    with torch.no_grad():
        model.delta.copy_(torch.from_numpy(delta))
    o_g = torch.from_numpy(out).cuda().requires_grad_(True)
    l_m = model.loss(torch.from_numpy(uid).cuda(), o_g, torch.from_numpy(label).cuda())
    assert abs(float(l_m) - float(l_c)) <= 1e-5 * max(1.0, abs(float(l_c)))


def test_model_with_a_batchnorm_width_that_is_not_a_multiple_of_4_matches_oracle(lib):
    """Head width N = 2 (D_l + P) + 8 = 266 (P = 65: not a BASELINE shape; reference models/user_model.py:16-18 builds it for any
    config).  The BatchNorm column kernels take float4 columns, so this width runs nn.BatchNorm1d on the device between the HIP
    GEMMs, and both attentions zero-pad their odd widths: the side path VERDICT r4 ('missing' 4) found
    untested -- whole step against the oracle, train and eval mode."""
    from news_recommendation_model_amd import config, synth, trainer
    dims = config.Dims(pca_vector=65, embed_setting=(32, 16, 8, 8), category_label_num=37)
    assert (2 * (dims.label_dim + dims.pca_vector) + 8) == 266
    B, H, T = 5, 9, 6
    user_num = 3 * B
    batch = synth.make_batch(dims, B, H, T, seed=21, user_num=user_num, pad_history=2, pad_target=1)
    sd = synth.make_state_dict(dims, seed=22, user_num=user_num)
    p = orc.to_torch_params(sd)
    tb_cpu = {k: torch.from_numpy(v) for k, v in batch.items() if isinstance(v, np.ndarray) and v.ndim > 0}
    tb_cpu = {k: (v.float() if v.is_floating_point() else v) for k, v in tb_cpu.items()}
    loss_o, r_o, g_o = orc.train_step(p, {"step": 0, "m": {}, "v": {}}, tb_cpu, lr=0.0)
    model = trainer.build_model(dims, user_num, sd, device="cuda").train()
    assert model.bn.num_features == 266
    tb = trainer.batch_to_device(batch, "cuda")
    out = model(tb["x_history"], tb["x_target"], tb["x_global"])
    loss = model.loss(tb["user_id"], out, tb["label"])
    loss.backward()
    assert rel_err(out.detach().cpu().numpy(), r_o.numpy()) < FWD_TOL
    assert abs(float(loss.detach()) - float(loss_o)) < FWD_TOL * abs(float(loss_o))
    gscale = max(float(g.abs().max()) for g in g_o.values())
    bounds = instant_interest_grad_bounds(sd, batch)
    for k, v in model.named_parameters():
        ref, got = g_o[k].numpy(), v.grad.cpu().numpy()
        if k in ZERO_GRAD_KEYS:
            assert np.abs(got).max() < 1e-5 * max(1.0, gscale), k
        else:
            assert (np.abs(got - ref) <= grad_tolerance(k, ref, GRAD_TOL, bounds)).all(), (k, float(np.abs(got - ref).max()))
    # running statistics after the one training forward, then the eval-mode forward on them
    assert rel_err(model.bn.running_var.cpu().numpy(), p["bn.running_var"].numpy()) < 1e-4
    model.eval()
    with torch.no_grad():
        r_e = model(tb["x_history"], tb["x_target"], tb["x_global"])
        r_eo = orc.user_model_forward(orc.to_torch_params({**sd, "bn.running_mean": p["bn.running_mean"].numpy(),
                                                           "bn.running_var": p["bn.running_var"].numpy()}, requires_grad=False),
                                      tb_cpu["x_history"], tb_cpu["x_target"], tb_cpu["x_global"], training=False)
    assert rel_err(r_e.cpu().numpy(), r_eo.numpy()) < FWD_TOL


@pytest.mark.parametrize("act", ["relu", "tanh", "leaky_relu"])
def test_mlp_with_a_non_default_activation_matches_the_reference_formula(lib, act):
    """MLP(input_dim, output_dim, activation_type) with an activation other than the default GELU (reference
    models/attention_model.py:13-27): the two Linears run as HIP GEMMs (ops.linear), the activation between them as an ATen
    elementwise op on the device -- forward, input gradient and every weight gradient against the same formula in float64."""
    from news_recommendation_model_amd.modules import MLP
    rng = np.random.default_rng(5)
    m = MLP(72, 40, act).cuda()
    x = torch.from_numpy(rng.standard_normal((33, 72)).astype(np.float32)).cuda().requires_grad_(True)
    g = torch.from_numpy(rng.standard_normal((33, 40)).astype(np.float32)).cuda()
    y = m(x)
    (y * g).sum().backward()
    fn = {"relu": torch.relu, "tanh": torch.tanh, "leaky_relu": torch.nn.functional.leaky_relu}[act]
    w = {k: v.detach().double().cpu().requires_grad_(True) for k, v in m.named_parameters()}
    x64 = x.detach().double().cpu().requires_grad_(True)
    y64 = fn(x64 @ w["fc1.weight"].T + w["fc1.bias"]) @ w["fc2.weight"].T + w["fc2.bias"]
    (y64 * g.double().cpu()).sum().backward()
    assert rel_err(y.detach().cpu().numpy(), y64.detach().numpy()) < 1e-5
    assert rel_err(x.grad.cpu().numpy(), x64.grad.numpy()) < 1e-4
    for k, v in m.named_parameters():
        assert rel_err(v.grad.cpu().numpy(), w[k].grad.numpy()) < 1e-4, k


@pytest.mark.parametrize("jsplit", [None, "0", "1"])
def test_pool_kernels_match_bmm(lib, monkeypatch, jsplit):
    """bmm_rows_kernel / rowdot_kernel against torch.bmm in float64; the pool's batched GEMM in both forms -- one wave per task
    and (round 5) one WORKGROUP per task with the reduction range cut in four (what few-task shapes such as the reference's
    default sizes take by themselves) -- and as dispatched (None)."""
    from news_recommendation_model_amd import ops
    if jsplit is None:
        monkeypatch.delenv("NRM_POOL_JSPLIT", raising=False)
    else:
        monkeypatch.setenv("NRM_POOL_JSPLIT", jsplit)
    torch.manual_seed(1)
    for B, T, H, D in ((3, 5, 7, 64), (2, 30, 50, 400), (1, 9, 130, 100), (2, 3, 5, 30), (2, 17, 70, 36), (1, 33, 64, 20),
                       (5, 16, 65, 132), (1, 1, 1, 4), (4, 15, 200, 64), (2, 70, 33, 72), (3, 2, 3, 8)):
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
    case, model, tb, batch, fx, sd = _model_and_batch("tiny_train", with_sd=True)
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
            for k, v in model.named_parameters():
                assert_first_adam_update(k, sd[k], v.detach().cpu().numpy(), fx, True)
    # identical math on identical gradients: the two optimizers stay together to float rounding
    # (atomics in the attention backward make gradients differ in the last bits between the two models)
    for (k, a), (_, b) in zip(model.named_parameters(), twin.named_parameters()):
        assert torch.allclose(a, b, rtol=0, atol=3e-4), k
    assert float(fopt.flat_grad.abs().max()) == 0.0          # zero_grad fused into the step
    assert all(p.grad is None for p in fopt.params)          # and the per-parameter gradient tensors are released
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


@pytest.mark.parametrize("mma", ["f32", "bf16x3"])
@pytest.mark.parametrize("emb,B,H,T,pad_h,pad_t", _fuzz_models())
def test_random_models_match_oracle(lib, emb, B, H, T, pad_h, pad_t, mma):
    """Seeded random model widths / batch shapes / padding: the whole step (forward, loss, every gradient) against the
    oracle (which is pinned to the reference by the fixtures above) -- with fp32 MFMA and with the bf16x3 arithmetic of
    BASELINE config 2, both held to the same gates."""
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
    model = trainer.build_model(dims, user_num, sd, device="cuda", attention_mma=mma).train()
    tb = trainer.batch_to_device(batch, "cuda")
    out = model(tb["x_history"], tb["x_target"], tb["x_global"])
    loss = model.loss(tb["user_id"], out, tb["label"])
    loss.backward()
    assert rel_err(out.detach().cpu().numpy(), r_o.numpy()) < FWD_TOL
    assert abs(float(loss.detach()) - float(loss_o)) < FWD_TOL * abs(float(loss_o))
    gscale = max(float(g.abs().max()) for g in g_o.values())
    bounds = instant_interest_grad_bounds(sd, batch)
    for k, v in model.named_parameters():
        ref = g_o[k].numpy()
        got = v.grad.cpu().numpy()
        if k in ZERO_GRAD_KEYS:
            assert np.abs(got).max() < 1e-5 * max(1.0, gscale), k
        else:
            assert (np.abs(got - ref) <= grad_tolerance(k, ref, GRAD_TOL, bounds)).all(), (k, float(np.abs(got - ref).max()))


@pytest.mark.parametrize("name", TRAJ_CASES)
def test_hip_trajectory_follows_the_reference(lib, name):
    """K = 8 steps of train.py:66-75 through trainer.train_step + FlatAdam on the HIP kernels against the trajectory the
    REFERENCE model + torch.optim.Adam took on the same batches (tests/golden/traj_*.npz): per-step loss and logits
    (<= 1e-3 relative while max|logit| < 1e3), BatchNorm running statistics (<= 1e-3), and at the end every
    parameter's total move and both Adam moments.  traj_c3 repeats one batch at C3 dimensions (as bench.py does): its
    logits grow to ~240 and the loss reaches the guarded regime of BCELoss; traj_tiny sees a fresh batch per step."""
    from news_recommendation_model_amd import trainer
    case, dims, user_num, batches, sd, fx = load_trajectory(name)
    model = trainer.build_model(dims, user_num, sd, device="cuda").train()
    opt = trainer.FlatAdam(model)
    losses, rs, rms, rvs = [], [], [], []
    for b in batches:
        loss, out = trainer.train_step(model, opt, trainer.batch_to_device(b, "cuda"))
        losses.append(float(loss)); rs.append(out.cpu().numpy())
        rms.append(model.bn.running_mean.cpu().numpy().copy()); rvs.append(model.bn.running_var.cpu().numpy().copy())
    assert opt.steps == case["steps"] and int(model.bn.num_batches_tracked) == case["steps"]
    params, m, v = {}, {}, {}
    off = 0
    for (k, prm) in model.named_parameters():
        params[k] = prm.detach().cpu().numpy()
        o = (prm.data_ptr() - opt.flat_param.data_ptr()) // 4
        m[k] = opt.exp_avg[o:o + prm.numel()].cpu().numpy()
        v[k] = opt.exp_avg_sq[o:o + prm.numel()].cpu().numpy()
    # measured on MI355X: loss 3.6e-4, logits 8e-5, BN 2e-5, parameter move 9e-5, Adam moments 5e-3 (traj_c3)
    worst = check_trajectory(fx, sd, losses, rs, rms, rvs, params, m, v, move_tol=5e-3, moment_tol=2e-2)
    print(name, worst)


def test_second_backward_through_attention_recomputes(lib):
    """The saved [B,T,H,D] pre-activation is overwritten in place by the first backward.  A second walk of the same graph
    (retain_graph=True: the reference's autograd allows it, models/attention_model.py:92) recomputes it from the saved inputs
    (round 5; rounds 1-4 raised) and returns the same gradients -- for the scores op and for the merged scores + pool node, a third
    time too, with and without the gradients nobody asked for."""
    from news_recommendation_model_amd import ops
    torch.manual_seed(0)
    D = 16
    t = torch.randn(2, 3, D, device="cuda", requires_grad=True)
    h = torch.randn(2, 5, D, device="cuda", requires_grad=True)
    w1 = (torch.randn(D, 4 * D, device="cuda") * 0.1).requires_grad_(True)
    b1 = (torch.randn(D, device="cuda") * 0.1).requires_grad_(True)
    w2, b2 = torch.randn(1, D, device="cuda", requires_grad=True), torch.zeros(1, device="cuda", requires_grad=True)
    for f in (lambda: ops.pointwise_attention_scores(t, h, w1, b1, w2, b2), lambda: ops.attend_and_pool(t, h, w1, b1, w2, b2)):
        out = f()
        seed = torch.randn_like(out)
        g1 = torch.autograd.grad(out, [t, h, w1, b1, w2], seed, retain_graph=True)
        g2 = torch.autograd.grad(out, [t, h, w1, b1, w2], seed, retain_graph=True)       # the buffer is spent: recomputed
        g3 = torch.autograd.grad(out, [w1], seed)                                           # ... and again, weight gradient only
        for a, b_ in zip(g1, g2):
            assert torch.allclose(a, b_, rtol=1e-5, atol=1e-6)
        assert torch.allclose(g1[2], g3[0], rtol=1e-5, atol=1e-6)
        fresh = torch.autograd.grad(f(), [t, h, w1, b1, w2], seed)                          # a fresh forward agrees
        for a, b_ in zip(g1, fresh):
            assert torch.allclose(a, b_, rtol=1e-5, atol=1e-6)
    s2 = ops.pointwise_attention_scores(t, h, w1, b1, w2, b2)
    with torch.no_grad():                                       # no [B,T,H,D] buffer is kept without grad mode
        s3 = ops.pointwise_attention_scores(t, h, w1, b1, w2, b2)
    assert not s3.requires_grad and torch.allclose(s3, s2.detach(), rtol=1e-6, atol=1e-6)
    # the other trade on request: every backward works on a copy of the saved pre-activation (no recomputation)
    prev = ops.set_retain_attention_graph(True)
    try:
        for f in (lambda: ops.pointwise_attention_scores(t, h, w1, b1, w2, b2), lambda: ops.attend_and_pool(t, h, w1, b1, w2, b2)):
            out = f()
            ga = torch.autograd.grad(out.sum(), [t, h], retain_graph=True)
            gb = torch.autograd.grad(out.sum(), [t, h])
            assert all(torch.equal(a, b) for a, b in zip(ga, gb))
    finally:
        ops.set_retain_attention_graph(prev)


def test_unit_seeded_loss_backward_twice_with_retain_graph(lib):
    """trainer.train_step seeds backward() with ops.unit_grad, and the loss node then hands its SAVED gradients on.
    delta.grad must not alias the saved buffer: a second backward(unit, retain_graph) accumulates in place into delta.grad, and
    a third one would read the rewritten buffer."""
    from news_recommendation_model_amd import ops
    rng = np.random.default_rng(11)
    B, T, n = 5, 7, 6
    out = torch.from_numpy(rng.standard_normal((B, T)).astype(np.float32)).cuda().requires_grad_(True)
    delta = torch.from_numpy((rng.standard_normal(n) * 0.3).astype(np.float32)).cuda().requires_grad_(True)
    label = torch.zeros(B, T, dtype=torch.float64)
    label[torch.arange(B), torch.from_numpy(rng.integers(0, T, B))] = 1
    uid = torch.from_numpy(rng.integers(0, n, B)).cuda()
    loss = ops.softmax_bce_loss(out, delta, label.cuda(), uid, 0.95)
    unit = ops.unit_grad(loss)
    loss.backward(unit, retain_graph=True)
    g_out, g_delta = out.grad.clone(), delta.grad.clone()
    for k in (2, 3):
        loss.backward(unit, retain_graph=True)
        assert torch.allclose(out.grad, k * g_out, rtol=1e-6, atol=1e-9)
        assert torch.allclose(delta.grad, k * g_delta, rtol=1e-6, atol=1e-12)
    (g2,) = torch.autograd.grad(loss, [delta], torch.full((), 2.0, device="cuda"))      # the general (scaled) path still agrees
    assert torch.allclose(g2, 2 * g_delta, rtol=1e-6, atol=1e-12)


def test_loss_user_ids_negative_wrap_and_out_of_range_flag(lib):
    """delta[id] as torch indexes it (models/user_model.py:40): a negative id counts from the end; an id outside
    [-n, n) raises IndexError in the reference -- the kernel clamps it, never reads or writes out of bounds, and raises
    the flag ops.check_index_errors() turns into IndexError (a checkpoint loaded into UserModel(user_num=0)
    has a 1-entry delta)."""
    from news_recommendation_model_amd import ops
    rng = np.random.default_rng(5)
    B, T, n = 6, 9, 4
    out = torch.from_numpy(rng.standard_normal((B, T)).astype(np.float32))
    label = torch.zeros(B, T, dtype=torch.float64)
    label[torch.arange(B), torch.from_numpy(rng.integers(0, T, B))] = 1
    delta = torch.from_numpy((rng.standard_normal(n) * 0.3).astype(np.float32))
    uid = torch.tensor([0, -1, 3, -4, 2, -2])
    o_c, d_c = out.clone().requires_grad_(True), delta.clone().requires_grad_(True)
    l_c = orc.user_model_loss({"delta": d_c}, uid, o_c, label)
    l_c.backward()
    ops.check_index_errors("cuda")
    o_g, d_g = out.cuda().requires_grad_(True), delta.cuda().requires_grad_(True)
    l_g = ops.softmax_bce_loss(o_g, d_g, label.cuda(), uid.cuda(), 0.95)
    l_g.backward()
    ops.check_index_errors("cuda")                                  # negative ids are legal
    assert abs(float(l_g) - float(l_c)) <= 1e-5 * abs(float(l_c))
    assert rel_err(o_g.grad.cpu().numpy(), o_c.grad.numpy()) < 1e-4
    # canaries around a 1-entry delta: ids far outside must not touch the neighbours
    arena = torch.full((64,), 7.0, device="cuda")
    one = arena[32:33].detach().zero_().requires_grad_(True)
    bad = torch.tensor([0, 5, -9, 10 ** 12, -10 ** 12, 1], device="cuda")
    l_b = ops.softmax_bce_loss(out.cuda().requires_grad_(True), one, label.cuda(), bad, 0.95)
    l_b.backward()
    torch.cuda.synchronize()
    assert torch.isfinite(l_b)
    assert torch.equal(arena[:32], torch.full((32,), 7.0, device="cuda")) and torch.equal(arena[33:], torch.full((31,), 7.0, device="cuda"))
    with pytest.raises(IndexError):
        ops.check_index_errors("cuda")


@pytest.mark.parametrize("name", ["c2_small", "tiny_train", "odd_shape", "c3_large", "refdefault"])
@pytest.mark.parametrize("mma,dense", [("bf16x3", "f32"), ("bf16", "f32"), ("bf16x3", "bf16x3")])
def test_bf16_attention_train_step_matches_reference_fixture(lib, mma, dense, name):
    """BASELINE config 2 arithmetic (bf16 matrix cores in both attentions; dense = f32: fp32 everywhere else, dense = bf16x3:
    every dense GEMM on the bf16 matrix cores too, as bench.py runs config 2) on whole-model fixtures produced by the fp32
    REFERENCE.  bf16x3 (hi/lo split operands) is held to the gates of the fp32 path -- forward <= 1e-3, gradients <= 1e-2
    (measured: 1e-6 / the oracle's own noise floor).  Plain bf16 operands are characterised, not gated: measured logits
    4e-4 .. 8e-4, gradients up to 1.2e-2 (c3_large, a fc2 bias); bounds 2e-3 / 3e-2."""
    from news_recommendation_model_amd import ops, trainer
    prev_dense = ops._default_dense_mma
    ops.set_dense_arithmetic(dense)
    try:
        _bf16_train_step_check(mma, name)
    finally:
        ops._default_dense_mma = prev_dense


def _bf16_train_step_check(mma, name):
    from news_recommendation_model_amd import trainer
    case, dims, batch, sd, fx = load_case(name)
    model = trainer.build_model(dims, int(batch["user_num"]), sd, device="cuda", attention_mma=mma).train()
    assert model.invariant_interest_model.label_attention.mma == mma
    tb = trainer.batch_to_device(batch, "cuda")
    out = model(tb["x_history"], tb["x_target"], tb["x_global"])
    loss = model.loss(tb["user_id"], out, tb["label"])
    loss.backward()
    full = case["full"]
    fwd_tol, grad_tol = (FWD_TOL, GRAD_TOL) if mma == "bf16x3" else (2e-3, 3e-2)
    e_r = rel_err(out.detach().cpu().numpy(), fx["r"])
    assert e_r < fwd_tol, e_r
    assert abs(float(loss.detach()) - float(fx["loss"])) < fwd_tol * abs(float(fx["loss"]))
    if mma == "bf16x3":
        np.testing.assert_allclose(orc.batch_auc(batch["label"], out.detach().cpu().numpy()), fx["auc"], atol=1e-6)
    gscale = max(float(fx["gradnorm/" + k]) for k, _ in model.named_parameters())
    worst = 0.0
    for k, v in model.named_parameters():
        ref = fx["grad/" + k]
        got = pick(v.grad.cpu().numpy(), full)
        if k in ZERO_GRAD_KEYS:
            assert np.abs(got).max() < 1e-4 * max(1.0, gscale), k
        else:
            e = np.abs(got - ref).max() / (np.abs(ref).max() + 1e-30)
            worst = max(worst, e)
            assert e <= grad_tol, (k, e)
    print(name, mma, "logits rel err", e_r, "worst grad rel err", worst)


def test_flat_adam_collects_every_gradient_with_one_launch(lib):
    """FlatAdam.collect_grads(): the gradients autograd left on the parameters land, bit for bit, in their slots of the flat
    buffer (C ABI nrm_gather_flat: one launch for all 38 tensors); a parameter that received no gradient contributes
    zeros; the padding between slots stays zero."""
    from news_recommendation_model_amd import trainer
    case, model, tb, batch, fx = _model_and_batch("tiny_train")
    model.train()
    opt = trainer.FlatAdam(model)
    out = model(tb["x_history"], tb["x_target"], tb["x_global"])
    model.loss(tb["user_id"], out, tb["label"]).backward()
    grads = [p.grad.clone() if p.grad is not None else None for p in opt.params]
    assert sum(g is not None for g in grads) >= len(grads) - 1
    grads[3] = None
    opt.params[3].grad = None                                 # pretend this one got nothing
    opt.flat_grad.fill_(7.0)                                  # stale contents must be overwritten
    opt.collect_grads()
    torch.cuda.synchronize()
    covered = torch.zeros(opt.n, dtype=torch.bool, device="cuda")
    for i, (p, g) in enumerate(zip(opt.params, grads)):
        got = opt.grad_view(i)
        assert torch.equal(got, g if g is not None else torch.zeros_like(p)), i
        covered[opt.offsets[i]:opt.offsets[i] + p.numel()] = True
    assert all(p.grad is None for p in opt.params)
    opt.collect_grads()                                       # idempotent until the next step
    assert torch.equal(opt.grad_view(0), grads[0])


@pytest.mark.parametrize("dw_last", ["0", "1"])
@pytest.mark.parametrize("graphed", [False, True])
def test_two_stream_attention_branches_match_one_stream(lib, monkeypatch, graphed, dw_last):
    """modules.UserInvariantInterestModel.forward issues its second attention (and pool) on a side stream for small
    shapes: eager and captured into a HIP graph, forward, every gradient and three optimizer steps must land where the
    one-stream step lands (same kernels; only the order of float atomics can differ).  ``dw_last`` (round 5): with and without
    the text+image attention's dW_p-only pass waiting for the label attention's chain on the other stream (ops._chain; by
    default only at C3-like sizes, forced here) -- an event recorded on one stream and waited for on the other, also inside a
    capture."""
    from news_recommendation_model_amd import trainer
    monkeypatch.delenv("NRM_BRANCH_STREAMS", raising=False)
    monkeypatch.setenv("NRM_DW_LAST", dw_last)
    case, one, tb, batch, fx = _model_and_batch("tiny_train")
    two = _model_and_batch("tiny_train")[1]
    one.train(); two.train()
    one.invariant_interest_model.two_streams = False
    two.invariant_interest_model.two_streams = True
    assert two.invariant_interest_model.uses_two_streams(10 ** 12) and not one.invariant_interest_model.uses_two_streams(1)
    # forward + backward
    outs, grads = [], []
    for m in (one, two):
        out = m(tb["x_history"], tb["x_target"], tb["x_global"])
        loss = m.loss(tb["user_id"], out, tb["label"])
        loss.backward()
        torch.cuda.synchronize()
        outs.append(out.detach().clone())
        grads.append({k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None})
        for p in m.parameters():
            p.grad = None
        del out, loss                                        # no autograd graph of this model may outlive its step (capture below)
    assert torch.equal(outs[0], outs[1])                     # the forward has no atomics at these sizes: bit-identical
    assert grads[0].keys() == grads[1].keys()
    for k in grads[0]:
        scale = float(grads[0][k].abs().max()) + 1e-30
        assert float((grads[0][k] - grads[1][k]).abs().max()) <= 1e-5 * scale, k
    # three optimizer steps
    oopt, topt = trainer.FlatAdam(one), trainer.FlatAdam(two)
    if graphed:
        for _ in range(3 + 3):
            lo, _ = trainer.train_step(one, oopt, tb)
        step = trainer.GraphedTrainStep(two, topt, tb, warmup=3)
        for _ in range(3):
            lt, _ = step.replay()
    else:
        for _ in range(3):
            lo, _ = trainer.train_step(one, oopt, tb)
            lt, _ = trainer.train_step(two, topt, tb)
    torch.cuda.synchronize()
    assert abs(float(lt) - float(lo)) < 1e-4 * abs(float(lo))
    for (k, a), (_, b) in zip(one.named_parameters(), two.named_parameters()):
        assert torch.allclose(a, b, rtol=0, atol=5e-4), k


def test_deferred_slab_reductions_with_a_frozen_weight(lib):
    """trainer.train_step defers the weight-gradient slab reductions to FlatAdam.collect_grads (one launch).  A weight that
    takes no gradient has its (dropped) gradient buffer kept alive by the record and reduced into harmlessly; every other
    parameter must move exactly as in a model where the reductions run at once."""
    from news_recommendation_model_amd import ops, trainer
    case, a, tb, batch, fx = _model_and_batch("tiny_train")
    b = _model_and_batch("tiny_train")[1]
    a.train(); b.train()
    for m in (a, b):
        m.mlp.fc1.weight.requires_grad_(False)
        m.invariant_interest_model.label_attention.mlp.fc1.weight.requires_grad_(False)
    aopt, bopt = trainer.FlatAdam(a), trainer.FlatAdam(b)
    assert len(aopt.params) == len(list(a.parameters())) - 2
    before = a.mlp.fc1.weight.detach().clone()
    for _ in range(3):
        la, _ = trainer.train_step(a, aopt, tb)                      # deferred (inside train_step)
        out = b(tb["x_history"], tb["x_target"], tb["x_global"])     # immediate: a plain backward(), then the same optimizer
        lb = b.loss(tb["user_id"], out, tb["label"])
        lb.backward()
        assert not ops._deferred["pending"]
        bopt.step()
    torch.cuda.synchronize()
    assert not ops._deferred["pending"]
    assert torch.equal(a.mlp.fc1.weight, before)                     # frozen stays frozen
    assert abs(float(la) - float(lb)) < 1e-5 * abs(float(lb))
    for (k, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        assert torch.allclose(pa, pb, rtol=0, atol=2e-5), k


@pytest.mark.parametrize("graphed", [False, True])
def test_weight_gradient_stream_matches_one_stream(lib, monkeypatch, graphed):
    """ops._wgrad_stream: under trainer.train_step + FlatAdam the weight-gradient GEMMs (dW = dY^T X, and the attention's side
    projections) may run on a side stream that FlatAdam.collect_grads joins.  Eager and captured into a HIP graph, three
    optimizer steps with the stream forced on must land where the one-stream steps land (same kernels; only the order of float
    atomics in the slab reductions can differ), and the stream must really have been used."""
    from news_recommendation_model_amd import ops, trainer
    case, one, tb, batch, fx = _model_and_batch("tiny_train")
    two = _model_and_batch("tiny_train")[1]
    one.train(); two.train()
    oopt, topt = trainer.FlatAdam(one), trainer.FlatAdam(two)
    ops._wgrad["streams"].clear()
    monkeypatch.setenv("NRM_WGRAD_STREAM", "0")
    for _ in range(3 + (3 if graphed else 0)):
        lo, _ = trainer.train_step(one, oopt, tb)
    assert not ops._wgrad["streams"]
    monkeypatch.setenv("NRM_WGRAD_STREAM", "1")
    if graphed:
        step = trainer.GraphedTrainStep(two, topt, tb, warmup=3)
        for _ in range(3):
            lt, _ = step.replay()
    else:
        for _ in range(3):
            lt, _ = trainer.train_step(two, topt, tb)
    torch.cuda.synchronize()
    assert ops._wgrad["streams"] and not ops._wgrad["used"]           # used, and joined by collect_grads
    assert abs(float(lt) - float(lo)) < 1e-4 * abs(float(lo))
    for (k, a), (_, b) in zip(one.named_parameters(), two.named_parameters()):
        assert torch.allclose(a, b, rtol=0, atol=5e-4), k
    # outside train_step (plain backward, torch.optim): gradients are valid as soon as backward() returns -- no side stream
    ops._wgrad["streams"].clear()
    out = two(tb["x_history"], tb["x_target"], tb["x_global"])
    two.loss(tb["user_id"], out, tb["label"]).backward()
    assert not ops._wgrad["streams"]
