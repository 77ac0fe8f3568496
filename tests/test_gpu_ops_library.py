"""GPU: the hot-path operators as torch.library custom ops (namespace ``nrm``; SURVEY.md §8b "torch.library ops in one
namespace").  Every op is reachable through ``torch.ops.nrm.*``, carries a fake-tensor shape function and an autograd
registration that ``torch.library.opcheck`` accepts, and the Modules -- which call these ops -- still pickle into a child
process the way reference test.py:172-182 hands its model list to a worker."""
import pickle
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
# Full default opcheck (schema incl. mutation annotations, fake-tensor function vs real outputs, autograd registration, AOT
# dispatch with dynamic shapes) for every op -- except the two attention ops, whose backward consumes the saved [B,T,H,D]
# pre-activation IN PLACE by design (2.46 GB at C3: DESIGN.md §4): AOT autograd refuses graphs that mutate a saved tensor in
# the backward, so `test_aot_dispatch_dynamic` is left out for them (and only for them).
CHECKS_ATTN = ("test_schema", "test_faketensor", "test_autograd_registration")


def _opcheck(op, args):
    torch.library.opcheck(op, args)                       # all four default checks


def _attn_args(B=2, T=3, H=5, D=16, grad=True):
    g = torch.Generator(device="cuda").manual_seed(0)
    r = lambda *s: torch.randn(*s, device="cuda", generator=g)          # noqa: E731
    t, h = r(B, T, D).requires_grad_(grad), r(B, H, D).requires_grad_(grad)
    w1, b1 = (0.1 * r(D, 4 * D)).requires_grad_(grad), (0.1 * r(D)).requires_grad_(grad)
    w2, b2 = r(1, D).requires_grad_(grad), r(1).requires_grad_(grad)
    return t, h, w1, b1, w2, b2


def test_every_op_is_registered_under_torch_ops_nrm(lib):
    from news_recommendation_model_amd import ops
    for name in ops.OPS:
        packet = getattr(torch.ops.nrm, name)
        assert packet.default._schema.name == "nrm::" + name
    assert "Tensor(a!) z" in str(torch.ops.nrm.pwattn_bwd.default._schema)             # in-place dz is declared


def test_opcheck_attention_and_pool(lib):
    from news_recommendation_model_amd import ops   # noqa: F401  (registers the ops)
    t, h, w1, b1, w2, b2 = _attn_args()
    torch.library.opcheck(torch.ops.nrm.pwattn_fwd.default, (t, h, w1, b1, w2, b2, True, 0), test_utils=CHECKS_ATTN)
    torch.library.opcheck(torch.ops.nrm.pwattn_fwd.default, (t.detach(), h.detach(), w1.detach(), b1.detach(), w2.detach(),
                                                             b2.detach(), False, 1), test_utils=CHECKS_ATTN)
    s, z = torch.ops.nrm.pwattn_fwd(t.detach(), h.detach(), w1.detach(), b1.detach(), w2.detach(), b2.detach(), True, 0)
    torch.library.opcheck(torch.ops.nrm.pwattn_bwd.default, (torch.randn_like(s), t.detach(), h.detach(), w1.detach(), w2.detach(), z, 0, True, True),
                          test_utils=CHECKS_ATTN)
    sc = torch.randn(2, 3, 5, device="cuda", requires_grad=True)
    _opcheck(torch.ops.nrm.weighted_pool_fwd.default, (sc, h))
    _opcheck(torch.ops.nrm.weighted_pool_bwd.default, (torch.randn(2, 3, 16, device="cuda"), sc.detach(), h.detach()))
    # scores + pool as one node
    torch.library.opcheck(torch.ops.nrm.attend_pool_fwd.default, (t, h, w1, b1, w2, b2, True, 0), test_utils=CHECKS_ATTN)
    pooled, s, z = torch.ops.nrm.attend_pool_fwd(t.detach(), h.detach(), w1.detach(), b1.detach(), w2.detach(), b2.detach(), True, 0)
    torch.library.opcheck(torch.ops.nrm.attend_pool_bwd.default, (torch.randn_like(pooled), t.detach(), h.detach(), w1.detach(),
                                                                  w2.detach(), s, z, 0, True, False), test_utils=CHECKS_ATTN)


@pytest.mark.parametrize("mma", ["f32", "bf16x3"])
@pytest.mark.parametrize("shape", [(3, 5, 7, 16), (4, 30, 32, 64), (2, 17, 40, 256)])
def test_attend_and_pool_node_matches_the_two_separate_nodes(lib, shape, mma):
    """ops.attend_and_pool (one autograd node; its backward reads the pooled gradient in place from a wider matrix, lets the
    pool's rowdot launch clear the attention's accumulators and adds the pool's history gradient with the last launch) against
    weighted_pool(pointwise_attention_scores(...)): same outputs, same gradients -- with a contiguous and with a column-block
    gradient."""
    from news_recommendation_model_amd import ops
    B, T, H, D = shape
    base = _attn_args(B, T, H, D, grad=False)
    gen = torch.Generator(device="cuda").manual_seed(5)
    wide = torch.randn(B * T, 2 * D + 8, device="cuda", generator=gen)
    for g in (torch.randn(B, T, D, device="cuda", generator=gen), wide[:, D:2 * D].unflatten(0, (B, T)), wide[:, 4:4 + D].unflatten(0, (B, T))):
        res = []
        for merged in (True, False):
            args = [a.clone().requires_grad_(True) for a in base]
            if merged:
                out = ops.attend_and_pool(*args, mma=mma)
            else:
                out = ops.weighted_pool(ops.pointwise_attention_scores(*args, mma=mma), args[1])
            out.backward(g)
            res.append((out.detach(), [a.grad for a in args]))
        assert torch.equal(res[0][0], res[1][0])
        for name, a, b in zip(("dt", "dh", "dfc1_w", "dfc1_b", "dfc2_w", "dfc2_b"), res[0][1], res[1][1]):
            scale = float(b.abs().max()) + 1e-30
            assert float((a - b).abs().max()) <= 2e-5 * scale, (name, float((a - b).abs().max()), scale)


def test_opcheck_dense_batchnorm_loss_frontend(lib):
    from news_recommendation_model_amd import config, ops, synth
    g = torch.Generator(device="cuda").manual_seed(1)
    r = lambda *s: torch.randn(*s, device="cuda", generator=g)          # noqa: E731
    x = r(37, 24).requires_grad_(True)
    w, b = (0.2 * r(10, 24)).requires_grad_(True), r(10).requires_grad_(True)
    for gelu in (False, True):
        _opcheck(torch.ops.nrm.linear_fwd.default, (x, w, b, gelu))
    w2, b2 = (0.2 * r(24, 10)).requires_grad_(True), r(24).requires_grad_(True)
    _opcheck(torch.ops.nrm.mlp_gelu_fwd.default, (x, w, b, w2, b2, None))
    m = r(37, 24).requires_grad_(True)
    _opcheck(torch.ops.nrm.mlp_gelu_fwd.default, (x, w, b, w2, b2, m))
    # BatchNorm: the statistics op mutates the running buffers (declared), the apply op is functional with autograd
    rm, rv = torch.zeros(24, device="cuda"), torch.ones(24, device="cuda")
    _opcheck(torch.ops.nrm.batch_norm_stats.default, (x.detach(), rm, rv, 0.1, 1e-5))
    mean, rstd = torch.ops.nrm.batch_norm_stats(x.detach(), rm, rv, 0.1, 1e-5)
    gamma, beta = r(24).requires_grad_(True), r(24).requires_grad_(True)
    _opcheck(torch.ops.nrm.batch_norm_apply.default, (x, mean, rstd, gamma, beta, True))
    # BatchNorm -> gate MLP -> product as one node, and the one-launch concat
    wg1, bg1 = (0.2 * r(6, 24)).requires_grad_(True), r(6).requires_grad_(True)
    wg2, bg2 = (0.2 * r(24, 6)).requires_grad_(True), r(24).requires_grad_(True)
    _opcheck(torch.ops.nrm.gate_block_fwd.default, (x, mean, rstd, gamma, beta, wg1, bg1, wg2, bg2, True))
    _opcheck(torch.ops.nrm.concat_cols.default, ([x, r(37, 3).requires_grad_(True), r(37, 10)[:, :7].requires_grad_(True)],))
    # loss
    out = r(6, 9).requires_grad_(True)
    delta = (0.1 * r(5)).requires_grad_(True)
    label = torch.zeros(6, 9, device="cuda")
    label[torch.arange(6), torch.arange(6)] = 1
    uid = torch.tensor([0, 1, 4, 2, 2, 3], device="cuda")
    _opcheck(torch.ops.nrm.softmax_bce_loss.default, (out, delta, label, uid, 0.95))
    # front end on packed rows
    dims = config.Dims.for_emb(16, 40)
    batch = synth.make_batch(dims, 3, 4, 2, seed=3, user_num=5)
    sd = synth.make_state_dict(dims, seed=2, user_num=5)
    inv = "invariant_interest_model."
    tabs = [torch.from_numpy(sd[inv + k]).cuda().requires_grad_(True) for k in (
        "category_embedding.0.weight", "sentiment_embedding.0.weight", "sentiment_embedding.0.bias", "type_embedding.0.weight",
        "year_embedding.0.weight", "month_embedding.0.weight", "day_embedding.0.weight", "hour_embedding.0.weight")]
    rows = torch.from_numpy(batch["x_history"]).cuda().reshape(-1, batch["x_history"].shape[-1])
    _opcheck(torch.ops.nrm.frontend_fwd.default, (rows, True, dims.n_subcat, dims.pca_vector, *tabs))
    rows_t = torch.from_numpy(batch["x_target"]).cuda().reshape(-1, batch["x_target"].shape[-1])
    _opcheck(torch.ops.nrm.frontend_pair_fwd.default, (rows, rows_t, dims.n_subcat, dims.pca_vector, *tabs))
    # one node for both row sets == the two separate nodes (table gradients accumulated in one arena vs summed by autograd)
    xh, xt = torch.from_numpy(batch["x_history"]).cuda(), torch.from_numpy(batch["x_target"]).cuda()
    res = []
    for pair in (True, False):
        tb = [t_.detach().clone().requires_grad_(True) for t_ in tabs]
        if pair:
            lab_h, ti_h, lab_t, ti_t = ops.frontend_pair(xh, xt, dims.n_subcat, dims.pca_vector, *tb)
        else:
            lab_h, ti_h = ops.frontend(xh, True, dims.n_subcat, dims.pca_vector, *tb)
            lab_t, ti_t = ops.frontend(xt, False, dims.n_subcat, dims.pca_vector, *tb)
        ((lab_h * lab_h).sum() + (lab_t.sin()).sum()).backward()
        res.append(([lab_h.detach(), ti_h, lab_t.detach(), ti_t], [t_.grad for t_ in tb]))
    for a, b in zip(res[0][0], res[1][0]):
        assert torch.equal(a, b)
    for a, b in zip(res[0][1], res[1][1]):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6)
    # evaluation + optimizer ops
    _opcheck(torch.ops.nrm.row_auc.default, (out.detach(), label, None))
    n = 64
    p_, g_, m_, v_ = r(n), r(n), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    st = torch.zeros(4, device="cuda")
    _opcheck(torch.ops.nrm.adam_step.default, (p_, g_, m_, v_, st, 1e-3, 0.9, 0.999, 1e-8, 1e-5, True))


@pytest.mark.parametrize("mma", ["f32", "bf16x3"])
@pytest.mark.parametrize("need", [(False, False), (True, False), (False, True)])
def test_attention_backward_skips_gradients_nobody_asked_for(lib, mma, need):
    """The text+image attention reads raw input columns: neither its target nor its history rows require a gradient, and the
    backward then skips the (b,h) contraction pass, the side-projection GEMMs and the pool's history product.  The weight
    gradients (and whichever row gradient IS wanted) must not change."""
    from news_recommendation_model_amd import ops
    base = _attn_args(3, 9, 33, 64, grad=False)
    g = torch.randn(3, 9, 64, device="cuda", generator=torch.Generator(device="cuda").manual_seed(7))
    res = []
    for flags in ((True, True), need):
        args = [a.clone().requires_grad_(True) for a in base]
        args[0].requires_grad_(flags[0])
        args[1].requires_grad_(flags[1])
        ops.attend_and_pool(*args, mma=mma).backward(g)
        res.append([a.grad for a in args])
    assert (res[1][0] is None) == (not need[0]) and (res[1][1] is None) == (not need[1])
    for name, a, b in zip(("dt", "dh", "dfc1_w", "dfc1_b", "dfc2_w", "dfc2_b"), res[1], res[0]):
        if a is not None:
            assert float((a - b).abs().max()) <= 2e-5 * (float(b.abs().max()) + 1e-30), name


@pytest.mark.parametrize("emb,B,H,T", [(16, 3, 4, 2), (64, 9, 37, 5), (400, 4, 50, 30)])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_category_gradient_by_counting_sort_matches_the_atomic_scatter(lib, monkeypatch, emb, B, H, T, dtype):
    """The category-table gradient has two implementations: one float atomic per (reference, column) inside the front-end
    backward, and (big batches) a counting sort of the references by category id + a gather-sum over runs.  Forced either way
    on the same inputs they agree to summation order; so do the other seven table gradients (which the sort does not touch)."""
    from news_recommendation_model_amd import config, ops, synth
    dims = config.Dims.for_emb(emb, 40)
    batch = synth.make_batch(dims, B, H, T, seed=emb + B, user_num=5, dtype=dtype)
    sd = synth.make_state_dict(dims, seed=2, user_num=5)
    inv = "invariant_interest_model."
    keys = ("category_embedding.0.weight", "sentiment_embedding.0.weight", "sentiment_embedding.0.bias", "type_embedding.0.weight",
            "year_embedding.0.weight", "month_embedding.0.weight", "day_embedding.0.weight", "hour_embedding.0.weight")
    xh, xt = torch.from_numpy(batch["x_history"]).cuda(), torch.from_numpy(batch["x_target"]).cuda()
    gen = torch.Generator(device="cuda").manual_seed(3)
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("NRM_FE_SORT", mode)
        tabs = [torch.from_numpy(sd[inv + k]).cuda().requires_grad_(True) for k in keys]
        lab_h, _, lab_t, _ = ops.frontend_pair(xh, xt, dims.n_subcat, dims.pca_vector, *tabs)
        if mode == "0":
            gh, gt = torch.randn(lab_h.shape, device="cuda", generator=gen), torch.randn(lab_t.shape, device="cuda", generator=gen)
        torch.autograd.backward([lab_h, lab_t], [gh, gt])
        res[mode] = [t_.grad for t_ in tabs]
        # one row set alone (the single-set op) through both implementations too
        tabs1 = [torch.from_numpy(sd[inv + k]).cuda().requires_grad_(True) for k in keys]
        lab1, _ = ops.frontend(xh, True, dims.n_subcat, dims.pca_vector, *tabs1)
        lab1.backward(gh)
        res[mode + "h"] = [t_.grad for t_ in tabs1]
    for a, b in zip(res["0"] + res["0h"], res["1"] + res["1h"]):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-5 * float(a.abs().max()) + 1e-12)
    assert float(res["1"][0].abs().max()) > 0


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("R,K,N", [(7, 3, 8), (3840, 3, 8), (1000, 4, 5), (65, 1, 1)])
def test_small_linear_relu_matches_torch(lib, R, K, N, dtype):
    """The instant-interest layer ReLU(Linear(3 -> 8)) as its own pair of kernels (reads float32 or the DataLoader's float64
    rows; the upstream gradient may be a column block of a wider matrix)."""
    from news_recommendation_model_amd import ops
    g = torch.Generator(device="cuda").manual_seed(R + K + N)
    x = torch.randn(R, K, device="cuda", generator=g).to(dtype)
    w = torch.randn(N, K, device="cuda", generator=g).requires_grad_(True)
    b = torch.randn(N, device="cuda", generator=g).requires_grad_(True)
    wide = torch.randn(R, N + 12, device="cuda", generator=g)
    y = ops.small_linear_relu(x, w, b)
    y.backward(wide[:, 4:4 + N])
    wr, br = w.detach().double().requires_grad_(True), b.detach().double().requires_grad_(True)
    yr = torch.relu(torch.nn.functional.linear(x.float().double(), wr, br))
    yr.backward(wide[:, 4:4 + N].double())
    assert torch.allclose(y.detach().double(), yr.detach(), rtol=1e-5, atol=1e-6)
    assert torch.allclose(w.grad.double(), wr.grad, rtol=1e-4, atol=1e-4 * float(wr.grad.abs().max()))
    assert torch.allclose(b.grad.double(), br.grad, rtol=1e-4, atol=1e-4 * float(br.grad.abs().max()))
    if dtype == torch.float32:
        _opcheck(torch.ops.nrm.small_linear_relu_fwd.default, (x, w, b))
    # shapes outside the small kernels fall back to linear + relu
    w9 = torch.randn(9, K, device="cuda", generator=g)
    assert torch.allclose(ops.small_linear_relu(x.float(), w9), torch.relu(torch.nn.functional.linear(x.float(), w9)), rtol=1e-5, atol=1e-5)


def test_loss_reads_strided_logits_and_float64_labels_and_passes_unit_gradients_through(lib):
    """The loss op reads the logits in place as column 0 of the padded [B*T, 4] GEMM output, takes the DataLoader's float64
    labels, writes dL/dout into the same padded layout, and -- seeded with ops.unit_grad -- hands its saved gradients on
    without multiplying them by one."""
    from news_recommendation_model_amd import ops
    B, T = 6, 9
    g = torch.Generator(device="cuda").manual_seed(11)
    buf = torch.randn(B * T, 4, device="cuda", generator=g)
    out_strided = buf[:, 0].view(B, T).requires_grad_(True)
    out_dense = out_strided.detach().contiguous().requires_grad_(True)
    delta = (0.1 * torch.randn(5, device="cuda", generator=g))
    label = torch.zeros(B, T, device="cuda", dtype=torch.float64)
    label[torch.arange(B), torch.arange(B)] = 1
    uid = torch.tensor([0, 1, 4, 2, 2, 3], device="cuda")
    res = []
    for out, y, seed in ((out_strided, label, True), (out_dense, label.float(), False)):
        d = delta.clone().requires_grad_(True)
        loss = ops.softmax_bce_loss(out, d, y, uid, 0.95)
        (gout, gd) = torch.autograd.grad(loss, [out, d], grad_outputs=ops.unit_grad(loss) if seed else None)
        res.append((loss.detach(), gout, gd))
    assert torch.allclose(res[0][0], res[1][0], rtol=1e-6, atol=0)          # (the loss is summed with float atomics: order varies)
    assert torch.equal(res[0][1], res[1][1]) and torch.allclose(res[0][2], res[1][2], rtol=1e-6, atol=1e-9)
    assert res[0][1].stride() == (4 * T, 4)                 # the padded layout the consuming GEMM streams
    # a non-unit upstream gradient still scales
    d = delta.clone().requires_grad_(True)
    loss = ops.softmax_bce_loss(out_dense, d, label, uid, 0.95)
    (g3,) = torch.autograd.grad(loss * 3.0, [d])
    assert torch.allclose(g3, 3.0 * res[1][2], rtol=1e-6, atol=1e-9)


def test_ops_called_through_torch_ops_match_the_python_entry_points(lib):
    from news_recommendation_model_amd import ops
    t, h, w1, b1, w2, b2 = _attn_args(grad=False)
    s_op, z = torch.ops.nrm.pwattn_fwd(t, h, w1, b1, w2, b2, False, 0)
    assert z.numel() == 0
    assert torch.equal(s_op, ops.pointwise_attention_scores(t, h, w1, b1, w2, b2))
    x = torch.randn(11, 12, device="cuda")
    w, b = torch.randn(8, 12, device="cuda"), torch.randn(8, device="cuda")
    y = torch.ops.nrm.linear_fwd(x, w, b, False)[0]
    assert torch.allclose(y, torch.nn.functional.linear(x, w, b), rtol=1e-5, atol=1e-5)
    with pytest.raises(RuntimeError):                                   # no CPU kernel exists for any nrm op
        torch.ops.nrm.linear_fwd(x.cpu(), w.cpu(), b.cpu(), False)


def test_modules_pickle_into_a_child_process_that_runs_the_ops(lib, tmp_path):
    """reference test.py:172-182: the CPU-resident model list is pickled to a child Process, which moves it to the device
    and runs it.  The child imports the package (registering the ops in ITS process) and must reproduce the parent's
    scores."""
    from golden_util import load_case
    from news_recommendation_model_amd import trainer
    case, dims, batch, sd, fx = load_case("tiny_eval")
    model = trainer.build_model(dims, int(batch["user_num"]), sd, device="cpu").eval()
    blob = tmp_path / "model.pkl"
    with open(blob, "wb") as f:
        pickle.dump({"model": model, "batch": {k: v for k, v in batch.items() if isinstance(v, np.ndarray)}}, f)
    code = (
        "import pickle, sys, numpy as np, torch\n"
        "sys.path.insert(0, %r)\n"
        "d = pickle.load(open(%r, 'rb'))\n"
        "m = d['model'].to('cuda').eval()\n"
        "b = {k: torch.from_numpy(v).cuda() for k, v in d['batch'].items() if v.ndim > 0}\n"
        "with torch.no_grad():\n"
        "    out = m(b['x_history'], b['x_target'], b['x_global'])\n"
        "np.save(%r, out.cpu().numpy())\n" % (str(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))),
                                                str(blob), str(tmp_path / "out.npy")))
    pr = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert pr.returncode == 0, pr.stderr.decode(errors="replace")[-2000:]
    got = np.load(tmp_path / "out.npy")
    assert np.abs(got - fx["r"]).max() / np.abs(fx["r"]).max() < 1e-3


def test_gate_block_and_concat_match_the_separate_ops(lib):
    """ops.gate_block (BatchNorm -> gate MLP -> product, one autograd node; the two gradients of the rows are joined inside the
    BatchNorm backward kernel) against the same computation from the separate ops and from plain PyTorch in fp64; and
    ops.concat_last against torch.cat."""
    from news_recommendation_model_amd import ops
    torch.manual_seed(3)
    R, N = 53, 40
    x0 = torch.randn(R, N, device="cuda")
    w1, b1 = 0.2 * torch.randn(N // 4, N, device="cuda"), 0.1 * torch.randn(N // 4, device="cuda")
    w2, b2 = 0.2 * torch.randn(N, N // 4, device="cuda"), 0.1 * torch.randn(N, device="cuda")
    gy = torch.randn(R, N, device="cuda")
    results = []
    for mode in ("fused", "separate", "torch64"):
        bn = torch.nn.BatchNorm1d(N).cuda().train()
        with torch.no_grad():
            bn.weight.copy_(torch.linspace(0.5, 1.5, N)); bn.bias.copy_(torch.linspace(-0.2, 0.2, N))
        dt = torch.float64 if mode == "torch64" else torch.float32
        if mode == "torch64":
            bn = bn.double()
        x = x0.to(dt).clone().requires_grad_(True)
        ps = [t.to(dt).clone().requires_grad_(True) for t in (w1, b1, w2, b2)]
        if mode == "fused":
            y = ops.gate_block(x, bn, *ps)
        elif mode == "separate":
            y = ops.mlp_gelu(ops.batch_norm(x, bn), *ps, mul=x)
        else:
            c = bn(x)
            y = torch.nn.functional.linear(torch.nn.functional.gelu(torch.nn.functional.linear(c, ps[0], ps[1])), ps[2], ps[3]) * x
        y.backward(gy.to(dt))
        results.append([y.detach().double(), x.grad.double(), bn.weight.grad.double(), bn.bias.grad.double(), bn.running_var.double()]
                       + [p.grad.double() for p in ps])
    for a, b, c in zip(*results):
        assert torch.allclose(a, c, rtol=1e-4, atol=1e-5)
        assert torch.allclose(b, c, rtol=1e-4, atol=1e-5)
    parts = [torch.randn(4, 6, 10, device="cuda", requires_grad=True), torch.randn(4, 6, 3, device="cuda", requires_grad=True),
             torch.randn(4, 6, 8, device="cuda")[..., :5].requires_grad_(True)]
    got = ops.concat_last(parts)
    want = torch.cat([p.detach() for p in parts], dim=-1)
    assert torch.equal(got, want)
    g = torch.randn_like(want)
    got.backward(g)
    assert torch.equal(parts[1].grad, g[..., 10:13]) and torch.equal(parts[2].grad, g[..., 13:18])


def test_deferred_reductions_refuse_a_weight_used_twice(lib):
    """inside ops.deferred_slab_reductions() a weight gradient is only valid after the flush.  A weight
    used twice in the graph makes autograd SUM two not-yet-reduced buffers into a new tensor the flush never reaches:
    FlatAdam.collect_grads must notice and raise instead of stepping on wrong gradients; with reductions run at once
    (train_step(..., defer_reductions=False) does this) the same graph gives the right gradient."""
    import torch
    from news_recommendation_model_amd import ops, trainer

    class Twice(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.fc = torch.nn.Linear(8, 8)

        def forward(self, x):
            return ops.linear(ops.linear(x, self.fc.weight, self.fc.bias), self.fc.weight, self.fc.bias)

    torch.manual_seed(0)
    mod = Twice().cuda()
    x = torch.randn(300, 8, device="cuda")
    opt = trainer.FlatAdam(mod)
    with ops.deferred_slab_reductions():
        mod(x).sum().backward()
    with pytest.raises(RuntimeError, match="deferred slab reductions"):
        opt.collect_grads()
    assert not ops._deferred["pending"]
    for p in mod.parameters():
        p.grad = None
    mod(x).sum().backward()                                     # immediate reductions: the reference gradient
    ref = torch.nn.Linear(8, 8).cuda()
    ref.load_state_dict({"weight": mod.fc.weight.detach().clone(), "bias": mod.fc.bias.detach().clone()})
    ref(ref(x)).sum().backward()
    assert torch.allclose(mod.fc.weight.grad, ref.weight.grad, rtol=1e-4, atol=1e-4)
    # a gradient that exists before backward(): train_step does not defer at all
    opt.zero_grad()


def test_packed_weight_cache_drops_dead_models_and_scopes_the_repack(lib):
    """entries of the packed-weight cache die with their parameter (no view keeps a dead model's storage
    alive), and FlatAdam.step() re-packs only its own parameters' images."""
    import gc
    import torch
    from news_recommendation_model_amd import ops, trainer
    ops.invalidate_packed_weights()
    a, b = torch.nn.Linear(16, 12).cuda(), torch.nn.Linear(16, 12).cuda()
    x = torch.randn(70, 16, device="cuda")
    ops.linear(x, a.weight, a.bias), ops.linear(x, b.weight, b.bias)
    n_two = len(ops._packs.entries)
    assert n_two >= 2
    assert ops.repack_persistent([a.weight, a.bias]) < ops.repack_persistent()       # scoped < everything
    del b
    gc.collect()
    assert len(ops._packs.entries) < n_two                       # b's entries went with b
    y = ops.linear(x, a.weight, a.bias)
    assert torch.allclose(y, torch.nn.functional.linear(x, a.weight, a.bias), rtol=1e-4, atol=1e-5)
