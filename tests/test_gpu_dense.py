"""GPU: the dense-layer kernels through the C ABI (nrm_gemm_pack / nrm_gemm_nt / nrm_gemm_tn) and the BatchNorm
column kernels, against float64 PyTorch on the same inputs.  Shapes cover the head of the model (K or N = 402,
not a multiple of 4 -> padded leading dimensions), tiny layers (3 -> 8), and ragged row counts."""
import numpy as np
import pytest
import torch

from golden_util import rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,K,N", [(300, 402, 1608), (257, 1608, 402), (64, 3, 8), (1000, 402, 1), (33, 66, 64),
                                   (5, 272, 68), (192, 400, 400), (700, 402, 400), (700, 400, 402), (3000, 1608, 402), (1300, 402, 1608)])
@pytest.mark.parametrize("gelu", [False, True])
def test_linear_forward_backward(lib, M, K, N, gelu):
    from news_recommendation_model_amd import ops
    g = torch.Generator(device="cpu").manual_seed(M * 7 + K + N)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / np.sqrt(K)
    b = torch.randn(N, generator=g) * 0.1
    gy = torch.randn(M, N, generator=g)
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    y_ref = torch.nn.functional.linear(xr, wr, br)
    if gelu:
        y_ref = torch.nn.functional.gelu(y_ref)
    y_ref.backward(gy.double())
    xg, wg, bg = (t.cuda().requires_grad_(True) for t in (x, w, b))
    y = ops.linear(xg, wg, bg, gelu=gelu)
    y.backward(gy.cuda())
    assert tuple(y.shape) == (M, N)
    assert rel_err(y.detach().cpu().numpy(), y_ref.detach().numpy()) < 1e-5
    assert rel_err(xg.grad.cpu().numpy(), xr.grad.numpy()) < 1e-5
    assert rel_err(wg.grad.cpu().numpy(), wr.grad.numpy()) < 1e-5
    assert rel_err(bg.grad.cpu().numpy(), br.grad.numpy()) < 1e-5


@pytest.mark.parametrize("M,K,N", [(5120, 1032, 258), (5120, 258, 1032), (3840, 264, 66), (300, 402, 1608), (257, 1608, 402), (33, 66, 64),
                                   (64, 3, 8), (700, 400, 402), (20000, 402, 400)])
def test_gemm_nt_stage_depths_are_bit_identical(lib, monkeypatch, M, K, N):
    """gemm_nt_kernel<..., KC>: one, two or four 16-wide K-chunks per LDS stage (round 5: built for launches of few workgroups -- the
    heads of C1 and of the reference's default sizes --, measured no faster there, kept behind NRM_NT_KC).  The chunks are contracted in the same order
    whatever the stage depth, so forward, input gradient and the GELU / GELU' epilogues must agree bit for bit with KC = 1 (the
    round-4 kernel, which the float64 comparisons of this file were written against); K / 16 not a multiple of KC, fewer chunks
    than one stage, a grid that would not pick KC by itself."""
    from news_recommendation_model_amd import ops
    g = torch.Generator(device="cpu").manual_seed(M + K + N)
    x = torch.randn(M, K, generator=g).cuda()
    w = (torch.randn(N, K, generator=g) / np.sqrt(K)).cuda()
    b = (torch.randn(N, generator=g) * 0.1).cuda()
    gy = torch.randn(M, N, generator=g).cuda()
    out = {}
    for kc in ("1", "2", "4", None):              # None: the default dispatch (KC = 1 since the measurement of round 5)
        if kc is None:
            monkeypatch.delenv("NRM_NT_KC", raising=False)
        else:
            monkeypatch.setenv("NRM_NT_KC", kc)
        res = []
        for gelu in (False, True):
            xg = x.clone().requires_grad_(True)
            y = ops.linear(xg, w, b, gelu=gelu)
            y.backward(gy)
            res += [y.detach().clone(), xg.grad.clone()]
        torch.cuda.synchronize()
        out[kc] = res
    for kc in ("2", "4", None):
        for a, r in zip(out[kc], out["1"]):
            assert torch.equal(a, r), (kc, float((a - r).abs().max()))
    ref = torch.nn.functional.linear(x.double(), w.double(), b.double())
    assert rel_err(out[None][0].cpu().numpy(), ref.cpu().numpy()) < 1e-5


@pytest.mark.parametrize("M,K,Hd,N", [(300, 1608, 402, 1608), (257, 1608, 402, 1), (64, 264, 66, 264), (33, 72, 18, 5),
                                      (7, 16, 4, 3), (1000, 256, 64, 1)])
def test_mlp_gelu_forward_backward(lib, M, K, Hd, N):
    """fc2(gelu(fc1 x)) as one node (models/attention_model.py:29-32): values and all five gradients against float64
    PyTorch; the backward goes through the GELU'-fused GEMM epilogue (NRM_EPI_DGELU)."""
    from news_recommendation_model_amd import ops
    g = torch.Generator(device="cpu").manual_seed(M + 3 * K + 5 * Hd + 7 * N)
    x = torch.randn(M, K, generator=g)
    w1 = torch.randn(Hd, K, generator=g) / np.sqrt(K)
    b1 = torch.randn(Hd, generator=g) * 0.1
    w2 = torch.randn(N, Hd, generator=g) / np.sqrt(Hd)
    b2 = torch.randn(N, generator=g) * 0.1
    gy = torch.randn(M, N, generator=g)
    ref = [t.double().requires_grad_(True) for t in (x, w1, b1, w2, b2)]
    F = torch.nn.functional
    y_ref = F.linear(F.gelu(F.linear(ref[0], ref[1], ref[2])), ref[3], ref[4])
    y_ref.backward(gy.double())
    dev = [t.cuda().requires_grad_(True) for t in (x, w1, b1, w2, b2)]
    y = ops.mlp_gelu(*dev)
    y.backward(gy.cuda())
    assert tuple(y.shape) == (M, N)
    assert rel_err(y.detach().cpu().numpy(), y_ref.detach().numpy()) < 1e-5
    for a, r, name in zip(dev, ref, ("x", "fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias")):
        assert rel_err(a.grad.cpu().numpy(), r.grad.numpy()) < 1e-5, name


@pytest.mark.parametrize("R,ni,nj,shape", [(1000, 30, 125, (2, 8)), (1000, 125, 30, (8, 2)), (999, 78, 30, (5, 2)), (1001, 30, 78, (2, 5)),
                                           (2051, 402, 1608, (2, 8)), (2051, 1608, 402, (8, 2)), (640, 400, 402, (5, 2)), (640, 402, 400, (2, 5)),
                                           (515, 400, 400, (5, 5)), (515, 64, 64, (4, 4)), (3, 402, 1608, (2, 8)), (130, 2, 1608, (2, 8))])
def test_gemm_tn_wave_tile_shapes(lib, R, ni, nj, shape):
    """dW = dY^T X on every wave-tile shape of gemm_tn (round 4: 2x8 / 8x2 / 5x2 / 2x5 beside 4x4 / 5x5, chosen by padding:
    csrc/gemm.hip tn_col) against float64: ragged widths, ragged row counts, several row splits, column sums."""
    from news_recommendation_model_amd import native, ops
    g = torch.Generator(device="cpu").manual_seed(R + 3 * ni + 5 * nj)
    a = ops._rows(torch.randn(R, ni, generator=g).cuda())
    b = ops._rows(torch.randn(R, nj, generator=g).cuda())
    # the plan really is the shape under test (16-column tiles: the fewest padded tiles win)
    t = lambda n, w: -(-(-(-n // 16)) // w) * w                     # noqa: E731
    cands = [(4, 4), (5, 5), (2, 8), (8, 2), (5, 2), (2, 5)]
    costs = [t(ni, k) * t(nj, d) for k, d in cands]
    assert cands[costs.index(min(costs))] == shape
    c, cs = ops._gemm_tn(a, b, True)
    torch.cuda.synchronize()
    ref = a.double().t() @ b.double()
    assert tuple(c.shape) == (ni, nj)
    assert rel_err(c.cpu().numpy(), ref.cpu().numpy()) < 1e-5
    assert rel_err(cs.cpu().numpy(), a.double().sum(0).cpu().numpy()) < 1e-5
    assert native.load().nrm_gemm_tn_nsplit(ni, nj, R, 0) >= 1


def test_linear_accepts_strided_and_3d_inputs(lib):
    from news_recommendation_model_amd import ops
    torch.manual_seed(0)
    x = torch.randn(4, 9, 70, device="cuda")[:, :, 3:69]            # non-contiguous, 66 features, offset start
    w = torch.randn(64, 66, device="cuda") * 0.1
    y = ops.linear(x, w, None)
    ref = torch.nn.functional.linear(x.double(), w.double())
    assert tuple(y.shape) == (4, 9, 64)
    assert rel_err(y.cpu().numpy(), ref.cpu().numpy()) < 1e-5


@pytest.mark.parametrize("R,N", [(1000, 264), (30720, 1608), (7, 8)])
@pytest.mark.parametrize("training", [True, False])
def test_batch_norm_matches_torch(lib, R, N, training):
    from news_recommendation_model_amd import ops
    torch.manual_seed(R + N)
    x = (torch.randn(R, N) * 3 + 5).cuda()                          # large mean: two-pass variance must hold up
    gy = torch.randn(R, N).cuda()
    ref = torch.nn.BatchNorm1d(N).cuda().double()
    mine = torch.nn.BatchNorm1d(N).cuda()
    with torch.no_grad():
        for m in (ref, mine):
            m.weight.copy_(torch.linspace(0.5, 1.5, N)); m.bias.copy_(torch.linspace(-1, 1, N))
            m.running_mean.copy_(torch.linspace(4, 6, N)); m.running_var.copy_(torch.linspace(8, 10, N))
    ref.train(training); mine.train(training)
    xr = x.double().requires_grad_(True)
    yr = ref(xr); yr.backward(gy.double())
    xm = x.clone().requires_grad_(True)
    ym = ops.batch_norm(xm, mine); ym.backward(gy)
    assert rel_err(ym.detach().cpu().numpy(), yr.detach().cpu().numpy()) < 1e-5
    assert rel_err(xm.grad.cpu().numpy(), xr.grad.cpu().numpy()) < 1e-4
    assert rel_err(mine.weight.grad.cpu().numpy(), ref.weight.grad.cpu().numpy()) < 1e-4
    assert rel_err(mine.bias.grad.cpu().numpy(), ref.bias.grad.cpu().numpy()) < 1e-4
    assert rel_err(mine.running_mean.cpu().numpy(), ref.running_mean.cpu().numpy()) < 1e-5
    assert rel_err(mine.running_var.cpu().numpy(), ref.running_var.cpu().numpy()) < 1e-5
    assert int(mine.num_batches_tracked) == int(ref.num_batches_tracked)


def test_packed_weight_cache_follows_the_weights(lib):
    """The packed GEMM operand of a parameter is cached (ops._PackedWeights): re-used while the weight is unchanged, re-packed when
    autograd's version counter moves (in-place update, load_state_dict), refreshed by ONE launch by ops.repack_persistent(), and
    droppable by hand after a write the counter cannot see (param.data...)."""
    from news_recommendation_model_amd import native, ops
    torch.manual_seed(0)
    lin = torch.nn.Linear(24, 10).cuda()
    x = torch.randn(7, 24, device="cuda")

    def run():
        native.kernel_events = []
        with torch.no_grad():
            y = ops.linear(x, lin.weight, lin.bias)
        packs = sum(1 for tag, _, _ in native.kernel_events if tag == "nrm_gemm_pack_multi")
        native.kernel_events = None
        return y, packs
    ref = lambda: torch.nn.functional.linear(x, lin.weight, lin.bias)      # noqa: E731
    ops.invalidate_packed_weights()
    y, n = run()
    assert n == 1 and torch.allclose(y, ref(), atol=1e-5)
    y, n = run()
    assert n == 0 and torch.allclose(y, ref(), atol=1e-5)                  # cached
    with torch.no_grad():
        lin.weight.mul_(2.0)                                               # version counter moves -> lazy re-pack
    y, n = run()
    assert n == 1 and torch.allclose(y, ref(), atol=1e-5)
    lin.weight.data.mul_(0.5)                                              # invisible to the counter ...
    ops.invalidate_packed_weights()                                        # ... so the caller says so
    y, n = run()
    assert n == 1 and torch.allclose(y, ref(), atol=1e-5)
    lin.weight.data.add_(1.0)
    assert ops.repack_persistent() >= 1                                    # what FlatAdam.step() does: one launch for all images
    y, n = run()
    assert n == 0 and torch.allclose(y, ref(), atol=1e-5)
    lin.load_state_dict({"weight": torch.ones(10, 24), "bias": torch.zeros(10)})
    y, n = run()
    assert n == 1 and torch.allclose(y, ref(), atol=1e-5)


def test_slab_reduce_multi_matches_single_launches_and_numpy(lib):
    """nrm_slab_reduce_multi: several slab sets (different shapes, strided destinations, a second signed destination, bias
    vectors) in one launch == the same sets through nrm_slab_reduce one by one == numpy; and ops' deferred mode records
    inside the context and produces the same gradients at the flush."""
    from news_recommendation_model_amd import native, ops
    rng = np.random.default_rng(3)
    sets = []
    for (nsplit, ni, nj, wide, second, vec) in [(5, 64, 64, 0, False, True), (37, 130, 33, 0, False, False), (160, 400, 400, 1600, True, True),
                                                (1, 7, 5, 0, False, True), (600, 64, 64, 256, True, False), (20, 402, 1608, 0, False, True)] * 5:
        ldws = (ni + 3) // 4 * 4
        ws = rng.standard_normal((nsplit, nj, ldws)).astype(np.float32)
        cs = rng.standard_normal((nsplit, ldws)).astype(np.float32) if vec else None
        sets.append((nsplit, ni, nj, ldws, wide, second, ws, cs))
    assert len(sets) > 24                                   # more than one table of the multi launch

    def run(mode):
        outs = []
        descs = (native.SlabDesc * len(sets))()
        keep = []
        for d, (nsplit, ni, nj, ldws, wide, second, ws, cs) in zip(descs, sets):
            w = torch.from_numpy(ws).cuda()
            ld = wide if wide else nj
            out = torch.zeros(ni, ld, device="cuda")
            o2 = out[:, nj:] if second else None                # a second column block of the same wide matrix
            v = torch.from_numpy(cs).cuda() if cs is not None else None
            vo = torch.empty(ni, device="cuda") if cs is not None else None
            keep.append((w, out, v, vo))
            if mode == "single":
                ops._slab_reduce(w, nsplit, nj, ldws, ni, out, ld, 1, out2=o2, out2_is=ld, out2_js=1, sign2=-1.0, vec=v, vec_out=vo)
            elif mode == "deferred":
                with ops.deferred_slab_reductions():
                    ops._slab_reduce(w, nsplit, nj, ldws, ni, out, ld, 1, out2=o2, out2_is=ld, out2_js=1, sign2=-1.0, vec=v, vec_out=vo)
            else:
                d.ws, d.nsplit, d.nj, d.ldws, d.ni = w.data_ptr(), nsplit, nj, ldws, ni
                d.out, d.out_istride, d.out_jstride = out.data_ptr(), ld, 1
                d.out2 = o2.data_ptr() if o2 is not None else None
                d.out2_istride, d.out2_jstride, d.sign2 = ld, 1, -1.0
                d.vec = v.data_ptr() if v is not None else None
                d.vec_out = vo.data_ptr() if vo is not None else None
            outs.append((out, vo))
        if mode == "deferred":
            assert len(ops._deferred["pending"]) == len(sets)
            assert float(outs[0][0].abs().max()) == 0.0      # recorded, not run
            ops.flush_slab_reductions()
            assert not ops._deferred["pending"]
        elif mode == "multi":
            native.call("nrm_slab_reduce_multi", descs, len(sets), native.stream_ptr())
        torch.cuda.synchronize()
        return [(o.cpu().numpy(), None if v is None else v.cpu().numpy()) for o, v in outs]

    single, multi, deferred = run("single"), run("multi"), run("deferred")
    for (nsplit, ni, nj, ldws, wide, second, ws, cs), (o1, v1), (o2, v2), (o3, v3) in zip(sets, single, multi, deferred):
        ref = ws.astype(np.float64).sum(0)[:, :ni].T          # [ni, nj]
        scale = np.abs(ref).max() + 1e-30
        for o in (o1, o2, o3):
            assert np.abs(o[:, :nj] - ref).max() <= 2e-6 * scale * max(1, nsplit) ** 0.5
            if second:
                assert np.abs(o[:, nj:2 * nj] + ref).max() <= 2e-6 * scale * max(1, nsplit) ** 0.5
            if wide:
                assert np.abs(o[:, (2 if second else 1) * nj:]).max() == 0.0      # nothing written outside the blocks
        if cs is not None:
            vref = cs.astype(np.float64).sum(0)[:ni]
            for v in (v1, v2, v3):
                assert np.abs(v - vref).max() <= 2e-6 * (np.abs(vref).max() + 1e-30) * max(1, nsplit) ** 0.5


@pytest.fixture
def dense_arithmetic():
    """ops.set_dense_arithmetic for one test, restored afterwards."""
    from news_recommendation_model_amd import ops
    prev = ops._default_dense_mma
    yield ops.set_dense_arithmetic
    ops._default_dense_mma = prev


# BASELINE config 2 names bf16: the dense layers then run on the bf16 matrix cores too (gemm_bf16.hip: resident-row forward /
# dX GEMM, split-row weight-gradient GEMM).  bf16x3 (hi/lo split operands) is held to 1e-4 against float64 -- well inside the
# fp32 gates it has to keep for the whole model; plain bf16 (one rounding per operand, 2^-9) is characterised at 3e-2.
@pytest.mark.parametrize("mma,tol", [("bf16x3", 1e-4), ("bf16", 3e-2)])
@pytest.mark.parametrize("M,K,N", [(300, 258, 1032), (257, 1032, 258), (64, 3, 8), (1000, 402, 1), (33, 66, 64), (5, 272, 68),
                                   (4099, 256, 256), (200, 1608, 402), (17, 40, 36), (51200, 64, 64), (15360, 258, 1032)])     # the last two: 128- and 64-row blocks
@pytest.mark.parametrize("gelu", [False, True])
def test_linear_forward_backward_on_bf16_matrix_cores(lib, dense_arithmetic, mma, tol, M, K, N, gelu):
    from news_recommendation_model_amd import ops
    dense_arithmetic(mma)
    g = torch.Generator(device="cpu").manual_seed(M * 7 + K + N)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / np.sqrt(K)
    b = torch.randn(N, generator=g) * 0.1
    gy = torch.randn(M, N, generator=g)
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    y_ref = torch.nn.functional.linear(xr, wr, br)
    if gelu:
        y_ref = torch.nn.functional.gelu(y_ref)
    y_ref.backward(gy.double())
    xg, wg, bg = (t.cuda().requires_grad_(True) for t in (x, w, b))
    y = ops.linear(xg, wg, bg, gelu=gelu)
    y.backward(gy.cuda())
    assert rel_err(y.detach().cpu().numpy(), y_ref.detach().numpy()) < tol
    assert rel_err(xg.grad.cpu().numpy(), xr.grad.numpy()) < tol
    assert rel_err(wg.grad.cpu().numpy(), wr.grad.numpy()) < tol
    # column sums stay fp32: exact without the activation; with GELU they sum dy * gelu'(z) of the bf16-rounded z
    assert rel_err(bg.grad.cpu().numpy(), br.grad.numpy()) < (tol if gelu else 1e-5)
    # and it is a different arithmetic from the fp32 path
    dense_arithmetic("f32")
    y32 = ops.linear(xg.detach(), wg.detach(), bg.detach(), gelu=gelu)
    if K >= 32 and N > 1:
        assert float((y32 - y.detach()).abs().max()) > 0


@pytest.mark.parametrize("M,K,Hd,N", [(300, 1032, 258, 1032), (257, 1032, 258, 1), (64, 264, 66, 264), (33, 72, 18, 5), (7, 16, 4, 3)])
def test_mlp_and_gate_block_on_bf16x3_matrix_cores(lib, dense_arithmetic, M, K, Hd, N):
    """fc2(gelu(fc1 x)) [* mul] with every GEMM (forward, dX with the fused GELU', dW) in bf16x3: all epilogues of gemm_nt_rx."""
    from news_recommendation_model_amd import ops
    dense_arithmetic("bf16x3")
    g = torch.Generator(device="cpu").manual_seed(M + 3 * K + 5 * Hd + 7 * N)
    x = torch.randn(M, K, generator=g)
    w1 = torch.randn(Hd, K, generator=g) / np.sqrt(K)
    b1 = torch.randn(Hd, generator=g) * 0.1
    w2 = torch.randn(N, Hd, generator=g) / np.sqrt(Hd)
    b2 = torch.randn(N, generator=g) * 0.1
    mul = torch.randn(M, N, generator=g)
    gy = torch.randn(M, N, generator=g)
    F = torch.nn.functional
    for with_mul in (False, True):
        if with_mul and N % 4:
            continue
        ref = [t.double().requires_grad_(True) for t in (x, w1, b1, w2, b2, mul)]
        y_ref = F.linear(F.gelu(F.linear(ref[0], ref[1], ref[2])), ref[3], ref[4])
        if with_mul:
            y_ref = y_ref * ref[5]
        y_ref.backward(gy.double())
        dev = [t.cuda().requires_grad_(True) for t in (x, w1, b1, w2, b2, mul)]
        y = ops.mlp_gelu(*dev[:5], mul=dev[5] if with_mul else None)
        y.backward(gy.cuda())
        assert rel_err(y.detach().cpu().numpy(), y_ref.detach().numpy()) < 1e-4
        names = ("x", "fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias") + (("mul",) if with_mul else ())
        for a, r, name in zip(dev, ref, names):
            assert rel_err(a.grad.cpu().numpy(), r.grad.numpy()) < 1e-4, (name, with_mul)
