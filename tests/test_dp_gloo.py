"""CPU, world_size 2, gloo: the data-parallel plumbing around the hot path:
impressions shard contiguously, ONE all-reduce of a single flat fp32 buffer averages the gradients,
and every replica ends the Adam step with identical parameters.  The per-shard gradients come from the
oracle here (the HIP path needs a GPU); on the GPU the same FlatGradReducer runs over RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from news_recommendation_model_amd import synth, trainer
from news_recommendation_model_amd.config import Dims


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out):
    from oracle import user_model_oracle as orc
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        dims = Dims.for_emb(64, category_label_num=40)
        B, H, T = 8, 6, 5
        batch = synth.make_batch(dims, B, H, T, seed=11)
        sd = synth.make_state_dict(dims, seed=1, user_num=int(batch["user_num"]))
        tb = {k: torch.from_numpy(v) for k, v in batch.items() if isinstance(v, np.ndarray) and v.ndim > 0}
        shard = trainer.shard_batch(tb, rank, world)
        assert shard["x_history"].shape[0] == B // world and shard["user_id"].shape[0] == B // world
        assert torch.equal(shard["x_target"], tb["x_target"][rank * 4:(rank + 1) * 4])

        # module with the product's parameter layout; gradients of this rank's shard from the oracle
        model = trainer.build_model(dims, int(batch["user_num"]), sd, device="cpu")
        p = orc.to_torch_params(sd)
        r = orc.user_model_forward(p, shard["x_history"], shard["x_target"], shard["x_global"], training=True)
        loss = orc.user_model_loss(p, shard["user_id"], r, shard["label"])
        names = [k for k, _ in model.named_parameters()]
        grads = torch.autograd.grad(loss, [p[k] for k in names], allow_unused=True)
        for (k, prm), g in zip(model.named_parameters(), grads):
            prm.grad = torch.zeros_like(prm) if g is None else g.detach().clone()
        local = {k: prm.grad.clone() for k, prm in model.named_parameters()}

        reducer = trainer.FlatGradReducer(model.parameters())
        assert reducer.nbytes == 4 * sum(q.numel() for q in model.parameters())
        calls = {"n": 0}
        real = dist.all_reduce

        def counting(*a, **kw):
            calls["n"] += 1
            return real(*a, **kw)
        dist.all_reduce = counting
        try:
            reducer.reduce()
        finally:
            dist.all_reduce = real
        assert calls["n"] == 1                          # exactly one collective per step

        # the reduced gradient is the mean over ranks
        gathered = [None] * world
        dist.all_gather_object(gathered, {k: v.numpy() for k, v in local.items()})
        for k, prm in model.named_parameters():
            want = sum(g[k] for g in gathered) / world
            np.testing.assert_allclose(prm.grad.numpy(), want, rtol=1e-6, atol=1e-9, err_msg=k)

        opt = trainer.make_optimizer(model)
        opt.step()
        flat = torch.cat([q.detach().reshape(-1) for q in model.parameters()])
        both = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(both, flat)
        assert torch.equal(both[0], both[1])           # replicas stay bit-identical after the step
        out.put((rank, "ok"))
    except Exception as e:                              # pragma: no cover
        out.put((rank, repr(e)))
        raise
    finally:
        dist.destroy_process_group()


def test_two_rank_gradient_allreduce_gloo():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(180)
    res = sorted(out.get(timeout=5) for _ in range(2))
    assert res == [(0, "ok"), (1, "ok")], res
    assert all(pr.exitcode == 0 for pr in procs)


def test_shard_batch_rejects_uneven_split():
    tb = {"user_id": torch.arange(5), "x_history": torch.zeros(5, 2, 3)}
    with pytest.raises(ValueError):
        trainer.shard_batch(tb, 0, 2)


def test_single_process_reducer_is_a_noop():
    lin = torch.nn.Linear(3, 2)
    lin.weight.grad = torch.ones_like(lin.weight)
    lin.bias.grad = torch.ones_like(lin.bias)
    trainer.FlatGradReducer(lin.parameters()).reduce()
    assert torch.equal(lin.weight.grad, torch.ones_like(lin.weight))


def _bench(args, env_extra, timeout=180):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=env, cwd=root,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)


@pytest.mark.skipif(torch.cuda.is_available(), reason="launcher plumbing on a box WITHOUT a GPU (the GPU leg is tests/test_gpu_dp.py)")
def test_bench_self_launcher_plumbing_without_gpu():
    """`bench.py --gpus N` starts its own ranks (no external torchrun).  Without a GPU every rank must refuse to
    run (there is no CPU path) and the parent must come back with a non-zero code instead of hanging."""
    pr = _bench(["--gpus", "2"], {})
    assert pr.returncode != 0 and b"device(s) visible" in pr.stderr                  # not enough devices, said so
    pr = _bench(["--gpus", "2", "--steps", "1", "--warmup", "0"], {"NRM_SINGLE_DEVICE": "1", "NRM_DIST_BACKEND": "gloo"})
    assert pr.returncode != 0 and b"needs an MI355X" in pr.stderr                   # children started and refused (the
    # launcher terminates the peers of the first rank that fails, so the message appears once or twice)


def test_bench_rejects_world_size_that_disagrees_with_gpus():
    pr = _bench(["--gpus", "2"], {"WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0"})
    assert pr.returncode != 0 and b"must agree" in pr.stderr


def test_launcher_deadline_terminates_hung_ranks():
    """bench.wait_for_ranks: ranks that never exit (a collective that never completes) are terminated -- then killed -- after
    the deadline, named on stderr, and the launcher returns 124 instead of polling until the driver's kill."""
    import io
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    try:
        import bench
    finally:
        sys.path.pop(0)
    sleeper = [sys.executable, "-c", "import time; time.sleep(120)"]
    stubborn = [sys.executable, "-c", "import signal, time; signal.signal(signal.SIGTERM, signal.SIG_IGN); time.sleep(120)"]
    quick = [sys.executable, "-c", "pass"]
    procs = [subprocess.Popen(quick), subprocess.Popen(sleeper), subprocess.Popen(stubborn)]
    time.sleep(0.5)                                    # let the stubborn rank install its handler
    log = io.StringIO()
    t0 = time.monotonic()
    rc = bench.wait_for_ranks(procs, timeout=1.0, grace=1.0, out=log)
    assert rc == 124 and time.monotonic() - t0 < 20
    assert all(pr.poll() is not None for pr in procs)
    text = log.getvalue()
    assert "ranks [1, 2] still running" in text and "rank 2 ignored SIGTERM" in text
    # a failing rank stops its peers and its code comes back
    procs = [subprocess.Popen([sys.executable, "-c", "import sys; sys.exit(3)"]), subprocess.Popen(sleeper)]
    log = io.StringIO()
    assert bench.wait_for_ranks(procs, timeout=30.0, out=log) == 3
    assert procs[1].poll() is not None and "rank 0 exited with code 3" in log.getvalue()
