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


def _import_bench():
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    try:
        import bench
    finally:
        sys.path.pop(0)
    return bench


def test_multi_rank_runs_step_eagerly_unless_graph_is_asked_for():
    """with a process group the DEFAULT launch mode is eager (an abort on ProcessGroupNCCL's watchdog thread
    during a capture never reaches an `except`, and the first multi-device contact of this code is the driver's scaling run);
    the captured step with the RCCL all-reduce inside is opt-in.  One process without a group picks by step time ("rule")."""
    bench = _import_bench()
    mode, rule = bench.launch_policy(graph=False, eager=False, use_dist=True, backend="nccl")
    assert mode == "eager" and "opt-in" in rule and "--graph" in rule
    assert bench.launch_policy(True, False, True, "nccl")[0] == "graph"
    assert bench.launch_policy(False, True, True, "nccl")[0] == "eager"
    assert bench.launch_policy(False, False, True, "gloo")[0] == "eager"
    assert bench.launch_policy(False, False, False, "nccl")[0] == "rule"            # by step time; both modes probed after the timed region
    assert bench.launch_policy(True, False, False, "nccl")[0] == "graph"
    assert bench.launch_policy(False, True, False, "nccl")[0] == "eager"
    with pytest.raises(SystemExit):
        bench.launch_policy(True, False, True, "gloo")         # gloo cannot be captured: refused, not silently eager
    with pytest.raises(SystemExit):
        bench.launch_policy(True, True, False, "nccl")


def test_heartbeat_fires_on_no_progress_only_and_names_the_phase():
    """the in-process deadline is a hang detector, not a whole-run limit.  A rank that keeps finishing steps is never
    ended however long it runs; one that stops making progress leaves with 124 and says in which phase."""
    import io
    import time
    bench = _import_bench()
    codes, log = [], io.StringIO()
    hb = bench.Heartbeat(1.0, rank=3, poll=0.05, out=log, _exit=codes.append)      # (generous against a loaded test machine)
    t0 = time.monotonic()
    while time.monotonic() - t0 < 3.0:                  # three deadlines' worth of healthy progress
        hb.beat("timed steps")
        time.sleep(0.05)
    assert codes == [] and hb.beats > 10
    hb.beat("final barrier")
    t1 = time.monotonic()
    while not codes and time.monotonic() - t1 < 10.0:    # ... and now a hang: it fires once the deadline has passed
        time.sleep(0.05)
    assert codes == [124] and time.monotonic() - t1 >= 0.9
    text = log.getvalue()
    assert "rank 3" in text and "'final barrier'" in text and "without progress" in text
    # a stopped heartbeat never fires (the final print of rank 0 may take long: fwd_auc_parity, cpu_baseline)
    codes2 = []
    hb2 = bench.Heartbeat(0.2, poll=0.05, out=io.StringIO(), _exit=codes2.append)
    hb2.stop()
    time.sleep(0.6)
    assert codes2 == []


def _line_worker(rank, world, port, out):
    """One rank of a `bench.py --gpus 2` run as far as a box without a GPU can take it: the process-group half of main() -- device
    identities, launch policy, per-rank times, replica check, the job's fields of the JSON line -- on CPU tensors over gloo."""
    import types
    bench = _import_bench()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cpu")
        ident = {"rank": rank, "current_device": rank, "name": "AMD Instinct MI355X", "uuid": f"GPU-{rank:04d}", "pci": f"0000:{rank + 5:02x}:00"}
        devices, distinct = bench.gather_device_identities(ident, world)
        assert distinct and [d["rank"] for d in devices] == list(range(world))
        same, not_distinct = bench.gather_device_identities(dict(ident, uuid="GPU-0000", pci="0000:05:00"), world)
        assert not not_distinct                                               # two ranks on one device are seen
        mode, rule = bench.launch_policy(False, False, True, "nccl")          # what the 8-GPU run decides
        assert mode == "eager"
        opt = types.SimpleNamespace(nbytes=21_170_420, collective_events=None, flat_param=torch.arange(16, dtype=torch.float32))
        elapsed = 0.632 + 0.010 * rank                                       # rank 1 is the straggler
        per_rank_ms, job_elapsed = bench.gather_rank_times(elapsed, 20, world, dev)
        assert len(per_rank_ms) == world and abs(job_elapsed - 0.642) < 1e-12
        in_sync = bench.replicas_hold_identical_weights(opt.flat_param)
        assert in_sync
        drift = opt.flat_param + (1e-3 if rank == 1 else 0.0)
        assert not bench.replicas_hold_identical_weights(drift)             # a replica that drifted is seen
        job = bench.job_fields(world, 1024, 20, job_elapsed, per_rank_ms, opt, True, devices, in_sync, None)
        out.put((rank, json_safe(job)))
    except Exception as e:                              # pragma: no cover
        out.put((rank, repr(e)))
        raise
    finally:
        dist.destroy_process_group()


def json_safe(obj):
    import json
    return json.loads(json.dumps(obj))


def test_two_rank_bench_line_job_fields_gloo():
    """The JSON line of a multi-rank run carries n_gpus == collective.world_size == N, devices_distinct, every rank's own
    ms_per_step, the collective's statistics keys, and the whole-job value over the SLOWEST rank's time."""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_line_worker, args=(r, 2, port, out)) for r in range(2)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(180)
    res = dict(out.get(timeout=5) for _ in range(2))
    assert all(isinstance(v, dict) for v in res.values()), res
    for job in res.values():                                               # every rank assembles the same job description
        assert job["n_gpus"] == 2 and job["collective"]["world_size"] == 2 and job["collective"]["backend"] == "gloo"
        assert job["collective"]["devices_distinct"] is True and len(job["collective"]["devices"]) == 2
        assert job["collective"]["all_reduce_per_step"] == 1
        assert {"allreduce_ms", "bus_GBps"} <= set(job["collective"])       # collective_stats (None here: no GPU events)
        pr_ = job["per_rank_ms_per_step"]
        assert pr_["ranks"] == [31.6, 32.1] and pr_["min"] == 31.6 and pr_["max"] == 32.1
        assert job["ms_per_step"] == 32.1                                   # the slowest rank's time is the job's
        assert abs(job["value"] - 2 * 1024 * 20 / 0.642) < 0.01
        assert job["grad_allreduce_bytes"] == 21_170_420 and job["replicas_in_sync"] is True
    assert res[0] == res[1]
    assert all(pr.exitcode == 0 for pr in procs)
