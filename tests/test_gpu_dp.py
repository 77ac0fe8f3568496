"""GPU: the data-parallel step of BASELINE config 4 on the HIP path (insertion point of the collective:
reference train.py:73-74).  Two ranks, both on cuda:0 of the 1-GPU test box, gloo process group (RCCL needs one
device per rank; the 8-GPU run of bench.py uses backend nccl over the same FlatAdam.all_reduce_grads).  The assertions
live in tests/dp_worker.py; here the two rank processes are started as children (nothing is exec'ed over this
process, which has initialised the GPU) and their verdicts collected."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run_ranks(cmd_for_rank, world, tmp_path, timeout=420, extra_env=None):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), NRM_DIST_BACKEND="gloo", NRM_SINGLE_DEVICE="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.update(extra_env or {})
        procs.append(subprocess.Popen(cmd_for_rank(r), env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    for pr in procs:
        try:
            out, _ = pr.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out.decode(errors="replace"))
    return procs, outs


def test_two_rank_hip_train_step_one_collective_identical_replicas(lib, tmp_path):
    world = 2
    paths = [str(tmp_path / f"rank{r}.json") for r in range(world)]
    procs, outs = _run_ranks(lambda r: [sys.executable, os.path.join(HERE, "dp_worker.py"), paths[r]], world, tmp_path)
    for r in range(world):
        assert os.path.exists(paths[r]), outs[r][-2000:]
        res = json.load(open(paths[r]))
        assert res["ok"], res.get("error", outs[r][-2000:])
        assert res["bn_stats_differ"]                       # per-replica BatchNorm, as DESIGN.md §6 states
        assert res["grad_max_rel_err_vs_oracle"] < 1e-2
    assert all(pr.returncode == 0 for pr in procs)


def test_bench_self_launches_two_ranks(lib, tmp_path):
    """`python bench.py --gpus 2` with no launcher around it: bench starts its own rank processes and rank 0 prints the
    one JSON line (the shape of command the driver runs for N > 1; here both ranks share cuda:0 over gloo)."""
    env = dict(os.environ, NRM_SINGLE_DEVICE="1", NRM_DIST_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    pr = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                         "--workload", "ref-default", "--batch", "16", "--no-cpu-baseline"],
                        env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=420)
    assert pr.returncode == 0, pr.stderr.decode(errors="replace")[-3000:]
    lines = [ln for ln in pr.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, pr.stdout.decode()[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 32 and line["scaling"] == "weak"
    assert line["replicas_in_sync"] is True
    assert line["grad_allreduce_bytes"] > 0
    assert line["collective"]["world_size"] == 2 and line["collective"]["all_reduce_per_step"] == 1
    # the step's one collective is timed (event pair) and priced as bus bandwidth; every rank's device is
    # reported, and here -- the one-GPU rehearsal -- they are known NOT to be distinct
    c = line["collective"]
    assert c["allreduce_ms"] > 0 and c["allreduce_events"] == 3 and c["bus_GBps"] > 0
    assert abs(c["bus_GBps"] - 2 * (2 - 1) / 2 * line["grad_allreduce_bytes"] / (c["allreduce_ms"] * 1e-3) / 1e9) < 0.02 * c["bus_GBps"] + 0.01
    assert len(c["devices"]) == 2 and {d["rank"] for d in c["devices"]} == {0, 1} and c["devices_distinct"] is False
    # round 4: every rank's own time (stragglers), and gloo cannot be captured -> the launch-mode probe leaves the step eager
    pr_ = line["per_rank_ms_per_step"]
    assert len(pr_["ranks"]) == 2 and pr_["min"] <= pr_["max"] <= line["ms_per_step"] * 1.001 and line["config"]["launch"] == "eager"


def test_bench_refuses_ranks_that_share_a_device(lib, tmp_path):
    """Two ranks that land on the SAME device without NRM_SINGLE_DEVICE=1 (a launcher that forgot LOCAL_RANK): every rank
    gathers the device identities and refuses to produce a number."""
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   NRM_DIST_BACKEND="gloo")
        env.pop("NRM_SINGLE_DEVICE", None)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                                       "--workload", "ref-default", "--batch", "8", "--no-cpu-baseline"],
                                      env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [pr.communicate(timeout=300)[0].decode(errors="replace") for pr in procs]
    assert all(pr.returncode != 0 for pr in procs)
    assert any("not distinct" in o for o in outs), outs[0][-1500:]


def test_bench_rank_leaves_with_124_when_a_peer_never_shows_up(lib, tmp_path):
    """A rank started by an EXTERNAL launcher (as the driver does for N > 1) whose peer never joins: the in-process deadline
    (--timeout) ends it with exit code 124 and a message instead of holding the node in a rendezvous / collective forever."""
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               NRM_DIST_BACKEND="gloo", NRM_SINGLE_DEVICE="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    pr = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--timeout", "15",
                         "--workload", "ref-default", "--batch", "8", "--no-cpu-baseline"],
                        env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240)
    assert pr.returncode == 124, (pr.returncode, pr.stderr.decode(errors="replace")[-1500:])
    assert "still running after 15 s" in pr.stderr.decode(errors="replace")


def _bench_line(env, *extra):
    pr = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                         "--workload", "ref-default", "--batch", "16", "--no-cpu-baseline", *extra],
                        env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=420)
    assert pr.returncode == 0, pr.stderr.decode(errors="replace")[-3000:]
    lines = [ln for ln in pr.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, pr.stdout.decode()[-2000:]
    return json.loads(lines[0])


def test_bench_one_rank_through_rccl(lib, tmp_path):
    """RCCL itself under the data-parallel step, as far as a one-GPU box can take it: a process group of ONE rank with
    backend nccl (NRM_DIST_WORLD1=1), so the communicator, the in-stream all-reduce of the flat gradient (ReduceOp.AVG),
    the barrier-bracketed timing and the replica check of bench.py all run through librccl on the code path of N > 1 --
    eagerly (forced, and as the default of any run with a process group) and as a CAPTURED step with the all-reduce inside
    the HIP graph (`--graph`, opt-in since round 5)."""
    env = dict(os.environ, NRM_DIST_WORLD1="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "NRM_DIST_BACKEND", "NRM_SINGLE_DEVICE"):
        env.pop(k, None)
    line = _bench_line(env, "--eager")
    assert line["collective"]["backend"] == "nccl" and line["collective"]["world_size"] == 1
    assert line["grad_allreduce_bytes"] > 0 and line["replicas_in_sync"] is True
    assert line["config"]["launch"] == "eager" and line["collective"]["allreduce_events"] == 3
    assert line["per_rank_ms_per_step"]["min"] == line["per_rank_ms_per_step"]["max"] == line["per_rank_ms_per_step"]["ranks"][0]
    # the captured step with the in-stream all-reduce inside the graph
    env["MASTER_PORT"] = str(_free_port())
    graphed = _bench_line(env, "--graph")
    assert graphed["config"]["launch"] == "hipGraph replay" and graphed["collective"]["backend"] == "nccl"
    assert graphed["replicas_in_sync"] is True and graphed["first_step_loss"] == line["first_step_loss"]
    assert graphed["collective"]["allreduce_timed_in"].startswith("3 eager") and graphed["collective"]["allreduce_ms"] > 0
    # the default WITH a process group: eager, not probed -- the captured step with RCCL inside is opt-in
    env["MASTER_PORT"] = str(_free_port())
    dflt = _bench_line(env)
    assert dflt["config"]["launch"] == "eager" and dflt["launch_probe"]["steps"] == 0
    assert "opt-in" in dflt["launch_probe"]["rule"] and dflt["replicas_in_sync"] is True
    assert dflt["collective"]["allreduce_timed_in"] == "timed region" and dflt["collective"]["allreduce_events"] == 3
    # the one-rank average must leave the step itself unchanged: same losses as the run without a process group
    env.pop("NRM_DIST_WORLD1")
    line2 = _bench_line(env, "--eager")
    assert line2["collective"] is None
    assert abs(line["loss"] - line2["loss"]) <= 1e-5 * max(1.0, abs(line2["loss"]))
    assert abs(line["first_step_loss"] - line2["first_step_loss"]) <= 1e-6 * max(1.0, abs(line2["first_step_loss"]))


def test_bench_default_launch_rule_and_probe_after_the_timed_region(lib, tmp_path):
    """One process, no flags (what the driver runs): the launch mode comes from the step time (this small shape is launch-bound ->
    hipGraph replay), and BOTH modes are timed after the timed region and reported -- so the W warm-up + K timed steps of the
    contract start without 20 probe steps and four captures in front of them (round 5)."""
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "NRM_DIST_BACKEND", "NRM_SINGLE_DEVICE", "NRM_DIST_WORLD1"):
        env.pop(k, None)
    pr = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "2", "--workload", "ref-default",
                         "--batch", "32", "--no-cpu-baseline"],
                        env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=420)
    assert pr.returncode == 0, pr.stderr.decode(errors="replace")[-3000:]
    d = json.loads([ln for ln in pr.stdout.decode().splitlines() if ln.startswith("{")][0])
    pb = d["launch_probe"]
    assert d["config"]["launch"] == "hipGraph replay" and pb["t_rule_ms"] < 10
    assert pb["steps"] >= 10 and pb["t_eager_ms"] > 0 and pb["t_graph_ms"] > 0 and pb["when"].startswith("after the timed region")
    assert d["steps"] == 4 and d["warmup"] == 2 and d["build"]["match"] is True
    assert d["config"]["weight_gradient_stream"] is False


def test_bench_line_carries_the_contract_fields(lib, tmp_path):
    """`python bench.py` (N = 1) prints ONE JSON line with the fields the driver and the judge read: metric / value / unit /
    n_gpus / steps / warmup / ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data / config.workload, plus
    `roofline` (bound, achieved, peak, unit, frac, traffic) and `cpu_baseline` (value, unit, cores, kind, sample).  Run on a
    small shape with a short CPU sample; the numbers themselves are not judged here."""
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "NRM_DIST_BACKEND", "NRM_SINGLE_DEVICE", "NRM_DIST_WORLD1"):
        env.pop(k, None)
    pr = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "2", "--workload", "ref-default",
                         "--batch", "32", "--eager", "--cpu-seconds", "2"],
                        env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=420)
    assert pr.returncode == 0, pr.stderr.decode(errors="replace")[-3000:]
    lines = [ln for ln in pr.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["vs_baseline"] is None and d["scaling"] == "weak" and "synthetic" in d["data"]
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 32 / (d["ms_per_step"] * 1e-3)) <= 1e-3 * d["value"]      # whole-job impressions per second
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and r["peak"] > 0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and "traffic" in r
    # round 4: the whole-step roofline, the first step's loss next to the last one's, the resident batches the steps cycle through
    st = r["step"]
    assert st["matrix_flops_per_step"] == st["contraction_flops"] + st["dense_flops"] > 0 and st["contraction_launches"] == 5
    assert abs(st["frac"] - st["matrix_flops_per_step"] / (d["ms_per_step"] * 1e-3) / 1e12 / st["peak"]) < 1e-3
    assert d["config"]["resident_batches_per_gpu"] == 4 and d["first_step_loss"] > 0 and "loss" in d
    chk = d["cpu_baseline"]["hip_vs_oracle_first_step"]
    assert chk["loss_rel_err"] < 1e-3 and chk["logit_max_rel_err"] < 1e-3 and chk["oracle_loss"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["sample"] and c["unit"] == d["unit"]
