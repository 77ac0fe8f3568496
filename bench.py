#!/usr/bin/env python3
"""bench.py -- train impressions/sec of the UserModel hot path on synthetic EBNeRD-shaped batches.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload C3-large]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is the reference's train.py:66-75 on one batch: forward -> loss -> backward -> [one gradient
all-reduce when N > 1] -> Adam(lr 1e-3, weight_decay 1e-5) -> zero_grad, train-mode BatchNorm, inputs
resident in HBM as fp32.  Every rank holds its own batch of B impressions (weak scaling: global batch
N*B, BASELINE config 4 at N=8).  Rank 0 prints ONE JSON line.

Extra objects on the line:
  roofline      the dominant HIP kernel (largest summed time in the timed region): algorithmic FLOPs per
                launch (2*B*T*H*D^2, the contraction only) / its mean launch duration measured
                with HIP events on the launch stream, against the dense fp32 MFMA peak of gfx950.  Inside the timed
                region only the four big attention kernels are bracketed by events; the rest of the `kernels`
                table is measured on 3 extra untimed steps after the warm-up (an event pair per launch on all ~600 launches of a
                step costs 1.5 % at C3 and 3-4x on the small shapes).
  cpu_baseline  the oracle (PyTorch-CPU restatement of the reference-literal algorithm) timed on this box's
                host cores on a bounded sample of the same workload (reduced B), rank 0, N=1 only.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# this pool's host driver only supports dmabuf IPC: without it RCCL's buffer exchange between rank processes fails with
# hipIpcGetMemHandle: invalid argument.  Set before anything initialises the GPU (also for ranks an external launcher started).
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np
import torch
import torch.distributed as dist

FP32_MFMA_PEAK_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_* dense peak
BF16_MFMA_PEAK_TFLOPS = 2500.0    # same guide: dense bf16 MFMA (the 5 PF headline figure includes 2:1 sparsity)
HBM_PEAK_GBS = 8000.0             # same guide: HBM3E spec peak (6.3 TB/s measured achievable)
# (workload, dtype) -> committed rocprofv3 --pmc summary of the same bench command (scripts/collect_profiles.sh + pmc_summary.py)
PMC_TRAFFIC_FILES = {("C3-large", "f32"): "r5_c3_pmc_traffic.json", ("C5-long", "f32"): "r5_c5_pmc_traffic.json",
                     ("C2-small", "bf16x3"): "r5_c2_bf16x3_pmc_traffic.json", ("C1-demo", "f32"): "r5_c1_pmc_traffic.json",
                     ("ref-default", "f32"): "r5_refdefault_pmc_traffic.json"}
INFER_TRAFFIC_FILES = {("C3-large", "f32"): "r5_infer_c3_pmc_traffic.json"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="C3-large")
    ap.add_argument("--mode", default="train", choices=["train", "infer"],
                    help="train (default): the headline metric, one step = reference train.py:66-75.  infer: the OTHER caller of "
                         "the hot path, reference test.py:31-74 model_test -- eval-mode BatchNorm, no saved pre-activation, "
                         "torch.no_grad, softmax / de-padding glue of evaluation.predict; one step = one batch")
    ap.add_argument("--dtype", default=None, choices=["f32", "bf16", "bf16x3"],
                    help="arithmetic of the attention contractions (everything else is fp32): f32 = fp32 MFMA (default, BASELINE "
                         "config 3); bf16x3 = bf16 MFMA on hi/lo split operands, fp32-class accuracy (default for --workload "
                         "C2-small, BASELINE config 2); bf16 = plain bf16 operands (misses the 1e-3 parity gate on raw scores)")
    ap.add_argument("--batch", type=int, default=None, help="override per-GPU batch")
    ap.add_argument("--graph", action="store_true",
                    help="capture the step in a HIP graph and time graph replays (per-kernel event timing then comes from 3 extra "
                         "eager steps outside the timed region).  With --gpus N > 1 (RCCL) this is OPT-IN: the captured step then "
                         "contains the in-stream all-reduce")
    ap.add_argument("--eager", action="store_true",
                    help="never replay a graph.  Default (neither flag), ONE process: hipGraph replay when an eager step takes < 10 ms, "
                         "eager stepping above; both modes are then timed over --probe-steps steps AFTER the timed region and reported "
                         "(launch_probe).  Default with a process group (--gpus N > 1): eager (launch_policy)")
    ap.add_argument("--batches", type=int, default=4,
                    help="resident synthetic batches per rank the steps cycle through (all in HBM before the timed region starts)")
    ap.add_argument("--probe-steps", type=int, default=10, help="steps per launch mode in the eager-vs-hipGraph probe (>= 10)")
    ap.add_argument("--no-probe", action="store_true", help="skip the probe of both launch modes that follows the timed region")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true",
                    help="(default since round 3) no HIP event pairs inside the timed region: the per-kernel durations and the "
                         "roofline come from 3 extra eager ONE-STREAM steps outside it, as with --graph")
    ap.add_argument("--timed-kernel-events", action="store_true",
                    help="eager mode only: bracket the heavy attention kernels with HIP events INSIDE the timed region (the "
                         "round-1/2 behaviour); the step then runs on one stream, i.e. without the weight-gradient stream")
    ap.add_argument("--pcie", action="store_true",
                    help="also time the steps with the float64 host batch copied to HBM every step (as the reference's "
                         "DataLoader + .to(device) does), prefetched one batch ahead on a copy stream; reported as "
                         "'pcie_inclusive', never as 'value'")
    ap.add_argument("--timeout", type=float, default=900.0,
                    help="--gpus N > 1: seconds WITHOUT PROGRESS (no step finished, no phase change: Heartbeat) after which a rank "
                         "exits with code 124 and names the phase it hung in; without an external launcher the parent applies the "
                         "same figure to the whole run, terminates, then kills, the rank processes and names the ranks that hung")
    ap.add_argument("--cpu-batch", type=int, default=None)
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


def host_cores():
    """CPUs this process may really use: the cgroup quota if there is one (the GPU box exposes 256 logical
    CPUs but grants a 16-CPU share; 256 threads on 16 CPUs ran the oracle 50x slower), else the affinity."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    quota_found = False
    for path, two in (("/sys/fs/cgroup/cpu.max", True), ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", False)):
        try:
            txt = open(path).read().split()
            if two:
                quota, period = txt[0], txt[1]
            else:
                quota, period = txt[0], open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().split()[0]
            if quota not in ("max", "-1"):
                n = min(n, max(1, int(int(quota) / int(period))))
                quota_found = True
                break
        except Exception:
            pass
    if not quota_found and n > 32:
        n = 16          # a 1-GPU box of this pool grants a 16-CPU share of its 256 logical CPUs
    env = os.environ.get("NRM_CPU_THREADS")
    if env:
        n = int(env)
    return max(1, min(n, 64))


def cpu_baseline(dims, wl, seconds, cpu_batch, mma="f32"):
    """Oracle train step (reference-literal: materialises the [B,T,H,4D] concat) on the host cores."""
    from news_recommendation_model_amd import synth
    from oracle import user_model_oracle as orc
    cores = host_cores()
    torch.set_num_threads(cores)
    # bounded sample: concat bytes = B*T*H*4D*4 per attention; keep it around 0.25 GB
    per_imp = wl["T"] * wl["H"] * 4 * wl["emb"] * 4
    Bc = cpu_batch or int(max(2, min(wl["B"], (256 << 20) // per_imp)))
    batch = synth.make_batch(dims, Bc, wl["H"], wl["T"], seed=0)
    sd = synth.make_state_dict(dims, seed=1, user_num=int(batch["user_num"]), perturb=False)
    p = orc.to_torch_params(sd)
    tb = {k: torch.from_numpy(v) for k, v in batch.items() if isinstance(v, np.ndarray) and v.ndim > 0}
    tb = {k: (v.float() if v.is_floating_point() else v) for k, v in tb.items()}
    st = {"step": 0, "m": {}, "v": {}}
    # the same first step through the HIP path (same weights, same batch): the checker's other job
    check = None
    try:
        from news_recommendation_model_amd import trainer
        model = trainer.build_model(dims, int(batch["user_num"]), sd, device="cuda", attention_mma=mma).train()
        gb = trainer.batch_to_device(batch, "cuda")
        out = model(gb["x_history"], gb["x_target"], gb["x_global"])
        gl = model.loss(gb["user_id"], out, gb["label"])
        gl.backward()
        g_dev = model.invariant_interest_model.label_attention.mlp.fc1.weight.grad.detach().cpu()
        r_dev, l_dev = out.detach().cpu(), float(gl.detach())
        del model, gb, out, gl
    except Exception as e:                          # never let the check break the throughput line
        check = {"error": repr(e)}
    l_cpu, r_cpu, g_cpu = orc.train_step(p, st, tb)                       # warm-up (and the reference values)
    if check is None:
        gk = g_cpu["invariant_interest_model.label_attention.mlp.fc1.weight"]
        check = {"logit_max_rel_err": float((r_dev - r_cpu).abs().max() / r_cpu.abs().max()),
                 "oracle_loss": round(float(l_cpu), 6), "hip_loss": round(l_dev, 6),
                 "loss_rel_err": abs(l_dev - float(l_cpu)) / abs(float(l_cpu)),
                 "attention_fc1_grad_max_rel_err": float((g_dev - gk).abs().max() / gk.abs().max())}
    n, t0 = 0, time.perf_counter()
    while True:
        orc.train_step(p, st, tb)
        n += 1
        el = time.perf_counter() - t0
        if (el >= seconds and n >= 3) or el > 3 * seconds:
            break
    return {"value": round(Bc * n / el, 2), "unit": "impressions/s", "cores": cores, "kind": "port",
            "sample": f"oracle/user_model_oracle.train_step, B={Bc} H={wl['H']} T={wl['T']} D={wl['emb']}, "
                      f"{n} steps after 1 warm-up, torch CPU fp32 {cores} threads",
            "hip_vs_oracle_first_step": check}


def fwd_auc_parity(dev, mma="f32", case_name="c3_large"):
    """The 'fwd AUC parity' half of the headline metric: forward of a golden case at the benchmarked shape's dimensions
    (c3_large: B=2, H=50, T=30, D=400; c2_small: B=2, H=32, T=30, D=256; outputs of the REFERENCE model,
    tests/golden/*.npz) through the HIP path IN THE ARITHMETIC BEING BENCHMARKED; max relative error of the logits and max
    |AUC difference| per impression against the fixture."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    try:
        from golden_util import load_case
        from news_recommendation_model_amd import evaluation, trainer
        case, dims, batch, sd, fx = load_case(case_name)
        model = trainer.build_model(dims, int(batch["user_num"]), sd, device=dev, attention_mma=mma).train()
        tb = trainer.batch_to_device(batch, dev)
        with torch.no_grad():
            out = model(tb["x_history"], tb["x_target"], tb["x_global"])
        auc, _ = evaluation.row_auc_top1(out, tb["label"])
        r = out.cpu().numpy()
        return {"case": f"{case_name} (reference fixture), attention arithmetic {mma}", "logit_max_rel_err": float(np.abs(r - fx["r"]).max() / np.abs(fx["r"]).max()),
                "auc_max_abs_diff": float(np.abs(auc.cpu().numpy() - fx["auc"]).max())}
    except Exception as e:                      # never let the parity probe break the throughput line
        return {"error": repr(e)}
    finally:
        sys.path.pop(0)


def collective_stats(opt, world, fallback_events=None):
    """Mean duration of the step's one all-reduce over the timed steps (event pair on the launch stream around it: the wait of
    the compute stream for the collective is inside) and its bus bandwidth 2 (N-1)/N * bytes / t -- what a scaling run needs to
    tell whether xGMI is the limiter.  When the timed region replays a captured step (no events inside a graph) the figure
    comes from the three eager one-stream steps before it (``fallback_events``)."""
    ev = getattr(opt, "collective_events", None)
    src = "timed region"
    if not ev and fallback_events:
        ev, src = fallback_events, "3 eager one-stream steps before the timed region"
    if not ev:
        return {"allreduce_ms": None, "bus_GBps": None}
    ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    bus = (2.0 * (world - 1) / world * opt.nbytes / (ms * 1e-3) / 1e9) if world > 1 and ms > 0 else 0.0
    return {"allreduce_ms": round(ms, 4), "bus_GBps": round(bus, 2), "allreduce_events": len(ev), "allreduce_timed_in": src}


class Heartbeat:
    """In-process hang detector of a rank (torch.distributed.run has none).  ``beat(phase)`` is called after the rendezvous, at
    every phase change and after every step; a daemon thread ends the process with exit code 124 when NO beat arrived for
    ``timeout`` seconds, naming the phase it was in -- a collective that never completes, a peer that died, a capture that
    deadlocks.  A healthy long run (large --steps, C5 on N GPUs) is never killed: the deadline is per beat, not per run
    (it used to be one threading.Timer(timeout) over the whole run).  ``stop()`` before the final print."""

    def __init__(self, timeout, rank=0, poll=None, out=None, _exit=os._exit):
        import threading
        self.timeout, self.rank, self.out, self._exit = float(timeout), rank, out, _exit
        self.phase, self.last, self.beats = "start", time.monotonic(), 0
        self._stop = threading.Event()
        self._poll = poll if poll is not None else min(5.0, max(0.05, self.timeout / 4))
        self._thread = threading.Thread(target=self._run, name="bench-heartbeat", daemon=True)
        self._thread.start()

    def beat(self, phase=None):
        if phase is not None:
            self.phase = phase
        self.last = time.monotonic()
        self.beats += 1

    def stop(self):
        self._stop.set()

    def _run(self):
        while not self._stop.wait(self._poll):
            idle = time.monotonic() - self.last
            if idle > self.timeout:
                print(f"bench.py: rank {self.rank} still running after {self.timeout:.0f} s without progress in phase "
                      f"'{self.phase}' ({self.beats} heartbeats so far; a collective that never completed?): exiting with 124",
                      file=self.out or sys.stderr, flush=True)
                self._exit(124)
                return


def launch_policy(graph, eager, use_dist, backend):
    """Launch mode of the timed region -> ("graph" | "eager" | "probe", rule).

    One process, no process group ("rule"): hipGraph replay when an eager step takes < 10 ms (the small shapes are launch-bound:
    1.3-1.9 ms replayed against 3-4 ms eager), eager stepping above -- what a multi-rank run does; at C3 the two modes are equal
    once both are timed in the same state of the chip (round 5: 30.96 eager / 30.89 replayed cool, 31.65-31.87 / 31.66-31.70 after
    1.5 s of load), C5 116 against 120 ms.  Both modes are still timed over --probe-steps steps each, but AFTER the timed region (round
    4 did it before: 20 probe steps and four captures put the chip under load for 1.5 s before the contract's W warm-up steps, and
    every mode then runs ~2 % slower -- DESIGN.md section 4f), and reported in launch_probe.
    With a process group (N > 1, or the one-rank RCCL rehearsal) the DEFAULT IS EAGER: a captured step
    contains the in-stream RCCL all-reduce, ProcessGroupNCCL's watchdog thread polls events while the main thread captures, and
    an abort on that thread never reaches an `except` here -- the first multi-device contact of this code is the driver's
    scaling run, where an abort means no number, and what capture buys is <= 2 % of a C3 step.  ``--graph`` opts in (thread-local
    capture mode, trainer.GraphedTrainStep); gloo synchronises with the host inside its collective and cannot be captured."""
    if graph and eager:
        raise SystemExit("bench.py: --graph and --eager exclude each other")
    if use_dist and backend != "nccl":
        if graph:
            raise SystemExit("bench.py --graph: a gloo all-reduce cannot be captured into a HIP graph (use the nccl backend)")
        return "eager", "eager: the all-reduce of this backend synchronises with the host and cannot be captured"
    if graph:
        return "graph", "--graph: forced" + (" (the captured step contains the in-stream RCCL all-reduce)" if use_dist else "")
    if eager:
        return "eager", "--eager: not probed"
    if use_dist:
        return "eager", ("eager by default with a process group: the captured step with the RCCL all-reduce inside is opt-in "
                         "(--graph); not probed")
    return "rule", ("hipGraph replay if an eager step takes < 10 ms (launch-bound), else eager; both modes are timed AFTER the timed region "
                    "and reported (launch_probe)")


def gather_device_identities(ident, world):
    """Every rank contributes its device identity; -> (list by rank, all distinct?)."""
    devices = [None] * world
    dist.all_gather_object(devices, ident)
    return devices, len({(d["uuid"], d["pci"]) for d in devices}) == world


def gather_rank_times(elapsed, steps, world, dev):
    """-> (every rank's own ms per step, the slowest rank's elapsed seconds = the job's time)."""
    mine = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    allr = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allr, mine)
    per_rank_ms = [float(x.item()) / steps * 1e3 for x in allr]
    tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    return per_rank_ms, float(tt.item())


def replicas_hold_identical_weights(flat_param):
    """Same init + averaged gradients: two checksums of the flat parameter buffer must be equal on every rank."""
    chk = torch.stack([flat_param.double().sum(), flat_param.double().abs().sum()])
    lo, hi = chk.clone(), chk.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    return bool(torch.equal(lo, hi))


def collective_object(opt, world, devices, table_events=None):
    """The `collective` object of the JSON line (what a scaling run is checked against: DESIGN.md section 6)."""
    return dict({"backend": dist.get_backend(), "world_size": dist.get_world_size(), "all_reduce_per_step": 1,
                 "launcher": "self (child processes)" if os.environ.get("NRM_BENCH_CHILD") else "external",
                 "devices": devices, "devices_distinct": len({(d["uuid"], d["pci"]) for d in devices}) == world},
                **collective_stats(opt, world, table_events))


def per_rank_object(per_rank_ms):
    return {"min": round(min(per_rank_ms), 3), "max": round(max(per_rank_ms), 3), "ranks": [round(x, 3) for x in per_rank_ms]}


def job_fields(world, per_gpu_batch, steps, elapsed, per_rank_ms, opt, use_dist, devices, replicas_in_sync, table_events=None):
    """The fields of the JSON line that describe the JOB (all N ranks): whole-job throughput over the slowest rank's time, the
    rank count as the process group saw it, every rank's own step time, the one collective of a step.  A scaling record is
    checked against these (n_gpus == collective.world_size == N, devices_distinct, per-rank times, all-reduce time and bus
    bandwidth); tests/test_dp_gloo.py runs this function under a two-rank gloo group."""
    return {"value": round(world * per_gpu_batch * steps / elapsed, 2), "n_gpus": world, "ms_per_step": round(elapsed / steps * 1e3, 3),
            "per_rank_ms_per_step": per_rank_object(per_rank_ms), "grad_allreduce_bytes": opt.nbytes if use_dist else 0,
            "replicas_in_sync": replicas_in_sync,
            "collective": collective_object(opt, world, devices, table_events) if use_dist else None}


HEAVY = ("nrm_pwattn_fwd", "pwattn_bwd_e_bt", "pwattn_bwd_e_dw", "pwattn_bwd_e_bh", "pwattn_bwd_dp_dtdh", "pwattn_bwd_rw_dtdh", "nrm_pwattn_bwd_dz")   # event-timed inside the timed region


def self_launch(n, timeout=900.0):
    """`python bench.py --gpus N` without a launcher: start the N ranks as CHILD processes of this one (the same
    command line, RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* in their environment) and leave with the worst exit code.
    Nothing in this process has touched the GPU yet (device_count() does not initialise it), and nothing is
    exec'ed over a process that has.  Rank 0 inherits stdout and prints the JSON line."""
    import socket
    import subprocess
    single = os.environ.get("NRM_SINGLE_DEVICE") == "1"
    ndev = torch.cuda.device_count()
    if ndev < n and not single:
        raise SystemExit(f"bench.py --gpus {n}: only {ndev} device(s) visible (a rehearsal of {n} ranks on one GPU needs "
                         "NRM_SINGLE_DEVICE=1 NRM_DIST_BACKEND=gloo)")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), NRM_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    raise SystemExit(wait_for_ranks(procs, timeout))


def wait_for_ranks(procs, timeout, grace=10.0, out=sys.stderr):
    """Wait for the rank processes.  One rank failing -> its peers (which would wait in a collective forever) are terminated
    and the first failure's code is returned.  ``timeout`` seconds without all of them exiting -> every survivor is terminated,
    after ``grace`` more seconds killed, the ranks that were still alive are named on ``out`` and 124 is returned.  Nothing is
    ever re-exec'ed."""
    t0 = time.monotonic()
    worst, alive = 0, {r: pr for r, pr in enumerate(procs)}
    while alive:
        for r, pr in list(alive.items()):
            rc = pr.poll()
            if rc is None:
                continue
            del alive[r]
            if rc != 0 and not worst:
                worst = rc
                print(f"bench.py: rank {r} exited with code {rc}; stopping ranks {sorted(alive)}", file=out, flush=True)
                for other in alive.values():            # one rank died: its peers would wait in a collective forever
                    other.terminate()
        if alive and time.monotonic() - t0 > timeout:
            hung = sorted(alive)
            print(f"bench.py: ranks {hung} still running after {timeout:.0f} s (a collective that never completed?): "
                  "terminating them", file=out, flush=True)
            for pr in alive.values():
                pr.terminate()
            t1 = time.monotonic()
            while any(pr.poll() is None for pr in alive.values()) and time.monotonic() - t1 < grace:
                time.sleep(0.1)
            for r, pr in alive.items():
                if pr.poll() is None:
                    print(f"bench.py: rank {r} ignored SIGTERM, killing it", file=out, flush=True)
                    pr.kill()
            for pr in alive.values():
                pr.wait()
            return 124
        time.sleep(0.2)
    return worst


def infer_cpu_baseline(dims, wl, seconds, cpu_batch):
    """orc.model_test_scores (restatement of reference test.py:31-74, parity unpinned: see its docstring) on the host cores, on
    a bounded sample of the same workload."""
    from news_recommendation_model_amd import synth
    from oracle import user_model_oracle as orc
    cores = host_cores()
    torch.set_num_threads(cores)
    per_imp = wl["T"] * wl["H"] * 4 * wl["emb"] * 4
    Bc = cpu_batch or int(max(2, min(wl["B"], (256 << 20) // per_imp)))
    batch = synth.make_batch(dims, Bc, wl["H"], wl["T"], seed=0)
    sd = synth.make_state_dict(dims, seed=1, user_num=int(batch["user_num"]), perturb=False)
    p = orc.to_torch_params(sd)
    tb = {k: torch.from_numpy(v) for k, v in batch.items() if isinstance(v, np.ndarray) and v.ndim > 0}
    tb = {k: (v.float() if v.is_floating_point() else v) for k, v in tb.items()}
    tb["empty_num"] = torch.zeros(Bc, dtype=torch.int64)
    orc.model_test_scores([p], tb)
    n, t0 = 0, time.perf_counter()
    while True:
        orc.model_test_scores([p], tb)
        n += 1
        el = time.perf_counter() - t0
        if (el >= seconds and n >= 3) or el > 3 * seconds:
            break
    return {"value": round(Bc * n / el, 2), "unit": "impressions/s", "cores": cores, "kind": "port",
            "sample": f"oracle/user_model_oracle.model_test_scores (one model), B={Bc} H={wl['H']} T={wl['T']} D={wl['emb']}, "
                      f"{n} batches after 1 warm-up, torch CPU fp32 {cores} threads"}


def main_infer(args):
    """--mode infer: impressions/s of evaluation.predict([model], batch) -- reference test.py:31-74 -- on one GPU."""
    if args.gpus != 1:
        raise SystemExit("bench.py --mode infer runs on one GPU (impressions are independent: N replicas scale trivially)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    from news_recommendation_model_amd import evaluation, native, synth, trainer
    from news_recommendation_model_amd import ops as _ops
    from news_recommendation_model_amd.config import Dims, WORKLOADS
    native.load()
    wl = dict(WORKLOADS[args.workload])
    if args.batch:
        wl["B"] = args.batch
    B, H, T, D = wl["B"], wl["H"], wl["T"], wl["emb"]
    if args.dtype is None:
        args.dtype = "bf16x3" if args.workload == "C2-small" else "f32"
    dims = Dims.for_emb(D)
    user_num = 10 * B
    if args.dtype != "f32":
        _ops.set_dense_arithmetic(args.dtype)
    sd = synth.make_state_dict(dims, seed=1, user_num=user_num, perturb=False)
    model = trainer.build_model(dims, user_num, sd, device=dev, attention_mma=args.dtype).eval()
    batch = synth.make_batch(dims, B, H, T, seed=0, user_num=user_num, dtype=np.float32)
    tb = trainer.batch_to_device(batch, dev)
    # throughput run: no padded candidates; on the host, where the reference's DataLoader leaves it (test.py:46) -- predict()
    # reads the common-padding trim there without a device synchronisation
    tb["empty_num"] = torch.zeros(B, dtype=torch.int64).pin_memory()
    models = [model]
    run = evaluation.GraphedPredict(models) if args.graph else (lambda b: evaluation.predict(models, b))
    for _ in range(max(args.warmup, 2)):
        scores, live = evaluation.predict(models, tb)
    torch.cuda.synchronize()
    native.kernel_events = []
    inv = model.invariant_interest_model
    inv.two_streams = False                             # kernel table: untimed, bracketed, every kernel alone on the chip
    for _ in range(3):
        evaluation.predict(models, tb)
    torch.cuda.synchronize()
    inv.two_streams = None
    events, native.kernel_events = native.kernel_events, None
    for _ in range(2):
        scores, live = run(tb)                          # (graph mode: capture happens here, outside the timed region)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        scores, live = run(tb)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    _ops.check_index_errors(dev)
    per = {}
    for tag, e0, e1 in events:
        per.setdefault(tag, []).append(e0.elapsed_time(e1))
    kern = {k: {"launches": len(v), "mean_ms": float(np.mean(v))} for k, v in per.items()}
    fwd_ms = kern["nrm_pwattn_fwd"]["mean_ms"]
    flops = 2.0 * B * T * H * D * D
    peak = FP32_MFMA_PEAK_TFLOPS if args.dtype == "f32" else BF16_MFMA_PEAK_TFLOPS
    ach = flops / (fwd_ms * 1e-3) / 1e12
    # without the z store the forward's algorithmic HBM bytes are its operands (t, h, u, v) and the scores: never the bound
    roof = {"bound": "mfma", "kernel": "nrm_pwattn_fwd (no z store)", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
            "frac": round(ach / peak, 4), "flops_per_launch": flops, "mean_launch_ms": round(fwd_ms, 4), "traffic": None,
            "traffic_source": "no committed PMC profile for the inference forward"}
    tpath = os.path.join(ROOT, "profiles", INFER_TRAFFIC_FILES.get((args.workload, args.dtype), ""))
    if not args.batch and os.path.isfile(tpath):
        from news_recommendation_model_amd import build as _build
        prof = json.load(open(tpath))
        if prof.get("kernel_sources_sha256") != _build.sources_digest():
            roof["traffic_source"] = f"stale: {os.path.basename(tpath)} was measured on other kernel sources (commit {prof.get('git_head', '?')})"
        else:
            roof["traffic"] = prof.get("nrm_pwattn_fwd", {}).get("corrected_bytes")
            roof["traffic_source"] = f"{os.path.basename(tpath)} (rocprofv3 --pmc, commit {prof.get('git_head', '?')}, same kernel sources)"
    ms = elapsed / args.steps * 1e3
    line = {"metric": "inference impressions/sec (reference test.py:31-74 model_test, one model)", "value": round(B * args.steps / elapsed, 2),
            "unit": "impressions/s", "n_gpus": 1, "steps": args.steps, "warmup": max(args.warmup, 2), "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": args.workload, "mode": "infer", "batch": B, "hist": H, "candidates": T, "emb": D, "launch": "hipGraph replay (evaluation.GraphedPredict)" if args.graph else "eager",
                       "attention_streams": 2 if inv.uses_two_streams(B * T * H * D) else 1,
                       "kernel_durations_from": "3 extra one-stream batches after the warm-up",
                       "step": "evaluation.predict: trim + eval-mode forward (no saved z) + softmax + de-padding softmax"},
            "roofline": roof,
            "kernels": {k: {"launches": v["launches"], "mean_ms": round(v["mean_ms"], 4)} for k, v in kern.items()},
            "gpu_kernel_ms_per_step": round(sum(v["mean_ms"] * v["launches"] for v in kern.values()) / 3, 4)}
    if not args.no_cpu_baseline:
        line["cpu_baseline"] = infer_cpu_baseline(dims, wl, args.cpu_seconds, args.cpu_batch)
    print(json.dumps(line), flush=True)


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.mode == "infer":
        return main_infer(args)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args.gpus, args.timeout)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; they must agree")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    # rehearsal on a one-GPU box: NRM_SINGLE_DEVICE=1 puts every rank on cuda:0 and NRM_DIST_BACKEND=gloo replaces
    # RCCL (which needs one device per rank); the real multi-GPU run uses neither.
    if os.environ.get("NRM_SINGLE_DEVICE") == "1":
        local = 0
    backend = os.environ.get("NRM_DIST_BACKEND", "nccl")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # NRM_DIST_WORLD1=1: bring the process group up even for one rank, so that a one-GPU box exercises RCCL itself
    # (library load, communicator, the in-stream all-reduce of the flat gradient, barrier) on the code path of N > 1
    use_dist = world > 1 or os.environ.get("NRM_DIST_WORLD1") == "1"
    # in-process hang detector for rank processes started by an EXTERNAL launcher (torch.distributed.run has none): a collective
    # that never completes must not hold the node forever.  The rank names the phase it hung in and leaves with 124; the
    # launcher then takes its peers down.  (Ranks started by `python bench.py --gpus N` itself are also watched by their parent.)
    hb = Heartbeat(args.timeout, rank) if use_dist else None

    def beat(phase=None):
        if hb is not None:
            hb.beat(phase)
    beat("rendezvous")
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    devices = None
    if use_dist:
        # every rank on its own physical device (unless this is the one-GPU rehearsal): gathered and checked by every rank
        pr_ = torch.cuda.get_device_properties(local)
        ident = {"rank": rank, "current_device": torch.cuda.current_device(), "name": pr_.name,
                 "uuid": str(getattr(pr_, "uuid", "")), "pci": f"{getattr(pr_, 'pci_domain_id', 0):04x}:{getattr(pr_, 'pci_bus_id', -1):02x}:"
                                                              f"{getattr(pr_, 'pci_device_id', -1):02x}"}
        beat("device identities")
        devices, distinct = gather_device_identities(ident, world)
        if not distinct and os.environ.get("NRM_SINGLE_DEVICE") != "1":
            raise SystemExit(f"bench.py: {world} ranks but their devices are not distinct: {devices} "
                             "(one process per GPU; NRM_SINGLE_DEVICE=1 only for the one-GPU rehearsal)")

    from news_recommendation_model_amd import native, synth, trainer
    from news_recommendation_model_amd.config import Dims, WORKLOADS
    native.load()
    wl = dict(WORKLOADS[args.workload])
    if args.batch:
        wl["B"] = args.batch
    B, H, T, D = wl["B"], wl["H"], wl["T"], wl["emb"]
    if args.dtype is None:
        args.dtype = "bf16x3" if args.workload == "C2-small" else "f32"
    dims = Dims.for_emb(D)
    user_num = 10 * B
    if args.dtype != "f32":
        # BASELINE config 2 names bf16: the dense layers run on the bf16 matrix cores as well (same hi/lo split as the attention)
        from news_recommendation_model_amd import ops as _o
        _o.set_dense_arithmetic(args.dtype)
    # same weights on every rank (seed 1); every rank holds NB different batches resident in HBM and cycles through them, so the
    # timed steps do not re-fit ONE batch (round 3: ~45 steps on one repeated batch drove the logits to 1e5 and the loss back up
    # to 6.5 -- the reference's arithmetic does the same, but the printed loss said nothing)
    from news_recommendation_model_amd import ops as _ops
    sd = synth.make_state_dict(dims, seed=1, user_num=user_num, perturb=False)
    model = trainer.build_model(dims, user_num, sd, device=dev, attention_mma=args.dtype).train()
    opt = trainer.FlatAdam(model)              # Adam(lr 1e-3, wd 1e-5) + zero_grad as one launch; flat grad buffer
    reducer = None                             # the all-reduce runs on opt.flat_grad (no gather copy)
    NB = max(1, args.batches)
    tbs = [trainer.batch_to_device(synth.make_batch(dims, B, H, T, seed=rank * NB + i, user_num=user_num, dtype=np.float32), dev)
           for i in range(NB)]
    counter = {"i": 0}

    def sync():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def eager_step():
        tb_ = tbs[counter["i"] % NB]
        counter["i"] += 1
        beat()
        return trainer.train_step(model, opt, tb_, reducer)

    graphs = []

    def capture_graphs():
        """One captured step per resident batch (static input buffers = the batches themselves), sharing one memory pool."""
        pool = None
        for j in range(NB):
            g = trainer.GraphedTrainStep(model, opt, tbs[j], warmup=3 if j == 0 else 1, pool=pool)
            pool = g.graph.pool()
            graphs.append(g)

    def graph_step():
        g = graphs[counter["i"] % NB]
        counter["i"] += 1
        beat()
        return g.replay()

    def time_steps(fn, n):
        sync()
        t_ = time.perf_counter()
        for _ in range(n):
            fn()
        sync()
        t_ = (time.perf_counter() - t_) / n
        if use_dist:                                   # every rank must take the same decision: the slowest rank's time
            tt_ = torch.tensor([t_], dtype=torch.float64, device=dev)
            dist.all_reduce(tt_, op=dist.ReduceOp.MAX)
            t_ = float(tt_.item())
        return t_

    # the very first step of the process (initial weights, batch 0): the loss the oracle reproduces (cpu_baseline leg)
    beat("first step")
    l0, _ = eager_step()
    first_step_loss = float(l0)

    # Small shapes run their two attentions on two streams (modules.UserInvariantInterestModel.forward).  A kernel that
    # shares the chip with a kernel of the other branch cannot be priced against a roofline, so every event-timed step runs
    # on ONE stream; the timed region uses both only where no kernel events are taken inside it.
    inv = model.invariant_interest_model
    probe = {"steps": 0, "t_eager_ms": None, "t_graph_ms": None, "rule": None}
    flop_meter = None
    if not args.timed_kernel_events:
        args.no_kernel_timing = True
    else:
        os.environ["NRM_WGRAD_STREAM"] = "0"           # event-timed kernels must own the machine
    if args.graph or args.no_kernel_timing:
        # per-kernel durations cannot be event-timed inside a graph, and a kernel that shares the chip with a kernel of another
        # stream cannot be priced against a roofline: take them from 3 eager ONE-STREAM steps first (no second attention stream,
        # no weight-gradient stream); the first of them also runs under the matrix-FLOP meter (roofline.step)
        beat("kernel table steps")
        inv.two_streams = False
        prev_wg = os.environ.get("NRM_WGRAD_STREAM")
        os.environ["NRM_WGRAD_STREAM"] = "0"
        for _ in range(2):
            eager_step()
        sync()
        native.kernel_events = []
        if use_dist:
            opt.collective_events = []                 # an event pair around the step's one all-reduce (eager steps only)
        for k_ in range(3):
            if k_ == 0:
                _ops.flop_meter_start()
            loss, _ = eager_step()
            if k_ == 0:
                flop_meter = _ops.flop_meter_stop()
        sync()
        events, native.kernel_events = native.kernel_events, None
        table_collective_events, opt.collective_events = opt.collective_events, None
        inv.two_streams = None
        if prev_wg is None:
            del os.environ["NRM_WGRAD_STREAM"]
        else:
            os.environ["NRM_WGRAD_STREAM"] = prev_wg
        # Launch mode of the timed region (launch_policy): --graph / --eager force it; one process without a process group times
        # BOTH over PROBE >= 10 steps in the same state of the box and takes the captured step when a step is launch-bound
        # (< 10 ms) or the replay is at least 1 % faster; with a process group the step is eager unless --graph asks otherwise.
        mode, rule = launch_policy(args.graph, args.eager, use_dist, backend)
        probe["rule"] = rule
        capture_error = None
        PROBE = max(10, args.probe_steps)

        def try_capture():
            """Capture one step per resident batch; False (and eager stepping on every rank) if it fails on any rank."""
            nonlocal capture_error
            beat("graph capture")
            try:
                capture_graphs()
                ok = 1.0
            except Exception as e:                      # noqa: BLE001 -- never let a failed capture cost the run its number
                capture_error, ok = repr(e), 0.0
                graphs.clear()
            if use_dist:                                # a capture that failed on ANY rank: everybody steps eagerly
                tt_ = torch.tensor([ok], dtype=torch.float64, device=dev)
                dist.all_reduce(tt_, op=dist.ReduceOp.MIN)
                ok = float(tt_.item())
            if not ok:
                graphs.clear()
            return bool(ok)

        if mode == "rule":
            beat("launch rule: three eager steps")
            t_first = time_steps(eager_step, 3)
            probe["t_rule_ms"] = round(t_first * 1e3, 4)
            args.graph = t_first < 10e-3
        if args.graph and not try_capture():
            if mode == "graph":
                raise SystemExit(f"bench.py --graph: capturing the step failed: {capture_error}")
            args.graph = False
            probe["rule"] = f"capture failed ({capture_error}): eager"
        beat("warm-up + timed steps")
        run = graph_step if args.graph else eager_step
        for _ in range(args.warmup):
            run()
        sync()
        if use_dist and not args.graph:
            opt.collective_events = []                 # an event pair around the step's one all-reduce
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss, _ = run()
        sync()
        elapsed = time.perf_counter() - t0
        steps_done = counter["i"]
        if mode == "rule" and not args.no_probe:
            # both launch modes over PROBE steps each, in the state the timed region left the chip in: information, not a decision
            beat("launch probe (after the timed region)")
            if graphs or try_capture():
                t_a = time_steps(graph_step if args.graph else eager_step, PROBE)       # the mode that was timed, again
                t_b = time_steps(eager_step if args.graph else graph_step, PROBE)
                t_graph, t_eager = (t_a, t_b) if args.graph else (t_b, t_a)
                probe.update(steps=PROBE, t_eager_ms=round(t_eager * 1e3, 4), t_graph_ms=round(t_graph * 1e3, 4),
                             when="after the timed region (the timed mode first)",
                             note="on big shapes the chip runs ~2 % slower once it has been under load for ~1.5 s (DESIGN.md 4f): these two "
                                  "figures are from that state, ms_per_step from the contract's W warm-up + K timed steps")
            else:
                probe.update(steps=0, when=f"capture failed: {capture_error}")
            if not args.graph:
                graphs.clear()
    else:
        # --timed-kernel-events (round-1/2 behaviour): the full per-kernel table comes from 3 extra eager steps outside the timed
        # region; inside it only the four big attention kernels (8 launches per step) are bracketed by events
        beat("warm-up + timed steps (kernel events inside)")
        inv.two_streams = False
        table_collective_events = None
        for _ in range(args.warmup):
            eager_step()
        sync()
        native.kernel_events = []
        for k_ in range(3):                        # table steps: untimed, after the W warm-up steps
            if k_ == 0:
                _ops.flop_meter_start()
            eager_step()
            if k_ == 0:
                flop_meter = _ops.flop_meter_stop()
        sync()
        table_events = native.kernel_events
        native.kernel_event_tags = set(HEAVY)
        native.kernel_events = []
        if use_dist:
            opt.collective_events = []
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss, _ = eager_step()
        sync()
        elapsed = time.perf_counter() - t0
        steps_done = counter["i"]
    _ops.check_index_errors(dev)               # a clamped table index / user id would make the number meaningless: fail loudly
    if not (args.graph or args.no_kernel_timing):
        events, native.kernel_events, native.kernel_event_tags = native.kernel_events, None, None
        # heavy kernels: timed region; everything else: the 3 bracketed warm-up steps
        events = events + [e for e in table_events if e[0] not in HEAVY]
    per_rank_ms = [elapsed / args.steps * 1e3]
    if use_dist:
        beat("gathering rank times")
        # every rank's own time for the K steps (stragglers show here); the job's time is the slowest rank's
        per_rank_ms, elapsed = gather_rank_times(elapsed, args.steps, world, dev)
        # replicas must still hold identical weights after the timed steps (same init, averaged gradients)
        replicas_in_sync = replicas_hold_identical_weights(opt.flat_param)
    else:
        replicas_in_sync = True

    pcie = None
    if args.pcie and world == 1:
        host = synth.make_batch(dims, B, H, T, seed=rank, user_num=user_num, dtype=np.float64)   # DataLoader dtype
        host = {k: torch.as_tensor(v).pin_memory() for k, v in host.items() if k != "user_num"}
        nbytes = sum(v.numel() * v.element_size() for v in host.values())
        pf = trainer.BatchPrefetcher((host for _ in range(args.steps + 3)), dev)
        it = iter(pf)
        for _ in range(3):
            b, slot = next(it)
            trainer.train_step(model, opt, b, reducer)
            pf.release(slot)
        sync()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            b, slot = next(it)
            loss_p, _ = trainer.train_step(model, opt, b, reducer)
            pf.release(slot)
        sync()
        el = time.perf_counter() - t1
        pcie = {"value": round(B * args.steps / el, 2), "unit": "impressions/s", "ms_per_step": round(el / args.steps * 1e3, 3),
                "host_dtype": "f64", "bytes_per_step": nbytes, "overlap": "pinned host batch, copy stream, one batch ahead"}

    # per-kernel launch durations from the event pairs recorded on the launch stream
    per = {}
    for tag, e0, e1 in events:
        per.setdefault(tag, []).append(e0.elapsed_time(e1))
    kern = {k: {"launches": len(v), "mean_ms": float(np.mean(v)), "total_ms": float(np.sum(v))} for k, v in per.items()}
    heavy = {k: v for k, v in kern.items() if k in HEAVY[:5]}      # the fp32 contraction kernels
    dom = max(heavy, key=lambda k: heavy[k]["total_ms"])
    flops_per_launch = 2.0 * B * T * H * D * D            # both attentions have width D in BASELINE shapes
    achieved = flops_per_launch / (heavy[dom]["mean_ms"] * 1e-3) / 1e12
    peak = FP32_MFMA_PEAK_TFLOPS if args.dtype == "f32" else BF16_MFMA_PEAK_TFLOPS
    roof = {"bound": "mfma", "kernel": dom, "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
            "frac": round(achieved / peak, 4), "flops_per_launch": flops_per_launch, "mean_launch_ms": round(heavy[dom]["mean_ms"], 4)}
    if args.dtype != "f32":
        # With bf16 matrix cores no attention kernel is bound by the algorithmic FLOPs any more: price the kernel with the largest
        # summed time against BOTH roofs and report the larger fraction --
        #   hbm:  algorithmic bytes (one pass over z / dz [B,T,H,D] fp32; two for the in-place dz pass) / t against 8 TB/s (spec; the
        #         guide measures 6.3 TB/s achievable: `frac_of_achievable_hbm`);
        #   mfma: ISSUED matrix-core FLOPs / t against the dense bf16 peak -- bf16x3 issues three MFMAs per product, so its matrix
        #         pipe does 3x the algorithmic 2 B T H D^2 (`mfma_frac_algorithmic` is the 1x figure).
        allheavy = {k: v for k, v in kern.items() if k in HEAVY}
        dom = max(allheavy, key=lambda k: allheavy[k]["total_ms"])
        zbytes = 4.0 * B * T * H * D
        bytes_per_launch = 2 * zbytes if dom == "nrm_pwattn_bwd_dz" else zbytes
        t_s = allheavy[dom]["mean_ms"] * 1e-3
        issue = 3.0 if args.dtype == "bf16x3" else 1.0
        f_alg = 0.0 if dom == "nrm_pwattn_bwd_dz" else flops_per_launch / t_s / 1e12 / BF16_MFMA_PEAK_TFLOPS
        f_mfma = issue * f_alg
        f_hbm = bytes_per_launch / t_s / 1e9 / HBM_PEAK_GBS
        both = {"hbm_frac": round(f_hbm, 4), "frac_of_achievable_hbm": round(bytes_per_launch / t_s / 1e9 / 6300.0, 4),
                "mfma_frac_issued": round(f_mfma, 4), "mfma_frac_algorithmic": round(f_alg, 4),
                "bytes_per_launch": bytes_per_launch, "flops_per_launch": flops_per_launch, "issued_flops_per_launch": issue * flops_per_launch,
                "mean_launch_ms": round(allheavy[dom]["mean_ms"], 4),
                "all_heavy_kernels": {k: {"mean_ms": round(v["mean_ms"], 4),
                                          "hbm_frac": round((2 if k == "nrm_pwattn_bwd_dz" else 1) * zbytes / (v["mean_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                          "mfma_frac_issued": 0.0 if k == "nrm_pwattn_bwd_dz" else round(issue * flops_per_launch / (v["mean_ms"] * 1e-3) / 1e12 / BF16_MFMA_PEAK_TFLOPS, 4)}
                                      for k, v in allheavy.items()}}
        if f_hbm >= f_mfma:
            roof = dict({"bound": "hbm", "kernel": dom, "achieved": round(bytes_per_launch / t_s / 1e9, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(f_hbm, 4)}, **both)
        else:
            roof = dict({"bound": "mfma", "kernel": dom, "achieved": round(issue * flops_per_launch / t_s / 1e12, 2), "peak": BF16_MFMA_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(f_mfma, 4)}, **both)

    # HBM traffic of the dominant kernel: PMC counters cannot be read from inside this process; the figure comes from
    # the committed rocprofv3 --pmc passes of this same command (profiles/<round>_<workload>_pmc_traffic.json, written by
    # scripts/pmc_summary.py, which stamps the digest of the kernel sources it measured).  A profile of other sources
    # is stale: traffic is then null and the note says so.
    traffic, traffic_note = None, "no committed PMC profile for this workload"
    from news_recommendation_model_amd import build as _build
    tpath = os.path.join(ROOT, "profiles", PMC_TRAFFIC_FILES.get((args.workload, args.dtype), ""))
    if not args.batch and os.path.isfile(tpath):
        try:
            prof = json.load(open(tpath))
            if prof.get("kernel_sources_sha256") != _build.sources_digest():
                traffic_note = f"stale: {os.path.basename(tpath)} was measured on other kernel sources (commit {prof.get('git_head', '?')})"
            else:
                traffic = prof.get(dom, {}).get("corrected_bytes")
                traffic_note = f"{os.path.basename(tpath)} (rocprofv3 --pmc, commit {prof.get('git_head', '?')}, same kernel sources)"
        except Exception as e:
            traffic_note = "unreadable profile: " + repr(e)

    beat("assembling the line")
    if rank == 0:
        ms = elapsed / args.steps * 1e3
        # whole-step roofline: the matrix FLOPs one step ISSUES on the MFMA pipe (attention contractions + every dense GEMM incl.
        # side projections and weight gradients; algorithmic 2MNK, no tile padding; x3 for the bf16x3 split) / ms_per_step / peak
        issue = 3.0 if args.dtype == "bf16x3" else 1.0
        step_roof = None
        if flop_meter is not None:
            mf = issue * (flop_meter["dense"] + flop_meter["contraction"])
            step_roof = {"matrix_flops_per_step": mf, "contraction_flops": issue * flop_meter["contraction"],
                         "dense_flops": issue * flop_meter["dense"], "contraction_launches": flop_meter["contraction_launches"],
                         "gemm_launches": flop_meter["dense_launches"], "issue_factor": issue, "peak": peak, "unit": "TFLOP/s",
                         "achieved": round(mf / (ms * 1e-3) / 1e12, 2), "frac": round(mf / (ms * 1e-3) / 1e12 / peak, 4),
                         "ms_at_peak": round(mf / (peak * 1e12) * 1e3, 3)}
        roof["step"] = step_roof
        job = job_fields(world, B, args.steps, elapsed, per_rank_ms, opt, use_dist, devices, replicas_in_sync, table_collective_events)
        line = {
            "metric": "train impressions/sec", "value": job["value"],
            "unit": "impressions/s", "n_gpus": job["n_gpus"], "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": job["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"f32": "f32", "bf16": "bf16 (attention contractions and dense GEMMs: bf16 MFMA operands, fp32 accumulate; rest f32)",
                      "bf16x3": "bf16x3 (attention contractions and dense GEMMs: bf16 MFMA on hi/lo split operands, fp32 accumulate; rest f32)"}[args.dtype],
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: EBNeRD-large-shape synthetic" if args.workload == "C3-large" else args.workload,
                       "per_gpu_batch": B, "global_batch": world * B, "resident_batches_per_gpu": NB, "hist": H, "candidates": T, "emb": D,
                       "user_num": user_num, "parallelism": f"dp{world}", "attention_streams": 2 if (inv.uses_two_streams(B * T * H * D) and not args.timed_kernel_events) else 1,
                       "weight_gradient_stream": bool(_ops._wgrad["streams"]) and not args.timed_kernel_events,
                       "kernel_durations_from": "timed region (event pairs on the launch stream)" if args.timed_kernel_events else "3 extra eager one-stream steps after the warm-up", "launch": "hipGraph replay" if args.graph else "eager",
                       "step": "fwd+loss+bwd+allreduce+Adam(wd=1e-5)" if use_dist else "fwd+loss+bwd+Adam(wd=1e-5)"},
            "loss": round(float(loss), 6), "first_step_loss": round(first_step_loss, 6),
            "loss_note": f"first_step_loss: initial weights, batch 0 (the oracle's value for its sample of the same batch: cpu_baseline."
                         f"hip_vs_oracle_first_step.oracle_loss); loss: last timed step, after {steps_done} steps cycling {NB} resident batches",
            "launch_probe": probe, "per_rank_ms_per_step": job["per_rank_ms_per_step"],
            "roofline": dict(roof, traffic=traffic, traffic_source=traffic_note),
            "kernels": {k: {"launches": v["launches"], "mean_ms": round(v["mean_ms"], 4)} for k, v in kern.items()},
            "grad_allreduce_bytes": job["grad_allreduce_bytes"], "replicas_in_sync": job["replicas_in_sync"],
            "collective": job["collective"],
        }
        # provenance: which kernel sources the loaded binary says it was built from, next to the sources this run sees
        line["build"] = dict(native.provenance(), kernel_sources_sha256=_build.sources_digest())
        line["build"]["match"] = line["build"]["library_sources_sha256"] == line["build"]["kernel_sources_sha256"]
        if pcie is not None:
            line["pcie_inclusive"] = pcie
        line["fwd_auc_parity"] = fwd_auc_parity(dev, args.dtype, {"C2-small": "c2_small", "C1-demo": "c1_demo", "C5-long": "c5_long",
                                                                  "ref-default": "refdefault"}.get(args.workload, "c3_large"))
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(dims, wl, args.cpu_seconds, args.cpu_batch, args.dtype)
        beat("printing")
        print(json.dumps(line), flush=True)
    if use_dist:
        beat("final barrier")
        dist.barrier()                 # rank 0 is still printing / probing parity: leave together
        hb.stop()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
