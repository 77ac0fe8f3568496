"""One rank of the data-parallel HIP-path test (started by tests/test_gpu_dp.py, one process per rank).

Every rank sits on cuda:0 (a 1-GPU box) and the process group is gloo, which moves the flat gradient through host
memory; on a multi-GPU node the same code path runs with backend nccl (= RCCL) and one device per rank
(bench.py --gpus N).  The rank
  1. runs forward + loss + backward of trainer.train_step's model on ITS shard through the HIP kernels and compares
     loss, logits and every gradient with the oracle run on the same shard (SURVEY.md §8e parity rule: per-replica
     batch, per-replica BatchNorm statistics);
  2. checks that FlatAdam.all_reduce_grads issues exactly ONE collective and leaves the mean of the ranks' gradients;
  3. takes two more full trainer.train_step()s (one collective each) and checks that all replicas hold
     bit-identical parameters and Adam moments afterwards.
Writes {"rank": r, "ok": true, ...} as JSON to the path in argv[1]."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import torch
import torch.distributed as dist


def main(out_path):
    from news_recommendation_model_amd import synth, trainer
    from news_recommendation_model_amd.config import Dims
    from oracle import user_model_oracle as orc
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group(os.environ.get("NRM_DIST_BACKEND", "gloo"), rank=rank, world_size=world)
    res = {"rank": rank, "ok": False}
    try:
        dims = Dims.for_emb(64, category_label_num=40)
        B, H, T = 8, 11, 6
        batch = synth.make_batch(dims, B, H, T, seed=11)
        user_num = int(batch["user_num"])
        sd = synth.make_state_dict(dims, seed=1, user_num=user_num)
        tb_cpu = {k: torch.from_numpy(v) for k, v in batch.items() if isinstance(v, np.ndarray) and v.ndim > 0}
        shard_cpu = trainer.shard_batch(tb_cpu, rank, world)
        shard = {k: v.cuda() for k, v in shard_cpu.items()}

        model = trainer.build_model(dims, user_num, sd, device="cuda").train()
        opt = trainer.FlatAdam(model)
        names = [k for k, _ in model.named_parameters()]

        # -- 1. this rank's shard through the HIP kernels vs the oracle on the same shard
        out = model(shard["x_history"], shard["x_target"], shard["x_global"])
        loss = model.loss(shard["user_id"], out, shard["label"])
        loss.backward()
        p = orc.to_torch_params(sd)
        sh32 = {k: (v.float() if v.is_floating_point() else v) for k, v in shard_cpu.items()}
        loss_o, r_o, g_o = orc.train_step(p, {"step": 0, "m": {}, "v": {}}, sh32, lr=0.0)
        err_r = float((out.detach().cpu() - r_o).abs().max() / r_o.abs().max())
        assert err_r < 1e-3, f"logits vs oracle {err_r}"
        assert abs(float(loss.detach()) - float(loss_o)) < 1e-3 * abs(float(loss_o))
        gscale = max(float(g.abs().max()) for g in g_o.values())
        worst = 0.0
        for k, prm in model.named_parameters():
            ref, got = g_o[k], prm.grad.detach().cpu()
            if k in ("delta", "out_mlp.fc2.bias"):
                assert float(got.abs().max()) < 1e-5 * max(1.0, gscale), k
            else:
                e = float((got - ref).abs().max() / (ref.abs().max() + 1e-30))
                worst = max(worst, e)
                assert e < 1e-2, (k, e)
        res["grad_max_rel_err_vs_oracle"] = worst
        opt.collect_grads()                                  # every p.grad -> its slot of the flat buffer, one launch
        local = opt.flat_grad.clone()
        for i, prm in enumerate(opt.params):
            assert prm.grad is None
        assert float(local.abs().max()) > 0

        # -- 2. exactly one collective, and it leaves the mean over ranks
        calls = {"n": 0}
        real = dist.all_reduce

        def counting(*a, **kw):
            calls["n"] += 1
            return real(*a, **kw)
        dist.all_reduce = counting
        try:
            opt.all_reduce_grads()
            assert calls["n"] == 1, calls
            gathered = [torch.empty_like(local) for _ in range(world)]
            dist.all_gather(gathered, local)
            want = sum(g.double() for g in gathered) / world
            assert float((opt.flat_grad.double() - want).abs().max()) <= 1e-6 * float(want.abs().max()) + 1e-12
            opt.step(zero_grad=True)

            # -- 3. two more whole steps: one collective each
            for step in range(2):
                calls["n"] = 0
                trainer.train_step(model, opt, shard)
                assert calls["n"] == 1, (step, calls)
        finally:
            dist.all_reduce = real
        torch.cuda.synchronize()
        assert opt.steps == 3
        for name, buf in (("param", opt.flat_param), ("exp_avg", opt.exp_avg), ("exp_avg_sq", opt.exp_avg_sq)):
            both = [torch.empty_like(buf) for _ in range(world)]
            dist.all_gather(both, buf)
            for other in both[1:]:
                assert torch.equal(both[0], other), f"replicas differ in {name}"
        # BatchNorm statistics are per replica (different shards): they must differ, i.e. nothing synchronised them
        rm = model.bn.running_mean.clone()
        rms = [torch.empty_like(rm) for _ in range(world)]
        dist.all_gather(rms, rm)
        res["bn_stats_differ"] = bool(not torch.equal(rms[0], rms[-1]))
        res["ok"] = True
        res["allreduce_bytes"] = opt.nbytes
    except Exception as e:                                  # noqa: BLE001 -- reported to the parent test
        import traceback
        res["error"] = repr(e) + "\n" + traceback.format_exc()
    finally:
        with open(out_path, "w") as f:
            json.dump(res, f)
        dist.barrier()
        dist.destroy_process_group()
    sys.exit(0 if res["ok"] else 1)


if __name__ == "__main__":
    main(sys.argv[1])
