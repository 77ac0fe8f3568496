#!/usr/bin/env python3
"""Diagnostic: which ATen ops (and from where) a training step still launches.  usage: aten_ops.py [workload]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from torch.profiler import ProfilerActivity, profile

from news_recommendation_model_amd import synth, trainer
from news_recommendation_model_amd.config import Dims, WORKLOADS

wl = WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "C1-demo"]
dims = Dims.for_emb(wl["emb"])
B = min(wl["B"], int(sys.argv[2]) if len(sys.argv) > 2 else 64)
batch = synth.make_batch(dims, B, wl["H"], wl["T"], seed=0, dtype=np.float32)
sd = synth.make_state_dict(dims, seed=1, user_num=int(batch["user_num"]), perturb=False)
model = trainer.build_model(dims, int(batch["user_num"]), sd).train()
opt = trainer.FlatAdam(model)
tb = trainer.batch_to_device(batch)
for _ in range(3):
    trainer.train_step(model, opt, tb)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    trainer.train_step(model, opt, tb)
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key.startswith("aten::") and e.device_time_total > 0
        and e.key in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::add", "aten::fill_", "aten::mul", "aten::sum", "aten::_to_copy")]
rows.sort(key=lambda e: (e.key, -e.count))
for e in rows:
    print(f"{e.count:4d} {e.key:20s} dev_us={e.device_time_total:7.1f}  shapes={e.input_shapes}")
