#!/usr/bin/env python3
"""Diagnostic: at a FULL-size small-shape workload (kernels really overlap there) the two-stream + graphed step must land where
the one-stream eager step lands.  usage: streams_fullsize.py [workload] [steps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from news_recommendation_model_amd import synth, trainer
from news_recommendation_model_amd.config import Dims, WORKLOADS

name = sys.argv[1] if len(sys.argv) > 1 else "C2-small"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
wl = WORKLOADS[name]
dims = Dims.for_emb(wl["emb"])
B = wl["B"]
sd = synth.make_state_dict(dims, seed=1, user_num=10 * B, perturb=True)
tb = trainer.batch_to_device(synth.make_batch(dims, B, wl["H"], wl["T"], seed=3, user_num=10 * B, dtype=np.float32))
mma = "bf16x3" if name == "C2-small" else "f32"


def run(two_streams, graphed):
    model = trainer.build_model(dims, 10 * B, sd, attention_mma=mma).train()
    model.invariant_interest_model.two_streams = two_streams
    opt = trainer.FlatAdam(model)
    losses = []
    if graphed:
        step = trainer.GraphedTrainStep(model, opt, tb, warmup=2)
        for _ in range(steps - 2):
            losses.append(float(step.replay()[0]))
    else:
        for _ in range(steps):
            losses.append(float(trainer.train_step(model, opt, tb)[0]))
    torch.cuda.synchronize()
    return losses, {k: p.detach().clone() for k, p in model.named_parameters()}, opt.exp_avg.clone()


ref_l, ref_p, ref_m = run(False, False)
worst = 0.0
for two, gr in ((True, False), (True, True), (False, True)):
    for rep in range(3):
        l, p, m = run(two, gr)
        assert abs(l[-1] - ref_l[-1]) <= 2e-4 * abs(ref_l[-1]) + 1e-7, (two, gr, l[-1], ref_l[-1])
        for k in ref_p:
            d = float((p[k] - ref_p[k]).abs().max())
            worst = max(worst, d)
            assert d <= 2e-4, (two, gr, rep, k, d)
        assert float((m - ref_m).abs().max()) <= 1e-3 * float(ref_m.abs().max()) + 1e-8
print(name, "losses", [round(x, 6) for x in ref_l], "worst parameter difference", worst)
print("streams ok")
