"""Where the HOST time of an eagerly stepped small shape goes (cProfile over 20 steps of the reference-default shape)."""
import cProfile
import pstats
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from news_recommendation_model_amd import synth, trainer
from news_recommendation_model_amd.config import Dims, WORKLOADS

wl = WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "ref-default"]
dims = Dims.for_emb(wl["emb"])
B = wl["B"]
model = trainer.build_model(dims, 10 * B, synth.make_state_dict(dims, seed=1, user_num=10 * B, perturb=False)).train()
opt = trainer.FlatAdam(model)
tb = trainer.batch_to_device(synth.make_batch(dims, B, wl["H"], wl["T"], seed=0, user_num=10 * B, dtype=np.float32), "cuda")
for _ in range(5):
    trainer.train_step(model, opt, tb)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    trainer.train_step(model, opt, tb)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
