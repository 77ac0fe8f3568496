#!/usr/bin/env python3
"""Diagnostic: mean launch time of the attention forward kernel alone (event pairs on the launch stream) at a BASELINE shape,
with the pre-activation saved.  NRM_HOTPATH_LIB selects a variant build (timing-only builds need NRM_ALLOW_DIAG_LIB=1).
    usage: fwd_probe.py [B T H D] [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from news_recommendation_model_amd import native, ops

B, T, H, D = (int(a) for a in sys.argv[1:5]) if len(sys.argv) > 4 else (1024, 30, 50, 400)
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 10
g = torch.Generator(device="cuda").manual_seed(0)
r = lambda *s: torch.randn(*s, device="cuda", generator=g)          # noqa: E731
t, h = r(B, T, D), r(B, H, D)
w1, b1 = (r(D, 4 * D) / (4 * D) ** 0.5).requires_grad_(True), (0.1 * r(D)).requires_grad_(True)
w2, b2 = (r(1, D) / D ** 0.5).requires_grad_(True), r(1).requires_grad_(True)
for _ in range(3):
    s = ops.pointwise_attention_scores(t, h, w1, b1, w2, b2, mma="f32")
    del s
torch.cuda.synchronize()
native.kernel_events = []
for _ in range(iters):
    s = ops.pointwise_attention_scores(t, h, w1, b1, w2, b2, mma="f32")
    del s
torch.cuda.synchronize()
ev, native.kernel_events = native.kernel_events, None
ms = [e0.elapsed_time(e1) for tag, e0, e1 in ev if tag == "nrm_pwattn_fwd"]
flops = 2.0 * B * T * H * D * D
print(f"{os.path.basename(os.environ.get('NRM_HOTPATH_LIB', 'product')):24s} fwd {np.mean(ms):.4f} ms (min {np.min(ms):.4f})  "
      f"{flops / np.mean(ms) / 1e9 / 157.3:.3f} of the fp32 MFMA peak", flush=True)
