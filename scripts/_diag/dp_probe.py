#!/usr/bin/env python3
"""Diagnostic: mean launch time of the fp32 dP-walk backward kernel (pwattn_bwd_dp.hip) and of its neighbours at a BASELINE shape
(event pairs on the launch stream).  NRM_HOTPATH_LIB selects a variant build (timing-only builds need NRM_ALLOW_DIAG_LIB=1).
    usage: dp_probe.py [B T H D] [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("NRM_BWD_DP", "1")
import numpy as np
import torch

from news_recommendation_model_amd import native, ops

B, T, H, D = (int(a) for a in sys.argv[1:5]) if len(sys.argv) > 4 else (1024, 30, 50, 400)
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 6
g = torch.Generator(device="cuda").manual_seed(0)
r = lambda *s: torch.randn(*s, device="cuda", generator=g)          # noqa: E731
t, h = r(B, T, D).requires_grad_(True), r(B, H, D).requires_grad_(True)
w1, b1 = (r(D, 4 * D) / (4 * D) ** 0.5).requires_grad_(True), (0.1 * r(D)).requires_grad_(True)
w2, b2 = (r(1, D) / D ** 0.5).requires_grad_(True), r(1).requires_grad_(True)
gs = r(B, T, H)


def step():
    s = ops.pointwise_attention_scores(t, h, w1, b1, w2, b2, mma="f32")
    s.backward(gs)
    for x in (t, h, w1, b1, w2, b2):
        x.grad = None


for _ in range(2):
    step()
torch.cuda.synchronize()
native.kernel_events = []
for _ in range(iters):
    step()
torch.cuda.synchronize()
ev, native.kernel_events = native.kernel_events, None
flops = 2.0 * B * T * H * D * D
out = []
for tag in ("pwattn_bwd_dp_dtdh", "pwattn_bwd_e_dw", "pwattn_bwd_e_bt", "pwattn_bwd_e_bh", "nrm_pwattn_fwd"):
    ms = [e0.elapsed_time(e1) for tg, e0, e1 in ev if tg == tag]
    if ms:
        out.append(f"{tag.replace('pwattn_', '').replace('nrm_', '')} {np.mean(ms):.4f} ms (min {np.min(ms):.4f}, {flops / np.mean(ms) / 1e9 / 157.3:.3f})")
print(f"{os.path.basename(os.environ.get('NRM_HOTPATH_LIB', 'product')):22s} [{B} {T} {H} {D}] " + "; ".join(out), flush=True)
