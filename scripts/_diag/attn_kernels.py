#!/usr/bin/env python3
"""Diagnostic: one attention (forward + backward) at a given shape with every kernel bracketed by events; prints the mean
launch time per kernel and the gradients' distance from the fp32-MFMA path of the same inputs.
    usage: attn_kernels.py <B> <T> <H> <D> <mma> [iters]        (NRM_HOTPATH_LIB selects a variant build)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from news_recommendation_model_amd import native, ops

B, T, H, D = (int(a) for a in sys.argv[1:5])
mma = sys.argv[5] if len(sys.argv) > 5 else "bf16x3"
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 10
g = torch.Generator(device="cuda").manual_seed(0)
r = lambda *s: torch.randn(*s, device="cuda", generator=g)          # noqa: E731
t, h = r(B, T, D).requires_grad_(True), r(B, H, D).requires_grad_(True)
w1, b1 = (r(D, 4 * D) / (4 * D) ** 0.5).requires_grad_(True), (0.1 * r(D)).requires_grad_(True)
w2, b2 = (r(1, D) / D ** 0.5).requires_grad_(True), r(1).requires_grad_(True)
gs = r(B, T, H)


def run(m):
    for x in (t, h, w1):
        x.grad = None
    s = ops.pointwise_attention_scores(t, h, w1, b1, w2, b2, mma=m)
    s.backward(gs)
    return t.grad.clone(), h.grad.clone(), w1.grad.clone(), s.detach().clone()


ref = run("f32")
for _ in range(2):
    got = run(mma)
torch.cuda.synchronize()
native.kernel_events = []
for _ in range(iters):
    run(mma)
torch.cuda.synchronize()
ev, native.kernel_events = native.kernel_events, None
per = {}
for tag, e0, e1 in ev:
    per.setdefault(tag, []).append(e0.elapsed_time(e1))
rel = lambda a, b: float((a - b).abs().max() / b.abs().max())          # noqa: E731
keys = ("nrm_pwattn_fwd", "nrm_pwattn_bwd_dz", "pwattn_bwd_rw_dtdh", "pwattn_bwd_e_bt", "pwattn_bwd_e_bh")
print(os.path.basename(os.environ.get("NRM_HOTPATH_LIB", "product")), " ".join(f"{k[4:]}={v}" for k, v in os.environ.items() if k.startswith("NRM_BRW")), mma, (B, T, H, D),
      " ".join(f"{k.replace('nrm_pwattn_', '').replace('pwattn_', '')}={np.mean(per[k]):.3f}" for k in keys if k in per),
      f"| vs f32: s {rel(got[3], ref[3]):.1e} dt {rel(got[0], ref[0]):.1e} dh {rel(got[1], ref[1]):.1e} dW1 {rel(got[2], ref[2]):.1e}", flush=True)
