#!/usr/bin/env python3
"""Diagnostic: the saved pre-activation z of the forward (walk form) against the fp32 forward's, element pattern of the mismatches."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from news_recommendation_model_amd import ops  # noqa: F401  (registers torch.ops.nrm)

B, T, H, D = (int(a) for a in sys.argv[1:5])
g = torch.Generator(device="cuda").manual_seed(0)
r = lambda *s: torch.randn(*s, device="cuda", generator=g)          # noqa: E731
t, h = r(B, T, D), r(B, H, D)
w1, b1 = r(D, 4 * D) / (4 * D) ** 0.5, 0.1 * r(D)
w2, b2 = r(1, D) / D ** 0.5, r(1)
s0, z0 = torch.ops.nrm.pwattn_fwd(t, h, w1, b1, w2, b2, True, 0)
s2, z2 = torch.ops.nrm.pwattn_fwd(t, h, w1, b1, w2, b2, True, 2)
torch.cuda.synchronize()
bad = (z2 - z0).abs() > 1e-3 * z0.abs().max()
print("s err", float((s2 - s0).abs().max() / s0.abs().max()), "z bad frac", float(bad.float().mean()))
if bool(bad.any()):
    idx = bad.nonzero()
    print("first bad (b,t,h,k):", idx[:5].tolist(), "bad k values:", sorted(set(idx[:, 3].tolist()))[:40], "bad h:", sorted(set(idx[:, 2].tolist()))[:40],
          "bad t:", sorted(set(idx[:, 1].tolist()))[:40])
if bool(bad.any()):
    for b_, t_, h_, k_ in idx[:6].tolist():
        val = float(z2[b_, t_, h_, k_])
        hits = (z0 == val).nonzero()[:4].tolist()
        near = ((z0[b_] - val).abs() < 1e-6 * max(1.0, abs(val))).nonzero()[:4].tolist()
        print((b_, t_, h_, k_), "z2", val, "z0", float(z0[b_, t_, h_, k_]), "z2 row", z2[b_, t_, h_, k_:k_ + 4].tolist(), "same value in z0 at", hits, "near in impression", near)
    # is it stale memory?  run again into a fresh poisoned buffer
