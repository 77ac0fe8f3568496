#!/usr/bin/env python3
"""Diagnostic: only the attention ops (forward [+ backward]) at a BASELINE shape, for rocprofv3 / PMC runs that should not
carry the rest of the model.   usage: attn_only.py <B> <T> <H> <D> <mma f32|bf16|bf16x3> [iters] [bwd]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from news_recommendation_model_amd import ops

B, T, H, D = (int(a) for a in sys.argv[1:5])
mma = sys.argv[5] if len(sys.argv) > 5 else "f32"
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 10
bwd = len(sys.argv) > 7
g = torch.Generator(device="cuda").manual_seed(0)
r = lambda *s: torch.randn(*s, device="cuda", generator=g)          # noqa: E731
t, h = r(B, T, D).requires_grad_(bwd), r(B, H, D).requires_grad_(bwd)
w1, b1 = (r(D, 4 * D) / (4 * D) ** 0.5).requires_grad_(True), (0.1 * r(D)).requires_grad_(True)
w2, b2 = (r(1, D) / D ** 0.5).requires_grad_(True), r(1).requires_grad_(True)
gs = r(B, T, H)
for i in range(iters + 2):
    if i == 2:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
    s = ops.pointwise_attention_scores(t, h, w1, b1, w2, b2, mma=mma)
    if bwd:
        s.backward(gs)
torch.cuda.synchronize()
print(f"{mma} B={B} T={T} H={H} D={D} {'fwd+bwd' if bwd else 'fwd'}: {(time.perf_counter() - t0) / iters * 1e3:.3f} ms per iteration")
