#!/usr/bin/env python3
"""Diagnostic: the dz pass (nrm_pwattn_bwd_dz) at a BASELINE shape, slab form (NRM_DZ_ROWS=0) against the full-row form (=1):
mean launch time and algorithmic TB/s (read z + write dz).   usage: dz_bench.py [B T H D] [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from news_recommendation_model_amd import native

B, T, H, D = (int(v) for v in sys.argv[1:5]) if len(sys.argv) > 4 else (1024, 30, 50, 400)
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 20
native.load()
z0 = torch.randn(B, T, H, D, device="cuda")
ds = torch.randn(B, T, H, device="cuda")
w2 = torch.randn(D, device="cuda")
out = {}
for mode in ("0", "1"):
    os.environ["NRM_DZ_ROWS"] = mode
    z = z0.clone()
    acc = torch.zeros(D + 4, device="cuda")
    du = torch.empty(B, H, D, device="cuda")
    dv = torch.empty(B, T, D, device="cuda")
    run = lambda: native.call("nrm_pwattn_bwd_dz", native.ptr(z), native.ptr(ds), native.ptr(w2), native.ptr(acc[:D]), native.ptr(acc[D:]),   # noqa: E731
                              native.ptr(du), native.ptr(dv), B, T, H, D, 0, native.stream_ptr())
    run()
    torch.cuda.synchronize()
    out[mode] = (z.clone(), du.clone(), dv.clone(), acc.clone())
    z.copy_(z0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        run()
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"dz pass B={B} T={T} H={H} D={D} NRM_DZ_ROWS={mode}: {ms:.4f} ms  {2 * 4.0 * B * T * H * D / ms / 1e9:.2f} TB/s algorithmic", flush=True)
for i, nm in enumerate(("dz", "du", "dv", "dw2|db2")):
    a, b = out["0"][i], out["1"][i]
    print(f"  {nm}: max rel diff slab vs rows {float((a - b).abs().max() / a.abs().max()):.2e}")
