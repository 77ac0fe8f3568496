#!/usr/bin/env python3
"""Diagnostic: one slab reduction (nrm_slab_reduce) on synthetic slabs: time and GB/s of slab bytes.
    usage: slab_reduce_bench.py [nsplit nj ni] ...   defaults: the C3 sets"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from news_recommendation_model_amd import native

native.load()
lib = native.load()
args = [int(a) for a in sys.argv[1:]]
sets = [tuple(args[i:i + 3]) for i in range(0, len(args), 3)] or [
    (324, 400, 400), (lib.nrm_gemm_tn_nsplit(1608, 402, 30720, 0), 402, 1608), (lib.nrm_gemm_tn_nsplit(402, 1608, 30720, 0), 1608, 402),
    (lib.nrm_gemm_tn_nsplit(400, 400, 51200, 0), 400, 400), (1920, 64, 64), (400, 64, 64)]
for nsplit, nj, ni in sets:
    ldws = (ni + 3) // 4 * 4
    ws = torch.randn(nsplit, nj, ldws, device="cuda")
    out = torch.zeros(ni, nj, device="cuda")
    st = native.stream_ptr()

    def run():
        native.call("nrm_slab_reduce", native.ptr(ws), nsplit, nj, ldws, ni, native.ptr(out), nj, 1, None, 0, 0, 0.0, None, None, st)
    run()
    torch.cuda.synchronize()
    ref = ws[:, :, :ni].sum(0).t()
    err = float((out - ref).abs().max() / ref.abs().max())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"nsplit={nsplit} [{nj} x {ni}] {ws.numel() * 4 / 1e6:.1f} MB: {ms * 1e3:.1f} us = {ws.numel() * 4 / ms / 1e6:.0f} GB/s  rel err {err:.1e}", flush=True)
