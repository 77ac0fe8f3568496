#!/usr/bin/env python3
"""Diagnostic: every dense GEMM shape of a C3 training step, one C-ABI launch kind at a time (gemm_nt per epilogue, gemm_tn),
mean launch time over `iters` launches (event pair around the batch), TFLOP/s and fraction of the fp32 MFMA peak; with the
product library also a check against torch.matmul in float64.   usage: gemm_shapes.py [iters] [only-substring]
NRM_HOTPATH_LIB=scripts/_diag/libnrm_<tag>.so NRM_ALLOW_DIAG_LIB=1 selects a variant build."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from news_recommendation_model_amd import native, ops

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
only = sys.argv[2] if len(sys.argv) > 2 else ""
PEAK = 157.3
BT, BH = 30720, 51200
# (name, M, K, N, epilogue, launches per step)
NT = [("w1 fwd", BH, 402, 400, ops.EPI_BIAS, 1), ("u / dh side", BH, 400, 400, ops.EPI_BIAS, 3), ("v / dt side", BT, 400, 400, ops.EPI_BIAS, 3),
      ("w1 dX", BH, 400, 402, ops.EPI_BIAS, 1),
      ("fc1 fwd (gelu)", BT, 1608, 402, ops.EPI_GELU, 3), ("gate.fc2 fwd (mul)", BT, 402, 1608, ops.EPI_MUL, 1),
      ("mlp.fc2 fwd / fc1 dX", BT, 402, 1608, ops.EPI_BIAS, 4), ("fc2 dX (dgelu)", BT, 1608, 402, ops.EPI_DGELU, 2)]
# (name, R, ni, nj, colsum, launches per step)
TN = [("fc2 dW [1608x402]", BT, 1608, 402, True, 2), ("fc1 dW [402x1608]", BT, 402, 1608, True, 3), ("w1 dW [400x402]", BH, 400, 402, True, 1),
      ("du^T h [400x400]", BH, 400, 400, True, 2), ("dv^T t [400x400]", BT, 400, 400, False, 2)]
diag = bool(os.environ.get("NRM_HOTPATH_LIB")) or os.environ.get("NRM_GEMM_NOCHECK") == "1"     # (the float64 check between two timings lowers the next one: compare variants with the same protocol)
tot = 0.0
tot_peak = 0.0


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for name, M, K, N, epi, per_step in NT:
    if only and only not in name:
        continue
    x = torch.randn(M, ops._pad4(K), device="cuda")[:, :K]
    x = ops._rows(x)
    w = torch.randn(N, K, device="cuda") / K ** 0.5
    b = torch.randn(N, device="cuda")
    zsav = torch.randn(M, ops._pad4(N), device="cuda")[:, :N] if epi == ops.EPI_DGELU else None
    mul = torch.randn(M, ops._pad4(N), device="cuda")[:, :N] if epi == ops.EPI_MUL else None
    run = lambda: ops._gemm_nt(x, w, K, 1, N, K, None if epi == ops.EPI_DGELU else b, epi, z=zsav, m=mul, owner=w)   # noqa: E731
    ms = timed(run)
    gf = 2.0 * M * K * N / 1e9
    err = ""
    if not diag:
        y, z = run()
        ref = x.double() @ w.double().t()
        if epi != ops.EPI_DGELU:
            ref = ref + b.double()
        if epi == ops.EPI_GELU:
            ref = torch.nn.functional.gelu(ref)
        elif epi == ops.EPI_MUL:
            ref = ref * mul.double()
        elif epi == ops.EPI_DGELU:
            zz = zsav.double()
            ref = ref * (0.5 * (1 + torch.erf(zz / 2 ** 0.5)) + zz * torch.exp(-0.5 * zz * zz) / (2 * torch.pi) ** 0.5)
        err = f" rel_err {float((y.double() - ref).abs().max() / ref.abs().max()):.1e}"
    print(f"gemm_nt {name:24s} M={M} K={K} N={N}: {ms * 1e3:7.1f} us  {gf / ms:6.1f} TF/s  {gf / ms / PEAK:.3f} of peak  (x{per_step}/step){err}", flush=True)
    tot += ms * per_step
    tot_peak += gf / PEAK * per_step
for name, R, ni, nj, colsum, per_step in TN:
    if only and only not in name:
        continue
    a = ops._rows(torch.randn(R, ops._pad4(ni), device="cuda")[:, :ni])
    bm = ops._rows(torch.randn(R, ops._pad4(nj), device="cuda")[:, :nj])
    out = torch.empty(ni, nj, device="cuda")
    run = lambda: ops._gemm_tn_slabs(a, bm, colsum, zero_out=out)     # noqa: E731
    ms = timed(run)
    gf = 2.0 * R * ni * nj / 1e9
    err = ""
    if not diag:
        c, cs = ops._gemm_tn(a, bm, colsum)
        ref = a.double().t() @ bm.double()
        err = f" rel_err {float((c.double() - ref).abs().max() / ref.abs().max()):.1e}"
        if colsum:
            err += f" / {float((cs.double() - a.double().sum(0)).abs().max() / a.double().sum(0).abs().max()):.1e}"
    nsplit = native.load().nrm_gemm_tn_nsplit(ni, nj, R, 0)
    print(f"gemm_tn {name:24s} R={R} {ni}x{nj} ({nsplit} splits): {ms * 1e3:7.1f} us  {gf / ms:6.1f} TF/s  {gf / ms / PEAK:.3f} of peak  (x{per_step}/step){err}", flush=True)
    tot += ms * per_step
    tot_peak += gf / PEAK * per_step
print(f"{os.path.basename(os.environ.get('NRM_HOTPATH_LIB', 'product'))}: dense GEMMs of a C3 step: {tot:.3f} ms (at the fp32 MFMA peak: {tot_peak:.3f} ms, {tot_peak / tot:.3f})", flush=True)
