"""Diagnostic only: load the s_memtime-stamped build of the library (scripts/_diag/libnrm_stamp.so, built outside the
repo from a patched copy of csrc/) and print where a wave of the backward contraction spends its cycles."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from news_recommendation_model_amd import native
native.LIB_PATH = os.path.join(ROOT, "scripts", "_diag", "libnrm_stamp.so")
from news_recommendation_model_amd import ops
lib = native.load()
lib.nrm_debug_read.restype = ctypes.c_int
lib.nrm_debug_read.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_long]
B, T, H, D = 1024, 30, 50, 400
torch.manual_seed(0)
k1, k2 = 1 / np.sqrt(4 * D), 1 / np.sqrt(D)
w1 = ((torch.rand(D, 4 * D, device="cuda") * 2 - 1) * k1).requires_grad_(True)
b1 = (torch.rand(D, device="cuda") * 2 - 1) * k1
w2 = (torch.rand(1, D, device="cuda") * 2 - 1) * k2
b2 = (torch.rand(1, device="cuda") * 2 - 1) * k2
t = torch.randn(B, T, D, device="cuda", requires_grad=True)
h = torch.randn(B, H, D, device="cuda", requires_grad=True)
for it in range(3):
    s = ops.pointwise_attention_scores(t, h, w1, b1, w2, b2)
    s.backward(torch.randn_like(s))
torch.cuda.synchronize()
for which, name in ((0, "bt (dt + dW_p)"), (1, "bh (dh)")):
    n = (1 << 22) // 8
    buf = np.zeros(n, dtype=np.uint64)
    rc = lib.nrm_debug_read(which, buf.ctypes.data, n)
    d = buf.reshape(-1, 8)
    d = d[d[:, 5] > 0].astype(np.float64)
    g = d[:, 5]
    print(f"{name}: {len(d)} waves, groups/wave {g.mean():.1f}; cycles per group (s_memtime ticks = shader clocks):")
    for i, lab in enumerate(("step loop (MFMA + operand waits)", "next-group prefetch issue", "epilogue (LDS, FMA, shuffles)", "flush (bounce + atomics)")):
        print(f"   {lab:36s} {np.mean(d[:, i] / g):9.0f}  ({100 * d[:, i].sum() / d[:, 4].sum():5.1f} % of wave time)")
    print(f"   total per group                      {np.mean(d[:, 4] / g):9.0f}   wave lifetime {d[:,4].mean()/1e6:.2f} M ticks (min {d[:,4].min()/1e6:.2f}, max {d[:,4].max()/1e6:.2f})")
