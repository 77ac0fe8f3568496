"""Diagnostic only: cycle shares of the forward attention kernel (stamped build, see run_stamp.py)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from news_recommendation_model_amd import native
native.LIB_PATH = os.path.join(ROOT, "scripts", "_diag", "libnrm_stamp_fwd.so")
from news_recommendation_model_amd import ops
lib = native.load()
lib.nrm_debug_read.restype = ctypes.c_int
lib.nrm_debug_read.argtypes = [ctypes.c_void_p, ctypes.c_long]
B, T, H, D = 1024, 30, 50, 400
torch.manual_seed(0)
k1, k2 = 1 / np.sqrt(4 * D), 1 / np.sqrt(D)
w1 = ((torch.rand(D, 4 * D, device="cuda") * 2 - 1) * k1).requires_grad_(True)
b1 = (torch.rand(D, device="cuda") * 2 - 1) * k1
w2 = (torch.rand(1, D, device="cuda") * 2 - 1) * k2
b2 = (torch.rand(1, device="cuda") * 2 - 1) * k2
t = torch.randn(B, T, D, device="cuda")
h = torch.randn(B, H, D, device="cuda")
for it in range(3):
    s = ops.pointwise_attention_scores(t, h, w1, b1, w2, b2)
torch.cuda.synchronize()
n = (8 << 20) // 8
buf = np.zeros(n, dtype=np.uint64)
lib.nrm_debug_read(buf.ctypes.data, n)
d = buf.reshape(-1, 8)
d = d[d[:, 5] > 0].astype(np.float64)
print(f"forward: {len(d)} waves; ticks per wave (one 16-row x 400-col tile, 2500 MFMAs = 80000 ticks of exclusive pipe):")
for i, lab in enumerate(("acc init (u+v loads)", "first DMA + barrier", "K loop (25 chunks)", "epilogue (z store, GELU, dot)")):
    print(f"   {lab:32s} {d[:, i].mean():9.0f}  ({100 * d[:, i].sum() / d[:, 4].sum():5.1f} %)")
print(f"   total                            {d[:, 4].mean():9.0f}")
