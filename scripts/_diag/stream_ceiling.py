"""What a plain read+write stream over a z-sized tensor (2.46 GB at C3) reaches on this GPU: the practical ceiling for
bwd_dz_kernel (one pass over z, in place)."""
import torch
n = 1024 * 30 * 50 * 400
x = torch.randn(n, device="cuda")
y = torch.empty_like(x)
for name, fn in (("copy_ (read 2.46 + write 2.46 GB)", lambda: y.copy_(x)), ("mul_ in place", lambda: x.mul_(1.0001)),
                 ("gelu_backward (read 2, write 1)", lambda: torch.ops.aten.gelu_backward(y, x))):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"{name:40s} {ms:7.3f} ms")
