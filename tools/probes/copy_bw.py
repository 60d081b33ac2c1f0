"""HBM streaming ceiling on this box: torch's device copy / add kernels at the LayerNorm kernels' footprint (diagnostic)."""
import torch
for mb in (237, 474):
    n = mb * (1 << 20) // 4 // 2
    x = torch.randn(n, device="cuda"); y = torch.empty_like(x); z = torch.randn(n, device="cuda")
    for name, fn, byt in (("copy", lambda: y.copy_(x), 8 * n), ("add", lambda: torch.add(x, z, out=y), 12 * n)):
        fn(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): fn()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10)
        print(f"{name} {byt / 1e6:.0f} MB: {best * 1e3:.1f} us = {byt / best / 1e9:.2f} TB/s", flush=True)
