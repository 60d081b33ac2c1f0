import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
import torch
from mudpt_amd import capi
lib = capi.load()
P = lambda t: C.c_void_p(t.data_ptr())
B, L, H = 256, 201, 12
Lp = 224
qkv = torch.randn(B, L, 3 * H * 64, device="cuda").to(torch.bfloat16)
dout = torch.randn(B, L, H * 64, device="cuda").to(torch.bfloat16)
out = torch.randn(B, L, H * 64, device="cuda").to(torch.bfloat16)
dqkv = torch.empty_like(qkv)
lse = torch.randn(B, H, Lp, device="cuda") + 5
delta = torch.zeros(B, H, Lp, device="cuda")
def run(flags):
    def f():
        assert lib.mudpt_attention_bwd(0, P(qkv), P(out), P(dout), P(lse), P(delta), P(dqkv), B, L, H, flags, None) == 0
    f(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10)
    return best * 1e3
for name, dbg in (("full", 0), ("no sweep", 1), ("loads + sweep only (return before K reload)", 2), ("no final dQ loop", 4), ("loads only", 1 | 2), ("no sweep, no final (loads + K reload)", 1 | 4)):
    print(f"{name}: {run(16 | (dbg << 8)):.1f} us", flush=True)
print(f"two kernels: {run(2):.1f} us")
