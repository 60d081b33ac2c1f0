"""Driver of tools/probes/mfma_f8.hip (GPU box): conversion, operand pairing / scales, relative rate of the MX-scaled fp8 MFMA."""
import ctypes as C
import os
import subprocess

import torch

here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "mfma_f8.so")
if not os.path.exists(so):
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", os.path.join(here, "mfma_f8.hip"), "-o", so])
lib = C.CDLL(so)
lib.probe_rate.restype = C.c_float
P = lambda t: C.c_void_p(t.data_ptr())

# 1. conversion: every value class (normal, subnormal, ties, beyond the maximum 448, negative)
g = torch.Generator().manual_seed(0)
x = torch.cat([torch.randn(4096, generator=g) * s for s in (1e-3, 0.02, 0.5, 4.0, 100.0, 1000.0)] +
              [torch.tensor([0.0, -0.0, 448.0, 464.0, 480.0, 1e6, -1e6, 2.0 ** -9, 2.0 ** -10, 1.5 * 2.0 ** -9, 0.0625 + 2.0 ** -8, 17.0, 18.0, 19.0])])
if x.numel() % 2:
    x = torch.cat([x, torch.zeros(1)])
xd = x.cuda()
y = torch.zeros(x.numel(), dtype=torch.uint8, device="cuda")
assert lib.probe_cvt(P(xd), P(y), x.numel()) == 0
ref = x.clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)
got = y.cpu()
bad = (got != ref).nonzero().flatten()
inrange = bad[(x[bad].abs() <= 448)]
print(f"   of those with |x| <= 448: {inrange.numel()}")
print(f"cvt_pk_fp8_f32 vs torch e4m3fn (inputs clamped to +-448): {bad.numel()} of {x.numel()} differ")
for i in bad[:10].tolist():
    print(f"   x {x[i].item():.6g}: got 0x{got[i].item():02x} = {got[i:i+1].view(torch.float8_e4m3fn).float().item()}, torch 0x{ref[i].item():02x} = {ref[i:i+1].view(torch.float8_e4m3fn).float().item()}")
big = torch.tensor([464.0, 480.0, 1e6, -1e6]).cuda()
yb = torch.zeros(4, dtype=torch.uint8, device="cuda")
lib.probe_cvt(P(big), P(yb), 4)
print("beyond the maximum (no clamp):", [f"0x{v:02x}" for v in yb.cpu().tolist()], "(0x7e = 448, 0x7f = NaN)")

# 2. one MFMA: random e4m3 operands, scales 2^-3 and 2^-5
A = (torch.randn(16, 128, generator=g) * 2).to(torch.float8_e4m3fn)
B = (torch.randn(16, 128, generator=g) * 0.5).to(torch.float8_e4m3fn)
D = torch.zeros(16, 16, device="cuda")
Ad, Bd = A.view(torch.uint8).cuda(), B.view(torch.uint8).cuda()  # both alive across the call
assert lib.probe_mfma(P(Ad), P(Bd), P(D), 127 - 3, 127 - 5) == 0
want = (A.double().float().double() @ B.float().double().t()) * 2.0 ** -8
err = (D.cpu().double() - want).abs().max().item()
print(f"mfma_scale 16x16x128 e4m3, lane (r, g) holds bytes k = 32 g .. 32 g + 31 of row r: max |D - A B^T 2^-8| = {err:.3e} (|D| max {want.abs().max():.3f})")
errT = (D.cpu().double().t() - want).abs().max().item()
print(f"   (against the transpose: {errT:.3e})")

# 3. relative rate
out = torch.zeros(4, device="cuda")
grid, iters = 256 * 2, 20000
for mode, name, flop in ((0, "scaled fp8 16x16x128", 2 * 16 * 16 * 128), (1, "f16 16x16x32", 2 * 16 * 16 * 32)):
    ms = lib.probe_rate(mode, grid, iters, P(out))
    tf = grid * 4 * iters * 8 * flop / (ms * 1e-3) / 1e12
    print(f"{name}: {ms:.2f} ms -> {tf:.0f} TFLOP/s (operands in registers, near-constant data: an upper bound)")
