"""Which byte of which lane multiplies which: the operand layout of v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3), probed with one-hot operands."""
import ctypes as C
import os
import torch

here = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(here, "mfma_f8.so"))
P = lambda t: C.c_void_p(t.data_ptr())
ONE = 0x38  # e4m3 1.0


def run(A, B, sa=127, sb=127):
    D = torch.zeros(16, 16, device="cuda")
    Ad, Bd = A.cuda(), B.cuda()  # keep both alive across the call (a temporary's block is reused by the next allocation)
    assert lib.probe_mfma(P(Ad), P(Bd), P(D), sa, sb) == 0
    return D.cpu()


ones = torch.full((16, 128), ONE, dtype=torch.uint8)
print("A = 1, B = 1:", run(ones, ones).unique().tolist(), "(expect 128)")
print("scale_a 126:", run(ones, ones, 126, 127).unique().tolist(), " scale_b 129:", run(ones, ones, 127, 129).unique().tolist())
# scale register bytes: only byte 0 used with opsel 0?
print("scale_a = 0x7f7f7e7f -> ", run(ones, ones, 0x7f7f7e7f, 127).unique().tolist(), " 0x7e7f7f7f ->", run(ones, ones, 0x7e7f7f7f, 127).unique().tolist())
# memory position p (0..127) of the probe's row = lane group p // 32, byte p % 32 of that lane's 32-byte fragment.
# pair[pa] = set of B positions pb with which A position pa is multiplied
pair = {}
for pa in list(range(0, 128, 1)):
    A = torch.zeros(16, 128, dtype=torch.uint8)
    A[:, pa] = ONE
    # B position pb carries the value code: use 4 runs with binary digits of pb (values 1 or 0) -> identify pb from 7 bits
    bits = []
    for bit in range(7):
        B = torch.zeros(16, 128, dtype=torch.uint8)
        sel = torch.tensor([(pb >> bit) & 1 for pb in range(128)], dtype=torch.bool)
        B[:, sel] = ONE
        bits.append(run(A, B)[0, 0].item())
    tot = run(A, ones)[0, 0].item()
    pb = sum(int(round(b)) << i for i, b in enumerate(bits))
    pair[pa] = (pb, tot)
bad = {pa: v for pa, v in pair.items() if v[0] != pa or v[1] != 1.0}
print("A position -> B position it multiplies (only where != identity):", bad if bad else "identity for all 128 positions")
# rows / columns: D[i][j] = sum_k A[i][k] B[j][k] with the probe's store map?
A = torch.zeros(16, 128, dtype=torch.uint8)
B = torch.zeros(16, 128, dtype=torch.uint8)
A[3, :] = ONE
B[5, :] = ONE
D = run(A, B)
print("A row 3 x B row 5 -> nonzero D entries:", D.nonzero().tolist(), D[D != 0].tolist())
