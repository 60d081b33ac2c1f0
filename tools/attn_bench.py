"""Attention micro-benchmark through the C ABI (GPU box): vision shape B=256, L=201, H=12 and text 11 x 77 x 8."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mudpt_amd import capi


def main():
    lib = capi.load()
    P = lambda t: C.c_void_p(t.data_ptr())
    if "--long" in sys.argv:  # BASELINE configs[4]: ViT-L/14@336, B 128, L 581, H 16 (the tiled kernels)
        B, L, H = 128, 581, 16
        Lp = lib.mudpt_attention_padded_len(L)
        qkv = torch.randn(B, L, 3 * H * 64, device="cuda").to(torch.bfloat16)
        dout = torch.randn(B, L, H * 64, device="cuda").to(torch.bfloat16)
        out = torch.empty(B, L, H * 64, device="cuda", dtype=torch.bfloat16)
        dqkv, lse, delta = torch.empty_like(qkv), torch.zeros(B, H, Lp, device="cuda"), torch.zeros(B, H, Lp, device="cuda")
        fl = 4.0 * B * H * L * L * 64
        for nm, fn, mult in (("fwd", lambda: lib.mudpt_attention_fwd(0, P(qkv), P(out), P(lse), B, L, H, 0, None), 1.0),
                             ("fwd 16-block tiled", lambda: lib.mudpt_attention_fwd(0, P(qkv), P(out), P(lse), B, L, H, 2, None), 1.0),
                             ("bwd (dQ kernel + dK/dV kernel)", lambda: lib.mudpt_attention_bwd(0, P(qkv), P(out), P(dout), P(lse), P(delta), P(dqkv), B, L, H, 0, None), 3.5)):
            assert fn() == 0
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 5)
            print(f"ViT-L/14@336 B {B} L {L} H {H} {nm}: {best * 1e3:7.1f} us ({mult * fl / best / 1e9:6.1f} TF/s executed)", flush=True)
        return
    for name, B, L, H, causal in (("vision", 256, 201, 12, 0), ("vision, head-contiguous (B*H seqs, H=1)", 3072, 201, 1, 0), ("text", 11, 77, 8, 1), ("text1000", 1000, 77, 8, 1)):
        Lp = lib.mudpt_attention_padded_len(L)
        qkv = torch.randn(B, L, 3 * H * 64, device="cuda").to(torch.bfloat16)
        dout = torch.randn(B, L, H * 64, device="cuda").to(torch.bfloat16)
        out = torch.empty(B, L, H * 64, device="cuda", dtype=torch.bfloat16)
        dqkv = torch.empty_like(qkv)
        lse = torch.zeros(B, H, Lp, device="cuda")
        delta = torch.zeros(B, H, Lp, device="cuda")

        def fwd(form=0):
            assert lib.mudpt_attention_fwd(0, P(qkv), P(out), P(lse), B, L, H, causal | form, None) == 0

        def bwd(form=0):
            assert lib.mudpt_attention_bwd(0, P(qkv), P(out), P(dout), P(lse), P(delta), P(dqkv), B, L, H, causal | form, None) == 0
        res = []
        for fn in (fwd, lambda: bwd(8), lambda: bwd(2), lambda: bwd(4), lambda: bwd(0 if causal else 16)):
            fn()
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 10)
            res.append(best)
        fl = 4.0 * B * H * L * L * 64
        print(f"{name}: fwd {res[0] * 1e3:7.1f} us ({fl / res[0] / 1e9:6.1f} TF/s)   bwd fused {res[1] * 1e3:7.1f} us ({2.5 * fl / res[1] / 1e9:6.1f} TF/s algorithmic)   two kernels {res[2] * 1e3:7.1f} us   fused, 2 blocks / wave {res[3] * 1e3:7.1f} us   {'default' if causal else 'single sweep'} {res[4] * 1e3:7.1f} us", flush=True)


if __name__ == "__main__":
    main()
