"""GEMM micro-benchmark through the C ABI (GPU box): the model's GEMM shapes at B = 256, random operands.

    python tools/gemm_bench.py [--dtype bf16] [--variants 0,1] [--rounds 5]
Variants are selected with mudpt_gemm's variant argument and interleaved in one process (A/B rule)."""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mudpt_amd import capi

SHAPES = [  # (name, M, N, K, epilogue)
    ("qkv    fwd", 51456, 2304, 768, 0),
    ("out    fwd", 51456, 768, 768, 5),
    ("fc     fwd", 51456, 3072, 768, 1),
    ("proj   fwd", 51456, 768, 3072, 5),
    ("dgelu  bwd", 51456, 3072, 768, 3),
    ("dfc    bwd", 51456, 768, 3072, 0),
    ("dout   bwd", 51456, 768, 768, 0),
    ("dqkv   bwd", 51456, 768, 2304, 0),
]

# the text tower after the trim to max(EOT) + 1 positions: 11 prompts x 9 rows (headline), 1000 x 19 (configs[2]), CoCoOp 64 x 11 x 9
TEXT_SHAPES = [(f"{nm} M={M}", M, N, K, e) for M in (99, 6336, 19000)
               for nm, N, K, e in (("qkv", 1536, 512, 0), ("out", 512, 512, 5), ("fc", 2048, 512, 1), ("proj", 512, 2048, 5))]


# the reference's own training batch (B 4: M = 4 x 201 = 804 rows) and the 50-class text tower beside it (50 prompts x ~20 positions)
SMALL_SHAPES = [(f"{nm} M={M}", M, N, K, e) for M, w in ((804, 768), (1000, 512), (15000, 768), (9000, 768), (4000, 768))
                for nm, N, K, e in (("qkv", 3 * w, w, 0), ("out", w, w, 5), ("fc", 4 * w, w, 1), ("proj", w, 4 * w, 5), ("dgelu", 4 * w, w, 3), ("dfc", w, 4 * w, 0), ("dqkv", w, 3 * w, 0))]


# between the small grids and the persistent kernel's territory: the text tower at 1000 classes (19 000 rows; width 512 with ViT-B, 768 with
# ViT-L/14@336), CoCoOp's 64 x 11 prompts (6 336 rows) and its 64-image vision tower (12 608 rows)
MID_SHAPES = [(f"{nm} M={M} w={w}", M, N, K, e) for M, w in ((19000, 512), (19000, 768), (6336, 512), (12608, 768))
              for nm, N, K, e in (("qkv", 3 * w, w, 0), ("out", w, w, 5), ("fc", 4 * w, w, 1), ("proj", w, 4 * w, 5), ("dgelu", 4 * w, w, 3), ("dfc", w, 4 * w, 0), ("dqkv", w, 3 * w, 0))]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--variants", default="0")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--only", default="")
    ap.add_argument("--epi", type=int, default=-1, help="only the shapes with this epilogue")
    ap.add_argument("--set", default="vision", choices=["vision", "text", "small", "mid"])
    ap.add_argument("--library-ref", action="store_true", help="also time torch.nn.functional.linear (rocBLAS / hipBLASLt) on the same operands: a "
                    "known-good reference for what this GPU does on the shape (diagnostic only; the product never calls it)")
    a = ap.parse_args()
    shapes = {"vision": SHAPES, "text": TEXT_SHAPES, "small": SMALL_SHAPES, "mid": MID_SHAPES}[a.set]
    lib = capi.load()
    dt, tt = (0, torch.bfloat16) if a.dtype == "bf16" else (1, torch.float16)
    variants = [int(v) for v in a.variants.split(",")]
    P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    total = {v: 0.0 for v in variants}
    for name, M, N, K, epi in shapes:
        if (a.only and a.only not in name) or (a.epi >= 0 and epi != a.epi):
            continue
        A = torch.randn(M, K, device="cuda").to(tt)
        B = (torch.randn(N, K, device="cuda") * K ** -0.5).to(tt)
        bias = torch.randn(N, device="cuda")
        f32 = epi in (2, 5)
        out0 = torch.empty(M, N, device="cuda", dtype=torch.float32 if f32 else tt)
        out1 = torch.empty(M, N, device="cuda", dtype=tt) if epi == 1 else None
        aux = torch.randn(M, N, device="cuda").to(torch.float32 if epi == 2 else tt) if epi in (2, 3) else None

        def run(v=0):
            rc = lib.mudpt_gemm(dt, epi, M, N, K, P(A), K, P(B), K, P(bias) if epi != 3 else None, P(out0), N, P(out1), N if out1 is not None else 0,
                                P(aux), N if aux is not None else 0, 0, 0, None, v, None)
            assert rc == 0, lib.mudpt_last_error()
        best = {v: 1e9 for v in variants}
        for r in range(a.rounds):
            for v in variants:
                run(v)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.iters):
                    run(v)
                e1.record()
                torch.cuda.synchronize()
                best[v] = min(best[v], e0.elapsed_time(e1) / a.iters)
        fl = 2.0 * M * N * K
        if a.library_ref:
            lin_b = bias.to(tt)
            tbest = 1e9
            for r in range(a.rounds):
                torch.nn.functional.linear(A, B, lin_b)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.iters):
                    torch.nn.functional.linear(A, B, lin_b)
                e1.record()
                torch.cuda.synchronize()
                tbest = min(tbest, e0.elapsed_time(e1) / a.iters)
            print(f"    library (torch linear, bias, {a.dtype} out): {tbest * 1e3:7.1f} us {fl / tbest / 1e9:7.1f} TF/s", flush=True)
        print(f"{name} M={M} N={N} K={K} epi={epi}: " + "  ".join(f"v{v}: {best[v] * 1e3:7.1f} us {fl / best[v] / 1e9:7.1f} TF/s" for v in variants), flush=True)
        for v in variants:
            total[v] += best[v]
    layer_fl = sum(2.0 * M * N * K for _, M, N, K, _ in shapes)
    print("sum over the 8 GEMMs of one block: " + "  ".join(f"v{v}: {total[v] * 1e3:.0f} us ({layer_fl / total[v] / 1e9:.0f} TF/s)" for v in variants))


if __name__ == "__main__":
    main()
