"""ms per step (BASELINE configs[1]: ViT-B/16, B 256, C 11) of every row of DESIGN.md 2's precision-ablation table, in ONE process on one box.
The logit errors of the same rows come from tests/test_exact_gpu.py::test_precision_ablation_on_the_gpu (reference fixtures at logit scale 100)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mudpt_amd import synth  # noqa: E402
from mudpt_amd.model import CustomCLIP, ModelShape  # noqa: E402

ROWS = [
    ("bf16 (throughput mode)", "bf16", {}),
    ("fp16 mode: vision fp16; text pairs, fp16 attention", "fp16", {}),
    ("text exact; vision fp16 everywhere", "fp32", {"vis_lo": 0}),
    ("text exact; vision e4m3 lo at c_fc, c_proj", "fp32", {"vis_sites": 12}),
    ("text exact; vision e4m3 lo at c_fc, c_proj, patch", "fp32", {"vis_sites": 28}),
    ("text exact; vision e4m3 lo at out_proj, c_fc, c_proj, patch", "fp32", {"vis_sites": 30}),
    ("text exact; vision e4m3 lo at all four GEMMs", "fp32", {"vis_sites": 15}),
    ("PARITY MODE (dtype fp32): + split pixels", "fp32", {}),
    ("parity mode, fp32 gradient stream (lp_grad 0)", "fp32", {"lp_grad": 0}),
    ("parity mode with fp16 pairs in the vision tower", "fp32", {"vis_lo": 1}),
    ("parity mode + vision fp32 attention", "fp32", {"vis_exact_attn": 1}),
    ("round 3 exact: pairs + fp32 attention (+ fp32 gradient stream)", "fp32", {"vis_lo": 1, "vis_exact_attn": 1, "lp_grad": 0}),
]


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    shape = ModelShape()
    B, C = 256, 11
    tok = synth.bench_tokenized_prompts()
    state = synth.random_clip_state(shape, seed=0)
    g = torch.Generator().manual_seed(1234)
    images = torch.randn(B, 3, shape.image_size, shape.image_size, generator=g).cuda()
    labels = torch.randint(0, C, (B,), generator=g).cuda()
    out = []
    for label, dtype, knobs in ROWS:
        m = CustomCLIP(shape, state, tok, ctx_token_ids=synth.CTX_INIT_TOKENS, max_batch=B, dtype=dtype, device="cuda:0", seed=1, knobs=knobs)
        for _ in range(3):
            m.forward_backward(images, labels)
            m.sgd_step(0.0025, momentum=0.9, weight_decay=5e-4)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            m.forward_backward(images, labels)
            m.sgd_step(0.0025, momentum=0.9, weight_decay=5e-4)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        print(f"{label:<70s} {dtype} {knobs}: {ms:.2f} ms/step", flush=True)
        out.append({"row": label, "dtype": dtype, "knobs": knobs, "ms_per_step": round(ms, 3)})
        m.close()
        del m
    print(json.dumps(out))


if __name__ == "__main__":
    main()
