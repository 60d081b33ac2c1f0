"""Timing of BASELINE configs[3]: CoCoOp ViT-B/16, batch 64 images x 11 classes = 704 text sequences per step, on one MI355X.

    python tools/cocoop_bench.py [--batch 64] [--classes 11] [--dtype bf16] [--steps 10]
Synthetic images, random-init CLIP.  Prints ms/step, images/s and the algorithmic TFLOP/s (vision forward only +
B*C text sequences forward + backward)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mudpt_amd import synth
from mudpt_amd.model import CustomCLIP, ModelShape


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--classes", type=int, default=11)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--steps", type=int, default=10)
    a = ap.parse_args()
    shape = ModelShape(depth=1)
    B, C = a.batch, a.classes
    tok = synth.bench_tokenized_prompts() if C == 11 else synth.synthetic_tokenized_prompts(C)
    m = CustomCLIP(shape, synth.random_clip_state(shape, 0), tok, ctx_token_ids=synth.CTX_INIT_TOKENS, max_batch=B, dtype=a.dtype,
                   seed=1, variant="cocoop")
    g = torch.Generator().manual_seed(0)
    images, labels = torch.randn(B, 3, 224, 224, generator=g).cuda(), torch.randint(0, C, (B,), generator=g).cuda()

    def step():
        loss = m.forward_backward(images, labels)
        m.sgd_step(0.002)
        return loss
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    dv, dtw, Lv, Lt = shape.v_width, shape.t_width, 197, shape.ctx_len
    vis = B * (Lv * shape.v_layers * (24 * dv * dv + 4 * Lv * dv) + 2 * 196 * 768 * dv)          # forward only
    txt = B * C * Lt * shape.t_layers * (2 * 24 * dtw * dtw + 3 * 4 * Lt * dtw)                   # forward + dX backward
    print(f"CoCoOp B={B} C={C} {a.dtype}: {dt * 1e3:.2f} ms/step, {B / dt:.0f} images/s, {B * C / dt:.0f} text sequences/s, "
          f"{(vis + txt) / dt / 1e12:.0f} TFLOP/s algorithmic ({(vis + txt) / 1e12:.2f} TFLOP/step), loss {loss.item():.4f}")
    m.close()


if __name__ == "__main__":
    main()
