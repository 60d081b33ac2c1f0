"""Diagnostic (GPU box): per-block error of the HIP path against the CPU oracle on the ViT-B/16 B=4 golden case."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import mudpt_oracle as O
from tests.helpers import GoldenCase
from tests.test_model_gpu import build

case = GoldenCase(sys.argv[1] if len(sys.argv) > 1 else "mudpt_vitb16_b4")
taps = {}
with torch.no_grad():
    ref_logits = O.forward(case.cfg, case.frozen, case.params, case.class_embedding, case.eot, case.images, taps)
B = len(case.labels)
for dtype in ("fp16", "bf16"):
    m = build(case, dtype)
    logits = m(case.images).cpu()
    print(f"== {dtype}: logits max err {(logits - ref_logits).abs().max():.3e} rms {(logits - ref_logits).pow(2).mean().sqrt():.3e}")
    for tower, pre, nl, nseq in (("vis", "visual.transformer", case.cfg.v_layers, B), ("txt", "transformer", case.cfg.t_layers, 11)):
        for i in range(nl):
            if i + 1 == nl:
                continue  # the last block output exists on the CLS / EOT rows only (debug_read "x_out"): covered by the feature errors below
            name = f"{tower}.x_in.{i + 1}"
            got = m.debug_read(name, B).view(nseq, -1, taps[f"{pre}.resblocks.{i}.out"].shape[-1])
            ref = taps[f"{pre}.resblocks.{i}.out"].clone()
            if i + 1 < nl and i < case.cfg.depth - 1:  # x_in.{i+1} holds the block output with the next layer's prompts spliced in
                n = case.cfg.n_ctx
                if tower == "vis":
                    got, ref = got[:, :-n], ref[:, :-n]
                else:
                    got, ref = torch.cat([got[:, :1], got[:, 1 + n:]], 1), torch.cat([ref[:, :1], ref[:, 1 + n:]], 1)
            rel = (got - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()
            cls = (got[:, 0] - ref[:, 0]).pow(2).mean().sqrt() / ref[:, 0].pow(2).mean().sqrt()
            print(f"  {tower} block {i:2d}: rel rms err all rows {rel:.3e}  row0 {cls:.3e}")
    for k, r in (("image_features", taps["image_features"]), ("text_features", taps["text_features"])):
        g = m.debug_read(k, B).view_as(r)
        print(f"  {k}: rel err {(g - r).pow(2).mean().sqrt() / r.pow(2).mean().sqrt():.3e}")
    m.close()
