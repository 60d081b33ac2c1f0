"""Synthetic stand-ins for what needs the network in the reference: a random-initialised CLIP state dict
(no ViT-B-16.pt offline; clip/clip.py:31-41 are download URLs) and class prompts without the BPE vocabulary.

Used by bench.py and the Dassl-free harness only; real runs pass a checkpoint's state dict instead."""
from __future__ import annotations

import math
from typing import Dict

import torch

from .model import ModelShape

# first Caltech-101 class names (datasets/caltech101.py:8-14 renaming applied), the benchmark's 11 classes
BENCH_CLASSNAMES = ["face", "leopard", "motorbike", "accordion", "airplane", "anchor", "ant", "barrel", "bass", "beaver",
                    "binocular"]
# clip.tokenize("a photo of a <name>.") for the names above (clip/clip.py:199-239): SOT 49406, "a photo of a"
# = 320 1125 539 320, name, "." = 269, EOT 49407.  Values recorded from the reference tokenizer (tests/golden).
_BENCH_NAME_TOKENS = [[1710], [15931], [33341], [48760], [16451], [13201], [773], [11703], [5992], [22874], [29172, 10940]]
CTX_INIT_TOKENS = [320, 1125, 539, 320]  # "a photo of a"
VOCAB = 49408


def bench_tokenized_prompts(ctx_len: int = 77) -> torch.Tensor:
    tok = torch.zeros(len(_BENCH_NAME_TOKENS), ctx_len, dtype=torch.int32)
    for i, name in enumerate(_BENCH_NAME_TOKENS):
        ids = [49406] + CTX_INIT_TOKENS + name + [269, 49407]
        tok[i, :len(ids)] = torch.tensor(ids, dtype=torch.int32)
    return tok


def synthetic_tokenized_prompts(n_cls: int, n_ctx: int = 4, ctx_len: int = 77, seed: int = 7) -> torch.Tensor:
    """Prompts shaped like "<ctx words> <name tokens> ." for synthetic class lists (e.g. 1000 ImageNet-sized).  Name lengths
    follow a long-tailed mix like BPE-tokenised ImageNet names: 70 % 1-3 tokens, 25 % 4-6, 5 % 7-12 (the longest name sets
    the number of positions the causal text tower has to run, see mudpt_set_class_prompts)."""
    g = torch.Generator().manual_seed(seed)
    tok = torch.zeros(n_cls, ctx_len, dtype=torch.int32)
    for c in range(n_cls):
        u = float(torch.rand((), generator=g))
        lo, hi = (1, 3) if u < 0.70 else ((4, 6) if u < 0.95 else (7, 12))
        k = lo + int(torch.randint(0, hi - lo + 1, (1,), generator=g))
        ids = [49406] + (CTX_INIT_TOKENS * n_ctx)[:n_ctx] + [int(v) for v in torch.randint(1000, 48000, (k,), generator=g)] + [269, 49407]
        tok[c, :len(ids)] = torch.tensor(ids, dtype=torch.int32)
    return tok


def clip_state_table(shape: ModelShape):
    """(key, tensor shape, init) of every entry of a CLIP ViT state dict (clip/model.py:667-808): init is a std for N(0, std^2)
    entries, "ones" / "zeros" for LayerNorm affine terms and attention biases."""
    dv, dt, e = shape.v_width, shape.t_width, shape.embed_dim
    P = (shape.image_size // shape.patch) ** 2
    rows = [
        ("visual.conv1.weight", (dv, 3, shape.patch, shape.patch), (3 * shape.patch ** 2) ** -0.5),
        ("visual.class_embedding", (dv,), dv ** -0.5),
        ("visual.positional_embedding", (P + 1, dv), dv ** -0.5),
        ("visual.ln_pre.weight", (dv,), "ones"), ("visual.ln_pre.bias", (dv,), "zeros"),
        ("visual.ln_post.weight", (dv,), "ones"), ("visual.ln_post.bias", (dv,), "zeros"),
        ("visual.proj", (dv, e), dv ** -0.5),
        ("token_embedding.weight", (VOCAB, dt), 0.02),
        ("positional_embedding", (shape.ctx_len, dt), 0.01),
        ("ln_final.weight", (dt,), "ones"), ("ln_final.bias", (dt,), "zeros"),
        ("text_projection", (dt, e), dt ** -0.5),
    ]
    for prefix, d, layers in (("visual.transformer", dv, shape.v_layers), ("transformer", dt, shape.t_layers)):
        for i in range(layers):
            p = f"{prefix}.resblocks.{i}."
            rows += [(p + "ln_1.weight", (d,), "ones"), (p + "ln_1.bias", (d,), "zeros"),
                     (p + "ln_2.weight", (d,), "ones"), (p + "ln_2.bias", (d,), "zeros"),
                     (p + "attn.in_proj_weight", (3 * d, d), d ** -0.5), (p + "attn.in_proj_bias", (3 * d,), "zeros"),
                     (p + "attn.out_proj.weight", (d, d), d ** -0.5 * (2 * layers) ** -0.5), (p + "attn.out_proj.bias", (d,), "zeros"),
                     (p + "mlp.c_fc.weight", (4 * d, d), (2 * d) ** -0.5), (p + "mlp.c_fc.bias", (4 * d,), 0.02),
                     (p + "mlp.c_proj.weight", (d, 4 * d), d ** -0.5 * (2 * layers) ** -0.5), (p + "mlp.c_proj.bias", (d,), 0.02)]
    return rows


def random_clip_state_shapes(shape: ModelShape) -> Dict[str, torch.Tensor]:
    """Uninitialised tensors of a CLIP state dict's shapes (under ``torch.device("meta")``: shapes only), for shape inference."""
    sd = {k: torch.empty(s) for k, s, _ in clip_state_table(shape)}
    sd["logit_scale"] = torch.empty(())
    return sd


def random_clip_state(shape: ModelShape, seed: int = 0, fp16_weights: bool = True) -> Dict[str, torch.Tensor]:
    """Random CLIP weights with the distributions of CLIP.initialize_parameters (clip/model.py:781-808) and the
    vision tower's width^-1/2 scale (clip/model.py:506-508,524); random, not zeros (zeros inflate clocks)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, s, init in clip_state_table(shape):
        if init == "ones":
            sd[k] = torch.ones(s)
        elif init == "zeros":
            sd[k] = torch.zeros(s)
        else:
            t = torch.randn(*s, generator=g) * init
            sd[k] = t.half().float() if fp16_weights else t
    sd["logit_scale"] = torch.tensor(math.log(1 / 0.07))
    return sd
