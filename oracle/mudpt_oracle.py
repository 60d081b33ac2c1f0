"""CPU oracle for the MuDPT prompt-tuning hot path (TEST INFRASTRUCTURE, not product).

This file is a from-scratch fp32 restatement, in plain torch CPU ops, of the arithmetic
the reference performs in

  * ``trainers/mudpt.py:117-130``  MuDPTPromptLearner.forward   -> :func:`prompt_learner`
  * ``clip/model.py:526-553``      VisionTransformer_MuDPT.forward -> :func:`vision_tower`
  * ``clip/model.py:275-301``      ResidualAttentionBlock_MuDPT.forward -> :func:`block`
  * ``trainers/mudpt.py:142-156``  TextEncoder.forward          -> :func:`text_tower`
  * ``trainers/mudpt.py:170-184``  CustomCLIP.forward           -> :func:`forward`
  * ``trainers/mudpt.py:249-251``  cross-entropy + backward     -> :func:`forward_backward`

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it.  The product path (``mudpt_amd``) never does and fails loudly when its HIP library is missing.

Pinning: the reference ships no tests or golden vectors for this path, so the
oracle is pinned by fixtures under ``tests/golden/`` that ``tests/golden/gen_golden.py`` produced by
running the reference's own modules (imported unmodified from /root/reference) on weights made by
:func:`make_frozen_state` / :func:`make_trainable_state`; ``tests/test_oracle_golden.py`` checks it.

Layout differs from the reference on purpose: activations are batch-first ``[B, L, d]`` (the
reference permutes to ``[L, B, d]`` for nn.MultiheadAttention); the arithmetic is identical.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, asdict
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor

# Names of the 10 trainable tensors in the order they sit in the flat parameter / gradient bucket.
# Keys are the reference's CustomCLIP state-dict names (trainers/mudpt.py:205-218 freeze rule).
TRAINABLE_ORDER = [
    "mudpt_prompt_learner.ctx",
    "mudpt_prompt_learner.deep_prompts",
    "mudpt_prompt_learner.embed_projection.weight",
    "mudpt_prompt_learner.embed_projection.bias",
    "mudpt_prompt_learner.deep_projections.weight",
    "mudpt_prompt_learner.deep_projections.bias",
    "image_encoder.visual_ctx",
    "image_encoder.visual_ctx_deep_prompts",
    "image_encoder.visual_ctx_deep_projections.weight",
    "image_encoder.visual_ctx_deep_projections.bias",
]


@dataclass(frozen=True)
class Config:
    """Shape of one MuDPT model.  Defaults = CLIP ViT-B/16 with n_ctx 4, depth 12 (BASELINE.json)."""
    image_size: int = 224
    patch: int = 16
    v_width: int = 768
    v_layers: int = 12
    v_heads: int = 12
    t_width: int = 512
    t_layers: int = 12
    t_heads: int = 8
    ctx_len: int = 77
    vocab: int = 49408
    embed_dim: int = 512
    n_ctx: int = 4
    depth: int = 12

    @property
    def n_patches(self) -> int:
        return (self.image_size // self.patch) ** 2

    @property
    def v_tokens(self) -> int:  # CLS + patches + prompt rows (clip/model.py:530-536)
        return 1 + self.n_patches + self.n_ctx

    def asdict(self):
        return asdict(self)


VIT_B16 = Config()
# CLIP ViT-L/14@336px with MuDPT depth 24 (BASELINE configs[4]): 576 patches + CLS + 4 prompt rows = 581 vision tokens; the text
# tower has 12 layers, so deep prompts 12..22 are never consumed and get zero gradient
VIT_L14_336 = Config(image_size=336, patch=14, v_width=1024, v_layers=24, v_heads=16, t_width=768, t_layers=12, t_heads=12,
                     ctx_len=77, vocab=49408, embed_dim=768, n_ctx=4, depth=24)
# Small shape used by fast tests: head_dim stays 64 (vision_heads = width // 64, clip/model.py:695),
# embed_dim == t_width as the reference requires, depth < layers so that
# "layers >= depth keep propagating prompt outputs" is exercised.
TINY = Config(image_size=32, patch=16, v_width=192, v_layers=3, v_heads=3, t_width=128, t_layers=3,
              t_heads=2, ctx_len=77, vocab=49408, embed_dim=128, n_ctx=2, depth=2)


# --------------------------------------------------------------------------------------------
# Weight recipes (seed + rule; fixtures store the recipe, not the weights)
# --------------------------------------------------------------------------------------------
def _fp16_round(t: Tensor) -> Tensor:
    """Round to fp16-representable values, as clip/model.py:857-878 convert_weights stores them."""
    return t.half().float()


def frozen_keys(cfg: Config) -> List[Tuple[str, Tuple[int, ...], str]]:
    """(state-dict key, shape, init kind) for every frozen CLIP tensor on the path.

    Key names are OpenAI CLIP's (clip/model.py:667-779 CLIP.__init__, :500-524 vision tower)."""
    out: List[Tuple[str, Tuple[int, ...], str]] = []
    dv, dt, e = cfg.v_width, cfg.t_width, cfg.embed_dim
    out += [
        ("visual.conv1.weight", (dv, 3, cfg.patch, cfg.patch), "conv"),
        ("visual.class_embedding", (dv,), "v_scale"),
        ("visual.positional_embedding", (cfg.n_patches + 1, dv), "v_scale"),
        ("visual.ln_pre.weight", (dv,), "ln_w"), ("visual.ln_pre.bias", (dv,), "ln_b"),
        ("visual.ln_post.weight", (dv,), "ln_w"), ("visual.ln_post.bias", (dv,), "ln_b"),
        ("visual.proj", (dv, e), "v_scale"),
    ]

    def blocks(prefix: str, d: int, layers: int):
        for i in range(layers):
            p = f"{prefix}.resblocks.{i}."
            yield (p + "ln_1.weight", (d,), "ln_w")
            yield (p + "ln_1.bias", (d,), "ln_b")
            yield (p + "attn.in_proj_weight", (3 * d, d), f"attn:{d}")
            yield (p + "attn.in_proj_bias", (3 * d,), "bias")
            yield (p + "attn.out_proj.weight", (d, d), f"proj:{d}:{layers}")
            yield (p + "attn.out_proj.bias", (d,), "bias")
            yield (p + "ln_2.weight", (d,), "ln_w")
            yield (p + "ln_2.bias", (d,), "ln_b")
            yield (p + "mlp.c_fc.weight", (4 * d, d), f"fc:{d}")
            yield (p + "mlp.c_fc.bias", (4 * d,), "bias")
            yield (p + "mlp.c_proj.weight", (d, 4 * d), f"proj:{d}:{layers}")
            yield (p + "mlp.c_proj.bias", (d,), "bias")

    out += list(blocks("visual.transformer", dv, cfg.v_layers))
    out += list(blocks("transformer", dt, cfg.t_layers))
    out += [
        ("token_embedding.weight", (cfg.vocab, dt), "emb"),
        ("positional_embedding", (cfg.ctx_len, dt), "pos"),
        ("ln_final.weight", (dt,), "ln_w"), ("ln_final.bias", (dt,), "ln_b"),
        ("text_projection", (dt, e), f"attn:{dt}"),
        ("logit_scale", (), "logit_scale"),
    ]
    return out


def make_frozen_state(cfg: Config, seed: int = 0) -> Dict[str, Tensor]:
    """Seeded random frozen CLIP weights, fp16-representable.

    Standard deviations follow CLIP.initialize_parameters (clip/model.py:781-808: attn d^-1/2,
    proj d^-1/2 (2 layers)^-1/2, fc (2d)^-1/2, emb 0.02, pos 0.01) and the vision tower's
    ``scale = width^-1/2`` (clip/model.py:506-508,524).  LayerNorm affine terms and biases are
    perturbed away from (1, 0) so that every parameter is exercised; the fixture generator loads
    exactly these tensors into the reference modules, so any values are legitimate."""
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, Tensor] = {}
    for key, shape, kind in frozen_keys(cfg):
        if kind == "logit_scale":
            sd[key] = torch.tensor(math.log(1 / 0.07))  # clip/model.py:777
            continue
        r = torch.randn(shape, generator=g)
        if kind == "conv":
            t = r * (3 * cfg.patch * cfg.patch) ** -0.5
        elif kind == "v_scale":
            t = r * cfg.v_width ** -0.5
        elif kind == "ln_w":
            t = 1.0 + 0.1 * r
        elif kind == "ln_b":
            t = 0.05 * r
        elif kind == "bias":
            t = 0.02 * r
        elif kind == "emb":
            t = 0.02 * r
        elif kind == "pos":
            t = 0.01 * r
        else:
            name, *args = kind.split(":")
            d = int(args[0])
            if name == "attn":
                t = r * d ** -0.5
            elif name == "fc":
                t = r * (2 * d) ** -0.5
            elif name == "proj":
                t = r * d ** -0.5 * (2 * int(args[1])) ** -0.5
            else:  # pragma: no cover
                raise ValueError(kind)
        sd[key] = _fp16_round(t)
    return sd


def trainable_shapes(cfg: Config) -> Dict[str, Tuple[int, ...]]:
    """Shapes of the 10 trainable tensors (trainers/mudpt.py:71-81, clip/model.py:512-519)."""
    n, D, dt, dv, e = cfg.n_ctx, cfg.depth, cfg.t_width, cfg.v_width, cfg.embed_dim
    return {
        "mudpt_prompt_learner.ctx": (n, dt),
        "mudpt_prompt_learner.deep_prompts": (D - 1, n, dt),
        "mudpt_prompt_learner.embed_projection.weight": (dv, dt),
        "mudpt_prompt_learner.embed_projection.bias": (dv,),
        "mudpt_prompt_learner.deep_projections.weight": (dv, dt),
        "mudpt_prompt_learner.deep_projections.bias": (dv,),
        "image_encoder.visual_ctx": (n, dv),
        "image_encoder.visual_ctx_deep_prompts": (D - 1, n, dv),
        "image_encoder.visual_ctx_deep_projections.weight": (e, dv),
        "image_encoder.visual_ctx_deep_projections.bias": (e,),
    }


def make_trainable_state(cfg: Config, seed: int = 1, frozen: Optional[Dict[str, Tensor]] = None,
                         ctx_token_ids: Optional[List[int]] = None) -> Dict[str, Tensor]:
    """Seeded values for the 10 trainable tensors.

    Prompts ~ N(0, 0.02^2) (trainers/mudpt.py:78-79, clip/model.py:513-517), Linear layers ~
    U(-1/sqrt(in), 1/sqrt(in)) like nn.Linear's default.  When ``frozen`` and ``ctx_token_ids`` are
    given, ``ctx`` is the token embedding of the init words (trainers/mudpt.py:57-64)."""
    g = torch.Generator().manual_seed(seed)
    out: Dict[str, Tensor] = {}
    for name, shape in trainable_shapes(cfg).items():
        if name.endswith("weight") or name.endswith("bias"):
            fan_in = shape[-1] if name.endswith("weight") else trainable_shapes(cfg)[name[:-4] + "weight"][-1]
            out[name] = (torch.rand(shape, generator=g) * 2 - 1) / math.sqrt(fan_in)
        else:
            out[name] = 0.02 * torch.randn(shape, generator=g)
    if frozen is not None and ctx_token_ids is not None:
        out["mudpt_prompt_learner.ctx"] = frozen["token_embedding.weight"][ctx_token_ids].clone()
    return out


def flatten(tensors: Dict[str, Tensor]) -> Tensor:
    return torch.cat([tensors[k].reshape(-1) for k in TRAINABLE_ORDER])


def unflatten(flat: Tensor, cfg: Config) -> Dict[str, Tensor]:
    out, off = {}, 0
    for k in TRAINABLE_ORDER:
        shp = trainable_shapes(cfg)[k]
        n = int(torch.tensor(shp).prod()) if len(shp) else 1
        out[k] = flat[off:off + n].reshape(shp)
        off += n
    return out


# --------------------------------------------------------------------------------------------
# Building blocks
# --------------------------------------------------------------------------------------------
def layer_norm(x: Tensor, w: Tensor, b: Tensor) -> Tensor:
    """clip/model.py:164-170: LayerNorm always evaluated in fp32, eps = nn.LayerNorm default."""
    return F.layer_norm(x.float(), (x.shape[-1],), w, b, 1e-5)


def quick_gelu(x: Tensor) -> Tensor:
    """clip/model.py:173-175."""
    return x * torch.sigmoid(1.702 * x)


def causal_mask(L: int) -> Tensor:
    """clip/model.py:810-816: additive mask, -inf strictly above the diagonal."""
    return torch.full((L, L), float("-inf")).triu_(1)


def attention(qkv: Tensor, heads: int, mask: Optional[Tensor]) -> Tensor:
    """Per-head softmax(Q K^T / sqrt(d_h) + mask) V on packed ``qkv [B, L, 3d]``.

    What nn.MultiheadAttention computes between its in- and out-projection
    (clip/model.py:271-273; packed in_proj order q, k, v)."""
    B, L, d3 = qkv.shape
    d = d3 // 3
    dh = d // heads
    q, k, v = qkv.split(d, dim=-1)
    q = q.reshape(B, L, heads, dh).transpose(1, 2)
    k = k.reshape(B, L, heads, dh).transpose(1, 2)
    v = v.reshape(B, L, heads, dh).transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) / math.sqrt(dh)
    if mask is not None:
        s = s + mask
    p = torch.softmax(s, dim=-1)
    o = p @ v
    return o.transpose(1, 2).reshape(B, L, d)


def block(x: Tensor, sd: Dict[str, Tensor], prefix: str, heads: int, mask: Optional[Tensor],
          taps: Optional[dict] = None) -> Tensor:
    """Pre-LN residual block, clip/model.py:299-300 (after the prompt splice of :275-297)."""
    h = layer_norm(x, sd[prefix + "ln_1.weight"], sd[prefix + "ln_1.bias"])
    qkv = h @ sd[prefix + "attn.in_proj_weight"].t() + sd[prefix + "attn.in_proj_bias"]
    a = attention(qkv, heads, mask)
    x = x + a @ sd[prefix + "attn.out_proj.weight"].t() + sd[prefix + "attn.out_proj.bias"]
    h2 = layer_norm(x, sd[prefix + "ln_2.weight"], sd[prefix + "ln_2.bias"])
    u = h2 @ sd[prefix + "mlp.c_fc.weight"].t() + sd[prefix + "mlp.c_fc.bias"]
    x = x + quick_gelu(u) @ sd[prefix + "mlp.c_proj.weight"].t() + sd[prefix + "mlp.c_proj.bias"]
    if taps is not None:
        taps[prefix + "out"] = x
    return x


def patchify(images: Tensor, patch: int) -> Tensor:
    """[B,3,H,W] -> [B, (H/p)(W/p), 3 p p]: the im2col of the stride-p conv (clip/model.py:527-529)."""
    B, C, H, W = images.shape
    gh, gw = H // patch, W // patch
    x = images.reshape(B, C, gh, patch, gw, patch).permute(0, 2, 4, 1, 3, 5)
    return x.reshape(B, gh * gw, C * patch * patch)


# --------------------------------------------------------------------------------------------
# The path
# --------------------------------------------------------------------------------------------
def prompt_learner(cfg: Config, params: Dict[str, Tensor], class_embedding: Tensor):
    """trainers/mudpt.py:117-130 (+ construct_prompts :97-115).

    ``class_embedding [C, ctx_len, d_t]`` is token_embedding(tokenized prompts); rows 1..n_ctx are
    replaced by ``ctx`` (prefix = row 0, suffix = rows 1+n_ctx.., :89-90)."""
    P = "mudpt_prompt_learner."
    ctx = params[P + "ctx"]
    n = cfg.n_ctx
    C = class_embedding.shape[0]
    prompts = torch.cat([class_embedding[:, :1], ctx.unsqueeze(0).expand(C, -1, -1),
                         class_embedding[:, 1 + n:]], dim=1)
    t2v = params[P + "deep_prompts"] @ params[P + "deep_projections.weight"].t() + params[P + "deep_projections.bias"]
    shared = ctx @ params[P + "embed_projection.weight"].t() + params[P + "embed_projection.bias"]
    return prompts, shared, params[P + "deep_prompts"], t2v


def vision_tower(cfg: Config, sd: Dict[str, Tensor], params: Dict[str, Tensor], images: Tensor,
                 shared: Tensor, t2v: Tensor, taps: Optional[dict] = None):
    """clip/model.py:526-553.  Returns (image_features [B, e], v->t text prompts [D-1, n, e])."""
    V = "image_encoder."
    B = images.shape[0]
    n = cfg.n_ctx
    w = sd["visual.conv1.weight"].reshape(cfg.v_width, -1)
    x = patchify(images.float(), cfg.patch) @ w.t()
    cls = sd["visual.class_embedding"].expand(B, 1, -1)
    x = torch.cat([cls, x], dim=1) + sd["visual.positional_embedding"]
    vp = (params[V + "visual_ctx"] + shared).unsqueeze(0).expand(B, -1, -1)  # no pos-emb on prompt rows
    x = torch.cat([x, vp], dim=1)
    deep = t2v + params[V + "visual_ctx_deep_prompts"]
    v2t = (params[V + "visual_ctx_deep_prompts"] @ params[V + "visual_ctx_deep_projections.weight"].t()
           + params[V + "visual_ctx_deep_projections.bias"])
    x = layer_norm(x, sd["visual.ln_pre.weight"], sd["visual.ln_pre.bias"])
    if taps is not None:
        taps["visual.ln_pre"] = x
    L = x.shape[1]
    for i in range(cfg.v_layers):
        # clip/model.py:279,290-297: layer i >= 1 replaces the LAST n rows by deep[i-1] while i-1 < D-1
        if i >= 1 and (i - 1) < deep.shape[0]:
            x = torch.cat([x[:, :L - n], deep[i - 1].unsqueeze(0).expand(B, -1, -1)], dim=1)
        x = block(x, sd, f"visual.transformer.resblocks.{i}.", cfg.v_heads, None, taps)
    f = layer_norm(x[:, 0], sd["visual.ln_post.weight"], sd["visual.ln_post.bias"]) @ sd["visual.proj"]
    return f, v2t


def text_tower(cfg: Config, sd: Dict[str, Tensor], prompts: Tensor, eot: Tensor, deep: Tensor,
               taps: Optional[dict] = None) -> Tensor:
    """trainers/mudpt.py:142-156 with the text branch of clip/model.py:281-289."""
    n = cfg.n_ctx
    x = prompts + sd["positional_embedding"]
    C, L, _ = x.shape
    mask = causal_mask(L)
    for i in range(cfg.t_layers):
        if i >= 1 and (i - 1) < deep.shape[0]:  # rows 1..n replaced, no pos-emb (appendix A.6)
            x = torch.cat([x[:, :1], deep[i - 1].unsqueeze(0).expand(C, -1, -1), x[:, 1 + n:]], dim=1)
        x = block(x, sd, f"transformer.resblocks.{i}.", cfg.t_heads, mask, taps)
    x = layer_norm(x, sd["ln_final.weight"], sd["ln_final.bias"])
    return x[torch.arange(C), eot] @ sd["text_projection"]


def forward(cfg: Config, sd: Dict[str, Tensor], params: Dict[str, Tensor], class_embedding: Tensor,
            eot: Tensor, images: Tensor, taps: Optional[dict] = None) -> Tensor:
    """trainers/mudpt.py:170-184 CustomCLIP.forward -> logits [B, C] fp32."""
    prompts, shared, text_deep, t2v = prompt_learner(cfg, params, class_embedding)
    img_f, v2t = vision_tower(cfg, sd, params, images, shared, t2v, taps)
    txt_f = text_tower(cfg, sd, prompts, eot, text_deep + v2t, taps)
    if taps is not None:
        taps["image_features"], taps["text_features"] = img_f, txt_f
    img_f = img_f / img_f.norm(dim=-1, keepdim=True)
    txt_f = txt_f / txt_f.norm(dim=-1, keepdim=True)
    return sd["logit_scale"].exp() * img_f @ txt_f.t()


def forward_backward(cfg: Config, sd: Dict[str, Tensor], params: Dict[str, Tensor],
                     class_embedding: Tensor, eot: Tensor, images: Tensor, labels: Tensor):
    """trainers/mudpt.py:249-251: mean cross-entropy and its gradient w.r.t. the 10 trainables."""
    leaf = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
    logits = forward(cfg, sd, leaf, class_embedding, eot, images)
    loss = F.cross_entropy(logits, labels.long())
    grads = torch.autograd.grad(loss, [leaf[k] for k in TRAINABLE_ORDER], allow_unused=True)
    grads = {k: (g if g is not None else torch.zeros_like(leaf[k])) for k, g in zip(TRAINABLE_ORDER, grads)}
    return loss.detach(), logits.detach(), grads


def sgd_step(param: Tensor, grad: Tensor, buf: Optional[Tensor], lr: float, momentum: float = 0.9,
             weight_decay: float = 5e-4, dampening: float = 0.0, nesterov: bool = False):
    """torch.optim.SGD's update (what Dassl's build_optimizer("sgd") runs; Dassl defaults
    momentum 0.9, weight_decay 5e-4, dampening 0, nesterov False).  Returns (param, buf)."""
    g = grad + weight_decay * param
    if momentum != 0:
        buf = g.clone() if buf is None else momentum * buf + (1 - dampening) * g
        g = g + momentum * buf if nesterov else buf
    return param - lr * g, buf


# --------------------------------------------------------------------------------------------
# Synthetic class prompts for shapes without a tokenizer fixture
# --------------------------------------------------------------------------------------------
def synthetic_tokens(cfg: Config, n_cls: int, seed: int = 7) -> Tensor:
    """[C, ctx_len] int32 shaped like clip.tokenize output (clip/clip.py:199-239): SOT, n_ctx
    context words, 1-3 class-name tokens, '.', EOT (the largest id), zero padding."""
    g = torch.Generator().manual_seed(seed)
    sot, eot_id = cfg.vocab - 2, cfg.vocab - 1
    tok = torch.zeros(n_cls, cfg.ctx_len, dtype=torch.int32)
    for c in range(n_cls):
        k = 1 + int(torch.randint(0, 3, (1,), generator=g))
        ids = [sot] + [320 + j for j in range(cfg.n_ctx)] + \
              [int(v) for v in torch.randint(1000, cfg.vocab - 1000, (k,), generator=g)] + [269, eot_id]
        tok[c, :len(ids)] = torch.tensor(ids, dtype=torch.int32)
    return tok
