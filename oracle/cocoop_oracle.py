"""CPU oracle for the CoCoOp path (TEST INFRASTRUCTURE, not product; same rules as mudpt_oracle.py).

A from-scratch fp32 restatement, in plain torch CPU ops, of what the reference computes in

  * ``trainers/cocoop.py:141-165``  PromptLearner.forward (meta_net bias, ctx shift, construct_prompts) -> :func:`prompts_for`
  * ``clip/model.py:478-496``       VisionTransformer.forward (the vanilla ViT: ``clip.load(..., cfg=None)``,
                                    ``trainers/cocoop.py:38``)                                            -> :func:`vision_tower`
  * ``trainers/cocoop.py:51-64``    TextEncoder.forward                                                  -> :func:`text_tower`
  * ``trainers/cocoop.py:178-198``  CustomCLIP.forward (per-image text features, logits, CE in training)  -> :func:`forward`
  * ``trainers/cocoop.py:258-261``  loss.backward() w.r.t. the ``prompt_learner`` parameters (:222-226)   -> :func:`forward_backward`

The reference loops over the images and runs the text encoder once per image (:187-194); the restatement batches all
(image, class) prompts, which is the same arithmetic.  Pinned by ``tests/golden/cocoop_*.npz`` (made by
``tests/golden/gen_golden.py`` from the reference's own ``trainers.cocoop.CustomCLIP``); ``tests/test_oracle_golden.py`` checks it.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

from . import mudpt_oracle as O

Tensor = torch.Tensor

# The 5 trainable tensors in flat-bucket order, under the reference's CustomCLIP state-dict names (trainers/cocoop.py:96-107,176)
TRAINABLE_ORDER = [
    "prompt_learner.ctx",
    "prompt_learner.meta_net.linear1.weight",
    "prompt_learner.meta_net.linear1.bias",
    "prompt_learner.meta_net.linear2.weight",
    "prompt_learner.meta_net.linear2.bias",
]


def trainable_shapes(cfg: O.Config) -> Dict[str, Tuple[int, ...]]:
    """trainers/cocoop.py:96-107: ctx [n_ctx, ctx_dim]; meta_net = Linear(vis_dim, vis_dim // 16), ReLU, Linear(vis_dim // 16, ctx_dim)
    with vis_dim = clip_model.visual.output_dim = embed_dim."""
    e, dt, h = cfg.embed_dim, cfg.t_width, cfg.embed_dim // 16
    return {TRAINABLE_ORDER[0]: (cfg.n_ctx, dt), TRAINABLE_ORDER[1]: (h, e), TRAINABLE_ORDER[2]: (h,),
            TRAINABLE_ORDER[3]: (dt, h), TRAINABLE_ORDER[4]: (dt,)}


def make_trainable_state(cfg: O.Config, seed: int = 1, frozen: Optional[Dict[str, Tensor]] = None,
                         ctx_token_ids: Optional[List[int]] = None) -> Dict[str, Tensor]:
    """Seeded values: ctx ~ N(0, 0.02^2) (:90-91) or the CTX_INIT words' token embeddings (:79-87); Linear layers ~
    U(-1/sqrt(in), 1/sqrt(in)) like nn.Linear's default."""
    g = torch.Generator().manual_seed(seed)
    shapes = trainable_shapes(cfg)
    out: Dict[str, Tensor] = {}
    for name, shape in shapes.items():
        if name.endswith("weight") or name.endswith("bias"):
            fan_in = shapes[name.rsplit(".", 1)[0] + ".weight"][-1]
            out[name] = (torch.rand(shape, generator=g) * 2 - 1) / math.sqrt(fan_in)
        else:
            out[name] = 0.02 * torch.randn(shape, generator=g)
    if frozen is not None and ctx_token_ids is not None:
        out[TRAINABLE_ORDER[0]] = frozen["token_embedding.weight"][ctx_token_ids].clone()
    return out


def flatten(tensors: Dict[str, Tensor]) -> Tensor:
    return torch.cat([tensors[k].reshape(-1) for k in TRAINABLE_ORDER])


def vision_tower(cfg: O.Config, sd: Dict[str, Tensor], images: Tensor) -> Tensor:
    """clip/model.py:478-496 (img_prompt False): conv-as-GEMM, CLS + positional embedding, ln_pre, blocks, ln_post(CLS) @ proj."""
    B = images.shape[0]
    w = sd["visual.conv1.weight"].reshape(cfg.v_width, -1)
    x = O.patchify(images.float(), cfg.patch) @ w.t()
    cls = sd["visual.class_embedding"].expand(B, 1, -1)
    x = torch.cat([cls, x], dim=1) + sd["visual.positional_embedding"]
    x = O.layer_norm(x, sd["visual.ln_pre.weight"], sd["visual.ln_pre.bias"])
    for i in range(cfg.v_layers):
        x = O.block(x, sd, f"visual.transformer.resblocks.{i}.", cfg.v_heads, None)
    return O.layer_norm(x[:, 0], sd["visual.ln_post.weight"], sd["visual.ln_post.bias"]) @ sd["visual.proj"]


def meta_net(params: Dict[str, Tensor], im_features: Tensor) -> Tensor:
    """trainers/cocoop.py:103-107,145."""
    P = "prompt_learner.meta_net."
    h = torch.relu(im_features @ params[P + "linear1.weight"].t() + params[P + "linear1.bias"])
    return h @ params[P + "linear2.weight"].t() + params[P + "linear2.bias"]


def prompts_for(cfg: O.Config, params: Dict[str, Tensor], class_embedding: Tensor, im_features: Tensor) -> Tensor:
    """trainers/cocoop.py:141-165 -> [B, C, ctx_len, d_t]: prefix (row 0) | ctx + bias_i | suffix (rows 1+n..)."""
    n, C, B = cfg.n_ctx, class_embedding.shape[0], im_features.shape[0]
    bias = meta_net(params, im_features)                                  # [B, d_t]
    ctx = params["prompt_learner.ctx"].unsqueeze(0) + bias.unsqueeze(1)     # [B, n, d_t]
    return torch.cat([class_embedding[:, :1].unsqueeze(0).expand(B, -1, -1, -1), ctx.unsqueeze(1).expand(-1, C, -1, -1),
                      class_embedding[:, 1 + n:].unsqueeze(0).expand(B, -1, -1, -1)], dim=2)


def text_tower(cfg: O.Config, sd: Dict[str, Tensor], prompts: Tensor, eot: Tensor) -> Tensor:
    """trainers/cocoop.py:51-64 on [S, ctx_len, d_t] prompts: + positional embedding, causal blocks, ln_final, EOT row @ projection."""
    x = prompts + sd["positional_embedding"]
    S, L, _ = x.shape
    mask = O.causal_mask(L)
    for i in range(cfg.t_layers):
        x = O.block(x, sd, f"transformer.resblocks.{i}.", cfg.t_heads, mask)
    x = O.layer_norm(x, sd["ln_final.weight"], sd["ln_final.bias"])
    return x[torch.arange(S), eot] @ sd["text_projection"]


def forward(cfg: O.Config, sd: Dict[str, Tensor], params: Dict[str, Tensor], class_embedding: Tensor, eot: Tensor,
            images: Tensor, taps: Dict[str, Tensor] = None) -> Tensor:
    """trainers/cocoop.py:178-195 -> logits [B, C].  taps (tests): receives the text-tower input "prompts" [B * C, ctx_len, d_t] with its
    gradient retained, i.e. the per-(image, class) terms whose sums are the ctx / meta_net gradients."""
    B, C = images.shape[0], class_embedding.shape[0]
    img = vision_tower(cfg, sd, images)
    img = img / img.norm(dim=-1, keepdim=True)
    prompts = prompts_for(cfg, params, class_embedding, img).reshape(B * C, cfg.ctx_len, cfg.t_width)
    if taps is not None and prompts.requires_grad:
        prompts.retain_grad()
        taps["prompts"] = prompts
    txt = text_tower(cfg, sd, prompts, eot.repeat(B)).reshape(B, C, -1)
    txt = txt / txt.norm(dim=-1, keepdim=True)
    return sd["logit_scale"].exp() * torch.einsum("be,bce->bc", img, txt)


def forward_backward(cfg: O.Config, sd: Dict[str, Tensor], params: Dict[str, Tensor], class_embedding: Tensor, eot: Tensor,
                     images: Tensor, labels: Tensor, taps: Dict[str, Tensor] = None):
    """trainers/cocoop.py:196-197 + :258-261: mean cross-entropy and its gradient w.r.t. the 5 trainables.
    taps (tests): "dprompts" [B, C, ctx_len, d_t] = gradient w.r.t. every (image, class) prompt (the terms the gradients sum over)."""
    leaf = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
    inner = {} if taps is not None else None
    logits = forward(cfg, sd, leaf, class_embedding, eot, images, inner)
    loss = F.cross_entropy(logits, labels.long())
    if taps is None:
        grads = torch.autograd.grad(loss, [leaf[k] for k in TRAINABLE_ORDER])
    else:
        loss.backward()
        grads = [leaf[k].grad for k in TRAINABLE_ORDER]
        taps["dprompts"] = inner["prompts"].grad.detach().reshape(images.shape[0], class_embedding.shape[0], cfg.ctx_len, cfg.t_width)
    return loss.detach(), logits.detach(), dict(zip(TRAINABLE_ORDER, grads))
