"""CPU-side precision ablation of the MuDPT forward (TEST INFRASTRUCTURE: a script, not a test; run it by hand).

Question: which operand roundings of the fp16 mode make up its 4.3e-3 logit error at logit scale 100, and what is
the cheapest set of sites that has to carry more bits to hold north_star's 1e-3?  The oracle's fp32 forward is re-run with a rounding
model at each site the kernels round at, against the reference's own logits in the `*_s100` fixtures:

  site      what is rounded                                       kernel that rounds it
  qkv       ln_1 output, the A operand of in_proj                 ln_fwd_kernel
  qk        q and k as the score product's operands               in_proj epilogue (qkv stored in T)
  pv        P = exp(s - m) and v as the P.V product's operands    attention kernels
  out       attention output, the A operand of out_proj           attention kernels
  fc        ln_2 output, the A operand of c_fc                    ln_fwd_kernel
  proj      QuickGELU(u), the A operand of c_proj                 c_fc epilogue
  patch     pixels, the A operand of the patch-embed GEMM         patchify (vision only)

  rounding models: f32 (none) | f16 | bf16 | pair (fp16 hi + fp16 lo: 22 bits, the exact mode) |
                   f8lo (fp16 hi + e4m3 lo x e4m3 weights: the lo term on the fp8 matrix pipe at twice the fp16 rate)

Usage:  python tests/precision_ablation.py [fixture] [--quick]
"""
import itertools
import math
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from oracle import mudpt_oracle as O  # noqa: E402
from tests.helpers import GoldenCase  # noqa: E402

SITES = ["qkv", "qk", "pv", "out", "fc", "proj"]


def r16(x):
    return x.half().float()


def rb16(x):
    return x.bfloat16().float()


def r8(x):
    """e4m3 with one power-of-two scale per tensor (what a fixed shift in the producing kernel gives)."""
    m = x.abs().max().item()
    if m == 0:
        return x
    s = 2.0 ** math.floor(math.log2(256.0 / m))
    return (x * s).to(torch.float8_e4m3fn).float() / s


def linear(x, w, b, mode):
    """x @ w.T + b with the A operand x rounded by `mode` (weights are fp16-exact)."""
    if mode == "f32":
        return x @ w.t() + b
    if mode == "f16":
        return r16(x) @ w.t() + b
    if mode == "bf16":
        return rb16(x) @ rb16(w).t() + b
    hi = r16(x)
    lo = x - hi
    if mode == "pair":
        return hi @ w.t() + r16(lo) @ w.t() + b
    if mode == "f8lo":
        return hi @ w.t() + r8(lo) @ r8(w).t() + b
    raise ValueError(mode)


def rnd(x, mode):
    return {"f32": lambda v: v, "f16": r16, "bf16": rb16, "pair": lambda v: r16(v) + r16(v - r16(v)), "f8lo": lambda v: r16(v) + r8(v - r16(v))}[mode](x)


def attention(qkv, heads, mask, m_qk, m_pv):
    B, L, d3 = qkv.shape
    d = d3 // 3
    q, k, v = (t.reshape(B, L, heads, 64).transpose(1, 2) for t in qkv.split(d, dim=-1))
    s = (rnd(q, m_qk) @ rnd(k, m_qk).transpose(-1, -2)) / 8.0
    if mask is not None:
        s = s + mask
    mx = s.max(dim=-1, keepdim=True).values
    p = torch.exp(s - mx)
    l = p.sum(dim=-1, keepdim=True)  # fp32 row sums of the unrounded p, as the kernels keep them
    o = (rnd(p, m_pv) @ rnd(v, m_pv)) / l
    return o.transpose(1, 2).reshape(B, L, d)


def block(x, sd, prefix, heads, mask, md):
    h = O.layer_norm(x, sd[prefix + "ln_1.weight"], sd[prefix + "ln_1.bias"])
    qkv = linear(h, sd[prefix + "attn.in_proj_weight"], sd[prefix + "attn.in_proj_bias"], md["qkv"])
    a = attention(qkv, heads, mask, md["qk"], md["pv"])
    x = x + linear(a, sd[prefix + "attn.out_proj.weight"], sd[prefix + "attn.out_proj.bias"], md["out"])
    h2 = O.layer_norm(x, sd[prefix + "ln_2.weight"], sd[prefix + "ln_2.bias"])
    u = linear(h2, sd[prefix + "mlp.c_fc.weight"], sd[prefix + "mlp.c_fc.bias"], md["fc"])
    return x + linear(O.quick_gelu(u), sd[prefix + "mlp.c_proj.weight"], sd[prefix + "mlp.c_proj.bias"], md["proj"])


def forward(case, vis, txt, layer_from=0):
    """logits with rounding models vis / txt (dicts site -> mode); layers below layer_from of the VISION tower run the f16 model."""
    cfg, sd, params = case.cfg, case.frozen, case.params
    prompts, shared, text_deep, t2v = O.prompt_learner(cfg, params, case.class_embedding)
    V = "image_encoder."
    B, n = case.images.shape[0], cfg.n_ctx
    w = sd["visual.conv1.weight"].reshape(cfg.v_width, -1)
    x = linear(O.patchify(case.images.float(), cfg.patch), w, 0.0, vis.get("patch", "f32"))
    x = torch.cat([sd["visual.class_embedding"].expand(B, 1, -1), x], dim=1) + sd["visual.positional_embedding"]
    x = torch.cat([x, (params[V + "visual_ctx"] + shared).unsqueeze(0).expand(B, -1, -1)], dim=1)
    deep = t2v + params[V + "visual_ctx_deep_prompts"]
    v2t = params[V + "visual_ctx_deep_prompts"] @ params[V + "visual_ctx_deep_projections.weight"].t() + params[V + "visual_ctx_deep_projections.bias"]
    x = O.layer_norm(x, sd["visual.ln_pre.weight"], sd["visual.ln_pre.bias"])
    L = x.shape[1]
    f16 = {s: "f16" for s in SITES}
    for i in range(cfg.v_layers):
        if i >= 1 and (i - 1) < deep.shape[0]:
            x = torch.cat([x[:, :L - n], deep[i - 1].unsqueeze(0).expand(B, -1, -1)], dim=1)
        x = block(x, sd, f"visual.transformer.resblocks.{i}.", cfg.v_heads, None, vis if i >= layer_from else f16)
    img_f = O.layer_norm(x[:, 0], sd["visual.ln_post.weight"], sd["visual.ln_post.bias"]) @ sd["visual.proj"]
    # text tower on the positions the library runs (up to the last EOT; causal: identical rows)
    Le = int(case.eot.max()) + 1
    x = (prompts + sd["positional_embedding"])[:, :Le]
    C = x.shape[0]
    mask = O.causal_mask(Le)
    tdeep = text_deep + v2t
    for i in range(cfg.t_layers):
        if i >= 1 and (i - 1) < tdeep.shape[0]:
            x = torch.cat([x[:, :1], tdeep[i - 1].unsqueeze(0).expand(C, -1, -1), x[:, 1 + n:]], dim=1)
        x = block(x, sd, f"transformer.resblocks.{i}.", cfg.t_heads, mask, txt)
    x = O.layer_norm(x, sd["ln_final.weight"], sd["ln_final.bias"])
    txt_f = x[torch.arange(C), case.eot] @ sd["text_projection"]
    img_f = img_f / img_f.norm(dim=-1, keepdim=True)
    txt_f = txt_f / txt_f.norm(dim=-1, keepdim=True)
    return sd["logit_scale"].exp() * img_f @ txt_f.t()


def modes(base, **over):
    d = {s: base for s in SITES + ["patch"]}
    d.update(over)
    return d


def main():
    name = next((a for a in sys.argv[1:] if not a.startswith("-")), "mudpt_vitb16_b4_s100")
    case = GoldenCase(name)
    torch.set_num_threads(8)

    def run(label, vis, txt, layer_from=0):
        t0 = time.time()
        with torch.no_grad():
            lg = forward(case, vis, txt, layer_from)
        e = (lg - case.logits).abs()
        print(f"{label:<64s} max {e.max().item():.2e}  rms {e.pow(2).mean().sqrt().item():.2e}   ({time.time() - t0:.1f} s)", flush=True)
        return e.max().item()

    print(f"fixture {name}: logit scale {case.frozen['logit_scale'].exp().item():.2f}, logits {tuple(case.logits.shape)}")
    run("fp32 everywhere (the oracle)", modes("f32"), modes("f32"))
    run("fp16 everywhere", modes("f16"), modes("f16"))
    run("bf16 everywhere", modes("bf16"), modes("bf16"))
    run("vision f16, text f32", modes("f16"), modes("f32"))
    run("vision f32, text f16", modes("f32"), modes("f16"))
    run("fp16 mode (r3): vision f16; text GEMM operands pairs, attention f16", modes("f16"), modes("pair", qk="f16", pv="f16"))
    run("exact mode (r3): pairs + f32 attention, both towers", modes("pair", qk="f32", pv="f32"), modes("pair", qk="f32", pv="f32"))
    print("-- one vision site at a time in f16, everything else f32")
    for s in SITES + ["patch"]:
        run(f"  only vision.{s} = f16", modes("f32", **{s: "f16"}), modes("f32"))
    print("-- one text site at a time in f16")
    for s in SITES:
        run(f"  only text.{s} = f16", modes("f32"), modes("f32", **{s: "f16"}))
    print("-- leave-one-out: all vision sites pair/f32 except one in f16 (text exact)")
    tx = modes("pair", qk="f32", pv="f32")
    for s in SITES + ["patch"]:
        run(f"  vision.{s} = f16, rest exact", modes("pair", **{"qk": "f32", "pv": "f32", s: "f16"}), tx)
    print("-- candidates (text tower exact throughout: it is 0.7 % of the step at C 11)")
    run("vision GEMM operands pairs, attention q/k/p/v f16", modes("pair", qk="f16", pv="f16"), tx)
    run("vision GEMM operands pairs, attention qk f32, pv f16", modes("pair", qk="f32", pv="f16"), tx)
    run("vision GEMM operands pairs, attention qk f16, pv f32", modes("pair", qk="f16", pv="f32"), tx)
    run("vision GEMM operands f8lo, attention f32", modes("f8lo", qk="f32", pv="f32"), tx)
    run("vision GEMM operands f8lo, attention pair (3-product split)", modes("f8lo", qk="pair", pv="pair"), tx)
    run("vision GEMM operands f8lo, attention f8lo", modes("f8lo"), tx)
    run("vision GEMM operands f8lo, attention f16", modes("f8lo", qk="f16", pv="f16"), tx)
    run("vision everything pair", modes("pair"), tx)
    if "--quick" not in sys.argv:
        print("-- exact from layer k of the vision tower on, f16 below")
        for k in (2, 4, 6, 8, 10):
            run(f"  vision exact from layer {k}", modes("pair", qk="f32", pv="f32"), tx, layer_from=k)
        print("-- pairs of GEMM sites")
        for a, b in itertools.combinations(["qkv", "out", "fc", "proj"], 2):
            run(f"  vision.{a},{b} = f16; rest exact", modes("pair", **{"qk": "f32", "pv": "f32", a: "f16", b: "f16"}), tx)


if __name__ == "__main__":
    main()
