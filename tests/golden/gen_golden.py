"""Generate the golden vectors under tests/golden/ by running the REFERENCE's own modules.

Run in the build container only (needs /root/reference, which never travels to the GPU box):

    python tests/golden/gen_golden.py            # writes tests/golden/*.npz
    python tests/golden/gen_golden.py --cocoop-only   # only the CoCoOp fixtures (trainers/cocoop.py)
    python tests/golden/gen_golden.py --defaults-only      # only the n_ctx 2 / depth 9 fixture (what the reference's scripts train)
    python tests/golden/gen_golden.py --cocoop-b32-only    # only the ViT-B/32 batch-1 CoCoOp fixture (the reference's CoCoOp yaml)
    python tests/golden/gen_golden.py --cocoop-many-only   # only the 48-class CoCoOp fixture (mixed prompt lengths)
    python tests/golden/gen_golden.py --many-only     # only the 208-class fixture (BASELINE configs[2]'s text-heavy shape)
    python tests/golden/gen_golden.py --scale100-only # the *_s100 fixtures: logit_scale = ln 100, what pretrained CLIP checkpoints hold

What runs: ``clip.model.CLIP`` and ``trainers.mudpt.CustomCLIP`` imported unmodified from
/root/reference; their parameters are overwritten with the seeded recipe of
``oracle.mudpt_oracle.make_frozen_state`` / ``make_trainable_state`` so that tests can rebuild the
same weights from (seed, rule) without storing them.  Four third-party packages the reference
imports are absent from this image (yacs, dassl, ftfy, torchvision — SURVEY.md §8c); the
generator puts inert placeholders for them on sys.path in a scratch directory: an attribute-dict
``CfgNode``, a no-op trainer registry/base class, ``fix_text = identity`` and empty transform
classes.  None of them touches arithmetic; every number in the fixtures comes out of the
reference's code running on torch CPU fp32.
"""
from __future__ import annotations

import os
import sys
import tempfile
import textwrap

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import mudpt_oracle as O  # noqa: E402

REFERENCE = "/root/reference"
CLASSNAMES = ["face", "leopard", "motorbike", "accordion", "airplane", "anchor", "ant", "barrel",
              "bass", "beaver", "binocular"]  # first Caltech-101 names, SURVEY.md §8(d)

PLACEHOLDERS = {
    "yacs/__init__.py": "",
    "yacs/config.py": """
        class CfgNode(dict):
            def __getattr__(self, k):
                try:
                    return self[k]
                except KeyError:
                    raise AttributeError(k)
            def __setattr__(self, k, v):
                self[k] = v
        """,
    "dassl/__init__.py": "",
    "dassl/engine.py": """
        class _Registry:
            def register(self):
                return lambda cls: cls
        TRAINER_REGISTRY = _Registry()
        class TrainerX:
            pass
        """,
    "dassl/metrics.py": "def compute_accuracy(*a, **k):\n    raise NotImplementedError\n",
    "dassl/utils.py": "def load_pretrained_weights(*a, **k):\n    raise NotImplementedError\n"
                      "def load_checkpoint(*a, **k):\n    raise NotImplementedError\n",
    "dassl/optim.py": "def build_optimizer(*a, **k):\n    raise NotImplementedError\n"
                      "def build_lr_scheduler(*a, **k):\n    raise NotImplementedError\n",
    "ftfy/__init__.py": "def fix_text(t):\n    return t\n",
    "torchvision/__init__.py": "",
    "torchvision/transforms.py": """
        class _T:
            def __init__(self, *a, **k):
                pass
        Compose = Resize = CenterCrop = ToTensor = Normalize = _T
        class InterpolationMode:
            BICUBIC = 3
        """,
}


def import_reference():
    sys.dont_write_bytecode = True
    d = tempfile.mkdtemp(prefix="mudpt_placeholders_")
    for rel, src in PLACEHOLDERS.items():
        p = os.path.join(d, rel)
        os.makedirs(os.path.dirname(p), exist_ok=True)
        with open(p, "w") as f:
            f.write(textwrap.dedent(src))
    sys.path.insert(0, d)
    sys.path.insert(0, REFERENCE)
    import clip  # noqa: F401
    from clip import model as clip_model
    from trainers import mudpt
    from yacs.config import CfgNode
    return clip, clip_model, mudpt, CfgNode


def seeded_images(cfg: O.Config, batch: int, seed: int) -> torch.Tensor:
    g = torch.Generator().manual_seed(seed)
    return torch.randn(batch, 3, cfg.image_size, cfg.image_size, generator=g)


# Word pool for the many-class fixture (ImageNet-style names): k-word names give class prompts of mixed token length, so the
# EOT positions spread over 7..17 of the 77 and the trimmed text tower runs Le > 9 with a different row count per class.
WORDS = ["tench", "goldfish", "shark", "hammerhead", "stingray", "rooster", "ostrich", "brambling", "goldfinch", "junco",
         "bunting", "robin", "bulbul", "jay", "magpie", "chickadee", "ouzel", "kite", "eagle", "vulture", "owl", "salamander",
         "newt", "axolotl", "bullfrog", "loggerhead", "terrapin", "iguana", "chameleon", "agama", "alligator", "triceratops",
         "thunder", "ringneck", "hognose", "vine", "night", "boa", "python", "cobra", "mamba", "rattlesnake", "sidewinder",
         "trilobite", "harvestman", "scorpion", "garden", "barn", "wolf", "tick", "centipede", "grouse", "ptarmigan", "prairie",
         "peacock", "quail", "partridge", "macaw", "cockatoo", "lorikeet", "coucal", "hornbill", "hummingbird", "toucan"]


def many_classnames(n: int = 208):
    """n distinct names of 1..9 pool words (deterministic): 'tench', 'goldfish shark', ... -- mixed BPE lengths."""
    names, k, i = [], 1, 0
    while len(names) < n:
        ws = [WORDS[(i + 7 * j * (k + 1)) % len(WORDS)] for j in range(k)]
        nm = " ".join(ws)
        if nm not in names:
            names.append(nm)
        i += 1
        k = 1 + (i * 5 + i // 3) % 9
    return names


def with_logit_scale(frozen: dict, logit_scale) -> dict:
    """logit_scale = None keeps CLIP's init value ln(1/0.07) (clip/model.py:777); a number stands for what a pretrained checkpoint
    loaded through clip/model.py:919 holds (every released CLIP has exp(logit_scale) = 100 within a fraction of a percent)."""
    if logit_scale is not None:
        frozen["logit_scale"] = torch.tensor(float(np.log(logit_scale)))
    return frozen


def run(cfg: O.Config, name: str, ctx_init: str, batch: int, frozen_seed: int, train_seed: int,
        image_seed: int, sample_big: bool, classnames=None, taps_wanted: bool = True, logit_scale=None):
    CLASSNAMES = classnames or globals()["CLASSNAMES"]
    clip, cm, mudpt, CN = import_reference()
    ycfg = CN(TRAINER=CN(NAME="MuDPT", MUDPT=CN(N_CTX=cfg.n_ctx, CTX_INIT=ctx_init,
                                                 DEEP_PROMPT_DEPTH=cfg.depth, PREC="fp32")),
              INPUT=CN(SIZE=(cfg.image_size, cfg.image_size)))
    ref_clip = cm.CLIP(cfg.embed_dim, cfg.image_size, cfg.v_layers, cfg.v_width, cfg.patch, cfg.ctx_len,
                       cfg.vocab, cfg.t_width, cfg.t_heads, cfg.t_layers, ycfg).float()
    frozen = with_logit_scale(O.make_frozen_state(cfg, frozen_seed), logit_scale)
    missing, unexpected = ref_clip.load_state_dict(frozen, strict=False)
    assert not unexpected, unexpected
    assert all("visual_ctx" in k for k in missing), missing  # only the vision-side trainables
    assert torch.equal(ref_clip.logit_scale.detach(), frozen["logit_scale"])
    model = mudpt.CustomCLIP(ycfg, CLASSNAMES, ref_clip)

    # ctx comes from the reference's own init (token embedding of ctx_init words, mudpt.py:57-64)
    tok = model.tokenized_prompts
    ctx_ids = [int(v) for v in clip.tokenize(ctx_init)[0, 1:1 + cfg.n_ctx]]
    params = O.make_trainable_state(cfg, train_seed, frozen, ctx_ids)
    ref_params = dict(model.named_parameters())
    assert torch.equal(ref_params["mudpt_prompt_learner.ctx"].detach(), params["mudpt_prompt_learner.ctx"])
    with torch.no_grad():
        for k in O.TRAINABLE_ORDER:
            ref_params[k].copy_(params[k])
    # reference freeze rule, trainers/mudpt.py:205-212
    for k, p in model.named_parameters():
        p.requires_grad_("prompt_learner" in k or "visual_ctx" in k)
    assert sorted(k for k, p in model.named_parameters() if p.requires_grad) == sorted(O.TRAINABLE_ORDER)

    taps = {}

    def hook(key):
        def fn(_m, _i, out):
            taps[key] = out[0].detach().permute(1, 0, 2).contiguous()  # LND -> NLD
        return fn

    if taps_wanted:
        for i, blk in enumerate(model.image_encoder.transformer.resblocks):
            blk.register_forward_hook(hook(f"visual.transformer.resblocks.{i}.out"))
        for i, blk in enumerate(model.text_encoder.transformer.resblocks):
            blk.register_forward_hook(hook(f"transformer.resblocks.{i}.out"))

    images = seeded_images(cfg, batch, image_seed)
    labels = torch.arange(batch) * 3 % len(CLASSNAMES) if classnames is None else (torch.arange(batch) * 101 + 17) % len(CLASSNAMES)
    model.train()
    logits = model(images)
    loss = torch.nn.functional.cross_entropy(logits, labels)
    loss.backward()
    grads = {k: ref_params[k].grad.detach().clone() for k in O.TRAINABLE_ORDER}

    out = {
        "config": np.array(repr(cfg.asdict())),
        "classnames": np.array(CLASSNAMES),
        "ctx_init": np.array(ctx_init),
        "seeds": np.array([frozen_seed, train_seed, image_seed], dtype=np.int64),
        "tokenized_prompts": tok.numpy().astype(np.int32),
        "ctx_token_ids": np.array(ctx_ids, dtype=np.int32),
        "labels": labels.numpy().astype(np.int64),
        "images_checksum": np.array([images.double().sum().item(), images.double().abs().sum().item()]),
        "frozen_checksum": np.array([frozen["visual.transformer.resblocks.0.attn.in_proj_weight"].double().sum().item(),
                                     frozen["token_embedding.weight"].double().abs().sum().item()]),
        "logits": logits.detach().numpy(),
        "loss": np.array(loss.item(), dtype=np.float64),
        "logit_scale": np.array(frozen["logit_scale"].item(), dtype=np.float32),  # the parameter (a log), as the state dict holds it
    }
    for k in O.TRAINABLE_ORDER:
        g = grads[k]
        out["grad_sum." + k] = np.array([g.double().sum().item(), g.double().pow(2).sum().sqrt().item()])
        if sample_big and g.dim() == 2 and g.numel() > 100000:
            out["grad_sample." + k] = g[::8, ::8].numpy()  # strided sample of the big Linear weights
        else:
            out["grad." + k] = g.numpy()
    keep = range(max(cfg.v_layers, cfg.t_layers)) if not sample_big else (0, 1, cfg.v_layers - 1)
    for i in keep:
        for pre in ("visual.transformer", "transformer"):
            key = f"{pre}.resblocks.{i}.out"
            if key in taps:
                t = taps[key]
                out["tap." + key] = (t[:, ::8, ::16] if sample_big else t).numpy()
    path = os.path.join(ROOT, "tests", "golden", name + ".npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: loss {loss.item():.6f}, logits[0,:3] {logits[0, :3].tolist()}, "
          f"{os.path.getsize(path) / 1e6:.2f} MB")


def run_cocoop(cfg: O.Config, name: str, ctx_init: str, batch: int, frozen_seed: int, train_seed: int, image_seed: int, classnames=None, logit_scale=None):
    """CoCoOp fixtures from the reference's own trainers/cocoop.py CustomCLIP over the vanilla CLIP (cfg=None, cocoop.py:38).
    classnames: None = the 11 benchmark names; a list = the many-class case (mixed prompt lengths: a different EOT row per class)."""
    CLASSNAMES = globals()["CLASSNAMES"] if classnames is None else list(classnames)
    from oracle import cocoop_oracle as CO
    clip, cm, _mudpt, CN = import_reference()
    from trainers import cocoop
    ycfg = CN(TRAINER=CN(NAME="CoCoOp", COCOOP=CN(N_CTX=cfg.n_ctx, CTX_INIT=ctx_init, PREC="fp32")),
              INPUT=CN(SIZE=(cfg.image_size, cfg.image_size)))
    ref_clip = cm.CLIP(cfg.embed_dim, cfg.image_size, cfg.v_layers, cfg.v_width, cfg.patch, cfg.ctx_len,
                       cfg.vocab, cfg.t_width, cfg.t_heads, cfg.t_layers, None).float()
    frozen = with_logit_scale(O.make_frozen_state(cfg, frozen_seed), logit_scale)
    missing, unexpected = ref_clip.load_state_dict(frozen, strict=False)
    assert not unexpected and not missing, (missing, unexpected)
    assert torch.equal(ref_clip.logit_scale.detach(), frozen["logit_scale"])
    model = cocoop.CustomCLIP(ycfg, CLASSNAMES, ref_clip)
    tok = model.tokenized_prompts
    ctx_ids = [int(v) for v in clip.tokenize(ctx_init)[0, 1:1 + cfg.n_ctx]]
    params = CO.make_trainable_state(cfg, train_seed, frozen, ctx_ids)
    ref_params = dict(model.named_parameters())
    assert torch.equal(ref_params["prompt_learner.ctx"].detach(), params["prompt_learner.ctx"])  # the reference's own CTX_INIT rule
    with torch.no_grad():
        for k in CO.TRAINABLE_ORDER:
            ref_params[k].copy_(params[k])
    for k, p in model.named_parameters():  # freeze rule, trainers/cocoop.py:222-226
        p.requires_grad_("prompt_learner" in k)
    assert sorted(k for k, p in model.named_parameters() if p.requires_grad) == sorted(CO.TRAINABLE_ORDER)
    images = seeded_images(cfg, batch, image_seed)
    labels = torch.arange(batch) * 3 % len(CLASSNAMES) if classnames is None else (torch.arange(batch) * 29 + 5) % len(CLASSNAMES)
    model.eval()
    with torch.no_grad():
        logits = model(images)          # eval mode returns logits (cocoop.py:198)
    model.train()
    loss = model(images, labels)        # training mode returns F.cross_entropy(logits, label) (cocoop.py:196-197)
    loss.backward()
    out = {
        "config": np.array(repr(cfg.asdict())), "classnames": np.array(CLASSNAMES), "ctx_init": np.array(ctx_init),
        "seeds": np.array([frozen_seed, train_seed, image_seed], dtype=np.int64),
        "tokenized_prompts": tok.numpy().astype(np.int32), "ctx_token_ids": np.array(ctx_ids, dtype=np.int32),
        "labels": labels.numpy().astype(np.int64),
        "images_checksum": np.array([images.double().sum().item(), images.double().abs().sum().item()]),
        "logits": logits.numpy(), "loss": np.array(loss.item(), dtype=np.float64),
        "logit_scale": np.array(frozen["logit_scale"].item(), dtype=np.float32),
    }
    for k in CO.TRAINABLE_ORDER:
        out["grad." + k] = ref_params[k].grad.detach().numpy()
    path = os.path.join(ROOT, "tests", "golden", name + ".npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: loss {loss.item():.6f}, logits[0,:3] {logits[0, :3].tolist()}, {os.path.getsize(path) / 1e6:.2f} MB")


TOKENIZER_STRINGS = [
    "a photo of a face.", "a photo of a binocular.", "X X X X water lily.", "a photo of a Boeing 737-200, a type of aircraft.",
    "a centered satellite photo of annual crop land.", "a photo of a 2012 Audi S5 Coupe.", "crème brûlée", "Dr. Strange's cat",
    "it's  a   DOG!!", "they're we've I'm you'll he'd can't", "hello &amp; goodbye &amp;amp; more", "1999 12345 3.14", "snake_case_name",
    "UPPER lower MiXeD", "a photo of a person doing Apply Eye Makeup.", "a photo of a cheese plate, a type of food.", "ñandú über naïve",
    "日本語 のテキスト", "emoji 😀 test", "tab\tand\nnewline", "  leading and trailing  ", "a-b-c d/e/f (g) [h] {i}", "<|startoftext|> inner <|endoftext|>",
    "a photo of a wandering albatross, a type of bird.", "itap of a motorbike.", "a bad photo of the accordion.", "!!!", "a", "",
    "a photo of a " + "very " * 40 + "long name.",
]


def run_tokenizer():
    """Token ids of the reference's own tokenizer (clip/simple_tokenizer.py via clip.tokenize) for a list of strings."""
    import json
    clip, _cm, _mudpt, _CN = import_reference()
    cases = []
    for t in TOKENIZER_STRINGS:
        ids = [int(v) for v in clip.tokenize(t, truncate=True)[0]]
        n = len(ids) - ids[::-1].index(49407)  # through the LAST end-of-text id (a string may contain the marker itself)
        cases.append({"text": t, "ids": ids[:n]})
    path = os.path.join(ROOT, "tests", "golden", "tokenizer_cases.json")
    with open(path, "w") as f:
        json.dump({"context_length": 77, "truncate": True, "cases": cases}, f, ensure_ascii=False, indent=0)
    print(f"wrote {path}: {len(cases)} strings")


if __name__ == "__main__":
    torch.manual_seed(0)
    if "--tokenizer-only" in sys.argv:
        run_tokenizer()
        sys.exit(0)
    if "--scale100-only" in sys.argv:
        # the same recipes with the logit scale pretrained CLIP checkpoints carry (exp(logit_scale) = 100 instead of the init value
        # 1/0.07 = 14.29): a 7x larger multiplier on the cosine, i.e. north_star's 1e-3 logit bound asks for 1e-5 on the cosine
        run(O.TINY, "mudpt_tiny_s100", "a photo", batch=3, frozen_seed=11, train_seed=12, image_seed=13, sample_big=True, taps_wanted=False, logit_scale=100.0)
        run(O.VIT_B16, "mudpt_vitb16_b4_s100", "a photo of a", batch=4, frozen_seed=0, train_seed=1, image_seed=1234, sample_big=True,
            taps_wanted=False, logit_scale=100.0)
        run_cocoop(O.TINY, "cocoop_tiny_s100", "a photo", batch=3, frozen_seed=21, train_seed=22, image_seed=23, logit_scale=100.0)
        run_cocoop(O.VIT_B16, "cocoop_vitb16_b2_s100", "a photo of a", batch=2, frozen_seed=0, train_seed=2, image_seed=4321, logit_scale=100.0)
        # the two other shapes of work at that scale: 208 class prompts of mixed length (length buckets, per-class EOT rows) and ViT-L/14@336
        # (depth 24, L = 581: the tiled attention kernels; embed 768)
        run(O.VIT_B16, "mudpt_vitb16_c208_b2_s100", "a photo of a", batch=2, frozen_seed=0, train_seed=3, image_seed=2468,
            sample_big=True, classnames=many_classnames(208), taps_wanted=False, logit_scale=100.0)
        run(O.VIT_L14_336, "mudpt_vitl14_336_b1_s100", "a photo of a", batch=1, frozen_seed=5, train_seed=6, image_seed=77, sample_big=True,
            taps_wanted=False, logit_scale=100.0)
        sys.exit(0)
    if "--many-only" in sys.argv:
        # BASELINE configs[2]'s shape of work at fixture size: ViT-B/16, 208 class prompts of mixed length (EOT 7..17), B = 2
        run(O.VIT_B16, "mudpt_vitb16_c208_b2", "a photo of a", batch=2, frozen_seed=0, train_seed=3, image_seed=2468,
            sample_big=True, classnames=many_classnames(208), taps_wanted=False)
        sys.exit(0)
    if "--vitl-only" in sys.argv:
        run(O.VIT_L14_336, "mudpt_vitl14_336_b1", "a photo of a", batch=1, frozen_seed=5, train_seed=6, image_seed=77, sample_big=True)
        sys.exit(0)
    if "--defaults-only" in sys.argv:
        # the configuration the reference's scripts actually train: they never pass TRAINER.MUDPT.*, so the code defaults apply -- N_CTX 2
        # (train.py:116-118) -- with the shipped yaml's depth 9 (configs/trainers/MuDPT/vit_b16_bz4_ep10_nctx4_depth9.yaml)
        import dataclasses
        run(dataclasses.replace(O.VIT_B16, n_ctx=2, depth=9), "mudpt_vitb16_n2_d9_b2", "a photo of a", batch=2, frozen_seed=0, train_seed=8, image_seed=99, sample_big=True)
        sys.exit(0)
    if "--cocoop-b32-only" in sys.argv:
        # the reference's own CoCoOp configuration: ViT-B/32, batch 1 (configs/trainers/CoCoOp/vit_b32_bz1_ep10_ctxv1.yaml)
        import dataclasses
        run_cocoop(dataclasses.replace(O.VIT_B16, patch=32), "cocoop_vitb32_b1", "a photo of a", batch=1, frozen_seed=3, train_seed=4, image_seed=55)
        sys.exit(0)
    if "--cocoop-many-only" in sys.argv:
        run_cocoop(O.VIT_B16, "cocoop_vitb16_c48_b2", "a photo of a", batch=2, frozen_seed=0, train_seed=6, image_seed=777, classnames=many_classnames(48))
        sys.exit(0)
    if "--cocoop-only" in sys.argv:
        run_cocoop(O.TINY, "cocoop_tiny", "a photo", batch=3, frozen_seed=21, train_seed=22, image_seed=23)
        run_cocoop(O.VIT_B16, "cocoop_vitb16_b2", "a photo of a", batch=2, frozen_seed=0, train_seed=2, image_seed=4321)
        run_cocoop(O.VIT_B16, "cocoop_vitb16_c48_b2", "a photo of a", batch=2, frozen_seed=0, train_seed=6, image_seed=777, classnames=many_classnames(48))
        sys.exit(0)
    run(O.TINY, "mudpt_tiny", "a photo", batch=3, frozen_seed=11, train_seed=12, image_seed=13, sample_big=False)
    run(O.VIT_B16, "mudpt_vitb16_b4", "a photo of a", batch=4, frozen_seed=0, train_seed=1, image_seed=1234,
        sample_big=True)
    run(O.VIT_L14_336, "mudpt_vitl14_336_b1", "a photo of a", batch=1, frozen_seed=5, train_seed=6, image_seed=77, sample_big=True)
    run(O.VIT_B16, "mudpt_vitb16_c208_b2", "a photo of a", batch=2, frozen_seed=0, train_seed=3, image_seed=2468,
        sample_big=True, classnames=many_classnames(208), taps_wanted=False)
    import dataclasses
    run(dataclasses.replace(O.VIT_B16, n_ctx=2, depth=9), "mudpt_vitb16_n2_d9_b2", "a photo of a", batch=2, frozen_seed=0, train_seed=8, image_seed=99, sample_big=True)
    run_cocoop(O.TINY, "cocoop_tiny", "a photo", batch=3, frozen_seed=21, train_seed=22, image_seed=23)
    run_cocoop(O.VIT_B16, "cocoop_vitb16_b2", "a photo of a", batch=2, frozen_seed=0, train_seed=2, image_seed=4321)
    run_cocoop(O.VIT_B16, "cocoop_vitb16_c48_b2", "a photo of a", batch=2, frozen_seed=0, train_seed=6, image_seed=777, classnames=many_classnames(48))
    run_cocoop(dataclasses.replace(O.VIT_B16, patch=32), "cocoop_vitb32_b1", "a photo of a", batch=1, frozen_seed=3, train_seed=4, image_seed=55)
    run_tokenizer()
