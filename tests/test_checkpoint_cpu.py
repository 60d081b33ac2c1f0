"""Checkpoint-FILE ingestion on CPU: the backbone loader of ``build_model`` and the plugin's ``load_model``.

Reference behaviour restated by the product code under test:
* ``clip.load`` (clip/clip.py:95-144): ``torch.jit.load`` of an OpenAI archive, falling back to ``torch.load`` of a plain state
  dict; ``build_model`` (clip/model.py:881-921) infers every dimension from tensor SHAPES and drops the three scalar entries
  ``input_resolution / context_length / vocab_size``; ``convert_weights`` (clip/model.py:857-878) stores Linear / Conv /
  attention weights, ``proj`` and ``text_projection`` in fp16;
* ``MuDPT.load_model`` (trainers/mudpt.py:270-302): ``model.pth.tar-<epoch>`` with ``state_dict`` / ``epoch``, the fixed token
  buffers dropped, ``strict=False`` -- a reference checkpoint also carries the whole frozen backbone (it registers the
  full CustomCLIP), which must be ignored.
No pretrained CLIP file exists offline, so the files are made from ``synth.random_clip_state`` (real-weight parity: unpinned)."""
import torch
from torch import nn

from mudpt_amd import dassl_lite, synth, trainer
from mudpt_amd.model import ModelShape

TINY = ModelShape(image_size=32, patch=16, v_width=192, v_layers=3, v_heads=3, t_width=128, t_layers=3, t_heads=2, ctx_len=77,
                  embed_dim=128, n_ctx=2, depth=2)
FP16_SUFFIXES = ("conv1.weight", "in_proj_weight", "in_proj_bias", "out_proj.weight", "out_proj.bias", "c_fc.weight", "c_fc.bias",
                 "c_proj.weight", "c_proj.bias", "visual.proj", "text_projection")


def as_checkpoint(sd):
    """What an OpenAI CLIP file holds: convert_weights' fp16 tensors + the three scalar entries build_model deletes."""
    out = {k: (v.half() if k.endswith(FP16_SUFFIXES) else v.clone()) for k, v in sd.items()}
    out["input_resolution"], out["context_length"], out["vocab_size"] = torch.tensor(32), torch.tensor(77), torch.tensor(synth.VOCAB)
    return out


class _Tree(nn.Module):
    """Container with buffers under dotted CLIP keys (what ``torch.jit.load(...).state_dict()`` of the OpenAI archive returns)."""

    def __init__(self, sd):
        super().__init__()
        for k, v in sd.items():
            mod = self
            *path, leaf = k.split(".")
            for part in path:
                if part not in mod._modules:
                    mod.add_module(part, _Tree({}))
                mod = mod._modules[part]
            mod.register_buffer(leaf, v.clone())


def cfg_with_path(path):
    cfg = dassl_lite.default_cfg()
    cfg.MODEL.BACKBONE.PATH = str(path)
    return cfg


def test_plain_state_dict_file_roundtrip(tmp_path):
    ck = as_checkpoint(synth.random_clip_state(TINY, seed=3))
    path = tmp_path / "tiny_clip_state.pt"
    torch.save(ck, path)
    got = trainer.load_clip_state_dict(cfg_with_path(path))  # torch.jit.load refuses a plain pickle -> weights-only torch.load
    assert sorted(got) == sorted(ck)
    for k, v in ck.items():
        assert got[k].dtype == v.dtype and torch.equal(got[k], v), k
    assert got["visual.transformer.resblocks.0.attn.in_proj_weight"].dtype == torch.float16
    assert got["visual.positional_embedding"].dtype == torch.float32
    assert ModelShape.from_state_dict(got, n_ctx=2, depth=2) == TINY


def test_jit_archive_roundtrip(tmp_path):
    ck = as_checkpoint(synth.random_clip_state(TINY, seed=4))
    path = tmp_path / "tiny_clip_jit.pt"
    torch.jit.save(torch.jit.script(_Tree(ck)), str(path))
    got = trainer.load_clip_state_dict(cfg_with_path(path))  # the torch.jit.load branch (clip/clip.py:121-123)
    assert sorted(got) == sorted(ck)
    for k, v in ck.items():
        assert got[k].dtype == v.dtype and torch.equal(got[k], v), k
    assert ModelShape.from_state_dict(got, n_ctx=2, depth=2) == TINY


def test_shape_inference_for_the_published_architectures():
    """clip/model.py:885-904 on ViT-B/16 and ViT-L/14@336 shaped dicts (meta tensors: shapes only, no 1.7 GB of weights)."""
    for want in (ModelShape(n_ctx=4, depth=12),
                 ModelShape(image_size=336, patch=14, v_width=1024, v_layers=24, v_heads=16, t_width=768, t_layers=12, t_heads=12,
                            embed_dim=768, n_ctx=4, depth=24)):
        with torch.device("meta"):
            sd = synth.random_clip_state_shapes(want)
        got = ModelShape.from_state_dict(sd, n_ctx=want.n_ctx, depth=want.depth)
        assert got == want, (got, want)


class _Holder(nn.Module):
    pass


def _trainable_module(shape: ModelShape, fill: float):
    """A module with the ten trainables under the reference's dotted names (what CustomCLIP registers, mudpt_amd/model.py)."""
    n, D1, dt, dv, e = shape.n_ctx, shape.depth - 1, shape.t_width, shape.v_width, shape.embed_dim
    shapes = {"mudpt_prompt_learner.ctx": (n, dt), "mudpt_prompt_learner.deep_prompts": (D1, n, dt),
              "mudpt_prompt_learner.embed_projection.weight": (dv, dt), "mudpt_prompt_learner.embed_projection.bias": (dv,),
              "mudpt_prompt_learner.deep_projections.weight": (dv, dt), "mudpt_prompt_learner.deep_projections.bias": (dv,),
              "image_encoder.visual_ctx": (n, dv), "image_encoder.visual_ctx_deep_prompts": (D1, n, dv),
              "image_encoder.visual_ctx_deep_projections.weight": (e, dv), "image_encoder.visual_ctx_deep_projections.bias": (e,)}
    root = _Holder()
    for k, s in shapes.items():
        mod = root
        *path, leaf = k.split(".")
        for part in path:
            if not hasattr(mod, part):
                setattr(mod, part, _Holder())
            mod = getattr(mod, part)
        mod.register_parameter(leaf, nn.Parameter(torch.full(s, fill)))
    return root, shapes


def test_load_model_reads_a_reference_shaped_checkpoint(tmp_path):
    """A checkpoint as the REFERENCE writes it: the ten trainables, the fixed token buffers and the whole frozen backbone under
    CustomCLIP's names (trainers/mudpt.py:227 registers the full model).  load_model keeps the ten, drops the rest."""
    model, shapes = _trainable_module(TINY, 0.0)
    g = torch.Generator().manual_seed(5)
    want = {k: torch.randn(s, generator=g) for k, s in shapes.items()}
    state = dict(want)
    state["mudpt_prompt_learner.token_prefix"] = torch.randn(11, 1, TINY.t_width, generator=g)
    state["mudpt_prompt_learner.token_suffix"] = torch.randn(11, 77 - 1 - TINY.n_ctx, TINY.t_width, generator=g)
    for k, v in synth.random_clip_state(TINY, seed=6).items():  # frozen backbone as CustomCLIP holds it (trainers/mudpt.py:159-168)
        if k.startswith("visual."):
            state["image_encoder." + k[len("visual."):]] = v
        elif k.startswith("transformer."):
            state["text_encoder." + k] = v
        elif k in ("positional_embedding", "ln_final.weight", "ln_final.bias", "text_projection"):
            state["text_encoder." + k] = v
    state["logit_scale"] = torch.tensor(4.6)
    d = tmp_path / "MultimodalDeepPromptTuning"
    d.mkdir()
    torch.save({"state_dict": state, "epoch": 7, "optimizer": None}, d / "model.pth.tar-7")
    t = object.__new__(trainer.MuDPT)
    t._models = {"MultimodalDeepPromptTuning": model}
    t.load_model(str(tmp_path), epoch=7)
    for k, p in model.named_parameters():
        assert torch.equal(p.detach(), want[k]), k
    torch.save({"state_dict": state, "epoch": 9}, d / "model-best.pth.tar")  # epoch=None -> the best model (:277-281)
    with torch.no_grad():
        for p in model.parameters():
            p.zero_()
    t.load_model(str(tmp_path))
    assert all(torch.equal(p.detach(), want[k]) for k, p in model.named_parameters())
