"""Host-side mirror of the reference's ``CustomCLIP`` (trainers/mudpt.py:159-184) over libmudpt_hip.so.

``CustomCLIP`` here is an ``nn.Module`` whose ONLY tensors are the 10 trainable ones, registered under the
reference's state-dict names (``mudpt_prompt_learner.ctx`` ... ``image_encoder.visual_ctx_deep_projections.bias``,
trainers/mudpt.py:205-218) as views of one flat fp32 bucket, so a torch / Dassl optimizer, ``state_dict()``
and the reference's ``load_model`` (``strict=False``) work unchanged.  The frozen CLIP weights live inside
the library in its own HBM layout.  torch supplies device memory and streams only.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass
from typing import Dict, Optional, Sequence

import torch
from torch import nn

from . import capi


@dataclass(frozen=True)
class ModelShape:
    """Dimensions the reference infers from the checkpoint (clip/model.py:885-904) plus the prompt config."""
    image_size: int = 224
    patch: int = 16
    v_width: int = 768
    v_layers: int = 12
    v_heads: int = 12
    t_width: int = 512
    t_layers: int = 12
    t_heads: int = 8
    ctx_len: int = 77
    embed_dim: int = 512
    n_ctx: int = 4
    depth: int = 12

    @staticmethod
    def from_state_dict(sd: Dict[str, torch.Tensor], n_ctx: int, depth: int) -> "ModelShape":
        """Same inference rules as clip/model.py:881-904 build_model (ViT branch)."""
        v_width = sd["visual.conv1.weight"].shape[0]
        v_layers = len([k for k in sd if k.startswith("visual.") and k.endswith(".attn.in_proj_weight")])
        patch = sd["visual.conv1.weight"].shape[-1]
        grid = round((sd["visual.positional_embedding"].shape[0] - 1) ** 0.5)
        t_width = sd["ln_final.weight"].shape[0]
        t_layers = len(set(k.split(".")[2] for k in sd if k.startswith("transformer.resblocks")))
        return ModelShape(image_size=patch * grid, patch=patch, v_width=v_width, v_layers=v_layers, v_heads=v_width // 64,
                          t_width=t_width, t_layers=t_layers, t_heads=t_width // 64,
                          ctx_len=sd["positional_embedding"].shape[0], embed_dim=sd["text_projection"].shape[1],
                          n_ctx=n_ctx, depth=depth)


class _Holder(nn.Module):
    """Namespace module so parameters get the reference's dotted state-dict keys."""


class CustomCLIP(nn.Module):
    def __init__(self, shape: ModelShape, clip_state: Dict[str, torch.Tensor], tokenized_prompts: torch.Tensor,
                 ctx_token_ids: Optional[Sequence[int]] = None, max_batch: int = 256, dtype: str = "bf16",
                 device: str = "cuda:0", seed: Optional[int] = None, variant: str = "mudpt", knobs: Optional[Dict[str, int]] = None,
                 class_shard: Optional[Sequence[int]] = None, group=None):
        super().__init__()
        if not torch.cuda.is_available():
            raise capi.MudptError("mudpt_amd needs an MI355X (HIP device); there is no CPU path in the product")
        self.lib = capi.load()
        self.shape = shape
        self.device = torch.device(device)
        self.dtype_name = dtype
        self.n_cls = int(tokenized_prompts.shape[0])
        self.max_batch = int(max_batch)
        self.tokenized_prompts = tokenized_prompts.clone()
        # "cocoop": the same library runs trainers/cocoop.py's CustomCLIP (vanilla vision tower, meta_net, one text-tower pass
        # per (image, class) pair); the module then owns ctx + meta_net under the reference's names (prompt_learner.*)
        self.variant = variant
        cfg = capi.Config(shape.image_size, shape.patch, shape.v_width, shape.v_layers, shape.v_heads, shape.t_width,
                          shape.t_layers, shape.t_heads, shape.ctx_len, shape.embed_dim, shape.n_ctx, shape.depth,
                          self.n_cls, self.max_batch, {"bf16": capi.BF16, "fp16": capi.F16, "fp32": capi.F32}[dtype],
                          {"mudpt": capi.VARIANT_MUDPT, "cocoop": capi.VARIANT_COCOOP}[variant])
        torch.cuda.set_device(self.device)
        h = C.c_void_p()
        capi.check(self.lib.mudpt_create(C.byref(cfg), C.byref(h)), "create")
        self._h = h
        for name, value in (knobs or {}).items():  # per-handle tuning knobs (mudpt_model_set): A/B runs and tests
            self.set_knob(name, value)
        # frozen weights, by OpenAI CLIP key (clip/model.py:919 load_state_dict)
        for k, v in clip_state.items():
            if k == "token_embedding.weight" or not isinstance(v, torch.Tensor):
                continue
            t = v.detach().to("cpu", torch.float32).contiguous()
            capi.check(self.lib.mudpt_set_weight(h, k.encode(), capi.ptr(t), t.numel()), f"set_weight({k})")
        # class-parallel text tower (include/mudpt.h): this rank encodes classes [c0, c1) of the n_cls; ``group`` is the process group
        # of the two exchanges (None = the default group)
        self.class_shard, self.group = None, group
        if class_shard is not None and tuple(class_shard) != (0, self.n_cls):
            assert variant == "mudpt", "CoCoOp's text features depend on the image: shard the batch, not the classes"
            self.class_shard = (int(class_shard[0]), int(class_shard[1]))
            capi.check(self.lib.mudpt_set_class_shard(h, *self.class_shard), "set_class_shard")
        # class prompts: token_embedding(tokenized) and the EOT position (trainers/mudpt.py:85-90,154)
        emb_w = clip_state["token_embedding.weight"].detach().to("cpu", torch.float32)
        tok = tokenized_prompts.to("cpu").long()
        emb = emb_w[tok].contiguous()
        eot = tok.argmax(dim=-1).to(torch.int32).contiguous()
        capi.check(self.lib.mudpt_set_class_prompts(h, capi.ptr(emb), capi.ptr(eot)), "set_class_prompts")
        # the flat parameter / gradient buckets and the 10 named views
        total = self.lib.mudpt_param_numel(h)
        self.flat_params = torch.zeros(total, dtype=torch.float32, device=self.device)
        self.flat_grads = torch.zeros(total, dtype=torch.float32, device=self.device)
        capi.check(self.lib.mudpt_bind_params(h, capi.ptr(self.flat_params), capi.ptr(self.flat_grads)), "bind_params")
        self.param_names = []
        for i in range(self.lib.mudpt_param_count(h)):
            name, off, numel, ndim, shp = C.c_char_p(), C.c_size_t(), C.c_size_t(), C.c_int32(), (C.c_int64 * 3)()
            capi.check(self.lib.mudpt_param_info(h, i, C.byref(name), C.byref(off), C.byref(numel), C.byref(ndim), C.byref(shp)))
            key = name.value.decode()
            view = self.flat_params[off.value:off.value + numel.value].view(*[int(shp[j]) for j in range(ndim.value)])
            p = nn.Parameter(view, requires_grad=True)
            p.grad = self.flat_grads[off.value:off.value + numel.value].view_as(view)
            mod = self
            *path, leaf = key.split(".")
            for part in path:  # namespace modules along the reference's dotted key
                if not hasattr(mod, part):
                    setattr(mod, part, _Holder())
                mod = getattr(mod, part)
            mod.register_parameter(leaf, p)
            self.param_names.append(key)
        self._init_trainables(emb_w, ctx_token_ids, seed)
        self._cp_feat = self._cp_dfeat = None
        if variant == "mudpt":  # the [n_cls, embed] text-feature table and its gradient (library-owned): operands of the class-parallel exchanges
            f, df, n = C.c_void_p(), C.c_void_p(), C.c_size_t()
            capi.check(self.lib.mudpt_cp_buffers(h, C.byref(f), C.byref(df), C.byref(n)), "cp_buffers")
            self._cp_feat, self._cp_dfeat = (capi.device_view(q.value, n.value, self.device).view(self.n_cls, shape.embed_dim) for q in (f, df))
        self._loss = torch.zeros(4, dtype=torch.float32, device=self.device)
        self._text_version = None  # flat_params._version the library's cached text features belong to
        self.loss_scale = 128.0    # the library's default (mudpt_set_loss_scale)

    # -- initialisation of the trainables, trainers/mudpt.py:57-81 and clip/model.py:512-519 ---------------------
    def _init_trainables(self, emb_w, ctx_token_ids, seed):
        g = torch.Generator().manual_seed(seed) if seed is not None else None
        sd = dict(self.named_parameters())
        with torch.no_grad():
            for k, p in sd.items():
                if k.endswith(".weight") or k.endswith(".bias"):
                    fan_in = sd[k.rsplit(".", 1)[0] + ".weight"].shape[1]
                    v = (torch.rand(p.shape, generator=g) * 2 - 1) / math.sqrt(fan_in)  # nn.Linear default
                else:
                    v = 0.02 * torch.randn(p.shape, generator=g)  # nn.init.normal_(std=0.02)
                p.copy_(v)
            if ctx_token_ids is not None:  # CTX_INIT words -> their token embeddings
                sd[self.ctx_key].copy_(emb_w[list(ctx_token_ids)])

    @property
    def ctx_key(self) -> str:
        return "mudpt_prompt_learner.ctx" if self.variant == "mudpt" else "prompt_learner.ctx"

    def set_knob(self, name: str, value: int):
        """``mudpt_model_set``: "gemm_variant", "lp_grad" and the split-operand knobs ("vis_lo", "txt_lo", "vis_sites", "txt_sites",
        "vis_exact_attn", "txt_exact_attn": include/mudpt.h MUDPT_F32) any time; "txt_trim" / "txt_buckets" only through
        ``knobs=`` at construction (before the class prompts are ingested)."""
        capi.check(self.lib.mudpt_model_set(self._h, name.encode(), int(value)), f"model_set({name})")

    def text_layout(self):
        """(token rows, length buckets, longest kept length) of one text-tower pass (``mudpt_text_layout``)."""
        r, b, l = C.c_int32(), C.c_int32(), C.c_int32()
        capi.check(self.lib.mudpt_text_layout(self._h, C.byref(r), C.byref(b), C.byref(l)), "text_layout")
        return r.value, b.value, l.value

    def set_params(self, tensors: Dict[str, torch.Tensor]):
        self._text_version = None
        with torch.no_grad():
            for k, p in self.named_parameters():
                if k in tensors:
                    p.copy_(tensors[k])

    def grads(self) -> Dict[str, torch.Tensor]:
        return {k: p.grad for k, p in self.named_parameters()}

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # -- trainers/mudpt.py:170-184 ---------------------------------------------------------------------------------
    def _check_images(self, image: torch.Tensor):
        S = self.shape.image_size
        # the library is told only B: a wrong-sized batch would make the patch gather read out of bounds (trainers/mudpt.py:55 asserts
        # the configured size; this asserts the data)
        assert image.dim() == 4 and tuple(image.shape[1:]) == (3, S, S), f"images must be [B, 3, {S}, {S}], got {tuple(image.shape)}"
        assert 0 < image.shape[0] <= self.max_batch, f"batch {image.shape[0]} outside 1..max_batch={self.max_batch}"

    def invalidate_text_cache(self):
        """Parameters changed behind the bucket's version counter (``p.data`` writes, optimizers, the library's SGD): the next eval
        forward recomputes the text features."""
        self._text_version = None

    def load_state_dict(self, *args, **kwargs):
        self._text_version = None
        return super().load_state_dict(*args, **kwargs)

    def forward(self, image: torch.Tensor) -> torch.Tensor:
        self._check_images(image)
        image = image.to(self.device, torch.float32).contiguous()
        B = image.shape[0]
        logits = torch.empty(B, self.n_cls, dtype=torch.float32, device=self.device)
        # eval mode: the text features only depend on the parameters; recompute them only when the flat bucket's version
        # counter moved (the reference re-runs the text tower for every test batch)
        version = self.flat_params._version
        reuse = (not self.training) and self._text_version == version and self.variant == "mudpt"  # CoCoOp's text features depend on the image
        if self.class_shard is not None:
            capi.check(self.lib.mudpt_cp_forward(self._h, capi.ptr(image), B, 1 if reuse else 0, self._stream()), "cp_forward")
            if not reuse:
                self._exchange(self._cp_feat)
            capi.check(self.lib.mudpt_cp_head(self._h, None, B, 1.0, None, capi.ptr(logits), 1 if reuse else 0, self._stream()), "cp_head")
        else:
            capi.check(self.lib.mudpt_forward_ex(self._h, capi.ptr(image), B, capi.ptr(logits), 1 if reuse else 0, self._stream()), "forward")
        self._text_version = version
        return logits

    def _exchange(self, table: torch.Tensor, async_op: bool = False):
        """Sum of a [n_cls, embed] table over the ranks of the class-parallel group (rows of other ranks are zero in the feature table,
        so the sum is the gather, bit for bit, also for uneven shards)."""
        import torch.distributed as dist
        # only a class-SHARDED handle exchanges: on an unsharded one every rank holds the full table already, and summing world
        # identical copies would scale d(features) -- and with it every text-side gradient -- by world without any error
        if self.class_shard is not None and dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            return dist.all_reduce(table, group=self.group, async_op=async_op)
        return None

    def forward_backward_cp(self, image: torch.Tensor, label: torch.Tensor, grad_scale: float = 1.0, return_logits: bool = False):
        """The training step in class-parallel phases (include/mudpt.h): towers forward with this rank's classes -> sum of the feature
        table -> head over the local images and all classes -> sum of d(features), overlapped with the vision backward -> text backward
        over this rank's classes.  On an unsharded handle (every rank encodes all classes) the phases run without any exchange."""
        self._check_images(image)
        assert label.shape == (image.shape[0],), f"labels must be [B], got {tuple(label.shape)}"
        image = image.to(self.device, torch.float32).contiguous()
        label = label.to(self.device, torch.int64).contiguous()
        B = image.shape[0]
        logits = torch.empty(B, self.n_cls, dtype=torch.float32, device=self.device) if return_logits else None
        capi.check(self.lib.mudpt_cp_forward(self._h, capi.ptr(image), B, capi.FWD_TRAINING, self._stream()), "cp_forward")
        self._exchange(self._cp_feat)
        capi.check(self.lib.mudpt_cp_head(self._h, capi.ptr(label), B, grad_scale, capi.ptr(self._loss), capi.ptr(logits), 0, self._stream()), "cp_head")
        work = self._exchange(self._cp_dfeat, async_op=True)
        capi.check(self.lib.mudpt_cp_backward(self._h, capi.CP_VISION, self._stream()), "cp_backward(vision)")
        if work is not None:
            work.wait()  # the current stream waits for the collective; the host does not
        capi.check(self.lib.mudpt_cp_backward(self._h, capi.CP_TEXT, self._stream()), "cp_backward(text)")
        self._text_version = self.flat_params._version
        return (self._loss[0], logits) if return_logits else self._loss[0]

    # -- trainers/mudpt.py:249-251 minus the optimizer step: loss (device scalar) + .grad of the 10 tensors ---------------
    def forward_backward(self, image: torch.Tensor, label: torch.Tensor, grad_scale: float = 1.0,
                         return_logits: bool = False):
        if self.class_shard is not None:
            return self.forward_backward_cp(image, label, grad_scale, return_logits)
        self._check_images(image)
        assert label.shape == (image.shape[0],), f"labels must be [B], got {tuple(label.shape)}"
        image = image.to(self.device, torch.float32).contiguous()
        label = label.to(self.device, torch.int64).contiguous()
        B = image.shape[0]
        logits = torch.empty(B, self.n_cls, dtype=torch.float32, device=self.device) if return_logits else None
        capi.check(self.lib.mudpt_forward_backward(self._h, capi.ptr(image), capi.ptr(label), B, grad_scale,
                                                   capi.ptr(self._loss), capi.ptr(logits), self._stream()), "forward_backward")
        self._text_version = self.flat_params._version  # the step's forward left this version's text features in the library
        return (self._loss[0], logits) if return_logits else self._loss[0]

    def set_loss_scale(self, scale: float):
        """Static scale of the backward pass (include/mudpt.h mudpt_set_loss_scale; default 128 per sample); the trainer plugins move it
        like torch's GradScaler does (halve on overflow, grow back after a run of clean steps)."""
        capi.check(self.lib.mudpt_set_loss_scale(self._h, float(scale)), "set_loss_scale")
        self.loss_scale = float(scale)

    def sgd_step(self, lr: float, momentum: float = 0.9, weight_decay: float = 5e-4, dampening: float = 0.0, nesterov: bool = False):
        capi.check(self.lib.mudpt_sgd_step(self._h, lr, momentum, weight_decay, dampening, int(nesterov), self._stream()), "sgd_step")
        self._text_version = None  # the library wrote the parameters behind torch's version counter

    def profile(self, enable):
        """False / 0: off; True / 1: bracket the persistent GEMM launches only; an int > 1: bit mask of PROF_CLASSES (31 = all)."""
        capi.check(self.lib.mudpt_profile_enable(self._h, int(enable)), "profile_enable")

    def profile_read(self):
        """(summed GEMM ms, summed algorithmic GEMM FLOPs, launches) since the last enable / read."""
        ms, fl, n = C.c_double(), C.c_double(), C.c_int64()
        capi.check(self.lib.mudpt_profile_read(self._h, C.byref(ms), C.byref(fl), C.byref(n)), "profile_read")
        return ms.value, fl.value, n.value

    PROF_CLASSES = ("gemm_pp", "ln_fwd", "ln_bwd", "attn_fwd", "attn_bwd")

    def profile_read_classes(self):
        """({class: (ms, work, launches)}, executed MFMA FLOPs) since the last enable / read; work = algorithmic FLOPs for "gemm_pp",
        algorithmic HBM bytes for the LayerNorm / attention classes (vision tower launches only)."""
        ms, work, n, ex = (C.c_double * 5)(), (C.c_double * 5)(), (C.c_int64 * 5)(), C.c_double()
        capi.check(self.lib.mudpt_profile_read_classes(self._h, C.byref(ms), C.byref(work), C.byref(n), C.byref(ex)), "profile_read_classes")
        return {k: (ms[i], work[i], n[i]) for i, k in enumerate(self.PROF_CLASSES)}, ex.value

    def debug_read(self, name: str, batch: int) -> torch.Tensor:
        """Flat fp32 host copy of an internal activation of the last call (test hook, see include/mudpt.h)."""
        n = C.c_size_t()
        capi.check(self.lib.mudpt_debug_read(self._h, name.encode(), batch, None, 0, C.byref(n)), "debug_read")
        out = torch.empty(n.value, dtype=torch.float32)
        capi.check(self.lib.mudpt_debug_read(self._h, name.encode(), batch, capi.ptr(out), n.value, C.byref(n)), "debug_read")
        return out

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            torch.cuda.synchronize(self.device)
            self.lib.mudpt_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
