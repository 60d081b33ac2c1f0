"""Drop-in ``MuDPT`` trainer plugin: the reference's ``trainers/mudpt.py:187-302`` surface over libmudpt_hip.so.

Same class name, registry name, hooks and error behaviour as the reference plugin:
``check_cfg`` (:189), ``build_model`` (:192), ``forward_backward`` (:235), ``parse_batch_train`` (:263),
inherited ``model_inference`` (``self.model(input)``) and ``load_model`` (:270).  With Dassl installed it subclasses
``dassl.engine.TrainerX`` and registers in Dassl's ``TRAINER_REGISTRY`` so ``train.py --trainer MuDPT`` and
``scripts/mudpt/*.sh`` run unchanged; without Dassl it falls back to ``dassl_lite``.
"""
from __future__ import annotations

import os
import os.path as osp

import torch

try:  # the real framework, when present
    from dassl.engine import TRAINER_REGISTRY, TrainerX
    from dassl.optim import build_optimizer, build_lr_scheduler
    from dassl.utils import load_checkpoint, load_pretrained_weights
    HAVE_DASSL = True
except ImportError:  # Dassl is not installed in the build image nor on the GPU box
    from .dassl_lite import TRAINER_REGISTRY, TrainerX, build_optimizer, build_lr_scheduler, load_checkpoint, load_pretrained_weights
    HAVE_DASSL = False

from . import parallel, synth
from .model import CustomCLIP, ModelShape

PREC_TO_DTYPE = {"fp16": "fp16", "fp32": "fp32", "amp": "bf16"}


def load_clip_state_dict(cfg):
    """trainers/mudpt.py:20-38 load_clip_to_cpu: a JIT archive or a plain state dict at MODEL.BACKBONE.PATH.
    The reference's name-based branch is a network download (and dead code); offline it is
    replaced by an explicit opt-in to seeded random weights (MODEL.BACKBONE.SYNTHETIC_SEED) for benchmarking."""
    path = cfg.MODEL.BACKBONE.PATH
    if path:
        print(f"Loading CLIP backbone: {cfg.MODEL.BACKBONE.NAME} from {path}")
        try:
            return torch.jit.load(path, map_location="cpu").eval().state_dict()
        except RuntimeError:
            return torch.load(path, map_location="cpu", weights_only=True)
    seed = cfg.MODEL.BACKBONE.get("SYNTHETIC_SEED", None) if hasattr(cfg.MODEL.BACKBONE, "get") else None
    if seed is None:
        raise RuntimeError("MODEL.BACKBONE.PATH is empty and no network is available to download "
                           f"{cfg.MODEL.BACKBONE.NAME}; set MODEL.BACKBONE.PATH to a local CLIP checkpoint")
    print(f"Building random-initialised CLIP {cfg.MODEL.BACKBONE.NAME} (seed {seed}) -- synthetic benchmark weights")
    return None


def tokenize_prompts(prompts, ctx_len=77, near=None):
    """clip.tokenize (clip/clip.py:199-239) by the native BPE tokenizer (mudpt_amd/tokenizer.py); the merge table is looked for
    at $MUDPT_BPE_VOCAB, next to the checkpoint (``near``) and in an importable clip package.  Without a merge table only the
    benchmark prompts (recorded ids) can be served."""
    from . import tokenizer
    try:
        return tokenizer.tokenize(list(prompts), ctx_len, near=near)
    except RuntimeError as no_vocab:
        if "merge table" not in str(no_vocab):
            raise
        table = {f"a photo of a {n}.": i for i, n in enumerate(synth.BENCH_CLASSNAMES)}
        tok = synth.bench_tokenized_prompts(ctx_len)
        try:
            return torch.stack([tok[table[p]] for p in prompts])
        except KeyError as e:
            raise RuntimeError(f"no BPE merge table available for prompt {e}: {no_vocab}") from None


LOSS_SCALE_INIT, LOSS_SCALE_MIN, LOSS_SCALE_GROWTH_INTERVAL = 128.0, 1.0, 2000


def data_parallel_step(trainer, batch):
    """One training step of either plugin (trainers/mudpt.py:235-261, trainers/cocoop.py:246-276) in data-parallel form:
    forward + cross-entropy + backward in ONE library call on this rank's images (gradient of loss / world), ONE all-reduce of the
    flat bucket, the consensus on what every rank sees (parallel.step_consensus), then the optimizer step -- so replicas stay bitwise
    identical.  The logged loss is the GLOBAL-batch mean, as the reference's (nn.DataParallel gathers the logits before
    F.cross_entropy, trainers/mudpt.py:249-256).

    Loss scaling follows torch.cuda.amp.GradScaler (the reference's amp path, trainers/mudpt.py:228,239-246): the backward runs on
    per-sample gradients times a power-of-two scale inside the library; gradients that come back non-finite (an fp16 copy of a token
    gradient overflowed) make every rank SKIP the optimizer step and halve the scale; LOSS_SCALE_GROWTH_INTERVAL clean steps in a row
    double it again, up to the initial value.  A non-finite LOSS is an error, as in Dassl's model_backward_and_update."""
    image, label = trainer.parse_batch_train(batch)
    model = trainer.model
    loss = model.forward_backward(image, label, grad_scale=parallel.grad_scale())
    parallel.allreduce_grads(model.flat_grads)
    loss_ok, grads_ok, global_loss = parallel.step_consensus(loss, model.flat_grads)
    if not loss_ok:
        raise FloatingPointError("Loss is infinite or NaN!")
    state = trainer.__dict__.setdefault("_loss_scale_state", {"scale": getattr(model, "loss_scale", LOSS_SCALE_INIT), "clean": 0, "skipped": 0})
    if grads_ok:
        trainer.optim.step()
        model.invalidate_text_cache()  # the optimizer wrote through .data views of the bucket
        state["clean"] += 1
        if state["clean"] >= LOSS_SCALE_GROWTH_INTERVAL and state["scale"] < LOSS_SCALE_INIT and hasattr(model, "set_loss_scale"):
            state["scale"], state["clean"] = state["scale"] * 2.0, 0
            model.set_loss_scale(state["scale"])
    else:
        if state["scale"] <= LOSS_SCALE_MIN or not hasattr(model, "set_loss_scale"):
            raise FloatingPointError("Gradients are infinite or NaN at the smallest loss scale!")
        state["scale"], state["clean"], state["skipped"] = state["scale"] * 0.5, 0, state["skipped"] + 1
        model.set_loss_scale(state["scale"])
        print(f"Gradient overflow: step skipped, loss scale halved to {state['scale']:g}")
    loss_summary = {"loss": global_loss}
    if (trainer.batch_idx + 1) == trainer.num_batches:
        trainer.update_lr()
    return loss_summary


def install_loader(trainer, local: int):
    """Rank-aware, prefetched training loader (called at the end of build_model).  world > 1: ``train_loader_x`` is rebuilt over the same
    dataset with its batch sampler wrapped in parallel.ShardedBatchSampler, so each rank loads and decodes 1/world of every global
    batch (the reference's nn.DataParallel scatters ONE loaded batch, trainers/mudpt.py:230-233; here nothing is loaded N times);
    loaders that cannot be rebuilt (list-like synthetic ones) keep the slice-after-load fallback (parallel.shard_batch).  Then the
    DevicePrefetcher overlaps the next batch's host -> device copy with the current step."""
    from .prefetch import DevicePrefetcher
    loader = getattr(trainer, "train_loader_x", None)
    trainer._loader_sharded = False
    if loader is None or isinstance(loader, DevicePrefetcher):
        return
    if parallel.world_size() > 1:
        sharded = parallel.shard_loader(loader)  # unchanged if the loader is rank-aware already (MUDPT_DATA_SHARDED=1, DistributedSampler)
        if sharded is not None:
            loader, trainer._loader_sharded = sharded, True
    trainer.train_loader_x = DevicePrefetcher(loader, device=f"cuda:{local}", shard=not trainer._loader_sharded)


def parse_batch(trainer, batch):
    """parse_batch_train of both plugins (trainers/mudpt.py:263-268, trainers/cocoop.py:278-283).  N > 1: this rank's share of the global
    batch -- already the whole batch when the loader is rank-aware or the batch came through the DevicePrefetcher (sliced, on the
    device: the two .to(device) are then no-ops), else the contiguous slice nn.DataParallel's scatter would give this GPU, taken on the
    host so that only 1/world of the images crosses PCIe."""
    input, label = batch["img"], batch["label"]
    if not batch.get("_mudpt_sharded", False) and not getattr(trainer, "_loader_sharded", False):
        input, label = parallel.shard_batch(input, label)
    return input.to(trainer.device), label.to(trainer.device)


def save_on_main(trainer, save, *args, **kwargs):
    """Replicas are identical: rank 0 alone writes OUTPUT_DIR; the others wait until the file is complete, so that a load_model that
    follows (Dassl's after_train with TEST.FINAL_MODEL = best_val, a resume) never reads a missing or half-written checkpoint.  The wait
    is an exchange of rank 0's success flag, not a bare barrier: if the save raised, EVERY rank raises (the others would otherwise walk
    on into the next collective, or into reading the missing file, while rank 0 unwinds)."""
    err = None
    if parallel.is_main():
        try:
            save(*args, **kwargs)
        except Exception as e:  # noqa: BLE001 -- re-raised below, after the other ranks have been told
            err = e
    if not parallel.all_ok(err is None):
        if err is not None:
            raise err
        raise RuntimeError("rank 0 failed to write the checkpoint (its traceback is on rank 0's stderr)")


def load_plugin_checkpoint(trainer, directory, epoch, drop_keys, skipped_note):
    """load_model of both plugins (trainers/mudpt.py:270-302, trainers/cocoop.py:285-307): {directory}/{name}/model.pth.tar-{epoch} or
    model-best.pth.tar, keys state_dict / epoch, the fixed token buffers dropped, strict=False (the frozen backbone entries of a
    reference checkpoint are ignored)."""
    if not directory:
        print(skipped_note)
        return
    names = trainer.get_model_names()
    model_file = "model-best.pth.tar"  # by default, the best model is loaded
    if epoch is not None:
        model_file = "model.pth.tar-" + str(epoch)
    for name in names:
        model_path = osp.join(directory, name, model_file)
        if not osp.exists(model_path):
            raise FileNotFoundError('Model not found at "{}"'.format(model_path))
        checkpoint = load_checkpoint(model_path)
        state_dict = checkpoint["state_dict"]
        epoch = checkpoint["epoch"]
        for k in drop_keys:  # ignore fixed token vectors
            state_dict.pop(k, None)
        print("Loading weights to {} " 'from "{}" (epoch = {})'.format(name, model_path, epoch))
        trainer._models[name].load_state_dict(state_dict, strict=False)


def class_parallel_shard(n_cls: int, setting=None):
    """This rank's class range for the class-parallel text tower, or None (every rank encodes all classes, as the reference does).
    ``setting``: cfg.TRAINER.MUDPT.CLASS_PARALLEL if the config defines it, else the environment variable MUDPT_CLASS_PARALLEL:
    "1" / True = on, "0" / False = off, unset / "auto" = on when there is more than one rank and at least 256 classes (ImageNet: the
    replicated text tower is 39 % of the step's FLOPs; at 11 classes it hides behind the vision tower and the two extra
    exchanges would only cost).  Every rank must then run the same sequence of forward calls (training steps, test batches)."""
    world = parallel.world_size()
    if setting is None:
        setting = os.environ.get("MUDPT_CLASS_PARALLEL", "auto")
    if isinstance(setting, str):
        setting = {"1": True, "true": True, "on": True, "0": False, "false": False, "off": False}.get(setting.lower(), "auto")
    on = (world > 1 and n_cls >= 256) if setting == "auto" else bool(setting)
    if not on or world <= 1 or world > n_cls:
        return None
    return parallel.class_range(n_cls)


def precision_to_dtype(prec: str) -> str:
    """TRAINER.*.PREC -> library mode.  The reference's model is fp32 on its CPU path whatever PREC says (clip/clip.py:142-143 floats it;
    trainers/mudpt.py:199-200).  "fp32" selects the library's parity mode (include/mudpt.h MUDPT_F32: split forward GEMM operands -- fp16
    pairs + fp32 attention in the text tower, fp16 + e4m3 remainders on the fp8 matrix pipe in the vision tower): logits within 3.5e-4 of the
    reference at the logit scale pretrained checkpoints carry (100), at 1.27x the bf16 step.  "fp16" is the fast mode that holds 1e-3 at the
    INIT logit scale 14.29 only (4e-3 at 100: see warn_if_fp16_misses_the_bound), "amp" the bf16 throughput mode."""
    return PREC_TO_DTYPE[prec]


_warned_fp16_scale = False


def warn_if_fp16_misses_the_bound(prec: str, state) -> bool:
    """PREC = "fp16" (the reference yamls' default) on a REAL checkpoint: every released CLIP carries exp(logit_scale) = 100
    (clip/model.py:777 initialises ln(1 / 0.07) = 14.29, training drives it to the clamp at 100), and at that scale the fp16 mode's logits
    are ~4e-3 from the reference's fp32 CPU path (trainers/mudpt.py:178-182) -- outside north_star's 1e-3.  Say so once, and which setting
    holds the bound.  Returns whether it warned."""
    global _warned_fp16_scale
    if prec != "fp16" or state is None or "logit_scale" not in state:
        return False
    scale = float(torch.as_tensor(state["logit_scale"]).float().exp())
    if abs(scale - 100.0) >= 1.0:
        return False
    if not _warned_fp16_scale:
        print(f'NOTE: PREC "fp16" with exp(logit_scale) = {scale:.1f}: this mode rounds GEMM operands to fp16 and is ~4e-3 from the reference\'s '
              'fp32 logits at this scale (1e-3 is only held at the init scale 14.29).  PREC "fp32" selects the parity mode (<= 3.5e-4 at '
              'scale 100, ~1.2x the step time of "fp16"); "amp" is the bf16 throughput mode.')
        _warned_fp16_scale = True
    return True


@TRAINER_REGISTRY.register()
class MuDPT(TrainerX):
    def check_cfg(self, cfg):
        assert cfg.TRAINER.MUDPT.PREC in ["fp16", "fp32", "amp"]  # trainers/mudpt.py:190

    def build_model(self):
        cfg = self.cfg
        classnames = self.dm.dataset.classnames
        mc = cfg.TRAINER.MUDPT
        assert mc.DEEP_PROMPT_DEPTH > 0, "PROMPT_DEPTH should be > 0"  # trainers/mudpt.py:52

        print(f"Loading CLIP (backbone: {cfg.MODEL.BACKBONE.NAME})")
        state = load_clip_state_dict(cfg)
        if state is None:
            shape = ModelShape(n_ctx=mc.N_CTX, depth=mc.DEEP_PROMPT_DEPTH)
            state = synth.random_clip_state(shape, cfg.MODEL.BACKBONE.SYNTHETIC_SEED)
        else:
            shape = ModelShape.from_state_dict(state, mc.N_CTX, mc.DEEP_PROMPT_DEPTH)
        cfg_imsize = cfg.INPUT.SIZE[0]
        assert cfg_imsize == shape.image_size, f"cfg_imsize ({cfg_imsize}) must equal to clip_imsize ({shape.image_size})"  # :55
        warn_if_fp16_misses_the_bound(mc.PREC, state)

        # trainers/mudpt.py:57-70,83-85: ctx init words, prompt prefix, "<prefix> <classname>." prompts
        ctx_init = mc.CTX_INIT
        if ctx_init:
            ctx_init = ctx_init.replace("_", " ")
            prompt_prefix = " ".join(ctx_init.split()[:mc.N_CTX])
            ctx_ids = [int(v) for v in tokenize_prompts([ctx_init], shape.ctx_len, near=cfg.MODEL.BACKBONE.PATH or None)[0, 1:1 + mc.N_CTX]] \
                if ctx_init != "a photo of a" else synth.CTX_INIT_TOKENS[:mc.N_CTX]
        else:
            print("Initializing A Generic Context")
            prompt_prefix, ctx_ids = " ".join(["X"] * mc.N_CTX), None
        print(f'Initial context: "{prompt_prefix}"')
        print(f"Number of context words (tokens): {mc.N_CTX}")
        print(f"Depth of deep prompt: {mc.DEEP_PROMPT_DEPTH}")
        prompts = [prompt_prefix + " " + name.replace("_", " ") + "." for name in classnames]
        tokenized = tokenize_prompts(prompts, shape.ctx_len, near=cfg.MODEL.BACKBONE.PATH or None)

        print("Building custom CLIP")
        # one process per GPU (the reference: nn.DataParallel in one process): join the process group torch.distributed.run set up
        # BEFORE the model exists, so grad_scale = 1 / world and the parameter broadcast below are in effect from step one
        rank, world, local = parallel.init()
        max_batch = max(-(-cfg.DATALOADER.TRAIN_X.BATCH_SIZE // world), cfg.DATALOADER.TEST.BATCH_SIZE)
        self.model = CustomCLIP(shape, state, tokenized, ctx_token_ids=ctx_ids, max_batch=max_batch,
                                dtype=precision_to_dtype(mc.PREC), device=f"cuda:{local}", seed=cfg.SEED,
                                class_shard=class_parallel_shard(len(classnames), getattr(mc, "CLASS_PARALLEL", None)))
        if self.model.class_shard is not None:
            print(f"Class-parallel text tower: this rank encodes classes {self.model.class_shard[0]}..{self.model.class_shard[1] - 1} of {len(classnames)}")
        # the freeze rule of trainers/mudpt.py:205-218 is structural here: the module only owns the 10 trainables
        print(f"Parameters to be updated: {set(self.model.param_names)}")
        if cfg.MODEL.INIT_WEIGHTS:
            # trainers/mudpt.py:220-221 passes self.model.prompt_learner, an attribute that does not exist; the
            # evident intent -- initialise the trainables from a checkpoint -- is applied to the module that owns them
            load_pretrained_weights(self.model, cfg.MODEL.INIT_WEIGHTS)

        self.optim = build_optimizer(self.model, cfg.OPTIM)
        self.sched = build_lr_scheduler(self.optim, cfg.OPTIM)
        self.register_model("MultimodalDeepPromptTuning", self.model, self.optim, self.sched)
        self.scaler = None  # loss scaling lives inside the library (mudpt_set_loss_scale)
        # the reference wraps in nn.DataParallel when device_count > 1 (:230-233); here: one process per GPU
        if parallel.world_size() > 1:
            parallel.broadcast_params(self.model.flat_params)
        install_loader(self, local)  # rank-aware (world > 1) and prefetched training loader

    def forward_backward(self, batch):
        return data_parallel_step(self, batch)

    def parse_batch_train(self, batch):
        return parse_batch(self, batch)

    def save_model(self, *args, **kwargs):
        save_on_main(self, super().save_model, *args, **kwargs)

    def load_model(self, directory, epoch=None):
        load_plugin_checkpoint(self, directory, epoch, ("mudpt_prompt_learner.token_prefix", "mudpt_prompt_learner.token_suffix"),  # trainers/mudpt.py:293-298
                               "Note that load_model() is skipped as no pretrained model is given")
