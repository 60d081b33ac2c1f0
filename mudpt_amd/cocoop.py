"""Drop-in ``CoCoOp`` trainer plugin: the reference's ``trainers/cocoop.py:201-307`` surface over libmudpt_hip.so
(BASELINE configs[3]).

Same class name, registry name, hooks and error behaviour: ``check_cfg`` (:203), ``build_model`` (:206), ``forward_backward``
(:246), ``parse_batch_train`` (:278), inherited ``model_inference`` (``self.model(input)`` in eval mode returns logits, :198) and
``load_model`` (:285).  As in the reference only the ``prompt_learner`` sub-module is given to the optimizer and registered
(:237-240), so checkpoints hold ``ctx`` and ``meta_net.*`` under the same keys.  The per-image text-encoder loop of
``CustomCLIP.forward`` (:187-194) is one batched pass over all (image, class) prompts inside the library.
"""
from __future__ import annotations

from . import parallel, synth
from .model import CustomCLIP, ModelShape
from .trainer import (TRAINER_REGISTRY, TrainerX, build_lr_scheduler, build_optimizer, data_parallel_step, install_loader, load_clip_state_dict,
                      load_plugin_checkpoint, load_pretrained_weights, parse_batch, precision_to_dtype, save_on_main, tokenize_prompts, warn_if_fp16_misses_the_bound)


@TRAINER_REGISTRY.register()
class CoCoOp(TrainerX):
    def check_cfg(self, cfg):
        assert cfg.TRAINER.COCOOP.PREC in ["fp16", "fp32", "amp"]  # trainers/cocoop.py:204

    def build_model(self):
        cfg = self.cfg
        classnames = self.dm.dataset.classnames
        cc = cfg.TRAINER.COCOOP
        print(f"Loading CLIP (backbone: {cfg.MODEL.BACKBONE.NAME})")
        state = load_clip_state_dict(cfg)
        n_ctx = cc.N_CTX
        ctx_init = cc.CTX_INIT
        near = cfg.MODEL.BACKBONE.PATH or None
        if ctx_init:  # trainers/cocoop.py:79-87: n_ctx follows the init words
            ctx_init = ctx_init.replace("_", " ")
            n_ctx = len(ctx_init.split(" "))
        if state is None:
            shape = ModelShape(n_ctx=n_ctx, depth=1)
            state = synth.random_clip_state(shape, cfg.MODEL.BACKBONE.SYNTHETIC_SEED)
        else:
            shape = ModelShape.from_state_dict(state, n_ctx, 1)
        cfg_imsize = cfg.INPUT.SIZE[0]
        assert cfg_imsize == shape.image_size, f"cfg_imsize ({cfg_imsize}) must equal to clip_imsize ({shape.image_size})"  # :77
        warn_if_fp16_misses_the_bound(cc.PREC, state)
        if ctx_init:
            ctx_ids = [int(v) for v in tokenize_prompts([ctx_init], shape.ctx_len, near=near)[0, 1:1 + n_ctx]] \
                if ctx_init != "a photo of a" else synth.CTX_INIT_TOKENS[:n_ctx]
            prompt_prefix = ctx_init
        else:
            ctx_ids, prompt_prefix = None, " ".join(["X"] * n_ctx)  # random N(0, 0.02^2) context (:90-92)
        print(f'Initial context: "{prompt_prefix}"')
        print(f"Number of context words (tokens): {n_ctx}")
        prompts = [prompt_prefix + " " + name.replace("_", " ") + "." for name in classnames]  # :110-112
        tokenized = tokenize_prompts(prompts, shape.ctx_len, near=near)

        print("Building custom CLIP")
        # one process per GPU (the reference: nn.DataParallel in one process): join the process group torch.distributed.run set up
        # BEFORE the model exists, so grad_scale = 1 / world and the parameter broadcast below are in effect from step one
        rank, world, local = parallel.init()
        max_batch = max(-(-cfg.DATALOADER.TRAIN_X.BATCH_SIZE // world), cfg.DATALOADER.TEST.BATCH_SIZE)
        self.model = CustomCLIP(shape, state, tokenized, ctx_token_ids=ctx_ids, max_batch=max_batch,
                                dtype=precision_to_dtype(cc.PREC), device=f"cuda:{local}", seed=cfg.SEED, variant="cocoop")
        print("Turning off gradients in both the image and the text encoder")  # structural: the module owns the 5 trainables only
        print(f"Parameters to be updated: {set(self.model.param_names)}")
        if cfg.MODEL.INIT_WEIGHTS:  # :234-235
            load_pretrained_weights(self.model.prompt_learner, cfg.MODEL.INIT_WEIGHTS)
        # NOTE: only give prompt_learner to the optimizer (:237)
        self.optim = build_optimizer(self.model.prompt_learner, cfg.OPTIM)
        self.sched = build_lr_scheduler(self.optim, cfg.OPTIM)
        self.register_model("prompt_learner", self.model.prompt_learner, self.optim, self.sched)
        self.scaler = None  # loss scaling lives inside the library
        if parallel.world_size() > 1:  # the reference's nn.DataParallel (:244-247) becomes one process per GPU
            parallel.broadcast_params(self.model.flat_params)
        install_loader(self, local)  # rank-aware (world > 1) and prefetched training loader

    def forward_backward(self, batch):
        # loss = model(image, label) (cross-entropy inside forward, :196-197) + backward in one library call
        return data_parallel_step(self, batch)

    def parse_batch_train(self, batch):
        return parse_batch(self, batch)

    def save_model(self, *args, **kwargs):
        save_on_main(self, super().save_model, *args, **kwargs)

    def load_model(self, directory, epoch=None):
        load_plugin_checkpoint(self, directory, epoch, ("token_prefix", "token_suffix"),  # trainers/cocoop.py:303-307
                               "Note that load_model() is skipped as no Pretrained model is given")
