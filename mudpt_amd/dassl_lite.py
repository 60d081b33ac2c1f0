"""Minimal stand-in for the parts of Dassl.pytorch the MuDPT plugin touches, used ONLY when Dassl is not installed
(it is an un-vendored, un-pinned dependency of the reference: train.py:6-9, trainers/mudpt.py:10-13).

It mirrors the hook protocol the plugin relies on -- ``check_cfg -> build_data_loader -> build_model ->
[run_epoch: forward_backward per batch] -> test: model_inference`` -- with a synthetic data manager, so the
drop-in trainer can be exercised and benchmarked on a box without datasets.  With Dassl installed the plugin
subclasses the real ``dassl.engine.TrainerX`` and none of this is imported.
"""
from __future__ import annotations

import math
import os
import time
from typing import Dict, List

import torch


class CfgNode(dict):
    """Attribute-style nested config (the subset of yacs.CfgNode behaviour the plugin reads)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


def default_cfg() -> CfgNode:
    """Defaults of the keys the plugin reads: train.py:68-119 extend_cfg + the shipped MuDPT yaml's OPTIM block."""
    C = CfgNode
    return C(
        SEED=1, OUTPUT_DIR="output/mudpt_amd", USE_CUDA=True,
        INPUT=C(SIZE=(224, 224)),
        MODEL=C(BACKBONE=C(NAME="ViT-B/16", PATH="", SYNTHETIC_SEED=0), INIT_WEIGHTS=""),
        DATASET=C(NAME="Synthetic", NUM_CLASSES=11, NUM_TRAIN=64, NUM_TEST=32),
        DATALOADER=C(TRAIN_X=C(BATCH_SIZE=4), TEST=C(BATCH_SIZE=100)),
        OPTIM=C(NAME="sgd", LR=0.0025, MAX_EPOCH=10, LR_SCHEDULER="cosine", WARMUP_EPOCH=1, WARMUP_TYPE="constant",
                WARMUP_CONS_LR=1e-5, MOMENTUM=0.9, WEIGHT_DECAY=5e-4, SGD_DAMPNING=0.0, SGD_NESTEROV=False),
        TRAIN=C(PRINT_FREQ=5),
        TRAINER=C(NAME="MuDPT", MUDPT=C(N_CTX=2, CTX_INIT="a photo of a", DEEP_PROMPT_DEPTH=8, PREC="fp16"),
                  COCOOP=C(N_CTX=4, CTX_INIT="a photo of a", PREC="fp16")),  # train.py:92-95 + configs/trainers/CoCoOp/*.yaml
    )


class _Registry:
    def __init__(self):
        self._obj: Dict[str, type] = {}

    def register(self):
        def deco(cls):
            self._obj[cls.__name__] = cls
            return cls
        return deco

    def get(self, name):
        return self._obj[name]

    def registered_names(self) -> List[str]:
        return list(self._obj)


TRAINER_REGISTRY = _Registry()


def build_trainer(cfg):
    return TRAINER_REGISTRY.get(cfg.TRAINER.NAME)(cfg)


def build_optimizer(model, optim_cfg):
    """Dassl's build_optimizer for NAME == "sgd": torch SGD over the module's parameters."""
    assert optim_cfg.NAME == "sgd", "dassl_lite only provides SGD"
    return torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=optim_cfg.LR, momentum=optim_cfg.MOMENTUM,
                           weight_decay=optim_cfg.WEIGHT_DECAY, dampening=optim_cfg.SGD_DAMPNING, nesterov=optim_cfg.SGD_NESTEROV)


class _ConstantWarmupCosine(torch.optim.lr_scheduler._LRScheduler):
    """Cosine annealing over MAX_EPOCH preceded by WARMUP_EPOCH epochs at WARMUP_CONS_LR (Dassl's ConstantWarmupScheduler)."""

    def __init__(self, optim, max_epoch, warmup_epoch, cons_lr):
        self.max_epoch, self.warmup_epoch, self.cons_lr = max_epoch, warmup_epoch, cons_lr
        super().__init__(optim)

    def get_lr(self):
        e = self.last_epoch
        if e < self.warmup_epoch:
            return [self.cons_lr for _ in self.base_lrs]
        return [0.5 * b * (1 + math.cos(math.pi * e / self.max_epoch)) for b in self.base_lrs]


def build_lr_scheduler(optim, optim_cfg):
    return _ConstantWarmupCosine(optim, optim_cfg.MAX_EPOCH, optim_cfg.WARMUP_EPOCH if optim_cfg.WARMUP_TYPE == "constant" else 0,
                                 optim_cfg.WARMUP_CONS_LR)


def load_checkpoint(path):
    if not os.path.exists(path):
        raise FileNotFoundError('File is not found at "{}"'.format(path))
    return torch.load(path, map_location="cpu", weights_only=True)


def load_pretrained_weights(model, weight_path):
    """Dassl's load_pretrained_weights: copy the checkpoint entries whose name and shape match, report the rest."""
    ckpt = load_checkpoint(weight_path)
    state = ckpt["state_dict"] if isinstance(ckpt, dict) and "state_dict" in ckpt else ckpt
    own = model.state_dict()
    matched, discarded = [], []
    for k, v in state.items():
        k = k[7:] if k.startswith("module.") else k
        if k in own and own[k].shape == v.shape:
            own[k].copy_(v)
            matched.append(k)
        else:
            discarded.append(k)
    if not matched:
        print(f'Cannot load {weight_path} (check the key names manually)')
    else:
        print(f"Successfully loaded pretrained weights from {weight_path}")
        if discarded:
            print(f"Layers discarded due to unmatched keys or size: {discarded}")


class _SyntheticDataset:
    def __init__(self, classnames):
        self.classnames = classnames
        self.lab2cname = dict(enumerate(classnames))
        self.num_classes = len(classnames)


class SyntheticDataManager:
    """Seeded N(0,1) "CLIP-normalised" images + uniform labels, shaped like Dassl's DataManager batches."""

    def __init__(self, cfg, classnames):
        self.dataset = _SyntheticDataset(classnames)
        g = torch.Generator().manual_seed(cfg.SEED)
        S = cfg.INPUT.SIZE[0]

        def make(n, bs):
            x, y = torch.randn(n, 3, S, S, generator=g), torch.randint(0, len(classnames), (n,), generator=g)
            return [{"img": x[i:i + bs], "label": y[i:i + bs]} for i in range(0, n - bs + 1, bs)] or [{"img": x, "label": y}]
        self.train_loader_x = make(cfg.DATASET.NUM_TRAIN, cfg.DATALOADER.TRAIN_X.BATCH_SIZE)
        self.test_loader = make(cfg.DATASET.NUM_TEST, min(cfg.DATALOADER.TEST.BATCH_SIZE, cfg.DATASET.NUM_TEST))
        self.num_classes = len(classnames)


class TrainerX:
    """The slice of dassl.engine.TrainerX / SimpleTrainer / TrainerBase the plugin uses."""

    def __init__(self, cfg):
        self._models, self._optims, self._scheds = {}, {}, {}
        self.check_cfg(cfg)
        self.cfg = cfg
        self.device = torch.device("cuda" if torch.cuda.is_available() and cfg.USE_CUDA else "cpu")
        self.start_epoch = self.epoch = 0
        self.max_epoch = cfg.OPTIM.MAX_EPOCH
        self.output_dir = cfg.OUTPUT_DIR
        self.build_data_loader()
        self.build_model()

    # -- hooks subclasses override ---------------------------------------------------------------------------
    def check_cfg(self, cfg):
        pass

    def build_data_loader(self):
        from .synth import BENCH_CLASSNAMES
        n = self.cfg.DATASET.NUM_CLASSES
        names = BENCH_CLASSNAMES[:n] if n <= len(BENCH_CLASSNAMES) else [f"class{i}" for i in range(n)]
        self.dm = SyntheticDataManager(self.cfg, names)
        self.train_loader_x, self.test_loader = self.dm.train_loader_x, self.dm.test_loader
        self.num_classes, self.lab2cname = self.dm.num_classes, self.dm.dataset.lab2cname

    def build_model(self):
        raise NotImplementedError

    def forward_backward(self, batch):
        raise NotImplementedError

    # -- TrainerBase services ---------------------------------------------------------------------------------------
    def register_model(self, name="model", model=None, optim=None, sched=None):
        self._models[name], self._optims[name], self._scheds[name] = model, optim, sched

    def get_model_names(self, names=None):
        return list(self._models) if names is None else ([names] if isinstance(names, str) else list(names))

    def set_model_mode(self, mode="train", names=None):
        for n in self.get_model_names(names):
            self._models[n].train(mode == "train")

    def update_lr(self, names=None):
        for n in self.get_model_names(names):
            if self._scheds[n] is not None:
                self._scheds[n].step()

    def get_current_lr(self, names=None):
        return self._optims[self.get_model_names(names)[0]].param_groups[0]["lr"]

    def detect_anomaly(self, loss):
        if not torch.isfinite(loss).all():
            raise FloatingPointError("Loss is infinite or NaN!")

    def save_model(self, epoch, directory, is_best=False, model_name=""):
        for n in self.get_model_names():
            d = os.path.join(directory, n)
            os.makedirs(d, exist_ok=True)
            ckpt = {"state_dict": self._models[n].state_dict(), "epoch": epoch + 1,
                    "optimizer": self._optims[n].state_dict() if self._optims[n] is not None else None,
                    "scheduler": self._scheds[n].state_dict() if self._scheds[n] is not None else None}
            torch.save(ckpt, os.path.join(d, model_name or f"model.pth.tar-{epoch + 1}"))
            if is_best:
                torch.save(ckpt, os.path.join(d, "model-best.pth.tar"))

    # -- loops -----------------------------------------------------------------------------------------------------------
    def train(self):
        for self.epoch in range(self.start_epoch, self.max_epoch):
            self.run_epoch()
        self.save_model(self.epoch, self.output_dir)
        return self.test()

    def run_epoch(self):
        self.set_model_mode("train")
        self.num_batches = len(self.train_loader_x)
        t0 = time.time()
        for self.batch_idx, batch in enumerate(self.train_loader_x):
            summary = self.forward_backward(batch)
            if (self.batch_idx + 1) % self.cfg.TRAIN.PRINT_FREQ == 0 or self.num_batches < self.cfg.TRAIN.PRINT_FREQ:
                print(f"epoch [{self.epoch + 1}/{self.max_epoch}] batch [{self.batch_idx + 1}/{self.num_batches}] "
                      f"time {time.time() - t0:.3f} loss {summary['loss']:.4f} lr {self.get_current_lr():.4e}")

    def parse_batch_test(self, batch):
        return batch["img"].to(self.device), batch["label"].to(self.device)

    def model_inference(self, input):
        return self.model(input)

    @torch.no_grad()
    def test(self, split=None):
        self.set_model_mode("eval")
        correct = total = 0
        for batch in self.test_loader:
            x, y = self.parse_batch_test(batch)
            pred = self.model_inference(x).argmax(dim=1)
            correct += int((pred == y).sum())
            total += int(y.numel())
        acc = 100.0 * correct / max(total, 1)
        print(f"=> result\n* total: {total}\n* correct: {correct}\n* accuracy: {acc:.1f}%")
        return acc
