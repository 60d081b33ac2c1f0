"""Dassl-free launcher for the MuDPT plugin on synthetic data (no datasets / checkpoints / network on the box).

    python -m mudpt_amd.harness --epochs 2 --batch 4 --n-ctx 4 --depth 12 [--prec fp16|amp] [--eval-only --model-dir D]

Mirrors what ``train.py`` (reference :153-173) does after config assembly: build_trainer(cfg) -> train() / test()."""
from __future__ import annotations

import argparse

import torch

from . import cocoop, dassl_lite, parallel, trainer  # noqa: F401  (importing trainer / cocoop registers MuDPT / CoCoOp)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--trainer", default="MuDPT", choices=["MuDPT", "CoCoOp"])
    ap.add_argument("--epochs", type=int, default=2)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--classes", type=int, default=11)
    ap.add_argument("--train-images", type=int, default=32)
    ap.add_argument("--n-ctx", type=int, default=4)
    ap.add_argument("--depth", type=int, default=12)
    ap.add_argument("--prec", default="fp16", choices=["fp16", "fp32", "amp"])
    ap.add_argument("--output-dir", default="output/mudpt_amd")
    ap.add_argument("--backbone-path", default="")
    ap.add_argument("--eval-only", action="store_true")
    ap.add_argument("--model-dir", default="")
    ap.add_argument("--load-epoch", type=int, default=None)
    a = ap.parse_args(argv)

    parallel.init()
    cfg = dassl_lite.default_cfg()
    cfg.OUTPUT_DIR = a.output_dir
    cfg.OPTIM.MAX_EPOCH = a.epochs
    cfg.DATALOADER.TRAIN_X.BATCH_SIZE = a.batch
    cfg.DATALOADER.TEST.BATCH_SIZE = max(a.batch, 8)
    cfg.DATASET.NUM_CLASSES, cfg.DATASET.NUM_TRAIN, cfg.DATASET.NUM_TEST = a.classes, a.train_images, 16
    cfg.MODEL.BACKBONE.PATH = a.backbone_path
    cfg.TRAINER.NAME = a.trainer
    cfg.TRAINER.MUDPT.N_CTX, cfg.TRAINER.MUDPT.DEEP_PROMPT_DEPTH, cfg.TRAINER.MUDPT.PREC = a.n_ctx, a.depth, a.prec
    cfg.TRAINER.COCOOP.PREC = a.prec
    torch.manual_seed(cfg.SEED)
    t = trainer.TRAINER_REGISTRY.get(a.trainer)(cfg) if not trainer.HAVE_DASSL else None
    if t is None:
        raise SystemExit("Dassl is installed: use the reference's train.py --trainer MuDPT / CoCoOp (see INTEGRATION.md)")
    if a.eval_only:
        t.load_model(a.model_dir, epoch=a.load_epoch)
        return t.test()
    return t.train()


if __name__ == "__main__":
    main()
