"""Images/s THROUGH THE PLUGIN (dassl_lite harness): host batches from the loader, parse_batch_train, forward_backward with the
torch optimizer step and loss.item() every step, as trainers/mudpt.py:235-261 -- with and without the DevicePrefetcher.

    python tools/plugin_bench.py [--batch 256] [--steps 12]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mudpt_amd import dassl_lite, trainer  # noqa: F401


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--steps", type=int, default=12)
    a = ap.parse_args()
    cfg = dassl_lite.default_cfg()
    cfg.OUTPUT_DIR = "/tmp/mudpt_plugin_bench"
    cfg.OPTIM.MAX_EPOCH, cfg.OPTIM.WARMUP_EPOCH = 1, 0
    cfg.DATASET.NUM_TRAIN, cfg.DATASET.NUM_TEST = a.batch * a.steps, 8
    cfg.DATALOADER.TRAIN_X.BATCH_SIZE, cfg.DATALOADER.TEST.BATCH_SIZE = a.batch, 8
    cfg.TRAINER.MUDPT.N_CTX, cfg.TRAINER.MUDPT.DEEP_PROMPT_DEPTH, cfg.TRAINER.MUDPT.PREC = 4, 12, "amp"
    t = dassl_lite.build_trainer(cfg)
    t.set_model_mode("train")
    t.num_batches = 10 ** 9
    for name, loader in (("prefetched (side-stream copy of the next batch)", t.train_loader_x), ("plain (.to(device) at the top of the step)", t.train_loader_x.loader)):
        for rep in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 0
            for t.batch_idx, batch in enumerate(loader):
                t.forward_backward(batch)
                n += a.batch
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        print(f"{name}: {dt / a.steps * 1e3:.2f} ms/step, {n / dt:.0f} images/s", flush=True)


if __name__ == "__main__":
    main()
