"""Worker of tests/test_classparallel_gpu.py::test_plugin_runs_class_parallel_at_world_2: one rank of the MuDPT trainer PLUGIN under
torch.distributed.run (gloo, every rank on cuda:0 of the one-GPU test box), with the class-parallel text tower switched on through the
environment (MUDPT_CLASS_PARALLEL=1; automatic only from 256 classes).  Two training steps on the loader's first batches, a test pass,
then the trainables are written per rank."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out = sys.argv[1]
    os.environ["MUDPT_BENCH_ONE_DEVICE"] = "1"
    os.environ["LOCAL_RANK"] = "0"  # every rank on the box's one GPU
    import torch.distributed as dist
    dist.init_process_group("gloo")  # before build_model, so parallel.init() finds the group (it would pick RCCL: one GPU cannot host two ranks)
    from mudpt_amd import dassl_lite, parallel, trainer  # noqa: F401  (importing trainer registers the plugin, as train.py:31-40 does)
    cfg = dassl_lite.default_cfg()
    cfg.OUTPUT_DIR = out + ".dir"
    cfg.OPTIM.MAX_EPOCH, cfg.OPTIM.WARMUP_EPOCH, cfg.OPTIM.LR = 1, 0, 0.02
    cfg.DATASET.NUM_TRAIN, cfg.DATASET.NUM_TEST = 8, 8
    cfg.DATALOADER.TEST.BATCH_SIZE = 8
    cfg.TRAINER.MUDPT.N_CTX, cfg.TRAINER.MUDPT.DEEP_PROMPT_DEPTH = 4, 12
    t = dassl_lite.build_trainer(cfg)
    rank, world = parallel.rank(), parallel.world_size()
    assert world == 2 and (t.model.class_shard is not None) == (os.environ.get("MUDPT_CLASS_PARALLEL") == "1"), (world, t.model.class_shard)
    t.batch_idx, t.num_batches = 0, 99
    losses = [t.forward_backward(t.train_loader_x[i % len(t.train_loader_x)])["loss"] for i in range(2)]
    acc = t.test()
    torch.cuda.synchronize()
    torch.save({"params": t.model.flat_params.detach().cpu().clone(), "losses": losses, "acc": acc, "shard": t.model.class_shard}, f"{out}.r{rank}")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
