"""Worker of tests/test_classparallel_gpu.py: one rank of a class-parallel MuDPT step (launched by torch.distributed.run, gloo, every rank
on cuda:0 of the one-GPU test box).  Rank r takes its slice of the fixture's batch and its share of the class prompts; after the step and the
bucket all-reduce every rank holds the global-batch gradient, which rank 0 writes out together with its local logits and loss."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    case_name, dtype, out_path = sys.argv[1:4]
    from tests.helpers import GoldenCase
    from tests.test_manyclass_gpu import build
    from mudpt_amd import parallel
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    c = GoldenCase(case_name)
    B = len(c.labels)
    per = B // world
    img, lab = c.images[rank * per:(rank + 1) * per], c.labels[rank * per:(rank + 1) * per]
    from mudpt_amd.model import CustomCLIP, ModelShape
    cfg = c.cfg
    shape = ModelShape(cfg.image_size, cfg.patch, cfg.v_width, cfg.v_layers, cfg.v_heads, cfg.t_width, cfg.t_layers, cfg.t_heads,
                       cfg.ctx_len, cfg.embed_dim, cfg.n_ctx, cfg.depth)
    m = CustomCLIP(shape, c.frozen, c.tokens, max_batch=per, dtype=dtype, class_shard=parallel.class_range(len(c.tokens), rank, world))
    m.set_params(c.params)
    loss, logits = m.forward_backward(img, lab, grad_scale=1.0 / world, return_logits=True)
    parallel.allreduce_grads(m.flat_grads)
    torch.cuda.synchronize()
    # eval path: sharded forward (text tower recomputed, then reused)
    m.eval()
    ev1 = m(img).cpu()
    ev2 = m(img).cpu()
    gathered = [None] * world
    dist.all_gather_object(gathered, {"logits": logits.cpu(), "loss": float(loss.item()), "eval": ev1, "eval_reuse": ev2})
    if rank == 0:
        torch.save({"ranks": gathered, "grads": {k: g.detach().cpu().clone() for k, g in m.grads().items()}, "shard": m.class_shard}, out_path)
    m.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
