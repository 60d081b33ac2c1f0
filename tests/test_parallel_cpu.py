"""N > 1 data-parallel path on CPU: 2 ranks, gloo.  The HIP path cannot run here, so the per-rank gradient comes from
the CPU oracle; what is under test is mudpt_amd.parallel (the bucket all-reduce, the 1/world convention, sharding)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from oracle import mudpt_oracle as O
from tests.helpers import GoldenCase


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from mudpt_amd import parallel
    r, w, _ = parallel.init("gloo")
    assert (r, w) == (rank, world) and parallel.world_size() == world
    case = GoldenCase("mudpt_tiny")
    B = 2 * world
    g = torch.Generator().manual_seed(99)
    images = torch.randn(B, 3, case.cfg.image_size, case.cfg.image_size, generator=g)
    labels = torch.randint(0, 11, (B,), generator=g)
    idx = list(parallel.shard(B, rank, world))
    params = {k: v.clone() for k, v in case.params.items()}
    if rank != 0:  # replicas must end up with rank 0's parameters
        params = {k: v + 1.0 for k, v in params.items()}
    flat_p = parallel.broadcast_params(O.flatten(params))
    params = O.unflatten(flat_p, case.cfg)
    _, _, grads = O.forward_backward(case.cfg, case.frozen, params, case.class_embedding, case.eot, images[idx], labels[idx])
    flat = O.flatten(grads) * parallel.grad_scale()  # what mudpt_forward_backward(grad_scale = 1 / world) writes
    parallel.allreduce_grads(flat)
    new_p, _ = O.sgd_step(flat_p, flat, None, 0.0025)
    if rank == 0:
        _, _, ref = O.forward_backward(case.cfg, case.frozen, case.params, case.class_embedding, case.eot, images, labels)
        torch.save({"got": flat, "ref": O.flatten(ref), "params": new_p}, out)
    else:
        torch.save({"params": new_p}, out + ".r1")
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gradient_equals_global_batch(tmp_path):
    out = str(tmp_path / "dp.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r0, r1 = torch.load(out), torch.load(out + ".r1")
    # sum over ranks of (1/world) * local-mean gradient == gradient of the global-batch mean (trainers/mudpt.py:249-250)
    # (fp32 autograd on different batch splits: agreement to rounding, measured max 2.6e-6 on values up to ~1e-1)
    torch.testing.assert_close(r0["got"], r0["ref"], atol=2e-5, rtol=1e-3)
    assert torch.equal(r0["params"], r1["params"])  # replicas stay bitwise identical after the step


class _OracleModel(torch.nn.Module):
    """Stands in for mudpt_amd.model.CustomCLIP on CPU: same surface (flat buckets, named views, forward_backward writing
    grad_scale * gradient into the bucket), arithmetic by the oracle.  What is under test is the PLUGIN's step, not the library."""

    def __init__(self, case, params):
        super().__init__()
        self.case = case
        self.flat_params = O.flatten(params).clone()
        self.flat_grads = torch.zeros_like(self.flat_params)
        off = 0
        self.views = {}
        for k in O.TRAINABLE_ORDER:
            n = params[k].numel()
            p = torch.nn.Parameter(self.flat_params[off:off + n].view_as(params[k]))
            p.grad = self.flat_grads[off:off + n].view_as(params[k])
            self.register_parameter(k.replace(".", "__"), p)
            off += n
        self.calls = []

    def forward_backward(self, image, label, grad_scale=1.0):
        c = self.case
        self.calls.append((tuple(image.shape), float(grad_scale)))
        loss, _, grads = O.forward_backward(c.cfg, c.frozen, O.unflatten(self.flat_params.clone(), c.cfg), c.class_embedding, c.eot, image, label)
        self.flat_grads.copy_(O.flatten(grads) * grad_scale)
        return loss.detach()

    def invalidate_text_cache(self):
        pass


def _plugin_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from mudpt_amd import parallel, trainer
    parallel.init("gloo")
    case = GoldenCase("mudpt_tiny")
    g = torch.Generator().manual_seed(123)
    B = 4  # the GLOBAL batch every rank's (rank-unaware) loader yields; the plugin slices it
    batches = [{"img": torch.randn(B, 3, case.cfg.image_size, case.cfg.image_size, generator=g), "label": torch.randint(0, 11, (B,), generator=g)}
               for _ in range(2)]
    t = object.__new__(trainer.MuDPT)
    t.model = _OracleModel(case, case.params)
    t.optim = torch.optim.SGD(t.model.parameters(), lr=0.05, momentum=0.9, weight_decay=5e-4)
    t.device, t.batch_idx, t.num_batches = torch.device("cpu"), 0, 99
    losses = [t.forward_backward(b)["loss"] for b in batches]
    assert t.model.calls == [((B // world, 3, case.cfg.image_size, case.cfg.image_size), 1.0 / world)] * 2, t.model.calls
    # a non-finite loss on ONE rank must stop every rank (no hang in the next collective)
    bad = torch.tensor(float("nan") if rank == 1 else 1.0)
    assert parallel.all_finite(bad, t.model.flat_grads) is False
    torch.save({"params": t.model.flat_params.clone(), "losses": losses}, f"{out}.r{rank}")
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_plugin_step_is_data_parallel_at_world_2(tmp_path):
    """The trainer plugin's own step under torch.distributed (gloo, 2 ranks): the process group exists, each rank takes its slice of
    the loader's global batch (nn.DataParallel's scatter, trainers/mudpt.py:230-233), gradients are all-reduced BEFORE optim.step,
    replicas are bitwise identical after two momentum-SGD steps and equal the single-process run on the whole batches."""
    from mudpt_amd import trainer
    out = str(tmp_path / "plugin")
    mp.spawn(_plugin_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r0, r1 = torch.load(out + ".r0"), torch.load(out + ".r1")
    assert torch.equal(r0["params"], r1["params"])
    # single process, same two global batches
    case = GoldenCase("mudpt_tiny")
    g = torch.Generator().manual_seed(123)
    t = object.__new__(trainer.MuDPT)
    t.model = _OracleModel(case, case.params)
    t.optim = torch.optim.SGD(t.model.parameters(), lr=0.05, momentum=0.9, weight_decay=5e-4)
    t.device, t.batch_idx, t.num_batches = torch.device("cpu"), 0, 99
    for _ in range(2):
        t.forward_backward({"img": torch.randn(4, 3, case.cfg.image_size, case.cfg.image_size, generator=g), "label": torch.randint(0, 11, (4,), generator=g)})
    assert t.model.calls[0][1] == 1.0
    torch.testing.assert_close(r0["params"], t.model.flat_params, atol=2e-6, rtol=1e-5)
    moved = (t.model.flat_params - O.flatten(case.params)).abs().max().item()
    assert moved > 1e-4  # the steps did something


def test_shard_batch_requires_divisible_global_batch(monkeypatch):
    from mudpt_amd import parallel
    x, y = torch.zeros(5, 3, 2, 2), torch.zeros(5, dtype=torch.long)
    assert parallel.shard_batch(x, y)[0] is x  # single process: untouched
    monkeypatch.setattr(parallel, "world_size", lambda: 2)
    monkeypatch.setattr(parallel, "rank", lambda: 1)
    with pytest.raises(ValueError, match="not divisible"):
        parallel.shard_batch(x, y)
    a, b = parallel.shard_batch(x[:4], y[:4])
    assert a.shape[0] == 2 and a.data_ptr() == x[2:4].data_ptr()
    monkeypatch.setenv("MUDPT_DATA_SHARDED", "1")   # a rank-aware loader opts out
    assert parallel.shard_batch(x, y)[0] is x


def test_single_process_is_a_noop():
    from mudpt_amd import parallel
    t = torch.arange(5.0)
    assert parallel.world_size() == 1 and parallel.grad_scale() == 1.0
    assert torch.equal(parallel.allreduce_grads(t.clone()), t) and torch.equal(parallel.broadcast_params(t.clone()), t)
    assert list(parallel.shard(10, 1, 4)) == [2, 3] and list(parallel.shard(8, 3, 4)) == [6, 7]


def test_class_range_partitions_the_classes():
    """Class-parallel text tower: contiguous, balanced, disjoint, covering; refuses more ranks than classes."""
    from mudpt_amd import parallel
    for n_cls, world in [(11, 2), (1000, 8), (208, 3), (8, 8)]:
        r = [parallel.class_range(n_cls, k, world) for k in range(world)]
        assert r[0][0] == 0 and r[-1][1] == n_cls
        assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
        sizes = [b - a for a, b in r]
        assert max(sizes) - min(sizes) <= 1 and min(sizes) >= 1
    with pytest.raises(AssertionError):
        parallel.class_range(3, 0, 4)


def test_class_parallel_setting(monkeypatch):
    """The plugin's switch: off for one process whatever the setting; "auto" needs >= 256 classes."""
    from mudpt_amd import trainer, parallel
    monkeypatch.setattr(parallel, "world_size", lambda: 1)
    assert trainer.class_parallel_shard(1000, True) is None
    monkeypatch.setattr(parallel, "world_size", lambda: 4)
    monkeypatch.setattr(parallel, "rank", lambda: 1)
    assert trainer.class_parallel_shard(1000, None) == (250, 500)
    assert trainer.class_parallel_shard(11, None) is None and trainer.class_parallel_shard(11, "1") == (3, 6)
    assert trainer.class_parallel_shard(1000, "off") is None
    monkeypatch.setenv("MUDPT_CLASS_PARALLEL", "0")
    assert trainer.class_parallel_shard(1000, None) is None
