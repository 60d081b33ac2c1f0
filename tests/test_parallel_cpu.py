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
    # a non-finite loss on ONE rank must stop every rank (no hang in the next collective): the consensus of the step
    bad = torch.tensor(float("nan") if rank == 1 else 1.0)
    assert parallel.step_consensus(bad, t.model.flat_grads)[0] is False
    # rank 0's checkpoint write fails: EVERY rank raises (no rank walks on into the next collective or a missing file)
    def failing_save():
        raise OSError("disk full")
    with pytest.raises(OSError if rank == 0 else RuntimeError):
        trainer.save_on_main(t, failing_save)
    done = []
    trainer.save_on_main(t, lambda: done.append(1))
    assert done == ([1] if rank == 0 else [])
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


class _ToyDataset(torch.utils.data.Dataset):
    """Dassl's DatasetWrapper in miniature: item i -> {"img", "label", "index"}; `reads` records which items THIS process decoded."""

    def __init__(self, n, size):
        g = torch.Generator().manual_seed(7)
        self.x, self.y, self.reads = torch.randn(n, 3, size, size, generator=g), torch.randint(0, 11, (n,), generator=g), []

    def __len__(self):
        return len(self.x)

    def __getitem__(self, i):
        self.reads.append(int(i))
        return {"img": self.x[i], "label": self.y[i], "index": i}


def _toy_loader(ds, batch):
    torch.manual_seed(321)  # Dassl's set_random_seed(cfg.SEED): the same on every rank, so RandomSampler draws the same permutation
    return torch.utils.data.DataLoader(ds, batch_size=batch, shuffle=True, drop_last=True, num_workers=0)


def _loader_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from mudpt_amd import parallel, trainer
    parallel.init("gloo")
    case = GoldenCase("mudpt_tiny")
    ds = _ToyDataset(16, case.cfg.image_size)
    t = object.__new__(trainer.MuDPT)
    t.model = _OracleModel(case, case.params)
    t.optim = torch.optim.SGD(t.model.parameters(), lr=0.05, momentum=0.9, weight_decay=5e-4)
    t.device, t.batch_idx, t.num_batches = torch.device("cpu"), 0, 99
    t.train_loader_x = _toy_loader(ds, 4)
    trainer.install_loader(t, rank)  # what build_model does last
    assert t._loader_sharded and len(t.train_loader_x) == 4
    seen, losses = [], []
    for batch in t.train_loader_x:
        assert batch["img"].shape[0] == 4 // world  # this rank's share only
        seen.append(batch["index"].tolist())
        losses.append(t.forward_backward(batch)["loss"])
    assert sorted(ds.reads) == sorted(i for b in seen for i in b)  # nothing beyond this rank's share was read / decoded
    assert t.model.calls == [((4 // world, 3, case.cfg.image_size, case.cfg.image_size), 1.0 / world)] * 4
    torch.save({"seen": seen, "losses": losses, "params": t.model.flat_params.clone()}, f"{out}.r{rank}")
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_rank_aware_loader_splits_every_global_batch(tmp_path):
    """world 2: each rank's rebuilt loader (parallel.shard_loader, installed by build_model) yields disjoint halves of every global batch;
    their union over the ranks is the single-process batch, in the single-process order; the plugin's steps on them equal the
    single-process run; and the logged loss is the GLOBAL-batch mean (nn.DataParallel gathers the logits, trainers/mudpt.py:249-256)."""
    from mudpt_amd import trainer
    out = str(tmp_path / "ld")
    mp.spawn(_loader_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r0, r1 = torch.load(out + ".r0"), torch.load(out + ".r1")
    case = GoldenCase("mudpt_tiny")
    ds = _ToyDataset(16, case.cfg.image_size)
    t = object.__new__(trainer.MuDPT)
    t.model = _OracleModel(case, case.params)
    t.optim = torch.optim.SGD(t.model.parameters(), lr=0.05, momentum=0.9, weight_decay=5e-4)
    t.device, t.batch_idx, t.num_batches = torch.device("cpu"), 0, 99
    single, losses = [], []
    for batch in _toy_loader(ds, 4):
        single.append(batch["index"].tolist())
        losses.append(t.forward_backward(batch)["loss"])
    assert [a + b for a, b in zip(r0["seen"], r1["seen"])] == single  # rank 0's half then rank 1's = the single-process batch
    assert all(set(a).isdisjoint(b) for a, b in zip(r0["seen"], r1["seen"]))
    assert r0["losses"] == r1["losses"]  # every rank logs the same (global) value
    torch.testing.assert_close(torch.tensor(r0["losses"]), torch.tensor(losses), atol=2e-6, rtol=1e-5)
    assert torch.equal(r0["params"], r1["params"])
    torch.testing.assert_close(r0["params"], t.model.flat_params, atol=2e-6, rtol=1e-5)


def _lockstep_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from mudpt_amd import parallel
    import torch.distributed as dist
    parallel.init("gloo")
    ds = _ToyDataset(16, 4)
    # (1) ranks in lock-step: two epochs pass the per-epoch check
    ld = parallel.shard_loader(_toy_loader(ds, 4))
    torch.manual_seed(99)
    epochs = [[b["index"].tolist() for b in ld] for _ in range(2)]
    # (2) rank 1 consumed ONE extra random number before the epoch's permutation is drawn: both ranks must raise, nobody hangs
    torch.manual_seed(99)
    if rank == 1:
        torch.rand(1)
    try:
        list(ld)
        raised = None
    except RuntimeError as e:
        raised = str(e)
    # (3) a loader that is rank-aware already is left alone: the documented opt-out, and torch's DistributedSampler
    from torch.utils.data.distributed import DistributedSampler
    dl = torch.utils.data.DataLoader(ds, batch_size=2, sampler=DistributedSampler(ds, num_replicas=world, rank=rank, shuffle=False))
    same = parallel.shard_loader(dl) is dl
    per_rank = [b["index"].tolist() for b in dl]
    os.environ["MUDPT_DATA_SHARDED"] = "1"
    plain = torch.utils.data.DataLoader(ds, batch_size=4)
    opt_out = parallel.shard_loader(plain) is plain
    del os.environ["MUDPT_DATA_SHARDED"]
    torch.save({"epochs": epochs, "raised": raised, "same": same, "per_rank": per_rank, "opt_out": opt_out}, f"{out}.r{rank}")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sharded_sampler_verifies_rank_lock_step_and_leaves_rank_aware_loaders_alone(tmp_path):
    """ShardedBatchSampler is only right while every rank draws the same permutation.  Once per epoch a checksum of the first global index
    batch is compared over the ranks (its own gloo group); a rank whose RNG stream slipped makes EVERY rank raise.  A loader that is
    per-rank already (MUDPT_DATA_SHARDED=1, DistributedSampler) is returned unchanged, not sharded twice."""
    out = str(tmp_path / "ls")
    mp.spawn(_lockstep_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r0, r1 = torch.load(out + ".r0"), torch.load(out + ".r1")
    for e0, e1 in zip(r0["epochs"], r1["epochs"]):
        assert len(e0) == 4 and all(len(a) == 2 and set(a).isdisjoint(b) for a, b in zip(e0, e1))
        assert sorted(i for b in e0 + e1 for i in b) == list(range(16))
    assert r0["raised"] and r1["raised"] and "lock-step" in r0["raised"] and "lock-step" in r1["raised"]
    assert r0["same"] and r1["same"] and r0["opt_out"] and r1["opt_out"]
    assert sorted(i for b in r0["per_rank"] + r1["per_rank"] for i in b) == list(range(16))  # each rank kept its WHOLE per-rank batches


def test_shard_loader_falls_back_for_list_like_loaders():
    from mudpt_amd import parallel
    ds = _ToyDataset(8, 4)
    assert parallel.shard_loader([{"img": 0}], 0, 2) is None          # dassl_lite's synthetic manager: slice after load instead
    ld = torch.utils.data.DataLoader(ds, batch_size=4)
    assert parallel.shard_loader(ld, 0, 1) is ld                      # one process: untouched
    halves = [list(parallel.shard_loader(ld, r, 2).batch_sampler) for r in range(2)]
    assert halves == [[[0, 1], [4, 5]], [[2, 3], [6, 7]]]
    with pytest.raises(ValueError, match="not divisible"):
        list(parallel.shard_loader(torch.utils.data.DataLoader(ds, batch_size=3), 0, 2).batch_sampler)


class _OverflowModel(torch.nn.Module):
    """forward_backward writes inf into the bucket on chosen steps: what an overflowed fp16 token gradient does to the real library."""

    def __init__(self, bad_steps):
        super().__init__()
        self.w = torch.nn.Parameter(torch.ones(4))
        self.flat_params, self.flat_grads = self.w.data, torch.zeros(4)
        self.w.grad = self.flat_grads
        self.bad, self.n, self.loss_scale, self.scales = set(bad_steps), 0, 128.0, []

    def forward_backward(self, image, label, grad_scale=1.0):
        self.flat_grads.fill_(float("inf") if self.n in self.bad else 1.0)
        self.n += 1
        return torch.tensor(0.5)

    def set_loss_scale(self, s):
        self.loss_scale = s
        self.scales.append(s)

    def invalidate_text_cache(self):
        pass


def test_overflow_skips_the_step_and_halves_the_loss_scale(monkeypatch):
    """GradScaler semantics (the reference's amp path, trainers/mudpt.py:228,243-246): non-finite gradients -> no optimizer step, scale
    halved; a run of clean steps doubles it again up to the initial value; a non-finite LOSS is an error (Dassl's detect_anomaly)."""
    from mudpt_amd import trainer
    monkeypatch.setattr(trainer, "LOSS_SCALE_GROWTH_INTERVAL", 3)
    t = object.__new__(trainer.MuDPT)
    t.model = _OverflowModel({1, 2})
    t.optim = torch.optim.SGD(t.model.parameters(), lr=0.1)
    t.device, t.batch_idx, t.num_batches = torch.device("cpu"), 0, 99
    batch = {"img": torch.zeros(2, 3, 4, 4), "label": torch.zeros(2, dtype=torch.long)}
    w = []
    for _ in range(9):
        assert t.forward_backward(batch) == {"loss": 0.5}
        w.append(t.model.w[0].item())
    # steps 1 and 2 overflow: parameters unchanged there, the scale goes 128 -> 64 -> 32, then two growth intervals bring it back
    assert w[0] == pytest.approx(0.9) and w[1] == w[0] and w[2] == w[0] and w[3] == pytest.approx(0.8)
    assert t.model.scales == [64.0, 32.0, 64.0, 128.0] and t._loss_scale_state["skipped"] == 2
    t.model.forward_backward = lambda *a, **k: torch.tensor(float("nan"))
    with pytest.raises(FloatingPointError, match="Loss is infinite or NaN"):
        t.forward_backward(batch)
    t2 = object.__new__(trainer.MuDPT)
    t2.model = _OverflowModel(set(range(100)))
    t2.optim = torch.optim.SGD(t2.model.parameters(), lr=0.1)
    t2.device, t2.batch_idx, t2.num_batches = torch.device("cpu"), 0, 99
    with pytest.raises(FloatingPointError, match="smallest loss scale"):
        for _ in range(20):
            t2.forward_backward(batch)
    assert t2.model.loss_scale == 1.0 and t2.model.w[0].item() == 1.0  # never stepped on garbage


def test_unsharded_handle_never_exchanges(monkeypatch):
    """world > 1 with every rank encoding ALL classes (class_shard None): the class-parallel phases must not sum the (identical) tables --
    d(features) would come out world times too large without any error."""
    import torch.distributed as dist
    from mudpt_amd.model import CustomCLIP
    m = object.__new__(CustomCLIP)
    torch.nn.Module.__init__(m)
    m.class_shard, m.group = None, None
    monkeypatch.setattr(dist, "is_initialized", lambda: True)
    monkeypatch.setattr(dist, "get_world_size", lambda group=None: 2)
    called = []
    monkeypatch.setattr(dist, "all_reduce", lambda *a, **k: called.append(1))
    assert m._exchange(torch.zeros(3)) is None and not called
    m.class_shard = (0, 5)
    m._exchange(torch.zeros(3))
    assert called == [1]
