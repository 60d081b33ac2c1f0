"""Data-parallel plumbing: one process per GPU, ONE all-reduce per step over the flat gradient bucket.

Replaces the reference's single-process ``nn.DataParallel`` (trainers/mudpt.py:230-233), which re-broadcasts all
125.6 M parameters every forward and reduces every gradient onto GPU 0.  Here the frozen backbone is resident on
every rank, images are independent units (no data-path collective) and the only exchange is the sum of the
4.97 MB fp32 bucket holding the 10 trainable tensors (RCCL over xGMI when the backend is "nccl").

Averaging convention: the reference's loss is ``F.cross_entropy`` (mean) over the concatenated batch
(trainers/mudpt.py:249-250).  Each rank computes the gradient of (1/world) * mean over its local batch
(``grad_scale = 1 / world``); the SUM over ranks is then the gradient of the global-batch mean (equal local batches).
"""
from __future__ import annotations

import os
from typing import Tuple

import torch


def env_rank() -> Tuple[int, int, int]:
    """(rank, world, local_rank) from the torch.distributed.run environment (defaults: single process)."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def init(backend: str | None = None):
    """Initialise the default process group when launched with WORLD_SIZE > 1; returns (rank, world, local_rank)."""
    import torch.distributed as dist
    rank, world, local = env_rank()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if torch.cuda.is_available():
            torch.cuda.set_device(local)  # RCCL binds the communicator to the current device
        # a bounded rendezvous / collective timeout (MUDPT_DIST_TIMEOUT_S, default 600 s as torch's own): a rank that never arrives fails the
        # job with an error instead of holding the others for torch's default on every later collective as well
        from datetime import timedelta
        dist.init_process_group(backend or ("nccl" if torch.cuda.is_available() else "gloo"),
                                timeout=timedelta(seconds=float(os.environ.get("MUDPT_DIST_TIMEOUT_S", "600"))))
    return rank, world, local


def world_size() -> int:
    import torch.distributed as dist
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank() -> int:
    import torch.distributed as dist
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def is_main() -> bool:
    """Rank 0 writes checkpoints and logs; every rank holds the same parameters after each step."""
    return rank() == 0


def grad_scale() -> float:
    return 1.0 / world_size()


def allreduce_grads(flat_grads: torch.Tensor) -> torch.Tensor:
    """Sum the flat gradient bucket over ranks in place (no-op for a single process)."""
    if world_size() > 1:
        import torch.distributed as dist
        dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM)
    return flat_grads


def broadcast_params(flat_params: torch.Tensor, src: int = 0) -> torch.Tensor:
    """Make every replica start from rank ``src``'s trainable tensors (replicas otherwise rely on equal seeds)."""
    if world_size() > 1:
        import torch.distributed as dist
        dist.broadcast(flat_params, src)
    return flat_params


def class_range(n_cls: int, r: int | None = None, world: int | None = None) -> Tuple[int, int]:
    """Rank r's contiguous share [c0, c1) of the class prompts for the class-parallel text tower (include/mudpt.h, SURVEY 8e second
    axis): balanced to within one class; with more ranks than classes the surplus ranks share the last class (every handle encodes
    at least one; the exchange is a sum, so a class held twice must not happen -- callers use world <= n_cls)."""
    r = rank() if r is None else r
    world = world_size() if world is None else world
    assert 0 < world <= n_cls, f"class-parallel text tower needs world ({world}) <= n_cls ({n_cls})"
    base, extra = divmod(n_cls, world)
    c0 = r * base + min(r, extra)
    return c0, c0 + base + (1 if r < extra else 0)


def shard_batch(image: torch.Tensor, label: torch.Tensor):
    """This rank's contiguous slice of a global batch, as ``nn.DataParallel``'s scatter along dim 0 gives GPU k
    (trainers/mudpt.py:230-233).  Dassl's loaders are not rank-aware: launched under torch.distributed.run every rank draws the
    SAME global batch (equal seeds), so slicing it here reproduces the reference's split with no data-path collective.  A loader
    that already yields per-rank batches (a DistributedSampler) opts out with MUDPT_DATA_SHARDED=1."""
    w = world_size()
    if w == 1 or os.environ.get("MUDPT_DATA_SHARDED") == "1":
        return image, label
    n = image.shape[0]
    if n % w != 0:
        raise ValueError(f"global batch {n} is not divisible by the {w} data-parallel ranks (equal shards are what makes the "
                         "sum of 1/world-scaled local gradients the global-batch mean)")
    idx = shard(n, rank(), w)
    return image[idx.start:idx.stop], label[idx.start:idx.stop]


def step_consensus(loss: torch.Tensor, flat_grads: torch.Tensor):
    """What every rank must agree on after the bucket all-reduce, in ONE 3-element all-reduce (sum): (loss finite on every rank,
    gradients finite on every rank, the GLOBAL-batch mean loss).  The reference's nn.DataParallel gathers the logits, so its
    ``F.cross_entropy`` and the ``loss.item()`` it logs are over the whole global batch (trainers/mudpt.py:249-256): the logged value here
    is the sum over ranks of loss_local / world, not the local slice's mean.  A rank that raised alone would leave the others hanging
    in the next collective, hence the consensus flags (Dassl's detect_anomaly checks the loss before backward)."""
    lf, gf = torch.isfinite(loss).all(), torch.isfinite(flat_grads).all()
    v = torch.stack([(~lf).float(), (~gf).float(), torch.where(lf, loss.detach().float().reshape(()) * grad_scale(), torch.zeros((), device=loss.device))]).to(flat_grads.device)
    if world_size() > 1:
        import torch.distributed as dist
        dist.all_reduce(v, op=dist.ReduceOp.SUM)
    v = v.tolist()  # the step's one host sync (the reference's loss.item(), trainers/mudpt.py:254)
    return v[0] == 0, v[1] == 0, v[2]


class ShardedBatchSampler:
    """This rank's contiguous 1/world slice of every index batch of a loader's own batch sampler.  Every rank iterates the SAME
    global batch sampler (equal seeds: Dassl seeds torch / numpy / random from cfg.SEED on every rank), so the slices of a step are
    disjoint and their concatenation over the ranks is exactly the single-process batch, in the single-process order: what
    nn.DataParallel's scatter of ONE loaded batch gives (trainers/mudpt.py:230-233) -- but each rank only reads and decodes its own
    images.

    That only holds while every rank's sampler draws the same permutation, i.e. while the ranks' RNG streams stay in lock-step; a rank
    that consumed one extra random number would silently train on overlapping / missing samples.  ``check_group`` (a gloo group made by
    ``shard_loader``; its collectives are independent of the gradient all-reduce's, so the loader's prefetch thread may issue them)
    verifies it once per epoch: a checksum of the epoch's FIRST global index batch must agree on all ranks, else every rank raises."""

    def __init__(self, batch_sampler, rank: int, world: int, check_group=None):
        self.batch_sampler, self.rank, self.world, self.check_group = batch_sampler, rank, world, check_group

    def _check_lock_step(self, idx):
        import zlib
        import torch.distributed as dist
        h = float(zlib.crc32(torch.tensor(idx, dtype=torch.int64).numpy().tobytes()))  # < 2^32: exact in float64
        t = torch.tensor([h, -h], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.check_group)
        if t[0].item() != -t[1].item():  # max(h) != min(h): some rank drew a different batch -- detected on EVERY rank
            raise RuntimeError("data-parallel ranks drew different global index batches (rank %d: first indices %s): the ranks' RNG streams are out of "
                               "lock-step, so their shards would overlap / miss samples.  Seed every rank identically before the loader is built, or use a "
                               "rank-aware sampler with MUDPT_DATA_SHARDED=1" % (self.rank, idx[:8]))

    def __iter__(self):
        first = True
        for idx in self.batch_sampler:
            idx = list(idx)
            if len(idx) % self.world != 0:
                raise ValueError(f"global batch {len(idx)} is not divisible by the {self.world} data-parallel ranks")
            if first and self.check_group is not None:
                self._check_lock_step(idx)
            first = False
            r = shard(len(idx), self.rank, self.world)
            yield idx[r.start:r.stop]

    def __len__(self):
        return len(self.batch_sampler)


def loader_is_rank_aware(loader) -> bool:
    """The documented opt-out (MUDPT_DATA_SHARDED=1), or a loader built on torch's DistributedSampler: its batches are per-rank already."""
    if os.environ.get("MUDPT_DATA_SHARDED") == "1":
        return True
    from torch.utils.data.distributed import DistributedSampler
    bs = getattr(loader, "batch_sampler", None)
    return isinstance(getattr(loader, "sampler", None), DistributedSampler) or isinstance(getattr(bs, "sampler", None), DistributedSampler)


def shard_loader(loader, r: int | None = None, world: int | None = None):
    """A rank-aware copy of a torch DataLoader (same dataset, workers, collate function; batch sampler wrapped in
    ShardedBatchSampler), or None when ``loader`` is not a DataLoader with a batch sampler (list-like synthetic loaders, iterable
    datasets): the caller then falls back to slicing the loaded global batch (``shard_batch``).  A loader that is rank-aware already
    (``loader_is_rank_aware``) is returned unchanged: sharding it again would keep 1/world of every per-rank batch."""
    from torch.utils.data import DataLoader
    import torch.distributed as dist
    r = rank() if r is None else r
    world = world_size() if world is None else world
    if world == 1 or loader_is_rank_aware(loader):
        return loader
    if not isinstance(loader, DataLoader) or getattr(loader, "batch_sampler", None) is None:
        return None
    kw = dict(num_workers=loader.num_workers, collate_fn=loader.collate_fn, pin_memory=loader.pin_memory, timeout=loader.timeout,
              worker_init_fn=loader.worker_init_fn, generator=loader.generator)
    if loader.num_workers > 0:
        kw.update(prefetch_factor=loader.prefetch_factor, persistent_workers=loader.persistent_workers)
    # the lock-step check's own group (every rank passes here: new_group is collective).  gloo on CPU tensors whatever the main backend is
    group = dist.new_group(backend="gloo") if dist.is_available() and dist.is_initialized() and dist.get_world_size() == world else None
    return DataLoader(loader.dataset, batch_sampler=ShardedBatchSampler(loader.batch_sampler, r, world, group), **kw)


def barrier():
    if world_size() > 1:
        import torch.distributed as dist
        dist.barrier()


def all_ok(ok: bool) -> bool:
    """True iff ``ok`` on every rank (one MIN all-reduce): lets all ranks take the same branch after a step only one of them can fail."""
    if world_size() > 1:
        import torch.distributed as dist
        t = torch.tensor([1.0 if ok else 0.0], device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        ok = bool(t.item() > 0)
    return ok


def shard(n_items: int, rank: int, world: int) -> range:
    """Contiguous, equal shards of a global batch; the remainder (if any) is dropped like DataLoader(drop_last=True)."""
    per = n_items // world
    return range(rank * per, (rank + 1) * per)
