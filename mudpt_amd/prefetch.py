"""Host -> device batch prefetch for the trainer plugins.

The C ABI takes DEVICE pointers; the reference's ``parse_batch_train`` (trainers/mudpt.py:263-268) moves a host batch with a
synchronous ``.to(device)`` at the top of every step: 256 x 3 x 224 x 224 fp32 = 154 MB, ~3.1 ms over PCIe Gen5 x16 in front of a
~26 ms step.  ``DevicePrefetcher`` wraps the loader the trainer iterates (``self.train_loader_x``): a worker thread
draws batches ahead (depth 2), slices each to this rank's share (``parallel.shard_batch``: nn.DataParallel's scatter) and copies it on
a side stream while the current step runs, so ``parse_batch_train`` finds device tensors and its ``.to(device)`` is a no-op.  The copy
goes straight from the loader's (pageable or pinned) memory: the runtime's own pageable path moves the 154 MB in ~3 ms, while a
hand-made pageable -> pinned staging copy on one host thread took 20-30 ms per batch (measured; removed).  Order, contents and length
of the wrapped loader are unchanged; without a GPU it passes batches through.
"""
from __future__ import annotations

import queue
import threading

import torch

from . import parallel

_END = object()


class DevicePrefetcher:
    DEPTH = 2  # batches staged ahead of the one the trainer is working on

    def __init__(self, loader, device=None, keys=("img", "label"), shard=True):
        self.loader, self.keys = loader, tuple(keys)
        self.shard = shard  # False: the loader is rank-aware already (parallel.shard_loader), its batches are this rank's slices
        self.device = torch.device(device) if device is not None else (torch.device("cuda") if torch.cuda.is_available() else None)
        self.enabled = self.device is not None and self.device.type == "cuda" and torch.cuda.is_available()
        self.stream = torch.cuda.Stream(self.device) if self.enabled else None

    def __len__(self):
        return len(self.loader)

    def __getattr__(self, name):  # dataset, batch_size, sampler ... of the wrapped loader
        return getattr(self.loader, name)

    def __getitem__(self, i):  # list-like loaders (dassl_lite's synthetic manager): the raw host batch
        return self.loader[i]

    def _stage(self, batch):
        """This rank's slice of the batch on the device (copy enqueued on the side stream), other entries untouched."""
        if not all(k in batch for k in self.keys):
            return batch, None
        img, label = parallel.shard_batch(batch[self.keys[0]], batch[self.keys[1]]) if self.shard else (batch[self.keys[0]], batch[self.keys[1]])
        out = dict(batch)
        out["_mudpt_sharded"] = True
        with torch.cuda.stream(self.stream):
            for k, t in ((self.keys[0], img), (self.keys[1], label)):
                out[k] = t.to(self.device, non_blocking=True)  # blocks THIS (worker) thread for pageable memory, never the trainer's
            ev = torch.cuda.Event()
            ev.record(self.stream)
        return out, ev

    def _worker(self, q, stop):
        try:
            if self.enabled:
                torch.cuda.set_device(self.device)
            for batch in self.loader:
                if stop.is_set():
                    return
                q.put(self._stage(batch))
            q.put(_END)
        except BaseException as e:  # surfaces in the consuming thread
            q.put(e)

    def __iter__(self):
        if not self.enabled:
            yield from self.loader
            return
        q, stop = queue.Queue(maxsize=self.DEPTH), threading.Event()
        th = threading.Thread(target=self._worker, args=(q, stop), daemon=True, name="mudpt-prefetch")
        th.start()
        try:
            while True:
                item = q.get()
                if item is _END:
                    return
                if isinstance(item, BaseException):
                    raise item
                batch, ev = item
                if ev is not None:
                    cur = torch.cuda.current_stream(self.device)
                    cur.wait_event(ev)  # the step's kernels wait for the copy, the host does not
                    # The staged tensors were allocated on the side stream: tell the caching allocator that the consumer's stream uses
                    # them, or a batch dropped without a host sync could hand its block back to the side-stream pool -- and to the NEXT
                    # batch's copy -- while a conversion / patch-gather kernel of this step still reads it (a uint8 or fp16 loader, a
                    # caller that does not sync on the loss).
                    for k in self.keys:
                        t = batch.get(k)
                        if isinstance(t, torch.Tensor) and t.is_cuda:
                            t.record_stream(cur)
                yield batch
        finally:
            stop.set()
            while th.is_alive():  # unblock a producer waiting on a full queue, then let it finish
                try:
                    q.get_nowait()
                except queue.Empty:
                    pass
                th.join(timeout=0.05)
