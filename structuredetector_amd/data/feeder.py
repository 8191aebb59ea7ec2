"""Directory-fed batches at step speed.

The reference feeds its training loop from a `DataLoader` with 4 persistent worker PROCESSES, `prefetch_factor=4` and pinned memory
(src/sdnet/model/trainer.py:62-72) over `CropDataset.__getitem__` (src/sdnet/data/dataset.py:41-49: JSON + PIL decode + the whole
transform chain per sample).  Here only the decode stays on the host -- resize, jitter, flips, normalisation and the target rendering
run for the whole batch on the GPU -- so the feed is:

  * a pool of decode THREADS (PIL's decoders and the copies release the GIL; no pickling of images between processes),
    kept `depth` batches ahead of the consumer;
  * one producer thread that groups a batch's images by size, copies each group into a PINNED staging buffer and uploads it with an
    asynchronous copy on a side stream, marked by an event;
  * the consumer (the training loop) makes its stream wait for that event: the decode and the upload of batch n+1 .. n+depth overlap
    step n, nothing in the loop blocks on the host.
"""
from __future__ import annotations

import os
import queue
import threading
from concurrent.futures import ThreadPoolExecutor

import torch


def cpu_share():
    """Cores this process may really use: the cgroup CPU quota where there is one (a container sees all host cores in
    os.cpu_count()), else the scheduler affinity, else the core count."""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]                       # cgroup v2
        if quota != "max":
            return max(1, int(int(quota) / int(period)))
    except (OSError, ValueError):
        pass
    try:
        quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())                         # cgroup v1
        period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if quota > 0 and period > 0:
            return max(1, quota // period)
    except (OSError, ValueError):
        pass
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 4


def default_decode_workers(world=1):
    """Half of this rank's cores, 2 .. 16: PNG / JPEG decoding is CPU work (~9 ms per 512x512 PNG), and the training loop's launch thread
    and the producer need cores of their own -- measured on a 16-core share, fp32 step of 76 ms: 8 workers 841 img/s (99 % of the
    synthetic-tensor step), 12: 822, 16: 815, 24: 732."""
    return max(2, min(16, cpu_share() // max(world, 1) // 2))


def prefetch_items(dataset, workers=None, depth=None):
    """`dataset[0], dataset[1], ...` in order, read `depth` items ahead by a pool of `workers` threads.  For the host-only readers of the
    batch-1 loops (`evaluate`: CropDataset(raw=False), `detect`: PredictionDataset -- PIL decode + resize + normalise, no GPU calls, the
    decoders release the GIL): the reference walks them sequentially (src/sdnet/cli/evaluate.py:34-45), which at ~7 ms of PNG decode per
    image leaves a 0.8 ms forward + decode waiting; items and their order are exactly those of the sequential walk."""
    from collections import deque
    n = len(dataset)
    workers = int(workers or default_decode_workers())
    depth = int(depth or 2 * workers)
    if n == 0:
        return
    with ThreadPoolExecutor(max_workers=workers, thread_name_prefix="sd-read") as pool:
        pending, nxt = deque(), 0
        try:
            while nxt < n and len(pending) < depth:
                pending.append(pool.submit(dataset.__getitem__, nxt)); nxt += 1
            while pending:
                item = pending.popleft().result()
                if nxt < n:
                    pending.append(pool.submit(dataset.__getitem__, nxt)); nxt += 1
                yield item
        finally:
            for f in pending:                                    # consumer stopped early (or a reader raised)
                f.cancel()


class GroupedBatch:
    """One batch as the GPU pipeline wants it: `groups` = {(height, width): (sample positions, (n, height, width, 3) uint8 DEVICE tensor)},
    `annotations` in sample order (ORIGINAL image pixels), `ready` = event after which the device tensors are complete."""

    def __init__(self, groups, annotations, ready, keep):
        self.groups, self.annotations, self.ready, self._keep = groups, annotations, ready, keep

    def __len__(self):
        return len(self.annotations)


class BatchFeeder:
    def __init__(self, dataset, index_batches, device, workers=None, depth=3):
        """dataset: CropDataset(raw=True) -- items ((H, W, 3) uint8 CPU tensor, annotation); index_batches: the epoch's batches of sample
        indices (trainer.shard_indices); workers: decode threads (default: default_decode_workers()); depth: batches in flight ahead of the consumer."""
        self.dataset, self.batches, self.device = dataset, [list(int(j) for j in b) for b in index_batches], torch.device(device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.workers = int(workers or default_decode_workers())
        self.depth = max(1, int(depth))
        self._pinned = {}             # (n, h, w) -> list of [buffer, event of the upload that last read it]
        self._stop = threading.Event()

    # ---- staging buffers: `depth + 2` pinned buffers per shape in rotation; a buffer is rewritten only after the upload that read it is done
    def _staging(self, n, h, w):
        ring = self._pinned.setdefault((n, h, w), {"next": 0, "slots": []})
        if len(ring["slots"]) < self.depth + 2:
            ring["slots"].append([torch.empty((n, h, w, 3), dtype=torch.uint8, pin_memory=True), None])
            return ring["slots"][-1]
        slot = ring["slots"][ring["next"] % len(ring["slots"])]
        ring["next"] += 1
        if slot[1] is not None:
            slot[1].synchronize()
        return slot

    def _produce(self, out):
        try:
            torch.cuda.set_device(self.device)
            side = torch.cuda.Stream(self.device)
            with ThreadPoolExecutor(max_workers=self.workers, thread_name_prefix="sd-decode") as pool:
                pending = []                                   # futures of the batches submitted so far, `depth` batches ahead
                nxt = 0

                def submit_more():
                    nonlocal nxt
                    while nxt < len(self.batches) and len(pending) <= self.depth:
                        pending.append([pool.submit(self.dataset.__getitem__, j) for j in self.batches[nxt]])
                        nxt += 1
                submit_more()
                while pending and not self._stop.is_set():
                    items = [f.result() for f in pending.pop(0)]
                    submit_more()
                    by_size = {}
                    for pos, (im, _) in enumerate(items):
                        by_size.setdefault((int(im.shape[0]), int(im.shape[1])), []).append(pos)
                    groups, keep = {}, []
                    with torch.cuda.stream(side):
                        for (h, w), idx in by_size.items():
                            slot = self._staging(len(idx), h, w)
                            for k, pos in enumerate(idx):
                                slot[0][k].copy_(items[pos][0])                  # pageable -> pinned (releases the GIL)
                            dev = slot[0].to(self.device, non_blocking=True)     # asynchronous: pinned source, side stream
                            ev = torch.cuda.Event()
                            ev.record(side)
                            slot[1] = ev
                            groups[(h, w)] = (idx, dev)
                            keep.append(slot)
                        ready = torch.cuda.Event()
                        ready.record(side)
                    batch = GroupedBatch(groups, [a for _, a in items], ready, keep)
                    while not self._stop.is_set():
                        try:
                            out.put(batch, timeout=0.2)
                            break
                        except queue.Full:
                            continue
                for fs in pending:                              # consumer stopped early
                    for f in fs:
                        f.cancel()
            out.put(None)
        except BaseException as err:                            # hand the failure to the consumer instead of dying silently
            out.put(err)

    def __iter__(self):
        out = queue.Queue(maxsize=self.depth)
        self._stop.clear()
        thread = threading.Thread(target=self._produce, args=(out,), name="sd-feeder", daemon=True)
        thread.start()
        try:
            while True:
                item = out.get()
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                cur = torch.cuda.current_stream(self.device)
                cur.wait_event(item.ready)                      # device-side wait: the host does not block
                for _, dev in item.groups.values():
                    dev.record_stream(cur)                      # allocated on the side stream, consumed on this one
                yield item
        finally:
            self._stop.set()
            while thread.is_alive():                            # unblock a producer waiting on a full queue
                try:
                    out.get_nowait()
                except queue.Empty:
                    pass
                thread.join(timeout=0.1)
