#!/usr/bin/env python3
"""Where does the host block in a directory-fed step?  Times the pieces of TrainAugmentation.__call__ with the feeder running."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from structuredetector_amd.data import BatchFeeder  # noqa: E402
from structuredetector_amd.data import augment as A  # noqa: E402
from structuredetector_amd.model.trainer import Trainer, shard_indices  # noqa: E402
from structuredetector_amd.utils.args import Arguments  # noqa: E402

root = Path("/tmp/sd_feed")
args = Arguments().parse(["--labels", str(root / "feed_labels.json"), "-s", "stem", "-b", "64", "--train_dir", str(root / "train"), "-e", "1000"])
tr = Trainer(args)
shards = shard_indices(len(tr.dataset), 64, 0, 1, 1)
T = {}
orig_pre = A.preprocess_images


def timed(name, fn):
    def w(*a, **k):
        t0 = time.perf_counter()
        r = fn(*a, **k)
        T[name] = T.get(name, 0.0) + time.perf_counter() - t0
        return r
    return w


A.preprocess_images = timed("preprocess_images", orig_pre)
tr.augment.draws_for = timed("draws_for", tr.augment.draws_for)
for with_step in (False, True):
    T.clear()
    feed = iter(BatchFeeder(tr.dataset, shards * 8, args.device, workers=16, depth=3))
    tot = {"wait": 0.0, "augment": 0.0, "encode": 0.0, "step": 0.0}
    for i in range(16):
        t0 = time.perf_counter(); batch = next(feed)
        t1 = time.perf_counter(); images, anns = tr.augment(batch, batch.annotations)
        t2 = time.perf_counter(); targets = tr.encode.batch(tr.augment.size, anns, args.device)
        t3 = time.perf_counter()
        if with_step:
            tr.step(images, targets)
        t4 = time.perf_counter()
        if i >= 4:
            for k, v in zip(tot, (t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
                tot[k] += v / 12 * 1e3
        else:
            T.clear()
    torch.cuda.synchronize()
    feed.close()
    print("with step" if with_step else "no step  ", {k: round(v, 1) for k, v in tot.items()}, "inside augment (ms/batch):", {k: round(v / 12 * 1e3, 1) for k, v in T.items()}, flush=True)
