#!/usr/bin/env python3
"""`train --train_dir` throughput (VERDICT r2 task 4): N synthetic PNG + JSON samples on disk (schema of the
reference's README.md:40-71), decoded by the thread-pool feed, resized / jittered / flipped / normalised and encoded on the GPU, through
the real TrainStep -- images/s next to the synthetic-tensor step of the same batch (what bench.py times).
usage: feed_bench.py [--n 512] [--batch 64] [--steps 24] [--workers 0] [--amp] [--no_augmentation] [--dir /tmp/sd_feed]"""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def write_samples(directory, n, size, seed=7):
    """photo-like content (smooth fields + fine noise: PNGs of ~1/2 the raw size, like camera images, not blocky noise that inflates fast);
    scenes from the product's own seeded generator (data/synthetic.py)"""
    from PIL import Image

    from structuredetector_amd.data.synthetic import synthetic_batch
    directory.mkdir(parents=True, exist_ok=True)
    rng = np.random.default_rng(seed)
    labels, parts = ["bean", "maize"], ["leaf"]
    for i in range(n):
        low = rng.integers(0, 256, (size // 32, size // 32, 3), dtype=np.uint8)
        img = np.asarray(Image.fromarray(low).resize((size, size), Image.BICUBIC), np.int16) + rng.integers(-12, 13, (size, size, 3), dtype=np.int16)
        Image.fromarray(np.clip(img, 0, 255).astype(np.uint8)).save(directory / f"img_{i:04d}.png", compress_level=3)
        n_obj, o_lab, o_xy, o_np, p_kind, p_xy = synthetic_batch(rng, 1, size, size, len(labels), len(parts))
        objs, j = [], 0
        for k in range(int(n_obj[0])):
            ps = [{"kind": parts[int(p_kind[j + q])], "location": {"x": float(p_xy[j + q][0]), "y": float(p_xy[j + q][1])}} for q in range(int(o_np[k]))]
            j += int(o_np[k])
            objs.append({"label": labels[int(o_lab[k])], "box": None,
                         "parts": [{"kind": "stem", "location": {"x": float(o_xy[k][0]), "y": float(o_xy[k][1])}}] + ps})
        js = {"image_path": str(directory / f"img_{i:04d}.png"), "img_size": [size, size], "objects": objs}
        (directory / f"img_{i:04d}.json").write_text(json.dumps(js))
    (directory.parent / "feed_labels.json").write_text(json.dumps({"labels": labels, "parts": parts}))
    return directory.parent / "feed_labels.json"


def run(n=512, batch=64, size=512, steps=24, workers=0, amp=False, no_augmentation=False, directory="/tmp/sd_feed", breakdown=True, synthetic=True):
    """Returns the figures as a dict (bench.py's `directory_feed` extra calls this with breakdown / synthetic off)."""
    import os
    from concurrent.futures import ThreadPoolExecutor

    from structuredetector_amd.data import CropDataset
    from structuredetector_amd.data.feeder import BatchFeeder
    from structuredetector_amd.model.trainer import Trainer, shard_indices
    from structuredetector_amd.utils.args import Arguments
    root = Path(directory)
    t0 = time.perf_counter()
    if len(list((root / "train").glob("*.png"))) == n and (root / "feed_labels.json").exists():
        labels = root / "feed_labels.json"                  # (a previous run's samples: same seed, same files)
    else:
        labels = write_samples(root / "train", n, size)
    t_write = time.perf_counter() - t0
    png_mb = sum(f.stat().st_size for f in (root / "train").glob("*.png")) / n / 1e6
    common = ["--labels", str(labels), "-s", "stem", "-b", str(batch), "-W", str(size), "-H", str(size), "-e", "1000"] + (["--amp"] if amp else []) + \
             (["-a"] if no_augmentation else [])
    out = {"samples": n, "png_mb_each": round(png_mb, 3), "write_s": round(t_write, 1), "batch": batch, "amp": amp, "augmentation": not no_augmentation}
    threads_before = torch.get_num_threads()

    def rate(argv, label):
        args = Arguments().parse(argv)
        tr = Trainer(args)
        it = tr.batches()
        k, t_start = 0, None
        warm = 4
        while k < warm + steps:
            try:
                images, targets = next(it)
            except StopIteration:
                tr.epoch += 1
                it = tr.batches()
                continue
            tr.step(images, targets)
            k += 1
            if k == warm:
                torch.cuda.synchronize(); t_start = time.perf_counter()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t_start
        it.close()
        out[label] = round(steps * batch / dt, 1)
        out[label.replace("img_s", "ms_per_step")] = round(dt / steps * 1e3, 2)

    def host_breakdown(argv):
        """where a directory-fed step spends its host time: waiting for the feeder, augmentation call, target encoding, step launch"""
        args = Arguments().parse(argv)
        tr = Trainer(args)
        shards = shard_indices(len(tr.dataset), batch, 0, 1, 1)
        feed = iter(BatchFeeder(tr.dataset, shards * 4, args.device, workers=workers or None, depth=3))
        acc = {"wait": 0.0, "augment": 0.0, "encode": 0.0, "step": 0.0}
        for i in range(4 + 12):
            t0 = time.perf_counter(); b = next(feed)
            t1 = time.perf_counter(); images, anns = tr.augment(b, b.annotations)
            t2 = time.perf_counter(); targets = tr.encode.batch(tr.augment.size, anns, args.device)
            t3 = time.perf_counter(); tr.step(images, targets)
            t4 = time.perf_counter()
            if i >= 4:
                for key, v in zip(acc, (t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
                    acc[key] += v / 12 * 1e3
        torch.cuda.synchronize()
        feed.close()
        out["host_ms_per_step"] = {key: round(v, 2) for key, v in acc.items()}

    try:
        if synthetic:
            rate(common + ["--synthetic", str(batch * 8)], "synthetic_img_s")
        if breakdown:
            host_breakdown(common + ["--train_dir", str(root / "train")])
        rate(common + ["--train_dir", str(root / "train"), "--decode_workers", str(workers)], "directory_img_s")
        if synthetic:
            out["directory_over_synthetic"] = round(out["directory_img_s"] / out["synthetic_img_s"], 3)
        # decode alone (what the pool sustains without the GPU step): every sample once through the dataset reader on the pool
        args = Arguments().parse(common + ["--train_dir", str(root / "train")])
        ds = CropDataset(args, root / "train", raw=True)
        from structuredetector_amd.data.feeder import default_decode_workers
        nw = workers or default_decode_workers()
        t0 = time.perf_counter()
        with ThreadPoolExecutor(nw) as pool:
            list(pool.map(ds.__getitem__, range(len(ds))))
        out["decode_only_img_s"] = round(len(ds) / (time.perf_counter() - t0), 1)
        out["decode_workers"] = nw
    finally:
        torch.set_num_threads(threads_before)          # (the directory Trainer runs torch's intra-op pool single-threaded)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=512); ap.add_argument("--batch", type=int, default=64); ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--steps", type=int, default=24); ap.add_argument("--workers", type=int, default=0)
    ap.add_argument("--amp", action="store_true"); ap.add_argument("--no_augmentation", action="store_true")
    ap.add_argument("--dir", default="/tmp/sd_feed")
    a = ap.parse_args()
    print(json.dumps(run(a.n, a.batch, a.size, a.steps, a.workers, a.amp, a.no_augmentation, a.dir)))


if __name__ == "__main__":
    main()
