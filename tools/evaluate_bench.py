#!/usr/bin/env python3
"""`evaluate --valid_dir` wall time per image over N synthetic PNG + JSON samples (the reference's batch-1 loop, src/sdnet/cli/evaluate.py:34-45):
one image per forward + decode on one decode thread (the reference's order of work) against the batched path (model/predictor.py).
usage: evaluate_bench.py [--n 128] [--size 512] [--dir /tmp/sd_eval]"""
import argparse
import contextlib
import io
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from tools.feed_bench import write_samples  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=128); ap.add_argument("--size", type=int, default=512); ap.add_argument("--dir", default="/tmp/sd_eval")
    a = ap.parse_args()
    from structuredetector_amd.cli import evaluate
    root = Path(a.dir)
    labels = write_samples(root / "valid", a.n, a.size)
    # the reference's evaluate REQUIRES a checkpoint (cli/evaluate.py:14-16): a seeded one is written once and loaded by every run
    ckpt = root / "seeded.pth"
    if not ckpt.exists():
        from argparse import Namespace
        from structuredetector_amd.model import Network
        Network(Namespace(labels={"bean": 0, "maize": 1}, parts={"leaf": 0}, fpn_depth=128), pretrained=False).save(ckpt)
    base = ["--valid_dir", str(root / "valid"), "--labels", str(labels), "-s", "stem", "-W", str(a.size), "-H", str(a.size), "-o", str(ckpt)]
    rows = (("batch of one, one decode thread (the reference's order of work)", ["--eval_batch", "1", "--decode_workers", "1"]),
            ("batch of one, decode pool", ["--eval_batch", "1"]),
            ("--eval_batch 16 (default), decode pool", []),
            ("--eval_batch 32, decode pool", ["--eval_batch", "32"]))
    for label, extra in rows:
        for rep in range(2):                                    # second pass: page cache warm, kernels loaded
            torch.cuda.synchronize(); t0 = time.perf_counter()
            with contextlib.redirect_stdout(io.StringIO()):
                evaluate.main(base + extra)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"{label:66s} {dt / a.n * 1e3:7.2f} ms per image  ({a.n / dt:7.1f} img/s)", flush=True)


if __name__ == "__main__":
    main()
