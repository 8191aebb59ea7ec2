#!/usr/bin/env python3
"""`evaluate --valid_dir` wall time per image over N synthetic PNG + JSON samples (the reference's batch-1 loop, src/sdnet/cli/evaluate.py:34-45):
the sequential reader (--decode_workers 1 = the reference's order of work) against the prefetching one (data/feeder.prefetch_items).
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
    base = ["--valid_dir", str(root / "valid"), "--labels", str(labels), "-s", "stem", "-W", str(a.size), "-H", str(a.size)]
    for label, extra in (("sequential reader (1 thread, no look-ahead)", ["--decode_workers", "1", "--prefetch", "1"]), ("prefetching reader (default pool)", [])):
        for rep in range(2):                                    # second pass: page cache warm, kernels loaded
            torch.cuda.synchronize(); t0 = time.perf_counter()
            with contextlib.redirect_stdout(io.StringIO()):
                evaluate.main(base + extra)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"{label:46s} {dt / a.n * 1e3:7.2f} ms per image  ({a.n / dt:7.1f} img/s)")


if __name__ == "__main__":
    main()
