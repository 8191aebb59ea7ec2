#!/usr/bin/env python3
"""GPU time of sd_preprocess_images with and without the ColorJitter pass (bs=64, 512x512 -> 512x512 and 600x800 -> 512x512)."""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from structuredetector_amd.data import preprocess_images  # noqa: E402
from structuredetector_amd.data.augment import jitter_words  # noqa: E402

dev = torch.device("cuda")
rng = np.random.default_rng(0)
for (H, W) in ((512, 512), (600, 800)):
    x = torch.from_numpy(rng.integers(0, 256, (64, H, W, 3), dtype=np.uint8)).to(dev)
    flips = [int(v) for v in rng.integers(0, 4, 64)]
    words, factors = zip(*(jitter_words(list(rng.permutation(4)), rng.uniform(0.75, 1.25), rng.uniform(0.75, 1.25), rng.uniform(0.85, 1.15), rng.uniform(-0.05, 0.05)) for _ in range(64)))
    for name, jit in (("plain", None), ("jitter", (list(words), list(factors)))):
        for _ in range(3):
            preprocess_images(x, (512, 512), flips, jitter=jit)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            preprocess_images(x, (512, 512), flips, jitter=jit)
        torch.cuda.synchronize()
        print(f"{H}x{W} -> 512x512 bs=64 {name}: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms per batch", flush=True)
