#!/usr/bin/env python3
"""Where a batch-1 `evaluate` iteration spends its wall time: reader, upload + forward, decoder (with metadata), Evaluator.accumulate."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from tools.feed_bench import write_samples  # noqa: E402
from structuredetector_amd.data import CropDataset, Decoder  # noqa: E402
from structuredetector_amd.model import Evaluator, Network  # noqa: E402
from structuredetector_amd.utils import Arguments  # noqa: E402

root = Path("/tmp/sd_eval"); n = 64
labels = write_samples(root / "valid", n, 512)
args = Arguments().parse(["--valid_dir", str(root / "valid"), "--labels", str(labels), "-s", "stem"])
t0 = time.perf_counter()
net = Network(args).eval().to(args.device)
print(f"Network(): {time.perf_counter() - t0:.2f} s")
ev, dec, ds = Evaluator(args), Decoder(args), CropDataset(args, args.valid_dir)
for label, full in (("warm-up", True), ("full metadata dict", True), ("annotation + raw_parts only", False)):
    acc = [0.0] * 4
    for i in range(n):
        t0 = time.perf_counter(); image, ann = ds[i]
        t1 = time.perf_counter()
        with torch.no_grad():
            out = net(image[None].to(args.device))
        torch.cuda.synchronize(); t2 = time.perf_counter()
        data = dec(out, return_metadata=True, metadata_fields=None if full else ("annotation", "raw_parts"))
        t3 = time.perf_counter()
        ev.accumulate(data["annotation"][0], ann, data["raw_parts"][0], True, True)
        t4 = time.perf_counter()
        for k, v in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
            acc[k] += v / n * 1e3
    if label != "warm-up":
        print(f"{label:28s} ms per image: reader %.2f, upload + forward %.2f, decoder %.2f, Evaluator.accumulate %.2f" % tuple(acc))
