#!/usr/bin/env python3
"""Same-box A/B of _Engine.overlap_wgrad (weight gradients on a second stream next to the data-gradient chain), fp32 and mixed precision."""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bench import make_args  # noqa: E402
from structuredetector_amd.data import Encode  # noqa: E402
from structuredetector_amd.data.synthetic import synthetic_batch  # noqa: E402
from structuredetector_amd.model import Network  # noqa: E402
from structuredetector_amd.model.trainer import TrainStep  # noqa: E402

dev = torch.device("cuda")
args = make_args(dev)
args.use_amp = False
net = Network(args, pretrained=False).to(dev).train()
enc = Encode(args)
step = TrainStep(net, args)
images = torch.randn(64, 3, 512, 512, device=dev)
plan = enc.upload(enc.plan(512, 512, *synthetic_batch(np.random.default_rng(0), 64, 512, 512, 2, 1)))


def timeit(n=10):
    for _ in range(3):
        step(images, enc.render_device(plan))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        step(images, enc.render_device(plan))
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for amp in (False, True):
    step.amp = amp
    res = {False: [], True: []}
    for _ in range(3):
        for ov in (False, True):
            net._engine.overlap_wgrad = ov
            res[ov].append(timeit())
    print(f"{'mixed-precision' if amp else 'fp32'} step: overlap_wgrad off {min(res[False]):.3f} ms, on {min(res[True]):.3f} ms ({min(res[True]) / min(res[False]) - 1:+.1%})", flush=True)
