#!/usr/bin/env python3
"""Same-process A/B of two values of sd_set_option("wgrad_bf16_ring") on the mixed-precision training step (bs = 64, 512x512), interleaved.
usage: ab_wgrad_ring.py [<value A> <value B>]   (default 0 3)"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bench import make_args  # noqa: E402
from structuredetector_amd import _lib as L  # noqa: E402
from structuredetector_amd.data import Encode  # noqa: E402
from structuredetector_amd.data.synthetic import synthetic_batch  # noqa: E402
from structuredetector_amd.model import Network  # noqa: E402
from structuredetector_amd.model.trainer import TrainStep  # noqa: E402

dev = torch.device("cuda")
args = make_args(dev); args.use_amp = True
torch.manual_seed(0)
net = Network(args, pretrained=False).to(dev).train()
step = TrainStep(net, args)
enc = Encode(args)
x = torch.randn(64, 3, 512, 512, device=dev)
plan = enc.upload(enc.plan(512, 512, *synthetic_batch(np.random.default_rng(0), 64, 512, 512, 2, 1)))
va, vb = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (0, 3)      # wgrad_bf16_ring values: 0 first form, 2 .. 4 row ring, 5 two-group row ring
res = {va: [], vb: []}
for _ in range(4):
    for v in (va, vb):
        L.check(L.lib().sd_set_option(b"wgrad_bf16_ring", v))
        for _ in range(3):
            step(x, enc.render_device(plan))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            step(x, enc.render_device(plan))
        torch.cuda.synchronize()
        res[v].append((time.perf_counter() - t0) / 20 * 1e3)
print(f"mixed-precision step: wgrad_bf16_ring={va} {min(res[va]):.3f} ms, ={vb} {min(res[vb]):.3f} ms ({min(res[vb]) / min(res[va]) - 1:+.1%})   all: "
      + " ".join(f"{a:.3f}/{b:.3f}" for a, b in zip(res[va], res[vb])))
