#!/usr/bin/env python3
"""Same-process A/B of two values of one sd_set_option switch on the fp32 (default) or mixed-precision (`amp`) training step, bs = 64, 512x512,
interleaved.  usage: ab_set_option.py <option> <value A> <value B> [amp]"""
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

name, va, vb = sys.argv[1].encode(), int(sys.argv[2]), int(sys.argv[3])
amp = "amp" in sys.argv[4:]
dev = torch.device("cuda")
args = make_args(dev); args.use_amp = amp
torch.manual_seed(0)
net = Network(args, pretrained=False).to(dev).train()
step = TrainStep(net, args)
enc = Encode(args)
x = torch.randn(64, 3, 512, 512, device=dev)
plan = enc.upload(enc.plan(512, 512, *synthetic_batch(np.random.default_rng(0), 64, 512, 512, 2, 1)))
res = {va: [], vb: []}
n = 20 if amp else 8
for _ in range(4):
    for v in (va, vb):
        L.check(L.lib().sd_set_option(name, v))
        for _ in range(2):
            step(x, enc.render_device(plan))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            step(x, enc.render_device(plan))
        torch.cuda.synchronize()
        res[v].append((time.perf_counter() - t0) / n * 1e3)
print(f"{'mixed-precision' if amp else 'fp32'} step: {name.decode()}={va} {min(res[va]):.3f} ms, ={vb} {min(res[vb]):.3f} ms ({min(res[vb]) / min(res[va]) - 1:+.2%})   all: "
      + " ".join(f"{a:.3f}/{b:.3f}" for a, b in zip(res[va], res[vb])))
