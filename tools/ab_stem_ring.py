#!/usr/bin/env python3
"""Same-process A/B of the mixed-precision step's stem backward: bf16 stem gradient + row-ring weight gradient (sd_maxpool_bn_relu_bwd_bf16_dx16 +
sd_conv2d_stem_wgrad_bf16) against the fp32 gradient + k_stem_wgrad_bf16, interleaved; bs = 64, 512x512."""
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
args = make_args(dev); args.use_amp = True
torch.manual_seed(0)
net = Network(args, pretrained=False).to(dev).train()
step = TrainStep(net, args)
enc = Encode(args)
x = torch.randn(64, 3, 512, 512, device=dev)
plan = enc.upload(enc.plan(512, 512, *synthetic_batch(np.random.default_rng(0), 64, 512, 512, 2, 1)))


def timeit(n=20):
    for _ in range(3):
        step(x, enc.render_device(plan))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        loss = step(x, enc.render_device(plan))
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, float(loss[0])


res = {True: [], False: []}
for _ in range(4):
    for v in (False, True):
        net._engine.stem_ring = v
        res[v].append(timeit())
a, b = min(t for t, _ in res[False]), min(t for t, _ in res[True])
print(f"mixed-precision step bs=64 512x512: fp32 stem gradient {a:.3f} ms, bf16 stem gradient + row ring {b:.3f} ms ({b / a - 1:+.1%})   all: "
      + " ".join(f"{p[0]:.2f}/{q[0]:.2f}" for p, q in zip(res[False], res[True])) + f"   last losses {res[False][-1][1]:.5f} / {res[True][-1][1]:.5f}")
