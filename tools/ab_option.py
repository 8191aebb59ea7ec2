#!/usr/bin/env python3
"""A/B of one sd_set_option switch on the same box, interleaved: bf16 eval forward (bs=64 512x512, and the 1024x1024 stress shape)
and the mixed-precision training step.  usage: ab_option.py <option> <value A> <value B>"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bench import make_args  # noqa: E402
from structuredetector_amd import _lib as L  # noqa: E402
from structuredetector_amd.model import Network  # noqa: E402

name, va, vb = sys.argv[1].encode(), int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda")


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def ab(label, fn):
    res = {va: [], vb: []}
    for _ in range(3):
        for v in (va, vb):
            L.check(L.lib().sd_set_option(name, v))
            res[v].append(timeit(fn))
    print(f"{label}: {name.decode()}={va}: {min(res[va]):.3f} ms   ={vb}: {min(res[vb]):.3f} ms   ({min(res[vb]) / min(res[va]) - 1:+.1%})", flush=True)


for (B, img, M, N, K, P) in (() if "--train-only" in sys.argv else ((64, 512, 2, 1, 20, 40), (16, 1024, 8, 8, 128, 512))):
    args = make_args(dev, M, N, K, P)
    net = Network(args, pretrained=False).to(dev).eval()
    net.bf16_inference = True
    x = torch.randn(B, 3, img, img, device=dev)
    with torch.no_grad():
        ab(f"bf16 eval forward B={B} {img}x{img}", lambda: net(x))
    del net, x

from structuredetector_amd.data import Encode  # noqa: E402
from structuredetector_amd.data.synthetic import synthetic_batch  # noqa: E402
from structuredetector_amd.model.trainer import TrainStep  # noqa: E402

args = make_args(dev)
args.use_amp = True
net = Network(args, pretrained=False).to(dev).train()
enc = Encode(args)
step = TrainStep(net, args)
images = torch.randn(64, 3, 512, 512, device=dev)
plan = enc.upload(enc.plan(512, 512, *synthetic_batch(np.random.default_rng(0), 64, 512, 512, 2, 1)))
ab("mixed-precision train step B=64", lambda: step(images, enc.render_device(plan)))
step.amp = False
ab("fp32 train step B=64", lambda: step(images, enc.render_device(plan)))
