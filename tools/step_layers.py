#!/usr/bin/env python3
"""Per-launch listing of the conv kernels of ONE fp32 training step at the bench workload (bs=64, 512x512): phase, geometry, kernel,
time (stream events around the launch), TFLOP/s, the launch's algorithmic HBM bytes and the time those bytes take at 8 TB/s -- which
launches are MFMA-bound and which sit on the HBM roofline.
usage: step_layers.py [--amp] [--batch 64] [--steps 3]"""
import argparse
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bench import make_args  # noqa: E402
from structuredetector_amd.data.synthetic import synthetic_batch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--amp", action="store_true"); ap.add_argument("--batch", type=int, default=64); ap.add_argument("--steps", type=int, default=3)
    a = ap.parse_args()
    import numpy as np
    from structuredetector_amd.data import Encode
    from structuredetector_amd.model import Network
    from structuredetector_amd.model.trainer import TrainStep
    dev = torch.device("cuda")
    args = make_args(dev)
    args.use_amp = bool(a.amp)
    torch.manual_seed(926354916)
    net = Network(args, pretrained=False).to(dev).train()
    step = TrainStep(net, args)
    enc = Encode(args)
    rng = np.random.default_rng(0)
    images = torch.randn(a.batch, 3, 512, 512, device=dev)
    plan = enc.upload(enc.plan(512, 512, *synthetic_batch(rng, a.batch, 512, 512, 2, 1)))
    targets = enc.render_device(plan)
    eng = net._engine
    for _ in range(2):
        step(images, targets)
    acc = {}
    for s in range(a.steps):
        eng.prof, eng.prof_shapes = [], []
        step(images, targets)
        torch.cuda.synchronize()
        for i, ((kind, flops, e0, e1, phase), shp) in enumerate(zip(eng.prof, eng.prof_shapes)):
            acc.setdefault(i, [kind, flops, phase, shp, 0.0])[4] += e0.elapsed_time(e1) / a.steps
    eng.prof = eng.prof_shapes = None
    es = 2 if a.amp else 4
    tot = 0.0
    print(f"{'phase':6s} {'B,Hi,Cin->Cout kxk/s':28s} {'kernel':34s} {'us':>8s} {'TF/s':>7s} {'MB':>7s} {'us@8TB/s':>8s} {'mfma us':>8s}")
    for i in sorted(acc):
        kind, flops, phase, (B, Hi, Wi, Cin, Cout, R, stride, Ho, Wo), ms = acc[i]
        x_b, y_b, w_b = B * Hi * Wi * Cin * es, B * Ho * Wo * Cout * es, Cout * Cin * R * R * es
        nbytes = x_b + y_b + w_b                                   # every operand once (a residual epilogue adds one more output-sized read)
        peak = 2500e12 if a.amp else 157.3e12
        print(f"{phase:6s} {f'{B},{Hi},{Cin}->{Cout} {R}x{R}/{stride}':28s} {kind[:34]:34s} {ms * 1e3:8.1f} {flops / ms / 1e9:7.1f} {nbytes / 1e6:7.0f} "
              f"{nbytes / 8e12 * 1e6:8.1f} {flops / peak * 1e6:8.1f}")
        tot += ms
    print(f"conv launches {len(acc)}  sum {tot:.2f} ms")


if __name__ == "__main__":
    main()
