#!/usr/bin/env python3
"""How much a training step stretches beside a simulated 8-rank all-reduce, over the collective's workgroup count and modelled bus
bandwidth (`TrainStep.attach_sim`, csrc/sd_commsim.hip): same-process alternating A/B at bs = 64, 512 x 512, fp32 and `--amp`.
usage: comm_sim_sweep.py [--steps 4]"""
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

steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 4
dev = torch.device("cuda")
M, N, K, P, B, img = 2, 1, 20, 40, 64, 512
args = make_args(dev, M, N, K, P)
torch.manual_seed(926354916)
net = Network(args, pretrained=False).to(dev).train()
step = TrainStep(net, args)
enc = Encode(args)
gen = torch.Generator(device=dev).manual_seed(1)
images = torch.randn(B, 3, img, img, device=dev, generator=gen)
tgt = enc.render(enc.plan(img, img, *synthetic_batch(np.random.default_rng(1), B, img, img, M, N)), dev)


def timed(n, warm=1):
    for _ in range(warm):
        step(images, tgt)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n):
        step(images, tgt)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


for amp in (False, True):
    step.amp = amp
    timed(2)
    print(f"{'mixed-precision' if amp else 'fp32'} step, bs {B}, {img} x {img}; 153 MB moved per step in five launches; ms per step (best of 3 alternating rounds)")
    for wgs, gbps, plan in ((16, 200.0, 5), (32, 200.0, 5), (64, 200.0, 5), (32, 100.0, 5), (32, 400.0, 5), (32, 200.0, 3), (32, 200.0, 2), (32, 200.0, 1), (32, 100.0, 2)):
        base, sim = [], []
        step.set_bucket_plan(plan)
        for _ in range(3):
            step.detach_sim(); base.append(timed(steps))
            step.attach_sim(ranks=8, workgroups=wgs, gbps=gbps); sim.append(timed(steps))
        step.detach_sim()
        step.set_bucket_plan(5)
        b_, s_ = min(base), min(sim)
        print(f"   {plan} launches, {wgs:3d} workgroups at {gbps:5.0f} GB/s ({153.0 / gbps:5.2f} ms of transfer): {b_:7.3f} -> {s_:7.3f}  exposed {s_ - b_:+.3f} ms ({100 * (s_ / b_ - 1):+.2f} %)")
