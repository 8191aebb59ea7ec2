#!/usr/bin/env python3
"""Training steps only (no eval / decode extras) for a clean rocprofv3 per-step kernel breakdown."""
import sys
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bench import make_args
from structuredetector_amd.data import Encode
from structuredetector_amd.data.synthetic import synthetic_batch
from structuredetector_amd.model import Network
from structuredetector_amd.model.trainer import TrainStep
dev = torch.device("cuda")
args = make_args(dev)
args.use_amp = len(sys.argv) > 2 and sys.argv[2] == "amp"      # prof_train.py <steps> amp: the mixed-precision step
net = Network(args, pretrained=False).to(dev).train()
step = TrainStep(net, args)
enc = Encode(args)
B, img, STEPS = 64, 512, int(sys.argv[1]) if len(sys.argv) > 1 else 4
x = torch.randn(B, 3, img, img, device=dev)
plan = enc.upload(enc.plan(img, img, *synthetic_batch(np.random.default_rng(0), B, img, img, 2, 1)))
for _ in range(STEPS):
    step(x, enc.render_device(plan))
torch.cuda.synchronize()
print("steps", STEPS)
