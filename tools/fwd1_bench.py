#!/usr/bin/env python3
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bench import make_args
from structuredetector_amd.model import Network
dev = torch.device("cuda")
net = Network(make_args(dev), pretrained=False, raw_output=True).to(dev).eval()
x = torch.randn(int(sys.argv[1]) if len(sys.argv) > 1 else 1, 3, 512, 512, device=dev)
with torch.no_grad():
    for _ in range(20):
        net(x)
torch.cuda.synchronize()
