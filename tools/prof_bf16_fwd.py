#!/usr/bin/env python3
"""bf16 eval forward only (bs=64, 512x512) for a per-kernel rocprofv3 breakdown."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bench import make_args
from structuredetector_amd.model import Network
dev = torch.device("cuda")
args = make_args(dev)
net = Network(args, pretrained=False).to(dev).eval()
net.bf16_inference = True
x = torch.randn(64, 3, 512, 512, device=dev)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 5
with torch.no_grad():
    for _ in range(N):
        net(x)
torch.cuda.synchronize()
print("forwards", N)
