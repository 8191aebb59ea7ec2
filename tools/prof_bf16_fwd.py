#!/usr/bin/env python3
"""bf16 eval forward only (bs=64, 512x512; `<n> stress`: bs=16, 1024x1024, 8 labels / 8 parts) for a per-kernel rocprofv3 breakdown."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bench import make_args
from structuredetector_amd.model import Network
dev = torch.device("cuda")
stress = "stress" in sys.argv[2:]        # configs[4]: 1024x1024, 8 labels / 8 parts, bs=16
args = make_args(dev, 8, 8, 128, 512) if stress else make_args(dev)
net = Network(args, pretrained=False).to(dev).eval()
net.bf16_inference = True
x = torch.randn(16, 3, 1024, 1024, device=dev) if stress else torch.randn(64, 3, 512, 512, device=dev)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 5
with torch.no_grad():
    for _ in range(N):
        net(x)
torch.cuda.synchronize()
print("forwards", N)
