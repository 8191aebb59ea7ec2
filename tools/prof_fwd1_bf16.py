#!/usr/bin/env python3
"""bf16-backbone eval forward at bs=1 (512x512) for a per-kernel rocprofv3 breakdown (BASELINE configs[1] on the bf16 backbone)."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bench import make_args
from structuredetector_amd.model import Network
dev = torch.device("cuda")
net = Network(make_args(dev), pretrained=False).to(dev).eval()
net.bf16_inference = True
x = torch.randn(1, 3, 512, 512, device=dev)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
with torch.no_grad():
    for _ in range(N):
        net(x)
torch.cuda.synchronize()
print("forwards", N)
