#!/usr/bin/env python3
"""fp32 eval forward at bs=1 (512x512) for a per-kernel rocprofv3 breakdown (BASELINE configs[1]).
usage: prof_fwd1.py [forwards=20] [graph]   -- `graph`: the same forwards replayed as one hipGraph (Network.graphed)"""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bench import make_args
from structuredetector_amd.model import Network
dev = torch.device("cuda")
net = Network(make_args(dev), pretrained=False).to(dev).eval()
x = torch.randn(1, 3, 512, 512, device=dev)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
graph = len(sys.argv) > 2 and sys.argv[2] == "graph"
with torch.no_grad():
    run = net.graphed(x) if graph else net
    for _ in range(N):
        run(x)
torch.cuda.synchronize()
print("forwards", N, "graph" if graph else "eager")
