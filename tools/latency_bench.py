#!/usr/bin/env python3
"""bs=1 inference latency (BASELINE configs[1]): Network eval forward + Decoder, 512x512."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bench import make_args  # noqa: E402
from structuredetector_amd.data import Decoder  # noqa: E402
from structuredetector_amd.model import Network  # noqa: E402

dev = torch.device("cuda")
args = make_args(dev)
net = Network(args, pretrained=False, raw_output=True).to(dev).eval()
dec = Decoder(args)
for B in (1, 2, 4, 8, 16):
    x = torch.randn(B, 3, 512, 512, device=dev)
    with torch.no_grad():
        for _ in range(3):
            out = net(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            out = net(x)
        torch.cuda.synchronize(); fwd = (time.perf_counter() - t0) / 10
        t0 = time.perf_counter()
        for _ in range(10):
            o = net(x)
            anns = dec({"anchor_hm": o[:, :2], "part_hm": o[:, 2:3], "offsets": o[:, 3:5], "embeddings": o[:, 5:7]})
        e2e = (time.perf_counter() - t0) / 10
        run = net.graphed(x)
        ref = net(x)
        assert torch.equal(run(x), ref), "graph replay must reproduce the eager forward bit for bit"
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            out = run(x)
        torch.cuda.synchronize(); gfwd = (time.perf_counter() - t0) / 20
        t0 = time.perf_counter()
        M, N = 2, 1
        for _ in range(20):
            o = run(x)
            anns = dec({"anchor_hm": o[:, :M], "part_hm": o[:, M:M + N], "offsets": o[:, M + N:M + N + 2], "embeddings": o[:, M + N + 2:]})
        ge2e = (time.perf_counter() - t0) / 20
    print(f"B={B}: hipGraph forward {gfwd * 1e3:.3f} ms, graph forward+decode e2e {ge2e * 1e3:.3f} ms ({ge2e / B * 1e3:.3f} ms/img)", flush=True)
    print(f"B={B}: forward {fwd * 1e3:.3f} ms ({B * 45.15 / fwd / 1e3:.1f} TFLOP/s), forward+decode e2e {e2e * 1e3:.3f} ms, per image {e2e / B * 1e3:.3f} ms", flush=True)
