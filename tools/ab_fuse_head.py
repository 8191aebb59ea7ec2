#!/usr/bin/env python3
"""Same-process A/B of the fused head (sd_conv2d_fwd_bf16_head: `up4.conv` + head in one launch) against the two launches, interleaved:
bf16 eval forward bs = 64 512x512 and the stress shape (bs = 16, 1024x1024, 8 + 8 maps)."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bench import make_args  # noqa: E402
from structuredetector_amd.model import Network  # noqa: E402

dev = torch.device("cuda")


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for (B, img, M, N, K, P) in ((64, 512, 2, 1, 20, 40), (16, 1024, 8, 8, 128, 512)):
    args = make_args(dev, M, N, K, P); args.use_amp = True
    net = Network(args, pretrained=False).to(dev).eval()
    net.bf16_inference = True
    x = torch.randn(B, 3, img, img, device=dev)
    res = {True: [], False: []}
    with torch.no_grad():
        for _ in range(4):
            for v in (False, True):
                net._engine.fuse_head = v
                res[v].append(timeit(lambda: net(x)))
    print(f"bf16 forward bs={B} {img}x{img} {M}+{N} maps: head as its own launch {min(res[False]):.3f} ms, fused {min(res[True]):.3f} ms "
          f"({min(res[True]) / min(res[False]) - 1:+.1%})   all: " + " ".join(f"{a:.3f}/{b:.3f}" for a, b in zip(res[False], res[True])), flush=True)
