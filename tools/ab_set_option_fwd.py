#!/usr/bin/env python3
"""Same-process A/B of two values of one sd_set_option switch on the bf16 eval forward (bs = 64 at 512x512 and the stress shape bs = 16 at
1024x1024, 8 + 8 maps), interleaved.  usage: ab_set_option_fwd.py <option> <value A> <value B>"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bench import make_args  # noqa: E402
from structuredetector_amd import _lib as L  # noqa: E402
from structuredetector_amd.model import Network  # noqa: E402

name, va, vb = sys.argv[1].encode(), int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda")
lib = L.lib()


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
    res = {va: [], vb: []}
    with torch.no_grad():
        for _ in range(4):
            for v in (va, vb):
                L.check(lib.sd_set_option(name, v))
                res[v].append(timeit(lambda: net(x)))
    a, b = min(res[va]), min(res[vb])
    print(f"bf16 forward bs={B} {img}x{img}: {name.decode()}={va} {a:.3f} ms, ={vb} {b:.3f} ms ({b / a - 1:+.1%})   all: "
          + " ".join(f"{p:.3f}/{q:.3f}" for p, q in zip(res[va], res[vb])), flush=True)
