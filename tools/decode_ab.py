#!/usr/bin/env python3
"""The bench line's three decoder figures (device time of one fused decode at bs=64, exact top-k, end-to-end bs=1 with the Python objects),
median of 5 rounds of 50 calls -- run once per library (SDNET_HIP_LIB=...) on the same box for an A/B."""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bench import make_args  # noqa: E402
from structuredetector_amd.data import Decoder, Encode  # noqa: E402
from structuredetector_amd.data.synthetic import synthetic_batch  # noqa: E402

dev = torch.device("cuda")
M, N, K, P, img, B = 2, 1, 20, 40, 512, 64
args = make_args(dev, M, N, K, P)
enc, dec = Encode(args), Decoder(args)
gen = torch.Generator(device=dev).manual_seed(0)
tgt = enc.render(enc.plan(img, img, *synthetic_batch(np.random.default_rng(B), B, img, img, M, N)), dev)
hm = torch.cat([tgt["anchor_hm"], tgt["part_hm"]], 1).clamp(1e-4, 0.95)
head = torch.cat([torch.log(hm / (1 - hm)) + 0.05 * torch.randn(hm.shape, device=dev, generator=gen),
                  0.1 * torch.randn(B, 4, img // 4, img // 4, device=dev, generator=gen)], 1)
outs = {"anchor_hm": head[:, :M], "part_hm": head[:, M:M + N], "offsets": head[:, M + N:M + N + 2], "embeddings": head[:, M + N + 2:]}
one = {k: v[:1] for k, v in outs.items()}


def timed(fn, n=50):
    for _ in range(5):
        fn()
    rs = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize(); rs.append((time.perf_counter() - t0) / n)
    return sorted(rs)[2] * 1e6


print("bs=64 device %.1f us   exact top-k %.1f us   bs=1 end to end %.1f us" % (
    timed(lambda: dec.decode_packed(outs, 0.5, 0.1, exact_topk=False)), timed(lambda: dec.decode_packed(outs, 0.5, 0.1, exact_topk=True)),
    timed(lambda: dec(one))))
