#!/usr/bin/env python3
"""Decoder on BASELINE.json configs[4] (stress): 1024x1024, 8 labels / 8 parts, K=128, P=512, bs=16, dense scenes
(64-96 objects/img) -- realistic planted heads and the worst case (N(0,1) logits: every NMS survivor is a candidate)."""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bench import make_args  # noqa: E402
from structuredetector_amd import _lib as L  # noqa: E402
from structuredetector_amd.data import Decoder, Encode  # noqa: E402
from structuredetector_amd.data.synthetic import synthetic_batch  # noqa: E402

dev = torch.device("cuda")
M, N, K, P, img, B = 8, 8, 128, 512, 1024, 16
args = make_args(dev, M, N, K, P)
enc, dec = Encode(args), Decoder(args)
gen = torch.Generator(device=dev).manual_seed(0)
tgt = enc.render(enc.plan(img, img, *synthetic_batch(np.random.default_rng(7), B, img, img, M, N, n_min=64, n_max=96)), dev)
hm = torch.cat([tgt["anchor_hm"], tgt["part_hm"]], 1).clamp(1e-4, 0.95)
h = img // 4
heads = {
    "dense scenes": torch.cat([torch.log(hm / (1 - hm)) + 0.05 * torch.randn(hm.shape, device=dev, generator=gen),
                               0.1 * torch.randn(B, 4, h, h, device=dev, generator=gen)], 1),
    "N(0,1) logits": torch.randn(B, M + N + 4, h, h, device=dev, generator=gen),
}
bytes_per_img = (M + N) * h * h * 4 + (2 * K + 4 * P) * 4 + (K * 4 + P * 7) * 4
for name, head in heads.items():
    outs = {"anchor_hm": head[:, :M], "part_hm": head[:, M:M + N], "offsets": head[:, M + N:M + N + 2], "embeddings": head[:, M + N + 2:]}
    res = {}
    for exact in (True, False):
        for _ in range(3):
            dec.decode_packed(outs, 0.5, 0.1, exact_topk=exact)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            dec.decode_packed(outs, 0.5, 0.1, exact_topk=exact)
        torch.cuda.synchronize()
        res[exact] = (time.perf_counter() - t0) / 20
    print(f"{name}: exact top-k {res[True] / B * 1e6:.2f} us/img ({B * bytes_per_img / res[True] / 1e9:.0f} GB/s), annotations-only "
          f"{res[False] / B * 1e6:.2f} us/img ({B * bytes_per_img / res[False] / 1e9:.0f} GB/s)", flush=True)
