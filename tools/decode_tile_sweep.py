#!/usr/bin/env python3
"""sd_decode_fused on 64x16 against 64x32 NMS tiles over batch sizes (sets sd_decode_set_option("tall_tiles_from"))."""
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
M, N, K, P, img = 2, 1, 20, 40, 512
args = make_args(dev, M, N, K, P)
enc, dec = Encode(args), Decoder(args)
gen = torch.Generator(device=dev).manual_seed(0)
for B in (1, 4, 8, 16, 24, 32, 48, 64, 96, 128):
    tgt = enc.render(enc.plan(img, img, *synthetic_batch(np.random.default_rng(B), B, img, img, M, N)), dev)
    hm = torch.cat([tgt["anchor_hm"], tgt["part_hm"]], 1).clamp(1e-4, 0.95)
    head = torch.cat([torch.log(hm / (1 - hm)) + 0.05 * torch.randn(hm.shape, device=dev, generator=gen),
                      0.1 * torch.randn(B, 4, img // 4, img // 4, device=dev, generator=gen)], 1)
    outs = {"anchor_hm": head[:, :M], "part_hm": head[:, M:M + N], "offsets": head[:, M + N:M + N + 2], "embeddings": head[:, M + N + 2:]}
    res = []
    for tall_from in (1 << 30, 1):
        L.check(L.lib().sd_decode_set_option(b"tall_tiles_from", tall_from))
        best = 1e9
        for _ in range(3):
            for _ in range(20):
                dec.decode_packed(outs, 0.5, 0.1, exact_topk=False, fused=True)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(400):
                dec.decode_packed(outs, 0.5, 0.1, exact_topk=False, fused=True)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 400)
        res.append(best * 1e6)
    print(f"B={B:4d} tile blocks(64x16)={B * 3 * 16:5d}: 64x16 {res[0]:6.2f} us  64x32 {res[1]:6.2f} us  ({res[1] / res[0] - 1:+.1%})", flush=True)
