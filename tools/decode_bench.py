#!/usr/bin/env python3
"""Decode-only micro-benchmark (run under rocprofv3 --kernel-trace --stats to split nms / select)."""
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
M, N, K, P, img = 2, 1, 20, 40, 512
args = make_args(dev, M, N, K, P)
enc, dec = Encode(args), Decoder(args)
gen = torch.Generator(device=dev).manual_seed(0)
for B in (64, 1):
    tgt = enc.render(enc.plan(img, img, *synthetic_batch(np.random.default_rng(B), B, img, img, M, N)), dev)
    hm = torch.cat([tgt["anchor_hm"], tgt["part_hm"]], 1).clamp(1e-4, 0.95)
    head = torch.cat([torch.log(hm / (1 - hm)) + 0.05 * torch.randn(hm.shape, device=dev, generator=gen),
                      0.1 * torch.randn(B, 4, img // 4, img // 4, device=dev, generator=gen)], 1)
    outs = {"anchor_hm": head[:, :M], "part_hm": head[:, M:M + N], "offsets": head[:, M + N:M + N + 2], "embeddings": head[:, M + N + 2:]}
    res = {}
    for exact in (True, False):
        for _ in range(5):
            packed, _ = dec.decode_packed(outs, 0.5, 0.1, exact_topk=exact)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50):
            packed, _ = dec.decode_packed(outs, 0.5, 0.1, exact_topk=exact)
        torch.cuda.synchronize()
        res[exact] = (time.perf_counter() - t0) / 50
    dt = res[True]
    print(f"B={B}: exact top-k {res[True] * 1e6:.1f} us/batch, annotations-only mode {res[False] * 1e6:.1f} us/batch = {res[False] / B * 1e6:.2f} us/img")
    packed, _ = dec.decode_packed(outs, 0.5, 0.1, exact_topk=True)
    import ctypes
    from structuredetector_amd import _lib as L
    cnt = L.workspace(1, dev)[:B * 2 * 128].view(torch.int32).cpu().numpy().reshape(B, 2, 32)[:, :, 0]
    print(f"B={B}: device {dt * 1e6:.1f} us/batch = {dt / B * 1e6:.2f} us/img; candidates per image (anchor, part): mean {cnt.mean(0)}, max {cnt.max(0)}")
