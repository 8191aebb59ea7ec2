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
    for fused in (False, True):
        res = {}
        for exact in (True, False):
            for _ in range(5):
                packed, _ = dec.decode_packed(outs, 0.5, 0.1, exact_topk=exact, fused=fused)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(200):
                packed, _ = dec.decode_packed(outs, 0.5, 0.1, exact_topk=exact, fused=fused)
            torch.cuda.synchronize()
            res[exact] = (time.perf_counter() - t0) / 200
        print(f"B={B} {'fused (1 launch)' if fused else 'two-launch     '}: exact top-k {res[True] * 1e6:6.1f} us/batch, annotations-only "
              f"{res[False] * 1e6:6.1f} us/batch = {res[False] / B * 1e6:.3f} us/img = {B * 199008 / res[False] / 1e9:.0f} GB/s")
    if B == 1:
        import cProfile, pstats, io
        for _ in range(20):
            dec(outs)
        t0 = time.perf_counter()
        for _ in range(200):
            dec(outs)
        print(f"B=1 end to end (launch + D2H + host objects): {(time.perf_counter() - t0) / 200 * 1e6:.1f} us")
        t0 = time.perf_counter()
        for _ in range(200):
            packed, _ = dec.decode_packed(outs, 0.5, 0.1, exact_topk=False)
        torch.cuda.synchronize()
        print(f"B=1 decode_packed host time: {(time.perf_counter() - t0) / 200 * 1e6:.1f} us")
        t0 = time.perf_counter()
        for _ in range(200):
            host = packed.cpu()
        print(f"B=1 packed.cpu(): {(time.perf_counter() - t0) / 200 * 1e6:.1f} us")
        pr = cProfile.Profile(); pr.enable()
        for _ in range(500):
            dec(outs)
        pr.disable()
        st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(14); print(st.getvalue()[:3500])
