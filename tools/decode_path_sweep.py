#!/usr/bin/env python3
"""sd_decode (two-launch decoder, the path taken where sd_decode_fused is not recommended): launch pair with one selector block per
image against the map-parallel path, over geometries and batch sizes; wall time per call over 30 back-to-back calls."""
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
lib = L.lib()
for (M, N, K, P, img, dense) in ((8, 8, 128, 512, 1024, True), (2, 1, 20, 40, 512, False), (4, 4, 64, 128, 768, True)):
    for B in (1, 2, 4, 8, 16, 64):
        if img == 1024 and B > 16:
            continue
        args = make_args(dev, M, N, K, P)
        enc, dec = Encode(args), Decoder(args)
        gen = torch.Generator(device=dev).manual_seed(0)
        tgt = enc.render(enc.plan(img, img, *synthetic_batch(np.random.default_rng(7), B, img, img, M, N, *((64, 96) if dense else (6, 12)))), dev)
        hm = torch.cat([tgt["anchor_hm"], tgt["part_hm"]], 1).clamp(1e-4, 0.95)
        h = img // 4
        head = torch.cat([torch.log(hm / (1 - hm)) + 0.05 * torch.randn(hm.shape, device=dev, generator=gen),
                          0.1 * torch.randn(B, 4, h, h, device=dev, generator=gen)], 1)
        outs = {"anchor_hm": head[:, :M], "part_hm": head[:, M:M + N], "offsets": head[:, M + N:M + N + 2], "embeddings": head[:, M + N + 2:]}
        row = []
        for exact in (True, False):
            for frm in (1 << 30, 1):
                L.check(lib.sd_decode_set_option(b"map_parallel_from", frm))
                for _ in range(5):
                    dec.decode_packed(outs, 0.5, 0.1, exact_topk=exact, fused=False)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(30):
                    dec.decode_packed(outs, 0.5, 0.1, exact_topk=exact, fused=False)
                torch.cuda.synchronize()
                row.append((time.perf_counter() - t0) / 30 * 1e6)
        fused = ""
        if lib.sd_decode_fused_supported(B, M, N, h, h, K, P):
            for _ in range(5):
                dec.decode_packed(outs, 0.5, 0.1, exact_topk=False, fused=True)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(30):
                dec.decode_packed(outs, 0.5, 0.1, exact_topk=False, fused=True)
            torch.cuda.synchronize()
            fused = f"  | one launch (fast) {(time.perf_counter() - t0) / 30 * 1e6:7.1f}"
        print(f"{img}x{img} {M}+{N} maps K={K} P={P} B={B:3d} ({B * (M + N) * (h // 64 or 1) * ((h + 15) // 16):6d} tile blocks): exact pair {row[0]:7.1f} / map {row[1]:7.1f} us"
              f"   fast pair {row[2]:7.1f} / map {row[3]:7.1f} us{fused}", flush=True)
L.check(lib.sd_decode_set_option(b"map_parallel_from", -1))
