#!/usr/bin/env python3
"""Decoder kernels alone for rocprofv3 --kernel-trace --stats: usage decode_prof.py <B> <fused 0|1> <exact 0|1> [stress]"""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bench import make_args  # noqa: E402
from structuredetector_amd.data import Decoder, Encode  # noqa: E402
from structuredetector_amd.data.synthetic import synthetic_batch  # noqa: E402

B, fused, exact = int(sys.argv[1]), bool(int(sys.argv[2])), bool(int(sys.argv[3]))
stress = len(sys.argv) > 4
dev = torch.device("cuda")
M, N, K, P, img = (8, 8, 128, 512, 1024) if stress else (2, 1, 20, 40, 512)
args = make_args(dev, M, N, K, P)
enc, dec = Encode(args), Decoder(args)
gen = torch.Generator(device=dev).manual_seed(0)
tgt = enc.render(enc.plan(img, img, *synthetic_batch(np.random.default_rng(B), B, img, img, M, N, *((64, 96) if stress else (6, 12)))), dev)
hm = torch.cat([tgt["anchor_hm"], tgt["part_hm"]], 1).clamp(1e-4, 0.95)
head = torch.cat([torch.log(hm / (1 - hm)) + 0.05 * torch.randn(hm.shape, device=dev, generator=gen),
                  0.1 * torch.randn(B, 4, img // 4, img // 4, device=dev, generator=gen)], 1)
outs = {"anchor_hm": head[:, :M], "part_hm": head[:, M:M + N], "offsets": head[:, M + N:M + N + 2], "embeddings": head[:, M + N + 2:]}
import os  # noqa: E402
from structuredetector_amd import _lib as L  # noqa: E402
if os.environ.get("SD_MAP_FROM"):
    L.check(L.lib().sd_decode_set_option(b"map_parallel_from", int(os.environ["SD_MAP_FROM"])))
if os.environ.get("SD_MAP_STREAM"):
    L.check(L.lib().sd_decode_set_option(b"map_stream", int(os.environ["SD_MAP_STREAM"])))
for env, key in (("SD_MAP_SPLIT", b"map_split"), ("SD_MAP_RG", b"map_rank_group"), ("SD_MAP_ROWS11", b"map_rows11"), ("SD_MAP_HALF", b"map_half"), ("SD_MAP_WAVES3", b"map_waves3")):
    if os.environ.get(env):
        L.check(L.lib().sd_decode_set_option(key, int(os.environ[env])))
if os.environ.get("SD_MAP_TH"):
    L.check(L.lib().sd_decode_set_option(b"map_tile_height", int(os.environ["SD_MAP_TH"])))
for _ in range(100):
    dec.decode_packed(outs, 0.5, 0.1, exact_topk=exact, fused=fused)
    torch.cuda.synchronize()
