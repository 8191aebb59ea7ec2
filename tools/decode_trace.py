#!/usr/bin/env python3
"""Stage timestamps of k_decode_fused (library built with EXTRA=-DSD_DECODE_TRACE into csrc/trace/; SDNET_ALLOW_ABLATION=1)."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bench import make_args  # noqa: E402
from structuredetector_amd import _lib as L  # noqa: E402
from structuredetector_amd.data import Decoder, Encode  # noqa: E402
from structuredetector_amd.data.synthetic import synthetic_batch  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
exact = bool(int(sys.argv[2])) if len(sys.argv) > 2 else False
dev = torch.device("cuda")
M, N, K, P, img = 2, 1, 20, 40, 512
args = make_args(dev, M, N, K, P)
enc, dec = Encode(args), Decoder(args)
gen = torch.Generator(device=dev).manual_seed(0)
tgt = enc.render(enc.plan(img, img, *synthetic_batch(np.random.default_rng(B), B, img, img, M, N)), dev)
hm = torch.cat([tgt["anchor_hm"], tgt["part_hm"]], 1).clamp(1e-4, 0.95)
head = torch.cat([torch.log(hm / (1 - hm)) + 0.05 * torch.randn(hm.shape, device=dev, generator=gen),
                  0.1 * torch.randn(B, 4, img // 4, img // 4, device=dev, generator=gen)], 1)
outs = {"anchor_hm": head[:, :M], "part_hm": head[:, M:M + N], "offsets": head[:, M + N:M + N + 2], "embeddings": head[:, M + N + 2:]}
lib = L.lib()
lib.sd_debug_read_trace.restype = C.c_int
lib.sd_debug_read_trace.argtypes = [C.c_void_p, C.c_int]
buf = (C.c_ulonglong * 8192)()
rows = []
for it in range(30):
    dec.decode_packed(outs, 0.5, 0.1, exact_topk=exact, fused=True)
    torch.cuda.synchronize()
    lib.sd_debug_read_trace(buf, 8192)
    t = np.frombuffer(buf, dtype=np.uint64).astype(np.int64)
    nt = min(B * 48, 1024)          # only the first 1024 tile blocks are traced
    tiles = t[:nt * 4].reshape(nt, 4)
    sel = t[4096:4096 + 8 * min(B, 64)].reshape(-1, 8)
    t0 = tiles[:, 0].min()
    if it >= 10:
        rows.append([tiles[:, 0].max() - t0, np.median(tiles[:, 1] - tiles[:, 0]), np.median(tiles[:, 2] - tiles[:, 1]),
                     np.median(tiles[:, 3] - tiles[:, 2]), tiles[:, 3].max() - t0, sel[:, 0].min() - t0] + [sel[-1, i] - t0 for i in range(1, 8)])
r = np.median(np.array(rows, np.float64), axis=0) * 0.01          # 100 MHz ticks -> us
names = ["last tile start", "tile: load+sigmoid", "tile: nms+key stores", "tile: drain+flag", "last tile end", "first selector start",
         "sel: records taken", "sel: prefix done", "sel: lists selected", "sel: zero slots filled", "(unused)", "(unused)", "sel: grouped (end)"]
print(f"B={B} exact={exact}: us since the first tile block started (medians over 20 runs; selector = last image's)")
for n, v in zip(names, r):
    print(f"   {n:24s} {v:7.2f}")
