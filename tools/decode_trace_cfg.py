#!/usr/bin/env python3
"""Stage timestamps inside the map-parallel decoder kernels at the cfg shape (bs = 64, 2 + 1 maps of 128 x 128, K = 20, P = 40).
Library built with `make -C structuredetector_amd/csrc SUFFIX=_trace EXTRA=-DSD_DECODE_TRACE`; run with
SDNET_HIP_LIB=structuredetector_amd/csrc/libsdnet_hip_trace.so SDNET_ALLOW_ABLATION=1 [SD_MAP_SPLIT=n] [SD_MAP_RG=0|1]."""
import ctypes as C
import os
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bench import make_args  # noqa: E402
from structuredetector_amd import _lib as L  # noqa: E402
from structuredetector_amd.data import Decoder, Encode  # noqa: E402
from structuredetector_amd.data.synthetic import synthetic_batch  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
exact = bool(int(sys.argv[2])) if len(sys.argv) > 2 else False
dev = torch.device("cuda")
stress = len(sys.argv) > 3          # configs[4]-like scene: 1024 x 1024, 8 + 8 maps, K = 128, P = 512 (usage: 16 1 stress)
M, N, K, P, img = (8, 8, 128, 512, 1024) if stress else (2, 1, 20, 40, 512)
args = make_args(dev, M, N, K, P)
enc, dec = Encode(args), Decoder(args)
gen = torch.Generator(device=dev).manual_seed(0)
tgt = enc.render(enc.plan(img, img, *synthetic_batch(np.random.default_rng(B), B, img, img, M, N, *((64, 96) if stress else (6, 12)))), dev)
hm = torch.cat([tgt["anchor_hm"], tgt["part_hm"]], 1).clamp(1e-4, 0.95)
head = torch.cat([torch.log(hm / (1 - hm)) + 0.05 * torch.randn(hm.shape, device=dev, generator=gen),
                  0.1 * torch.randn(B, 4, img // 4, img // 4, device=dev, generator=gen)], 1)
outs = {"anchor_hm": head[:, :M], "part_hm": head[:, M:M + N], "offsets": head[:, M + N:M + N + 2], "embeddings": head[:, M + N + 2:]}
lib = L.lib()
L.check(lib.sd_decode_set_option(b"map_parallel_from", 1))
for env, key in (("SD_MAP_SPLIT", b"map_split"), ("SD_MAP_RG", b"map_rank_group")):
    if os.environ.get(env):
        L.check(lib.sd_decode_set_option(key, int(os.environ[env])))
lib.sd_debug_read_trace.restype = C.c_int
lib.sd_debug_read_trace.argtypes = [C.c_void_p, C.c_int]
buf = (C.c_ulonglong * 8192)()
rows, blocks = [], []
for it in range(30):
    lib.sd_debug_read_trace(buf, 0)
    dec.decode_packed(outs, 0.5, 0.1, exact_topk=exact, fused=False)
    torch.cuda.synchronize()
    lib.sd_debug_read_trace(buf, 8192)
    t = np.frombuffer(buf, dtype=np.uint64).astype(np.int64)
    if it >= 10:
        t0 = t[6400]
        nb = min(B * (M + N) * int(os.environ.get("SD_MAP_SPLIT", "1") or 1), 592)
        st, en = t[7000:7000 + nb] - t0, t[7600:7600 + nb] - t0
        blocks.append([st.min(), np.percentile(st, 50), np.percentile(st, 90), st.max(), np.median(en - st), (en - st).max(), en.max()])
        rows.append(np.concatenate([t[6400:6406] - t0, t[6420:6426] - t0, t[6500:6503] - t0, t[6600:6604] - t0, t[6610:6614] - t0,
                                    t[6700:6705] - t0, t[6710:6715] - t0, t[6604:6606] - t0]))
r = np.median(np.array(rows, np.float64), axis=0) * 0.01
f = lambda v: " ".join(f"{x:7.2f}" for x in v)
print(f"{'stress' if stress else 'cfg'} decode B={B} exact={exact} split={os.environ.get('SD_MAP_SPLIT', 'auto')} rank_group={os.environ.get('SD_MAP_RG', '1')}: us since block 0 of "
      "k_map_stream_select started (medians over 20 runs; thread 0's view)")
print("  stream block 0      [start, streamed(wave 0), keys converted, extra, selected, stored]:", f(r[0:6]))
print("  stream first part map block                                                      :", f(r[6:12]))
print("  k_rank_maps block 0 [start, lists in LDS, ranked]                                 :", f(r[12:15]))
print("  k_group_wide (0,0)  [start, anchors posted, barrier, parts done]                  :", f(r[15:19]))
print("  k_group_wide last   [start, anchors posted, barrier, parts done]                  :", f(r[19:23]))
print("  k_rank_group block 0 [start, lists in LDS, ranked, decoded, grouped]              :", f(r[23:28]))
print("  k_rank_group last    [start, lists in LDS, ranked, decoded, grouped]              :", f(r[28:33]))
print("  k_group_wide (0,0)  short lists: [lists in LDS, zero slots filled] (0 / negative = path not taken):", f(r[33:35]))
bl = np.median(np.array(blocks, np.float64), axis=0) * 0.01
print(f"  all blocks of k_map_stream_select: start min {bl[0]:.2f} median {bl[1]:.2f} p90 {bl[2]:.2f} max {bl[3]:.2f}; run time median {bl[4]:.2f} max {bl[5]:.2f}; last end {bl[6]:.2f}")
