#!/usr/bin/env python3
"""Stage timestamps inside the map-parallel decoder kernels (k_select_map, k_merge_group) on the stress geometry.
Library built with `make -C structuredetector_amd/csrc SUFFIX=_trace EXTRA=-DSD_DECODE_TRACE`; run with
SDNET_HIP_LIB=structuredetector_amd/csrc/libsdnet_hip_trace.so SDNET_ALLOW_ABLATION=1."""
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

exact = bool(int(sys.argv[1])) if len(sys.argv) > 1 else True
dev = torch.device("cuda")
M, N, K, P, img, B = 8, 8, 128, 512, 1024, 16
args = make_args(dev, M, N, K, P)
enc, dec = Encode(args), Decoder(args)
gen = torch.Generator(device=dev).manual_seed(0)
tgt = enc.render(enc.plan(img, img, *synthetic_batch(np.random.default_rng(B), B, img, img, M, N, 64, 96)), dev)
hm = torch.cat([tgt["anchor_hm"], tgt["part_hm"]], 1).clamp(1e-4, 0.95)
head = torch.cat([torch.log(hm / (1 - hm)) + 0.05 * torch.randn(hm.shape, device=dev, generator=gen),
                  0.1 * torch.randn(B, 4, img // 4, img // 4, device=dev, generator=gen)], 1)
outs = {"anchor_hm": head[:, :M], "part_hm": head[:, M:M + N], "offsets": head[:, M + N:M + N + 2], "embeddings": head[:, M + N + 2:]}
lib = L.lib()
L.check(lib.sd_decode_set_option(b"map_parallel_from", 1))
lib.sd_debug_read_trace.restype = C.c_int
lib.sd_debug_read_trace.argtypes = [C.c_void_p, C.c_int]
buf = (C.c_ulonglong * 8192)()
rows, more = [], []
for it in range(30):
    dec.decode_packed(outs, 0.5, 0.1, exact_topk=exact, fused=False)
    torch.cuda.synchronize()
    lib.sd_debug_read_trace(buf, 8192)
    t = np.frombuffer(buf, dtype=np.uint64).astype(np.int64)
    if it >= 10:
        more.append(np.concatenate([t[6500:6503] - t[6500], t[6520:6523] - t[6520], [t[6520] - t[6500]], t[6600:6604] - t[6600], t[6610:6614] - t[6610],
                                    [t[6610] - t[6600], t[6500] - t[6400], t[6600] - t[6500]]]))
        rows.append(np.concatenate([t[6400:6406] - t[6400], t[6420:6426] - t[6420], t[6440:6446] - t[6440], t[6300:6313] - t[6300], [t[6440] - t[6400]]]))
r = np.median(np.array(rows, np.float64), axis=0) * 0.01
print(f"stress decode, exact={exact}: k_map_stream_select, us since the block started (wave 0's view; medians over 20 runs)")
for name, o in (("anchor map 0 (block 0)", 0), ("part map 0 (block 8)", 6), ("last block (255)", 12)):
    print(f"  {name:24s} wave 0 streamed {r[o + 1]:.2f}, all waves {r[o + 2]:.2f}, keys {r[o + 3]:.2f}, selected {r[o + 4]:.2f}, stored {r[o + 5]:.2f}")
print(f"  block 255 started {r[-1]:.2f} us after block 0")
print("  radix select of block 0: start 0, after passes: " + " ".join(f"{v:.2f}" for v in r[19:27] if abs(v) < 1e6)
      + f"; passes done {r[28]:.2f}, collected {r[29]:.2f}, sorted {r[30]:.2f}")
q = np.median(np.array(more, np.float64), axis=0) * 0.01
print(f"k_rank_maps block 0: lists in LDS {q[1]:.2f}, ranked {q[2]:.2f}; last block: {q[4]:.2f}, {q[5]:.2f} (started {q[6]:.2f} after block 0)")
print(f"k_group_wide block (0, 0): anchors posted {q[8]:.2f}, barrier {q[9]:.2f}, parts done {q[10]:.2f}; last block: {q[12]:.2f}, {q[13]:.2f}, {q[14]:.2f} (started {q[15]:.2f} after block 0)")
print(f"block 0 of k_rank_maps starts {q[16]:.2f} us after block 0 of k_map_stream_select, block 0 of k_group_wide {q[17]:.2f} us after block 0 of k_rank_maps")
