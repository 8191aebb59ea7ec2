#!/usr/bin/env python3
"""Phase times of k_conv_fwd_sb blocks (experiment build: make -C structuredetector_amd/csrc SUFFIX=_sbtrace EXTRA=-DSD_SB_TRACE;
run with SDNET_HIP_LIB=structuredetector_amd/csrc/libsdnet_hip_sbtrace.so SDNET_ALLOW_ABLATION=1).  100 MHz timestamps per block:
0 start, 1 prologue done (3 chunks issued), 2 first chunk landed, 3 loop done, 4 slab stored + drained, 5 ticket drawn, 6 output stored."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from structuredetector_amd import _lib as L  # noqa: E402

lib = L.lib()
raw = C.CDLL(str(L.LIB_PATH))
dev = torch.device("cuda")
CASES = {"layer1": (1, 128, 128, 64, 64, 3, 1, 1), "layer2": (1, 64, 64, 128, 128, 3, 1, 1), "layer3": (1, 32, 32, 256, 256, 3, 1, 1),
         "layer4": (1, 16, 16, 512, 512, 3, 1, 1), "ds2": (1, 128, 128, 64, 128, 1, 2, 0), "up4conv": (1, 128, 128, 128, 128, 3, 1, 1)}
bf16 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
dt = torch.bfloat16 if bf16 else torch.float32
for name, (B, H, W, cin, cout, k, s, p) in CASES.items():
    d = L.ConvDesc(); d.B, d.Hi, d.Wi, d.Cin, d.Cout, d.R, d.S, d.stride, d.pad = B, H, W, cin, cout, k, k, s, p
    d.Ho = (H + 2 * p - k) // s + 1; d.Wo = (W + 2 * p - k) // s + 1
    x = torch.randn(B, H, W, cin, device=dev).to(dt); w = (torch.randn(cout, k, k, cin, device=dev) / (cin * k * k) ** 0.5).to(dt)
    y = torch.empty(B, d.Ho, d.Wo, cout, device=dev, dtype=dt)
    ws = torch.empty(max(lib.sd_conv2d_fwd_sb_workspace_bytes(C.byref(d), bf16), 256), dtype=torch.uint8, device=dev)
    st = torch.zeros(max(lib.sd_conv2d_fwd_sb_state_bytes(C.byref(d), bf16), 256), dtype=torch.uint8, device=dev)
    for _ in range(5):
        L.check(lib.sd_conv2d_fwd_sb(x.data_ptr(), w.data_ptr(), y.data_ptr(), C.byref(d), 0, 0, 0, 0, 1, bf16, ws.data_ptr(), ws.numel(), st.data_ptr(), st.numel(), L.stream()))
    torch.cuda.synchronize()
    nb = 1024
    buf = (C.c_ulonglong * (8 * nb))()
    assert raw.sd_debug_sb_trace(buf, nb) == 0
    t = np.frombuffer(buf, dtype=np.uint64).reshape(nb, 8).astype(np.int64)
    blocks = int(lib.sd_conv2d_fwd_sb_workspace_bytes(C.byref(d), bf16) // 16384) or ((B * d.Ho * d.Wo + 63) // 64) * (cout // 64)
    t = t[:min(blocks, nb)]
    t0 = t[:, 0].min()
    us = lambda a: a / 100.0
    split = t[:, 4].max() > 0 and blocks > ((B * d.Ho * d.Wo + 63) // 64) * (cout // 64)
    last = t[:, 6] > t[:, 0]
    print(f"{name} {'bf16' if bf16 else 'fp32'}: blocks {blocks}  start spread {us(t[:, 0].max() - t0):.2f} us | per block (median / max): "
          f"prologue {us(np.median(t[:, 1] - t[:, 0])):.2f}/{us((t[:, 1] - t[:, 0]).max()):.2f}  first chunk {us(np.median(t[:, 2] - t[:, 1])):.2f}/{us((t[:, 2] - t[:, 1]).max()):.2f}  "
          f"loop {us(np.median(t[:, 3] - t[:, 2])):.2f}/{us((t[:, 3] - t[:, 2]).max()):.2f}" +
          (f"  slab+drain {us(np.median(t[:, 4] - t[:, 3])):.2f}/{us((t[:, 4] - t[:, 3]).max()):.2f}  ticket {us(np.median(t[:, 5] - t[:, 4])):.2f}/{us((t[:, 5] - t[:, 4]).max()):.2f}"
           f"  reduce+store {us(np.median((t[:, 6] - t[:, 5])[last])):.2f}/{us((t[:, 6] - t[:, 5])[last].max()):.2f}" if split else
           f"  epilogue {us(np.median(t[:, 6] - t[:, 3])):.2f}/{us((t[:, 6] - t[:, 3]).max()):.2f}") +
          f" | last loop end {us(t[:, 3].max() - t0):.2f}  kernel end {us(t[:, 6].max() - t0):.2f} us"
          f" | args loaded after {us(np.median(t[:, 7] - t[:, 0])):.2f} us" + ("" if split else f" | loop shader cycles {int(np.median(t[:, 4]))}"))
