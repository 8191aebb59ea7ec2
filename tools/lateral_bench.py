#!/usr/bin/env python3
"""FPN lateral 1x1 conv (64 -> 128 @128x128, bs=64) with and without its fused epilogue (bias, upsample-add)."""
import ctypes as C
import sys
import time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from structuredetector_amd import _lib as L
lib = L.lib()
dev = torch.device("cuda")
B, H, cin, cout = 64, 128, 64, 128
d = L.ConvDesc()
d.B, d.Hi, d.Wi, d.Cin, d.Cout, d.R, d.S, d.stride, d.pad = B, H, H, cin, cout, 1, 1, 1, 0
d.Ho = d.Wo = H
x = torch.randn(B, H, H, cin, device=dev)
w = torch.randn(cout, 1, 1, cin, device=dev) * 0.1
y = torch.empty(B, H, H, cout, device=dev)
bias = torch.randn(cout, device=dev)
f = torch.randn(B, H // 2, H // 2, cout, device=dev)
full = torch.randn(B, H, H, cout, device=dev)
def run(shift, res, up2):
    L.check(lib.sd_conv2d_fwd(x.data_ptr(), w.data_ptr(), y.data_ptr(), C.byref(d), 0, shift, res, up2, 0, 0, 0, L.stream()))
for name, args in (("plain", (0, 0, 0)), ("bias", (bias.data_ptr(), 0, 0)), ("bias + full-size residual", (bias.data_ptr(), full.data_ptr(), 0)),
                   ("bias + upsampled residual", (bias.data_ptr(), f.data_ptr(), 1))):
    for _ in range(3):
        run(*args)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        run(*args)
    torch.cuda.synchronize()
    print(f"{name:30s} {(time.perf_counter() - t0) / 20 * 1e6:8.1f} us", flush=True)
