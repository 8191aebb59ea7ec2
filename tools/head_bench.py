#!/usr/bin/env python3
"""fp32 head (1x1 conv 128 -> M + N + 4, NHWC in, NCHW out) at the bench workload: forward and backward launches, microseconds and the
fraction of the HBM roofline (algorithmic bytes: the 537 MB activation once per pass)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from structuredetector_amd import _lib as L  # noqa: E402

lib = L.lib(); dev = "cuda"
B, H, W, C, Co = 64, 128, 128, 128, 7
x = torch.randn(B, H, W, C, device=dev); w = torch.randn(Co, C, device=dev) / 11; b = torch.randn(Co, device=dev)
y = torch.empty(B, Co, H, W, device=dev); dy = torch.randn(B, Co, H, W, device=dev)
dx = torch.empty_like(x); dw = torch.empty_like(w); db = torch.empty_like(b)
ws = torch.empty(lib.sd_head_bwd_workspace_bytes(B, H * W, C, Co), dtype=torch.uint8, device=dev)
fwd = lambda: lib.sd_head_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), B, H * W, C, Co, L.stream())
bwd = lambda: lib.sd_head_bwd(dy.data_ptr(), x.data_ptr(), w.data_ptr(), dx.data_ptr(), dw.data_ptr(), db.data_ptr(), B, H * W, C, Co, 0, ws.data_ptr(), ws.numel(), L.stream())
for name, fn, nbytes in (("head fwd", fwd, x.numel() * 4 + y.numel() * 4), ("head bwd (dgrad + wgrad + finalize)", bwd, 2 * x.numel() * 4 + 2 * dy.numel() * 4)):
    for _ in range(3):
        L.check(fn(), name)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        L.check(fn(), name)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"{name:38s} {us:8.1f} us  {nbytes / us / 1e6:6.2f} TB/s ({nbytes / us / 1e6 / 8 * 100:.0f} % of 8 TB/s)")
