#!/usr/bin/env python3
"""sd_conv2d_wgrad_bf16 alone on the 3x3 / stride 1 layer shapes of the mixed-precision step (bs = 64, 512x512): device time per call
(events over 20 calls), all-taps kernel + partial-sum reduce.  With SDNET_HIP_LIB / SDNET_ALLOW_ABLATION=1 for timing-only builds
(make SUFFIX=_w16ablN EXTRA=-DSD_W16_ABL=N)."""
import ctypes as C
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from structuredetector_amd import _lib as L  # noqa: E402
from tests.test_gpu_network import make_desc  # noqa: E402

lib = L.lib()
import os
if os.environ.get('SD_RING'):
    L.check(lib.sd_set_option(b'wgrad_bf16_ring', int(os.environ['SD_RING'])))
dev = "cuda"
SHAPES = [("layer1 64->64 @128", 128, 64, 64), ("layer2 128->128 @64", 64, 128, 128), ("layer3 256->256 @32", 32, 256, 256), ("layer4 512->512 @16", 16, 512, 512),
          ("up4.conv 128->128 @128", 128, 128, 128)]
for name, H, cin, cout in SHAPES:
    d = make_desc(L, 64, H, H, cin, cout, 3, 1, 1)
    x = torch.randn(64, H, H, cin, device=dev).to(torch.bfloat16)
    dy = torch.randn(64, H, H, cout, device=dev).to(torch.bfloat16)
    dw = torch.empty(cout, 3, 3, cin, device=dev)
    ws = torch.empty(lib.sd_conv2d_wgrad_bf16_workspace_bytes(C.byref(d)), dtype=torch.uint8, device=dev)
    run = lambda: L.check(lib.sd_conv2d_wgrad_bf16(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), C.byref(d), 0, ws.data_ptr(), ws.numel(), L.stream()))
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    gf = 2.0 * 64 * H * H * cin * cout * 9 / 1e9
    print(f"{name:26s} {us:8.1f} us per call (kernel + reduce)  {gf / (us * 1e-6) / 1e3:7.1f} TFLOP/s", flush=True)
