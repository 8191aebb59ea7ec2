#!/usr/bin/env python3
"""Phase times of k_stem_wgrad_bf16_ring (experiment build: make -C structuredetector_amd/csrc SUFFIX=_pptrace EXTRA=-DSD_PP_TRACE), bs = 64, 512x512.
usage: SDNET_HIP_LIB=structuredetector_amd/csrc/libsdnet_hip_pptrace.so SDNET_ALLOW_ABLATION=1 python3 tools/stem_wgrad_trace.py"""
import ctypes as C
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from structuredetector_amd import _lib as L  # noqa: E402
lib = L.lib(); dev = "cuda"
B, H, W = 64, 512, 512
d = L.ConvDesc()
d.B, d.Hi, d.Wi, d.Cin, d.Cout, d.R, d.S, d.stride, d.pad = B, H, W, 3, 64, 7, 7, 2, 3
d.Ho, d.Wo = H // 2, W // 2
x = torch.randn(B, 3, H, W, device=dev)
dy = torch.randn(B, H // 2, W // 2, 64, device=dev).bfloat16()
dw = torch.empty(64, 7, 7, 3, device=dev)
ws = torch.empty(lib.sd_conv2d_stem_wgrad_workspace_bytes(C.byref(d)), dtype=torch.uint8, device=dev)
for _ in range(3):
    L.check(lib.sd_conv2d_stem_wgrad_bf16(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), C.byref(d), 0, ws.data_ptr(), ws.numel(), L.stream()))
torch.cuda.synchronize()
raw = C.CDLL(str(L.LIB_PATH))
buf = (C.c_ulonglong * 64)()
assert raw.sd_debug_pp_trace(buf) == 0
names = ["wait + barrier A", "prep + raw reads + DMA issue", "commit (convert + ring stores)", "store wait + barrier B", "issue operand reads", "wait + 40 MFMAs"]
for wv in range(4):
    a = [buf[wv * 8 + k] for k in range(8)]
    n = max(a[7], 1)
    print(f"wave {wv}: {n} steps; shader-clock ticks per step: " + ", ".join(f"{names[k]} {a[k] / n:.0f}" for k in range(6)) + f"; sum {sum(a[:6]) / n:.0f}")
