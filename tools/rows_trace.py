#!/usr/bin/env python3
"""Phase times of k_conv3x3_c64_rows_bf16 (experiment build: make -C structuredetector_amd/csrc SUFFIX=_pptrace EXTRA=-DSD_PP_TRACE)."""
import ctypes as C
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from structuredetector_amd import _lib as L  # noqa: E402
lib = L.lib(); dev = "cuda"
d = L.ConvDesc()
d.B, d.Hi, d.Wi, d.Cin, d.Cout, d.R, d.S, d.stride, d.pad = 64, 128, 128, 64, 64, 3, 3, 1, 1
d.Ho = d.Wo = 128
x = torch.randn(64, 128, 128, 64, device=dev).bfloat16(); w = (torch.randn(64, 3, 3, 64, device=dev) * 0.05).bfloat16()
y = torch.empty_like(x)
for _ in range(3):
    L.check(lib.sd_conv2d_fwd_bf16(x.data_ptr(), w.data_ptr(), y.data_ptr(), C.byref(d), 0, 0, 0, 0, 0, 0, 0, L.stream()))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    L.check(lib.sd_conv2d_fwd_bf16(x.data_ptr(), w.data_ptr(), y.data_ptr(), C.byref(d), 0, 0, 0, 0, 0, 0, 0, L.stream()))
e1.record(); e1.synchronize()
print(f"{L.LIB_PATH.name}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per launch (bs 64, 128x128, 64 -> 64; no residual, no affine)")
raw = C.CDLL(str(L.LIB_PATH))
buf = (C.c_ulonglong * 64)()
assert raw.sd_debug_pp_trace(buf) == 0
names = ["issue row DMA", "store previous row (+ residual fetch)", "36 reads + 72 MFMAs", "vmcnt(0)", "acc -> scratch", "barrier"]
for wv in range(4):
    a = [buf[wv * 8 + k] for k in range(8)]
    n = max(a[7], 1)
    print(f"wave {wv}: {n} rows; cycles per row: " + ", ".join(f"{names[k]} {a[k] / n:.0f}" for k in range(6)) + f"; sum {sum(a[:6]) / n:.0f}")
