#!/usr/bin/env python3
"""Phase times of k_conv3x3_c64_rows_f32 (experiment build: make -C structuredetector_amd/csrc SUFFIX=_pptrace EXTRA=-DSD_PP_TRACE;
run with SDNET_HIP_LIB=structuredetector_amd/csrc/libsdnet_hip_pptrace.so SDNET_ALLOW_ABLATION=1)."""
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
x = torch.randn(64, 128, 128, 64, device=dev); w = torch.randn(64, 3, 3, 64, device=dev) * 0.05
y = torch.empty_like(x)
assert lib.sd_conv2d_kernel_name(C.byref(d), 0) == b"k_conv3x3_c64_rows_f32"
for _ in range(3):
    L.check(lib.sd_conv2d_fwd(x.data_ptr(), w.data_ptr(), y.data_ptr(), C.byref(d), 0, 0, 0, 0, 0, 0, 0, L.stream()))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    L.check(lib.sd_conv2d_fwd(x.data_ptr(), w.data_ptr(), y.data_ptr(), C.byref(d), 0, 0, 0, 0, 0, 0, 0, L.stream()))
e1.record(); torch.cuda.synchronize()
print(f"launch {e0.elapsed_time(e1) / 10 * 1e3:.1f} us = {77.3 / (e0.elapsed_time(e1) / 10):.1f} TFLOP/s")
raw = C.CDLL(str(L.LIB_PATH))
buf = (C.c_ulonglong * 64)()
assert raw.sd_debug_pp_trace(buf) == 0
names = ["row start (residual, DMA issue, zero acc)", "72 reads + 288 MFMAs", "vmcnt(0)", "epilogue", "barrier"]
for wv in range(4):
    a = [buf[wv * 8 + k] for k in range(8)]
    n = max(a[7], 1)
    print(f"wave {wv}: {n} rows; cycles per row: " + ", ".join(f"{names[k]} {a[k] / n:.0f}" for k in range(5)) + f"; sum {sum(a[:5]) / n:.0f}")
