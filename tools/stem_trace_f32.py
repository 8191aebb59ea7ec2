#!/usr/bin/env python3
"""Phase times of the fp32 stem forward k_stem_fwd<false> (experiment build: make -C structuredetector_amd/csrc SUFFIX=_pptrace EXTRA=-DSD_PP_TRACE).
usage: SDNET_HIP_LIB=structuredetector_amd/csrc/libsdnet_hip_pptrace.so SDNET_ALLOW_ABLATION=1 python3 tools/stem_trace_f32.py"""
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
x = torch.randn(B, 3, H, W, device=dev); w = torch.randn(64, 7, 7, 3, device=dev) / 12
y = torch.empty(B, H // 2, W // 2, 64, device=dev)
mean, invstd = torch.empty(64, device=dev), torch.empty(64, device=dev)
rm, rv = torch.zeros(64, device=dev), torch.ones(64, device=dev)
if len(sys.argv) > 1:
    L.check(lib.sd_set_option(b"stem_fwd_blocks", int(sys.argv[1])))
ws = L.workspace(lib.sd_conv2d_stem_fwd_bn_stats_workspace_bytes(C.byref(d)), x.device)
for _ in range(3):
    L.check(lib.sd_conv2d_stem_fwd_bn_stats(x.data_ptr(), w.data_ptr(), y.data_ptr(), C.byref(d), 1e-5, 0.1, rm.data_ptr(), rv.data_ptr(),
                                            mean.data_ptr(), invstd.data_ptr(), ws.data_ptr(), ws.numel(), L.stream()))
torch.cuda.synchronize()
raw = C.CDLL(str(L.LIB_PATH))
buf = (C.c_ulonglong * 64)()
assert raw.sd_debug_stem_trace(buf) == 0
names = ["wait for the patch DMA", "barrier 1", "MFMA loop", "barrier 2", "issue next DMA", "epilogue (LDS + stores)", "statistics"]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    lib.sd_conv2d_stem_fwd_bn_stats(x.data_ptr(), w.data_ptr(), y.data_ptr(), C.byref(d), 1e-5, 0.1, rm.data_ptr(), rv.data_ptr(), mean.data_ptr(), invstd.data_ptr(), ws.data_ptr(), ws.numel(), L.stream())
e1.record(); torch.cuda.synchronize()
print(f"launch: {e0.elapsed_time(e1) * 100:.1f} us (incl. finalize)")
for wv in range(4):
    a = [buf[wv * 8 + k] for k in range(8)]
    n = max(a[7], 1)
    print(f"wave {wv}: {n} tiles; cycles per tile: " + ", ".join(f"{names[k]} {a[k] / n:.0f}" for k in range(7)) + f"; sum {sum(a[:7]) / n:.0f}")
