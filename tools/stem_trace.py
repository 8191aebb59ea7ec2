#!/usr/bin/env python3
"""Phase times of k_stem_pool_bf16 (experiment build: make -C structuredetector_amd/csrc SUFFIX=_pptrace EXTRA=-DSD_PP_TRACE)."""
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
sc = torch.ones(64, device=dev); sh = torch.zeros(64, device=dev)
y = torch.empty(B, H // 4, W // 4, 64, dtype=torch.bfloat16, device=dev)
for _ in range(3):
    L.check(lib.sd_stem_bn_relu_maxpool_fwd_bf16(x.data_ptr(), w.data_ptr(), sc.data_ptr(), sh.data_ptr(), y.data_ptr(), C.byref(d), L.stream()))
torch.cuda.synchronize()
raw = C.CDLL(str(L.LIB_PATH))
buf = (C.c_ulonglong * 64)()
assert raw.sd_debug_pp_trace(buf) == 0
names = ["patch->LDS (waits for the prefetch)", "issue next fetch", "barrier 1", "MFMA + rowbuf", "barrier 2", "pool + store", "barrier 3 + halo"]
for wv in range(4):
    a = [buf[wv * 8 + k] for k in range(8)]
    n = max(a[7], 1)
    print(f"wave {wv}: {n} tiles; cycles per tile: " + ", ".join(f"{names[k]} {a[k] / n:.0f}" for k in range(7)) + f"; sum {sum(a[:7]) / n:.0f}")
