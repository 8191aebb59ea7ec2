#!/usr/bin/env python3
"""Phase times of k_conv3x3_bf16_pp (experiment build: make -C structuredetector_amd/csrc SUFFIX=_pptrace EXTRA=-DSD_PP_TRACE).
usage: SDNET_HIP_LIB=structuredetector_amd/csrc/libsdnet_hip_pptrace.so SDNET_ALLOW_ABLATION=1 python3 tools/pp_trace.py [H C]"""
import ctypes as C
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from structuredetector_amd import _lib as L  # noqa: E402

H, ch = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (32, 256)
zeros = len(sys.argv) > 3
lib = L.lib()
dev = torch.device("cuda")
d = L.ConvDesc()
d.B, d.Hi, d.Wi, d.Cin, d.Cout, d.R, d.S, d.stride, d.pad = 64, H, H, ch, ch, 3, 3, 1, 1
d.Ho = d.Wo = H
x = torch.randn(64, H, H, ch, device=dev).bfloat16()
w = (torch.randn(ch, 3, 3, ch, device=dev) * 0.05).bfloat16()
if zeros:
    x.zero_(); w.zero_()
y = torch.empty(64, H, H, ch, dtype=torch.bfloat16, device=dev)
raw = C.CDLL(str(L.LIB_PATH))
for _ in range(5):
    L.check(lib.sd_conv2d_fwd_bf16(x.data_ptr(), w.data_ptr(), y.data_ptr(), C.byref(d), 0, 0, 0, 0, 0, 0, 0, L.stream()))
torch.cuda.synchronize()
buf = (C.c_ulonglong * 64)()
assert raw.sd_debug_pp_trace(buf) == 0
taps = 9 * ch // 32
print(f"layer {H}x{H} {ch}->{ch}, {taps} taps per tile; shader-clock cycles per tap, one line per wave of block 8 (waves 0-3 = group 0)")
print("wave  load+barrier1  mfma   barrier2   loop total/tap | prologue  loop  epilogue (cycles)")
for wv in range(8):
    a = [buf[wv * 8 + k] for k in range(8)]
    print(f"{wv:4d}  {a[0] / taps:12.0f} {a[1] / taps:6.0f} {a[2] / taps:9.0f} {a[3] / taps:15.0f} | {a[4]:8d} {a[3]:8d} {a[5]:8d}   loop clock {a[3] / max(a[7], 1) / 10:.3f} GHz")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    lib.sd_conv2d_fwd_bf16(x.data_ptr(), w.data_ptr(), y.data_ptr(), C.byref(d), 0, 0, 0, 0, 0, 0, 0, L.stream())
e1.record(); torch.cuda.synchronize()
print(f"launch to launch: {e0.elapsed_time(e1) * 100:.1f} us")

# chip-level timeline of the last launch: every block's start / loop begin / loop end / epilogue end (100 MHz clock)
import numpy as np
nb = (64 * H * H // 512) * (ch // 128)
nb = min(nb, 4096)
tl = (C.c_ulonglong * (4 * nb))()
assert raw.sd_debug_pp_timeline(tl, nb) == 0
t = np.frombuffer(tl, dtype=np.uint64).astype(np.int64).reshape(nb, 4)
t0 = t[:, 0].min()
t = (t - t0) * 0.01
order = np.argsort(t[:, 0])
print(f"timeline of {nb} blocks (us since the first block started): kernel span {t[:, 3].max():.1f}")
print("  start: first %.1f  median %.1f  last %.1f" % (t[:, 0].min(), np.median(t[:, 0]), t[:, 0].max()))
pro, loop, epi = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]
for name, v in (("prologue", pro), ("loop", loop), ("epilogue", epi)):
    print(f"  {name:9s} min {v.min():6.2f}  median {np.median(v):6.2f}  p90 {np.percentile(v, 90):6.2f}  max {v.max():6.2f}")
rounds = max(1, int(round(nb / 256)))
for r in range(rounds):
    sel = order[r * 256:(r + 1) * 256]
    print(f"  round {r}: start {t[sel, 0].min():6.1f}..{t[sel, 0].max():6.1f}  loop begin med {np.median(t[sel, 1]):6.1f}  loop end med {np.median(t[sel, 2]):6.1f}  "
          f"end med {np.median(t[sel, 3]):6.1f} max {t[sel, 3].max():6.1f}  | pro {np.median(pro[sel]):5.2f} loop {np.median(loop[sel]):5.2f} epi {np.median(epi[sel]):5.2f}")
