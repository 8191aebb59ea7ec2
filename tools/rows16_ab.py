#!/usr/bin/env python3
"""Same-process A/B of the two bf16 row-stream kernels of layer1 (64 -> 64 channels, 3x3 / 1 / 1) at the production shape:
k_conv3x3_c64_rows16_bf16 (sd_set_option("conv_rows16", 1), default) against k_conv3x3_c64_rows_bf16 (0).
usage: rows16_ab.py [B] [H=W]   (default 64 128: bs = 64 at 512 x 512 inputs; 16 256 = the stress geometry)"""
import ctypes as C
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from structuredetector_amd import _lib as L  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
HW = int(sys.argv[2]) if len(sys.argv) > 2 else 128
lib = L.lib(); dev = "cuda"
d = L.ConvDesc()
d.B, d.Hi, d.Wi, d.Cin, d.Cout, d.R, d.S, d.stride, d.pad = B, HW, HW, 64, 64, 3, 3, 1, 1
d.Ho = d.Wo = HW
x = torch.randn(B, HW, HW, 64, device=dev).bfloat16(); w = (torch.randn(64, 3, 3, 64, device=dev) * 0.05).bfloat16()
res = torch.randn(B, HW, HW, 64, device=dev).bfloat16()
y = torch.empty_like(x)
scale = torch.rand(64, device=dev) + 0.5; shift = torch.randn(64, device=dev)
mean = torch.empty(64, device=dev); invstd = torch.empty(64, device=dev); rm = torch.zeros(64, device=dev); rv = torch.ones(64, device=dev)
ws = torch.empty(lib.sd_conv2d_fwd_bf16_bn_stats_workspace_bytes(C.byref(d)), dtype=torch.uint8, device=dev)
gflop = 2 * B * HW * HW * 64 * 64 * 9 / 1e9


def fwd_plain():
    L.check(lib.sd_conv2d_fwd_bf16(x.data_ptr(), w.data_ptr(), y.data_ptr(), C.byref(d), 0, 0, 0, 0, 0, 0, 0, L.stream()))


def fwd_affine_relu():
    L.check(lib.sd_conv2d_fwd_bf16(x.data_ptr(), w.data_ptr(), y.data_ptr(), C.byref(d), scale.data_ptr(), shift.data_ptr(), 0, 0, 1, 0, 0, L.stream()))


def fwd_affine_res_relu():
    L.check(lib.sd_conv2d_fwd_bf16(x.data_ptr(), w.data_ptr(), y.data_ptr(), C.byref(d), scale.data_ptr(), shift.data_ptr(), res.data_ptr(), 0, 1, 0, 0, L.stream()))


def fwd_stats():
    L.check(lib.sd_conv2d_fwd_bf16_bn_stats(x.data_ptr(), w.data_ptr(), y.data_ptr(), C.byref(d), 1e-5, 0.1, rm.data_ptr(), rv.data_ptr(),
                                            mean.data_ptr(), invstd.data_ptr(), ws.data_ptr(), ws.numel(), L.stream()))


def dgrad_plain():
    L.check(lib.sd_conv2d_dgrad_bf16(x.data_ptr(), w.data_ptr(), y.data_ptr(), C.byref(d), 0, 0, L.stream()))


def dgrad_res():
    L.check(lib.sd_conv2d_dgrad_bf16(x.data_ptr(), w.data_ptr(), y.data_ptr(), C.byref(d), res.data_ptr(), 1, L.stream()))


def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print(f"layer1 conv, bs {B}, {HW} x {HW}, 64 -> 64, {gflop:.1f} GFLOP per launch; us per launch (TFLOP/s), three alternating rounds")
for name, fn in (("fwd plain", fwd_plain), ("fwd affine + ReLU (eval conv1)", fwd_affine_relu), ("fwd affine + residual + ReLU (eval conv2)", fwd_affine_res_relu),
                 ("fwd + BatchNorm statistics (training; incl. finalize launch)", fwd_stats), ("dgrad plain", dgrad_plain), ("dgrad + residual", dgrad_res)):
    rows = {0: [], 1: []}
    for _ in range(3):
        for v in (0, 1):
            L.check(lib.sd_set_option(b"conv_rows16", v))
            rows[v].append(timed(fn))
    L.check(lib.sd_set_option(b"conv_rows16", 1))
    a, b_ = min(rows[0]), min(rows[1])
    print(f"  {name:62s} rows (32x32x16) {a:7.1f} ({gflop / a * 1e-3 * 1e3:6.0f})   rows16 {b_:7.1f} ({gflop / b_ * 1e-3 * 1e3:6.0f})   {100 * (b_ / a - 1):+.1f} %")
