#!/usr/bin/env python3
"""Randomised comparison of k_conv3x3_c64_rows16_bf16 with the 32x32x16 row-stream kernel it replaces (same fp32 sums: the outputs must be
BIT-IDENTICAL) over batch sizes, heights (odd, 1, 2, not a multiple of the unit), one and two strips per row, forward / data-gradient,
affine / residual / ReLU combinations; every case run twice (run-to-run determinism).  usage: rows16_fuzz.py [cases=60] [seed=0]"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from structuredetector_amd import _lib as L  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
lib = L.lib(); dev = "cuda"
L.check(lib.sd_set_option(b"conv_rows64_min_units", 1)); L.check(lib.sd_set_option(b"conv_fwd_split_k", 0))
bad = 0
for case in range(n_cases):
    B = int(rng.integers(1, 9)); H = int(rng.choice([1, 2, 3, 5, 8, 9, 16, 17, 31, 40, 64])); W = int(rng.choice([128, 256]))
    mode = str(rng.choice(["fwd", "dgrad"]))
    affine = bool(rng.integers(0, 2)) and mode == "fwd"; use_res = bool(rng.integers(0, 2)); relu = bool(rng.integers(0, 2)) and mode == "fwd"
    d = L.ConvDesc()
    d.B, d.Hi, d.Wi, d.Cin, d.Cout, d.R, d.S, d.stride, d.pad = B, H, W, 64, 64, 3, 3, 1, 1
    d.Ho, d.Wo = H, W
    g = torch.Generator(device=dev).manual_seed(case)
    x = torch.randn(B, H, W, 64, device=dev, generator=g).bfloat16()
    w = (torch.randn(64, 3, 3, 64, device=dev, generator=g) / 24).bfloat16()
    res = torch.randn(B, H, W, 64, device=dev, generator=g).bfloat16()
    scale = torch.rand(64, device=dev, generator=g) + 0.5; shift = torch.randn(64, device=dev, generator=g)
    outs = {}
    for v in (0, 1, 1):
        L.check(lib.sd_set_option(b"conv_rows16", v))
        y = torch.full((B, H, W, 64), 7.0, device=dev).bfloat16()
        if mode == "fwd":
            L.check(lib.sd_conv2d_fwd_bf16(x.data_ptr(), w.data_ptr(), y.data_ptr(), C.byref(d), scale.data_ptr() if affine else 0, shift.data_ptr() if affine else 0,
                                           res.data_ptr() if use_res else 0, 0, int(relu), 0, 0, L.stream()))
        else:
            L.check(lib.sd_conv2d_dgrad_bf16(x.data_ptr(), w.data_ptr(), y.data_ptr(), C.byref(d), res.data_ptr() if use_res else 0, 1 if use_res else 0, L.stream()))
        name = lib.sd_conv2d_kernel_name(C.byref(d), 16 if mode == "fwd" else 17).decode()
        outs.setdefault(v, []).append((y.clone(), name))
    (y0, n0), = outs[0]
    (y1, n1), (y2, _) = outs[1]
    ok = n0 == "k_conv3x3_c64_rows_bf16" and n1 == "k_conv3x3_c64_rows16_bf16" and torch.equal(y0, y1) and torch.equal(y1, y2)
    if not ok:
        bad += 1
        diff = (y0.float() - y1.float()).abs().max().item()
        print(f"MISMATCH case {case}: B={B} H={H} W={W} {mode} affine={affine} res={use_res} relu={relu}: kernels {n0} / {n1}, max |diff| {diff}, rerun equal {torch.equal(y1, y2)}")
L.check(lib.sd_set_option(b"conv_rows16", 1))
print(f"rows16 fuzz: {n_cases} cases, {bad} mismatches")
raise SystemExit(1 if bad else 0)
