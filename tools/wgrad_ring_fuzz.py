#!/usr/bin/env python3
"""Randomised differential check of the row-ring weight-gradient kernels (fp32 k_wgrad3x3_ring / _ring2, bf16 k_wgrad3x3_bf16_ring<D> / _ring2)
against the first forms on the same operands: random batch, map height, map width (multiples of 32) and channel counts.
usage: wgrad_ring_fuzz.py [cases=40] [seed=0]"""
import ctypes as C
import random
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from structuredetector_amd import _lib as L  # noqa: E402
from tests.test_gpu_network import make_desc  # noqa: E402

lib = L.lib()
dev = "cuda"
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = {"f32": 0.0, "bf16": 0.0}
for case in range(n_cases):
    B = rnd.choice([1, 2, 3, 5, 8, 13, 24])
    H = rnd.choice([1, 2, 3, 5, 8, 17, 32, 40])
    W = 32 * rnd.choice([1, 2, 3, 4])
    cin, cout = 64 * rnd.choice([1, 2, 3]), 64 * rnd.choice([1, 2, 4])
    d = make_desc(L, B, H, W, cin, cout, 3, 1, 1)
    g = torch.Generator(device=dev).manual_seed(case)
    x = torch.randn(B, H, W, cin, device=dev, generator=g)
    dy = torch.randn(B, H, W, cout, device=dev, generator=g)
    ws = torch.empty(max(lib.sd_conv2d_wgrad_workspace_bytes(C.byref(d)), lib.sd_conv2d_wgrad_bf16_workspace_bytes(C.byref(d)), 256), dtype=torch.uint8, device=dev)

    def run32(form):
        L.check(lib.sd_set_option(b"wgrad_f32_ring", form))
        dw = torch.full((cout, 3, 3, cin), float("nan"), device=dev)
        L.check(lib.sd_conv2d_wgrad(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), C.byref(d), 0, ws.data_ptr(), ws.numel(), L.stream()))
        return dw

    def run16(form, x16, dy16):
        L.check(lib.sd_set_option(b"wgrad_bf16_ring", form))
        dw = torch.full((cout, 3, 3, cin), float("nan"), device=dev)
        L.check(lib.sd_conv2d_wgrad_bf16(dy16.data_ptr(), x16.data_ptr(), dw.data_ptr(), C.byref(d), 0, ws.data_ptr(), ws.numel(), L.stream()))
        return dw

    ref = run32(0)
    scale = ref.abs().max().item() + 1e-30
    for form in (1, 2):
        err = (run32(form) - ref).abs().max().item() / scale
        worst["f32"] = max(worst["f32"], err)
        assert err <= 2e-5, (case, "f32", form, B, H, W, cin, cout, err)
    x16, dy16 = x.to(torch.bfloat16), dy.to(torch.bfloat16)
    ref = run16(0, x16, dy16)
    scale = ref.abs().max().item() + 1e-30
    for form in (2, 3, 4, 5):
        err = (run16(form, x16, dy16) - ref).abs().max().item() / scale
        worst["bf16"] = max(worst["bf16"], err)
        assert err <= 2e-5, (case, "bf16", form, B, H, W, cin, cout, err)
L.check(lib.sd_set_option(b"wgrad_f32_ring", 2)); L.check(lib.sd_set_option(b"wgrad_bf16_ring", 5))
print(f"{n_cases} random shapes: every ring form equals the first form; worst relative difference fp32 {worst['f32']:.2e}, bf16 {worst['bf16']:.2e}")
