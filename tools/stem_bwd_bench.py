#!/usr/bin/env python3
"""Stem backward of the mixed-precision step, bs = 64 at 512x512, kernel by kernel (stream events, best of 5 x 10 launches):
BatchNorm + ReLU + max-pool backward with an fp32 / a bf16 input gradient, and the 7x7 weight gradient from either."""
import ctypes as C
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from structuredetector_amd import _lib as L

lib = L.lib()
dev = torch.device("cuda")
B, H, W = 64, 512, 512
d0 = L.ConvDesc()
d0.B, d0.Hi, d0.Wi, d0.Cin, d0.Cout, d0.R, d0.S, d0.stride, d0.pad, d0.Ho, d0.Wo = B, H, W, 3, 64, 7, 7, 2, 3, H // 2, W // 2
img = torch.randn(B, 3, H, W, device=dev)
x16 = torch.randn(B, H // 2, W // 2, 64, device=dev).bfloat16()
dp16 = torch.randn(B, H // 4, W // 4, 64, device=dev).bfloat16()
idx = torch.randint(0, 9, (B, H // 4, W // 4, 64), dtype=torch.uint8, device=dev)
mean = torch.zeros(64, device=dev); invstd = torch.ones(64, device=dev); gamma = torch.ones(64, device=dev); beta = torch.zeros(64, device=dev)
dx32 = torch.empty(B, H // 2, W // 2, 64, device=dev); dx16 = torch.empty_like(dx32, dtype=torch.bfloat16)
dg = torch.empty(64, device=dev); db = torch.empty(64, device=dev)
ws = torch.empty(max(lib.sd_col_reduce_workspace_bytes(B * (H // 2) * (W // 2), 64), lib.sd_conv2d_stem_wgrad_workspace_bytes(C.byref(d0))), dtype=torch.uint8, device=dev)
dw = torch.empty(64, 7, 7, 3, device=dev)


def timed(fn, n=10, reps=5):
    for _ in range(3):
        fn()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


def pool(fn, dx):
    return lambda: L.check(fn(dp16.data_ptr(), idx.data_ptr(), x16.data_ptr(), B, H // 2, W // 2, 64, mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(),
                              beta.data_ptr(), dx.data_ptr(), dg.data_ptr(), db.data_ptr(), 0, ws.data_ptr(), ws.numel(), L.stream()))


def wgrad(fn, dy):
    return lambda: L.check(fn(dy.data_ptr(), img.data_ptr(), dw.data_ptr(), C.byref(d0), 0, ws.data_ptr(), ws.numel(), L.stream()))


with torch.cuda.stream(torch.cuda.Stream()):
    print(f"sd_maxpool_bn_relu_bwd_bf16 (fp32 dx: reduce + finalize + apply)      {timed(pool(lib.sd_maxpool_bn_relu_bwd_bf16, dx32)):8.1f} us")
    print(f"sd_maxpool_bn_relu_bwd_bf16_dx16 (bf16 dx)                             {timed(pool(lib.sd_maxpool_bn_relu_bwd_bf16_dx16, dx16)):8.1f} us")
    print(f"sd_conv2d_stem_wgrad_bf16mm (fp32 dy, k_stem_wgrad_bf16 + reduce)      {timed(wgrad(lib.sd_conv2d_stem_wgrad_bf16mm, dx32)):8.1f} us")
    print(f"sd_conv2d_stem_wgrad_bf16 (bf16 dy, k_stem_wgrad_bf16_ring + reduce)   {timed(wgrad(lib.sd_conv2d_stem_wgrad_bf16, dx16)):8.1f} us")
