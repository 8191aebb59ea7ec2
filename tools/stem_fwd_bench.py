#!/usr/bin/env python3
"""Stem forward of the mixed-precision step, bs = 64 at 512x512 (stream events, best of 5 x 10): sd_conv2d_stem_fwd_bn_stats_bf16 on the
row-ring kernel and on k_stem_fwd<true> (sd_set_option("stem_fwd_ring", 0)); conv + the statistics finalize launches."""
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
w = torch.randn(64, 7, 7, 3, device=dev) / 12
y = torch.empty(B, H // 2, W // 2, 64, dtype=torch.bfloat16, device=dev)
mean = torch.empty(64, device=dev); invstd = torch.empty(64, device=dev); rm = torch.zeros(64, device=dev); rv = torch.ones(64, device=dev)
ws = torch.empty(lib.sd_conv2d_stem_fwd_bn_stats_workspace_bytes(C.byref(d0)), dtype=torch.uint8, device=dev)


def run():
    L.check(lib.sd_conv2d_stem_fwd_bn_stats_bf16(img.data_ptr(), w.data_ptr(), y.data_ptr(), C.byref(d0), 1e-5, 0.1, rm.data_ptr(), rv.data_ptr(),
                                                 mean.data_ptr(), invstd.data_ptr(), ws.data_ptr(), ws.numel(), L.stream()))


def timed(n=10, reps=5):
    for _ in range(3):
        run()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            run()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


pool16 = torch.empty(B, H // 4, W // 4, 64, dtype=torch.bfloat16, device=dev); idx = torch.empty(B, H // 4, W // 4, 64, dtype=torch.uint8, device=dev)
x32 = torch.randn(B, H // 2, W // 2, 64, device=dev); pool32 = torch.empty(B, H // 4, W // 4, 64, device=dev)
gamma = torch.ones(64, device=dev); beta = torch.zeros(64, device=dev)


def pool(bf16):
    def f():
        if bf16:
            L.check(lib.sd_bn_relu_maxpool_fwd_bf16(y.data_ptr(), B, H // 2, W // 2, 64, mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                                    pool16.data_ptr(), idx.data_ptr(), L.stream()))
        else:
            L.check(lib.sd_bn_relu_maxpool_fwd(x32.data_ptr(), B, H // 2, W // 2, 64, mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                               pool32.data_ptr(), idx.data_ptr(), L.stream()))
    return f


def timed_fn(fn, n=10, reps=5):
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


with torch.cuda.stream(torch.cuda.Stream()):
    run()
    for pair in (0, 1, 0, 1):
        L.check(lib.sd_set_option(b"pool_fwd_pair", pair))
        print(f"sd_bn_relu_maxpool_fwd, pool_fwd_pair={pair}: fp32 {timed_fn(pool(False)):8.1f} us, bf16 {timed_fn(pool(True)):8.1f} us")
    for ring in (0, 1, 0, 1):
        L.check(lib.sd_set_option(b"stem_fwd_ring", ring))
        print(f"sd_conv2d_stem_fwd_bn_stats_bf16, stem_fwd_ring={ring}: {timed():8.1f} us")
