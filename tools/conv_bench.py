#!/usr/bin/env python3
"""Per-layer micro-benchmark of the MFMA conv kernels (fwd / dgrad / wgrad) on the shapes of the SDNet backbone
at bs=64, 512x512.  Run on the GPU box:  python tools/conv_bench.py [--iters 10] [--only fwd]"""
import argparse
import ctypes as C
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from structuredetector_amd import _lib as L  # noqa: E402

SHAPES = [  # name, H(in), Cin, Cout, k, stride, pad, count in the network
    ("l1.3x3 64->64 @128", 128, 64, 64, 3, 1, 1, 6),
    ("l2.0 3x3s2 64->128 @128", 128, 64, 128, 3, 2, 1, 1),
    ("l2 3x3 128->128 @64", 64, 128, 128, 3, 1, 1, 7),
    ("l2.ds 1x1s2 64->128", 128, 64, 128, 1, 2, 0, 1),
    ("l3.0 3x3s2 128->256 @64", 64, 128, 256, 3, 2, 1, 1),
    ("l3 3x3 256->256 @32", 32, 256, 256, 3, 1, 1, 11),
    ("l3.ds 1x1s2 128->256", 64, 128, 256, 1, 2, 0, 1),
    ("l4.0 3x3s2 256->512 @32", 32, 256, 512, 3, 2, 1, 1),
    ("l4 3x3 512->512 @16", 16, 512, 512, 3, 1, 1, 5),
    ("l4.ds 1x1s2 256->512", 32, 256, 512, 1, 2, 0, 1),
    ("up1 1x1 512->128 @16", 16, 512, 128, 1, 1, 0, 1),
    ("up2.lat 1x1 256->128 @32", 32, 256, 128, 1, 1, 0, 1),
    ("up2.conv 3x3 128->128 @32", 32, 128, 128, 3, 1, 1, 1),
    ("up3.lat 1x1 128->128 @64", 64, 128, 128, 1, 1, 0, 1),
    ("up3.conv 3x3 128->128 @64", 64, 128, 128, 3, 1, 1, 1),
    ("up4.lat 1x1 64->128 @128", 128, 64, 128, 1, 1, 0, 1),
    ("up4.conv 3x3 128->128 @128", 128, 128, 128, 3, 1, 1, 1),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--only", default="")
    ap.add_argument("--zeros", action="store_true", help="all-zero operands: same instruction stream at lower power (DVFS check)")
    a = ap.parse_args()
    lib = L.lib()
    dev = torch.device("cuda")
    B = a.batch
    tot = {"fwd": [0.0, 0.0], "dgrad": [0.0, 0.0], "wgrad": [0.0, 0.0]}
    print(f"{'layer':30s} {'GFLOP':>8s} | {'fwd us':>9s} {'TF':>6s} | {'dgrad us':>9s} {'TF':>6s} | {'wgrad us':>9s} {'TF':>6s}")
    for name, H, cin, cout, k, s, pad, cnt in SHAPES:
        d = L.ConvDesc()
        d.B, d.Hi, d.Wi, d.Cin, d.Cout, d.R, d.S, d.stride, d.pad = B, H, H, cin, cout, k, k, s, pad
        d.Ho = d.Wo = (H + 2 * pad - k) // s + 1
        x = torch.randn(B, H, H, cin, device=dev)
        w = torch.randn(cout, k, k, cin, device=dev) * 0.05
        if a.zeros:
            x.zero_(); w.zero_()
        wt = torch.empty(cin * k * k * cout, device=dev)
        y = torch.empty(B, d.Ho, d.Wo, cout, device=dev)
        dy = torch.randn(B, d.Ho, d.Wo, cout, device=dev)
        if a.zeros:
            dy.zero_()
        dx = torch.empty_like(x)
        dw = torch.empty_like(w)
        ws = torch.empty(max(lib.sd_conv2d_wgrad_workspace_bytes(C.byref(d)), 256), dtype=torch.uint8, device=dev)
        L.check(lib.sd_conv2d_transpose_weights(w.data_ptr(), wt.data_ptr(), cout, k * k, cin, L.stream()))
        gflop = 2.0 * B * d.Ho * d.Wo * cout * cin * k * k / 1e9
        fns = {
            "fwd": lambda: lib.sd_conv2d_fwd(x.data_ptr(), w.data_ptr(), y.data_ptr(), C.byref(d), 0, 0, 0, 0, 0, 0, 0, L.stream()),
            "dgrad": lambda: lib.sd_conv2d_dgrad(dy.data_ptr(), wt.data_ptr(), dx.data_ptr(), C.byref(d), 0, L.stream()),
            "wgrad": lambda: lib.sd_conv2d_wgrad(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), C.byref(d), 0, ws.data_ptr(), ws.numel(), L.stream()),
        }
        row = f"{name:30s} {gflop:8.1f} |"
        for kind, fn in fns.items():
            if a.only and kind != a.only:
                row += f" {'-':>9s} {'-':>6s} |"
                continue
            for _ in range(2):
                L.check(fn())
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                fn()
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / a.iters
            tot[kind][0] += us * cnt; tot[kind][1] += gflop * cnt
            row += f" {us:9.1f} {gflop / us * 1e3:6.1f} |"
        print(row, flush=True)
    for kind, (us, gf) in tot.items():
        if us:
            print(f"network total {kind:6s}: {us / 1e3:8.2f} ms  {gf / us * 1e3:6.1f} TFLOP/s ({gf / us * 1e3 / 157.3 * 100:.1f}% of fp32 MFMA peak)")


if __name__ == "__main__":
    main()
