#!/usr/bin/env python3
"""Per-layer micro-benchmark of the bf16 MFMA conv kernels (forward / data-gradient) on the shapes of the SDNet backbone at
bs=64, 512x512, with an A/B against the previous kernel choice (sd_set_option) and a bit-exactness check between the two.
Run on the GPU box:  python tools/conv_bench_bf16.py [--iters 20] [--only l2] [--no-ab]"""
import argparse
import ctypes as C
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from structuredetector_amd import _lib as L  # noqa: E402

SHAPES = [  # name, H(in), Cin, Cout, k, stride, pad, count in the network
    ("l1 3x3 64->64 @128", 128, 64, 64, 3, 1, 1, 6),
    ("l2.0 3x3s2 64->128 @128", 128, 64, 128, 3, 2, 1, 1),
    ("l2 3x3 128->128 @64", 64, 128, 128, 3, 1, 1, 8),
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
    ("up4.lat 1x1 64->128 @128", 128, 64, 128, 1, 1, 0, 1),
    ("up4.conv 3x3 128->128 @128", 128, 128, 128, 3, 1, 1, 1),
]


def timed(fn, iters):
    for _ in range(3):
        L.check(fn())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--only", default="")
    ap.add_argument("--no-ab", action="store_true")
    ap.add_argument("--zeros", action="store_true")
    ap.add_argument("--min-tiles", type=int, default=200, help="value of the option for the NEW side of the A/B")
    ap.add_argument("--option", default="conv_pp_min_tiles", help="the sd_set_option switch of the A/B (value 1 << 30 = previous kernel)")
    a = ap.parse_args()
    lib = L.lib()
    dev = torch.device("cuda")
    B = a.batch
    tot = {}
    L.check(lib.sd_set_option(a.option.encode(), a.min_tiles))
    print(f"{'layer':30s} {'GFLOP':>8s} | {'fwd us':>8s} {'TF':>6s} {'old us':>8s} {'same':>5s} | {'dgrad us':>8s} {'TF':>6s} {'old us':>8s} {'same':>5s}")
    for name, H, cin, cout, k, s, pad, cnt in SHAPES:
        if a.only and a.only not in name:
            continue
        d = L.ConvDesc()
        d.B, d.Hi, d.Wi, d.Cin, d.Cout, d.R, d.S, d.stride, d.pad = B, H, H, cin, cout, k, k, s, pad
        d.Ho = d.Wo = (H + 2 * pad - k) // s + 1
        x = torch.randn(B, H, H, cin, device=dev).bfloat16()
        w = (torch.randn(cout, k, k, cin, device=dev) * 0.05)
        w16 = w.bfloat16()
        dy = torch.randn(B, d.Ho, d.Wo, cout, device=dev).bfloat16()
        if a.zeros:
            x.zero_(); w16.zero_(); dy.zero_()
        wt = torch.empty(cin * k * k * cout, dtype=torch.bfloat16, device=dev)
        L.check(lib.sd_conv2d_transpose_weights_bf16(w.data_ptr(), wt.data_ptr(), cout, k * k, cin, L.stream()))
        y = torch.empty(B, d.Ho, d.Wo, cout, dtype=torch.bfloat16, device=dev)
        dx = torch.empty(B, H, H, cin, dtype=torch.bfloat16, device=dev)
        nws = lib.sd_conv2d_fwd_bf16_workspace_bytes(C.byref(d))
        ws = torch.empty(max(nws, 256), dtype=torch.uint8, device=dev)
        gflop = 2.0 * B * d.Ho * d.Wo * cout * cin * k * k / 1e9
        fns = {
            "fwd": (lambda: lib.sd_conv2d_fwd_bf16(x.data_ptr(), w16.data_ptr(), y.data_ptr(), C.byref(d), 0, 0, 0, 0, 0, ws.data_ptr(), nws, L.stream()), y),
            "dgrad": (lambda: lib.sd_conv2d_dgrad_bf16(dy.data_ptr(), wt.data_ptr(), dx.data_ptr(), C.byref(d), 0, 0, L.stream()), dx),
        }
        row = f"{name:30s} {gflop:8.1f} |"
        for kind, (fn, out) in fns.items():
            us = timed(fn, a.iters)
            new = out.clone()
            if a.no_ab:
                old_us, same = float("nan"), "-"
            else:
                L.check(lib.sd_set_option(a.option.encode(), 1 << 30))
                old_us = timed(fn, a.iters)
                same = "yes" if torch.equal(new, out) else "NO"
                L.check(lib.sd_set_option(a.option.encode(), a.min_tiles))
            t = tot.setdefault(kind, [0.0, 0.0, 0.0])
            t[0] += us * cnt; t[1] += gflop * cnt; t[2] += old_us * cnt
            row += f" {us:8.1f} {gflop / us * 1e3:6.1f} {old_us:8.1f} {same:>5s} |"
        print(row, flush=True)
    for kind, (us, gf, old) in tot.items():
        print(f"network total {kind:6s}: {us / 1e3:8.3f} ms  {gf / us * 1e3:6.1f} TFLOP/s ({gf / us * 1e3 / 2500 * 100:.1f}% of the bf16 MFMA peak)   previous {old / 1e3:8.3f} ms")


if __name__ == "__main__":
    main()
