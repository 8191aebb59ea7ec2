#!/usr/bin/env python3
"""fp32 stem pair at the bench workload (bs=64, 512x512): sd_conv2d_stem_fwd_bn_stats and sd_conv2d_stem_wgrad, microseconds per launch,
TFLOP/s, and a check of both results against torch (conv2d on the same operands; fp64 yardstick for the weight gradient).
usage: stem_bench.py [--iters 20] [--batch 64]"""
import argparse
import ctypes as C
import hashlib
import sys
from pathlib import Path

import torch
import torch.nn.functional as F

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from structuredetector_amd import _lib as L  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20); ap.add_argument("--batch", type=int, default=64); ap.add_argument("--size", type=int, default=512)
    a = ap.parse_args()
    lib = L.lib(); dev = "cuda"
    B, H, W = a.batch, a.size, a.size
    d = L.ConvDesc()
    d.B, d.Hi, d.Wi, d.Cin, d.Cout, d.R, d.S, d.stride, d.pad = B, H, W, 3, 64, 7, 7, 2, 3
    d.Ho, d.Wo = H // 2, W // 2
    g = torch.Generator(device=dev).manual_seed(3)
    x = torch.randn(B, 3, H, W, device=dev, generator=g)
    w = torch.randn(64, 7, 7, 3, device=dev, generator=g) / 12          # [Cout][R][S][Cin]
    dy = torch.randn(B, H // 2, W // 2, 64, device=dev, generator=g)
    y = torch.empty(B, H // 2, W // 2, 64, device=dev)
    dw = torch.empty_like(w)
    mean, invstd = torch.empty(64, device=dev), torch.empty(64, device=dev)
    rm, rv = torch.zeros(64, device=dev), torch.ones(64, device=dev)
    ws_f = L.workspace(lib.sd_conv2d_stem_fwd_bn_stats_workspace_bytes(C.byref(d)), x.device)
    ws_w = L.workspace(lib.sd_conv2d_stem_wgrad_workspace_bytes(C.byref(d)), x.device)
    fwd = lambda: lib.sd_conv2d_stem_fwd_bn_stats(x.data_ptr(), w.data_ptr(), y.data_ptr(), C.byref(d), 1e-5, 0.1, rm.data_ptr(), rv.data_ptr(),
                                                  mean.data_ptr(), invstd.data_ptr(), ws_f.data_ptr(), ws_f.numel(), L.stream())
    wg = lambda: lib.sd_conv2d_stem_wgrad(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), C.byref(d), 0, ws_w.data_ptr(), ws_w.numel(), L.stream())
    flops = 2.0 * B * (H // 2) * (W // 2) * 64 * 147
    for name, fn in (("stem fwd + stats", fwd), ("stem wgrad", wg)):
        for _ in range(3):
            L.check(fn(), name)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            L.check(fn(), name)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / a.iters * 1e3
        print(f"{name:18s} {us:8.1f} us  {flops / us / 1e6:6.1f} TFLOP/s ({flops / us / 1e6 / 157.3 * 100:.1f} % of the fp32 MFMA peak)")
    # checks (on 8 images: the torch reference needs the NCHW copies)
    nb = min(B, 8)
    wt = w.permute(0, 3, 1, 2).contiguous()
    ref = F.conv2d(x[:nb], wt, stride=2, padding=3).permute(0, 2, 3, 1)
    err = (y[:nb] - ref).abs().max().item() / ref.abs().max().item()
    x64, dy64 = x.double(), dy.permute(0, 3, 1, 2).double()
    gw = torch.nn.grad.conv2d_weight(x64, wt.shape, dy64, stride=2, padding=3).permute(0, 2, 3, 1)
    gerr = (dw.double() - gw).abs().max().item() / gw.abs().max().item()
    m_ref = ref.reshape(-1, 64).mean(0) if nb == B else None
    print(f"fwd max rel err vs torch conv2d {err:.2e}; wgrad max rel err vs fp64 {gerr:.2e}"
          + (f"; mean err {(mean - m_ref).abs().max().item():.2e}" if m_ref is not None else ""))
    print("sha256 y", hashlib.sha256(y.cpu().numpy().tobytes()).hexdigest()[:16], " dw", hashlib.sha256(dw.cpu().numpy().tobytes()).hexdigest()[:16],
          " mean", hashlib.sha256(mean.cpu().numpy().tobytes()).hexdigest()[:16])
    assert err < 1e-5 and gerr < 1e-5


if __name__ == "__main__":
    main()
