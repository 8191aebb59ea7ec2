#!/usr/bin/env python3
"""Randomised check of the bf16 forward convs that changed in round 4 against a PyTorch reference on the same bf16-rounded operands:
the two-group 3x3 kernel (v_mfma_f32_16x16x32_bf16; forced onto small grids) with scale / shift / residual / ReLU, the same conv with the
head fused into its epilogue, and the 1x1 lateral stream kernel with bias and a half-size residual.
usage: conv_bf16_fuzz.py [cases=30] [seed=0]"""
import ctypes as C
import random
import sys
from pathlib import Path

import torch
import torch.nn.functional as F

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from structuredetector_amd import _lib as L  # noqa: E402
from tests.test_gpu_network import make_desc  # noqa: E402

lib = L.lib()
dev = "cuda"
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)


def nhwc16(t):
    return t.permute(0, 2, 3, 1).contiguous().to(dev).to(torch.bfloat16)


def back(t):
    return t.float().permute(0, 3, 1, 2).cpu()


L.check(lib.sd_set_option(b"conv_pp_min_tiles", 1)); L.check(lib.sd_set_option(b"conv_fwd_split_k", 0)); L.check(lib.sd_set_option(b"conv1x1_stream_min_pixels", 32))
worst = [0.0, 0.0, 0.0]
done = [0, 0, 0]
for case in range(n_cases):
    W = rnd.choice([16, 32, 64, 128, 256])
    H = rnd.choice([4, 8, 16, 32]) if W >= 128 else rnd.choice([W // 2, W, 2 * W])
    B = rnd.choice([1, 2, 3, 4])
    while (B * H * W) % 512:
        B += 1
    cin, cout = 64 * rnd.choice([1, 2, 4]), 128 * rnd.choice([1, 2])
    g = torch.Generator().manual_seed(case)
    x = torch.randn(B, cin, H, W, generator=g).bfloat16().float()
    w = (torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5).bfloat16().float()
    scale = torch.rand(cout, generator=g) + 0.5; shift = torch.randn(cout, generator=g)
    res = torch.randn(B, cout, H, W, generator=g).bfloat16().float()
    d = make_desc(L, B, H, W, cin, cout, 3, 1, 1)
    name = lib.sd_conv2d_kernel_name(C.byref(d), 16).decode()
    xd, wd, rd, sc, sh = nhwc16(x), nhwc16(w), nhwc16(res), scale.to(dev), shift.to(dev)
    if name == "k_conv3x3_bf16_pp":
        y = torch.full((B, H, W, cout), float("nan"), dtype=torch.bfloat16, device=dev)
        L.check(lib.sd_conv2d_fwd_bf16(xd.data_ptr(), wd.data_ptr(), y.data_ptr(), C.byref(d), sc.data_ptr(), sh.data_ptr(), rd.data_ptr(), 0, 1, 0, 0, L.stream()))
        ref = torch.relu(F.conv2d(x, w, None, 1, 1) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1) + res)
        err = (back(y) - ref).abs().max().item() / (ref.abs().max().item() + 1e-30)
        assert err <= 8e-3, (case, name, B, H, W, cin, cout, err)
        worst[0] = max(worst[0], err); done[0] += 1
        co = rnd.choice([3, 7, 16, 20, 32])
        if cout == 128 and lib.sd_conv2d_fwd_bf16_head_supported(C.byref(d), co):
            hw = torch.randn(co, 128, generator=g); hb = torch.randn(co, generator=g)
            prep = torch.empty(lib.sd_head_split_bf16_bytes(), dtype=torch.uint8, device=dev)
            hwd, hbd = hw.to(dev), hb.to(dev)
            L.check(lib.sd_head_split_bf16(hwd.data_ptr(), hbd.data_ptr(), co, prep.data_ptr(), L.stream()))
            out = torch.full((B, co, H, W), float("nan"), device=dev)
            L.check(lib.sd_conv2d_fwd_bf16_head(xd.data_ptr(), wd.data_ptr(), C.byref(d), sc.data_ptr(), sh.data_ptr(), 1, prep.data_ptr(), co, out.data_ptr(), L.stream()))
            yb = torch.empty_like(y)
            L.check(lib.sd_conv2d_fwd_bf16(xd.data_ptr(), wd.data_ptr(), yb.data_ptr(), C.byref(d), sc.data_ptr(), sh.data_ptr(), 0, 0, 1, 0, 0, L.stream()))
            want = torch.einsum("bhwc,oc->bohw", yb.float().cpu().double(), hw.double()) + hb.double().view(1, -1, 1, 1)
            err = (out.cpu().double() - want).abs().max().item() / (want.abs().max().item() + 1e-30)
            assert err <= 2e-5, (case, "head", B, H, W, co, err)
            worst[1] = max(worst[1], err); done[1] += 1
    # lateral: 1x1 onto 128 channels with a half-size residual
    ci = rnd.choice([64, 128])
    if H % 2 == 0 and (B * H * W) % 32 == 0:
        xl = torch.randn(B, ci, H, W, generator=g).bfloat16().float()
        wl = (torch.randn(128, ci, 1, 1, generator=g) / ci ** 0.5).bfloat16().float()
        bias = torch.randn(128, generator=g)
        half = torch.randn(B, 128, H // 2, W // 2, generator=g).bfloat16().float()
        dl = make_desc(L, B, H, W, ci, 128, 1, 1, 0)
        assert lib.sd_conv2d_kernel_name(C.byref(dl), 16).decode().startswith("k_conv1x1_stream_bf16")
        yl = torch.full((B, H, W, 128), float("nan"), dtype=torch.bfloat16, device=dev)
        xld, wld, hd, bd = nhwc16(xl), nhwc16(wl), nhwc16(half), bias.to(dev)
        L.check(lib.sd_conv2d_fwd_bf16(xld.data_ptr(), wld.data_ptr(), yl.data_ptr(), C.byref(dl), 0, bd.data_ptr(), hd.data_ptr(), 1, 0, 0, 0, L.stream()))
        ref = F.conv2d(xl, wl, bias) + F.interpolate(half, scale_factor=2, mode="nearest")
        err = (back(yl) - ref).abs().max().item() / (ref.abs().max().item() + 1e-30)
        assert err <= 8e-3, (case, "stream", B, H, W, ci, err)
        worst[2] = max(worst[2], err); done[2] += 1
print(f"two-group 3x3: {done[0]} shapes, worst {worst[0]:.2e}; fused head: {done[1]} shapes, worst {worst[1]:.2e}; lateral stream: {done[2]} shapes, worst {worst[2]:.2e}")
