#!/usr/bin/env python3
"""Eval forward: fp32 vs bf16 backbone at bs=64 512x512 and the stress config (1024x1024, 8 labels / 8 parts)."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bench import make_args  # noqa: E402
from structuredetector_amd.data import Decoder  # noqa: E402
from structuredetector_amd.model import Network  # noqa: E402

dev = torch.device("cuda")
import os
if os.environ.get("SD_PATCH_BN64"):       # A/B switch: 64-channel layers through the patch-staging kernel too
    from structuredetector_amd import _lib as _L
    _L.check(_L.lib().sd_set_option(b"conv_patch_bn64", 1))
if os.environ.get("SD_NO_PATCH"):          # A/B switch: the layers the patch-staging kernel would take run the generic igemm kernels
    from structuredetector_amd import _lib as _L
    _L.check(_L.lib().sd_set_option(b"conv_patch_min_tiles", 1 << 30))


def timeit(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


for (B, img, M, N, K, P, gf) in ((64, 512, 2, 1, 20, 40, 45.15), (16, 1024, 8, 8, 128, 512, 180.8), (1, 512, 2, 1, 20, 40, 45.15)):
    args = make_args(dev, M, N, K, P)
    net = Network(args, pretrained=False).to(dev).eval()
    dec = Decoder(args)
    x = torch.randn(B, 3, img, img, device=dev)
    with torch.no_grad():
        net.bf16_inference = False
        t32 = timeit(lambda: net(x))
        net.bf16_inference = True
        t16 = timeit(lambda: net(x))
        tdec = timeit(lambda: dec.decode_packed(net(x), 0.5, 0.1))
    print(f"B={B} {img}x{img} M={M} N={N}: fp32 forward {t32 * 1e3:.2f} ms ({B * gf / t32 / 1e3:.1f} TFLOP/s) | bf16 forward {t16 * 1e3:.2f} ms "
          f"({B * gf / t16 / 1e3:.1f} TFLOP/s, {t32 / t16:.2f}x) | bf16 forward + fp32 decode {tdec * 1e3:.2f} ms = {B / tdec:.0f} img/s", flush=True)
