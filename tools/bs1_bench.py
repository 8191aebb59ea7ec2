#!/usr/bin/env python3
"""bs=1 eval forward (BASELINE configs[1]), fp32 and bf16 backbone, with sd_conv2d_fwd_sb on (default) and off (the two-launch
split-K path): wall time per forward, eager and as a hipGraph replay; per-layer event times with --layers."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bench import make_args  # noqa: E402
from structuredetector_amd.model import Network  # noqa: E402

dev = torch.device("cuda")
args = make_args(dev)
B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 1
x = torch.randn(B, 3, 512, 512, device=dev)


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for bf16 in (False, True):
    net = Network(args, pretrained=False, raw_output=True).to(dev).eval()
    net.bf16_inference = bf16
    with torch.no_grad():
        for sb in (True, False):
            net._engine.small_batch_kernel = sb
            eager = timed(lambda: net(x))
            if bf16:
                run = net.graphed(x)
                run.static_in.copy_(x)
                graph = timed(lambda: run(run.static_in))
                print(f"B={B} bf16 sb={int(sb)}: eager {eager:.3f} ms, hipGraph {graph:.3f} ms", flush=True)
                continue
            run = net.graphed(x)
            graph = timed(lambda: run(x))
            print(f"B={B} fp32 sb={int(sb)}: eager {eager:.3f} ms ({B * 45.15 / eager:.1f} TFLOP/s), hipGraph {graph:.3f} ms ({B * 45.15 / graph:.1f} TFLOP/s)", flush=True)
        if "--layers" in sys.argv and not bf16:
            net._engine.small_batch_kernel = True
            for _ in range(3):
                net._engine.prof = []
                net(x)
                torch.cuda.synchronize()
            rows = [(k, f, a.elapsed_time(b) * 1e3) for (k, f, a, b, _) in net._engine.prof]
            net._engine.prof = None
            for k, f, us in rows:
                print(f"    {k:28s} {f / 1e9:7.3f} GFLOP {us:7.1f} us {f / us / 1e6:6.1f} TFLOP/s")
            print(f"    sum of conv launches {sum(r[2] for r in rows):.1f} us")
