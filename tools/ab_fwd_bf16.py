#!/usr/bin/env python3
"""Same-box A/B of two builds of libsdnet_hip.so on the bf16 eval forward (bs = 64, 512x512) and the stress forward (bs = 16, 1024x1024,
8 + 8 maps): each build in its own child process, alternating, best of the repetitions.  Timing-only experiment builds are allowed
(SDNET_ALLOW_ABLATION=1): the outputs are not compared.
usage: ab_fwd_bf16.py <libA.so> <libB.so> [<libC.so> ...] [reps=3]"""
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CHILD = r'''
import sys, time
import torch
sys.path.insert(0, sys.argv[1])
from bench import make_args
from structuredetector_amd.model import Network
dev = torch.device("cuda")
out = []
for (B, S, M, N, K, P) in ((64, 512, 2, 1, 20, 40), (16, 1024, 8, 8, 128, 512)):
    args = make_args(dev, M, N, K, P); args.use_amp = True
    torch.manual_seed(0)
    net = Network(args, pretrained=False).to(dev).eval()
    net.bf16_inference = True
    x = torch.randn(B, 3, S, S, device=dev)
    with torch.no_grad():
        for _ in range(5): net(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30): net(x)
        torch.cuda.synchronize()
    out.append((time.perf_counter() - t0) / 30 * 1e3)
    del net, x
print("RESULT", *out)
'''
libs = [str(Path(p).resolve()) for p in sys.argv[1:] if p.endswith(".so")]          # two or more builds
reps = int(sys.argv[-1]) if sys.argv[-1].isdigit() else 3
res = {lib: [] for lib in libs}
for _ in range(reps):
    for lib in libs:
        env = dict(os.environ, SDNET_HIP_LIB=lib, SDNET_ALLOW_ABLATION="1")
        out = subprocess.run([sys.executable, "-c", CHILD, str(ROOT)], env=env, capture_output=True, text=True, timeout=900)
        line = [ln for ln in out.stdout.splitlines() if ln.startswith("RESULT")]
        if not line:
            print(out.stderr[-1500:]); raise SystemExit(1)
        res[lib].append([float(v) for v in line[0].split()[1:]])
for lib in libs:
    r = res[lib]
    print(f"{Path(lib).name:32s} bf16 forward bs=64 512x512 {min(x[0] for x in r):7.3f} ms   stress bs=16 1024x1024 {min(x[1] for x in r):7.3f} ms   all: "
          + " ".join(f"{x[0]:.3f}/{x[1]:.3f}" for x in r), flush=True)
