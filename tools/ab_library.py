#!/usr/bin/env python3
"""Same-box A/B of two builds of libsdnet_hip.so (a code change, not an option): the fp32 and the mixed-precision training step,
bs = 64, 512x512, each build in its own child process, alternating, best of the repetitions.
usage: ab_library.py <libA.so> <libB.so> [<libC.so> ...] [reps=3]"""
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CHILD = r'''
import sys, time
import numpy as np, torch
sys.path.insert(0, sys.argv[1])
from bench import make_args
from structuredetector_amd.data import Encode
from structuredetector_amd.data.synthetic import synthetic_batch
from structuredetector_amd.model import Network
from structuredetector_amd.model.trainer import TrainStep
dev = torch.device("cuda")
out = []
for amp in (False, True):
    args = make_args(dev); args.use_amp = amp
    torch.manual_seed(0)
    net = Network(args, pretrained=False).to(dev).train()
    step = TrainStep(net, args)
    enc = Encode(args)
    x = torch.randn(64, 3, 512, 512, device=dev)
    plan = enc.upload(enc.plan(512, 512, *synthetic_batch(np.random.default_rng(0), 64, 512, 512, 2, 1)))
    for _ in range(3): step(x, enc.render_device(plan))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 20 if amp else 8
    for _ in range(n): loss = step(x, enc.render_device(plan))
    torch.cuda.synchronize()
    out.append((time.perf_counter() - t0) / n * 1e3)
    out.append(float(loss[0]))
    del net, step, x
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
    print(f"{Path(lib).name:32s} fp32 step {min(x[0] for x in r):7.3f} ms ({64e3 / min(x[0] for x in r):6.1f} img/s, loss {r[0][1]:.6f})   "
          f"mixed precision {min(x[2] for x in r):7.3f} ms (loss {r[0][3]:.6f})   all: " + " ".join(f"{x[0]:.2f}/{x[2]:.2f}" for x in r))
