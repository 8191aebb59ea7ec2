#!/usr/bin/env python3
"""BatchNorm passes against torch kernels with the same streams on the same tensors (layer1 at bs = 64: M = 2^20 pixels x 64 channels; layer2: 2^18 x 128):
apply (1 read + 1 write [+ residual]), backward reduce (2 reads), backward apply (2 reads + 1 write).  Stream events, best of 5 x 10."""
import ctypes as C
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from structuredetector_amd import _lib as L

lib = L.lib()
dev = torch.device("cuda")


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


for (M, Cc) in ((1 << 20, 64), (1 << 18, 128)):
    for dt, tag in ((torch.bfloat16, "bf16"), (torch.float32, "fp32")):
        x = torch.randn(M, Cc, device=dev).to(dt); dy = torch.randn(M, Cc, device=dev).to(dt); y = torch.empty_like(x); dx = torch.empty_like(x)
        res = torch.randn(M, Cc, device=dev).to(dt)
        mean = torch.zeros(Cc, device=dev); invstd = torch.ones(Cc, device=dev); gamma = torch.ones(Cc, device=dev); beta = torch.zeros(Cc, device=dev)
        dg = torch.empty(Cc, device=dev); db = torch.empty(Cc, device=dev)
        ws = torch.empty(lib.sd_col_reduce_workspace_bytes(M, Cc), dtype=torch.uint8, device=dev)
        mb = x.numel() * x.element_size() / 1e6
        ap = lib.sd_bn_apply_bf16 if dt == torch.bfloat16 else lib.sd_bn_apply
        bw = lib.sd_bn_bwd_bf16 if dt == torch.bfloat16 else lib.sd_bn_bwd
        t_ap = timed(lambda: L.check(ap(x.data_ptr(), y.data_ptr(), M, Cc, mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 0, 1, 0, L.stream())))
        t_apr = timed(lambda: L.check(ap(x.data_ptr(), y.data_ptr(), M, Cc, mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), res.data_ptr(), 1, 0, L.stream())))
        t_bw = timed(lambda: L.check(bw(dy.data_ptr(), x.data_ptr(), 0, 2, M, Cc, mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), dx.data_ptr(), 0,
                                        dg.data_ptr(), db.data_ptr(), 0, ws.data_ptr(), ws.numel(), L.stream())))
        t_t2 = timed(lambda: torch.relu(x, out=y) if False else torch.add(x, 1.0, out=y))
        t_t3 = timed(lambda: torch.mul(x, dy, out=dx))
        print(f"M={M} C={Cc} {tag} ({mb:.0f} MB per tensor): apply {t_ap:7.1f} us = {2 * mb / t_ap:5.2f} TB/s, apply + residual {t_apr:7.1f} us = {3 * mb / t_apr:5.2f} TB/s, "
              f"backward (reduce + finalize + apply, 5 tensor passes) {t_bw:7.1f} us = {5 * mb / t_bw:5.2f} TB/s | torch add {t_t2:7.1f} us = {2 * mb / t_t2:5.2f} TB/s, "
              f"torch mul {t_t3:7.1f} us = {3 * mb / t_t3:5.2f} TB/s", flush=True)
