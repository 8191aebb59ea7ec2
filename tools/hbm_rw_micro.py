#!/usr/bin/env python3
"""HBM rates of plain torch kernels on 1 GiB buffers (stream events, best of 5 x 5): a write-only fill, a read-only sum, a copy (read + write).
The write-heavy kernels of this library (stem forward, BatchNorm apply, pool backward) top out near the fill's rate."""
import torch
dev = torch.device("cuda")
n = 1 << 28                                     # 1 GiB of fp32
a = torch.empty(n, device=dev); b = torch.empty(n, device=dev)


def timed(fn, reps=5, k=5):
    for _ in range(2):
        fn()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(k):
            fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / k)
    return best


gb = n * 4 / 1e9
t = timed(lambda: a.fill_(1.0)); print(f"fill (write only)   {t * 1e3:8.1f} us  {gb / t:6.2f} TB/s written")
t = timed(lambda: a.sum());     print(f"sum (read only)     {t * 1e3:8.1f} us  {gb / t:6.2f} TB/s read")
t = timed(lambda: b.copy_(a));  print(f"copy (read + write) {t * 1e3:8.1f} us  {2 * gb / t:6.2f} TB/s total")
t = timed(lambda: torch.add(a, 1.0, out=b)); print(f"add scalar (r + w)  {t * 1e3:8.1f} us  {2 * gb / t:6.2f} TB/s total")
