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
# three streams, bf16 (the shape of a BatchNorm backward apply pass: two reads, one write, 16 bytes per lane)
n2 = 1 << 28
x16 = torch.empty(n2, device=dev, dtype=torch.bfloat16); y16 = torch.empty_like(x16); z16 = torch.empty_like(x16)
t = timed(lambda: torch.mul(x16, y16, out=z16)); print(f"mul bf16 (2 r + 1 w) {t * 1e3:8.1f} us  {3 * n2 * 2 / 1e9 / t:6.2f} TB/s total")
x32 = torch.empty(n2 // 2, device=dev); y32 = torch.empty_like(x32); z32 = torch.empty_like(x32)
t = timed(lambda: torch.mul(x32, y32, out=z32)); print(f"mul fp32 (2 r + 1 w) {t * 1e3:8.1f} us  {3 * (n2 // 2) * 4 / 1e9 / t:6.2f} TB/s total")
