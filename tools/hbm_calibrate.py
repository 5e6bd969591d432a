#!/usr/bin/env python3
"""What this box's HBM delivers to simple kernels (torch ops): device-to-device copy (read + write), read-only
reduction, write-only fill. Calibration for the roofline fractions in DESIGN.md; the 8 TB/s peak is the guide's."""
import time
import torch
dev = torch.device("cuda", 0)
n = 1 << 30  # 1 GiB
a = torch.empty(n, dtype=torch.uint8, device=dev)
b = torch.empty(n, dtype=torch.uint8, device=dev)
a.fill_(1)
def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
t = timed(lambda: b.copy_(a))
print(f"copy 1 GiB: {t*1e3:.3f} ms -> {2*n/t/1e12:.2f} TB/s (read + write)")
af = a.view(torch.int32)
t = timed(lambda: af.sum())
print(f"sum  1 GiB: {t*1e3:.3f} ms -> {n/t/1e12:.2f} TB/s (read)")
t = timed(lambda: b.fill_(7))
print(f"fill 1 GiB: {t*1e3:.3f} ms -> {n/t/1e12:.2f} TB/s (write)")
# the mix of the piece kernel: 2/3 reads, 1/3 writes
c = torch.empty(n // 2, dtype=torch.uint8, device=dev)
t = timed(lambda: torch.add(a[: n // 2], a[n // 2:], out=c))
print(f"add  (1 GiB in, 0.5 GiB out): {t*1e3:.3f} ms -> {1.5*n/t/1e12:.2f} TB/s (2:1 read:write)")
