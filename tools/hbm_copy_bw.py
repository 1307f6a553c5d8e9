#!/usr/bin/env python3
"""Measured device-memory bandwidth of the box (SURVEY.md section 8d asks for it next to the 8 TB/s spec figure):
device-to-device copy and a fill over 4 GiB, timed with events."""
import torch

assert torch.cuda.is_available()
n = 1 << 30                                   # 4 GiB of int32
a = torch.ones(n, dtype=torch.int32, device="cuda")
b = torch.empty_like(a)
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for name, fn, bytes_moved in (("copy (read + write)", lambda: b.copy_(a), 2 * 4 * n),
                              ("fill (write only)", lambda: b.fill_(7), 4 * n)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev0.record()
    reps = 10
    for _ in range(reps):
        fn()
    ev1.record()
    torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / reps
    print(f"{name:22s} {bytes_moved / ms / 1e9:8.2f} TB/s  ({ms:.3f} ms per {bytes_moved / 2**30:.0f} GiB)")
print(torch.cuda.get_device_name(0))
