"""Calibration: what a plain device-to-device copy of the fc1-update volume reaches on this box (read + write bytes)."""
import torch
dev = torch.device("cuda:0")
n = 128 * 1003520
a = torch.empty(n, dtype=torch.float32, device=dev).normal_()
b = torch.empty_like(a)
for nbytes_label, fn, vol in [("copy 0.51 GB -> 0.51 GB", lambda: b.copy_(a), 2 * 4 * n),
                              ("read-only sum 0.51 GB", lambda: a.sum(), 4 * n),
                              ("fill 0.51 GB", lambda: b.zero_(), 4 * n)]:
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"{nbytes_label}: {ms * 1e3:7.1f} us  {vol / ms / 1e9:5.2f} TB/s")
