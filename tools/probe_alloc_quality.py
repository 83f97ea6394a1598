"""Is streaming bandwidth a property of the ALLOCATION?  Twelve 490 MiB buffers, an in-place scale (read + write) timed on
each, three rounds.   python tools/probe_alloc_quality.py"""
import torch
dev = torch.device("cuda:0")
n = 128 * 1003520
bufs = [torch.empty(n, device=dev) for _ in range(12)]
for b in bufs:
    b.fill_(1.0)


def t_scale(b, it=10):
    for _ in range(2):
        b.mul_(1.0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        b.mul_(1.0)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


for rnd in range(3):
    print("round", rnd, " ".join(f"{t_scale(b):6.1f}" for b in bufs))
print("ptr>>21", " ".join(hex(b.data_ptr() >> 21) for b in bufs))
