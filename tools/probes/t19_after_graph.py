"""The T = 19 step read 5.2 ms (1.8 ms on its own) at its place in bench.py's sequence of legs: which leg before it does that,
and what does the caching allocator do meanwhile?   python tools/probes/t19_after_graph.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
dev = torch.device("cuda:0")


def stats(tag):
    s = torch.cuda.memory_stats()
    print(f"   [{tag}] device allocs {s.get('num_device_alloc')} frees {s.get('num_device_free')} retries {s.get('num_alloc_retries')} "
          f"reserved {s.get('reserved_bytes.all.current', 0) >> 20} MiB active {s.get('active_bytes.all.current', 0) >> 20} MiB", flush=True)


def t19(tag):
    stats(tag + " before")
    print(tag, bench.measure_batch_sweep(dev, 60, batches=(32,), steps=30, warmup=8), flush=True)
    stats(tag + " after")


which = sys.argv[1] if len(sys.argv) > 1 else "config3,graph"
t19("t19 first")
for leg in which.split(","):
    if leg == "config3":
        c = bench.measure_config3(dev, 32, 55)
        print("config3", c["pipeline_ms"], c["joined_train_step"], flush=True)
    elif leg == "graph":
        print("graph", bench.measure_graph_step(dev, 32, 55)["replay_ms"], flush=True)
    elif leg == "sweep":
        print("sweep", bench.measure_batch_sweep(dev, 55), flush=True)
    torch.cuda.empty_cache()
    t19("t19 after " + leg)
