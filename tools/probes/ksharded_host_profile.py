"""Where the HOST spends a K-sharded rank's step (eager Python): cProfile over 30 emulated steps at per-GPU batch 32.
python tools/probes/ksharded_host_profile.py [per_gpu_batch]"""
import cProfile
import os
import pstats
import sys
import time
from unittest import mock

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench

b = int(sys.argv[1]) if len(sys.argv) > 1 else 32
prof = cProfile.Profile()
orig = bench.time_steps if hasattr(bench, "time_steps") else None
# measure_ksharded_rank_compute times `step` itself; run it once with a profiler wrapped around its timed loop
real_perf = time.perf_counter
state = {"n": 0}


def perf():
    state["n"] += 1
    if state["n"] == 1:
        prof.enable()
    elif state["n"] == 2:
        prof.disable()
    return real_perf()


with mock.patch.object(bench.time, "perf_counter", perf):
    out = bench.measure_ksharded_rank_compute(torch.device("cuda:0"), 55, per_gpu_batch=b, steps=30, warmup=5)
print(out["ms_per_step"], "ms per step (with the profiler's overhead)")
st = pstats.Stats(prof)
st.sort_stats("cumulative").print_stats(45)
