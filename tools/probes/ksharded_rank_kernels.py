"""One rank's kernels of the K-sharded 8-GPU run on this GPU (bench.measure_ksharded_rank_compute), next to the row-sharded mix:
python tools/probes/ksharded_rank_kernels.py [per_gpu_batch ...]"""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench

dev = torch.device("cuda:0")
for b in [int(a) for a in sys.argv[1:]] or [64, 32]:
    print("ksharded", b, json.dumps(bench.measure_ksharded_rank_compute(dev, 55, per_gpu_batch=b)))
    print("sharded ", b, json.dumps(bench.measure_sharded_rank_compute(dev, 55, global_batch=8 * b)))
