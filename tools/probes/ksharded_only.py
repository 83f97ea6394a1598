"""python tools/probes/ksharded_only.py <per_gpu_batch>: the K-sharded one-rank emulation alone (for rocprofv3 --kernel-trace)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
print(json.dumps(bench.measure_ksharded_rank_compute(torch.device("cuda:0"), 55, per_gpu_batch=int(sys.argv[1]), steps=20, warmup=5)))
