"""bench.py's matched-validation leg (val_nmae + cpu_baseline) on its own: python tools/run_val_experiment.py [seeds] [steps]"""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

seeds = int(sys.argv[1]) if len(sys.argv) > 1 and int(sys.argv[1]) > 0 else None      # None / 0: the sequential rule
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 512
oracle_s = float(sys.argv[3]) if len(sys.argv) > 3 else 60.0
val, cpu = bench.matched_validation_and_cpu_baseline(torch.device("cuda:0"), 55, seeds=seeds, n_steps=steps, oracle_seconds=oracle_s)
print(json.dumps({"val_nmae": val, "cpu_baseline": cpu}, indent=1))
