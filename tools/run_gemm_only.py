"""Workload for PMC passes over the tiled GEMM (gemm_bf16x3_kernel): the weight-gradient product dW = dY^T X of a
64 -> 1024 linear over 19 456 rows (K = 19 456 cut over workgroups), 20 launches.   python tools/run_gemm_only.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from predict_pv_yield_amd import hip_ops as K
dev = torch.device("cuda:0")
x, dy = torch.randn(19456, 64, device=dev), torch.randn(19456, 1024, device=dev)
for _ in range(20):
    K.gemm_splitk(dy.t(), x)
torch.cuda.synchronize()
