import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from predict_pv_yield_amd import hip_ops as K
x = torch.randn(32, 32, 16, 62, 62, device="cuda")
for _ in range(10):
    K.pack_split2_ncdhw_f32_to_ndhwc_f16(x)
    K.pack_split3_ncdhw_f32_to_ndhwc_bf16(x)
torch.cuda.synchronize()
