import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["PV_YIELD_LIB"] = os.path.join(ROOT, "predict_pv_yield_amd", "lib", "libpvyield_diag.so")
import numpy as np, torch
from predict_pv_yield_amd import hip_ops as K
from predict_pv_yield_amd._lib import get_lib
d = "cuda"
b, nq, nk, h = 152, 128, 16384, 1
q = torch.randn(b, nq, 64, device=d); kv = torch.randn(b, nk, 128, device=d)
for name, kvx in (("f32 K/V", kv), ("bf16 K/V", kv.to(torch.bfloat16))):
    for _ in range(3): K.attention_fwd(q, kvx, h, 0.125, bf16_operands=True)
    torch.cuda.synchronize()
    buf = np.zeros((1 << 14) * 8, dtype=np.uint64)
    f = get_lib().pv_diag_read_attn_fwd; f.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    assert f(buf.ctypes.data, buf.size) == 0
    w = buf.reshape(-1, 8).astype(np.float64); w = w[w[:, 7] > 0]
    labels = ["barrier at the tile's top", "K / V arrive + conversion + LDS write", "barrier after the LDS write", "next tile's loads issued + 4 QK products",
              "softmax", "pack + 4 PV products"]
    tot = w[:, :6].sum(1).mean()
    print(f"{name}: {len(w)} waves, {w[:, 7].mean():.0f} tiles per wave, {tot / w[:, 7].mean():.0f} cycles / tile")
    for i, lab in enumerate(labels):
        print(f"   {lab:45s} {(w[:, i] / w[:, 7]).mean():8.0f} cycles / tile  {100 * w[:, i].mean() / tot:5.1f} %")
