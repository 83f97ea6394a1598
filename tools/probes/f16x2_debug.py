import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from predict_pv_yield_amd import hip_ops as K
dev = torch.device("cuda:0")
shape, pad, mag = (6, 32, 6, 40, 52), (1, 1, 1), 3e-6
g = torch.Generator().manual_seed(1)
x = torch.randn(shape, generator=g).abs() * mag
wt = torch.randn(32, 32, 3, 3, 3, generator=g) * 0.05
bias = torch.randn(32, generator=g) * 0.1 * mag
ref = F.relu(F.conv3d(x.double(), wt.double(), bias.double(), padding=pad))
xd = x.to(dev)
xh, xl, xs = K.pack_split2_ncdhw_f32_to_ndhwc_f16(xd)
wp, ws = K.conv3d_pack_weight_split2_f16(wt.to(dev))
for trial in range(3):
    y, st = K.conv3d_f32_on_f16x2(xh, xl, xs, wp[0], wp[1], ws, 32, 32, pad, bias=bias.to(dev), relu=True, want_max=True)
    torch.cuda.synchronize()
    ymax = y.abs().max()
    smax = st[0:1].view(torch.int32).view(torch.float32) if False else st[0:1]
    err = (y.cpu().double() - ref).abs()
    print("trial", trial, "y max", float(ymax), "state", float(st[0]), "ref max", float(ref.max()), "max err", float(err.max()),
          "n above state", int((y > st[0]).sum()))
    idx = torch.nonzero(y > st[0])
    print(idx[:10].tolist())
    bad = torch.nonzero(err > 1e-9)
    print("bad elements", bad.shape[0], bad[:10].tolist())
print("---- the test's case")
from predict_pv_yield_amd import functional as Fn
g = torch.Generator().manual_seed(sum(shape) + pad[0])
x = (torch.randn(shape, generator=g).abs() * mag)
x[torch.rand(shape, generator=g) < 0.3] = 0.0
wt = torch.randn(32, 32, 3, 3, 3, generator=g) * 0.05
bias = torch.randn(32, generator=g) * 0.1 * mag
ref = F.relu(F.conv3d(x.double(), wt.double(), bias.double(), padding=pad))
xd = x.to(dev).requires_grad_(True)
wd, bd = wt.to(dev).requires_grad_(True), bias.to(dev).requires_grad_(True)
y = Fn.conv3d_general_f32(xd, wd, bd, stride=(1, 1, 1), padding=pad, relu=True, x_is_relu_output=True, dy_pregated=False)
st = y._pv_maxabs
yd = y.detach()
err = (yd.cpu().double() - ref).abs()
print("y max", float(yd.abs().max()), "state", float(st[0]), "ref max", float(ref.max()), "max err", float(err.max()), "n above", int((yd > st[0]).sum()))
idx = torch.nonzero(yd > st[0])
print(idx[:20].tolist())
print("values", [float(yd[tuple(i)]) for i in idx[:5]], "ref", [float(ref[tuple(i)]) for i in idx[:5]])
print("---- direct API on the test's data")
xh, xl, xs = K.pack_split2_ncdhw_f32_to_ndhwc_f16(x.to(dev))
wp, ws = K.conv3d_pack_weight_split2_f16(wt.to(dev))
top = torch.topk(yd.flatten(), 4).values.tolist()
print("top-4 of y", top)
for trial in range(4):
    y2, st2 = K.conv3d_f32_on_f16x2(xh, xl, xs, wp[0], wp[1], ws, 32, 32, pad, bias=bias.to(dev), relu=True, want_max=True)
    torch.cuda.synchronize()
    print("direct", trial, "equal y", bool(torch.equal(y2, yd)), "state", float(st2[0]))
for trial in range(3):
    y3 = Fn.conv3d_general_f32(xd, wd, bd, stride=(1, 1, 1), padding=pad, relu=True, x_is_relu_output=True, dy_pregated=False)
    print("autograd", trial, "state", float(y3._pv_maxabs[0]), "ymax", float(y3.detach().max()))
