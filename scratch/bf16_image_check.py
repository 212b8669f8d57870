"""bf16 configuration, image-input layers and the 4-channel head: the bf16-I/O entries against the fp32 entries on bf16-rounded operands."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kpx_amd
from kpx_amd import ops
dev = torch.device('cuda:0')
BF = torch.bfloat16
def rel(a, b): return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
g = torch.Generator(device='cpu').manual_seed(5)
def rnd(*s): return torch.randn(*s, generator=g).to(dev)

for (name, n, hi, cin, cout, k, st, pt, pl, act) in [('pose conv_1 7x7', 8, 128, 3, 32, 7, 1, 3, 3, ops.ACT_NONE), ('D conv_0 4x4/s2', 8, 128, 3, 64, 4, 2, 1, 1, ops.ACT_LRELU),
                                                       ('vgg conv1_1', 8, 128, 3, 64, 3, 1, 1, 1, ops.ACT_RELU)]:
    ho = hi // st
    x = rnd(n, hi, hi, cin); w = rnd(k, k, cin, cout) * 0.1; b = rnd(cout) * 0.1
    y32 = torch.empty(n, ho, ho, cout, device=dev); y16 = torch.empty(n, ho, ho, cout, device=dev, dtype=BF)
    ops.conv_fwd_raw(x, cin, cin, w, b, y32, cout, st, pt, pl, act)
    before = dict(ops.fallback_uses)
    ops.conv_fwd_raw(x, cin, cin, w, b, y16, cout, st, pt, pl, act)
    print(name, 'fwd bf16-out vs fp32', rel(y16.float(), y32), 'exact-after-rounding', bool((y16 == y32.to(BF)).all()))
    dy = rnd(n, ho, ho, cout); dy16 = dy.to(BF); dyr = dy16.float()
    dw32 = torch.empty_like(w); dw16 = torch.empty_like(w)
    ops.conv_wgrad_raw(x, cin, cin, dyr, cout, dw32, st, pt, pl)
    ops.conv_wgrad_raw(x, cin, cin, dy16, cout, dw16, st, pt, pl)
    print('   wgrad bf16-dy vs fp32 on rounded dy', rel(dw16, dw32))
    if k != 7:
        dx32 = torch.empty_like(x); dx16 = torch.empty_like(x)
        ops.conv_dgrad_raw(dyr, cout, w, dx32, cin, cin, st, pt, pl)
        ops.conv_dgrad_raw(dy16, cout, w, dx16, cin, cin, st, pt, pl)
        print('   dgrad bf16-dy vs fp32 on rounded dy', rel(dx16, dx32))
    print('   fallbacks', {k2: ops.fallback_uses[k2] - before[k2] for k2 in before})

# 4-channel head through autograd
ops.set_compute_dtype('bf16')
n, h, cin, cout = 8, 128, 64, 4
x = rnd(n, h, h, cin).to(BF); w = (rnd(3, 3, cin, cout) * 0.05).requires_grad_(True); b = (rnd(cout) * 0.1).requires_grad_(True)
xr = x.clone().requires_grad_(True)
before = dict(ops.fallback_uses)
y = ops.conv2d(xr, w, b, 1, 0, ops.ACT_NONE, out_dtype=torch.float32)
dy = rnd(n, h, h, cout)
y.backward(dy); ops.join_side_stream()
torch.cuda.synchronize()
print('head fallbacks', {k2: ops.fallback_uses[k2] - before[k2] for k2 in before})
ops.set_compute_dtype('f32')
x32 = x.float().requires_grad_(True); w32 = w.detach().clone().requires_grad_(True); b32 = b.detach().clone().requires_grad_(True)
y2 = ops.conv2d(x32, w32, b32, 1, 0, ops.ACT_NONE)
y2.backward(dy.to(BF).float()); torch.cuda.synchronize()
print('head y', rel(y, y2), 'dx', rel(xr.grad.float(), x32.grad), 'dw', rel(w.grad, w32.grad), 'db', rel(b.grad, b32.grad))
