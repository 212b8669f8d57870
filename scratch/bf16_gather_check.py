"""Strided / 4x4 layers on bf16 tensors (kpx_conv2d_{fwd,dgrad,wgrad}_bf16) through ops.conv2d against the oracle on the rounded operands."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import kpx_amd
from kpx_amd import ops
from oracle import restatement as R
dev = torch.device('cuda:0')
def rel(a, b):
    a = a.astype(np.float64); b = b.astype(np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))
ops.set_compute_dtype('bf16')
ok = True
torch.manual_seed(2)
for (n, h, cin, cout, k, s, pad, act) in [(4, 33, 64, 128, 4, 2, 1, 2), (4, 18, 256, 512, 4, 2, 1, 2), (2, 32, 32, 64, 3, 2, 0, 0), (8, 6, 1024, 2048, 4, 2, 1, 2), (2, 64, 64, 128, 3, 2, 0, 0)]:
    for key in ops.fallback_uses: ops.fallback_uses[key] = 0
    x = torch.randn(n, h, h, cin, device=dev).bfloat16().requires_grad_(True)
    w = (torch.randn(k, k, cin, cout, device=dev) / (k * k * cin) ** 0.5).requires_grad_(True)
    b = torch.randn(cout, device=dev).requires_grad_(True)
    y = ops.conv2d(x, w, b, stride=s, pad=pad, act=act)
    gy = torch.randn(*y.shape, device=dev).bfloat16()
    y.backward(gy)
    torch.cuda.synchronize()
    xo = x.detach().float().cpu().requires_grad_(True); wo = w.detach().bfloat16().float().cpu().requires_grad_(True); bo = b.detach().cpu().requires_grad_(True)
    zo = R.conv(xo, wo, bo, s, pad)
    yo = torch.relu(zo) if act == 1 else torch.nn.functional.leaky_relu(zo, 0.01) if act == 2 else zo
    # backward of the oracle at the GPU's activation pattern and rounded gradient
    pos = (y.detach().float().cpu() > 0)
    fac = torch.where(pos, torch.tensor(1.0), torch.tensor(0.01)) if act == 2 else torch.ones_like(zo)
    gz = (gy.float().cpu() * fac).bfloat16().float() if act else gy.float().cpu()
    zo.backward(gz)
    e = [rel(y.detach().float().cpu().numpy(), yo.detach().bfloat16().float().numpy()), rel(x.grad.float().cpu().numpy(), xo.grad.bfloat16().float().numpy()),
         rel(w.grad.cpu().numpy(), wo.grad.numpy()), rel(b.grad.cpu().numpy(), bo.grad.numpy())]
    good = e[0] < 4e-3 and e[1] < 6e-3 and e[2] < 3e-3 and e[3] < 1e-4 and sum(ops.fallback_uses.values()) == 0
    ok &= good
    print((n, h, cin, cout, k, s), 'y %.2e dx %.2e dw %.2e db %.2e' % tuple(e), dict(ops.fallback_uses), 'ok' if good else 'FAIL')
print('ALL OK' if ok else 'FAILURES')
