"""bf16 3x3 weight-gradient kernel (csrc/conv_bf16s_wgrad.hip) against the oracle on the bf16-rounded operands; launch times of bench layers."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import kpx_amd
from kpx_amd import ops
from kpx_amd._lib import lib, check
from oracle import restatement as R
dev = torch.device('cuda:0')

def rel(a, b):
    a = a.astype(np.float64); b = b.astype(np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))

def run(x, dy, cin, cout):
    n, h, w, ldx = x.shape; lddy = dy.shape[3]
    dw = torch.full((3, 3, cin, cout), 7.0, device=dev)
    nbytes = lib.kpx_conv3x3_wgrad_bf16_workspace_bytes(n, h, w, cin, cout)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    check(lib.kpx_conv3x3_wgrad_bf16(x.data_ptr(), n, h, w, cin, ldx, dy.data_ptr(), cout, lddy, dw.data_ptr(), ws.data_ptr(), nbytes, ops._stream()), 'wgrad')
    return dw, ws

ok = True
torch.manual_seed(1)
for (n, h, w_, cin, cout, ldx) in [(2, 16, 32, 64, 128, 64), (2, 32, 32, 64, 64, 64), (3, 16, 16, 128, 128, 128), (2, 8, 32, 32, 32, 32), (2, 32, 32, 16, 16, 16),
                                   (2, 16, 32, 64, 16, 64), (1, 32, 32, 158, 256, 160), (2, 64, 64, 128, 64, 128), (2, 16, 32, 32, 128, 32), (4, 16, 16, 256, 256, 256)]:
    x = torch.randn(n, h, w_, ldx, device=dev).bfloat16()
    if ldx > cin: x[..., cin:] = 0
    dy = torch.randn(n, h, w_, cout, device=dev).bfloat16()
    if not lib.kpx_conv3x3_wgrad_bf16_eligible(n, h, w_, cin, ldx, cout, cout, x.data_ptr(), dy.data_ptr()):
        print((n, h, w_, cin, cout), 'not eligible'); continue
    dw, _ = run(x, dy, cin, cout)
    torch.cuda.synchronize()
    wo = torch.zeros(3, 3, cin, cout, requires_grad=True)
    z = R.conv(x[..., :cin].float().cpu(), wo, None, 1, 0)
    z.backward(dy.float().cpu())
    e = rel(dw.cpu().numpy(), wo.grad.numpy())
    ok &= e < 1e-5
    print('%-30s wgrad %.2e %s' % ((n, h, w_, cin, cout), e, 'ok' if e < 1e-5 else 'FAIL'))
print('ALL OK' if ok else 'FAILURES')
if len(sys.argv) > 1:
    def tm(fn, iters=30, warm=5):
        for _ in range(warm): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters
    for (n, h, cin, cout, name) in [(32, 64, 128, 128, 'translator conv_3_1'), (32, 64, 256, 128, 'translator conv_3_0'), (32, 32, 256, 256, 'translator conv_1_1'),
                                    (32, 128, 64, 64, 'translator conv_5_1'), (32, 128, 128, 64, 'translator conv_5_0'), (64, 16, 128, 128, 'pose conv_1_1'), (64, 64, 32, 32, 'pose conv_5_1'),
                                    (64, 128, 64, 16, 'pose conv_7_0'), (64, 128, 16, 16, 'pose conv_7_1'), (64, 128, 32, 32, 'enc conv_2'), (64, 32, 128, 128, 'enc conv_6')]:
        x = torch.randn(n, h, h, cin, device=dev).bfloat16(); dy = torch.randn(n, h, h, cout, device=dev).bfloat16()
        dw = torch.empty(3, 3, cin, cout, device=dev)
        nbytes = lib.kpx_conv3x3_wgrad_bf16_workspace_bytes(n, h, h, cin, cout)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        f = lambda: lib.kpx_conv3x3_wgrad_bf16(x.data_ptr(), n, h, h, cin, cin, dy.data_ptr(), cout, cout, dw.data_ptr(), ws.data_ptr(), nbytes, ops._stream())
        ms = tm(f)
        fl = 2.0 * 9 * cin * cout * h * h * n
        print('%-22s N=%d %dx%d %d->%d: %.4f ms  %.0f TF  (slabs %.0f MB)' % (name, n, h, h, cin, cout, ms, fl / ms / 1e9, nbytes / 1e6))
