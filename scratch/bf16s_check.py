"""bf16-storage 3x3 kernel (csrc/conv_bf16s.hip): parity on a set of shapes against the oracle convolution evaluated on the bf16-rounded
operands, LDS-DMA out-of-range semantics, and launch times of the bench layers.  python scratch/bf16s_check.py [time]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import kpx_amd
from kpx_amd import ops
from kpx_amd._lib import lib, check
from oracle import restatement as R

dev = torch.device('cuda:0')

def run(x16, w, bias, act, dgrad=False, out_f32=False, mask=None, stats=False, nn=None, ldout=None):
    n, h, wd, k = x16.shape
    cin, cout = w.shape[2], w.shape[3]
    nn = (cin if dgrad else cout) if nn is None else nn
    kk = cout if dgrad else cin
    wf = torch.empty(lib.kpx_conv3x3_bf16s_weights_bytes(kk, nn), dtype=torch.uint8, device=dev)
    check(lib.kpx_conv3x3_bf16s_prepare_f32(w.data_ptr(), cin, cout, 1 if dgrad else 0, wf.data_ptr(), ops._stream()), 'prep')
    ld = nn if ldout is None else ldout
    out = torch.full((n, h, wd, ld), 7.0, dtype=torch.float32 if out_f32 else torch.bfloat16, device=dev)
    st = None
    if stats:
        tiles = lib.kpx_conv3x3_bf16s_stats_tiles(n, h, wd, kk, nn)
        st = torch.zeros(tiles * 2 * nn, dtype=torch.float32, device=dev)
    rc = lib.kpx_conv3x3_bf16s(x16.data_ptr(), n, h, wd, kk, x16.stride(2), wf.data_ptr(), bias.data_ptr() if bias is not None else None, out.data_ptr(), nn, ld,
                               1 if out_f32 else 0, act, mask.data_ptr() if mask is not None else None, mask.shape[3] if mask is not None else 0,
                               st.data_ptr() if st is not None else None, ops._stream())
    check(rc, 'conv')
    return out, st, wf

def rel(a, b):
    a = a.astype(np.float64); b = b.astype(np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))

def oracle(x16, w, bias, act):
    xr = x16.float().cpu(); wr = w.bfloat16().float().cpu()
    z = R.conv(xr, wr, bias.cpu() if bias is not None else None, 1, 0)
    return torch.relu(z) if act == 1 else torch.nn.functional.leaky_relu(z, 0.01) if act == 2 else z

torch.manual_seed(0)
CASES = [(2, 16, 32, 32, 128, 1), (2, 32, 64, 64, 128, 0), (1, 32, 32, 160, 256, 1), (3, 16, 32, 16, 64, 0), (2, 32, 32, 64, 32, 1), (2, 32, 32, 32, 16, 0),
         (4, 16, 16, 128, 128, 1), (8, 8, 8, 64, 128, 0), (2, 64, 64, 128, 128, 1), (2, 16, 16, 256, 64, 2), (16, 8, 8, 32, 32, 0)]
ok = True
for (n, h, w_, cin, cout, act) in CASES:
    x = torch.randn(n, h, w_, cin, device=dev).bfloat16()
    wt = (torch.randn(3, 3, cin, cout, device=dev) / (9 * cin) ** 0.5)
    b = torch.randn(cout, device=dev)
    if not lib.kpx_conv3x3_bf16s_eligible(n, h, w_, cin, cout, cin, x.data_ptr()):
        print('case', (n, h, w_, cin, cout), 'not eligible'); continue
    want = oracle(x, wt, b, act).numpy()
    got32, _, _ = run(x, wt, b, act, out_f32=True)
    got16, st, _ = run(x, wt, b, act, stats=True)
    torch.cuda.synchronize()
    e32 = rel(got32.cpu().numpy(), want)
    e16 = rel(got16.float().cpu().numpy(), torch.from_numpy(want).bfloat16().float().numpy())
    # statistics: sums of the pre-activation outputs
    z = oracle(x, wt, b, 0).double().numpy().reshape(-1, cout)
    tiles = st.numel() // (2 * cout)
    s = st.view(tiles, 2, cout).double().sum(0).cpu().numpy()
    es = max(rel(s[0], z.sum(0)), rel(s[1], (z * z).sum(0)))
    # dgrad: conv with flipped, transposed filters == gradient of the forward
    dy = torch.randn(n, h, w_, cout, device=dev).bfloat16()
    xo = torch.zeros(n, h, w_, cin, requires_grad=True)
    zo = R.conv(xo, wt.bfloat16().float().cpu(), None, 1, 0)
    zo.backward(dy.float().cpu())
    if lib.kpx_conv3x3_bf16s_eligible(n, h, w_, cout, cin, cout, dy.data_ptr()) and cin % 8 == 0:
        dx, _, _ = run(dy, wt, None, 0, dgrad=True, out_f32=True) if cin % 4 == 0 else (None, None, None)
        torch.cuda.synchronize()
        ed = rel(dx.cpu().numpy(), xo.grad.numpy())
    else:
        ed = float('nan')
    good = e32 < 1e-5 and e16 < 3e-3 and es < 1e-4 and (ed != ed or ed < 1e-5)
    ok &= good
    print('%-28s fp32-out %.2e  bf16-out vs rounded %.2e  stats %.2e  dgrad %.2e  %s' % ((n, h, w_, cin, cout, act), e32, e16, es, ed, 'ok' if good else 'FAIL'))
# mask + channel-slice input / strided output
x_full = torch.randn(2, 32, 32, 160, device=dev).bfloat16()
wt = torch.randn(3, 3, 128, 64, device=dev) * 0.03
m = torch.randn(2, 32, 32, 64, device=dev).bfloat16()
out, _, _ = run(x_full[..., 16:144], wt, None, 0, mask=m, ldout=96)
want = oracle(x_full[..., 16:144], wt, None, 0) * (m.float().cpu() > 0)
torch.cuda.synchronize()
e = rel(out[..., :64].float().cpu().numpy(), want.bfloat16().float().numpy())
keep = bool((out[..., 64:].float() == 7.0).all())
print('slice + mask + strided out: %.2e, untouched tail %s' % (e, keep)); ok &= e < 3e-3 and keep
print('ALL OK' if ok else 'FAILURES')

if len(sys.argv) > 1:
    def tm(fn, iters=50, warm=10):
        for _ in range(warm): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters
    for (n, h, cin, cout, name) in [(32, 64, 128, 128, 'translator conv_3_1'), (32, 64, 256, 128, 'translator conv_3_0'), (32, 32, 256, 256, 'translator conv_1_1'),
                                    (32, 128, 64, 64, 'translator conv_5_1'), (64, 128, 64, 64, 'vgg conv1_2'), (64, 32, 256, 256, 'vgg conv3_2'),
                                    (64, 16, 512, 512, 'vgg conv4_2'), (64, 8, 512, 512, 'vgg conv5_1'), (64, 16, 128, 128, 'pose conv_1_1'), (64, 64, 32, 32, 'pose conv_5_1'),
                                    (64, 128, 64, 16, 'pose conv_7_0'), (64, 128, 32, 32, 'enc conv_2')]:
        x = torch.randn(n, h, h, cin, device=dev).bfloat16()
        wt = torch.randn(3, 3, cin, cout, device=dev) * 0.03
        b = torch.zeros(cout, device=dev)
        if not lib.kpx_conv3x3_bf16s_eligible(n, h, h, cin, cout, cin, x.data_ptr()):
            print(name, 'not eligible'); continue
        _, _, wf = run(x, wt, b, 1)
        out = torch.empty(n, h, h, cout, dtype=torch.bfloat16, device=dev)
        f = lambda: lib.kpx_conv3x3_bf16s(x.data_ptr(), n, h, h, cin, cin, wf.data_ptr(), b.data_ptr(), out.data_ptr(), cout, cout, 0, 1, None, 0, None, ops._stream())
        ms = tm(f)
        fl = 2.0 * 9 * cin * cout * h * h * n
        by = n * h * h * (cin + cout) * 2
        print('%-22s N=%d %dx%d %d->%d: %.4f ms  %.0f TF (%.2f of 2.5 PF)  %.2f TB/s algorithmic' % (name, n, h, h, cin, cout, ms, fl / ms / 1e9, fl / ms / 1e9 / 2500, by / ms / 1e9))
