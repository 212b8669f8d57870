"""GPU check of csrc/conv_wino43b.hip against a float64 convolution and the fp32-MFMA F(4x4,3x3) kernel, plus timing.
   python scratch/w43b_check.py [time]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kpx_amd
from kpx_amd import ops
from kpx_amd._lib import lib, check

dev = torch.device('cuda:0')


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def ref64(x, w, b, act):
    y = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(3, 2, 0, 1), b.double() if b is not None else None, padding=1).permute(0, 2, 3, 1)
    if act == 1:
        y = torch.relu(y)
    elif act == 2:
        y = torch.nn.functional.leaky_relu(y, 0.01)
    return y


def run_new(x, cin, w, b, cout, act, dgrad=0, ldx=None, stats=False, mask=None, pool=False):
    n, h, wd = x.shape[:3]
    ldx = ldx or x.shape[3]
    u = torch.empty(lib.kpx_wino43b_u_bytes(w.shape[2], w.shape[3]), dtype=torch.uint8, device=dev)
    check(lib.kpx_wino43b_filter_transform_f32(w.data_ptr(), w.shape[2], w.shape[3], dgrad, u.data_ptr(), ops._stream()), 'tf')
    y = torch.full((n, h, wd, cout), float('nan'), device=dev)
    slab = torch.empty(lib.kpx_conv3x3_wino43_stats_tiles(n, h, wd) * 2 * cout, device=dev) if stats else None
    py = torch.empty((n, h // 2, wd // 2, cout), device=dev) if pool else None
    check(lib.kpx_conv3x3_wino43b_f32(x.data_ptr(), n, h, wd, cin, ldx, u.data_ptr(), b.data_ptr() if b is not None else None, y.data_ptr(), cout, cout, act,
                                      mask.data_ptr() if mask is not None else None, cout if mask is not None else 0, py.data_ptr() if pool else None, cout if pool else 0,
                                      slab.data_ptr() if stats else None, None, 0, None, ops._stream()), 'conv')
    return y, slab, py, u


def run_old(x, cin, w, b, cout, act, dgrad=0, ldx=None):
    n, h, wd = x.shape[:3]
    ldx = ldx or x.shape[3]
    u = torch.empty(lib.kpx_wino43_u_bytes(w.shape[2], w.shape[3]), dtype=torch.uint8, device=dev)
    check(lib.kpx_wino43_filter_transform_f32(w.data_ptr(), w.shape[2], w.shape[3], dgrad, u.data_ptr(), ops._stream()), 'tf')
    y = torch.empty((n, h, wd, cout), device=dev)
    check(lib.kpx_conv3x3_wino43_f32(x.data_ptr(), n, h, wd, cin, ldx, u.data_ptr(), b.data_ptr() if b is not None else None, y.data_ptr(), cout, cout, act, ops._stream()), 'conv')
    return y, u


def t_ms(fn, iters=50, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


CASES = [(2, 16, 16, 64, 64, 1), (4, 16, 16, 40, 72, 0), (2, 16, 32, 16, 64, 0), (1, 32, 32, 64, 64, 1), (2, 32, 64, 32, 40, 2), (1, 16, 64, 24, 96, 0), (2, 64, 64, 128, 128, 0), (1, 32, 32, 256, 128, 1)]
if len(sys.argv) < 2 or sys.argv[1] != 'time':
    for (n, h, wd, cin, cout, act) in CASES:
        g = torch.Generator().manual_seed(n * 1000 + cin + cout)
        x = torch.randn(n, h, wd, cin, generator=g).to(dev)
        w = (torch.randn(3, 3, cin, cout, generator=g) / (9 * cin) ** 0.5).to(dev)
        b = torch.randn(cout, generator=g).to(dev)
        want = ref64(x, w, b, act)
        y, _, _, _ = run_new(x, cin, w, b, cout, act)
        yo, _ = run_old(x, cin, w, b, cout, act)
        print('fwd  n%d %dx%d %d->%d act%d  new %.3e  old %.3e  (new vs old %.3e)' % (n, h, wd, cin, cout, act, rel(y, want), rel(yo, want), rel(y, yo)), flush=True)
        # data gradient: conv of dy [.., cout] with the flipped / transposed filter
        dy = torch.randn(n, h, wd, cout, generator=g).to(dev)
        if cout % 4 == 0 and cout >= 16 and cin >= 33:
            wt = w.flip(0, 1).permute(0, 1, 3, 2).contiguous()
            want = ref64(dy, wt, None, 0)
            dx, _, _, _ = run_new(dy, cout, w, None, cin, 0, dgrad=1)
            print('dgrad                              new %.3e' % rel(dx, want), flush=True)
    # statistics strips, mask, pool
    n, h, wd, c = 2, 32, 64, 64
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, h, wd, c, generator=g).to(dev); w = (torch.randn(3, 3, c, c, generator=g) / 24).to(dev); b = torch.randn(c, generator=g).to(dev)
    y, slab, _, _ = run_new(x, c, w, b, c, 0, stats=True)
    s = slab.view(-1, 2, c).double().sum(0)
    print('stats: sum %.3e  sumsq %.3e' % (rel(s[0], y.double().sum((0, 1, 2))), rel(s[1], (y.double() ** 2).sum((0, 1, 2)))))
    m = torch.randn(n, h, wd, c, generator=g).to(dev)
    y2, _, py, _ = run_new(x, c, w, b, c, 1, mask=m, pool=True)
    want = torch.relu(ref64(x, w, b, 0)) * (m > 0)
    print('mask+relu %.3e  pool %.3e' % (rel(y2, want), rel(py, torch.nn.functional.max_pool2d(want.permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1))))
    n, h, wd, c = 4, 16, 16, 64                          # packed 16x16 images with the mask / pool epilogue
    x = torch.randn(n, h, wd, c, generator=g).to(dev); m = torch.randn(n, h, wd, c, generator=g).to(dev)
    y2, _, py, _ = run_new(x, c, w, b, c, 1, mask=m, pool=True)
    want = torch.relu(ref64(x, w, b, 0)) * (m > 0)
    print('packed: mask+relu %.3e  pool %.3e' % (rel(y2, want), rel(py, torch.nn.functional.max_pool2d(want.permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1))))
else:
    shapes = [(int(a), int(b), int(c_), int(d)) for a, b, c_, d in (t.split(',') for t in sys.argv[2:])] if len(sys.argv) > 2 else None
    for (n, h, c, co) in shapes or [(32, 64, 128, 128), (32, 128, 64, 64), (32, 64, 256, 128), (32, 128, 128, 64), (64, 128, 64, 64), (64, 64, 128, 128), (64, 32, 256, 256), (32, 32, 256, 256), (64, 64, 64, 128), (64, 16, 512, 512), (64, 16, 256, 512), (64, 16, 256, 128)]:
        x = torch.randn(n, h, h, c, device=dev); w = torch.randn(3, 3, c, co, device=dev) * 0.03; b = torch.zeros(co, device=dev)
        y, _, _, u = run_new(x, c, w, b, co, 0)
        yo, uo = run_old(x, c, w, b, co, 0)
        yy = torch.empty_like(y)
        tn = t_ms(lambda: lib.kpx_conv3x3_wino43b_f32(x.data_ptr(), n, h, h, c, c, u.data_ptr(), b.data_ptr(), yy.data_ptr(), co, co, 0, None, 0, None, 0, None, None, 0, None, ops._stream()))
        to = t_ms(lambda: lib.kpx_conv3x3_wino43_f32(x.data_ptr(), n, h, h, c, c, uo.data_ptr(), b.data_ptr(), yy.data_ptr(), co, co, 0, ops._stream()))
        fl = 2.0 * 9 * c * co * h * h * n
        print('N%d %dx%d %d->%d : new %.4f ms (%.0f TF alg)  old %.4f ms (%.0f TF)  new/old %.2f  diff %.2e' % (n, h, h, c, co, tn, fl / tn / 1e9, to, fl / to / 1e9, tn / to, rel(y, yo)), flush=True)
