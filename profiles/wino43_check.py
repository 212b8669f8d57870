"""F(4x4,3x3) kernel: accuracy against a float64 convolution and timing beside the F(2x2,3x3) kernel."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import kpx_amd
from kpx_amd import ops
from kpx_amd._lib import lib, check
dev = torch.device('cuda:0')
torch.manual_seed(0)
def run43(x, w, bias, act, dgrad=False, cin=None):
    n, h, wd, ld = x.shape
    k = cin or ld
    nn = w.shape[2] if dgrad else w.shape[3]
    u = torch.empty(lib.kpx_wino43_u_bytes(w.shape[2], w.shape[3]) // 4, device=dev)
    check(lib.kpx_wino43_filter_transform_f32(w.data_ptr(), w.shape[2], w.shape[3], int(dgrad), u.data_ptr(), ops._stream()), 'xf')
    y = torch.empty(n, h, wd, nn, device=dev)
    assert lib.kpx_conv3x3_wino43_eligible(n, h, wd, k, nn, ld, x.data_ptr())
    check(lib.kpx_conv3x3_wino43_f32(x.data_ptr(), n, h, wd, k, ld, u.data_ptr(), bias.data_ptr() if bias is not None else None, y.data_ptr(), nn, nn, act, ops._stream()), 'conv')
    return y, u
def ref64(x, w, bias, act, dgrad=False):
    xx = x.double().permute(0, 3, 1, 2)
    if dgrad:
        ww = w.double().flip(0, 1).permute(2, 3, 0, 1)      # [cin, cout, kh, kw] -> conv with swapped roles
    else:
        ww = w.double().permute(3, 2, 0, 1)
    y = torch.nn.functional.conv2d(xx, ww, bias.double() if bias is not None else None, padding=1).permute(0, 2, 3, 1)
    if act == 1: y = y.relu()
    if act == 2: y = torch.nn.functional.leaky_relu(y, 0.01)
    return y
ok = True
for (n, h, wd, cin, cout, act, dg) in [(2, 16, 32, 16, 64, 0, False), (2, 32, 32, 64, 64, 1, False), (1, 16, 64, 20, 40, 2, False), (3, 48, 96, 134, 128, 0, False),
                                       (2, 32, 64, 64, 128, 0, True), (2, 128, 128, 128, 128, 1, False), (1, 16, 32, 256, 70, 0, True)]:
    kk = cin if not dg else cout
    x = torch.randn(n, h, wd, (kk + 7) // 8 * 8, device=dev)
    w = torch.randn(3, 3, cin, cout, device=dev) * 0.1
    b = torch.randn(cout, device=dev) if not dg else None
    y, _ = run43(x, w, b, act, dg, cin=kk)
    r = ref64(x[..., :kk], w, b, act, dg)
    err = ((y.double() - r).norm() / r.norm()).item()
    print((n, h, wd, cin, cout, act, dg), 'rel-L2 %.3e' % err, 'max %.3e' % (y.double() - r).abs().max().item())
    ok &= err < 1e-5
print('OK' if ok else 'FAIL')
# timing: the 32x128x128x128->128 layer
x = torch.randn(32, 128, 128, 128, device=dev); w = torch.randn(3, 3, 128, 128, device=dev) * 0.05; b = torch.zeros(128, device=dev)
y = torch.empty(32, 128, 128, 128, device=dev)
u43 = torch.empty(lib.kpx_wino43_u_bytes(128, 128) // 4, device=dev); u23 = torch.empty(lib.kpx_wino_u_bytes(128, 128) // 4, device=dev)
check(lib.kpx_wino43_filter_transform_f32(w.data_ptr(), 128, 128, 0, u43.data_ptr(), ops._stream()), 'xf')
check(lib.kpx_wino_filter_transform_f32(w.data_ptr(), 128, 128, 0, u23.data_ptr(), ops._stream()), 'xf')
def t(fn, it=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it
for shp, ci, co in [((32, 128, 128), 128, 128), ((32, 128, 128), 64, 64), ((32, 64, 64), 128, 128), ((32, 32, 32), 256, 256), ((32, 64, 64), 256, 256)]:
    n, h, wd = shp
    x = torch.randn(n, h, wd, ci, device=dev); w = torch.randn(3, 3, ci, co, device=dev) * 0.05; y = torch.empty(n, h, wd, co, device=dev)
    u43 = torch.empty(lib.kpx_wino43_u_bytes(ci, co) // 4, device=dev); u23 = torch.empty(lib.kpx_wino_u_bytes(ci, co) // 4, device=dev)
    check(lib.kpx_wino43_filter_transform_f32(w.data_ptr(), ci, co, 0, u43.data_ptr(), ops._stream()), 'xf')
    check(lib.kpx_wino_filter_transform_f32(w.data_ptr(), ci, co, 0, u23.data_ptr(), ops._stream()), 'xf')
    s = ops._stream()
    t43 = t(lambda: lib.kpx_conv3x3_wino43_f32(x.data_ptr(), n, h, wd, ci, ci, u43.data_ptr(), None, y.data_ptr(), co, co, 0, s))
    t23 = t(lambda: lib.kpx_conv3x3_wino_f32(x.data_ptr(), n, h, wd, ci, ci, u23.data_ptr(), None, y.data_ptr(), co, co, 0, s))
    fl = 2.0 * n * h * wd * 9 * ci * co
    print(shp, ci, co, 'F(4,3) %.4f ms (%.1f eff TF, executed frac %.2f) | F(2,3) %.4f ms (%.1f eff TF)' % (t43, fl / t43 / 1e9, fl / 4 / t43 / 1e9 / 157.3, t23, fl / t23 / 1e9))
print('packed 16x16:')
for shp, ci, co in [((64, 16, 16), 512, 512), ((64, 16, 16), 256, 512), ((32, 16, 16), 512, 512), ((64, 16, 16), 128, 128)]:
    n, h, wd = shp
    x = torch.randn(n, h, wd, ci, device=dev); w = torch.randn(3, 3, ci, co, device=dev) * 0.05; y = torch.empty(n, h, wd, co, device=dev)
    u43 = torch.empty(lib.kpx_wino43_u_bytes(ci, co) // 4, device=dev); u23 = torch.empty(lib.kpx_wino_u_bytes(ci, co) // 4, device=dev)
    check(lib.kpx_wino43_filter_transform_f32(w.data_ptr(), ci, co, 0, u43.data_ptr(), ops._stream()), 'xf')
    check(lib.kpx_wino_filter_transform_f32(w.data_ptr(), ci, co, 0, u23.data_ptr(), ops._stream()), 'xf')
    s = ops._stream()
    t43 = t(lambda: check(lib.kpx_conv3x3_wino43_f32(x.data_ptr(), n, h, wd, ci, ci, u43.data_ptr(), None, y.data_ptr(), co, co, 0, s), 'c43'))
    t23 = t(lambda: check(lib.kpx_conv3x3_wino_f32(x.data_ptr(), n, h, wd, ci, ci, u23.data_ptr(), None, y.data_ptr(), co, co, 0, s), 'c23'))
    fl = 2.0 * n * h * wd * 9 * ci * co
    print(shp, ci, co, 'F(4,3) %.4f ms (%.1f eff TF, executed frac %.2f) | F(2,3) %.4f ms (%.1f eff TF)' % (t43, fl / t43 / 1e9, fl / 4 / t43 / 1e9 / 157.3, t23, fl / t23 / 1e9))
