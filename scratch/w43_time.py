import sys, os, torch
sys.path.insert(0, '.')
import kpx_amd
from kpx_amd import ops
from kpx_amd._lib import lib, check
dev = torch.device('cuda:0')
def t(fn, it=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it
n, h, wd, ci, co = 32, 128, 128, 128, 128
x = torch.randn(n, h, wd, ci, device=dev); w = torch.randn(3, 3, ci, co, device=dev) * 0.05; y = torch.empty(n, h, wd, co, device=dev)
u43 = torch.empty(lib.kpx_wino43_u_bytes(ci, co) // 4, device=dev)
check(lib.kpx_wino43_filter_transform_f32(w.data_ptr(), ci, co, 0, u43.data_ptr(), ops._stream()), 'xf')
s = ops._stream()
for dbg in [int(v) for v in sys.argv[1:]] or [0]:
    os.environ['KPX_W43_DBG'] = str(dbg)
    t43 = t(lambda: lib.kpx_conv3x3_wino43_f32(x.data_ptr(), n, h, wd, ci, ci, u43.data_ptr(), None, y.data_ptr(), co, co, 0, s))
    print('dbg %2d: %.4f ms  (MFMA-only floor %.4f)' % (dbg, t43, 2.0 * n * h * wd * 9 * ci * co / 4 / 157.3e9))
