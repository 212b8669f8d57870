"""In-kernel s_memtime stamps of conv_wino43_kernel (diagnostic build: profiles/wino43_stamps.sh): per wavefront, cycles of the prologue,
the chunk loop, the two epilogue passes and the stores; per chunk the barrier wait and the two phases of each role; the shader clock."""
import sys, os, torch, ctypes, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import kpx_amd
from kpx_amd import ops
from kpx_amd._lib import lib, check
dev = torch.device('cuda:0')
n, h, wd, ci, co = int(os.environ.get("W43_N", 32)), 128, 128, 128, 128
x = torch.randn(n, h, wd, ci, device=dev); w = torch.randn(3, 3, ci, co, device=dev) * 0.05; y = torch.empty(n, h, wd, co, device=dev)
u43 = torch.empty(lib.kpx_wino43_u_bytes(ci, co) // 4, device=dev)
check(lib.kpx_wino43_filter_transform_f32(w.data_ptr(), ci, co, 0, u43.data_ptr(), ops._stream()), 'xf')
s = ops._stream()

import kpx_amd._lib as L
clib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libkpx_hip_dbg43.so'))
clib.kpx_conv3x3_wino43_f32.argtypes = lib.kpx_conv3x3_wino43_f32.argtypes
lib = clib
clib.kpx_debug_w43_stamps.argtypes = [ctypes.c_void_p]
for dbg in [int(v) for v in sys.argv[1:]] or [0]:
    os.environ['KPX_W43_DBG'] = str(dbg)
    run = lambda: lib.kpx_conv3x3_wino43_f32(x.data_ptr(), n, h, wd, ci, ci, u43.data_ptr(), None, y.data_ptr(), co, co, 0, s)
    buf = torch.zeros(64 * 8 * 64, dtype=torch.int64, device=dev)
    clib.kpx_debug_w43_stamps(None)
    for _ in range(100): run()
    torch.cuda.synchronize()
    clib.kpx_debug_w43_stamps(buf.data_ptr())
    run(); torch.cuda.synchronize()
    clib.kpx_debug_w43_stamps(None)
    d = buf.cpu().numpy().reshape(64, 8, 64)
    print('dbg', dbg)
    for wv in range(8):
        q = d[:, wv]
        tot = q[:, 7] - q[:, 0]
        clk = tot / np.maximum(q[:, 9] - q[:, 8], 1) * 100.0
        f = lambda a, b: np.median(q[:, b] - q[:, a])
        if True:
            print(' wave %d: prologue %6.0f loop %7.0f (%.0f/chunk) bar %5.0f deposit %5.0f bar %5.0f outxf %6.0f store %6.0f | total %7.0f  clk %.0f MHz' %
                  (wv, f(0, 1), f(1, 2), f(1, 2) / 16, f(2, 3), f(3, 4), f(4, 5), f(5, 6), f(6, 7), np.median(tot), np.median(clk)))
        else:
            print(' wave %d: prologue %6.0f loop %7.0f (%.0f/chunk) bar %5.0f rest %6.0f | total %7.0f  clk %.0f MHz' % (wv, f(0, 1), f(1, 2), f(1, 2) / 16, f(2, 3), f(3, 6), np.median(tot), np.median(clk)))

    print(' per chunk (median over chunks 4-11, workgroups): wait-at-barrier | phase 1 | phase 2 | whole chunk')
    for wv in range(8):
        q = d[:, wv, 16:48].reshape(64, 8, 4).astype(np.float64)
        bar = np.median(q[:, :, 1] - q[:, :, 0]); p1 = np.median(q[:, :, 2] - q[:, :, 1]); p2 = np.median(q[:, :, 3] - q[:, :, 2])
        whole = np.median(q[:, 1:, 0] - q[:, :-1, 0])
        print('  wave %d (%s): %6.0f | %6.0f | %6.0f | %6.0f' % (wv, 'T then M' if wv < 4 else 'M then T' if wv < 6 else 'L then M', bar, p1, p2, whole))
    print(' prologue: setup | (loaders: first patch landed + staged) | first barrier passed | transform(0) / second stage')
    for wv in (0, 4, 6):
        q = d[:, wv].astype(np.float64)
        print('  wave %d: %6.0f | %6.0f | %6.0f | %6.0f' % (wv, np.median(q[:, 10] - q[:, 0]), np.median(q[:, 11] - q[:, 0]) if wv >= 6 else 0, np.median(q[:, 12] - q[:, 0]), np.median(q[:, 1] - q[:, 0])))
