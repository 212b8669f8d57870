"""In-kernel s_memtime stamps of conv_wino43b_kernel (diagnostic build scratch/exp/libkpx_w4bstamp.so = csrc/conv_wino43b.hip with -DKPX_W4B_STAMP):
per wavefront the prologue, the K loop, the epilogue stages; per K step the DMA issue, the two transform and the two multiply phases, the counted
wait and the barrier; the shader clock.   python scratch/w43b_stamps.py [N H C]"""
import sys, os, ctypes, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kpx_amd
from kpx_amd import ops
from kpx_amd._lib import lib, check
dev = torch.device('cuda:0')
n, h, c = [int(v) for v in sys.argv[1:4]] if len(sys.argv) >= 4 else (32, 64, 128)
co = int(sys.argv[4]) if len(sys.argv) >= 5 else c
x = torch.randn(n, h, h, c, device=dev); w = torch.randn(3, 3, c, co, device=dev) * 0.05
if os.environ.get('W4B_ZERO'): x.zero_(); w.zero_()        # (power experiment: all-zero operands)
y = torch.empty(n, h, h, co, device=dev)
u = torch.empty(lib.kpx_wino43b_u_bytes(c, co), dtype=torch.uint8, device=dev)
check(lib.kpx_wino43b_filter_transform_f32(w.data_ptr(), c, co, 0, u.data_ptr(), ops._stream()), 'xf')
s = ops._stream()
clib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'exp', os.environ.get('W4B_STAMP_LIB', 'libkpx_w4bstamp.so')))
clib.kpx_conv3x3_wino43b_f32.argtypes = lib.kpx_conv3x3_wino43b_f32.argtypes
clib.kpx_debug_w4b_stamps.argtypes = [ctypes.c_void_p]
run = lambda: clib.kpx_conv3x3_wino43b_f32(x.data_ptr(), n, h, h, c, c, u.data_ptr(), None, y.data_ptr(), co, co, 0, None, 0, None, 0, None, None, 0, None, s)
buf = torch.zeros(64 * 4 * 512, dtype=torch.int64, device=dev)
clib.kpx_debug_w4b_stamps(None)
for _ in range(200): run()
torch.cuda.synchronize()
clib.kpx_debug_w4b_stamps(buf.data_ptr())
run(); torch.cuda.synchronize()
clib.kpx_debug_w4b_stamps(None)
dfull = buf.cpu().numpy().reshape(64, 4, 512).astype(np.float64)
d = dfull[:, :, :64]
ks = (c + 15) // 16
print('N%d %dx%d %d->%d: %d K steps' % (n, h, h, c, co, ks))
for wv in range(4):
    q = d[:, wv]
    f = lambda a, b: np.median(q[:, b] - q[:, a])
    clk = np.median((q[:, 7] - q[:, 0]) / np.maximum(q[:, 9] - q[:, 8], 1) * 100.0)
    print(' wave %d: prologue %6.0f | loop %7.0f (%.0f / K step) | epilogue: bar %5.0f deposit %5.0f bar %5.0f outxf %5.0f stores+ %5.0f pass2 %6.0f | total %7.0f  clk %.0f MHz'
          % (wv, f(0, 1), f(1, 2), f(1, 2) / ks, f(2, 3), f(3, 4), f(4, 5), f(5, 6), f(6, 10), f(10, 7), f(0, 7), clk))
print(' per K step (median over steps 1-%d and workgroups): dma issue | T single | M single | T pair | M pair | wait | barrier+ | whole' % min(4, ks - 1))
nst = min(4, ks - 1) if ks > 1 else 0
for wv in range(4):
    q = d[:, wv, 16:16 + 8 * nst].reshape(64, nst, 8)
    seg = [np.median(q[:, :, j + 1] - q[:, :, j]) for j in range(6)]
    whole = np.median(q[:, 1:, 0] - q[:, :-1, 0]) if nst > 1 else float('nan')
    print('  wave %d: %6.0f | %6.0f | %6.0f | %6.0f | %6.0f | %6.0f | %6.0f | %6.0f' % (wv, seg[0], seg[1], seg[2], seg[3], seg[4], seg[5], whole - sum(seg) if nst > 1 else float('nan'), whole))

