#!/usr/bin/env python3
"""In-kernel s_memtime stamps of the Winograd forward kernel (diagnostic build of csrc/conv_wino.hip with -DKPX_WINO_STAMP,
linked as profiles/libkpx_hip_dbg.so by profiles/wino_stamps.sh).  Prints, for wavefront 0 (transform first) and wavefront 4
(MFMA first) of the first 64 workgroups, the median cycles per chunk spent waiting at the barrier, in the first half and in the second
half of the chunk body, plus prologue / epilogue cycles and the shader clock (s_memtime ticks per s_memrealtime tick x 100 MHz)."""
import ctypes
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(HERE, 'libkpx_hip_dbg.so'))
P = ctypes.c_void_p
lib.kpx_conv2d_fwd_workspace_bytes.restype = ctypes.c_size_t
lib.kpx_conv2d_fwd_workspace_bytes.argtypes = [ctypes.c_int] * 7
lib.kpx_conv2d_fwd_f32.restype = ctypes.c_int
lib.kpx_conv2d_fwd_f32.argtypes = [P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, P, ctypes.c_int, ctypes.c_int, P,
                                   P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, P, ctypes.c_size_t, P]
lib.kpx_debug_wino_stamps.argtypes = [P]


def main():
    n, h, c = 32, 64, 128
    if len(sys.argv) > 1:
        n, h, c = (int(v) for v in sys.argv[1:4])
    dev = torch.device('cuda:0')
    x = torch.randn(n, h, h, c, device=dev)
    w = torch.randn(3, 3, c, c, device=dev) * 0.03
    b = torch.zeros(c, device=dev)
    y = torch.empty(n, h, h, c, device=dev)
    nbytes = lib.kpx_conv2d_fwd_workspace_bytes(n, h, h, c, c, 3, 3)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    dbg = torch.zeros(64 * 2 * 256, dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def run():
        rc = lib.kpx_conv2d_fwd_f32(x.data_ptr(), n, h, h, c, c, w.data_ptr(), 3, 3, b.data_ptr(), y.data_ptr(), h, h, c, c, 1, 1, 1, 0, 0, ws.data_ptr(), nbytes, st)
        assert rc == 0, rc
    for _ in range(200):                      # warm the clocks (DVFS) with back-to-back launches
        run()
    torch.cuda.synchronize()
    assert lib.kpx_debug_wino_stamps(dbg.data_ptr()) == 0
    run()
    torch.cuda.synchronize()
    d = dbg.cpu().numpy().reshape(64, 2, 256)
    nch = (c + 7) // 8
    for wv, name in ((0, 'wave 0 (transform, then MFMA)'), (1, 'wave 4 (MFMA, then transform)')):
        s = d[:, wv]
        ok = s[:, 0] > 0
        s = s[ok]
        total = s[:, 3] - s[:, 0]
        clk = (s[:, 3] - s[:, 0]) / np.maximum(s[:, 4] - s[:, 1], 1) * 100.0
        pro = s[:, 8] - s[:, 0]
        epi = s[:, 3] - s[:, 2]
        st_ = s[:, 8:8 + 4 * nch].reshape(-1, nch, 4)
        bar = st_[:, :, 1] - st_[:, :, 0]
        h1 = st_[:, :, 2] - st_[:, :, 1]
        h2 = st_[:, :, 3] - st_[:, :, 2]
        chunk = np.diff(st_[:, :, 0], axis=1)
        print('%s: %d workgroups; kernel %.0f cycles, clock %.0f MHz; prologue %.0f, epilogue %.0f; per chunk: total %.0f, barrier wait %.0f, first half %.0f, second half %.0f'
              % (name, s.shape[0], np.median(total), np.median(clk), np.median(pro), np.median(epi), np.median(chunk), np.median(bar), np.median(h1), np.median(h2)))
        print('   per-chunk medians (barrier / half1 / half2):', ' '.join('%d/%d/%d' % (np.median(bar[:, k]), np.median(h1[:, k]), np.median(h2[:, k])) for k in range(nch)))


if __name__ == '__main__':
    main()
