"""Weight gradients of the tiny-filter layers (conv_wsmall.hip) at the bench batch, each alone."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import bench
import kpx_amd  # noqa: F401
from kpx_amd import ops
dev = torch.device('cuda:0')
for name, n, h, ci, co, k, s, pad in (('enc conv_2 32->32 @128 N=64', 64, 128, 32, 32, 3, 1, 1), ('enc conv_1 7x7 3->32 N=64', 64, 128, 3, 32, 7, 1, 3),
                                      ('pose conv_7_1 16->16 N=64', 64, 128, 16, 16, 3, 1, 1), ('discr conv_0 4x4s2 3->64 N=64', 64, 128, 3, 64, 4, 2, 1),
                                      ('pose 5_1 32->32 @64 N=64', 64, 64, 32, 32, 3, 1, 1)):
    ho = (h + 2 * (1 if k == 4 else 0) + s - 1) // s if k == 4 else h
    x = torch.randn(n, h, h, ci, device=dev); dy = torch.randn(n, ho, ho, co, device=dev); dw = torch.empty(k, k, ci, co, device=dev)
    pt = pad if k != 4 else 2
    ms = bench.time_kernel(lambda: ops.conv_wgrad_raw(x, ci, ci, dy, co, dw, s, pt, pt), iters=50, warm=10)
    print('%-34s %.4f ms' % (name, ms))
