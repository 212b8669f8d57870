"""One shape of conv_wino43b_kernel repeated (for rocprofv3 --pmc / --kernel-trace runs):  python scratch/w43b_one.py N H C [Cout] [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kpx_amd
from kpx_amd import ops
from kpx_amd._lib import lib, check
dev = torch.device('cuda:0')
n, h, c = [int(v) for v in sys.argv[1:4]]
co = int(sys.argv[4]) if len(sys.argv) > 4 else c
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 30
x = torch.randn(n, h, h, c, device=dev); w = torch.randn(3, 3, c, co, device=dev) * 0.05; y = torch.empty(n, h, h, co, device=dev)
u = torch.empty(lib.kpx_wino43b_u_bytes(c, co), dtype=torch.uint8, device=dev)
check(lib.kpx_wino43b_filter_transform_f32(w.data_ptr(), c, co, 0, u.data_ptr(), ops._stream()), 'xf')
for _ in range(reps):
    check(lib.kpx_conv3x3_wino43b_f32(x.data_ptr(), n, h, h, c, c, u.data_ptr(), None, y.data_ptr(), co, co, 0, None, 0, None, 0, None, None, 0, None, ops._stream()), 'conv')
torch.cuda.synchronize()
print('done')
