"""One bf16-storage conv launch shape, repeated (for rocprofv3 --pmc):  python scratch/bf16s_one.py N H Cin Cout [iters]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kpx_amd
from kpx_amd import ops
from kpx_amd._lib import lib, check
dev = torch.device('cuda:0')
n, h, cin, cout = (int(v) for v in sys.argv[1:5])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 30
x = torch.randn(n, h, h, cin, device=dev).bfloat16()
w = torch.randn(3, 3, cin, cout, device=dev) * 0.03
b = torch.zeros(cout, device=dev)
y = torch.empty(n, h, h, cout, dtype=torch.bfloat16, device=dev)
wf = torch.empty(lib.kpx_conv3x3_bf16s_weights_bytes(cin, cout), dtype=torch.uint8, device=dev)
check(lib.kpx_conv3x3_bf16s_prepare_f32(w.data_ptr(), cin, cout, 0, wf.data_ptr(), ops._stream()), 'prep')
for _ in range(iters):
    check(lib.kpx_conv3x3_bf16s(x.data_ptr(), n, h, h, cin, cin, wf.data_ptr(), b.data_ptr(), y.data_ptr(), cout, cout, 0, 1, None, 0, None, ops._stream()), 'conv')
torch.cuda.synchronize()
