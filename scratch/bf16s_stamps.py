"""In-kernel s_memtime stamps of conv3x3_bf16s_kernel (diagnostic build: csrc/conv_bf16s.hip with scratch/exp/conv_bf16s_stamps.patch applied, built to scratch/exp/libkpx_stamp.so, selected with KPX_LIB): per phase and wavefront the cycles
spent in the request block, the fragment reads + MFMAs, the counted wait and the barrier.   KPX_LIB=... python scratch/bf16s_stamps.py N H Cin Cout"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import kpx_amd
from kpx_amd import ops
from kpx_amd._lib import lib, check
dev = torch.device('cuda:0')
n, h, cin, cout = (int(v) for v in sys.argv[1:5])
raw = ctypes.CDLL(os.environ['KPX_LIB'])
x = torch.randn(n, h, h, cin, device=dev).bfloat16()
w = torch.randn(3, 3, cin, cout, device=dev) * 0.03
b = torch.zeros(cout, device=dev)
y = torch.empty(n, h, h, cout, dtype=torch.bfloat16, device=dev)
wf = torch.empty(lib.kpx_conv3x3_bf16s_weights_bytes(cin, cout), dtype=torch.uint8, device=dev)
check(lib.kpx_conv3x3_bf16s_prepare_f32(w.data_ptr(), cin, cout, 0, wf.data_ptr(), ops._stream()), 'prep')
def run():
    check(lib.kpx_conv3x3_bf16s(x.data_ptr(), n, h, h, cin, cin, wf.data_ptr(), b.data_ptr(), y.data_ptr(), cout, cout, 0, 1, None, 0, None, ops._stream()), 'conv')
for _ in range(5): run()
st = torch.zeros(64 * 8 * 64 * 5, dtype=torch.int64, device=dev)
raw.kpx_conv3x3_bf16s_stamps.argtypes = [ctypes.c_void_p]
raw.kpx_conv3x3_bf16s_stamps(ctypes.c_void_p(st.data_ptr()))
run(); torch.cuda.synchronize()
raw.kpx_conv3x3_bf16s_stamps(ctypes.c_void_p(0))
s = st.cpu().numpy().reshape(64, 8, 64, 5)
nph = 3 * ((cin + 31) // 32)
t0 = s[:, :, 63, 0]                                  # before the K loop
ph = s[:, :, :nph, :]
issue = ph[..., 1] - ph[..., 0]; mma = ph[..., 2] - ph[..., 1]; wait = ph[..., 3] - ph[..., 2]; bar = ph[..., 4] - ph[..., 3]
print('shape', (n, h, cin, cout), 'phases', nph, '(s_memtime ticks; 100 MHz constant clock or shader clock, see ratio to the known MFMA time)')
print('per phase and wavefront, mean over 64 workgroups x 8 wavefronts:  request block %.0f   reads + MFMAs %.0f   counted wait %.0f   barrier %.0f' % (issue.mean(), mma.mean(), wait.mean(), bar.mean()))
print('by phase index (mean):')
for p in range(nph):
    print('  phase %2d: issue %6.0f  mma %6.0f  wait %6.0f  barrier %6.0f' % (p, issue[:, :, p].mean(), mma[:, :, p].mean(), wait[:, :, p].mean(), bar[:, :, p].mean()))
kloop = s[:, :, 62, 0] - t0; epi = s[:, :, 61, 0] - s[:, :, 62, 0]
print('K loop %.0f   epilogue %.0f   (first tile)' % (kloop.mean(), epi.mean()))
print('by wavefront, K loop total: ', [int(v) for v in kloop.mean(0)])
