"""In-kernel s_memtime stamps of conv3x3_wgrad_bf16_kernel (diagnostic build: csrc/conv_bf16s_wgrad.hip with scratch/exp/conv_bf16s_wgrad_stamps.patch,
built to scratch/exp/libkpx_wstamp.so, selected with KPX_LIB): per stage and wavefront the cycles of the request block, the reads + MFMAs, the
wait for the next stage's operands and the barrier.   KPX_LIB=... python scratch/wgrad16_stamps.py N H Cin Cout"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import kpx_amd
from kpx_amd import ops
from kpx_amd._lib import lib, check
dev = torch.device('cuda:0')
n, h, cin, cout = (int(v) for v in sys.argv[1:5])
raw = ctypes.CDLL(os.environ['KPX_LIB'])
x = torch.randn(n, h, h, cin, device=dev).bfloat16(); dy = torch.randn(n, h, h, cout, device=dev).bfloat16()
dw = torch.empty(3, 3, cin, cout, device=dev)
nbytes = lib.kpx_conv3x3_wgrad_bf16_workspace_bytes(n, h, h, cin, cout)
ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
def run():
    check(lib.kpx_conv3x3_wgrad_bf16(x.data_ptr(), n, h, h, cin, cin, dy.data_ptr(), cout, cout, dw.data_ptr(), ws.data_ptr(), nbytes, ops._stream()), 'wgrad')
for _ in range(5): run()
st = torch.zeros(64 * 8 * 64 * 4, dtype=torch.int64, device=dev)
raw.kpx_conv3x3_wgrad_bf16_stamps.argtypes = [ctypes.c_void_p]
raw.kpx_conv3x3_wgrad_bf16_stamps(ctypes.c_void_p(st.data_ptr()))
run(); torch.cuda.synchronize()
raw.kpx_conv3x3_wgrad_bf16_stamps(ctypes.c_void_p(0))
s = st.cpu().numpy().reshape(64, 8, 64, 4)
nst = int((s[0, 0, :60, 0] != 0).sum())
ph = s[:, :, :nst, :]
issue = ph[..., 1] - ph[..., 0]; mma = ph[..., 2] - ph[..., 1]; wait = ph[..., 3] - ph[..., 2]
nxt = np.concatenate([ph[:, :, 1:, 0], s[:, :, 62:63, 0]], axis=2); bar = nxt - ph[..., 3]
print('shape', (n, h, cin, cout), 'stages per workgroup', nst)
print('per stage and wavefront (mean): request block %.0f   reads + MFMAs %.0f   wait for the next stage %.0f   barrier %.0f   = %.0f' % (issue.mean(), mma.mean(), wait.mean(), bar.mean(), (issue + mma + wait + bar).mean()))
for p in range(min(nst, 6)):
    print('  stage %2d: issue %6.0f  mma %6.0f  wait %6.0f  barrier %6.0f' % (p, issue[:, :, p].mean(), mma[:, :, p].mean(), wait[:, :, p].mean(), bar[:, :, p].mean()))
print('whole stage loop %.0f cycles' % (s[:, :, 62, 0] - s[:, :, 0, 0]).mean())
