import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import kpx_amd
from kpx_amd import ops
from oracle import restatement as R
dev = torch.device('cuda:0')
def rel(a, b):
    a = a.astype(np.float64); b = b.astype(np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))
for (n, h, w, cin, cout, k, s, pad) in [(4,17,17,64,128,4,2,1), (4,10,10,128,256,4,2,1), (4,6,6,256,512,4,2,1), (4,4,4,512,1024,4,2,1), (4,3,3,1024,2048,4,2,1), (2,3,3,1024,2048,4,2,1), (4,65,65,64,128,4,2,1),(4,34,34,128,256,4,2,1),(4,18,18,256,512,4,2,1),(4,10,10,512,1024,4,2,1),(4,6,6,1024,2048,4,2,1), (4,16,16,32,64,3,2,0), (4,8,8,64,128,3,2,0), (4,4,4,128,256,3,2,0)]:
    rs = np.random.RandomState(1)
    x = rs.randn(n, h, w, cin).astype(np.float32); wt = (rs.randn(k, k, cin, cout) / np.sqrt(k*k*cin)).astype(np.float32)
    xg = torch.from_numpy(x).to(dev).requires_grad_(True); wg = torch.from_numpy(wt).to(dev)
    y = ops.conv2d(xg, wg, None, stride=s, pad=pad)
    xo = torch.from_numpy(x).requires_grad_(True)
    yo = R.conv(xo, torch.from_numpy(wt), None, s, pad)
    gy = rs.randn(*yo.shape).astype(np.float32)
    yo.backward(torch.from_numpy(gy)); y.backward(torch.from_numpy(gy).to(dev))
    yn, yon = y.detach().cpu().numpy(), yo.detach().numpy()
    print((n,h,w,cin,cout,k,s,pad), 'fwd', '%.2e' % rel(yn, yon), 'dgrad', '%.2e' % rel(xg.grad.cpu().numpy(), xo.grad.numpy()),
          'per-cout-block fwd err', ['%.1e' % rel(yn[..., c:c+16], yon[..., c:c+16]) for c in range(0, min(cout, 64), 16)])
