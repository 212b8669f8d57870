import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
import kpx_amd
from kpx_amd import ops
from oracle import restatement as R
dev = torch.device('cuda:0')
def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64); return float(np.linalg.norm(a - b) / np.linalg.norm(b))
n, h, w, c0, c1, c2, groups = 2, 64, 64, 64, 128, 128, 1
rs = np.random.RandomState(c0 + c1 + c2)
x = rs.randn(n, h, w, c0).astype(np.float32)
wa = (rs.randn(3, 3, c0, c1) / np.sqrt(9 * c0)).astype(np.float32); wb = (rs.randn(3, 3, c1, c2) / np.sqrt(9 * c1)).astype(np.float32)
ga = rs.uniform(0.5, 1.5, c1).astype(np.float32); be = (rs.randn(c1) * 0.3).astype(np.float32)
gy = rs.randn(n, h, w, c2).astype(np.float32)
to = {k_: torch.from_numpy(v).double().requires_grad_(True) for k_, v in dict(x=x, wa=wa, wb=wb, ga=ga, be=be).items()}
za = R.conv(to['x'], to['wa'], None, 1, 0)
zb = torch.relu(R.batch_norm_train(za, to['ga'], to['be'])[0])
oo = R.conv(zb, to['wb'], None, 1, 0)
oo.backward(torch.from_numpy(gy).double())
for name, fuse, w43, reg_b in (('F43 fused', True, True, True), ('F43 dgrad, separate reduce', False, True, True), ('F23 fused', True, False, True), ('F23 separate', False, False, True)):
    ops.FUSE_BN_BWD, ops.WINO43 = fuse, w43
    t = {k_: torch.from_numpy(v).to(dev).requires_grad_(True) for k_, v in dict(x=x, wa=wa, wb=wb, ga=ga, be=be).items()}
    keys = ops.register_constant_filter(t['wa'].detach()) + ops.register_constant_filter(t['wb'].detach())
    ya = ops.conv2d(t['x'], t['wa'], None, stride=1, pad=0, act=0, bn_stats=True)
    mm, mv = torch.zeros(c1, device=dev), torch.ones(c1, device=dev)
    yb = ops.batch_norm(ya, t['ga'], t['be'], mm, mv, train=True, act=1, groups=groups)
    out = ops.conv2d(yb, t['wb'], None, stride=1, pad=0, act=0)
    ops.begin_backward()
    u = ops.fused_bn_uses['backward_sums_from_dgrad_epilogue']
    out.backward(torch.from_numpy(gy).to(dev)); ops.join_side_stream()
    print('%-28s hits %d ' % (name, ops.fused_bn_uses['backward_sums_from_dgrad_epilogue'] - u) + ' '.join('%s %.2e' % (k_, rel(t[k_].grad.cpu().numpy(), to[k_].grad.numpy())) for k_ in ('x', 'wa', 'wb', 'ga', 'be')))
    ops.release_filters(keys)
