import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import kpx_amd
from kpx_amd import ops
from oracle import restatement as R
dev = torch.device('cuda:0')
def rel(a, b): return float(np.linalg.norm((a.astype(np.float64) - b).ravel()) / np.linalg.norm(b.ravel()))
n, h, w, c0, c1, c2, groups = 2, 64, 64, 64, 128, 128, 1
rs = np.random.RandomState(c0 + c1 + c2)
x = rs.randn(n, h, w, c0).astype(np.float32)
wa = (rs.randn(3, 3, c0, c1) / np.sqrt(9 * c0)).astype(np.float32); wb = (rs.randn(3, 3, c1, c2) / np.sqrt(9 * c1)).astype(np.float32)
ga = rs.uniform(0.5, 1.5, c1).astype(np.float32); be = (rs.randn(c1) * 0.3).astype(np.float32)
gy = rs.randn(n, h, w, c2).astype(np.float32)
def run(fused, b3, b3_bwd=None):
    ops.FUSE_BN_BWD, ops.WINO43B = fused, True
    t = {k_: torch.from_numpy(v).to(dev).requires_grad_(True) for k_, v in dict(x=x, wa=wa, wb=wb, ga=ga, be=be).items()}
    keys = ops.register_constant_filter(t['wa'].detach()) + ops.register_constant_filter(t['wb'].detach())
    try:
        ops.WINO43B = b3
        ya = ops.conv2d(t['x'], t['wa'], None, stride=1, pad=0, act=0, bn_stats=True)
        mm, mv = torch.zeros(c1, device=dev), torch.ones(c1, device=dev)
        yb = ops.batch_norm(ya, t['ga'], t['be'], mm, mv, train=True, act=1, groups=groups)
        out = ops.conv2d(yb, t['wb'], None, stride=1, pad=0, act=0)
        ops.begin_backward()
        ops.WINO43B = b3 if b3_bwd is None else b3_bwd
        ub = ops.conv_kernel_uses['wino43b']
        out.backward(torch.from_numpy(gy).to(dev))
        print('   wino43b launches in backward:', ops.conv_kernel_uses['wino43b'] - ub)
    finally:
        ops.release_filters(keys)
    return {k_: v.grad.cpu().numpy() for k_, v in t.items()}
to = {k_: torch.from_numpy(v).double().requires_grad_(True) for k_, v in dict(x=x, wa=wa, wb=wb, ga=ga, be=be).items()}
za = torch.nn.functional.conv2d(to['x'].permute(0, 3, 1, 2), to['wa'].permute(3, 2, 0, 1), padding=1)
mu = za.mean((0, 2, 3), keepdim=True); var = za.var((0, 2, 3), unbiased=False, keepdim=True)
zb = torch.relu((za - mu) / torch.sqrt(var + 1e-5) * to['ga'].view(1, -1, 1, 1) + to['be'].view(1, -1, 1, 1))
oo = torch.nn.functional.conv2d(zb, to['wb'].permute(3, 2, 0, 1), padding=1)
oo.backward(torch.from_numpy(gy).double().permute(0, 3, 1, 2))
want = {k_: v.grad.numpy() for k_, v in to.items()}
for fused, b3, b3b in ((False, True, True), (False, True, False), (False, False, True), (False, False, False)):
    g = run(fused, b3, b3b)
    print('fused %s bf16x3 fwd %s bwd %s:' % (fused, b3, b3b), ' '.join('%s %.2e' % (k_, rel(g[k_], want[k_])) for k_ in ('x', 'wa', 'wb', 'ga', 'be')))

# ---- where does the forward differ?
def fwd_only(b3):
    ops.WINO43B = b3
    t = {k_: torch.from_numpy(v).to(dev) for k_, v in dict(x=x, wa=wa, ga=ga, be=be).items()}
    keys = ops.register_constant_filter(t['wa'])
    try:
        ya = ops.conv2d(t['x'], t['wa'], None, stride=1, pad=0, act=0, bn_stats=True)
        ts = getattr(ya, '_kpx_tile_stats', None)
        mm, mv = torch.zeros(c1, device=dev), torch.ones(c1, device=dev)
        yb = ops.batch_norm(ya, t['ga'], t['be'], mm, mv, train=True, act=1, groups=groups)
        return ya.cpu().double(), yb.cpu().double(), mm.cpu().double(), mv.cpu().double(), (ts[0].cpu().double().view(-1, 2, c1), ts[1]) if ts is not None else None
    finally:
        ops.release_filters(keys)
ya1, yb1, mm1, mv1, ts1 = fwd_only(True)
ya0, yb0, mm0, mv0, ts0 = fwd_only(False)
za64 = za.detach().permute(0, 2, 3, 1)
print('conv_a out vs f64: new %.2e old %.2e' % (float((ya1 - za64).norm() / za64.norm()), float((ya0 - za64).norm() / za64.norm())))
print('tiles per image new %s old %s; slab rows %d / %d' % (ts1[1], ts0[1], ts1[0].shape[0], ts0[0].shape[0]))
s1, s0 = ts1[0].sum(0), ts0[0].sum(0)
true_s, true_q = za64.sum((0, 1, 2)), (za64 ** 2).sum((0, 1, 2))
print('channel sums from the slab vs f64: new sum %.2e sumsq %.2e | old sum %.2e sumsq %.2e' % (float((s1[0] - true_s).norm() / true_s.norm()), float((s1[1] - true_q).norm() / true_q.norm()),
      float((s0[0] - true_s).norm() / true_s.norm()), float((s0[1] - true_q).norm() / true_q.norm())))
print('slab sums vs the stored tensor own sums: new %.2e old %.2e' % (float((s1[0] - ya1.sum((0, 1, 2))).norm() / true_s.norm()), float((s0[0] - ya0.sum((0, 1, 2))).norm() / true_s.norm())))
print('moving mean diff new-old %.2e  moving var diff %.2e ; relu mask flips new vs old: %d' % (float((mm1 - mm0).abs().max()), float((mv1 - mv0).abs().max()), int(((yb1 > 0) != (yb0 > 0)).sum())))
zb64 = zb.detach().permute(0, 2, 3, 1)
print('BN+relu out vs f64: new %.2e old %.2e ; mask flips vs f64: new %d old %d' % (float((yb1 - zb64).norm() / zb64.norm()), float((yb0 - zb64).norm() / zb64.norm()), int(((yb1 > 0) != (zb64 > 0)).sum()), int(((yb0 > 0) != (zb64 > 0)).sum())))
