"""Gradient error against the float64 arbiter (as tests/test_model_gpu.py::test_configs0...) under the current KPX_WINO43* policy.
At B=4 most policy layers launch <= 128 workgroups and would fall back to F(2x2,3x3): the threshold is set to 0 here (KPX_WINO43_MIN_WGS
overrides) so that EVERY policy layer runs the F(4x4,3x3) kernel, as in the B=32 benchmark; the launch count is printed."""
import os, sys, torch, numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from test_model_gpu import make_model, grad_error_vs_f64, R, rel_l2
from kpx_amd import ops
if 'KPX_WINO43_MIN_WGS' not in os.environ:
    ops.WINO43_MIN_WORKGROUPS = 0
dev = torch.device('cuda:0')
res, k, b = 128, 15, 4
model = make_model(res, k, b, dev, width_div=1)
torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
cache = '/tmp/w43_policy_cache.pt'
im, fut = R.synthetic_pair(b, res=res)
if os.path.exists(cache):
    want, want64 = torch.load(cache, weights_only=False)
else:
    st = R.TrainState(R.init_variables(k, res=res, seed=1234), R.synthetic_vgg(seed=19))
    want = R.train_step(st, im, fut)
    st64 = R.TrainState(R.init_variables(k, res=res, seed=1234), R.synthetic_vgg(seed=19), dtype=torch.float64)
    want64 = R.train_step(st64, im, fut)
    torch.save((want, want64), cache)
used = ops.conv_kernel_uses['wino43']
model.train_step(None, {'image': torch.from_numpy(im).to(dev), 'future_image': torch.from_numpy(fut).to(dev)}, 0, b)
print('F(4x4,3x3) launches in this step: %d (min workgroups %d)' % (ops.conv_kernel_uses['wino43'] - used, ops.WINO43_MIN_WORKGROUPS))
fwd = model.last['fwd']
print('policy fwd-excl=%s dgrad-excl=%s on=%s' % (os.environ.get('KPX_WINO43_EXCLUDE_FWD'), os.environ.get('KPX_WINO43_EXCLUDE_DGRAD'), os.environ.get('KPX_WINO43', '1')))
print('  final_output rel-L2 vs fp32 oracle %.2e  vs f64 %.2e (oracle32 vs f64 %.2e)' % (rel_l2(fwd['final_output'].cpu().numpy(), want['final_output'].numpy()),
      rel_l2(fwd['final_output'].cpu().numpy(), want64['final_output'].numpy()), rel_l2(want['final_output'].numpy(), want64['final_output'].numpy())))
print('  points max abs diff %.2e' % np.abs(fwd['current_points'].cpu().numpy() - want['current_points'].numpy()).max())
got = model.loss_values()
print('  losses', {k_: '%.3e' % abs(got[k_] - want[k_]) for k_ in ('loss_D', 'loss_G_recon', 'loss_G_adv')})
for which, g32, g64 in (('G', want['grads_G'], want64['grads_G']), ('D', want['grads_D'], want64['grads_D'])):
    names = [n for n in g32 if n.endswith('/kernel') and 'conv_6' not in n]
    e_h, e_o = grad_error_vs_f64(model, g32, g64, names)
    line = '  %s: hip %.2e oracle32 %.2e ratio %.2f |' % (which, e_h, e_o, e_h / e_o)
    for scope in ('image_encoder', 'pose_encoder', 'translator', 'img_discr'):
        sub = [n for n in names if n.startswith(scope)]
        if sub:
            a, c = grad_error_vs_f64(model, g32, g64, sub)
            line += ' %s %.2e/%.2e=%.1f' % (scope[:8], a, c, a / c)
    print(line)
