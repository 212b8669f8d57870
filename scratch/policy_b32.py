"""Forward parity at the bench batch (B=32, 128x128, K=15) under the current F(4x4,3x3) policy environment: frame / crude / mask rel-L2 and
key-point distance from the fp32 oracle, losses.  Run once per policy (KPX_WINO43_FWD_ALL=0 / 1)."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from test_model_gpu import make_model, R, rel_l2
dev = torch.device('cuda:0')
res, k, b = 128, 15, 32
torch.set_num_threads(min(32, len(os.sched_getaffinity(0))))
cache = '/tmp/policy_b32_oracle.pt'
im, fut = R.synthetic_pair(b, res=res)
if os.path.exists(cache):
    want = torch.load(cache, weights_only=False)
else:
    st = R.TrainState(R.init_variables(k, res=res, seed=1234), R.synthetic_vgg(seed=19, width_div=4))
    want = R.train_step(st, im, fut)
    torch.save({k_: v for k_, v in want.items() if not k_.startswith('grads')}, cache)
model = make_model(res, k, b, dev, width_div=4)
model.train_step(None, {'image': torch.from_numpy(im).to(dev), 'future_image': torch.from_numpy(fut).to(dev)}, 0, b)
fwd = model.last['fwd']; got = model.loss_values()
print('FWD_ALL=%s: frame %.2e crude %.2e mask %.2e points %.2e | loss_G_recon %.3e loss_D %.3e' % (
    os.environ.get('KPX_WINO43_FWD_ALL', '0'), rel_l2(fwd['final_output'].cpu().numpy(), want['final_output'].numpy()),
    rel_l2(fwd['crude_output'].cpu().numpy(), want['crude_output'].numpy()), rel_l2(fwd['mask'].cpu().numpy(), want['mask'].numpy()),
    np.abs(fwd['current_points'].cpu().numpy() - want['current_points'].numpy()).max(),
    abs(got['loss_G_recon'] - want['loss_G_recon']) / abs(want['loss_G_recon']), abs(got['loss_D'] - want['loss_D']) / abs(want['loss_D'])))
