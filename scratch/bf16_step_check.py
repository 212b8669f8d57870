"""bf16 configuration end to end: a tiny and a mid-size train step in both configurations (losses side by side, fallback counters)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import kpx_amd
from kpx_amd import ops
from oracle import restatement as R
dev = torch.device('cuda:0')

def run(dtype, res, k, b, steps, div):
    ops.set_compute_dtype(dtype)
    for key in ops.fallback_uses: ops.fallback_uses[key] = 0
    cfg = {'training': {'lr': {'start_val': 1e-4, 'step': 20000, 'decay': 0.95}, 'batch_size': b}, 'model': {'n_pts': k}, 'paths': {'log_dir': '/tmp/kpx_t', 'vggnet': None}}
    vgg = kpx_amd.Vgg19(weights=kpx_amd.synthetic_vgg19_weights(seed=19, width_div=div), device=dev)
    m = kpx_amd.DetectorTranslatorModel(cfg, device=dev, vgg=vgg, image_size=res)
    m.build()
    out = []
    for s in range(steps):
        im, fut = R.synthetic_pair(b, res=res, seed0=10 + 2 * s, seed1=11 + 2 * s)
        m.train_step(None, {'image': torch.from_numpy(im).to(dev), 'future_image': torch.from_numpy(fut).to(dev)}, s, b)
        lv = m.loss_values()
        out.append([lv['loss_D'], lv['loss_G_recon'], lv['loss_G_adv']])
    torch.cuda.synchronize()
    fb = dict(ops.fallback_uses)
    ops.set_compute_dtype('f32')
    return np.asarray(out), fb, m.launch_mode(), ops.conv_kernel_uses_bf16s[0]

for (res, k, b, div) in [(32, 3, 2, 8), (128, 15, 4, 1)]:
    a, _, _, _ = run('f32', res, k, b, 3, div)
    c, fb, mode, uses = run('bf16', res, k, b, 3, div)
    print('res %d K %d B %d' % (res, k, b)); print(' f32 ', a.tolist()); print(' bf16', c.tolist()); print(' fallbacks', fb, 'launch mode', mode, 'bf16s launches', uses)
