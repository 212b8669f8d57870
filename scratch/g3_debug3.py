import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['KPX_GRAPH'] = '0'
import numpy as np, torch
import kpx_amd
from kpx_amd import ops
from kpx_amd._lib import lib
from kpx_amd.synthetic import synthetic_pair
dev = torch.device('cuda:0')
res, k, b = int(sys.argv[1]), 3, 2
def rel(a, b): return float(np.linalg.norm(a.astype(np.float64) - b) / max(np.linalg.norm(b.astype(np.float64)), 1e-30))
out = {}
for mode in ('fp32', 'gemm3'):
    if mode == 'fp32': os.environ['KPX_NO_GEMM3'] = '1'
    else: os.environ.pop('KPX_NO_GEMM3', None)
    lib.kpx_reload_env()
    cfg = {'training': {'lr': {'start_val': 1e-4, 'step': 20000, 'decay': 0.95}, 'batch_size': b}, 'model': {'n_pts': k}, 'paths': {'log_dir': '/tmp/x', 'vggnet': None}}
    vgg = kpx_amd.Vgg19(weights=kpx_amd.synthetic_vgg19_weights(seed=19, width_div=8), device=dev)
    model = kpx_amd.DetectorTranslatorModel(cfg, device=dev, vgg=vgg, image_size=res); model.build()
    feed = {k_: torch.from_numpy(v).to(dev) for k_, v in synthetic_pair(b, res=res, seed0=10, seed1=20).items()}
    model.train_step(None, feed, 0, b)
    torch.cuda.synchronize()
    fwd = model.last['fwd']
    out[mode] = ({k_: v.cpu().numpy().copy() for k_, v in fwd.items()}, {n: model.store.grad(n).cpu().numpy().copy() for bk in ('G', 'D') for n in model.store.buckets[bk].entries if n.endswith('kernel')}, model.loss_values())
for k_ in out['fp32'][0]:
    print('%-20s %.2e' % (k_, rel(out['gemm3'][0][k_], out['fp32'][0][k_])))
print(out['fp32'][2]); print(out['gemm3'][2])
for n in out['fp32'][1]:
    r = rel(out['gemm3'][1][n], out['fp32'][1][n])
    if r > 1e-4: print('%-50s %.2e  norm %.3e' % (n, r, np.linalg.norm(out['fp32'][1][n])))
