import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['KPX_GRAPH'] = '0'
import numpy as np, torch
import kpx_amd
from kpx_amd import ops, networks, variables
from kpx_amd._lib import lib
dev = torch.device('cuda:0')
res, k, b = 32, 3, 2
cfg = {'training': {'lr': {'start_val': 1e-4, 'step': 20000, 'decay': 0.95}, 'batch_size': b}, 'model': {'n_pts': k}, 'paths': {'log_dir': '/tmp/x', 'vggnet': None}}
vgg = kpx_amd.Vgg19(weights=kpx_amd.synthetic_vgg19_weights(seed=19, width_div=8), device=dev)
model = kpx_amd.DetectorTranslatorModel(cfg, device=dev, vgg=vgg, image_size=res); model.build()
rs = np.random.RandomState(0)
x = torch.from_numpy(rs.uniform(-1, 1, (4, res, res, 3)).astype(np.float32)).to(dev)
out = {}
for mode in ('fp32', 'gemm3'):
    if mode == 'fp32': os.environ['KPX_NO_GEMM3'] = '1'
    else: os.environ.pop('KPX_NO_GEMM3', None)
    lib.kpx_reload_env()
    acts = []
    with variables.as_default(model.store):
        logits = networks.img_discr(x)
        loss = ops.sigmoid_xent(logits, logits.numel() // 2, 1.0, logits.numel() // 2, 0.0)
        ops.begin_backward()
        torch.autograd.backward([loss], [model._e0])
        ops.join_side_stream(dev)
    torch.cuda.synchronize()
    out[mode] = (logits.detach().cpu().numpy().copy(), {n: model.store.grad(n).cpu().numpy().copy() for n in model.store.buckets['D'].entries})
def rel(a, b): return float(np.linalg.norm(a.astype(np.float64) - b) / max(np.linalg.norm(b.astype(np.float64)), 1e-30))
print('logits', rel(out['gemm3'][0], out['fp32'][0]))
for n in out['fp32'][1]:
    print('%-40s %.2e  norm %.3e' % (n, rel(out['gemm3'][1][n], out['fp32'][1][n]), np.linalg.norm(out['fp32'][1][n])))
