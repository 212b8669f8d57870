"""Which part of the train step breaks HIP graph capture?  python scratch/graph_probe.py MODE  (each mode in its own process)."""
import os, sys
mode = sys.argv[1]
env = {'fwd1': dict(KPX_AUX_STREAM='0', KPX_AUX_STREAM_FWD='0', KPX_AUX_STREAM_ADV='0', KPX_SIDE_WGRAD='0'),
       'fwd_aux': dict(KPX_SIDE_WGRAD='0'),
       'step1': dict(KPX_AUX_STREAM='0', KPX_AUX_STREAM_FWD='0', KPX_AUX_STREAM_ADV='0', KPX_SIDE_WGRAD='0'),
       'step_side': dict(KPX_AUX_STREAM='0', KPX_AUX_STREAM_FWD='0', KPX_AUX_STREAM_ADV='0'),
       'step_aux': dict(KPX_SIDE_WGRAD='0'),
       'step_full': {},
       'p_fwd_side': dict(KPX_AUX_STREAM='0', KPX_AUX_STREAM_ADV='0'), 'p_d_side': dict(KPX_AUX_STREAM_FWD='0', KPX_AUX_STREAM_ADV='0'),
       'p_dadv_side': dict(KPX_AUX_STREAM_FWD='0'), 'p_full_noadv': dict(KPX_AUX_STREAM_ADV='0'),
       'p_fwd_noside': dict(KPX_AUX_STREAM='0', KPX_AUX_STREAM_ADV='0', KPX_SIDE_WGRAD='0'), 'p_d_noside': dict(KPX_AUX_STREAM_FWD='0', KPX_AUX_STREAM_ADV='0', KPX_SIDE_WGRAD='0'), 'torch_only': {}, 'torch_bwd': {}, 'one_conv': {}, 'dloss': dict(KPX_AUX_STREAM='0', KPX_AUX_STREAM_FWD='0', KPX_AUX_STREAM_ADV='0', KPX_SIDE_WGRAD='0')}[mode]
os.environ.update(env)
os.environ['KPX_GRAPH'] = '0'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import faulthandler; faulthandler.enable()
import numpy as np, torch
import kpx_amd
from kpx_amd import ops
from kpx_amd.synthetic import synthetic_pair
dev = torch.device('cuda:0')
if mode == 'torch_only':
    x = torch.randn(1024, 1024, device=dev); g = torch.cuda.CUDAGraph()
    y = x @ x
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        y = (x @ x).relu()
    g.replay(); torch.cuda.synchronize(); print('OK torch_only', float(y.sum())); sys.exit(0)
if mode == 'torch_bwd':
    w = torch.randn(256, 256, device=dev, requires_grad=True); x = torch.randn(64, 256, device=dev)
    (x @ w).relu().sum().backward(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); w.grad = None
    with torch.cuda.graph(g):
        (x @ w).relu().sum().backward()
    g.replay(); torch.cuda.synchronize(); print('OK torch_bwd', float(w.grad.sum())); sys.exit(0)
if mode == 'one_conv':
    x = torch.randn(4, 64, 64, 64, device=dev); w = torch.randn(3, 3, 64, 64, device=dev) * .05
    y = ops.conv2d(x, w); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        y = ops.conv2d(x, w)
    g.replay(); torch.cuda.synchronize(); print('OK one_conv', float(y.sum())); sys.exit(0)
res, k, b = 64, 5, 4
cfg = {'training': {'lr': {'start_val': 1e-4, 'step': 20000, 'decay': 0.95}, 'batch_size': b}, 'model': {'n_pts': k}, 'paths': {'log_dir': '/tmp/kpx_probe', 'vggnet': None}}
vgg = kpx_amd.Vgg19(weights=kpx_amd.synthetic_vgg19_weights(seed=19, width_div=8), device=dev)
model = kpx_amd.DetectorTranslatorModel(cfg, device=dev, vgg=vgg, image_size=res)
model.build()
feed = {k_: torch.from_numpy(v).to(dev) for k_, v in synthetic_pair(b, res=res).items()}
model.train_step(None, feed, 0, b)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
if mode.startswith('fwd'):
    model.forward(feed['image'], feed['future_image'], with_vis_maps=False); torch.cuda.synchronize()
    with torch.cuda.graph(g):
        out = model.forward(feed['image'], feed['future_image'], with_vis_maps=False)
    g.replay(); torch.cuda.synchronize(); print('OK', mode, float(out['final_output'].sum()))
elif mode == 'dloss':
    from kpx_amd import variables
    with variables.as_default(model.store):
        fwd = model._define_forward_pass(feed['image'], feed['future_image'])
        final_d = fwd['final_output'].detach()
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            d_losses = model._loss_D(final_d, feed['future_image'])
            ops.begin_backward()
            torch.autograd.backward([d_losses], [model._e0])
            ops.join_side_stream(dev)
    g.replay(); torch.cuda.synchronize(); print('OK dloss', d_losses.tolist())
else:
    model._alpha_dev = {w: torch.zeros(1, device=dev) for w in ('D', 'G')}
    model._capturing = True
    with torch.cuda.graph(g):
        model._train_step_eager(feed)
    model._capturing = False
    for w in ('D', 'G'):
        ops.fill_raw_(model._alpha_dev[w], 1e-4)
    g.replay(); torch.cuda.synchronize(); print('OK', mode, model.loss_values())
