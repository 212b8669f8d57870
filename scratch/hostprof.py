import sys, cProfile, pstats, io
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import kpx_amd
from kpx_amd.synthetic import synthetic_pair
dev = torch.device('cuda:0')
cfg = {'training': {'lr': {'start_val': 1e-4, 'step': 20000, 'decay': 0.95}, 'batch_size': 32}, 'model': {'n_pts': 15}, 'paths': {'log_dir': '/tmp/kpx_bench', 'vggnet': None}}
vgg = kpx_amd.Vgg19(weights=kpx_amd.synthetic_vgg19_weights(seed=19), device=dev)
model = kpx_amd.DetectorTranslatorModel(cfg, device=dev, vgg=vgg, image_size=128)
model.build()
pair = synthetic_pair(32, res=128, seed0=0, seed1=1)
feed = {k: torch.from_numpy(v).to(dev) for k, v in pair.items()}
for i in range(3):
    model.train_step(None, feed, i, 32)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(5):
    model.train_step(None, feed, 3 + i, 32)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(35)
print(s.getvalue()[:6000])
