"""Which layers of one bf16 train step still go through the fp32 kernels between conversions (ops.fallback_uses), with their shapes."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kpx_amd
from kpx_amd import ops
from oracle import restatement as R
dev = torch.device('cuda:0')
seen = collections.Counter()
class Tracker(dict):
    def __setitem__(self, key, val):
        if key in self and val > self[key]:
            f = sys._getframe(1); loc = f.f_locals
            desc = [key, f.f_code.co_name]
            for nm in ('x', 'dy', 'dx', 'y', 'w', 'dw'):
                t = loc.get(nm)
                if torch.is_tensor(t): desc.append('%s%s:%s' % (nm, tuple(t.shape), str(t.dtype).replace('torch.', '')))
            for nm in ('stride', 'cin', 'c'):
                if nm in loc: desc.append('%s=%s' % (nm, loc[nm]))
            seen[' '.join(desc)] += 1
        dict.__setitem__(self, key, val)
ops.fallback_uses = Tracker(ops.fallback_uses)
os.environ['KPX_GRAPH'] = '0'
ops.set_compute_dtype('bf16')
b = 4
cfg = {'training': {'lr': {'start_val': 1e-4, 'step': 20000, 'decay': 0.95}, 'batch_size': b}, 'model': {'n_pts': 15}, 'paths': {'log_dir': '/tmp/kpx_t', 'vggnet': None}}
vgg = kpx_amd.Vgg19(weights=kpx_amd.synthetic_vgg19_weights(seed=19), device=dev)
m = kpx_amd.DetectorTranslatorModel(cfg, device=dev, vgg=vgg, image_size=128)
m.build()
im, fut = R.synthetic_pair(b, res=128, seed0=10, seed1=11)
m.train_step(None, {'image': torch.from_numpy(im).to(dev), 'future_image': torch.from_numpy(fut).to(dev)}, 0, b)
torch.cuda.synchronize()
for k, v in sorted(seen.items()): print(v, k)
