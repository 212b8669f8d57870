import os, sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from test_model_gpu import make_model, R
from kpx_amd import ops
dev = torch.device('cuda:0')
model = make_model(128, 15, 4, dev, width_div=1)
im, fut = R.synthetic_pair(4, res=128)
model.train_step(None, {'image': torch.from_numpy(im).to(dev), 'future_image': torch.from_numpy(fut).to(dev)}, 0, 4)
print('dgrad-excl=%r uses=%s  n43 descriptors=%s' % (os.environ.get('KPX_WINO43_EXCLUDE_DGRAD'), ops.conv_kernel_uses, model.store.filter_bank.n_desc43))
