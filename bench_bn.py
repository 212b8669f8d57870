"""Micro-benchmark of the batch-norm kernels (HBM-bound): per-kernel GB/s on the step's typical tensor shapes."""
import sys, time, torch
sys.path.insert(0, '.')
import kpx_amd
from kpx_amd import ops
dev = torch.device('cuda:0')
shapes = [(32, 128, 128, 64), (32, 64, 64, 128), (32, 32, 32, 256), (64, 128, 128, 16), (64, 64, 64, 32), (64, 16, 16, 128), (64, 65, 65, 128)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in sys.argv[1:5])]
for shp in shapes:
    n, h, w, c = shp
    x = torch.randn(*shp, device=dev, requires_grad=True)
    g = torch.ones(c, device=dev, requires_grad=True); b = torch.zeros(c, device=dev, requires_grad=True)
    mm = torch.zeros(c, device=dev); mv = torch.ones(c, device=dev)
    gy = torch.randn(*shp, device=dev)
    nbytes = x.numel() * 4
    def fwd():
        return ops.batch_norm(x, g, b, mm, mv, train=True, act=1)
    for _ in range(3): y = fwd(); y.backward(gy)
    torch.cuda.synchronize()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    it = 20; tf = tb = 0.0
    for _ in range(it):
        e0.record(); y = fwd(); e1.record(); y.backward(gy); e2.record(); torch.cuda.synchronize()
        tf += e0.elapsed_time(e1); tb += e1.elapsed_time(e2)
    tf /= it; tb /= it
    print('%-22s %6.1f MB | fwd %.3f ms = %5.2f TB/s of 3 passes | bwd %.3f ms = %5.2f TB/s of 5 passes' %
          (str(shp), nbytes / 1e6, tf, 3 * nbytes / tf / 1e9, tb, 5 * nbytes / tb / 1e9))
