"""Heat-map render at the configs[1] and configs[3] shapes for a KPX_GAUSS_BLOCKS / KPX_GAUSS_NT setting (set in the environment)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import bench
import kpx_amd  # noqa: F401
dev = torch.device('cuda:0')
for res, k, b in ((128, 15, 32), (256, 40, 16)):
    d = bench.roofline_render(dev, res, k, b)
    print('blocks=%s nt=%s  [%d,%d,%d,%d]  %.1f GB/s  frac %.4f  %.5f ms' % (os.environ.get('KPX_GAUSS_BLOCKS'), os.environ.get('KPX_GAUSS_NT'), 2 * b, res, res, k, d['achieved'], d['frac'], d['avg_launch_ms']))
