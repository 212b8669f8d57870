"""One roofline leg of bench.py alone:  python scratch/leg_only.py roofline_conv | roofline_wgrad | roofline_conv_f23 | ...   (for A/B and --pmc runs)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import bench
import kpx_amd  # noqa: F401
for name in sys.argv[1:]:
    d = getattr(bench, name)(torch.device('cuda:0'))
    print(name, d['achieved'], d['frac'], d['avg_launch_ms'])
