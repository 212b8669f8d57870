"""The Winograd weight gradient of translator conv_3_0 alone (bench.py's roofline_wgrad leg), for rocprofv3 --pmc runs."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import bench
import kpx_amd  # noqa: F401
print(bench.roofline_wgrad(torch.device('cuda:0')))
