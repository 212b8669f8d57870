#!/bin/bash
# Diagnostic build of the F(4x4,3x3) kernel with in-kernel s_memtime stamps (-DKPX_WINO_STAMP) + the run that prints them.
set -e
cd "$(dirname "$0")/../unsupervised-keypoint-learning-for-guiding-class-conditional-video-prediction_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DKPX_WINO_STAMP -c conv_wino43.hip -o /tmp/conv_wino43_dbg.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC conv_igemm.o conv_wino.o /tmp/conv_wino43_dbg.o conv_bf16.o conv_rgb.o pointwise.o keypoints.o loss_optim.o rollout.o -o ../../profiles/libkpx_hip_dbg43.so
cd ../../profiles && python3 wino43_stamps.py "$@"
