#!/bin/bash
# Diagnostic build of the Winograd kernel with in-kernel s_memtime stamps + the run that prints them (see wino_stamps.py).
set -e
cd "$(dirname "$0")/../unsupervised-keypoint-learning-for-guiding-class-conditional-video-prediction_amd/csrc"
make -s
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DKPX_WINO_STAMP -c conv_wino.hip -o /tmp/conv_wino_dbg.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC conv_igemm.o /tmp/conv_wino_dbg.o conv_wino43.o conv_bf16.o conv_rgb.o pointwise.o keypoints.o loss_optim.o rollout.o -o ../../profiles/libkpx_hip_dbg.so
cd ../../profiles && python3 wino_stamps.py "$@"
