#!/bin/bash
# diagnostic build of csrc/conv_wino43b.hip with in-kernel stamps (and optional -DW4B_EXP=<bits>): scratch/exp/libkpx_w4bstamp.so
# usage: bash scratch/w4b_stampbuild.sh [extra hipcc flags]
cd "$(dirname "$0")/../unsupervised-keypoint-learning-for-guiding-class-conditional-video-prediction_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -DKPX_W4B_STAMP "$@" conv_wino43b.hip kpx_env.hip -o ../../scratch/exp/libkpx_w4bstamp.so
