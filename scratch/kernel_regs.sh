#!/bin/bash
# Register / LDS / spill figures of every kernel of one csrc/*.hip file:   bash scratch/kernel_regs.sh conv_wino.hip [filter]
PKG=/root/repo/unsupervised-keypoint-learning-for-guiding-class-conditional-video-prediction_amd/csrc
T=$(mktemp -d); cd $T
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -c $PKG/$1 -o k.co -I$PKG ${EXTRA_FLAGS} || exit 1
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=k.co --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=k.elf
/opt/rocm/lib/llvm/bin/llvm-readelf --notes k.elf | python3 -c "
import sys,re
txt=sys.stdin.read()
for blk in txt.split('- .agpr_count:')[1:]:
    blk='.agpr_count:'+blk
    g=lambda k: (re.search(r'\.'+k+r':\s*(\S+)', blk) or [None,'?'])[1]
    name=g('name')
    if len(sys.argv)>1 and sys.argv[1] not in name: continue
    print('%-90s vgpr %s agpr %s sgpr %s lds %s scratch %s vspill %s' % (name[:90], g('vgpr_count'), g('agpr_count'), g('sgpr_count'), g('group_segment_fixed_size'), g('private_segment_fixed_size'), g('vgpr_spill_count')))
" $2 | cut -c1-200
rm -rf $T
