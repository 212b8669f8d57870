#!/bin/bash
# instruction mix of the four K-loop bodies of conv_wino43b_kernel<0> (cross-compiled ISA; no GPU)
cd "$(dirname "$0")/../unsupervised-keypoint-learning-for-guiding-class-conditional-video-prediction_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only $EXTRA -S conv_wino43b.hip -o /tmp/w4b.s 2>/dev/null
awk '/^_Z19conv_wino43b_kernelILi0ELb0EEv11Wino43bGeom:/,/s_endpgm/' /tmp/w4b.s > /tmp/k0.s
python3 - <<'PY'
import re
lines=open('/tmp/k0.s').read().split('\n')
hdr=[i for i,l in enumerate(lines) if 'Inner Loop Header' in l]
for h in hdr:
    # loop end: first s_cbranch back to this label
    lab=lines[h].split(':')[0]
    tag='Header='+lab.lstrip('.L')          # the loop = the header block + the following blocks marked "in Loop: Header=<it>"
    end=None
    for i in range(h+1,len(lines)):
        if re.match(r'^\.LBB\d+_\d+:',lines[i]) and tag not in lines[i]: end=i-1; break
    if end is None: continue
    seg=lines[h:end+1]
    cnt=lambda p: sum(1 for l in seg if re.search(p,l))
    n=sum(1 for l in seg if l.startswith('\t') and not l.strip().startswith(';') and not l.strip().startswith('.'))
    if cnt('v_mfma')==0: continue
    print('loop %s: %d instr | mfma %d | valu(v_ non-mfma) %d | pk %d | accw %d accr %d | scratch ld %d st %d | ds_read %d ds_write %d | buffer_load %d | waitcnt %d (vmcnt0 %d) | s_nop %d' % (
        lab,n,cnt('v_mfma'),cnt(r'^\tv_(?!mfma)'),cnt('v_pk_'),cnt('v_accvgpr_write'),cnt('v_accvgpr_read'),cnt('scratch_load'),cnt('scratch_store'),cnt('ds_read'),cnt('ds_write'),cnt('buffer_load'),cnt('s_waitcnt'),cnt(r'vmcnt\(0\)'),cnt('s_nop')))
PY
