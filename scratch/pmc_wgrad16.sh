#!/bin/bash
# PMC of the bf16 weight-gradient roofline leg (translator conv_3_0): bash scratch/pmc_wgrad16.sh
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/pmc_wg16; rm -rf $O; mkdir -p $O
i=0
for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA" "SQ_WAIT_ANY SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE SQ_WAVES SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_LDS_ADDR_CONFLICT"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d $O/p$i -o p --output-format csv -- python3 bench.py --roofline-only --dtype bf16 > $O/out$i.txt 2>&1
done
python3 - "$O" <<'PY'
import csv, glob, collections, sys
O = sys.argv[1]
pmc = collections.defaultdict(list)
for f in glob.glob(O + '/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'conv3x3_wgrad_bf16_kernel' in r['Kernel_Name']:
            pmc[r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(pmc):
    print('%-34s %16.0f  (%d dispatches)' % (k, sum(pmc[k]) / len(pmc[k]), len(pmc[k])))
PY
