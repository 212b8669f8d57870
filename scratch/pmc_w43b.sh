#!/bin/bash
# Issue-cycle budget of conv_wino43b_kernel from PMC (three passes):  bash scratch/pmc_w43b.sh <tag> N H C [Cout]
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=$1; shift
O=gpurun_out/pmcw4b_$TAG; rm -rf $O; mkdir -p $O
i=0
for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA" "SQ_WAIT_ANY SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE SQ_WAVES SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_FLAT" "SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_IFETCH SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d $O/p$i -o p --output-format csv -- python3 scratch/w43b_one.py "$@" > $O/out$i.txt 2>&1
done
python3 - "conv_wino43b_kernel" "$O" <<'PY'
import csv, glob, collections, sys
ksub, O = sys.argv[1], sys.argv[2]
pmc = collections.defaultdict(list)
for f in glob.glob(O + '/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if ksub in r['Kernel_Name']:
            pmc[r['Counter_Name']].append(float(r['Counter_Value']))
out = open(O + '/summary.txt', 'w')
for k in sorted(pmc):
    line = '%-34s %16.0f  (%d dispatches)' % (k, sum(pmc[k]) / len(pmc[k]), len(pmc[k]))
    print(line); out.write(line + '\n')
PY
