#!/bin/bash
# instruction-cache counters of conv_wino43b_kernel:  bash scratch/pmc_w43b_icache.sh <tag> N H C [Cout]
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=$1; shift
O=gpurun_out/pmcw4b_ic_$TAG; rm -rf $O; mkdir -p $O
i=0
for c in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQC_ICACHE_BUSY_CYCLES SQC_ICACHE_INPUT_VALID_READYB GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY"; do
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
