#!/bin/bash
# L2 / memory counters of conv_wino43b_kernel:  bash scratch/pmc_w43b_mem.sh <tag> N H C [Cout]
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=$1; shift
O=gpurun_out/pmcw4bm_$TAG; rm -rf $O; mkdir -p $O
i=0
for c in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
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
for k in sorted(pmc):
    print('%-34s %16.0f  (%d dispatches)' % (k, sum(pmc[k]) / len(pmc[k]), len(pmc[k])))
PY
tail -3 $O/out1.txt
