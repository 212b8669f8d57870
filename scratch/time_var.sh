#!/bin/bash
# forced tile variants of the bf16 3x3 kernel on one shape: bash scratch/time_var.sh N H Cin Cout "variants"
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
for v in $5; do
  O=gpurun_out/tv_$v; rm -rf $O; mkdir -p $O
  KPX_BF16S_VARIANT=$v timeout 300 rocprofv3 --kernel-trace --stats -d $O -o s --output-format csv -- python3 scratch/bf16s_one.py $1 $2 $3 $4 60 > $O/out.txt 2>&1
  grep bf16s_kernel $O/s_kernel_stats.csv | python3 -c "
import csv,sys
for r in csv.reader(sys.stdin): print('variant+1=$v', r[0][:45], 'avg us %.1f' % (float(r[3])/1e3), 'min %.1f' % (float(r[5])/1e3))"
done
