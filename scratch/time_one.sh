#!/bin/bash
# kernel durations of one bf16s shape for several library builds: bash scratch/time_one.sh "tags" N H Cin Cout
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
for tag in $1; do
  O=gpurun_out/t1_$tag; rm -rf $O; mkdir -p $O
  KPX_LIB=$PWD/scratch/exp/libkpx_$tag.so timeout 300 rocprofv3 --kernel-trace --stats -d $O -o s --output-format csv -- python3 scratch/bf16s_one.py $2 $3 $4 $5 60 > $O/out.txt 2>&1
  grep bf16s_kernel $O/s_kernel_stats.csv | python3 -c "
import csv,sys
for r in csv.reader(sys.stdin): print('$tag', '$2 $3 $4 $5', 'calls', r[1], 'avg us %.1f' % (float(r[3])/1e3), 'min %.1f' % (float(r[5])/1e3))"
done
