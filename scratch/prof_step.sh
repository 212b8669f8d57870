#!/bin/bash
# Per-family kernel-time breakdown of one bench configuration:  bash scratch/prof_step.sh <tag> [bench.py args...]
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=$1; shift
O=gpurun_out/prof_$TAG; rm -rf $O; mkdir -p $O
KPX_GRAPH=0 timeout 600 rocprofv3 --kernel-trace --stats -d $O/t -o s --output-format csv -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline "$@" > $O/bench.json 2>/dev/null
f=$(find $O/t -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats.csv
python3 profiles/step_breakdown.py $O/kernel_stats.csv 13 | tee $O/breakdown.txt
rm -rf $O/t
