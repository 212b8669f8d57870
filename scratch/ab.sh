#!/bin/bash
# A/B of library builds inside ONE session (boxes differ by +-1.5 %):  bash scratch/ab.sh "a b c" [reps] [legs]   -> scratch/exp/libkpx_<tag>.so via KPX_LIB
REPS=${2:-3}; LEGS=${3:-roofline_conv roofline_wgrad}
for rep in $(seq $REPS); do for tag in $1; do
  KPX_LIB=$PWD/scratch/exp/libkpx_$tag.so timeout 100 python3 scratch/leg_only.py $LEGS 2>/dev/null | sed "s/^/$tag /"
  KPX_LIB=$PWD/scratch/exp/libkpx_$tag.so timeout 200 python3 bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$tag step', d['ms_per_step'])"
done; done
