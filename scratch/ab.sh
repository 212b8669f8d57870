#!/bin/bash
# A/B of library builds inside ONE session (boxes differ by +-1.5 %):  bash scratch/ab.sh a b [reps]   -> scratch/exp/libkpx_<tag>.so via KPX_LIB
REPS=${3:-3}
for rep in $(seq $REPS); do for tag in $1 $2; do
  KPX_LIB=$PWD/scratch/exp/libkpx_$tag.so timeout 100 python3 scratch/wgrad_only.py 2>/dev/null | tail -1 | python3 -c "
import sys,ast
d=ast.literal_eval(sys.stdin.read()); print('$tag wgrad leg', d['achieved'], d['avg_launch_ms'])"
  KPX_LIB=$PWD/scratch/exp/libkpx_$tag.so timeout 200 python3 bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$tag step', d['ms_per_step'])"
done; done
