#!/bin/bash
# A/B of library builds inside ONE session, bf16 configuration:  bash scratch/ab16.sh "a b" [reps]   -> scratch/exp/libkpx_<tag>.so via KPX_LIB
REPS=${2:-2}
for rep in $(seq $REPS); do for tag in $1; do
  KPX_LIB=$PWD/scratch/exp/libkpx_$tag.so timeout 200 python3 bench.py --roofline-only --dtype bf16 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$tag conv', d['roofline']['avg_launch_ms'], d['roofline']['frac'], 'wgrad', d['roofline_wgrad']['avg_launch_ms'])"
  KPX_LIB=$PWD/scratch/exp/libkpx_$tag.so timeout 300 python3 bench.py --dtype bf16 --steps 20 --warmup 4 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$tag step', d['ms_per_step'])"
done; done
