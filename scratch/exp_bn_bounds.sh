#!/bin/bash
# Timing-only bounds: what the step would take if a batch-norm phase cost nothing (KPX_EXP_SKIP bit mask: 1 fwd apply, 2 bwd reduce,
# 4 bwd apply, 8 finalizes).  Results are numerically meaningless.
O=gpurun_out/exp1; mkdir -p $O
for rep in 1 2; do
for m in 0 1 2 4 8 6 15; do
  KPX_EXP_SKIP=$m python3 bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('skip=$m rep=$rep', d['ms_per_step'], d['value'])" | tee -a $O/bounds.txt
done; done
