#!/bin/bash
O=gpurun_out/dpprobe; mkdir -p $O
B="bench.py --gpus 1 --steps 20 --warmup 4 --no-cpu-baseline --no-roofline"
L="timeout 200 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1"
run() { name=$1; shift; env "$@" $L --master-port $((29600 + RANDOM % 300)) $B 2>/dev/null | tail -1 > $O/$name.json; python3 -c "
import json
try:
    d=json.loads(open('$O/$name.json').read().strip().split(chr(10))[-1]); print('%-34s %8.3f ms  mode %s' % ('$name', d['ms_per_step'], d.get('launch_mode_by_rank')))
except Exception as e: print('$name ERR', e)" | tee -a $O/summary.txt; }
for rep in 1 2; do
run launcher_only_graph_$rep KPX_X=1
run launcher_only_eager_$rep KPX_GRAPH=0
run inline_eager_nocoll_$rep KPX_GRAPH=0 KPX_DP_PHASED=0 KPX_DP_FORCE_EXCHANGE=1 KPX_DP_NO_COLLECTIVES=1
run inline_eager_coll_$rep KPX_GRAPH=0 KPX_DP_PHASED=0 KPX_DP_FORCE_EXCHANGE=1
run single_c_nocoll_$rep KPX_DP_SINGLE_GRAPH=1 KPX_DP_FORCE_EXCHANGE=1 KPX_DP_NO_COLLECTIVES=1
run single_c_coll_$rep KPX_DP_SINGLE_GRAPH=1 KPX_DP_FORCE_EXCHANGE=1
run eager_c_nocoll_$rep KPX_GRAPH=0 KPX_DP_SCHEDULE=c KPX_DP_FORCE_EXCHANGE=1 KPX_DP_NO_COLLECTIVES=1
run segments_b_coll_$rep KPX_DP_FORCE_EXCHANGE=1
done
