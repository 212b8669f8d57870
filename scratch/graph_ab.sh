#!/bin/bash
mkdir -p gpurun_out/ab
run() { name=$1; shift; env "$@" python bench.py --no-cpu-baseline --no-roofline --steps 30 > gpurun_out/ab/$name.json 2> gpurun_out/ab/$name.err; python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/ab/$name.json').read().strip().split('\n')[-1]); print('$name', d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'])
except Exception as e: print('$name ERR', e)
PY
}
run eager KPX_GRAPH=0
run graph KPX_GRAPH=1
run graph_forkafter KPX_GRAPH=1 KPX_FORK_BEFORE_DGRAD=after
run graph_q2 DEBUG_HIP_FORCE_GRAPH_QUEUES=2
run graph_q3 DEBUG_HIP_FORCE_GRAPH_QUEUES=3
run graph_q6 DEBUG_HIP_FORCE_GRAPH_QUEUES=6
run graph_noadv KPX_AUX_STREAM_ADV=0
run graph_nofwd KPX_AUX_STREAM_FWD=0
run eager2 KPX_GRAPH=0
