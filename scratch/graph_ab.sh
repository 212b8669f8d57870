#!/bin/bash
mkdir -p gpurun_out/ab
run() { name=$1; shift; env "$@" python bench.py --no-cpu-baseline --steps 20 > gpurun_out/ab/$name.json 2> gpurun_out/ab/$name.err; python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/ab/$name.json').read().strip().split('\n')[-1]); print('$name', d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'])
except Exception as e: print('$name ERR', e)
PY
}
run eager_forkbefore KPX_GRAPH=0
run graph_forkbefore KPX_GRAPH=1
run eager_forkafter KPX_GRAPH=0 KPX_FORK_BEFORE_DGRAD=0
run graph_forkafter KPX_GRAPH=1 KPX_FORK_BEFORE_DGRAD=0
run graph_forkbefore_sideonly KPX_GRAPH=1 KPX_AUX_STREAM=0 KPX_AUX_STREAM_FWD=0 KPX_AUX_STREAM_ADV=0
