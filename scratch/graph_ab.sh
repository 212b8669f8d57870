#!/bin/bash
mkdir -p gpurun_out/ab
run() { name=$1; shift; env "$@" python bench.py --no-cpu-baseline --no-roofline --steps 30 > gpurun_out/ab/$name.json 2> gpurun_out/ab/$name.err; python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/ab/$name.json').read().strip().split('\n')[-1]); print('$name', d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'])
except Exception as e: print('$name ERR', e)
PY
}
for i in 1 2; do
run eager_vggaux KPX_GRAPH=0 KPX_VGG_ON_AUX=1
run eager_daux KPX_GRAPH=0 KPX_VGG_ON_AUX=0
run graph_vggaux KPX_GRAPH=1 KPX_VGG_ON_AUX=1
run graph_daux KPX_GRAPH=1 KPX_VGG_ON_AUX=0
done
