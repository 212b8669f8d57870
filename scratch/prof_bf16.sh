#!/bin/bash
# kernel statistics of the bf16 train step (eager launches: one trace record per kernel): gpurun_out/bf16prof/
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
O=gpurun_out/bf16prof; rm -rf $O; mkdir -p $O
KPX_GRAPH=0 rocprofv3 --kernel-trace --stats -d $O/step -o s --output-format csv -- python3 bench.py --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $O/bench_profiled.json 2>/dev/null
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/bf16prof/step/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms per step (13 steps): %.2f' % (tot / 13 / 1e6))
for r in rows[:45]:
    print('%8.3f ms/step %6.1f calls/step avg %8.1f us  %s' % (float(r['TotalDurationNs']) / 13 / 1e6, float(r['Calls']) / 13, float(r['AverageNs']) / 1e3, r['Name'][:110]))
PY
