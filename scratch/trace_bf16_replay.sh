#!/bin/bash
# kernel trace of the REPLAYED bf16 step (graph launches): gpurun_out/bf16replay/
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
O=gpurun_out/bf16replay; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace -d $O/step -o s --output-format csv -- python3 bench.py --dtype ${1:-bf16} --steps 6 --warmup 3 --no-cpu-baseline --no-roofline > $O/bench.json 2>/dev/null
python3 profiles/overlap.py $O/step/s_kernel_trace.csv 9 | head -5
tail -1 $O/bench.json | cut -c1-200
