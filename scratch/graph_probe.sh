#!/bin/bash
mkdir -p gpurun_out/probe
for m in ${MODES:-step_full}; do
  timeout 300 python scratch/graph_probe.py $m > gpurun_out/probe/$m.log 2>&1
  echo "$m rc=$? $(grep -a '^OK' gpurun_out/probe/$m.log | head -1)"
done
