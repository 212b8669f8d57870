#!/bin/bash
# in-step A/B of dispatch knobs (each chosen from kernels measured alone)
run() { env "$@" python bench.py --no-roofline --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for rep in 1 2; do
for e in X=1 KPX_WW_TARGET=128 KPX_WW_TARGET=512 KPX_WGRAD_TARGET=1024 KPX_WGRAD_TARGET=8192 KPX_SPLITK_MAXTILES=128 KPX_SPLITK_MAXTILES=512 KPX_WINO_STAGGER=0 KPX_NO_C16=1 KPX_NO_SMALLCOUT=1 KPX_NO_WTAPROWS=1 KPX_NO_WROWS=1 KPX_WW_COMIN=16 KPX_WSMALL_C64_MAX=16 KPX_SIDE_WGRAD=0; do
  echo "rep$rep $e: $(run $e)"
done; done
