#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/dpprof; mkdir -p $O
export WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29655
B="python3 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/nondp -o s --output-format csv -- $B > $O/nondp.json 2>/dev/null
export MASTER_PORT=29656
KPX_DP_SINGLE_GRAPH=1 KPX_DP_FORCE_EXCHANGE=1 KPX_DP_NO_COLLECTIVES=1 timeout 300 rocprofv3 --kernel-trace --stats -d $O/dpc -o s --output-format csv -- $B > $O/dpc.json 2>/dev/null
export MASTER_PORT=29657
KPX_DP_FORCE_EXCHANGE=1 timeout 300 rocprofv3 --kernel-trace --stats -d $O/dpb -o s --output-format csv -- $B > $O/dpb.json 2>/dev/null
for n in nondp dpc dpb; do f=$(find $O/$n -name "*kernel_stats.csv" | head -1); cp $f $O/${n}_kernel_stats.csv; python3 profiles/step_breakdown.py $f 13 > $O/${n}_breakdown.txt; tail -1 $O/$n.json | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$n', d['ms_per_step'])"; done
find $O -name "*kernel_trace.csv" | head; for n in nondp dpc dpb; do f=$(find $O/$n -name "*kernel_trace.csv" | head -1); python3 profiles/overlap.py $f 13 > $O/${n}_overlap.txt; done
rm -rf $O/nondp $O/dpc $O/dpb
