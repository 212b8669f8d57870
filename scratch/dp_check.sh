#!/bin/bash
# Data-parallel step on the one GPU of a build session: the 2-rank gloo tests, then the cost of the RCCL path at one rank
# (KPX_DP_FORCE_EXCHANGE=1 keeps both collectives in) against the plain single-GPU step, replayed and eager.
O=gpurun_out/dp; mkdir -p $O
[ -z "$SKIP_TESTS" ] && timeout 900 python3 -m pytest tests/test_model_gpu.py -x -q -k "data_parallel" --timeout=420 2>&1 | tail -5 | tee $O/tests.log
B="python3 bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-roofline"
timeout 200 $B 2>/dev/null | tail -1 > $O/single_graph.json
KPX_GRAPH=0 timeout 200 $B 2>/dev/null | tail -1 > $O/single_eager.json
KPX_DP_FORCE_EXCHANGE=1 timeout 200 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 4 --no-cpu-baseline --no-roofline 2>$O/nccl1_one.err | tail -1 > $O/nccl1_one.json
KPX_DP_GRAPH=segments KPX_DP_FORCE_EXCHANGE=1 timeout 200 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29521 bench.py --gpus 1 --steps 20 --warmup 4 --no-cpu-baseline --no-roofline 2>$O/nccl1_segments.err | tail -1 > $O/nccl1_segments.json
KPX_GRAPH=0 KPX_DP_GRAPH=segments KPX_DP_FORCE_EXCHANGE=1 timeout 200 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 1 --steps 20 --warmup 4 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $O/nccl1_eager_segments.json
KPX_GRAPH=0 KPX_DP_GRAPH=inline KPX_DP_FORCE_EXCHANGE=1 timeout 200 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 1 --steps 20 --warmup 4 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $O/nccl1_inline.json
KPX_DIST_BACKEND=gloo timeout 200 python3 bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $O/gloo2_segments.json
for f in single_graph single_eager nccl1_one nccl1_segments nccl1_segments_nocoll nccl1_eager_segments nccl1_inline gloo2_segments; do python3 -c "
import json
try:
    d=json.loads(open('$O/$f.json').read().strip().split(chr(10))[-1]); print('%-22s %8.3f ms  %8.1f pairs/s  host_work_min %s  mode %s' % ('$f', d['ms_per_step'], d['value'], d.get('host_work_ms_per_step_min_by_rank'), d.get('launch_mode_by_rank')))
except Exception as e: print('$f ERR', e)"; done | tee $O/summary.txt
grep -i "warn\|error\|fail" $O/nccl1_one.err | tail -5
