#!/bin/bash
O=gpurun_out/dpdbg; mkdir -p $O
KPX_GRAPH=0 KPX_DP_FORCE_EXCHANGE=1 timeout 120 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $O/eager_b.out 2> $O/eager_b.err; echo "eager_b rc=$?" | tee -a $O/rc.txt
KPX_DP_FORCE_EXCHANGE=1 timeout 120 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29532 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $O/graph_b.out 2> $O/graph_b.err; echo "graph_b rc=$?" | tee -a $O/rc.txt
timeout 300 python3 -m pytest tests/test_model_gpu.py -x -q -s -k "data_parallel_train_steps" > $O/test1.log 2>&1; echo "test1 rc=$?" | tee -a $O/rc.txt
tail -5 $O/eager_b.err $O/graph_b.err $O/test1.log
